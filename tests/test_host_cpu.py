"""CPU (-m "not gpu"): the C-ABI library loads and exports every declared symbol, host-side logic
(schedules, metrics, adjacency build, module surface) matches the reference fixtures, and the
product path refuses to run without a GPU instead of falling back."""
import os
import re

import numpy as np
import pytest
import torch

import gdmcf_amd
from gdmcf_amd import _lib
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_exports_every_declared_symbol():
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "gdmcf_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gdmcf_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/gdmcf_hip.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.gdmcf_version() == 1


def test_library_is_built_from_the_current_sources():
    """The in-tree libgdmcf_hip.so (the file that travels to the GPU box) was built from the sources as they are now: a
    header that stopped compiling once went unnoticed because every later run loaded the previous binary."""
    from gdmcf_amd import build as b
    stamp = os.path.join(b.CSRC, ".build_stamp")
    assert os.path.exists(b.LIB) and os.path.exists(stamp), "run python -m gdmcf_amd.build"
    assert open(stamp).read() == b._digest(), "sources changed since libgdmcf_hip.so was built: run python -m gdmcf_amd.build"


def test_schedule_build_matches_reference_tables():
    fx = H.load("schedules")
    kinds = {"linear": 0, "linear-var": 1, "cosine": 2, "binomial": 3}
    for c in fx["combos"]:
        key, sch, scale, mn, mx, T = str(c).split("|")
        tabs = _lib.schedule_tables(kinds[sch], float(scale), float(mn), float(mx), int(T), True)
        for name, row in zip(_lib.TABLE_NAMES, tabs):
            ref = fx[f"{key}.{name}"]
            if "log" in name or "sqrt" in name or "coef" in name:
                # libm sqrt/log are correctly rounded; torch's vectorised f64 sqrt/log (the reference)
                # are 1 ulp off in a few entries (e.g. 3/100 of sqrt_recip_alphas_cumprod): allow 2 ulp
                np.testing.assert_allclose(row, ref, rtol=5e-16, atol=0, err_msg=f"{c} {name}")
            else:
                np.testing.assert_array_equal(row, ref, err_msg=f"{c} {name}")


def test_schedule_errors_follow_reference_conventions():
    with pytest.raises(NotImplementedError):
        _lib.schedule_tables(9, 0.1, 0.001, 0.01, 5)
    with pytest.raises(AssertionError):  # betas out of range (reference gaussian_diffusion.py:83)
        _lib.schedule_tables(0, -1.0, 0.001, 0.01, 5, False)
    with pytest.raises(NotImplementedError):
        gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "nope", 0.1, 0.001, 0.01, 5, "cpu")


def test_product_normal_kl_matches_reference_vector():
    """gdmcf_amd.gaussian_diffusion.normal_kl (reference gaussian_diffusion.py:1165-1192, a utility: dead code there,
    SURVEY F10) against the KL vector the reference itself produced (schedules.npz `kl.*`), tensor and scalar forms."""
    from gdmcf_amd.gaussian_diffusion import normal_kl
    fx = H.load("schedules")
    m1, lv1, m2, lv2 = [torch.from_numpy(fx[f"kl.{k}"]) for k in ("m1", "lv1", "m2", "lv2")]
    np.testing.assert_array_equal(normal_kl(m1, lv1, m2, lv2).numpy(), fx["kl.out"])
    # scalar log-variances are promoted to the tensor's dtype (reference :1181-1184)
    a = normal_kl(m1, 0.25, m2, -0.5).numpy()
    b = normal_kl(m1, torch.full_like(m1, 0.25), m2, torch.full_like(m1, -0.5)).numpy()
    np.testing.assert_array_equal(a, b)
    assert np.abs(normal_kl(m1, lv1, m1, lv1).numpy()).max() < 1e-6  # KL(p || p) = 0 up to f32 rounding
    with pytest.raises(AssertionError):
        normal_kl(0.0, 0.0, 1.0, 0.0)


def test_diffusion_object_surface_and_weights():
    fx = H.load("schedules")
    d = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 5, "cpu")
    assert list(d.parameters()) == []  # main.py:263 sums diffusion.parameters()
    assert d.Lt_history.shape == (5, 10) and d.Lt_history.dtype == torch.float64
    assert d.Lt_count.dtype == torch.int64
    np.testing.assert_array_equal(d._weights["x0"].numpy(), fx["c0.snr_weight_x0"])
    np.testing.assert_array_equal(d.posterior_mean_coef1.numpy(), fx["c0.posterior_mean_coef1"])
    d.gcn, d.indexIn = 0, None  # attributes main.py pokes (:222, :241)
    # main.py:192-193 builds the Discrete class with these extra keywords and calls .to(device)
    dd = gdmcf_amd.GaussianDiffusionDiscrete(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 5, "cpu",
                                             discrete=0.99, CatOneHot=False, epps=0.9995, args=None).to("cpu")
    np.testing.assert_array_equal(dd.betas.numpy(), d.betas.numpy())
    # CatOneHot: built for the Discrete class (with a DNNOneHot denoiser), not for the base class's own one-hot branch
    dd1 = gdmcf_amd.GaussianDiffusionDiscrete(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 5, "cpu",
                                              CatOneHot=True)
    assert dd1.CatOneHot and dd1.indexIn is False and dd1.discrete == 0.99
    with pytest.raises(NotImplementedError):
        gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 5, "cpu",
                                    CatOneHot=True)


def test_onehot_backbone_surface_matches_reference_init_and_names():
    """DNNOneHot: same state_dict names / shapes and the same initialisation draws as the reference (the fixture's
    weights were drawn by the reference class under the same seed), incl. the growth of the caller's out_dims[0]."""
    fx = H.load("onehot_train_deep_x0")
    meta = H.onehot_train_meta(fx)
    torch.manual_seed(33)
    I, dims = meta["I"], meta["dims"]
    out_dims = dims[::-1] + [I]
    m = gdmcf_amd.DNNOneHot([I] + dims, out_dims, 10)
    assert out_dims[0] == 2 * dims[-1]
    sd = m.state_dict()
    ref = H.state_dict_from(fx)
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k


def test_dnn_surface_matches_reference_init_and_names():
    fx = H.load("train_tiny_x0")
    meta = H.train_meta(fx)
    torch.manual_seed(1)  # gen_golden's seed for this case: same draw order -> same weights
    m = gdmcf_amd.DNN([meta["I"]] + meta["dims"], meta["dims"][::-1] + [meta["I"]], 10, time_type="cat", norm=False)
    sd = m.state_dict()
    assert list(sd.keys()) == [k[3:] for k in fx if k.startswith("sd.")]
    for k, v in sd.items():
        np.testing.assert_array_equal(v.numpy(), fx["sd." + k], err_msg=k)
    with pytest.raises(AssertionError):
        gdmcf_amd.DNN([64, 16], [8, 64], 10)
    with pytest.raises(ValueError):
        gdmcf_amd.DNN([64, 16], [16, 64], 10, time_type="add")
    np.testing.assert_array_equal(gdmcf_amd.timestep_embedding(torch.from_numpy(H.load("schedules")["temb.ts"]), 10).numpy(),
                                  H.load("schedules")["temb.10"])


def test_product_path_has_no_cpu_fallback():
    m = gdmcf_amd.DNN([64, 16], [16, 64], 10)
    d = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 5, "cpu")
    x = torch.zeros(4, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, torch.zeros(4, dtype=torch.long))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        d.training_losses(m, x, True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        d.p_sample(m, x, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gdmcf_amd.masked_topk(torch.zeros(2, 8), 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):  # the device noise draw (gdmcf_randn_f32) does not become torch.randn
        gdmcf_amd._lib.philox_randn((2, 8), "cpu", 1, 1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gdmcf_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+\S*oracle", src, flags=re.M), f"{fn} imports the oracle"


def test_metrics_match_reference():
    from gdmcf_amd import computeTopNAccuracy
    assert computeTopNAccuracy([[1, 2], [3], []], [[1, 5, 2], [4, 3, 9], [0, 1, 2]], [1, 3]) == \
        ([0.3333, 0.3333], [0.1667, 0.6667], [0.3333, 0.5169], [0.3333, 0.5])
    for case in H.SAMPLE_CASES:
        fx = H.load("sample_" + case)
        gt = [fx["gt_flat"][a:b].tolist() for a, b in zip(fx["gt_ptr"][:-1], fx["gt_ptr"][1:])]
        res = computeTopNAccuracy(gt, fx["topk_idx"].tolist(), fx["topN"].tolist())
        np.testing.assert_array_equal(np.array(res, dtype=np.float64), fx["metrics"])


@pytest.mark.parametrize("case", ["small", "mid"])
def test_lightgcn_adjacency_build_matches_reference(case):
    from gdmcf_amd.lightgcn import normalized_bipartite_csr
    fx = H.load("lightgcn_" + case)
    U, It, d, L = [int(v) for v in str(fx["meta"][0]).split("|")]
    indptr, indices, vals = normalized_bipartite_csr(fx["users"], fx["items"], U, It)
    rows = np.repeat(np.arange(U + It), np.diff(indptr))
    np.testing.assert_array_equal(rows, fx["A_row"])
    np.testing.assert_array_equal(indices, fx["A_col"])
    np.testing.assert_allclose(vals, fx["A_val"], rtol=2e-7, atol=0)
    assert indptr.dtype == np.int64 and indices.dtype == np.int32 and vals.dtype == np.float32


def test_data_load_matches_reference(tmp_path):
    """reference data_utils.data_load (npy [nnz,2] pairs -> CSR float64, duplicates summed) on the fixture lists."""
    from gdmcf_amd import data_utils
    fx = H.load("data_load")
    paths = []
    for n in ("train", "valid", "test"):
        paths.append(str(tmp_path / f"{n}_list.npy"))
        np.save(paths[-1], fx[f"{n}_list"])
    tr, va, te, nu, ni = data_utils.data_load(*paths)
    assert (nu, ni) == (int(fx["n_user"]), int(fx["n_item"]))
    for got, key in ((tr, "train"), (va, "valid"), (te, "test")):
        assert got.dtype == np.float64 and got.format == "csr"
        np.testing.assert_array_equal(got.toarray(), fx[key])
    ds = data_utils.DataDiffusion(torch.from_numpy(tr.toarray()))
    row, idx = ds[3]
    assert idx == 3 and len(ds) == nu and torch.equal(row, torch.from_numpy(tr.toarray())[3])


def test_bf16_shadow_registry_and_gemm_precision_are_host_side_state():
    """include/gdmcf_hip.h: gdmcf_gemm_precision is per calling thread and returns the previous mode; the bf16 shadow
    registry is a host-side map keyed by the float32 base pointer (no GPU needed to exercise its contract)."""
    import ctypes
    import threading
    from gdmcf_amd import _lib
    lib = _lib.load()
    assert lib.gdmcf_gemm_precision(1) == 0 and lib.gdmcf_gemm_precision(-1) == 1  # -1 only queries
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.gdmcf_gemm_precision(-1)))
    t.start(), t.join()
    assert seen == [0]  # another thread still has the default
    assert lib.gdmcf_gemm_precision(0) == 1
    f32 = (ctypes.c_float * 16)()
    b16 = (ctypes.c_uint16 * (64 * 64 + 8))()
    base = ctypes.addressof(b16)
    aligned = (base + 15) & ~15
    pf = ctypes.addressof(f32)
    assert _lib.shadow_info(pf) is None and lib.gdmcf_bf16_shadow_get(pf) is None
    _lib.check(lib.gdmcf_bf16_shadow_set(pf, aligned, 4, 4, 64))
    assert _lib.shadow_info(pf) == (aligned, 4, 4, 64) and lib.gdmcf_bf16_shadow_get(pf) == aligned
    with pytest.raises(AssertionError):  # row stride must be a multiple of 64 and >= cols
        _lib.check(lib.gdmcf_bf16_shadow_set(pf, aligned, 4, 4, 8))
    with pytest.raises(ValueError):  # 16-byte alignment
        _lib.check(lib.gdmcf_bf16_shadow_set(pf, aligned + 2, 4, 4, 64))
    with pytest.raises(ValueError):  # syncing something that was never registered
        _lib.check(lib.gdmcf_bf16_shadow_sync(pf + 4, 4, None))
    _lib.check(lib.gdmcf_bf16_shadow_clear(pf))
    assert _lib.shadow_info(pf) is None


@pytest.mark.parametrize("d", [64, 20])
def test_spmm_plan_covers_every_nonzero_once(d):
    """Host-built execution plan of gdmcf_spmm_csr_f32 (gdmcf_amd/lightgcn.py:spmm_plan): emulate in numpy exactly
    what the kernels do with it -- whole short rows, pieces of long rows, slot-ordered combination of cut rows --
    and compare with the CSR product; also the invariants the kernels rely on.  Includes empty rows, a hub row cut
    into many pieces, a row of exactly `chunk` nonzeros, and a width without the short-row path (d = 20)."""
    import scipy.sparse as sp
    from gdmcf_amd.lightgcn import spmm_plan
    rng = np.random.default_rng(0)
    n, m, chunk, short = 300, 120, 16, 5
    deg = rng.integers(0, 12, n)
    deg[7], deg[8], deg[9], deg[200] = 0, 119, chunk, short  # empty, hub, exactly one chunk, short-limit row
    rows = np.repeat(np.arange(n), deg)
    cols = np.concatenate([rng.choice(m, k, replace=False) for k in deg]) if deg.sum() else np.zeros(0, int)
    A = sp.csr_matrix((rng.standard_normal(len(rows)), (rows, cols)), shape=(n, m))
    A.sort_indices()
    X = rng.standard_normal((m, d))
    pl = spmm_plan(A.indptr, chunk=chunk, short=short, d=d)
    nv = len(pl["vrow"])
    assert len(pl["vbeg"]) == len(pl["vend"]) == len(pl["vslot"]) == nv
    # every nonzero belongs to exactly one virtual row, pieces never exceed the chunk, short rows come first, whole
    cover = np.zeros(A.nnz, int)
    for b, e in zip(pl["vbeg"], pl["vend"]):
        cover[b:e] += 1
    assert (cover == 1).all()
    ns = pl["n_short"]
    assert (d == 20 and ns == 0) or (d == 64 and ns == int((np.diff(A.indptr) <= short).sum()))
    assert (pl["vslot"][:ns] == -1).all() and ((pl["vend"] - pl["vbeg"])[ns:] <= chunk).all()
    assert (pl["vbeg"][:ns] == A.indptr[pl["vrow"][:ns]]).all() and (pl["vend"][:ns] == A.indptr[pl["vrow"][:ns] + 1]).all()
    # emulate: direct rows write Y, cut rows write partial slots that are added in slot order
    Y = np.zeros((n, d))
    partial = np.zeros((max(pl["n_slots"], 1), d))
    for v in range(nv):
        acc = np.zeros(d)
        for j in range(pl["vbeg"][v], pl["vend"][v]):
            acc += A.data[j] * X[A.indices[j]]
        if pl["vslot"][v] < 0:
            Y[pl["vrow"][v]] = acc
        else:
            partial[pl["vslot"][v]] = acc
    for i, r in enumerate(pl["lrow"]):
        Y[r] = partial[pl["lptr"][i]:pl["lptr"][i + 1]].sum(0)
    np.testing.assert_allclose(Y, A @ X, rtol=1e-12, atol=1e-12)
    assert 8 in pl["lrow"] and 9 not in pl["lrow"] and len(pl["lptr"]) == len(pl["lrow"]) + 1  # only cut rows combine


@pytest.mark.parametrize("d,n_waves,split_at", [(64, 32, None), (64, 64, None), (16, 32, None), (256, 32, None), (8, 32, None),
                                                (64, 32, 100), (16, 64, 100)])
def test_spmm_bundle_plan_is_a_partition_of_the_work(d, n_waves, split_at):
    """Static schedule of gdmcf_spmm_bundled_f32 (gdmcf_amd/lightgcn.py:spmm_bundle_plan), emulated in numpy wave by wave
    exactly as csrc/spmm_bundle.hip walks it: every nonzero is gathered exactly once, every row is written exactly once
    (whole rows by their wave, cut rows from their partial slots in slot order), bundles hold rows of the same class
    sorted by length, the waves of a class own contiguous runs, empty rows still get written."""
    import scipy.sparse as sp
    from gdmcf_amd.lightgcn import spmm_bundle_plan
    rng = np.random.default_rng(d + n_waves)
    n, m = 257, 190
    deg = np.minimum(rng.zipf(1.6, n), m)
    deg[5], deg[6], deg[7] = 0, m, 150  # an empty row, a full row, a long one
    rows = np.repeat(np.arange(n), deg)
    cols = np.concatenate([np.sort(rng.choice(m, k, replace=False)) for k in deg])
    A = sp.csr_matrix((rng.standard_normal(len(rows)), (rows, cols)), shape=(n, m))
    A.sort_indices()
    X = rng.standard_normal((m, d))
    s_max, piece = 12, 40
    # (split_at: the round-4 experiment that phases the rows below / from a row index apart -- still a partition)
    pl = spmm_bundle_plan(A.indptr, A.indices, d=d, n_waves=n_waves, s_max=s_max, piece=piece, split_at=split_at)
    G = pl["G"]
    assert G == 64 // (d // 4) and pl["n_waves"] == n_waves
    wd = pl["wdesc"].reshape(n_waves, 4)
    wpc = n_waves // 8
    cover = np.zeros(A.nnz, int)
    written = np.zeros(n, int)
    Y = np.full((n, d), np.nan)
    partial = np.full((max(pl["n_slots"], 1), d), np.nan)
    seen_p, seen_b = np.zeros(pl["n_pieces"], int), np.zeros(pl["n_bundles"], int)
    for w in range(n_waves):
        l0, l1, s0, s1 = wd[w]
        assert l0 <= l1 and s0 <= s1
        if w % wpc:  # contiguous runs inside a class
            assert l0 == wd[w - 1][1] and s0 == wd[w - 1][3]
        for p in range(l0, l1):
            seen_p[p] += 1
            b, ln, r = pl["lbeg"][p], pl["llen"][p], pl["lrow"][p]
            assert 0 < ln <= piece and A.indptr[r] <= b and b + ln <= A.indptr[r + 1]
            cover[b:b + ln] += 1
            acc = (A.data[b:b + ln, None] * X[A.indices[b:b + ln]]).sum(0)
            if pl["lslot"][p] >= 0:
                partial[pl["lslot"][p]] = acc
            else:
                Y[r] = acc
                written[r] += 1
        for bd in range(s0, s1):
            seen_b[bd] += 1
            lens = pl["slen"][bd * G:(bd + 1) * G]
            assert pl["smax"][bd] & 0x3FFFFFFF == lens.max()  # bit 30 = streaming-load hint
            for e in range(bd * G, (bd + 1) * G):
                r = pl["srow"][e]
                if r < 0:
                    assert pl["slen"][e] == 0
                    continue
                b, ln = pl["sbeg"][e], pl["slen"][e]
                assert b == A.indptr[r] and ln == A.indptr[r + 1] - A.indptr[r] and ln <= s_max
                cover[b:b + ln] += 1
                Y[r] = (A.data[b:b + ln, None] * X[A.indices[b:b + ln]]).sum(0)
                written[r] += 1
    assert (seen_p == 1).all() and (seen_b == 1).all() and (cover == 1).all()
    for i, r in enumerate(pl["crow"]):
        sl = partial[pl["cptr"][i]:pl["cptr"][i + 1]]
        assert len(sl) >= 2 and not np.isnan(sl).any()
        Y[r] = sl.sum(0)
        written[r] += 1
    assert (written == 1).all()
    np.testing.assert_allclose(Y, A @ X, rtol=1e-12, atol=1e-12)
    assert 6 in pl["crow"] and 7 in pl["crow"] and 5 not in pl["crow"] and (Y[5] == 0).all()
    # a piece never crosses a column-range boundary: the pieces of one class gather from one slice of the table
    lo = np.array([A.indices[b] for b in pl["lbeg"]]); hi = np.array([A.indices[b + n - 1] for b, n in zip(pl["lbeg"], pl["llen"])])
    pcls = np.searchsorted(wd[::wpc, 0], np.arange(pl["n_pieces"]), side="right") - 1  # class of a piece from the wave table
    for c in range(7):
        if (pcls == c).any() and (pcls > c).any():
            assert hi[pcls == c].max() <= lo[pcls > c].min()
    # classes hold equal shares of the cost (mean-column order), waves of a class equal shares of the class
    from gdmcf_amd.lightgcn import SPMM_COST_PIECE, SPMM_COST_ROW
    lc = np.concatenate([[0], np.cumsum(pl["llen"] + SPMM_COST_PIECE)])
    sc = np.concatenate([[0], np.cumsum(((pl["smax"] & 0x3FFFFFFF) + SPMM_COST_ROW) * G)])
    cost = (lc[wd[:, 1]] - lc[wd[:, 0]]) + (sc[wd[:, 3]] - sc[wd[:, 2]])
    per_class = cost.reshape(8, wpc).sum(1)
    assert per_class.max() <= 1.5 * per_class.mean() + (piece + 32)


@pytest.mark.parametrize("d,n_waves", [(64, 32), (64, 64), (16, 32), (256, 32), (8, 32)])
def test_spmm_stream_pack_reproduces_the_product(d, n_waves):
    """gdmcf_spmm_stream_f32's input (gdmcf_amd/lightgcn.py:spmm_stream_pack), walked in numpy exactly as the kernel does
    (csrc/spmm_bundle.hip: spmm_stream_kernel): wave by wave, batch by batch, UN steps at a time, lane group g taking entry
    step*G + g; pieces are summed over the lane groups and go to their slot or row, bundles write one row per lane group.
    The result must be A @ X, every real nonzero must appear exactly once with its value, padding must have weight 0."""
    import scipy.sparse as sp
    from gdmcf_amd.lightgcn import spmm_bundle_plan, spmm_stream_pack
    rng = np.random.default_rng(3 * d + n_waves)
    n, m = 257, 190
    deg = np.minimum(rng.zipf(1.6, n), m)
    deg[5], deg[6], deg[7] = 0, m, 150
    rows = np.repeat(np.arange(n), deg)
    cols = np.concatenate([np.sort(rng.choice(m, k, replace=False)) for k in deg])
    A = sp.csr_matrix((rng.standard_normal(len(rows)).astype(np.float32), (rows, cols)), shape=(n, m))
    A.sort_indices()
    X = rng.standard_normal((m, d))
    plan = spmm_bundle_plan(A.indptr, A.indices, d=d, n_waves=n_waves, s_max=12, piece=40, n_cols=m)
    sp_ = spmm_stream_pack(plan, A.indptr, A.indices, A.data, d=d)
    G, UN, DW = sp_["G"], sp_["UN"], sp_["DW"]
    LPR = 64 // G
    assert UN == min(4, LPR) and DW == 1 + max(G, 2) and sp_["n_entries"] % 64 == 0
    cw = sp_["cw"].reshape(-1, 2)
    cc, ww = cw[:, 0], cw[:, 1].copy().view(np.float32)
    ud = sp_["ud"].reshape(-1, DW)
    wd = sp_["wdesc"].reshape(n_waves, 4)
    Y = np.full((n, d), np.nan)
    written = np.zeros(n, int)
    partial = np.full((max(sp_["n_slots"], 1), d), np.nan)
    units_seen = np.zeros(sp_["n_units"], int)
    real = []  # (col, val) of every entry with a weight
    for w in range(n_waves):
        sb, nb, u0, u1 = wd[w]
        pos = sb * 64  # entry index of the current step's group-0 lane
        for u in range(u0, u1):
            units_seen[u] += 1
            hdr = int(ud[u, 0])
            ngroups = hdr & 0x7FFFFFFF
            acc = np.zeros((G, d))
            for _ in range(ngroups * UN):
                assert pos + G <= (sb + nb) * 64
                for g in range(G):
                    c, wt = cc[pos + g], ww[pos + g]
                    assert 0 <= c < m
                    acc[g] += float(wt) * X[c]
                    if wt != 0:
                        real.append((u, g, c, wt))
                pos += G
            if hdr < 0:
                row, slot = ud[u, 1], ud[u, 2]
                if slot >= 0:
                    partial[slot] = acc.sum(0)
                else:
                    Y[row] = acc.sum(0)
                    written[row] += 1
            else:
                for g in range(G):
                    a = int(ud[u, 1 + g])
                    if a < 0:
                        assert not acc[g].any()
                        continue
                    r = a & 0x3FFFFFFF
                    Y[r] = 0.0 if a & 0x40000000 else acc[g]
                    assert (a & 0x40000000 != 0) == (deg[r] == 0)
                    written[r] += 1
        assert pos <= (sb + nb) * 64 and ((sb + nb) * 64 - pos < 64 or u0 == u1)  # the run is exactly the wave's units
        kinds = [int(ud[u, 0]) < 0 for u in range(u0, u1)]
        assert kinds == sorted(kinds, reverse=True)  # a wave runs its pieces first, then its bundles
    assert (units_seen == 1).all()
    for i, r in enumerate(sp_["crow"]):
        sl = partial[sp_["cptr"][i]:sp_["cptr"][i + 1]]
        assert len(sl) >= 2 and not np.isnan(sl).any()
        Y[r] = sl.sum(0)
        written[r] += 1
    assert (written == 1).all()
    np.testing.assert_allclose(Y, A @ X, rtol=1e-6, atol=1e-6)
    assert len(real) == (A.data != 0).sum()


@pytest.mark.parametrize("n,n_nonempty", [(2000, 10), (200, 5)])
def test_spmm_stream_pack_gives_every_wave_with_units_a_batch(n, n_nonempty):
    """Graphs of mostly isolated nodes: a wave may own nothing but bundles of EMPTY rows (no entries of its own).  The
    stream kernel pre-loads batch 0 and batch min(1, nb-1) of the run of every wave that has units (csrc/spmm_bundle.hip:
    spmm_stream_kernel), so the packer must give such a wave one (weight-0) batch inside the cw array -- with nb == 0 those
    loads fell before / behind the run (round-2 advisor finding)."""
    import scipy.sparse as sp
    from gdmcf_amd.lightgcn import spmm_bundle_plan, spmm_stream_pack
    rng = np.random.default_rng(n)
    rows = np.repeat(rng.choice(n, n_nonempty, replace=False), 3)
    cols = rng.integers(0, n, len(rows))
    A = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n, n))
    A.sum_duplicates()
    A.sort_indices()
    plan = spmm_bundle_plan(A.indptr, A.indices, d=64, n_cols=n)
    sp_ = spmm_stream_pack(plan, A.indptr, A.indices, A.data, d=64)
    wd = sp_["wdesc"].reshape(-1, 4)
    n_batches = sp_["n_entries"] // 64
    assert len(sp_["cw"]) == 2 * sp_["n_entries"]
    has_units = wd[:, 3] > wd[:, 2]
    assert has_units.sum() > n_nonempty  # the case is real: waves that own only empty rows exist
    assert (wd[has_units, 1] >= 1).all()
    assert (wd[has_units, 0] + wd[has_units, 1] <= n_batches).all()
    cw = sp_["cw"].reshape(-1, 2)
    assert (cw[:, 0] >= 0).all() and (cw[:, 0] < n).all()


def test_bench_gpus_n_starts_n_ranks_or_fails_loudly():
    """`python bench.py --gpus N` (the shape of the command the driver runs) must produce an N-rank line or no line: the
    parent starts N fresh rank processes itself (dry run here: the ranks meet in a gloo group on the CPU, nothing is timed),
    and a rank count that does not match --gpus is an error, never a silent n_gpus = 1."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, GDMCF_BENCH_DRY_RUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_in_group"] == 2 and line["dry_run"] is True
    # the driver's largest case: eight ranks of one node (dry run: they only have to find each other and report as eight)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["ranks_in_group"] == 8 and line["dry_run"] is True
    # launched as ONE rank but asked for two GPUs: refuse (before anything touches a GPU)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr and "{" not in r.stdout
    # no dry run, no GPUs here: the parent refuses to start fewer ranks than asked for
    if not torch.cuda.is_available():
        env2 = {k: v for k, v in env.items() if k != "GDMCF_BENCH_DRY_RUN"}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and "{" not in r.stdout


def test_header_is_plain_c(tmp_path):
    """include/gdmcf_hip.h is the C ABI: it must compile as C99 with nothing but the standard headers (no C++, no HIP,
    no torch types in any signature)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = tmp_path / "use_header.c"
    src.write_text('#include "%s"\nint probe(void) { return GDMCF_OK + GDMCF_GEMM_BF16; }\n' % os.path.join(ROOT, "include", "gdmcf_hip.h"))
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-c", str(src), "-o", str(tmp_path / "o.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gdmcf_hip.h")).read(), flags=re.S)  # code only
    assert "at::" not in hdr and "Tensor" not in hdr and "hipStream_t" not in hdr and "std::" not in hdr


def test_c_program_links_and_uses_the_abi(tmp_path):
    """examples/c_abi_smoke.c: a plain C99 program linked against libgdmcf_hip.so builds the reference's schedule
    tables through the C ABI (pinned values of SURVEY 8a) and sees the error convention -- no Python in between."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    _lib.load()  # builds the library if it is missing
    csrc = os.path.join(ROOT, "gdmcf_amd", "csrc")
    exe = str(tmp_path / "c_abi_smoke")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", os.path.join(ROOT, "examples", "c_abi_smoke.c"),
                        "-I" + os.path.join(ROOT, "include"), "-L" + csrc, "-lgdmcf_hip", "-lm", "-Wl,-rpath," + csrc,
                        "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "betas[1] = 2.2500225002275442e-05" in r.stdout and "rc -3" in r.stdout


def test_onehot_embedding_backbone_surface_matches_reference_init_and_names():
    """DNNOneHotEmbedding: the reference's state_dict names / shapes and initialisation draws (layers, then xavier-uniform
    item and user tables), out_dims[0] grown like DNNOneHot's."""
    fx = H.load("onehot_emb_ragged_eps_wd")
    meta = H.onehot_emb_meta(fx)
    torch.manual_seed(52)
    I, dims = meta["I"], meta["dims"]
    out_dims = dims[::-1] + [I]
    m = gdmcf_amd.DNNOneHotEmbedding([I] + dims, out_dims, 10, item_num=I, user_num=meta["U"])
    assert out_dims[0] == 2 * dims[-1]
    sd, ref = m.state_dict(), H.state_dict_from(fx)
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k
    assert m.embedding_item.weight.shape == (I, 3 * dims[-1]) and m.embedding_user.weight.shape == (meta["U"], dims[-1])


def _fake_kernel(body):
    return "_Z4fakev:                                ; @fake\n" + body + "\n\ts_endpgm\n"


def test_build_lint_finds_registers_touched_before_their_counted_wait():
    """gdmcf_amd/build.py: the kernels with hand-counted `s_waitcnt vmcnt(N)` behind inline-asm loads are disassembled at build
    time; a consumer placed before its wait, a compiler copy of a register whose load is still in flight, or an address that
    uses one are build errors (the failure mode behind DESIGN 4.1b's corrupted accumulator lanes)."""
    from gdmcf_amd.build import lint_vmcnt, lint_ring_registers
    load = lambda d, a: f"\t;;#ASMSTART\n\tbuffer_load_dwordx4 v[{d}:{d + 3}], v{a}, s[0:3], s4 offen\n\t;;#ASMEND\n"
    ok = _fake_kernel(load(10, 1) + load(14, 1) + "\t;;#ASMSTART\n\ts_waitcnt vmcnt(1)\n\t;;#ASMEND\n"
                      "\tv_mfma_f32_16x16x4_f32 v[20:23], v10, v2, v[20:23]\n" + load(10, 1) +
                      "\t;;#ASMSTART\n\ts_waitcnt vmcnt(1)\n\t;;#ASMEND\n\tv_mfma_f32_16x16x4_f32 v[20:23], v14, v2, v[20:23]\n"
                      "\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v30, v10")
    assert lint_vmcnt(ok) == []
    early = ok.replace("vmcnt(1)\n\t;;#ASMEND\n\tv_mfma_f32_16x16x4_f32 v[20:23], v10", "vmcnt(2)\n\t;;#ASMEND\n\tv_mfma_f32_16x16x4_f32 v[20:23], v10")
    assert any("v10 used" in b for b in lint_vmcnt(early))
    copy = ok.replace("\ts_waitcnt vmcnt(0)\n", "")
    assert any("copied" in b for b in lint_vmcnt(copy))
    addr = _fake_kernel(load(10, 1) + load(14, 11))
    assert any("address" in b for b in lint_vmcnt(addr))
    # a loop: the load issued at the bottom of the body is consumed at its top -- behind the wait: clean; without it: flagged
    loop = _fake_kernel(load(10, 1) + ".LBB0_1:\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n"
                        "\tv_mfma_f32_16x16x4_f32 v[20:23], v10, v2, v[20:23]\n" + load(10, 1) + "\ts_cbranch_scc1 .LBB0_1\n\ts_waitcnt vmcnt(0)")
    assert lint_vmcnt(loop) == []
    assert lint_vmcnt(loop.replace(".LBB0_1:\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)", ".LBB0_1:\n\t;;#ASMSTART\n\ts_nop 0"))
    assert lint_vmcnt("nothing here")  # no kernel with asm loads: the lint says so instead of passing silently
    assert lint_ring_registers("nothing here")


def test_build_lint_finds_store_data_rewritten_too_early():
    """gdmcf_amd/build.py:lint_store_data -- a 16-byte vector-memory store whose data register is rewritten by the next
    instruction (what hipcc emitted for `buffer_store_dwordx4 .., sN offen`: it pads this hazard only for immediate scalar offsets)."""
    from gdmcf_amd.build import lint_store_data
    bad = _fake_kernel("\tbuffer_store_dwordx4 v[146:149], v0, s[36:39], s10 offen\n\tv_mov_b32_e32 v146, v114\n\tv_mov_b32_e32 v147, v110")
    assert len(lint_store_data(bad)) == 1 and "v146" in lint_store_data(bad)[0]
    ok = _fake_kernel("\tbuffer_store_dwordx4 v[148:151], v1, s[36:39], 0 offen\n\tv_mov_b32_e32 v146, v126\n\tv_mov_b32_e32 v147, v122\n"
                      "\tv_mov_b32_e32 v148, v118")
    assert lint_store_data(ok) == []
    nop = _fake_kernel("\tglobal_store_dwordx4 v[0:1], v[4:7], off\n\ts_nop 1\n\tv_mov_b32_e32 v4, v9")
    assert lint_store_data(nop) == []
    assert len(lint_store_data(_fake_kernel("\tglobal_store_dwordx4 v[0:1], v[4:7], off\n\ts_nop 0\n\tv_mov_b32_e32 v5, v9"))) == 1


def test_fuse_into_backward_seats_weight_rows_on_128_byte_lines(tmp_path, monkeypatch):
    """FusedAdamW.fuse_into_backward (optim.py; reference main.py:258 constructs the optimiser, models/DNN.py:71-72 the weights):
    a fused weight keeps its Parameter object, shape and values, its rows move onto 128-byte lines; state_dict / torch.save /
    load_state_dict go through the view; the moments follow the weight; unfuse() gives contiguous tensors back."""
    torch.manual_seed(0)
    m = gdmcf_amd.DNN([3000, 700], [700, 3000], 10, time_type="cat", norm=False)
    ref = {k: v.clone() for k, v in m.state_dict().items()}
    ids = [id(w) for (w, _, _) in m.layer_list()]
    opt = gdmcf_amd.FusedAdamW(m.parameters(), lr=1e-3)
    assert opt.fuse_into_backward(m, min_numel=1 << 12) is opt and m.engine.fused_opt is opt
    ws = [w for (w, _, _) in m.layer_list()]
    assert [id(w) for w in ws] == ids and all(isinstance(w, torch.nn.Parameter) and w.requires_grad for w in ws)
    assert [tuple(w.shape) for w in ws] == [(700, 3010), (3000, 700)]
    assert [w.stride() for w in ws] == [(3040, 1), (704, 1)]
    assert all(torch.equal(m.state_dict()[k], ref[k]) for k in ref)
    path = tmp_path / "sd.pt"
    torch.save(m.state_dict(), path)
    back = torch.load(path)
    assert all(torch.equal(back[k], ref[k]) for k in ref)
    m.load_state_dict({k: 2 * v for k, v in ref.items()})
    assert [w.stride() for w in ws] == [(3040, 1), (704, 1)]
    assert all(torch.equal(m.state_dict()[k], 2 * ref[k]) for k in ref)
    # what lies between the rows is zero and stays zero
    pad = torch.as_strided(ws[1].data, (3000, 4), (704, 1), 700)
    assert float(pad.abs().max()) == 0.0
    # moments created for a seated weight share its leading dimension; moments that do not are re-seated with their values
    fs = opt.fused_state(ws[1])
    assert fs["exp_avg"].stride() == (704, 1) and fs["exp_avg_sq"].stride() == (704, 1) and fs["step"] == 1
    opt.state[ws[0]].update(step=3, exp_avg=torch.full((700, 3010), 0.5), exp_avg_sq=torch.full((700, 3010), 0.25))
    fs0 = opt.fused_state(ws[0])
    assert fs0["exp_avg"].stride() == (3040, 1) and float(fs0["exp_avg"].min()) == 0.5 and float(fs0["exp_avg_sq"].max()) == 0.25
    for w in ws:
        opt.state[w]["_fused_pending"] = False
    # back to the separate pass: contiguous weights and moments, same values, nothing fused
    assert opt.unfuse(m) is opt and m.engine.fused_opt is None
    assert all(w.is_contiguous() for w in ws) and [id(w) for (w, _, _) in m.layer_list()] == ids
    assert all(torch.equal(m.state_dict()[k], 2 * ref[k]) for k in ref)
    assert opt.state[ws[0]]["exp_avg"].is_contiguous() and float(opt.state[ws[0]]["exp_avg"].min()) == 0.5
    # moments that arrive in the other layout (a checkpoint written while the weight was seated) follow the contiguous weight in the
    # separate pass too: W, exp_avg, exp_avg_sq are addressed with one leading dimension
    seated = torch.zeros(700, 3040)[:, :3010]
    seated.copy_(opt.state[ws[0]]["exp_avg"])
    opt.state[ws[0]]["exp_avg"] = seated
    st = opt._init_state(ws[0])
    assert st["exp_avg"].is_contiguous() and float(st["exp_avg"].min()) == 0.5 and st["exp_avg_sq"].is_contiguous()
    # the switch: GDMCF_ALIGN_ROWS=0 fuses without moving anything
    monkeypatch.setenv("GDMCF_ALIGN_ROWS", "0")
    opt.fuse_into_backward(m, min_numel=1 << 12)
    assert all(w.is_contiguous() for w in ws) and m.engine.fused_opt is opt
