"""GPU (-m gpu): LightGCN propagation at the graph sizes its kernels are SELECTED for (reference lightGCN.py:180-194), and
a data-parallel rehearsal at the per-rank shape of BASELINE configs[3].

* `LightGCN.propagate_through_layers` with the auto-selected kernel (no GDMCF_SPMM_GEN override) on the synthetic Yelp
  graph (88 969 nodes, streamed third-generation schedule: 4 096 waves, 8 XCD column ranges), the Amazon-Book graph
  (203 771 nodes, first-generation kernels) and the config-5 graph (1.2 M nodes, 4e7 directed nonzeros, d = 64, 3 layers:
  "HBM-bound SpMM stress", BASELINE configs[4]) against the float64 scipy CSR product of the SAME float32 adjacency;
  tolerance as tests/test_gpu_parity.py::test_spmm_random_graphs_against_float64 (2e-5 of max|ref|, absolute).
* the config-5 graph row-sharded over 2 ranks (`shard_rows=True`, gloo, both ranks on the one GPU of the test box):
  every rank's propagated tables equal the float64 product too, and each other bit for bit.
* 2 ranks x 50 rows at the Yelp width (I = 34 395, dims=[1000], T = 5: the per-GPU share of a global batch of 400 on 8
  GPUs is 50 rows) through DataParallelStep: the mean loss of the GLOBAL batch and the replicas' weights against the CPU
  oracle (loss <= 1e-4 relative: north_star).
"""
import functools
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gdmcf_amd
from gdmcf_amd import data as D
from oracle import gdmcf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LAYERS, DIM = 3, 64


@functools.lru_cache(maxsize=1)
def _graph(shape):
    indptr, indices, n_items = D.synth_csr(shape, seed=0)
    n_users = D.SHAPES[shape]["n_users"]
    users = np.repeat(np.arange(n_users), np.diff(indptr))
    return users, indices.astype(np.int64), n_users, n_items


def _float64_mean_of_layers(users, items, U, It, E0, L):
    """mean_l (A~^l E0), l = 0..L, in float64 on the float32 adjacency the product builds (reference :145-178, :184-189)."""
    import scipy.sparse as sp
    from gdmcf_amd.lightgcn import normalized_bipartite_csr
    indptr, indices, vals = normalized_bipartite_csr(users, items, U, It)
    A = sp.csr_matrix((vals.astype(np.float64), indices, indptr), shape=(U + It, U + It))
    cur = acc = E0.astype(np.float64)
    for _ in range(L):
        cur = A @ cur
        acc = acc + cur
    return acc / (L + 1), int(indices.size)


@pytest.mark.parametrize("shape,expect_streamed,min_nodes,min_nnz", [
    ("yelp", True, 88969, 1_900_000),
    ("amazon-book", False, 203771, 4_300_000),
    ("stress", False, 1_200_000, 39_000_000),
])
def test_propagation_at_the_selected_kernels_graph_size(shape, expect_streamed, min_nodes, min_nnz, monkeypatch):
    monkeypatch.delenv("GDMCF_SPMM_GEN", raising=False)  # the product's own choice (lightgcn.py: SPMM_SLICE_MB)
    users, items, U, It = _graph(shape)
    assert U + It >= min_nodes
    rng = np.random.default_rng(7)
    E0 = (rng.standard_normal((U + It, DIM)) * 0.1).astype(np.float32)
    m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": items}, U, It, LAYERS, DIM, device=DEV)
    assert m._streamed == expect_streamed and not m._bundled, (shape, m._streamed, m._bundled)
    if expect_streamed:
        assert m._plan["n_waves"] == 4096
    with torch.no_grad():
        m.E0.weight.copy_(torch.from_numpy(E0))
    m = m.to(DEV)
    with torch.no_grad():
        fu, fi, iu, ii = m.propagate_through_layers()
        fu2, fi2, _, _ = m.propagate_through_layers()
    got = torch.cat([fu, fi]).cpu().numpy()
    assert torch.equal(fu, fu2) and torch.equal(fi, fi2)  # fixed schedule, no atomics: bit-identical reruns
    np.testing.assert_array_equal(torch.cat([iu, ii]).detach().cpu().numpy(), E0)
    ref, nnz = _float64_mean_of_layers(users, items, U, It, E0, LAYERS)
    assert nnz >= min_nnz and m.nnz == nnz
    tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max())
    assert err <= tol, (shape, err, tol)
    # the layers themselves, not only their mean (a wrong row would be diluted by 1/(L+1) otherwise)
    with torch.no_grad():
        _, _, _, _, layers = m.propagate_through_layers(return_layers=True)
    import scipy.sparse as sp
    from gdmcf_amd.lightgcn import normalized_bipartite_csr
    indptr, indices, vals = normalized_bipartite_csr(users, items, U, It)
    A = sp.csr_matrix((vals.astype(np.float64), indices, indptr), shape=(U + It, U + It))
    l1 = A @ E0.astype(np.float64)
    assert float(np.abs(layers[0].cpu().numpy() - l1).max()) <= 2e-5 * max(1.0, float(np.abs(l1).max()))


def _shard_worker(rank, port, path, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    z = np.load(path)
    users, items, E0 = z["users"], z["items"], z["E0"]
    U, It = int(z["U"]), int(z["It"])
    m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": items}, U, It, LAYERS, DIM, device=DEV, shard_rows=True)
    with torch.no_grad():
        m.E0.weight.copy_(torch.from_numpy(E0))
    m = m.to(DEV)
    with torch.no_grad():
        fu, fi, _, _ = m.propagate_through_layers()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"shard{rank}.npy"), torch.cat([fu, fi]).cpu().numpy())
    dist.destroy_process_group()


def test_config5_graph_row_sharded_over_two_ranks(tmp_path, monkeypatch):
    """config 5's graph (1.2 M nodes) with every rank owning half of the adjacency's rows: a layer = local SpMM over the
    full table + all-gather of the row blocks.  Both ranks end with the full, identical, correct tables."""
    monkeypatch.delenv("GDMCF_SPMM_GEN", raising=False)
    users, items, U, It = _graph("stress")
    rng = np.random.default_rng(11)
    E0 = (rng.standard_normal((U + It, DIM)) * 0.1).astype(np.float32)
    path = str(tmp_path / "graph.npz")
    np.savez(path, users=users, items=items, E0=E0, U=U, It=It)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_shard_worker, args=(port, path, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "shard0.npy"), np.load(tmp_path / "shard1.npy")
    np.testing.assert_array_equal(a, b)
    ref, _ = _float64_mean_of_layers(users, items, U, It, E0, LAYERS)
    assert float(np.abs(a - ref).max()) <= 2e-5 * max(1.0, float(np.abs(ref).max()))


# ---- data-parallel rehearsal at the per-rank shape of BASELINE configs[3] -----------------------------------------------
B_RANK, I_Y, HID_Y, T_Y, DP_STEPS = 50, 34395, 1000, 5, 2


def _dp_inputs(step, world):
    g = torch.Generator().manual_seed(500 + step)
    B = B_RANK * world
    x = (torch.rand(B, I_Y, generator=g) < 0.00075).float()
    ts = torch.randint(0, T_Y, (B,), generator=g)
    noise = torch.randn(B, I_Y, generator=g)
    keep = (torch.rand(B, I_Y, generator=g) < 0.5).float()
    return x, ts, torch.ones(B, dtype=torch.float64), noise, keep


def _dp_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    from gdmcf_amd.parallel import DataParallelStep
    torch.manual_seed(0)
    model = gdmcf_amd.DNN([I_Y, HID_Y], [HID_Y, I_Y], 10)
    torch.manual_seed(0)
    model.load_state_dict(O.DNN([I_Y, HID_Y], [HID_Y, I_Y], 10).state_dict())  # the oracle's initial weights, literally
    model = model.to(DEV).train()
    diff = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T_Y, DEV)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-5, weight_decay=0.0)
    step = DataParallelStep(diff, model, opt)
    losses = []
    lo, hi = rank * B_RANK, (rank + 1) * B_RANK
    for s in range(DP_STEPS):
        x, ts, pt, noise, keep = [t[lo:hi].to(DEV) for t in _dp_inputs(s, 2)]
        losses.append(float(step(x, True, ts=ts, pt=pt, noise=noise, drop_mask=keep)))
    torch.cuda.synchronize()
    keys = ["in_layers.0.bias", "out_layers.0.bias", "emb_layer.weight"]
    sd = model.state_dict()
    torch.save(dict(losses=losses, small={k: sd[k].cpu() for k in keys}, w2_rows=sd["out_layers.0.weight"][:64].cpu(),
                    hist=diff.Lt_history.cpu(), cnt=diff.Lt_count.cpu()), os.path.join(out_dir, f"dp{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_of_fifty_rows_match_the_oracle_on_the_global_batch(tmp_path):
    """BASELINE configs[3] splits a batch of 400 over 8 GPUs: 50 rows per rank.  Two such ranks (global batch 100) against
    the CPU ORACLE on the global batch: mean loss per step (<= 1e-4 relative), Lt-history bookkeeping, weights after AdamW."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "dp0.pt"), torch.load(tmp_path / "dp1.pt")
    torch.manual_seed(0)
    om = O.DNN([I_Y, HID_Y], [HID_Y, I_Y], 10).train()
    od = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T_Y)
    oopt = O.make_optimizer(om, 1e-5)
    ref = []
    for s in range(DP_STEPS):
        x, ts, pt, noise, keep = _dp_inputs(s, 2)
        oloss, _ = O.train_step(od, om, oopt, x, True, ts=ts, pt=pt, noise=noise, drop_mask=keep)
        ref.append(float(oloss))
    got = 0.5 * (np.array(r0["losses"]) + np.array(r1["losses"]))  # equal shard sizes: global mean = mean of local means
    np.testing.assert_allclose(got, ref, rtol=1e-4)
    osd = om.state_dict()
    for r in (r0, r1):
        np.testing.assert_array_equal(r["cnt"].numpy(), od.Lt_count.numpy())
        np.testing.assert_allclose(r["hist"].numpy(), od.Lt_history.numpy(), rtol=1e-4)
        for k, v in r["small"].items():
            assert float((v - osd[k]).abs().max()) < 0.25 * 1e-5 * DP_STEPS, k
        assert float((r["w2_rows"] - osd["out_layers.0.weight"][:64]).abs().max()) < 0.25 * 1e-5 * DP_STEPS
    for k in r0["small"]:
        assert torch.equal(r0["small"][k], r1["small"][k])  # replicas stay bit-identical
    assert torch.equal(r0["w2_rows"], r1["w2_rows"])


@pytest.mark.parametrize("B,N,K", [(400, 34395, 1000), (400, 1000, 34405), (400, 94949, 1000), (400, 1000, 94959), (256, 5000, 777),
                                   (130, 4100, 515)])
def test_weight_gradient_product_every_element_against_float64(B, N, K):
    """gdmcf_linear_bwd_weight_f32 (dW = dZ^T A, db = sum_m rs_m dZ_m; reference main.py:350) through the C ABI at the shapes
    the register-streaming kernel serves (csrc/gemm_dr.hip: operands travel in registers that hand-counted waits guard), EVERY
    element against a float64 product, three launches each: a compiler-inserted copy of such a register -- seen during
    development -- shows up as a handful of elements off by O(1) among tens of millions, which sampled checks miss.
    Also: bit-identical from launch to launch (dynamic tile queue, fixed arithmetic)."""
    from gdmcf_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(B + N + K)
    ldz, lda = (N + 63) // 64 * 64, (K + 63) // 64 * 64
    dZ = torch.randn(B, ldz, generator=g).to(DEV)
    A = torch.randn(B, lda, generator=g).to(DEV)
    rs = (torch.rand(B, generator=g) + 0.5).to(DEV)
    ref = dZ[:, :N].double().t() @ A[:, :K].double()
    scale = float(ref.abs().max())
    outs = []
    for use_db, use_rs in [(0, 0), (1, 0), (1, 1)]:
        dW = torch.full((N, K), float("nan"), device=DEV)
        db = torch.full((N,), float("nan"), device=DEV)
        _lib.check(lib.gdmcf_linear_bwd_weight_f32(dZ.data_ptr(), ldz, A.data_ptr(), lda, rs.data_ptr() if use_rs else None, 0, B, N, K,
                                                   dW.data_ptr(), K, db.data_ptr() if use_db else None, 0, _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert float((dW.double() - ref).abs().max()) <= 2e-6 * scale * max(1.0, (B / 400) ** 0.5), (use_db, use_rs)
        if use_db:
            dref = (dZ[:, :N].double() * (rs.double()[:, None] if use_rs else 1.0)).sum(0)
            assert float((db.double() - dref).abs().max()) <= 2e-6 * float(dref.abs().max()), (use_db, use_rs)
        outs.append(dW)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # sporadic corruption (seen in development variants of this kernel in about half of the launches) needs repetition to show:
    # twenty more launches, each bit-identical to the first
    for _ in range(20):
        dW = torch.full((N, K), float("nan"), device=DEV)
        _lib.check(lib.gdmcf_linear_bwd_weight_f32(dZ.data_ptr(), ldz, A.data_ptr(), lda, None, 0, B, N, K, dW.data_ptr(), K, None, 0,
                                                   _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(dW, outs[0])


_HL_SCRIPT = r'''
import sys, torch
sys.path.insert(0, {root!r})
from gdmcf_amd import _lib
lib = _lib.load()
dev = "cuda:0"
for (B, N, K, bits) in [(400, 34395, 1000, 0), (400, 34395, 1000, 1), (400, 94949, 1000, 1), (240, 40000, 520, 0)]:
    g = torch.Generator(device="cpu").manual_seed(B + N + K + bits)
    h = torch.randn(B, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    alpha = (torch.rand(B, generator=g) + 0.5).to(dev)
    tgt = (torch.rand(B, N, generator=g) < 0.02).float().to(dev)
    ldd = (N + 31) // 32 * 32
    diff = torch.full((B, ldd), float("nan"), device=dev)
    nt = lib.gdmcf_loss_tiles(N)
    rowpart = torch.zeros(B * nt, device=dev)
    rowsum = torch.zeros(B, device=dev)
    if bits:
        words = (N + 31) // 32
        pad = torch.zeros(B, words * 32, device=dev)
        pad[:, :N] = tgt
        packed = (pad.view(B, words, 32).to(torch.int64) << torch.arange(32, device=dev)).sum(-1)  # bit n & 31 of word n >> 5
        packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32).contiguous()
        _lib.check(lib.gdmcf_linear_loss_fwd_bits_f32(h.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), packed.data_ptr(), words,
                                                      alpha.data_ptr(), B, N, K, None, 0, diff.data_ptr(), ldd, rowpart.data_ptr(),
                                                      rowsum.data_ptr(), _lib.stream_ptr()))
    else:
        _lib.check(lib.gdmcf_linear_loss_fwd_f32(h.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), tgt.data_ptr(), N,
                                                 alpha.data_ptr(), B, N, K, None, 0, diff.data_ptr(), ldd, rowpart.data_ptr(),
                                                 rowsum.data_ptr(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    ref = alpha.double()[:, None] * (h.double() @ W.double().t() + bias.double()) - tgt.double()
    err = float((diff[:, :N].double() - ref).abs().max())
    rs = float(((rowsum.double() - (ref * ref).sum(1)).abs() / (ref * ref).sum(1)).max())
    print("case", B, N, K, bits, "max err", err, "row sums", rs)
    used = (B * ((N + 63) // 64) + 3) // 4 * 4  # the hybrid kernel keeps its transposed copy of h behind the row partials:
    assert torch.equal(rowpart[used:used + K * B].view(K, B), h.t()), "the hybrid kernel did not run"  # proof that it ran
    assert err <= 2e-5 * float(ref.abs().max()), err
    assert rs <= 2e-6, rs
print("HL-OK")
'''


def test_output_layer_loss_product_on_the_opt_in_hybrid_kernel(tmp_path):
    """csrc/gemm_dr.hip dr_hl_kernel (GDMCF_GEMM_DR bit 3: small operand pre-transposed and register-streamed, large operand
    through wave-private LDS) through the C ABI of the fused loss layer (gaussian_diffusion.py:335): every element
    of alpha * (h W^T + b) - target and the row sums of its square against float64, float and bitmap targets, Yelp and
    Amazon-Book widths and a ragged shape.  The switch is read once per process, so the check runs in ONE child process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "hl_check.py"
    script.write_text(_HL_SCRIPT.format(root=root))
    env = dict(os.environ, GDMCF_GEMM_DR="9")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "HL-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("which,B,N,K", [("fwd", 400, 1000, 34405), ("fwd", 400, 1000, 94959), ("dinput", 400, 34395, 1000),
                                         ("dinput", 400, 94949, 1000), ("dinput_lds", 400, 34395, 1000), ("dinput", 230, 20003, 1000),
                                         ("fwd_wt", 400, 1000, 34405), ("fwd_wt", 400, 1000, 94959), ("fwd_wt", 230, 1000, 20003),
                                         ("fwd", 200, 999, 20003), ("loss_lds", 400, 34395, 1000), ("loss_lds", 400, 94949, 1000)])
def test_lds_tiled_products_every_element_against_float64_twenty_launches(which, B, N, K, tmp_path):
    """The LDS-tiled f32 kernel (csrc/gemm_f32.hip) keeps two tiles in flight in registers that inline-asm loads write and
    hand-counted s_waitcnt guard; its waits are chosen by the same predicate as its loads, which the build's path-insensitive
    lint (build.py:lint_vmcnt) cannot follow -- so its guard is THIS test: the first-layer forward product
    (gdmcf_linear_fwd_f32: split-K slabs + reducer, bias, tanh; reference models/DNN.py:79-81), the input-gradient product
    (gdmcf_linear_bwd_input_f32; main.py:350: on dr_kn_kernel, both operands straight into registers, by default -- a ragged shape
    with row / column / K tails included -- and on the LDS-tiled kernel with GDMCF_GEMM_DR=17) and the fused-loss product ON THE
    LDS-TILED KERNEL (GDMCF_GEMM_DR=1: the fat-tile kernel that serves it by default since round 4 is switched off) at the Yelp and Amazon-Book shapes and a ragged one: EVERY
    element against float64, then twenty more launches bit for bit (a register consumed before its load has landed shows up as
    a few elements off in SOME launches)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "lds_check.py"
    script.write_text(_LDS_SCRIPT.format(root=root, which=which.replace("_lds", "") if which.startswith("dinput") else which, B=B, N=N, K=K))
    env = dict(os.environ, GDMCF_GEMM_DR="1") if which == "loss_lds" else dict(os.environ)
    if which == "dinput_lds":  # the input gradient ON THE LDS-TILED KERNEL (dr_kn_kernel, its default since round 4, switched off)
        env["GDMCF_GEMM_DR"] = "17"
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "LDS-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


_LDS_SCRIPT = r'''
import sys, torch
sys.path.insert(0, {root!r})
from gdmcf_amd import _lib
lib = _lib.load()
dev = "cuda:0"
which, B, N, K = {which!r}, {B}, {N}, {K}
g = torch.Generator(device="cpu").manual_seed(B + N + K)
st = _lib.stream_ptr()
if which == "fwd":        # C[B, N] = tanh(A[B, K] W[N, K]^T + b)
    A = torch.randn(B, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.01).to(dev); bias = torch.randn(N, generator=g).to(dev)
    ws = torch.empty(max(int(lib.gdmcf_linear_ws_bytes(B, N, K)), 256), dtype=torch.uint8, device=dev)
    def run():
        out = torch.full((B, N), float("nan"), device=dev)
        _lib.check(lib.gdmcf_linear_fwd_f32(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), 1, B, N, K, out.data_ptr(), N,
                                            ws.data_ptr(), ws.numel(), st))
        torch.cuda.synchronize()
        return out
    ref = torch.tanh(A.double() @ W.double().t() + bias.double())
    tol = 2e-5
elif which == "fwd_wt":   # the same layer through the TRANSPOSED weight (gdmcf_linear_fwd_wt_f32: the reverse loop's hidden layer)
    A = torch.randn(B, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.01).to(dev); bias = torch.randn(N, generator=g).to(dev)
    ldn = (N + 31) // 32 * 32
    Wt = torch.full((K, ldn), float("nan"), device=dev); Wt[:, :N] = W.t()
    ws = torch.empty(max(int(lib.gdmcf_linear_ws_bytes(B, N, K)), 256), dtype=torch.uint8, device=dev)
    def run():
        out = torch.full((B, N), float("nan"), device=dev)
        _lib.check(lib.gdmcf_linear_fwd_wt_f32(A.data_ptr(), K, Wt.data_ptr(), ldn, bias.data_ptr(), 1, B, N, K, out.data_ptr(), N,
                                               ws.data_ptr(), ws.numel(), st))
        torch.cuda.synchronize()
        assert lib.gdmcf_debug_last_gemm() == 5, "dr_kn_kernel did not serve this product"
        return out
    ref = torch.tanh(A.double() @ W.double().t() + bias.double())
    tol = 2e-5
elif which == "dinput":   # dA[B, K] = rs * (dZ[B, N] W[N, K]) * (1 - act^2)
    dZ = torch.randn(B, N, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.01).to(dev)
    rs = (torch.rand(B, generator=g) + 0.5).to(dev); act = torch.tanh(torch.randn(B, K, generator=g)).to(dev)
    ws = torch.empty(max(int(lib.gdmcf_linear_ws_bytes(B, N, K)), 256), dtype=torch.uint8, device=dev)
    def run():
        out = torch.full((B, K), float("nan"), device=dev)
        _lib.check(lib.gdmcf_linear_bwd_input_f32(dZ.data_ptr(), N, W.data_ptr(), K, rs.data_ptr(), act.data_ptr(), K, 1, B, N, K,
                                                  out.data_ptr(), K, ws.data_ptr(), ws.numel(), st))
        torch.cuda.synchronize()
        # since the end of round 4 the register-streaming kernel dr_kn_kernel serves this product (5); GDMCF_GEMM_DR=17 -> LDS-tiled (1)
        import os
        want = 1 if os.environ.get("GDMCF_GEMM_DR") == "17" else 5
        assert lib.gdmcf_debug_last_gemm() == want, (lib.gdmcf_debug_last_gemm(), want)
        return out
    ref = rs.double()[:, None] * (dZ.double() @ W.double()) * (1 - act.double() ** 2)
    tol = 2e-5
else:                     # the fused-loss product on the LDS-tiled kernel
    h = torch.randn(B, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.05).to(dev); bias = torch.randn(N, generator=g).to(dev)
    tgt = (torch.rand(B, N, generator=g) < 0.02).float().to(dev)
    ldd = (N + 31) // 32 * 32
    rowpart = torch.zeros(B * lib.gdmcf_loss_tiles(N), device=dev); rowsum = torch.zeros(B, device=dev)
    def run():
        diff = torch.full((B, ldd), float("nan"), device=dev)
        _lib.check(lib.gdmcf_linear_loss_fwd_f32(h.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), tgt.data_ptr(), N, None, B, N, K,
                                                 None, 0, diff.data_ptr(), ldd, rowpart.data_ptr(), rowsum.data_ptr(), st))
        torch.cuda.synchronize()
        assert lib.gdmcf_debug_last_gemm() == 1, "the LDS-tiled kernel did not serve this product"
        return diff[:, :N]
    ref = h.double() @ W.double().t() + bias.double() - tgt.double()
    tol = 2e-5
first = run()
err = float((first.double() - ref).abs().max()) / float(ref.abs().max())
print("max err / max|ref|", err)
assert err <= tol, err
for rep in range(20):
    assert torch.equal(run(), first), rep
print("LDS-OK")
'''


@pytest.mark.parametrize("B,N,K,bits", [(400, 34395, 1000, 0), (400, 34395, 1000, 1), (400, 94949, 1000, 1), (240, 40000, 520, 0),
                                        (400, 200000, 2000, 1)])
def test_output_layer_loss_product_on_the_fat_tile_kernel(B, N, K, bits):
    """csrc/gemm_dr.hip dr_fat_kernel (round 4, the DEFAULT for batch-sized M: one 80 x 16 NB tile per wave, one wave per SIMD, one
    pass) through the C ABI of the fused loss layer (reference models/DNN.py:83-86 + gaussian_diffusion.py:335): EVERY element
    of alpha * (h W^T + b) - target and the row sums of its square against float64 -- float and bitmap targets, the Yelp,
    Amazon-Book and stress widths (tile widths 11, 10, ... blocks; one and several rounds of tiles) and a ragged shape with a
    K tail -- twenty launches bit for bit, and the proof that this kernel (not the LDS-tiled one) served the call."""
    from gdmcf_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(B + N + K + bits)
    h = torch.randn(B, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    alpha = (torch.rand(B, generator=g) + 0.5).to(DEV)
    tgt = (torch.rand(B, N, generator=g) < 0.02).float().to(DEV)
    ldd = (N + 31) // 32 * 32
    nt = lib.gdmcf_loss_tiles(N)
    words = (N + 31) // 32
    packed = None
    if bits:
        pad = torch.zeros(B, words * 32, device=DEV)
        pad[:, :N] = tgt
        packed = (pad.view(B, words, 32).to(torch.int64) << torch.arange(32, device=DEV)).sum(-1)  # bit n & 31 of word n >> 5
        packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32).contiguous()
        del pad

    def run():
        diff = torch.full((B, ldd), float("nan"), device=DEV)
        rowpart = torch.zeros(B * nt, device=DEV)
        rowsum = torch.zeros(B, device=DEV)
        if bits:
            _lib.check(lib.gdmcf_linear_loss_fwd_bits_f32(h.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), packed.data_ptr(), words,
                                                          alpha.data_ptr(), B, N, K, None, 0, diff.data_ptr(), ldd, rowpart.data_ptr(),
                                                          rowsum.data_ptr(), _lib.stream_ptr()))
        else:
            _lib.check(lib.gdmcf_linear_loss_fwd_f32(h.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), tgt.data_ptr(), N,
                                                     alpha.data_ptr(), B, N, K, None, 0, diff.data_ptr(), ldd, rowpart.data_ptr(),
                                                     rowsum.data_ptr(), _lib.stream_ptr()))
        torch.cuda.synchronize()
        return diff, rowsum

    diff, rowsum = run()
    assert lib.gdmcf_debug_last_gemm() == 4, "the fat-tile kernel did not serve this product"
    ref = alpha.double()[:, None] * (h.double() @ W.double().t() + bias.double()) - tgt.double()
    assert float((diff[:, :N].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    rs = (ref * ref).sum(1)
    assert float(((rowsum.double() - rs).abs() / rs).max()) <= 2e-6
    del ref, rs
    for _ in range(20 if N < 100000 else 3):
        d2, r2 = run()
        assert torch.equal(d2[:, :N], diff[:, :N]) and torch.equal(r2, rowsum)


@pytest.mark.parametrize("eps_mode,noise", [(False, False), (True, True)])
def test_reverse_step_product_on_the_fat_tile_kernel(eps_mode, noise):
    """The same kernel with the posterior epilogue (gdmcf_linear_posterior_fwd_f32; reference gaussian_diffusion.py:451-471,
    :495-498, :210-217) at the Yelp width: every element of x_{t-1} and of pred_xstart against float64."""
    from gdmcf_amd import _lib
    lib = _lib.load()
    B, N, K = 400, 34395, 1000
    g = torch.Generator(device="cpu").manual_seed(5 + eps_mode)
    h = torch.randn(B, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    xt = torch.randn(B, N, generator=g).to(DEV)
    c1, c2 = (torch.rand(B, generator=g) + 0.2).to(DEV), (torch.rand(B, generator=g) + 0.2).to(DEV)
    r1 = (torch.rand(B, generator=g) + 1.0).to(DEV) if eps_mode else None
    r2 = (torch.rand(B, generator=g) * 0.1).to(DEV) if eps_mode else None
    sg = (torch.rand(B, generator=g) * 0.01).to(DEV) if noise else None
    z = torch.randn(B, N, generator=g).to(DEV) if noise else None
    xn = torch.full((B, N), float("nan"), device=DEV)
    pred = torch.full((B, N), float("nan"), device=DEV)
    _lib.check(lib.gdmcf_linear_posterior_fwd_f32(h.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), xt.data_ptr(), N, c1.data_ptr(),
                                                  c2.data_ptr(), _lib.ptr(r1), _lib.ptr(r2), _lib.ptr(sg), _lib.ptr(z),
                                                  N if noise else 0, B, N, K, xn.data_ptr(), N, pred.data_ptr(), N,
                                                  _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert lib.gdmcf_debug_last_gemm() == 4
    out = h.double() @ W.double().t() + bias.double()
    p = (r1.double()[:, None] * xt.double() - r2.double()[:, None] * out) if eps_mode else out
    mean = c1.double()[:, None] * p + c2.double()[:, None] * xt.double()
    if noise:
        mean = mean + sg.double()[:, None] * z.double()
    assert float((pred.double() - p).abs().max()) <= 2e-5 * float(p.abs().max())
    assert float((xn.double() - mean).abs().max()) <= 2e-5 * float(mean.abs().max())


@pytest.mark.parametrize("B,N,K", [(400, 34395, 1000), (400, 1000, 34405), (400, 94949, 1000), (130, 4100, 515)])
def test_fused_adamw_weight_gradient_every_element_against_float64(B, N, K):
    """gdmcf_linear_bwd_weight_adamw_f32 (dW = dZ^T A never leaves the accumulators; W, exp_avg, exp_avg_sq updated in the
    epilogue with torch.optim.AdamW's single-tensor math; reference main.py:350-351) at the shapes the register-streaming
    kernel serves: EVERY element of the three tensors against the float64 update of the float64 product, and bit-identical
    when repeated from the same state."""
    from gdmcf_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(7 * B + N + K)
    ldz, lda = (N + 63) // 64 * 64, (K + 63) // 64 * 64
    dZ = (torch.randn(B, ldz, generator=g) * 0.1).to(DEV)
    A = torch.randn(B, lda, generator=g).to(DEV)
    W0 = (torch.randn(N, K, generator=g) * 0.05).to(DEV)
    m0 = (torch.randn(N, K, generator=g) * 0.01).to(DEV)
    v0 = (torch.rand(N, K, generator=g) * 1e-3).to(DEV)
    lr, b1, b2, eps, wd, step, gs = 1e-3, 0.9, 0.999, 1e-8, 0.01, 3, 0.5
    gref = (dZ[:, :N].double().t() @ A[:, :K].double()) * gs
    p = W0.double() * (1 - lr * wd)
    m = m0.double() + (gref - m0.double()) * (1 - b1)
    v = v0.double() * b2 + (1 - b2) * gref * gref
    p = p - (lr / (1 - b1 ** step)) * m / (v.sqrt() / (1 - b2 ** step) ** 0.5 + eps)
    outs = []
    for _ in range(2):
        W, me, ve = W0.clone(), m0.clone(), v0.clone()
        _lib.check(lib.gdmcf_linear_bwd_weight_adamw_f32(dZ.data_ptr(), ldz, A.data_ptr(), lda, None, 0, B, N, K, W.data_ptr(), K,
                                                         me.data_ptr(), ve.data_ptr(), None, lr, b1, b2, eps, wd, step, gs,
                                                         _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert float((me.double() - m).abs().max()) <= 4e-6 * float(m.abs().max())
        assert float((ve.double() - v).abs().max()) <= 4e-6 * float(v.abs().max())
        assert float((W.double() - p).abs().max()) <= 4e-6 * float(p.abs().max()) + 2e-5 * lr
        outs.append((W, me, ve))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))


@pytest.mark.parametrize("B,N,K", [(400, 34395, 1000), (400, 94949, 1000), (256, 5000, 777), (130, 4100, 515)])
def test_bias_gradient_as_a_column_of_the_weight_gradient_product(B, N, K):
    """gdmcf_rowscale_f32 writes the row scale into column K of its scaled copy (ldo > K); gdmcf_linear_bwd_weight_f32 on that copy
    then takes db[n] = sum_m rs[m] dZ[m, n] (reference: the bias gradient of main.py:350's backward) out of the product as its
    column K instead of reading dZ a second time.  Every element of dW and db against float64; the same call on a copy without
    the spare column (ldo == K) goes through the column-sum pass and must agree; twenty repetitions bit for bit (the variant of
    the kernel that first carried this column is the one that exposed the store-data hazard of DESIGN 4.1b)."""
    from gdmcf_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(3 * B + N + K)
    ldz = (N + 63) // 64 * 64
    dZ = torch.randn(B, ldz, generator=g).to(DEV)
    h = torch.randn(B, K, generator=g).to(DEV)
    rs = (torch.rand(B, generator=g) + 0.5).to(DEV)
    ref = dZ[:, :N].double().t() @ (h.double() * rs.double()[:, None])
    dref = (dZ[:, :N].double() * rs.double()[:, None]).sum(0)
    res = []
    for ldo in (K + 24, K):
        hs = torch.zeros(B, ldo, device=DEV)
        for rep in range(21 if ldo > K else 1):
            _lib.check(lib.gdmcf_rowscale_f32(h.data_ptr(), K, rs.data_ptr(), B, K, hs.data_ptr(), ldo, _lib.stream_ptr()))
            dW = torch.full((N, K), float("nan"), device=DEV)
            db = torch.full((N,), float("nan"), device=DEV)
            # a_scale_col: the caller's statement that column K of the operand holds the row scale (ldo > K only)
            _lib.check(lib.gdmcf_linear_bwd_weight_f32(dZ.data_ptr(), ldz, hs.data_ptr(), ldo, rs.data_ptr(), int(ldo > K), B, N, K,
                                                       dW.data_ptr(), K, db.data_ptr(), 0, _lib.stream_ptr()))
            torch.cuda.synchronize()
            if rep == 0:
                if ldo > K:
                    assert torch.equal(hs[:, K], rs)
                assert float((dW.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) * max(1.0, (B / 400) ** 0.5), ldo
                assert float((db.double() - dref).abs().max()) <= 2e-6 * float(dref.abs().max()) * max(1.0, (B / 400) ** 0.5), ldo
                res.append((dW, db))
            else:
                assert torch.equal(dW, res[-1][0]) and torch.equal(db, res[-1][1]), rep
    # No state travels between the two calls (the former thread-local record matched on addresses): a caller that
    # REWRITES the scaled copy after gdmcf_rowscale_f32 -- same address, same row scale, column K now garbage -- and does not
    # claim the column gets db from the column-sum pass, right; and the flag alone decides, whatever ran before.
    ldo = K + 24
    hs = torch.zeros(B, ldo, device=DEV)
    _lib.check(lib.gdmcf_rowscale_f32(h.data_ptr(), K, rs.data_ptr(), B, K, hs.data_ptr(), ldo, _lib.stream_ptr()))
    hs[:, K] = 123.0   # the column is gone
    hs[:, :K] = h * rs[:, None] * 2.0  # and the operand is another one
    dW = torch.full((N, K), float("nan"), device=DEV)
    db = torch.full((N,), float("nan"), device=DEV)
    _lib.check(lib.gdmcf_linear_bwd_weight_f32(dZ.data_ptr(), ldz, hs.data_ptr(), ldo, rs.data_ptr(), 0, B, N, K, dW.data_ptr(), K,
                                               db.data_ptr(), 0, _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert float((db.double() - dref).abs().max()) <= 2e-6 * float(dref.abs().max()) * max(1.0, (B / 400) ** 0.5)
    assert float((dW.double() - 2.0 * ref).abs().max()) <= 4e-6 * float(ref.abs().max()) * max(1.0, (B / 400) ** 0.5)
    # the same operand without any preceding gdmcf_rowscale_f32 call, the column written by the caller itself: flag honoured
    hs2 = torch.zeros(B, ldo, device=DEV)
    hs2[:, :K] = h * rs[:, None]
    hs2[:, K] = rs
    db2 = torch.full((N,), float("nan"), device=DEV)
    _lib.check(lib.gdmcf_linear_bwd_weight_f32(dZ.data_ptr(), ldz, hs2.data_ptr(), ldo, rs.data_ptr(), 1, B, N, K, dW.data_ptr(), K,
                                               db2.data_ptr(), 0, _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(db2, res[0][1])
    assert torch.equal(res[0][0], res[1][0])  # the product itself does not change with the extra column
    assert float((res[0][1].double() - res[1][1].double()).abs().max()) <= 4e-6 * float(dref.abs().max())


def test_products_take_weights_whose_rows_sit_on_128_byte_lines():
    """FusedAdamW.fuse_into_backward seats a fused weight on rows of round_up(in, 32) floats (optim.py).  Every product that
    reads or updates a weight (reference models/DNN.py:79-86, main.py:350-351) must give the SAME BITS with that leading
    dimension as with PyTorch's contiguous one, at the Yelp shapes: the hidden layer, the fused-loss layer on the fat-tile
    kernel -- whose last 16-k chunk reaches past K = 1000 into the padding, filled with NaN here: both operands are masked --
    the input gradient, and the fused weight-gradient + AdamW products, which must also leave the padding of W / exp_avg /
    exp_avg_sq untouched."""
    from gdmcf_amd import _lib
    lib = _lib.load()
    B, I, H, E = 400, 34395, 1000, 10
    up = lambda n, a: (n + a - 1) // a * a
    g = torch.Generator(device="cpu").manual_seed(5)
    ldk, ldi, ldh = up(I + E, 64), up(I, 64), 1024
    xin = torch.zeros(B, ldk, device=DEV)
    xin[:, :I + E] = torch.randn(B, I + E, generator=g).to(DEV)
    hs = torch.zeros(B, ldh, device=DEV)
    hs[:, :H] = torch.randn(B, H, generator=g).to(DEV)
    dz2 = torch.zeros(B, ldi, device=DEV)
    dz2[:, :I] = (torch.randn(B, I, generator=g) * 0.01).to(DEV)
    tgt = (torch.rand(B, ldi, generator=g) < 0.002).float().to(DEV)
    b1, b2 = torch.randn(H, generator=g).to(DEV), torch.randn(I, generator=g).to(DEV)
    W1c = (torch.randn(H, I + E, generator=g) * 0.01).to(DEV)
    W2c = (torch.randn(I, H, generator=g) * 0.01).to(DEV)
    ws_bytes = max(lib.gdmcf_linear_ws_bytes(B, H, I + E), lib.gdmcf_linear_ws_bytes(B, I, H), 1 << 20)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    st = _lib.stream_ptr()

    def seated(t, fill):
        n, k = t.shape
        buf = torch.full((n, up(k, 32)), fill, device=DEV)
        buf[:, :k] = t
        return buf

    def products(W1, W2):
        ld1, ld2 = W1.stride(0), W2.stride(0)
        h_out = torch.full((B, ldh), float("nan"), device=DEV)
        _lib.check(lib.gdmcf_linear_fwd_f32(xin.data_ptr(), ldk, W1.data_ptr(), ld1, b1.data_ptr(), 1, B, H, I + E, h_out.data_ptr(),
                                            ldh, ws.data_ptr(), ws_bytes, st))
        diff = torch.full((B, ldi), float("nan"), device=DEV)
        rowpart = torch.zeros(B * lib.gdmcf_loss_tiles(I), device=DEV)
        rowsum = torch.zeros(B, device=DEV)
        _lib.check(lib.gdmcf_linear_loss_fwd_f32(hs.data_ptr(), ldh, W2.data_ptr(), ld2, b2.data_ptr(), tgt.data_ptr(), ldi, None, B, I,
                                                 H, None, 0, diff.data_ptr(), ldi, rowpart.data_ptr(), rowsum.data_ptr(), st))
        fat = lib.gdmcf_debug_last_gemm()
        dh = torch.full((B, ldh), float("nan"), device=DEV)
        _lib.check(lib.gdmcf_linear_bwd_input_f32(dz2.data_ptr(), ldi, W2.data_ptr(), ld2, None, hs.data_ptr(), ldh, 1, B, I, H,
                                                  dh.data_ptr(), ldh, ws.data_ptr(), ws_bytes, st))
        torch.cuda.synchronize()
        return h_out[:, :H].clone(), diff[:, :I].clone(), rowsum, dh[:, :H].clone(), fat

    ref = products(W1c, W2c)
    assert ref[4] == 4 and all(bool(torch.isfinite(t).all()) for t in ref[:4])
    # hidden layer / input gradient: zero padding (what the optimiser's seating leaves); fused-loss layer: NaN padding
    got = products(seated(W1c, 0.0)[:, :I + E], seated(W2c, float("nan"))[:, :H])
    assert got[4] == 4, "the fat-tile kernel refused the padded leading dimension"
    assert torch.equal(got[0], ref[0]), "hidden layer"
    assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), "fused-loss layer (padding behind K must be masked)"
    got0 = products(W1c, seated(W2c, 0.0)[:, :H])
    assert torch.equal(got0[3], ref[3]), "input gradient"
    del got, got0
    # the fused weight-gradient + AdamW products: real elements bit-identical, the padding (sentinel 7) untouched
    for (N, K, dZ, ldz, A, lda, Wc) in ((I, H, dz2, ldi, hs, ldh, W2c), (H, I + E, ref[0], None, xin, ldk, W1c)):
        if ldz is None:
            dZp = torch.zeros(B, ldh, device=DEV)
            dZp[:, :H] = dZ
            dZ, ldz = dZp, ldh
        outs = []
        for fill in (None, 7.0):
            if fill is None:
                W, m, v = Wc.clone(), torch.full_like(Wc, 0.01), torch.full_like(Wc, 1e-4)
            else:
                W, m, v = seated(Wc, fill), seated(torch.full_like(Wc, 0.01), fill), seated(torch.full_like(Wc, 1e-4), fill)
            _lib.check(lib.gdmcf_linear_bwd_weight_adamw_f32(dZ.data_ptr(), ldz, A.data_ptr(), lda, None, 0, B, N, K, W.data_ptr(),
                                                             W.stride(0), m.data_ptr(), v.data_ptr(), None, 1e-3, 0.9, 0.999, 1e-8,
                                                             0.01, 3, 1.0, st))
            torch.cuda.synchronize()
            assert lib.gdmcf_debug_last_gemm() == 3
            if fill is not None and W.stride(0) > K:
                assert all(bool((t[:, K:] == fill).all()) for t in (W, m, v)), "padding written"
            outs.append((W[:, :K], m[:, :K], v[:, :K]))
        assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), (N, K)
        del outs


def test_training_steps_with_seated_weights_equal_the_separate_pass_bit_for_bit(monkeypatch):
    """Five Yelp-shape training steps (reference main.py:343-351) three ways -- AdamW as a separate pass over contiguous weights;
    inside the weight-gradient products on contiguous weights (GDMCF_ALIGN_ROWS=0); inside them on weights seated on 128-byte
    rows (the default of fuse_into_backward) -- every loss, weight and moment identical bit for bit."""
    import scipy.sparse as sp
    from gdmcf_amd.data_utils import DeviceCSR
    from gdmcf_amd.parallel import DataParallelStep
    dev = torch.device(DEV)
    B, hid, T = 400, 1000, 5
    indptr, indices, I = D.synth_csr("yelp", n_rows=2 * B, seed=0)
    dcsr = DeviceCSR(sp.csr_matrix((np.ones(len(indices), np.float32), indices, indptr), shape=(2 * B, I)), dev)
    res = []
    for mode in ("separate", "fused-contiguous", "fused-seated"):
        monkeypatch.setenv("GDMCF_ALIGN_ROWS", "0" if mode == "fused-contiguous" else "1")
        torch.manual_seed(0)
        model = gdmcf_amd.DNN([I, hid], [hid, I], 10, time_type="cat", norm=False).to(dev).train()
        diffusion = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, dev)
        opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
        if mode != "separate":
            opt.fuse_into_backward(model)
        strides = [w.stride(0) for (w, _, _) in model.layer_list()]
        assert strides == ([34432, 1024] if mode == "fused-seated" else [I + 10, hid])
        step = DataParallelStep(diffusion, model, opt)
        torch.manual_seed(99)
        model.engine.manual_seed(7)
        losses = [float(step(dcsr.batch(torch.arange((i % 2) * B, (i % 2 + 1) * B, device=dev)), True)) for i in range(5)]
        torch.cuda.synchronize()
        assert [w.stride(0) for (w, _, _) in model.layer_list()] == strides
        res.append((losses, [p.detach().clone().contiguous() for p in model.parameters()],
                    [opt.state[p]["exp_avg"].clone().contiguous() for p in model.parameters()],
                    [opt.state[p]["exp_avg_sq"].clone().contiguous() for p in model.parameters()]))
        del model, opt, step
    for other in res[1:]:
        assert other[0] == res[0][0]
        for k in (1, 2, 3):
            assert all(torch.equal(a, b) for a, b in zip(res[0][k], other[k]))


def test_first_layer_bias_gradient_from_the_ones_column_of_the_input_builder():
    """gdmcf_dnn_prep_input_csr_f32 leaves 1 in column I + E of xin (its first padding column); the first layer's weight-gradient
    product (reference main.py:350: dW1 = dz1^T [x_t, emb], db1 = sum_m dz1) then carries db1 as its column K (a_scale_col with
    rowscale NULL).  Yelp shape through the C ABI: the column as written, dW1 / db1 of the plain and the fused-AdamW entry against
    float64 and against the column-sum pass (a_scale_col = 0), and the hidden layer's forward -- which reads xin up to column
    I + E only -- unchanged by what the column holds."""
    import scipy.sparse as sp
    from gdmcf_amd import _lib
    from gdmcf_amd.data_utils import DeviceCSR
    lib = _lib.load()
    dev = torch.device(DEV)
    B, H, E = 400, 1000, 10
    indptr, indices, I = D.synth_csr("yelp", n_rows=B, seed=0)
    dcsr = DeviceCSR(sp.csr_matrix((np.ones(len(indices), np.float32), indices, indptr), shape=(B, I)), dev)
    batch = dcsr.batch(torch.arange(B, device=dev))
    g = torch.Generator(device="cpu").manual_seed(17)
    ldk = (I + E + 63) // 64 * 64
    assert ldk > I + E
    xin = torch.full((B, ldk), float("nan"), device=DEV)
    temb = torch.zeros(B, E, device=DEV)
    bits = torch.zeros(B, (I + 31) // 32, dtype=torch.int32, device=DEV)
    ts = torch.randint(0, 5, (B,), generator=g).to(DEV)
    ca = (torch.rand(5, generator=g) * 0.5 + 0.5).to(DEV)
    cb = (torch.rand(5, generator=g) * 0.1).to(DEV)
    ew, eb = torch.randn(E, E, generator=g).to(DEV), torch.randn(E, generator=g).to(DEV)
    c = batch.csr
    st = _lib.stream_ptr()
    _lib.check(lib.gdmcf_dnn_prep_input_csr_f32(c.indptr.data_ptr(), c.indices.data_ptr(), batch.row_ids.data_ptr(), ts.data_ptr(),
                                                ca.data_ptr(), cb.data_ptr(), 2, None, 0, 2, None, 0, 0.5, 1234, 1, ew.data_ptr(),
                                                eb.data_ptr(), E, B, I, xin.data_ptr(), ldk, temb.data_ptr(), bits.data_ptr(),
                                                bits.stride(0), st))
    torch.cuda.synchronize()
    assert bool((xin[:, I + E] == 1.0).all()) and bool((xin[:, I + E + 1:] == 0.0).all()) and bool(torch.isfinite(xin).all())
    dz1 = torch.zeros(B, 1024, device=DEV)
    dz1[:, :H] = (torch.randn(B, H, generator=g) * 0.01).to(DEV)
    K = I + E
    ref = dz1[:, :H].double().t() @ xin[:, :K].double()
    dref = dz1[:, :H].double().sum(0)
    out = {}
    for scol in (1, 0):
        dW = torch.full((H, K), float("nan"), device=DEV)
        db = torch.full((H,), float("nan"), device=DEV)
        _lib.check(lib.gdmcf_linear_bwd_weight_f32(dz1.data_ptr(), 1024, xin.data_ptr(), ldk, None, scol, B, H, K, dW.data_ptr(), K,
                                                   db.data_ptr(), 0, st))
        torch.cuda.synchronize()
        assert float((dW.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
        assert float((db.double() - dref).abs().max()) <= 2e-6 * float(dref.abs().max())
        out[scol] = (dW, db)
    assert torch.equal(out[0][0], out[1][0])
    assert float((out[0][1] - out[1][1]).abs().max()) <= 4e-6 * float(dref.abs().max())
    # the fused entry: same db, from the column
    W = (torch.randn(H, K, generator=g) * 0.01).to(DEV)
    m, v = torch.zeros_like(W), torch.zeros_like(W)
    db = torch.full((H,), float("nan"), device=DEV)
    _lib.check(lib.gdmcf_linear_bwd_weight_adamw_f32(dz1.data_ptr(), 1024, xin.data_ptr(), ldk, None, 1, B, H, K, W.data_ptr(), K,
                                                     m.data_ptr(), v.data_ptr(), db.data_ptr(), 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, st))
    torch.cuda.synchronize()
    assert torch.equal(db, out[1][1])
    assert float((m.double() - 0.1 * ref).abs().max()) <= 4e-6 * float(0.1 * ref.abs().max())
    # the hidden layer reads K columns of xin: whatever column K holds
    W1 = (torch.randn(H, K, generator=g) * 0.01).to(DEV)
    b1 = torch.randn(H, generator=g).to(DEV)
    ws_bytes = max(lib.gdmcf_linear_ws_bytes(B, H, K), 1 << 20)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    hs = []
    for fill in (1.0, 0.0, float("nan")):
        xin[:, K] = fill
        h = torch.full((B, 1024), float("nan"), device=DEV)
        _lib.check(lib.gdmcf_linear_fwd_f32(xin.data_ptr(), ldk, W1.data_ptr(), K, b1.data_ptr(), 1, B, H, K, h.data_ptr(), 1024,
                                            ws.data_ptr(), ws_bytes, st))
        torch.cuda.synchronize()
        hs.append(h[:, :H].clone())
    assert bool(torch.isfinite(hs[0]).all()) and torch.equal(hs[0], hs[1]) and torch.equal(hs[0], hs[2])


def test_reverse_loop_through_cached_transposed_weights_follows_weight_updates():
    """engine._transposed: the reverse-diffusion loop (reference gaussian_diffusion.py:161-220) runs its hidden layer through a
    cached W^T (gdmcf_linear_fwd_wt_f32 -> dr_kn_kernel).  Yelp shape: predictions with the cache equal the plain path within
    float32 summation order and give the same top-k lists (up to ties inside that noise); after a training step the cache must follow the new weights (version
    check) -- compared with a model that never uses it."""
    import scipy.sparse as sp
    from gdmcf_amd.data_utils import DeviceCSR
    from gdmcf_amd.parallel import DataParallelStep
    dev = torch.device(DEV)
    B, hid, T = 400, 1000, 5
    indptr, indices, I = D.synth_csr("yelp", n_rows=B, seed=0)
    dcsr = DeviceCSR(sp.csr_matrix((np.ones(len(indices), np.float32), indices, indptr), shape=(B, I)), dev)
    x = torch.zeros(B, I, device=dev)
    x[torch.as_tensor(np.repeat(np.arange(B), np.diff(indptr)), device=dev), torch.as_tensor(indices.astype(np.int64), device=dev)] = 1.0
    torch.manual_seed(0)
    model = gdmcf_amd.DNN([I, hid], [hid, I], 10, time_type="cat", norm=False).to(dev)
    diffusion = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, dev)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-3)
    step = DataParallelStep(diffusion, model, opt)
    noise0 = torch.randn(B, I, generator=torch.Generator().manual_seed(3)).to(dev)  # the same x_T for every prediction

    def predict(cached):
        model.eval()
        model.engine._wt_on = cached
        with torch.no_grad():
            p = diffusion.p_sample(model, x, T, False, noise0=noise0)
        torch.cuda.synchronize()
        model.train()
        return p

    for round_ in range(2):
        a, b = predict(True), predict(False)
        assert model.engine._wt, "the transposed cache was not used"
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), round_
        # the same top-20 lists up to ties inside the two paths' summation-order noise: every item one path picks scores, in the
        # OTHER path, no lower than that path's own 20th score minus the noise bound (index-for-index equality would test luck:
        # scores 1e-9 apart swap places between two float32 summation orders)
        tol = 2e-5 * float(b.abs().max())
        ia, ib = a.topk(20, dim=1).indices, b.topk(20, dim=1).indices
        assert bool((b.gather(1, ia) >= b.topk(20, dim=1).values[:, -1:] - tol).all()), round_
        assert bool((a.gather(1, ib) >= a.topk(20, dim=1).values[:, -1:] - tol).all()), round_
        assert float((ia == ib).float().mean()) > 0.98, round_
        torch.manual_seed(5)
        step(dcsr.batch(torch.arange(B, device=dev)), True)  # the weights move: the next prediction must see them
        c = predict(True)
        assert float((c - a).abs().max()) > 0.0
