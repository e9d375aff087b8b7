"""GPU (-m gpu): parity of the HIP hot path (through gdmcf_amd -> C ABI) with
  * the committed golden fixtures (outputs of the real reference), and
  * the CPU oracle on the same seeded inputs.
Tolerances: training loss <= 1e-4 relative (north_star); q_sample / history bookkeeping / top-k
index sets bit-exact; everything else fp32 summation-order noise (stated per test)."""
import copy

import numpy as np
import pytest
import torch

import gdmcf_amd
from gdmcf_amd import ModelMeanType
from oracle import gdmcf_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gpu_model(meta, fx):
    I, dims = meta["I"], meta["dims"]
    m = gdmcf_amd.DNN([I] + dims, dims[::-1] + [I], meta.get("emb", 10), time_type="cat", norm=meta.get("norm", False))
    m.load_state_dict(H.state_dict_from(fx))
    return m.to(DEV)


def gpu_diffusion(meta):
    mt = {"x0": ModelMeanType.START_X, "eps": ModelMeanType.EPSILON}[meta["mean_type"]]
    return gdmcf_amd.GaussianDiffusion(mt, meta.get("schedule", "linear-var"), meta["scale"], meta["nmin"],
                                       meta["nmax"], meta["T"], DEV).to(DEV)


def cu(t):
    return t.to(DEV)


def test_native_library_is_loaded():
    import ctypes  # noqa: F401
    from gdmcf_amd import _lib
    lib = _lib.load()
    ncu, wave = ctypes.c_int(), ctypes.c_int()
    arch = ctypes.create_string_buffer(64)
    _lib.check(lib.gdmcf_device_info(ctypes.byref(ncu), ctypes.byref(wave), arch, 64))
    assert wave.value == 64 and ncu.value >= 64
    assert arch.value.decode().startswith("gfx950"), arch.value


@pytest.mark.parametrize("case", H.TRAIN_CASES)
def test_q_sample_bit_exact(case):
    fx = H.load("train_" + case)
    meta = H.train_meta(fx)
    diff = gpu_diffusion(meta)
    inp = H.step_inputs(fx, 0)
    x_t = diff.q_sample(cu(inp["x"]), cu(inp["ts"]), cu(inp["noise"]))
    np.testing.assert_array_equal(x_t.cpu().numpy(), fx["s0.x_t"])


@pytest.mark.parametrize("case", H.TRAIN_CASES)
def test_train_steps_match_reference(case):
    """zero_grad -> training_losses -> mean -> backward -> AdamW.step, injected randomness."""
    fx = H.load("train_" + case)
    meta = H.train_meta(fx)
    model = gpu_model(meta, fx)
    diff = gpu_diffusion(meta)
    diff.Lt_history.copy_(torch.from_numpy(fx["Lt_history0"]))
    diff.Lt_count.copy_(torch.from_numpy(fx["Lt_count0"]))
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=meta["lr"], weight_decay=meta["wd"])
    model.train()
    for s in range(meta["n_steps"]):
        inp = H.step_inputs(fx, s)
        if f"s{s}.p_all" in fx:
            np.testing.assert_allclose(diff.importance_probs().cpu().numpy(), fx[f"s{s}.p_all"], rtol=1e-5)
        opt.zero_grad()
        terms = diff.training_losses(model, cu(inp["x"]), True, ts=cu(inp["ts"]), pt=cu(inp["pt"]),
                                     noise=cu(inp["noise"]), drop_mask=cu(inp["drop_mask"]))
        assert terms["loss"].dtype == torch.float64 and terms["loss"].shape == (meta["B"],)
        loss = terms["loss"].mean()
        loss.backward()
        lv = terms["loss"].detach().cpu().numpy()
        # north_star: training loss within 1e-4 relative (observed ~1e-6: fp32 summation order only)
        np.testing.assert_allclose(lv, fx[f"s{s}.loss_vec"], rtol=1e-4, atol=0)
        assert abs(float(loss) - float(fx[f"s{s}.loss"])) <= 1e-4 * abs(float(fx[f"s{s}.loss"]))
        if s == 0:
            for k, v in model.named_parameters():
                assert H.relerr(v.grad.cpu().numpy(), fx["g0." + k]) < 2e-4, k
        opt.step()
        np.testing.assert_array_equal(diff.Lt_count.cpu().numpy(), fx[f"s{s}.Lt_count"])
        np.testing.assert_allclose(diff.Lt_history.cpu().numpy(), fx[f"s{s}.Lt_history"], rtol=1e-4, atol=0)
    for k, v in model.named_parameters():
        # AdamW's first steps move each weight by ~lr whatever the gradient: compare on the lr scale
        d = np.abs(v.detach().cpu().numpy() - fx["pN." + k]).max()
        assert d < 0.02 * meta["lr"] * meta["n_steps"], (k, d)
        assert H.relerr(opt.state[v]["exp_avg"].cpu().numpy(), fx["m." + k]) < 2e-4, k
        assert H.relerr(opt.state[v]["exp_avg_sq"].cpu().numpy(), fx["v." + k]) < 4e-4, k


@pytest.mark.parametrize("case", H.TRAIN_CASES)
def test_bf16_mode_on_every_reference_configuration(case):
    """The bf16 GEMM mode (shadows included) on every fixture configuration of the f32 parity suite -- eps / x0 mean
    type, weight decay, T=40 importance sampling, two hidden layers, F.normalize -- against the reference's own
    numbers with bf16-rounding tolerances: loss 3e-3 (mean) / 2e-2 (per row), first-step gradients 5e-2 relative L2,
    and the bookkeeping that does not depend on the GEMM precision (Lt_count) exact."""
    fx = H.load("train_" + case)
    meta = H.train_meta(fx)
    model = gpu_model(meta, fx)
    model.gemm_dtype = "bf16"
    diff = gpu_diffusion(meta)
    diff.Lt_history.copy_(torch.from_numpy(fx["Lt_history0"]))
    diff.Lt_count.copy_(torch.from_numpy(fx["Lt_count0"]))
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=meta["lr"], weight_decay=meta["wd"])
    model.train()
    for s in range(meta["n_steps"]):
        inp = H.step_inputs(fx, s)
        opt.zero_grad()
        terms = diff.training_losses(model, cu(inp["x"]), True, ts=cu(inp["ts"]), pt=cu(inp["pt"]),
                                     noise=cu(inp["noise"]), drop_mask=cu(inp["drop_mask"]))
        loss = terms["loss"].mean()
        loss.backward()
        np.testing.assert_allclose(terms["loss"].detach().cpu().numpy(), fx[f"s{s}.loss_vec"], rtol=2e-2, atol=0)
        assert abs(float(loss) - float(fx[f"s{s}.loss"])) <= 3e-3 * abs(float(fx[f"s{s}.loss"]))
        if s == 0:
            for k, v in model.named_parameters():
                assert H.relerr(v.grad.cpu().numpy(), fx["g0." + k]) < 5e-2, k
        opt.step()
        np.testing.assert_array_equal(diff.Lt_count.cpu().numpy(), fx[f"s{s}.Lt_count"])
    assert model.engine.buffers(meta["B"], torch.device(DEV)).shadows is not None  # the shadow path is what ran


@pytest.mark.parametrize("case", ["tiny_x0", "tiny_eps"])
def test_per_step_methods_match_oracle(case):
    """The reference's per-step methods (p_mean_variance :473-515, q_posterior_mean_variance :451-471,
    _predict_xstart_from_eps :518-523) exist with the reference's names and results; p_sample fuses them, these
    are for callers that use them directly."""
    fx = H.load("train_" + case)
    meta = H.train_meta(fx)
    model = gpu_model(meta, fx).eval()
    diff = gpu_diffusion(meta)
    om = O.DNN([meta["I"]] + meta["dims"], meta["dims"][::-1] + [meta["I"]], 10)
    om.load_state_dict(H.state_dict_from(fx))
    om.eval()
    omt = {"x0": O.ModelMeanType.START_X, "eps": O.ModelMeanType.EPSILON}[meta["mean_type"]]
    od = O.GaussianDiffusion(omt, meta.get("schedule", "linear-var"), meta["scale"], meta["nmin"], meta["nmax"], meta["T"])
    inp = H.step_inputs(fx, 0)
    x, t = inp["x"] + 0.1 * inp["noise"], inp["ts"]
    with torch.no_grad():
        got = diff.p_mean_variance(model, cu(x), cu(t))
        want = od.p_mean_variance(om, x, t)
    for k in ("mean", "variance", "log_variance", "pred_xstart"):
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k].numpy(), rtol=2e-5, atol=1e-6, err_msg=k)
    m2 = diff.q_posterior_mean_variance(cu(inp["x"]), cu(x), cu(t))
    w2 = od.q_posterior_mean_variance(inp["x"], x, t)
    for a, b in zip(m2, w2):
        np.testing.assert_array_equal(a.cpu().numpy(), b.numpy())  # elementwise f32 on the same tables
    np.testing.assert_array_equal(diff._predict_xstart_from_eps(cu(x), cu(t), cu(inp["noise"])).cpu().numpy(),
                                  od._predict_xstart_from_eps(x, t, inp["noise"]).numpy())
    with pytest.raises(AssertionError):
        diff.p_mean_variance(model, cu(x), cu(t[:-1]))


@pytest.mark.parametrize("T,B,I,hid", [(2, 5, 33, 8), (2, 3, 17, 4), (5, 1, 130, 16), (3, 7, 1, 4)])
def test_degenerate_sizes_match_oracle(T, B, I, hid):
    """Edges of the size space against the oracle: two diffusion steps (one is refused on both sides: the reference's
    table code indexes entry 1), one user per batch, a single item (its products take the element-wise kernel of
    gemm_small.hip), hidden widths below one MFMA block, odd everything -- training step and reverse loop."""
    with pytest.raises((IndexError, AssertionError)):
        O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 1)
    with pytest.raises((IndexError, AssertionError)):
        gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 1, DEV)
    torch.manual_seed(T * 100 + B)
    om = O.DNN([I, hid], [hid, I], 10)
    gm = gdmcf_amd.DNN([I, hid], [hid, I], 10)
    gm.load_state_dict(om.state_dict())
    gm = gm.to(DEV)
    od = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T)
    gd_ = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV)
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(B, I, generator=g) < 0.4).float()
    ts = torch.randint(0, T, (B,), generator=g)
    noise = torch.randn(B, I, generator=g)
    keep = (torch.rand(B, I, generator=g) < 0.5).float()
    oopt, gopt = O.make_optimizer(om, 1e-3), gdmcf_amd.FusedAdamW(gm.parameters(), lr=1e-3, weight_decay=0.0)
    om.train(), gm.train()
    oloss, ovec = O.train_step(od, om, oopt, x, True, ts=ts, pt=torch.ones(B), noise=noise, drop_mask=keep)
    gopt.zero_grad()
    terms = gd_.training_losses(gm, cu(x), True, ts=cu(ts), pt=cu(torch.ones(B)), noise=cu(noise), drop_mask=cu(keep))
    terms["loss"].mean().backward()
    gopt.step()
    np.testing.assert_allclose(terms["loss"].detach().cpu().numpy(), ovec.numpy(), rtol=1e-5)
    for p, q in zip(gm.parameters(), om.parameters()):
        assert H.relerr(p.grad.cpu().numpy(), q.grad.numpy()) < 1e-4
    np.testing.assert_array_equal(gd_.Lt_count.cpu().numpy(), od.Lt_count.numpy())
    om.eval(), gm.eval()
    with torch.no_grad():
        want = od.p_sample(om, x, 0, False)
        got = gd_.p_sample(gm, cu(x), 0, False)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-6)
    k = min(3, I)
    rows, cols = x.nonzero(as_tuple=True)
    csr = x.to_sparse_csr()
    idx = gdmcf_amd.masked_topk(got, k, csr.crow_indices().to(DEV), csr.col_indices().to(DEV))
    assert idx.shape == (B, k)


@pytest.mark.parametrize("emb", [7, 3, 16])
def test_odd_embedding_widths_match_oracle(emb):
    """timestep_embedding zero-pads an odd width (reference models/DNN.py:1823-1824)."""
    B, I, hid, T = 9, 70, 12, 5
    torch.manual_seed(emb)
    om = O.DNN([I, hid], [hid, I], emb)
    gm = gdmcf_amd.DNN([I, hid], [hid, I], emb)
    gm.load_state_dict(om.state_dict())
    gm = gm.to(DEV)
    od = O.GaussianDiffusion(O.ModelMeanType.EPSILON, "linear-var", 0.01, 0.001, 0.01, T)
    gd_ = gdmcf_amd.GaussianDiffusion(ModelMeanType.EPSILON, "linear-var", 0.01, 0.001, 0.01, T, DEV)
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(B, I, generator=g) < 0.2).float()
    ts = torch.randint(0, T, (B,), generator=g)
    noise, keep = torch.randn(B, I, generator=g), (torch.rand(B, I, generator=g) < 0.5).float()
    np.testing.assert_allclose(gdmcf_amd.timestep_embedding(cu(ts), emb).cpu().numpy(), O.timestep_embedding(ts, emb).numpy(),
                               rtol=1e-6, atol=1e-7)
    om.train(), gm.train()
    oopt = O.make_optimizer(om, 1e-3)
    oloss, ovec = O.train_step(od, om, oopt, x, True, ts=ts, pt=torch.ones(B), noise=noise, drop_mask=keep)
    terms = gd_.training_losses(gm, cu(x), True, ts=cu(ts), pt=cu(torch.ones(B)), noise=cu(noise), drop_mask=cu(keep))
    terms["loss"].mean().backward()
    np.testing.assert_allclose(terms["loss"].detach().cpu().numpy(), ovec.numpy(), rtol=1e-5)
    for (k, p), q in zip(gm.named_parameters(), om.parameters()):
        assert H.relerr(p.grad.cpu().numpy(), q.grad.numpy()) < 1e-4 or float(q.grad.abs().max()) < 1e-12, k


@pytest.mark.parametrize("p", [0.0, 0.25, 0.9])
def test_dropout_rates_match_oracle(p):
    """DNN(dropout=p): survivors are scaled by 1/(1-p) (reference models/DNN.py:37,77); p = 0 keeps everything.  Also
    the in-kernel Philox mask: the kept fraction matches 1-p and kept elements carry exactly the 1/(1-p) scale."""
    B, I, hid, T = 16, 300, 24, 5
    torch.manual_seed(5)
    om = O.DNN([I, hid], [hid, I], 10, dropout=p)
    gm = gdmcf_amd.DNN([I, hid], [hid, I], 10, dropout=p)
    gm.load_state_dict(om.state_dict())
    gm = gm.to(DEV)
    od = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T)
    gd_ = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV)
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(B, I, generator=g) < 0.1).float()
    ts = torch.randint(0, T, (B,), generator=g)
    noise = torch.randn(B, I, generator=g)
    keep = (torch.rand(B, I, generator=g) < (1 - p)).float()
    om.train(), gm.train()
    oloss, ovec = O.train_step(od, om, O.make_optimizer(om, 1e-3), x, True, ts=ts, pt=torch.ones(B), noise=noise, drop_mask=keep)
    terms = gd_.training_losses(gm, cu(x), True, ts=cu(ts), pt=cu(torch.ones(B)), noise=cu(noise), drop_mask=cu(keep))
    terms["loss"].mean().backward()
    np.testing.assert_allclose(terms["loss"].detach().cpu().numpy(), ovec.numpy(), rtol=1e-5)
    for pg, q in zip(gm.parameters(), om.parameters()):
        assert H.relerr(pg.grad.cpu().numpy(), q.grad.numpy()) < 1e-4
    # Philox path: statistics of the mask
    xs = torch.ones(64, 4096, device=DEV)
    big = gdmcf_amd.DNN([4096, 8], [8, 4096], 10, dropout=p).to(DEV).train()
    big.engine.manual_seed(7)
    big(xs, torch.zeros(64, dtype=torch.int64, device=DEV))
    xin = big.engine.buffers(64, torch.device(DEV)).xin[:, :4096]
    kept = xin != 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 0.01
    assert bool(torch.all(torch.abs(xin[kept] * (1.0 - p) - 1.0) < 1e-6))


def test_unweighted_loss_branch():
    """reweight=False: the reference multiplies unit weights with an undefined `loss` (gaussian_diffusion.py:349-352 --
    it raises); this build and the oracle take DiffRec's reading: unit weights on the mse, for both targets."""
    fx = H.load("train_ragged_x0")
    meta = H.train_meta(fx)
    model, diff = gpu_model(meta, fx).train(), gpu_diffusion(meta)
    om = H.oracle_model(meta, fx).train()
    od = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", meta["scale"], meta["nmin"], meta["nmax"], meta["T"])
    inp = H.step_inputs(fx, 0)
    want = od.training_losses(om, inp["x"], False, ts=inp["ts"], pt=inp["pt"], noise=inp["noise"], drop_mask=inp["drop_mask"])["loss"]
    got = diff.training_losses(model, cu(inp["x"]), False, ts=cu(inp["ts"]), pt=cu(inp["pt"]), noise=cu(inp["noise"]),
                               drop_mask=cu(inp["drop_mask"]))["loss"]
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5)
    got.mean().backward()
    want.mean().backward()
    for p, q in zip(model.parameters(), om.parameters()):
        assert H.relerr(p.grad.cpu().numpy(), q.grad.numpy()) < 2e-4
    fe = H.load("train_tiny_eps")
    me = H.train_meta(fe)
    ie = H.step_inputs(fe, 0)
    oe = O.GaussianDiffusion(O.ModelMeanType.EPSILON, "linear-var", me["scale"], me["nmin"], me["nmax"], me["T"])
    want = oe.training_losses(H.oracle_model(me, fe).train(), ie["x"], False, ts=ie["ts"], pt=ie["pt"], noise=ie["noise"],
                              drop_mask=ie["drop_mask"])["loss"]
    got = gpu_diffusion(me).training_losses(gpu_model(me, fe).train(), cu(ie["x"]), False, ts=cu(ie["ts"]), pt=cu(ie["pt"]),
                                            noise=cu(ie["noise"]), drop_mask=cu(ie["drop_mask"]))["loss"]
    assert bool((ie["ts"] == 0).any())  # the t == 0 rows are the ones whose treatment differs from the weighted branch
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5)


def test_noise_scale_zero_feeds_the_rows_unnoised():
    """noise_scale == 0: no schedule tables are built and x_t = x_start (reference gaussian_diffusion.py:86-89,
    :299-302); only the unweighted loss is defined then."""
    fx = H.load("train_ragged_x0")
    meta = H.train_meta(fx)
    model, om = gpu_model(meta, fx).train(), H.oracle_model(meta, fx).train()
    od = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.0, meta["nmin"], meta["nmax"], meta["T"])
    gd_ = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.0, meta["nmin"], meta["nmax"], meta["T"], DEV)
    inp = H.step_inputs(fx, 0)
    want = od.training_losses(om, inp["x"], False, ts=inp["ts"], pt=inp["pt"], noise=inp["noise"], drop_mask=inp["drop_mask"])["loss"]
    got = gd_.training_losses(model, cu(inp["x"]), False, ts=cu(inp["ts"]), pt=cu(inp["pt"]), noise=cu(inp["noise"]),
                              drop_mask=cu(inp["drop_mask"]))["loss"]
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5)
    with pytest.raises(AttributeError):  # the SNR weights need tables that were never built -- as in the reference
        gd_.training_losses(model, cu(inp["x"]), True, ts=cu(inp["ts"]), pt=cu(inp["pt"]))


def test_optimizer_state_interchanges_with_torch_adamw():
    """FusedAdamW's state_dict (step / exp_avg / exp_avg_sq per parameter, the reference optimiser's layout, main.py:258)
    loads into torch.optim.AdamW and back; both then take the same next step."""
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(37, 53, device=DEV)), torch.nn.Parameter(torch.randn(11, device=DEV))]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    fo = gdmcf_amd.FusedAdamW(ps, lr=1e-2, weight_decay=0.05)
    g = torch.Generator(device=DEV).manual_seed(1)
    for _ in range(3):
        for p in ps:
            p.grad = torch.randn(p.shape, device=DEV, generator=g)
        fo.step()
    to = torch.optim.AdamW(qs, lr=1e-2, weight_decay=0.05)
    for q, p in zip(qs, ps):
        q.data.copy_(p.data)
    # state_dict() hands out the live moment tensors (torch's does too): copy as a checkpoint round trip would
    to.load_state_dict(copy.deepcopy(fo.state_dict()))  # ours -> torch
    grads = [torch.randn(p.shape, device=DEV, generator=g) for p in ps]
    for p, q, gr in zip(ps, qs, grads):
        p.grad, q.grad = gr.clone(), gr.clone()
    fo.step()
    to.step()
    for p, q in zip(ps, qs):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    fo2 = gdmcf_amd.FusedAdamW(ps, lr=1e-2, weight_decay=0.05)
    fo2.load_state_dict(copy.deepcopy(to.state_dict()))  # torch -> ours (torch keeps `step` as a tensor)
    for p, q, gr in zip(ps, qs, grads):
        p.grad, q.grad = gr.clone(), gr.clone()
    fo2.step()
    to.step()
    for p, q in zip(ps, qs):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
        assert int(fo2.state[p]["step"]) == int(to.state[q]["step"]) == 5


def test_thousand_diffusion_steps_match_oracle():
    """T = 1000 (DDPM-scale): the [T, 10] float64 loss history no longer fits the default 48 KB of LDS -- the FIFO
    kernels ask for up to 150 KB.  Training step, history bookkeeping, importance probabilities vs the oracle."""
    T, B, I, hid = 1000, 64, 200, 32
    torch.manual_seed(1)
    om = O.DNN([I, hid], [hid, I], 10)
    gm = gdmcf_amd.DNN([I, hid], [hid, I], 10)
    gm.load_state_dict(om.state_dict())
    gm = gm.to(DEV)
    od = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.0001, 0.0005, 0.005, T)
    gd_ = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.0001, 0.0005, 0.005, T, DEV)
    g = torch.Generator().manual_seed(2)
    hist0 = torch.rand(T, 10, generator=g, dtype=torch.float64) * 5 + 0.1
    od.Lt_history, od.Lt_count = hist0.clone(), torch.full((T,), 10, dtype=torch.int64)
    gd_.Lt_history.copy_(hist0)
    gd_.Lt_count.fill_(10)
    np.testing.assert_allclose(gd_.importance_probs().cpu().numpy(), od.importance_probs().numpy(), rtol=1e-12)
    om.train(), gm.train()
    oopt, gopt = O.make_optimizer(om, 1e-3), gdmcf_amd.FusedAdamW(gm.parameters(), lr=1e-3, weight_decay=0.0)
    for _ in range(2):
        x = (torch.rand(B, I, generator=g) < 0.1).float()
        ts = torch.randint(0, T, (B,), generator=g)
        ts[:4] = torch.tensor([0, T - 1, 7, 7])  # first / last step and a repeated step in one batch
        noise, keep = torch.randn(B, I, generator=g), (torch.rand(B, I, generator=g) < 0.5).float()
        pt = torch.rand(B, generator=g, dtype=torch.float64) + 0.5
        oloss, ovec = O.train_step(od, om, oopt, x, True, ts=ts, pt=pt, noise=noise, drop_mask=keep)
        gopt.zero_grad()
        terms = gd_.training_losses(gm, cu(x), True, ts=cu(ts), pt=cu(pt), noise=cu(noise), drop_mask=cu(keep))
        terms["loss"].mean().backward()
        gopt.step()
        np.testing.assert_allclose(terms["loss"].detach().cpu().numpy(), ovec.numpy(), rtol=1e-5)
        np.testing.assert_allclose(gd_.Lt_history.cpu().numpy(), od.Lt_history.numpy(), rtol=1e-5)
    t_dev, pt_dev = gd_.sample_timesteps(B, DEV, "importance")
    assert int(t_dev.min()) >= 0 and int(t_dev.max()) < T and bool((pt_dev > 0).all())


def test_plain_forward_backward_matches_oracle():
    """model(x, t) + autograd through the HIP kernels vs the oracle's eager autograd."""
    fx = H.load("train_ragged_x0")
    meta = H.train_meta(fx)
    model, om = gpu_model(meta, fx), H.oracle_model(meta, fx)
    inp = H.step_inputs(fx, 0)
    x = torch.from_numpy(fx["s0.x_t"])
    g = torch.Generator().manual_seed(3)
    w = torch.randn(meta["B"], meta["I"], generator=g)
    model.train(), om.train()
    out = model(cu(x), cu(inp["ts"]), drop_mask=cu(inp["drop_mask"]))
    ref = om(x, inp["ts"], inp["drop_mask"])
    assert H.relerr(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    np.testing.assert_allclose(out.detach().cpu().numpy(), fx["s0.model_output"], rtol=0, atol=2e-5)
    (out * cu(w)).sum().backward()
    (ref * w).sum().backward()
    for (k, a), (_, b) in zip(model.named_parameters(), om.named_parameters()):
        assert H.relerr(a.grad.cpu().numpy(), b.grad.numpy()) < 1e-4, k
    model.eval(), om.eval()
    with torch.no_grad():
        assert H.relerr(model(cu(x), cu(inp["ts"])).cpu().numpy(), om(x, inp["ts"]).numpy()) < 1e-5


def test_backward_after_second_forward_is_refused():
    fx = H.load("train_tiny_x0")
    meta = H.train_meta(fx)
    model, diff = gpu_model(meta, fx), gpu_diffusion(meta)
    x = cu(H.step_inputs(fx, 0)["x"])
    a = diff.training_losses(model, x, True)["loss"].mean()
    diff.training_losses(model, x, True)
    with pytest.raises(RuntimeError, match="overwritten"):
        a.backward()


@pytest.mark.parametrize("case", H.SAMPLE_CASES)
def test_p_sample_topk_and_metrics_match_reference(case):
    fx = H.load("sample_" + case)
    meta = H.sample_meta(fx)
    model = gpu_model(meta, fx).eval()
    diff = gpu_diffusion(meta)
    x = cu(torch.from_numpy(fx["x_start"].astype(np.float32)))
    T, k = meta["T"], meta["k"]
    cap = {}
    p0 = diff.p_sample(model, x, 0, False, capture=cap)
    scale = np.abs(fx["pred_steps0"]).max()
    assert np.abs(p0.cpu().numpy() - fx["pred_steps0"]).max() < 2e-5 * max(scale, 1.0)
    for n in range(T):
        assert H.relerr(cap["pred_xstart"][n].cpu().numpy(), fx["step_pred_xstart"][n]) < 2e-5
        assert H.relerr(cap["mean"][n].cpu().numpy(), fx["step_mean"][n]) < 2e-5
    pT = diff.p_sample(model, x, T, False, noise0=cu(torch.from_numpy(fx["noise_stepsT"])))
    assert H.relerr(pT.cpu().numpy(), fx["pred_stepsT"]) < 2e-5
    pn = diff.p_sample(model, x, 2, True, noise0=cu(torch.from_numpy(fx["noise_noisy0"])),
                       step_noise=cu(torch.from_numpy(fx["noise_noisy_steps"])))
    assert H.relerr(pn.cpu().numpy(), fx["pred_noisy"]) < 2e-5
    with pytest.raises(AssertionError):
        diff.p_sample(model, x, T + 1, False)

    # history mask + top-k on the device prediction: bit-exact index SETS (north_star); the ordered
    # list is compared wherever neighbouring scores are further apart than fp32 noise.
    his = torch.from_numpy(fx["x_start"].astype(np.float32)).to_sparse_csr()
    idx = gdmcf_amd.masked_topk(p0, k, his.crow_indices(), his.col_indices()).cpu().numpy()
    tol = 1e-4 * max(scale, 1.0)
    for b in range(meta["B"]):
        if fx["topk_gap"][b] > tol:
            assert set(idx[b].tolist()) == set(fx["topk_idx"][b].tolist()), b
        if fx["topk_min_adjacent_gap"][b] > tol:
            np.testing.assert_array_equal(idx[b], fx["topk_idx"][b])
    assert (fx["topk_gap"] > tol).mean() > 0.9  # the fixture really exercises the comparison
    # the same kernel on the reference's own scores must reproduce its lists exactly
    idx_ref = gdmcf_amd.masked_topk(cu(torch.from_numpy(fx["pred_steps0"])), k, his.crow_indices(), his.col_indices())
    np.testing.assert_array_equal(idx_ref.cpu().numpy(), fx["topk_idx"])
    gt = [fx["gt_flat"][a:b].tolist() for a, b in zip(fx["gt_ptr"][:-1], fx["gt_ptr"][1:])]
    res = gdmcf_amd.computeTopNAccuracy(gt, idx_ref.cpu().tolist(), fx["topN"].tolist())
    np.testing.assert_array_equal(np.array(res, dtype=np.float64), fx["metrics"])  # Recall@k parity


def test_topk_edge_cases_against_oracle():
    g = torch.Generator().manual_seed(0)
    B, I = 37, 1003
    pred = torch.randn(B, I, generator=g)
    pred[:, ::7] = pred[:, 1::7][:, : pred[:, ::7].shape[1]]  # many exact ties
    pred[3] = 0.25  # a constant row: pure index order
    pred[4, 10:20] = float("-inf")
    mask = (torch.rand(B, I, generator=g) < 0.3)
    mask[5] = True  # everything masked: all -inf, index order
    mask[6] = False
    rows, cols = mask.nonzero(as_tuple=True)
    csr = mask.float().to_sparse_csr()
    for k in (1, 2, 100, 128, 129, I):
        ref = O.masked_topk(pred, rows, cols, k)
        got = gdmcf_amd.masked_topk(cu(pred), k, csr.crow_indices(), csr.col_indices())
        np.testing.assert_array_equal(got.cpu().numpy(), ref.numpy(), err_msg=f"k={k}")
    val, idx = gdmcf_amd.masked_topk(cu(pred), 5, return_values=True)
    tv, ti = torch.topk(pred, 5)
    np.testing.assert_array_equal(val.cpu().numpy(), tv.numpy())
    with pytest.raises(AssertionError):
        gdmcf_amd.masked_topk(cu(pred), I + 1)


def test_topk_wide_rows_take_the_unstaged_path():
    """Rows wider than the LDS staging limit (~37 k items; the Amazon-Book shape has 94 949): the keys are recomputed
    from global memory in every radix pass.  Same lists as the oracle, ties included."""
    g = torch.Generator().manual_seed(4)
    B, I = 6, 60011
    pred = torch.randn(B, I, generator=g)
    n5 = pred[:, 1::5].shape[1]
    pred[:, 0:5 * n5:5] = pred[:, 1::5]  # exact ties
    pred[2] = -1.5  # constant row: pure index order
    mask = (torch.rand(B, I, generator=g) < 0.01)
    rows, cols = mask.nonzero(as_tuple=True)
    csr = mask.float().to_sparse_csr()
    for k in (1, 100, 257):
        ref = O.masked_topk(pred, rows, cols, k)
        got = gdmcf_amd.masked_topk(cu(pred), k, csr.crow_indices().to(DEV), csr.col_indices().to(DEV))
        np.testing.assert_array_equal(got.cpu().numpy(), ref.numpy(), err_msg=f"k={k}")


@pytest.mark.parametrize("gen", ["1", "3"])
@pytest.mark.parametrize("case", ["small", "mid"])
def test_lightgcn_propagation_matches_reference(case, gen, monkeypatch):
    monkeypatch.setenv("GDMCF_SPMM_GEN", gen)
    fx = H.load("lightgcn_" + case)
    U, It, d, L = [int(v) for v in str(fx["meta"][0]).split("|")]
    data = {"user_id_idx": fx["users"], "item_id_idx": fx["items"]}
    m = gdmcf_amd.LightGCN(data, U, It, L, d, device=DEV)
    with torch.no_grad():
        m.E0.weight.copy_(torch.from_numpy(fx["E0"]))
    m = m.to(DEV)
    with torch.no_grad():
        fu, fi, iu, ii, layers = m.propagate_through_layers(return_layers=True)
    for l in range(L):
        np.testing.assert_allclose(layers[l].cpu().numpy(), fx["layers"][l], rtol=0, atol=3e-7)
    np.testing.assert_allclose(fu.cpu().numpy(), fx["final_user"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(fi.cpu().numpy(), fx["final_item"], rtol=0, atol=3e-7)
    np.testing.assert_array_equal(iu.cpu().numpy(), fx["E0"][:U])
    with torch.no_grad():
        fu2, fi2, _, _ = m.propagate_through_layers()  # production path: layer mean fused into the last SpMM
    np.testing.assert_allclose(fu2.cpu().numpy(), fx["final_user"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(fi2.cpu().numpy(), fx["final_item"], rtol=0, atol=3e-7)


@pytest.mark.parametrize("gen", ["1", "2", "3"])
def test_spmm_hub_rows_are_split_and_exact(gen, monkeypatch):
    """A hub item with thousands of neighbours goes through the split + combine path of every kernel generation
    (1: virtual rows, 2: bundled schedule, 3: streamed schedule -- gdmcf_amd/lightgcn.py picks by table size)."""
    monkeypatch.setenv("GDMCF_SPMM_GEN", gen)
    rng = np.random.default_rng(0)
    U, It, d = 3000, 50, 64
    users = np.concatenate([np.arange(U), rng.integers(0, U, 500)])
    items = np.concatenate([np.zeros(U, dtype=np.int64), rng.integers(1, It, 500)])  # item 0 has 3000 neighbours
    A = O.lightgcn_norm_adj(users, items, U, It)
    E0 = rng.standard_normal((U + It, d)).astype(np.float32)
    ref = O.lightgcn_propagate(A, E0, 3, U)
    m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": items}, U, It, 3, d, device=DEV)
    cut_rows = m._plan["crow"] if "crow" in m._plan else m._plan["lrow"]  # rows cut into several pieces
    assert cut_rows.numel() >= 1 and m._plan["n_slots"] >= 12
    with torch.no_grad():
        m.E0.weight.copy_(torch.from_numpy(E0))
    m = m.to(DEV)
    with torch.no_grad():
        fu, fi, _, _ = m.propagate_through_layers()
    # a 3000-term fp32 sum is order sensitive: judge both against the float64 evaluation of the SAME
    # float32 adjacency.  The oracle (sequential fp32) is within 3e-5 of it, the kernel (tree order) within 5e-6.
    A64, E64 = A.astype(np.float64), E0.astype(np.float64)
    acc, cur = E64.copy(), E64
    for _ in range(3):
        cur = A64 @ cur
        acc = acc + cur
    mean64 = acc / 4
    np.testing.assert_allclose(ref[1], mean64[U:], rtol=0, atol=3e-5)
    np.testing.assert_allclose(fu.cpu().numpy(), mean64[:U], rtol=0, atol=5e-6)
    np.testing.assert_allclose(fi.cpu().numpy(), mean64[U:], rtol=0, atol=5e-6)
    with torch.no_grad():
        fu2, fi2, _, _ = m.propagate_through_layers()
    assert torch.equal(fi, fi2) and torch.equal(fu, fu2)  # deterministic (no atomics)


@pytest.mark.parametrize("gen", ["1", "2", "3"])
def test_spmm_isolated_nodes_and_non_finite_strangers(gen, monkeypatch):
    """Nodes without any edge keep mean_l(A~^l E0) = E0 / (L+1) (only the l = 0 term), and a non-finite embedding reaches
    exactly the nodes the reference's sparse product lets it reach: its neighbourhood -- not rows that merely share a wave,
    a bundle or a padded step with it (the schedules pad with weight-0 entries; 0 * inf must never be formed with a
    stranger's row)."""
    monkeypatch.setenv("GDMCF_SPMM_GEN", gen)
    rng = np.random.default_rng(5)
    U, It, d, L = 400, 300, 64, 2
    users = rng.integers(0, U - 40, 5000)  # the last 40 users and ...
    items = rng.integers(10, It, 5000)  # ... the first 10 items have no edge at all
    A = O.lightgcn_norm_adj(users, items, U, It)
    E0 = rng.standard_normal((U + It, d)).astype(np.float32)
    E0[U + 3] = np.inf  # an isolated item (column 0..9 of the item block): nobody gathers it
    E0[0, 5] = np.nan  # user 0: reaches its items after one layer, their users after two
    m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": items}, U, It, L, d, device=DEV)
    with torch.no_grad():
        m.E0.weight.copy_(torch.from_numpy(E0))
    m = m.to(DEV)
    with torch.no_grad():
        fu, fi, _, _ = m.propagate_through_layers()
    got = np.concatenate([fu.cpu().numpy(), fi.cpu().numpy()])
    with np.errstate(invalid="ignore"):
        A64 = A.astype(np.float64)
        cur, acc = E0.astype(np.float64), E0.astype(np.float64)
        for _ in range(L):
            nxt = np.zeros_like(cur)  # sparse product: only stored entries multiply (0-weight structural zeros do not exist)
            for r in range(A64.shape[0]):
                lo, hi = A64.indptr[r], A64.indptr[r + 1]
                if hi > lo:
                    nxt[r] = (A64.data[lo:hi, None] * cur[A64.indices[lo:hi]]).sum(0)
            cur = nxt
            acc = acc + cur
        ref = acc / (L + 1)
    bad_ref, bad_got = ~np.isfinite(ref), ~np.isfinite(got)
    np.testing.assert_array_equal(bad_got, bad_ref)
    np.testing.assert_allclose(got[~bad_ref], ref[~bad_ref], rtol=0, atol=2e-6)
    iso = np.concatenate([np.arange(U - 40, U), U + np.arange(10)])
    iso = iso[iso != U + 3]
    np.testing.assert_array_equal(got[iso], (E0[iso] * np.float32(1.0 / (L + 1))))


@pytest.mark.parametrize("gen", ["1", "3"])
def test_spmm_random_graphs_against_float64(gen, monkeypatch):
    """Random bipartite graphs with awkward degree patterns (hub users AND hub items, isolated nodes on both sides, a
    single edge, more items than users) through LightGCN.propagate_through_layers on both kernel generations, against the
    float64 evaluation of the same float32 adjacency."""
    monkeypatch.setenv("GDMCF_SPMM_GEN", gen)
    rng = np.random.default_rng(int(gen))
    for trial in range(12):
        U, It = int(rng.integers(2, 400)), int(rng.integers(2, 600))
        d = int(rng.choice([8, 64, 128]))
        nnz = int(rng.integers(1, 6000))
        users = np.minimum(rng.zipf(1.4, nnz) - 1, U - 1) if trial % 2 else rng.integers(0, U, nnz)
        items = np.minimum(rng.zipf(1.2, nnz) - 1, It - 1) if trial % 3 else rng.integers(0, It, nnz)
        if trial == 0:
            users, items = np.array([0]), np.array([0])
        L = int(rng.integers(1, 4))
        A = O.lightgcn_norm_adj(users, items, U, It)
        E0 = rng.standard_normal((U + It, d)).astype(np.float32)
        m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": items}, U, It, L, d, device=DEV)
        with torch.no_grad():
            m.E0.weight.copy_(torch.from_numpy(E0))
        m = m.to(DEV)
        with torch.no_grad():
            fu, fi, _, _ = m.propagate_through_layers()
        A64 = A.astype(np.float64)
        cur = acc = E0.astype(np.float64)
        for _ in range(L):
            cur = A64 @ cur
            acc = acc + cur
        ref = acc / (L + 1)
        got = np.concatenate([fu.cpu().numpy(), fi.cpu().numpy()])
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=f"trial {trial}")


def test_lt_history_kernel_matches_serial_fifo():
    """The parallel rank-based FIFO update == the reference's row-by-row loop (gaussian_diffusion.py:355-368)."""
    from gdmcf_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    for T, H, B in [(3, 4, 50), (5, 10, 400), (40, 10, 16), (7, 3, 1), (2, 10, 257)]:
        od = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear", 0.1, 0.001, 0.01, T, history_num_per_term=H)
        hist = torch.zeros(T, H, dtype=torch.float64, device=DEV)
        cnt = torch.zeros(T, dtype=torch.int64, device=DEV)
        for rounds in range(4):
            ts = torch.randint(0, T, (B,), generator=g)
            if rounds == 1:
                ts[:] = ts[0]  # every row on one timestep: deep overflow
            lu = torch.rand(B, generator=g, dtype=torch.float64)
            od.update_history(ts, lu)
            tsd, lud = ts.to(DEV), lu.to(DEV)
            _lib.check(lib.gdmcf_lt_history_update(tsd.data_ptr(), lud.data_ptr(), B, T, H, hist.data_ptr(),
                                                   cnt.data_ptr(), _lib.stream_ptr()))
            np.testing.assert_array_equal(cnt.cpu().numpy(), od.Lt_count.numpy())
            live = (torch.arange(H)[None, :] < od.Lt_count[:, None]).numpy()
            np.testing.assert_array_equal(hist.cpu().numpy()[live], od.Lt_history.numpy()[live])


def test_sample_timesteps_kernel():
    fx = H.load("train_imp_T40")
    meta = H.train_meta(fx)
    d = gpu_diffusion(meta)
    B = 4096
    t, pt = d.sample_timesteps(B, DEV, "importance")  # history empty -> uniform branch, pt == 1
    assert t.dtype == torch.int64 and int(t.min()) >= 0 and int(t.max()) < meta["T"]
    assert torch.equal(pt, torch.ones(B, dtype=torch.float64, device=DEV))
    cnts = torch.bincount(t, minlength=meta["T"]).float() / B
    assert float((cnts - 1.0 / meta["T"]).abs().max()) < 0.02
    d.Lt_history.copy_(torch.from_numpy(fx["Lt_history0"]))
    d.Lt_count.copy_(torch.from_numpy(fx["Lt_count0"]))
    from gdmcf_amd import _lib
    p_out = torch.zeros(meta["T"], dtype=torch.float64, device=DEV)
    ts = torch.empty(B, dtype=torch.int64, device=DEV)
    ptv = torch.empty(B, dtype=torch.float64, device=DEV)
    _lib.check(_lib.load().gdmcf_sample_timesteps(d.Lt_history.data_ptr(), d.Lt_count.data_ptr(), meta["T"], 10, B, 0.001,
                                                  7, 1, ts.data_ptr(), ptv.data_ptr(), p_out.data_ptr(),
                                                  _lib.stream_ptr()))
    np.testing.assert_allclose(p_out.cpu().numpy(), fx["s0.p_all"], rtol=1e-12)  # the reference's p vector
    np.testing.assert_allclose(ptv.cpu().numpy(), (p_out[ts] * meta["T"]).cpu().numpy(), rtol=0, atol=0)
    emp = torch.bincount(ts, minlength=meta["T"]).double().cpu().numpy() / B
    assert np.abs(emp - fx["s0.p_all"]).max() < 0.02
    t2, _ = d.sample_timesteps(B, DEV, "importance")
    t3, _ = d.sample_timesteps(B, DEV, "importance")
    assert not torch.equal(t2, t3)


@pytest.mark.parametrize("gen", ["1", "2", "3"])
@pytest.mark.parametrize("d", [8, 16, 32, 64, 128, 256, 20, 7])
def test_spmm_widths_against_oracle(d, gen, monkeypatch):
    monkeypatch.setenv("GDMCF_SPMM_GEN", gen)
    rng = np.random.default_rng(d)
    U, It, nnz = 300, 200, 4000
    users, items = rng.integers(0, U, nnz), rng.integers(0, It, nnz)
    A = O.lightgcn_norm_adj(users, items, U, It)
    E0 = rng.standard_normal((U + It, d)).astype(np.float32)
    ref = O.lightgcn_propagate(A, E0, 2, U)
    m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": items}, U, It, 2, d, device=DEV)
    with torch.no_grad():
        m.E0.weight.copy_(torch.from_numpy(E0))
    m = m.to(DEV)
    with torch.no_grad():
        fu, fi, _, _ = m.propagate_through_layers()
    np.testing.assert_allclose(fu.cpu().numpy(), ref[0], rtol=0, atol=2e-6)
    np.testing.assert_allclose(fi.cpu().numpy(), ref[1], rtol=0, atol=2e-6)


def test_philox_noise_and_dropout_statistics():
    torch.manual_seed(0)
    B, I = 64, 4099
    m = gdmcf_amd.DNN([I, 32], [32, I], 10).to(DEV)
    d = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear", 1.0, 0.5, 0.5, 4, DEV)
    x0 = torch.zeros(B, I, device=DEV)
    ts = torch.zeros(B, dtype=torch.long, device=DEV)
    z = d.q_sample(x0, ts) / d._t32["sqrt_1mab"][0]  # pure noise
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    assert abs(float((z ** 4).mean()) - 3.0) < 0.15  # gaussian kurtosis
    z2 = d.q_sample(x0, ts) / d._t32["sqrt_1mab"][0]
    assert not torch.equal(z, z2)  # a fresh stream per call
    # dropout keep fraction and scaling through the engine's input builder
    eng = m.engine
    bufs = eng.buffers(B, torch.device(DEV))
    ones = torch.ones(B, I, device=DEV)
    eng.manual_seed(123)
    eng._prep(bufs, ones, ts, None, None, None, None, True)
    a = bufs.xin[:, :I].clone()
    assert set(torch.unique(a).tolist()) == {0.0, 2.0}
    assert abs(float((a > 0).float().mean()) - 0.5) < 0.01
    eng.manual_seed(123)
    eng._prep(bufs, ones, ts, None, None, None, None, True)
    assert torch.equal(a, bufs.xin[:, :I])  # counter based: same (seed, offset) -> same mask
    # p = 0.3: keep fraction 0.7 (16-bit thresholds), exact scale 1/(1-p); the two column groups that share a Philox block
    # (columns i and i + 1024: low / high halves of the same words) are independent
    m.drop.p = 0.3
    eng.manual_seed(5)
    eng._prep(bufs, ones, ts, None, None, None, None, True)
    k = bufs.xin[:, :I] > 0
    assert abs(float(k.float().mean()) - 0.7) < 0.01
    uq = torch.unique(bufs.xin[:, :I]).tolist()
    assert len(uq) == 2 and uq[0] == 0.0 and abs(uq[1] - 1.0 / 0.7) < 1e-6
    lo, hi = k[:, :1024].float(), k[:, 1024:2048].float()
    corr = float(((lo - lo.mean()) * (hi - hi.mean())).mean() / (lo.std() * hi.std()))
    assert abs(corr) < 0.02
    m.drop.p = 0.5


def test_randn_kernel_is_the_input_builders_noise_stream():
    """gdmcf_randn_f32 (stands where the reference calls th.randn_like: gaussian_diffusion.py:328-331, :210-217): stream 0 with
    the input builder's (seed, offset) IS the noise the builder draws in place -- x_t from the in-kernel draw equals x_t from the
    filled buffer bit for bit (odd width: 4-byte-aligned rows, ragged last group); moments; counter-based repeatability."""
    from gdmcf_amd import _lib
    torch.manual_seed(7)
    B, I = 48, 4099
    d = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear", 1.0, 0.5, 0.5, 4, DEV)
    x0 = (torch.rand(B, I, device=DEV) < 0.1).float()
    ts = torch.randint(0, 4, (B,), device=DEV)
    a = d.q_sample(x0, ts)  # in-kernel Philox draw: offset (1 << 40) + _q_calls, seed torch.initial_seed()
    seed = int(torch.initial_seed()) & (2 ** 63 - 1)
    nz = _lib.philox_randn((B, I), DEV, seed, (1 << 40) + d._q_calls, stream_id=0)
    b = d.q_sample(x0, ts, noise=nz)
    assert torch.equal(a, b)
    # the callers' streams: N(0,1) moments, a different stream / offset gives different numbers, the same ones the same
    z = _lib.philox_randn((256, 8191), DEV, 99, 1, stream_id=4)
    assert abs(float(z.mean())) < 0.005 and abs(float(z.std()) - 1.0) < 0.005
    assert abs(float((z ** 4).mean()) - 3.0) < 0.05 and abs(float((z ** 3).mean())) < 0.02
    assert float(z.abs().max()) < 7.0 and bool(torch.isfinite(z).all())
    assert torch.equal(z, _lib.philox_randn((256, 8191), DEV, 99, 1, stream_id=4))
    assert not torch.equal(z, _lib.philox_randn((256, 8191), DEV, 99, 2, stream_id=4))
    assert not torch.equal(z, _lib.philox_randn((256, 8191), DEV, 99, 1, stream_id=7))
    # into a wider buffer: the columns past `cols` are not touched
    buf = torch.full((5, 40), -1.0, device=DEV)
    _lib.philox_randn((5, 33), DEV, 3, 3, out=buf[:, :33])
    assert torch.equal(buf[:, 33:], torch.full((5, 7), -1.0, device=DEV)) and bool((buf[:, :33] != -1.0).all())
    with pytest.raises(RuntimeError):
        _lib.philox_randn((4, 4), "cpu", 1, 1)


@pytest.mark.parametrize("t0_likelihood", [True, False])
def test_eps_target_kernel_equals_the_elementwise_expressions(t0_likelihood):
    """gdmcf_eps_target_f32 against the reference's element-wise passes (gaussian_diffusion.py:344-348: target = eps except the
    t == 0 rows, whose target is r1[0]*x_t - x0 with weight r2[0] and twice the divisor): bit-exact, out of place and in place."""
    from gdmcf_amd import _lib
    torch.manual_seed(3)
    B, I, ld = 37, 1003, 1024
    noise = torch.randn(B, I, device=DEV)
    xt = torch.randn(B, ld, device=DEV)
    x0 = (torch.rand(B, I, device=DEV) < 0.2).float()
    ts = torch.randint(0, 3, (B,), device=DEV)
    ts[0], ts[B - 1] = 0, 0
    r1 = torch.tensor([1.2345678, 9.0], device=DEV)
    r2 = torch.tensor([0.7654321, 9.0], device=DEV)
    is0 = (ts == 0) if t0_likelihood else torch.zeros_like(ts, dtype=torch.bool)
    want_t = torch.where(is0[:, None], r1[0] * xt[:, :I] - x0, noise)
    want_a = torch.where(is0, r2[0], torch.ones((), device=DEV))
    want_d = torch.where(is0, 2.0 * I, 1.0 * I).float()
    lib = _lib.load()
    for in_place in (False, True):
        nz = noise.clone()
        tgt = nz if in_place else torch.full((B, I), float("nan"), device=DEV)
        al, rd = torch.empty(B, device=DEV), torch.empty(B, device=DEV)
        _lib.check(lib.gdmcf_eps_target_f32(nz.data_ptr(), nz.stride(0), xt.data_ptr(), xt.stride(0), x0.data_ptr(), x0.stride(0),
                                            ts.data_ptr(), r1.data_ptr(), r2.data_ptr(), int(t0_likelihood), B, I, tgt.data_ptr(),
                                            tgt.stride(0), al.data_ptr(), rd.data_ptr(), _lib.stream_ptr()))
        assert torch.equal(tgt, want_t) and torch.equal(al, want_a) and torch.equal(rd, want_d)


def test_eps_training_on_the_device_noise_draw():
    """ModelMeanType.EPSILON without injected noise: the target is drawn by gdmcf_randn_f32 (no ATen pass), the x0-likelihood rows
    are patched in place; the loss equals, bit for bit, the injected-noise path's (pinned to the reference fixtures) on the SAME noise, re-drawn from the
    stream's (seed, offset); training steps run on it."""
    torch.manual_seed(11)
    B, I, T = 64, 515, 5
    m = gdmcf_amd.DNN([I, 48], [48, I], 10).to(DEV)
    m.eval()  # no dropout: the oracle sees the same input
    d = gdmcf_amd.GaussianDiffusion(ModelMeanType.EPSILON, "linear-var", 0.1, 0.001, 0.02, T, DEV)
    x = (torch.rand(B, I, device=DEV) < 0.05).float()
    ts = torch.randint(0, T, (B,), device=DEV)
    ts[:4] = 0
    pt = torch.full((B,), 1.0 / T, dtype=torch.float64, device=DEV)
    from gdmcf_amd import _lib
    calls = getattr(d, "_randn_calls", 0)
    got = d.training_losses(m, x, True, ts=ts, pt=pt)["loss"]
    assert d._randn_calls == calls + 1
    nz = _lib.philox_randn((B, I), DEV, int(torch.initial_seed()) & (2 ** 63 - 1), d._randn_calls, stream_id=4)
    d2 = gdmcf_amd.GaussianDiffusion(ModelMeanType.EPSILON, "linear-var", 0.1, 0.001, 0.02, T, DEV)
    want = d2.training_losses(m, x, True, ts=ts, pt=pt, noise=nz)["loss"]  # the injected-noise path (pinned to the fixtures)
    assert torch.equal(got, want)
    m.train()
    opt = gdmcf_amd.FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.0)
    w0 = m.out_layers[-1].weight.detach().clone()
    losses = []
    for _ in range(20):
        opt.zero_grad()
        l = d.training_losses(m, x, True)["loss"].mean()
        l.backward()
        opt.step()
        losses.append(float(l))
    # (the SNR-weighted eps loss of 20 importance-sampled steps is too noisy to demand a decrease: finite, a fresh draw per step,
    # and the step reaches the weights)
    assert all(np.isfinite(losses)) and len(set(losses)) == 20
    assert d._randn_calls == calls + 21 and not torch.equal(w0, m.out_layers[-1].weight.detach())


def test_rng_paths_train_and_decrease_loss():
    """Unseeded end-to-end run on the fused-Philox path: loss is finite, history fills, loss falls."""
    torch.manual_seed(0)
    B, I, T = 64, 777, 5
    m = gdmcf_amd.DNN([I, 64], [64, I], 10).to(DEV)
    d = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV)
    opt = gdmcf_amd.FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.0)
    x = (torch.rand(B, I, device=DEV) < 0.05).float()
    losses = []
    for _ in range(60):
        opt.zero_grad()
        l = d.training_losses(m, x, True)["loss"].mean()
        l.backward()
        opt.step()
        losses.append(float(l))
    assert np.isfinite(losses).all()
    assert bool((d.Lt_count == 10).all())  # importance sampling is live by now
    assert np.mean(losses[-10:]) < 0.7 * np.mean(losses[:10])


def test_full_size_step_matches_oracle_yelp_shape():
    """BASELINE config[1] shape (B=400, I=34395, dims=[1000], T=5): one full train step vs the oracle."""
    torch.manual_seed(0)
    B, I, hid, T = 400, 34395, 1000, 5
    om = O.DNN([I, hid], [hid, I], 10)
    model = gdmcf_amd.DNN([I, hid], [hid, I], 10)
    model.load_state_dict(om.state_dict())
    model = model.to(DEV)
    od = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T)
    gdif = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV)
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(B, I, generator=g) < 0.00075).float()
    ts = torch.randint(0, T, (B,), generator=g)
    pt = torch.ones(B)
    noise = torch.randn(B, I, generator=g)
    keep = (torch.rand(B, I, generator=g) < 0.5).float()
    oopt = O.make_optimizer(om, 1e-5)
    gopt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-5, weight_decay=0.0)
    om.train(), model.train()
    # size-independent property at full size: the loss is linear in the per-row weights 1/pt
    gdif.update_history = False
    with torch.no_grad():
        l1 = gdif.training_losses(model, cu(x), True, ts=cu(ts), pt=cu(pt), noise=cu(noise), drop_mask=cu(keep))["loss"]
        l2 = gdif.training_losses(model, cu(x), True, ts=cu(ts), pt=cu(pt * 2), noise=cu(noise), drop_mask=cu(keep))["loss"]
    np.testing.assert_array_equal((l2 * 2).cpu().numpy(), l1.cpu().numpy())  # deterministic kernels: exact
    gdif.update_history = True
    oloss, ovec = O.train_step(od, om, oopt, x, True, ts=ts, pt=pt, noise=noise, drop_mask=keep)
    gopt.zero_grad()
    terms = gdif.training_losses(model, cu(x), True, ts=cu(ts), pt=cu(pt), noise=cu(noise), drop_mask=cu(keep))
    gl = terms["loss"].mean()
    gl.backward()
    grads = {k: v.grad.clone() for k, v in model.named_parameters()}
    gopt.step()
    assert abs(float(gl) - float(oloss)) <= 1e-4 * abs(float(oloss)), (float(gl), float(oloss))
    np.testing.assert_allclose(terms["loss"].detach().cpu().numpy(), ovec.numpy(), rtol=1e-4)
    for (k, p), (_, q) in zip(model.named_parameters(), om.named_parameters()):
        assert H.relerr(grads[k].cpu().numpy(), q.grad.numpy()) < 2e-4, k
        # first AdamW step = -lr * g/(|g|+eps): ill-conditioned where |g| ~ eps (fp32 noise decides), so
        # bound the worst element by a fraction of lr and require the well-conditioned ones to agree tightly
        dp = np.abs(p.detach().cpu().numpy() - q.detach().numpy())
        big = np.abs(q.grad.numpy()) > 1e-5
        assert dp.max() < 0.25 * 1e-5, k
        assert (not big.any()) or dp[big].max() < 0.01 * 1e-5, k
    np.testing.assert_allclose(gdif.Lt_history.cpu().numpy(), od.Lt_history.numpy(), rtol=1e-4)


@pytest.mark.parametrize("shape", [(32, 515, 100), (400, 34395, 1000)])
def test_bf16_gemm_path_tracks_oracle(shape):
    """BASELINE configs[2] (bf16 denoiser GEMM inputs, f32 accumulate / state): the same injected-randomness
    train step as the f32 parity tests, against the f32 CPU oracle.  Tolerances are those of bf16 operand rounding
    (2^-9 relative per operand, averaging out over the reduction), measured with tools/bf16_check.py:
    mean training loss 3e-5 .. 3e-4 relative (<= 1e-4 at the Yelp shape), per-row loss <= 4e-3, gradients
    1e-3 .. 1e-2 relative L2.  Asserted with ~3x margin.  Also: p_sample + top-20 keeps >= 95 % of the f32 picks."""
    B, I, hid = shape
    T = 5
    torch.manual_seed(0)
    om = O.DNN([I, hid], [hid, I], 10)
    model = gdmcf_amd.DNN([I, hid], [hid, I], 10, gemm_dtype="bf16")
    model.load_state_dict(om.state_dict())
    model = model.to(DEV)
    od = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T)
    gdif = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV)
    oopt = O.make_optimizer(om, 1e-3)
    gopt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.0)
    om.train(), model.train()
    g = torch.Generator().manual_seed(1)
    full = I > 10000
    for step in range(1 if full else 3):
        x = (torch.rand(B, I, generator=g) < (0.00075 if full else 0.02)).float()
        ts = torch.randint(0, T, (B,), generator=g)
        noise = torch.randn(B, I, generator=g)
        keep = (torch.rand(B, I, generator=g) < 0.5).float()
        pt = torch.ones(B)
        oloss, ovec = O.train_step(od, om, oopt, x, True, ts=ts, pt=pt, noise=noise, drop_mask=keep)
        ograds = [q.grad.clone() for q in om.parameters()]
        gopt.zero_grad()
        terms = gdif.training_losses(model, cu(x), True, ts=cu(ts), pt=cu(pt), noise=cu(noise), drop_mask=cu(keep))
        gl = terms["loss"].mean()
        gl.backward()
        rel = abs(float(gl) - float(oloss)) / abs(float(oloss))
        assert rel <= (1.5e-4 if full else 1e-3), (step, rel)
        np.testing.assert_allclose(terms["loss"].detach().cpu().numpy(), ovec.numpy(), rtol=(1.5e-3 if full else 1.2e-2))
        for p, og in zip(model.parameters(), ograds):
            assert H.relerr(p.grad.cpu().numpy(), og.numpy()) < 3e-2
        gopt.step()
        # keep the two models in lock-step so that step s+1 again isolates one step of rounding
        model.load_state_dict(om.state_dict())
        for p, oq in zip(model.parameters(), om.parameters()):
            gopt.state[p]["exp_avg"].copy_(oopt.state[oq]["exp_avg"])
            gopt.state[p]["exp_avg_sq"].copy_(oopt.state[oq]["exp_avg_sq"])
    if not full:
        f32 = gdmcf_amd.DNN([I, hid], [hid, I], 10)
        f32.load_state_dict(om.state_dict())
        f32 = f32.to(DEV).eval()
        model.eval()
        x = (torch.rand(B, I, generator=g) < 0.02).float()
        with torch.no_grad():
            pa = gdif.p_sample(f32, cu(x), 0, False)
            pb = gdif.p_sample(model, cu(x), 0, False)
        ta, tb = torch.topk(pa, 20).indices.cpu().numpy(), torch.topk(pb, 20).indices.cpu().numpy()
        overlap = np.mean([len(set(a) & set(b)) / 20.0 for a, b in zip(ta, tb)])
        assert overlap >= 0.95, overlap


def test_bf16_shadows_stay_in_sync():
    """bf16 mode streams bf16 shadows of every GEMM operand.  After full train steps every shadow must equal the
    float32 tensor rounded to bfloat16 (bit for bit, zero padding intact): written by the input builder, the split-K
    reducers, the fused-loss epilogue, rowscale and -- for the weights -- by the AdamW kernel itself; a weight changed
    behind the library's back is picked up through its version counter."""
    from gdmcf_amd import _lib
    B, I, hid, T = 48, 1301, 255, 5  # odd row lengths: the AdamW shadow stores are then only 2-byte aligned
    torch.manual_seed(0)
    model = gdmcf_amd.DNN([I, hid], [hid, I], 10, gemm_dtype="bf16").to(DEV).train()
    gdif = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    g = torch.Generator().manual_seed(1)

    def step():
        x = (torch.rand(B, I, generator=g) < 0.03).float().to(DEV)
        opt.zero_grad()
        l = gdif.training_losses(model, x, True)["loss"].mean()
        l.backward()
        opt.step()
        return x, float(l.detach())

    def check(sh, t):
        rows, cols = sh.rows, sh.cols
        want = t[:rows, :cols].detach().bfloat16()
        assert torch.equal(sh.buf[:rows, :cols], want)
        assert float(sh.buf[rows:].abs().sum()) == 0.0 and float(sh.buf[:, cols:].abs().sum()) == 0.0  # padding

    for _ in range(3):
        step()
    eng = model.engine
    bufs = eng.buffers(B, torch.device(DEV))
    assert bufs.shadows is not None and len(eng._wshadow) == 2
    srcs = [bufs.xin, bufs.diff, bufs.acts[0], bufs.dzs[0], bufs.hs]
    for sh, t in zip(bufs.shadows, srcs):
        assert _lib.shadow_info(t.data_ptr())[0] == sh.buf.data_ptr()
        check(sh, t)
    for w in (model.in_layers[0].weight, model.out_layers[0].weight):
        check(eng._wshadow[id(w)][0], w)  # maintained by gdmcf_adamw_bf16s_f32, no re-cast
    # a weight edited by plain torch: the next step must see the new values (version-keyed refresh)
    with torch.no_grad():
        model.out_layers[0].weight.mul_(0.5)
    twin = gdmcf_amd.DNN([I, hid], [hid, I], 10, gemm_dtype="bf16").to(DEV).train()
    twin.load_state_dict(model.state_dict())
    x = (torch.rand(B, I, generator=g) < 0.03).float().to(DEV)
    ts = torch.randint(0, T, (B,), generator=g).to(DEV)
    noise, keep = torch.randn(B, I, generator=g).to(DEV), (torch.rand(B, I, generator=g) < 0.5).float().to(DEV)
    gdif.update_history = False
    with torch.no_grad():
        a = gdif.training_losses(model, x, True, ts=ts, pt=torch.ones(B, device=DEV), noise=noise, drop_mask=keep)["loss"]
        b = gdif.training_losses(twin, x, True, ts=ts, pt=torch.ones(B, device=DEV), noise=noise, drop_mask=keep)["loss"]
    assert torch.equal(a, b)
    check(eng._wshadow[id(model.out_layers[0].weight)][0], model.out_layers[0].weight)


@pytest.mark.parametrize("gemm_dtype", ["f32", "bf16"])
def test_sparse_rows_train_exactly_like_dense_rows(gemm_dtype):
    """CSR input path (gdmcf_dnn_prep_input_csr_f32 + gdmcf_linear_loss_fwd_bits_f32): handing
    `training_losses` the rows as a data_utils.CsrBatch gives bit-identical per-row losses, gradients, Lt-history and
    weights after the step as handing it the densified rows -- with in-kernel Philox noise / dropout (same seed, same
    offsets) and with injected noise / dropout masks, at a ragged width (odd I, rows crossing the 4096-column workgroup
    span, an empty row, a full row)."""
    import scipy.sparse as sp
    from gdmcf_amd.data_utils import DeviceCSR
    rng = np.random.default_rng(7)
    U, I, hid, T, B = 300, 9003, 48, 5, 37
    dense = (rng.random((U, I)) < 0.004).astype(np.float32)
    dense[5] = 0.0
    dense[6] = 1.0
    dcsr = DeviceCSR(sp.csr_matrix(dense), DEV)
    ids = torch.from_numpy(rng.permutation(U)[:B].astype(np.int64))
    ids[0], ids[1] = 5, 6
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(B, I, generator=g)
    keep = (torch.rand(B, I, generator=g) < 0.5).to(torch.uint8)
    ts = torch.randint(0, T, (B,), generator=g)
    for inject in (False, True):
        out = []
        for sparse in (False, True):
            torch.manual_seed(11)
            model = gdmcf_amd.DNN([I, hid], [hid, I], 10, gemm_dtype=gemm_dtype).to(DEV).train()
            d = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.1, 0.001, 0.01, T, DEV)
            opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-3)
            x = dcsr.batch(ids) if sparse else dcsr.rows(ids)
            kw = dict(ts=cu(ts), pt=torch.ones(B, dtype=torch.float64, device=DEV))
            if inject:
                kw.update(noise=cu(noise), drop_mask=cu(keep))
            opt.zero_grad()
            terms = d.training_losses(model, x, True, **kw)
            terms["loss"].mean().backward()
            grads = [p.grad.clone() for p in model.parameters()]
            opt.step()
            out.append((terms["loss"].detach().clone(), grads, [p.detach().clone() for p in model.parameters()],
                        d.Lt_history.clone(), d.Lt_count.clone()))
        (l0, g0, w0, h0, c0), (l1, g1, w1, h1, c1) = out
        assert torch.equal(l0, l1) and torch.equal(h0, h1) and torch.equal(c0, c1)
        for a, b in zip(g0 + w0, g1 + w1):
            assert torch.equal(a, b)
    # what cannot stay sparse densifies by itself: eps target, F.normalize
    torch.manual_seed(11)
    model = gdmcf_amd.DNN([I, hid], [hid, I], 10, norm=True).to(DEV).train()
    d = gdmcf_amd.GaussianDiffusion(ModelMeanType.EPSILON, "linear-var", 0.1, 0.001, 0.01, T, DEV)
    a = d.training_losses(model, dcsr.batch(ids), True, ts=cu(ts), pt=torch.ones(B, dtype=torch.float64, device=DEV),
                          noise=cu(noise), drop_mask=cu(keep))["loss"]
    b = d.training_losses(model, dcsr.rows(ids), True, ts=cu(ts), pt=torch.ones(B, dtype=torch.float64, device=DEV),
                          noise=cu(noise), drop_mask=cu(keep))["loss"]
    assert torch.equal(a.detach(), b.detach())


def test_sparse_rows_match_the_reference_fixture():
    """The CSR input path against the reference's own numbers: the `ragged_x0` fixture's rows fed as a CsrBatch."""
    import scipy.sparse as sp
    from gdmcf_amd.data_utils import DeviceCSR
    fx = H.load("train_ragged_x0")
    meta = H.train_meta(fx)
    model, diff = gpu_model(meta, fx).train(), gpu_diffusion(meta)
    diff.Lt_history.copy_(torch.from_numpy(fx["Lt_history0"]))
    diff.Lt_count.copy_(torch.from_numpy(fx["Lt_count0"]))
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=meta["lr"], weight_decay=meta["wd"])
    for s in range(meta["n_steps"]):
        inp = H.step_inputs(fx, s)
        dcsr = DeviceCSR(sp.csr_matrix(inp["x"].numpy()), DEV)
        opt.zero_grad()
        terms = diff.training_losses(model, dcsr.batch(torch.arange(meta["B"])), True, ts=cu(inp["ts"]), pt=cu(inp["pt"]),
                                     noise=cu(inp["noise"]), drop_mask=cu(inp["drop_mask"]))
        loss = terms["loss"].mean()
        loss.backward()
        assert abs(float(loss.detach()) - float(fx[f"s{s}.loss"])) <= 1e-4 * abs(float(fx[f"s{s}.loss"]))
        np.testing.assert_allclose(terms["loss"].detach().cpu().numpy(), fx[f"s{s}.loss_vec"], rtol=1e-4)
        if s == 0:
            for k, v in model.named_parameters():
                assert H.relerr(v.grad.cpu().numpy(), fx["g0." + k]) < 2e-4, k
        opt.step()


def test_device_metrics_equal_the_reference_loop():
    """gdmcf_topn_metrics_f64 (device) == computeTopNAccuracy (the reference's Python loop, pinned by the golden
    vectors in the CPU suite): identical 4-decimal results on random rankings incl. users with empty ground truth,
    ground truths longer and shorter than N, hits at rank 1 and at the last rank; cut-off validation errors."""
    import scipy.sparse as sp
    from gdmcf_amd import evaluate_utils as EU
    rng = np.random.default_rng(3)
    U, I, K = 700, 900, 100
    topN = [1, 10, 20, 50, 100]
    dens = rng.choice([0.0, 0.002, 0.02, 0.2], size=U, p=[0.1, 0.4, 0.4, 0.1])
    gt = sp.csr_matrix((rng.random((U, I)) < dens[:, None]).astype(np.float32))
    pred = np.stack([rng.permutation(I)[:K] for _ in range(U)]).astype(np.int64)
    for u in range(0, U, 7):  # plant certain hits at the first / last rank
        row = gt[u].indices
        if len(row):
            pred[u, 0 if u % 2 else K - 1] = row[0]
            pred[u] = np.concatenate([pred[u][:1], [x for x in pred[u][1:] if x != pred[u][0]], rng.permutation(I)[:5]])[:K]
    want = EU.computeTopNAccuracy([gt[u].indices.tolist() for u in range(U)], pred.tolist(), topN)
    got = EU.computeTopNAccuracy_device(gt, torch.from_numpy(pred).to(DEV), topN)
    assert got == want
    with pytest.raises(ValueError):
        EU.computeTopNAccuracy_device(gt, torch.from_numpy(pred).to(DEV), [20, 10])
    with pytest.raises(AssertionError):
        EU.computeTopNAccuracy_device(gt, torch.from_numpy(pred[:, :50]).to(DEV), [10, 100])


def test_driver_train_and_evaluate_match_oracle_loop():
    """reference main.py:327-351 + :267-310 end to end on a small synthetic problem: the HIP driver and the
    oracle loop start from the same weights, see the same batches and the same injected randomness, and must
    report the same Recall/NDCG@N."""
    import scipy.sparse as sp
    from gdmcf_amd import driver
    rng = np.random.default_rng(0)
    U, I, hid, T, B = 96, 300, 48, 5, 32
    dense = (rng.random((U, I)) < 0.06).astype(np.float32)
    test = ((rng.random((U, I)) < 0.03) & (dense == 0)).astype(np.float32)
    train_csr, test_csr = sp.csr_matrix(dense), sp.csr_matrix(test)
    torch.manual_seed(0)
    om = O.DNN([I, hid], [hid, I], 10)
    gm = gdmcf_amd.DNN([I, hid], [hid, I], 10)
    gm.load_state_dict(om.state_dict())
    gm = gm.to(DEV)
    od = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T)
    gd_ = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV)
    oopt = O.make_optimizer(om, 1e-3)
    gopt = gdmcf_amd.FusedAdamW(gm.parameters(), lr=1e-3, weight_decay=0.0)
    g = torch.Generator().manual_seed(5)
    om.train(), gm.train()
    for epoch in range(3):  # the same three epochs on both sides, randomness injected
        for lo in range(0, U, B):
            x = torch.from_numpy(dense[lo:lo + B])
            ts = torch.randint(0, T, (B,), generator=g)
            noise = torch.randn(B, I, generator=g)
            keep = (torch.rand(B, I, generator=g) < 0.5).float()
            O.train_step(od, om, oopt, x, True, ts=ts, pt=torch.ones(B), noise=noise, drop_mask=keep)
            gopt.zero_grad()
            l = gd_.training_losses(gm, cu(x), True, ts=cu(ts), pt=cu(torch.ones(B)), noise=cu(noise), drop_mask=cu(keep))
            l["loss"].mean().backward()
            gopt.step()
    topN = [10, 20, 50]
    res = driver.evaluate(gd_, gm, train_csr, test_csr, train_csr, topN, 0, False, B, DEV)
    # the oracle's evaluate loop
    om.eval()
    preds = []
    with torch.no_grad():
        for lo in range(0, U, B):
            x = torch.from_numpy(dense[lo:lo + B])
            p = od.p_sample(om, x, 0, False)
            rows, cols = x.nonzero(as_tuple=True)
            preds.extend(O.masked_topk(p, rows, cols, topN[-1]).tolist())
    gt = [test_csr[i, :].nonzero()[1].tolist() for i in range(U)]
    ref = O.computeTopNAccuracy(gt, preds, topN)
    # three Adam steps amplify fp32 noise a little: metrics (4 decimals, averaged over 96 users) agree to 1e-3
    np.testing.assert_allclose(np.array(res), np.array(ref), rtol=0, atol=1.1e-3)
    # unseeded smoke of the epoch loop itself (Philox path): runs, finite, counts batches
    total, count = driver.train_one_epoch(gd_, gm, gopt, train_csr, B, DEV, generator=torch.Generator().manual_seed(1))
    assert count == U // B and np.isfinite(total)


def test_device_batch_loader_matches_dense_rows():
    import scipy.sparse as sp
    from gdmcf_amd import data_utils
    fx = H.load("data_load")
    csr = sp.csr_matrix(fx["train"])  # contains a 2.0 and a 3.0 (duplicated pairs)
    dcsr = data_utils.DeviceCSR(csr, DEV)
    ids = torch.tensor([5, 0, 11, 11, 3])
    np.testing.assert_array_equal(dcsr.rows(ids).cpu().numpy(), fx["train"][ids.numpy()].astype(np.float32))
    rng = np.random.default_rng(0)
    big = sp.random(257, 4099, density=0.01, format="csr", random_state=1, data_rvs=lambda n: np.ones(n))
    loader = data_utils.DeviceBatchLoader(big, 64, shuffle=True, drop_last=True, device=DEV,
                                          generator=torch.Generator().manual_seed(0))
    assert len(loader) == 4
    seen = []
    for batch, idx in loader:
        assert batch.shape == (64, 4099) and batch.is_cuda
        np.testing.assert_array_equal(batch.cpu().numpy(), np.asarray(big[idx.numpy()].todense(), dtype=np.float32))
        seen.extend(idx.tolist())
    assert len(set(seen)) == 256
    # the same batches as CSR rows (never densified) and as bare row ids (for graph.GraphedTrainStep)
    mk = lambda **kw: data_utils.DeviceBatchLoader(big, 64, shuffle=True, drop_last=True, device=DEV,
                                                   generator=torch.Generator().manual_seed(0), **kw)
    for (dense, i0), (sparse, i1), (none, i2) in zip(mk(), mk(sparse=True), mk(ids_only=True)):
        assert torch.equal(i0, i1) and torch.equal(i0, i2) and none is None
        assert isinstance(sparse, data_utils.CsrBatch) and torch.equal(sparse.dense(), dense)


def test_driver_epoch_is_the_same_dense_sparse_and_graphed():
    """driver.train_one_epoch with dense rows, with CSR rows (sparse=True) and with the step replayed from a hipGraph
    (graph_step=): same epoch loss and the same weights bit for bit (what examples/train_synthetic.py --sparse-rows /
    --graph run)."""
    import scipy.sparse as sp
    from gdmcf_amd import data_utils, driver
    from gdmcf_amd.graph import GraphedTrainStep
    U, I, B = 640, 3001, 64
    csr = sp.random(U, I, density=0.01, format="csr", random_state=3, data_rvs=lambda n: np.ones(n)).astype(np.float32)
    out = []
    for mode in ("dense", "sparse", "graph"):
        torch.manual_seed(2)
        model = gdmcf_amd.DNN([I, 96], [96, I], 10).to(DEV)
        diff = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.1, 0.001, 0.01, 5, DEV)
        opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-3)
        dcsr = data_utils.DeviceCSR(csr, DEV)
        gen = torch.Generator().manual_seed(9)
        gstep = GraphedTrainStep(diff, model, opt, dcsr, B) if mode == "graph" else None
        tot = 0.0
        for _ in range(2):
            t, n = driver.train_one_epoch(diff, model, opt, dcsr, B, DEV, generator=gen, sparse=(mode == "sparse"), graph_step=gstep)
            assert n == U // B
            tot += t
        if gstep is not None:
            gstep.close()
        out.append((tot, [p.detach().clone() for p in model.parameters()], diff.Lt_history.clone()))
    for tot, ws, hist in out[1:]:
        assert tot == out[0][0]
        assert torch.equal(hist, out[0][2])
        for a, b in zip(ws, out[0][1]):
            assert torch.equal(a, b)


def test_checkpoint_resume_is_bit_exact(tmp_path):
    """2 steps + save + 2 steps  ==  load + 2 steps, bit for bit, on the unseeded Philox path (weights, AdamW
    moments, Lt-history and the random stream position all restored)."""
    from gdmcf_amd import checkpoint

    def build():
        torch.manual_seed(7)
        m = gdmcf_amd.DNN([515, 64], [64, 515], 10).to(DEV)
        d = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 5, DEV)
        o = gdmcf_amd.FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
        return m.train(), d, o

    g = torch.Generator().manual_seed(0)
    xs = [(torch.rand(32, 515, generator=g) < 0.05).float().to(DEV) for _ in range(4)]

    def run(m, d, o, batches):
        out = []
        for x in batches:
            o.zero_grad()
            l = d.training_losses(m, x, True)["loss"].mean()
            l.backward()
            o.step()
            out.append(float(l.detach()))
        return out

    m, d, o = build()
    run(m, d, o, xs[:2])
    checkpoint.save_checkpoint(tmp_path / "ck.pt", m, d, o, epoch=3)
    ref_losses = run(m, d, o, xs[2:])
    m2, d2, o2 = build()
    epoch, _ = checkpoint.load_checkpoint(tmp_path / "ck.pt", m2, d2, o2)
    assert epoch == 3
    got = run(m2, d2, o2, xs[2:])
    assert got == ref_losses
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)
    assert torch.equal(d.Lt_history, d2.Lt_history) and torch.equal(d.Lt_count, d2.Lt_count)
    # the model part loads into the oracle (= reference layout) as well
    om = O.DNN([515, 64], [64, 515], 10)
    om.load_state_dict(torch.load(tmp_path / "ck.pt", weights_only=False)["model"])


def test_lightgcn_bpr_step_gradients_match_oracle():
    """reference lightGCN.py:196-219 + :291-298: BPR loss and dE0 through the HIP SpMM (backward = the same
    propagation, A~ symmetric) vs torch.sparse autograd on the CPU; then a few optimiser steps reduce the loss."""
    from gdmcf_amd.lightgcn import bpr_loss, sample_bpr_batch
    rng = np.random.default_rng(0)
    U, It, d, L, nnz = 400, 250, 64, 3, 5000
    users = np.concatenate([rng.integers(0, U, nnz), np.arange(U)])
    items = np.concatenate([np.minimum((rng.pareto(1.2, nnz) * It / 20).astype(np.int64), It - 1), rng.integers(0, It, U)])
    A = O.lightgcn_norm_adj(users, items, U, It)
    m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": items}, U, It, L, d, device=DEV).to(DEV)
    E0 = m.E0.weight.detach().cpu().numpy().copy()
    R = A[:U, U:].tocsr()
    bu, bp, bn = sample_bpr_batch(R.indptr, R.indices, U, It, 128, rng)
    assert all(p in R.indices[R.indptr[u]:R.indptr[u + 1]] for u, p in zip(bu, bp))
    assert not any(n in R.indices[R.indptr[u]:R.indptr[u + 1]] for u, n in zip(bu, bn))
    decay = 1e-4
    mf_ref, reg_ref, g_ref = O.lightgcn_bpr_step(A, E0, L, U, bu, bp, bn, decay)
    ue, pe, ne, u0, p0, n0 = m(cu(torch.from_numpy(bu)), cu(torch.from_numpy(bp)), cu(torch.from_numpy(bn)))
    mf, reg = bpr_loss(bu, ue, pe, ne, u0, p0, n0)
    (mf + decay * reg).backward()
    assert abs(float(mf) - mf_ref) < 1e-6 and abs(float(reg) - reg_ref) < 1e-4 * reg_ref
    assert H.relerr(m.E0.weight.grad.cpu().numpy(), g_ref) < 1e-5
    opt = torch.optim.Adam(m.parameters(), lr=0.005)  # as the reference script (lightGCN.py:255)
    losses = []
    for _ in range(30):
        opt.zero_grad()
        bu, bp, bn = sample_bpr_batch(R.indptr, R.indices, U, It, 128, rng)
        out = m(cu(torch.from_numpy(bu)), cu(torch.from_numpy(bp)), cu(torch.from_numpy(bn)))
        mf, reg = bpr_loss(bu, *out)
        (mf + decay * reg).backward()
        opt.step()
        losses.append(float(mf))
    assert np.mean(losses[-5:]) < np.mean(losses[:5])


@pytest.mark.parametrize("prec", ["f32", "bf16", "bf16-shadows", "f32x3"])
def test_linear_entry_points_random_shapes(prec):
    """Adversarial shapes for the branch-free edge loaders (clamped addresses, in-register shifts, K tails,
    odd leading dimensions, split-K on/off): every dense entry point of the C ABI vs float64 matmul.
    bf16 mode (gdmcf_gemm_precision): the same, against float64 matmul of the bfloat16-rounded operands --
    bf16 x bf16 products are exact in f32, so the only difference left is f32 accumulation order.
    f32x3 mode (three-term bf16 split of both operands, six bf16 MFMAs per block: gemm_split.hip): held to the SAME
    reference and tolerance as the native f32 kernels -- no operand is rounded."""
    from gdmcf_amd import _lib
    lib = _lib.load()
    prev = lib.gdmcf_gemm_precision({"f32": 0, "bf16": 1, "bf16-shadows": 1, "f32x3": 2}[prec])
    try:
        _linear_entry_points_random_shapes(lib, "bf16" if prec.startswith("bf16") else "f32", shadows=(prec == "bf16-shadows"))
    finally:
        lib.gdmcf_gemm_precision(prev)
        lib.gdmcf_bf16_shadow_clear(None)


def _linear_entry_points_random_shapes(lib, prec, shadows=False, shapes=None, seed=0):
    """shadows=True: every operand gets a registered bf16 shadow (the kernels then stream those, no edge predicates,
    zero padding) -- same reference, same tolerance."""
    from gdmcf_amd import _lib
    D = (lambda t: t.bfloat16().double()) if prec == "bf16" else (lambda t: t.double())
    rng = np.random.default_rng(seed)
    st = _lib.stream_ptr()
    given = shapes
    # the first three need the element-wise kernel of gemm_small.hip in f32 (K < 4, or fewer than 4 rows of a
    # row-contiguous operand); the bf16 loaders take them as they are
    shapes = [(2, 3, 3), (4, 5, 2), (3, 2, 9), (1, 4, 4), (3, 5, 7), (16, 16, 16), (17, 33, 65), (80, 128, 32), (81, 129, 33),
              (100, 257, 36), (400, 130, 1000), (7, 1000, 515), (129, 70, 4099), (65, 64, 8195), (33, 300, 31), (5, 6, 20000)]
    if prec == "bf16":  # the last four take the 208x256 tile class (fused epilogues and weight gradients included)
        shapes += [(5, 2, 130), (200, 300, 515), (413, 1000, 700), (400, 28001, 70), (64, 1100, 28000)]
    if given is not None:
        shapes = given
    keep = []
    for (M, N, K) in shapes:
        for pad in (0, 3):
            lda, ldw, ldc = K + pad, K + (1 if pad else 0), N + pad
            A = torch.zeros(M, lda, device=DEV); A[:, :K] = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(DEV)
            W = torch.zeros(N, ldw, device=DEV); W[:, :K] = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(DEV)
            bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(DEV)
            for sh in keep:  # unregister before the allocator can hand the same addresses to new tensors
                sh.close()
            keep = [_lib.Bf16Shadow(A[:, :K]), _lib.Bf16Shadow(W[:, :K])] if shadows else []
            assert not shadows or lib.gdmcf_bf16_shadow_get(A.data_ptr()) == keep[0].buf.data_ptr()
            ref = D(A[:, :K]) @ D(W[:, :K]).T + bias.double()
            close = lambda got, want: float((got - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
            ws_bytes = int(lib.gdmcf_linear_ws_bytes(M, N, K))
            ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=DEV)
            # forward (+tanh)
            C = torch.full((M, ldc), float("nan"), device=DEV)
            _lib.check(lib.gdmcf_linear_fwd_f32(A.data_ptr(), lda, W.data_ptr(), ldw, bias.data_ptr(), 1, M, N, K, C.data_ptr(),
                                                ldc, ws.data_ptr(), ws_bytes, st))
            assert close(C[:, :N].double(), torch.tanh(ref)), ("fwd", M, N, K, pad)
            assert pad == 0 or bool(torch.isnan(C[:, N:]).all())  # nothing written outside [M, N]
            # fused loss
            tgt = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).to(DEV)
            alpha = torch.from_numpy(rng.uniform(0.5, 1.5, M).astype(np.float32)).to(DEV)
            diff = torch.full((M, ldc), float("nan"), device=DEV)
            out = torch.empty(M, N, device=DEV)
            rowpart = torch.zeros(M, lib.gdmcf_loss_tiles(N), device=DEV)
            rowsum = torch.zeros(M, device=DEV)
            _lib.check(lib.gdmcf_linear_loss_fwd_f32(A.data_ptr(), lda, W.data_ptr(), ldw, bias.data_ptr(), tgt.data_ptr(), N,
                                                     alpha.data_ptr(), M, N, K, out.data_ptr(), N, diff.data_ptr(), ldc,
                                                     rowpart.data_ptr(), rowsum.data_ptr(), st))
            dref = alpha.double()[:, None] * ref - tgt.double()
            assert close(diff[:, :N].double(), dref) and close(out.double(), ref), ("loss", M, N, K, pad)
            # (rows whose residual nearly cancels carry the f32 rounding of alpha*out - target relative to their tiny sum)
            want_rs = (dref ** 2).sum(1).cpu().numpy()
            np.testing.assert_allclose(rowsum.cpu().numpy(), want_rs, rtol=2e-5, atol=2e-6 * float(want_rs.max()))
            # backward wrt input: dA = rs * (dZ @ W) * (1 - act^2)
            dZ = torch.zeros(M, ldc, device=DEV); dZ[:, :N] = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).to(DEV)
            rs = torch.from_numpy(rng.uniform(0.5, 1.5, M).astype(np.float32)).to(DEV)
            act = torch.zeros(M, lda, device=DEV); act[:, :K] = torch.from_numpy(rng.uniform(-0.9, 0.9, (M, K)).astype(np.float32)).to(DEV)
            dA = torch.full((M, lda), float("nan"), device=DEV)
            if shadows:
                keep.append(_lib.Bf16Shadow(dZ[:, :N]))
            _lib.check(lib.gdmcf_linear_bwd_input_f32(dZ.data_ptr(), ldc, W.data_ptr(), ldw, rs.data_ptr(), act.data_ptr(), lda, 1,
                                                      M, N, K, dA.data_ptr(), lda, ws.data_ptr(), ws_bytes, st))
            r2 = rs.double()[:, None] * (D(dZ[:, :N]) @ D(W[:, :K])) * (1 - act[:, :K].double() ** 2)
            assert close(dA[:, :K].double(), r2), ("bwd_input", M, N, K, pad)
            # backward wrt weight: dW = dZ^T @ A, db = sum_m rs*dZ
            dW = torch.full((N, ldw), float("nan"), device=DEV)
            db = torch.empty(N, device=DEV)
            _lib.check(lib.gdmcf_linear_bwd_weight_f32(dZ.data_ptr(), ldc, A.data_ptr(), lda, rs.data_ptr(), 0, M, N, K, dW.data_ptr(),
                                                       ldw, db.data_ptr(), 0, st))
            r3 = D(dZ[:, :N]).T @ D(A[:, :K])
            assert close(dW[:, :K].double(), r3), ("bwd_weight", M, N, K, pad)
            np.testing.assert_allclose(db.cpu().numpy(), (rs.double()[:, None] * dZ[:, :N].double()).sum(0).cpu().numpy(),
                                       rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f32x3"])
@pytest.mark.parametrize("shape", [(1280, 256, 64), (1301, 255, 48)])
def test_fused_optimizer_row_epilogue_matches_separate_pass(shape, dtype):
    """fuse_into_backward at shapes that take the 128x128 wave-specialised f32 kernel / the bf16 kernels, whose fused
    epilogue hands the gradient tile through LDS and streams W, exp_avg, exp_avg_sq row-wise (gemm_epilogue_rows):
    same weights, moments and losses as the separate AdamW pass over 5 steps (Philox path, weight decay on); in bf16
    mode the weight shadows refreshed by the epilogue must equal the rounded weights."""
    I, hid, B = shape

    def run(fuse):
        torch.manual_seed(0)
        m = gdmcf_amd.DNN([I, hid], [hid, I], 10, gemm_dtype=dtype).to(DEV).train()
        d = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 5, DEV)
        o = gdmcf_amd.FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
        if fuse:
            o.fuse_into_backward(m)
        m.engine.manual_seed(99)
        g = torch.Generator().manual_seed(1)
        losses = []
        for _ in range(5):
            x = (torch.rand(B, I, generator=g) < 0.02).float().to(DEV)
            ts = torch.randint(0, 5, (B,), generator=g).to(DEV)
            o.zero_grad()
            l = d.training_losses(m, x, True, ts=ts, pt=torch.ones(B, device=DEV))["loss"].mean()
            l.backward()
            o.step()
            losses.append(float(l.detach()))
        return m, o, losses

    m0, o0, l0 = run(False)
    m1, o1, l1 = run(True)
    np.testing.assert_allclose(l1, l0, rtol=1e-6)
    for a, b in zip(m0.parameters(), m1.parameters()):
        assert float((a - b).abs().max()) <= 2e-3 * 1e-3 * 5  # same math in two compilation contexts (FMA contraction)
        assert H.relerr(o1.state[b]["exp_avg_sq"].cpu().numpy(), o0.state[a]["exp_avg_sq"].cpu().numpy()) < 1e-6
    if dtype == "bf16":
        for w in (m1.in_layers[0].weight, m1.out_layers[0].weight):
            sh = m1.engine._wshadow[id(w)][0]
            assert torch.equal(sh.buf[:sh.rows, :sh.cols], w.detach().bfloat16())


def test_bench_emits_the_contract_line():
    """bench.py prints ONE JSON line with the driver's contract fields, the roofline object of the dominant kernel
    (HIP-event timing inside the run) and, with the default flags, a cpu_baseline object."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "8", "--warmup", "2", "--cpu-seconds", "1"],
                       capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "training users/sec" and d["unit"] == "users/s" and d["n_gpus"] == 1 and d["steps"] == 8
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    ro = d["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s") and 0 < ro["frac"] < 1
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    # roofline.traffic is MEASURED in the run (two rocprofv3 --pmc child passes of the same command), not looked up
    assert ro["traffic"] and ro["traffic"] > 1e6 and "measured in this run" in ro["traffic_source"], ro["traffic_source"]
    # single GPU: AdamW of the two large weights runs inside their weight-gradient products; the separate pass is timed beside it
    assert "inside the weight-gradient products" in d["optimizer"] and d["fused_optimizer_leg"]["is_main_line"] is True
    assert d["separate_optimizer_leg"]["ms_per_step"] > 0 and d["clock_preheat"]["seconds"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert abs(d["value"] - 400 * 8 / (d["ms_per_step"] * 8e-3)) / d["value"] < 1e-3
    assert cb["physical_cores"] is None or cb["physical_cores"] >= 1
    assert cb["one_thread"]["value"] > 0  # the one-thread orientation figure of BASELINE.md section 3
    # the SpMM and the evaluation path are part of the default line (north_star sets a target on the SpMM)
    sp = d["spmm"]
    assert sp["bound"] == "hbm" and sp["ms_per_layer"] > 0 and abs(sp["frac"] - sp["achieved"] / sp["peak"]) < 1e-3
    assert d["sampling"]["users_per_s"] > 0 and d["ranks_in_group"] == 1
    # BASELINE configs[2] (Amazon-Book shape, bf16 GEMM inputs) rides in the default line so the driver's record carries it
    c2 = d["configs2_leg"]
    assert "error" not in c2, c2
    assert c2["dtype"] == "bf16" and c2["n_items"] == 94949 and c2["ms_per_step"] > 0 and c2["users_per_s"] > 0
    dk = c2["dominant_kernel"]
    assert dk["bound"] in ("hbm", "mfma") and abs(dk["frac"] - dk["achieved"] / dk["peak"]) < 1e-3
    assert c2["fused_optimizer"]["ms_per_step"] > 0
    # asked for two GPUs on a one-GPU box: no line, non-zero exit
    if __import__("torch").cuda.device_count() < 2:
        r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                            capture_output=True, text=True, cwd=root, timeout=600)
        assert r2.returncode != 0 and "{" not in r2.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("I,k", [(34395, 100), (34395, 1), (34395, 256), (5000, 20), (40960, 50), (41000, 50), (300, 300)])
def test_topk_fast_path_and_its_redo_rows_against_a_stable_sort(I, k):
    """gdmcf_topk_masked_f32 (reference main.py:296-301: history items to -inf, torch.topk) selects through topk_fast_kernel since
    round 4 (k <= 256, rows up to 40 960 wide) and hands the rows it cannot bound -- more than 1 024 elements at or above the k-th
    thread maximum -- to the radix-select kernel in a second launch.  Row kinds here: random scores; all scores equal (every
    element ties: redo); two distinct values with the threshold inside the large tie group (redo); the k largest values all in
    one thread's columns (index = 7 mod 1024: the bound is weak, redo); masked rows with fewer than k unmasked items (the tail is
    -inf in index order); a row with NaN-free negatives only.  Expected: indices of a STABLE descending sort (lower index first
    among equal scores), exactly; GDMCF_TOPK_FAST=0 (radix select only) is covered by the other top-k tests via small k > 256."""
    from gdmcf_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(I + k)
    rows = []
    rows.append(torch.randn(I, generator=g))
    rows.append(torch.full((I,), 0.25))
    two = torch.where(torch.rand(I, generator=g) < 0.001, 1.0, 0.5)
    rows.append(two)
    adv = torch.randn(I, generator=g) * 0.01
    sel = torch.arange(7, I, 1024)[: max(1, min(k, (I - 7 + 1023) // 1024))]
    adv[sel] = 5.0 + torch.rand(len(sel), generator=g)
    rows.append(adv)
    rows.append(torch.randn(I, generator=g))       # masked heavily below
    rows.append(-torch.rand(I, generator=g) - 1.0)
    pred = torch.stack(rows).to(DEV)
    B = pred.shape[0]
    # history: row 4 keeps only k // 2 + 1 items unmasked; the others mask a random 1 %
    indptr, indices = [0], []
    for b in range(B):
        if b == 4:
            keep = torch.randperm(I, generator=g)[: k // 2 + 1]
            m = torch.ones(I, dtype=torch.bool)
            m[keep] = False
            cols = torch.nonzero(m).flatten()
        else:
            cols = torch.nonzero(torch.rand(I, generator=g) < 0.01).flatten()
        indices.append(cols)
        indptr.append(indptr[-1] + len(cols))
    indptr_t = torch.tensor(indptr, dtype=torch.int64, device=DEV)
    indices_t = torch.cat(indices).to(torch.int32).to(DEV)
    idx = torch.full((B, k), -7, dtype=torch.int64, device=DEV)
    val = torch.full((B, k), float("nan"), device=DEV)
    _lib.check(lib.gdmcf_topk_masked_f32(pred.data_ptr(), pred.stride(0), B, I, indptr_t.data_ptr(), indices_t.data_ptr(), k,
                                         idx.data_ptr(), val.data_ptr(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    masked = pred.clone()
    for b in range(B):
        masked[b, indices[b].to(DEV)] = float("-inf")
    want_val, want_idx = torch.sort(masked, dim=1, descending=True, stable=True)
    assert torch.equal(idx, want_idx[:, :k]), [int((idx[b] != want_idx[b, :k]).sum()) for b in range(B)]
    assert torch.equal(val, want_val[:, :k])


def test_checkpoint_resume_with_the_fused_optimiser_on_seated_weights(tmp_path):
    """The same resume (reference main.py:258, :343-351 with a save between steps) with AdamW inside the weight-gradient products:
    the weights and their moments sit on 128-byte rows (FusedAdamW.fuse_into_backward), the checkpoint goes through the views,
    and a fresh model / optimiser -- fused before OR after loading -- continues bit for bit; the model part still loads into the
    oracle's (= the reference's) contiguous layout, and into a model that never fuses."""
    from gdmcf_amd import checkpoint

    def build(fuse):
        torch.manual_seed(7)
        m = gdmcf_amd.DNN([515, 64], [64, 515], 10).to(DEV)
        d = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 5, DEV)
        o = gdmcf_amd.FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
        if fuse:
            o.fuse_into_backward(m, min_numel=1 << 12)
        return m.train(), d, o

    g = torch.Generator().manual_seed(0)
    xs = [(torch.rand(32, 515, generator=g) < 0.05).float().to(DEV) for _ in range(4)]

    def run(m, d, o, batches):
        out = []
        for x in batches:
            o.zero_grad()
            l = d.training_losses(m, x, True)["loss"].mean()
            l.backward()
            o.step()
            out.append(float(l.detach()))
        return out

    m, d, o = build(True)
    assert [w.stride(0) for (w, _, _) in m.layer_list()] == [544, 64]  # 525 -> 544 floats per row; 64 already on lines
    run(m, d, o, xs[:2])
    checkpoint.save_checkpoint(tmp_path / "ck.pt", m, d, o, epoch=3)
    ref_losses = run(m, d, o, xs[2:])
    for order in ("fuse-then-load", "load-then-fuse", "never-fused"):
        m2, d2, o2 = build(order == "fuse-then-load")
        epoch, _ = checkpoint.load_checkpoint(tmp_path / "ck.pt", m2, d2, o2)
        if order == "load-then-fuse":
            o2.fuse_into_backward(m2, min_numel=1 << 12)
        assert epoch == 3
        got = run(m2, d2, o2, xs[2:])
        assert got == ref_losses, order
        for a, b in zip(m.parameters(), m2.parameters()):
            assert torch.equal(a, b), order
        for a, b in zip(m.parameters(), m2.parameters()):
            assert torch.equal(o.state[a]["exp_avg"], o2.state[b]["exp_avg"]), order
            assert torch.equal(o.state[a]["exp_avg_sq"], o2.state[b]["exp_avg_sq"]), order
        assert torch.equal(d.Lt_history, d2.Lt_history) and torch.equal(d.Lt_count, d2.Lt_count)
    om = O.DNN([515, 64], [64, 515], 10)
    om.load_state_dict(torch.load(tmp_path / "ck.pt", weights_only=False)["model"])
    assert all(p.is_contiguous() for p in om.parameters())
