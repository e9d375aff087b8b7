"""GPU (-m gpu): the one-hot variant (first slice) -- GaussianDiffusionDiscrete(CatOneHot=True) driving a
DNNOneHot denoiser -- through the C ABI against the committed outputs of the real reference (tests/golden/onehot_*.npz)
and the CPU oracle.  Tolerances as for the plain DNN: loss <= 1e-4 relative (north_star), kept one-hot bits bit-exact,
everything else fp32 summation-order noise."""
import numpy as np
import pytest
import torch

import gdmcf_amd
from gdmcf_amd import ModelMeanType
from oracle import gdmcf_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cu(t):
    return t.to(DEV)


def gpu_pair(meta, fx):
    I, dims = meta["I"], meta["dims"]
    m = gdmcf_amd.DNNOneHot([I] + dims, dims[::-1] + [I], meta.get("emb", 10), norm=meta.get("norm", False))
    m.load_state_dict(H.state_dict_from(fx))
    mt = {"x0": ModelMeanType.START_X, "eps": ModelMeanType.EPSILON}[meta["mean_type"]]
    d = gdmcf_amd.GaussianDiffusionDiscrete(mt, meta.get("schedule", "linear-var"), meta["scale"], meta["nmin"],
                                            meta["nmax"], meta["T"], DEV, discrete=meta["discrete"], CatOneHot=True)
    return m.to(DEV), d


@pytest.mark.parametrize("case", H.ONEHOT_TRAIN_CASES)
def test_onehot_train_steps_match_reference(case):
    """zero_grad -> training_losses -> mean -> backward -> AdamW.step with the reference's randomness injected (both
    timestep draws, the sampled classes, noise, both dropout keep-masks)."""
    fx = H.load("onehot_train_" + case)
    meta = H.onehot_train_meta(fx)
    model, diff = gpu_pair(meta, fx)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=meta["lr"], weight_decay=meta["wd"])
    model.train()
    for s in range(meta["n_steps"]):
        inp = H.onehot_step_inputs(fx, s)
        # the discrete-noise kernel alone: kept bits are bit-exact
        xU, _ = model.engine.onehot_rows(cu(inp["x"]), None, cu(inp["sampled"]), meta["discrete"])
        np.testing.assert_array_equal(xU.cpu().numpy().reshape(meta["B"], meta["I"], 2).astype(np.uint8), fx[f"s{s}.x_tU"])
        opt.zero_grad()
        terms = diff.training_losses(model, cu(inp["x"]), True, ts=cu(inp["ts"]), pt=cu(inp["pt"]), noise=cu(inp["noise"]),
                                     drop_mask=cu(inp["drop_mask"]), ts_U=cu(inp["ts_U"]), sampled=cu(inp["sampled"]),
                                     drop_mask_U=cu(inp["drop_mask_U"]))
        assert terms["loss"].dtype == torch.float64 and terms["loss"].shape == (meta["B"],)
        loss = terms["loss"].mean()
        loss.backward()
        np.testing.assert_allclose(terms["loss"].detach().cpu().numpy(), fx[f"s{s}.loss_vec"], rtol=1e-4, atol=0)
        assert abs(float(loss.detach()) - float(fx[f"s{s}.loss"])) <= 1e-4 * abs(float(fx[f"s{s}.loss"]))
        if s == 0:
            for k, v in model.named_parameters():
                assert H.relerr(v.grad.cpu().numpy(), fx["g0." + k]) < 2e-4, k
        opt.step()
        np.testing.assert_array_equal(diff.Lt_count.cpu().numpy(), fx[f"s{s}.Lt_count"])
        np.testing.assert_allclose(diff.Lt_history.cpu().numpy(), fx[f"s{s}.Lt_history"], rtol=1e-4, atol=0)
    for k, v in model.named_parameters():
        d = np.abs(v.detach().cpu().numpy() - fx["pN." + k]).max()
        assert d < 0.02 * meta["lr"] * meta["n_steps"], (k, d)
        assert H.relerr(opt.state[v]["exp_avg"].cpu().numpy(), fx["m." + k]) < 2e-4, k
        assert H.relerr(opt.state[v]["exp_avg_sq"].cpu().numpy(), fx["v." + k]) < 4e-4, k


@pytest.mark.parametrize("case", H.ONEHOT_SAMPLE_CASES)
def test_onehot_p_sample_matches_reference(case):
    fx = H.load("onehot_sample_" + case)
    meta = H.onehot_sample_meta(fx)
    model, diff = gpu_pair(meta, fx)
    model.eval()
    x = cu(torch.from_numpy(fx["x_start"].astype(np.float32)))
    T = meta["T"]
    p0 = diff.p_sample(model, x, 0, False)
    assert H.relerr(p0.cpu().numpy(), fx["pred_steps0"]) < 2e-5
    pT = diff.p_sample(model, x, T, False, noise0=cu(torch.from_numpy(fx["noise_stepsT"])),
                       sampled0=cu(torch.from_numpy(fx["sampled_stepsT"])))
    assert H.relerr(pT.cpu().numpy(), fx["pred_stepsT"]) < 2e-5
    pn = diff.p_sample(model, x, 2, True, noise0=cu(torch.from_numpy(fx["noise_noisy0"])),
                       sampled0=cu(torch.from_numpy(fx["sampled_noisy0"])),
                       step_noise=cu(torch.from_numpy(fx["noise_noisy_steps"])))
    assert H.relerr(pn.cpu().numpy(), fx["pred_noisy"]) < 2e-5
    with pytest.raises(AssertionError):
        diff.p_sample(model, x, T + 1, False)
    with pytest.raises(TypeError):  # the one-hot path needs the one-hot backbone
        diff.p_sample(gdmcf_amd.DNN([meta["I"], 8], [8, meta["I"]], 10).to(DEV), x, 0, False)
    # a backbone whose forward keeps the reference's signature model(x, t, x_tU) (no `posterior=`): the reverse loop applies
    # the posterior element-wise instead of inside the output GEMM -- same results; `capture` holds the real means
    base = type(model)
    plain = type("PlainForward", (base,), {"forward": lambda self, a, t, u: base.forward(self, a, t, u)})
    model.__class__ = plain
    cap = {}
    pn2 = diff.p_sample(model, x, 2, True, noise0=cu(torch.from_numpy(fx["noise_noisy0"])),
                        sampled0=cu(torch.from_numpy(fx["sampled_noisy0"])),
                        step_noise=cu(torch.from_numpy(fx["noise_noisy_steps"])), capture=cap)
    assert H.relerr(pn2.cpu().numpy(), fx["pred_noisy"]) < 2e-5
    assert len(cap["mean"]) == T and all(m is not None and torch.isfinite(m).all() for m in cap["mean"])
    model.__class__ = base
    cap = {}
    pn3 = diff.p_sample(model, x, 2, True, noise0=cu(torch.from_numpy(fx["noise_noisy0"])),
                        sampled0=cu(torch.from_numpy(fx["sampled_noisy0"])),
                        step_noise=cu(torch.from_numpy(fx["noise_noisy_steps"])), capture=cap)
    assert H.relerr(pn3.cpu().numpy(), fx["pred_noisy"]) < 2e-5
    assert all(m is not None for m in cap["mean"]) and H.relerr(cap["mean"][-1].cpu().numpy(), pn3.cpu().numpy()) < 1e-6


def test_onehot_noise_kernel_statistics_and_determinism():
    """Device RNG path of gdmcf_onehot_noise_f32: P(class 1 | c0) = a*[c0 == 1] + (1 - a)*(1 - e), a = ts/B (the
    reference's own scaling); the true bit survives iff the draw reproduces the class.  Checked against the oracle's
    transition matrix on 2 x 10^6 draws (4 sigma), same (seed, offset) -> same draw, new offset -> new draw."""
    B, I, e = 16, 125_003, 0.9
    lib = gdmcf_amd._lib.load()
    g = torch.Generator().manual_seed(3)
    x0 = (torch.rand(B, I, generator=g) < 0.3).float().to(DEV)
    ts = torch.arange(B, dtype=torch.int64, device=DEV) % 7
    od = O.GaussianDiffusionDiscrete(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 7, discrete=e, CatOneHot=True)
    Q = od.get_Qt_bar(ts.cpu().float() / B)  # [B, 2, 2]

    def draw(offset):
        xU = torch.empty(B, 2 * I, device=DEV)
        s = torch.empty(B, I, dtype=torch.uint8, device=DEV)
        gdmcf_amd._lib.check(lib.gdmcf_onehot_noise_f32(x0.data_ptr(), x0.stride(0), ts.data_ptr(), B, I, e, None, 0, 1234,
                                                        offset, xU.data_ptr(), xU.stride(0), s.data_ptr(), s.stride(0),
                                                        gdmcf_amd._lib.stream_ptr()))
        return xU.view(B, I, 2), s

    xU, s = draw(1)
    xU2, s2 = draw(1)
    _, s3 = draw(2)
    assert torch.equal(s, s2) and torch.equal(xU, xU2) and not torch.equal(s, s3)
    c0 = x0 != 0
    keep = (s != 0) == c0
    assert torch.equal(xU[..., 0] != 0, keep & ~c0) and torch.equal(xU[..., 1] != 0, keep & c0)
    for b in (0, 3, 6, 15):
        for c in (0, 1):
            sel = c0[b] == bool(c)
            n = int(sel.sum())
            p = float(Q[b, c, 1])
            got = float((s[b][sel] != 0).float().mean())
            assert abs(got - p) < 4 * np.sqrt(p * (1 - p) / n) + 1e-6, (b, c, got, p)


def test_onehot_rng_path_trains_and_matches_oracle_on_its_own_draws():
    """No injected randomness: Philox noise / dropout / class draws inside the kernels.  The kernels' own draws are read
    back (sampled classes via the noise kernel's output, masks and noise by replaying with the same seeds is not needed):
    the loss must fall over a few steps and stay finite; parameter names / optimizer state as the reference's."""
    torch.manual_seed(0)
    I, hid, B, T = 300, 32, 64, 5
    out_dims = [hid, I]
    model = gdmcf_amd.DNNOneHot([I, hid], out_dims, 10).to(DEV).train()
    assert out_dims == [2 * hid, I]
    diff = gdmcf_amd.GaussianDiffusionDiscrete(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV, CatOneHot=True)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=2e-3, weight_decay=0.0)
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(B, I, generator=g) < 0.05).float().to(DEV)
    pt = torch.ones(B, dtype=torch.float64, device=DEV)
    ts0 = torch.zeros(B, dtype=torch.int64, device=DEV)  # t = 0: unit weight, so the loss is the plain mse
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = diff.training_losses(model, x, True, ts=ts0, pt=pt)["loss"].mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and losses[-1] < 0.7 * losses[0], losses[::5]
    assert [k for k, _ in model.named_parameters()][:6] == ["emb_layer.weight", "emb_layer.bias", "in_layers.0.weight",
                                                            "in_layers.0.bias", "in_layers2.0.weight", "in_layers2.0.bias"]


def test_onehot_full_width_step_matches_oracle():
    """Yelp-width rows (I = 34 395, hidden 1000 + 1000) at a small batch: one training step against the CPU oracle on the
    same injected randomness -- the [B, 2I] branch runs the same split-K GEMMs as the plain denoiser at K = 68 800."""
    torch.manual_seed(2)
    I, hid, B, T = 34395, 1000, 48, 5
    om = O.DNNOneHot([I, hid], [hid, I], 10)
    gm = gdmcf_amd.DNNOneHot([I, hid], [hid, I], 10)
    gm.load_state_dict(om.state_dict())
    gm = gm.to(DEV).train()
    om.train()
    od = O.GaussianDiffusionDiscrete(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, CatOneHot=True)
    gd_ = gdmcf_amd.GaussianDiffusionDiscrete(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV, CatOneHot=True)
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(B, I, generator=g) < 0.001).float()
    ts = torch.randint(0, T, (B,), generator=g)
    ts_U = torch.randint(0, T, (B,), generator=g)
    sampled = (torch.rand(B, I, generator=g) < 0.02).long()
    noise = torch.randn(B, I, generator=g)
    keep = (torch.rand(B, I, generator=g) < 0.5).float()
    keep_U = (torch.rand(B, 2 * I, generator=g) < 0.5).float()
    pt = torch.ones(B, dtype=torch.float64)
    ot = od.training_losses(om, x, True, ts=ts, pt=pt, noise=noise, drop_mask=keep, ts_U=ts_U, sampled=sampled,
                            drop_mask_U=keep_U)
    ot["loss"].mean().backward()
    gt = gd_.training_losses(gm, cu(x), True, ts=cu(ts), pt=cu(pt), noise=cu(noise), drop_mask=cu(keep), ts_U=cu(ts_U),
                             sampled=cu(sampled), drop_mask_U=cu(keep_U))
    gt["loss"].mean().backward()
    np.testing.assert_allclose(gt["loss"].detach().cpu().numpy(), ot["loss"].detach().numpy(), rtol=1e-4, atol=0)
    for (k, a), (_, b) in zip(gm.named_parameters(), om.named_parameters()):
        assert H.relerr(a.grad.cpu().numpy(), b.grad.numpy()) < 3e-4, k


def test_apply_noise_and_qt_bar_by_name():
    """The reference's public methods of the discrete noise: get_Qt_bar equals the oracle's matrices exactly; apply_noise
    returns one-hot int64 [B, I, 2] whose class-1 rate follows row c0 of Q_bar(ts / B) (4 sigma on 10^5 draws per row)."""
    B, I, T = 8, 100_000, 5
    d = gdmcf_amd.GaussianDiffusionDiscrete(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV, discrete=0.8,
                                            CatOneHot=True)
    od = O.GaussianDiffusionDiscrete(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, discrete=0.8, CatOneHot=True)
    ts = torch.tensor([0, 1, 2, 3, 4, 4, 2, 0])
    Q = d.get_Qt_bar(cu(ts).float() / B)
    np.testing.assert_array_equal(Q.cpu().numpy(), od.get_Qt_bar(ts.float() / B).numpy())
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(B, I, generator=g) < 0.4).long()
    out = d.apply_noise(cu(ts), cu(torch.nn.functional.one_hot(x, 2).float()))
    assert out.shape == (B, I, 2) and out.dtype == torch.int64 and bool((out.sum(-1) == 1).all())
    s = out[..., 1].cpu()
    for b in range(B):
        for c in (0, 1):
            sel = x[b] == c
            p, n = float(Q[b, c, 1]), int(sel.sum())
            assert abs(float(s[b][sel].float().mean()) - p) < 4 * np.sqrt(p * (1 - p) / n) + 1e-6, (b, c)
    assert not torch.equal(out, d.apply_noise(cu(ts), cu(torch.nn.functional.one_hot(x, 2).float())))  # a new draw per call


# ---- indexIn backbone DNNOneHotEmbedding ---------------------------------------------------------------------------
def gpu_emb_pair(meta, fx):
    I, dims = meta["I"], meta["dims"]
    m = gdmcf_amd.DNNOneHotEmbedding([I] + dims, dims[::-1] + [I], 10, item_num=I, user_num=meta["U"])
    m.load_state_dict(H.state_dict_from(fx))
    mt = {"x0": ModelMeanType.START_X, "eps": ModelMeanType.EPSILON}[meta["mean_type"]]
    d = gdmcf_amd.GaussianDiffusionDiscrete(mt, meta["schedule"], meta["scale"], meta["nmin"], meta["nmax"], meta["T"], DEV,
                                            discrete=meta["discrete"], CatOneHot=True)
    d.indexIn = True  # main.py:241
    return m.to(DEV), d


@pytest.mark.parametrize("case", H.ONEHOT_EMB_CASES)
def test_onehot_embedding_backbone_matches_reference(case):
    """DNNOneHotEmbedding (cosine scores against the item table, user rows, NT-Xent term x 0.1) under
    GaussianDiffusionDiscrete(CatOneHot=True, indexIn): training steps with the reference's randomness injected, then
    p_sample on the trained weights, against the reference's own numbers."""
    fx = H.load("onehot_emb_" + case)
    meta = H.onehot_emb_meta(fx)
    model, diff = gpu_emb_pair(meta, fx)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=meta["lr"], weight_decay=meta["wd"])
    model.train()
    for s in range(meta["n_steps"]):
        inp = H.onehot_step_inputs(fx, s)
        opt.zero_grad()
        terms = diff.training_losses(model, cu(inp["x"]), True, index=torch.from_numpy(fx[f"s{s}.index"]), ts=cu(inp["ts"]),
                                     pt=cu(inp["pt"]), noise=cu(inp["noise"]), drop_mask=cu(inp["drop_mask"]),
                                     ts_U=cu(inp["ts_U"]), sampled=cu(inp["sampled"]), drop_mask_U=cu(inp["drop_mask_U"]))
        loss = terms["loss"].mean()
        loss.backward()
        assert abs(float(model.engine.last_closs) - float(fx[f"s{s}.closs"])) <= 2e-5 * abs(float(fx[f"s{s}.closs"]))
        np.testing.assert_allclose(terms["loss"].detach().cpu().numpy(), fx[f"s{s}.loss_vec"], rtol=1e-4, atol=0)
        assert abs(float(loss.detach()) - float(fx[f"s{s}.loss"])) <= 1e-4 * abs(float(fx[f"s{s}.loss"]))
        if s == 0:
            for k, v in model.named_parameters():
                if k.startswith("out_layers"):
                    assert v.grad is None  # never applied by this backbone (reference: no gradient either)
                else:
                    assert H.relerr(v.grad.cpu().numpy(), fx["g0." + k]) < 3e-4, k
        opt.step()
        np.testing.assert_array_equal(diff.Lt_count.cpu().numpy(), fx[f"s{s}.Lt_count"])
        np.testing.assert_allclose(diff.Lt_history.cpu().numpy(), fx[f"s{s}.Lt_history"], rtol=1e-4, atol=0)
    for k, v in model.named_parameters():
        d = np.abs(v.detach().cpu().numpy() - fx["pN." + k]).max()
        assert d < 0.02 * meta["lr"] * meta["n_steps"], (k, d)
    model.eval()
    x, idx = cu(torch.from_numpy(fx["e.x_start"].astype(np.float32))), torch.from_numpy(fx["e.index"])
    p0 = diff.p_sample(model, x, 0, False, index=idx)
    # (weights after two AdamW steps differ by ~1e-5 relative from the reference's: compare on that scale)
    assert H.relerr(p0.cpu().numpy(), fx["e.pred_steps0"]) < 2e-4
    pT = diff.p_sample(model, x, meta["T"], False, index=idx, noise0=cu(torch.from_numpy(fx["e.noise_stepsT"])),
                       sampled0=cu(torch.from_numpy(fx["e.sampled_stepsT"])))
    assert H.relerr(pT.cpu().numpy(), fx["e.pred_stepsT"]) < 2e-4
    with pytest.raises(RuntimeError):  # the embedding backbone cannot run without the users' ids
        diff.p_sample(model, x, 0, False)


def test_onehot_embedding_full_width_step_matches_oracle():
    """Yelp-width item table (I = 34 395, three 1000-wide blocks) and 54 574 users at a small batch: one training step
    against the CPU oracle on the same injected randomness."""
    torch.manual_seed(3)
    I, hid, B, T, U = 34395, 1000, 32, 5, 54574
    om = O.DNNOneHotEmbedding([I, hid], [hid, I], 10, item_num=I, user_num=U)
    gm = gdmcf_amd.DNNOneHotEmbedding([I, hid], [hid, I], 10, item_num=I, user_num=U)
    gm.load_state_dict(om.state_dict())
    gm = gm.to(DEV).train()
    om.train()
    od = O.GaussianDiffusionDiscrete(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, CatOneHot=True)
    gd_ = gdmcf_amd.GaussianDiffusionDiscrete(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV, CatOneHot=True)
    od.indexIn = gd_.indexIn = True
    g = torch.Generator().manual_seed(6)
    x = (torch.rand(B, I, generator=g) < 0.001).float()
    ts, ts_U = torch.randint(0, T, (B,), generator=g), torch.randint(0, T, (B,), generator=g)
    sampled = (torch.rand(B, I, generator=g) < 0.02).long()
    noise = torch.randn(B, I, generator=g)
    keep, keep_U = (torch.rand(B, I, generator=g) < 0.5).float(), (torch.rand(B, 2 * I, generator=g) < 0.5).float()
    index = torch.randperm(U, generator=g)[:B]
    pt = torch.ones(B, dtype=torch.float64)
    ot = od.training_losses(om, x, True, ts=ts, pt=pt, noise=noise, drop_mask=keep, ts_U=ts_U, sampled=sampled,
                            drop_mask_U=keep_U, index=index)
    ot["loss"].mean().backward()
    gt = gd_.training_losses(gm, cu(x), True, index=index, ts=cu(ts), pt=cu(pt), noise=cu(noise), drop_mask=cu(keep),
                             ts_U=cu(ts_U), sampled=cu(sampled), drop_mask_U=cu(keep_U))
    gt["loss"].mean().backward()
    np.testing.assert_allclose(gt["loss"].detach().cpu().numpy(), ot["loss"].detach().numpy(), rtol=1e-4, atol=0)
    for (k, a), (_, b) in zip(gm.named_parameters(), om.named_parameters()):
        if b.grad is None:
            assert a.grad is None, k
        else:
            assert H.relerr(a.grad.cpu().numpy(), b.grad.numpy()) < 5e-4, k


@pytest.mark.parametrize("backbone", ["onehot", "onehot-emb"])
def test_onehot_backbones_with_bf16_gemm_inputs(backbone):
    """gemm_dtype="bf16" on the one-hot backbones (operands rounded to bf16 on chip, f32 accumulate and state): one
    training step within bf16 rounding of the f32 path -- loss 1e-3, gradients a few 1e-2 of their max-norm."""
    torch.manual_seed(4)
    I, hid, B, T, U = 1500, 96, 64, 5, 300
    g = torch.Generator().manual_seed(8)
    x = (torch.rand(B, I, generator=g) < 0.03).float().to(DEV)
    rand = dict(ts=torch.randint(0, T, (B,), generator=g).to(DEV), pt=torch.ones(B, dtype=torch.float64, device=DEV),
                noise=torch.randn(B, I, generator=g).to(DEV), drop_mask=(torch.rand(B, I, generator=g) < 0.5).to(torch.uint8).to(DEV),
                sampled=(torch.rand(B, I, generator=g) < 0.02).to(torch.uint8).to(DEV),
                drop_mask_U=(torch.rand(B, 2 * I, generator=g) < 0.5).to(torch.uint8).to(DEV))
    if backbone == "onehot-emb":
        rand["index"] = torch.randperm(U, generator=g)[:B]
    res = {}
    for dtype in ("f32", "bf16", "f32x3"):
        torch.manual_seed(11)
        if backbone == "onehot":
            m = gdmcf_amd.DNNOneHot([I, hid], [hid, I], 10, gemm_dtype=dtype)
        else:
            m = gdmcf_amd.DNNOneHotEmbedding([I, hid], [hid, I], 10, item_num=I, user_num=U, gemm_dtype=dtype)
        m = m.to(DEV).train()
        d = gdmcf_amd.GaussianDiffusionDiscrete(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV, CatOneHot=True)
        d.indexIn = backbone == "onehot-emb"
        loss = d.training_losses(m, x, True, **rand)["loss"].mean()
        loss.backward()
        res[dtype] = (float(loss.detach()), [p.grad.clone() for p in m.parameters() if p.grad is not None])
    assert abs(res["bf16"][0] - res["f32"][0]) <= 2e-3 * abs(res["f32"][0]) and res["bf16"][0] != res["f32"][0]
    for a, b in zip(res["bf16"][1], res["f32"][1]):
        assert H.relerr(a.cpu().numpy(), b.cpu().numpy()) < 5e-2
    # "f32x3" (float32 products from three-term bf16 splits) is no rounding mode: f32-level agreement
    assert abs(res["f32x3"][0] - res["f32"][0]) <= 2e-6 * abs(res["f32"][0])
    for a, b in zip(res["f32x3"][1], res["f32"][1]):
        assert H.relerr(a.cpu().numpy(), b.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("layers", [2, 1, 0])
def test_gcn_backbone_matches_the_restated_oracle(layers):
    """DNNOneHotEmbeddingGCN (PARITY UNPINNED: torch_geometric's GCNConv is restated in the oracle from its published
    semantics): two training steps and p_sample against the oracle's FULL-GRAPH evaluation on the same injected randomness,
    with sumW away from its initial 1 so that the GCN branch carries weight -- loss, every gradient incl. the GCNConv
    parameters and sumW, weights after AdamW."""
    torch.manual_seed(7)
    I, hid, B, T, U = 210, 24, 20, 5, 70
    om = O.DNNOneHotEmbeddingGCN([I, hid], [hid, I], 10, item_num=I, user_num=U, gcn_layers=layers)
    with torch.no_grad():
        om.sumW.fill_(0.4)
        for k, p in om.named_parameters():
            if k.startswith("gcn_model") and k.endswith("bias"):
                p.normal_(0.0, 0.1)
    gm = gdmcf_amd.DNNOneHotEmbeddingGCN([I, hid], [hid, I], 10, item_num=I, user_num=U, gcn_layers=layers)
    gm.load_state_dict(om.state_dict())
    gm = gm.to(DEV).train()
    om.train()
    od = O.GaussianDiffusionDiscrete(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, CatOneHot=True)
    gd_ = gdmcf_amd.GaussianDiffusionDiscrete(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV, CatOneHot=True)
    od.indexIn = gd_.indexIn = True
    oo, go = O.make_optimizer(om, 1e-3, 0.01), gdmcf_amd.FusedAdamW(gm.parameters(), lr=1e-3, weight_decay=0.01)
    g = torch.Generator().manual_seed(9)
    for s in range(2):
        x = (torch.rand(B, I, generator=g) < 0.06).float()
        r = dict(ts=torch.randint(0, T, (B,), generator=g), pt=torch.ones(B, dtype=torch.float64),
                 noise=torch.randn(B, I, generator=g), drop_mask=(torch.rand(B, I, generator=g) < 0.5).float(),
                 ts_U=torch.randint(0, T, (B,), generator=g), sampled=(torch.rand(B, I, generator=g) < 0.05).long(),
                 drop_mask_U=(torch.rand(B, 2 * I, generator=g) < 0.5).float(), index=torch.randperm(U, generator=g)[:B])
        oo.zero_grad()
        ol = od.training_losses(om, x, True, **r)["loss"]
        ol.mean().backward()
        go.zero_grad()
        gl = gd_.training_losses(gm, cu(x), True, **{k: (v if k == "index" else cu(v)) for k, v in r.items()})["loss"]
        gl.mean().backward()
        np.testing.assert_allclose(gl.detach().cpu().numpy(), ol.detach().numpy(), rtol=1e-4, atol=0)
        for (k, a), (_, b) in zip(gm.named_parameters(), om.named_parameters()):
            if b.grad is None:
                assert a.grad is None, k
            else:
                scale = max(float(b.grad.abs().max()), 1e-30)
                assert float((a.grad.cpu() - b.grad).abs().max()) < 3e-4 * scale + 1e-9, (k, s)
        oo.step()
        go.step()
    for (k, a), (_, b) in zip(gm.named_parameters(), om.named_parameters()):
        assert float((a.detach().cpu() - b.detach()).abs().max()) < 0.02 * 1e-3 * 2, k
    gm.eval()
    om.eval()
    x = (torch.rand(B, I, generator=g) < 0.06).float()
    idx = torch.randperm(U, generator=g)[:B]
    with torch.no_grad():
        want = od.p_sample(om, x, 0, False, sampled0=torch.zeros(1), index=idx)
    got = gd_.p_sample(gm, cu(x), 0, False, index=idx)
    assert H.relerr(got.cpu().numpy(), want.numpy()) < 2e-4


@pytest.mark.parametrize("case", ["plain", "guided", "guided_T9"])
def test_degree_guided_graph_of_the_reverse_loop_matches_reference(case):
    """GaussianDiffusionDiscrete.p_sample with indexIn: the per-step degree-guided graph (reference :706-744) built on the
    device by gdmcf_graph_guided_step_u8.  With the reference's draws injected (classes per item, one bit per user) the graph
    handed to the model at every reverse step equals the one the reference handed to its model, bit for bit."""
    fx = H.load("graph_guided_" + case)
    B, I, T, guided, scale, disc, _ = str(fx["meta"][0]).split("|")
    B, I, T, guided = int(B), int(I), int(T), bool(int(guided))

    class A:
        user_guided = guided

    d = gdmcf_amd.GaussianDiffusionDiscrete(gdmcf_amd.ModelMeanType.START_X, "linear-var", float(scale), 0.001, 0.01, T, DEV,
                                            discrete=float(disc), CatOneHot=True, args=A())
    d.indexIn = True
    torch.manual_seed(0)
    model = gdmcf_amd.DNNOneHotEmbedding([I, 16], [16, I], 10, item_num=I, user_num=B).to(DEV).eval()
    seen = []
    inner = model.forward

    def spy(x, t, x_U, index=None, graph=None, **kw):
        seen.append(graph.clone())
        return inner(x, t, x_U, index=index, graph=graph, **kw)

    model.forward = spy
    x = torch.from_numpy(fx["x_start"].astype(np.float32)).to(DEV)
    cap = {}
    d.p_sample(model, x, 0, False, index=torch.arange(B), capture=cap, graph_sampled=torch.from_numpy(fx["sampled"]).to(DEV),
               graph_pick=torch.from_numpy(fx["pick"]).to(DEV))
    assert len(seen) == T and seen[0].dtype == torch.uint8
    np.testing.assert_array_equal(torch.stack(seen).cpu().numpy(), fx["graph"])
    np.testing.assert_array_equal(d.last_graph.cpu().numpy(), fx["graph"][-1])
    np.testing.assert_array_equal(torch.stack(cap["graph"]).cpu().numpy(), fx["graph"])


def test_degree_guided_graph_draws_follow_the_transition_rows():
    """Un-injected: the in-kernel Philox draws of gdmcf_graph_guided_step_u8 follow the reference's probabilities --
    P(edge appears) = (1-a)(1-e), P(edge stays) = a + (1-a)(1-e), a = t/B (:775), user bit ~ deg/maxdeg (:710-716) -- and an
    edge that is in the graph stays in it."""
    from gdmcf_amd import _lib
    B, I, e = 64, 20000, 0.9
    d = gdmcf_amd.GaussianDiffusionDiscrete(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 40, DEV,
                                            discrete=e, CatOneHot=True)
    g = torch.zeros(B, I, dtype=torch.uint8, device=DEV)
    g[:, ::2] = 1
    before = g.clone()
    t = torch.full((B,), 16, dtype=torch.int64, device=DEV)  # a = 16/64 = 0.25
    degp = torch.linspace(0, 1, B, device=DEV)
    sampled = torch.empty_like(g)
    pick = torch.empty(B, dtype=torch.uint8, device=DEV)
    d.user_guided = False
    d._graph_step(g, t, degp, sampled_out=sampled, pick_out=pick)
    s = sampled.float().cpu().numpy()
    a = 0.25
    assert abs(s[:, 1::2].mean() - (1 - a) * (1 - e)) < 0.002  # state 0 -> 1
    assert abs(s[:, ::2].mean() - (a + (1 - a) * (1 - e))) < 0.004  # state 1 -> 1
    np.testing.assert_array_equal(g.cpu().numpy(), (before | sampled).cpu().numpy())  # not user guided: every draw counts
    assert (g[:, ::2] == 1).all()
    # user bits: frequency follows the relative degree; guided: rows whose bit is 0 do not change
    d.user_guided = True
    picks = []
    for _ in range(200):
        g2 = before.clone()
        d._graph_step(g2, t, degp, sampled_out=sampled, pick_out=pick)
        picks.append(pick.float().cpu().numpy())
        unchanged = (g2 == before).all(dim=1).cpu().numpy()
        assert unchanged[pick.cpu().numpy() == 0].all()
    freq = np.mean(picks, axis=0)
    assert np.abs(freq - degp.cpu().numpy()).max() < 0.15 and freq[0] == 0.0 and freq[-1] == 1.0


def test_lightgcn_tables_are_handed_to_the_embedding_backbone():
    """the propagated LightGCN tables (HIP SpMM) initialise embedding_user / the user-facing columns of
    embedding_item of DNNOneHotEmbedding (reference models/DNN.py:1148-1149, :1263-1274); the score of (u, i) then carries
    <e_u, e_i> of the graph model, and the diffusion step trains on from there."""
    rng = np.random.default_rng(3)
    U, I, h = 96, 70, 16
    users, items = rng.integers(0, U, 900), rng.integers(0, I, 900)
    lg = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": items}, U, I, 2, h, device=DEV).to(DEV)
    torch.manual_seed(0)
    model = gdmcf_amd.DNNOneHotEmbedding([I, h], [h, I], 10, item_num=I, user_num=U).to(DEV)
    before_items = model.embedding_item.weight.detach().clone()
    fu, fi = model.load_lightgcn_embeddings(lg)
    A = O.lightgcn_norm_adj(users, items, U, I)
    ref_u, ref_i = O.lightgcn_propagate(A, lg.E0.weight.detach().cpu().numpy(), 2, U)[:2]
    np.testing.assert_allclose(fu.cpu().numpy(), ref_u, atol=2e-6)
    assert torch.equal(model.embedding_user.weight, fu) and torch.equal(model.embedding_item.weight[:, -h:], fi)
    assert torch.equal(model.embedding_item.weight[:, :-h], before_items[:, :-h])  # the [h, h_U] columns keep their init
    with pytest.raises(ValueError):
        model.load_lightgcn_embeddings(gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": items}, U, I, 1, 8, device=DEV).to(DEV))
    # one diffusion step on top of the handed-over tables: loss finite, both tables receive gradients
    d = gdmcf_amd.GaussianDiffusionDiscrete(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 5, DEV, CatOneHot=True)
    d.indexIn = True
    x = torch.from_numpy((rng.random((32, I)) < 0.1).astype(np.float32)).to(DEV)
    model.train()
    loss = d.training_losses(model, x, True, index=torch.arange(32))["loss"].mean()
    loss.backward()
    assert torch.isfinite(loss) and model.embedding_user.weight.grad.abs().sum() > 0 and model.embedding_item.weight.grad.abs().sum() > 0
