"""gemm_dtype="f32x3": float32 products assembled from six bf16 MFMAs of three-term operand splits (gemm_split.hip).
The mode claims f32-LEVEL error, so it is measured against float64 beside the native f32 MFMA kernels and held to the
native path's tolerances everywhere (see also test_linear_entry_points_random_shapes[f32x3] and
test_amazon_book_shape_step_matches_oracle[f32x3])."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gdmcf_amd
import helpers as H
from gdmcf_amd import _lib
from gdmcf_amd.gaussian_diffusion import ModelMeanType
from oracle import gdmcf_oracle as O

DEV = torch.device("cuda:0")
cu = lambda t: t.to(DEV)


def _errors(got, want):
    err = got.double() - want
    return float(err.abs().max()) / float(want.abs().max()), float((err ** 2).mean().sqrt()) / float((want ** 2).mean().sqrt())


def test_split_products_carry_f32_level_error():
    """The four dense products of a training step (forward K = 9 010 split-K, fused-loss forward, input gradient, weight
    gradient) at B = 400, hidden 1000: error against float64 of the f32x3 mode vs the native v_mfma_f32_16x16x4_f32 mode.
    The split never rounds an operand (a = a0 + a1 + a2 exactly); what it drops is below one ulp of each product."""
    lib = _lib.load()
    st = _lib.stream_ptr()
    B, I, Hd = 400, 9000, 1000
    g = torch.Generator().manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g).to(DEV)
    K1 = I + 10
    xin, W1, b1 = rn(B, K1), rn(Hd, K1) / K1 ** 0.5, rn(Hd)
    h, W2, b2 = torch.tanh(rn(B, Hd)), rn(I, Hd) / Hd ** 0.5, rn(I)
    tgt, alpha, rs = (torch.rand(B, I, generator=g) < 0.01).float().to(DEV), torch.ones(B, device=DEV), torch.ones(B, device=DEV)
    dz = rn(B, I) * 1e-3
    ws_bytes = max(int(lib.gdmcf_linear_ws_bytes(B, Hd, K1)), int(lib.gdmcf_linear_ws_bytes(B, I, Hd)))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    D = lambda t: t.double()
    want = {"fwd": D(xin) @ D(W1).T + D(b1), "loss": D(h) @ D(W2).T + D(b2), "dh": D(dz) @ D(W2), "dW": D(dz).T @ D(h)}
    res = {}
    for mode, code in (("f32", 0), ("f32x3", 2)):
        prev = lib.gdmcf_gemm_precision(code)
        try:
            C1, out, diff = torch.empty(B, Hd, device=DEV), torch.empty(B, I, device=DEV), torch.empty(B, I, device=DEV)
            rowpart, rowsum = torch.zeros(B, lib.gdmcf_loss_tiles(I), device=DEV), torch.zeros(B, device=DEV)
            dA, dW, db = torch.empty(B, Hd, device=DEV), torch.empty(I, Hd, device=DEV), torch.empty(I, device=DEV)
            _lib.check(lib.gdmcf_linear_fwd_f32(xin.data_ptr(), K1, W1.data_ptr(), K1, b1.data_ptr(), 0, B, Hd, K1, C1.data_ptr(), Hd,
                                                ws.data_ptr(), ws_bytes, st))
            _lib.check(lib.gdmcf_linear_loss_fwd_f32(h.data_ptr(), Hd, W2.data_ptr(), Hd, b2.data_ptr(), tgt.data_ptr(), I,
                                                     alpha.data_ptr(), B, I, Hd, out.data_ptr(), I, diff.data_ptr(), I,
                                                     rowpart.data_ptr(), rowsum.data_ptr(), st))
            _lib.check(lib.gdmcf_linear_bwd_input_f32(dz.data_ptr(), I, W2.data_ptr(), Hd, rs.data_ptr(), h.data_ptr(), Hd, 0, B, I, Hd,
                                                      dA.data_ptr(), Hd, ws.data_ptr(), ws_bytes, st))
            _lib.check(lib.gdmcf_linear_bwd_weight_f32(dz.data_ptr(), I, h.data_ptr(), Hd, rs.data_ptr(), 0, B, I, Hd, dW.data_ptr(), Hd,
                                                       db.data_ptr(), 0, st))
            res[mode] = {"fwd": _errors(C1, want["fwd"]), "loss": _errors(out, want["loss"]), "dh": _errors(dA, want["dh"]),
                         "dW": _errors(dW, want["dW"])}
        finally:
            lib.gdmcf_gemm_precision(prev)
    for k in want:
        (mx_n, rms_n), (mx_s, rms_s) = res["f32"][k], res["f32x3"][k]
        print(f"{k:5s} native max {mx_n:.2e} rms {rms_n:.2e} | f32x3 max {mx_s:.2e} rms {rms_s:.2e}")
        assert rms_s <= max(2.0 * rms_n, 1.5e-7), (k, rms_n, rms_s)
        assert mx_s <= max(3.0 * mx_n, 1e-6), (k, mx_n, mx_s)


def test_split_mode_trains_like_the_f32_oracle():
    """Three training steps (injected noise / masks) against the CPU oracle at the f32 tolerances of the native path, then
    p_sample + masked top-k: same index lists as the native f32 mode wherever the k/(k+1) score gap exceeds f32 noise."""
    B, I, hid, T = 96, 5003, 384, 5
    torch.manual_seed(3)
    om = O.DNN([I, hid], [hid, I], 10)
    models = {}
    for mode in ("f32", "f32x3"):
        m = gdmcf_amd.DNN([I, hid], [hid, I], 10, gemm_dtype=mode)
        m.load_state_dict(om.state_dict())
        models[mode] = m.to(DEV).train()
    od = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.1, 0.001, 0.01, T)
    oopt = O.make_optimizer(om, 1e-5)
    gds = {k: gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.1, 0.001, 0.01, T, DEV) for k in models}
    opts = {k: gdmcf_amd.FusedAdamW(m.parameters(), lr=1e-5, weight_decay=0.0) for k, m in models.items()}
    g = torch.Generator().manual_seed(5)
    for step in range(3):
        x = (torch.rand(B, I, generator=g) < 0.01).float()
        ts = torch.randint(0, T, (B,), generator=g)
        noise = torch.randn(B, I, generator=g)
        keep = (torch.rand(B, I, generator=g) < 0.5).float()
        pt = torch.ones(B)
        oloss, ovec = O.train_step(od, om.train(), oopt, x, True, ts=ts, pt=pt, noise=noise, drop_mask=keep)
        for k, m in models.items():
            opts[k].zero_grad()
            terms = gds[k].training_losses(m, cu(x), True, ts=cu(ts), pt=cu(pt), noise=cu(noise), drop_mask=cu(keep))
            terms["loss"].mean().backward()
            np.testing.assert_allclose(terms["loss"].detach().cpu().numpy(), ovec.numpy(), rtol=1e-4, err_msg=k)
            for (name, p), (_, q) in zip(m.named_parameters(), om.named_parameters()):
                assert H.relerr(p.grad.cpu().numpy(), q.grad.numpy()) < 2e-4, (k, step, name)
            opts[k].step()
    # evaluation path
    for m in models.values():
        m.eval()
    x = (torch.rand(64, I, generator=g) < 0.01).float()
    with torch.no_grad():
        scores = {k: gds[k].p_sample(m, cu(x), 2, False) for k, m in models.items()}
    a, b = scores["f32"], scores["f32x3"]
    assert H.relerr(b.cpu().numpy(), a.cpu().numpy()) < 2e-5
    k = 20
    va, ia = torch.topk(a, k + 1)
    ib = torch.topk(b, k)[1]
    noise_floor = 4e-6 * float(a.abs().max())
    safe = (va[:, k - 1] - va[:, k]) > noise_floor
    assert int(safe.sum()) > 32
    for r in torch.nonzero(safe).flatten().tolist():
        assert set(ia[r, :k].tolist()) == set(ib[r].tolist()), r
