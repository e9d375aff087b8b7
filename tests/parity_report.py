"""Parity numbers in one table (what the GPU tests assert, measured): HIP path vs the committed outputs of the real
reference (tests/golden) and vs the CPU oracle.  Test infrastructure (it uses the oracle as the checker, like the tests).  Run on the MI355X:  python tests/parity_report.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gdmcf_amd  # noqa: E402
from gdmcf_amd import ModelMeanType  # noqa: E402
from oracle import gdmcf_oracle as O  # noqa: E402  (checker)
from tests import helpers as H  # noqa: E402

DEV = "cuda:0"
cu = lambda t: t.to(DEV)


def model_of(meta, fx, dtype="f32"):
    I, dims = meta["I"], meta["dims"]
    m = gdmcf_amd.DNN([I] + dims, dims[::-1] + [I], meta.get("emb", 10), norm=meta.get("norm", False), gemm_dtype=dtype)
    m.load_state_dict(H.state_dict_from(fx))
    return m.to(DEV)


def diff_of(meta):
    mt = {"x0": ModelMeanType.START_X, "eps": ModelMeanType.EPSILON}[meta["mean_type"]]
    return gdmcf_amd.GaussianDiffusion(mt, meta.get("schedule", "linear-var"), meta["scale"], meta["nmin"], meta["nmax"],
                                       meta["T"], DEV)


print("== training steps vs the reference's own numbers (golden fixtures) ==")
print(f"{'case':16s} {'steps':>5s} {'max rel err loss':>18s} {'max rel err row loss':>22s} {'grad rel-L2 (step 0)':>22s} {'q_sample':>10s}")
for case in H.TRAIN_CASES:
    fx = H.load("train_" + case)
    meta = H.train_meta(fx)
    model, diff = model_of(meta, fx).train(), diff_of(meta)
    diff.Lt_history.copy_(torch.from_numpy(fx["Lt_history0"]))
    diff.Lt_count.copy_(torch.from_numpy(fx["Lt_count0"]))
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=meta["lr"], weight_decay=meta["wd"])
    e_loss = e_row = e_grad = 0.0
    inp0 = H.step_inputs(fx, 0)
    qs = bool(np.array_equal(diff.q_sample(cu(inp0["x"]), cu(inp0["ts"]), cu(inp0["noise"])).cpu().numpy(), fx["s0.x_t"]))
    for s in range(meta["n_steps"]):
        inp = H.step_inputs(fx, s)
        opt.zero_grad()
        terms = diff.training_losses(model, cu(inp["x"]), True, ts=cu(inp["ts"]), pt=cu(inp["pt"]), noise=cu(inp["noise"]),
                                     drop_mask=cu(inp["drop_mask"]))
        loss = terms["loss"].mean()
        loss.backward()
        lv = terms["loss"].detach().cpu().numpy()
        e_loss = max(e_loss, abs(float(loss.detach()) - float(fx[f"s{s}.loss"])) / abs(float(fx[f"s{s}.loss"])))
        e_row = max(e_row, float(np.max(np.abs(lv - fx[f"s{s}.loss_vec"]) / np.abs(fx[f"s{s}.loss_vec"]))))
        if s == 0:
            e_grad = max(H.relerr(v.grad.cpu().numpy(), fx["g0." + k]) for k, v in model.named_parameters())
        opt.step()
    print(f"{case:16s} {meta['n_steps']:5d} {e_loss:18.2e} {e_row:22.2e} {e_grad:22.2e} {'bit-exact' if qs else 'DIFFERS':>10s}")

print("\n== reverse diffusion + masked top-k + metrics vs the reference (golden fixtures) ==")
for case in H.SAMPLE_CASES:
    fx = H.load("sample_" + case)
    meta = H.sample_meta(fx)
    model, diff = model_of(meta, fx).eval(), diff_of(meta)
    x = cu(torch.from_numpy(fx["x_start"].astype(np.float32)))
    p0 = diff.p_sample(model, x, 0, False)
    his = torch.from_numpy(fx["x_start"].astype(np.float32)).to_sparse_csr()
    idx = gdmcf_amd.masked_topk(p0, meta["k"], his.crow_indices().to(DEV), his.col_indices().to(DEV)).cpu().numpy()
    same = sum(set(idx[b].tolist()) == set(fx["topk_idx"][b].tolist()) for b in range(meta["B"]))
    gt = [fx["gt_flat"][a:b].tolist() for a, b in zip(fx["gt_ptr"][:-1], fx["gt_ptr"][1:])]
    res = np.array(gdmcf_amd.computeTopNAccuracy(gt, idx.tolist(), fx["topN"].tolist()))
    print(f"{case:12s} prediction rel err {H.relerr(p0.cpu().numpy(), fx['pred_steps0']):.2e};  top-{meta['k']} index sets equal "
          f"on {same}/{meta['B']} rows;  metrics (P/R/NDCG/MRR @ {fx['topN'].tolist()}) equal: {bool(np.array_equal(res, fx['metrics']))}")

print("\n== full Yelp-shape step (B=400, I=34395, dims=[1000]) vs the CPU oracle, fp32 and bf16 GEMM inputs ==")
B, I, hid, T = 400, 34395, 1000, 5
torch.manual_seed(0)
om = O.DNN([I, hid], [hid, I], 10).train()
od = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T)
g = torch.Generator().manual_seed(1)
x = (torch.rand(B, I, generator=g) < 0.00075).float()
ts = torch.randint(0, T, (B,), generator=g)
noise = torch.randn(B, I, generator=g)
keep = (torch.rand(B, I, generator=g) < 0.5).float()
oloss, ovec = O.train_step(od, om, O.make_optimizer(om, 0.0), x, True, ts=ts, pt=torch.ones(B), noise=noise, drop_mask=keep)
for dtype in ("f32", "bf16"):
    m = gdmcf_amd.DNN([I, hid], [hid, I], 10, gemm_dtype=dtype)
    m.load_state_dict(om.state_dict())
    m = m.to(DEV).train()
    d = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV)
    terms = d.training_losses(m, cu(x), True, ts=cu(ts), pt=cu(torch.ones(B)), noise=cu(noise), drop_mask=cu(keep))
    gl = terms["loss"].mean()
    gl.backward()
    eg = max(H.relerr(p.grad.cpu().numpy(), q.grad.numpy()) for p, q in zip(m.parameters(), om.parameters()))
    print(f"{dtype:5s} loss {float(gl.detach()):.8f} (oracle {float(oloss):.8f}, rel {abs(float(gl.detach()) - float(oloss)) / float(oloss):.2e});  "
          f"worst gradient rel-L2 {eg:.2e}")

print("\n== one-hot variant (GaussianDiffusionDiscrete(CatOneHot=True) + DNNOneHot) vs the reference (golden fixtures) ==")
print(f"{'case':16s} {'steps':>5s} {'max rel err loss':>18s} {'max rel err row loss':>22s} {'grad rel-L2 (step 0)':>22s} {'kept bits':>10s}")
for case in H.ONEHOT_TRAIN_CASES:
    fx = H.load("onehot_train_" + case)
    meta = H.onehot_train_meta(fx)
    I, dims = meta["I"], meta["dims"]
    model = gdmcf_amd.DNNOneHot([I] + dims, dims[::-1] + [I], meta.get("emb", 10), norm=meta.get("norm", False))
    model.load_state_dict(H.state_dict_from(fx))
    model = model.to(DEV).train()
    mt = {"x0": ModelMeanType.START_X, "eps": ModelMeanType.EPSILON}[meta["mean_type"]]
    diff = gdmcf_amd.GaussianDiffusionDiscrete(mt, meta["schedule"], meta["scale"], meta["nmin"], meta["nmax"], meta["T"], DEV,
                                               discrete=meta["discrete"], CatOneHot=True)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=meta["lr"], weight_decay=meta["wd"])
    e_loss = e_row = e_grad = 0.0
    bits = True
    for s in range(meta["n_steps"]):
        inp = H.onehot_step_inputs(fx, s)
        xU, _ = model.engine.onehot_rows(cu(inp["x"]), None, cu(inp["sampled"]), meta["discrete"])
        bits = bits and np.array_equal(xU.cpu().numpy().reshape(meta["B"], I, 2).astype(np.uint8), fx[f"s{s}.x_tU"])
        opt.zero_grad()
        terms = diff.training_losses(model, cu(inp["x"]), True, ts=cu(inp["ts"]), pt=cu(inp["pt"]), noise=cu(inp["noise"]),
                                     drop_mask=cu(inp["drop_mask"]), ts_U=cu(inp["ts_U"]), sampled=cu(inp["sampled"]),
                                     drop_mask_U=cu(inp["drop_mask_U"]))
        loss = terms["loss"].mean()
        loss.backward()
        lv = terms["loss"].detach().cpu().numpy()
        e_loss = max(e_loss, abs(float(loss.detach()) - float(fx[f"s{s}.loss"])) / abs(float(fx[f"s{s}.loss"])))
        e_row = max(e_row, float(np.max(np.abs(lv - fx[f"s{s}.loss_vec"]) / np.abs(fx[f"s{s}.loss_vec"]))))
        if s == 0:
            e_grad = max(H.relerr(v.grad.cpu().numpy(), fx["g0." + k]) for k, v in model.named_parameters())
        opt.step()
    print(f"{case:16s} {meta['n_steps']:5d} {e_loss:18.2e} {e_row:22.2e} {e_grad:22.2e} {'bit-exact' if bits else 'DIFFER':>10s}")
for case in H.ONEHOT_SAMPLE_CASES:
    fx = H.load("onehot_sample_" + case)
    meta = H.onehot_sample_meta(fx)
    I, dims = meta["I"], meta["dims"]
    model = gdmcf_amd.DNNOneHot([I] + dims, dims[::-1] + [I], 10)
    model.load_state_dict(H.state_dict_from(fx))
    model = model.to(DEV).eval()
    mt = {"x0": ModelMeanType.START_X, "eps": ModelMeanType.EPSILON}[meta["mean_type"]]
    diff = gdmcf_amd.GaussianDiffusionDiscrete(mt, "linear-var", meta["scale"], meta["nmin"], meta["nmax"], meta["T"], DEV,
                                               discrete=meta["discrete"], CatOneHot=True)
    x = cu(torch.from_numpy(fx["x_start"].astype(np.float32)))
    p0 = diff.p_sample(model, x, 0, False)
    pT = diff.p_sample(model, x, meta["T"], False, noise0=cu(torch.from_numpy(fx["noise_stepsT"])),
                       sampled0=cu(torch.from_numpy(fx["sampled_stepsT"])))
    print(f"{case:12s} p_sample rel err: steps=0 {H.relerr(p0.cpu().numpy(), fx['pred_steps0']):.2e}, "
          f"steps=T {H.relerr(pT.cpu().numpy(), fx['pred_stepsT']):.2e}")

print("\n== indexIn backbone DNNOneHotEmbedding vs the reference (golden fixtures) ==")
for case in H.ONEHOT_EMB_CASES:
    fx = H.load("onehot_emb_" + case)
    meta = H.onehot_emb_meta(fx)
    I, dims = meta["I"], meta["dims"]
    model = gdmcf_amd.DNNOneHotEmbedding([I] + dims, dims[::-1] + [I], 10, item_num=I, user_num=meta["U"])
    model.load_state_dict(H.state_dict_from(fx))
    model = model.to(DEV).train()
    mt = {"x0": ModelMeanType.START_X, "eps": ModelMeanType.EPSILON}[meta["mean_type"]]
    diff = gdmcf_amd.GaussianDiffusionDiscrete(mt, meta["schedule"], meta["scale"], meta["nmin"], meta["nmax"], meta["T"], DEV,
                                               discrete=meta["discrete"], CatOneHot=True)
    diff.indexIn = True
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=meta["lr"], weight_decay=meta["wd"])
    e_loss = e_row = e_grad = e_closs = 0.0
    for s in range(meta["n_steps"]):
        inp = H.onehot_step_inputs(fx, s)
        opt.zero_grad()
        terms = diff.training_losses(model, cu(inp["x"]), True, index=torch.from_numpy(fx[f"s{s}.index"]), ts=cu(inp["ts"]),
                                     pt=cu(inp["pt"]), noise=cu(inp["noise"]), drop_mask=cu(inp["drop_mask"]),
                                     ts_U=cu(inp["ts_U"]), sampled=cu(inp["sampled"]), drop_mask_U=cu(inp["drop_mask_U"]))
        loss = terms["loss"].mean()
        loss.backward()
        lv = terms["loss"].detach().cpu().numpy()
        e_loss = max(e_loss, abs(float(loss.detach()) - float(fx[f"s{s}.loss"])) / abs(float(fx[f"s{s}.loss"])))
        e_row = max(e_row, float(np.max(np.abs(lv - fx[f"s{s}.loss_vec"]) / np.abs(fx[f"s{s}.loss_vec"]))))
        e_closs = max(e_closs, abs(float(model.engine.last_closs) - float(fx[f"s{s}.closs"])) / abs(float(fx[f"s{s}.closs"])))
        if s == 0:
            e_grad = max(H.relerr(v.grad.cpu().numpy(), fx["g0." + k]) for k, v in model.named_parameters() if v.grad is not None)
        opt.step()
    print(f"{case:16s} steps {meta['n_steps']}: loss {e_loss:.2e}, row loss {e_row:.2e}, NT-Xent term {e_closs:.2e}, "
          f"gradients (step 0) {e_grad:.2e}")
