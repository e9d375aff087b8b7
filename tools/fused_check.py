"""Exploratory: FusedAdamW.fuse_into_backward vs the separate optimiser pass, several shapes, unseeded Philox path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gdmcf_amd  # noqa: E402
from gdmcf_amd import ModelMeanType  # noqa: E402

DEV = "cuda:0"


def run(I, hid, B, fuse, steps, dtype="f32", lr=1e-5):
    torch.manual_seed(0)
    m = gdmcf_amd.DNN([I, hid], [hid, I], 10, gemm_dtype=dtype).to(DEV).train()
    d = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, 5, DEV)
    o = gdmcf_amd.FusedAdamW(m.parameters(), lr=lr, weight_decay=0.0)
    if fuse:
        o.fuse_into_backward(m)
    m.engine.manual_seed(1234)
    g = torch.Generator().manual_seed(1)
    losses = []
    for s in range(steps):
        x = (torch.rand(B, I, generator=g) < 0.01).float().to(DEV)
        ts = torch.randint(0, 5, (B,), generator=g).to(DEV)
        o.zero_grad()
        l = d.training_losses(m, x, True, ts=ts, pt=torch.ones(B, device=DEV))["loss"].mean()
        l.backward()
        o.step()
        losses.append(float(l.detach()))
    return m, o, losses


for (I, hid, B) in [(515, 100, 32), (1280, 256, 64), (34395, 1000, 400)]:
    steps = 6
    m0, o0, l0 = run(I, hid, B, False, steps)
    m1, o1, l1 = run(I, hid, B, True, steps)
    print(I, hid, B, "losses unfused", ["%.6f" % v for v in l0])
    print(I, hid, B, "losses fused  ", ["%.6f" % v for v in l1])
    for (k, a), (_, b) in zip(m0.named_parameters(), m1.named_parameters()):
        print("   ", k, "max|dw|/lr = %.3e" % (float((a - b).abs().max()) / 1e-5),
              "exp_avg rel %.2e" % float((o0.state[a]["exp_avg"] - o1.state[b]["exp_avg"]).norm() / o0.state[a]["exp_avg"].norm()),
              "step", o0.state[a]["step"], o1.state[b]["step"], flush=True)
