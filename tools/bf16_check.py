"""Exploratory: how far does the bf16-input GEMM path move the training loss / gradients from the f32 path?
Run on the GPU box:  python tools/bf16_check.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gdmcf_amd  # noqa: E402
from gdmcf_amd import ModelMeanType  # noqa: E402

DEV = "cuda:0"


def run(I, hid, B, T, dtype, steps=3, seed=0, lr=1e-3, density=0.01):
    torch.manual_seed(seed)
    m = gdmcf_amd.DNN([I, hid], [hid, I], 10, gemm_dtype=dtype).to(DEV).train()
    d = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, DEV)
    o = gdmcf_amd.FusedAdamW(m.parameters(), lr=lr, weight_decay=0.0)
    g = torch.Generator().manual_seed(1)
    out = []
    for s in range(steps):
        x = (torch.rand(B, I, generator=g) < density).float().to(DEV)
        ts = torch.randint(0, T, (B,), generator=g).to(DEV)
        noise = torch.randn(B, I, generator=g).to(DEV)
        keep = (torch.rand(B, I, generator=g) < 0.5).float().to(DEV)
        o.zero_grad()
        l = d.training_losses(m, x, True, ts=ts, pt=torch.ones(B, device=DEV), noise=noise, drop_mask=keep)["loss"]
        l.mean().backward()
        grads = [p.grad.detach().clone() for p in m.parameters()]
        o.step()
        out.append((l.detach().double().cpu().numpy(), grads))
    return out


for (I, hid, B, T) in [(515, 100, 32, 5), (4099, 256, 64, 5), (34395, 1000, 400, 5)]:
    a = run(I, hid, B, T, "f32")
    b = run(I, hid, B, T, "bf16")
    for s, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        rel_mean = abs(lb.mean() - la.mean()) / abs(la.mean())
        rel_row = np.max(np.abs(lb - la) / np.maximum(np.abs(la), 1e-30))
        gerr = [float((y - x).norm() / x.norm().clamp_min(1e-30)) for x, y in zip(ga, gb)]
        print(f"I={I} hid={hid} B={B} step {s}: loss {la.mean():.6g} rel(mean) {rel_mean:.2e} max rel(row) {rel_row:.2e} "
              f"grad rel-L2 {['%.1e' % e for e in gerr]}", flush=True)
