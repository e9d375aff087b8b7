"""Reference point: the same training step written in plain PyTorch on the same GPU (nn.Linear, F.dropout,
torch.optim.AdamW(fused=True / foreach), float32, TF32 off) -- what running the reference's model on PyTorch-ROCm as it is
would cost per step.  Timing only (independent of oracle/ and of the parity tests)."""
import math
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

dev = "cuda:0"
torch.backends.cuda.matmul.allow_tf32 = False
B, I, H, E, T = 400, 34395, 1000, 10, 5


class Denoiser(nn.Module):
    def __init__(self):
        super().__init__()
        self.emb_layer = nn.Linear(E, E)
        self.inl = nn.Linear(I + E, H)
        self.outl = nn.Linear(H, I)

    def forward(self, x, t):
        half = E // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(half, device=x.device, dtype=torch.float32) / half)
        args = t[:, None].float() * freqs[None]
        temb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        h = torch.cat([F.dropout(x, 0.5, self.training), self.emb_layer(temb)], dim=-1)
        return self.outl(torch.tanh(self.inl(h)))


torch.manual_seed(0)
model = Denoiser().to(dev).train()
ab = torch.linspace(0.99999, 0.9999, T, device=dev, dtype=torch.float64)
sa, sb = ab.sqrt().float(), (1 - ab).sqrt().float()
w = torch.ones(T, device=dev, dtype=torch.float64)
x = (torch.rand(B, I, device=dev) < 0.00075).float()
for name, kw in (("foreach", dict(foreach=True)), ("fused", dict(fused=True)), ("fused + autocast bf16", dict(fused=True))):
    opt = torch.optim.AdamW(model.parameters(), lr=1e-5, weight_decay=0.0, **kw)
    amp = "autocast" in name

    def step():
        opt.zero_grad()
        ts = torch.randint(0, T, (B,), device=dev)
        noise = torch.randn_like(x)
        x_t = sa[ts][:, None] * x + sb[ts][:, None] * noise
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out = model(x_t, ts)
        mse = ((x - out.float()) ** 2).mean(dim=1)
        loss = (w[ts] * mse).mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    print(f"PyTorch eager, AdamW({name}): {ms:.3f} ms/step = {B / ms * 1e3:,.0f} users/s", flush=True)

# the optimiser alone: torch's fused AdamW next to gdmcf_amd.FusedAdamW on the same parameters / gradients
import os  # noqa: E402
import sys  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gdmcf_amd  # noqa: E402

for p in model.parameters():
    p.grad = torch.randn_like(p)
for name, opt in (("torch.optim.AdamW(fused=True)", torch.optim.AdamW(model.parameters(), lr=1e-5, weight_decay=0.0, fused=True)),
                  ("gdmcf_amd.FusedAdamW", gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-5, weight_decay=0.0))):
    for _ in range(5):
        opt.step()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30):
        opt.step()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 30
    n = sum(p.numel() for p in model.parameters())
    print(f"{name}: {ms:.3f} ms per step over {n / 1e6:.1f} M parameters = {28.0 * n / ms / 1e9:.2f} TB/s of 28 B/param")
