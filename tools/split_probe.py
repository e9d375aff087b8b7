"""Time and accuracy of the dense products of one Yelp-shape training step in the three GEMM modes
(native f32 MFMA / bf16-rounded operands / three-term bf16 split = "f32x3"), through the C ABI.
    python tools/split_probe.py [I] [hid] [B]
Error columns: max |got - f64| / max |f64| and the RMS error relative to the RMS of the result."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from gdmcf_amd import _lib

I = int(sys.argv[1]) if len(sys.argv) > 1 else 34395
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 400
dev = torch.device("cuda:0")
lib = _lib.load()
st = _lib.stream_ptr()
g = torch.Generator(device="cpu").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
K1 = I + 10
xin, W1, b1 = rn(B, K1), rn(H, K1) / K1 ** 0.5, rn(H)
h, W2, b2 = torch.tanh(rn(B, H)), rn(I, H) / H ** 0.5, rn(I)
tgt, alpha = (torch.rand(B, I, generator=g) < 0.002).float().to(dev), torch.ones(B, device=dev)
dz, rs = rn(B, I) * 1e-3, torch.ones(B, device=dev)
dh = rn(B, H) * 1e-3
ws_bytes = max(int(lib.gdmcf_linear_ws_bytes(B, H, K1)), int(lib.gdmcf_linear_ws_bytes(B, I, H)))
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
C1, out, diff = torch.empty(B, H, device=dev), torch.empty(B, I, device=dev), torch.empty(B, I, device=dev)
rowpart, rowsum = torch.zeros(B, lib.gdmcf_loss_tiles(I), device=dev), torch.zeros(B, device=dev)
dA, dW2, db2, dW1, db1 = torch.empty(B, H, device=dev), torch.empty(I, H, device=dev), torch.empty(I, device=dev), \
    torch.empty(H, K1, device=dev), torch.empty(H, device=dev)
act = h

calls = {
    "fwd1 [B,I]x[I,H]": lambda: lib.gdmcf_linear_fwd_f32(xin.data_ptr(), K1, W1.data_ptr(), K1, b1.data_ptr(), 0, B, H, K1, C1.data_ptr(), H,
                                                    ws.data_ptr(), ws_bytes, st),
    "loss [B,H]x[H,I]": lambda: lib.gdmcf_linear_loss_fwd_f32(h.data_ptr(), H, W2.data_ptr(), H, b2.data_ptr(), tgt.data_ptr(), I, alpha.data_ptr(),
                                                          B, I, H, out.data_ptr(), I, diff.data_ptr(), I, rowpart.data_ptr(),
                                                          rowsum.data_ptr(), st),
    "dh   [B,I]x[I,H]": lambda: lib.gdmcf_linear_bwd_input_f32(dz.data_ptr(), I, W2.data_ptr(), H, rs.data_ptr(), act.data_ptr(), H, 0, B, I, H,
                                                           dA.data_ptr(), H, ws.data_ptr(), ws_bytes, st),
    "dW2  [I,B]x[B,H]": lambda: lib.gdmcf_linear_bwd_weight_f32(dz.data_ptr(), I, h.data_ptr(), H, rs.data_ptr(), 0, B, I, H, dW2.data_ptr(), H,
                                                            db2.data_ptr(), 0, st),
    "dW1  [H,B]x[B,I]": lambda: lib.gdmcf_linear_bwd_weight_f32(dh.data_ptr(), H, xin.data_ptr(), K1, rs.data_ptr(), 0, B, H, K1, dW1.data_ptr(), K1,
                                                            db1.data_ptr(), 0, st),
}
D = lambda t: t.double()
refs = {
    "fwd1 [B,I]x[I,H]": (lambda: C1, D(xin) @ D(W1).T + D(b1)),
    "loss [B,H]x[H,I]": (lambda: out, D(h) @ D(W2).T + D(b2)),
    "dh   [B,I]x[I,H]": (lambda: dA, D(dz) @ D(W2)),
    "dW2  [I,B]x[B,H]": (lambda: dW2, D(dz).T @ D(h)),
    "dW1  [H,B]x[B,I]": (lambda: dW1, D(dh).T @ D(xin)),
}
flops = 2.0 * B * I * H
print(f"shape: B={B} I={I} hid={H}")
print(f"{'product':18s} {'mode':6s} {'us':>8s} {'TFLOP/s':>8s} {'max err':>10s} {'rms err':>10s}")
tot = {}
for name, fn in calls.items():
    for mode, code in (("f32", 0), ("bf16", 1), ("f32x3", 2)):
        prev = lib.gdmcf_gemm_precision(code)
        try:
            for _ in range(3):
                _lib.check(fn())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 20
            for _ in range(n):
                _lib.check(fn())
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            got, want = refs[name][0]().double(), refs[name][1]
            err = got - want
            mx = float(err.abs().max()) / float(want.abs().max())
            rms = float((err ** 2).mean().sqrt()) / float((want ** 2).mean().sqrt())
            tot[mode] = tot.get(mode, 0.0) + us
            print(f"{name:18s} {mode:6s} {us:8.1f} {flops / us / 1e6:8.1f} {mx:10.2e} {rms:10.2e}", flush=True)
        finally:
            lib.gdmcf_gemm_precision(prev)
print("sum of the five products [us]:", {k: round(v, 1) for k, v in tot.items()})
