"""Reference point: the vendor GEMM library (rocBLAS / hipBLASLt through torch.matmul, float32, TF32 off) on the five
products of the Yelp-shape training step, next to this library's kernels (bench.py kernel averages).  Plain products
only -- no fused bias / tanh / loss / posterior / AdamW epilogues, no split-K reducer."""
import torch

dev = "cuda:0"
torch.backends.cuda.matmul.allow_tf32 = False
B, I, H, E = 400, 34395, 1000, 10
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
xin, W1, h, W2, diff = rnd(B, I + E), rnd(H, I + E), rnd(B, H), rnd(I, H), rnd(B, I)
cases = [("GEMM1  xin @ W1^T", lambda: xin @ W1.t(), 2.0 * B * H * (I + E)),
         ("GEMM2  h @ W2^T  ", lambda: h @ W2.t(), 2.0 * B * I * H),
         ("dh     diff @ W2 ", lambda: diff @ W2, 2.0 * B * I * H),
         ("dW2    diff^T @ h", lambda: diff.t() @ h, 2.0 * B * I * H),
         ("dW1    h^T @ xin ", lambda: h.t() @ xin, 2.0 * B * H * (I + E))]
for name, fn, flops in cases:
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 30
    print(f"{name}: {ms:.4f} ms  {flops / ms / 1e9:.1f} TFLOP/s", flush=True)
for dt in (torch.bfloat16,):
    a, b = xin.to(dt), W1.to(dt)
    for _ in range(5):
        a @ b.t()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30):
        a @ b.t()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 30
    print(f"GEMM1 with bf16 operands resident ({dt}): {ms:.4f} ms  {2.0 * B * H * (I + E) / ms / 1e9:.1f} TFLOP/s")
