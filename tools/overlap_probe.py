"""Does an HBM-bound kernel (fused AdamW over 137.6 MB) co-run with an MFMA-bound GEMM (dW1) on two streams?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gdmcf_amd
from gdmcf_amd import _lib
lib = _lib.load()
dev = "cuda:0"
B, I, H = 400, 34395, 1000
dz = torch.randn(B, 1024, device=dev)
xin = torch.randn(B, 34432, device=dev)
dW = torch.empty(H, I + 10, device=dev)
p = torch.nn.Parameter(torch.randn(I, H, device=dev))
p.grad = torch.randn(I, H, device=dev)
opt = gdmcf_amd.FusedAdamW([p], lr=1e-5, weight_decay=0.0)
s2 = torch.cuda.Stream()

def gemm():
    _lib.check(lib.gdmcf_linear_bwd_weight_f32(dz.data_ptr(), dz.stride(0), xin.data_ptr(), xin.stride(0), None, 0, B, H, I + 10,
                                               dW.data_ptr(), dW.stride(0), None, 0, _lib.stream_ptr()))
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def both():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        opt.step()
    gemm()
    torch.cuda.current_stream().wait_stream(s2)
def serial():
    opt.step(); gemm()
print("gemm alone  %.4f ms" % timeit(gemm))
print("adamw alone %.4f ms" % timeit(lambda: opt.step()))
print("serial      %.4f ms" % timeit(serial))
print("2 streams   %.4f ms" % timeit(both))
