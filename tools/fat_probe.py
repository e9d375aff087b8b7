"""Times the output-layer products (fused loss / posterior epilogue) through the C ABI: fat-tile kernel (default) against the
LDS-tiled kernel (GDMCF_GEMM_DR=1), Yelp and Amazon-Book widths.   python tools/fat_probe.py [reps]"""
import os, sys
import torch
sys.path.insert(0, ".")
from gdmcf_amd import _lib
if os.environ.get("GDMCF_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["GDMCF_PROBE_LIB"])
lib = _lib.load()
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B, K = 400, 1000
for N in (34395, 94949):
    h = torch.randn(B, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; bias = torch.randn(N, device=dev)
    words = (N + 31) // 32
    packed = torch.randint(-2**31, 2**31 - 1, (B, words), dtype=torch.int32, device=dev)
    ldd = (N + 63) // 64 * 64
    diff = torch.empty(B, ldd, device=dev); rowpart = torch.zeros(B * lib.gdmcf_loss_tiles(N), device=dev); rowsum = torch.zeros(B, device=dev)
    xt = torch.randn(B, ldd, device=dev); xn = torch.empty(B, ldd, device=dev); c1 = torch.rand(B, device=dev); c2 = torch.rand(B, device=dev)
    st = _lib.stream_ptr()
    def loss():
        _lib.check(lib.gdmcf_linear_loss_fwd_bits_f32(h.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), packed.data_ptr(), words, None, B, N, K,
                                                      None, 0, diff.data_ptr(), ldd, rowpart.data_ptr(), rowsum.data_ptr(), st))
    def post():
        _lib.check(lib.gdmcf_linear_posterior_fwd_f32(h.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), xt.data_ptr(), ldd, c1.data_ptr(), c2.data_ptr(),
                                                      None, None, None, None, 0, B, N, K, xn.data_ptr(), ldd, None, 0, st))
    for name, fn in (("loss", loss), ("posterior", post)):
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"N={N} {name}: {ms:.4f} ms  {2.0*B*N*K/ms/1e9:.1f} TF  frac {2.0*B*N*K/ms/1e9/157.3:.3f}  (kernel family {lib.gdmcf_debug_last_gemm()})", flush=True)
