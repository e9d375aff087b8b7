"""Times the weight-gradient product with and without the AdamW epilogue (gdmcf_linear_bwd_weight_f32 / _adamw_f32, C ABI) at the
two Yelp weight shapes for several reduction lengths (batch sizes): a short reduction isolates the optimiser stream of the
fused kernel (K = 128: 0.07 ms of matrix time), the full one shows how much of it runs under the k loops.
    python tools/fused_probe.py [reps]"""
import sys, time
import torch
sys.path.insert(0, ".")
import os
from gdmcf_amd import _lib
if os.environ.get("GDMCF_PROBE_LIB"):  # a probe build (tools/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["GDMCF_PROBE_LIB"])
lib = _lib.load()
only = os.environ.get("PROBE_ONLY")  # e.g. "400" -> only that reduction length
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
I, H, E = 34395, 1000, 10
for name, N, K in (("dW2 [I,H]", I, H), ("dW1 [H,I+E]", H, I + E)):
    for B in (128, 256, 400):
        if only and str(B) != only:
            continue
        ldz, lda = (N + 63) // 64 * 64, (K + 63) // 64 * 64
        dZ = torch.randn(B, ldz, device=dev) * 0.1
        A = torch.randn(B, lda, device=dev)
        ldw = (K + 31) // 32 * 32 if os.environ.get("PROBE_LDW_ALIGN") else K  # rows of W / exp_avg / exp_avg_sq on 128-byte lines
        W = torch.randn(N, ldw, device=dev) * 0.05
        m = torch.zeros(N, ldw, device=dev)
        v = torch.zeros(N, ldw, device=dev)
        dW = torch.empty(N, ldw, device=dev)
        st = _lib.stream_ptr()
        def plain():
            _lib.check(lib.gdmcf_linear_bwd_weight_f32(dZ.data_ptr(), ldz, A.data_ptr(), lda, None, 0, B, N, K, dW.data_ptr(), ldw, None, 0, st))
        def fused():
            _lib.check(lib.gdmcf_linear_bwd_weight_adamw_f32(dZ.data_ptr(), ldz, A.data_ptr(), lda, None, 0, B, N, K, W.data_ptr(), ldw,
                                                             m.data_ptr(), v.data_ptr(), None, 1e-5, 0.9, 0.999, 1e-8, 0.0, 3, 1.0, st))
        out = []
        for fn in (plain, fused):
            for _ in range(150):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / reps)
        flop = 2.0 * B * N * K
        byt = 24.0 * N * K
        print(f"{name} B={B}: plain {out[0]:.4f} ms ({flop/out[0]/1e9:.1f} TF)  fused {out[1]:.4f} ms  (+{out[1]-out[0]:.4f}; optimiser stream alone at 6.3 TB/s = {byt/6.3e9:.4f} ms; fused stream rate if serial {byt/(out[1]-out[0])/1e9:.2f} TB/s)", flush=True)
