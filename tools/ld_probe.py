"""Do the products that READ the weights care whether a weight row starts on a 128-byte line?  Yelp shapes, batch 400: the hidden
layer (W1 [H, I+E], row = 137 620 B), the fused-loss output layer and the input-gradient product (W2 [I, H], row = 4 000 B), each
with the weight's leading dimension as PyTorch leaves it (= columns) and rounded up to 32 floats.  C ABI calls, HIP events.
    python tools/ld_probe.py [reps]"""
import sys, os
import torch
sys.path.insert(0, ".")
from gdmcf_amd import _lib
if os.environ.get("GDMCF_PROBE_LIB"):  # a probe build (tools/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["GDMCF_PROBE_LIB"])
lib = _lib.load()
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B, I, H, E = 400, 34395, 1000, 10
up = lambda n, a: (n + a - 1) // a * a
ldk, ldi, ldh = up(I + E, 64), up(I, 64), 1024
xin = torch.randn(B, ldk, device=dev)
hs = torch.randn(B, ldh, device=dev)
hs[:, H:] = 0
dz2 = torch.randn(B, ldi, device=dev) * 0.01
tgt = (torch.rand(B, ldi, device=dev) < 0.002).float()
b1, b2 = torch.zeros(H, device=dev), torch.zeros(I, device=dev)
ws_bytes = max(lib.gdmcf_linear_ws_bytes(B, H, I + E), lib.gdmcf_linear_ws_bytes(B, I, H), 1 << 20)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
h_out = torch.empty(B, ldh, device=dev)
diff = torch.empty(B, ldi, device=dev)
rowpart = torch.empty(B, lib.gdmcf_loss_tiles(I), device=dev)
rowsum = torch.empty(B, device=dev)
dh = torch.empty(B, ldh, device=dev)
st = _lib.stream_ptr()

def timed(fn):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

cases = [(I + E, H, False), (up(I + E, 32), up(H, 32), True)]
if os.environ.get("LD_SWEEP"):  # "ld1:ld2,ld1:ld2,..."
    cases = [(int(a), int(b), True) for a, b in (c.split(":") for c in os.environ["LD_SWEEP"].split(","))]
for ld1, ld2, pad in cases:
    W1 = torch.randn(H, ld1, device=dev) * 0.01
    W2 = torch.randn(I, ld2, device=dev) * 0.01
    W1[:, I + E:] = 0
    W2[:, H:] = 0
    fwd = lambda: _lib.check(lib.gdmcf_linear_fwd_f32(xin.data_ptr(), ldk, W1.data_ptr(), ld1, b1.data_ptr(), 1, B, H, I + E,
                                                      h_out.data_ptr(), ldh, ws.data_ptr(), ws_bytes, st))
    loss = lambda: _lib.check(lib.gdmcf_linear_loss_fwd_f32(hs.data_ptr(), ldh, W2.data_ptr(), ld2, b2.data_ptr(), tgt.data_ptr(), ldi,
                                                            None, B, I, H, None, 0, diff.data_ptr(), ldi, rowpart.data_ptr(),
                                                            rowsum.data_ptr(), st))
    bwd = lambda: _lib.check(lib.gdmcf_linear_bwd_input_f32(dz2.data_ptr(), ldi, W2.data_ptr(), ld2, None, hs.data_ptr(), ldh, 1,
                                                            B, I, H, dh.data_ptr(), ldh, ws.data_ptr(), ws_bytes, st))
    print(f"weight rows {'on 128-byte lines (ld %d / %d)' % (ld1, ld2) if pad else 'as PyTorch leaves them (ld %d / %d)' % (ld1, ld2)}:"
          f"  hidden fwd {timed(fwd):.4f} ms   loss fwd {timed(loss):.4f} ms   input grad {timed(bwd):.4f} ms", end="", flush=True)
    # the fused weight-gradient + AdamW products with W / exp_avg / exp_avg_sq on that leading dimension
    out = []
    for (N, K, ldw, dZ, ldz, A, lda) in ((I, H, ld2, dz2, ldi, hs, ldh), (H, I + E, ld1, h_out, ldh, xin, ldk)):
        W = torch.randn(N, ldw, device=dev) * 0.05
        m = torch.zeros(N, ldw, device=dev)
        v = torch.zeros(N, ldw, device=dev)
        fused = lambda: _lib.check(lib.gdmcf_linear_bwd_weight_adamw_f32(dZ.data_ptr(), ldz, A.data_ptr(), lda, None, 0, B, N, K,
                                                                         W.data_ptr(), ldw, m.data_ptr(), v.data_ptr(), None, 1e-5,
                                                                         0.9, 0.999, 1e-8, 0.0, 3, 1.0, st))
        out.append(timed(fused))
        del W, m, v
    print(f"   fused dW2 {out[0]:.4f} ms   fused dW1 {out[1]:.4f} ms", flush=True)
