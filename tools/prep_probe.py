"""Time gdmcf_dnn_prep_input_f32 at the Yelp shape with its random streams switched on and off (where do the 44 us go?)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gdmcf_amd  # noqa: E402
from gdmcf_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
B, I, E = 400, 34395, 10
ld = (I + E + 63) // 64 * 64
x = (torch.rand(B, I, device=dev) < 0.001).float()
ts = torch.randint(0, 5, (B,), device=dev)
ca, cb = torch.rand(5, device=dev), torch.rand(5, device=dev)
w, b = torch.randn(E, E, device=dev), torch.randn(E, device=dev)
xin = torch.zeros(B, ld, device=dev)
temb, rn = torch.zeros(B, E, device=dev), torch.zeros(B, device=dev)
noise = torch.randn(B, I, device=dev)
keep = (torch.rand(B, I, device=dev) < 0.5).to(torch.uint8)
E_use = int(os.environ.get("PROBE_E", E))
for name, nm, dm in (("copy only", 0, 0), ("philox noise", 2, 0), ("philox dropout", 0, 2), ("both (training)", 2, 2),
                     ("explicit noise + mask", 1, 1)):
    def run():
        _lib.check(lib.gdmcf_dnn_prep_input_f32(x.data_ptr(), x.stride(0), ts.data_ptr(), ca.data_ptr() if nm else None,
                                                cb.data_ptr() if nm else None, nm, noise.data_ptr() if nm == 1 else None, I, dm,
                                                keep.data_ptr() if dm == 1 else None, I, 0.5, 1, 7, 0, w.data_ptr(), b.data_ptr(),
                                                E_use, B, I, xin.data_ptr(), ld, None, 0, temb.data_ptr(), rn.data_ptr(),
                                                _lib.stream_ptr()))
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:24s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
