"""One-off fuzz of the input builder (test infrastructure, not collected by pytest): widths around the 4096-column workgroup
span, embedding columns straddling it, odd embedding sizes, F.normalize on/off, against the CPU oracle's forward."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gdmcf_amd  # noqa: E402
from oracle import gdmcf_oracle as O  # noqa: E402  (checker)

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
n = 0
for I in (5, 1023, 1024, 4085, 4090, 4095, 4096, 4097, 8186, 8192, 12290):
    for emb in (10, 7, 2):
        for norm in (False, True):
            B, hid = 9, 12
            torch.manual_seed(I + emb)
            om = O.DNN([I, hid], [hid, I], emb, norm=norm)
            gm = gdmcf_amd.DNN([I, hid], [hid, I], emb, norm=norm)
            gm.load_state_dict(om.state_dict())
            gm = gm.to(DEV).train()
            om.train()
            x = torch.randn(B, I, generator=g)
            ts = torch.randint(0, 50, (B,), generator=g)
            keep = (torch.rand(B, I, generator=g) < 0.5).float()
            want = om(x, ts, keep)
            got = gm(x.to(DEV), ts.to(DEV), keep.to(DEV))
            err = float((got.detach().cpu() - want.detach()).abs().max() / want.detach().abs().max())
            assert err < 2e-5, (I, emb, norm, err)
            n += 1
print("fuzz_prep ok:", n, "configurations")
