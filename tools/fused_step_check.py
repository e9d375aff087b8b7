"""Yelp-shape training steps with AdamW inside the weight-gradient products against the separate pass: loss per step, every weight
and moment afterwards (the two must agree to rounding: same gradient bits, same gd_adam_elem)."""
import sys
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, ".")
import gdmcf_amd
from gdmcf_amd import data
from gdmcf_amd.data_utils import DeviceCSR
from gdmcf_amd.parallel import DataParallelStep
import os
dev = torch.device("cuda:0")
LR, WD, NS = float(os.environ.get("LR", "1e-3")), float(os.environ.get("WD", "0.01")), int(os.environ.get("NS", "5"))
B, hid, T = 400, 1000, 5
indptr, indices, I = data.synth_csr("yelp", n_rows=2 * B, seed=0)
dcsr = DeviceCSR(sp.csr_matrix((np.ones(len(indices), np.float32), indices, indptr), shape=(2 * B, I)), dev)
res = []
for fuse in (False, True):
    torch.manual_seed(0)
    model = gdmcf_amd.DNN([I, hid], [hid, I], 10, time_type="cat", norm=False).to(dev).train()
    diffusion = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, dev)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=LR, weight_decay=WD)
    if fuse:
        opt.fuse_into_backward(model)
    step = DataParallelStep(diffusion, model, opt)
    torch.manual_seed(99)
    model.engine.manual_seed(7)
    losses = [float(step(dcsr.batch(torch.arange((i % 2) * B, (i % 2 + 1) * B, device=dev)), True)) for i in range(NS)]
    torch.cuda.synchronize()
    res.append((losses, [p.detach().clone() for p in model.parameters()],
                [opt.state[p]["exp_avg"].clone() for p in model.parameters()], [opt.state[p]["exp_avg_sq"].clone() for p in model.parameters()]))
print("losses separate", res[0][0])
print("losses fused   ", res[1][0])
for name, k in (("weights", 1), ("exp_avg", 2), ("exp_avg_sq", 3)):
    for a, b, (pn, _) in zip(res[0][k], res[1][k], model.named_parameters()):
        d = (a - b).abs()
        print(f"{name:10s} {pn:24s} max|diff| {float(d.max()):.3e}  max|a| {float(a.abs().max()):.3e}  differing {int((d > 0).sum())} of {a.numel()}")
