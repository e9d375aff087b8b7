"""How long does the host need to enqueue one training step?  (If this approaches the GPU time per step the GPU starves.)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gdmcf_amd  # noqa: E402
from gdmcf_amd.parallel import DataParallelStep  # noqa: E402

dev = torch.device("cuda:0")
B, I, hid, T = 400, 34395, 1000, 5
for dtype in ("f32", "bf16"):
    torch.manual_seed(0)
    model = gdmcf_amd.DNN([I, hid], [hid, I], 10, gemm_dtype=dtype).to(dev).train()
    diff = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, dev)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-5, weight_decay=0.0)
    step = DataParallelStep(diff, model, opt)
    x = (torch.rand(B, I, device=dev) < 0.00075).float()
    for _ in range(20):
        step(x, True)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        step(x, True)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{dtype}: host enqueue {1e3 * t_host / n:.3f} ms/step, wall {1e3 * t_all / n:.3f} ms/step", flush=True)
