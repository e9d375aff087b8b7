"""End-to-end run of the reference's training / evaluation loop (main.py:327-378) on synthetic rows of a named shape:

    python examples/train_synthetic.py --users 8000 --epochs 3            # fp32
    python examples/train_synthetic.py --users 8000 --epochs 3 --bf16     # bf16 GEMM inputs

Synthetic users interact with popularity-skewed items (gdmcf_amd/data.py); 20 % of every user's interactions are held
out as the test set.  Prints the mean training loss per epoch, users/s, and Precision / Recall / NDCG / MRR @ topN.
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gdmcf_amd  # noqa: E402
from gdmcf_amd import data, driver  # noqa: E402
from gdmcf_amd.evaluate_utils import print_results  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="yelp", choices=list(data.SHAPES))
    ap.add_argument("--users", type=int, default=8000, help="number of synthetic users (rows) to train on")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--batch", type=int, default=400)
    ap.add_argument("--hidden", type=int, default=1000)
    ap.add_argument("--T", type=int, default=5)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--bf16", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    indptr, indices, I = data.synth_csr(args.shape, n_rows=args.users, seed=0)
    U = len(indptr) - 1
    rng = np.random.default_rng(0)
    held = rng.random(len(indices)) < 0.2
    rows = np.repeat(np.arange(U), np.diff(indptr))
    mk = lambda m: sp.csr_matrix((np.ones(int(m.sum()), np.float32), (rows[m], indices[m])), shape=(U, I))
    train, test = mk(~held), mk(held)
    torch.manual_seed(0)
    model = gdmcf_amd.DNN([I, args.hidden], [args.hidden, I], 10, time_type="cat", norm=False,
                          gemm_dtype="bf16" if args.bf16 else "f32").to(dev)
    diffusion = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, args.T, dev)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=args.lr, weight_decay=0.0)
    gen = torch.Generator().manual_seed(0)
    topN = [10, 20, 50, 100]
    for epoch in range(args.epochs):
        t0 = time.perf_counter()
        total, count = driver.train_one_epoch(diffusion, model, opt, train, args.batch, dev, generator=gen)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"epoch {epoch}: mean loss {total / max(count, 1):.4f}, {count * args.batch / dt:,.0f} users/s", flush=True)
    t0 = time.perf_counter()
    res = driver.evaluate(diffusion, model, train, test, train, topN, 0, False, args.batch, dev)
    print(f"evaluation of {U} users: {time.perf_counter() - t0:.2f} s")
    print_results(None, None, res)


if __name__ == "__main__":
    main()
