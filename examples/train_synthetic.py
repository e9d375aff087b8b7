"""End-to-end run of the reference's training / evaluation loop (main.py:327-378) on synthetic rows of a named shape:

    python examples/train_synthetic.py --users 8000 --epochs 3            # fp32
    python examples/train_synthetic.py --users 8000 --epochs 3 --bf16     # bf16 GEMM inputs
    python examples/train_synthetic.py --users 8000 --epochs 3 --graph    # CSR rows, one hipGraph launch per batch
    python examples/train_synthetic.py --users 8000 --epochs 3 --gemm f32x3   # f32 products from three-term bf16 splits
    python examples/train_synthetic.py --users 8000 --epochs 3 --backbone onehot   # one-hot variant (DNNOneHot)
    python examples/train_synthetic.py --users 4000 --epochs 2 --backbone onehot-emb --hidden 256 --lightgcn-init 50
                                                                                   # LightGCN BPR -> tables handed to the denoiser
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/train_synthetic.py --users 64000 --epochs 3               # data parallel: one process per GPU (RCCL)

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
    ap.add_argument("--gemm", default=None, choices=["f32", "bf16", "f32x3"],
                    help="GEMM mode of the DNN denoiser (f32x3: float32 products from three-term bf16 splits, DESIGN 4.4b)")
    ap.add_argument("--sparse-rows", action="store_true", help="feed the training rows as CSR batches (never densified)")
    ap.add_argument("--graph", action="store_true", help="replay the training step from one hipGraph (single GPU, dnn backbone)")
    ap.add_argument("--backbone", default="dnn", choices=["dnn", "onehot", "onehot-emb", "onehot-gcn"],
                    help="onehot: GaussianDiffusionDiscrete(CatOneHot=True) + DNNOneHot; onehot-emb: + user / item embedding "
                         "tables (DNNOneHotEmbedding, indexIn); both fp32")
    ap.add_argument("--lightgcn-init", type=int, default=0, metavar="STEPS",
                    help="onehot-emb / onehot-gcn: first train a LightGCN (latent dim = --hidden) on the training graph for "
                         "STEPS BPR steps (HIP SpMM propagation, reference lightGCN.py:273-338) and hand its propagated tables "
                         "to the backbone's embedding_user / embedding_item (load_lightgcn_embeddings)")
    args = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    say = print if rank == 0 else (lambda *a, **k: None)
    indptr, indices, I = data.synth_csr(args.shape, n_rows=args.users, seed=0)
    U = len(indptr) - 1
    rng = np.random.default_rng(0)
    held = rng.random(len(indices)) < 0.2
    rows = np.repeat(np.arange(U), np.diff(indptr))
    mk = lambda m: sp.csr_matrix((np.ones(int(m.sum()), np.float32), (rows[m], indices[m])), shape=(U, I))
    train, test = mk(~held), mk(held)
    # data parallel: rank r trains on the users r, r + world, ... (every step then covers `world` x batch users whose
    # gradients are all-reduced); evaluation runs on rank 0 over all users
    n_dp = (U // (world * args.batch)) * world * args.batch  # every rank must run the same number of steps
    my_train = train[:n_dp][rank::world] if world > 1 else train
    torch.manual_seed(0)
    if args.backbone != "dnn":  # what main.py builds for CatOneHot with args.backbone == 'DNNOneHot' / 'DNNOneHotEmbedding'
        if args.backbone == "onehot":
            model = gdmcf_amd.DNNOneHot([I, args.hidden], [args.hidden, I], 10, time_type="cat", norm=False).to(dev)
        else:
            cls = gdmcf_amd.DNNOneHotEmbedding if args.backbone == "onehot-emb" else gdmcf_amd.DNNOneHotEmbeddingGCN
            model = cls([I, args.hidden], [args.hidden, I], 10, time_type="cat", norm=False, item_num=I, user_num=U).to(dev)
        diffusion = gdmcf_amd.GaussianDiffusionDiscrete(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01,
                                                        args.T, dev, CatOneHot=True)
        diffusion.indexIn = args.backbone != "onehot"  # main.py:241, :245
    else:
        model = gdmcf_amd.DNN([I, args.hidden], [args.hidden, I], 10, time_type="cat", norm=False,
                              gemm_dtype=args.gemm or ("bf16" if args.bf16 else "f32")).to(dev)
        diffusion = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, args.T, dev)
    if args.lightgcn_init > 0 and args.backbone in ("onehot-emb", "onehot-gcn"):
        # "fed by the LightGCN propagation" (north star): BPR-train the graph model on the same interactions, then
        # initialise the denoiser's user / item tables from mean_l(A~^l E0)
        from gdmcf_amd.lightgcn import bpr_loss, sample_bpr_batch
        coo = train.tocoo()
        lg = gdmcf_amd.LightGCN({"user_id_idx": coo.row, "item_id_idx": coo.col}, U, I, 3, args.hidden, device=dev).to(dev)
        lopt = torch.optim.Adam(lg.parameters(), lr=0.005)
        tr = train.tocsr()
        active = np.nonzero(np.diff(tr.indptr) > 0)[0]
        t0 = time.perf_counter()
        for s_ in range(args.lightgcn_init):
            bu = np.sort(rng.choice(active, min(1024, len(active)), replace=False))
            deg = tr.indptr[bu + 1] - tr.indptr[bu]
            bp = tr.indices[tr.indptr[bu] + (rng.random(len(bu)) * deg).astype(np.int64)]
            bn = rng.integers(0, I, len(bu))
            bu_, bp_, bn_ = (torch.from_numpy(np.asarray(a, dtype=np.int64)).to(dev) for a in (bu, bp, bn))
            lopt.zero_grad()
            mf, reg = bpr_loss(bu_, *lg(bu_, bp_, bn_))
            (mf + 1e-4 * reg).backward()
            lopt.step()
        torch.cuda.synchronize()
        model.load_lightgcn_embeddings(lg)
        say(f"LightGCN: {args.lightgcn_init} BPR steps in {time.perf_counter() - t0:.2f} s (last mf loss {float(mf.detach()):.4f}); "
            f"tables handed to {type(model).__name__}", flush=True)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=args.lr, weight_decay=0.0)
    gen = torch.Generator().manual_seed(0)
    from gdmcf_amd.parallel import DataParallelStep
    step = DataParallelStep(diffusion, model, opt)  # broadcasts rank 0's weights when world > 1
    gstep = None
    if args.graph:
        from gdmcf_amd.data_utils import DeviceCSR
        from gdmcf_amd.graph import GraphedTrainStep
        my_train = DeviceCSR(my_train, dev)
        gstep = GraphedTrainStep(diffusion, model, opt, my_train, args.batch)
    topN = [10, 20, 50, 100]
    for epoch in range(args.epochs):
        t0 = time.perf_counter()
        total, count = driver.train_one_epoch(diffusion, model, opt, my_train, args.batch, dev, generator=gen, step=step,
                                              sparse=args.sparse_rows, graph_step=gstep)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        say(f"epoch {epoch}: mean loss {total / max(count, 1):.4f}, {world * count * args.batch / dt:,.0f} users/s", flush=True)
    if gstep is not None:
        gstep.close()
    if rank == 0:
        t0 = time.perf_counter()
        res = driver.evaluate(diffusion, model, train, test, train, topN, 0, False, args.batch, dev)
        print(f"evaluation of {U} users: {time.perf_counter() - t0:.2f} s")
        print_results(None, None, res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
