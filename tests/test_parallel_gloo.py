"""CPU, world_size 2, gloo: the data-parallel exchange of gdmcf_amd.parallel.

The compute kernels need the GPU, so here the per-rank gradients / losses come from the oracle and
only the distributed plumbing under test is the product's: (1) SUM all-reduce of gradients + 1/world
scaling == gradient of the global-batch mean loss; (2) the (ts, loss) all-gather replays the
order-dependent Lt-history FIFO identically on every rank."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gdmcf_amd import parallel
from oracle import gdmcf_oracle as O

WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(B, I, hid, T, seed=0):
    torch.manual_seed(seed)
    model = O.DNN([I, hid], [hid, I], 10)
    g = torch.Generator().manual_seed(seed + 1)
    x = (torch.rand(B, I, generator=g) < 0.1).float()
    ts = torch.randint(0, T, (B,), generator=g)
    noise = torch.randn(B, I, generator=g)
    keep = (torch.rand(B, I, generator=g) < 0.5).float()
    return model, x, ts, noise, keep


def _worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(1)
    B, I, hid, T = 12, 40, 8, 3
    model, x, ts, noise, keep = _setup(B, I, hid, T)
    diff = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, history_num_per_term=4)
    lo, hi = rank * B // WORLD, (rank + 1) * B // WORLD
    pt = torch.ones(B)
    model.train()
    terms = diff.training_losses(model, x[lo:hi], True, ts=ts[lo:hi], pt=pt[lo:hi], noise=noise[lo:hi],
                                 drop_mask=keep[lo:hi])
    terms["loss"].mean().backward()
    # --- (1) gradient exchange (product code) ---
    parallel.allreduce_grads(list(model.parameters()), bucket_bytes=1 << 20)
    grads = [p.grad / WORLD for p in model.parameters()]  # FusedAdamW applies grad_scale = 1/world
    # --- (2) history exchange (product code) + replay in global batch order ---
    lu_local = terms["loss"].detach() * pt[lo:hi]
    ts_all, lu_all = parallel.gather_history_inputs(ts[lo:hi].contiguous(), lu_local.contiguous())
    rep = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, history_num_per_term=4)
    rep.update_history(ts_all, lu_all)
    # broadcast: rank 1 perturbs its weights, rank 0's must win
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    parallel.broadcast_parameters(model)
    torch.save(dict(grads=grads, ts_all=ts_all, lu_all=lu_all, hist=rep.Lt_history, cnt=rep.Lt_count,
                    params=[p.detach().clone() for p in model.parameters()]), os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_data_parallel_exchange_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    # single-process truth on the full global batch
    B, I, hid, T = 12, 40, 8, 3
    model, x, ts, noise, keep = _setup(B, I, hid, T)
    diff = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, history_num_per_term=4)
    model.train()
    terms = diff.training_losses(model, x, True, ts=ts, pt=torch.ones(B), noise=noise, drop_mask=keep)
    terms["loss"].mean().backward()
    for g0, g1, p in zip(r0["grads"], r1["grads"], model.parameters()):
        assert torch.equal(g0, g1)  # every rank ends with identical gradients
        np.testing.assert_allclose(g0.numpy(), p.grad.numpy(), rtol=2e-5, atol=1e-7 * float(p.grad.abs().max()))
    for r in (r0, r1):
        assert torch.equal(r["ts_all"], ts)  # rank order == global batch order
        # half-batch vs full-batch CPU GEMMs block differently: fp32 summation-order noise only
        np.testing.assert_allclose(r["lu_all"].numpy(), terms["loss"].detach().numpy(), rtol=2e-6)
        np.testing.assert_array_equal(r["cnt"].numpy(), diff.Lt_count.numpy())
        np.testing.assert_allclose(r["hist"].numpy(), diff.Lt_history.numpy(), rtol=2e-6)
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)


def test_single_process_paths_are_noops():
    assert not dist.is_initialized()
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    parallel.allreduce_grads([p])
    assert torch.equal(p.grad, torch.full((3,), 2.0))
    ts, lu = torch.arange(4), torch.rand(4, dtype=torch.float64)
    a, b = parallel.gather_history_inputs(ts, lu)
    assert a is ts and b is lu


def _rows_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    g = torch.Generator().manual_seed(10 + rank)
    rows = torch.randn(6, 5, generator=g)            # this rank's contribution, [world * 3, 5]
    mine, h = parallel.reduce_scatter_rows(rows.clone())
    if h is not None:
        h.wait()
    full = torch.full((6, 5), float("nan"))
    full[rank * 3:(rank + 1) * 3] = mine * 10 + rank   # "updated rows" of this rank
    h = parallel.all_gather_rows_inplace(full)
    if h is not None:
        h.wait()
    block = torch.full((4, 2), float(rank))            # ceil(7 / 2) = 4 rows per rank, 7 in total
    table = parallel.all_gather_rows(block, 7)
    torch.save(dict(rows=rows, mine=mine, full=full, table=table), os.path.join(out_dir, f"q{rank}.pt"))
    dist.destroy_process_group()


def test_row_collectives_world2(tmp_path):
    """reduce_scatter_rows / all_gather_rows_inplace (sharded optimiser) and all_gather_rows (row-sharded LightGCN)."""
    port = _free_port()
    mp.spawn(_rows_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    q0, q1 = torch.load(tmp_path / "q0.pt"), torch.load(tmp_path / "q1.pt")
    total = q0["rows"] + q1["rows"]
    assert torch.equal(q0["mine"], total[:3]) and torch.equal(q1["mine"], total[3:])
    want = torch.cat([total[:3] * 10 + 0, total[3:] * 10 + 1])
    assert torch.equal(q0["full"], want) and torch.equal(q1["full"], want)
    want_t = torch.cat([torch.zeros(4, 2), torch.ones(3, 2)])
    assert torch.equal(q0["table"], want_t) and torch.equal(q1["table"], want_t)


def _shard4_worker(rank, port, out_dir):
    """What DataParallelStep._sink / _finish_exchange do to ONE large weight gradient under shard_optimizer=True, with the
    product's collectives, on a weight whose row count is NOT a multiple of the world size (world 4, 11 rows: blocks of
    2 rows per rank + 3 leftover rows that every rank updates from an all-reduced gradient).  The update rule is a stand-in
    (w -= lr * g / world; the AdamW kernel needs the GPU): what is under test is the row arithmetic and the exchange."""
    world = 4
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R, C, lr = 11, 5, 0.1
    w = torch.arange(R * C, dtype=torch.float32).reshape(R, C) / 7.0          # identical replicas
    g = torch.Generator().manual_seed(40 + rank)
    grad = torch.randn(R, C, generator=g)                                      # this rank's local gradient
    n_eq = R // world * world
    shard, h = parallel.reduce_scatter_rows(grad[:n_eq].contiguous())
    tail = grad[n_eq:].clone()
    parallel._all_reduce(tail, None)
    if h is not None:
        h.wait()
    nr = n_eq // world
    w[rank * nr:(rank + 1) * nr] -= lr * shard / world                         # own block
    w[n_eq:] -= lr * tail / world                                              # leftover rows: every rank, same values
    h = parallel.all_gather_rows_inplace(w[:n_eq])
    if h is not None:
        h.wait()
    torch.save(dict(grad=grad, w=w), os.path.join(out_dir, f"w4_{rank}.pt"))
    dist.destroy_process_group()


def test_sharded_update_world4_with_leftover_rows(tmp_path):
    port = _free_port()
    mp.spawn(_shard4_worker, args=(port, str(tmp_path)), nprocs=4, join=True)
    rs = [torch.load(tmp_path / f"w4_{r}.pt") for r in range(4)]
    total = sum(r["grad"] for r in rs)
    want = torch.arange(11 * 5, dtype=torch.float32).reshape(11, 5) / 7.0 - 0.1 * total / 4
    for r in rs:
        np.testing.assert_allclose(r["w"].numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
        assert torch.equal(r["w"], rs[0]["w"])  # replicas bit-identical
