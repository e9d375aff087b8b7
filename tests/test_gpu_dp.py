"""GPU (-m gpu): end-to-end rehearsal of DataParallelStep with 2 ranks sharing the one GPU of the test
box (gloo, host-staged collectives; production uses nccl = RCCL).  Two ranks on half batches must
reproduce a single process on the global batch: loss, parameters after AdamW, Lt-history."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
WORLD, B, I, HID, T, STEPS = 2, 64, 515, 100, 5, 3


def _inputs(step):
    g = torch.Generator().manual_seed(100 + step)
    x = (torch.rand(B, I, generator=g) < 0.05).float()
    ts = torch.randint(0, T, (B,), generator=g)
    noise = torch.randn(B, I, generator=g)
    keep = (torch.rand(B, I, generator=g) < 0.5).to(torch.uint8)
    return x, ts, torch.ones(B, dtype=torch.float64), noise, keep


def _build(dev):
    import gdmcf_amd
    torch.manual_seed(0)
    model = gdmcf_amd.DNN([I, HID], [HID, I], 10).to(dev)
    diff = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, dev)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.0)
    model.train()
    return model, diff, opt


def _worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from gdmcf_amd.parallel import DataParallelStep
    dev = "cuda:0"
    model, diff, opt = _build(dev)
    if rank == 1:  # must be overwritten by the broadcast from rank 0
        with torch.no_grad():
            model.out_layers[0].bias.add_(3.0)
    step = DataParallelStep(diff, model, opt)
    losses = []
    lo, hi = rank * B // WORLD, (rank + 1) * B // WORLD
    for s in range(STEPS):
        x, ts, pt, noise, keep = [t[lo:hi].to(dev) for t in _inputs(s)]
        losses.append(float(step(x, True, ts=ts, pt=pt, noise=noise, drop_mask=keep)))
    torch.cuda.synchronize()
    torch.save(dict(losses=losses, params=[p.detach().cpu() for p in model.parameters()], hist=diff.Lt_history.cpu(),
                    cnt=diff.Lt_count.cpu()), os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_the_global_batch(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    from gdmcf_amd.parallel import DataParallelStep
    dev = "cuda:0"
    model, diff, opt = _build(dev)
    step = DataParallelStep(diff, model, opt)
    ref_losses = []
    for s in range(STEPS):
        x, ts, pt, noise, keep = [t.to(dev) for t in _inputs(s)]
        ref_losses.append(float(step(x, True, ts=ts, pt=pt, noise=noise, drop_mask=keep)))
    # global mean loss = mean of the two local means (equal shard sizes)
    np.testing.assert_allclose(0.5 * (np.array(r0["losses"]) + np.array(r1["losses"])), ref_losses, rtol=1e-5)
    for a, b, p in zip(r0["params"], r1["params"], model.parameters()):
        assert torch.equal(a, b)  # replicas stay bit-identical
        assert float((a - p.detach().cpu()).abs().max()) < 0.05 * 1e-3 * STEPS
    for r in (r0, r1):
        np.testing.assert_array_equal(r["cnt"].numpy(), diff.Lt_count.cpu().numpy())
        np.testing.assert_allclose(r["hist"].numpy(), diff.Lt_history.cpu().numpy(), rtol=1e-5)


def test_single_process_early_update_equals_sequential_step():
    """world == 1, early_update=True: DataParallelStep issues the AdamW update of each large weight on a side stream as soon as its
    gradient GEMM is enqueued (input gradient first).  Same kernels on the same data -> bit-identical to the plain
    zero_grad / training_losses / backward / step loop."""
    import gdmcf_amd
    from gdmcf_amd.parallel import DataParallelStep
    dev, I2, H2, B2 = "cuda:0", 3000, 128, 48

    def build():
        torch.manual_seed(3)
        m = gdmcf_amd.DNN([I2, H2], [H2, I2], 10).to(dev).train()
        d = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, dev)
        o = gdmcf_amd.FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
        return m, d, o

    def inputs(s):
        g = torch.Generator().manual_seed(7 + s)
        return dict(x=(torch.rand(B2, I2, generator=g) < 0.03).float().to(dev), ts=torch.randint(0, T, (B2,), generator=g).to(dev),
                    noise=torch.randn(B2, I2, generator=g).to(dev), keep=(torch.rand(B2, I2, generator=g) < 0.5).float().to(dev))

    m0, d0, o0 = build()
    l0 = []
    for s in range(4):
        i = inputs(s)
        o0.zero_grad()
        l = d0.training_losses(m0, i["x"], True, ts=i["ts"], pt=torch.ones(B2, device=dev), noise=i["noise"], drop_mask=i["keep"])["loss"].mean()
        l.backward()
        o0.step()
        l0.append(float(l.detach()))
    m1, d1, o1 = build()
    step = DataParallelStep(d1, m1, o1, early_update=True)
    assert m1.engine.grad_sink is not None and m1.engine.input_grad_first  # the overlapped path is the one under test
    l1 = []
    for s in range(4):
        i = inputs(s)
        l1.append(float(step(i["x"], True, ts=i["ts"], pt=torch.ones(B2, device=dev), noise=i["noise"], drop_mask=i["keep"])))
    torch.cuda.synchronize()
    # the reported scalar: torch's mean of the float64 row losses vs the loss-tail kernel's fixed-order mean of the same
    # values (last bit may differ); everything the step COMPUTES WITH is compared bit for bit below
    np.testing.assert_allclose(l0, l1, rtol=1e-14, atol=0)
    for a, b in zip(m0.parameters(), m1.parameters()):
        assert torch.equal(a, b)
        assert torch.equal(o0.state[a]["exp_avg_sq"], o1.state[b]["exp_avg_sq"]) and o0.state[a]["step"] == o1.state[b]["step"]
    assert torch.equal(d0.Lt_history, d1.Lt_history)


def _lightgcn_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    import gdmcf_amd
    dev = "cuda:0"
    rng = np.random.default_rng(5)
    U, It, d, L = 301, 203, 64, 3
    users = rng.integers(0, U, 4000)
    items = np.where(rng.random(4000) < 0.3, 7, rng.integers(0, It, 4000))  # item 7 is a hub (split rows)
    pairs = np.unique(np.stack([users, items], 1), axis=0)
    data = {"user_id_idx": pairs[:, 0], "item_id_idx": pairs[:, 1]}
    torch.manual_seed(11)
    ref = gdmcf_amd.LightGCN(data, U, It, L, d, dev).to(dev)
    torch.manual_seed(11)
    sh = gdmcf_amd.LightGCN(data, U, It, L, d, dev, shard_rows=True).to(dev)
    assert sh._world == WORLD and torch.equal(ref.E0.weight, sh.E0.weight)
    with torch.no_grad():
        a, b = ref.propagate_through_layers(), sh.propagate_through_layers()
    ok_fwd = all(torch.equal(x, y) for x, y in zip(a, b))
    # backward: rank r has its own cotangent; the sharded backward must deliver the propagated SUM over ranks
    gs = [torch.randn(U + It, d, generator=torch.Generator().manual_seed(100 + r)).to(dev) for r in range(WORLD)]
    fu, fi, _, _ = sh.propagate_through_layers()
    (torch.cat([fu, fi]) * gs[rank]).sum().backward()
    want = ref._propagate((gs[0] + gs[1]).contiguous())[0]
    ok_bwd = torch.equal(sh.E0.weight.grad, want)
    torch.save(dict(ok_fwd=ok_fwd, ok_bwd=ok_bwd, rows=sh._rows), os.path.join(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


def test_row_sharded_lightgcn_equals_single_process(tmp_path):
    """LightGCN(shard_rows=True): every rank owns a row block of the adjacency, layers are local SpMMs + an all-gather,
    the backward all-reduces the cotangent first.  Bit-identical to the unsharded propagation (rows are independent
    and the kernels deterministic), forward and backward, on a graph with a hub row that is split across waves."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_lightgcn_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    r0, r1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    assert r0["ok_fwd"] and r1["ok_fwd"] and r0["ok_bwd"] and r1["ok_bwd"]
    assert r0["rows"][:2] == (0, 252) and r1["rows"][:2] == (252, 504)


def _shard_worker(rank, port, out_dir, world=WORLD, I2=3001):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gdmcf_amd
    from gdmcf_amd.parallel import DataParallelStep
    dev, H2, B2 = "cuda:0", 128, 32  # 3001 rows / 2 ranks: one leftover row that every rank updates (3003 / 4: three)

    def run(shard):
        torch.manual_seed(5)
        m = gdmcf_amd.DNN([I2, H2], [H2, I2], 10).to(dev).train()
        d = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, dev)
        o = gdmcf_amd.FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
        step = DataParallelStep(d, m, o, shard_optimizer=bool(shard))
        losses = []
        for s in range(4):
            if shard == "toggle":  # bench.py's warm-up autotune: sharded, all-reduce, sharded, all-reduce
                assert step.set_shard_optimizer(s % 2 == 0) == (s % 2 == 0)
            g = torch.Generator().manual_seed(50 + s)
            x = (torch.rand(world * B2, I2, generator=g) < 0.03).float()[rank * B2:(rank + 1) * B2].to(dev)
            ts = torch.randint(0, T, (world * B2,), generator=g)[rank * B2:(rank + 1) * B2].to(dev)
            noise = torch.randn(world * B2, I2, generator=g)[rank * B2:(rank + 1) * B2].to(dev)
            keep = (torch.rand(world * B2, I2, generator=g) < 0.5).float()[rank * B2:(rank + 1) * B2].to(dev)
            losses.append(float(step(x, True, ts=ts, pt=torch.ones(B2, device=dev), noise=noise, drop_mask=keep)))
        step.gather_optimizer_state()
        torch.cuda.synchronize()
        return m, o, losses, step

    m0, o0, l0, _ = run(False)
    m1, o1, l1, st1 = run(True)
    m2, o2, l2, st2 = run("toggle")
    ok = st1.shard_optimizer and l0 == l1 and l0 == l2
    for a, b, c in zip(m0.parameters(), m1.parameters(), m2.parameters()):
        ok = ok and torch.equal(a, b) and torch.equal(o0.state[a]["exp_avg"], o1.state[b]["exp_avg"]) \
            and torch.equal(o0.state[a]["exp_avg_sq"], o1.state[b]["exp_avg_sq"]) and o0.state[a]["step"] == o1.state[b]["step"]
        ok = ok and torch.equal(a, c) and torch.equal(o0.state[a]["exp_avg"], o2.state[c]["exp_avg"]) \
            and torch.equal(o0.state[a]["exp_avg_sq"], o2.state[c]["exp_avg_sq"])
    # beyond two ranks a ring all-reduce adds an element's contributions in an order that depends on where the element sits in
    # the buffer, and the sharded path reduces differently cut buffers: equal up to fp32 summation order, not bit for bit
    close = st1.shard_optimizer and bool(np.allclose(l0, l1, rtol=1e-6) and np.allclose(l0, l2, rtol=1e-6))
    for a, b, c in zip(m0.parameters(), m1.parameters(), m2.parameters()):
        for x, y in ((a, b), (a, c), (o0.state[a]["exp_avg"], o1.state[b]["exp_avg"]), (o0.state[a]["exp_avg_sq"], o1.state[b]["exp_avg_sq"])):
            close = close and bool(torch.allclose(x, y, rtol=2e-4, atol=1e-6 * float(x.abs().max()) + 1e-12))
    torch.save(dict(ok=bool(ok), close=bool(close), params=[p.detach().cpu() for p in m1.parameters()]), os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_sharded_optimizer_equals_all_reduce_path(tmp_path):
    """DataParallelStep(shard_optimizer=True): reduce-scatter of the row blocks, AdamW on the own rows, in-place
    all-gather of the updated rows (+ all-reduce for the R mod world leftover row).  Same losses, weights, moments and
    step counts as the all-reduce path, bit for bit, and identical replicas."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_shard_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    r0, r1 = torch.load(tmp_path / "s0.pt"), torch.load(tmp_path / "s1.pt")
    assert r0["ok"] and r1["ok"]
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)


def test_sharded_optimizer_world4_with_leftover_rows(tmp_path):
    """The same through FOUR ranks (gloo, all on the one GPU of the test box) and a weight of 3003 rows: 750 rows per rank
    + 3 leftover rows that every rank updates from the all-reduced tail -- `R mod world != 0` beyond world 2."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_shard_worker, args=(port, str(tmp_path), 4, 3003), nprocs=4, join=True)
    rs = [torch.load(tmp_path / f"s{r}.pt") for r in range(4)]
    assert all(r["close"] for r in rs)  # (bit-identical only for two ranks: see _shard_worker)
    for r in rs[1:]:
        for a, b in zip(rs[0]["params"], r["params"]):
            assert torch.equal(a, b)


def _rccl_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    import gdmcf_amd
    from gdmcf_amd import parallel
    from gdmcf_amd.parallel import DataParallelStep
    dev, I2, H2, B2 = "cuda:0", 3001, 128, 32

    def run(**kw):
        torch.manual_seed(5)
        m = gdmcf_amd.DNN([I2, H2], [H2, I2], 10).to(dev).train()
        d = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, dev)
        o = gdmcf_amd.FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
        step = DataParallelStep(d, m, o, **kw)
        losses = []
        for s in range(4):
            g = torch.Generator().manual_seed(50 + s)
            x = (torch.rand(B2, I2, generator=g) < 0.03).float().to(dev)
            ts = torch.randint(0, T, (B2,), generator=g).to(dev)
            noise = torch.randn(B2, I2, generator=g).to(dev)
            keep = (torch.rand(B2, I2, generator=g) < 0.5).float().to(dev)
            losses.append(float(step(x, True, ts=ts, pt=torch.ones(B2, device=dev), noise=noise, drop_mask=keep)))
        deferred = len(m.engine.weight_waiters)  # all-gathers of updated rows not waited for yet
        m.state_dict()
        deferred = (deferred, len(m.engine.weight_waiters))
        step.gather_optimizer_state()
        torch.cuda.synchronize()
        return m, o, losses, d, step, deferred

    ref = run()  # one rank, no exchange
    res = {}
    for name, kw in (("allreduce", dict(force_exchange=True)), ("sharded", dict(force_exchange=True, shard_optimizer=True)),
                     ("no_overlap", dict(force_exchange=True, overlap=False))):
        got = run(**kw)
        ok = got[4].exchange and ref[2] == got[2]
        for a, b in zip(ref[0].parameters(), got[0].parameters()):
            ok = ok and torch.equal(a, b) and torch.equal(ref[1].state[a]["exp_avg_sq"], got[1].state[b]["exp_avg_sq"])
        ok = ok and torch.equal(ref[3].Lt_history, got[3].Lt_history) and torch.equal(ref[3].Lt_count, got[3].Lt_count)
        ok = ok and got[5] == ((2, 0) if name == "sharded" else (0, 0))
        res[name] = bool(ok)
    # the row collectives on device tensors
    t = torch.arange(12, dtype=torch.float32, device=dev).view(4, 3)
    full = t.clone()
    h = parallel.all_gather_rows_inplace(full)
    if h is not None:
        h.wait()
    shard, h = parallel.reduce_scatter_rows(t.clone())
    if h is not None:
        h.wait()
    torch.cuda.synchronize()
    res["rows"] = bool(torch.equal(full, t) and torch.equal(shard, t) and torch.equal(parallel.all_gather_rows(t, 3), t[:3]))
    torch.save(res, os.path.join(out_dir, "rccl.pt"))
    dist.destroy_process_group()


def test_rccl_one_rank_group_runs_every_collective(tmp_path):
    """The production backend ("nccl" = RCCL) cannot put two ranks on the one GPU of the test box, so the device-side
    collective calls (async all-reduce handles, reduce_scatter_tensor, the in-place all_gather_into_tensor, broadcast)
    are rehearsed in a group of ONE rank with force_exchange=True: a one-rank SUM is the identity, so losses, weights,
    moments and the Lt history must equal the plain single-process step bit for bit."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rccl_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    res = torch.load(tmp_path / "rccl.pt")
    assert res == dict(allreduce=True, sharded=True, no_overlap=True, rows=True), res


def test_bench_data_parallel_path_in_a_one_rank_rccl_group():
    """bench.py --rehearse-dp: the line the driver gets at N > 1 (sharded optimiser, deferred all-gathers, barrier +
    max-over-ranks timing, replica check) produced through a one-rank RCCL group; same loss as the plain run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for extra in (["--rehearse-dp", "--shard-optimizer"], ["--rehearse-dp", "--allreduce-optimizer"], [],
                  ["--rehearse-dp", "--autotune-dp"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2",
                            "--no-cpu-baseline", "--no-live-traffic", "--separate-optimizer"] + extra, capture_output=True,
                           text=True, cwd=root, timeout=600, env=dict(os.environ, MASTER_PORT="29571"))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]))
    sh, ar, plain, auto = outs
    # neither variant requested: both are timed for a few untimed steps after the warm-up, the faster one runs, and the
    # switch between them (moments of the sharded weights gathered first) leaves the replicas consistent
    tune = auto["dp_autotune"]
    assert tune["chosen"] in ("sharded", "allreduce") and tune["allreduce_ms_per_step"] > 0 and tune["sharded_ms_per_step"] > 0
    assert auto["replicas_in_sync"] is True and sh["dp_autotune"] is None
    assert sh["replicas_in_sync"] is True and ar["replicas_in_sync"] is True and plain["replicas_in_sync"] is None
    assert "row-sharded" in sh["optimizer"] and "all-reduce" in ar["optimizer"]
    assert sh["final_loss"] == ar["final_loss"] == plain["final_loss"]
    assert sh["n_gpus"] == 1 and sh["scaling"] == "weak"


def _onehot_inputs(step, B2, I2):
    g = torch.Generator().manual_seed(300 + step)
    return dict(x=(torch.rand(B2, I2, generator=g) < 0.05).float(), ts=torch.randint(0, T, (B2,), generator=g),
                sampled=(torch.rand(B2, I2, generator=g) < 0.03).to(torch.uint8), noise=torch.randn(B2, I2, generator=g),
                drop_mask=(torch.rand(B2, I2, generator=g) < 0.5).to(torch.uint8),
                drop_mask_U=(torch.rand(B2, 2 * I2, generator=g) < 0.5).to(torch.uint8))


def _onehot_build(dev, I2, H2):
    import gdmcf_amd
    torch.manual_seed(9)
    m = gdmcf_amd.DNNOneHot([I2, H2], [H2, I2], 10).to(dev).train()
    d = gdmcf_amd.GaussianDiffusionDiscrete(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, dev,
                                            CatOneHot=True)
    return m, d, gdmcf_amd.FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.0)


def _onehot_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from gdmcf_amd.parallel import DataParallelStep
    dev, I2, H2, B2 = "cuda:0", 257, 48, 32
    m, d, o = _onehot_build(dev, I2, H2)
    step = DataParallelStep(d, m, o)
    lo, hi = rank * B2 // WORLD, (rank + 1) * B2 // WORLD
    losses = []
    for s in range(3):
        inp = {k: v[lo:hi].to(dev) for k, v in _onehot_inputs(s, B2, I2).items()}
        x = inp.pop("x")
        losses.append(float(step(x, True, pt=torch.ones(hi - lo, device=dev), **inp)))
    torch.cuda.synchronize()
    torch.save(dict(losses=losses, params=[p.detach().cpu() for p in m.parameters()], hist=d.Lt_history.cpu()),
               os.path.join(out_dir, f"o{rank}.pt"))
    dist.destroy_process_group()


def test_onehot_backbone_two_ranks_equal_one_process(tmp_path):
    """The one-hot backbone under DataParallelStep (gradients handed to the sink as their kernels are enqueued): two
    ranks on half batches reproduce one process on the global batch (sampled classes injected -- the reference's noise
    level a = ts / batch_size would otherwise depend on the LOCAL batch size)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_onehot_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    r0, r1 = torch.load(tmp_path / "o0.pt"), torch.load(tmp_path / "o1.pt")
    from gdmcf_amd.parallel import DataParallelStep
    dev, I2, H2, B2 = "cuda:0", 257, 48, 32
    m, d, o = _onehot_build(dev, I2, H2)
    step = DataParallelStep(d, m, o)
    ref = []
    for s in range(3):
        inp = {k: v.to(dev) for k, v in _onehot_inputs(s, B2, I2).items()}
        x = inp.pop("x")
        ref.append(float(step(x, True, pt=torch.ones(B2, device=dev), **inp)))
    np.testing.assert_allclose(0.5 * (np.array(r0["losses"]) + np.array(r1["losses"])), ref, rtol=1e-5)
    for a, b, p in zip(r0["params"], r1["params"], m.parameters()):
        assert torch.equal(a, b)
        assert float((a - p.detach().cpu()).abs().max()) < 0.05 * 1e-3 * 3
    np.testing.assert_allclose(r0["hist"].numpy(), d.Lt_history.cpu().numpy(), rtol=1e-5)
