"""The training step replayed from a hipGraph (gdmcf_amd/graph.py) against the eager step: same kernels, same Philox
offsets, same AdamW scalars -> bit-identical losses, weights, moments and Lt-history."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gdmcf_amd
from gdmcf_amd.gaussian_diffusion import ModelMeanType

DEV = torch.device("cuda:0")


def _setup(gemm_dtype, seed=5, fuse=False, big=False):
    import scipy.sparse as sp
    from gdmcf_amd.data_utils import DeviceCSR
    rng = np.random.default_rng(seed)
    U, I, hid, T = (500, 6001, 96, 7) if not big else (800, 34395, 1000, 5)  # big: the Yelp width (register-streaming kernels)
    dense = (rng.random((U, I)) < 0.006).astype(np.float32)
    dcsr = DeviceCSR(sp.csr_matrix(dense), DEV)
    torch.manual_seed(21)
    model = gdmcf_amd.DNN([I, hid], [hid, I], 10, gemm_dtype=gemm_dtype).to(DEV).train()
    diff = gdmcf_amd.GaussianDiffusion(ModelMeanType.START_X, "linear-var", 0.1, 0.001, 0.01, T, DEV)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    if fuse:  # AdamW of the two large weights inside their weight-gradient products (reads the step's scalars from the device)
        opt.fuse_into_backward(model, min_numel=1 << 12)
    return dcsr, model, diff, opt


def _state(model, diff, opt):
    out = [p.detach().clone() for p in model.parameters()]
    for p in model.parameters():
        out += [opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone()]
    return out + [diff.Lt_history.clone(), diff.Lt_count.clone()]


@pytest.mark.parametrize("gemm_dtype,table_steps,fuse,big", [("f32", 8192, False, False), ("bf16", 8192, False, False),
                                                             ("f32", 4, False, False), ("f32", 8192, True, False),
                                                             ("bf16", 5, True, False), ("f32", 4, True, True)])
def test_graph_replay_equals_eager_steps(gemm_dtype, table_steps, fuse, big):
    """(fuse: FusedAdamW.fuse_into_backward -- the N = 1 default of bench.py since round 4: the products that carry the optimiser
    read the step's AdamW scalars from the graph's device state; big: batch 400 at the Yelp width, where those products are the
    register-streaming kernel with the optimiser stream inside its k loop)"""
    from gdmcf_amd.graph import GraphedTrainStep
    from gdmcf_amd.parallel import DataParallelStep
    B, n_graph, n_after = (64, 14, 3) if not big else (400, 9, 2)
    U = 500 if not big else 800
    batches = [torch.from_numpy(np.random.default_rng(100 + k).permutation(U)[:B].astype(np.int64)) for k in range(n_graph + n_after)]

    dcsr, model, diff, opt = _setup(gemm_dtype, fuse=fuse, big=big)
    step = DataParallelStep(diff, model, opt)
    eager_losses = [step(dcsr.batch(b.to(DEV)), True).clone() for b in batches]
    eager = _state(model, diff, opt)
    del step, model, opt
    torch.cuda.empty_cache()

    dcsr, model, diff, opt = _setup(gemm_dtype, fuse=fuse, big=big)
    losses = []
    with GraphedTrainStep(diff, model, opt, dcsr, B, warmup=3, table_steps=table_steps) as gstep:
        for b in batches[:n_graph]:
            losses.append(gstep(b))
        assert gstep.graph is not None
    # the counters came back: eager steps continue where the graph stopped
    assert model.engine.offset == n_graph and diff._ts_calls == n_graph
    assert {int(opt.state[p]["step"]) for p in model.parameters()} == {n_graph}
    step = DataParallelStep(diff, model, opt)
    for b in batches[n_graph:]:
        losses.append(step(dcsr.batch(b.to(DEV)), True).clone())
    for k, (a, b) in enumerate(zip(eager_losses, losses)):
        assert torch.equal(a, b), (k, float(a), float(b))
    for a, b in zip(eager, _state(model, diff, opt)):
        assert torch.equal(a, b)


def test_graph_step_rejects_what_it_cannot_capture():
    from gdmcf_amd.graph import GraphedTrainStep
    dcsr, model, diff, opt = _setup("f32")
    with pytest.raises(ValueError):
        with GraphedTrainStep(diff, model, opt, dcsr, 32) as g:
            g(torch.arange(31))
    eps = gdmcf_amd.GaussianDiffusion(ModelMeanType.EPSILON, "linear-var", 0.1, 0.001, 0.01, 5, DEV)
    with pytest.raises(NotImplementedError):
        GraphedTrainStep(eps, model, opt, dcsr, 32)
    with pytest.raises(NotImplementedError):
        GraphedTrainStep(diff, model, torch.optim.AdamW(model.parameters()), dcsr, 32)


def _dp_graph_worker(rank, port, out_dir):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from gdmcf_amd.graph import GraphedTrainStep
    from gdmcf_amd.parallel import DataParallelStep
    B, n_graph = 64, 12
    batches = [torch.from_numpy(np.random.default_rng(300 + k).permutation(500)[:B].astype(np.int64)) for k in range(n_graph)]
    dcsr, model, diff, opt = _setup("f32")
    step = DataParallelStep(diff, model, opt, force_exchange=True)
    assert step.exchange
    eager_losses = [step(dcsr.batch(b.to(DEV)), True).clone() for b in batches]
    step.flush()
    eager = _state(model, diff, opt)
    dcsr, model, diff, opt = _setup("f32")
    losses = []
    with GraphedTrainStep(diff, model, opt, dcsr, B, warmup=3, force_exchange=True) as gstep:
        assert gstep.step.exchange
        for b in batches:
            losses.append(gstep(b))
        captured, err = isinstance(gstep.graph, torch.cuda.CUDAGraph), gstep.capture_error
    same = all(torch.equal(a, b) for a, b in zip(eager_losses, losses)) and \
        all(torch.equal(a, b) for a, b in zip(eager, _state(model, diff, opt)))
    torch.save(dict(captured=captured, err=err, same=same), os.path.join(out_dir, "dpgraph.pt"))
    dist.destroy_process_group()


def test_data_parallel_step_replayed_from_a_graph_in_a_one_rank_rccl_group(tmp_path):
    """The data-parallel body (gradient all-reduces + the small float64 exchange through RCCL, history replay on the gathered
    batch, AdamW with 1/world folded in) captured into ONE hipGraph: a group of one rank with force_exchange=True runs every
    collective (a one-rank SUM is the identity), so 3 eager + 9 replayed steps must equal 12 eager data-parallel steps bit for
    bit -- and the capture itself must succeed on this RCCL build."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_graph_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    res = torch.load(tmp_path / "dpgraph.pt")
    assert res["captured"], res["err"]
    assert res["same"]


def test_graph_step_follows_a_changed_learning_rate():
    """The AdamW scalars of the coming steps live in a device table: a learning-rate change between replays (scheduler) must
    reach the replayed update (round-2 advisor finding: it was silently ignored until the next table refill)."""
    from gdmcf_amd.graph import GraphedTrainStep
    from gdmcf_amd.parallel import DataParallelStep
    B, n = 64, 10
    batches = [torch.from_numpy(np.random.default_rng(400 + k).permutation(500)[:B].astype(np.int64)) for k in range(n)]
    dcsr, model, diff, opt = _setup("f32")
    step = DataParallelStep(diff, model, opt)
    for k, b in enumerate(batches):
        if k == 6:
            opt.param_groups[0]["lr"] = 3e-4
        step(dcsr.batch(b.to(DEV)), True)
    eager = _state(model, diff, opt)
    dcsr, model, diff, opt = _setup("f32")
    with GraphedTrainStep(diff, model, opt, dcsr, B, warmup=3) as gstep:
        for k, b in enumerate(batches):
            if k == 6:
                opt.param_groups[0]["lr"] = 3e-4
            gstep(b)
        assert isinstance(gstep.graph, torch.cuda.CUDAGraph)
    for a, b in zip(eager, _state(model, diff, opt)):
        assert torch.equal(a, b)
