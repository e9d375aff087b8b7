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
