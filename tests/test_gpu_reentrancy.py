"""GPU (-m gpu): launches of the library that overlap in time must not share hidden state ("latent bugs").

* The register-streaming weight-gradient kernel (csrc/gemm_dr.hip: dr_tn_kernel, reference main.py:350) pulls its tiles from
  ticket counters in device memory.  The counters used to belong to the CALL SITE: two launches of the same entry point that
  overlap -- the two weight gradients of a step on two streams (GDMCF_GEMM_SIDE=1), two host threads -- drew from the same
  queues and each computed a subset of its tiles.  They now belong to the launch (dr_ticket_slot); here the two Yelp-shape
  weight-gradient products run CONCURRENTLY on two streams, twenty times, and every result must equal the one-stream result
  bit for bit, and the whole training step with GDMCF_GEMM_SIDE=1 must leave the same gradients and weights as the default order.
* `bench.py --gpus 2` end to end through the launcher's real Popen path on a one-GPU box (ranks share the device, gloo group).
"""
import hashlib
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_weight_gradient_products_overlapping_on_two_streams():
    from gdmcf_amd import _lib
    lib = _lib.load()
    B, I, H, E = 400, 34395, 1000, 10
    g = torch.Generator(device="cpu").manual_seed(11)
    ldi, ldk, ldh = (I + 63) // 64 * 64, (I + E + 63) // 64 * 64, 1024
    dz2 = torch.randn(B, ldi, generator=g).to(DEV)      # d(loss)/d(out)      [B, I]
    hs = torch.randn(B, ldh, generator=g).to(DEV)       # scaled hidden act   [B, H]
    dz1 = torch.randn(B, ldh, generator=g).to(DEV)      # d(loss)/d(hidden)   [B, H]
    xin = torch.randn(B, ldk, generator=g).to(DEV)      # first-layer input   [B, I + E]

    def dw2(out, stream):
        _lib.check(lib.gdmcf_linear_bwd_weight_f32(dz2.data_ptr(), ldi, hs.data_ptr(), ldh, None, 0, B, I, H, out.data_ptr(), H,
                                                   None, 0, stream))

    def dw1(out, stream):
        _lib.check(lib.gdmcf_linear_bwd_weight_f32(dz1.data_ptr(), ldh, xin.data_ptr(), ldk, None, 0, B, H, I + E, out.data_ptr(),
                                                   I + E, None, 0, stream))

    ref2 = torch.full((I, H), float("nan"), device=DEV)
    ref1 = torch.full((H, I + E), float("nan"), device=DEV)
    dw2(ref2, _lib.stream_ptr())
    torch.cuda.synchronize()
    dw1(ref1, _lib.stream_ptr())
    torch.cuda.synchronize()
    r64 = dz2[:, :I].double().t() @ hs[:, :H].double()
    assert float((ref2.double() - r64).abs().max()) <= 2e-6 * float(r64.abs().max())
    del r64
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(20):
        o2 = torch.full((I, H), float("nan"), device=DEV)
        o1 = torch.full((H, I + E), float("nan"), device=DEV)
        torch.cuda.synchronize()
        # both launches are enqueued before either can have finished (0.2 ms each): they share the chip
        with torch.cuda.stream(s1):
            dw2(o2, s1.cuda_stream)
        with torch.cuda.stream(s2):
            dw1(o1, s2.cuda_stream)
            if rep % 2:  # and a second pair right behind, so that four launches are in flight on two streams
                dw1(o1, s2.cuda_stream)
        if rep % 2:
            with torch.cuda.stream(s1):
                dw2(o2, s1.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(o2, ref2), f"dW2 differs when launched beside dW1 (repetition {rep})"
        assert torch.equal(o1, ref1), f"dW1 differs when launched beside dW2 (repetition {rep})"


_STEP_SCRIPT = r'''
import hashlib, sys, numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, {root!r})
import gdmcf_amd
from gdmcf_amd import data
from gdmcf_amd.data_utils import DeviceCSR
from gdmcf_amd.parallel import DataParallelStep
dev = torch.device("cuda:0")
B, hid, T = 400, 1000, 5
indptr, indices, I = data.synth_csr("yelp", n_rows=2 * B, seed=0)
dcsr = DeviceCSR(sp.csr_matrix((np.ones(len(indices), np.float32), indices, indptr), shape=(2 * B, I)), dev)
torch.manual_seed(0)
model = gdmcf_amd.DNN([I, hid], [hid, I], 10, time_type="cat", norm=False).to(dev).train()
diffusion = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, dev)
opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.0)
step = DataParallelStep(diffusion, model, opt)
torch.manual_seed(99)
h = hashlib.sha256()
for i in range(4):
    loss = step(dcsr.batch(torch.arange((i % 2) * B, (i % 2 + 1) * B, device=dev)), True)
    torch.cuda.synchronize()
    for p in model.parameters():
        h.update(p.grad.detach().cpu().numpy().tobytes())
        h.update(p.detach().cpu().numpy().tobytes())
    h.update(np.float64(float(loss)).tobytes())
print("SIDE", model.engine._gemm_side, "HASH", h.hexdigest())
'''


def test_training_step_with_the_side_stream_weight_gradient_is_bit_identical(tmp_path):
    """GDMCF_GEMM_SIDE=1 (engine.py: the last layer's weight-gradient product on a second stream, beside the input-gradient
    product and -- once that has drained -- the first layer's weight gradient): four Yelp-shape training steps, every gradient
    and every weight after every step hashed; the hash must equal the default order's."""
    script = tmp_path / "side_step.py"
    script.write_text(_STEP_SCRIPT.format(root=ROOT))
    out = {}
    for side in ("0", "1"):
        r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, GDMCF_GEMM_SIDE=side), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("SIDE")][-1].split()
        assert line[1] == ("True" if side == "1" else "False")
        out[side] = line[3]
    assert out["0"] == out["1"]


def test_bench_gpus_2_end_to_end_on_one_gpu():
    """First-contact rehearsal for the 8-GPU node (reference main.py:343-351 has no distributed path): the command
    shape the driver runs, `python bench.py --gpus 2`, through launch_ranks' real Popen path -- two rank processes, rendezvous,
    the data-parallel step with its gradient exchange, dp_autotune, the strong-scaling leg, the replica check, one line from
    rank 0.  One GPU here, so the ranks share it and meet in a gloo group (GDMCF_BENCH_SHARE_GPU=1; the line says `rehearsal`);
    --graph-dp cannot capture host-staged collectives and must stay eager IN the process with capture_error set."""
    env = dict(os.environ, GDMCF_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--workload", "tiny",
           "--hidden", "96", "--graph-dp", "--preheat-seconds", "0.05"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_in_group"] == 2 and line["steps"] == 3
    assert line["replicas_in_sync"] is True
    assert line["rehearsal"]
    at = line["dp_autotune"]
    assert at and at["chosen"] in ("sharded", "allreduce") and at["allreduce_ms_per_step"] > 0
    ss = line["strong_scaling"]
    assert ss and ss["global_batch"] == 400 and ss["batch_per_gpu"] == 200 and ss["ms_per_step"] > 0
    gl = line["graph_leg"]
    assert gl and gl.get("captured") is False and "nccl" in (gl.get("capture_error") or ""), gl
    assert line["value"] > 0 and line["config"]["parallelism"] == "dp2"
    # without the rehearsal switch two ranks on one GPU must refuse: non-zero exit, no line
    env.pop("GDMCF_BENCH_SHARE_GPU")
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and "{" not in r.stdout
