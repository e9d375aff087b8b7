#!/usr/bin/env python3
"""Benchmark of the GDMCF diffusion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload yelp|amazon-book]

One "step" = the body of the reference training loop (main.py:345-351): zero_grad ->
GaussianDiffusion.training_losses -> mean -> backward -> AdamW.step on one batch of 400 dense user
rows resident in HBM.  Metric: training users/sec (BASELINE.json), fp32, synthetic Yelp-shape rows
(the reference ships no dataset), random-init weights.  N > 1: one process per GPU (torchrun), each
rank takes its own 400-row batch (weak scaling), gradients all-reduced with RCCL.

Besides the contract fields the JSON line carries
  roofline     -- the dominant kernel's achieved rate, timed live with HIP events on the launch
                  stream inside the timed region (gdmcf_prof_*), against the gfx950 peak;
  cpu_baseline -- the oracle (a PyTorch-CPU restatement of the reference path, parity-pinned to
                  the reference by tests/golden) timed on this host's cores on the same workload.

Other legs / variants (not part of the default line): --gemm-dtype bf16 (BASELINE configs[2]), --workload
amazon-book|stress, --fuse-optimizer (AdamW inside the weight-gradient GEMM epilogues, N = 1), --allreduce-optimizer (N > 1: plain
all-reduce + full AdamW on every rank) / --shard-optimizer (reduce-scatter + AdamW on 1/N of the rows + deferred
all-gather); with neither, N > 1 times both for a few untimed steps after the warm-up and keeps the faster one
(`dp_autotune` in the output), --rehearse-dp (N = 1: every collective through a one-rank RCCL group), --global-batch G (strong scaling), --spmm (LightGCN
propagation; row-sharded when N > 1), --bpr (LightGCN BPR training step), --sampling (p_sample + masked top-k).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_* = 64 FLOP/clk/SIMD
PEAK_BF16_MATRIX_TFLOPS = 2516.6  # dense bf16 MFMA = 16 x the f32 matrix rate (same guide)
PEAK_HBM_GBPS = 8000.0
TAGS = {1: "linear_fwd_gemm", 2: "loss_fwd_gemm", 3: "posterior_gemm", 4: "bwd_input_gemm", 5: "bwd_weight_gemm",
        6: "adamw", 7: "prep_input", 8: "spmm_csr", 9: "topk", 10: "onehot_noise", 11: "randn"}
GEMM_TAGS = (1, 2, 3, 4, 5)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r04_hbm_traffic.json")


def measured_traffic(kernel_tag, workload, gemm_dtype="f32"):
    """FALLBACK only (live_traffic below is the measurement): HBM bytes per launch of the kernel behind `kernel_tag` from the
    committed rocprofv3 PMC passes (profiles/summarize.py writes one row per bench tag).  Returns (bytes or None, note)."""
    if not os.path.exists(TRAFFIC_FILE):
        return None, f"no {os.path.relpath(TRAFFIC_FILE, ROOT)}"
    prof = json.load(open(TRAFFIC_FILE))
    if prof.get("workload") != workload or prof.get("gemm_dtype") != gemm_dtype:
        return None, f"profile is for {prof.get('workload')}/{prof.get('gemm_dtype')}, this run is {workload}/{gemm_dtype}"
    row = prof.get("tags", {}).get(kernel_tag)
    if row is None:
        msg = f"{os.path.relpath(TRAFFIC_FILE, ROOT)} has no row for tag '{kernel_tag}' (re-run tools/profile_round.sh)"
        print(f"[bench] WARNING: roofline.traffic unavailable: {msg}", file=sys.stderr)
        return None, msg
    return int(row["hbm_total_MB"] * 1e6), f"{row['kernel']} ({row['launches']} launches profiled; committed profile, not this run)"


def live_traffic(kernel_tag, argv):
    """HBM bytes per launch of the kernel behind `kernel_tag`, MEASURED in this run: PMC counters cannot be read from inside
    the process, so two short CHILD runs of this same command (4 steps, no extra legs) are started under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes: the counters do not fit one; FETCH_SIZE doubled per
    the gfx950 note of MI355X_MICROARCH.md, both in KiB) and their per-kernel averages are mapped to bench tags by
    profiles/summarize.py.  A child is a new process started with Popen (never an exec from this GPU-initialised one).
    Returns (bytes or None, note)."""
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import summarize
    keep = []
    skip_next = False
    for a in argv:  # the parent's variant flags (workload, dtype, optimiser placement ...) minus what the child fixes
        if skip_next:
            skip_next = False
            continue
        if a in ("--steps", "--warmup", "--preheat-seconds", "--cpu-seconds", "--gpus"):
            skip_next = True
            continue
        keep.append(a)
    child = [sys.executable, os.path.abspath(__file__)] + keep + [
        "--steps", "4", "--warmup", "2", "--preheat-seconds", "0", "--no-cpu-baseline", "--no-prof", "--no-fused-leg",
        "--no-graph-leg", "--no-configs2-leg", "--no-spmm", "--no-sampling", "--no-live-traffic"]
    tmp = tempfile.mkdtemp(prefix="gdmcf_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    csvs = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            r = subprocess.run([rp, "--pmc", ctr, "--kernel-trace", "-d", out, "-o", "p", "--output-format", "csv", "--"] + child,
                               cwd=tmp, env=env, capture_output=True, text=True, timeout=240)
            found = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
            if r.returncode != 0 or not found:
                return None, f"rocprofv3 --pmc {ctr} child failed (rc {r.returncode}): {(r.stderr or '')[-160:]}"
            csvs[ctr] = found[0]
        rows = summarize.traffic(csvs["FETCH_SIZE"], csvs["WRITE_SIZE"])
        tags = summarize.by_tag(rows, "", "")["tags"]
        row = tags.get(kernel_tag)
        if row is None:
            return None, f"no kernel of tag '{kernel_tag}' in the PMC passes"
        return int(row["hbm_total_MB"] * 1e6), (f"measured in this run: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE child "
                                                f"passes of this command, {row['kernel']}, {row['launches']} launches: read "
                                                f"{row['hbm_read_MB']} MB + written {row['hbm_write_MB']} MB per launch")
    except Exception as exc:  # never fatal for the line
        return None, f"live traffic failed: {type(exc).__name__}: {exc}"[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="yelp", choices=["yelp", "amazon-book", "stress", "tiny"])
    ap.add_argument("--T", type=int, default=5, help="diffusion steps")
    ap.add_argument("--batch", type=int, default=400)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: split this many rows over the ranks (per-rank batch = G / N) instead of --batch per rank")
    ap.add_argument("--hidden", type=int, default=1000)
    ap.add_argument("--gemm-dtype", default="f32", choices=["f32", "bf16"],
                    help="input precision of the denoiser GEMMs (bf16 = BASELINE configs[2]; f32 is the parity path)")
    ap.add_argument("--dense-rows", action="store_true",
                    help="densify the batch on the device (gdmcf_densify_rows_f32) and hand the dense rows to training_losses as "
                         "the reference's loop does, instead of leaving them sparse (CsrBatch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-1thread", action="store_true", help="skip the one-thread CPU step (tens of seconds)")
    ap.add_argument("--preheat-seconds", type=float, default=1.0,
                    help="untimed: run dense products on scratch buffers for this long before the warm-up steps so that the timed "
                         "region does not start on a cold clock (0 = off; reported as clock_preheat)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with two rocprofv3 --pmc child passes of this command (N = 1 only; ~40 s)")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--prof-every", type=int, default=0,
                    help="bracket the tagged launches of every Nth timed step with HIP events.  An event pair keeps the next kernel "
                         "from starting behind the previous one: ~5 us per launch, ~80 us per bracketed step (rocprofv3 kernel "
                         "trace, profiles/README.md) -- every 4th step still inflates the line by 1.5 %%.  0 (default) = three "
                         "bracketed steps spread over the timed region (every max(4, ceil(steps / 3))th step)")
    ap.add_argument("--spmm", action="store_true", help="(default on) time the LightGCN SpMM, reported under 'spmm'")
    ap.add_argument("--no-spmm", action="store_true", help="skip the LightGCN SpMM leg")
    ap.add_argument("--spmm-only", action="store_true",
                    help="time ONLY the LightGCN propagation of --workload (yelp / amazon-book / stress: BASELINE configs[4]'s 1.2 M-node "
                         "graph) and print its line; no training step")
    ap.add_argument("--spmm-sharded", action="store_true",
                    help="N > 1: row-shard the adjacency over the ranks (local SpMM + all-gather per layer) instead of timing "
                         "the replica on rank 0")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling as the main line: global batch 400 (BASELINE configs[3]) split over the ranks; without "
                         "it the main line is weak scaling (400 rows per rank) and at N > 1 a short strong-scaling leg is "
                         "reported under 'strong_scaling'")
    ap.add_argument("--no-strong-leg", action="store_true", help="N > 1: skip the extra strong-scaling leg")
    ap.add_argument("--fuse-optimizer", action="store_true",
                    help="(the default at N = 1 with the plain denoiser since round 4) update the two large weights inside "
                         "their weight-gradient products (FusedAdamW.fuse_into_backward); same update rule, gradient never "
                         "materialised")
    ap.add_argument("--separate-optimizer", action="store_true",
                    help="N = 1: keep AdamW as a separate pass after the backward in the main line (what N > 1 always does: the "
                         "gradients are exchanged first)")
    ap.add_argument("--no-fused-leg", action="store_true",
                    help="N = 1: skip the extra leg that times the OTHER optimiser placement (separate pass when the main line "
                         "is fused, and vice versa)")
    ap.add_argument("--f32x3-leg", action="store_true",
                    help="N = 1, fp32: also time the step with gemm_dtype='f32x3' (opt-in product mode, DESIGN 4.4b)")
    ap.add_argument("--no-configs2-leg", action="store_true",
                    help="N = 1: skip the BASELINE configs[2] leg (Amazon-Book shape, bf16 GEMM inputs, B = 400)")
    ap.add_argument("--no-graph-leg", action="store_true", help="N = 1: skip the extra leg that replays the step from a hipGraph")
    ap.add_argument("--graph-dp", action="store_true",
                    help="N > 1: also time the data-parallel step (all-reduce exchange) replayed from one hipGraph per rank, RCCL "
                         "calls captured (default on with --rehearse-dp)")
    ap.add_argument("--bpr", action="store_true", help="also time a LightGCN BPR training step (reported under 'bpr')")
    ap.add_argument("--backbone", default="dnn", choices=["dnn", "onehot", "onehot-emb", "onehot-gcn"],
                    help="dnn: the plain denoiser (BASELINE configs); onehot: GaussianDiffusionDiscrete(CatOneHot=True) + "
                         "DNNOneHot; onehot-emb / onehot-gcn: the indexIn backbones DNNOneHotEmbedding / DNNOneHotEmbeddingGCN "
                         "(SURVEY 8 f1)")
    ap.add_argument("--rehearse-dp", action="store_true",
                    help="N = 1 only: create a one-rank RCCL group and run every data-parallel collective through it "
                         "(rehearsal of the N > 1 code path on a single-GPU box)")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="reduce-scatter + AdamW on 1/N of the rows + all-gather (the default when N > 1; with "
                         "--rehearse-dp it selects that path at N = 1)")
    ap.add_argument("--autotune-dp", action="store_true",
                    help="with --rehearse-dp: time both exchange variants during warm-up and keep the faster one (what "
                         "happens by default when N > 1 and neither variant is requested)")
    ap.add_argument("--allreduce-optimizer", action="store_true",
                    help="N > 1: all-reduce of the gradients + full AdamW on every rank instead of the sharded optimiser")
    ap.add_argument("--sampling", action="store_true", help="(default on) time p_sample + masked top-k, reported under 'sampling'")
    ap.add_argument("--no-sampling", action="store_true", help="skip the evaluation-path leg")
    args = ap.parse_args()
    if args.strong and not args.global_batch:
        args.global_batch = 400
    default_line = args.backbone == "dnn" and not args.rehearse_dp
    # single GPU, plain denoiser: AdamW of the two large weights runs inside their weight-gradient products unless asked otherwise
    args.fuse_main = (args.gpus == 1 and args.backbone == "dnn" and not args.rehearse_dp and not args.separate_optimizer) \
        or args.fuse_optimizer
    args.spmm = (args.spmm or default_line) and not args.no_spmm
    args.sampling = (args.sampling or default_line) and not args.no_sampling
    return args


def default_line_only(args):
    """the command the driver runs (plain denoiser, Yelp shape, fp32, no variant flags): only that line carries the extra legs"""
    return (args.backbone == "dnn" and args.workload == "yelp" and args.gemm_dtype == "f32" and not args.rehearse_dp
            and not args.separate_optimizer and not args.global_batch and args.batch == 400 and args.hidden == 1000)


def collect_prof(lib, cap=65536):
    tags = (ctypes.c_int * cap)()
    ms = (ctypes.c_float * cap)()
    work = (ctypes.c_double * cap)()
    n = lib.gdmcf_prof_collect(cap, tags, ms, work)
    out = {}
    for i in range(n):
        d = out.setdefault(int(tags[i]), dict(ms=0.0, work=0.0, n=0))
        d["ms"] += float(ms[i])
        d["work"] += float(work[i])
        d["n"] += 1
    return out


def kernel_table(kernels, gemm_dtype, B, hid, I, steps, n_profiled, el):
    """One entry per tagged kernel (HIP-event totals of the profiled steps), largest share of the step first: achieved
    rate on the ALGORITHMIC work, the peak that bounds it, the fraction, the average launch and its share of the step."""
    klist = []
    for tag, d in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"]):
        sec = d["ms"] * 1e-3
        extra = {}
        if tag in GEMM_TAGS and gemm_dtype == "bf16":
            # bf16 products at batch 400 are bound by moving operands and results, not by the matrix pipe: report
            # the HBM fraction on the compulsory bytes (each operand and the result once) and the bf16-MFMA
            # fraction beside it.
            E_ = 10
            ob = 2.0  # operands are streamed from their bf16 shadows (2 B/elem); results and the loss target are f32
            per_launch = {1: ob * (B + hid) * (I + E_) + 4.0 * B * hid,
                          2: ob * (B * hid + I * hid) + 4.0 * 2 * B * I + 2.0 * B * I,
                          3: ob * (B * hid + I * hid) + 4.0 * 2 * B * I,
                          4: ob * (B * I + I * hid) + 4.0 * B * hid,
                          5: (ob * (B * I + 2 * B * hid + B * (I + E_)) + 4.0 * (I * hid + hid * (I + E_))) / 2.0}[tag]
            ach, peak, unit, bound = per_launch * d["n"] / sec / 1e9, PEAK_HBM_GBPS, "GB/s", "hbm"
            extra = dict(mfma_tflops=round(d["work"] / sec / 1e12, 1),
                         mfma_frac=round(d["work"] / sec / 1e12 / PEAK_BF16_MATRIX_TFLOPS, 4))
        elif tag in GEMM_TAGS:
            ach, peak, unit, bound = d["work"] / sec / 1e12, PEAK_F32_MATRIX_TFLOPS, "TFLOP/s", "mfma"
        else:
            ach, peak, unit, bound = d["work"] / sec / 1e9, PEAK_HBM_GBPS, "GB/s", "hbm"
        klist.append(dict(kernel=TAGS.get(tag, str(tag)), bound=bound, achieved=round(ach, 2), peak=peak, unit=unit,
                          frac=round(ach / peak, 4), avg_ms=round(d["ms"] / d["n"], 4), launches=d["n"],
                          share_of_step=round(d["ms"] * steps / max(n_profiled, 1) / (el * 1e3), 4), **extra))
    return klist


def cpu_baseline(args, I, x_batches, seconds):
    """The oracle's train step on the host cores, same shape / same rows (bounded sample)."""
    from oracle import gdmcf_oracle as O
    torch.manual_seed(0)
    from gdmcf_amd import data as _data
    n_users = _data.SHAPES[args.workload]["n_users"]
    if args.backbone != "dnn":
        if args.backbone == "onehot":
            om = O.DNNOneHot([I, args.hidden], [args.hidden, I], 10)
        elif args.backbone == "onehot-emb":
            om = O.DNNOneHotEmbedding([I, args.hidden], [args.hidden, I], 10, item_num=I, user_num=n_users)
        else:
            om = O.DNNOneHotEmbeddingGCN([I, args.hidden], [args.hidden, I], 10, item_num=I, user_num=n_users)
        od = O.GaussianDiffusionDiscrete(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, args.T, CatOneHot=True)
        od.indexIn = args.backbone != "onehot"
    else:
        om = O.DNN([I, args.hidden], [args.hidden, I], 10)
        od = O.GaussianDiffusion(O.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, args.T)
    opt = O.make_optimizer(om, 1e-5)
    om.train()
    xs = [torch.from_numpy(b) for b in x_batches[:2]]
    Bc = xs[0].shape[0]
    extra = dict(index=torch.arange(Bc)) if args.backbone in ("onehot-emb", "onehot-gcn") else {}
    O.train_step(od, om, opt, xs[0], True, **extra)  # warm-up (allocations, thread pool)
    n, t0 = 0, time.perf_counter()
    while True:
        O.train_step(od, om, opt, xs[n % len(xs)], True, **extra)
        n += 1
        el = time.perf_counter() - t0
        if (el >= seconds and n >= 3) or n >= 50:
            break
    threads = torch.get_num_threads()
    out = dict(value=round(Bc * n / el, 2), unit="users/s", cores=threads, kind="port",
               sample=f"{n} train steps of B={Bc}, I={I}, dims=[{args.hidden}], T={args.T} (oracle, PyTorch-CPU eager, "
                      f"one process, {threads} intra-op threads)",
               ms_per_step=round(1e3 * el / n, 2), physical_cores=physical_cores(), logical_cpus=os.cpu_count())
    if not args.no_cpu_1thread:
        # orientation figure: the same step on ONE thread, one step only (it takes tens of seconds)
        torch.set_num_threads(1)
        t1 = time.perf_counter()
        O.train_step(od, om, opt, xs[0], True, **extra)
        e1 = time.perf_counter() - t1
        torch.set_num_threads(threads)
        out["one_thread"] = dict(value=round(Bc / e1, 2), unit="users/s", ms_per_step=round(1e3 * e1, 1), sample="1 train step")
    return out


def clock_preheat(lib, dev, seconds):
    """A fixed TIME of dense products on scratch operands (no training step; no model / optimiser / RNG state touched): brings the
    clock up after an idle period.  Returns the `clock_preheat` record of the line, or None when switched off."""
    from gdmcf_amd import _lib
    if seconds <= 0:
        return None
    pa = torch.randn(400, 4096, device=dev)
    pw = torch.randn(4096, 4096, device=dev)
    pc = torch.empty(400, 4096, device=dev)
    pws = torch.empty(max(int(lib.gdmcf_linear_ws_bytes(400, 4096, 4096)), 256), dtype=torch.uint8, device=dev)
    t_ph, n_ph = time.perf_counter(), 0
    while time.perf_counter() - t_ph < seconds:
        for _ in range(16):
            _lib.check(lib.gdmcf_linear_fwd_f32(pa.data_ptr(), 4096, pw.data_ptr(), 4096, None, 0, 400, 4096, 4096,
                                                pc.data_ptr(), 4096, pws.data_ptr(), pws.numel(), _lib.stream_ptr()))
        torch.cuda.synchronize()
        n_ph += 16
    return dict(seconds=round(time.perf_counter() - t_ph, 3), launches=n_ph,
                what="untimed dense products on scratch buffers before the warm-up steps (clock ramp after idle); "
                     "no training step, no model / optimiser / RNG state touched")


def prof_stride(steps, asked=0):
    """Which timed steps are bracketed with HIP events: every `asked`-th, or -- 0 -- three of them spread over the region.  The
    brackets cost ~80 us per step (each event pair holds the next launch back ~5 us); the kernels' averages need a handful of
    samples, not a quarter of the steps."""
    return max(1, asked) if asked > 0 else max(4, -(-steps // 3))


def physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo; None when the file does not say"""
    try:
        seen, phys = set(), None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                seen.add((phys, ln.split(":")[1].strip()))
        return len(seen) or None
    except OSError:
        return None


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: this parent -- which never touches the GPU -- starts N fresh child
    processes (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relays rank 0's JSON line and fails
    loudly when a rank fails -- e.g. because fewer than N GPUs are visible to it."""
    import socket
    import subprocess
    n = args.gpus
    # a profiler's preloaded library has initialised the GPU in THIS process before main() ran: starting the ranks from here
    # would be an exec from a GPU-initialised process (forbidden on this pool).  Profile one rank started directly instead.
    preload = " ".join(os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_LIBRARY_PATH"))
    if any(tag in preload.lower() for tag in ("rocprof", "roctracer", "rocprofiler")) and not os.environ.get("GDMCF_BENCH_DRY_RUN"):
        raise SystemExit("bench.py --gpus N under a profiler: the parent would have to start ranks from a GPU-initialised "
                         "process.  Profile a single rank (rocprofv3 ... -- python bench.py) or launch the ranks with "
                         "torch.distributed.run and profile inside them.")
    # (the parent does not even COUNT devices: torch.cuda.device_count() can fall back to hipGetDeviceCount, i.e. initialise the
    # GPU in the process that is about to start the ranks; every rank checks the count itself and exits non-zero)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    codes = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    if any(codes):
        raise SystemExit(f"bench.py --gpus {n}: rank exit codes {codes}")
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if not lines or json.loads(lines[-1]).get("n_gpus") != n:
        raise SystemExit(f"bench.py --gpus {n}: rank 0 did not report n_gpus == {n}")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not args.rehearse_dp:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if os.environ.get("GDMCF_BENCH_DRY_RUN"):
        # launcher rehearsal on a box without GPUs (tests/test_host_cpu.py): the ranks meet in a gloo group, nothing is timed
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        one = torch.ones(1)
        dist.all_reduce(one)
        if rank == 0:
            print(json.dumps({"metric": "training users/sec", "value": None, "n_gpus": world, "dry_run": True,
                              "ranks_in_group": int(one.item())}))
        dist.destroy_process_group()
        return
    # GDMCF_BENCH_SHARE_GPU=1 (tests only: the rehearsal of the N > 1 launcher and step on a one-GPU box): the ranks share the
    # visible devices (rank r -> device r % count) and meet in a gloo group, collectives staged through the host; the line says
    # so (`rehearsal`).  Without it a rank that does not find a GPU of its own refuses to run.
    share = os.environ.get("GDMCF_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < (1 if share else max(world, 1)):
        raise SystemExit(f"bench.py --gpus {args.gpus}: rank {rank} sees {have} GPU(s) visible; refusing to report an "
                         f"N-GPU line from fewer GPUs than ranks")
    if share:
        local = local % have
    if world > 1 or args.rehearse_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL's kernels compete with MFMA-bound GEMMs for CUs while the exchange overlaps the backward: run them on a
        # high-priority stream (read by ProcessGroupNCCL when the group is created; an explicit setting wins)
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        if share and world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    if args.spmm_only:
        # the SpMM leg alone (reference lightGCN.py:180-194): one line in the contract's shape, metric = layers per second
        import gdmcf_amd
        from gdmcf_amd import _lib
        sp_ = bench_spmm(gdmcf_amd, _lib.load(), args.workload, dev, world=world if args.spmm_sharded else 1)
        if rank == 0:
            print(json.dumps({"metric": "LightGCN propagation layers/sec", "value": round(1e3 / sp_["ms_per_layer"], 1), "unit": "layers/s",
                              "n_gpus": world, "steps": 20, "warmup": 3, "ms_per_step": sp_["ms_per_layer"], "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": f"{args.workload}-shape synthetic bipartite graph, d=64, 3 layers "
                                                     "(lightGCN.py:180-194)" + (" (BASELINE configs[4]: HBM-bound SpMM stress)"
                                                                                if args.workload == "stress" else "")},
                              "roofline": {k: sp_[k] for k in ("bound", "achieved", "peak", "unit", "frac")}, "spmm": sp_,
                              "spmm_split": os.environ.get("GDMCF_SPMM_SPLIT", "0")}))
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return

    import gdmcf_amd
    from gdmcf_amd import _lib, data
    from gdmcf_amd.parallel import DataParallelStep
    if os.environ.get("GDMCF_PROBE_LIB"):  # a probe build of the library (tools/build_variant.sh): experiments only, the line says so
        _lib.LIB_PATH = os.path.abspath(os.environ["GDMCF_PROBE_LIB"])
    lib = _lib.load()

    B, hid, T = args.batch, args.hidden, args.T
    strong = args.global_batch > 0
    if strong:
        if args.global_batch % world:
            raise SystemExit("--global-batch must be divisible by the number of ranks")
        B = args.global_batch // world
    n_pool = 4
    indptr, indices, I = data.synth_csr(args.workload, n_rows=(world * n_pool) * B, seed=0)
    lo = rank * n_pool * B
    sub_ptr = indptr[lo:lo + n_pool * B + 1] - indptr[lo]
    sub_idx = indices[indptr[lo]:indptr[lo + n_pool * B]]
    x_host = data.dense_batches(sub_ptr, sub_idx, I, B, n_pool)  # only for the CPU baseline / sampling legs
    # the interaction matrix is resident in HBM as CSR before the timed region; every step densifies its own
    # 400 rows on the device (gdmcf_densify_rows_f32) -- the batch provider is part of the step
    import scipy.sparse as sp
    from gdmcf_amd.data_utils import DeviceCSR
    dcsr = DeviceCSR(sp.csr_matrix((np.ones(len(sub_idx), np.float32), sub_idx, sub_ptr), shape=(n_pool * B, I)), dev)
    row_ids = [torch.arange(i * B, (i + 1) * B, device=dev) for i in range(n_pool)]
    x_buf = torch.empty(B, I, dtype=torch.float32, device=dev)
    x_dev = torch.from_numpy(x_host[:1]).to(dev)
    # what the step is handed: the rows left sparse (default for the plain denoiser: the CSR-fed input builder and the bitmap
    # loss target, gdmcf_amd.data_utils.CsrBatch -- bit-identical to the dense path), or densified on the device first
    # (--dense-rows; always for the one-hot backbones, which read the dense row)
    sparse_rows = args.backbone == "dnn" and not args.dense_rows

    def rows_of(k):
        return dcsr.batch(row_ids[k]) if sparse_rows else dcsr.rows(row_ids[k], out=x_buf)

    torch.manual_seed(0)
    if args.backbone != "dnn":
        if args.fuse_optimizer:
            raise SystemExit("--backbone onehot*: separate AdamW pass only")
        args.fuse_main = False
        if args.backbone == "onehot":
            model = gdmcf_amd.DNNOneHot([I, hid], [hid, I], 10, time_type="cat", norm=False, gemm_dtype=args.gemm_dtype).to(dev)
        else:
            cls = gdmcf_amd.DNNOneHotEmbedding if args.backbone == "onehot-emb" else gdmcf_amd.DNNOneHotEmbeddingGCN
            model = cls([I, hid], [hid, I], 10, time_type="cat", norm=False, item_num=I,
                        user_num=data.SHAPES[args.workload]["n_users"], gemm_dtype=args.gemm_dtype).to(dev)
        diffusion = gdmcf_amd.GaussianDiffusionDiscrete(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T,
                                                        dev, CatOneHot=True)
        diffusion.indexIn = args.backbone != "onehot"  # main.py:241, :245
    else:
        model = gdmcf_amd.DNN([I, hid], [hid, I], 10, time_type="cat", norm=False, gemm_dtype=args.gemm_dtype).to(dev)
        diffusion = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, dev)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-5, weight_decay=0.0)
    model.train()
    torch.manual_seed(1234 + rank)  # (before anything creates the model's engine: its Philox seed is torch.initial_seed() then)
    fuse_main = bool(args.fuse_main and world == 1)
    if fuse_main:
        opt.fuse_into_backward(model)
    # N > 1: the same bytes cross xGMI either way (reduce-scatter + all-gather == all-reduce), but the sharded
    # optimiser touches 1/N of the AdamW state per GPU and its all-gathers overlap the next step's first GEMMs
    sharded = (args.shard_optimizer or world > 1) and not args.allreduce_optimizer
    # neither flag given at N > 1: both variants are timed during warm-up (untimed region) and the faster one runs
    autotune = (world > 1 or args.autotune_dp) and not (args.shard_optimizer or args.allreduce_optimizer)
    step = DataParallelStep(diffusion, model, opt, shard_optimizer=sharded and not autotune, force_exchange=args.rehearse_dp)

    step_kw = [dict(index=r + lo) if args.backbone in ("onehot-emb", "onehot-gcn") else {} for r in row_ids]  # user ids

    def sync():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # ---- clock pre-heat (untimed, not a training step): after an idle period the chip runs its first tens of milliseconds
    # at 2.1-2.2 GHz instead of 2.38 GHz (in-kernel s_memtime / s_memrealtime stamps); a timed region of 20 steps
    # = 32 ms behind 5 warm-up steps = 8 ms would measure the ramp, which no training run longer than a blink sees.  A fixed
    # TIME of dense products on scratch operands brings the clock up; the model, the optimiser and the random streams are
    # not touched (the W warm-up steps that follow are the only steps before the timed ones).  Reported as `clock_preheat`.
    preheat = clock_preheat(lib, dev, args.preheat_seconds)
    loss = None
    trace = os.environ.get("GDMCF_BENCH_TRACE") == "1"  # debugging: the loss of every step on stderr (synchronises each step)
    for i in range(args.warmup):
        loss = step(rows_of(i % n_pool), True, **step_kw[i % n_pool])
        if trace:
            print(f"[trace] warm-up step {i}: loss {float(loss)!r}", file=sys.stderr)
    sync()
    dp_autotune = None
    if autotune and step.exchange:
        # untimed: a few steps of each exchange variant on this node's links, max over ranks, keep the faster one
        trial, failed = {}, None
        for name, flag in (("allreduce", False), ("sharded", True)):
            try:
                step.set_shard_optimizer(flag)
                for i in range(2):
                    step(rows_of(i % n_pool), True, **step_kw[i % n_pool])
                sync()
                t1 = time.perf_counter()
                for i in range(6):
                    step(rows_of(i % n_pool), True, **step_kw[i % n_pool])
                sync()
                tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                trial[name] = float(tt) / 6 * 1e3
            except Exception as exc:  # a variant this RCCL build refuses must not cost the whole run
                if name == "allreduce":
                    raise
                failed = f"{type(exc).__name__}: {exc}"[:200]
                print(f"[bench] sharded exchange failed on rank {rank}: {failed}", file=sys.stderr)
                trial[name] = float("inf")
        sharded = trial["sharded"] <= trial["allreduce"]
        step.set_shard_optimizer(sharded)
        sync()
        dp_autotune = dict(allreduce_ms_per_step=round(trial["allreduce"], 4),
                           sharded_ms_per_step=None if failed else round(trial["sharded"], 4),
                           chosen="sharded" if sharded else "allreduce")
        if failed:
            dp_autotune["sharded_error"] = failed
    prof = not args.no_prof
    every = prof_stride(args.steps, args.prof_every)
    n_profiled = len(range(0, args.steps, every)) if prof else 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        if prof:
            lib.gdmcf_prof_enable(1 if i % every == 0 else 2)  # 2 = pause, records kept
        loss = step(rows_of(i % n_pool), True, **step_kw[i % n_pool])
        if trace:
            print(f"[trace] step {i}: loss {float(loss)!r}", file=sys.stderr)
    host_el = time.perf_counter() - t0  # enqueue time only: well below `el` when the host runs ahead of the GPU
    sync()
    el = time.perf_counter() - t0
    kernels = collect_prof(lib) if prof else {}
    lib.gdmcf_prof_enable(0)
    final_loss = float(loss)
    if dist.is_initialized():
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t)

    # ---- data parallel: every replica must hold bit-identical parameters after the run (outside the timed region) ----
    in_sync = None
    if dist.is_initialized():
        step.flush()
        cs = torch.stack([p.detach().double().sum() for p in model.parameters()] +
                         [p.detach().double().abs().max() for p in model.parameters()])
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool(torch.equal(lo, hi))

    # ---- the evaluation-path and SpMM legs run on the model as the main line left it (`final_loss` describes it); the extra
    # N = 1 legs below keep training it and come afterwards.  At N > 1 rank 0 runs them alone, after the collective legs. ----
    legs = {}

    def eval_legs():
        if args.spmm and (rank == 0 or (world > 1 and args.spmm_sharded)):
            legs["spmm"] = bench_spmm(gdmcf_amd, lib, args.workload, dev, world=world if args.spmm_sharded else 1)
        if args.bpr and rank == 0:
            legs["bpr"] = bench_bpr(gdmcf_amd, args.workload, dev)
        if args.sampling and rank == 0:
            legs["sampling"] = bench_sampling(gdmcf_amd, lib, model, diffusion, x_dev[0], sub_ptr[:B + 1], sub_idx, dev)

    if world == 1:
        eval_legs()

    # ---- N > 1: strong-scaling leg (BASELINE configs[3] as stated: global batch 400 split over the ranks) ----
    strong_leg = None
    if world > 1 and not strong and not args.no_strong_leg and 400 % world == 0:
        Bs = 400 // world
        ids_s = [r[:Bs] for r in row_ids]
        xs_buf = torch.empty(Bs, I, dtype=torch.float32, device=dev)

        def rows_of_s(k):
            return dcsr.batch(ids_s[k]) if sparse_rows else dcsr.rows(ids_s[k], out=xs_buf)
        for i in range(max(3, args.warmup // 4)):
            step(rows_of_s(i % n_pool), True, **{k: v[:Bs] for k, v in step_kw[i % n_pool].items()})
        sync()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(rows_of_s(i % n_pool), True, **{k: v[:Bs] for k, v in step_kw[i % n_pool].items()})
        sync()
        ts_ = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(ts_, op=dist.ReduceOp.MAX)
        strong_leg = dict(global_batch=400, batch_per_gpu=Bs, steps=args.steps, ms_per_step=round(1e3 * float(ts_) / args.steps, 4),
                          users_per_s=round(400 * args.steps / float(ts_), 1), scaling="strong",
                          what=f"BASELINE configs[3]: Yelp_clean batch=400 steps=5 split over {world} GPUs ({Bs} rows per GPU), "
                               "data parallel, RCCL gradient exchange over xGMI")
        step.flush()

    # ---- N = 1: the same step with AdamW inside the weight-gradient GEMM epilogues (FusedAdamW.fuse_into_backward:
    # same update rule, bit-identical weights by test; opt-in because `.grad` of the two big weights is then never
    # materialised).  Reported beside the main line, which keeps backward(); optimizer.step() as the reference's loop. ----
    fused_leg = separate_leg = None
    if world == 1 and args.backbone == "dnn" and not args.rehearse_dp and not args.no_fused_leg:
        if fuse_main:
            opt.fuse_into_backward(model, min_numel=1 << 62)  # off: the separate pass
        else:
            opt.fuse_into_backward(model)
        for i in range(max(3, args.warmup // 4)):
            step(rows_of(i % n_pool), True)
        sync()
        t1 = time.perf_counter()
        for i in range(args.steps):
            loss_f = step(rows_of(i % n_pool), True)
        sync()
        ef = time.perf_counter() - t1
        other = dict(ms_per_step=round(1e3 * ef / args.steps, 4), users_per_s=round(B * args.steps / ef, 1), steps=args.steps,
                     final_loss=float(loss_f))
        if fuse_main:
            separate_leg = dict(other, what="the same step with AdamW as a separate pass after the backward (28 B/param, one "
                                            "launch): bench.py --separate-optimizer makes it the main line; N > 1 always runs it")
            opt.fuse_into_backward(model)  # back to the main line's placement
        else:
            fused_leg = dict(other, what="AdamW of the two large weights inside their weight-gradient products (the default main "
                                         "line at N = 1)")
            opt.fuse_into_backward(model, min_numel=1 << 62)
    if fuse_main:
        fused_leg = dict(ms_per_step=round(1e3 * el / args.steps, 4), users_per_s=round(world * B * args.steps / el, 1),
                         steps=args.steps, is_main_line=True,
                         what="the main line: AdamW of the two large weights inside their weight-gradient products")

    # ---- N = 1: the same step captured once in a hipGraph and replayed (gdmcf_amd/graph.py: counters and AdamW scalars in
    # device memory, batch = a fixed id buffer over the resident CSR matrix; bit-identical to the eager step by test).
    # `host_enqueue_ms` is what the host spends per step: one 3 KB id copy + one graph launch. ----
    graph_leg = None
    # With --rehearse-dp (one-rank RCCL group) the captured body includes every collective of the data-parallel step; at N > 1
    # the leg is opt-in (--graph-dp): all ranks capture the same collectives.
    graph_dp = (args.rehearse_dp and world == 1) or (world > 1 and args.graph_dp)
    if (world == 1 or graph_dp) and args.backbone == "dnn" and sparse_rows \
            and (not args.rehearse_dp or graph_dp) and not args.no_graph_leg:
        from gdmcf_amd.graph import GraphedTrainStep
        try:
            step.flush()
            with GraphedTrainStep(diffusion, model, opt, dcsr, B, warmup=3, force_exchange=args.rehearse_dp) as gstep:
                for i in range(max(5, args.warmup // 4)):
                    gstep(row_ids[i % n_pool])
                sync()
                t1 = time.perf_counter()
                for i in range(args.steps):
                    loss_g = gstep(row_ids[i % n_pool])
                hg = time.perf_counter() - t1
                sync()
                eg = time.perf_counter() - t1
                captured, cap_err = isinstance(gstep.graph, torch.cuda.CUDAGraph), gstep.capture_error
            if dist.is_initialized():
                tg = torch.tensor([eg, hg], dtype=torch.float64, device=dev)
                dist.all_reduce(tg, op=dist.ReduceOp.MAX)
                eg, hg = float(tg[0]), float(tg[1])
            graph_leg = dict(captured=captured, capture_error=cap_err, exchange=bool(gstep.step.exchange),
                             ms_per_step=round(1e3 * eg / args.steps, 4), users_per_s=round(world * B * args.steps / eg, 1),
                             host_enqueue_ms=round(1e3 * hg / args.steps, 4), eager_host_enqueue_ms=round(1e3 * host_el / args.steps, 4),
                             steps=args.steps, final_loss=float(loss_g),
                             optimizer="fused into the weight-gradient products" if fuse_main else "separate pass",
                             what="the training step (the main line's) replayed from one hipGraph (gdmcf_amd.graph.GraphedTrainStep)")
        except Exception as exc:  # reported, never fatal for the main line
            graph_leg = dict(error=f"{type(exc).__name__}: {exc}"[:300])

    # ---- N = 1, fp32: the same step with the dense products in "f32x3" mode (float32 operands split on chip into three
    # bfloat16 terms, six bf16 MFMAs per block, f32 accumulate: f32-level error -- tests/test_gpu_split.py measures it against
    # float64 beside the native f32 MFMA kernels).  Reported beside the main line, which stays on v_mfma_f32_16x16x4_f32. ----
    x3_leg = None
    if world == 1 and args.backbone == "dnn" and args.gemm_dtype == "f32" and not args.rehearse_dp \
            and args.f32x3_leg:
        model.gemm_dtype = "f32x3"
        try:
            for i in range(max(3, args.warmup // 4)):
                step(rows_of(i % n_pool), True)
            sync()
            t1 = time.perf_counter()
            for i in range(args.steps):
                loss_x = step(rows_of(i % n_pool), True)
            sync()
            ex = time.perf_counter() - t1
            x3_leg = dict(ms_per_step=round(1e3 * ex / args.steps, 4), users_per_s=round(B * args.steps / ex, 1), steps=args.steps,
                          final_loss=float(loss_x), what="dense products as three-term bf16 splits on the bf16 matrix pipe "
                          "(DNN(gemm_dtype='f32x3'), gemm_split.hip); f32-level error, opt-in")
        finally:
            model.gemm_dtype = "f32"

    # ---- roofline of the dominant kernel (rank 0's events) ----
    roofline = None
    klist = kernel_table(kernels, args.gemm_dtype, B, hid, I, args.steps, n_profiled, el)
    if fuse_main:
        # the weight-gradient launches also carry the optimiser stream of their weight (W, exp_avg, exp_avg_sq read and written
        # once: 24 B per parameter): report its rate beside the matrix rate -- the launch is bound by max(MFMA time, stream time)
        for kk in klist:
            if kk["kernel"] == "bwd_weight_gemm":
                byt = 24.0 * (I * hid + hid * (I + 10)) / 2.0
                kk["kernel"] = "bwd_weight_gemm + AdamW stream"
                kk["optimizer_stream"] = dict(bytes_per_launch=int(byt), achieved=round(byt / (kk["avg_ms"] * 1e-3) / 1e9, 1),
                                              peak=PEAK_HBM_GBPS, unit="GB/s", frac=round(byt / (kk["avg_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                                              what="24 B per parameter of the updated weight, inside the same launch")
    if klist:
        k0 = klist[0]
        tag0 = k0["kernel"].split(" + ")[0]
        traffic, traffic_note = (None, "off (--no-live-traffic)")
        if world == 1 and rank == 0 and not args.no_live_traffic and not args.rehearse_dp:
            traffic, traffic_note = live_traffic(tag0, sys.argv[1:])
        if traffic is None:
            t2, n2 = measured_traffic(tag0, args.workload, args.gemm_dtype)
            traffic, traffic_note = t2, f"{traffic_note}; fallback: {n2}"
        roofline = dict(bound=k0["bound"], achieved=k0["achieved"], peak=k0["peak"], unit=k0["unit"], frac=k0["frac"],
                        traffic=traffic, traffic_source=traffic_note, kernel=k0["kernel"], avg_ms=k0["avg_ms"],
                        launches_per_step=k0["launches"] // max(n_profiled, 1), profiled_steps=n_profiled,
                        traffic_unit="HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) + WRITE_SIZE, separate passes",
                        algorithmic_unit=("2*M*N*K FLOP per launch" if k0["bound"] == "mfma" else
                                          "compulsory bytes per launch (operands and results once; AdamW 28 B/param)"))
        if "optimizer_stream" in k0:
            roofline["optimizer_stream"] = k0["optimizer_stream"]

    if world > 1:
        eval_legs()
    spmm, bpr, sampling = legs.get("spmm"), legs.get("bpr"), legs.get("sampling")

    # ---- N = 1, default line: BASELINE configs[2] (Amazon-Book shape, bf16 GEMM inputs) so that the driver's record carries it ----
    configs2_leg = None
    if world == 1 and default_line_only(args) and not args.no_configs2_leg:
        try:
            configs2_leg = bench_configs2(gdmcf_amd, lib, dev, args.steps, max(3, args.warmup // 4), preheat_seconds=args.preheat_seconds)
        except Exception as exc:  # reported, never fatal for the main line
            configs2_leg = dict(error=f"{type(exc).__name__}: {exc}"[:300])

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, I, x_host, args.cpu_seconds)

    if rank == 0:
        users = world * B * args.steps
        out = {
            "metric": "training users/sec", "value": round(users / el, 1), "unit": "users/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.gemm_dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}-shape synthetic rows, batch={B}/GPU, dims=[{hid}], T={T}, "
                                   f"noise_scale=0.01, linear-var, mean_type=x0, reweight, AdamW lr=1e-5"
                                   + (", backbone DNNOneHot under GaussianDiffusionDiscrete(CatOneHot) (SURVEY 8 f1)"
                                      if args.backbone == "onehot" else
                                      ", backbone DNNOneHotEmbedding under GaussianDiffusionDiscrete(CatOneHot, indexIn) "
                                      "(SURVEY 8 f1)" if args.backbone == "onehot-emb" else
                                      ", backbone DNNOneHotEmbeddingGCN (the shipped YAML's; parity unpinned, SURVEY 8 f1)"
                                      if args.backbone == "onehot-gcn" else "")
                                   + (" (BASELINE configs[1])" if args.workload == "yelp" and T == 5 and hid == 1000
                                      and args.gemm_dtype == "f32" and args.backbone == "dnn" else "")
                                   + (" (BASELINE configs[2]: bf16 denoiser GEMM inputs, f32 accumulate/state)"
                                      if args.workload == "amazon-book" and args.gemm_dtype == "bf16" else ""),
                       "n_items": I, "global_batch": world * B, "parallelism": f"dp{world}",
                       "batch_rows": "device CSR rows (CsrBatch)" if sparse_rows else "dense rows densified from the device CSR"},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": klist, "final_loss": final_loss,
            "host_enqueue_ms_per_step": round(1e3 * host_el / args.steps, 4), "clock_preheat": preheat,
            "probe_lib": os.environ.get("GDMCF_PROBE_LIB") or None,
            "rehearsal": ("ranks share the visible GPU(s), gloo group with host-staged collectives (GDMCF_BENCH_SHARE_GPU=1): "
                          "launcher / step rehearsal, not a measurement") if (share and world > 1) else None,
            "replicas_in_sync": in_sync, "dp_autotune": dp_autotune, "strong_scaling": strong_leg, "fused_optimizer_leg": fused_leg, "separate_optimizer_leg": separate_leg, "graph_leg": graph_leg, "f32x3_leg": x3_leg,
            "configs2_leg": configs2_leg,
            "ranks_in_group": dist.get_world_size() if dist.is_initialized() else 1,
            "optimizer": "FusedAdamW" + (" inside the weight-gradient products of the two large weights (optimiser stream interleaved "
                                         "into their k loops); separate pass for the small tensors" if fuse_main else
                                         " (row-sharded over the ranks: reduce-scatter, AdamW on 1/N rows, deferred all-gather)"
                                         if (sharded and step.exchange) else
                                         " (separate pass after the gradient all-reduce)" if step.exchange else " (separate pass)"),
        }
        if cpu:
            out["speedup_vs_cpu"] = round(out["value"] / cpu["value"], 1)
        if spmm:
            out["spmm"] = spmm
        if sampling:
            out["sampling"] = sampling
        if bpr:
            out["bpr"] = bpr
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()  # rank 0 runs the extra legs (SpMM, evaluation path) alone: leave together
        dist.destroy_process_group()


def bench_sampling(gdmcf_amd, lib, model, diffusion, x, indptr, indices, dev, iters=10, k=100):
    """Evaluation path of reference main.py:288-301: p_sample (T denoiser forwards, steps=0) + history mask +
    top-100, for one 400-user batch; users/s and the per-kernel split."""
    model.eval()
    ip = torch.from_numpy(np.asarray(indptr, dtype=np.int64)).to(dev)
    ix = torch.from_numpy(np.asarray(indices[:int(indptr[-1])], dtype=np.int32)).to(dev)
    for _ in range(2):
        pred = diffusion.p_sample(model, x, 0, False)
        idx = gdmcf_amd.masked_topk(pred, k, ip, ix)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(iters):
        lib.gdmcf_prof_enable(1 if it % 4 == 0 else 2)  # HIP-event brackets on three of the ten batches (~5 us per launch)
        pred = diffusion.p_sample(model, x, 0, False)
        idx = gdmcf_amd.masked_topk(pred, k, ip, ix)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / iters
    prof = collect_prof(lib)
    lib.gdmcf_prof_enable(0)
    model.train()
    out = dict(ms_per_batch=round(el * 1e3, 3), users_per_s=round(x.shape[0] / el, 1), topk=k, T=diffusion.steps)
    for tag, name in ((1, "hidden_gemm"), (3, "posterior_gemm"), (7, "prep_input"), (9, "topk")):
        if tag in prof:
            d = prof[tag]
            out[name + "_avg_ms"] = round(d["ms"] / d["n"], 4)
            if tag == 9:
                out["topk_GBps"] = round(d["work"] / (d["ms"] * 1e-3) / 1e9, 1)
    return out


def bench_configs2(gdmcf_amd, lib, dev, steps, warmup, B=400, hid=1000, T=5, prof_every=0, preheat_seconds=0.0):
    """BASELINE configs[2] beside the main line: "Amazon-Book_clean batch=400 dims=[1000] steps=5, 1xMI355X, bf16 denoiser GEMM
    on MFMA" -- the same training step (zero_grad -> training_losses -> mean -> backward -> AdamW.step) on synthetic rows of
    the Amazon-Book shape (I = 94 949), dense products with bf16-rounded inputs on v_mfma_f32_16x16x32_bf16, f32 accumulation,
    f32 master weights / gradients / AdamW state.  Own model, own optimizer, own rows; kernels timed with HIP events like
    the main line.  Also timed: the same step with AdamW inside the weight-gradient epilogues."""
    import scipy.sparse as sp
    from gdmcf_amd import data
    from gdmcf_amd.data_utils import DeviceCSR
    from gdmcf_amd.parallel import DataParallelStep
    n_pool = 4
    prof_every = prof_stride(steps, prof_every)
    indptr, indices, I = data.synth_csr("amazon-book", n_rows=n_pool * B, seed=0)
    dcsr = DeviceCSR(sp.csr_matrix((np.ones(len(indices), np.float32), indices, indptr), shape=(n_pool * B, I)), dev)
    row_ids = [torch.arange(i * B, (i + 1) * B, device=dev) for i in range(n_pool)]
    torch.manual_seed(0)
    model = gdmcf_amd.DNN([I, hid], [hid, I], 10, time_type="cat", norm=False, gemm_dtype="bf16").to(dev).train()
    diffusion = gdmcf_amd.GaussianDiffusion(gdmcf_amd.ModelMeanType.START_X, "linear-var", 0.01, 0.001, 0.01, T, dev)
    opt = gdmcf_amd.FusedAdamW(model.parameters(), lr=1e-5, weight_decay=0.0)
    step = DataParallelStep(diffusion, model, opt)
    torch.manual_seed(4321)

    def timed(n, prof):
        for i in range(max(3, warmup)):
            step(dcsr.batch(row_ids[i % n_pool]), True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            if prof:
                lib.gdmcf_prof_enable(1 if i % prof_every == 0 else 2)
            loss = step(dcsr.batch(row_ids[i % n_pool]), True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        k = collect_prof(lib) if prof else {}
        lib.gdmcf_prof_enable(0)
        return el, k, float(loss)

    # main figure of the leg: AdamW of the two large weights inside their weight-gradient products (the default optimiser
    # placement at N = 1 since round 4); the separate pass is timed beside it
    opt.fuse_into_backward(model)
    # the leg starts behind the main line's rocprofv3 child passes (tens of seconds without GPU work in this process): the same
    # untimed clock pre-heat as the main line before its warm-up steps
    leg_preheat = clock_preheat(lib, dev, preheat_seconds)
    el, kernels, loss = timed(steps, True)
    klist = kernel_table(kernels, "bf16", B, hid, I, steps, len(range(0, steps, prof_every)), el)
    opt.fuse_into_backward(model, min_numel=1 << 62)
    els, ks, losss = timed(steps, True)
    kls = kernel_table(ks, "bf16", B, hid, I, steps, len(range(0, steps, prof_every)), els)
    # the fused step's dominant kernel is the weight-gradient product WITH the optimiser stream in its epilogue: HBM-bound,
    # 26 B per parameter (W, exp_avg, exp_avg_sq read and written, the bf16 shadow written) + the bf16 operands once
    dom = None
    kw = kernels.get(5)
    if kw:
        avg = kw["ms"] / kw["n"]
        byt = 26.0 * I * hid + 2.0 * B * (I + hid)
        dom = dict(kernel="bwd_weight_gemm + AdamW epilogue", bound="hbm", avg_ms=round(avg, 4), launches_per_step=2,
                   achieved=round(byt / (avg * 1e-3) / 1e9, 1), peak=PEAK_HBM_GBPS, unit="GB/s",
                   frac=round(byt / (avg * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                   share_of_step=round(2 * avg / (1e3 * el / steps), 4),
                   algorithmic_unit="26 B per parameter of the [I, hid] weight + its bf16 operands once, per launch")
    k0s = kls[0] if kls else None
    out = dict(what="BASELINE configs[2]: Amazon-Book-shape synthetic rows, batch=400, dims=[1000], T=5, bf16 denoiser GEMM inputs on "
                    "the bf16 MFMA (f32 accumulate, f32 master weights and AdamW state), 1 GPU; AdamW of the two large weights inside "
                    "their weight-gradient products (FusedAdamW.fuse_into_backward, the N = 1 default)",
               n_items=I, dtype="bf16", steps=steps, ms_per_step=round(1e3 * el / steps, 4), users_per_s=round(B * steps / el, 1),
               clock_preheat=leg_preheat, final_loss=loss, kernels=klist, optimizer="fused into the weight-gradient products", dominant_kernel=dom,
               fused_optimizer=dict(ms_per_step=round(1e3 * el / steps, 4), users_per_s=round(B * steps / el, 1), final_loss=loss,
                                    dominant_kernel=dom, is_main_figure=True),
               separate_optimizer=dict(ms_per_step=round(1e3 * els / steps, 4), users_per_s=round(B * steps / els, 1), final_loss=losss,
                                       dominant_kernel=None if k0s is None else dict(
                                           kernel=k0s["kernel"], bound=k0s["bound"], achieved=k0s["achieved"], peak=k0s["peak"],
                                           unit=k0s["unit"], frac=k0s["frac"], avg_ms=k0s["avg_ms"], share_of_step=k0s["share_of_step"],
                                           algorithmic_unit="AdamW: 30 B/param (28 + the bf16 shadow of the two large weights)"),
                                       what="the same step with AdamW as a separate pass (backward(); optimizer.step())"))
    del step, opt, model, dcsr
    torch.cuda.empty_cache()
    return out


def bench_spmm(gdmcf_amd, lib, workload, dev, layers=3, d=64, iters=20, world=1):
    """LightGCN propagation over the whole synthetic graph: ms per layer and HBM-roofline fraction
    (algorithmic bytes = nnz*8 + (N+1)*8 + 2*N*d*4; rowptr is int64 here).  With world > 1 the adjacency is
    row-sharded over the ranks (LightGCN(shard_rows=True): local SpMM + all-gather per layer); `ms_per_layer` then is
    the SpMM kernel of this rank's block, `ms_per_propagation` the wall time of all layers including the all-gathers."""
    from gdmcf_amd import data
    cfg = data.SHAPES[workload]
    indptr, indices, I = data.synth_csr(workload, seed=0)
    users = np.repeat(np.arange(cfg["n_users"]), np.diff(indptr))
    torch.manual_seed(0)  # identical E0 on every rank
    m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": indices}, cfg["n_users"], I, layers, d, device=dev,
                           shard_rows=world > 1).to(dev)
    nnz = m.nnz
    N = cfg["n_users"] + I
    torch.set_grad_enabled(False)
    for _ in range(3):
        m.propagate_through_layers()
    torch.cuda.synchronize()
    lib.gdmcf_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(iters):
        m.propagate_through_layers()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / iters
    k = collect_prof(lib).get(8)
    lib.gdmcf_prof_enable(0)
    torch.set_grad_enabled(True)
    ms = k["ms"] / k["n"]
    alg = m.algorithmic_bytes()
    gbps = alg / (ms * 1e-3) / 1e9
    sched = "streamed (one launch + combine)" if getattr(m, "_streamed", False) else \
        "bundled" if getattr(m, "_bundled", False) else "virtual rows (short / long / combine launches)"
    # second fraction: what the kernel has to move is one d*4-byte row PER NONZERO; the chip's rate for random
    # rows of that size depends on the table size (tools/gather_probe.hip, profiles/r02_spmm_gather_probe*.txt)
    table_mb = N * d * 4 / 1e6
    gather_peak = float(np.interp(table_mb, [4.0, 16.8, 33.0, 268.0], [32.0, 10.0, 8.5, 7.3]))
    gathered = nnz * d * 4 / (ms * 1e-3) / 1e12
    return dict(ms_per_layer=round(ms, 4), ms_per_propagation=round(wall_ms, 4), row_shards=world, nnz=nnz, nodes=N, d=d,
                schedule=sched, algorithmic_MB=round(alg / 1e6, 2), gathered_MB=round(nnz * d * 4 / 1e6, 1),
                gathered_TBps=round(gathered, 2),
                gather_roofline=dict(table_MB=round(table_mb, 1), achieved=round(gathered, 2), peak=round(gather_peak, 2),
                                     unit="TB/s of gathered rows", frac=round(gathered / gather_peak, 3),
                                     peak_source="random 256-byte row gathers from a table of this size, measured "
                                                 "(tools/gather_probe.hip)"),
                achieved=round(gbps, 1), peak=PEAK_HBM_GBPS, unit="GB/s", frac=round(gbps / PEAK_HBM_GBPS, 4), bound="hbm")


def bench_bpr(gdmcf_amd, workload, dev, layers=3, d=64, batch=1024, iters=20):
    """One LightGCN BPR training step (reference lightGCN.py:291-298): propagate over the whole graph (3 SpMM layers),
    gather the batch rows, BPR + regularisation loss, backward (the same propagation on the gradient), Adam."""
    from gdmcf_amd import data
    from gdmcf_amd.lightgcn import bpr_loss, sample_bpr_batch
    cfg = data.SHAPES[workload]
    indptr, indices, I = data.synth_csr(workload, seed=0)
    U = cfg["n_users"]
    users = np.repeat(np.arange(U), np.diff(indptr))
    torch.manual_seed(0)
    m = gdmcf_amd.LightGCN({"user_id_idx": users, "item_id_idx": indices}, U, I, layers, d, device=dev).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=0.005)
    rng = np.random.default_rng(0)
    batches = [[torch.from_numpy(a).to(dev) for a in sample_bpr_batch(indptr, indices, U, I, batch, rng)] for _ in range(4)]

    def step(i):
        bu, bp, bn = batches[i % 4]
        opt.zero_grad()
        out = m(bu, bp, bn)
        mf, reg = bpr_loss(bu, *out)
        (mf + 1e-4 * reg).backward()
        opt.step()
        return mf

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        mf = step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / iters
    return dict(ms_per_step=round(ms, 4), batch=batch, layers=layers, d=d, nodes=U + I, nnz=m.nnz, loss=float(mf))


if __name__ == "__main__":
    main()
