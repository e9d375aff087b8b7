#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (gpurun_out/<run>/...) into the small summaries committed here.

  python profiles/summarize.py stats  <kernel_stats.csv>                    -> per-kernel time table
  python profiles/summarize.py traffic <fetch counter csv> <write counter csv> -> HBM bytes per launch
  python profiles/summarize.py counters <counter csv>                       -> per-launch average of each counter

HBM traffic follows MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE come from separate
--pmc passes and are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores (dword-per-lane stores,
as in the GEMM epilogues, are uncalibrated -- marked)."""
import collections
import csv
import json
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def stats(path):
    rows = list(csv.DictReader(open(path)))
    out = []
    for r in rows:
        if float(r["Percentage"]) < 0.05:
            continue
        out.append(dict(kernel=short(r["Name"]), calls=int(r["Calls"]), avg_us=round(float(r["AverageNs"]) / 1e3, 2),
                        total_ms=round(float(r["TotalDurationNs"]) / 1e6, 3), pct=float(r["Percentage"])))
    return out


def per_kernel(path, counter):
    agg, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        agg[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return {k: agg[k] / len(n[k]) for k in agg}, {k: len(v) for k, v in n.items()}


def traffic(fetch_csv, write_csv):
    f, nf = per_kernel(fetch_csv, "FETCH_SIZE")
    w, _ = per_kernel(write_csv, "WRITE_SIZE")
    out = []
    for k in sorted(f, key=lambda k: -(f[k] + w.get(k, 0))):
        rd = 2.0 * f[k] * 1024  # gfx950 correction: FETCH_SIZE counts 128-B requests as 64 B
        wr = w.get(k, 0.0) * 1024
        if rd + wr < 1e6:
            continue
        out.append(dict(kernel=k, launches=nf[k], hbm_read_MB=round(rd / 1e6, 1), hbm_write_MB=round(wr / 1e6, 1),
                        hbm_total_MB=round((rd + wr) / 1e6, 1)))
    return out


def counters(path):
    """per-kernel per-launch average of every counter in a --pmc csv"""
    names = sorted({r["Counter_Name"] for r in csv.DictReader(open(path))})
    table = collections.defaultdict(dict)
    for c in names:
        avg, n = per_kernel(path, c)
        for k, v in avg.items():
            table[k][c] = round(v, 1)
            table[k]["launches"] = n[k]
    return [dict(kernel=k, **v) for k, v in table.items()]


# bench.py tag of a kernel, from its name and its layout / epilogue template arguments (never from tile sizes, which change
# with tuning): gemm_f32_kernel<LAY_A, LAY_B, BM, BN, BK, ..., EPI>, gemm_f32_spec_kernel<LAY_A, LAY_B, ...>
import re

TAG_RULES = [
    (r"^dr_tn_kernel<", "bwd_weight_gemm"),
    (r"^dr_tn_adamw_kernel<", "bwd_weight_gemm"),
    (r"^dr_kn_kernel<", "bwd_input_gemm"),  # (also the reverse loop's hidden layer through the transposed weight)
    (r"^dr_fat_kernel<\d+, 2>$", "loss_fwd_gemm"),
    (r"^dr_fat_kernel<\d+, 3>$", "posterior_gemm"),
    (r"^dr_nt_kernel<.*, 2>$", "loss_fwd_gemm"),
    (r"^dr_nt_kernel<.*, 3>$", "posterior_gemm"),
    (r"^gemm_f32_spec_kernel<1, 1,", "bwd_weight_gemm"),
    (r"^gemm_f32_kernel<0, 0, .*, 2>$", "loss_fwd_gemm"),
    (r"^gemm_f32_kernel<0, 0, .*, 3>$", "posterior_gemm"),
    (r"^gemm_f32_kernel<0, 1, .*, 0>$", "bwd_input_gemm"),
    (r"^gemm_f32_kernel<0, 0, .*, 0>$", "linear_fwd_gemm"),
    (r"^gemm_bf16", "bf16_gemm"),
    (r"^adamw_kernel", "adamw"),
    (r"^prep_input_kernel", "prep_input"),
    (r"^spmm_stream_kernel|^spmm_short_kernel", "spmm_csr"),
    (r"^spmm_vec_kernel", "spmm_csr_long_rows"),
    (r"^topk_fast_kernel", "topk"),
    (r"^topk_kernel", "topk"),
    (r"^onehot_noise", "onehot_noise"),
]


def by_tag(rows, workload, gemm_dtype):
    """rows of traffic() keyed by bench tag (the heaviest kernel wins when several map to one tag)"""
    tags = {}
    for r in rows:
        for pat, tag in TAG_RULES:
            if re.search(pat, r["kernel"]):
                if tag not in tags or r["hbm_total_MB"] * r["launches"] > tags[tag]["hbm_total_MB"] * tags[tag]["launches"]:
                    tags[tag] = r
                break
    return dict(workload=workload, gemm_dtype=gemm_dtype, note="HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and "
                "WRITE_SIZE passes of `python bench.py` (tools/profile_round.sh)", tags=tags)


if __name__ == "__main__":
    if sys.argv[1] == "traffic_by_tag":  # <fetch csv> <write csv> <workload> <gemm dtype>
        print(json.dumps(by_tag(traffic(sys.argv[2], sys.argv[3]), sys.argv[4], sys.argv[5]), indent=1))
    elif sys.argv[1] == "counters":
        print(json.dumps(counters(sys.argv[2]), indent=1))
    elif sys.argv[1] == "stats":
        print(json.dumps(stats(sys.argv[2]), indent=1))
    else:
        print(json.dumps(traffic(sys.argv[2], sys.argv[3]), indent=1))
