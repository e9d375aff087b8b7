# PMC passes over tools/fused_probe.py (plain and fused weight-gradient products, Yelp shapes): wave-cycle split, L2 hits, HBM bytes.
# Usage (GPU box): bash tools/profile_fused_probe.sh <out dir under gpurun_out>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-pmc_fused}
mkdir -p $O
export PROBE_ONLY=400
P="python3 tools/fused_probe.py 10"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/c -o c --output-format csv -- $P > $O/c.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/a -o a --output-format csv -- $P > $O/a.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f -o f --output-format csv -- $P > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w -o w --output-format csv -- $P > $O/w.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --kernel-trace -d $O/d -o d --output-format csv -- $P > $O/d.log 2>&1 || true
for x in c a f w d; do python3 profiles/summarize.py counters $O/$x/${x}_counter_collection.csv > $O/$x.json 2>/dev/null || true; done
ls $O
