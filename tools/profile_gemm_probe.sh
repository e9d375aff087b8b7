# PMC passes over tools/bin/gemm_probe (the five Yelp-shape products, DR and LDS-tiled kernels): L2 requests / hits, MFMA busy,
# wave-cycle split.  Usage (GPU box): bash tools/profile_gemm_probe.sh <out dir under gpurun_out> [env assignments...]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-pmc_probe}
shift || true
for kv in "$@"; do export "$kv"; done
mkdir -p $O
P="tools/bin/gemm_probe 6"
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/a -o a --output-format csv -- $P > $O/a.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/b -o b --output-format csv -- $P > $O/b.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/c -o c --output-format csv -- $P > $O/c.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f -o f --output-format csv -- $P > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w -o w --output-format csv -- $P > $O/w.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr --kernel-trace -d $O/d -o d --output-format csv -- $P > $O/d.log 2>&1 || true
for x in a b c f w d; do python3 profiles/summarize.py counters $O/$x/${x}_counter_collection.csv > $O/$x.json 2>/dev/null || true; done
ls $O
