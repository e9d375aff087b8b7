# MFMA-busy / wave-cycle counters of the default f32 bench command (separate --pmc passes; program directly after --)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04m}
mkdir -p $O
B="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof --no-fused-leg --no-graph-leg --no-configs2-leg --no-spmm --no-sampling --no-live-traffic --preheat-seconds 0"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/mfma_a -o a --output-format csv -- $B > $O/mfma_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/mfma_c -o c --output-format csv -- $B > $O/mfma_c.log 2>&1
for x in mfma_a/a mfma_c/c; do python3 profiles/summarize.py counters $O/${x}_counter_collection.csv > $O/$(dirname $x).json; done
ls $O
