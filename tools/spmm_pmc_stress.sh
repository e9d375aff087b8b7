# The config-5 graph (1.2 M nodes, 4e7 directed nonzeros, BASELINE configs[4] "HBM-bound SpMM stress"): the bench line of the
# propagation and three PMC passes over its kernels (L2 hits, HBM-side fetch / write bytes).  Usage (GPU box):
#   bash tools/spmm_pmc_stress.sh <out dir under gpurun_out>
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04_stress}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py --spmm-only --workload stress > $O/bench_spmm_stress.json 2> $O/bench_spmm_stress.err
echo bench done
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d $O/p$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/spmm_only.py stress 2 > $O/p$i.log 2>&1
  echo pass $i done
done
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python3 profiles/summarize.py counters $O/p$i/p_counter_collection.csv ; done > $O/counters.json
