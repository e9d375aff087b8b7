O=$GRAFT_REPO_ROOT/gpurun_out/r2b/hit; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift
  ( export "$@"; rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $O/$n -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/spmm_only.py yelp 3 > $O/$n.log 2>&1 )
  echo "== $n"; python3 $GRAFT_REPO_ROOT/profiles/summarize.py counters $O/$n/p_counter_collection.csv | grep -A6 "spmm_\(stream\|bundle_k\|short\|vec\)" | grep -E "kernel|TCC"
}
run gen3_nt GDMCF_SPMM_GEN=3
cd $GRAFT_REPO_ROOT
echo "== time gen3"; GDMCF_SPMM_GEN=3 python tools/spmm_probe2.py yelp 2>&1 | grep -E "real|mod  2048"
echo "== time gen3 stress"; GDMCF_SPMM_GEN=3 python tools/spmm_probe2.py stress 2>&1 | grep -E "real|mod  2048"
