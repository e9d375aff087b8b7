# PMC passes over the SpMM kernels: tools/spmm_pmc.sh <out dir> [shape]   (env GDMCF_SPMM_GEN etc. select the kernel)
set -e
O=$GRAFT_REPO_ROOT/$1; S=${2:-yelp}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_BUSY_avr TCC_TAG_STALL_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d $O/p$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/spmm_only.py $S 3 > $O/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do python3 profiles/summarize.py counters $O/p$i/p_counter_collection.csv ; done > $O/counters.json
