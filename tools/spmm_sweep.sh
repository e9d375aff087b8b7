echo "== gen3 pipelined"; GDMCF_SPMM_GEN=3 python tools/spmm_probe2.py yelp 2>&1 | grep -E "real|mod  2048"
echo "== gen3 unpipelined"; GDMCF_SPMM_PIPE=0 GDMCF_SPMM_GEN=3 python tools/spmm_probe2.py yelp 2>&1 | grep -E "real|mod  2048"
echo "== gen3 pipelined 8192 waves"; GDMCF_SPMM_WAVES=8192 GDMCF_SPMM_GEN=3 python tools/spmm_probe2.py yelp 2>&1 | grep -E "real|mod  2048"
echo "== gen3 pipelined stress"; GDMCF_SPMM_GEN=3 python tools/spmm_probe2.py stress 2>&1 | grep -E "real|mod  2048"
