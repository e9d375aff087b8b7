for D in 0 8 1 9 11 15; do echo "== gen3 dbg $D (1 no stores, 2 no piece reduction, 4 no descriptor loads, 8 no combine kernel)"; GDMCF_SPMM_DBG=$D GDMCF_SPMM_GEN=3 python tools/spmm_probe2.py yelp 2>&1 | grep -E "real|mod  2048"; done
for D in 0 9 15; do echo "== gen3 stress dbg $D"; GDMCF_SPMM_DBG=$D GDMCF_SPMM_GEN=3 python tools/spmm_probe2.py stress 2>&1 | grep -E "real|mod  2048"; done
