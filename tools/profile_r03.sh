# Round-3 evidence run (GPU box): bash tools/profile_r03.sh
#  1. tools/profile_round.sh r03       -- kernel stats + FETCH/WRITE passes of the default bench command, bf16 Amazon stats, bench lines
#  2. MFMA-busy / wave-cycle counters of the same command (f32) and of the Amazon-Book bf16 command
#  3. FETCH/WRITE + L2 counters of the first-generation SpMM kernels on the Amazon-Book graph (the Yelp graph runs the streamed one)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03
O=gpurun_out/r03
B="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof --no-fused-leg --no-graph-leg --no-configs2-leg --no-spmm --no-sampling"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/mfma_a -o a --output-format csv -- $B > $O/mfma_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/mfma_c -o c --output-format csv -- $B > $O/mfma_c.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/mfma_b16 -o b --output-format csv -- $B --gemm-dtype bf16 --workload amazon-book > $O/mfma_b16.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_b16 -o f --output-format csv -- $B --gemm-dtype bf16 --workload amazon-book > $O/fetch_b16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write_b16 -o w --output-format csv -- $B --gemm-dtype bf16 --workload amazon-book > $O/write_b16.log 2>&1
for x in mfma_a/a mfma_c/c mfma_b16/b; do python3 profiles/summarize.py counters $O/${x}_counter_collection.csv > $O/$(dirname $x).json; done
python3 profiles/summarize.py traffic_by_tag $O/fetch_b16/f_counter_collection.csv $O/write_b16/w_counter_collection.csv amazon-book bf16 > $O/hbm_traffic_bf16_amazon.json
echo mfma done
bash tools/spmm_pmc.sh gpurun_out/r03/spmm_amazon amazon-book > $O/spmm_amazon.log 2>&1
echo spmm done
ls $O
