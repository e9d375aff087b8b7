# Round profile: kernel statistics + HBM traffic of the default bench command, the bf16 / Amazon-Book line, the default lines.
# Usage (GPU box): bash tools/profile_round.sh <tag, e.g. r02>
set -e
T=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$T
mkdir -p $O
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fused-leg --no-graph-leg --no-configs2-leg --no-live-traffic --preheat-seconds 0"
rocprofv3 --kernel-trace --stats -d $O/stats -o f32 --output-format csv -- $B > $O/bench_under_rocprof.json 2> $O/stats.log
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o fetch --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof --no-fused-leg --no-graph-leg --no-configs2-leg --no-live-traffic --preheat-seconds 0 > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o write --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof --no-fused-leg --no-graph-leg --no-configs2-leg --no-live-traffic --preheat-seconds 0 > $O/write.log 2>&1
echo write done
python3 profiles/summarize.py traffic_by_tag $O/fetch/fetch_counter_collection.csv $O/write/write_counter_collection.csv yelp f32 > $O/hbm_traffic.json
python3 profiles/summarize.py stats $O/stats/f32_kernel_stats.csv > $O/kernel_stats_summary.json
rocprofv3 --kernel-trace --stats -d $O/stats_bf16 -o bf16 --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --gemm-dtype bf16 --workload amazon-book --no-spmm --no-sampling --no-graph-leg --no-live-traffic --preheat-seconds 0 > $O/bench_bf16_amazon_under_rocprof.json 2> $O/stats_bf16.log
echo bf16 stats done
python3 bench.py > $O/bench_default.json 2> $O/bench_default.log
python3 bench.py --gemm-dtype bf16 --workload amazon-book --no-cpu-baseline --no-live-traffic --preheat-seconds 0 > $O/bench_bf16_amazon.json
python3 bench.py --workload amazon-book --no-cpu-baseline --no-live-traffic --preheat-seconds 0 > $O/bench_f32_amazon.json
ls $O
