set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/bf16s
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o bf16 --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --gemm-dtype bf16 --workload amazon-book > $O/bench_bf16_amazon_under_rocprof.json 2> $O/stats.log
python3 bench.py --gemm-dtype bf16 --workload amazon-book --no-cpu-baseline > $O/bench_bf16_amazon.json
python3 bench.py --gemm-dtype bf16 --no-cpu-baseline > $O/bench_bf16_yelp.json
