set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
B="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --spmm --sampling"
rocprofv3 --kernel-trace --stats -d $O/stats -o f32 --output-format csv -- $B > $O/bench_under_rocprof.json 2> $O/stats.log
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o fetch --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o write --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof > $O/write.log 2>&1
echo write done
rocprofv3 --kernel-trace --stats -d $O/stats_bf16 -o bf16 --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --gemm-dtype bf16 --workload amazon-book > $O/bench_bf16_amazon_under_rocprof.json 2> $O/stats_bf16.log
echo bf16 stats done
python3 bench.py > $O/bench_default.json 2> $O/bench_default.log
python3 bench.py --gemm-dtype bf16 --workload amazon-book --no-cpu-baseline > $O/bench_bf16_amazon.json
python3 bench.py --gemm-dtype bf16 --no-cpu-baseline > $O/bench_bf16_yelp.json
python3 bench.py --workload amazon-book --no-cpu-baseline > $O/bench_f32_amazon.json
ls $O
