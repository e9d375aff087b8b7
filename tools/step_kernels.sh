# rocprofv3 per-kernel statistics of the default training step only (no extra legs): quick check of the small kernels.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/step
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/s -o s --output-format csv -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-prof > $O/bench.json 2> $O/log.txt
python3 profiles/summarize.py stats $O/s/s_kernel_stats.csv > $O/summary.json
