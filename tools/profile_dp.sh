# Per-kernel statistics of the data-parallel step rehearsed in a one-rank RCCL group (all-reduce and sharded-optimiser paths).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/dp
mkdir -p $O
for mode in ar sh; do
  extra=""; [ $mode = sh ] && extra="--shard-optimizer"
  rocprofv3 --kernel-trace --stats -d $O/$mode -o $mode --output-format csv -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-prof --rehearse-dp $extra > $O/bench_$mode.json 2> $O/$mode.log
  echo $mode done
done
find $O -name "*kernel_stats.csv"
