# Per-kernel statistics of the one-hot variants (bench.py --backbone onehot / onehot-emb).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/onehot_prof
mkdir -p $O
for bb in onehot onehot-emb; do
  rocprofv3 --kernel-trace --stats -d $O/$bb -o $bb --output-format csv -- python3 bench.py --backbone $bb --steps 30 --warmup 5 --no-cpu-baseline --no-prof > $O/bench_$bb.json 2> $O/$bb.log
  echo $bb done
done
