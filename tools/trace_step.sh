cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/tr -o tr --output-format csv -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-prof > gpurun_out/tr.log 2>&1
