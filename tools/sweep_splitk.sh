for t in 440 520 600 680 760 840 960 1120; do
  echo "TARGET $t $(GDMCF_TARGET_WGS=$t python bench.py --no-cpu-baseline --steps 40 --warmup 8 --prof-every 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], [(k['kernel'][:10],k['avg_ms']) for k in d['kernels'] if 'linear_fwd' in k['kernel'] or 'bwd_input' in k['kernel']])")"
done
