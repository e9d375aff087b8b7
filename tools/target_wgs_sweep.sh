# split-K workgroup target of the two products with a small [batch, hidden] output (forward layer, input gradient)
O=gpurun_out/${1:-wgs_sweep}
mkdir -p $O
F="--steps 80 --warmup 15 --no-cpu-baseline --no-spmm --no-sampling --no-fused-leg --no-graph-leg --no-configs2-leg"
for w in 760 512 600 680 840 1000 1280; do
  GDMCF_TARGET_WGS=$w python bench.py $F > $O/w$w.json 2> $O/w$w.err
  python - <<PY
import json
d=json.loads(open("$O/w$w.json").read().strip().splitlines()[-1])
k={x["kernel"]:x["avg_ms"] for x in d["kernels"]}
print("target $w: ms/step", d["ms_per_step"], "fwd", k.get("linear_fwd_gemm"), "dh", k.get("bwd_input_gemm"))
PY
done
