# fused weight-gradient + AdamW products of configs[2]: whole kernel / epilogue only (k loop skipped) / k loop only, per tile class
O=gpurun_out/${1:-bf16_abl}
mkdir -p $O
F="--workload amazon-book --gemm-dtype bf16 --fuse-optimizer --steps 40 --warmup 8 --no-cpu-baseline --no-spmm --no-sampling --no-fused-leg --no-graph-leg --no-configs2-leg"
for c in 3 0; do
  for d in 0 1 2; do
    GDMCF_BF16_DW_CLASS=$c GDMCF_BF16_DBG=$d python bench.py $F > $O/c${c}_d$d.json 2> $O/c${c}_d$d.err
    python - <<PY
import json
d=json.loads(open("$O/c${c}_d$d.json").read().strip().splitlines()[-1])
print("class $c dbg $d: ms/step", d["ms_per_step"], [(k["kernel"], k["avg_ms"]) for k in d["kernels"] if k["kernel"]=="bwd_weight_gemm"])
PY
  done
done
