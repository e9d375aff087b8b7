# configs[2] (Amazon-Book shape, bf16 GEMM inputs) with AdamW fused into the weight-gradient epilogues: tile class of the
# weight-gradient products (GDMCF_BF16_DW_CLASS: 3 = 208x256 on 8 waves, 1 = 128x128, 0 = 80x128).  Usage: bash tools/bf16_fused_sweep.sh <out>
O=gpurun_out/${1:-bf16_sweep}
mkdir -p $O
F="--workload amazon-book --gemm-dtype bf16 --fuse-optimizer --steps 60 --warmup 10 --no-cpu-baseline --no-spmm --no-sampling --no-fused-leg --no-graph-leg --no-configs2-leg"
for c in default 1 0 2; do
  if [ $c = default ]; then unset GDMCF_BF16_DW_CLASS; else export GDMCF_BF16_DW_CLASS=$c; fi
  python bench.py $F > $O/class_$c.json 2> $O/class_$c.err
  python - <<PY
import json
d=json.loads(open("$O/class_$c.json").read().strip().splitlines()[-1])
print("class $c: ms/step", d["ms_per_step"], [(k["kernel"], k["avg_ms"]) for k in d["kernels"]])
PY
done
