# f32x3 kernel ablations (GDMCF_SPLIT_DBG bits: 2 no MFMAs, 4 no global loads, 8 no split / LDS writes,
# 16 MFMA waves at s_setprio 1, 32 loader waves at s_setprio 1); usage: bash tools/split_ablate.sh [bits ...]
mkdir -p gpurun_out/r2s
for d in ${@:-0 2 4 8 12 14}; do
  echo "== dbg $d" >> gpurun_out/r2s/abl.txt
  GDMCF_SPLIT_DBG=$d timeout -k 10 200 python tools/split_probe.py 2>&1 | grep "f32x3" >> gpurun_out/r2s/abl.txt || exit 1
done
cat gpurun_out/r2s/abl.txt
