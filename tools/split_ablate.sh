mkdir -p gpurun_out/r2s
for d in 0 2 4 8 12 14; do
  echo "== dbg $d" >> gpurun_out/r2s/abl.txt
  GDMCF_SPLIT_DBG=$d timeout -k 10 200 python tools/split_probe.py 2>&1 | grep "f32x3" >> gpurun_out/r2s/abl.txt || exit 1
done
cat gpurun_out/r2s/abl.txt
