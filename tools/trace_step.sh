# kernel trace of the default training step: every launch between two AdamW launches, with start offsets and gaps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/trace; mkdir -p $O
rocprofv3 --kernel-trace -d $O/t -o t --output-format csv -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-prof --no-spmm --no-sampling --no-fused-leg "$@" > $O/bench.json 2> $O/log.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/trace/t/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
a,b=idx[8],idx[9]
t0=int(rows[a]['End_Timestamp']); prev=t0; tot=0
for r in rows[a+1:b+1]:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:90]
    print(f"{(st-t0)/1e3:8.1f} us  gap {(st-prev)/1e3:5.1f}  dur {(en-st)/1e3:7.1f}  {n}")
    prev=en; tot+=en-st
print("step span %.1f us, sum of kernels %.1f us, launches %d" % ((int(rows[b]['End_Timestamp'])-t0)/1e3, tot/1e3, b-a))
PY
