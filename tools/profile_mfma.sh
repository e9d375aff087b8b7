set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/mfma
mkdir -p $O
B="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/a -o a --output-format csv -- $B > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --kernel-trace -d $O/b -o b --output-format csv -- $B > $O/b.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/c -o c --output-format csv -- $B > $O/c.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/d -o d --output-format csv -- $B > $O/d.log 2>&1
ls $O/*
