#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
bash tools/gpu_call.sh g
cd /tmp && export TMPDIR=/tmp
A="--no-cpu --no-sizes --no-ensemble --steps 365 --warmup 0 --preheat-days 0 --agents 100000000"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/g_trace_100m -- python3 $R/bench.py $A > $OUT/g_trace_100m.json 2>/dev/null
echo "trace rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/g_sq_100m -- python3 $R/bench.py $A > /dev/null 2>&1
echo "sq rc=$?"
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/g_sq2_100m -- python3 $R/bench.py $A > /dev/null 2>&1
echo "sq2 rc=$?"
