#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel traces + PMC passes of the final binary.
# Usage: bash tools/collect_profiles.sh <tag>      (writes gpurun_out/<tag>_*; tools/summarize_profiles.py turns them
# into the tracked files under profiles/)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
sha256sum $R/reina_model_amd/csrc/libreina_hip.so | cut -d' ' -f1 > $OUT/${TAG}_lib_sha256.txt
python3 $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
COMMON="--no-cpu --no-sizes --no-ensemble --steps 365 --warmup 0 --preheat-days 0"
for cfg in "hus:" "50m:--agents 50000000" "100m:--agents 100000000" "200m:--agents 200000000"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_$name -- python3 $R/bench.py $COMMON $args > $OUT/${TAG}_trace_$name.json 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch_$name -- python3 $R/bench.py $COMMON $args > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write_$name -- python3 $R/bench.py $COMMON $args > /dev/null 2>&1
done
# the driver's window (bench.py's defaults there: --steps 20 --warmup 5) has its own traffic figure: all quiet days
W="--no-cpu --no-sizes --no-ensemble --steps 20 --warmup 5 --preheat-days 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch_husw -- python3 $R/bench.py $W > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write_husw -- python3 $R/bench.py $W > /dev/null 2>&1
# the 128-member ensemble (BASELINE config 5): the bytes of a group step -- the same bench command with only the ensemble beside the window
E="--no-cpu --no-sizes --steps 20 --warmup 5 --preheat-days 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch_ens -- python3 $R/bench.py $E > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write_ens -- python3 $R/bench.py $E > /dev/null 2>&1
for cfg in "hus:" "50m:--agents 50000000" "100m:--agents 100000000" "200m:--agents 200000000"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/${TAG}_sq1_$name -- python3 $R/bench.py $COMMON $args > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/${TAG}_sq2_$name -- python3 $R/bench.py $COMMON $args > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_sq3_$name -- python3 $R/bench.py $COMMON $args > /dev/null 2>&1
done
echo collected
