#!/bin/bash
# compiler-flag variants of the library, built on the GPU box and measured side by side: the driver's window, the HUS year, the 1e8 year
# usage: bash tools/gpu_flag_sweep.sh        (variants below)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F="--offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math"
declare -a NAMES=(base preload16 maxilp maxmem prealloc)
declare -a FLAGS=("" "-mllvm -amdgpu-kernarg-preload-count=16" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-prealloc-sgpr-spill-vgprs")
for i in "${!NAMES[@]}"; do
  n=${NAMES[$i]}; f=${FLAGS[$i]}
  ( /opt/rocm/bin/hipcc $F $f -o /tmp/libreina_$n.so reina_model_amd/csrc/reina_hip.hip 2>&1 | grep -E "error|warning: unknown" | head -3 ) &
done
wait
pyline='import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=b["roofline"]["kernels"]; print(b["ms_per_step"], b.get("ms_per_step_warm"), {x: k[x]["us"] for x in k})'
for i in "${!NAMES[@]}"; do
  n=${NAMES[$i]}
  [ -f /tmp/libreina_$n.so ] || { echo "== $n: build failed"; continue; }
  echo "== $n  (${FLAGS[$i]})"
  export REINA_HIP_LIB=/tmp/libreina_$n.so
  timeout 200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "test_mini_default or test_mini_kitchen" 2>&1 | tail -1
  for r in 1 2 3; do timeout 200 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python3 -c "$pyline"; done
  timeout 200 python3 bench.py --steps 365 --warmup 0 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python3 -c "$pyline"
  timeout 300 python3 bench.py --agents 100000000 --steps 365 --warmup 0 --preheat-days 0 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python3 -c "$pyline"
done 2>&1 | tee $OUT/flag_sweep.txt
