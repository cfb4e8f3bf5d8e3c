#!/bin/bash
# the driver's window N times with and without the spin-up burst in front of the measured region (bench.py: _spin_gpu), alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
N=${1:-20}
py='import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b["ms_per_step"], b.get("ms_per_step_warm"))'
for i in $(seq 1 $N); do
  echo -n "spin    "; python3 bench.py --steps 20 --warmup 5 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python3 -c "$py"
  echo -n "no_spin "; REINA_BENCH_NO_SPIN=1 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python3 -c "$py"
done | tee $OUT/window_ab.txt
