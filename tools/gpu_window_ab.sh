#!/bin/bash
# the driver's window N times with and without the spin-up burst in front of the measured region (bench.py: _spin_gpu), alternating
# (the burst lived in bench.py of commit e779d9d only -- `_spin_gpu`, switched off by REINA_BENCH_NO_SPIN=1 --: it made the window slower and was taken out; the A/B of that commit: profiles/r06_evidence/window_spin_ab.txt)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
N=${1:-20}
py='import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b["ms_per_step"], b.get("ms_per_step_warm"))'
for i in $(seq 1 $N); do
  echo -n "spin    "; python3 bench.py --steps 20 --warmup 5 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python3 -c "$py"
  echo -n "no_spin "; REINA_BENCH_NO_SPIN=1 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python3 -c "$py"
done | tee $OUT/window_ab.txt
