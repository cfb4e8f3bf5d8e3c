#!/bin/bash
# longer randomised soaks on the round's final binary (other scenarios than the collection's: REINA_SOAK_OFFSET): unsharded (the three
# launches a day), unsharded with the one-launch form switched on, sharded (both attribution modes; exact: trailers instead of the all-reduce)
R=${GRAFT_REPO_ROOT:-$(pwd)}; EV=$R/gpurun_out/r06_evidence; mkdir -p $EV; cd $R
export REINA_SOAK_OFFSET=${1:-500000}
T=${2:-420}
run() { name=$1; shift; timeout $((T + 60)) "$@" > $EV/$name 2>&1 & SP=$!; sleep $T; kill $SP 2>/dev/null; wait $SP 2>/dev/null; tail -1 $EV/$name; grep -c MISMATCH $EV/$name; }
run soak2_unsharded.txt python tools/parity_soak.py 100000
REINA_FUSED_DAY=1 run soak2_one_launch_form.txt python tools/parity_soak.py 100000
run soak2_sharded.txt python tools/parity_soak.py 100000 sharded
REINA_SOAK_ATTRIBUTION=exact run soak2_sharded_exact.txt python tools/parity_soak.py 100000 sharded
