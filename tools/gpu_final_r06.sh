#!/bin/bash
# evidence of round 6 on its final binary: randomised soak (both day forms, both attribution modes), profile collection (kernel traces,
# FETCH / WRITE / SQ passes, the ensemble's bytes -> profiles/traffic.json), per-day kernel times, the driver's window six times, the
# line as the driver's command prints it
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r06}; SOAK=${2:-150}
OUT=$R/gpurun_out; EV=$OUT/${TAG}_evidence; mkdir -p $EV; cd $R
sha256sum reina_model_amd/csrc/libreina_hip.so | cut -d' ' -f1 > $EV/lib_sha256.txt
timeout $((SOAK + 60)) python tools/parity_soak.py 100000 > $EV/soak_unsharded.txt 2>&1 &
SP=$!; sleep $SOAK; kill $SP 2>/dev/null; wait $SP 2>/dev/null
timeout $((SOAK + 60)) python tools/parity_soak.py 100000 sharded > $EV/soak_sharded.txt 2>&1 &
SP=$!; sleep $SOAK; kill $SP 2>/dev/null; wait $SP 2>/dev/null
tail -1 $EV/soak_unsharded.txt; tail -1 $EV/soak_sharded.txt; grep -c MISMATCH $EV/soak_unsharded.txt $EV/soak_sharded.txt
bash tools/collect_profiles.sh $TAG > $EV/collect.log 2>&1; tail -2 $EV/collect.log
for n in 1685983 100000000; do echo "== $n agents"; python tools/day_modes.py $n 365 auto 2>/dev/null | awk 'NR<=2 || NR%15==3 || /^#/'; done > $EV/kernel_times_by_day.txt
for i in 1 2 3 4 5 6; do python bench.py --steps 20 --warmup 5 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], b['value'], b.get('ms_per_step_warm'), b['roofline']['kernel_us_per_day'])"; done > $EV/driver_window_20_steps.txt
cat $EV/driver_window_20_steps.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $EV/driver_line.json 2> $EV/driver_line.err; wc -c $EV/driver_line.json; cp profiles/bench_detail.json $EV/driver_line_detail.json
ls $OUT | grep ${TAG}_ | head -60
