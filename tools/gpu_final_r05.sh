#!/bin/bash
# final evidence of round 5 on its final binary: randomised soak (both day forms, both attribution modes), profile collection (kernel
# traces, FETCH / WRITE / SQ passes -> profiles/traffic.json with per-kernel bytes and VALU utilisation), per-day kernel times, the
# sharded day's kernels in both attribution modes, the driver's window
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r05}; SOAK=${2:-300}
OUT=$R/gpurun_out; EV=$OUT/${TAG}_evidence; mkdir -p $EV; cd $R
sha256sum reina_model_amd/csrc/libreina_hip.so | cut -d' ' -f1 > $EV/lib_sha256.txt
timeout $((SOAK + 60)) python tools/parity_soak.py 100000 > $EV/soak_unsharded.txt 2>&1 &
SP=$!; sleep $SOAK; kill $SP 2>/dev/null; wait $SP 2>/dev/null
timeout $((SOAK + 60)) python tools/parity_soak.py 100000 sharded > $EV/soak_sharded.txt 2>&1 &
SP=$!; sleep $SOAK; kill $SP 2>/dev/null; wait $SP 2>/dev/null
tail -1 $EV/soak_unsharded.txt; tail -1 $EV/soak_sharded.txt; grep -c MISMATCH $EV/soak_unsharded.txt $EV/soak_sharded.txt
bash tools/collect_profiles.sh $TAG > $EV/collect.log 2>&1; tail -2 $EV/collect.log
for n in 1685983 50000000 100000000 200000000; do echo "== $n agents"; python tools/day_modes.py $n 365 auto 2>/dev/null | awk 'NR<=2 || NR%15==3 || /^#/'; done > $EV/kernel_times_by_day.txt
for mode in exact mirror; do
  echo "== [$mode] 8 shards x 1685983 (BASELINE configs[1] per GPU), days 92-104 and 300-312"; python tools/sharded_kernels.py 8 13487864 92:104 $mode 2>/dev/null; python tools/sharded_kernels.py 8 13487864 300:312 $mode 2>/dev/null
  echo "== [$mode] 8 shards x 12.5e6 = 1e8 in total (north_star's target), days 92-104 and 300-312"; python tools/sharded_kernels.py 8 100000000 92:104 $mode 2>/dev/null; python tools/sharded_kernels.py 8 100000000 300:312 $mode 2>/dev/null
  echo "== [$mode] 2 shards x 5e7, days 92-104 and 300-312"; python tools/sharded_kernels.py 2 100000000 92:104 $mode 2>/dev/null; python tools/sharded_kernels.py 2 100000000 300:312 $mode 2>/dev/null
done > $EV/sharded_day_kernels.txt
for i in 1 2 3 4 5 6; do python bench.py --steps 20 --warmup 5 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], b['value'], b['roofline']['kernel_us_per_day'], b.get('cold_count_rows', {}).get('ms_per_step'))"; done > $EV/driver_window_20_steps.txt
cat $EV/driver_window_20_steps.txt
ls $OUT | grep ${TAG}_ | head -60
