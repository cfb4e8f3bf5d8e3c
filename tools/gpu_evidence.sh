#!/bin/bash
# Run on the GPU box (via gpurun): the text evidence of a round on its final binary -- GPU suite log, randomised soak, in-kernel
# stamps of the day's opening, peak / quiet day kernel times, the sharded day's kernels, the ablation of the peak day, the
# microbenchmarks.  usage: bash tools/gpu_evidence.sh <tag> [soak seconds]   -> gpurun_out/<tag>_evidence/*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r03}; SOAK=${2:-420}
OUT=$R/gpurun_out/${TAG}_evidence; mkdir -p $OUT; cd $R
sha256sum reina_model_amd/csrc/libreina_hip.so | cut -d' ' -f1 > $OUT/lib_sha256.txt
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > $OUT/gpu_suite.txt 2>&1; echo "pytest rc=$?" >> $OUT/gpu_suite.txt; tail -2 $OUT/gpu_suite.txt
timeout $((SOAK + 60)) python tools/parity_soak.py 100000 > $OUT/soak_unsharded.txt 2>&1 &
SP=$!; sleep $SOAK; kill $SP 2>/dev/null; wait $SP 2>/dev/null
timeout $((SOAK + 60)) python tools/parity_soak.py 100000 sharded > $OUT/soak_sharded.txt 2>&1 &
SP=$!; sleep $((SOAK * 2 / 3)); kill $SP 2>/dev/null; wait $SP 2>/dev/null
tail -1 $OUT/soak_unsharded.txt; tail -1 $OUT/soak_sharded.txt; grep -c MISMATCH $OUT/soak_unsharded.txt $OUT/soak_sharded.txt
F="--offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math"
if [ -z "${EVIDENCE_QUICK:-}" ]; then   # (EVIDENCE_QUICK=1: what a change outside k_open / k_day leaves as it was is not measured again)
/opt/rocm/bin/hipcc $F -DREINA_OPEN_STAMPS -o /tmp/libreina_os.so reina_model_amd/csrc/reina_hip.hip 2>/dev/null
for n in 1e8 1685983; do REINA_HIP_LIB=/tmp/libreina_os.so python tools/open_stamps_big.py $n 2>/dev/null; done > $OUT/stamps_open.txt
fi
for n in 1685983 50000000 100000000 200000000; do echo "== $n agents"; python tools/peak_day.py $n 2>/dev/null; done > $OUT/peak_and_quiet_days.txt
{ echo "== 8 shards x 1685983 (BASELINE configs[1] per GPU), days 92-104 and 300-312"; python tools/sharded_kernels.py 8 13487864 92:104 2>/dev/null; python tools/sharded_kernels.py 8 13487864 300:312 2>/dev/null
  echo "== 2 shards x 5e7, days 92-104 and 300-312"; python tools/sharded_kernels.py 2 100000000 92:104 2>/dev/null; python tools/sharded_kernels.py 2 100000000 300:312 2>/dev/null; } > $OUT/sharded_day_kernels.txt
if [ -z "${EVIDENCE_QUICK:-}" ]; then
bash tools/gpu_ablate.sh > $OUT/ablation_1e8.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_scatter tools/ubench_scatter.hip 2>/dev/null && timeout 120 /tmp/ubench_scatter > $OUT/ubench_scatter.txt
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/ubench_prims tools/ubench_prims.hip 2>/dev/null && timeout 120 /tmp/ubench_prims > $OUT/ubench_prims.txt
# the bench line of this binary with `traffic` reported (profiles/traffic.json must carry its hash), and the round driver's window six times
python bench.py > $OUT/final_bench.json 2> $OUT/final_bench.err
for i in 1 2 3 4 5 6; do python bench.py --steps 20 --warmup 5 --no-cpu --no-sizes --no-ensemble 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['value'], b['roofline']['kernel_us_per_day'])"; done > $OUT/driver_window_20_steps.txt
python tools/run20_wall.py > $OUT/run20_wall.txt 2>/dev/null
ls -la $OUT
