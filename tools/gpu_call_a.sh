#!/bin/bash
# round 2, first GPU call: full GPU suite, bench line, two-rank self-launch hook, counter list, SQ pass
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
free -g > $OUT/a_free.txt; nproc >> $OUT/a_free.txt
timeout 1500 python -m pytest tests -x -q -m gpu --durations=25 > $OUT/a_pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/a_pytest.log
tail -40 $OUT/a_pytest.log
timeout 600 python bench.py > $OUT/a_bench.json 2> $OUT/a_bench.err
echo "bench rc=$?"; tail -c 600 $OUT/a_bench.err
REINA_BENCH_BACKEND=gloo REINA_BENCH_ONE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-large --no-ensemble --no-cpu > $OUT/a_bench2.json 2> $OUT/a_bench2.err
echo "bench2 rc=$?"; tail -c 400 $OUT/a_bench2.err; head -c 600 $OUT/a_bench2.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/a_counters.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/a_sq_50m -- python3 $R/bench.py --agents 50000000 --no-cpu --no-sizes --no-ensemble --steps 365 --warmup 0 --preheat-days 0 > $OUT/a_sq_50m.json 2> $OUT/a_sq_50m.err
echo "sq rc=$?"
