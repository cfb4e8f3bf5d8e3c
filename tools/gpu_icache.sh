#!/bin/bash
# instruction-cache counters of one 365-day scenario (is a quiet day's k_day waiting for its own code?): usage gpu_icache.sh <tag> [agents]
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-ic}; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
A="--no-cpu --no-sizes --no-ensemble --steps 365 --warmup 0 --preheat-days 0 --agents ${2:-100000000}"
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_ic -- python3 $R/bench.py $A > /dev/null 2>&1
echo "ic rc=$?"
python3 - <<PY
import csv, glob, collections
f = glob.glob('$OUT/${TAG}_ic/**/*counter_collection.csv', recursive=True)
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    rows[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in rows.items():
    if not k.startswith(('void k_', 'k_')): continue
    print(k)
    for n, v in sorted(c.items()):
        v2 = sorted(v)
        print('   %-28s launches %4d mean %.4g  5%% %.4g  median %.4g  max %.4g' % (n, len(v), sum(v) / len(v), v2[len(v2) // 20], v2[len(v2) // 2], v2[-1]))
PY
