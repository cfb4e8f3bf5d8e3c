"""per-day kernel durations + SQ counters of one profiled size: python tools/prof_summary.py <tag>"""
import csv, glob, sys, collections
import numpy as np
tag = sys.argv[1]
f = glob.glob('gpurun_out/%s_trace/*/*kernel_trace.csv' % tag)[0]
ks = {}
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    ks.setdefault(n, []).append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000))
for n, v in ks.items():
    if not n.startswith('k_') or n in ('k_init', 'k_upload'):
        continue
    d = np.array([x[1] for x in sorted(v)[-365:]])
    print('  %-16s n=%d mean %.1f med %.1f p90 %.1f max %.1f | by day/30: %s' % (n, len(d), d.mean(), np.median(d), np.percentile(d, 90), d.max(), ' '.join('%.0f' % x for x in d[::30])))
fs = glob.glob('gpurun_out/%s_sq/*/*counter_collection.csv' % tag)
if fs:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        n = r['Kernel_Name'].split('(')[0]
        if n.startswith(('k_day', 'k_hosp_install')):
            agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
    for n, d in agg.items():
        for c, v in sorted(d.items()):
            v = np.array(v[-365:])
            print('%-16s %-20s mean %.4g  day300 %.4g  day95 %.4g' % (n, c, v.mean(), v[300], v[95]))
