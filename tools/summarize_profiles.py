#!/usr/bin/env python3
"""Turn the raw rocprofv3 output merged into gpurun_out/ by tools/collect_profiles.sh into the
small tracked summaries under profiles/:  <tag>_kernel_stats_<cfg>.csv (rocprofv3 --stats table),
<tag>_kernel_by_day_<cfg>.csv (mean us of each kernel over the timed days, sampled days),
<tag>_pmc_k_scan.csv and traffic.json (HBM bytes per k_scan launch, FETCH_SIZE doubled per the
gfx950 note in MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
G = os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')
os.makedirs(P, exist_ok=True)
traffic = {}
pmc_rows = [['config', 'counter', 'launches', 'mean_KB', 'min_KB', 'max_KB']]
for cfg, key in (('hus', 'hus'), ('50m', '50000000'), ('200m', '200000000')):
    newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime, reverse=True)   # several collections may share a tag
    st = newest(os.path.join(G, '%s_trace_%s' % (tag, cfg), '*', '*kernel_stats.csv'))
    if st:
        shutil.copy(st[0], os.path.join(P, '%s_kernel_stats_%s.csv' % (tag, cfg)))
    tr = [t for t in newest(os.path.join(G, '%s_trace_%s' % (tag, cfg), '*', '*kernel_trace.csv')) if os.path.getsize(t) > 100000]
    if tr:
        rows = list(csv.DictReader(open(tr[0])))
        # the timed region = from the 365th-last k_scan launch on (preheat and warm-up runs come before)
        scans = sorted(int(r['Start_Timestamp']) for r in rows if r['Kernel_Name'].startswith('k_scan'))
        t0 = scans[-365] - 60000 if len(scans) >= 365 else 0
        rows = [r for r in rows if int(r['Start_Timestamp']) >= t0]
        names = sorted(set(r['Kernel_Name'].split('(')[0] for r in rows if r['Kernel_Name'].startswith(('k_', 'void k_'))))
        with open(os.path.join(P, '%s_kernel_by_day_%s.csv' % (tag, cfg)), 'w') as f:
            w = csv.writer(f)
            w.writerow(['kernel (timed region only)', 'launches', 'mean_us', 'min_us', 'max_us'] + ['launch%d_us' % d for d in range(5, 370, 30)])
            for n in names:
                d = np.array([(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000 for r in rows
                              if r['Kernel_Name'].split('(')[0] == n])
                w.writerow([n, len(d), round(d.mean(), 2), round(d.min(), 2), round(d.max(), 2)] +
                           [round(x, 1) for x in d[5::30]])
    tot = 0.0
    ok = True
    for kind, cname, mult in (('fetch', 'FETCH_SIZE', 2.0), ('write', 'WRITE_SIZE', 1.0)):
        f = newest(os.path.join(G, '%s_%s_%s' % (tag, kind, cfg), '*', '*counter_collection.csv'))
        if not f:
            ok = False
            continue
        vals = np.array([float(r['Counter_Value']) for r in csv.DictReader(open(f[0]))
                         if r['Kernel_Name'].startswith('k_scan') and r['Counter_Name'] == cname])[-365:]
        pmc_rows.append([cfg, cname, len(vals), round(vals.mean(), 3), vals.min(), vals.max()])
        tot += mult * vals.mean() * 1024
    if ok:
        traffic[key] = int(round(tot))
with open(os.path.join(P, '%s_pmc_k_scan.csv' % tag), 'w') as f:
    csv.writer(f).writerows(pmc_rows)
traffic['_comment'] = ('HBM bytes per k_scan launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes with '
                       '--kernel-trace), mean over the 365 timed launches; FETCH_SIZE (KB) doubled per '
                       'MI355X_MICROARCH.md (gfx950 reports half the bytes of a 16-B/lane stream); raw in %s_pmc_k_scan.csv' % tag)
json.dump(traffic, open(os.path.join(P, 'traffic.json'), 'w'), indent=1)
b = os.path.join(G, '%s_bench.json' % tag)
if os.path.exists(b):
    shutil.copy(b, os.path.join(P, '%s_bench.json' % tag))
print(json.dumps(traffic))
