#!/usr/bin/env python3
"""Turn the raw rocprofv3 output merged into gpurun_out/ by tools/collect_profiles.sh into the small tracked
summaries under profiles/:

  <tag>_bench.json                  the bench line of the same binary
  <tag>_kernel_stats_<cfg>.csv      rocprofv3 --kernel-trace --stats table (one 365-day scenario, no warm-up)
  <tag>_kernel_by_day_<cfg>.csv     every kernel of the day: launches, mean / median / p90 / max us, sampled days
  <tag>_pmc_hbm_<cfg>.csv           FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes), KB per launch and per day
  <tag>_sq_<cfg>.csv                SQ counters per kernel (three passes), mean per launch
  traffic.json                      HBM bytes per simulated day (all kernels of the day), stamped with the sha256 of
                                    the libreina_hip.so they were collected on -- bench.py reports `traffic` only when
                                    that matches the binary it runs

FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of a 16-B/lane stream at 64 B);
that calibration holds for k_day's and k_open's streams; for the scattered dword reads of the other kernels it is
an upper bound (noted in the csv)."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
G = os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')
os.makedirs(P, exist_ok=True)
DAYS = 365
SKIP = ('k_init', 'k_upload')


def newest(pat):
    return sorted(glob.glob(pat), key=os.path.getmtime, reverse=True)


def kname(s):
    return s.split('(')[0].replace('void ', '')


def day_kernel(n):
    return n.startswith('k_') and n not in SKIP


per_day = {}
per_kernel = {}
for cfg, key in (('hus', 'hus'), ('50m', '50000000'), ('100m', '100000000'), ('200m', '200000000')):
    st = newest(os.path.join(G, '%s_trace_%s' % (tag, cfg), '*', '*kernel_stats.csv'))
    if st:
        shutil.copy(st[0], os.path.join(P, '%s_kernel_stats_%s.csv' % (tag, cfg)))
    tr = newest(os.path.join(G, '%s_trace_%s' % (tag, cfg), '*', '*kernel_trace.csv'))
    if tr:
        ks = collections.defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            n = kname(r['Kernel_Name'])
            if day_kernel(n):
                ks[n].append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000))
        with open(os.path.join(P, '%s_kernel_by_day_%s.csv' % (tag, cfg)), 'w') as f:
            w = csv.writer(f)
            w.writerow(['kernel', 'launches', 'us_per_simulated_day', 'mean_us', 'median_us', 'p90_us', 'max_us'] +
                       ['launch_at_%d%%' % p for p in range(0, 100, 10)])
            tot = 0.0
            for n in sorted(ks):
                d = np.array([x[1] for x in sorted(ks[n])])
                tot += d.sum() / DAYS
                w.writerow([n, len(d), round(d.sum() / DAYS, 2), round(d.mean(), 2), round(float(np.median(d)), 2),
                            round(float(np.percentile(d, 90)), 2), round(d.max(), 2)] +
                           [round(d[int(len(d) * p / 100)], 1) for p in range(0, 100, 10)])
            w.writerow(['SUM of kernel time per simulated day', '', round(tot, 2)])
    rows = [['kernel', 'counter', 'launches', 'mean_KB_per_launch', 'max_KB', 'KB_per_simulated_day', 'HBM_bytes_per_day (FETCH x2)']]
    tot, ok, pk = 0.0, True, {}
    for kind, cname, mult in (('fetch', 'FETCH_SIZE', 2.0), ('write', 'WRITE_SIZE', 1.0)):
        f = newest(os.path.join(G, '%s_%s_%s' % (tag, kind, cfg), '*', '*counter_collection.csv'))
        if not f:
            ok = False
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            n = kname(r['Kernel_Name'])
            if day_kernel(n) and r['Counter_Name'] == cname:
                agg[n].append(float(r['Counter_Value']))
        for n in sorted(agg):
            v = np.array(agg[n])
            b = mult * v.sum() * 1024 / DAYS
            rows.append([n, cname, len(v), round(v.mean(), 2), round(v.max(), 1), round(v.sum() / DAYS, 2), int(round(b))])
            pk[n] = pk.get(n, 0) + int(round(b))
            tot += b
    if ok and tot:
        rows.append(['ALL', 'FETCH_SIZE x2 + WRITE_SIZE', '', '', '', '', int(round(tot))])
        per_day[key] = int(round(tot))
        per_kernel[key] = pk
        with open(os.path.join(P, '%s_pmc_hbm_%s.csv' % (tag, cfg)), 'w') as f:
            csv.writer(f).writerows(rows)
    sq = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in ('sq1', 'sq2', 'sq3'):
        f = newest(os.path.join(G, '%s_%s_%s' % (tag, p, cfg), '*', '*counter_collection.csv'))
        if f:
            for r in csv.DictReader(open(f[0])):
                n = kname(r['Kernel_Name'])
                if day_kernel(n):
                    sq[n][r['Counter_Name']].append(float(r['Counter_Value']))
    if sq:
        with open(os.path.join(P, '%s_sq_%s.csv' % (tag, cfg), ), 'w') as f:
            w = csv.writer(f)
            w.writerow(['kernel', 'counter', 'launches', 'mean_per_launch', 'quiet_day(launch at 5%)', 'peak_day(max)'])
            for n in sorted(sq):
                for c in sorted(sq[n]):
                    v = np.array(sq[n][c])
                    w.writerow([n, c, len(v), '%.5g' % v.mean(), '%.5g' % v[int(len(v) * 0.05)], '%.5g' % v.max()])

sha_f = os.path.join(G, '%s_lib_sha256.txt' % tag)
if per_day and os.path.exists(sha_f):
    try:
        commit = subprocess.check_output(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD']).decode().strip()
    except Exception:
        commit = None
    sys.path.insert(0, ROOT)
    import bench as _bench
    json.dump({'lib_sha256': open(sha_f).read().strip(), 'src_sha256': _bench.src_sha256(), 'commit': commit, 'tag': tag, 'per_day_bytes': per_day,
               'per_kernel_bytes_per_day': per_kernel,
               '_comment': 'HBM bytes per simulated day summed over every kernel of the day: rocprofv3 --pmc FETCH_SIZE and '
                           '--pmc WRITE_SIZE (separate passes, --kernel-trace only) over one 365-day scenario; FETCH_SIZE (KB) '
                           'doubled per MI355X_MICROARCH.md; raw per kernel in %s_pmc_hbm_<cfg>.csv' % tag},
              open(os.path.join(P, 'traffic.json'), 'w'), indent=1)
b = os.path.join(G, '%s_bench.json' % tag)
if os.path.exists(b) and os.path.getsize(b):
    shutil.copy(b, os.path.join(P, '%s_bench.json' % tag))
print(json.dumps(per_day))
