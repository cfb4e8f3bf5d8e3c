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

FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of a 16-B/lane stream at 64 B).  Round 6
calibrated the other shapes (tools/ubench_pmc.hip -> profiles/pmc_calibration.json): a SCATTERED 4-byte load is one request tallied
at 64 B as well, and two such loads 64 bytes apart in one 128-byte line are still ONE request -- every read request that leaves L2
is a 128-byte line, so x 2 is the correction for every kernel of the day, not an upper bound; WRITE_SIZE is exact: 32 B per scattered
store or atomic (one sector), the bytes themselves for a 16-B/lane stream."""
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
util = {}
trace_us = {}   # key -> kernel -> [us per simulated day, mean us per launch, launches]: rocprofv3 --kernel-trace of the same command
N_SIMD, CLOCK_GHZ = 1024, 2.4   # MI355X: 256 CUs x 4 SIMDs; the shader clock the cycle figures are priced at
for cfg, key in (('hus', 'hus'), ('husw', 'hus_window'), ('50m', '50000000'), ('100m', '100000000'), ('200m', '200000000')):
    DAYS = 50 if cfg == 'husw' else 365   # (husw: the driver's window, 5 untimed + 20 timed days -- all of them quiet, all counted --, twice:
    #                                        bench.py runs such a short window a second time for its `cold_count_rows` figure)
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
            trace_us[key] = {n.split('<')[0]: [round(float(np.array([x[1] for x in ks[n]]).sum()) / DAYS, 3),
                                               round(float(np.array([x[1] for x in ks[n]]).mean()), 3), len(ks[n])] for n in sorted(ks)}
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
    disp = collections.defaultdict(dict)   # (pass, kernel) -> dispatch id -> {counter: value, 'ns': duration of THAT launch}
    for p in ('sq1', 'sq2', 'sq3'):
        f = newest(os.path.join(G, '%s_%s_%s' % (tag, p, cfg), '*', '*counter_collection.csv'))
        if f:
            for r in csv.DictReader(open(f[0])):
                n = kname(r['Kernel_Name'])
                if day_kernel(n):
                    sq[n][r['Counter_Name']].append(float(r['Counter_Value']))
                    d = disp[(p, n)].setdefault(r['Dispatch_Id'], {})
                    d[r['Counter_Name']] = float(r['Counter_Value'])
                    d['ns'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    # VALU issue utilisation of a launch = SQ_ACTIVE_INST_VALU (units of 4 clocks, summed over the chip) x 4 / (SIMDs x the launch's
    # cycles at the shader clock); waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES -- counters and duration of the SAME launch of the same
    # pass; "mean day": the sums over the year's launches, "peak day": the longest launch
    u = {}
    for n in sorted(set(k[1] for k in disp)):
        ent = {}
        d2 = [x for x in disp.get(('sq2', n), {}).values() if 'SQ_ACTIVE_INST_VALU' in x and x['ns'] > 0]
        if d2:
            pk = max(d2, key=lambda x: x['ns'])
            ent['valu_mean_day'] = round(sum(x['SQ_ACTIVE_INST_VALU'] for x in d2) * 4 / (N_SIMD * sum(x['ns'] for x in d2) * CLOCK_GHZ), 4)
            ent['valu_peak_day'] = round(pk['SQ_ACTIVE_INST_VALU'] * 4 / (N_SIMD * pk['ns'] * CLOCK_GHZ), 4)
            ent['peak_launch_us_under_counters'] = round(pk['ns'] / 1000, 1)
        d1 = [x for x in disp.get(('sq1', n), {}).values() if x.get('SQ_WAVE_CYCLES', 0) > 0 and 'SQ_WAIT_ANY' in x]
        if d1:
            pk = max(d1, key=lambda x: x['ns'])
            ent['waiting_mean_day'] = round(sum(x['SQ_WAIT_ANY'] for x in d1) / sum(x['SQ_WAVE_CYCLES'] for x in d1), 4)
            ent['waiting_peak_day'] = round(pk['SQ_WAIT_ANY'] / pk['SQ_WAVE_CYCLES'], 4)
            if all('SQ_INSTS_SALU' in x and x.get('SQ_INSTS_VALU', 0) > 0 for x in d1):
                ent['salu_per_valu_peak_day'] = round(pk['SQ_INSTS_SALU'] / pk['SQ_INSTS_VALU'], 4)
        if ent:
            u[n] = ent
    if u:
        util[key] = u
    if sq:
        with open(os.path.join(P, '%s_sq_%s.csv' % (tag, cfg), ), 'w') as f:
            w = csv.writer(f)
            w.writerow(['kernel', 'counter', 'launches', 'mean_per_launch', 'quiet_day(launch at 5%)', 'peak_day(max)'])
            for n in sorted(sq):
                for c in sorted(sq[n]):
                    v = np.array(sq[n][c])
                    w.writerow([n, c, len(v), '%.5g' % v.mean(), '%.5g' % v[int(len(v) * 0.05)], '%.5g' % v.max()])

# the ensemble (BASELINE config 5): bytes per GROUP STEP of the 128-member group -- the group kernels' dispatches of the full-size
# group (the largest grid of each kernel: a two-member warm-up group runs the same kernels first), 365 steps
ens_rows, ens_tot, ens_ok = [['kernel', 'counter', 'dispatches', 'mean_KB_per_group_step', 'HBM_bytes_per_group_step (FETCH x2)']], 0.0, True
for kind, cname, mult in (('fetch', 'FETCH_SIZE', 2.0), ('write', 'WRITE_SIZE', 1.0)):
    f = newest(os.path.join(G, '%s_%s_ens' % (tag, kind), '*', '*counter_collection.csv'))
    if not f:
        ens_ok = False
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        n = r['Kernel_Name'].replace('void ', '')
        if n.startswith('k_') and '<true' in n and r['Counter_Name'] == cname:
            agg[n.split('(')[0]].append((int(r['Grid_Size']), float(r['Counter_Value'])))
    for n in sorted(agg):
        big = max(g for g, _ in agg[n])
        v = np.array([x for g, x in agg[n] if g == big])
        steps = 365.0
        b = mult * v.sum() * 1024 / steps
        ens_rows.append([n, cname, len(v), round(v.sum() / steps, 2), int(round(b))])
        ens_tot += b
if ens_ok and ens_tot:
    ens_rows.append(['ALL', 'FETCH_SIZE x2 + WRITE_SIZE', '', '', int(round(ens_tot))])
    per_day['ensemble_128'] = int(round(ens_tot))
    with open(os.path.join(P, '%s_pmc_hbm_ensemble128.csv' % tag), 'w') as f:
        csv.writer(f).writerows(ens_rows)

sha_f = os.path.join(G, '%s_lib_sha256.txt' % tag)
if per_day and os.path.exists(sha_f):
    try:
        commit = subprocess.check_output(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD']).decode().strip()
    except Exception:
        commit = None
    sys.path.insert(0, ROOT)
    import bench as _bench
    json.dump({'lib_sha256': open(sha_f).read().strip(), 'src_sha256': _bench.src_sha256(), 'commit': commit, 'tag': tag, 'per_day_bytes': per_day,
               'per_kernel_bytes_per_day': per_kernel, 'utilisation': util, 'kernel_trace_us': trace_us,
               '_kernel_trace_us': 'per kernel [us per simulated day, mean us per launch, launches] from rocprofv3 --kernel-trace of the same 365-day '
                                   'command (every dispatch timestamped alike)',
               '_utilisation': 'per kernel, from the SQ passes of the same binary: valu = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x launch cycles at '
                               '2.4 GHz), waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES; mean day = sums over the 365 launches, peak day = the longest launch',
               '_comment': 'HBM bytes per simulated day summed over every kernel of the day: rocprofv3 --pmc FETCH_SIZE and '
                           '--pmc WRITE_SIZE (separate passes, --kernel-trace only) over one 365-day scenario; FETCH_SIZE (KB) '
                           'doubled per MI355X_MICROARCH.md; raw per kernel in %s_pmc_hbm_<cfg>.csv' % tag},
              open(os.path.join(P, 'traffic.json'), 'w'), indent=1)
b = os.path.join(G, '%s_bench.json' % tag)
if os.path.exists(b) and os.path.getsize(b):
    shutil.copy(b, os.path.join(P, '%s_bench.json' % tag))
print(json.dumps(per_day))
