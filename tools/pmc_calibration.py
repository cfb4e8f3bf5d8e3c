#!/usr/bin/env python3
"""gpurun_out/pmc_cal/ (tools/gpu_pmc_calibration.sh) -> profiles/pmc_calibration.json: what rocprofv3's FETCH_SIZE / WRITE_SIZE
report PER ACCESS for every access shape of tools/ubench_pmc.hip (known number of accesses per launch over a 9.6 GB footprint),
and the factor that turns the reported KB into bytes that crossed the L2's memory side for that shape.
usage: python tools/pmc_calibration.py [dir]"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'pmc_cal')
shapes = [json.loads(l) for l in open(os.path.join(D, 'shapes.jsonl')) if l.startswith('{')]


def counters(name):
    p = os.path.join(D, name + '_counters.csv')
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            k = r['Kernel_Name']
            if 'k_pmc' in k:
                kind = int(k.split('k_pmc<')[1].split('>')[0])
                out[kind][r['Counter_Name']].append(float(r['Counter_Value']))
    return out


fetch, write, rd, wr = counters('FETCH_SIZE'), counters('WRITE_SIZE'), counters('RDREQ'), counters('WRREQ')
res = {}
for s in shapes:
    kind = int(s['kernel'].split('<')[1].split('>')[0])
    acc = s['accesses_per_launch']
    mean = lambda d, c: (sum(d[kind][c]) / len(d[kind][c])) if d[kind].get(c) else None
    f_kb, w_kb = mean(fetch, 'FETCH_SIZE'), mean(write, 'WRITE_SIZE')
    e = dict(accesses_per_launch=acc, program_bytes_per_access=s['bytes_per_access'], us_per_launch=s['us_per_launch'],
             fetch_reported_bytes_per_access=None if f_kb is None else round(f_kb * 1024 / acc, 3),
             write_reported_bytes_per_access=None if w_kb is None else round(w_kb * 1024 / acc, 3))
    for name, d, c in (('rdreq_per_access', rd, 'TCC_EA0_RDREQ_sum'), ('rdreq_32B_per_access', rd, 'TCC_EA0_RDREQ_32B_sum'),
                       ('wrreq_per_access', wr, 'TCC_EA0_WRREQ_sum'), ('wrreq_64B_per_access', wr, 'TCC_EA0_WRREQ_64B_sum'),
                       ('atomic_req_per_access', wr, 'TCC_EA0_ATOMIC_sum')):
        v = mean(d, c)
        if v is not None:
            e[name] = round(v / acc, 4)
    res[s['shape']] = e
out = dict(_comment='tools/ubench_pmc.hip under rocprofv3 --pmc (one pass per counter, program directly behind `--`), one MI355X; '
                    'reported bytes = counter KB x 1024 / known accesses; footprint 9.6 GB (scattered) / 4 GiB per launch (streams)',
           shapes=res)
json.dump(out, open(os.path.join(ROOT, 'profiles', 'pmc_calibration.json'), 'w'), indent=1)
for k, v in res.items():
    print('%-10s program %3.0f B/access  FETCH %8s  WRITE %8s  %s' % (k, v['program_bytes_per_access'], v['fetch_reported_bytes_per_access'],
          v['write_reported_bytes_per_access'], {a: b for a, b in v.items() if a.endswith('_per_access') and 'reported' not in a and 'program' not in a}))
