"""Per-kernel duration distribution and inter-kernel gaps over the timed region (the last `days`
k_day launches) of a rocprofv3 kernel trace.  usage: trace_breakdown.py <kernel_trace.csv> [days]"""
import collections
import csv
import sys

import numpy as np

f = sys.argv[1]
days = int(sys.argv[2]) if len(sys.argv) > 2 else 365
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    by[n].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
scan = sorted(by['k_day'])[-days:]
t0 = scan[0][0] - 60000
tot = 0
allk = []
for n, v in sorted(by.items()):
    v = [x for x in sorted(v) if x[0] >= t0]
    if not n.startswith('k_') or not v:
        continue
    allk += v
    d = np.array([(b - a) / 1e3 for a, b in v])
    tot += d.sum()
    print('%-16s n=%3d mean %6.1f  p10 %6.1f p50 %6.1f p90 %6.1f max %7.1f  sum/day %6.1f' % (
        n, len(d), d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90), d.max(), d.sum() / days))
allk.sort()
gaps = np.array([allk[i + 1][0] - allk[i][1] for i in range(len(allk) - 1)]) / 1e3
print('kernel sum per day %.1f us; gaps mean %.2f p50 %.2f p90 %.2f sum/day %.1f; wall/day %.1f' % (
    tot / days, gaps.mean(), np.percentile(gaps, 50), np.percentile(gaps, 90), gaps.sum() / days,
    (allk[-1][1] - allk[0][0]) / 1e3 / days))
# gaps attributed to the kernel that FOLLOWS them
names = {}
for n, v in by.items():
    for x in v:
        names[x] = n
g = collections.defaultdict(list)
for i in range(len(allk) - 1):
    g[names[allk[i + 1]]].append((allk[i + 1][0] - allk[i][1]) / 1e3)
for n, v in sorted(g.items()):
    v = np.array(v)
    print('gap before %-16s mean %5.2f p50 %5.2f p90 %5.2f sum/day %5.1f' % (n, v.mean(), np.percentile(v, 50), np.percentile(v, 90), v.sum() / days))
