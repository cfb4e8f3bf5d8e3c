"""Does sharding shift the epidemic?  HUS x 365 d on G shards (in-process on one GPU), `n` seeds, against (a) the 128
recorded reference runs and (b) an unsharded HIP ensemble of 512 seeds -- per-run final size, deaths, detections,
peak height and day: means, relative difference, Welch z.  The 48-seed test of tests/test_reference_ensembles.py
has a tolerance of 1.4-6 % on these; this tool is how a smaller systematic shift is looked for.
usage: python tools/shard_effect.py [G=8] [n=256]"""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np

import ref_stats

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ref, meta = ref_stats.load_ref('hus_default')
idx = {k: i for i, k in enumerate(meta['pop13'])}


def outcomes(tot):   # tot[S, D, 13]
    tot = np.asarray(tot, dtype=np.float64)
    return {'final all_infected': tot[:, -1, idx['all_infected']], 'final dead': tot[:, -1, idx['dead']],
            'final all_detected': tot[:, -1, idx['all_detected']], 'peak infected': tot[:, :, idx['infected']].max(axis=1),
            'peak day': tot[:, :, idx['infected']].argmax(axis=1).astype(np.float64),
            'all_infected day 60': tot[:, 60, idx['all_infected']], 'all_infected day 90': tot[:, 90, idx['all_infected']],
            'all_infected day 120': tot[:, 120, idx['all_infected']],
            'mean in_icu days 90-130': tot[:, 90:130, idx['in_icu']].mean(axis=1),
            'mean in_ward days 90-130': tot[:, 90:130, idx['in_ward']].mean(axis=1)}


def show(name, a, b, la, lb):
    se = np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
    print('  %-26s %s %11.1f  %s %11.1f  diff %+6.2f %%  z %+5.2f' % (name, la, a.mean(), lb, b.mean(),
          100 * (a.mean() - b.mean()) / b.mean(), (a.mean() - b.mean()) / se if se else 0.0))


o_ref = outcomes(ref['tot'])
un, _ = ref_stats.run_parallel_ensemble('hus_default', range(70000, 70512))
o_un = outcomes(un['ag'].sum(axis=3))
sh, _ = ref_stats.run_sharded_ensemble('hus_default', range(90000, 90000 + n), G)
o_sh = outcomes(sh['ag'].sum(axis=3))
print('unsharded HIP (512 seeds) vs reference (128 runs)')
for k in o_ref:
    show(k, o_un[k], o_ref[k], 'hip', 'ref')
print('%d shards (%d seeds) vs reference (128 runs)' % (G, n))
for k in o_ref:
    show(k, o_sh[k], o_ref[k], 'shd', 'ref')
print('%d shards (%d seeds) vs unsharded HIP (512 seeds)' % (G, n))
for k in o_ref:
    show(k, o_sh[k], o_un[k], 'shd', 'hip')
