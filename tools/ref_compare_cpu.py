#!/usr/bin/env python3
"""Oracle B (CPU, bit-identical to the HIP engine) against a recorded reference ensemble, seeds spread over
host processes: `python tools/ref_compare_cpu.py hus_default 128 [jobs] [key=value ...]` (variables patches).
The GPU suite does this with 512 HIP seeds (tests/test_reference_ensembles.py); this is the same comparison
without a GPU, for investigating a discrepancy."""
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def work(args):
    family, seeds, patch = args
    import par_backend
    import ref_stats
    G = int(os.environ.get('REF_COMPARE_SHARDS', '0'))
    if G:
        par, _ = ref_stats.run_sharded_ensemble(family, seeds, G, engine_factory=par_backend.par_engine_factory)
        return par
    par, _ = ref_stats.run_parallel_ensemble(family, seeds, engine_factory=par_backend.par_engine_factory, group=len(seeds),
                                             variables_patch=patch)
    return par


def main():
    family, n = sys.argv[1], int(sys.argv[2])
    jobs = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    patch = {}
    for kv in sys.argv[4:]:
        k, v = kv.split('=')
        patch[k] = float(v)
    import ref_stats
    seeds = list(range(60000, 60000 + n))
    chunks = [seeds[i::jobs] for i in range(jobs) if seeds[i::jobs]]
    cache = '/tmp/ref_compare_%s_%d_%s_g%s.npz' % (family, n, '_'.join('%s%g' % kv for kv in sorted(patch.items())), os.environ.get('REF_COMPARE_SHARDS', '0'))
    if os.path.exists(cache):
        par = dict(np.load(cache))
    else:
        with Pool(jobs) as p:
            parts = p.map(work, [(family, c, patch) for c in chunks])
        par = {k: np.concatenate([q[k] for q in parts]) for k in parts[0]}
        np.savez_compressed(cache, **par)
    ref, meta = ref_stats.load_ref(family)
    rep = ref_stats.compare(par, ref, meta)
    print(ref_stats.tolerance_report(rep, meta))
    print('%d failures' % len(rep['failures']))
    for f in sorted(rep['failures'], key=lambda f: -abs(f[1]))[:30]:
        print(f)
    zs = {m[0]: m[1] for m in rep['means']}
    names = ['infected total', 'new_infections total', 'all_infected total', 'all_detected total', 'detected total', 'dead total',
             'in_icu total', 'in_ward total', 'exposed_per_day', 'ct_cases_per_day', 'r']
    print('z by day: ' + ' | '.join(names))
    for d in ref['ck_days']:
        print('%4d ' % d + ' '.join('%+6.2f' % zs.get('day %d %s' % (d, nm), float('nan')) for nm in names))
    top = sorted(rep['means'], key=lambda m: -abs(m[1]))[:15]
    print('largest |z| among the means:')
    for m in top:
        print('  %-40s z=%+.2f ref=%.2f par=%.2f' % m[:4])


if __name__ == '__main__':
    main()
