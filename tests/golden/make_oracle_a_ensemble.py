#!/usr/bin/env python3
"""More reference-EQUIVALENT runs of the headline configuration, for the power of the statistical tier:

    python tests/golden/make_oracle_a_ensemble.py [--seeds 256] [--jobs 8]

oracle/reina_seq.c (oracle A) restates cythonsim bit for bit -- it reproduces all 37 recorded runs of the real reference,
every per-day age-group histogram, `r`, contacts and variants (tests/test_oracle_seq.py) -- so a run of A with a new seed is a
run the reference would have produced with that seed.  This script makes 256 of them for the HUS default scenario
(BASELINE configs[1]: 1 685 983 agents x 365 days; seeds 2000.., disjoint from the 128 recorded cythonsim runs' 1000..1127 and
from the single-run goldens) and writes `oracle_a_ens_hus_default.npz` (data only):

  tot[S,D,13]  int32  per seed, per day population totals, POP13 order of make_golden.py (row d = the state BEFORE day d)
  seeds[S], meta (json: what generated it)

The 128 runs of the REAL cythonsim (`ref_ens_hus_default.npz`, make_ref_ensemble.py) stay the anchor: a test without any
engine checks that these 256 runs and those 128 are samples of one distribution; the GPU tier then compares the HIP engine
with all 384.  About 5 s per run and core."""
import argparse
import json
import multiprocessing
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def run_one(seed):
    import ref_stats
    from oracle import seq_oracle as so
    ref, meta = ref_stats.load_ref('hus_default')
    v = ref_stats.variables_for(meta)
    ages = np.asarray(meta['age_counts'])
    D = meta['days']
    ctx = so.make_context(v, ages, seed, interventions=meta['interventions'], ipc=meta.get('ipc'))
    out = np.zeros((D, 13), dtype=np.int32)
    for d in range(D):
        c = ctx.counters()
        out[d] = [int(c[n].sum()) for n in meta['pop13']]
        ctx.iterate()
    return seed, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=256)
    ap.add_argument('--jobs', type=int, default=8)
    a = ap.parse_args()
    seeds = list(range(2000, 2000 + a.seeds))
    res = {}
    with multiprocessing.Pool(a.jobs) as pool:
        for seed, out in pool.imap_unordered(run_one, seeds):
            res[seed] = out
            if len(res) % 16 == 0:
                print('%d / %d runs' % (len(res), len(seeds)), flush=True)
    tot = np.stack([res[s] for s in seeds])
    meta = dict(family='hus_default', generator='oracle/reina_seq.c through oracle/seq_oracle.py (bit-exact restatement of cythonsim, '
                'pinned by the 37 recorded runs)', seeds_first=seeds[0], runs=len(seeds))
    np.savez_compressed(os.path.join(HERE, 'oracle_a_ens_hus_default.npz'), tot=tot, seeds=np.asarray(seeds, dtype=np.int32),
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print('wrote oracle_a_ens_hus_default.npz: %d runs; final all_infected mean %.1f sd %.1f' % (
        len(seeds), tot[:, -1, 3].mean(), tot[:, -1, 3].std(ddof=1)))


if __name__ == '__main__':
    main()
