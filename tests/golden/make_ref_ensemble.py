#!/usr/bin/env python3
"""Ensembles of the REAL reference simulator (`/root/reference/cythonsim`, imported through
`_harness/ref_harness.py`, build container only) for the statistical tier "parallel engine vs
reference" (tests/test_ref_ensemble.py, tests/test_parity_gpu.py):

    python tests/golden/make_ref_ensemble.py [--only FAMILY] [--jobs 7] [--seeds N]

The parallel formulation cannot replay the reference's single sequential PCG64 stream, so the HIP
engine is compared with the DISTRIBUTION of reference runs instead of with one run.  The power of
that comparison is set by the number of reference seeds recorded here: 128 runs per family, the
HUS family at BASELINE configs[1] (1 685 983 agents x 365 days).

One `ref_ens_<family>.npz` per family (data only: inputs + expected outputs):
  tot[S,D,13]      int32    per seed, per day population totals (POP13 order of make_golden.py)
  scal[S,D,7]      float64  per seed, per day scalars (SCALARS order)
  dc[S,D,6]        int32    daily_contacts by place
  ibv[S,D,V]       int32    infected_by_variant
  ag_mean/ag_var[D,13,9]    per-day mean / unbiased variance over seeds of every age-group series
  ag_ck[S,K,13,9]  int32    every age-group series on the checkpoint days `ck_days`
  per_age_final[S,3,101]    get_population_stats('dead'|'all_infected'|'all_detected') at the end
  seeds[S], meta (json: scenario, interventions, population)
Row d is the state BEFORE the d-th iterate() (calc/simulation.py:195 vs :270).
"""
import argparse
import json
import multiprocessing
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '_harness'))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402

CK_EVERY = 15


def families():
    """family -> spec (same scenario definitions as the single-run goldens of make_golden.py)."""
    sc = mg.scenarios()
    fam = {}
    for name, key in (('hus_default', 'hus_default_s0'), ('mini_default', 'mini_default_s0'),
                      ('mini_imports', 'mini_imports_s0'), ('mini_kitchen', 'mini_kitchen_s0'),
                      ('mini_initial', 'mini_initial_s0')):
        spec = dict(sc[key])
        spec.pop('seed')
        fam[name] = spec
    return fam


def run_one(args):
    fam, spec, seed = args
    import ref_harness as rh
    rh.setup()
    ages = rh.hus_age_structure() if spec['pop'] == 'hus' else mg.mini_age_structure(spec['pop'])
    ctx = rh.make_context(seed, variables=spec['variables'], age_structure=ages,
                          interventions=spec['interventions'], ipc=spec.get('ipc'))
    D = spec['days']
    vnames = list(ctx.disease.variant_names)
    pop = np.zeros((D, 13, 9), dtype=np.int32)
    scal = np.zeros((D, 7), dtype=np.float64)
    dc = np.zeros((D, 6), dtype=np.int32)
    ibv = np.zeros((D, len(vnames)), dtype=np.int32)
    for d in range(D):
        s = ctx.generate_state()
        for i, k in enumerate(mg.POP13):
            pop[d, i] = s[k]
        for i, k in enumerate(mg.SCALARS):
            scal[d, i] = s[k]
        for i, k in enumerate(mg.PLACES):
            dc[d, i] = s['daily_contacts'][k]
        for i, k in enumerate(vnames):
            ibv[d, i] = s['infected_by_variant'][k]
        ctx.iterate()
    per_age = np.stack([ctx.get_population_stats(w) for w in ('dead', 'all_infected', 'all_detected')])
    return fam, seed, pop, scal, dc, ibv, per_age.astype(np.int32), vnames, [int(x) for x in ages.values]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    ap.add_argument('--jobs', type=int, default=7)
    ap.add_argument('--seeds', type=int, default=128)
    a = ap.parse_args()
    import ref_harness as rh
    rh.setup()  # build the extension once before forking workers
    fams = families()
    for fam, spec in fams.items():
        if a.only and fam != a.only:
            continue
        seeds = list(range(1000, 1000 + a.seeds))  # disjoint from the single-run goldens' seeds
        todo = [(fam, spec, s) for s in seeds]
        res = {}
        with multiprocessing.Pool(a.jobs) as pool:
            for r in pool.imap_unordered(run_one, todo):
                res[r[1]] = r
                print('%s seed %d done (%d/%d)' % (fam, r[1], len(res), len(seeds)), flush=True)
        pop = np.stack([res[s][2] for s in seeds])          # [S,D,13,9]
        D = pop.shape[1]
        ck_days = np.arange(0, D, CK_EVERY)
        if ck_days[-1] != D - 1:
            ck_days = np.append(ck_days, D - 1)
        ivs = rh.default_variables()['interventions'] if spec['interventions'] == 'default' \
            else spec['interventions']
        meta = dict(family=fam, days=D, variables=spec['variables'], interventions=ivs,
                    variant_names=res[seeds[0]][7], ipc=spec.get('ipc'), age_counts=res[seeds[0]][8],
                    pop13=mg.POP13, scalars=mg.SCALARS, places=mg.PLACES)
        popf = pop.astype(np.float64)
        np.savez_compressed(
            os.path.join(HERE, 'ref_ens_%s.npz' % fam),
            tot=pop.sum(axis=3).astype(np.int32),
            scal=np.stack([res[s][3] for s in seeds]),
            dc=np.stack([res[s][4] for s in seeds]),
            ibv=np.stack([res[s][5] for s in seeds]),
            ag_mean=popf.mean(axis=0), ag_var=popf.var(axis=0, ddof=1),
            ag_ck=pop[:, ck_days], ck_days=ck_days.astype(np.int32),
            per_age_final=np.stack([res[s][6] for s in seeds]),
            seeds=np.asarray(seeds, dtype=np.int32),
            meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
        print('wrote ref_ens_%s.npz: %d seeds x %d days; final all_infected mean %.1f sd %.1f'
              % (fam, len(seeds), D, pop[:, -1, 3].sum(axis=1).mean(), pop[:, -1, 3].sum(axis=1).std(ddof=1)),
              flush=True)


if __name__ == '__main__':
    main()
