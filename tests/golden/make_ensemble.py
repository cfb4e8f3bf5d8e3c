#!/usr/bin/env python3
"""Ensemble statistics of the SEQUENTIAL oracle (oracle/reina_seq.c, pinned bit-exact to the
reference cythonsim by test_oracle_seq.py) for the tolerance tier "parallel engine vs reference":

    python tests/golden/make_ensemble.py      # writes tests/golden/seq_ensemble_200k.npz

Scenario: HUS age shape scaled to 200 000 agents, 300 beds / 35 ICU units (capacity binds at the
peak), the reference's default interventions, 240 days, seeds 0..31.  Stored: per-day mean and
unbiased variance over seeds of the population totals listed in ATTRS.  The parallel formulation
(CPU oracle B, and the HIP engine) cannot replay the reference's sequential PCG64 stream, so its
trajectories are compared with these moments instead (tests/test_par_vs_seq.py).
"""
import copy
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

N = 200000
DAYS = 240
SEEDS = 32
ATTRS = ('susceptible', 'infected', 'all_infected', 'all_detected', 'in_ward', 'in_icu', 'dead',
         'recovered', 'non_hospital_deaths', 'new_infections')
OVERRIDES = dict(hospital_beds=300, icu_units=35)


def scenario():
    from reina_model_amd import datasets
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    v.update(OVERRIDES)
    return v, datasets.scaled_population(N)


def run_seq(seed):
    from oracle import seq_oracle as so
    v, ages = scenario()
    ctx = so.make_context(v, ages, seed)
    out = np.zeros((DAYS, len(ATTRS)))
    for d in range(DAYS):
        c = ctx.counters()
        out[d] = [c[a].sum() for a in ATTRS]
        ctx.iterate()
    return out


def main():
    with Pool(8) as p:
        runs = np.array(p.map(run_seq, range(SEEDS)))
    np.savez_compressed(os.path.join(HERE, 'seq_ensemble_200k.npz'),
                        mean=runs.mean(axis=0), var=runs.var(axis=0, ddof=1), n=SEEDS,
                        attrs=np.array(ATTRS), n_agents=N, days=DAYS)
    print('wrote seq_ensemble_200k.npz; day-239 all_infected mean %.1f' % runs[:, -1, 2].mean())


if __name__ == '__main__':
    main()
