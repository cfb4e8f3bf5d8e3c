"""Tolerance tier between the parallel formulation and the reference.

The reference consumes ONE PCG64 stream in scan order (cythonsim/simrandom.pyx:13-55), which no
parallel engine can replay; the parallel formulation (oracle B on CPU, the HIP kernels on GPU)
keys each decision by (agent, day, purpose) instead.  Same model, different random numbers, so
the claim is distributional.  Stated tolerance, per compared day and quantity, ensembles of n
seeds:  |mean_par - mean_seq| <= 4 * sqrt(var_par/n_par + var_seq/n_seq) + 0.5 % of the seq mean
(+1 agent).  mean_seq / var_seq come from the sequential oracle A, which is pinned bit-exact to
the real cythonsim (test_oracle_seq.py).
"""
import copy
import os

import numpy as np
import pytest

from golden_util import GOLDEN

DAYS_CHECKED = (20, 40, 60, 80, 100, 120, 160, 200, 239)
N_SEEDS = 16


def _ensemble(engine_factory=None, device='cuda:0'):
    import sys
    sys.path.insert(0, GOLDEN)
    import make_ensemble as me
    from reina_model_amd import engine as eng, simulation
    z = np.load(os.path.join(GOLDEN, 'seq_ensemble_200k.npz'))
    v, ages = me.scenario()
    runs = []
    A = eng.MAX_AGES
    for seed in range(5000, 5000 + N_SEEDS):
        ctx = simulation.make_context(v, age_counts=ages, seed=seed, engine_factory=engine_factory, device=device)
        h = ctx.run(me.DAYS)
        out = np.zeros((me.DAYS, len(me.ATTRS)))
        for k, a in enumerate(me.ATTRS):
            i = eng.C_NAMES.index(a)
            out[:, k] = h[:, i * A:(i + 1) * A].sum(axis=1)
        runs.append(out)
    return np.array(runs), z, me.ATTRS


def _check(runs, z, attrs, skip=None):
    mean_p, var_p = runs.mean(axis=0), runs.var(axis=0, ddof=1)
    n_p, n_s = runs.shape[0], int(z['n'])
    worst = 0.0
    for d in DAYS_CHECKED:
        for k, a in enumerate(attrs):
            if skip is not None and skip(d, a, z['mean'][d, k]):
                continue
            se = np.sqrt(var_p[d, k] / n_p + z['var'][d, k] / n_s)
            tol = 4.0 * se + 0.005 * abs(z['mean'][d, k]) + 1.0
            diff = abs(mean_p[d, k] - z['mean'][d, k])
            worst = max(worst, diff / tol)
            assert diff <= tol, 'day %d %s: parallel %.1f vs sequential %.1f (tol %.1f)' % (
                d, a, mean_p[d, k], z['mean'][d, k], tol)
    return worst


@pytest.mark.slow
def test_oracle_b_matches_sequential_oracle_statistically():
    import par_backend
    runs, z, attrs = _ensemble(engine_factory=par_backend.par_engine_factory)
    _check(runs, z, attrs)


@pytest.mark.slow
@pytest.mark.parametrize('G', [4, 8])
def test_sharded_formulation_matches_sequential_oracle_statistically(G):
    """SURVEY 8e deviation check: the population split over G shards (4, and the 8 of a full node) with import /
    vaccination quotas partitioned, free beds / ICU units pooled and re-divided by demand every day, and cross-shard
    contacts exchanged as pressure histograms stays inside the same tolerance (12 seeds)."""
    import sys
    sys.path.insert(0, GOLDEN)
    import make_ensemble as me
    import par_backend
    from reina_model_amd import engine as eng, sharding, simulation
    z = np.load(os.path.join(GOLDEN, 'seq_ensemble_200k.npz'))
    v, ages = me.scenario()
    A = eng.MAX_AGES
    runs = []
    for seed in range(7000, 7012):
        members = []
        ctxs = [simulation.make_context(v, age_counts=ages, seed=seed, engine_factory=par_backend.par_engine_factory,
                                        comm=sharding.InProcessComm(r, G, members)) for r in range(G)]
        out = np.zeros((me.DAYS, len(me.ATTRS)))
        for d in range(me.DAYS):
            c = sharding.reduce_counters(ctxs)
            for k, a in enumerate(me.ATTRS):
                i = eng.C_NAMES.index(a)
                out[d, k] = c[i * A:(i + 1) * A].sum()
            sharding.step_shards_together(ctxs)
        runs.append(out)
    # Beds and ICU units: every shard takes its demand-proportional share of the pooled FREE capacity every day, so even
    # 35 ICU units split 8 ways stay as busy as the undivided pool (28 of 33 occupied before that).  What remains is
    # second order: a bed released during the day stays on its shard until the next morning's split -- with 300 beds
    # over 8 shards the saturated ward runs about 2 % below the undivided one (294 vs 299); bounded here at 3 %.
    runs = np.array(runs)
    ward = list(me.ATTRS).index('in_ward')
    sat = lambda d, a, ref_mean: G == 8 and a == 'in_ward' and ref_mean > 0.95 * v['hospital_beds']
    _check(runs, z, me.ATTRS, sat)
    for d in DAYS_CHECKED:
        if G == 8 and z['mean'][d, ward] > 0.95 * v['hospital_beds']:
            assert runs[:, d, ward].mean() >= 0.97 * z['mean'][d, ward], (d, runs[:, d, ward].mean(), z['mean'][d, ward])


@pytest.mark.gpu
def test_hip_engine_matches_sequential_oracle_statistically():
    runs, z, attrs = _ensemble()
    _check(runs, z, attrs)


def test_initial_condition_parallel_form_matches_sequential_oracle():
    """set_initial_state: the parallel form (distinct agents, slot-ordered capacity) against the
    sequential restatement (draws with replacement), 24 seeds each, right after construction and on
    days 15 / 30; same tolerance as above."""
    import par_backend
    from oracle import seq_oracle as so
    from reina_model_amd import datasets, simulation
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    v.update(hospital_beds=9, icu_units=2, p_icu_death_no_beds=50.0)
    ages = datasets.scaled_population(60000)
    ipc = dict(dead=30, in_icu=12, in_ward=20, confirmed_cases=230, incubating=200, ill=150, recovered=900)
    names = ['susceptible', 'infected', 'all_infected', 'recovered', 'dead', 'in_ward', 'in_icu', 'detected',
             'all_detected', 'available_hospital_beds', 'available_icu_units']

    def series(ctx):
        out = []
        for d in range(31):
            if d in (0, 15, 30):
                s = ctx.generate_state()
                out.append([float(np.sum(s[n])) for n in names])
            ctx.iterate()
        return out

    A = np.array([series(so.make_context(v, ages, seed, ipc=ipc)) for seed in range(24)])
    B = np.array([series(simulation.make_context(v, age_counts=ages, seed=seed, ipc=ipc,
                                                 engine_factory=par_backend.par_engine_factory)) for seed in range(100, 124)])
    for d in range(3):
        for k, n in enumerate(names):
            se = np.sqrt(A[:, d, k].var(ddof=1) / 24 + B[:, d, k].var(ddof=1) / 24)
            tol = 4.0 * se + 0.005 * abs(A[:, d, k].mean()) + 1.0
            assert abs(A[:, d, k].mean() - B[:, d, k].mean()) <= tol, (d, n, A[:, d, k].mean(), B[:, d, k].mean(), tol)


@pytest.mark.gpu
def test_hip_engine_against_the_recorded_reference_runs_at_hus_scale():
    """The BASELINE configuration itself: 1 685 983 agents, default scenario, 365 days.  24 seeds on
    the GPU (one engine group) against the SIX runs recorded from the real cythonsim
    (tests/golden/hus_default_s*.npz), same tolerance: |mean_gpu - mean_ref| <= 4 * sqrt(var_gpu/24 +
    var_ref/6) + 0.5 % of the reference mean + 1, every 30th day, 11 quantities."""
    import glob
    import json
    from reina_model_amd import datasets, engine as eng, ensemble
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    files = sorted(glob.glob(os.path.join(GOLDEN, 'hus_default_s*.npz')))
    assert len(files) == 6
    meta = json.loads(bytes(np.load(files[0])['meta']))
    ref = np.array([np.load(f)['pop'].sum(axis=2) for f in files]).astype(np.float64)   # [6, 365, 13]
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    hist = ensemble.run_ensemble(v, range(9000, 9024), 365, age_counts=datasets.get_population_for_area(), concurrent=24)
    A = eng.MAX_AGES
    names = [n for n in meta['pop13'] if n != 'vaccinated']
    worst = 0.0
    for d in range(30, 365, 30):
        for n in names:
            g = hist[:, d, eng.C_NAMES.index(n) * A:(eng.C_NAMES.index(n) + 1) * A].sum(axis=1).astype(np.float64)
            r = ref[:, d, meta['pop13'].index(n)]
            tol = 4.0 * np.sqrt(g.var(ddof=1) / len(g) + r.var(ddof=1) / len(r)) + 0.005 * abs(r.mean()) + 1.0
            worst = max(worst, abs(g.mean() - r.mean()) / tol)
            assert abs(g.mean() - r.mean()) <= tol, (d, n, g.mean(), r.mean(), tol)
    print('worst |diff|/tol = %.2f' % worst)


@pytest.mark.gpu
@pytest.mark.parametrize('family,n_ref', [('mini_kitchen', 6), ('mini_default', 8), ('mini_imports', 4), ('mini_initial', 4)])
def test_hip_engine_against_the_recorded_mini_runs(family, n_ref):
    """Every intervention type against the REAL reference: 64 GPU seeds vs the recorded cythonsim runs
    of the same scenario (kitchen sink: vaccination, new beds / ICU, masks, variant imports, tracing;
    default interventions; imports only; initial population condition), 13 quantities every 25th day,
    tolerance 4 * sqrt(var_gpu/64 + var_ref/n_ref) + 0.5 % + 1."""
    from golden_util import load_run, variables_for
    from reina_model_amd import engine as eng, ensemble
    runs = [load_run('%s_s%d' % (family, k)) for k in range(n_ref)]
    meta = runs[0][1]
    ref = np.array([z['pop'].sum(axis=2) for z, _ in runs]).astype(np.float64)   # [n_ref, days, 13]
    v = variables_for(meta)
    ages = np.asarray(meta['age_counts'])
    members = []
    from reina_model_amd import simulation
    plan_ctx = simulation.make_context(v, age_counts=ages, seed=0, interventions=meta['interventions'], ipc=meta.get('ipc'))
    plan = plan_ctx.make_plan(meta['days'])
    members = [simulation.make_context(v, age_counts=ages, seed=31000 + s, interventions=meta['interventions'], ipc=meta.get('ipc'))
               for s in range(64)]
    hist = ensemble.run_group_plan(members, plan)
    A = eng.MAX_AGES
    for d in range(25, meta['days'], 25):
        for i, n in enumerate(meta['pop13']):
            g = hist[:, d, eng.C_NAMES.index(n) * A:(eng.C_NAMES.index(n) + 1) * A].sum(axis=1).astype(np.float64)
            r = ref[:, d, i]
            tol = 4.0 * np.sqrt(g.var(ddof=1) / len(g) + r.var(ddof=1) / len(r)) + 0.005 * abs(r.mean()) + 1.0
            assert abs(g.mean() - r.mean()) <= tol, (family, d, n, g.mean(), r.mean(), tol)
