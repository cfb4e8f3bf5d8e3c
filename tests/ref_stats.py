"""Statistical comparison of an ensemble of the PARALLEL formulation (HIP engine on the GPU, oracle B on
the CPU) with an ensemble of the REAL reference (tests/golden/ref_ens_<family>.npz, recorded from
/root/reference/cythonsim by tests/golden/make_ref_ensemble.py).  Test infrastructure only.

Why statistics: the reference draws every random number from one sequential PCG64 stream in scan
order (cythonsim/simrandom.pyx:13-55); a parallel engine cannot replay that stream, so the same
seed gives a different, equally valid trajectory.  What CAN be checked is that both engines sample
the same distribution of trajectories.  The checks, all on the days `ck_days` (every 15th day and the
last one):

  means      every one of the 13 age-group series x 9 age groups and its total, the 7 scalars of
             generate_state() (free ICU units / beds, r, exposed_per_day, ct_cases_per_day, ...),
             daily_contacts by place and infected_by_variant:
                 |mean_par - mean_ref| <= Z_MAX * sqrt(var_par / n_par + var_ref / n_ref)
             NO relative slack.  Age-group cells whose pooled mean is below MIN_MEAN agents are skipped (a handful
             of events in hundreds of runs: no normal approximation) -- they are covered by their totals.
  variances  totals only: |log(var_par / var_ref)| <= Z_MAX * sqrt((k - 1) / n_par + (k - 1) / n_ref)
             with k the pooled sample kurtosis (the large-sample standard error of a log variance).
  shapes     two-sample Kolmogorov-Smirnov on per-run final attack rate, deaths, detected cases, peak
             day and peak height of `infected`: p >= KS_P_MIN.

Z_MAX = 4.5: about 3 000 comparisons are made per family; if they were independent, a correct engine
would exceed 4.5 sigma somewhere in 2 % of all test outcomes (they are strongly correlated, so less).
The seeds are fixed, so a given build either passes or fails, it does not flicker.  The implied
relative tolerance on the cumulative counts is reported (`tolerance_report`).
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

Z_MAX = 4.5
MIN_MEAN = 5.0          # age-group cells
MIN_MEAN_TOTAL = 1.0    # totals, scalars, places, variants: the mean of >= 128 runs of a count with mean >= 1 is in the
#                         normal regime (e.g. in_icu of a 20 000-agent family: 2-5 agents, 6 units -- the old 4-run comparisons
#                         were the only check of those until round 2)
KS_P_MIN = 1e-3
EARLY_DAYS = (1, 2, 3, 5, 7, 10)


def load_ref(family):
    z = np.load(os.path.join(GOLDEN, 'ref_ens_%s.npz' % family))
    meta = json.loads(bytes(z['meta']))
    return z, meta


def variables_for(meta):
    import golden_util
    return golden_util.variables_for(meta)    # (the override set and the scenario a family was recorded under included)


def series_from_history(hist, meta, ctx):
    """history[S, D, COUNTER_WORDS] of the engine -> the arrays of a ref_ens file:
    ag[S, D, 13, 9], scal[S, D, 7], dc[S, D, 6], ibv[S, D, V]."""
    from reina_model_amd import engine as eng
    from reina_model_amd.contacts import PLACES
    A = eng.MAX_AGES
    S, D, _ = hist.shape
    ngroups = len(ctx.age_group_labels)
    onehot = np.zeros((ctx.nr_ages, ngroups), dtype=np.int64)
    onehot[np.arange(ctx.nr_ages), ctx.age_group_indices[:ctx.nr_ages]] = 1
    ag = np.zeros((S, D, 13, ngroups), dtype=np.int64)
    for i, n in enumerate(meta['pop13']):
        ci = eng.C_NAMES.index(n)
        ag[:, :, i, :] = hist[:, :, ci * A: ci * A + ctx.nr_ages].astype(np.int64) @ onehot
    sc = hist[:, :, eng.C_NR * A:].astype(np.float64)
    scal = np.zeros((S, D, 7))
    infections, infectors = sc[..., eng.S_TOTAL_INFECTIONS], sc[..., eng.S_TOTAL_INFECTORS]
    r = np.where(infectors > 5, infections / np.maximum(infectors, 1), 0.0)   # main.pyx:1817
    mob = 1.0 - np.asarray(ctx.mobility_history[:D], dtype=np.float64)
    cols = dict(available_icu_units=sc[..., eng.S_AVAILABLE_ICU], available_hospital_beds=sc[..., eng.S_AVAILABLE_BEDS],
                total_icu_units=sc[..., eng.S_ICU_UNITS], r=r, exposed_per_day=sc[..., eng.S_EXPOSED_PER_DAY],
                ct_cases_per_day=sc[..., eng.S_CT_CASES_PER_DAY], mobility_limitation=np.broadcast_to(mob, (S, D)))
    for i, n in enumerate(meta['scalars']):
        scal[:, :, i] = cols[n]
    dc = np.stack([sc[..., eng.S_DAILY_CONTACTS + PLACES.index(p)] for p in meta['places']], axis=-1)
    ibv = sc[..., eng.S_INFECTED_BY_VARIANT: eng.S_INFECTED_BY_VARIANT + len(meta['variant_names'])]
    return dict(ag=ag, scal=scal, dc=dc, ibv=ibv)


def run_parallel_ensemble(family, seeds, engine_factory=None, device='cuda:0', group=128, variables_patch=None):
    """`seeds` runs of the family's scenario on the parallel formulation (HIP engine groups by default,
    another implementation of the ABI through `engine_factory`).  Returns (series dict, meta)."""
    from reina_model_amd import ensemble, simulation
    z, meta = load_ref(family)
    v = variables_for(meta)
    if variables_patch:
        v.update(variables_patch)
    ages = np.asarray(meta['age_counts'])
    kw = dict(age_counts=ages, interventions=meta['interventions'], ipc=meta.get('ipc'), engine_factory=engine_factory,
              device=device)
    seeds = list(seeds)
    parts = []
    planner = None
    for s0 in range(0, len(seeds), group):
        planner = simulation.make_context(v, seed=0, **kw)
        plan = planner.make_plan(meta['days'])
        members = [simulation.make_context(v, seed=sd, **kw) for sd in seeds[s0:s0 + group]]
        if engine_factory is None:
            hist = ensemble.run_group_plan(members, plan)
        else:   # a CPU implementation of the ABI has no groups: replay the plan member by member
            hist = np.stack([m.run_plan(plan) for m in members])
        planner.mobility_history = plan['mobility_history']
        parts.append(series_from_history(hist, meta, planner))
        del members
    out = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    return out, meta


def run_sharded_ensemble(family, seeds, G, engine_factory=None, device='cuda:0'):
    """the family's scenario with the population split over G shards stepped in lock-step in ONE process (the
    per-day pressure exchange emulated on the host, reina_model_amd.sharding.step_shards_together); the global
    history is the sum over the shards.  Returns (series dict, meta)."""
    from reina_model_amd import engine as eng, sharding, simulation
    z, meta = load_ref(family)
    v = variables_for(meta)
    ages = np.asarray(meta['age_counts'])
    D = meta['days']
    hist = np.zeros((len(list(seeds)), D, eng.COUNTER_WORDS), dtype=np.int64)
    ctxs = None
    for k, seed in enumerate(seeds):
        members = []
        ctxs = [simulation.make_context(v, age_counts=ages, seed=seed, interventions=meta['interventions'], ipc=meta.get('ipc'),
                                        engine_factory=engine_factory, device=device, comm=sharding.InProcessComm(r, G, members))
                for r in range(G)]
        mob = []
        for d in range(D):
            hist[k, d] = sharding.reduce_counters(ctxs)
            mob.append(float(ctxs[0].contact_matrix.mobility_factor))
            sharding.step_shards_together(ctxs)
        ctxs[0].mobility_history = mob
    return series_from_history(hist, meta, ctxs[0]), meta


def _welch(name, g, r, out, min_mean=MIN_MEAN):
    """g, r: per-run samples of one quantity -> appends (name, z, mean_ref, mean_par, tol_rel)"""
    g = np.asarray(g, dtype=np.float64)
    r = np.asarray(r, dtype=np.float64)
    mg, mr = g.mean(), r.mean()
    pooled = (g.sum() + r.sum()) / (len(g) + len(r))
    se = np.sqrt(g.var(ddof=1) / len(g) + r.var(ddof=1) / len(r))
    if abs(pooled) < min_mean or se == 0.0:
        if se == 0.0 and mg != mr and abs(pooled) >= min_mean:
            out.append((name, np.inf, mr, mg, 0.0))
        return
    out.append((name, (mg - mr) / se, mr, mg, Z_MAX * se / abs(mr) if mr else np.inf))


def compare(par, ref, meta, z_max=Z_MAX):
    """par: series dict of the parallel ensemble; ref: the loaded ref_ens npz.  Returns a report dict
    with every comparison made; `failures` lists those outside the stated tolerance."""
    from scipy import stats
    ck = [int(d) for d in ref['ck_days']]
    means, variances, ks = [], [], []
    ref_tot, ref_ck = ref['tot'].astype(np.float64), ref['ag_ck'].astype(np.float64)
    # the first days in full: what Population.set_initial_state leaves behind decays within days (agents of any severity
    # put into ward / ICU leave on day 1: the reference's get_hospitalization_days / get_icu_days give them 0 days), and
    # the 15-day grid of the fixtures' age-group tables does not see it.  Totals only (the fixtures hold every day's totals).
    for d in EARLY_DAYS:
        if d in ck or d >= ref_tot.shape[1]:
            continue
        for i, n in enumerate(meta['pop13']):
            _welch('day %d %s total' % (d, n), par['ag'][:, d, i].sum(axis=1), ref_tot[:, d, i], means, min_mean=MIN_MEAN_TOTAL)
    for ki, d in enumerate(ck):
        for i, n in enumerate(meta['pop13']):
            _welch('day %d %s total' % (d, n), par['ag'][:, d, i].sum(axis=1), ref_tot[:, d, i], means, min_mean=MIN_MEAN_TOTAL)
            for gidx in range(ref_ck.shape[3]):
                _welch('day %d %s group %d' % (d, n, gidx), par['ag'][:, d, i, gidx], ref_ck[:, ki, i, gidx], means)
        for i, n in enumerate(meta['scalars']):
            if n in ('mobility_limitation', 'total_icu_units'):
                # host-side values: identical in every run, compared exactly
                if not np.allclose(par['scal'][:, d, i], ref['scal'][0, d, i], rtol=0, atol=1e-12):
                    means.append(('day %d %s' % (d, n), np.inf, float(ref['scal'][0, d, i]), float(par['scal'][0, d, i]), 0.0))
                continue
            _welch('day %d %s' % (d, n), par['scal'][:, d, i], ref['scal'][:, d, i], means,
                   min_mean=0.05 if n == 'r' else MIN_MEAN_TOTAL)
        for i, n in enumerate(meta['places']):
            _welch('day %d daily_contacts %s' % (d, n), par['dc'][:, d, i], ref['dc'][:, d, i], means, min_mean=MIN_MEAN_TOTAL)
        for i, n in enumerate(meta['variant_names']):
            _welch('day %d infected_by_variant %s' % (d, n), par['ibv'][:, d, i], ref['ibv'][:, d, i], means, min_mean=MIN_MEAN_TOTAL)
        # variance ratios of the totals
        for i, n in enumerate(meta['pop13']):
            g, r = par['ag'][:, d, i].sum(axis=1).astype(np.float64), ref_tot[:, d, i]
            vg, vr = g.var(ddof=1), r.var(ddof=1)
            if (g.sum() + r.sum()) / (len(g) + len(r)) < 10 * MIN_MEAN or vg == 0 or vr == 0:
                continue
            zs = np.concatenate([(g - g.mean()) / np.sqrt(vg), (r - r.mean()) / np.sqrt(vr)])
            kurt = max(float((zs ** 4).mean()), 1.5)
            se = np.sqrt((kurt - 1.0) * (1.0 / len(g) + 1.0 / len(r)))
            variances.append(('day %d var(%s)' % (d, n), float(np.log(vg / vr) / se), vr, vg, Z_MAX * se))
    # distribution shapes, per run
    idx = {n: i for i, n in enumerate(meta['pop13'])}
    gt = par['ag'].sum(axis=3).astype(np.float64)     # [S, D, 13]
    for name, fg, fr in (
            ('final all_infected', gt[:, -1, idx['all_infected']], ref_tot[:, -1, idx['all_infected']]),
            ('final dead', gt[:, -1, idx['dead']], ref_tot[:, -1, idx['dead']]),
            ('final all_detected', gt[:, -1, idx['all_detected']], ref_tot[:, -1, idx['all_detected']]),
            ('peak day of infected', gt[:, :, idx['infected']].argmax(axis=1), ref_tot[:, :, idx['infected']].argmax(axis=1)),
            ('peak infected', gt[:, :, idx['infected']].max(axis=1), ref_tot[:, :, idx['infected']].max(axis=1)),
            ('peak in_icu', gt[:, :, idx['in_icu']].max(axis=1), ref_tot[:, :, idx['in_icu']].max(axis=1))):
        if np.all(fg == fg[0]) and np.all(fr == fr[0]) and fg[0] == fr[0]:
            continue
        res = stats.ks_2samp(fg, fr)
        ks.append((name, float(res.pvalue), float(res.statistic), float(np.median(fr)), float(np.median(fg))))
    failures = [m for m in means if not abs(m[1]) <= z_max] + [m for m in variances if not abs(m[1]) <= z_max] + \
               [k for k in ks if not k[1] >= KS_P_MIN]
    return dict(means=means, variances=variances, ks=ks, failures=failures, n_par=par['ag'].shape[0], n_ref=ref_tot.shape[0])


def tolerance_report(rep, meta):
    """worst |z|, and the relative tolerance the comparison implies on the cumulative totals"""
    worst = max(rep['means'], key=lambda m: abs(m[1]))
    worst_v = max(rep['variances'], key=lambda m: abs(m[1])) if rep['variances'] else None
    cum = [m for m in rep['means'] if m[0].endswith(' total') and any(
        (' %s ' % n) in m[0] for n in ('all_infected', 'all_detected', 'dead', 'recovered', 'cum_icu'))]
    last_day = max(int(m[0].split()[1]) for m in cum)
    tail = [m for m in cum if int(m[0].split()[1]) >= last_day // 2]
    lines = ['%d parallel runs vs %d reference runs: %d mean comparisons, worst z = %+.2f (%s: ref %.2f, par %.2f)' % (
        rep['n_par'], rep['n_ref'], len(rep['means']), worst[1], worst[0], worst[2], worst[3])]
    if worst_v:
        lines.append('%d variance comparisons, worst z = %+.2f (%s)' % (len(rep['variances']), worst_v[1], worst_v[0]))
    lines.append('implied tolerance on cumulative totals, second half of the run: %.2f %% .. %.2f %%' % (
        100 * min(m[4] for m in tail), 100 * max(m[4] for m in tail)))
    lines.append('measured |difference| there: max %.2f %%' % (100 * max(abs(m[3] - m[2]) / abs(m[2]) for m in tail)))
    for k in rep['ks']:
        lines.append('KS %-22s p = %.3g  D = %.3f  (median ref %.1f, par %.1f)' % k)
    return '\n'.join(lines)


def assert_same_distribution(rep, meta):
    print(tolerance_report(rep, meta))
    if rep['failures']:
        msg = '\n'.join(str(f) for f in sorted(rep['failures'], key=lambda f: -abs(f[1]) if len(f) == 5 and f[0].startswith('day') else 0)[:25])
        raise AssertionError('%d comparisons outside the stated tolerance:\n%s' % (len(rep['failures']), msg))
