#!/usr/bin/env python3
"""Turku override set (reference variables.py:10-216, selected by VARIABLE_OVERRIDE_SET, :218-220 and :436-437): data fixtures and
golden runs recorded from the REAL reference in this container.

    python tests/golden/make_turku.py [--jobs 6] [--skip-ensemble]

The reference selects an override set at import time from the environment, so this script sets VARIABLE_OVERRIDE_SET=turku
BEFORE the harness imports the reference's `variables` module -- the runs below go through the reference's own
`get_population_for_area()` (Turku is a municipality of data/005_11re_2019.csv), `get_initial_population_condition()`
(data/hosp_cases_turku.csv) and `get_active_interventions()` (the scenario's `add_interventions`).

Writes (data only; no reference source text):
  reina_model_amd/data/fi_turku.json      age histogram + case-file rows of the area (package input data)
  tests/golden/turku_inputs.json          the override set as the reference holds it (what reina_model_amd.variables must equal)
  tests/golden/turku_<scenario>_s<seed>.npz   per-day state of the reference run (layout of make_golden.py)
  tests/golden/ref_ens_turku_<scenario>.npz   64-run ensembles (layout of make_ref_ensemble.py)
"""
import argparse
import copy
import json
import multiprocessing
import os
import sys

os.environ['VARIABLE_OVERRIDE_SET'] = 'turku'

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '_harness'))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402

DAYS = 470          # 2020-02-18 .. 2021-06-01: all nine tracing steps, the `vaccinate` programme from 2021-03-15, the 2021-05 imports
AUTUMN = dict(start_date='2020-09-01', incubating_at_simulation_start=150, ill_at_simulation_start=50,
              recovered_at_simulation_start=1000)   # the start the override set keeps commented out (variables.py:16, :210-214)


def runs():
    out = {}
    for seed in (0, 1, 2):
        out['turku_default_s%d' % seed] = dict(seed=seed, days=DAYS, scenario='default', variables={})
        out['turku_astra-zeneca_s%d' % seed] = dict(seed=seed, days=DAYS, scenario='astra-zeneca', variables={})
    for seed in (0, 1):
        out['turku_stop-wearing-masks_s%d' % seed] = dict(seed=seed, days=DAYS, scenario='stop-wearing-masks', variables={})
        out['turku_autumn_s%d' % seed] = dict(seed=10 + seed, days=240, scenario='astra-zeneca', variables=AUTUMN)
    return out


def turku_variables(spec):
    import ref_harness as rh
    v = rh.default_variables()
    assert v['area_name'] == 'Turku'
    v.update(copy.deepcopy(spec['variables']))
    v['active_scenario'] = spec['scenario']
    v['random_seed'] = spec['seed']
    return v


def make_context(spec):
    """the reference's own construction (calc/simulation.py:148-180) for the Turku set"""
    import ref_harness as rh
    st = rh.setup()
    sim, ds = st['simulation'], st['datasets']
    v = turku_variables(spec)
    ages = ds.get_population_for_area(variable_store=v).sum(axis=1)
    ipc = ds.get_initial_population_condition(variable_store=v)
    age_to_group = sim.make_age_groups()
    age_groups = list(np.unique(age_to_group))
    pop_params = dict(age_structure=ages, contacts_per_day=rh.contacts_per_day(), initial_population_condition=ipc,
                      age_groups=dict(labels=age_groups, age_indices=[age_groups.index(x) for x in age_to_group]),
                      imported_infection_ages=v['imported_infection_ages'])
    ctx = st['model'].Context(population_params=pop_params,
                              healthcare_params=dict(hospital_beds=v['hospital_beds'], icu_units=v['icu_units']),
                              disease_params=sim.create_disease_params(v), start_date=v['start_date'], random_seed=spec['seed'])
    ivs = st['interventions'].get_active_interventions(v)
    for iv in ivs:
        ctx.add_intervention(iv)
    ipc_d = {k: int(getattr(ipc, k)) for k in ('dead', 'in_icu', 'in_ward', 'confirmed_cases', 'infected_cases', 'incubating', 'ill', 'recovered')}
    return ctx, v, ages, ipc_d, [iv.make_iv_tuple() for iv in ivs]


def run_one(args):
    name, spec = args
    ctx, v, ages, ipc_d, iv_tuples = make_context(spec)
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
    meta = dict(name=name, seed=spec['seed'], days=D, override_set='turku', scenario=spec['scenario'], variables=spec['variables'],
                interventions=[list(t) for t in iv_tuples], variant_names=vnames, ipc=ipc_d if any(ipc_d.values()) else None,
                age_counts=[int(x) for x in ages.values], pop13=mg.POP13, scalars=mg.SCALARS, places=mg.PLACES)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), pop=pop, scalars=scal, daily_contacts=dc, infected_by_variant=ibv,
                        per_age_final=per_age.astype(np.int32), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    return name, int(pop[-1, 3].sum()), int(pop[-1, 1].sum())


def gen_inputs():
    import pandas as pd
    import ref_harness as rh
    st = rh.setup()
    ds = st['datasets']
    v = rh.default_variables()
    ages = ds.get_population_for_area().sum(axis=1)
    df = pd.read_csv(ds.AREA_CASEFILES['Turku'], header=0, index_col=0)
    rows = [[str(d), int(r['dead']), int(r['in_icu']), int(r['in_ward']), int(r['confirmed'])] for d, r in df.iterrows()]
    pkg = dict(country=v['country'], area_name='Turku', age_counts=[int(x) for x in ages.values], case_rows=rows,
               case_columns=['date', 'dead', 'in_icu', 'in_ward', 'confirmed'])
    with open(os.path.join(HERE, '..', '..', 'reina_model_amd', 'data', 'fi_turku.json'), 'w') as f:
        json.dump(pkg, f)
    ov = st['variables'].VARIABLE_OVERRIDE_SETS['turku']
    # the override set itself as package data (like fi_hus.json: values recorded from the module the reference imports, UI texts
    # -- long names, scenario descriptions -- left out): reina_model_amd.variables.VARIABLE_OVERRIDE_SETS reads it
    pkg_ov = {k: ov[k] for k in ov if k != 'scenarios'}
    pkg_ov['scenarios'] = [{k: s_[k] for k in ('id', 'label', 'add_interventions') if k in s_} for s_ in ov['scenarios']]
    with open(os.path.join(HERE, '..', '..', 'reina_model_amd', 'data', 'override_sets.json'), 'w') as f:
        json.dump({'turku': pkg_ov}, f, ensure_ascii=False, indent=0)
    with open(os.path.join(HERE, 'turku_inputs.json'), 'w') as f:
        json.dump(dict(override_set={k: ov[k] for k in ov if k not in ('area_name_long',)},
                       scenario_ids=[s['id'] for s in ov['scenarios']],
                       add_interventions={s['id']: s.get('add_interventions', []) for s in ov['scenarios']},
                       variable_defaults={k: v[k] for k in v if k not in ('scenarios', 'area_name_long')},
                       age_counts=pkg['age_counts']), f)
    print('fi_turku.json: N=%d, %d case rows (%s .. %s)' % (ages.sum(), len(rows), rows[0][0], rows[-1][0]))
    # the third area the reference has a case file for (AREA_CASEFILES, calc/datasets.py:82-86): a hospital district, its population
    # the sum over its member municipalities (get_population_for_area's other branch, :55-58)
    vs = dict(v, area_name='Varsinais-Suomi')
    ages_vs = ds.get_population_for_area(variable_store=vs).sum(axis=1)
    dfv = pd.read_csv(ds.AREA_CASEFILES['Varsinais-Suomi'], header=0, index_col=0)
    rows_vs = [[str(d), int(r['dead']), int(r['in_icu']), int(r['in_ward']), int(r['confirmed'])] for d, r in dfv.iterrows()]
    with open(os.path.join(HERE, '..', '..', 'reina_model_amd', 'data', 'fi_varsinais-suomi.json'), 'w') as f:
        json.dump(dict(country=v['country'], area_name='Varsinais-Suomi', age_counts=[int(x) for x in ages_vs.values], case_rows=rows_vs,
                       case_columns=['date', 'dead', 'in_icu', 'in_ward', 'confirmed']), f)
    print('fi_varsinais-suomi.json: N=%d, %d case rows' % (ages_vs.sum(), len(rows_vs)))


CK_EVERY = 15


def ens_one(args):
    fam, spec = args
    ctx, v, ages, ipc_d, iv_tuples = make_context(spec)
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
    return fam, spec['seed'], pop, scal, dc, ibv, per_age.astype(np.int32), vnames, [int(x) for x in ages.values], [list(t) for t in iv_tuples]


def gen_ensembles(jobs, n_seeds):
    """ref_ens_turku_<scenario>.npz, the layout of make_ref_ensemble.py: `n_seeds` runs of the real reference per scenario"""
    for scenario in ('astra-zeneca',):   # (the vaccine changes severities only, main.pyx:1049-1054: the default scenario's infection curves are the same runs)
        fam = 'turku_' + scenario
        seeds = list(range(1000, 1000 + n_seeds))   # disjoint from the single-run goldens' seeds
        todo = [(fam, dict(seed=s_, days=DAYS, scenario=scenario, variables={})) for s_ in seeds]
        res = {}
        with multiprocessing.Pool(jobs) as pool:
            for r in pool.imap_unordered(ens_one, todo):
                res[r[1]] = r
        pop = np.stack([res[s_][2] for s_ in seeds])
        D = pop.shape[1]
        ck_days = np.arange(0, D, CK_EVERY)
        if ck_days[-1] != D - 1:
            ck_days = np.append(ck_days, D - 1)
        meta = dict(family=fam, days=D, override_set='turku', scenario=scenario, variables={}, interventions=res[seeds[0]][9],
                    variant_names=res[seeds[0]][7], ipc=None, age_counts=res[seeds[0]][8], pop13=mg.POP13, scalars=mg.SCALARS,
                    places=mg.PLACES)
        popf = pop.astype(np.float64)
        np.savez_compressed(
            os.path.join(HERE, 'ref_ens_%s.npz' % fam), tot=pop.sum(axis=3).astype(np.int32),
            scal=np.stack([res[s_][3] for s_ in seeds]), dc=np.stack([res[s_][4] for s_ in seeds]), ibv=np.stack([res[s_][5] for s_ in seeds]),
            ag_mean=popf.mean(axis=0), ag_var=popf.var(axis=0, ddof=1), ag_ck=pop[:, ck_days], ck_days=ck_days.astype(np.int32),
            per_age_final=np.stack([res[s_][6] for s_ in seeds]), seeds=np.asarray(seeds, dtype=np.int32),
            meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
        print('wrote ref_ens_%s.npz: %d seeds x %d days; final all_infected mean %.1f sd %.1f' % (
            fam, len(seeds), D, pop[:, -1, 3].sum(axis=1).mean(), pop[:, -1, 3].sum(axis=1).std(ddof=1)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--jobs', type=int, default=6)
    ap.add_argument('--only', default=None)
    ap.add_argument('--skip-ensemble', action='store_true')
    ap.add_argument('--ensemble-seeds', type=int, default=64)
    a = ap.parse_args()
    import ref_harness as rh
    rh.setup()
    if not a.only:
        gen_inputs()
    todo = [(k, v) for k, v in runs().items() if not a.only or k.startswith(a.only)]
    with multiprocessing.Pool(a.jobs) as pool:
        for name, tot, vacc in pool.imap_unordered(run_one, todo):
            print('%s done (all_infected at end: %d, vaccinated: %d)' % (name, tot, vacc), flush=True)
    if not a.skip_ensemble and not a.only:
        gen_ensembles(a.jobs, a.ensemble_seeds)


if __name__ == '__main__':
    main()
