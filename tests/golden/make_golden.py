#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference simulator
(`/root/reference/cythonsim`, imported through `_harness/ref_harness.py`) in this container.

    python tests/golden/make_golden.py [--only NAME_PREFIX] [--jobs 6]

Outputs are data only (inputs + expected outputs). The reference's sources are never copied.
Every fixture records the scenario (variables overrides, interventions, population) that made it,
so the oracle tests can rebuild the same inputs without the reference present.

Fixture layout (one .npz per run):
  pop[D,13,9]  int32   generate_state() age-group series, attr order = POP13 below
  scalars[D,7] float64 SCALARS order below
  daily_contacts[D,6] int32 (home, work, school, transport, leisure, other)
  infected_by_variant[D,V] int32
  per_age_final[3,101] int32  get_population_stats('dead'|'all_infected'|'all_detected') at the end
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

POP13 = ['susceptible', 'vaccinated', 'infected', 'all_infected', 'detected', 'all_detected',
         'in_icu', 'cum_icu', 'in_ward', 'dead', 'recovered', 'non_hospital_deaths',
         'new_infections']
SCALARS = ['available_icu_units', 'available_hospital_beds', 'total_icu_units', 'r',
           'exposed_per_day', 'ct_cases_per_day', 'mobility_limitation']
PLACES = ['home', 'work', 'school', 'transport', 'leisure', 'other']


def mini_age_structure(total):
    """HUS age histogram scaled to ~`total` agents (every age keeps >= 1 agent)."""
    import ref_harness as rh
    hus = rh.hus_age_structure()
    s = (hus * (total / hus.sum())).round().astype(int).clip(lower=1)
    return s


KITCHEN_IVS = [
    ['test-all-with-symptoms', '2020-02-20'],
    ['test-only-severe-symptoms', '2020-03-10', 40],
    ['test-with-contact-tracing', '2020-03-25', 60],
    ['test-all-with-symptoms', '2020-05-20'],
    ['test-with-contact-tracing', '2020-06-10', 100],
    ['limit-mobility', '2020-03-15', 60, 0, 70, 'other'],
    ['limit-mobility', '2020-03-20', 30],
    ['limit-mobility', '2020-04-01', 100, 7, 18, 'school'],
    ['limit-mobility', '2020-04-10', 40, 20, 64, 'work'],
    ['limit-mobility', '2020-05-15', 10],
    ['limit-mobility', '2020-06-20', 0, 7, 18, 'school'],
    ['wear-masks', '2020-03-18', 50, None, None, 'transport'],
    ['wear-masks', '2020-04-08', 80, 15, None, None],
    ['wear-masks', '2020-05-10', 100, None, 60, 'leisure'],
    ['build-new-hospital-beds', '2020-04-05', 6],
    ['build-new-icu-units', '2020-04-12', 2],
    ['vaccinate', '2020-03-01', 700, 70, None],
    ['vaccinate', '2020-03-20', 1400, 30, 69],
    ['vaccinate', '2020-04-20', 0, 70, None],
    ['vaccinate', '2020-05-01', 2100, None, None],
    ['import-infections', '2020-02-19', 30],
    ['import-infections', '2020-02-25', 40, 'b1.1.7'],
    ['import-infections', '2020-03-05', 60],
    ['import-infections-weekly', '2020-03-10', 30, 40],
    ['import-infections-weekly', '2020-05-10', 12, 100],
    ['import-infections', '2020-06-01', 50, 'b1.1.7'],
]

IMPORT_ONLY_IVS = [
    ['import-infections', '2020-02-19', 40],
    ['import-infections', '2020-02-22', 40],
    ['import-infections-weekly', '2020-03-01', 20],
]


def scenarios():
    """name -> dict(seed, days, pop ('hus' | int total), variables overrides, interventions)."""
    sc = {}
    for seed in (0, 1, 2, 3, 7, 1234):
        sc['hus_default_s%d' % seed] = dict(seed=seed, days=365, pop='hus', variables={},
                                            interventions='default')
    for seed in range(8):
        sc['mini_default_s%d' % seed] = dict(seed=seed, days=200, pop=20000,
                                             variables=dict(hospital_beds=12, icu_units=2),
                                             interventions='default')
    for seed in range(4):
        sc['mini_imports_s%d' % seed] = dict(seed=seed, days=150, pop=20000,
                                             variables=dict(hospital_beds=2600, icu_units=300),
                                             interventions=IMPORT_ONLY_IVS)
    for seed in range(6):
        sc['mini_kitchen_s%d' % seed] = dict(seed=100 + seed, days=200, pop=30000,
                                             variables=dict(hospital_beds=10, icu_units=1,
                                                            p_icu_death_no_beds=60.0),
                                             interventions=KITCHEN_IVS)
    # set_initial_state (main.pyx:1452-1516): every branch populated; a second shape where beds and
    # ICU units run out during the initial hospitalisations
    for seed in range(4):
        sc['mini_initial_s%d' % seed] = dict(seed=seed, days=120, pop=20000,
                                             variables=dict(hospital_beds=40, icu_units=6),
                                             interventions='default',
                                             ipc=dict(dead=5, in_icu=4, in_ward=12, confirmed_cases=230,
                                                      incubating=60, ill=45, recovered=300))
    for seed in range(2):
        sc['mini_initial_full_s%d' % seed] = dict(seed=20 + seed, days=80, pop=20000,
                                                  variables=dict(hospital_beds=9, icu_units=2,
                                                                 p_icu_death_no_beds=50.0),
                                                  interventions=KITCHEN_IVS,
                                                  ipc=dict(dead=3, in_icu=6, in_ward=14, confirmed_cases=90,
                                                           incubating=40, ill=30, recovered=100))
    # round 2: the corners a randomised differential test (tests/diff_a_b.py) found the parallel formulation wrong in, and
    # the one it still does not reproduce -- recorded so that the sequential oracle they are judged against is itself
    # pinned on them.  (a) fewer recovered than incubating: the walk of set_initial_state stops short of the ward / ICU
    # slots; (b) an initial condition with ZERO hospital beds: transfer_to_icu of agents just refused a bed; (c) imports
    # listed before / after a contact-tracing intervention of the same date (interventions run in list order).
    for seed in range(2):
        sc['mini_initial_short_s%d' % seed] = dict(seed=40 + seed, days=60, pop=20000,
                                                   variables=dict(hospital_beds=32, icu_units=3),
                                                   interventions='default',
                                                   ipc=dict(dead=2, in_icu=5, in_ward=7, confirmed_cases=135,
                                                            incubating=45, ill=11, recovered=33))
        sc['mini_initial_nobeds_s%d' % seed] = dict(seed=50 + seed, days=60, pop=20000,
                                                    variables=dict(hospital_beds=0, icu_units=2, p_icu_death_no_beds=50.0,
                                                                   p_hospital_death_no_beds=50.0),
                                                    interventions='default',
                                                    ipc=dict(dead=3, in_icu=5, in_ward=7, confirmed_cases=90,
                                                             incubating=30, ill=20, recovered=100))
        sc['mini_order_import_first_s%d' % seed] = dict(seed=60 + seed, days=60, pop=20000,
                                                        variables=dict(hospital_beds=11, icu_units=1, infectiousness_multiplier=0.45),
                                                        interventions=[['import-infections', '2020-02-18', 38],
                                                                       ['test-with-contact-tracing', '2020-02-18', 85]])
        sc['mini_order_tracing_first_s%d' % seed] = dict(seed=60 + seed, days=60, pop=20000,
                                                         variables=dict(hospital_beds=11, icu_units=1, infectiousness_multiplier=0.45),
                                                         interventions=[['test-with-contact-tracing', '2020-02-18', 85],
                                                                        ['import-infections', '2020-02-18', 38]])
    sc['hus_initial_s5'] = dict(seed=5, days=120, pop='hus', variables={}, interventions='default',
                                ipc=dict(dead=20, in_icu=30, in_ward=80, confirmed_cases=1500,
                                         incubating=600, ill=400, recovered=5000))
    return sc


def run_scenario(args):
    name, spec = args
    import ref_harness as rh
    st = rh.setup()
    if spec['pop'] == 'hus':
        ages = rh.hus_age_structure()
    else:
        ages = mini_age_structure(spec['pop'])
    ctx = rh.make_context(spec['seed'], variables=spec['variables'], age_structure=ages,
                          interventions=spec['interventions'], ipc=spec.get('ipc'))
    D = spec['days']
    vnames = list(ctx.disease.variant_names)
    pop = np.zeros((D, 13, 9), dtype=np.int32)
    scal = np.zeros((D, 7), dtype=np.float64)
    dc = np.zeros((D, 6), dtype=np.int32)
    ibv = np.zeros((D, len(vnames)), dtype=np.int32)
    for d in range(D):
        s = ctx.generate_state()
        for i, k in enumerate(POP13):
            pop[d, i] = s[k]
        for i, k in enumerate(SCALARS):
            scal[d, i] = s[k]
        for i, k in enumerate(PLACES):
            dc[d, i] = s['daily_contacts'][k]
        for i, k in enumerate(vnames):
            ibv[d, i] = s['infected_by_variant'][k]
        ctx.iterate()
    per_age = np.stack([ctx.get_population_stats(w) for w in ('dead', 'all_infected', 'all_detected')])
    ivs = rh.default_variables()['interventions'] if spec['interventions'] == 'default' \
        else spec['interventions']
    meta = dict(name=name, seed=spec['seed'], days=D, variables=spec['variables'],
                interventions=ivs, variant_names=vnames, ipc=spec.get('ipc'),
                age_counts=[int(x) for x in ages.values], pop13=POP13, scalars=SCALARS,
                places=PLACES)
    out = os.path.join(HERE, name + '.npz')
    np.savez_compressed(out, pop=pop, scalars=scal, daily_contacts=dc, infected_by_variant=ibv,
                        per_age_final=per_age.astype(np.int32),
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    return name, int(pop[-1, 3].sum())


def gen_inputs():
    """Input data fixtures: HUS age histogram, FI contact rows (data/contact_matrix.csv) in the
    long form the reference feeds to ContactMatrix (calc/simulation.py:74-100), defaults."""
    import ref_harness as rh
    rh.setup()
    ages = rh.hus_age_structure()
    df = rh.contacts_per_day()
    rows = [[t.place_type, int(t.participant_age), int(t.contact_age[0]), int(t.contact_age[1]),
             float(t.contacts)] for t in df.itertuples()]
    v = rh.default_variables()
    inputs = dict(
        hus_age_counts=[int(x) for x in ages.values],
        contacts_per_day_rows=rows,
        contacts_per_day_columns=['place_type', 'participant_age', 'contact_age_min',
                                  'contact_age_max', 'contacts'],
        variable_defaults={k: v[k] for k in v if k not in ('scenarios', 'area_name_long')},
    )
    with open(os.path.join(HERE, 'inputs.json'), 'w') as f:
        json.dump(inputs, f)
    print('inputs.json: N=%d, %d contact rows' % (ages.sum(), len(rows)))

    # Package input data (reina_model_amd/data/): the FI rows of data/contact_matrix.csv in their
    # wide form + the HUS age histogram. Data only; `reina_model_amd.datasets` re-expands it.
    # values are taken from the DataFrame the reference itself parsed (pandas' C float parser is
    # not round-trip exact, so re-parsing the CSV text with float() would differ in the last ulp)
    wdf = rh.setup()['datasets'].get_contacts_for_country()
    ccols = [c for c in wdf.columns if c not in ('place_type', 'participant_age')]
    cgroups = [[int(y) for y in c.split('-')] for c in ccols]
    wide = []
    for _, t in wdf.iterrows():
        lo, hi = [int(y) for y in t['participant_age'].split('-')]
        wide.append([t['place_type'], lo, hi, [float(t[c]) for c in ccols]])
    pkg = dict(country=v['country'], area_name=v['area_name'], contact_groups=cgroups,
               contact_rows=wide, age_counts=[int(x) for x in ages.values])
    pkg_dir = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'reina_model_amd', 'data')
    os.makedirs(pkg_dir, exist_ok=True)
    with open(os.path.join(pkg_dir, 'fi_hus.json'), 'w') as f:
        json.dump(pkg, f)
    print('reina_model_amd/data/fi_hus.json: %d wide rows' % len(wide))


def gen_casefile():
    """Adds the area's case file rows (data/hosp_cases_hus.csv as the reference reads it in
    calc/datasets.py:143-177: date -> dead, in_icu, in_ward, confirmed) to the package data."""
    import pandas as pd
    import ref_harness as rh
    rh.setup()
    import calc.datasets as ds
    df = pd.read_csv(ds.AREA_CASEFILES['HUS'], header=0, index_col=0)
    rows = [[str(d), int(r['dead']), int(r['in_icu']), int(r['in_ward']), int(r['confirmed'])]
            for d, r in df.iterrows()]
    path = os.path.join(HERE, '..', '..', 'reina_model_amd', 'data', 'fi_hus.json')
    with open(path) as f:
        data = json.load(f)
    data['case_rows'] = rows
    data['case_columns'] = ['date', 'dead', 'in_icu', 'in_ward', 'confirmed']
    with open(path, 'w') as f:
        json.dump(data, f)
    print('fi_hus.json: %d case rows (%s .. %s)' % (len(rows), rows[0][0], rows[-1][0]))


def gen_rng_kat():
    """G4: known answers at the RandomPool boundary (simrandom.pyx:13-55)."""
    import ref_harness as rh
    rh.setup()
    import rngshim
    from cythonsim.simrandom import RandomPool
    out = {}
    for seed in (0, 1, 4321):
        rp = RandomPool(seed)
        out['s%d_double' % seed] = rngshim.draw_pattern(rp, 'd' * 1000)
        rp = RandomPool(seed)
        out['s%d_uint32' % seed] = rngshim.draw_pattern(rp, 'u' * 1001)
        rp = RandomPool(seed)
        pat = ('uduuddudu' * 120)[:1000]
        out['s%d_mixed' % seed] = rngshim.draw_pattern(rp, pat)
        rp = RandomPool(seed)
        out['s%d_lognormal_0_0.5' % seed] = rngshim.draw_pattern(rp, 'l' * 20000, 0.0, 0.5)
        for mu, cv in ((5.1, 0.86), (21.0, 0.45), (18.8, 0.45)):
            rp = RandomPool(seed)
            out['s%d_gamma_%g_%g' % (seed, mu, cv)] = rngshim.draw_pattern(rp, 'g' * 20000, mu, cv)
        rp = RandomPool(seed)
        # interleave everything, including the uint32 half-word buffer across other draws
        pat = ('ulugdgul' * 500)
        out['s%d_interleaved_5.1_0.86' % seed] = rngshim.draw_pattern(rp, pat, 5.1, 0.86)
        # legacy global shuffle used for agent->age assignment (main.pyx:1435-1436)
        RandomPool(seed)
        idx = np.arange(1000, dtype=np.int32)
        np.random.shuffle(idx)
        out['s%d_legacy_shuffle_1000' % seed] = idx
    out['mixed_pattern'] = np.frombuffer(('uduuddudu' * 120)[:1000].encode(), dtype=np.uint8)
    out['interleaved_pattern'] = np.frombuffer(('ulugdgul' * 500).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'rng_kat.npz'), **out)
    print('rng_kat.npz written')


def gen_samples():
    """G3: Context.sample() draws (main.pyx:2047-2101), 10 000 each, fresh Context per query so
    the stream position is known (seed 4321, mini population, no interventions)."""
    import ref_harness as rh
    rh.setup()
    ages = mini_age_structure(5000)
    out = {}
    queries = []
    for age in (5, 25, 45, 65, 85):
        queries.append(('contacts_per_day', age, None))
        queries.append(('symptom_severity', age, None))
    queries.append(('incubation_period', 45, None))
    for sev in ('ASYMPTOMATIC', 'MILD', 'SEVERE', 'CRITICAL', 'FATAL'):
        queries.append(('illness_period', 45, sev))
        queries.append(('hospitalization_period', 45, sev))
        queries.append(('icu_period', 45, sev))
        queries.append(('onset_to_removed_period', 45, sev))
    for what, age, sev in queries:
        ctx = rh.make_context(4321, age_structure=ages, interventions=[])
        out['%s|%d|%s' % (what, age, sev or '')] = np.asarray(ctx.sample(what, age, sev), dtype=np.int32)
    # the same query issued twice on one context (stream continues)
    ctx = rh.make_context(77, age_structure=ages, interventions=[])
    a = np.asarray(ctx.sample('contacts_per_day', 33, None), dtype=np.int32)
    b = np.asarray(ctx.sample('incubation_period', 33, None), dtype=np.int32)
    out['chain77|contacts_per_day|33'] = a
    out['chain77|incubation_period|33'] = b
    out['age_counts'] = np.asarray(ages.values, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, 'samples.npz'), **out)
    print('samples.npz written (%d queries)' % len(queries))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    ap.add_argument('--jobs', type=int, default=6)
    ap.add_argument('--skip-aux', action='store_true')
    a = ap.parse_args()
    import ref_harness as rh
    rh.setup()  # build the extension once before forking workers
    if not a.skip_aux and not a.only:
        gen_inputs()
        gen_casefile()
        gen_rng_kat()
        gen_samples()
    sc = scenarios()
    todo = [(k, v) for k, v in sc.items() if not a.only or k.startswith(a.only)]
    with multiprocessing.Pool(a.jobs) as pool:
        for name, tot in pool.imap_unordered(run_scenario, todo):
            print('%s done (all_infected at end: %d)' % (name, tot), flush=True)


if __name__ == '__main__':
    main()
