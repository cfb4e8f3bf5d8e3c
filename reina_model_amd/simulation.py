"""Driver with the contract of the reference's `calc/simulation.py:17-290`.

`simulate_individuals(variables)` builds the population / healthcare / disease parameter dicts the
same way (`create_disease_params` :50-61, `make_age_groups` :103-116), constructs the engine
`Context`, runs the day loop and returns the same two DataFrames `(df, adf)`:
  df  : one row per date, columns POP_ATTRS + STATE_ATTRS + EXPOSURES_ATTRS + ['us_per_infected']
  adf : per-date x (attr, age group) population counts.
Row d is the state BEFORE day d is simulated (the reference calls generate_state() before
iterate(), :195 vs :270).  The day loop itself runs on the GPU without per-day host round trips
(`Context.run`); `step_callback` is honoured every `callback_day_interval` days by running the
simulation in stretches.

Also here, with the reference's names: `sample_model_parameters` (:301-347), `simulate_monte_carlo`
/ `run_monte_carlo` (:350-385; the reference maps 1000 seeds over a pool of 8 processes, here the
seeds run as engine groups on the GPU, reina_model_amd/ensemble.py) and the `python -m` day table
(:388-460): `python -m reina_model_amd.simulation [--days N] [--seed S] [--scenario ID]`.
"""
import time
from datetime import date, timedelta

import numpy as np

from . import datasets
from . import engine as _eng
from . import model
from .interventions import get_active_interventions, iv_tuple_to_obj
from .variables import copy_variables

POP_ATTRS = ['susceptible', 'vaccinated', 'infected', 'detected', 'all_detected', 'in_ward', 'in_icu',
             'dead', 'non_hospital_deaths', 'recovered', 'all_infected', 'new_infections']
EXPOSURES_ATTRS = ['exposures_home', 'exposures_work', 'exposures_school', 'exposures_transport',
                   'exposures_leisure', 'exposures_other']
STATE_ATTRS = ['exposed_per_day', 'available_hospital_beds', 'available_icu_units', 'total_icu_units',
               'ct_cases_per_day', 'r', 'mobility_limitation']


class ExecutionInterrupted(Exception):
    pass


def create_disease_params(variables):
    """calc/simulation.py:50-61: every p_* / ratio_* variable is a percentage."""
    kwargs = {}
    for key in model.DISEASE_PARAMS:
        val = variables[key]
        if key.startswith('p_') or key.startswith('ratio_'):
            if isinstance(val, list):
                val = [(age, sev / 100) for age, sev in val]
            else:
                val = val / 100
        kwargs[key] = val
    return kwargs


def make_context(variables, age_counts=None, seed=None, interventions=None, device='cuda:0',
                 engine_factory=None, comm=None, ipc=None, strict=False):
    """Build a Context the way calc/simulation.py:148-180 does.  `ipc`: an InitialPopulationCondition,
    a dict of its fields, None (no initial condition), or 'auto' = what simulate_individuals passes,
    datasets.get_initial_population_condition(variables) (calc/simulation.py:152)."""
    if isinstance(ipc, str) and ipc == 'auto':
        ipc = datasets.get_initial_population_condition(variables)
    if age_counts is None:
        age_counts = datasets.get_population_for_area(variables['area_name'])
    age_to_group = datasets.make_age_groups(variables['max_age'])
    groups = list(np.unique(age_to_group))
    pop_params = dict(
        age_structure=np.asarray(age_counts),
        contacts_per_day=datasets.get_contacts_per_day(variables['country']),
        initial_population_condition=datasets.InitialPopulationCondition(**ipc) if isinstance(ipc, dict) else ipc,
        age_groups=dict(labels=groups, age_indices=[groups.index(x) for x in age_to_group]),
        imported_infection_ages=variables['imported_infection_ages'],
    )
    hc = dict(hospital_beds=variables['hospital_beds'], icu_units=variables['icu_units'])
    ctx = model.Context(pop_params, hc, create_disease_params(variables), variables['start_date'],
                        random_seed=variables['random_seed'] if seed is None else seed,
                        device=device, engine_factory=engine_factory, comm=comm, strict=strict)
    if interventions is None:
        ivs = get_active_interventions(variables)
    else:
        vnames = tuple(v['name'] for v in variables['variants'])
        ivs = [iv_tuple_to_obj(iv, vnames) for iv in interventions]
    for iv in ivs:
        ctx.add_intervention(iv)
    return ctx


def simulate_individuals(variables=None, step_callback=None, callback_day_interval=1, device='cuda:0',
                         engine_factory=None, age_counts=None):
    import pandas as pd
    if variables is None:
        variables = copy_variables()
    t0 = time.perf_counter()
    ctx = make_context(variables, age_counts=age_counts, device=device, engine_factory=engine_factory,
                       ipc=datasets.get_initial_population_condition(variables))   # calc/simulation.py:152
    start_date = date.fromisoformat(variables['start_date'])
    days = variables['simulation_days']
    age_groups = ctx.age_group_labels
    date_index = pd.date_range(start_date, periods=days)
    cols = POP_ATTRS + STATE_ATTRS + EXPOSURES_ATTRS + ['us_per_infected']
    rows = []
    ag_array = np.empty((days, len(POP_ATTRS), len(age_groups)), dtype='i')

    if step_callback is None:   # no progress reports wanted: one run, frames for all days at once
        last = time.perf_counter()
        hist = ctx.run(days)
        return _frames_from_history(ctx, hist, ctx.mobility_history, start_date, (time.perf_counter() - last) * 1000 / days)

    done = 0
    stretch = max(1, int(callback_day_interval))
    last = time.perf_counter()
    while done < days:
        n = min(stretch, days - done)
        hist = ctx.run(n)
        now = time.perf_counter()
        ms_per_day = (now - last) * 1000 / n
        last = now
        for k in range(n):
            s = ctx.state_from_counters(hist[k], mobility_factor=ctx.mobility_history[k])
            for idx, attr in enumerate(POP_ATTRS):
                ag_array[done + k, idx, :] = s[attr]
            rec = {attr: s[attr].sum() for attr in POP_ATTRS}
            for a in STATE_ATTRS:
                rec[a] = s[a]
            for place, nr in s['daily_contacts'].items():
                rec['exposures_%s' % place] = nr
            rec['us_per_infected'] = ms_per_day * 1000 / rec['infected'] if rec['infected'] else 0
            rows.append(rec)
        done += n
        df = pd.DataFrame(rows, index=date_index[:done], columns=cols).reindex(date_index)
        if not step_callback(df):
            raise ExecutionInterrupted()
    df = pd.DataFrame(rows, index=date_index, columns=cols).astype(np.float64).astype(object)
    adf = pd.DataFrame(
        ag_array.flatten(),
        index=pd.MultiIndex.from_product([date_index, POP_ATTRS, age_groups], names=['date', 'attr', 'age_group']),
        columns=['pop'])
    adf = adf.unstack('attr').unstack('age_group')
    adf.columns = adf.columns.droplevel()
    return df, adf


def _frames_from_history(ctx, hist, mobility_history, start_date, ms_per_day=0.0, want_adf=True):
    """(df, adf) of simulate_individuals from a recorded counter history[days, COUNTER_WORDS]; all
    days at once (the per-day dict path, Context.state_from_counters, gives the same numbers -- checked
    in tests/test_host_logic.py)."""
    import pandas as pd
    days = hist.shape[0]
    A = _eng.MAX_AGES
    age_groups = ctx.age_group_labels
    ngroups = len(age_groups)
    date_index = pd.date_range(start_date, periods=days)
    cols = POP_ATTRS + STATE_ATTRS + EXPOSURES_ATTRS + ['us_per_infected']
    # age -> report group sums as one matrix product per attribute: [days, nr_ages] @ [nr_ages, groups]
    onehot = np.zeros((ctx.nr_ages, ngroups), dtype=np.int64)
    onehot[np.arange(ctx.nr_ages), ctx.age_group_indices[:ctx.nr_ages]] = 1
    ag_array = np.empty((days, len(POP_ATTRS), ngroups), dtype='i')
    data = {}
    for idx, attr in enumerate(POP_ATTRS):
        ci = _eng.C_NAMES.index(attr)
        per_age = hist[:, ci * A: ci * A + ctx.nr_ages].astype(np.int64)
        ag_array[:, idx, :] = per_age @ onehot
        data[attr] = ag_array[:, idx, :].sum(axis=1)
    sc = hist[:, _eng.C_NR * A:].astype(np.int64)
    infections, infectors = sc[:, _eng.S_TOTAL_INFECTIONS], sc[:, _eng.S_TOTAL_INFECTORS]
    r = np.zeros(days, dtype=np.float64)   # 0 until more than 5 infectors were seen (main.pyx:1827)
    ok = infectors > 5
    r[ok] = infections[ok] / infectors[ok]
    data['exposed_per_day'] = sc[:, _eng.S_EXPOSED_PER_DAY]
    data['available_hospital_beds'] = sc[:, _eng.S_AVAILABLE_BEDS]
    data['available_icu_units'] = sc[:, _eng.S_AVAILABLE_ICU]
    data['total_icu_units'] = sc[:, _eng.S_ICU_UNITS]
    data['ct_cases_per_day'] = sc[:, _eng.S_CT_CASES_PER_DAY]
    data['r'] = r
    data['mobility_limitation'] = 1 - np.asarray(mobility_history[:days], dtype=np.float64)
    for i, place in enumerate(model.PLACES):
        data['exposures_%s' % place] = sc[:, _eng.S_DAILY_CONTACTS + i]
    infected = data['infected'].astype(np.float64)
    data['us_per_infected'] = np.where(infected > 0, ms_per_day * 1000 / np.where(infected > 0, infected, 1), 0)
    # (the reference assembles its frame from per-day dicts and reindexes it on every progress report: what reaches the caller
    # holds Python floats in object columns -- recorded from the real driver in tests/golden/frames_ref.json; same here)
    df = pd.DataFrame({c: np.asarray(data[c], dtype=np.float64) for c in cols}, index=date_index, columns=cols).astype(object)
    if not want_adf:
        return df, None
    adf = pd.DataFrame(
        ag_array.flatten(),
        index=pd.MultiIndex.from_product([date_index, POP_ATTRS, age_groups], names=['date', 'attr', 'age_group']),
        columns=['pop'])
    adf = adf.unstack('attr').unstack('age_group')
    adf.columns = adf.columns.droplevel()
    return df, adf


def sample_model_parameters(what, age, severity=None, variables=None, device='cuda:0', engine_factory=None):
    """calc/simulation.py:301-347: distribution of one per-agent quantity (`Context.sample`) in a
    one-agent-per-age population.  Returns the value -> share Series (severity names for
    'symptom_severity'; the day -> infectiousness Series for 'infectiousness').  The reference
    also prints the table and opens a matplotlib window; that is left to the caller."""
    import pandas as pd
    if variables is None:
        variables = copy_variables()
    max_age = variables['max_age']
    age_to_group = datasets.make_age_groups(max_age)
    groups = list(np.unique(age_to_group))
    pop_params = dict(
        age_structure=np.ones(max_age + 1, dtype=np.int64),
        contacts_per_day=datasets.get_contacts_per_day(variables['country']),
        age_groups=dict(labels=groups, age_indices=[groups.index(x) for x in age_to_group]),
        imported_infection_ages=variables['imported_infection_ages'],   # (the reference omits it and fails)
    )
    ctx = model.Context(pop_params, dict(hospital_beds=0, icu_units=0), create_disease_params(variables),
                        '2020-01-01', device=device, engine_factory=engine_factory)
    if variables.get('sample_limit_mobility', 0) != 0:
        # the reference passes (type, value) to apply_intervention, which only accepts an
        # intervention object (main.pyx:1880); the evident intent is a population-wide limit
        ctx.apply_intervention(iv_tuple_to_obj(['limit-mobility', None, variables['sample_limit_mobility']],
                                               tuple(v['name'] for v in variables['variants'])))
        ctx.contact_matrix.init_day()
    samples = ctx.sample(what, age, severity)
    if what == 'infectiousness':
        s = pd.Series(index=samples['day'], data=samples['val'])
        return s[s != 0].sort_index()
    s = pd.Series(samples)
    c = s.value_counts().sort_index()
    if what == 'symptom_severity':
        c.index = c.index.map(model.SEVERITY_TO_STR)
    return c / c.sum()


def simulate_monte_carlo(seed, variables=None, device='cuda:0', engine_factory=None, age_counts=None):
    """calc/simulation.py:350-359: one run with `random_seed = seed`; df gets a 'run' column."""
    v = copy_variables() if variables is None else dict(variables)
    v['random_seed'] = seed
    df, _ = simulate_individuals(v, device=device, engine_factory=engine_factory, age_counts=age_counts)
    df['run'] = seed
    return df


def run_monte_carlo(scenario_name, seeds=range(1000), device='cuda:0', group_size=64, days=None, write_csv=True,
                    age_counts=None, engine_factory=None, variables=None, ipc='auto'):
    """calc/simulation.py:362-385 with the seeds stepped as engine groups on one GPU instead of a
    process pool: one DataFrame with every run's per-date rows, columns as simulate_individuals
    plus 'run' and 'scenario'; written to reina_<scenario>.csv like the reference.  Every run starts from
    the scenario's initial population condition, as the reference's do (run_monte_carlo ->
    simulate_monte_carlo -> simulate_individuals, calc/simulation.py:152); `ipc` overrides it."""
    import pandas as pd
    from . import ensemble
    from .scenarios import scenario_variables
    v = scenario_variables(scenario_name, base=variables)
    days = v['simulation_days'] if days is None else days
    seeds = list(seeds)
    dfs = []
    for start in range(0, len(seeds), group_size):
        part = seeds[start:start + group_size]
        planner = make_context(v, age_counts=age_counts, seed=part[0], device=device, engine_factory=engine_factory, ipc=ipc)
        plan = planner.make_plan(days)
        members = [make_context(v, age_counts=age_counts, seed=sd, device=device, engine_factory=engine_factory, ipc=ipc)
                   for sd in part]
        t0 = time.perf_counter()
        hist = ensemble.run_group_plan(members, plan)
        ms_per_day = (time.perf_counter() - t0) * 1000 / days / len(part)
        for m, sd in enumerate(part):
            df, _ = _frames_from_history(members[m], hist[m], plan['mobility_history'],
                                         date.fromisoformat(v['start_date']), ms_per_day, want_adf=False)
            df['run'] = sd
            dfs.append(df)
        del members, planner
    df = pd.concat(dfs)
    df.index.name = 'date'
    df = df.reset_index()
    df['scenario'] = scenario_name
    if write_csv:
        df.to_csv('reina_%s.csv' % scenario_name, index=False)
    return df


def table_header():
    """header line of the day table the reference prints when run as a script (calc/simulation.py:411-421)"""
    header = '%-10s' % 'day'
    for attr in POP_ATTRS + ['ct_cases_per_day', 'r'] + ['exposures', 'us_per_infected']:
        header += '%15s' % attr
    return header


def table_row(rec):
    """one line of that table from a row of df (calc/simulation.py:423-441)"""
    s = '%-12s' % rec.name.date().isoformat()
    for attr in POP_ATTRS:
        s += '%15d' % rec[attr]
    s += '%15d' % rec['ct_cases_per_day']
    s += '%13.2f' % rec['r']
    s += '%15d' % sum(rec[x] for x in rec.index if 'exposures_' in x)
    if rec['infected']:
        s += '%13.2f' % rec['us_per_infected']
    return s


def main(argv=None):
    """The day table the reference prints when run as a script (calc/simulation.py:411-447)."""
    import argparse
    from .scenarios import scenario_variables
    ap = argparse.ArgumentParser(prog='python -m reina_model_amd.simulation')
    ap.add_argument('--days', type=int, default=None)
    ap.add_argument('--seed', type=int, default=None)
    ap.add_argument('--scenario', default='default')
    ap.add_argument('--device', default='cuda:0')
    ap.add_argument('--interval', type=int, default=1, help='days between printed rows')
    a = ap.parse_args(argv)
    v = scenario_variables(a.scenario)
    if a.days is not None:
        v['simulation_days'] = a.days
    if a.seed is not None:
        v['random_seed'] = a.seed
    print(table_header())

    def step_callback(df):
        for _, rec in df.dropna().iloc[-a.interval:].iterrows():
            print(table_row(rec))
        return True

    df, adf = simulate_individuals(v, step_callback=step_callback, callback_day_interval=a.interval, device=a.device)
    print(adf)
    return df, adf


if __name__ == '__main__':
    main()
