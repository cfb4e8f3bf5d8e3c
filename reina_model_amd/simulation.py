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
                 engine_factory=None, comm=None):
    """Build a Context the way calc/simulation.py:148-180 does."""
    if age_counts is None:
        age_counts = datasets.get_population_for_area(variables['area_name'])
    age_to_group = datasets.make_age_groups(variables['max_age'])
    groups = list(np.unique(age_to_group))
    pop_params = dict(
        age_structure=np.asarray(age_counts),
        contacts_per_day=datasets.get_contacts_per_day(variables['country']),
        initial_population_condition=None,
        age_groups=dict(labels=groups, age_indices=[groups.index(x) for x in age_to_group]),
        imported_infection_ages=variables['imported_infection_ages'],
    )
    hc = dict(hospital_beds=variables['hospital_beds'], icu_units=variables['icu_units'])
    ctx = model.Context(pop_params, hc, create_disease_params(variables), variables['start_date'],
                        random_seed=variables['random_seed'] if seed is None else seed,
                        device=device, engine_factory=engine_factory, comm=comm)
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
    ctx = make_context(variables, age_counts=age_counts, device=device, engine_factory=engine_factory)
    start_date = date.fromisoformat(variables['start_date'])
    days = variables['simulation_days']
    age_groups = ctx.age_group_labels
    date_index = pd.date_range(start_date, periods=days)
    cols = POP_ATTRS + STATE_ATTRS + EXPOSURES_ATTRS + ['us_per_infected']
    rows = []
    ag_array = np.empty((days, len(POP_ATTRS), len(age_groups)), dtype='i')

    done = 0
    stretch = days if step_callback is None else max(1, int(callback_day_interval))
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
        if step_callback is not None:
            df = pd.DataFrame(rows, index=date_index[:done], columns=cols).reindex(date_index)
            if not step_callback(df):
                raise ExecutionInterrupted()
    df = pd.DataFrame(rows, index=date_index, columns=cols)
    adf = pd.DataFrame(
        ag_array.flatten(),
        index=pd.MultiIndex.from_product([date_index, POP_ATTRS, age_groups], names=['date', 'attr', 'age_group']),
        columns=['pop'])
    adf = adf.unstack('attr').unstack('age_group')
    adf.columns = adf.columns.droplevel()
    return df, adf
