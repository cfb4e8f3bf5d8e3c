"""Model variables: names, units and default values of the reference's `variables.py:227-435`, its override sets
(`VARIABLE_OVERRIDE_SETS`, variables.py:10-216) and the module-level accessors a driver script uses (variables.py:451-536).

The engine keeps the reference's variable NAMES and UNITS (every `p_*` / `ratio_*` is a
percentage, converted by `simulation.create_disease_params` exactly like
calc/simulation.py:50-61) so existing scenario definitions run unchanged.  Default VALUES are the
reference's HUS defaults (recorded from the reference by tests/golden/make_golden.py into
tests/golden/inputs.json; this literal is generated from that record).

Override sets (round 6): `VARIABLE_OVERRIDE_SETS['turku']` -- Turku's beds / ICU units, its intervention history to 2021-06 (nine
contact-tracing steps, place-specific mask ladders, weekly imports with a growing variant share) and its scenarios with
`add_interventions` (`astra-zeneca`: ['vaccinate', '2021-03-15', 2000, 25, 55]) -- is data recorded from the module the reference
imports (tests/golden/make_turku.py -> data/override_sets.json).  As in the reference, the environment variable
VARIABLE_OVERRIDE_SET selects a set for the process (variables.py:218-220, :436-437: VARIABLE_DEFAULTS.update(set)); a caller
can also ask for one explicitly: `copy_variables(override_set='turku')`.

The Flask-session store behind the reference's accessors is web plumbing (out of scope); `get_variable` /
`set_variable` / `reset_variable(s)` / `allow_set_variable` keep their names and their behaviour OUTSIDE a request context: a
process-wide override table that `set_variable` may only write inside `allow_set_variable()`.
"""
import copy
import json
import os
from contextlib import contextmanager

VARIABLE_DEFAULTS = {'area_name': 'HUS',
 'country': 'FI',
 'max_age': 100,
 'simulation_days': 565,
 'start_date': '2020-02-18',
 'hospital_beds': 2600,
 'icu_units': 300,
 'p_mask_protects_wearer': 10.0,
 'p_mask_protects_others': 70.0,
 'infectiousness_multiplier': 0.55,
 'p_susceptibility': [[0, 34.0],
                      [10, 67.0],
                      [20, 100.0],
                      [30, 100.0],
                      [40, 100.0],
                      [50, 100.0],
                      [60, 124.0],
                      [70, 147.0],
                      [80, 147.0],
                      [90, 147.0]],
 'p_asymptomatic_infection': 0.8,
 'p_symptomatic': [[0, 50.0],
                   [10, 55.0],
                   [20, 60.0],
                   [30, 65.0],
                   [40, 70.0],
                   [50, 75.0],
                   [60, 80.0],
                   [70, 85.0],
                   [80, 90.0],
                   [90, 90.0]],
 'p_severe': [[0, 0.05],
              [10, 0.165],
              [20, 0.72],
              [30, 2.08],
              [40, 3.43],
              [50, 7.65],
              [60, 13.28],
              [70, 20.655],
              [80, 24.57],
              [90, 24.57]],
 'p_critical': [[0, 0.003],
                [10, 0.008],
                [20, 0.036],
                [30, 0.104],
                [40, 0.216],
                [50, 0.933],
                [60, 3.639],
                [70, 8.923],
                [80, 17.42],
                [90, 17.42]],
 'p_fatal': [[0, 0.002],
             [10, 0.002],
             [20, 0.01],
             [30, 0.032],
             [40, 0.098],
             [50, 0.265],
             [60, 0.766],
             [70, 2.439],
             [80, 8.292],
             [90, 16.19]],
 'p_death_outside_hospital': [[0, 0.0],
                              [10, 0.0],
                              [20, 0.0],
                              [30, 0.0],
                              [40, 0.0],
                              [50, 0.0],
                              [60, 1.0],
                              [70, 6.0],
                              [80, 50.0],
                              [90, 55.0]],
 'p_hospital_death_no_beds': 20.0,
 'p_icu_death_no_beds': 100.0,
 'mean_incubation_duration': 5.1,
 'mean_duration_from_onset_to_death': 18.8,
 'mean_duration_from_onset_to_recovery': 21.0,
 'ratio_of_duration_before_hospitalisation': 30.0,
 'ratio_of_duration_in_ward': 15.0,
 'imported_infection_ages': [[0, 15.0], [20, 40.0], [40, 40.0], [60, 5.0], [70, 0]],
 'incubating_at_simulation_start': 0,
 'ill_at_simulation_start': 0,
 'recovered_at_simulation_start': 0,
 'interventions': [['test-all-with-symptoms', '2020-02-20'],
                   ['test-only-severe-symptoms', '2020-03-15', 25],
                   ['test-only-severe-symptoms', '2020-03-30', 50],
                   ['test-only-severe-symptoms', '2020-04-15', 70],
                   ['test-with-contact-tracing', '2020-06-15', 30],
                   ['test-with-contact-tracing', '2020-09-15', 30],
                   ['limit-mobility', '2020-03-15', 80, 0, 70, 'other'],
                   ['limit-mobility', '2020-08-15', 50, 0, 70, 'other'],
                   ['limit-mobility', '2020-04-01', 5],
                   ['limit-mobility', '2020-05-01', 20],
                   ['limit-mobility', '2020-07-01', 10],
                   ['limit-mobility', '2020-09-01', 10],
                   ['limit-mobility', '2020-09-15', 10],
                   ['limit-mobility', '2020-10-01', 0],
                   ['wear-masks', '2020-07-01', 80, 65, None, None],
                   ['limit-mobility', '2020-03-12', 0, 7, 12, 'school'],
                   ['limit-mobility', '2020-04-01', 100, 19, None, 'school'],
                   ['limit-mobility', '2020-05-30', 100, 7, 12, 'school'],
                   ['limit-mobility', '2020-05-30', 100, 13, 15, 'school'],
                   ['limit-mobility', '2020-05-30', 100, 16, 18, 'school'],
                   ['limit-mobility', '2020-08-12', 0, 7, 12, 'school'],
                   ['limit-mobility', '2020-08-12', 0, 13, 15, 'school'],
                   ['limit-mobility', '2020-08-12', 0, 16, 18, 'school'],
                   ['limit-mobility', '2020-08-12', 20, 19, None, 'school'],
                   ['import-infections', '2020-02-22', 20],
                   ['import-infections', '2020-03-05', 50],
                   ['import-infections', '2020-03-07', 80],
                   ['import-infections', '2020-03-09', 120],
                   ['import-infections', '2020-03-11', 80],
                   ['import-infections', '2020-03-13', 20],
                   ['import-infections', '2020-03-15', 20],
                   ['import-infections-weekly', '2020-07-01', 50],
                   ['import-infections', '2020-08-15', 50],
                   ['import-infections', '2020-09-01', 100],
                   ['import-infections', '2020-09-07', 100],
                   ['import-infections', '2020-09-15', 100],
                   ['import-infections', '2020-10-01', 50],
                   ['import-infections', '2020-10-15', 100],
                   ['import-infections', '2020-11-01', 100],
                   ['import-infections', '2020-11-15', 100]],
 'variants': [{'name': 'b1.1.7', 'infectiousness_multiplier': 0.9075}],
 'active_scenario': 'default',
 'sample_limit_mobility': 0,
 'random_seed': 0,
 'scenarios': [{'id': 'default', 'label': 'default', 'description': ''}]}



def _load_override_sets():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'override_sets.json')
    with open(path, encoding='utf-8') as f:
        return json.load(f)


VARIABLE_OVERRIDE_SETS = _load_override_sets()
HUS_DEFAULTS = copy.deepcopy(VARIABLE_DEFAULTS)    # the defaults with no override set applied, whatever the environment says

_variable_override_set = os.getenv('VARIABLE_OVERRIDE_SET')
if _variable_override_set:
    assert _variable_override_set in VARIABLE_OVERRIDE_SETS
    VARIABLE_DEFAULTS.update(copy.deepcopy(VARIABLE_OVERRIDE_SETS[_variable_override_set]))   # variables.py:436-437

# variables.py:440-441, :449: overrides set later programmatically
_variable_overrides = {}
_allow_variable_set = False


def set_variable(var_name, value):
    """variables.py:452-467 outside a request context: only inside `allow_set_variable()`"""
    assert var_name in VARIABLE_DEFAULTS
    assert isinstance(value, type(VARIABLE_DEFAULTS[var_name]))
    if not _allow_variable_set:
        raise Exception('Should not set variable outside of request context')
    _variable_overrides[var_name] = value


def get_variable(var_name, var_store=None):
    """variables.py:470-491: the store given, else the process-wide override, else the default; lists are handed out as copies"""
    out = None
    if var_store is not None:
        out = var_store.get(var_name)
    elif var_name in _variable_overrides:
        out = _variable_overrides[var_name]
    if out is None:
        out = VARIABLE_DEFAULTS[var_name]
    if isinstance(out, list):
        return list(out)
    return out


def reset_variable(var_name):
    _variable_overrides.pop(var_name, None)


def reset_variables():
    _variable_overrides.clear()


@contextmanager
def allow_set_variable():
    """variables.py:525-536"""
    global _allow_variable_set
    old = _allow_variable_set
    _allow_variable_set = True
    try:
        yield None
    finally:
        _allow_variable_set = old


def copy_variables(override_set=None, **overrides):
    """A fresh, caller-owned copy of the variables (what a driver passes as `variables`; variables.py:518-522: every name through
    `get_variable`, so process-wide overrides show), then -- extensions of this package -- the named override set on top of the
    plain defaults (`override_set='turku'`, or `'hus'` / `'none'` for the defaults whatever VARIABLE_OVERRIDE_SET says) and
    `overrides`, every key of which must be a known variable."""
    unknown = sorted(set(overrides) - set(VARIABLE_DEFAULTS))
    if unknown:
        raise KeyError('unknown variable(s): %s' % ', '.join(unknown))
    if override_set is None:
        out = {name: copy.deepcopy(get_variable(name)) for name in VARIABLE_DEFAULTS}
    else:
        out = copy.deepcopy(HUS_DEFAULTS)
        if str(override_set).lower() not in ('hus', 'none', ''):
            if override_set not in VARIABLE_OVERRIDE_SETS:
                raise KeyError('unknown override set %r (have %s)' % (override_set, ', '.join(sorted(VARIABLE_OVERRIDE_SETS))))
            out.update(copy.deepcopy(VARIABLE_OVERRIDE_SETS[override_set]))
    out.update(copy.deepcopy(overrides))
    return out
