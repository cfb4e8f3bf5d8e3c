"""Preset scenarios of the reference (`scenarios.py:56-198`), reduced to what the simulator needs:
an id, extra interventions appended to the default list, variable overrides, and the one scenario
that rewrites the default list (`looser-restrictions-to-start-with` halves every limit-mobility
reduction, scenarios.py:181-190).  Names/descriptions/translations are UI and out of scope."""
import copy

from .variables import VARIABLE_DEFAULTS


def _halve_mobility(ivs):
    out = []
    for iv in ivs:
        iv = list(iv)
        if iv[0] == 'limit-mobility':
            iv[2] = iv[2] // 2
        out.append(iv)
    return out


SCENARIOS = {
    'default': dict(interventions=[]),
    'summer-boogie': dict(interventions=[['limit-mobility', '2020-05-15', 30]]),
    'mitigation': dict(interventions=(
        [[k, d, n] for d in ('2020-06-30', '2020-07-15', '2020-07-30', '2020-08-15', '2020-08-30')
         for k, n in (('build-new-icu-units', 150), ('build-new-hospital-beds', 300))] +
        [['limit-mobility', d, r] for d, r in (
            ('2020-06-01', 30), ('2020-07-01', 40), ('2020-08-01', 30), ('2020-09-15', 40),
            ('2020-10-15', 30), ('2020-12-15', 20), ('2021-01-15', 5), ('2021-02-15', 0))])),
    'hammer-and-dance': dict(interventions=(
        [['test-with-contact-tracing', d, e] for d, e in (
            ('2020-05-01', 30), ('2020-06-01', 40), ('2020-07-01', 50), ('2020-08-01', 60))] +
        [['limit-mobility', d, r] for d, r in (
            ('2020-05-01', 30), ('2020-06-24', 25), ('2020-08-15', 10), ('2020-12-06', 15))])),
    'looser-restrictions-to-start-with': dict(interventions=[], rewrite=_halve_mobility),
}


def scenario_variables(scenario_id, base=None):
    """Variables dict after `Scenario.apply()` (scenarios.py:41-53) on the defaults (or `base`)."""
    if scenario_id not in SCENARIOS:
        raise Exception('Scenario not found')
    sc = SCENARIOS[scenario_id]
    v = copy.deepcopy(VARIABLE_DEFAULTS if base is None else base)
    v['interventions'] = [list(iv) for iv in v['interventions']] + [list(iv) for iv in sc.get('interventions', [])]
    for key, val in sc.get('variables', {}).items():
        v[key] = val
    if 'rewrite' in sc:
        v['interventions'] = sc['rewrite'](v['interventions'])
    v['preset_scenario'] = scenario_id
    return v
