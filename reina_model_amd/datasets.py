"""Input data for the agent engine: age structure and POLYMOD-style contact rows.

Host-side counterpart of the reference loaders `calc/datasets.py:48-79`
(`get_population_for_area`, `get_contacts_for_country`) and of the long-form expansion in
`calc/simulation.py:74-100` (`get_contacts_per_day`).  The reference reads CSV/XLS with pandas;
here the FI rows of its contact matrix and the HUS age histogram ship as one small JSON
(`data/fi_hus.json`, produced by tests/golden/make_golden.py) and are expanded with plain
Python so that the row ORDER (which fixes the floating-point summation order of the per-age
totals, see contacts.py) is identical to the reference's melted DataFrame.
"""
import json
import os
from dataclasses import dataclass

import numpy as np

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')
_DATA = os.path.join(_DATA_DIR, 'fi_hus.json')
# area name (variables['area_name']) -> bundled file: the age histogram the reference computes from data/005_11re_2019.csv
# (a municipality, or the municipalities of a hospital district: calc/datasets.py:48-61) and the rows of the area's case file
# (AREA_CASEFILES, calc/datasets.py:82-86), recorded by tests/golden/make_golden.py / make_turku.py
_AREAS = {'HUS': 'fi_hus.json', 'Turku': 'fi_turku.json', 'Varsinais-Suomi': 'fi_varsinais-suomi.json'}
_cache = {}


def _load(area_name='HUS'):
    if area_name not in _AREAS:
        raise KeyError('no bundled data for area %r (have %s)' % (area_name, ', '.join(sorted(_AREAS))))
    if area_name not in _cache:
        with open(os.path.join(_DATA_DIR, _AREAS[area_name])) as f:
            _cache[area_name] = json.load(f)
    return _cache[area_name]


def get_population_for_area(area_name='HUS'):
    """Age histogram int64[A] (index = age) of a bundled area (calc/datasets.py:48-61)."""
    return np.asarray(_load(area_name)['age_counts'], dtype=np.int64)


def scaled_population(total, base=None):
    """Synthetic population with the HUS age shape: count[a] = round(base[a] * total / sum(base)),
    every age keeps >= 1 agent."""
    base = get_population_for_area() if base is None else np.asarray(base, dtype=np.int64)
    s = np.round(base * (float(total) / float(base.sum()))).astype(np.int64)
    return np.maximum(s, 1)


def get_contacts_per_day(country='FI'):
    """Long-form contact rows in the reference's order: for each contact-age column, for each
    (place, participant group) CSV row, one row per participant age in the group.

    Returns a list of tuples (place_type: str, participant_age: int, (cmin, cmax), contacts: float).
    """
    d = _load('HUS')            # (the FI rows of data/contact_matrix.csv travel with the HUS file)
    if country != d['country']:
        raise KeyError('no bundled contact matrix for country %r' % country)
    if 'rows' in _cache:
        return _cache['rows']
    rows = []
    for ci, (cmin, cmax) in enumerate(d['contact_groups']):
        for place, pmin, pmax, vals in d['contact_rows']:
            c = vals[ci]
            for p in range(pmin, pmax + 1):
                rows.append((place, p, (cmin, cmax), c))
    _cache['rows'] = rows   # callers treat the list as read-only
    return rows


def make_age_groups(max_age=100):
    """age -> report group label, 10-year bins with '80+' on top (calc/simulation.py:103-116)."""
    out = []
    for i in range(0, max_age + 1):
        grp = i // 10
        out.append('80+' if grp >= 8 else '%d–%d' % (grp * 10, grp * 10 + 9))
    return out


@dataclass
class InitialPopulationCondition:
    """calc/datasets.py:106-134: how many people are in which state when the simulation starts."""
    dead: int = 0
    in_icu: int = 0
    in_ward: int = 0
    confirmed_cases: int = 0
    infected_cases: int = 0
    incubating: int = 0
    ill: int = 0
    recovered: int = 0

    def has_initial_state(self):
        return bool(self.dead or self.in_icu or self.in_ward or self.confirmed_cases
                    or self.infected_cases or self.incubating or self.ill or self.recovered)

    def were_incubating(self):
        """everyone who contracted the virus at some point before the start"""
        return sum([self.dead, self.recovered, self.in_icu, self.in_ward, self.ill, self.incubating])

    def recovered_without_illness(self):
        return self.were_incubating() - self.were_ill()

    def were_ill(self):
        return sum([self.dead, self.recovered, self.in_icu, self.in_ward, self.ill])


def get_initial_population_condition(variables):
    """calc/datasets.py:143-177: measured numbers (dead, in ICU, in ward, confirmed) from the
    area's case file row of the start date, unmeasured ones from the variables; a start date the
    file does not list means an empty initial condition (as the reference, which prints a note)."""
    d = _load(variables['area_name'])
    for row in d['case_rows']:
        if row[0] == variables['start_date']:
            return InitialPopulationCondition(
                dead=row[1], in_icu=row[2], in_ward=row[3], confirmed_cases=row[4],
                ill=variables['ill_at_simulation_start'], incubating=variables['incubating_at_simulation_start'],
                recovered=variables['recovered_at_simulation_start'])
    return InitialPopulationCondition()
