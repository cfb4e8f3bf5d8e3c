"""Intervention objects the engine consumes.

Mirror of the reference's `common/interventions.py:59-156,159-323,337-339` as far as the
simulator needs it: `Context.apply_intervention` (cythonsim/main.pyx:1880-1960) reads only
`.type`, `.date` and `.get_param_values()`.  The UI-facing parts of the reference class (labels,
translations, choice widgets) are out of scope; parameter ids, order and kinds are kept so the
reference's tuple form (`['limit-mobility', '2020-03-15', 80, 0, 70, 'other']`) converts the same.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

PLACES = ('home', 'work', 'school', 'transport', 'leisure', 'other')

# type -> ordered parameters (id, kind); kind 'int' or 'choice'
INTERVENTION_SCHEMA: Dict[str, List[Tuple[str, str]]] = {
    'test-all-with-symptoms': [],
    'test-only-severe-symptoms': [('mild_detection_rate', 'int')],
    'test-with-contact-tracing': [('efficiency', 'int')],
    'limit-mobility': [('reduction', 'int'), ('min_age', 'int'), ('max_age', 'int'), ('place', 'choice')],
    'wear-masks': [('share_of_contacts', 'int'), ('min_age', 'int'), ('max_age', 'int'), ('place', 'choice')],
    'vaccinate': [('weekly_vaccinations', 'int'), ('min_age', 'int'), ('max_age', 'int')],
    'import-infections': [('amount', 'int'), ('variant', 'choice')],
    'import-infections-weekly': [('weekly_amount', 'int')],  # + one 'variant_<name>' int per variant
    'build-new-hospital-beds': [('beds', 'int')],
    'build-new-icu-units': [('units', 'int')],
}


@dataclass
class Intervention:
    type: str
    date: Optional[str] = None
    values: Dict[str, object] = field(default_factory=dict)
    id: Optional[str] = None
    variant_names: Tuple[str, ...] = ('b1.1.7',)

    def parameters(self):
        if self.type not in INTERVENTION_SCHEMA:
            raise Exception('Invalid intervention type: %s' % self.type)
        params = list(INTERVENTION_SCHEMA[self.type])
        if self.type == 'import-infections-weekly':
            params += [('variant_%s' % v, 'int') for v in self.variant_names]
        return params

    def get_param_values(self):
        """{param_id: value}; int parameters are always present (None when unset), choices only
        when set (common/interventions.py:103-117)."""
        out = {}
        if not self.values:
            return out
        for pid, kind in self.parameters():
            if kind == 'int':
                out[pid] = self.values.get(pid)
            else:
                c = self.values.get(pid)
                if not c:
                    continue
                out[pid] = c
        return out

    def make_iv_tuple(self):
        return [self.type, self.date] + [self.values.get(pid) for pid, _ in self.parameters()]


def iv_tuple_to_obj(iv, variant_names=('b1.1.7',)):
    """['type', 'YYYY-MM-DD', p0, p1, ...] -> Intervention (common/interventions.py:74-101,337-339)."""
    obj = Intervention(type=iv[0], date=iv[1], variant_names=tuple(variant_names))
    rest = list(iv)[2:]
    for pid, kind in obj.parameters():
        if not rest:
            break
        val = rest.pop(0)
        if val is None:
            continue
        if kind == 'int':
            assert isinstance(val, int)
        else:
            assert isinstance(val, str)
            if pid == 'place' and val not in PLACES:
                raise Exception('Invalid choice value: %s' % val)
            if pid == 'variant' and val not in variant_names:
                raise Exception('Invalid choice value: %s' % val)
        obj.values[pid] = val
    return obj


def get_active_interventions(variables):
    """Interventions of the active scenario, in list order (common/interventions.py:342-376).
    The reference also appends Google-mobility and THL-vaccination derived interventions when its
    (git-ignored) datasets exist; those loaders are out of scope, so this is the variables list
    plus the scenario's `add_interventions`."""
    vnames = tuple(v['name'] for v in variables.get('variants', []))
    out = []
    for idx, iv in enumerate(variables['interventions']):
        obj = iv_tuple_to_obj(iv, vnames)
        obj.id = str(idx)
        out.append(obj)
    active = variables.get('active_scenario')
    if active:
        for s in variables.get('scenarios', []):
            if s['id'] == active:
                break
        else:
            raise Exception('Invalid active scenario: %s' % active)
        for iv in s.get('add_interventions', []):
            out.append(iv_tuple_to_obj(iv, vnames))
    return out
