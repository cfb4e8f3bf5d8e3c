"""Helpers to load the golden fixtures recorded from the reference (tests/golden/make_golden.py)."""
import copy
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_run(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    meta = json.loads(bytes(z['meta']))
    return z, meta


def variables_for(meta):
    from reina_model_amd.variables import copy_variables
    # (runs recorded under an override set of the reference -- VARIABLE_OVERRIDE_SET=turku, tests/golden/make_turku.py -- say so)
    v = copy_variables(override_set=meta.get('override_set') or 'hus')
    v.update(copy.deepcopy(meta['variables']))
    if meta.get('scenario'):
        v['active_scenario'] = meta['scenario']
    return v


def compare_day(s, z, meta, d):
    """Assert one generate_state() dict equals golden row d exactly (ints) / exactly (floats)."""
    for i, k in enumerate(meta['pop13']):
        assert np.array_equal(np.asarray(s[k]), z['pop'][d, i]), (d, k, s[k], z['pop'][d, i])
    for i, k in enumerate(meta['scalars']):
        assert s[k] == z['scalars'][d, i], (d, k, s[k], z['scalars'][d, i])
    dc = [s['daily_contacts'][p] for p in meta['places']]
    assert np.array_equal(dc, z['daily_contacts'][d]), (d, 'daily_contacts')
    ibv = [s['infected_by_variant'][n] for n in meta['variant_names']]
    assert np.array_equal(ibv, z['infected_by_variant'][d]), (d, 'infected_by_variant')
