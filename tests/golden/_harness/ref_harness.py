"""Imports the REINA reference (read-only, from /root/reference) inside THIS container so golden
vectors can be generated from the real Cython simulator.

Harness-only. Nothing here travels to the GPU box as a dependency of tests: the tests read the
fixtures this produces (`tests/golden/*.npz|json`). The reference's files are never copied.

What is worked around (and why it does not touch the arithmetic under test):
  * missing modules `faker`, `flask_babel`, `flask_caching`, `dotenv` -> label/caching stubs in
    `stubs/`;
  * `calc.datasets.get_healthcare_districts` needs `xlrd` -> replaced with `biff8.py`;
  * `generate_mobility_ivs` / `generate_vaccination_ivs` read git-ignored datasets -> `[]`;
  * `DATASET_PATH` must hold `hosp_cases_turku.csv` (a filedep) -> temp dir copy.
"""
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('REINA_REFERENCE', '/root/reference')

_state = {}


def setup():
    if _state:
        return _state
    if not os.path.isdir(REF):
        raise RuntimeError('reference tree not found at %s' % REF)
    ds = tempfile.mkdtemp(prefix='reina_ds_')
    shutil.copy(os.path.join(REF, 'data', 'hosp_cases_turku.csv'), ds)
    os.environ['DATASET_PATH'] = ds
    build_dir = os.environ.get('REINA_PYXBLD', os.path.join(tempfile.gettempdir(), 'reina_pyxbld'))
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(HERE, 'stubs'))
    sys.path.insert(0, HERE)

    import pandas as pd
    import pyximport
    pyximport.install(build_dir=build_dir, language_level=3)

    import calc.datasets as datasets
    import biff8

    def get_healthcare_districts(*a, **k):
        cells = biff8.read_sheet_cells(os.path.join(REF, 'data', 'shp_jasenkunnat_2020.xls'),
                                       'shp_jäsenkunnat_2020_lkm')
        rows = []
        for r in sorted(set(rc[0] for rc in cells)):
            if r < 5:
                continue
            k, s, e = cells.get((r, 1)), cells.get((r, 3)), cells.get((r, 4))
            if k is None or s is None or e is None:
                continue
            rows.append((k, s, e))
        return pd.DataFrame(rows, columns=['kunta', 'sairaanhoitopiiri', 'erva-alue'])

    datasets.get_healthcare_districts = get_healthcare_districts
    datasets.generate_mobility_ivs = lambda *a, **k: []
    datasets.generate_vaccination_ivs = lambda *a, **k: []

    import common.interventions as interventions
    interventions.generate_mobility_ivs = lambda *a, **k: []
    interventions.generate_vaccination_ivs = lambda *a, **k: []

    import variables
    import calc.simulation as simulation
    from cythonsim import model

    _state.update(dict(datasets=datasets, interventions=interventions, variables=variables,
                       simulation=simulation, model=model, pd=pd))
    return _state


def default_variables():
    st = setup()
    import copy
    return copy.deepcopy(st['variables'].VARIABLE_DEFAULTS)


def hus_age_structure():
    """pd.Series age(0..100) -> count for the default area (HUS)."""
    st = setup()
    df = st['datasets'].get_population_for_area()
    return df.sum(axis=1)


def contacts_per_day():
    st = setup()
    return st['simulation'].get_contacts_per_day()


def make_ipc(spec):
    """dict -> the reference's InitialPopulationCondition (calc/datasets.py:106-134)"""
    if not spec:
        return None
    st = setup()
    import calc.datasets as ds
    return ds.InitialPopulationCondition(**spec)


def make_context(seed, variables=None, age_structure=None, interventions='default', ipc=None):
    """Build a reference `model.Context` the way calc/simulation.py:148-180 does."""
    st = setup()
    sim = st['simulation']
    v = default_variables()
    if variables:
        v.update(variables)
    if age_structure is None:
        age_structure = hus_age_structure()
    age_to_group = sim.make_age_groups()
    import numpy as np
    age_groups = list(np.unique(age_to_group))
    pop_params = dict(
        age_structure=age_structure,
        contacts_per_day=contacts_per_day(),
        initial_population_condition=make_ipc(ipc),
        age_groups=dict(labels=age_groups, age_indices=[age_groups.index(x) for x in age_to_group]),
        imported_infection_ages=v['imported_infection_ages'],
    )
    hc_params = dict(hospital_beds=v['hospital_beds'], icu_units=v['icu_units'])
    disease_params = sim.create_disease_params(v)
    ctx = st['model'].Context(
        population_params=pop_params,
        healthcare_params=hc_params,
        disease_params=disease_params,
        start_date=v['start_date'],
        random_seed=seed,
    )
    if interventions == 'default':
        ivs = v['interventions']
    else:
        ivs = interventions
    for iv in ivs:
        ctx.add_intervention(st['interventions'].iv_tuple_to_obj(iv))
    return ctx
