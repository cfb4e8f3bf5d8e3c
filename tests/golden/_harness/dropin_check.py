"""Harness-only (build container; needs /root/reference): the reference's OWN driver
`calc.simulation.simulate_individuals` run with `model` swapped for `reina_model_amd.model` -- the
Python-level switch of INTEGRATION.md section 1 -- exactly as a maintainer would.  No GPU here, so the
engine behind the product's Context is the CPU oracle B (par_ ABI); the point is the PROTOCOL: the
reference's population / disease / healthcare dicts, its real Intervention objects, generate_state()
and iterate() as called by the reference, and the (df, adf) frames it assembles.

    python tests/golden/_harness/dropin_check.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import ref_harness as rh

st = rh.setup()
sim, variables = st['simulation'], st['variables']

import par_backend
from reina_model_amd import engine as eng, model as our_model
eng.hip_engine = lambda cfg, dis, device='cuda:0': par_backend.par_engine_factory(cfg, dis)   # no GPU in this container

DAYS = 60
with variables.allow_set_variable():
    variables.set_variable('simulation_days', DAYS)
    variables.set_variable('random_seed', 5)
    ref_df, ref_adf = sim.simulate_individuals(skip_cache=True)          # the real cythonsim
    sim.model = our_model                                                # <- the switch
    our_df, our_adf = sim.simulate_individuals(skip_cache=True)          # same driver, our Context

assert list(ref_df.columns) == list(our_df.columns), (list(ref_df.columns), list(our_df.columns))
assert list(ref_adf.columns) == list(our_adf.columns) and ref_df.index.equals(our_df.index)
assert (ref_df.dtypes.astype(str).values == our_df.dtypes.astype(str).values).all()
assert (ref_adf.dtypes.astype(str).values == our_adf.dtypes.astype(str).values).all() and ref_adf.index.equals(our_adf.index)
tot = int(our_df[['susceptible', 'infected', 'recovered', 'dead']].iloc[-1].sum())
assert tot == 1685983, tot
# same model, different random streams: the two runs agree loosely already on one seed
a, b = ref_df['all_infected'].iloc[-1], our_df['all_infected'].iloc[-1]
print('reference driver, cythonsim  : all_infected after %d days = %d' % (DAYS, a))
print('reference driver, our Context: all_infected after %d days = %d' % (DAYS, b))
assert 0.3 < b / max(a, 1) < 3.0
print('DROPIN_OK columns=%d rows=%d' % (len(our_df.columns), len(our_df)))
