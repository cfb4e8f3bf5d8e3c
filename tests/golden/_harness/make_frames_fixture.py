"""Harness-only (build container; needs /root/reference): records what the reference's OWN driver
`calc.simulation.simulate_individuals` (real cythonsim behind it) hands to its callers -- the (df, adf) frames -- and
what `python -m calc.simulation` prints, as DATA: column order, dtypes, index, the adf MultiIndex layout, every value of a
40-day HUS run, the header line and the first rows of the printed table.  tests/test_host_logic.py /
tests/test_parity_gpu.py rebuild the same frames with reina_model_amd.simulation from the same per-day numbers and
compare.  Nothing of the reference's source is stored.

    python tests/golden/_harness/make_frames_fixture.py     ->  tests/golden/frames_ref.json
"""
import contextlib
import io
import json
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import numpy as np
import ref_harness as rh

st = rh.setup()
sim, variables = st['simulation'], st['variables']
DAYS, SEED = 40, 5
with variables.allow_set_variable():
    variables.set_variable('simulation_days', DAYS)
    variables.set_variable('random_seed', SEED)
    df, adf = sim.simulate_individuals(skip_cache=True)

    # the script entry of the same module: header line + one row per day + the adf print
    variables.set_variable('simulation_days', 8)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        runpy.run_module('calc.simulation', run_name='__main__')
    printed = buf.getvalue().splitlines()


def col_values(s):
    if s.dtype == object:
        return [v if isinstance(v, (int, float)) and not isinstance(v, bool) else float(v) for v in s.tolist()]
    return s.tolist()


out = dict(
    days=DAYS, seed=SEED, start=str(df.index[0].date()),
    df=dict(columns=list(df.columns), dtypes=[str(t) for t in df.dtypes], index_dtype=str(df.index.dtype),
            index_name=df.index.name, index_freq=str(df.index.freqstr), index_type=type(df.index).__name__,
            values={c: col_values(df[c]) for c in df.columns},
            element_types={c: sorted(set(type(v).__name__ for v in df[c].values)) for c in df.columns}),
    adf=dict(columns=[list(c) for c in adf.columns], column_names=list(adf.columns.names), nlevels=adf.columns.nlevels,
             dtypes=sorted(set(str(t) for t in adf.dtypes)), index_dtype=str(adf.index.dtype), index_name=adf.index.name,
             index_type=type(adf.index).__name__, values=adf.values.tolist()),
    printed=dict(header=printed[0], rows=printed[1:9]),
)
path = os.path.join(ROOT, 'tests', 'golden', 'frames_ref.json')
with open(path, 'w') as f:
    json.dump(out, f)
print('wrote', path, os.path.getsize(path), 'bytes; df', df.shape, 'adf', adf.shape)
print(printed[0])
print(printed[1])
