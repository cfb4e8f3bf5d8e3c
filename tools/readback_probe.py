import copy, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from reina_model_amd import datasets, simulation, engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
ages = datasets.get_population_for_area()
ctx = simulation.make_context(copy.deepcopy(VARIABLE_DEFAULTS), age_counts=ages, seed=1)
ctx.run(25, record_history=True); torch.cuda.synchronize()
e = ctx.engine; a = e.alloc
hist = ctx._history_buffer(20)
ctx.run(20, record_history=True); torch.cuda.synchronize()
rows = 20; n = (rows + 1) * eng.COUNTER_WORDS
acc = {}
def T(k, f):
    t = time.perf_counter(); r = f(); acc.setdefault(k, []).append((time.perf_counter() - t) * 1e6); return r
for rep in range(30):
    pin = T('torch.empty pinned', lambda: torch.empty(n, dtype=torch.int32, pin_memory=True))
    out = T('numpy view + ptr', lambda: (pin.numpy(), pin.data_ptr()))
    T('library call (export + sync)', lambda: e._check(e.f['read_history'](e._h, a.ptr(hist), rows, out[1], a.stream()), 'read_history'))
    T('raise_on_problem + slice', lambda: (ctx._raise_on_problem(out[0].reshape(rows + 1, eng.COUNTER_WORDS)[rows]), out[0].reshape(rows + 1, eng.COUNTER_WORDS)[:rows]))
    T('whole read_history', lambda: e.read_history(hist, rows))
    T('torch.cuda.synchronize (idle)', lambda: torch.cuda.synchronize())
    del pin, out
for k, v in acc.items():
    print('%-34s median %7.1f us  min %7.1f' % (k, float(np.median(v[5:])), min(v[5:])))
