"""One-off: HIP engine vs oracle B bit for bit at BASELINE configs[2] size (50 M agents), first 130 days
(through the peak of the first wave).  Too slow for the test suite (B needs about a minute)."""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import bench, par_backend
from reina_model_amd import simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
days = int(sys.argv[1]) if len(sys.argv) > 1 else 130
v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), 50_000_000)
gpu = simulation.make_context(v, age_counts=ages, seed=0)
t0 = time.time(); hg = gpu.run(days); print('gpu %.2f s' % (time.time() - t0), flush=True)
cpu = simulation.make_context(v, age_counts=ages, seed=0, engine_factory=par_backend.par_engine_factory)
t0 = time.time(); hc = cpu.run(days); print('oracle B %.1f s' % (time.time() - t0), flush=True)
assert np.array_equal(hg, hc), 'history differs at day %d' % np.nonzero((hg != hc).any(axis=1))[0][0]
tg, tc = gpu.engine.tensors, cpu.engine.tensors
for name in ('hot', 'infector', 'n_infected', 'vacc_day', 'onset_days'):
    a = gpu.engine.alloc.to_host(tg[name]).view(np.uint32); b = np.asarray(tc[name]).view(np.uint32)
    assert np.array_equal(a, b), name
print('PARITY_50M_OK days=%d all_infected=%d' % (days, int(gpu.per_age_counters()['all_infected'].sum())))
