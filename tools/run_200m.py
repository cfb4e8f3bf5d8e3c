"""One-off: SURVEY 8d's largest single-GPU point, 2e8 agents (default scenario scaled).  Prints the
device memory held, the 365-day rate, the busiest day's bed / ICU event count, the size-independent
properties of tests/test_parity_gpu.py::test_conservation_at_scale, and -- with a second argument --
compares the first <that many> days bit for bit with oracle B (about 0.5 s of CPU per day)."""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import bench
from reina_model_amd import simulation, engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
par_days = int(sys.argv[2]) if len(sys.argv) > 2 else 0
v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), N)
t0 = time.time(); ctx = simulation.make_context(v, age_counts=ages, seed=1); torch.cuda.synchronize()
print('context %.1f s, HBM held %.2f GB' % (time.time() - t0, torch.cuda.memory_allocated() / 1e9), flush=True)
ctx.run(5)
t0 = time.time(); hist = ctx.run(365); dt = time.time() - t0
n = int(ages.sum())
print('N=%d  365 days in %.3f s = %.4f ms/day = %.3e agent-days/s' % (n, dt, dt / 365 * 1e3, n * 365 / dt), flush=True)
A = eng.MAX_AGES
def tot(name):
    i = eng.C_NAMES.index(name)
    return hist[:, i * A:(i + 1) * A].sum(axis=1)
sc = hist[:, eng.C_NR * A:]
assert np.all(tot('susceptible') + tot('infected') + tot('recovered') + tot('dead') == n)
assert np.all(tot('all_infected') == tot('infected') + tot('recovered') + tot('dead'))
assert np.all(tot('hospitalized') == tot('in_ward') + tot('in_icu'))
assert np.all(sc[:, eng.S_DAILY_CONTACTS:eng.S_DAILY_CONTACTS + 6].sum(axis=1) == sc[:, eng.S_EXPOSED_PER_DAY])
assert np.all(sc[:, eng.S_PROBLEM] == 0), sc[:, eng.S_PROBLEM].max()
print('conservation ok, all_infected=%d dead=%d, busiest multi-range day: %d events' % (
    tot('all_infected')[-1], tot('dead')[-1], int(ctx.engine.alloc.to_host(ctx.engine.tensors['control'])[eng.L_HOSP_PEAK])), flush=True)
del ctx
if par_days:
    import par_backend
    g = simulation.make_context(v, age_counts=ages, seed=1); hg = g.run(par_days)
    c = simulation.make_context(v, age_counts=ages, seed=1, engine_factory=par_backend.par_engine_factory)
    t0 = time.time(); hc = c.run(par_days); print('oracle B %.1f s' % (time.time() - t0), flush=True)
    assert np.array_equal(hg, hc), 'history differs at day %d' % np.nonzero((hg != hc).any(axis=1))[0][0]
    for name in ('hot', 'infector', 'n_infected', 'vacc_day', 'onset_days'):
        a = g.engine.alloc.to_host(g.engine.tensors[name]).view(np.uint32); b = np.asarray(c.engine.tensors[name]).view(np.uint32)
        assert np.array_equal(a, b), name
    print('PARITY_OK N=%d days=%d' % (n, par_days))
