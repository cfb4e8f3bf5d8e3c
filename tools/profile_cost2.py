import copy, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from reina_model_amd import datasets, simulation, engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
v = copy.deepcopy(VARIABLE_DEFAULTS); ages = datasets.get_population_for_area()
pre = simulation.make_context(v, age_counts=ages, seed=99); pre.run(200); pre.synchronize(); del pre
orig = eng.Engine.run_day_array
log = []
def timed(self, arr, n, hp):
    t = time.perf_counter(); orig(self, arr, n, hp); log.append((n, (time.perf_counter() - t) * 1e6))
eng.Engine.run_day_array = timed
orig_up = eng.Engine.upload_contact_tables
ups = []
def timed_up(self, *a):
    t = time.perf_counter(); orig_up(self, *a); ups.append((time.perf_counter() - t) * 1e6)
eng.Engine.upload_contact_tables = timed_up
for stride in (0, 64, 8, 8):
    c = simulation.make_context(v, age_counts=ages, seed=0)
    c.engine.profile_enable(stride)
    c.run(5, record_history=False); c.synchronize(); c.engine.profile_read()
    log.clear(); ups.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c.run(365, record_history=False)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('stride %d: total %.1f us/day host %.1f' % (stride, (t2 - t0) / 365 * 1e6, (t1 - t0) / 365 * 1e6))
    print('   chunks (days: us/day):', ' '.join('%d:%.0f' % (n, us / n) for n, us in log), flush=True)
    print('   uploads us:', ' '.join('%.0f' % u for u in ups), flush=True)
    del c
