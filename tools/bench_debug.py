import copy, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from reina_model_amd import datasets, engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
ups = []
orig_up = eng.Engine.upload_contact_tables
def timed_up(self, *a):
    t = time.perf_counter(); orig_up(self, *a); ups.append(round((time.perf_counter() - t) * 1e6))
eng.Engine.upload_contact_tables = timed_up
log = []
orig = eng.Engine.run_day_array
def timed(self, arr, n, hp):
    t = time.perf_counter(); orig(self, arr, n, hp); log.append((n, round((time.perf_counter() - t) * 1e6 / n)))
eng.Engine.run_day_array = timed
v = copy.deepcopy(VARIABLE_DEFAULTS); ages = datasets.get_population_for_area()
for preheat_runs in (2, 4):
    ups.clear(); log.clear()
    dt, prof, stats, n = bench.run_gpu(v, ages, 0, 365, 5, 'cuda:0', None, preheat=365, stride=8, preheat_runs=preheat_runs)
    print('preheat_runs %d: %.1f us/day' % (preheat_runs, dt / 365 * 1e6))
    print('  uploads', ups[-12:]); print('  chunks', log[-17:], flush=True)
