import copy, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch
from reina_model_amd import datasets, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
for prof in (True, False, True, False):
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    ctx = simulation.make_context(v, age_counts=datasets.get_population_for_area(), seed=0)
    ctx.run(5, record_history=False); ctx.synchronize()
    ctx.engine.profile_enable(prof)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = ctx.run(365)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print('profile=%s total %.1f ms' % (prof, (t1 - t0) * 1e3))
    if prof: print(ctx.engine.profile_read())
