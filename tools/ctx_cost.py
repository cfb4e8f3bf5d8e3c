"""Where does constructing a Context (one ensemble member) spend its time?"""
import copy, cProfile, os, pstats, sys, time
sys.path.insert(0, os.getcwd())
import torch
from reina_model_amd import datasets, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
v = copy.deepcopy(VARIABLE_DEFAULTS); ages = datasets.get_population_for_area()
cs = [simulation.make_context(v, age_counts=ages, seed=s) for s in range(4)]
torch.cuda.synchronize()
t0 = time.perf_counter()
cs2 = [simulation.make_context(v, age_counts=ages, seed=10 + s) for s in range(16)]
torch.cuda.synchronize()
print('per context: %.2f ms' % ((time.perf_counter() - t0) / 16 * 1e3))
cProfile.run('cs3 = [simulation.make_context(v, age_counts=ages, seed=100 + s) for s in range(16)]; torch.cuda.synchronize()', '/tmp/ctx.prof')
pstats.Stats('/tmp/ctx.prof').sort_stats('cumtime').print_stats(18)
