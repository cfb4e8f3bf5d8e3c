"""One engine group of K HUS members, 365 days (for rocprofv3 --kernel-trace)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reina_model_amd import datasets, ensemble, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
v = copy.deepcopy(VARIABLE_DEFAULTS); ages = datasets.get_population_for_area()
for rep in range(2):
    planner = simulation.make_context(v, age_counts=ages, seed=0)
    plan = planner.make_plan(365)
    ctxs = [simulation.make_context(v, age_counts=ages, seed=s) for s in range(K)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ensemble.run_group_plan(ctxs, plan)
    torch.cuda.synchronize()
    print('K=%d: %.3f ms/day' % (K, (time.perf_counter() - t0) / 365 * 1e3), flush=True)
    del ctxs
