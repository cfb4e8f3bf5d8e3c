import copy, sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from reina_model_amd import simulation, datasets
from reina_model_amd.variables import VARIABLE_DEFAULTS
v = copy.deepcopy(VARIABLE_DEFAULTS)
for rep in range(3):
    ctx = simulation.make_context(v, age_counts=datasets.get_population_for_area(), seed=rep)
    for _ in range(20): ctx.generate_state(); ctx.iterate()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): s = ctx.generate_state(); ctx.iterate()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('eager generate_state + iterate: %.1f us/day' % (dt / 200 * 1e6), flush=True)
