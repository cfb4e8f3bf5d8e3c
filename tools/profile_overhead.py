"""What do the HIP-event-timed launches cost the day loop?  HUS x 365 d (and a 20-day window) with kernel timing off and
with one timed launch every 16th / 4th day -- the numbers behind bench.py's choice of stride."""
import copy, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from reina_model_amd import datasets, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
ages = datasets.get_population_for_area()


def run(stride, days, warm=5, reps=3):
    best = 1e9
    for r in range(reps):
        ctx = simulation.make_context(copy.deepcopy(VARIABLE_DEFAULTS), age_counts=ages, seed=r)
        if stride:
            ctx.engine.profile_enable(stride)
        ctx.run(warm, record_history=False)
        ctx.synchronize()
        if stride:
            ctx.engine.profile_read_kernels()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.run(days, record_history=True)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / days * 1e6)
        if stride:
            ctx.engine.profile_read_kernels()
    return best


for days in (365, 20):
    print('%3d days: timing off %.2f us/day | stride 16 %.2f | stride 8 %.2f | stride 4 %.2f' % (
        days, run(0, days), run(16, days), run(8, days), run(4, days)), flush=True)
