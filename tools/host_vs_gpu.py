"""Is the day loop host-bound?  Time until the library call returns (all launches issued) vs until
the stream is idle, for HUS and a large population."""
import copy, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from reina_model_amd import datasets, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
sys.path.insert(0, os.path.join(os.getcwd()))
import bench
for n in (0, 50_000_000):
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    if n:
        v, ages = bench.scaled_scenario(v, n)
    else:
        ages = datasets.get_population_for_area()
    for rep in range(2):
        ctx = simulation.make_context(v, age_counts=ages, seed=rep)
        ctx.run(150, record_history=False)
        ctx.synchronize()
        plan = ctx.make_plan(215)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.run_plan(plan, record_history=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('N=%d rep %d: issue %.1f us/day, complete %.1f us/day' % (int(ages.sum()), rep, (t1 - t0) / 215 * 1e6, (t2 - t0) / 215 * 1e6), flush=True)
        del ctx
