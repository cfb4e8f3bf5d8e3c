"""Why a 20-step timed window scatters (bench.py --steps 20): the same window ten times in one process, wall time of
ctx.run(20) + synchronize, with the garbage collector on / off, and with the day descriptors planned before the clock starts."""
import copy, gc, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from reina_model_amd import datasets, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
ages = datasets.get_population_for_area()
pre = simulation.make_context(copy.deepcopy(VARIABLE_DEFAULTS), age_counts=ages, seed=99)
pre.run(365); pre.synchronize(); del pre


def window(mode, seed):
    ctx = simulation.make_context(copy.deepcopy(VARIABLE_DEFAULTS), age_counts=ages, seed=seed)
    ctx.run(5, record_history=False)
    ctx.synchronize()
    if mode == 'nogc':
        gc.collect(); gc.disable()
    plan = ctx.make_plan(20) if mode == 'plan' else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if plan is not None:
        ctx.run_plan(plan)
    else:
        ctx.run(20, record_history=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20 * 1e6
    gc.enable()
    return dt


for mode in ('default', 'nogc', 'plan', 'default', 'nogc'):
    print('%-8s us/step: %s' % (mode, ' '.join('%.1f' % window(mode, s) for s in range(10))), flush=True)
