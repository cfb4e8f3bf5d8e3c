"""Where the wall time of Context.run goes on the host (HUS, 365 days)."""
import copy, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch
from reina_model_amd import datasets, simulation, engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
v = copy.deepcopy(VARIABLE_DEFAULTS)
ctx = simulation.make_context(v, age_counts=datasets.get_population_for_area(), seed=0)
ctx.run(5, record_history=False); ctx.synchronize()
tb = tr = tu = 0.0
eng_run = ctx.engine.run_days
def timed_run(days):
    global tr
    t = time.perf_counter(); eng_run(days); tr += time.perf_counter() - t
ctx.engine.run_days = timed_run
orig_build = ctx._build_day
def timed_build(ptr=None):
    global tb
    t = time.perf_counter(); r = orig_build(ptr); tb += time.perf_counter() - t
    return r
ctx._build_day = timed_build
orig_up = ctx._upload_tables
def timed_up():
    global tu
    t = time.perf_counter(); orig_up(); tu += time.perf_counter() - t
ctx._upload_tables = timed_up
for prof in (False, True):
    tb = tr = tu = 0.0
    ctx.engine.profile_enable(prof)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = ctx.run(365)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print('profile=%s total %.1f ms: build %.1f  run_days(host) %.1f  upload %.1f  rest(sync, history) %.1f' % (
        prof, (t1 - t0) * 1e3, tb * 1e3, tr * 1e3, tu * 1e3, (t1 - t0 - tb - tr - tu) * 1e3))
    if prof: ctx.engine.profile_read()
