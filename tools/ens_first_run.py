"""Why is the bench's ONE timed ensemble slower than the steady state?  bench.ensemble_line's preparation (pinned block requested, a
two-member warm-up group), then three ensembles of fresh members, each split into its parts.  python tools/ens_first_run.py"""
import copy, gc, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from reina_model_amd import datasets, ensemble, simulation, engine as _eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
v = copy.deepcopy(VARIABLE_DEFAULTS); ages = datasets.get_population_for_area()
seeds, days = 128, 365
planner = simulation.make_context(v, age_counts=ages, seed=0)
plan = planner.make_plan(days)
members = [simulation.make_context(v, age_counts=ages, seed=100 + k) for k in range(seeds)]
if 'profile' in sys.argv:
    members[0].engine.profile_enable(16)
warm = torch.empty(seeds * days * _eng.COUNTER_WORDS, dtype=torch.int32, pin_memory=True); del warm
os.environ['REINA_DAY_MODE'] = 'sparse'
pre = [simulation.make_context(v, age_counts=ages, seed=90 + k) for k in range(2)]
if 'warmprofile' in sys.argv:
    pre[0].engine.profile_enable(1)
os.environ.pop('REINA_DAY_MODE')
ensemble.run_group_plan(pre, pre[0].make_plan(5)); del pre
for rep in range(3):
    if rep:
        members = [simulation.make_context(v, age_counts=ages, seed=100 + k + 1000 * rep) for k in range(seeds)]
    if len(sys.argv) > 1 and sys.argv[1] == 'nogc':
        gc.collect(); gc.disable()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    group = _eng.EngineGroup([c.engine for c in members]); a = group.alloc
    t1 = time.perf_counter()
    hist = a.zeros(seeds * days * _eng.COUNTER_WORDS, np.int32)
    t2 = time.perf_counter()
    row = 4 * _eng.COUNTER_WORDS; done = 0
    for si, (tables, arr, n) in enumerate(plan['segments']):
        if tables is not None: group.upload_contact_tables(*tables)
        ptrs = [a.ptr(hist) + row * (m * days + done) for m in range(seeds)]
        group.run_day_array(arr, n, ptrs); done += n
    t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    out = a.to_host(hist)
    t5 = time.perf_counter()
    finals = a.to_host(torch.stack([c.engine.tensors['counters'] for c in members]))
    t6 = time.perf_counter()
    group.close(); t7 = time.perf_counter()
    gc.enable()
    print('ensemble %d: group %.1f ms | hist zeros %.1f | issue %.1f | wait for the GPU %.1f | to_host %.1f | finals %.1f | close %.1f | total %.1f ms = %.4f ms per step' % (
        rep, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3, (t6-t5)*1e3, (t7-t6)*1e3, (t7-t0)*1e3, (t7-t0)*1e3/days), flush=True)
    del members, out, hist
