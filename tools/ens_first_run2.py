"""bench.ensemble_line's own sequence around ONE ensemble.run_group_plan call, with switches: python tools/ens_first_run2.py [noprofile] [nopre] [manual]"""
import copy, gc, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from reina_model_amd import datasets, ensemble, simulation, engine as _eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
v = copy.deepcopy(VARIABLE_DEFAULTS); ages = datasets.get_population_for_area()
seeds, days, device = 128, 365, 'cuda:0'
planner = simulation.make_context(v, age_counts=ages, seed=0, device=device)
plan = planner.make_plan(days)
members = [simulation.make_context(v, age_counts=ages, seed=100 + k, device=device) for k in range(seeds)]
if 'noprofile' not in sys.argv:
    members[0].engine.profile_enable(16)
warm = torch.empty(seeds * days * _eng.COUNTER_WORDS, dtype=torch.int32, pin_memory=True)
del warm
if 'nopre' not in sys.argv:
    os.environ['REINA_DAY_MODE'] = 'sparse'
    pre = [simulation.make_context(v, age_counts=ages, seed=90 + k, device=device) for k in range(2)]
    os.environ.pop('REINA_DAY_MODE')
    ensemble.run_group_plan(pre, pre[0].make_plan(5))
    del pre
gc.collect(); torch.cuda.synchronize(); gc.disable()
t0 = time.perf_counter()
if 'manual' in sys.argv:
    group = _eng.EngineGroup([c.engine for c in members]); a = group.alloc
    hist = a.zeros(seeds * days * _eng.COUNTER_WORDS, np.int32)
    row = 4 * _eng.COUNTER_WORDS; done = 0
    for si, (tables, arr, n) in enumerate(plan['segments']):
        if tables is not None: group.upload_contact_tables(*tables)
        ptrs = [a.ptr(hist) + row * (m * days + done) for m in range(seeds)]
        group.run_day_array(arr, n, ptrs); done += n
    t_issue = time.perf_counter()
    torch.cuda.synchronize(); t_gpu = time.perf_counter()
    out = a.to_host(hist); t_host = time.perf_counter()
    finals = a.to_host(torch.stack([c.engine.tensors['counters'] for c in members]))
    group.close()
    print('  issue %.1f | gpu %.1f | to_host %.1f' % ((t_issue - t0) * 1e3, (t_gpu - t_issue) * 1e3, (t_host - t_gpu) * 1e3))
else:
    hist = ensemble.run_group_plan(members, plan)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('%s: %.1f ms = %.4f ms per step' % (' '.join(sys.argv[1:]) or 'as bench', dt * 1e3, dt * 1e3 / days), members[0].engine.profile_read_kernels() if 'noprofile' not in sys.argv else '')
