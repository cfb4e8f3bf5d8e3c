"""Where the wall time of a 128-member engine-group run goes (group set-up, issue, GPU, history read-back, final counters), then the same
run through ensemble.run_group_plan.  (Round 4 tried the history rows copied back behind every segment of the plan, on a stream of their
own: the kernels ran 9 ms longer beside the copies and the last segment's copy cannot overlap -- 93 ms against 96, all of it from
reading the members' final counters in one copy, which stayed.)  python tools/ens_probe.py"""
import copy, time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from reina_model_amd import datasets, ensemble, simulation, engine as _eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
v = copy.deepcopy(VARIABLE_DEFAULTS); ages = datasets.get_population_for_area()
planner = simulation.make_context(v, age_counts=ages, seed=0)
days = 365
plan = planner.make_plan(days)
for rep in range(2):
    members = [simulation.make_context(v, age_counts=ages, seed=100 + k + 1000 * rep) for k in range(128)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    group = _eng.EngineGroup([c.engine for c in members]); a = group.alloc
    torch.cuda.synchronize(); t1 = time.perf_counter()
    K = 128
    hist = a.zeros(K * days * _eng.COUNTER_WORDS, np.int32)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    row = 4 * _eng.COUNTER_WORDS; done = 0
    for si, (tables, arr, n) in enumerate(plan['segments']):
        if tables is not None: group.upload_contact_tables(*tables)
        ptrs = [a.ptr(hist) + row * (m * days + done) for m in range(K)]
        group.run_day_array(arr, n, ptrs); done += n
    t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    out = a.to_host(hist)
    t5 = time.perf_counter()
    for c in members: c._raise_on_problem(c.engine.read_counters())
    t6 = time.perf_counter()
    group.close(); t7 = time.perf_counter()
    print('rep %d: group %.1f ms | hist zeros %.1f | issue %.1f | wait for the GPU %.1f | to_host (%d MB) %.1f | counters of 128 members %.1f | close %.1f | total %.1f' % (
        rep, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, hist.numel()*4>>20, (t5-t4)*1e3, (t6-t5)*1e3, (t7-t6)*1e3, (t7-t0)*1e3), flush=True)
    del members

for rep in range(3):
    members = [simulation.make_context(v, age_counts=ages, seed=5000 + k + 1000 * rep) for k in range(128)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = ensemble.run_group_plan(members, plan)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print('run_group_plan rep %d: %.1f ms = %.4f ms per 128 member-days; history %s, infected on the last day %d' % (rep, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / days, out.shape, int(out[:, -1, :128].sum())), flush=True)
    del members, out
