"""Where the wall time of a 20-day Context.run() goes at HUS (what the round driver times): host planning, library calls,
waiting for the GPU, the read-back."""
import copy, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from reina_model_amd import datasets, simulation, engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
ages = datasets.get_population_for_area()
acc = {}
for rep in range(12):
    ctx = simulation.make_context(copy.deepcopy(VARIABLE_DEFAULTS), age_counts=ages, seed=rep)
    ctx.run(5, record_history=False); ctx.synchronize(); torch.cuda.synchronize()
    a = ctx.engine.alloc
    T = {}
    t0 = time.perf_counter()
    hist = ctx._history_buffer(20); base = a.ptr(hist); row = 4 * eng.COUNTER_WORDS
    T['alloc history'] = time.perf_counter() - t0
    pending, issued, chunk = [], 0, 4
    tp = tl = 0.0
    for _ in range(20):
        t1 = time.perf_counter()
        d, changed = ctx._build_day(None)
        pending.append(d); ctx.day += 1
        tp += time.perf_counter() - t1
        if len(pending) >= chunk or _ == 19:
            t1 = time.perf_counter()
            arr = (eng.Day * len(pending))(*pending)
            ctx.engine.run_day_array(arr, len(pending), base + row * issued)
            issued += len(pending); pending = []; chunk = min(chunk * 2, 64)
            tl += time.perf_counter() - t1
    T['plan 20 days'] = tp; T['library calls (launches)'] = tl
    t1 = time.perf_counter(); torch.cuda.synchronize(); T['wait for the GPU'] = time.perf_counter() - t1
    t1 = time.perf_counter(); out = ctx._history_to_host(hist, 20); T['history + final counters to host'] = time.perf_counter() - t1
    T['total'] = time.perf_counter() - t0
    if rep >= 2:
        for k, v in T.items():
            acc.setdefault(k, []).append(v * 1e6)
for k, v in acc.items():
    print('%-30s median %7.1f us   min %7.1f' % (k, float(np.median(v)), min(v)))
