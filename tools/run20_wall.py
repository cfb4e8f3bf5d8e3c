"""Wall time of Context.run(20) after 5 days at HUS -- the window the round driver times -- and what the contact-table change
inside it (2020-03-12, day 23) costs the host: python tools/run20_wall.py"""
import copy, os, sys, time, gc
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from reina_model_amd import datasets, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
ages = datasets.get_population_for_area()
for warm in (5, 30):
    ts, up, pk = [], [], []
    for rep in range(14):
        ctx = simulation.make_context(copy.deepcopy(VARIABLE_DEFAULTS), age_counts=ages, seed=rep)
        ctx.run(warm, record_history=False); ctx.synchronize(); torch.cuda.synchronize()
        t_up = [0.0, 0.0]
        orig, origp = ctx.engine.upload_contact_tables, ctx._packed_tables
        def timed(*a, _o=orig):
            t = time.perf_counter(); r = _o(*a); t_up[0] += time.perf_counter() - t; return r
        def timedp(*a, _o=origp):
            t = time.perf_counter(); r = _o(*a); t_up[1] += time.perf_counter() - t; return r
        ctx.engine.upload_contact_tables = timed; ctx._packed_tables = timedp
        gc.collect(); gc.disable()
        t0 = time.perf_counter()
        h = ctx.run(20, record_history=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        gc.enable()
        if rep >= 2:
            ts.append((t1 - t0) * 1e6); up.append(t_up[0] * 1e6); pk.append(t_up[1] * 1e6)
    print('after %2d days: run(20) median %.1f us (min %.1f) = %.2f us/step; of it table rebuild (host) %.1f us + upload call %.1f us' % (
        warm, np.median(ts), min(ts), np.median(ts) / 20, np.median(pk), np.median(up)))
