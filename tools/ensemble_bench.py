"""Throughput of a Monte-Carlo ensemble of HUS simulations on one GPU (BASELINE config 5 shape)."""
import copy, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from reina_model_amd import datasets, ensemble, engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
v = copy.deepcopy(VARIABLE_DEFAULTS)
ages = datasets.get_population_for_area()
N = int(ages.sum()); days = 365
ensemble.run_ensemble(v, [999], 30, age_counts=ages)  # warm up
for members, threads in ((1, 1), (8, 4), (16, 8), (32, 8), (32, 16), (64, 16)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = ensemble.run_ensemble(v, list(range(members)), days, age_counts=ages, threads=threads)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    A = eng.MAX_AGES; i = eng.C_NAMES.index('all_infected')
    fin = h[:, -1, i * A:(i + 1) * A].sum(axis=1)
    print('members %3d threads %2d: %.3f s  -> %.2e agent-days/s  (all_infected mean %.0f sd %.0f)' % (
        members, threads, dt, members * N * days / dt, fin.mean(), fin.std()), flush=True)
