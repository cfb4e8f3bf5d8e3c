"""Throughput of a Monte-Carlo ensemble of HUS simulations on one GPU (BASELINE config 5 shape):
batched engine groups (one launch per phase for all members) vs one stream + host thread per member."""
import copy, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from reina_model_amd import datasets, ensemble, engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
v = copy.deepcopy(VARIABLE_DEFAULTS)
ages = datasets.get_population_for_area()
N = int(ages.sum()); days = 365
ensemble.run_ensemble(v, [999, 998], 60, age_counts=ages)  # warm up
# step time alone (members built beforehand): the launch-count argument for groups
from reina_model_amd import simulation
for members in (1, 8, 32, 64):
    planner = simulation.make_context(v, age_counts=ages, seed=0)
    plan = planner.make_plan(days)
    ctxs = [simulation.make_context(v, age_counts=ages, seed=s) for s in range(members)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ensemble.run_group_plan(ctxs, plan)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('group of %3d, stepping only: %.3f s = %.3f ms/day -> %.2e agent-days/s' % (
        members, dt, dt * 1e3 / days, members * N * days / dt), flush=True)
    del ctxs
rows = []
for members, mode in ((1, 'batched'), (8, 'batched'), (16, 'batched'), (32, 'batched'), (64, 'batched'),
                      (128, 'batched'), (16, 'threads'), (64, 'threads')):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = ensemble.run_ensemble(v, list(range(members)), days, age_counts=ages, threads=16, batched=(mode == 'batched'))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    A = eng.MAX_AGES; i = eng.C_NAMES.index('all_infected')
    fin = h[:, -1, i * A:(i + 1) * A].sum(axis=1)
    rows.append(dict(members=members, mode=mode, seconds=round(dt, 3), agent_days_per_s=members * N * days / dt,
                     all_infected_mean=float(fin.mean()), all_infected_sd=float(fin.std())))
    print('members %3d %-8s: %.3f s  -> %.2e agent-days/s  (all_infected mean %.0f sd %.0f)' % (
        members, mode, dt, members * N * days / dt, fin.mean(), fin.std()), flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], 'w'), indent=1)
