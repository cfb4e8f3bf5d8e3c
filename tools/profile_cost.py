"""What do timestamped k_scan dispatches cost?  365 HUS days, stride s (0 = none)."""
import copy, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from reina_model_amd import datasets, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
v = copy.deepcopy(VARIABLE_DEFAULTS); ages = datasets.get_population_for_area()
pre = simulation.make_context(v, age_counts=ages, seed=99); pre.run(200); pre.synchronize(); del pre
for rep in range(2):
    for stride in (0, 64, 8, 1, 0, 8):
        c = simulation.make_context(v, age_counts=ages, seed=rep)
        c.engine.profile_enable(stride)
        c.run(5, record_history=False); c.synchronize(); c.engine.profile_read()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c.run(365, record_history=False)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        p = c.engine.profile_read()
        print('stride %2d: %.1f us/day (host returned at %.1f); %d samples avg %.2f us' % (
            stride, (t2 - t0) / 365 * 1e6, (t1 - t0) / 365 * 1e6, p['scan_launches'],
            p['scan_ms_total'] / max(1, p['scan_launches']) * 1e3), flush=True)
        del c
