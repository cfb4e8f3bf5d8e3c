"""What does asking PyTorch's caching host allocator for a page-locked block cost while the device is busy / idle?  (round 6: 22 ms for the
325 MB block of an engine group's history while its kernels ran; is the 150 KB block of a 20-day window's read-back free?)"""
import copy, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from reina_model_amd import datasets, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
v = copy.deepcopy(VARIABLE_DEFAULTS); ages = datasets.get_population_for_area()
ctx = simulation.make_context(v, age_counts=ages, seed=0)
ctx.run(30)
for n_bytes in (153216, 2_700_000, 40_000_000):
    n = n_bytes // 4
    a = torch.empty(n, dtype=torch.int32, pin_memory=True); del a     # the block exists and is back in the pool
    torch.cuda.synchronize()
    idle, busy = [], []
    for rep in range(20):
        t0 = time.perf_counter(); a = torch.empty(n, dtype=torch.int32, pin_memory=True); idle.append(time.perf_counter() - t0); del a
    for rep in range(20):
        ctx.run(40, record_history=False) if False else None
        # queue ~2 ms of GPU work without waiting for it, then ask
        x = torch.empty(64 << 20, dtype=torch.int32, device='cuda'); [x.add_(1) for _ in range(20)]
        t0 = time.perf_counter(); a = torch.empty(n, dtype=torch.int32, pin_memory=True); busy.append(time.perf_counter() - t0); del a
        torch.cuda.synchronize()
    idle.sort(); busy.sort()
    print('%9d bytes: idle median %.1f us (min %.1f)   device busy median %.1f us (min %.1f, max %.1f)' % (
        n_bytes, idle[10] * 1e6, idle[0] * 1e6, busy[10] * 1e6, busy[0] * 1e6, busy[-1] * 1e6), flush=True)
