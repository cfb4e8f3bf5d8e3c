"""Monte-Carlo ensembles: many independent simulations of the same scenario on one GPU.

Counterpart of the reference's `run_monte_carlo` (calc/simulation.py:349-385: a
`multiprocessing.Pool(8)` over seeds).  A single HUS-sized simulation keeps only a few per cent of
an MI355X busy (its day is a chain of short, latency-bound kernels), so an ensemble is run as K
engine instances side by side: every member owns its HBM state and its own HIP stream, and a
small pool of host threads issues the members' days (the C ABI call releases the GIL), letting the
GPU overlap the members' kernels.  Members are fully independent (BASELINE config 5: "replicas
only", no collective); over several GPUs the seeds are simply partitioned across ranks.
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import simulation


def run_ensemble(variables, seeds, days, age_counts=None, device='cuda:0', threads=8, concurrent=None,
                 interventions=None):
    """Run one simulation per seed for `days` days. Returns history[len(seeds), days, COUNTER_WORDS]
    (row d = counters before day d, as Context.run). `concurrent` bounds how many members hold HBM
    state at once (default: all)."""
    import torch
    seeds = list(seeds)
    concurrent = len(seeds) if concurrent is None else max(1, int(concurrent))
    out = [None] * len(seeds)
    dev = torch.device(device)
    lock = threading.Lock()
    # the day descriptors do not depend on the seed: plan the scenario once, replay it per member
    planner = simulation.make_context(variables, age_counts=age_counts, seed=seeds[0], device=device,
                                      interventions=interventions)
    plan = planner.make_plan(days)
    del planner

    def work(k):
        torch.cuda.set_device(dev)
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            with lock:  # context construction touches shared Python state (allocator, caches)
                ctx = simulation.make_context(variables, age_counts=age_counts, seed=seeds[k], device=device,
                                              interventions=interventions)
            hist = ctx.run_plan(plan)
            stream.synchronize()
        out[k] = hist
        del ctx
        return k

    for start in range(0, len(seeds), concurrent):
        batch = range(start, min(len(seeds), start + concurrent))
        with ThreadPoolExecutor(max_workers=min(threads, len(batch))) as pool:
            list(pool.map(work, batch))
    return np.stack(out)
