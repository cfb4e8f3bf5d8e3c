"""Monte-Carlo ensembles: many independent simulations of the same scenario on one GPU.

Counterpart of the reference's `run_monte_carlo` (calc/simulation.py:349-385: a
`multiprocessing.Pool(8)` over seeds).  A single HUS-sized simulation keeps only a few per cent of
an MI355X busy (its day is a chain of short, latency-bound kernels), so an ensemble is run as K
engine instances side by side, every member owning its HBM state.  Two ways to issue the work:

  * batched (default): the members form an engine group (include/reina_hip.h: reina_group_*) and
    every phase of a day is ONE kernel launch covering all members (member = blockIdx.y) -- the
    launch count per day does not grow with K, so the host never becomes the limit;
  * threaded: every member has its own HIP stream and a small pool of host threads issues the
    members' days (the C ABI call releases the GIL); kept for members of differing scenarios.

Members are fully independent (BASELINE config 5: "replicas only", no collective); over several
GPUs the seeds are simply partitioned across ranks.  Results are bit-identical either way and
identical to running each seed alone (tests/test_parity_gpu.py).
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import simulation


def run_group_plan(contexts, plan, record_history=True, member_plans=None):
    """Execute `plan` (Context.make_plan) for all `contexts` as one engine group.  Returns
    history[len(contexts), days, COUNTER_WORDS] (host) or None.

    `member_plans` (one plan per context, see run_sweep): an intervention sweep -- the members' scenarios
    differ in the VALUES of their mobility limits / mask shares only, so they share the day descriptors
    of `plan` while every member gets its own contact tables at each table change."""
    from . import engine as _eng
    import os as _os, time as _time
    _T = [] if _os.environ.get('REINA_ENS_TIMING') else None   # (diagnostic: where a group run's wall time goes, tools/ens_first_run2.py)
    def _t(name):
        if _T is not None:
            _T.append((name, _time.perf_counter()))
    _t('start')
    group = _eng.EngineGroup([c.engine for c in contexts])
    a = group.alloc
    days = plan['days']
    K = len(contexts)
    hist = a.zeros(K * days * _eng.COUNTER_WORDS, np.int32) if record_history else None
    row = 4 * _eng.COUNTER_WORDS
    done = 0
    for si, (tables, arr, n) in enumerate(plan['segments']):
        if member_plans is not None:
            for c, mp in zip(contexts, member_plans):
                if mp['segments'][si][0] is not None:
                    c.engine.upload_contact_tables(*mp['segments'][si][0])
        elif tables is not None:
            group.upload_contact_tables(*tables)
        ptrs = [a.ptr(hist) + row * (m * days + done) for m in range(K)] if record_history else None
        group.run_day_array(arr, n, ptrs)
        done += n
    _t('issued')
    for m, c in enumerate(contexts):
        c.mobility_history = (member_plans[m] if member_plans is not None else plan)['mobility_history']
        c.day = plan['start_day'] + days
    out = None
    if record_history:
        out = a.to_host(hist).reshape(K, days, _eng.COUNTER_WORDS)
    _t('history on the host')
    torch = getattr(a, 'torch', None)
    if torch is not None:
        # the members' final counter blocks (the problem word among them) in one copy instead of one per member (4 ms for 128)
        finals = a.to_host(torch.stack([c.engine.tensors['counters'] for c in contexts]))
        for m, c in enumerate(contexts):
            c._raise_on_problem(np.asarray(finals[m]))
    else:
        for c in contexts:
            c._raise_on_problem(c.engine.read_counters())
    _t('final counters')
    group.close()
    _t('closed')
    if _T is not None:
        print('run_group_plan: ' + ' | '.join('%s %.1f ms' % (n, (t - _T[k][1]) * 1e3) for k, (n, t) in enumerate(_T[1:])), flush=True)
    return out


def _same_day_descriptors(p, q):
    import ctypes
    if len(p['segments']) != len(q['segments']):
        return False
    for (_, a, n), (_, b, m) in zip(p['segments'], q['segments']):
        if n != m or ctypes.string_at(a, ctypes.sizeof(a)) != ctypes.string_at(b, ctypes.sizeof(b)):
            return False
    return True


def run_sweep(variables_list, seeds, days, age_counts=None, device='cuda:0', engine_factory=None, ipc='auto'):
    """BASELINE config 5's "intervention sweep": member m runs scenario variables_list[m] with seed
    seeds[m], all as ONE engine group.  The scenarios must agree in everything that goes into the day
    descriptors (dates, testing modes, imports, vaccination, capacities) and in the dates of their
    limit-mobility / wear-masks interventions; they may differ in those interventions' values (each member
    gets its own contact tables).  Returns (history[len(seeds), days, COUNTER_WORDS], contexts)."""
    assert len(variables_list) == len(seeds)
    plans, ctxs = [], []
    for v, sd in zip(variables_list, seeds):
        planner = simulation.make_context(v, age_counts=age_counts, seed=sd, device=device, engine_factory=engine_factory, ipc=ipc)
        plans.append(planner.make_plan(days))
        del planner
        ctxs.append(simulation.make_context(v, age_counts=age_counts, seed=sd, device=device, engine_factory=engine_factory, ipc=ipc))
    for k, p in enumerate(plans[1:], 1):
        if not _same_day_descriptors(plans[0], p):
            raise ValueError('scenario %d differs from scenario 0 in more than the values of its mobility / mask '
                             'interventions: it cannot share a group' % k)
    return run_group_plan(ctxs, plans[0], member_plans=plans), ctxs


def run_ensemble_distributed(variables, seeds, days, group=None, concurrent=64, **kw):
    """BASELINE config 5: an ensemble over the GPUs of a node.  Replicas only -- rank r of the
    torch.distributed group runs seeds[r::world] as engine groups on its own GPU, no data-path
    collective; the histories are gathered on rank 0 (returned there in seed order, None elsewhere)."""
    import torch.distributed as dist
    seeds = list(seeds)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = seeds[rank::world]
    hist = run_ensemble(variables, mine, days, concurrent=concurrent, **kw) if mine else None
    parts = [None] * world if rank == 0 else None
    dist.gather_object((mine, hist), parts, dst=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if rank != 0:
        return None
    by_seed = {}
    for sds, h in parts:
        for k, sd in enumerate(sds):
            by_seed[sd] = h[k]
    return np.stack([by_seed[sd] for sd in seeds])


def run_ensemble(variables, seeds, days, age_counts=None, device='cuda:0', threads=8, concurrent=None,
                 interventions=None, batched=True, engine_factory=None, ipc='auto'):
    """Run one simulation per seed for `days` days. Returns history[len(seeds), days, COUNTER_WORDS]
    (row d = counters before day d, as Context.run). `concurrent` bounds how many members hold HBM
    state at once (default: all).  `ipc`: the initial population condition of every member; 'auto' = the
    one simulate_individuals applies for these variables (calc/simulation.py:152), None = none."""
    import torch
    seeds = list(seeds)
    concurrent = len(seeds) if concurrent is None else max(1, int(concurrent))
    if batched:
        planner = simulation.make_context(variables, age_counts=age_counts, seed=seeds[0], device=device,
                                          interventions=interventions, engine_factory=engine_factory, ipc=ipc)
        plan = planner.make_plan(days)
        del planner
        outs = []
        for start in range(0, len(seeds), concurrent):
            ctxs = [simulation.make_context(variables, age_counts=age_counts, seed=sd, device=device,
                                            interventions=interventions, engine_factory=engine_factory, ipc=ipc)
                    for sd in seeds[start:start + concurrent]]
            outs.append(run_group_plan(ctxs, plan))
            del ctxs
        return np.concatenate(outs)
    out = [None] * len(seeds)
    dev = torch.device(device)
    lock = threading.Lock()
    # the day descriptors do not depend on the seed: plan the scenario once, replay it per member
    planner = simulation.make_context(variables, age_counts=age_counts, seed=seeds[0], device=device,
                                      interventions=interventions, ipc=ipc)
    plan = planner.make_plan(days)
    del planner

    def work(k):
        torch.cuda.set_device(dev)
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            with lock:  # context construction touches shared Python state (allocator, caches)
                ctx = simulation.make_context(variables, age_counts=age_counts, seed=seeds[k], device=device,
                                              interventions=interventions, ipc=ipc)
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
