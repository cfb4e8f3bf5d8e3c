#!/usr/bin/env python3
"""bench.py -- agent-days/s of the MI355X-native day step, with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--agents A] [--no-cpu] [--no-large]

A "step" is one simulated day (Context.iterate of the reference, cythonsim/main.pyx:2011-2018)
over the whole population.  At N=1 the workload is BASELINE.json configs[1]: the HUS population
(1 685 983 agents, real age structure + FI contact matrix), default scenario, K=365 days.
W warm-up days are simulated first (untimed), then exactly K days are timed between
barrier + torch.cuda.synchronize() pairs; rank 0 prints ONE JSON line.

Inputs are resident in HBM when the timed region starts (state tensors, tables); per-day host
work inside the region is the intervention schedule -> day descriptors (+ a 100 KB table upload
on the 11 days the mobility factors change), exactly what the reference's iterate() does on host.

Extra objects on the line:
  roofline     dominant kernel k_scan: algorithmic bytes per launch (4 B hot word read per agent +
               4 B written back per infected agent) / mean launch duration
               from HIP events recorded on the launch stream inside the timed region (every --time-every-th day), vs 8 TB/s.
  cpu_baseline the sequential C oracle (oracle/reina_seq.c, bit-exact vs the reference cythonsim)
               timed on one host core on a bounded sample (first days of the same workload).
  ensemble     BASELINE config 5 shape: 32 seeds of the HUS scenario as one engine group.
  large        the same measurement on BASELINE configs[2] (synthetic 50 M agents, HUS age shape,
               beds/ICU/imports scaled) -- the HBM-resident regime the roofline is meant for.
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def scaled_scenario(variables, total_agents):
    """BASELINE configs[2]/[3]: HUS age shape scaled to `total_agents`; beds, ICU
    units and every import amount scaled by the same factor so prevalence stays comparable."""
    from reina_model_amd import datasets
    base = datasets.get_population_for_area()
    S = total_agents / float(base.sum())
    v = copy.deepcopy(variables)
    v['hospital_beds'] = int(round(v['hospital_beds'] * S))
    v['icu_units'] = int(round(v['icu_units'] * S))
    ivs = []
    for iv in v['interventions']:
        iv = list(iv)
        if iv[0] in ('import-infections', 'import-infections-weekly'):
            iv[2] = int(round(iv[2] * S))
        ivs.append(iv)
    v['interventions'] = ivs
    return v, datasets.scaled_population(total_agents)


_COMM = []


def _shared_comm(sharding):
    """one communicator (torch.distributed group + our RCCL communicator) for every run of this process"""
    if not _COMM:
        _COMM.append(sharding.TorchComm())
    return _COMM[0]


def run_gpu(variables, ages, seed, steps, warmup, device, dist=None, preheat=0, stride=1, preheat_runs=2):
    import numpy as np
    import torch
    from reina_model_amd import engine as eng
    from reina_model_amd import sharding, simulation
    comm = _shared_comm(sharding) if dist is not None else None
    for rep in range(preheat_runs if preheat else 0):
        # throw-away runs of the same workload (untimed, separate state): bring the GPU out of its
        # idle power state and pay one-time runtime costs before the measured simulation exists.
        # (event pools, allocator pools, first timestamped dispatches), profiled like the timed one.
        pre = simulation.make_context(variables, age_counts=ages, seed=seed + 1000003 + rep, device=device, comm=comm)
        pre.engine.profile_enable(stride)
        pre.run(preheat, record_history=True)
        pre.synchronize()
        pre.engine.profile_read()
        del pre
    ctx = simulation.make_context(variables, age_counts=ages, seed=seed, device=device, comm=comm)
    # the event-timed launch path is switched on BEFORE the warm-up so its one-time costs (event
    # pool, first timestamped dispatches) are not billed to the timed region
    ctx.engine.profile_enable(stride)
    if warmup:
        ctx.run(warmup, record_history=False)
    ctx.synchronize()
    ctx.engine.profile_read()  # discard the warm-up launches
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hist = ctx.run(steps, record_history=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    prof = ctx.engine.profile_read()
    ctx.engine.profile_enable(False)
    if comm is not None:
        hist = hist // 1  # already the global (summed) history on every rank
    A = eng.MAX_AGES
    inf = hist[:, eng.C_NAMES.index('infected') * A:(eng.C_NAMES.index('infected') + 1) * A].sum(axis=1)
    sc = hist[:, eng.C_NR * A:]
    timed = (np.arange(steps) + warmup) % stride == 0   # the days whose scan launch carried timestamps
    stats = dict(
        mean_infected=float(inf[timed].mean() if timed.any() else inf.mean()),
        mean_infected_all_days=float(inf.mean()),
        contacts=float(sc[:, eng.S_EXPOSED_PER_DAY].sum()),
        new_infections=float(hist[:, eng.C_NAMES.index('new_infections') * A:(eng.C_NAMES.index('new_infections') + 1) * A].sum()),
        final_all_infected=int(hist[-1, eng.C_NAMES.index('all_infected') * A:(eng.C_NAMES.index('all_infected') + 1) * A].sum()),
    )
    if comm is not None:
        stats['mean_infected'] /= comm.world  # per-shard share for the per-launch byte count
        stats['mean_infected_all_days'] /= comm.world
    return t1 - t0, prof, stats, ctx.total_people


def roofline_obj(n_agents, steps, prof, stats, stride=1, ms_per_step=None):
    # algorithmic bytes of one k_scan launch: every agent's 4-byte hot word read once, the hot
    # word of every infected agent written back (the 4*N + 4*N_inf terms)
    bytes_per_launch = 4.0 * n_agents + 4.0 * stats['mean_infected']
    launches = max(1, int(prof['scan_launches']))
    ms = prof['scan_ms_total'] / launches
    achieved = bytes_per_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    day_bytes = (4.0 * n_agents * steps + 4.0 * stats['mean_infected_all_days'] * steps
                 + 4.0 * stats['contacts'] + 12.0 * stats['new_infections']) / steps
    extra = {}
    if ms_per_step:
        # the whole day against the same roofline (sum of B_alg / wall): small populations are
        # latency-bound, this is the honest figure next to the streaming kernel's
        extra = dict(day_achieved=round(day_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                     day_frac=round(day_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5))
    return dict(bound='hbm', kernel='k_scan', achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit='GB/s',
                frac=round(achieved / HBM_PEAK_GBS, 5), traffic=None,
                bytes_per_launch=bytes_per_launch, avg_launch_ms=round(ms, 6), launches=launches,
                launches_note='timestamped launches: every %d-th day of the timed region' % stride,
                day_algorithmic_bytes=round(day_bytes, 1), **extra)


def cpu_baseline(variables, ages, seed, days):
    from oracle import seq_oracle
    import numpy as np
    ctx = seq_oracle.make_context(variables, ages, seed)
    t0 = time.perf_counter()
    for _ in range(days):
        ctx.iterate()
    dt = time.perf_counter() - t0
    n = int(np.asarray(ages).sum())
    return dict(value=round(n * days / dt, 1), unit='agent-days/s', cores=1, kind='port',
                sample='sequential C restatement of cythonsim (bit-exact vs reference goldens), '
                       'HUS %d agents, first %d days of the default scenario, 1 thread, %.1f s' % (n, days, dt))


_CPU_WORKER = r"""
import copy, sys, time
sys.path.insert(0, %(root)r)
from oracle import seq_oracle
from reina_model_amd import datasets
from reina_model_amd.variables import VARIABLE_DEFAULTS
ages = datasets.get_population_for_area()
ctx = seq_oracle.make_context(copy.deepcopy(VARIABLE_DEFAULTS), ages, %(seed)d)
while time.time() < %(start)f:
    time.sleep(0.01)
t0 = time.time()
for _ in range(%(days)d):
    ctx.iterate()
print(t0, time.time())
"""


_CPU_HELPER = r"""
import json, os, subprocess, sys, time
root, days, procs, go = %(root)r, %(days)d, %(procs)d, %(go)r
worker = %(worker)r
while not os.path.exists(go):          # the GPU measurements come first: all-core load slows the host
    time.sleep(0.05)
    if os.getppid() == 1:
        sys.exit(0)
start = time.time() + 10.0
ps = [subprocess.Popen([sys.executable, '-c', worker %% dict(root=root, seed=1000 + k, start=start, days=days)],
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for k in range(procs)]
spans = []
for p in ps:
    out, _ = p.communicate(timeout=600)
    if p.returncode == 0:
        a, b = out.decode().split()[-2:]
        spans.append((float(a), float(b)))
print(json.dumps(spans))
"""


class CpuAllCores:
    """The reference's Monte-Carlo shape (calc/simulation.py:376: a pool of processes, one
    simulation each) with the sequential C restatement: one HUS simulation per host core, all
    started together.  A helper process is spawned BEFORE this process touches the GPU (no exec
    after GPU initialisation); it starts its workers only when told to, after the GPU measurements,
    so that the all-core load cannot disturb them."""

    def __init__(self, days, max_procs=64):
        import subprocess
        import tempfile
        self.days = days
        self.procs = min(os.cpu_count() or 1, max_procs)
        self.max_procs = max_procs
        self.go = os.path.join(tempfile.gettempdir(), 'reina_bench_go_%d' % os.getpid())
        code = _CPU_HELPER % dict(root=ROOT, days=days, procs=self.procs, go=self.go, worker=_CPU_WORKER)
        self.helper = subprocess.Popen([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)

    def run(self):
        open(self.go, 'w').close()
        try:
            out, _ = self.helper.communicate(timeout=900)
            spans = json.loads(out.decode().strip().splitlines()[-1])
        except Exception:
            return None
        finally:
            try:
                os.unlink(self.go)
            except OSError:
                pass
        if not spans:
            return None
        wall = max(b for _, b in spans) - min(a for a, _ in spans)
        n = 1685983
        return dict(value=round(n * self.days * len(spans) / wall, 1), unit='agent-days/s', cores=len(spans), kind='port',
                    sample='%d concurrent sequential simulations (one per core, cores capped at %d), HUS %d agents, '
                           'first %d days each, %.1f s wall' % (len(spans), self.max_procs, n, self.days, wall))


def ensemble_line(seeds, days, device):
    """BASELINE config 5 shape: a Monte-Carlo ensemble of HUS simulations on one GPU, stepped as an
    engine group (one launch per phase for all members)."""
    import torch
    from reina_model_amd import datasets, ensemble, simulation
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    ages = datasets.get_population_for_area()
    planner = simulation.make_context(v, age_counts=ages, seed=0, device=device)
    plan = planner.make_plan(days)
    members = [simulation.make_context(v, age_counts=ages, seed=100 + k, device=device) for k in range(seeds)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ensemble.run_group_plan(members, plan)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = int(ages.sum())
    return dict(workload='%d seeds x HUS %d agents x %d days, one engine group' % (seeds, n, days),
                value=round(seeds * n * days / dt, 1), unit='agent-days/s', ms_per_step=round(dt * 1000 / days, 6),
                members=seeds)


def main():
    t_process = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=365)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--agents', type=int, default=0, help='synthetic population per GPU (0 = HUS)')
    ap.add_argument('--large-agents', type=int, default=50_000_000)
    ap.add_argument('--xlarge-agents', type=int, default=200_000_000,
                    help='SURVEY 8d: a population well past the 256 MB Infinity Cache (0 = skip)')
    ap.add_argument('--cpu-days', type=int, default=120)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-large', action='store_true')
    ap.add_argument('--no-ensemble', action='store_true')
    ap.add_argument('--ensemble-seeds', type=int, default=32)
    ap.add_argument('--cpu-all-cores-days', type=int, default=120)
    ap.add_argument('--cpu-max-procs', type=int, default=64, help='concurrent CPU simulations of the all-cores baseline')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--time-every', type=int, default=8,
                    help='k_scan launches carry HIP event timestamps on every k-th day of the timed region')
    ap.add_argument('--preheat-days', type=int, default=365,
                    help='days of a throw-away simulation run before the measured one (GPU clocks, one-time costs)')
    a = ap.parse_args()
    if a.steps < 8 * a.time_every:      # short runs: at least ~8 timestamped launches, every day if need be
        a.time_every = max(1, a.steps // 8)

    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    cpu_all = None
    if world_env == 1 and not a.no_cpu and not a.agents and a.cpu_all_cores_days > 0:
        cpu_all = CpuAllCores(a.cpu_all_cores_days, a.cpu_max_procs)   # helper spawned before anything initialises the GPU; runs last

    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # (test hooks: REINA_BENCH_BACKEND=gloo REINA_BENCH_ONE_GPU=1 lets two ranks share the single GPU
        # of a test box to exercise this path; the driver's runs use nccl, one GPU per rank)
        if os.environ.get('REINA_BENCH_ONE_GPU'):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        backend = os.environ.get('REINA_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    device = 'cuda:%d' % local_rank

    from reina_model_amd import datasets
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    per_gpu = a.agents if a.agents else 1685983
    if a.agents or world > 1:
        # weak scaling: the GLOBAL population is per_gpu x world agents, sharded over the ranks
        # (BASELINE configs[3] shape: HUS age structure, beds / ICU / imports scaled with it)
        v, ages = scaled_scenario(v, per_gpu * world)
        workload = 'synthetic %d agents (%d per GPU; HUS age structure + FI contact matrix, beds/ICU/imports scaled), default scenario, %d days' % (per_gpu * world, per_gpu, a.steps)
    else:
        ages = datasets.get_population_for_area()
        workload = 'HUS 1685983 agents, default scenario (variables.py:227-435), %d days' % a.steps

    dt, prof, stats, n_local = run_gpu(v, ages, a.seed, a.steps, a.warmup, device, dist, preheat=a.preheat_days, stride=a.time_every)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    import numpy as _np
    total_agents = int(_np.asarray(ages).sum())
    n_agents = total_agents // world   # agents one k_scan launch streams on this rank
    value = total_agents * a.steps / dt

    large_sharded = None
    if world > 1 and not a.no_large and not a.agents:
        # BASELINE configs[3] shape on the same ranks: 50 M agents per GPU (4 x 10^8 on 8), sharded
        vl, agesl = scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), a.large_agents * world)
        dtl, profl, statsl, nl = run_gpu(vl, agesl, a.seed, a.steps, a.warmup, device, dist,
                                         preheat=min(a.preheat_days, 120), stride=a.time_every, preheat_runs=1)
        t = torch.tensor([dtl], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dtl = float(t.item())
        tot_l = int(_np.asarray(agesl).sum())
        large_sharded = {
            'workload': 'synthetic %d agents (%d per GPU, BASELINE configs[3] shape), default scenario scaled, %d days' % (
                tot_l, tot_l // world, a.steps),
            'value': round(tot_l * a.steps / dtl, 1), 'unit': 'agent-days/s', 'ms_per_step': round(dtl * 1000 / a.steps, 6),
            'roofline': roofline_obj(tot_l // world, a.steps, profl, statsl, a.time_every, dtl * 1000 / a.steps),
            'final_all_infected': statsl['final_all_infected'],
        }

    out = None
    if rank == 0:
        out = {
            'metric': 'agent-days/sec', 'value': round(value, 1), 'unit': 'agent-days/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt * 1000 / a.steps, 6), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'u32', 'data': 'synthetic',
            'config': {'workload': workload, 'agents_total': total_agents,
                       'parallelism': 'single GPU' if world == 1 else 'agents sharded x%d, one 8 KB RCCL all-reduce of infection pressure per day' % world,
                       'final_all_infected': stats['final_all_infected']},
            'roofline': roofline_obj(n_agents, a.steps, prof, stats, a.time_every, dt * 1000 / a.steps),
        }
        traffic_file = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(traffic_file):
            try:
                tj = json.load(open(traffic_file))
                key = 'hus' if not a.agents else str(a.agents)
                if key in tj:
                    out['roofline']['traffic'] = tj[key]
            except Exception:
                pass
        def extra(key, fn):
            # the additional workloads must not take the headline line down with them
            try:
                out[key] = fn()
            except Exception as e:   # noqa: BLE001 -- reported in the line itself
                out[key] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}

        def sized_line(n_agents_cfg, label, preheat, preheat_runs):
            vl, agesl = scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), n_agents_cfg)
            dtl, profl, statsl, nl = run_gpu(vl, agesl, a.seed, a.steps, a.warmup, device, preheat=preheat,
                                             stride=a.time_every, preheat_runs=preheat_runs)
            line = {
                'workload': 'synthetic %d agents (%s), default scenario scaled, %d days' % (nl, label, a.steps),
                'value': round(nl * a.steps / dtl, 1), 'unit': 'agent-days/s',
                'ms_per_step': round(dtl * 1000 / a.steps, 6),
                'roofline': roofline_obj(nl, a.steps, profl, statsl, a.time_every, dtl * 1000 / a.steps),
                'final_all_infected': statsl['final_all_infected'],
            }
            try:   # PMC traffic of the same workload (profiles/traffic.json, collected in separate --pmc passes)
                line['roofline']['traffic'] = json.load(open(traffic_file)).get(str(n_agents_cfg))
            except Exception:
                pass
            return line

        if not a.no_large and world == 1 and not a.agents:
            extra('large', lambda: sized_line(a.large_agents, 'BASELINE configs[2]', min(a.preheat_days, 120), 2))
            if a.xlarge_agents:
                # SURVEY 8d's second point: a hot array (0.8 GB) that no cache level holds
                extra('xlarge', lambda: sized_line(a.xlarge_agents, 'SURVEY 8d: HBM-resident regime', min(a.preheat_days, 30), 1))
        if large_sharded is not None:
            out['large'] = large_sharded
        if not a.no_ensemble and world == 1 and not a.agents:
            extra('ensemble', lambda: ensemble_line(a.ensemble_seeds, a.steps, device))
        if not a.no_cpu and world == 1:   # the CPU baseline is an N=1 figure
            hus = datasets.get_population_for_area()
            out['cpu_baseline'] = cpu_baseline(copy.deepcopy(VARIABLE_DEFAULTS), hus, a.seed, a.cpu_days)
            out['cpu_baseline']['cores_available'] = os.cpu_count()
            try:
                with open('/proc/cpuinfo') as f:
                    out['cpu_baseline']['cpu_model'] = next(l.split(':', 1)[1].strip() for l in f if l.startswith('model name'))
            except Exception:
                pass
            if cpu_all is not None:
                res = cpu_all.run()
                if res is not None:
                    out['cpu_baseline']['all_cores'] = res
        out['timing'] = {
            'timed_region_s': round(dt, 6), 'process_wall_s': round(time.perf_counter() - t_process, 2),
            'note': 'the process also builds contexts, runs untimed preheat simulations, the extra 50 M / ensemble '
                    'configurations and the CPU baselines; value = agents x steps / timed_region_s'}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
