#!/usr/bin/env python3
"""bench.py -- agent-days/s of the MI355X-native day step, with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--agents A] [--no-cpu] [--no-sizes] [--no-ensemble]

A "step" is one simulated day (Context.iterate of the reference, cythonsim/main.pyx:2011-2018)
over the whole population.  At N=1 the workload is BASELINE.json configs[1]: the HUS population
(1 685 983 agents, real age structure + FI contact matrix), default scenario.  W warm-up days are
simulated first (untimed), then exactly K days are timed between barrier + torch.cuda.synchronize()
pairs; rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU and starts
the N ranks itself (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 ...` as a child process), relays the child's output and exits with its code.  Under
torchrun (WORLD_SIZE set) it is one rank: the population is `--agents` (default 1 685 983) agents PER
GPU, age-stratified shards, one 8 KB RCCL all-reduce of infection pressure per day ("scaling": "weak").

Inputs are resident in HBM when the timed region starts (state tensors, tables); per-day host
work inside the region is the intervention schedule -> day descriptors (+ a 100 KB table upload
on the 11 days the mobility factors change), exactly what the reference's iterate() does on host.

Objects on the line:
  roofline       THE DAY against HBM: achieved = sum over the timed days of
                 B_alg(day) = 4 N + 4 N_inf + 4 C + 12 I_new bytes / wall time of the timed region; frac =
                 achieved / 8 TB/s.  `kernels`: every kernel of the day with its mean launch duration from
                 HIP events on the launch stream inside the timed region (one kind of kernel per profiled
                 day, the kinds taking turns), its share of the day, and -- where the kernel has
                 algorithmic bytes of its own -- bytes per launch and achieved GB/s.  `dominant_kernel`
                 names the streaming kernel k_day with its per-launch figure.  `traffic`: HBM bytes per
                 day from PMC counters (profiles/traffic.json: separate rocprofv3 --pmc passes, FETCH_SIZE
                 doubled per MI355X_MICROARCH.md); only reported when that file was collected on the very
                 library binary being timed (sha256 match), else null.
  full_scenario  the same figures over the whole 365-day default scenario (epidemic peak included) for HUS,
                 5 x 10^7 (configs[2]), 10^8 (BASELINE's "100 M agents") and 2 x 10^8 agents (SURVEY 8d's
                 HBM-resident point) -- a short --steps window right after the start is all quiet days.
  ensemble       BASELINE config 5 per-GPU batch: 128 seeds of the HUS scenario as one engine group.
  cpu_baseline   the sequential C restatement of cythonsim (oracle/reina_seq.c, bit-exact vs the reference)
                 on one host core over THE SAME day window (W untimed, K timed days), repeated over seeds
                 until about 10 s of timed work; `all_cores`: one simulation per core, all started together,
                 over the same window (--cpu-all-cores-days n: the first n days instead, labelled).
"""
import argparse
import copy
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
HUS_AGENTS = 1685983
DAY_KERNELS = ('k_small_day', 'k_open', 'k_test_trace1', 'k_vaccinate', 'k_day', 'k_hospital', 'k_hosp_sort', 'k_hosp_walk', 'k_remote', 'k_hosp_install', 'k_xchg', 'collective')
# the chip's RANDOM-ACCESS rates, G accesses/s (tools/ubench_random.hip, profiles/r05_evidence/ubench_random.txt: independent of the requests
# in flight per lane and of the waves per CU -- throughput limits): what bounds k_hosp_install, whose bytes are nothing
RANDOM_RATES = dict(load=50.0, store=23.0, atomic=18.0, atomic_cached=26.0)
DAY_IMAGE_BYTES = 101 * 1024   # k_day's LDS image of the contact tables, staged once per workgroup (k_contacts.inc: DayShared + rows)


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


_COMM = {}


def _shared_comm(sharding, attribution='mirror'):
    """one communicator (torch.distributed group + our RCCL communicator) per attribution mode for every run of this process"""
    if attribution not in _COMM:
        _COMM[attribution] = sharding.TorchComm(attribution=attribution)
    return _COMM[attribution]


TIMED_LAUNCH_COST_US = 5.5   # wall time a timestamped dispatch adds to a short run (tools/window_probe.py: 20 HUS days with 15 / 3 / 0 of them;
#                              8 us with events of the default, system-scope release: the engine's timing events release to the device)


def stride_for(steps, time_every):
    """profiled days: one kind of kernel per profiled day, the four kinds at phases 0, 1/4, 1/2, 3/4 of the stride -- any `stride`
    consecutive days time every kernel once.  A timestamped dispatch adds 5.5-8 us of wall time to a short run
    (TIMED_LAUNCH_COST_US): until round 5 a run of fewer than 64 days timed one kernel EVERY day, and the round driver's 20-day
    window carried 15 of them -- 6 us a step of a 39 us step, the instrument a seventh of the measurement.  Now: stride 8 below
    64 days (the window: two or three samples of each kernel, 7 dispatches), 16 from there on, 4 for runs shorter than 16 days."""
    if time_every:
        return max(4, int(time_every) // 4 * 4)
    return 4 if steps < 16 else 8 if steps < 64 else 16


def run_gpu(variables, ages, seed, steps, warmup, device, dist=None, preheat=0, stride=16, preheat_runs=2, attribution='mirror', day_only=False):
    import numpy as np
    import torch
    from reina_model_amd import engine as eng
    from reina_model_amd import sharding, simulation
    comm = _shared_comm(sharding, attribution) if dist is not None else None
    for rep in range(preheat_runs if preheat else 0):
        # throw-away runs of the same workload (untimed, separate state): bring the GPU out of its
        # idle power state and pay one-time runtime costs before the measured simulation exists.
        # (event pools, allocator pools, first timestamped dispatches), profiled like the timed one.
        pre = simulation.make_context(variables, age_counts=ages, seed=seed + 1000003 + rep, device=device, comm=comm)
        pre.engine.profile_enable(stride)
        pre.run(preheat, record_history=True)
        pre.run(min(steps, 64), record_history=True)   # a read-back of the timed run's size class too (its pinned staging block)
        pre.synchronize()
        pre.engine.profile_read_kernels()
        del pre
    # A Context is part of reference cycles: the throw-away ones die when the collector gets to them -- destroying
    # their engines (hipFree of a hundred MB each, event destruction: synchronising calls) -- and if that happens inside a
    # short timed window it doubles it (tools/short_window.py: 20 steps took 39-41 us each, or 52-65 right after a
    # collection that freed a Context).  Collect them here, before the measured simulation exists, and keep the
    # collector off while the clock runs.
    import gc
    gc.collect()
    torch.cuda.synchronize()
    ctx = simulation.make_context(variables, age_counts=ages, seed=seed, device=device, comm=comm)
    # the event-timed launch path is switched on BEFORE the warm-up so its one-time costs (event
    # pool, first timestamped dispatches) are not billed to the timed region
    # (day_only: the short cold window times its dominant kernel alone -- a timestamped dispatch costs 5.5 us of a 740 us window; the
    # other kernels' figures come from the warm window beside it)
    ctx.engine.profile_enable(-stride if day_only else stride)
    if warmup:
        ctx.run(warmup, record_history=False)
    ctx.synchronize()
    ctx.engine.profile_read_kernels()  # discard the warm-up launches
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    gc.disable()
    try:   # (an exception out of the timed region must not leave the collector off for the workloads that follow)
        t0 = time.perf_counter()
        hist = ctx.run(steps, record_history=True)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t1 = time.perf_counter()
    finally:
        gc.enable()
    prof = ctx.engine.profile_read_kernels()
    ctx.engine.profile_enable(False)
    A = eng.MAX_AGES

    def tot(name):
        i = eng.C_NAMES.index(name)
        return hist[:, i * A:(i + 1) * A].sum(axis=1).astype(np.float64)

    world = comm.world if comm is not None else 1
    inf = tot('infected') / world                      # (sharded: the history is global, a launch streams one shard)
    sc = hist[:, eng.C_NR * A:]
    days = np.arange(steps) + warmup
    contacts = sc[:, eng.S_EXPOSED_PER_DAY].astype(np.float64) / world
    # row d is the state BEFORE day d ran: day d's own new infections / contacts are in row d + 1
    new_inf = tot('new_infections') / world
    stats = dict(
        infected_on_scan_days=float(inf[days % stride == 0].mean() if (days % stride == 0).any() else inf.mean()),
        mean_infected=float(inf.mean()),
        contacts_per_day=float(contacts[1:].mean() if steps > 1 else contacts.mean()),
        contacts_on_scan_days=float(contacts[1:][(days[:-1] % stride) == 0].mean()
                                    if steps > 1 and ((days[:-1] % stride) == 0).any() else contacts.mean()),
        new_infections_per_day=float(new_inf[1:].mean() if steps > 1 else new_inf.mean()),
        removed_per_day=float(sc[1:, eng.S_TOTAL_INFECTORS].mean() / world if steps > 1 else sc[:, eng.S_TOTAL_INFECTORS].mean() / world),
        final_all_infected=int(tot('all_infected')[-1]),
        peak_infected=int(tot('infected').max()),
    )
    rccl_world = None
    if comm is not None and getattr(comm, 'direct', None) is not None:
        rccl_world = comm.direct.count()
    return dict(dt=t1 - t0, prof=prof, stats=stats, n_local=ctx.total_people, rccl_world=rccl_world, exchange_fill=ctx.exchange_fill())


def lib_sha256():
    from reina_model_amd import engine as eng
    h = hashlib.sha256()
    with open(eng.HIP_LIB_PATH, 'rb') as f:
        h.update(f.read())
    return h.hexdigest()


def src_sha256():
    """sha256 over the sources the library is built from (csrc/*, include/reina_hip.h), in name order: the second way a
    traffic.json is matched to the tree -- hipcc's builds of identical sources differ in a build id"""
    from reina_model_amd import build as _b
    h = hashlib.sha256()
    for f in sorted(set(_b.DEPS)):
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


_TRACE_US = {}


def traffic_for(key):
    """PMC figures of profiles/traffic.json for configuration `key`, only if collected on this very binary (or on a build of
    the very sources): (bytes per day, {kernel: bytes per day}, {kernel: utilisation figures}, note)"""
    try:
        tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    except Exception:
        return None, {}, {}, 'no profiles/traffic.json'
    how = None
    if tj.get('lib_sha256') == lib_sha256():
        how = 'the binary being timed (sha256 match)'
    elif tj.get('src_sha256') and tj.get('src_sha256') == src_sha256():
        how = 'a build of the very sources this binary was built from (sha256 over csrc/ and include/ matches; the binary was rebuilt since)'
    if how is None:
        return None, {}, {}, 'profiles/traffic.json was collected on another binary (sha256 %s..., commit %s) and other sources: not reported' % (
            str(tj.get('lib_sha256'))[:12], tj.get('commit'))
    val = tj.get('per_day_bytes', {}).get(key)
    strip = lambda d: {k.split('<')[0]: v for k, v in d.items()}
    # (the SQ passes are collected over the 365-day scenarios: a short window of the same population carries the year's figures)
    util = tj.get('utilisation', {}).get(key) or (tj.get('utilisation', {}).get('hus', {}) if key == 'hus_window' else {})
    global _TRACE_US
    _TRACE_US = tj.get('kernel_trace_us', {}).get(key, {})   # rocprofv3 --kernel-trace of the same 365-day command (roofline_obj)
    return (val, strip(tj.get('per_kernel_bytes_per_day', {}).get(key, {})), strip(util),
            'rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE summed over the kernels of a day, mean over the scenario; collected on %s; commit %s' % (how, tj.get('commit')))


def roofline_obj(n_agents, res, steps, stride, traffic_key=None):
    """the day against HBM three ways -- priced at SURVEY 8d's algorithmic bytes (what a hot-word streamer must move), at the
    bytes the PMC counters saw move, and at a restated byte model of the engine as built -- + every kernel's HIP-event time +
    the VALU-issue utilisation of the two kernels that own the day"""
    st, prof = res['stats'], res['prof']
    ms_per_step = res['dt'] * 1000 / steps
    day_s = ms_per_step * 1e-3
    day_bytes = 4.0 * n_agents + 4.0 * st['mean_infected'] + 4.0 * st['contacts_per_day'] + 12.0 * st['new_infections_per_day']
    achieved = day_bytes / day_s / 1e9
    alg = {   # algorithmic bytes per launch of the kernels that own a term of B_alg
        'k_day': 4.0 * n_agents + 4.0 * st['infected_on_scan_days'] + 4.0 * st['contacts_on_scan_days'],
        'k_hosp_install': 12.0 * st['new_infections_per_day'],
        # (round 6: a small unsharded population's whole day is ONE launch -- opening, stream, installs between launch-wide barriers --:
        # its algorithmic bytes are the day's)
        'k_small_day': day_bytes,
    }
    moved_day, moved_k, util, note = (None, {}, {}, None)
    trace = {}
    if traffic_key is not None:
        moved_day, moved_k, util, note = traffic_for(traffic_key)
        trace = dict(_TRACE_US) if moved_day else {}
    kernels, ksum = {}, 0.0
    for k in DAY_KERNELS:
        ms, n = prof.get(k, (0.0, 0))
        if not n:
            continue
        us = ms * 1000.0 / n
        ent = dict(avg_launch_us=round(us, 3), timed_launches=n)
        if k in alg:
            # what a kernel that streamed every hot word would have to move, over this kernel's time: a ratio that EXCEEDS 1
            # where the kernel reads less than that (a sparse day streams one bit per agent) -- not a fraction of anything
            ent['algorithmic_bytes_per_launch'] = round(alg[k], 1)
            ent['vs_hot_word_streamer'] = round(alg[k] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
        if k in moved_k:
            # bytes the kernel moved (PMC, mean per simulated day; the kernel runs once a day) over its mean launch time
            ent['moved_bytes_per_launch'] = moved_k[k]
            ent['moved_GBs'] = round(moved_k[k] / (us * 1e-6) / 1e9, 2)
            ent['moved'] = round(moved_k[k] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
        if k == 'k_hosp_install':
            # the kernel's bound is the chip's random-access rate, not bytes: an infection is one load (the target's claim), two atomics
            # on the bit planes (Infinity-Cache resident), one on the source's count and two stores (the target's word, its infector); an
            # onset two stores; an agent booked into the R statistics one load -- priced at the measured rates (RANDOM_RATES)
            r_ = RANDOM_RATES
            ns = (st['new_infections_per_day'] * (1 / r_['load'] + 2 / r_['atomic_cached'] + 1 / r_['atomic'] + 2 / r_['store'])
                  + st['new_infections_per_day'] * 2 / r_['store'] + st.get('removed_per_day', 0.0) / r_['load'])
            ent['random_access'] = {'floor_us': round(ns / 1000.0, 2), 'frac': round(ns / 1000.0 / us, 4),
                                    'rates_G_per_s': r_, 'note': 'mean day; the launch also pays its 4-5 us dispatch floor and, on ordered days, the bed / ICU walk'}
        if k in trace:
            # the same kernel in rocprofv3's kernel trace of the same command (profiles/): mean over ALL its launches of the year, every
            # dispatch timestamped alike -- a timestamped dispatch among plain ones (the HIP-event figure above) measures about 1 us more
            ent['trace_avg_launch_us'] = trace[k][1]
            ent['trace_launches'] = trace[k][2]
        if k in util:
            ent['valu'] = {'mean_day': util[k].get('valu_mean_day'), 'peak_day': util[k].get('valu_peak_day')}
            ent['waiting'] = {'mean_day': util[k].get('waiting_mean_day'), 'peak_day': util[k].get('waiting_peak_day')}
        kernels[k] = ent
    every_day = [k for k in ('k_small_day', 'k_open', 'k_day', 'k_hospital', 'k_remote', 'k_hosp_install', 'k_hosp_sort', 'k_hosp_walk', 'k_xchg') if k in kernels]
    ksum = sum(kernels[k]['avg_launch_us'] for k in every_day)
    for k in every_day:
        kernels[k]['share_of_kernel_time'] = round(kernels[k]['avg_launch_us'] / ksum, 4) if ksum else None
    # the engine AS BUILT, restated in bytes at the memory side's CALIBRATED granularity (round 6: tools/ubench_pmc.hip,
    # profiles/pmc_calibration.json -- every read request that leaves L2 is a 128-byte line, whatever the load's width, two loads
    # of one line 64 bytes apart are one request; scattered stores and atomics are 32-byte sectors): a sparse day streams the
    # ACTIVE bit plane (N / 8) and fetches one LINE per active agent's word, brings k_day's table image into each of the 8
    # XCDs' L2, looks up one line of the infected plane per contact that can transmit (1 in 50), and an infection reads two lines
    # (the target's record at install, the source's count at the R bookkeeping) and writes eighteen sectors over its course (word,
    # record, two plane bits, the source's count, its onset's two stores, three later transitions of its word, queue appends)
    sparse = n_agents >= 8_000_000
    model = ((n_agents / 8.0 if sparse else 4.0 * n_agents) + (128.0 if sparse else 4.0) * st['mean_infected'] + 8 * DAY_IMAGE_BYTES
             + 128.0 * st['contacts_per_day'] / 50.0 + (2 * 128.0 + 18 * 32.0) * st['new_infections_per_day'])
    out = dict(bound='hbm', achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(achieved / HBM_PEAK_GBS, 5),
               traffic=moved_day, scope='whole day: sum of B_alg over the timed days / wall time of the timed region (SURVEY.md 8d)',
               algorithmic_bytes_per_day=round(day_bytes, 1),
               algorithmic_bytes_formula='4*N + 4*N_infected + 4*contacts + 12*new_infections (per day, means over the timed days)',
               frac_meaning='vs_hot_word_streamer: the day priced at the bytes a kernel that read every agent\'s hot word would move. The engine '
                            'does not (sparse days stream one bit per agent): `moved` is the measured figure',
               moved=None, model_bytes_per_day=round(model, 1),
               model_formula=('N/8 + 128*N_infected' if sparse else '4*N + 4*N_infected') + ' + 8 XCDs * table image + 128*contacts/50 + 832*new_infections '
                             '(128-byte read lines, 32-byte write sectors: profiles/pmc_calibration.json)',
               wasted=None,
               ms_per_step=round(ms_per_step, 6), kernel_us_per_day=round(ksum, 3), kernels=kernels,
               kernel_timing='HIP events (start/stop of the dispatch packet, launch stream) inside the timed region; on a profiled day '
                             'one kind of kernel is timed: stride %d days per kind (%d timestamped dispatches in the timed region, about '
                             '%.0f us of wall time each: tools/window_probe.py)' % (
                                 stride, sum(int(x.get('timed_launches', 0)) for x in kernels.values()), TIMED_LAUNCH_COST_US))
    if trace:
        # the kernels' time per simulated day by rocprofv3's trace of the same 365-day command (kernels that do not run every day
        # count with the days they run): what the step is to be compared with -- the sum of the HIP-event means above carries the
        # price of a timestamped dispatch among plain ones and can exceed the mean step
        out['kernel_us_per_day_trace'] = round(sum(v[0] for v in trace.values()), 3)
    if moved_day:
        # `moved`: HBM bytes per day from the PMC counters / the day's wall time / peak -- north_star's "achieved HBM GB/s against the
        # chip's peak" read literally; `wasted`: moved bytes over the restated model's (sector granularity included)
        out['moved'] = dict(bytes_per_day=moved_day, GBs=round(moved_day / day_s / 1e9, 2), frac=round(moved_day / day_s / 1e9 / HBM_PEAK_GBS, 5))
        out['wasted'] = round(moved_day / model, 3)
    if traffic_key is not None:
        out['traffic_note'] = note
    # what bounds the day (the SQ passes of profiles/r05_sq_*.csv): not HBM bandwidth in either regime
    vd = (util.get('k_day') or {})
    out['valu'] = {k: kernels[k]['valu'] for k in ('k_day', 'k_hosp_install') if k in kernels and 'valu' in kernels[k]} or None
    out['bound_by_regime'] = {
        'quiet_day': 'latency: three to four launches of dependent round trips at the dispatch floor (a 12.5 MB bit plane that HBM moves in 1.6 us)',
        'peak_day': 'valu issue in k_day (contact sampling: Philox + table search per contact)' + (
            ', %.2f of the chip\'s VALU issue slots over the launch' % vd['valu_peak_day'] if vd.get('valu_peak_day') else '') +
            '; scattered-access latency in k_hosp_install'}
    dom = 'k_day' if 'k_day' in kernels and kernels['k_day']['avg_launch_us'] * kernels['k_day']['timed_launches'] >= \
        kernels.get('k_small_day', {}).get('avg_launch_us', 0.0) * kernels.get('k_small_day', {}).get('timed_launches', 0) else \
        'k_small_day' if 'k_small_day' in kernels else None
    if dom:
        out['dominant_kernel'] = dict(name=dom, **kernels[dom])
    return out


LINE_LIMIT = 8000          # the round driver parses ONE line of stdout; round 5's 24.5 KB line did not parse
DETAIL_PATH = os.path.join(ROOT, 'profiles', 'bench_detail.json')


def _kernel_brief(r, name):
    """one kernel of a roofline object in five numbers: HIP-event us, trace us, PMC bytes per launch, moved fraction, VALU issue"""
    k = (r.get('kernels') or {}).get(name)
    if not k:
        return None
    out = {'us': k.get('avg_launch_us')}
    for short, key in (('trace_us', 'trace_avg_launch_us'), ('pmc_bytes', 'moved_bytes_per_launch'), ('moved', 'moved'),
                       ('alg_bytes', 'algorithmic_bytes_per_launch')):
        if k.get(key) is not None:
            out[short] = k[key]
    if k.get('valu'):
        out['valu'] = [k['valu'].get('mean_day'), k['valu'].get('peak_day')]
    if k.get('random_access'):
        out['random_access_frac'] = k['random_access']['frac']
    return out


def _roofline_brief(r):
    """the contract's roofline object (bound / achieved / peak / unit / frac / traffic) + what the verdict reads: the measured
    fraction, the day's kernels, the dominant kernel by name"""
    if not r or 'error' in r:
        return r
    out = {k: r.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')}
    out['alg_bytes_per_day'] = r.get('algorithmic_bytes_per_day')
    out['moved_frac'] = (r.get('moved') or {}).get('frac')
    out['wasted'] = r.get('wasted')
    out['kernel_us_per_day'] = r.get('kernel_us_per_day')
    if 'kernel_us_per_day_trace' in r:
        out['kernel_us_per_day_trace'] = r['kernel_us_per_day_trace']
    out['dominant_kernel'] = (r.get('dominant_kernel') or {}).get('name')
    out['kernels'] = {k: _kernel_brief(r, k) for k in (r.get('kernels') or {})}
    return out


def compact_line(out):
    """The ONE line the driver parses: the contract's fields, `roofline` (headline + a per-size table), `cpu_baseline`, `ensemble`
    -- at most LINE_LIMIT bytes.  Everything else (per-kernel objects in full, formulae, notes) is in profiles/bench_detail.json."""
    line = {k: out[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                'vs_baseline', 'dtype', 'data') if k in out}
    for k in ('value_warm', 'ms_per_step_warm', 'headline_is'):
        if k in out:
            line[k] = out[k]
    cfg = out['config']
    line['config'] = {k: cfg[k] for k in ('workload', 'agents_total', 'parallelism', 'attribution', 'rccl_world', 'collective') if k in cfg}
    rl = _roofline_brief(out['roofline'])
    rl['scope'] = 'whole day: B_alg (4N + 4N_inf + 4C + 12I_new, SURVEY 8d) / wall time; moved_frac = PMC bytes / wall time / peak'
    sizes = {}
    for key, fs in (out.get('full_scenario') or {}).items():
        if 'error' in fs:
            sizes[key] = {'error': fs['error'][:120]}
            continue
        r = fs['roofline']
        sizes[key] = {'agents': fs.get('agents'), 'value': fs['value'], 'ms_per_step': fs['ms_per_step'], 'frac': r['frac'],
                      'moved_frac': (r.get('moved') or {}).get('frac'), 'traffic': r.get('traffic'), 'wasted': r.get('wasted'),
                      'kernel_us_per_day': r.get('kernel_us_per_day_trace', r.get('kernel_us_per_day')),
                      'k_day': _kernel_brief(r, 'k_day'), 'k_hosp_install': _kernel_brief(r, 'k_hosp_install'),
                      'k_open_us': ((r.get('kernels') or {}).get('k_open') or {}).get('avg_launch_us')}
        if 'k_small_day' in (r.get('kernels') or {}):
            sizes[key]['k_small_day'] = _kernel_brief(r, 'k_small_day')
        sizes[key] = {k: v for k, v in sizes[key].items() if v is not None}
    if sizes:
        rl['full_scenario_365d'] = sizes
    line['roofline'] = rl
    cb = out.get('cpu_baseline')
    if cb:
        c = {k: cb[k] for k in ('value', 'unit', 'cores', 'kind') if k in cb}
        c['sample'] = 'oracle A (C restatement of cythonsim, bit-exact vs its goldens), HUS, the same %d+%d-day window, 1 thread' % (out['warmup'], out['steps'])
        c['cpu_model'] = cb.get('cpu_model')
        if cb.get('all_cores'):
            c['all_cores'] = {'value': cb['all_cores']['value'], 'cores': cb['all_cores']['cores']}
        cal = cb.get('calibration')
        if cal:
            c['port_over_cythonsim'] = list(cal['port_over_reference_365_days'].values())
        line['cpu_baseline'] = c
    for key in ('ensemble', 'large', 'strong'):
        o = out.get(key)
        if not o:
            continue
        if 'error' in o:
            line[key] = {'error': o['error'][:160]}
            continue
        e = {k: o[k] for k in ('value', 'unit', 'ms_per_step', 'members_per_gpu', 'n_gpus', 'scaling', 'attribution', 'rccl_world') if k in o}
        ks = o.get('kernels') or (o.get('roofline') or {}).get('kernels') or {}
        e['kernel_us'] = {k: v.get('avg_launch_us') for k, v in ks.items()}
        if o.get('roofline') and 'frac' in o['roofline']:
            e['roofline'] = {k: o['roofline'].get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')}
            if o['roofline'].get('moved_frac') is not None:
                e['roofline']['moved_frac'] = o['roofline']['moved_frac']
        for other in ('exact', 'mirror'):
            if isinstance(o.get(other), dict):
                e[other] = {k: o[other][k] for k in ('value', 'ms_per_step', 'error') if k in o[other]}
        line[key] = e
    line['timing'] = {k: out['timing'][k] for k in ('timed_region_s', 'process_wall_s')}
    line['detail'] = 'profiles/bench_detail.json'
    text = json.dumps(line, separators=(',', ':'))
    if len(text) > LINE_LIMIT:
        # never an unparseable line: drop the widest optional parts first
        for victim in (('roofline', 'full_scenario_365d'), ('roofline', 'kernels'), ('strong',), ('large',), ('ensemble',)):
            d = line
            for k in victim[:-1]:
                d = d.get(k, {})
            d.pop(victim[-1], None)
            line['truncated'] = True
            if len(json.dumps(line, separators=(',', ':'))) <= LINE_LIMIT:
                break
    return line


def write_detail(out):
    """the full objects of the run, beside the line (best effort: a read-only tree must not cost the line)"""
    for path in (DETAIL_PATH, os.path.join(ROOT, 'gpurun_out', 'bench_detail.json')):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, 'w') as f:
                    json.dump(out, f, indent=1)
        except OSError:
            pass


def cpu_baseline(variables, ages, seed, steps, warmup, budget_s=10.0, max_runs=16):
    """oracle A (sequential C restatement, bit-exact vs the reference) on one core over the SAME window"""
    from oracle import seq_oracle
    import numpy as np
    n = int(np.asarray(ages).sum())
    timed, runs = 0.0, 0
    while timed < budget_s and runs < max_runs:
        ctx = seq_oracle.make_context(variables, ages, seed + runs)
        for _ in range(warmup):
            ctx.iterate()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.iterate()
        timed += time.perf_counter() - t0
        runs += 1
    return dict(value=round(n * steps * runs / timed, 1), unit='agent-days/s', cores=1, kind='port',
                sample='sequential C restatement of cythonsim (bit-exact vs reference goldens), HUS %d agents, the same window as '
                       'the GPU line (%d untimed + %d timed days of the default scenario), %d run(s) with seeds %d.., 1 thread, '
                       '%.1f s timed' % (n, warmup, steps, runs, seed, timed))


_CPU_WORKER = r"""
import copy, sys, time
sys.path.insert(0, %(root)r)
from oracle import seq_oracle
from reina_model_amd import datasets
from reina_model_amd.variables import VARIABLE_DEFAULTS
ages = datasets.get_population_for_area()
ctx = seq_oracle.make_context(copy.deepcopy(VARIABLE_DEFAULTS), ages, %(seed)d)
for _ in range(%(warmup)d):          # the same untimed days as the GPU line
    ctx.iterate()
while time.time() < %(start)f:
    time.sleep(0.01)
t0 = time.time()
for _ in range(%(days)d):
    ctx.iterate()
print(t0, time.time())
"""


_CPU_HELPER = r"""
import json, os, subprocess, sys, time
root, days, warmup, procs, go = %(root)r, %(days)d, %(warmup)d, %(procs)d, %(go)r
worker = %(worker)r
while not os.path.exists(go):          # the GPU measurements come first: all-core load slows the host
    time.sleep(0.05)
    if os.getppid() == 1:
        sys.exit(0)
start = time.time() + 10.0 + 0.05 * warmup
ps = [subprocess.Popen([sys.executable, '-c', worker %% dict(root=root, seed=1000 + k, start=start, days=days, warmup=warmup)],
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

    def __init__(self, days, max_procs=64, warmup=0):
        import tempfile
        self.days = days
        self.warmup = warmup
        self.same_window = False
        self.procs = min(os.cpu_count() or 1, max_procs)
        self.max_procs = max_procs
        self.go = os.path.join(tempfile.gettempdir(), 'reina_bench_go_%d' % os.getpid())
        code = _CPU_HELPER % dict(root=ROOT, days=days, warmup=warmup, procs=self.procs, go=self.go, worker=_CPU_WORKER)
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
        return dict(value=round(HUS_AGENTS * self.days * len(spans) / wall, 1), unit='agent-days/s', cores=len(spans), kind='port',
                    sample='%d concurrent sequential simulations (one per core, cores capped at %d), HUS %d agents, '
                           '%d untimed + %d timed days each%s, %.1f s wall' % (
                               len(spans), self.max_procs, HUS_AGENTS, self.warmup, self.days,
                               ' (the GPU line\'s window)' if self.same_window else ' (NOT the GPU line\'s window)', wall))


def ensemble_line(seeds, days, device, dist=None):
    """BASELINE config 5: a Monte-Carlo ensemble of HUS simulations stepped as an engine group (one launch
    per phase for all members); 128 seeds = the per-GPU batch of the 1024-seed configuration.  Over several
    ranks: replicas only, every rank runs its own `seeds` members, no data-path collective."""
    import torch
    from reina_model_amd import datasets, ensemble, simulation
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    ages = datasets.get_population_for_area()
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    planner = simulation.make_context(v, age_counts=ages, seed=0, device=device)
    plan = planner.make_plan(days)
    members = [simulation.make_context(v, age_counts=ages, seed=100 + rank * seeds + k, device=device) for k in range(seeds)]
    members[0].engine.profile_enable(16)
    # The members' history comes back through ONE page-locked block (325 MB for 128 members x 365 days); PyTorch's caching host
    # allocator hands the same block to every later run of the process, but page-locking it the first time costs 20-30 ms of a run
    # whose kernels take 80 (tools/ens_probe.py).  Like the headline's warm-up steps, the block is requested once before the timed
    # region: the figure is that of the second and every later ensemble of a process (said in the object's `note`).
    from reina_model_amd import engine as _eng
    warm = torch.empty(seeds * days * _eng.COUNTER_WORDS, dtype=torch.int32, pin_memory=True)
    del warm
    # (and the group kernels' first launches -- the runtime loads a kernel's code when it is first launched -- by a warm-up group
    # of two throw-away members over five days: the ensemble's counterpart of the headline's warm-up steps)
    # (round 6: a two-member group streams the hot words -- 1.6 tiles per wave --, the timed group of `seeds` members the bit planes: another
    # instantiation of k_day, whose first launch cost 1.1 ms inside the timed region on some boxes -- its mean over 23 timed launches read
    # 154 us there and 107 elsewhere.  The warm-up members are made with the sparse form forced, so the kernels it loads are the timed ones.)
    prev_mode = os.environ.get('REINA_DAY_MODE')
    os.environ['REINA_DAY_MODE'] = 'sparse'
    try:
        pre = [simulation.make_context(v, age_counts=ages, seed=90 + k, device=device) for k in range(2)]
    finally:
        if prev_mode is None:
            os.environ.pop('REINA_DAY_MODE', None)
        else:
            os.environ['REINA_DAY_MODE'] = prev_mode
    ensemble.run_group_plan(pre, pre[0].make_plan(5))
    del pre
    # (round 6: as in run_gpu -- the Contexts of the runs before this one sit in reference cycles and die when the collector gets to them,
    # destroying their engines with synchronising calls; inside the timed region that was 20-25 ms of a 80 ms ensemble on every box:
    # tools/ens_first_run.py measures 78.6 ms for this very sequence with nothing to collect, the line read 104)
    import gc
    gc.collect()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    gc.disable()
    try:
        t0 = time.perf_counter()
        hist = ensemble.run_group_plan(members, plan)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
    finally:
        gc.enable()
    prof = members[0].engine.profile_read_kernels()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n = int(ages.sum())
    # the group's step against HBM, like a single population's day: B_alg summed over the members (means over members and days of
    # infected agents, contacts and new infections from their histories) / the wall time of a group step; `traffic`: PMC bytes per
    # group step when profiles/traffic.json holds a pass over this very command on this binary (key ensemble_<seeds>)
    import numpy as np
    A = _eng.MAX_AGES
    i_inf, i_new = _eng.C_NAMES.index('infected'), _eng.C_NAMES.index('new_infections')
    mean_inf = float(hist[:, :, i_inf * A:(i_inf + 1) * A].sum(axis=2).mean())
    mean_new = float(hist[:, 1:, i_new * A:(i_new + 1) * A].sum(axis=2).mean())
    mean_con = float(hist[:, 1:, _eng.C_NR * A + _eng.S_EXPOSED_PER_DAY].mean())
    step_bytes = seeds * (4.0 * n + 4.0 * mean_inf + 4.0 * mean_con + 12.0 * mean_new)
    step_s = dt / days
    moved, _, _, note = traffic_for('ensemble_%d' % seeds)
    roof = dict(bound='hbm', achieved=round(step_bytes / step_s / 1e9, 2), peak=HBM_PEAK_GBS, unit='GB/s',
                frac=round(step_bytes / step_s / 1e9 / HBM_PEAK_GBS, 5), traffic=moved,
                moved_frac=round(moved / step_s / 1e9 / HBM_PEAK_GBS, 5) if moved else None,
                algorithmic_bytes_per_group_step=round(step_bytes, 1), scope='one step of the whole group (%d member-days): B_alg summed over the members / wall time' % seeds,
                traffic_note=note)
    return dict(workload='%d seeds x HUS %d agents x %d days per GPU, one engine group per GPU (config 5: replicas only)' % (seeds, n, days), roofline=roof,
                value=round(world * seeds * n * days / dt, 1), unit='agent-days/s', ms_per_step=round(dt * 1000 / days, 6),
                members_per_gpu=seeds, n_gpus=world,
                note='timed: group construction, every launch of the year, the read-back of all members\' history rows and final counters; the page-locked block the history comes back through was requested once before the timed region and a two-member group ran five days first (a process\'s first ensemble pays some 30 ms more: page-locking, the kernels\' first launches)',
                kernels={k: dict(avg_launch_us=round(ms * 1000 / c, 2), timed_launches=c) for k, (ms, c) in prof.items() if c})


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n_gpus):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD process tree before this
    process has made any GPU call (it never makes one), relay the output, exit with the child's code."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=env)


def main():
    t_process = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=365)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--agents', type=int, default=0, help='synthetic population per GPU (0 = HUS)')
    ap.add_argument('--sizes', default='50000000,100000000,200000000',
                    help='synthetic populations of the full_scenario object (HUS is always there)')
    ap.add_argument('--large-agents', type=int, default=50_000_000, help='N > 1: agents per GPU of the `large` object')
    ap.add_argument('--strong-agents', type=int, default=100_000_000,
                    help='N > 1: TOTAL agents of the `strong` object (north_star\'s target configuration: 10^8 agents over the ranks)')
    ap.add_argument('--no-strong', action='store_true')
    ap.add_argument('--attribution', default='mirror', choices=('mirror', 'exact'),
                    help='N > 1: cross-shard infector links of the headline, `large` and `strong` runs -- mirror: stand-in infectors, ONE '
                         'all-reduce per day (north_star\'s exchange); exact: true links, two exchanges per day, four on contact-tracing days (SURVEY 8 f-4). '
                         '`strong` carries the other mode\'s figure beside it')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-sizes', action='store_true', help='skip the full_scenario object')
    ap.add_argument('--no-large', action='store_true')
    ap.add_argument('--no-ensemble', action='store_true')
    ap.add_argument('--ensemble-seeds', type=int, default=128)
    ap.add_argument('--cpu-all-cores-days', type=int, default=-1,
                    help='all-cores CPU baseline: -1 = the window of the GPU line (--warmup untimed + --steps timed days), '
                         '0 = off, n > 0 = the first n days')
    ap.add_argument('--cpu-max-procs', type=int, default=64, help='concurrent CPU simulations of the all-cores baseline')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--time-every', type=int, default=0,
                    help='each kind of kernel carries HIP event timestamps on every k-th day (0: 4 / 8 / 16 by run length)')
    ap.add_argument('--preheat-days', type=int, default=365,
                    help='days of a throw-away simulation run before the measured one (GPU clocks, one-time costs)')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(a.gpus))

    stride = stride_for(a.steps, a.time_every)
    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    cpu_all = None
    if world_env == 1 and not a.no_cpu and not a.agents and a.cpu_all_cores_days != 0:
        # helper spawned before anything initialises the GPU; runs last
        if a.cpu_all_cores_days < 0:
            cpu_all = CpuAllCores(a.steps, a.cpu_max_procs, warmup=a.warmup)
            cpu_all.same_window = True
        else:
            cpu_all = CpuAllCores(a.cpu_all_cores_days, a.cpu_max_procs)

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

    import numpy as np
    from reina_model_amd import datasets
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    per_gpu = a.agents if a.agents else HUS_AGENTS
    if a.agents or world > 1:
        # weak scaling: the GLOBAL population is per_gpu x world agents, sharded over the ranks
        # (BASELINE configs[3] shape: HUS age structure, beds / ICU / imports scaled with it)
        v, ages = scaled_scenario(v, per_gpu * world)
        workload = 'synthetic %d agents (%d per GPU; HUS age structure + FI contact matrix, beds/ICU/imports scaled), default scenario, days %d..%d' % (
            per_gpu * world, per_gpu, a.warmup, a.warmup + a.steps - 1)
        traffic_key = str(a.agents) if world == 1 and a.steps == 365 else None
    else:
        ages = datasets.get_population_for_area()
        workload = 'HUS %d agents (BASELINE configs[1]), default scenario (variables.py:227-435), days %d..%d' % (
            HUS_AGENTS, a.warmup, a.warmup + a.steps - 1)
        # (profiles/traffic.json holds the 365-day scenario's mean bytes per day: a shorter window gets no traffic figure)
        traffic_key = 'hus' if a.steps == 365 else 'hus_window' if (a.steps, a.warmup) == (20, 5) else None

    def max_over_ranks(dt):
        if world == 1:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # The headline of a short window is the COLD one (round-5 verdict): the library keeps the count-threshold rows of contact tables it
    # has built before in a process-wide table, and the untimed preheat runs of the same scenario would fill it -- 8 % of a 20-day
    # window.  REINA_COUNT_ROW_CACHE=0 computes every row afresh, in the preheat runs and in the timed window alike; the window with
    # the table filled is reported beside it as value_warm.
    cold_headline = world == 1 and not a.agents and a.steps <= 64
    if cold_headline:
        os.environ['REINA_COUNT_ROW_CACHE'] = '0'
    try:
        res = run_gpu(v, ages, a.seed, a.steps, a.warmup, device, dist, preheat=a.preheat_days, stride=stride, attribution=a.attribution,
                      day_only=cold_headline)
    finally:
        os.environ.pop('REINA_COUNT_ROW_CACHE', None)
    res['dt'] = max_over_ranks(res['dt'])
    total_agents = int(np.asarray(ages).sum())
    n_agents = total_agents // world   # agents one k_day launch streams on this rank
    value = total_agents * a.steps / res['dt']

    large_sharded = None
    if world > 1 and not a.no_large and not a.agents:
        # BASELINE configs[3] shape on the same ranks: 50 M agents per GPU (4 x 10^8 on 8), sharded, full scenario
        vl, agesl = scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), a.large_agents * world)
        rl = run_gpu(vl, agesl, a.seed, 365, 0, device, dist, preheat=60, stride=16, preheat_runs=1, attribution=a.attribution)
        rl['dt'] = max_over_ranks(rl['dt'])
        tot_l = int(np.asarray(agesl).sum())
        large_sharded = {
            'workload': 'synthetic %d agents (%d per GPU, BASELINE configs[3] shape), default scenario scaled, 365 days' % (tot_l, tot_l // world),
            'value': round(tot_l * 365 / rl['dt'], 1), 'unit': 'agent-days/s', 'ms_per_step': round(rl['dt'] * 1000 / 365, 6),
            'roofline': roofline_obj(tot_l // world, rl, 365, 16), 'final_all_infected': rl['stats']['final_all_infected'],
        }
    strong_sharded = None
    if world > 1 and not a.no_strong and not a.agents:
        # north_star's target configuration: 10^8 agents in TOTAL over the ranks -- strong scaling of the metric's size
        # (the N = 1 counterpart is full_scenario["100000000"] of the single-GPU line)
        vs_, ages_s = scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), a.strong_agents)
        rs = run_gpu(vs_, ages_s, a.seed, 365, 0, device, dist, preheat=60, stride=16, preheat_runs=1, attribution=a.attribution)
        rs['dt'] = max_over_ranks(rs['dt'])
        tot_s = int(np.asarray(ages_s).sum())
        strong_sharded = {
            'workload': 'synthetic %d agents in total (%d per GPU), default scenario scaled, 365 days' % (tot_s, tot_s // world),
            'scaling': 'strong', 'attribution': a.attribution, 'value': round(tot_s * 365 / rs['dt'], 1), 'unit': 'agent-days/s',
            'ms_per_step': round(rs['dt'] * 1000 / 365, 6), 'rccl_world': rs['rccl_world'],
            'roofline': roofline_obj(tot_s // world, rs, 365, 16), 'final_all_infected': rs['stats']['final_all_infected'],
            'expect': 'strong scaling of 10^8 agents is NEGATIVE on quiet days: a shard of 1.25 x 10^7 spends 67-86 us of kernels on a quiet '
                      'day (two more launches, the cross-shard branches; profiles/r05_evidence/sharded_day_kernels.txt) plus the collectives\' '
                      'latency, against 51 us for the whole 10^8 on one GPU; only the peak days gain (137-142 us per shard against 340). '
                      'Sharding buys capacity (4 x 10^8 agents), not speed, at these sizes.  roofline.kernels: k_remote / k_hosp_sort are '
                      'the launches sharding adds, `collective` is the event-timed cost of RCCL itself per exchange point',
        }
        # the other attribution mode on the same ranks (exact: the true links, contact / feedback / tracing records through
        # ncclAllToAll -- two exchanges a day, four on contact-tracing days, instead of one)
        other = 'exact' if a.attribution == 'mirror' else 'mirror'
        try:
            ro = run_gpu(vs_, ages_s, a.seed, 365, 0, device, dist, preheat=0, stride=16, preheat_runs=0, attribution=other)
            ro['dt'] = max_over_ranks(ro['dt'])
            strong_sharded[other] = {'value': round(tot_s * 365 / ro['dt'], 1), 'ms_per_step': round(ro['dt'] * 1000 / 365, 6),
                                     'kernels': {k: round(ms * 1000 / c, 2) for k, (ms, c) in ro['prof'].items() if c},
                                     'exchange_segment_fill': dict(zip(('peak_records', 'capacity'), (ro if other == 'exact' else rs)['exchange_fill']))}
        except Exception as e:   # noqa: BLE001 -- reported in the line itself
            strong_sharded[other] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    ens_dist = None
    if world > 1 and not a.no_ensemble and not a.agents:
        try:
            ens_dist = ensemble_line(min(a.ensemble_seeds, 32), 365, device, dist)
        except Exception as e:   # noqa: BLE001 -- reported in the line itself
            ens_dist = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}

    if rank == 0:
        out = {
            'metric': 'agent-days/sec', 'value': round(value, 1), 'unit': 'agent-days/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(res['dt'] * 1000 / a.steps, 6), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'u32', 'data': 'synthetic',
            'config': {'workload': workload, 'agents_total': total_agents, **({'attribution': a.attribution} if world > 1 else {}),
                       'parallelism': 'single GPU' if world == 1 else (
                           'agents sharded x%d, one RCCL all-reduce per day (infection pressure + the shards\' bed / ICU event maps)' % world if a.attribution == 'mirror' else
                           'agents sharded x%d, exact attribution: two ncclAllToAll per day (four on contact-tracing days), the capacity words and event maps in their trailers' % world),
                       'final_all_infected': res['stats']['final_all_infected'], 'peak_infected_in_window': res['stats']['peak_infected']},
            'roofline': roofline_obj(n_agents, res, a.steps, stride, traffic_key),
            'notes': ['the timed region covers host planning, launches, every kernel of the days and the read-back of their history rows; '
                      'on a day whose contact tables change, the host-side count-threshold rows of values seen before in this '
                      'process (the untimed warm-up runs of the same scenario) are taken from a process-wide table instead of '
                      'being recomputed (reina_hip.hip count_row_for; 83 us of host time per table change, DESIGN section 5); '
                      'all GPU work of such a day, the table upload included, is inside the timed region',
                      'roofline.achieved / frac price the day at its ALGORITHMIC bytes (SURVEY 8d: every agent\'s 4-byte hot word once a '
                      'day, ...).  Since round 4 a population of >= 4 tiles per wave (about 8 M agents on a whole chip) does not read them: '
                      'k_day streams one ACTIVE bit per agent and fetches the words of the agents whose bit is set (the reference leaves '
                      'everybody who is not infected at once too, main.pyx:1974-1975), so the HBM bytes actually moved are about half the '
                      'algorithmic ones at 5e7-2e8 agents.  roofline.moved is the measured figure (PMC bytes / wall time / 8 TB/s: north_star\'s '
                      '"achieved HBM GB/s against the chip\'s peak"), roofline.wasted prices it against a restated byte model of the engine as '
                      'built, roofline.valu / bound_by_regime say what bounds the day now: latency on quiet days, VALU issue on peak days -- '
                      'not HBM bandwidth.  A kernel\'s vs_hot_word_streamer may exceed 1: it is a comparison with a streamer, not a fraction'],
        }
        if world > 1:
            out['notes'].append(
                'reading the scaling curve: the headline keeps %d agents PER GPU, a size at which a day is a chain of launches at the '
                'dispatch floor -- a sharded day has two more launches (k_hosp_presort, k_remote) and the all-reduce, about 56 us of '
                'kernels per shard against 37 unsharded (DESIGN section 6) plus RCCL\'s small-message latency: expect 50-60 %% '
                'weak-scaling efficiency there by construction.  `large` (5 x 10^7 agents per GPU, BASELINE configs[3]) is the weak-'
                'scaling figure to read, `strong` (10^8 agents in total, north_star\'s target) the strong-scaling one; its N = 1 '
                'counterpart is full_scenario["100000000"] of the single-GPU line' % per_gpu)
            # how the per-day exchange ran: ncclCommCount of the communicator the day stream's in-stream all-reduce uses, or null
            # when the direct communicator could not be built and the exchange fell back to torch.distributed.all_reduce
            out['rccl_world'] = res['rccl_world']
            out['config']['rccl_world'] = res['rccl_world']
            out['config']['collective'] = ('ncclAllReduce queued on the day stream (own RCCL communicator)' if res['rccl_world']
                                           else 'torch.distributed.all_reduce (%s backend; direct RCCL communicator unavailable)' % dist.get_backend())

        if cold_headline:
            out['headline_is'] = 'cold: count-threshold rows of every table change computed inside the timed window (REINA_COUNT_ROW_CACHE=0)'
            # the same window once the process has seen the scenario's mobility values (one preheat run fills the table of rows)
            try:
                rw_ = run_gpu(v, ages, a.seed, a.steps, a.warmup, device, None, preheat=a.preheat_days, stride=stride, preheat_runs=1)
                out['value_warm'] = round(total_agents * a.steps / rw_['dt'], 1)
                out['ms_per_step_warm'] = round(rw_['dt'] * 1000 / a.steps, 6)
                # the kernels the cold window did not time (it carries timestamps on k_day alone): from the warm window
                rw_['prof'].update({k: x for k, x in res['prof'].items() if x[1]})
                warm_roof = roofline_obj(n_agents, dict(res, prof=rw_['prof']), a.steps, stride, traffic_key)
                for k, ent in warm_roof['kernels'].items():
                    if k not in out['roofline']['kernels']:
                        out['roofline']['kernels'][k] = dict(ent, timed_in='the warm window')
                out['roofline']['kernel_us_per_day'] = warm_roof['kernel_us_per_day']
            except Exception as e:   # noqa: BLE001
                out['value_warm'] = None
                out['warm_error'] = str(e)[:200]

        def extra(store, key, fn):
            # the additional workloads must not take the headline line down with them
            try:
                store[key] = fn()
            except Exception as e:   # noqa: BLE001 -- reported in the line itself
                store[key] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}

        def full_line(n_cfg, label):
            import gc
            gc.collect()   # (the previous size's Contexts sit in reference cycles: several GB of HBM each at 1e8-2e8 agents)
            if n_cfg:
                vl, agesl = scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), n_cfg)
                key = str(n_cfg)
            else:
                vl, agesl, key = copy.deepcopy(VARIABLE_DEFAULTS), datasets.get_population_for_area(), 'hus'
            r = run_gpu(vl, agesl, a.seed, 365, 0, device, preheat=60 if n_cfg else 365, stride=16, preheat_runs=1 if n_cfg else 2)
            nl = r['n_local']
            return {'workload': '%s: %d agents, default scenario%s, all 365 days' % (label, nl, ' scaled' if n_cfg else ''), 'agents': nl,
                    'value': round(nl * 365 / r['dt'], 1), 'unit': 'agent-days/s', 'ms_per_step': round(r['dt'] * 1000 / 365, 6),
                    'roofline': roofline_obj(nl, r, 365, 16, key), 'final_all_infected': r['stats']['final_all_infected'],
                    'peak_infected': r['stats']['peak_infected']}

        if world == 1 and not a.agents and not a.no_sizes:
            fs = out['full_scenario'] = {}
            extra(fs, 'hus', lambda: full_line(0, 'HUS (BASELINE configs[1])'))
            labels = {50_000_000: 'synthetic 5e7 (BASELINE configs[2])', 100_000_000: "synthetic 1e8 (BASELINE metric's 100 M agents)",
                      200_000_000: 'synthetic 2e8 (SURVEY 8d: HBM-resident regime)'}
            for n_cfg in [int(x) for x in a.sizes.split(',') if x]:
                extra(fs, str(n_cfg), lambda n_cfg=n_cfg: full_line(n_cfg, labels.get(n_cfg, 'synthetic')))
        if large_sharded is not None:
            out['large'] = large_sharded
        if strong_sharded is not None:
            out['strong'] = strong_sharded
        if ens_dist is not None:
            out['ensemble'] = ens_dist
        if not a.no_ensemble and world == 1 and not a.agents:
            extra(out, 'ensemble', lambda: ensemble_line(a.ensemble_seeds, 365, device))
        if not a.no_cpu and world == 1:   # the CPU baseline is an N=1 figure
            hus = datasets.get_population_for_area()
            out['cpu_baseline'] = cpu_baseline(copy.deepcopy(VARIABLE_DEFAULTS), hus, a.seed, a.steps, a.warmup)
            out['cpu_baseline']['cores_available'] = os.cpu_count()
            try:
                with open('/proc/cpuinfo') as f:
                    out['cpu_baseline']['cpu_model'] = next(l.split(':', 1)[1].strip() for l in f if l.startswith('model name'))
            except Exception:
                pass
            # how the port relates to the reference it stands for: both timed on one core of the BUILD container over the same days
            # (tools/cpu_calibration.py, needs /root/reference; a recorded constant here -- the reference never travels)
            try:
                cal = json.load(open(os.path.join(ROOT, 'profiles', 'cpu_calibration.json')))
                r_ = cal['results']
                out['cpu_baseline']['calibration'] = {
                    'where': 'build container, %s, one core; %s' % (cal['cpu_model'], cal['workload']),
                    'oracle_a_agent_days_per_s': {k: x['agent_days_per_s'] for k, x in r_['oracle_a'].items()},
                    'cythonsim_agent_days_per_s': {k: x['agent_days_per_s'] for k, x in r_['cythonsim'].items()},
                    'cythonsim_noexcept_build_agent_days_per_s': {k: x['agent_days_per_s'] for k, x in r_['cythonsim_noexcept'].items()},
                    'port_over_reference_365_days': {
                        'container Cython (3.2: GIL round trip per cdef nogil call)': round(r_['oracle_a']['365']['agent_days_per_s'] / r_['cythonsim']['365']['agent_days_per_s'], 2),
                        'legacy_implicit_noexcept build (what the reference pins, cython 3.0a6)': round(r_['oracle_a']['365']['agent_days_per_s'] / r_['cythonsim_noexcept']['365']['agent_days_per_s'], 2)},
                    'note': 'the port is FASTER than the reference it stands for: divide cpu_baseline.value by the ratio to estimate cythonsim on this host'}
            except Exception:
                pass
            if cpu_all is not None:
                r_all = cpu_all.run()
                if r_all is not None:
                    out['cpu_baseline']['all_cores'] = r_all
        out['timing'] = {
            'timed_region_s': round(res['dt'], 6), 'process_wall_s': round(time.perf_counter() - t_process, 2),
            'note': 'the process also builds contexts, runs untimed preheat simulations, the full_scenario / ensemble '
                    'configurations and the CPU baselines; value = agents x steps / timed_region_s'}
        write_detail(out)
        print(json.dumps(compact_line(out), separators=(',', ':')), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
