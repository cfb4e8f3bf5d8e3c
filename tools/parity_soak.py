"""One-off soak: many more randomised scenarios than the suite runs (HIP vs oracle B, bit for bit)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_parity_gpu as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
OFF = int(os.environ.get('REINA_SOAK_OFFSET', '0'))   # other scenarios than the last soak's
bad = 0
t0 = time.time()
if len(sys.argv) > 2 and sys.argv[2] == 'sharded':
    # the same randomised scenarios on a population split over 2-4 in-process shards, HIP vs oracle B
    import par_backend
    from reina_model_amd import sharding, simulation
    for case in range(n):
        rng = np.random.default_rng(300000 + OFF + case)
        v, ages, days, ivs, ipc = T._random_scenario(rng)
        os.environ['REINA_DAY_MODE'] = ('dense', 'sparse')[case % 2]   # (round 4: both forms of k_day's stream; small populations would all be dense)
        G = int(rng.integers(2, 5))
        seed = int(rng.integers(0, 2 ** 31))
        if ipc is not None and v['hospital_beds'] == 0 and ipc.get('in_icu', 0) > 0:
            ipc = dict(ipc, in_icu=0)   # (the reference refuses ICU patients without beds, and so do both engines: tested elsewhere)
        if case % 100 == 99:
            print('sharded soak: %d scenarios so far, %d mismatches, %.0f s' % (case + 1, bad, time.time() - t0), flush=True)
        gm, cm = [], []
        # (round 5: both attribution modes -- exact: the records through the exchange segments, links checked to be the true ones)
        attribution = os.environ.get('REINA_SOAK_ATTRIBUTION') or ('exact', 'exact', 'mirror')[case % 3]
        try:
            gpu = [simulation.make_context(v, age_counts=ages, seed=seed, interventions=ivs, ipc=ipc,
                                           comm=sharding.InProcessComm(r, G, gm, attribution=attribution)) for r in range(G)]
            cpu = [simulation.make_context(v, age_counts=ages, seed=seed, interventions=ivs, ipc=ipc,
                                           comm=sharding.InProcessComm(r, G, cm, attribution=attribution),
                                           engine_factory=par_backend.par_engine_factory) for r in range(G)]
            for d in range(min(days, 100)):
                sharding.step_shards_together(gpu)
                sharding.step_shards_together(cpu)
                if d % 10 == 9:
                    # (a capacity that runs out -- an exchange segment, a list -- fails BOTH runs loudly, but what was dropped
                    # differs: once both engines have raised the same problem the scenario is over)
                    pg = int(sharding.reduce_counters(gpu)[T.eng.C_NR * T.eng.MAX_AGES + T.eng.S_PROBLEM])
                    pc = int(sharding.reduce_counters(cpu)[T.eng.C_NR * T.eng.MAX_AGES + T.eng.S_PROBLEM])
                    if pg >= 100 or pc >= 100:
                        assert pg == pc, 'day %d: problem %d on the GPU, %d on the CPU' % (d, pg, pc)
                        raise StopIteration
                    for a, b in zip(gpu, cpu):
                        assert np.array_equal(a.engine.read_counters(), b.engine.read_counters()), 'day %d' % d
            for a, b in zip(gpu, cpu):
                assert np.array_equal(a.engine.read_counters(), b.engine.read_counters()), 'final'
                T._assert_state_equal(a, b)
            if attribution == 'exact' and not sharding.reduce_counters(gpu)[T.eng.C_NR * T.eng.MAX_AGES + T.eng.S_PROBLEM]:
                from shard_util import assert_links_are_true
                assert_links_are_true(gpu)
        except StopIteration:
            print('sharded case %d (G=%d, %s): both engines ran out of a capacity (problem %d), skipped' % (case, G, attribution, pg), flush=True)
        except AssertionError as e:
            bad += 1
            print('MISMATCH sharded case %d (G=%d, %s): %s' % (case, G, attribution, str(e)[:300]), flush=True)
    print('sharded soak: %d scenarios, %d mismatches, %.0f s' % (n, bad, time.time() - t0))
    sys.exit(0)
for case in range(n):
    if case % 250 == 249:
        print('soak: %d scenarios so far, %d mismatches, %.0f s' % (2 * (case + 1), bad, time.time() - t0), flush=True)
    for kind, seed0 in (('random', 100000), ('extreme', 200000)):
        rng = np.random.default_rng(seed0 + OFF + case)
        v, ages, days, ivs, ipc = T._random_scenario(rng)
        if os.environ.get('REINA_FUSED_DAY') == '1':
            os.environ.pop('REINA_DAY_MODE', None)   # (round 6: the one-launch form of a small population's days is taken only when no form of k_day is forced)
        else:
            os.environ['REINA_DAY_MODE'] = ('dense', 'sparse')[(case + (kind == 'extreme')) % 2]   # (round 4: both forms of k_day's stream)
        try:
            if kind == 'extreme':
                total = int(rng.integers(600, 6000))
                ages = T.datasets.scaled_population(total)
                v['infectiousness_multiplier'] = float(rng.uniform(1.0, 3.0))
                v['hospital_beds'] = int(rng.integers(0, 3)); v['icu_units'] = int(rng.integers(0, 2))
                from datetime import date, timedelta
                d0 = date.fromisoformat(v['start_date'])
                ivs = list(ivs) + [['import-infections', (d0 + timedelta(days=int(rng.integers(0, 20)))).isoformat(), int(total * rng.uniform(0.2, 1.5))],
                                   ['test-with-contact-tracing', (d0 + timedelta(days=int(rng.integers(0, 30)))).isoformat(), 100]]
                if ipc is not None:
                    ipc = {k: min(val, total // 12) for k, val in ipc.items()}
                days = min(days, 90)
            T._run_and_compare(v, ages, int(rng.integers(0, 2 ** 31)), days, interventions=ivs, chunk=40, ipc=ipc)
        except AssertionError as e:
            bad += 1
            print('MISMATCH %s case %d: %s' % (kind, case, str(e)[:300]), flush=True)
print('soak: %d scenarios, %d mismatches, %.0f s' % (2 * n, bad, time.time() - t0))
