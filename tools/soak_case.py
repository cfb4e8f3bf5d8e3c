"""Re-run ONE scenario of the sharded randomised soak (tools/parity_soak.py ... sharded) and show where HIP and oracle B part:
python tools/soak_case.py <case> [offset]  -> first day / shard / counter that differs, with both values"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_parity_gpu as T
import par_backend
from reina_model_amd import engine as eng, sharding, simulation
case = int(sys.argv[1]); OFF = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(300000 + OFF + case)
v, ages, days, ivs, ipc = T._random_scenario(rng)
G = int(rng.integers(2, 5))
seed = int(rng.integers(0, 2 ** 31))
if ipc is not None and v['hospital_beds'] == 0 and ipc.get('in_icu', 0) > 0:
    ipc = dict(ipc, in_icu=0)
print('case', case, 'G', G, 'agents', int(np.sum(ages)), 'days', days, 'beds', v['hospital_beds'], 'icu', v['icu_units'], 'ipc', ipc)
for iv in ivs:
    print('  ', iv)
gm, cm = [], []
gpu = [simulation.make_context(v, age_counts=ages, seed=seed, interventions=ivs, ipc=ipc, comm=sharding.InProcessComm(r, G, gm)) for r in range(G)]
cpu = [simulation.make_context(v, age_counts=ages, seed=seed, interventions=ivs, ipc=ipc, comm=sharding.InProcessComm(r, G, cm),
                               engine_factory=par_backend.par_engine_factory) for r in range(G)]
names = {}
for k in dir(eng):
    if k.startswith('C_') and isinstance(getattr(eng, k), int) and k != 'C_NR':
        names[getattr(eng, k)] = k
snames = {getattr(eng, k): k for k in dir(eng) if k.startswith('S_') and isinstance(getattr(eng, k), int) and k != 'S_NR'}
A = eng.MAX_AGES
for d in range(min(days, 100)):
    sharding.step_shards_together(gpu)
    sharding.step_shards_together(cpu)
    bad = False
    for r, (a, b) in enumerate(zip(gpu, cpu)):
        x, y = a.engine.read_counters(), b.engine.read_counters()
        if not np.array_equal(x, y):
            bad = True
            idx = np.nonzero(x != y)[0]
            for i in idx[:12]:
                if i < eng.C_NR * A:
                    print('day %d shard %d: %s[age %d] hip %d oracle %d' % (d, r, names.get(i // A, i // A), i % A, x[i], y[i]))
                else:
                    print('day %d shard %d: %s hip %d oracle %d' % (d, r, snames.get(i - eng.C_NR * A, i - eng.C_NR * A), x[i], y[i]))
            cx = a.engine.alloc.to_host(a.engine.tensors['control']); cy = np.asarray(b.engine.tensors['control']) if 'control' in getattr(b.engine, 'tensors', {}) else None
            print('   hip control[0:20]', cx[:20].tolist())
    if not bad:
        for r, (a, b) in enumerate(zip(gpu, cpu)):
            try:
                T._assert_state_equal(a, b)
            except AssertionError as e:
                bad = True
                print('day %d shard %d: state differs: %s' % (d, r, str(e)[:200]))
                tg, tc = a.engine.tensors, b.engine.tensors
                for name in ('hot', 'infector', 'n_infected'):
                    x = a.engine.alloc.to_host(tg[name]).view(np.uint32); y = np.asarray(tc[name]).view(np.uint32)
                    idx = np.nonzero(x != y)[0]
                    for i in idx[:6]:
                        print('   %s[%d]: hip %#x oracle %#x' % (name, i, x[i], y[i]))
    if bad:
        break
else:
    print('no difference in the counters over', min(days, 100), 'days')
