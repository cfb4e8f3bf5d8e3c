"""diagnostic: where do the HIP engine and oracle B part on the first ordered day of test_large_bed_event_sets?"""
import copy, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from reina_model_amd import datasets, simulation, engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
import par_backend
v = copy.deepcopy(VARIABLE_DEFAULTS)
v.update(hospital_beds=900, icu_units=60)
ivs = [['import-infections', '2020-02-19', 3000], ['import-infections', '2020-02-25', 3000, 'b1.1.7'], ['test-all-with-symptoms', '2020-02-20']]
ages = datasets.scaled_population(2_500_000)
days = int(sys.argv[1]) if len(sys.argv) > 1 else 22
gpu = simulation.make_context(v, age_counts=ages, seed=3, interventions=ivs)
cpu = simulation.make_context(v, age_counts=ages, seed=3, interventions=ivs, engine_factory=par_backend.par_engine_factory)
hg, hc = gpu.run(days), cpu.run(days)
bad = np.nonzero((hg != hc).any(axis=1))[0]
print('rows differing', bad[:5])
cg, cc = gpu.engine.read_counters(), cpu.engine.read_counters()
w = np.nonzero(cg != cc)[0]
print('final counters differ at', w[:20], cg[w[:20]], cc[w[:20]])
A = eng.MAX_AGES
for row in bad[:4]:
    print('row', row, {eng.C_NAMES[k]: int(hg[row, k * A:(k + 1) * A].sum() - hc[row, k * A:(k + 1) * A].sum()) for k in range(eng.C_NR)
                       if (hg[row, k * A:(k + 1) * A] != hc[row, k * A:(k + 1) * A]).any()},
          'scalars', np.nonzero(hg[row, eng.C_NR * A:] != hc[row, eng.C_NR * A:])[0])
    for k in range(eng.C_NR):
        dd = hg[row, k * A:(k + 1) * A] - hc[row, k * A:(k + 1) * A]
        if dd.any():
            print('   ', eng.C_NAMES[k], {int(a): int(dd[a]) for a in np.nonzero(dd)[0]})
a = gpu.engine.alloc.to_host(gpu.engine.tensors['hot']).view(np.uint32)
b = np.asarray(cpu.engine.tensors['hot']).view(np.uint32)
d = np.nonzero(a != b)[0]
print('hot words differ for', len(d), 'agents')
for i in d[:20]:
    print(i, hex(a[i]), hex(b[i]), 'state', a[i] & 7, b[i] & 7, 'sev', (a[i] >> 3) & 7)
ctl = gpu.engine.alloc.to_host(gpu.engine.tensors['control'])
print('control', ctl[:20])
he = gpu.engine.alloc.to_host(gpu.engine.tensors['hosp_events']).view(np.uint64)
R = eng.hosp_ranges(int(np.sum(ages)))
counts = he[:R // 2].view(np.uint32)
print('bucket counts', counts.tolist())
keys_off = R // 2 + 2 * R
cap = eng.hosp_bucket_cap(int(np.sum(ages)), gpu.engine.config.max_hosp_events)
allk = np.concatenate([he[keys_off + r * cap: keys_off + r * cap + counts[r]] for r in range(R)])
ids = ((allk >> np.uint64(2)) & np.uint64(0xFFFFFFFF)).astype(np.int64)
print('events', len(allk), 'distinct agents', len(np.unique(ids)), 'types', np.bincount((allk & np.uint64(3)).astype(np.int64), minlength=4))
agg = he[R // 2: R // 2 + R]
print('published', int((agg >> np.uint64(63)).sum()), 'of', R)
