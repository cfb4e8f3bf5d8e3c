"""dense vs sparse HIP on the three-variants scenario with DAY_F_DEBUG: which sources differ on the first bad day"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from reina_model_amd import datasets, engine as eng, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
v = copy.deepcopy(VARIABLE_DEFAULTS)
v.update(hospital_beds=15, icu_units=3)
v['variants'] = [{'name': 'b1.1.7', 'infectiousness_multiplier': 1.3},
                 {'name': 'p.1', 'infectiousness_multiplier': 1.6, 'mean_incubation_duration': 4.0, 'p_asymptomatic_infection': 50.0}]
ivs = [['import-infections', '2020-02-19', 40], ['import-infections', '2020-02-25', 30, 'b1.1.7'],
       ['import-infections', '2020-03-01', 30, 'p.1'], ['test-all-with-symptoms', '2020-02-22'],
       ['import-infections-weekly', '2020-03-05', 70, 30, 20], ['test-with-contact-tracing', '2020-03-20', 60],
       ['limit-mobility', '2020-03-25', 40], ['import-infections-weekly', '2020-04-20', 35, 0, 100]]
ages = datasets.scaled_population(50000)
os.environ['REINA_DAY_FLAGS'] = '16'
ctx = {}
for mode in ('dense', 'sparse'):
    os.environ['REINA_DAY_MODE'] = mode
    ctx[mode] = simulation.make_context(v, age_counts=ages, seed=17, interventions=ivs)
N = int(ages.sum())
prev = {m: np.zeros(N, dtype=np.int64) for m in ('dense', 'sparse')}
prevp = {m: np.zeros(N, dtype=np.int64) for m in ('dense', 'sparse')}
for d in range(40):
    dbg = {}
    for mode in ('dense', 'sparse'):
        ctx[mode].run(1)
        wi = ctx[mode].engine.alloc.to_host(ctx[mode].engine.tensors['work_items']).view(np.uint32)[:2 * N].reshape(N, 2).copy()
        dbg[mode] = wi
        sc = ctx[mode].engine.alloc.to_host(ctx[mode].engine.tensors['scan_lists']).view(np.uint32)[:N].astype(np.int64)
        pc = ctx[mode].engine.alloc.to_host(ctx[mode].engine.tensors['scan_lists']).view(np.uint32)[N:2 * N].astype(np.int64)
        today_push = pc - prevp[mode]
        prevp[mode] = pc
        dbg[mode + '_push'] = today_push
        today_scans = sc - prev[mode]
        prev[mode] = sc
        odd = np.nonzero(today_scans > 1)[0]
        if len(odd):
            print('day', d, mode, 'agents scanned more than once:', [(int(i), int(today_scans[i])) for i in odd[:10]])
        dbg[mode + '_scans'] = today_scans
    a, b = dbg['dense'], dbg['sparse']
    today = lambda x: ((x[:, 0] >> 8) & 0xFFF) == d
    ta, tb = today(a) & (a[:, 0] != 0), today(b) & (b[:, 0] != 0)
    diff = np.nonzero((ta != tb) | (ta & tb & ((a[:, 0] != b[:, 0]) | (a[:, 1] != b[:, 1]))))[0]
    ca = ctx['dense'].engine.read_counters(); cb = ctx['sparse'].engine.read_counters()
    if len(diff) or not np.array_equal(ca, cb):
        print('day', d, 'sources dense', ta.sum(), 'sparse', tb.sum(), 'differing', len(diff), 'counters equal', np.array_equal(ca, cb))
        print('  sum of counts dense', int(((a[ta, 0] & 0xFF) - 1).sum()), 'sparse', int(((b[tb, 0] & 0xFF) - 1).sum()))
        sa, sb = dbg['dense_scans'], dbg['sparse_scans']
        for i in np.nonzero(sa != sb)[0][:10]:
            print('  scans of agent %d: dense %d sparse %d' % (i, sa[i], sb[i]))
        mir = ctx['dense'].engine.alloc.to_host(ctx['dense'].engine.tensors['mirror']).view(np.uint64)
        for k in range(int(min(mir[0], 62))):
            e = int(mir[1 + k])
            print('   log kind %d wave %d lane %d pos %d a %d b %d i&0xffff %d' % (e >> 60, (e >> 48) & 0xFFF, (e >> 40) & 0xFF, (e >> 32) & 0xFF, (e >> 24) & 0xFF, (e >> 16) & 0xFF, e & 0xFFFF))
        for i in diff[:10]:
            print('  agent %d: dense %#x %#x (today %s)  sparse %#x %#x (today %s)' % (i, a[i, 0], a[i, 1], ta[i], b[i, 0], b[i, 1], tb[i]))
        break
else:
    print('no difference in 40 days')
