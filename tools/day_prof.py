"""Where a wave of k_day spends its cycles (-DREINA_DAY_PROF build: tools/build_variant.sh PROF): shader-clock cycles summed over
the waves, per part of the loop, on chosen days of the 365-day scenario.  python tools/day_prof.py [agents] [day ...]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import bench
from reina_model_amd import simulation, datasets
from reina_model_amd.variables import VARIABLE_DEFAULTS
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
days = [int(x) for x in sys.argv[2:]] or [20, 60, 93, 200]
if n > 2_000_000:
    v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), n)
else:
    v, ages = copy.deepcopy(VARIABLE_DEFAULTS), datasets.get_population_for_area()
ctx = simulation.make_context(v, age_counts=ages, seed=0)
what = os.environ.get('REINA_PROF_WHAT', '?')
d = 0
for day in days:
    ctx.run(day - d); d = day
    ctx.synchronize()
    ctx.engine.tensors['mirror'].zero_()
    ctx.engine.profile_enable(1)
    ctx.run(1); d += 1
    ctx.synchronize()
    k = ctx.engine.profile_read_kernels()
    ctx.engine.profile_enable(0)
    m = ctx.engine.alloc.to_host(ctx.engine.tensors['mirror']).view(np.uint64).astype(np.float64)
    rows = m[64:].reshape(-1, 4)
    rows = rows[rows[:, 3] > 0]
    waves = max(1.0, float(len(rows)))
    allr = m[64:].reshape(-1, 4)
    nwg = len(allr) // 16
    wg = allr[:nwg * 16, 0].reshape(nwg, 16)
    wg = wg[wg.max(axis=1) > 0]
    if len(wg):
        last = wg.max(axis=1) / 1000.0
        print('   workgroups %d: loop of the LAST wave of a workgroup min %.1f mean %.1f max %.1f kcycles; of the FIRST wave mean %.1f' % (
            len(wg), last.min(), last.mean(), last.max(), (wg.min(axis=1) / 1000.0).mean()))
    if len(wg):   # the pieces of this part by a wave's rank in its workgroup's finishing order (the pool hands batches to the early ones)
        pieces = allr[:nwg * 16, 2].reshape(nwg, 16)[allr[:nwg * 16, 0].reshape(nwg, 16).max(axis=1) > 0]
        order = np.argsort(wg, axis=1)
        byrank = np.take_along_axis(pieces, order, axis=1).mean(axis=0)
        loops = np.take_along_axis(wg, order, axis=1).mean(axis=0) / 1000.0
        print('   pieces by finishing rank: ' + ' '.join('%.1f' % x for x in byrank))
        print('   loop kcycles by rank:     ' + ' '.join('%.0f' % x for x in loops))
    print('part %s day %d: k_day %.1f us; per wave: whole loop %.1f kcycles (slowest %.1f), this part %.1f kcycles in %.1f pieces (%.2f kcycles each)' % (
        what, day, k['k_day'][0] * 1000.0, rows[:, 0].sum() / waves / 1000.0, (rows[:, 0].max() if len(rows) else 0.0) / 1000.0, rows[:, 1].sum() / waves / 1000.0, rows[:, 2].sum() / waves, rows[:, 1].sum() / max(1.0, rows[:, 2].sum()) / 1000.0))
