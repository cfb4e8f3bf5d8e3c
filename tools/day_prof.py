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
    waves = max(1.0, m[31])
    print('part %s day %d: k_day %.1f us; per wave: whole loop %.1f kcycles, this part %.1f kcycles in %.1f pieces (%.2f kcycles each)' % (
        what, day, k['k_day'][0] * 1000.0, m[32] / waves / 1000.0, m[33] / waves / 1000.0, m[34] / waves, m[33] / max(1.0, m[34]) / 1000.0))
