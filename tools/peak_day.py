"""Kernel times (HIP events on every launch) on the peak days and on quiet days of the default scenario:
python tools/peak_day.py [agents]  -> us per launch of each kernel on days 92-103 and 300-311"""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import bench
from reina_model_amd import simulation, datasets
from reina_model_amd.variables import VARIABLE_DEFAULTS
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
if n > 2_000_000:
    v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), n)
else:
    v, ages = copy.deepcopy(VARIABLE_DEFAULTS), datasets.get_population_for_area()
ctx = simulation.make_context(v, age_counts=ages, seed=0)
day = 0
import time
wins = ((92, 104, 'peak'), (300, 312, 'quiet'))
if len(sys.argv) > 2:   # python tools/peak_day.py <agents> lo:hi[,lo:hi...]
    wins = tuple((int(x.split(':')[0]), int(x.split(':')[1]), 'win') for x in sys.argv[2].split(','))
for lo, hi, label in wins:
    ctx.run(lo - day, record_history=False)
    ctx.synchronize()
    ctx.engine.profile_enable(1)
    ctx.engine.profile_read_kernels()
    t0 = time.perf_counter()
    ctx.run(hi - lo, record_history=True)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) * 1e6 / (hi - lo)
    prof = ctx.engine.profile_read_kernels()
    ctx.engine.profile_enable(False)
    day = hi
    tot = sum(ms for ms, c in prof.values())
    print('%s %-5s days %d-%d: kernels %.1f us/day, wall (every launch timestamped) %.1f |' % (sys.argv[1] if len(sys.argv) > 1 else '1e8', label, lo, hi, tot * 1000 / (hi - lo), wall),
          ' '.join('%s %.1f' % (k, ms * 1000 / c) for k, (ms, c) in prof.items() if c), flush=True)
