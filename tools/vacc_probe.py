"""What does a vaccination day cost?  The default scenario scaled to `agents` with a programme of `weekly` vaccinations from day 15 on
(ages 16-100; `tiers`: three programmes side by side, ages 70-100 / 50-69 / 16-49, a third of the number each): every kernel of
days 10..25 timed.  python tools/vacc_probe.py [agents] [weekly] [tiers]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from reina_model_amd import simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
weekly = int(float(sys.argv[2])) if len(sys.argv) > 2 else n // 30
v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), n)
ivs = [['vaccinate', '2020-03-04', weekly, 16, 100]]
if len(sys.argv) > 3 and sys.argv[3] == 'tiers':
    ivs = [['vaccinate', '2020-03-04', weekly // 3, lo, hi] for lo, hi in ((70, 100), (50, 69), (16, 49))]
ctx = simulation.make_context(v, age_counts=ages, seed=0, interventions=ivs)
ctx.run(10)
ctx.engine.profile_enable(1)
for d in range(10, 26):
    ctx.run(1)
    k = ctx.engine.profile_read_kernels()
    print('day %d: %s | vaccinated so far %d' % (d, ' '.join('%s %.1f' % (nm, ms * 1000.0) for nm, (ms, c) in k.items() if c), int(ctx.generate_state()['vaccinated'].sum())), flush=True)
