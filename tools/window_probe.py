"""The round driver's window (bench.py --steps 20 --warmup 5) taken apart: Context.run(20) as bench.py times it -- the call itself, the
closing torch.cuda.synchronize, and the same window again and again in one process (is the first one different?)."""
import copy, gc, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from reina_model_amd import datasets, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
ages = datasets.get_population_for_area()
strides = [int(x) for x in os.environ.get('PROBE_STRIDES', '4,0,8,16').split(',')]
rows = {}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
  for stride in strides:
      gc.collect(); torch.cuda.synchronize()
      ctx = simulation.make_context(copy.deepcopy(VARIABLE_DEFAULTS), age_counts=ages, seed=rep)
      ctx.engine.profile_enable(stride)
      ctx.run(5, record_history=False); ctx.synchronize(); ctx.engine.profile_read_kernels(); torch.cuda.synchronize()
      gc.disable()
      t0 = time.perf_counter(); hist = ctx.run(20, record_history=True); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
      gc.enable()
      k = ctx.engine.profile_read_kernels()
      rows.setdefault(stride, []).append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, sum(c for _, c in k.values())))
      pass
for stride, v in rows.items():
    v = v[2:]
    tot = sorted(a + b for a, b, _ in v)
    print('stride %2d: timed launches %d; run(20) + synchronize: median %.1f us (%.3f us/step), min %.1f, max %.1f; synchronize alone median %.1f' % (
        stride, v[0][2], tot[len(tot) // 2], tot[len(tot) // 2] / 20, tot[0], tot[-1], sorted(b for _, b, _ in v)[len(v) // 2]))
