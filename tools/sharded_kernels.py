"""Kernel time of a sharded day on one GPU: G in-process shards stepped in lock-step, every launch timestamped (HIP events):
python tools/sharded_kernels.py <G> <total agents> [lo:hi] [exact|mirror]  -> us per launch of each kernel of shard 0 over days lo..hi"""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import bench
from reina_model_amd import sharding, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
G, total = int(sys.argv[1]), int(float(sys.argv[2]))
lo, hi = (int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '92:104').split(':'))
v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), total)
members = []
attribution = sys.argv[4] if len(sys.argv) > 4 else 'exact'
ctxs = [simulation.make_context(v, age_counts=ages, seed=0, comm=sharding.InProcessComm(r, G, members, attribution=attribution)) for r in range(G)]
for d in range(lo):
    sharding.step_shards_together(ctxs)
ctxs[0].engine.read_counters()
ctxs[0].engine.profile_enable(1)
ctxs[0].engine.profile_read_kernels()
for d in range(lo, hi):
    sharding.step_shards_together(ctxs)
for c in ctxs:
    c.engine.read_counters()
prof = ctxs[0].engine.profile_read_kernels()
tot = sum(ms for ms, c in prof.values())
print('%s: %d shards x %d agents, days %d-%d, shard 0: kernels %.1f us/day |' % (attribution, G, ctxs[0].total_people, lo, hi, tot * 1000 / (hi - lo)),
      ' '.join('%s %.1f' % (k, ms * 1000 / c) for k, (ms, c) in prof.items() if c), flush=True)
