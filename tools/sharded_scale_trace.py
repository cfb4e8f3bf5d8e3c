"""4 in-process shards x 25 M agents, 120 days (for rocprofv3 --kernel-trace): cost of k_remote etc. at scale."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reina_model_amd import sharding, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
total = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
days = int(sys.argv[3]) if len(sys.argv) > 3 else 120
v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), total)
members = []
ctxs = [simulation.make_context(v, age_counts=ages, seed=2, comm=sharding.InProcessComm(r, G, members)) for r in range(G)]
for d in range(days):
    sharding.step_shards_together(ctxs)
for c in ctxs:
    c.synchronize()
print('done')
