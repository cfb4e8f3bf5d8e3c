"""One-off: 4 shards x 25 M agents (10^8 in total, BASELINE configs[3] shape) stepped in lock-step on
one GPU vs the same sharded run on oracle B, bit for bit.  Too slow for the suite."""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import bench, par_backend
from reina_model_amd import sharding, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
days = int(sys.argv[1]) if len(sys.argv) > 1 else 110
G = 4
v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), 100_000_000)
gm, cm = [], []
gpu = [simulation.make_context(v, age_counts=ages, seed=6, comm=sharding.InProcessComm(r, G, gm)) for r in range(G)]
cpu = [simulation.make_context(v, age_counts=ages, seed=6, comm=sharding.InProcessComm(r, G, cm),
                               engine_factory=par_backend.par_engine_factory) for r in range(G)]
t0 = time.time()
for d in range(days):
    sharding.step_shards_together(gpu)
    sharding.step_shards_together(cpu)
    if d % 10 == 9 or d == days - 1:
        for a, b in zip(gpu, cpu):
            assert np.array_equal(a.engine.read_counters(), b.engine.read_counters()), 'day %d' % d
        print('day %d ok (%.0f s)' % (d, time.time() - t0), flush=True)
for a, b in zip(gpu, cpu):
    for name in ('hot', 'infector', 'n_infected'):
        x = a.engine.alloc.to_host(a.engine.tensors[name]).view(np.uint32); y = np.asarray(b.engine.tensors[name]).view(np.uint32)
        assert np.array_equal(x, y), name
tot = sharding.reduce_counters(gpu)
print('PARITY_SHARDED_100M_OK days=%d' % days)
