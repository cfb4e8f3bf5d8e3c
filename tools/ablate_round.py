"""What do a state-machine round's stores cost a wave of k_day?  Diagnostic build (-DREINA_ABLATE -DREINA_DAY_PROF=12: the
wait for everything a round left in flight is timed), the scenario runs normally to the given day, then single days are run
with parts of the round switched off (tools/ablate_day.py lists the bits; the state that follows is meaningless).
python tools/ablate_round.py [agents] [day] [bits ...]"""
import copy, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from reina_model_amd import simulation, engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
day = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bits = [int(x) for x in sys.argv[3:]] or [0, 8, 16, 32, 56, 0]
v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), n)
ctx = simulation.make_context(v, age_counts=ages, seed=0)
ctx.run(day, record_history=False)
ctx.synchronize()
lib = eng.load_hip_library()
lib.reina_debug_ablate.argtypes = [ctypes.c_uint32]
for b in bits:
    assert lib.reina_debug_ablate(b) == 0
    ctx.engine.tensors['mirror'].zero_()
    ctx.engine.profile_enable(1)
    ctx.engine.profile_read_kernels()
    ctx.run(1, record_history=False)
    ctx.synchronize()
    k = ctx.engine.profile_read_kernels()
    ctx.engine.profile_enable(0)
    m = ctx.engine.alloc.to_host(ctx.engine.tensors['mirror']).view(np.uint64).astype(np.float64)
    rows = m[64:].reshape(-1, 4)
    rows = rows[rows[:, 3] > 0]
    waves = max(1.0, float(len(rows)))
    print('ablate %3d: k_day %.1f us k_hosp_install %.1f us; per wave: loop %.1f kcycles, timed part %.1f kcycles in %.1f pieces' % (
        b, k['k_day'][0] * 1000.0, k['k_hosp_install'][0] * 1000.0, rows[:, 0].sum() / waves / 1000.0, rows[:, 1].sum() / waves / 1000.0, rows[:, 2].sum() / waves), flush=True)
