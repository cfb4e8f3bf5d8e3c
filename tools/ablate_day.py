"""Where does k_day spend a peak day?  Diagnostic build (-DREINA_ABLATE, REINA_HIP_LIB): the scenario runs normally to
day 92, then parts of k_day are switched off (timing only -- the state that follows is meaningless) and days 92-98 are
timed.  python tools/ablate_day.py <bits> [agents]: 1 no target resolution, 2 no normal / exp in the count draw,
4 no contact sampling at all,
8 no clearing of ACTIVE bits in the plane, 16 no list stores in the state-machine rounds, 32 no store of changed hot words,
64 no store of onset_days.  ABLATE_START=<day> (default 92), ABLATE_PROF=<lib built with -DREINA_DAY_PROF too>: per-wave cycles"""
import copy, ctypes, os, sys
sys.path.insert(0, os.getcwd())
import bench
from reina_model_amd import simulation, engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS
bits = int(sys.argv[1])
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), n)
ctx = simulation.make_context(v, age_counts=ages, seed=0)
start = int(os.environ.get('ABLATE_START', '92'))
ctx.run(start, record_history=False)
ctx.synchronize()
lib = eng.load_hip_library()
lib.reina_debug_ablate.argtypes = [ctypes.c_uint32]
assert lib.reina_debug_ablate(bits) == 0
ctx.engine.profile_enable(1)
ctx.engine.profile_read_kernels()
ctx.run(3, record_history=False)
ctx.synchronize()
prof = ctx.engine.profile_read_kernels()
print('ablate %d day %d: ' % (bits, start) + ' '.join('%s %.1f' % (k, ms * 1000 / c) for k, (ms, c) in prof.items() if c), flush=True)
