"""Diagnostic (library built with -DREINA_HOSP_STAMPS -DREINA_INSTALL_STAMPS, see tools/gpu_stamps.sh): where
k_hosp_install spends its time, by window of the scenario.  Stamps accumulate in buffers.mirror (100 MHz ticks):
[1..6] the ordered event walk's phases, [0..2]/[8..10] install_block's phases per role (sum over workgroups),
[4..6]/[12..14] the slowest workgroup."""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from reina_model_amd import simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), n)
ctx = simulation.make_context(v, age_counts=ages, seed=0)
which = sys.argv[2] if len(sys.argv) > 2 else 'hosp'
for lo, hi in ((0, 60), (60, 85), (85, 100), (100, 125), (125, 160), (160, 365)):
    ctx.engine.tensors['mirror'].zero_()
    ctx.run(hi - lo)
    ctx.synchronize()
    m = ctx.engine.alloc.to_host(ctx.engine.tensors['mirror']).astype(np.float64)[:16] / 100.0
    d = hi - lo
    if which == 'hosp':
        names = ['-', 'count+type', 'setup', 'sort', 'scan', 'apply', 'flush']
        print('days %3d-%3d event walk us/day:' % (lo, hi), ' '.join('%s %.1f' % (names[k], m[k] / d) for k in range(1, 7)), 'walks %.1f' % (m[1:7].sum() / d), '| whole block %.1f us/day over %d ordered days = %.1f us per ordered day' % (m[7] / d, int(m[8] * 100), m[7] / max(1.0, m[8] * 100)), flush=True)
    else:
        print('days %3d-%3d install: candidates sum-us/day setup %.0f work %.0f flush %.0f (slowest wg: %.1f %.1f %.1f) | deferred setup %.0f work %.0f flush %.0f (slowest %.1f %.1f %.1f)' % (
            lo, hi, m[0] / d, m[1] / d, m[2] / d, m[4], m[5], m[6], m[8] / d, m[9] / d, m[10] / d, m[12], m[13], m[14]), flush=True)
