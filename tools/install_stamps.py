"""Diagnostic (build with -DREINA_INSTALL_STAMPS): phases of k_install per workgroup role, HUS."""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from reina_model_amd import datasets, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
ctx = simulation.make_context(copy.deepcopy(VARIABLE_DEFAULTS), age_counts=datasets.get_population_for_area(), seed=0)
ctx.run(130); ctx.synchronize()
ctx.engine.tensors['mirror'].zero_()
ctx.run(200); ctx.synchronize()
m = ctx.engine.alloc.to_host(ctx.engine.tensors['mirror']).astype(np.float64) / 100.0
nb = 512 * 200.0   # workgroups per role x days (grid 1024)
for role, off in (('candidates', 0), ('deferred', 8)):
    print('%-10s mean us/workgroup: setup %.2f work %.2f flush %.2f | max us: setup %.1f work %.1f flush %.1f' % (
        role, m[off] / nb, m[off + 1] / nb, m[off + 2] / nb, m[off + 4], m[off + 5], m[off + 6]))
