"""Diagnostic (-DREINA_INSTALL_STAMPS): k_hosp_install's phases for member 0 of an engine group of K HUS members
(python tools/stamps_install_group.py [K])"""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from reina_model_amd import datasets, ensemble, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS
K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
v = copy.deepcopy(VARIABLE_DEFAULTS)
ages = datasets.get_population_for_area()
planner = simulation.make_context(v, age_counts=ages, seed=0)
members = [simulation.make_context(v, age_counts=ages, seed=100 + k) for k in range(K)]
for lo, hi in ((0, 60), (60, 125), (125, 365)):
    plan = planner.make_plan(hi - lo)
    for m in members[:2]:
        m.engine.tensors['mirror'].zero_()
    ensemble.run_group_plan(members, plan)
    m = members[0].engine.alloc.to_host(members[0].engine.tensors['mirror']).astype(np.float64)[:32] / 100.0
    d = hi - lo
    print('days %3d-%3d member 0: installing workgroups sum/day prologue %.1f work %.1f flush %.1f, slowest ever %.1f %.1f %.1f | events-only workgroup: %.1f %.1f %.1f, slowest %.1f %.1f %.1f' % (
        lo, hi, m[0] / d, m[1] / d, m[2] / d, m[4], m[5], m[6], m[8] / d, m[9] / d, m[10] / d, m[12], m[13], m[14]), flush=True)
