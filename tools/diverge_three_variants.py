"""Where do HIP and oracle B part on the three-variants scenario of the GPU suite?  Steps both a day at a time and compares
counters and state after every day; for each REINA_DAY_MODE given (default: auto dense sparse).
usage: python tools/diverge_three_variants.py [modes...]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_parity_gpu as T
import par_backend
from reina_model_amd import datasets, engine as eng, simulation
from reina_model_amd.variables import VARIABLE_DEFAULTS

v = copy.deepcopy(VARIABLE_DEFAULTS)
v.update(hospital_beds=15, icu_units=3)
v['variants'] = [{'name': 'b1.1.7', 'infectiousness_multiplier': 1.3},
                 {'name': 'p.1', 'infectiousness_multiplier': 1.6, 'mean_incubation_duration': 4.0, 'p_asymptomatic_infection': 50.0}]
ivs = [['import-infections', '2020-02-19', 40], ['import-infections', '2020-02-25', 30, 'b1.1.7'],
       ['import-infections', '2020-03-01', 30, 'p.1'], ['test-all-with-symptoms', '2020-02-22'],
       ['import-infections-weekly', '2020-03-05', 70, 30, 20], ['test-with-contact-tracing', '2020-03-20', 60],
       ['limit-mobility', '2020-03-25', 40], ['import-infections-weekly', '2020-04-20', 35, 0, 100]]
ages = datasets.scaled_population(50000)
A = eng.MAX_AGES
names = {getattr(eng, k): k for k in dir(eng) if k.startswith('S_') and isinstance(getattr(eng, k), int) and k != 'S_NR'}
for mode in (sys.argv[1:] or ["auto", "dense", "sparse"]):
    os.environ.pop('REINA_DAY_MODE', None)
    if mode != 'auto':
        os.environ['REINA_DAY_MODE'] = mode
    gpu = simulation.make_context(v, age_counts=ages, seed=17, interventions=ivs)
    cpu = simulation.make_context(v, age_counts=ages, seed=17, interventions=ivs, engine_factory=par_backend.par_engine_factory)
    print('mode', mode)
    for d in range(130):
        gpu.run(1); cpu.run(1)
        x, y = gpu.engine.read_counters(), cpu.engine.read_counters()
        bad = False
        if not np.array_equal(x, y):
            bad = True
            for i in np.nonzero(x != y)[0][:16]:
                if i < eng.C_NR * A:
                    print('  day %d: %s[age %d] hip %d oracle %d' % (d, eng.C_NAMES[i // A], i % A, x[i], y[i]))
                else:
                    print('  day %d: %s hip %d oracle %d' % (d, names.get(i - eng.C_NR * A, i - eng.C_NR * A), x[i], y[i]))
        tg, tc = gpu.engine.tensors, cpu.engine.tensors
        for name in ('hot', 'infector', 'n_infected'):
            a = gpu.engine.alloc.to_host(tg[name]).view(np.uint32); b = np.asarray(tc[name]).view(np.uint32)
            idx = np.nonzero(a != b)[0]
            if len(idx):
                bad = True
                print('  day %d: %s differs at %d agents' % (d, name, len(idx)))
                for i in idx[:8]:
                    print('     %s[%d]: hip %#x oracle %#x   (hot hip %#x oracle %#x)' % (name, i, a[i], b[i],
                          gpu.engine.alloc.to_host(tg['hot']).view(np.uint32)[i], np.asarray(tc['hot']).view(np.uint32)[i]))
        if bad:
            print('  control[24:28]', gpu.engine.alloc.to_host(tg['control'])[24:28])
            break
    else:
        print('  130 days identical')
