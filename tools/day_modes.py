"""Per-day kernel times of one scenario with k_day forced dense, forced sparse, and choosing by itself (round 4).
usage: python tools/day_modes.py [agents] [days] [modes...]    (modes: dense sparse auto alternate tickets imports_open; default: dense sparse auto)
Each day is stepped alone and every kernel of it timed (HIP events, profile stride 1), so the times carry the event
overhead of a fully timed day (a few us) -- the comparison between modes is what this is for: from which share of active
agents on does streaming the hot words beat fetching the active ones?  (Measured, rounds 4-5: never where the sparse form
is possible -- the launch picks the form by population size alone.)"""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def run(agents, days, mode):
    import numpy as np
    import bench
    from reina_model_amd import engine as eng
    from reina_model_amd import simulation
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    for k in ('REINA_DAY_MODE', 'REINA_DAY_FLAGS', 'REINA_OPEN_TICKETS', 'REINA_IMPORTS_IN_OPEN'):
        os.environ.pop(k, None)
    if mode == 'tickets':
        os.environ['REINA_OPEN_TICKETS'] = '1'
    if mode == 'imports_open':
        os.environ['REINA_IMPORTS_IN_OPEN'] = '1' 
    if mode in ('dense', 'sparse', 'alternate'):
        os.environ['REINA_DAY_MODE'] = mode
    v, ages = bench.scaled_scenario(copy.deepcopy(VARIABLE_DEFAULTS), agents)
    ctx = simulation.make_context(v, age_counts=ages, seed=0)
    ctx.engine.profile_enable(1)
    rows = []
    A = eng.MAX_AGES
    for d in range(days):
        h = ctx.run(1, record_history=True)
        k = ctx.engine.profile_read_kernels()
        infected = int(h[0][0:A].sum())
        rows.append(dict(day=d, infected=infected, **{n: round(ms * 1000.0, 2) for n, (ms, c) in k.items() if c}))
    final = ctx.engine.read_counters()
    return rows, final


if __name__ == '__main__':
    import numpy as np
    agents = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    days = int(sys.argv[2]) if len(sys.argv) > 2 else 365
    modes = sys.argv[3:] or ['dense', 'sparse', 'auto']
    out, finals = {}, {}
    for m in modes:
        out[m], finals[m] = run(agents, days, m)
    ref = finals[modes[0]]
    for m in modes[1:]:
        assert np.array_equal(ref, finals[m]), 'mode %s ends in a different state' % m
    print('# agents %d, us per kernel per day; columns per mode: k_day k_hosp_install k_open' % agents)
    print('day infected(at open) ' + ' '.join('%s' % m for m in modes))
    for d in range(days):
        cells = []
        for m in modes:
            r = out[m][d]
            cells.append('%7.1f %6.1f %6.1f' % (r.get('k_day', 0), r.get('k_hosp_install', 0), r.get('k_open', 0)))
        print('%3d %9d | %s' % (d, out[modes[0]][d]['infected'], ' | '.join(cells)))
    for m in modes:
        tot = {}
        for r in out[m]:
            for k, x in r.items():
                if k not in ('day', 'infected'):
                    tot[k] = tot.get(k, 0.0) + x
        print('# mean us/day %-9s %s  sum %.1f' % (m, ' '.join('%s %.1f' % (k, x / days) for k, x in sorted(tot.items())), sum(tot.values()) / days))
        print('# max  us     %-9s %s' % (m, ' '.join('%s %.1f' % (k, max(r.get(k, 0.0) for r in out[m])) for k in sorted(tot))))
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'day_modes_%d.json' % agents), 'w'))
