"""BASELINE configs[0] plumbing: the reference's `python -m calc.simulation` day table, produced by the
sequential CPU oracle A (TEST INFRASTRUCTURE -- no GPU, not part of the product).

    python -m oracle.cli [--days 180] [--seed 0] [--check]

--check compares every printed day with the vectors recorded from the real cythonsim
(tests/golden/hus_default_s<seed>.npz) and fails on the first difference.
"""
import argparse
import copy
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    from oracle import seq_oracle as so
    from reina_model_amd import datasets
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    ap = argparse.ArgumentParser(prog='python -m oracle.cli')
    ap.add_argument('--days', type=int, default=180)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--check', action='store_true')
    a = ap.parse_args(argv)
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    ctx = so.make_context(v, datasets.get_population_for_area(), a.seed)
    z = meta = None
    if a.check:
        z = np.load(os.path.join(ROOT, 'tests', 'golden', 'hus_default_s%d.npz' % a.seed))
        meta = json.loads(bytes(z['meta']))
    attrs = ['susceptible', 'infected', 'all_infected', 'detected', 'all_detected', 'in_ward', 'in_icu', 'dead', 'recovered']
    print('%-12s' % 'day' + ''.join('%13s' % x for x in attrs) + '%8s' % 'r')
    for d in range(a.days):
        s = ctx.generate_state()
        if z is not None:
            for i, k in enumerate(meta['pop13']):
                if not np.array_equal(np.asarray(s[k]), z['pop'][d, i]):
                    raise SystemExit('day %d: %s differs from the recorded cythonsim run' % (d, k))
        print('%-12s' % ctx.get_date_for_today() + ''.join('%13d' % int(np.sum(s[x])) for x in attrs) + '%8.2f' % s['r'])
        ctx.iterate()
    if z is not None:
        print('all %d days identical to tests/golden/hus_default_s%d.npz' % (a.days, a.seed))


if __name__ == '__main__':
    main()
