"""HUS 1 685 983 agents x 365 days of the default scenario on the CPU oracles, for more reference-equivalent runs than the 128
recorded ones: python tests/hus_a_vs_b.py A|B <first seed> <last seed + 1> writes gpurun_out/hus_ab_<A|B>_<lo>_<hi>.npy
(totals [runs, days, 13]); profiles/r02_hus_oracle_a_vs_b_384_runs.txt is the comparison of 256 runs of A, 258 of B and the
128 recorded runs (about 5 s per run and core)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_stats, par_backend
from oracle import seq_oracle as so
from reina_model_amd import simulation, engine as eng
which, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ref, meta = ref_stats.load_ref('hus_default')
v = ref_stats.variables_for(meta); ages = np.asarray(meta['age_counts']); D = meta['days']
names = meta['pop13']
out = np.zeros((hi - lo, D, 13))
for k, seed in enumerate(range(lo, hi)):
    if which == 'A':
        ctx = so.make_context(v, ages, seed, interventions=meta['interventions'], ipc=meta.get('ipc'))
        for d in range(D):
            c = ctx.counters()
            out[k, d] = [c[n].sum() for n in names]
            ctx.iterate()
    else:
        ctx = simulation.make_context(v, age_counts=ages, seed=seed, interventions=meta['interventions'], ipc=meta.get('ipc'), device='cpu', engine_factory=par_backend.par_engine_factory)
        h = ctx.run(D)
        A_ = eng.MAX_AGES
        for i, n in enumerate(names):
            ci = eng.C_NAMES.index(n); out[k, :, i] = h[:, ci * A_:(ci + 1) * A_].sum(axis=1)
np.save('gpurun_out/hus_ab_%s_%d_%d.npy' % (which, lo, hi), out)
print(which, lo, hi, 'done', flush=True)
