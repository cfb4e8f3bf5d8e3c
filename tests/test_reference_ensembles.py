"""The parallel formulation against ENSEMBLES OF THE REAL REFERENCE (tests/golden/ref_ens_*.npz: 128
cythonsim runs per scenario family, recorded by tests/golden/make_ref_ensemble.py in the build
container).  This is the link "HIP engine -> reference" of DESIGN.md section 2: the reference's single
sequential PCG64 stream cannot be replayed in parallel, so the claim is that both engines sample the same
DISTRIBUTION of trajectories; tests/ref_stats.py states the tolerance (means of every age-group series
and scalar at 4.5 sigma with no relative slack, variance ratios, KS on per-run outcomes).

  -m gpu       512 HIP-engine seeds per family, the HUS family at BASELINE configs[1] (1 685 983 agents x
               365 days); plus the NEGATIVE CONTROL: the same comparison with infectiousness_multiplier
               biased by 3 % must fail.
  -m "not gpu" oracle B (bit-identical to the HIP engine, tests/test_parity_gpu.py) on the mini families.

Round 6: the family `turku_astra-zeneca` -- the reference's other deployment (VARIABLE_OVERRIDE_SET=turku, variables.py:10-216):
192 962 agents x 470 days, scenario `astra-zeneca` (the `vaccinate` programme from 2021-03-15), 64 runs of the real reference
(tests/golden/make_turku.py).
"""
import pytest

import ref_stats

N_GPU = 512
N_CPU = 128


@pytest.mark.gpu
@pytest.mark.parametrize('family', ['hus_default', 'mini_default', 'mini_imports', 'mini_kitchen', 'mini_initial', 'turku_astra-zeneca'])
def test_hip_engine_samples_the_reference_distribution(family):
    par, meta = ref_stats.run_parallel_ensemble(family, range(70000, 70000 + N_GPU))
    ref, _ = ref_stats.load_ref(family)
    rep = ref_stats.compare(par, ref, meta)
    ref_stats.assert_same_distribution(rep, meta)
    if family == 'hus_default':
        # the power the comparison has at the BASELINE configuration: cumulative counts are pinned to <= 3 %
        cum = [m for m in rep['means'] if m[0].endswith(' total') and ' all_infected ' in m[0] and int(m[0].split()[1]) >= 180]
        assert cum and max(m[4] for m in cum) <= 0.03, max(m[4] for m in cum)


@pytest.mark.gpu
@pytest.mark.parametrize('what,factor', [('infectiousness_multiplier', 1.03), ('infectiousness_multiplier', 0.97)])
def test_negative_control_a_three_percent_bias_is_detected(what, factor):
    """the comparison can fail: HUS x 365 d with the transmission probability off by 3 %"""
    ref, meta = ref_stats.load_ref('hus_default')
    base = ref_stats.variables_for(meta)
    par, _ = ref_stats.run_parallel_ensemble('hus_default', range(80000, 80000 + 256), variables_patch={what: base[what] * factor})
    rep = ref_stats.compare(par, ref, meta)
    print(ref_stats.tolerance_report(rep, meta))
    assert len(rep['failures']) > 20, len(rep['failures'])
    worst = max(abs(m[1]) for m in rep['means'])
    assert worst > 2 * ref_stats.Z_MAX, worst


@pytest.mark.gpu
def test_eight_shards_against_the_reference_distribution():
    """SURVEY 8 f-4.  HUS x 365 d split over 8 shards (in-process on one GPU), 48 seeds, against the 128 runs of the REAL
    reference with the tolerance of ref_stats -- EVERY series, no exemption: the epidemic curves and age-group cells,
    everything that walks infector links (`r`, `ct_cases_per_day`, detections: stand-in infectors for cross-shard
    infections), and the capacity series at saturation (beds and ICU units are one pool over the shards: round 2, which
    re-divided the free capacity every morning, ran the saturated ICU 3.2 % low and this test exempted it)."""
    par, meta = ref_stats.run_sharded_ensemble('hus_default', range(90000, 90048), 8)
    ref, _ = ref_stats.load_ref('hus_default')
    rep = ref_stats.compare(par, ref, meta)
    print(ref_stats.tolerance_report(rep, meta))
    assert not rep['failures'], rep['failures'][:10]
    zs = {m[0]: m for m in rep['means']}
    for d in ref['ck_days']:
        m = zs.get('day %d in_icu total' % d)
        if m is not None and m[2] > 0.9 * 300:   # saturated in the reference: as full here (1 % at most idle beyond it)
            assert m[3] >= 0.99 * m[2], m


@pytest.mark.slow
@pytest.mark.parametrize('family', ['mini_default', 'mini_imports', 'mini_kitchen', 'mini_initial', 'turku_astra-zeneca'])
def test_oracle_b_samples_the_reference_distribution(family):
    import par_backend
    par, meta = ref_stats.run_parallel_ensemble(family, range(60000, 60000 + N_CPU), engine_factory=par_backend.par_engine_factory)
    ref, _ = ref_stats.load_ref(family)
    ref_stats.assert_same_distribution(ref_stats.compare(par, ref, meta), meta)


def test_reference_ensembles_are_what_the_generator_describes():
    """fixture sanity (no engine): shapes, seeds disjoint from the single-run goldens, conservation in
    every recorded reference run"""
    import numpy as np
    for family in ('hus_default', 'mini_default', 'mini_imports', 'mini_kitchen', 'mini_initial', 'turku_astra-zeneca'):
        z, meta = ref_stats.load_ref(family)
        S, D, _ = z['tot'].shape
        assert S >= (64 if family.startswith('turku') else 128) and D == meta['days'] and len(set(z['seeds'].tolist())) == S and z['seeds'].min() >= 1000
        idx = {n: i for i, n in enumerate(meta['pop13'])}
        n = sum(meta['age_counts'])
        t = z['tot'].astype(np.int64)
        assert np.all(t[..., idx['susceptible']] + t[..., idx['infected']] + t[..., idx['recovered']] + t[..., idx['dead']] == n)
        assert np.array_equal(z['ag_ck'].sum(axis=3), z['tot'][:, z['ck_days']])
        assert np.allclose(z['ag_mean'].sum(axis=2), z['tot'].mean(axis=0))


# ---------------------------------------------------------------------------------------------------------------------
# More power at the headline configuration: 256 further reference-EQUIVALENT runs (oracle A restates cythonsim bit for bit --
# 37 recorded runs -- so its runs with new seeds are runs of the reference; tests/golden/make_oracle_a_ensemble.py).
def _welch(a, b):
    import numpy as np
    ma, mb = a.mean(axis=0), b.mean(axis=0)
    se = np.sqrt(a.var(axis=0, ddof=1) / a.shape[0] + b.var(axis=0, ddof=1) / b.shape[0])
    return ma, mb, se


def _load_oracle_a():
    import json
    import os
    import numpy as np
    z = np.load(os.path.join(ref_stats.GOLDEN, 'oracle_a_ens_hus_default.npz'))
    return z['tot'].astype(np.float64), z['seeds'], json.loads(bytes(z['meta']))


def test_oracle_a_runs_and_the_recorded_cythonsim_runs_are_one_distribution():
    """no engine: the 256 oracle-A runs against the 128 runs of the REAL cythonsim -- every population total on the check
    days at 4.5 sigma, KS on the per-run outcomes; and the fixture's own sanity"""
    import numpy as np
    from scipy import stats
    ref, meta = ref_stats.load_ref('hus_default')
    a, seeds, _ = _load_oracle_a()
    real = ref['tot'].astype(np.float64)
    assert a.shape[1:] == real.shape[1:] and a.shape[0] >= 256
    assert not set(seeds.tolist()) & set(ref['seeds'].tolist())
    idx = {n: i for i, n in enumerate(meta['pop13'])}
    n = sum(meta['age_counts'])
    assert np.all(a[..., idx['susceptible']] + a[..., idx['infected']] + a[..., idx['recovered']] + a[..., idx['dead']] == n)
    ck = ref['ck_days']
    ma, mb, se = _welch(a[:, ck], real[:, ck])
    keep = (ma + mb) / 2 >= 5
    z = np.where(keep & (se > 0), (ma - mb) / np.where(se > 0, se, 1), 0.0)
    assert np.abs(z).max() <= ref_stats.Z_MAX, (np.abs(z).max(), np.unravel_index(np.abs(z).argmax(), z.shape))
    inf = idx['infected']
    for name, f in (('final size', lambda t: t[:, -1, idx['all_infected']]), ('deaths', lambda t: t[:, -1, idx['dead']]),
                    ('detections', lambda t: t[:, -1, idx['all_detected']]), ('peak height', lambda t: t[:, :, inf].max(axis=1)),
                    ('peak day', lambda t: t[:, :, inf].argmax(axis=1))):
        p = stats.ks_2samp(f(a), f(real)).pvalue
        assert p >= 1e-3, (name, p)


@pytest.mark.gpu
def test_hip_engine_against_all_384_reference_equivalent_runs():
    """512 HIP seeds against the 128 recorded cythonsim runs AND the 256 oracle-A runs together: every population total on the
    check days at 4.5 sigma with no slack, and the power that buys -- the tolerance on the cumulative counts in the second half
    of the year is at most 1.7 % for infections, detections and recoveries and 2.6 % for deaths (a few hundred per run), the
    measured difference at most 1 % (round 2 asserted 3 % for the infection count alone, against the 128)."""
    import numpy as np
    ref, meta = ref_stats.load_ref('hus_default')
    a, _, _ = _load_oracle_a()
    pooled = np.concatenate([ref['tot'].astype(np.float64), a])
    par, _ = ref_stats.run_parallel_ensemble('hus_default', range(70000, 70000 + N_GPU))
    hip = par['ag'].sum(axis=3).astype(np.float64)
    ck = ref['ck_days']
    mh, mr, se = _welch(hip[:, ck], pooled[:, ck])
    keep = (mh + mr) / 2 >= 5
    z = np.where(keep & (se > 0), (mh - mr) / np.where(se > 0, se, 1), 0.0)
    worst = np.unravel_index(np.abs(z).argmax(), z.shape)
    print('worst |z| %.2f on day %d, %s; ' % (np.abs(z).max(), ck[worst[0]], meta['pop13'][worst[1]]))
    assert np.abs(z).max() <= ref_stats.Z_MAX, (np.abs(z).max(), ck[worst[0]], meta['pop13'][worst[1]])
    idx = {n: i for i, n in enumerate(meta['pop13'])}
    late = ck >= 180
    for name in ('all_infected', 'all_detected', 'dead', 'recovered'):
        i = idx[name]
        tol = ref_stats.Z_MAX * se[late, i] / mr[late, i]
        diff = np.abs(mh[late, i] - mr[late, i]) / mr[late, i]
        print('%-13s tolerance %.2f-%.2f %%, measured difference at most %.2f %%' % (name, 100 * tol.min(), 100 * tol.max(), 100 * diff.max()))
        assert tol.max() <= (0.026 if name == 'dead' else 0.017), (name, tol.max())
        assert diff.max() <= 0.01, (name, diff.max())
