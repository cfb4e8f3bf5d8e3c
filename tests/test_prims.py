"""Numeric primitives shared by the HIP kernels and oracle B (reina_model_amd/csrc/reina_prims.h),
exercised through oracle B's test hooks on the CPU build: Philox known answers (Random123
kat_vectors), exp/log accuracy, inverse-normal accuracy, gamma moments, and the contact-count
sampler against the reference's own `Context.sample('contacts_per_day')` draws."""
import ctypes
import os

import numpy as np
import pytest

import par_backend
from golden_util import GOLDEN

vp = ctypes.c_void_p


@pytest.fixture(scope='module')
def L():
    lib = par_backend.lib()
    lib.par_test_philox.argtypes = [vp, vp, vp]
    for n in ('par_test_expf', 'par_test_logf', 'par_test_normal'):
        getattr(lib, n).argtypes = [vp, vp, ctypes.c_int]
    lib.par_test_gamma.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint32,
                                   ctypes.c_uint32, vp, ctypes.c_int]
    lib.par_test_nr_contacts.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_int, vp, ctypes.c_int]
    return lib


def test_philox4x32_10_known_answers(L):
    def ph(key, ctr):
        k = np.array(key, dtype=np.uint32)
        c = np.array(ctr, dtype=np.uint32)
        o = np.zeros(4, dtype=np.uint32)
        L.par_test_philox(k.ctypes.data, c.ctypes.data, o.ctypes.data)
        return [int(x) for x in o]
    assert ph([0, 0], [0, 0, 0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ph([0xffffffff] * 2, [0xffffffff] * 4) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ph([0xa4093822, 0x299f31d0], [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_philox2x32_10_known_answers(L):
    """Random123 kat_vectors, philox2x32 10: (ctr0 ctr1 key) -> (out0 out1)"""
    L.par_test_philox2.argtypes = [vp, vp, vp]

    def ph(key, ctr):
        k = np.array([key], dtype=np.uint32)
        c = np.array(ctr, dtype=np.uint32)
        o = np.zeros(2, dtype=np.uint32)
        L.par_test_philox2(k.ctypes.data, c.ctypes.data, o.ctypes.data)
        return [int(x) for x in o]
    assert ph(0, [0, 0]) == [0xff1dae59, 0x6cd10df2]
    assert ph(0xffffffff, [0xffffffff, 0xffffffff]) == [0x2c3f628b, 0xab4fd7ad]
    assert ph(0x13198a2e, [0x243f6a88, 0x85a308d3]) == [0xdd7ce038, 0xf62a4c12]


def test_the_abi_test_hook_evaluates_the_same_primitives():
    """include/reina_hip.h: reina_test_prims, host build (par_test_prims) -- the device build answers the same records in
    tests/test_prims_gpu.py"""
    from reina_model_amd import engine as eng
    f = eng.bind_abi(par_backend.lib(), 'par_')
    assert [int(x) for x in eng.test_prims(f, 'philox4', [[0xa4093822, 0x299f31d0, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]])[0]] == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    assert [int(x) for x in eng.test_prims(f, 'philox2', [[0x13198a2e, 0x243f6a88, 0x85a308d3]])[0]] == [0xdd7ce038, 0xf62a4c12]
    x = np.linspace(-5, 5, 1001).astype(np.float32)
    y = eng.test_prims(f, 'expf', x.view(np.uint32))[:, 0].view(np.float32)
    assert np.max(np.abs(y - np.exp(x.astype(np.float64))) / np.exp(x.astype(np.float64))) < 2e-7
    bad = np.zeros(4, dtype=np.uint32)
    assert f['test_prims'](99, bad.ctypes.data, 1, bad.ctypes.data) != 0   # unknown primitive: refused


def test_expf_logf_accuracy(L):
    x = np.linspace(-20, 20, 400001).astype(np.float32)
    y = np.zeros_like(x)
    L.par_test_expf(x.ctypes.data, y.ctypes.data, len(x))
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(y - ref) / ref) < 2e-7
    x = np.exp(np.linspace(-40, 40, 400001)).astype(np.float32)
    y = np.zeros_like(x)
    L.par_test_logf(x.ctypes.data, y.ctypes.data, len(x))
    ref = np.log(x.astype(np.float64))
    assert np.max(np.abs(y - ref)) < 4e-6


def test_inverse_normal_accuracy_and_symmetry(L):
    from scipy.special import ndtri
    r = np.linspace(0, 2 ** 32 - 1, 1000001).astype(np.uint64).astype(np.uint32)
    y = np.zeros(len(r), dtype=np.float32)
    L.par_test_normal(r.ctypes.data, y.ctypes.data, len(r))
    ref = ndtri((r.astype(np.float64) + 0.5) / 2 ** 32)
    assert np.max(np.abs(y - ref)) < 2e-6
    rc = (np.uint64(2 ** 32 - 1) - r.astype(np.uint64)).astype(np.uint32)
    yc = np.zeros(len(r), dtype=np.float32)
    L.par_test_normal(rc.ctypes.data, yc.ctypes.data, len(r))
    tails = (ref < -1.98) | (ref > 1.98)
    assert np.array_equal(y[tails], -yc[tails])


@pytest.mark.parametrize('mu,cv', [(5.1, 0.86), (21.0, 0.45), (18.8, 0.45)])
def test_gamma_moments(L, mu, cv):
    g = np.zeros(400000, dtype=np.float32)
    L.par_test_gamma(mu, cv, 99, 3, 3, g.ctypes.data, len(g))
    assert abs(g.mean() - mu) < 0.01 * mu
    assert abs(g.std() / g.mean() - cv) < 0.01
    assert g.min() > 0


def test_gamma_matches_reference_distribution(L):
    """round(gamma(5.1, 0.86)) vs the reference's 10 000 `incubation_period` draws (two-sample,
    chi-square on the day histogram)."""
    z = np.load(os.path.join(GOLDEN, 'samples.npz'))
    ref = z['incubation_period|45|']
    g = np.zeros(200000, dtype=np.float32)
    L.par_test_gamma(5.1, 0.86, 7, 11, 3, g.ctypes.data, len(g))
    mine = (g + 0.5).astype(np.int32)
    bins = np.arange(0, 26)
    h_ref = np.bincount(np.clip(ref, 0, 25), minlength=26)[bins].astype(np.float64)
    h_me = np.bincount(np.clip(mine, 0, 25), minlength=26)[bins].astype(np.float64)
    p = h_me / h_me.sum()
    exp = p * h_ref.sum()
    chi2 = ((h_ref - exp) ** 2 / np.maximum(exp, 1e-9))[exp > 5].sum()
    dof = (exp > 5).sum() - 1
    assert chi2 < dof + 5 * np.sqrt(2 * dof), (chi2, dof)


@pytest.mark.parametrize('age', [5, 25, 45, 65, 85])
def test_contact_count_matches_reference_samples(L, age):
    """nr_contacts sampler vs the reference's `contacts_per_day` sample (main.pyx:1308-1320)."""
    from reina_model_amd import contacts, datasets
    z = np.load(os.path.join(GOLDEN, 'samples.npz'))
    ref = z['contacts_per_day|%d|' % age]
    cm = contacts.ContactMatrix(datasets.get_contacts_per_day(), 101)
    nrc = np.float32(cm.tables.nr_contacts_by_age[age])
    y = np.zeros(200000, dtype=np.int32)
    L.par_test_nr_contacts(5, 1, float(nrc), 1.0, 100, y.ctypes.data, len(y))
    assert abs(y.mean() - ref.mean()) < 4 * ref.std() / np.sqrt(len(ref)) + 0.02
    for q in (10, 50, 90, 99):
        assert abs(np.percentile(y, q) - np.percentile(ref, q)) <= max(1.0, 0.06 * np.percentile(ref, q))


def test_an_age_without_contacts_never_draws_one(L):
    """A count row of an age with nr_contacts_by_age <= 0 is all 0xFFFFFFFF ("never": the reference returns 0 contacts for
    it, main.pyx:1311-1320 with c = 0) and the searches test r >= threshold: the draw r = 0xFFFFFFFF must not pass it
    (round-3 advisor finding: it returned the full limit of 100 contacts, once in 2^32 draws)."""
    L.par_test_count_from_draw.argtypes = [ctypes.c_float, ctypes.c_int, ctypes.c_uint32]
    L.par_test_count_from_draw.restype = ctypes.c_int
    for r in (0, 1, 0x7FFFFFFF, 0xFFFFFFFE, 0xFFFFFFFF):
        assert L.par_test_count_from_draw(0.0, 0, r) == 0
        assert L.par_test_count_from_draw(-1.0, 1, r) == 0
    # an ordinary age: the largest draw gives the largest count the thresholds allow, monotone in the draw
    prev = 0
    for r in (0, 0x40000000, 0x80000000, 0xC0000000, 0xFFFFFFFE, 0xFFFFFFFF):
        n = L.par_test_count_from_draw(11.5, 0, r)
        assert n >= prev
        prev = n
    assert L.par_test_count_from_draw(11.5, 0, 0xFFFFFFFF) == L.par_test_count_from_draw(11.5, 0, 0xFFFFFFFE)


def test_a_chance_is_one_integer_comparison(L):
    """k_day tests every contact's transmission draw against its source's thinning bound as (draw >> 8) < threshold with the
    threshold computed once per source (csrc/reina_prims.h: rp_chance_threshold); oracle B calls rp_chance, the definition
    (RandomPool.chance, simrandom.pyx:32-39).  The two must agree on every probability and every draw: probabilities at and
    around 0 and 1, negative, NaN, infinite, denormal, the neighbours of k x 2^-24 (where the ceiling steps), random bit
    patterns; draws at the threshold, one below, one above and at random."""
    L.par_test_chance_threshold.argtypes = [vp, vp, ctypes.c_int]
    L.par_test_chance_threshold.restype = ctypes.c_int
    rng = np.random.default_rng(11)
    special = np.array([0.0, -0.0, 1.0, -1.0, 2.0, np.nan, np.inf, -np.inf, 1e-45, 1e-38, 5.96e-8, 0.5, 0.25, 1.0 - 2.0 ** -24,
                        np.nextafter(np.float32(1), np.float32(0)), np.nextafter(np.float32(1), np.float32(2))], dtype=np.float32)
    k = rng.integers(0, 1 << 24, 4000).astype(np.float64)
    edges = (k * 2.0 ** -24).astype(np.float32)
    near = np.concatenate([edges, np.nextafter(edges, np.float32(0)), np.nextafter(edges, np.float32(1))])
    rand_bits = rng.integers(0, 1 << 32, 20000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    unit = rng.random(20000).astype(np.float32)
    small = (rng.random(5000) * 1e-6).astype(np.float32)
    ps = np.concatenate([special, near, rand_bits, unit, small]).astype(np.float32)
    # per probability: draws whose 24 bits sit at the step of p x 2^24, and random ones
    with np.errstate(invalid='ignore', over='ignore'):
        step = np.nan_to_num(np.clip(np.ceil(ps.astype(np.float64) * 2.0 ** 24), 0, (1 << 24) - 1), nan=0.0).astype(np.int64)
    draws, probs = [], []
    for d in (-2, -1, 0, 1, 2):
        u = np.clip(step + d, 0, (1 << 24) - 1).astype(np.uint64)
        draws.append(((u << np.uint64(8)) | rng.integers(0, 256, len(u)).astype(np.uint64)).astype(np.uint32))
        probs.append(ps)
    for _ in range(3):
        draws.append(rng.integers(0, 1 << 32, len(ps), dtype=np.uint64).astype(np.uint32))
        probs.append(ps)
    p_bits = np.ascontiguousarray(np.concatenate(probs).view(np.uint32))
    r = np.ascontiguousarray(np.concatenate(draws))
    assert len(p_bits) == len(r) > 400000
    assert L.par_test_chance_threshold(p_bits.ctypes.data_as(vp), r.ctypes.data_as(vp), len(r)) == 0


def test_a_contacts_place_from_its_rows_place_groups(L):
    """k_day takes the PLACE of a contact from at most five comparisons against the thresholds at which the place of the selected
    table entry changes (Tables::grp, derived when the tables are uploaded) and leaves the entry search to the contacts that can
    transmit; oracle B searches the entry for every contact.  The derivation and both look-ups, restated in the oracle library, on
    random rows -- 1 to 96 entries sorted by place, places without entries, entries of probability zero (repeated thresholds) at
    the start, in the middle and at the end, a row that saturates early (trailing 0xFFFFFFFF) -- and on the draws that matter:
    every threshold, its neighbours, 0, 0xFFFFFFFF, random ones.  (The GPU suite holds the library's own derivation against oracle
    B on real tables; a draw of exactly 0xFFFFFFFF never occurs there.)"""
    L.par_test_place_groups.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int]
    L.par_test_place_groups.restype = ctypes.c_int
    rng = np.random.default_rng(23)
    for case in range(400):
        cnt = int(rng.integers(1, 97))
        nplaces = int(rng.integers(1, 7))
        places = np.sort(rng.choice(6, size=nplaces, replace=False))
        place = np.sort(rng.choice(places, size=cnt))                    # sorted by place; some of the chosen places may get no entry
        p = rng.random(cnt)
        p[rng.random(cnt) < (0.0, 0.3, 0.7)[case % 3]] = 0.0               # entries of probability zero
        if case % 5 == 0:
            p[int(rng.integers(0, cnt)):] = 0.0                           # the row saturates early
        if p.sum() == 0.0:
            p[int(rng.integers(0, cnt))] = 1.0
        cum = np.cumsum(p) / p.sum()
        thr = np.clip(np.floor(cum * 4294967296.0), 0, 4294967295.0).astype(np.uint64).astype(np.uint32)
        meta = (place.astype(np.uint32) | (rng.integers(0, 100, cnt).astype(np.uint32) << 8))
        edges = thr.astype(np.int64)
        draws = np.concatenate([edges, edges - 1, edges + 1, [0, 1, 0xFFFFFFFE, 0xFFFFFFFF], rng.integers(0, 1 << 32, 300)])
        draws = np.clip(draws, 0, 0xFFFFFFFF).astype(np.uint32)
        thr = np.ascontiguousarray(thr); meta = np.ascontiguousarray(meta); draws = np.ascontiguousarray(draws)
        bad = L.par_test_place_groups(thr.ctypes.data_as(vp), meta.ctypes.data_as(vp), cnt, draws.ctypes.data_as(vp), len(draws))
        assert bad == 0, (case, cnt, bad)
    # a row whose entries are NOT sorted by place has more than six groups: the library keeps the full search for such tables
    place = np.tile(np.arange(6, dtype=np.uint32), 4)
    thr = np.ascontiguousarray((np.arange(1, 25, dtype=np.uint64) * (1 << 27)).astype(np.uint32))
    draws = np.ascontiguousarray(np.array([5], dtype=np.uint32))
    assert L.par_test_place_groups(thr.ctypes.data_as(vp), np.ascontiguousarray(place).ctypes.data_as(vp), 24, draws.ctypes.data_as(vp), 1) == -1


def test_saturating_maps_compose_like_the_functions_they_stand_for(L):
    """The ordered bed / ICU walk -- and what a sharded population's shards exchange -- rests on one piece of algebra shared by
    the kernels and oracle B (csrc/reina_prims.h): an event acts on a free count as f(x) = max(x + a, m), and two such maps
    compose to one of the same form.  Held here against the plain definition, evaluated in Python integers: random chains of
    admissions (a = -1, m = 0), releases (a = +1, never binds) and composed maps, applied to every free count from 0 to 40;
    composition is associative; a bucket's pair of maps survives its 57-bit packing (round-3 verdict: the algebra was held
    only by the statistical tier)."""
    L.par_test_sat.argtypes = [ctypes.c_int] * 5 + [vp]
    NEG = -(1 << 29)
    out = (ctypes.c_int * 7)()

    def then(f, g):
        L.par_test_sat(f[0], f[1], g[0], g[1], 0, out)
        return (out[0], out[1])

    def apply_c(f, x):
        L.par_test_sat(0, NEG, f[0], f[1], x, out)   # identity, then f
        return out[2]

    ev = {'take': (-1, 0), 'give': (1, NEG)}
    rng = np.random.default_rng(5)
    for _ in range(300):
        chain = [ev['take'] if rng.random() < 0.6 else ev['give'] for _ in range(int(rng.integers(1, 60)))]
        # the chain composed left to right, and in a random bracketing (associativity)
        acc = (0, NEG)
        for f in chain:
            acc = then(acc, f)
        cut = int(rng.integers(0, len(chain) + 1))
        left, right = (0, NEG), (0, NEG)
        for f in chain[:cut]:
            left = then(left, f)
        for f in chain[cut:]:
            right = then(right, f)
        assert then(left, right) == acc
        for x in range(0, 41):
            y = x
            for a, m in chain:        # the definition, event by event: take one if any, give one back
                y = max(y + a, m)
            assert apply_c(acc, x) == y, (chain, x)
        # packing: what a walker publishes / a shard sends for a bucket (|a|, |m| <= 4096, or "never binds")
        if -4096 <= acc[0] <= 4096 and (acc[1] <= NEG // 2 or -4096 <= acc[1] <= 4096):
            L.par_test_sat(acc[0], acc[1], left[0], left[1], 0, out)
            for got, want in (((out[3], out[4]), acc), ((out[5], out[6]), left)):
                assert got[0] == want[0]
                assert got[1] == want[1] or (got[1] <= NEG // 2 and want[1] <= NEG // 2)
