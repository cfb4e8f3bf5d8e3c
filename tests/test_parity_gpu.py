"""GPU parity tests proper: the HIP engine (through the C ABI, libreina_hip.so) against the CPU
oracle B (oracle/reina_par.c) on the same seeded inputs.  Everything compared is integer or raw
float bits and must match EXACTLY: per-day counter blocks, the final hot words, infector links,
infection counts, onset durations, vaccination days, testing-queue contents (as sets).
"""
import copy

import numpy as np
import pytest

from golden_util import load_run, variables_for
from reina_model_amd import datasets, simulation
from reina_model_amd import engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS

pytestmark = pytest.mark.gpu


def _pair(variables, ages, seed, interventions=None):
    import par_backend
    gpu = simulation.make_context(variables, age_counts=ages, seed=seed, interventions=interventions)
    cpu = simulation.make_context(variables, age_counts=ages, seed=seed, interventions=interventions,
                                  engine_factory=par_backend.par_engine_factory)
    return gpu, cpu


def _assert_state_equal(gpu, cpu):
    tg, tc = gpu.engine.tensors, cpu.engine.tensors
    for name in ('hot', 'infector', 'n_infected', 'vacc_day'):
        a = gpu.engine.alloc.to_host(tg[name]).view(np.uint32)
        b = np.asarray(tc[name]).view(np.uint32)
        assert np.array_equal(a, b), name
    a = gpu.engine.alloc.to_host(tg['onset_days']).view(np.uint32)
    b = np.asarray(tc['onset_days']).view(np.uint32)
    assert np.array_equal(a, b), 'onset_days bits'
    cg = gpu.engine.alloc.to_host(tg['control'])
    cc = np.asarray(tc['control'])
    for l, q in ((2, 'queue0'), (3, 'queue1')):
        assert cg[l] == cc[l], 'queue length'
        qa = np.sort(gpu.engine.alloc.to_host(tg[q])[:cg[l]].view(np.uint32))
        qb = np.sort(np.asarray(tc[q])[:cc[l]].view(np.uint32))
        assert np.array_equal(qa, qb), q
    # infectee lists: same sets per infector (insertion order is free)
    fa = gpu.engine.alloc.to_host(tg['first_infectee'])
    na = gpu.engine.alloc.to_host(tg['next_sibling'])
    fb, nb = np.asarray(tc['first_infectee']), np.asarray(tc['next_sibling'])
    heads = np.nonzero(fb >= 0)[0]
    assert np.array_equal(heads, np.nonzero(fa >= 0)[0])
    for h in heads[:2000]:
        def chain(f, n):
            out, c = [], f[h]
            while c >= 0:
                out.append(c)
                c = n[c]
            return sorted(out)
        assert chain(fa, na) == chain(fb, nb)


def _run_and_compare(variables, ages, seed, days, interventions=None, chunk=None):
    gpu, cpu = _pair(variables, ages, seed, interventions)
    done = 0
    chunk = chunk or days
    while done < days:
        n = min(chunk, days - done)
        hg = gpu.run(n)
        hc = cpu.run(n)
        if not np.array_equal(hg, hc):
            bad = np.nonzero((hg != hc).any(axis=1))[0][0]
            words = np.nonzero(hg[bad] != hc[bad])[0]
            raise AssertionError('first divergence at day %d, counter words %s gpu=%s cpu=%s' % (
                done + bad, words[:8], hg[bad][words[:8]], hc[bad][words[:8]]))
        done += n
    assert np.array_equal(gpu.engine.read_counters(), cpu.engine.read_counters())
    _assert_state_equal(gpu, cpu)
    return gpu, cpu


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_mini_default_scenario(seed):
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    v.update(hospital_beds=12, icu_units=2)
    _run_and_compare(v, datasets.scaled_population(20000), seed, 200)


@pytest.mark.parametrize('name', ['mini_kitchen_s0', 'mini_kitchen_s3'])
def test_mini_kitchen_sink(name):
    """all intervention types: vaccination cursors, new beds/ICU, variant imports, weekly shares,
    masks, p_icu_death_no_beds < 1"""
    _, meta = load_run(name)
    _run_and_compare(variables_for(meta), np.asarray(meta['age_counts']), meta['seed'], meta['days'],
                     interventions=meta['interventions'])


def test_mini_imports_only():
    _, meta = load_run('mini_imports_s1')
    _run_and_compare(variables_for(meta), np.asarray(meta['age_counts']), meta['seed'], meta['days'],
                     interventions=meta['interventions'])


def test_eager_iterate_equals_batched_run():
    """iterate()+generate_state() per day (the reference's calling pattern) == run(days)"""
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    v.update(hospital_beds=12, icu_units=2)
    ages = datasets.scaled_population(20000)
    a = simulation.make_context(v, age_counts=ages, seed=5)
    b = simulation.make_context(v, age_counts=ages, seed=5)
    hist = b.run(60)
    for d in range(60):
        assert np.array_equal(a.engine.read_counters(), hist[d])
        a.iterate()
    assert np.array_equal(a.engine.read_counters(), b.engine.read_counters())


def test_hus_full_population_150_days():
    """BASELINE config 2 population: 1 685 983 agents, default scenario, first wave + saturated
    beds/ICU + start of contact tracing"""
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    _run_and_compare(v, datasets.get_population_for_area(), 0, 150, chunk=50)


def test_ragged_population_sizes():
    """N not a multiple of 4 / of the wave size; single-agent ages"""
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    v.update(hospital_beds=3, icu_units=1)
    for total in (4099, 10007):
        _run_and_compare(v, datasets.scaled_population(total), 11, 120)


def test_conservation_at_scale():
    """size-independent properties on a 20M-agent synthetic population (no oracle run): every day
    susceptible+infected+recovered+dead == N, all_infected == infected+recovered+dead,
    hospitalized == in_ward+in_icu, sum(daily_contacts) == exposed_per_day"""
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    S = 20_000_000 / 1685983
    v.update(hospital_beds=int(2600 * S), icu_units=int(300 * S))
    ivs = []
    for iv in v['interventions']:
        iv = list(iv)
        if iv[0] in ('import-infections', 'import-infections-weekly'):
            iv[2] = int(iv[2] * S)
        ivs.append(iv)
    ages = datasets.scaled_population(20_000_000)
    ctx = simulation.make_context(v, age_counts=ages, seed=1, interventions=ivs)
    hist = ctx.run(120)
    A = eng.MAX_AGES
    N = int(ages.sum())
    def tot(name):
        i = eng.C_NAMES.index(name)
        return hist[:, i * A:(i + 1) * A].sum(axis=1)
    sc = hist[:, eng.C_NR * A:]
    assert np.all(tot('susceptible') + tot('infected') + tot('recovered') + tot('dead') == N)
    assert np.all(tot('all_infected') == tot('infected') + tot('recovered') + tot('dead'))
    assert np.all(tot('hospitalized') == tot('in_ward') + tot('in_icu'))
    assert np.all(sc[:, eng.S_DAILY_CONTACTS:eng.S_DAILY_CONTACTS + 6].sum(axis=1) == sc[:, eng.S_EXPOSED_PER_DAY])
    assert tot('all_infected')[-1] > 100000
    assert np.all(sc[:, eng.S_PROBLEM] == 0)


def test_sharded_population_two_shards_on_one_gpu():
    """the population split over G=2 engine instances (both on this GPU, stepped in
    lock-step with the pressure buffers summed in between) == the same on the CPU oracle."""
    import par_backend
    from reina_model_amd import sharding
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    v.update(hospital_beds=30, icu_units=4)
    ages = datasets.scaled_population(60000)
    G = 2
    gm, cm = [], []
    gpu = [simulation.make_context(v, age_counts=ages, seed=4, comm=sharding.InProcessComm(r, G, gm)) for r in range(G)]
    cpu = [simulation.make_context(v, age_counts=ages, seed=4, comm=sharding.InProcessComm(r, G, cm),
                                   engine_factory=par_backend.par_engine_factory) for r in range(G)]
    for d in range(160):
        sharding.step_shards_together(gpu)
        sharding.step_shards_together(cpu)
        if d % 20 == 19:
            for a, b in zip(gpu, cpu):
                assert np.array_equal(a.engine.read_counters(), b.engine.read_counters()), d
    for a, b in zip(gpu, cpu):
        _assert_state_equal(a, b)
    tot = sharding.reduce_counters(gpu)
    assert tot[eng.C_NAMES.index('all_infected') * eng.MAX_AGES:][:101].sum() > 5000
