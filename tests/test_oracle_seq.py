"""Pins the sequential CPU oracle (oracle/reina_seq.c, "A") against vectors recorded from the REAL
reference cythonsim in the build container (tests/golden/make_golden.py).  Bit-exact everywhere:
integer histograms, float scalars (`r`, `mobility_limitation`), RNG doubles.
"""
import os

import numpy as np
import pytest

from golden_util import GOLDEN, compare_day, load_run, variables_for
from oracle import seq_oracle as so

POP13 = ['susceptible', 'vaccinated', 'infected', 'all_infected', 'detected', 'all_detected', 'in_icu', 'cum_icu', 'in_ward',
         'dead', 'recovered', 'non_hospital_deaths', 'new_infections']


def _run_and_compare(name, max_days=None):
    z, meta = load_run(name)
    ctx = so.make_context(variables_for(meta), meta['age_counts'], meta['seed'],
                          interventions=meta['interventions'], ipc=meta.get('ipc'))
    days = meta['days'] if max_days is None else min(max_days, meta['days'])
    for d in range(days):
        compare_day(ctx.generate_state(), z, meta, d)
        ctx.iterate()
    if days == meta['days']:
        for i, w in enumerate(('dead', 'all_infected', 'all_detected')):
            assert np.array_equal(ctx.get_population_stats(w), z['per_age_final'][i])


@pytest.mark.parametrize('name', ['mini_default_s%d' % s for s in range(8)])
def test_mini_default_scenario_bit_exact(name):
    """HUS default interventions (testing modes, contact tracing, masks, mobility, imports) on a
    20k-agent population with binding bed/ICU capacity."""
    _run_and_compare(name)


@pytest.mark.parametrize('name', ['mini_imports_s%d' % s for s in range(4)])
def test_mini_imports_only_bit_exact(name):
    """Transmission isolated: imports only, no testing / mobility / masks."""
    _run_and_compare(name)


@pytest.mark.parametrize('name', ['mini_kitchen_s%d' % s for s in range(6)])
def test_mini_kitchen_sink_bit_exact(name):
    """Every intervention type incl. vaccination, new beds/ICU, variant imports, weekly variant
    shares, p_icu_death_no_beds < 1 (ICU accounting drift, quirk Q7)."""
    _run_and_compare(name)


@pytest.mark.parametrize('name', ['mini_initial_s%d' % s for s in range(4)] + ['mini_initial_full_s0', 'mini_initial_full_s1'])
def test_initial_population_condition_bit_exact(name):
    """Population.set_initial_state (main.pyx:1452-1516): people incubating / ill / in ward / in ICU
    / dead / recovered at the start, confirmed cases spread over ages; the `_full` runs exhaust
    beds and ICU units during the initial hospitalisations."""
    _run_and_compare(name)


@pytest.mark.parametrize('name', ['mini_initial_short_s0', 'mini_initial_short_s1'])
def test_initial_condition_walk_cut_short_bit_exact(name):
    """fewer recovered (33) than incubating (45) people: the boundaries of set_initial_state end at 2 * 45 + 11 + 2 + 5 + 7
    = 115 while the walk covers were_incubating() = 103 slots -- nobody starts in ward or ICU (recorded in round 2)"""
    z, meta = load_run(name)
    assert z['pop'][0].sum(axis=1)[POP13.index('in_icu')] == 0 and z['pop'][0].sum(axis=1)[POP13.index('in_ward')] == 0
    assert z['pop'][0].sum(axis=1)[POP13.index('all_infected')] == 103
    _run_and_compare(name)


@pytest.mark.parametrize('name', ['mini_order_import_first_s0', 'mini_order_import_first_s1',
                                  'mini_order_tracing_first_s0', 'mini_order_tracing_first_s1'])
def test_same_day_interventions_run_in_list_order_bit_exact(name):
    """imports listed before / after a contact-tracing intervention of the same date (main.pyx:2013-2015: list order;
    import-infections infects at once, so only imports AFTER the tracing intervention keep an infectee list)"""
    _run_and_compare(name)


def test_initial_condition_with_icu_patients_and_no_beds_is_refused():
    """what the real reference does (build container, tests/golden/make_golden.py scenario `mini_initial_nobeds`):
    AssertionError out of Context.__init__ -- set_initial_state main.pyx:1495 -> person_transfer_to_icu :350 ->
    Population.transfer_to_icu :1603 `assert person.state == PersonState.HOSPITALIZED`, the agent having been refused a bed.
    No run could be recorded; the oracle and the product refuse the configuration the same way."""
    import copy
    import par_backend
    from reina_model_amd import datasets, simulation
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    v.update(hospital_beds=0, icu_units=2)
    ages = datasets.scaled_population(20000)
    ipc = dict(dead=3, in_icu=5, in_ward=7, confirmed_cases=90, incubating=30, ill=20, recovered=100)
    with pytest.raises(AssertionError):
        so.make_context(v, ages, 50, ipc=ipc)
    with pytest.raises(AssertionError):
        simulation.make_context(v, age_counts=ages, seed=50, ipc=ipc, engine_factory=par_backend.par_engine_factory, device='cpu')
    # no ICU slot left by the shortened walk, or no ICU patients at all: constructs (ward patients without a bed die or recover)
    for ok in (dict(ipc, in_icu=0), dict(dead=1, in_icu=5, in_ward=7, incubating=45, ill=2, recovered=0)):
        so.make_context(v, ages, 50, ipc=ok)
        simulation.make_context(v, age_counts=ages, seed=50, ipc=ok, engine_factory=par_backend.par_engine_factory, device='cpu')


def test_hus_initial_condition_first_40_days_bit_exact():
    _run_and_compare('hus_initial_s5', max_days=40)


@pytest.mark.slow
def test_hus_default_seed0_first_150_days_bit_exact():
    """Full HUS population (1 685 983 agents), default scenario; 150 days covers the first wave,
    bed/ICU saturation and the start of contact tracing (day 118)."""
    _run_and_compare('hus_default_s0', max_days=150)


def test_hus_age_structure_fixture():
    from reina_model_amd import datasets
    _, meta = load_run('hus_default_s0')
    assert sum(meta['age_counts']) == 1685983
    assert list(datasets.get_population_for_area()) == meta['age_counts']


# ---- RandomPool known answers (simrandom.pyx:13-55) ----

@pytest.fixture(scope='module')
def kat():
    return np.load(os.path.join(GOLDEN, 'rng_kat.npz'))


@pytest.mark.parametrize('seed', [0, 1, 4321])
def test_pcg64_double_uint32_and_halfword_buffering(kat, seed):
    assert np.array_equal(so.rng_pattern(seed, 'd' * 1000), kat['s%d_double' % seed])
    assert np.array_equal(so.rng_pattern(seed, 'u' * 1001), kat['s%d_uint32' % seed])
    pat = bytes(kat['mixed_pattern']).decode()
    assert np.array_equal(so.rng_pattern(seed, pat), kat['s%d_mixed' % seed])


@pytest.mark.parametrize('seed', [0, 1, 4321])
def test_lognormal_and_gamma_known_answers(kat, seed):
    got = so.rng_pattern(seed, 'l' * 20000, 0.0, 0.5)
    assert np.array_equal(got, kat['s%d_lognormal_0_0.5' % seed])
    for mu, cv in ((5.1, 0.86), (21.0, 0.45), (18.8, 0.45)):
        got = so.rng_pattern(seed, 'g' * 20000, mu, cv)
        assert np.array_equal(got, kat['s%d_gamma_%g_%g' % (seed, mu, cv)]), (mu, cv)
    pat = bytes(kat['interleaved_pattern']).decode()
    got = so.rng_pattern(seed, pat, 5.1, 0.86)
    assert np.array_equal(got, kat['s%d_interleaved_5.1_0.86' % seed])


@pytest.mark.parametrize('seed', [0, 1, 4321])
def test_legacy_shuffle_is_frozen(kat, seed):
    assert np.array_equal(so.legacy_shuffled_indices(seed, 1000), kat['s%d_legacy_shuffle_1000' % seed])


# ---- Context.sample() (main.pyx:2047-2101) ----

def test_sample_draws_match_reference():
    z = np.load(os.path.join(GOLDEN, 'samples.npz'))
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    import copy
    ages = z['age_counts']
    n = 0
    for key in z.files:
        if key == 'age_counts' or key.startswith('chain77'):
            continue
        what, age, sev = key.split('|')
        ctx = so.make_context(copy.deepcopy(VARIABLE_DEFAULTS), ages, 4321, interventions=[])
        got = ctx.sample(what, int(age), sev or None)
        assert np.array_equal(got, z[key]), key
        n += 1
    assert n == 31
    ctx = so.make_context(copy.deepcopy(VARIABLE_DEFAULTS), ages, 77, interventions=[])
    assert np.array_equal(ctx.sample('contacts_per_day', 33), z['chain77|contacts_per_day|33'])
    assert np.array_equal(ctx.sample('incubation_period', 33), z['chain77|incubation_period|33'])


def test_cli_day_table_matches_the_recorded_run(capsys):
    """BASELINE configs[0] plumbing: `python -m oracle.cli --check` prints the reference's day table from
    oracle A and verifies every printed day against the recorded cythonsim run"""
    from oracle import cli
    cli.main(['--days', '25', '--seed', '1', '--check'])
    out = capsys.readouterr().out
    assert 'all 25 days identical' in out and out.count('\n') >= 27


TURKU_RUNS = (['turku_default_s%d' % s for s in range(3)] + ['turku_astra-zeneca_s%d' % s for s in range(3)] +
              ['turku_stop-wearing-masks_s%d' % s for s in range(2)] + ['turku_autumn_s%d' % s for s in range(2)])


@pytest.mark.parametrize('name', TURKU_RUNS)
def test_turku_override_set_bit_exact(name):
    """The reference's other deployment (variables.py:10-216, VARIABLE_OVERRIDE_SET=turku; recorded by tests/golden/make_turku.py
    through the reference's own get_population_for_area / get_initial_population_condition / get_active_interventions): Turku's
    192 962 agents over 470 days -- nine contact-tracing steps, place-specific mask ladders, weekly imports with a growing
    variant share, and in the `astra-zeneca` scenario the `vaccinate` programme from 2021-03-15; `autumn`: the start date the set
    keeps commented out (2020-09-01), whose initial condition comes from the rows of Turku's case file."""
    z, meta = load_run(name)
    # the interventions the package derives from its own variables are the ones the reference applied
    from reina_model_amd import interventions as ivs
    from reina_model_amd import datasets
    v = variables_for(meta)
    mine = [iv.make_iv_tuple() for iv in ivs.get_active_interventions(v)]
    assert mine == meta['interventions']
    assert list(datasets.get_population_for_area(v['area_name'])) == meta['age_counts']
    ipc = datasets.get_initial_population_condition(v)
    if meta['ipc'] is None:
        assert not ipc.has_initial_state()
    else:
        assert {k: int(getattr(ipc, k)) for k in meta['ipc']} == meta['ipc']
    if 'astra' in name or 'autumn' in name:
        assert z['pop'][-1, POP13.index('vaccinated')].sum() > 10000
    _run_and_compare(name, max_days=200 if name.endswith('_s2') else None)   # (the third seeds: through the first autumn)
