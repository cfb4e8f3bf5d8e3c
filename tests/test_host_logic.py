"""Host-side logic of the product: contact-table builder, intervention conversion, day descriptors,
parameter expansion.  CPU only; the engine is replaced by oracle B where one is needed."""
import copy
import os

import numpy as np
import pytest

import par_backend
from reina_model_amd import contacts, datasets, interventions, model, simulation
from reina_model_amd import engine as eng
from reina_model_amd.variables import VARIABLE_DEFAULTS


def test_contact_tables_equal_pandas_formulation():
    """generate_contact_probabilities (main.pyx:1184-1235) uses pandas groupby-sum (Kahan),
    sort_index, divide, cumsum; the numpy builder must give the same float64 bits."""
    pd = pytest.importorskip('pandas')
    rows = datasets.get_contacts_per_day()
    cm = contacts.ContactMatrix(rows, 101)
    cm.set_mobility_factor((100 - 80) / 100.0, place=5, min_age=0, max_age=70)
    cm.set_mobility_factor(0.95)
    cm.set_mobility_factor(0.0, place=2, min_age=19, max_age=None)
    cm.set_mask_probability(0.8, min_age=65)
    assert cm.init_day() is True and cm.init_day() is False
    t = cm.tables
    df = pd.DataFrame(rows, columns=['place_type', 'participant_age', 'contact_age', 'contacts'])
    for place, mn, mx, f in cm.mobility_factors:
        filt = (df.participant_age >= mn) & (df.participant_age <= mx)
        if place != contacts.PLACE_ALL:
            filt &= df.place_type == contacts.PLACES[place]
        df.loc[filt, 'contacts'] *= float(f)
    tot = df.groupby('participant_age')['contacts'].sum()
    d2 = df.set_index(['place_type', 'participant_age', 'contact_age']).sort_index().unstack('participant_age')
    d2.columns = d2.columns.droplevel(0)
    d2 = d2.divide(tot, axis=1).cumsum()
    assert np.array_equal(tot.values, t.nr_contacts_by_age)
    for age in d2.columns:
        o = t.offset[age]
        for k, ((place, (a, b)), cum) in enumerate(d2[age].to_dict().items()):
            assert contacts.PLACES[t.place[o + k]] == place and t.cmin[o + k] == a and t.cmax[o + k] == b
            assert t.cum_p[o + k] == cum
    assert t.mask_p[t.offset[70]] == np.float32(0.8) and t.mask_p[t.offset[64]] == 0


def test_mobility_factor_is_float32_like_the_reference():
    cm = contacts.ContactMatrix(datasets.get_contacts_per_day(), 101)
    cm.set_mobility_factor((100 - 30) / 100.0)
    assert 1 - float(cm.mobility_factor) == 0.30000001192092896  # value seen in the reference goldens


def test_intervention_tuples_round_trip():
    iv = interventions.iv_tuple_to_obj(['limit-mobility', '2020-03-15', 80, 0, 70, 'other'])
    assert iv.type == 'limit-mobility' and iv.date == '2020-03-15'
    assert iv.get_param_values() == dict(reduction=80, min_age=0, max_age=70, place='other')
    iv = interventions.iv_tuple_to_obj(['wear-masks', '2020-07-01', 80, 65, None, None])
    assert iv.get_param_values() == dict(share_of_contacts=80, min_age=65, max_age=None)
    iv = interventions.iv_tuple_to_obj(['import-infections-weekly', '2020-03-10', 30, 40])
    assert iv.get_param_values() == {'weekly_amount': 30, 'variant_b1.1.7': 40}
    iv = interventions.iv_tuple_to_obj(['test-all-with-symptoms', '2020-02-20'])
    assert iv.get_param_values() == {}
    with pytest.raises(Exception):
        interventions.iv_tuple_to_obj(['no-such-thing', '2020-01-01'])
    with pytest.raises(Exception):
        interventions.iv_tuple_to_obj(['limit-mobility', '2020-01-01', 10, None, None, 'moon'])
    ivs = interventions.get_active_interventions(copy.deepcopy(VARIABLE_DEFAULTS))
    assert len(ivs) == 40 and ivs[0].id == '0'


def test_disease_struct_expansion():
    d, names = model.build_disease_struct(simulation.create_disease_params(VARIABLE_DEFAULTS), 101,
                                          VARIABLE_DEFAULTS['imported_infection_ages'])
    assert names == ['wild-type', 'b1.1.7']
    assert d.p_asymptomatic_infection[0] == np.float32(0.008)  # quirk Q1: 0.8 is treated as a percentage
    assert d.infectiousness_multiplier[1] == np.float32(0.9075)
    assert d.p_susceptibility[0][9] == np.float32(0.34) and d.p_susceptibility[0][10] == np.float32(0.67)
    assert d.p_susceptibility[0][100] == np.float32(1.47)
    assert d.p_severe_given_symptomatic[45] == np.float32(0.0343 / 0.70)
    assert d.infectiousness_over_time[0][10] == np.float32(0.18539)
    assert list(d.import_class_min_age[:5]) == [0, 20, 40, 60, 70]
    assert list(d.import_class_max_age[:5]) == [19, 39, 59, 69, 100]
    assert d.import_class_cum[3] == np.float32(1.0)


def test_day_descriptors_follow_the_schedule():
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    ivs = [['import-infections', '2020-02-18', 7], ['import-infections', '2020-02-18', 3, 'b1.1.7'],
           ['import-infections-weekly', '2020-02-18', 21, 100],
           ['vaccinate', '2020-02-19', 700, 70, None], ['vaccinate', '2020-02-19', 70, None, None],
           ['build-new-hospital-beds', '2020-02-19', 5], ['test-with-contact-tracing', '2020-02-20', 30],
           ['limit-mobility', '2020-02-20', 50]]
    ctx = simulation.make_context(v, age_counts=datasets.scaled_population(5000), seed=1, interventions=ivs,
                                  engine_factory=par_backend.par_engine_factory)
    d0, changed = ctx._build_day()
    ctx.day += 1
    assert not changed
    b = [(x.count, x.variant, x.pre_init) for x in d0.import_batches[:d0.n_import_batches]]
    assert b == [(7, 0, 1), (3, 1, 1), (3, 1, 0)]  # weekly 21/7 = 3/day, all variant share 100 %
    assert d0.testing_mode == model.NO_TESTING and d0.n_vaccinations == 0
    d1, changed = ctx._build_day()
    ctx.day += 1
    assert d1.add_beds == 5 and d1.n_vaccinations == 2
    assert (d1.vaccinations[0].nr, d1.vaccinations[0].slot) == (100, 0)
    assert d1.vaccinations[0].idx_start == ctx.age_start[70] and d1.vaccinations[0].idx_end == ctx.total_people
    assert (d1.vaccinations[1].nr, d1.vaccinations[1].idx_start) == (10, 0)
    d2, changed = ctx._build_day()
    assert changed and d2.testing_mode == model.ALL_WITH_SYMPTOMS_CT
    assert abs(d2.p_successful_tracing - 0.3) < 1e-7


def test_interventions_are_found_by_day_number_whatever_carries_their_date():
    """the schedule is indexed by day number (not by comparing date strings every day): an ISO string and a datetime.date land
    on the same day, list order within a date is kept, a date before the start never applies, a malformed date is an error when the intervention is added"""
    from datetime import date
    from reina_model_amd.interventions import Intervention
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    ctx = simulation.make_context(v, age_counts=datasets.scaled_population(5000), seed=1, interventions=[],
                                  engine_factory=par_backend.par_engine_factory)
    ctx.add_intervention(Intervention('build-new-hospital-beds', date(2020, 2, 20), {'beds': 5}))
    ctx.add_intervention(Intervention('build-new-hospital-beds', '2020-02-20', {'beds': 7}))
    ctx.add_intervention(Intervention('build-new-hospital-beds', '2020-01-01', {'beds': 1000}))   # before the start date
    adds = []
    for _ in range(4):
        d, _ = ctx._build_day()
        ctx.day += 1
        adds.append(d.add_beds)
    assert adds == [0, 0, 12, 0]
    with pytest.raises(ValueError, match='not an ISO date'):   # (refused when it is added, not on the first day of the run)
        ctx.add_intervention(Intervention('build-new-hospital-beds', '20/02/2020', {'beds': 1}))
    # the index follows the list: an intervention added later, or put in place of another one, is found
    ctx.add_intervention(Intervention('build-new-hospital-beds', '2020-02-23', {'beds': 3}))
    ctx.interventions[-1] = Intervention('build-new-hospital-beds', '2020-02-22', {'beds': 4})
    d, _ = ctx._build_day()
    assert d.add_beds == 4


def test_unknown_intervention_type_raises_like_the_reference():
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    ctx = simulation.make_context(v, age_counts=datasets.scaled_population(3000), seed=1, interventions=[],
                                  engine_factory=par_backend.par_engine_factory)

    class Fake:
        type, date = 'limit-mass-gatherings', '2020-02-18'

        def get_param_values(self):
            return {}
    with pytest.raises(Exception):
        ctx.apply_intervention(Fake())
    with pytest.raises(Exception, match='Variant nope not found'):
        ctx.find_variant('nope')


def test_simulate_individuals_frame_contract():
    """(df, adf) shapes/columns of calc/simulation.py:186-190,278-290 with the CPU checker engine."""
    pytest.importorskip('pandas')
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    v['simulation_days'] = 30
    seen = []
    df, adf = simulation.simulate_individuals(v, step_callback=lambda d: seen.append(len(d)) or True,
                                              callback_day_interval=10, age_counts=datasets.scaled_population(8000),
                                              engine_factory=par_backend.par_engine_factory)
    assert list(df.columns) == simulation.POP_ATTRS + simulation.STATE_ATTRS + simulation.EXPOSURES_ATTRS + ['us_per_infected']
    assert len(df) == 30 and str(df.index[0].date()) == '2020-02-18'
    assert adf.shape == (30, 12 * 9)
    assert df['susceptible'].iloc[0] == datasets.scaled_population(8000).sum()
    assert (df['susceptible'] + df['infected'] + df['recovered'] + df['dead'] == df['susceptible'].iloc[0]).all()
    assert len(seen) == 3
    with pytest.raises(simulation.ExecutionInterrupted):
        simulation.simulate_individuals(v, step_callback=lambda d: False, age_counts=datasets.scaled_population(8000),
                                        engine_factory=par_backend.par_engine_factory)


def test_context_sample_matches_reference_distributions():
    """Context.sample() (main.pyx:2047-2101) with the engine's own samplers vs the 10 000-draw
    samples recorded from the reference: same support, mean within 4 standard errors (+2 %)."""
    import os
    from golden_util import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'samples.npz'))
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    ctx = simulation.make_context(v, age_counts=z['age_counts'], seed=4321, interventions=[],
                                  engine_factory=par_backend.par_engine_factory)
    n = 0
    for key in z.files:
        if key == 'age_counts' or key.startswith('chain77'):
            continue
        what, age, sev = key.split('|')
        ref = z[key].astype(np.float64)
        got = ctx.sample(what, int(age), sev or None).astype(np.float64)
        assert got.shape == ref.shape
        se = np.sqrt(ref.var() / len(ref) + got.var() / len(got))
        assert abs(got.mean() - ref.mean()) <= 4 * se + 0.02 * abs(ref.mean()) + 1e-9, (key, got.mean(), ref.mean())
        assert got.min() >= 0 and got.max() <= max(ref.max() * 3, 4)
        if what == 'symptom_severity':
            for s in range(5):
                assert abs((got == s).mean() - (ref == s).mean()) < 0.02, (key, s)
        n += 1
    assert n == 31
    # 'infectiousness' is advertised by the reference (main.pyx:2059,2066-2070) but calls a method
    # Disease does not define; here it returns the (day, val) record array the caller expects
    r = ctx.sample('infectiousness', 30)
    assert len(r) == 200 and r['day'][0] == -100 and abs(r['val'][100] - 0.18539) < 1e-6 and r['val'][0] == 0
    with pytest.raises(Exception):
        ctx.sample('no-such-thing', 30)


def test_scenarios_apply_like_the_reference():
    """scenarios.py:41-53,181-190: extra interventions are appended, the 'looser' scenario halves
    every limit-mobility reduction of the default list."""
    from reina_model_amd import scenarios
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    base = VARIABLE_DEFAULTS['interventions']
    v = scenarios.scenario_variables('default')
    assert [list(i) for i in v['interventions']] == [list(i) for i in base] and v['preset_scenario'] == 'default'
    v = scenarios.scenario_variables('mitigation')
    assert len(v['interventions']) == len(base) + 18
    assert v['interventions'][len(base)] == ['build-new-icu-units', '2020-06-30', 150]
    assert v['interventions'][-1] == ['limit-mobility', '2021-02-15', 0]
    v = scenarios.scenario_variables('hammer-and-dance')
    assert v['interventions'][len(base)] == ['test-with-contact-tracing', '2020-05-01', 30]
    v = scenarios.scenario_variables('looser-restrictions-to-start-with')
    for a, b in zip(base, v['interventions']):
        assert list(b[:2]) == list(a[:2]) and list(b[3:]) == list(a[3:])
        if len(a) > 2:
            assert b[2] == (a[2] // 2 if a[0] == 'limit-mobility' else a[2])
    import pytest
    with pytest.raises(Exception):
        scenarios.scenario_variables('nope')


def test_native_contact_table_builder_equals_numpy_form():
    """reina_build_contact_tables / par_build_contact_tables (csrc/reina_contacts.h, what a Context uses
    when a mobility limitation changes) against the numpy form of generate_contact_probabilities +
    pack_contact_tables: totals, cumulative probabilities, float32 totals and uint32 thresholds bit for
    bit, over stacked / overlapping / zeroing factors (an age whose contacts are all switched off has
    NaN probabilities in the reference: thresholds 0)."""
    from reina_model_amd.contacts import PLACES, ContactMatrix
    from reina_model_amd.model import pack_contact_tables
    build = eng.bind_abi(par_backend.lib(), 'par_')['build_contact_tables']
    A = 101
    ROWS = datasets.get_contacts_per_day()
    rng = np.random.default_rng(5)

    def pair():
        a, b = ContactMatrix(ROWS, A), ContactMatrix(ROWS, A)
        b.native_build, b.pack_ages, b.pack_entries = build, eng.MAX_AGES, eng.MAX_ENTRIES
        return a, b

    def same(a, b):
        ta, tb = a.generate_contact_probabilities(), b.generate_contact_probabilities()
        assert getattr(ta, 'packed', None) is None and tb.packed is not None
        for f in ('nr_contacts_by_age', 'cum_p'):
            assert np.array_equal(getattr(ta, f).view(np.uint64), getattr(tb, f).view(np.uint64)), f
        for f in ('offset', 'count', 'place', 'cmin', 'cmax'):
            assert np.array_equal(getattr(ta, f), getattr(tb, f)), f
        assert np.array_equal(ta.mask_p, tb.mask_p)
        pa, pb = pack_contact_tables(ta, A), pack_contact_tables(tb, A)
        assert np.array_equal(pa[0].view(np.uint32), pb[0].view(np.uint32))       # nr_contacts float32
        for k in (1, 2, 3):
            assert np.array_equal(pa[k], pb[k]), k
        assert pa[4] == pb[4]

    a, b = pair()
    same(a, b)                                                    # no factors
    for cm in (a, b):
        cm.set_mobility_factor(0.5, place=PLACES.index('work'))
        cm.set_mobility_factor(0.8, place=PLACES.index('school'), min_age=7, max_age=18)
        cm.set_mobility_factor(0.95)                              # every place, every age
        cm.set_mask_probability(0.6, place=PLACES.index('transport'), min_age=15)
    same(a, b)
    for cm in (a, b):
        cm.set_mobility_factor(0.0, min_age=80, max_age=100)      # everything off for the oldest: 0 / 0
        cm.set_mobility_factor(0.3, place=PLACES.index('work'))   # replaces the earlier work factor
    same(a, b)
    for _ in range(15):                                           # random stacks
        a, b = pair()
        for _k in range(int(rng.integers(1, 12))):
            lo = int(rng.integers(0, A)); hi = int(rng.integers(lo, A))
            place = None if rng.random() < 0.25 else int(rng.integers(0, len(PLACES)))
            f = float(np.float32(rng.choice([0.0, 0.05, 0.33, 0.5, 0.9, 1.0, 1.7])))
            for cm in (a, b):
                cm.set_mobility_factor(f, place=place, min_age=lo, max_age=hi)
        same(a, b)


def test_import_batches_carry_the_testing_mode_in_force_when_they_were_applied():
    """interventions of one date run in list order (main.pyx:2013-2015); import-infections infects at once, so each
    batch records the mode of that moment (it decides whether the imported agents keep an infectee list)"""
    import copy
    import par_backend
    from reina_model_amd import datasets, engine as eng, simulation
    from reina_model_amd.variables import VARIABLE_DEFAULTS
    v = copy.deepcopy(VARIABLE_DEFAULTS)
    d0 = v['start_date']
    ivs = [['import-infections', d0, 5], ['test-with-contact-tracing', d0, 50], ['import-infections', d0, 7],
           ['test-only-severe-symptoms', d0, 10], ['import-infections', d0, 9]]
    ctx = simulation.make_context(v, age_counts=datasets.scaled_population(5000), seed=1, interventions=ivs,
                                  engine_factory=par_backend.par_engine_factory, device='cpu')
    day, _ = ctx._build_day(None)
    got = [(b.count, b.testing_mode) for b in day.import_batches[:day.n_import_batches] if b.pre_init]
    assert got == [(5, 0), (7, 1), (9, 3)], got      # TestingMode: none 0, tracing 1, all with symptoms 2, only severe 3
    assert day.testing_mode == 3


def test_initial_condition_walk_is_cut_short_like_the_reference_walks_it():
    """recovered_without_illness() = were_incubating() - were_ill() = incubating (calc/datasets.py:120-134), so the slot
    boundaries of set_initial_state (main.pyx:1456-1463) end at 2 * incubating + ill + dead + in_icu + in_ward while the
    walk covers were_incubating() slots: with fewer recovered than incubating people the last categories lose slots"""
    from reina_model_amd import datasets
    ipc = datasets.InitialPopulationCondition(dead=2, in_icu=5, in_ward=7, confirmed_cases=135, incubating=45, ill=11, recovered=33)
    assert ipc.recovered_without_illness() == 45 and ipc.were_incubating() == 103
    seen = {}

    class Eng:
        def set_initial_state(self, ic):
            seen['ic'] = ic
    from reina_model_amd import model
    ctx = model.Context.__new__(model.Context)
    ctx._split = lambda x: x
    ctx.shard_rank, ctx.n_shards, ctx.engine, ctx.beds = 0, 1, Eng(), 10
    ctx._set_initial_state(ipc)
    ic = seen['ic']
    assert ic.were_incubating == 103                      # not 45 + 45 + 11 + 2 + 5 + 7 = 115: ward and ICU slots are lost
    assert ic.incubating + ic.recovered_without_illness + ic.ill + ic.dead == 103
    ipc2 = datasets.InitialPopulationCondition(dead=5, in_icu=4, in_ward=12, incubating=60, ill=45, recovered=300)
    ctx._set_initial_state(ipc2)
    assert seen['ic'].were_incubating == ipc2.were_incubating() == 426


def _reference_frames():
    import os
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'frames_ref.json')) as f:
        return json.load(f)


def test_frames_equal_the_reference_drivers_own_frames():
    """SURVEY 8 row f-3, pinned to the REAL reference: tests/golden/frames_ref.json holds the (df, adf) frames that
    calc.simulation.simulate_individuals itself returned for a 40-day HUS run of the real cythonsim (recorded by
    tests/golden/_harness/make_frames_fixture.py: column order, dtypes, index, the adf MultiIndex layout, every value).
    The same per-day numbers, put back into a counter history, must come out of reina_model_amd.simulation as the same
    frames: labels, order, dtypes, index, element types and values."""
    pd = pytest.importorskip('pandas')
    from datetime import date
    from fractions import Fraction
    from reina_model_amd import engine as eng
    f = _reference_frames()
    days, A = f['days'], eng.MAX_AGES
    ref_cols = f['df']['columns']
    assert ref_cols == simulation.POP_ATTRS + simulation.STATE_ATTRS + simulation.EXPOSURES_ATTRS + ['us_per_infected']
    age_to_group = datasets.make_age_groups(VARIABLE_DEFAULTS['max_age'])
    groups = list(np.unique(age_to_group))
    adf_cols = [tuple(c) for c in f['adf']['columns']]
    assert [g for a, g in adf_cols if a == adf_cols[0][0]] == groups          # the age-group labels and their order
    indices = np.array([groups.index(x) for x in age_to_group])
    first_age = [int(np.nonzero(indices == g)[0][0]) for g in range(len(groups))]
    # the reference's numbers as a counter history: a group's value sits at the group's first age
    hist = np.zeros((days, eng.COUNTER_WORDS), dtype=np.int32)
    adf_vals = np.asarray(f['adf']['values'], dtype=np.int64)
    for k, (attr, g) in enumerate(adf_cols):
        hist[:, eng.C_NAMES.index(attr) * A + first_age[groups.index(g)]] = adf_vals[:, k]
    sc = hist[:, eng.C_NR * A:]
    v = f['df']['values']
    for name, slot in (('exposed_per_day', eng.S_EXPOSED_PER_DAY), ('available_hospital_beds', eng.S_AVAILABLE_BEDS),
                       ('available_icu_units', eng.S_AVAILABLE_ICU), ('total_icu_units', eng.S_ICU_UNITS),
                       ('ct_cases_per_day', eng.S_CT_CASES_PER_DAY)):
        sc[:, slot] = np.asarray(v[name], dtype=np.int64)
    for i, place in enumerate(('home', 'work', 'school', 'transport', 'leisure', 'other')):
        sc[:, eng.S_DAILY_CONTACTS + i] = np.asarray(v['exposures_' + place], dtype=np.int64)
    for d, r in enumerate(v['r']):   # r = total_infections / total_infectors (more than 5 infectors), else 0
        fr = Fraction(r).limit_denominator(100000)
        k = 1 if fr.denominator > 5 else 6
        assert r == 0 or fr.numerator * k / (fr.denominator * k) == r
        sc[d, eng.S_TOTAL_INFECTIONS], sc[d, eng.S_TOTAL_INFECTORS] = (fr.numerator * k, fr.denominator * k) if r else (0, 0)
    mobility = [float(np.float32(1.0 - x)) for x in v['mobility_limitation']]   # (the reference keeps the factor as a C float)
    assert [1 - m for m in mobility] == v['mobility_limitation']

    class Ctx:
        nr_ages = len(age_to_group)
        age_group_labels = groups
        age_group_indices = indices

    df, adf = simulation._frames_from_history(Ctx, hist, mobility, date.fromisoformat(f['start']), ms_per_day=1.0)
    # df: labels, order, dtypes, index
    assert list(df.columns) == ref_cols
    assert [str(t) for t in df.dtypes] == f['df']['dtypes']
    assert type(df.index).__name__ == f['df']['index_type'] and str(df.index.dtype) == f['df']['index_dtype']
    assert df.index.name == f['df']['index_name'] and str(df.index.freqstr) == f['df']['index_freq']
    assert str(df.index[0].date()) == f['start'] and len(df) == days
    for c in ref_cols:
        assert sorted(set(type(x).__name__ for x in df[c].values)) == f['df']['element_types'][c], c
        if c != 'us_per_infected':   # (a timing column)
            assert df[c].tolist() == v[c], c
    # adf: the two-level column index (attr, age_group) in the reference's order, names, dtype, index, values
    assert adf.columns.nlevels == f['adf']['nlevels'] and list(adf.columns.names) == f['adf']['column_names']
    assert [tuple(c) for c in adf.columns] == adf_cols
    assert sorted(set(str(t) for t in adf.dtypes)) == f['adf']['dtypes']
    assert type(adf.index).__name__ == f['adf']['index_type'] and adf.index.name == f['adf']['index_name']
    assert str(adf.index.dtype) == f['adf']['index_dtype']
    assert np.array_equal(adf.values, adf_vals)
    # ... and the day table `python -m calc.simulation` prints: header line and rows, character for character
    assert simulation.table_header() == f['printed']['header']
    for d, line in enumerate(f['printed']['rows']):
        ours = simulation.table_row(df.iloc[d])
        if df['infected'].iloc[d]:   # the line ends with the timing column: same text up to it, then a number in its own width
            cut = len(ours) - 13
            assert ours[:cut] == line[:cut] and float(line[cut:]) >= 0, (d, ours, line)
        else:
            assert ours == line, (d, ours, line)


def test_turku_override_set_equals_the_references():
    """variables.py:10-216 as the reference's module holds it (tests/golden/turku_inputs.json, recorded by make_turku.py with
    VARIABLE_OVERRIDE_SET=turku): every variable of the merged defaults, the scenario ids and their add_interventions"""
    import json
    import os
    from reina_model_amd import datasets, variables as V
    t = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'turku_inputs.json')))
    v = V.copy_variables(override_set='turku')
    for k, x in t['variable_defaults'].items():
        assert v[k] == x, k
    assert [s['id'] for s in v['scenarios']] == t['scenario_ids']
    assert {s['id']: s.get('add_interventions', []) for s in v['scenarios']} == t['add_interventions']
    assert ['vaccinate', '2021-03-15', 2000, 25, 55] in t['add_interventions']['astra-zeneca']
    assert v['area_name'] == 'Turku' and v['hospital_beds'] == 900 and v['icu_units'] == 55
    assert sum(1 for iv in v['interventions'] if iv[0] == 'test-with-contact-tracing') == 9
    assert list(datasets.get_population_for_area('Turku')) == t['age_counts'] and sum(t['age_counts']) == 192962
    # the defaults themselves are untouched, an unknown set is refused, and 'hus' means "no set"
    assert V.copy_variables()['area_name'] == 'HUS' and V.copy_variables(override_set='hus') == V.copy_variables()
    with pytest.raises(KeyError):
        V.copy_variables(override_set='tampere')
    with pytest.raises(KeyError):
        datasets.get_population_for_area('Tampere')
    # the third area the reference keeps a case file for (AREA_CASEFILES, calc/datasets.py:82-86): a hospital DISTRICT, whose population the
    # reference sums over its member municipalities (recorded through its own get_population_for_area by make_turku.py)
    assert int(datasets.get_population_for_area('Varsinais-Suomi').sum()) == 479861
    vs = V.copy_variables(area_name='Varsinais-Suomi', start_date='2020-03-08')
    assert datasets.get_initial_population_condition(vs).confirmed_cases == 1
    # Turku's case file feeds the initial condition (calc/datasets.py:138-173): a listed start date, and one it does not list
    v.update(start_date='2020-09-01', incubating_at_simulation_start=150, ill_at_simulation_start=50, recovered_at_simulation_start=1000)
    ipc = datasets.get_initial_population_condition(v)
    assert (ipc.dead, ipc.in_icu, ipc.in_ward, ipc.confirmed_cases, ipc.incubating, ipc.ill, ipc.recovered) == (7, 0, 0, 479, 150, 50, 1000)
    v['start_date'] = '2020-02-18'
    assert not datasets.get_initial_population_condition(v).has_initial_state()


def test_override_set_is_selected_by_the_environment_like_the_reference():
    """variables.py:218-220, :436-437: VARIABLE_OVERRIDE_SET picks the process's defaults at import time"""
    import subprocess
    import sys
    code = ('import reina_model_amd.variables as V; v = V.copy_variables(); '
            'print(v["area_name"], v["hospital_beds"], len(v["scenarios"]), V.copy_variables(override_set="hus")["area_name"])')
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, VARIABLE_OVERRIDE_SET='turku'), capture_output=True, text=True,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ['Turku', '900', '3', 'HUS']
    bad = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, VARIABLE_OVERRIDE_SET='nowhere'), capture_output=True, text=True,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert bad.returncode != 0 and 'AssertionError' in bad.stderr


def test_reference_shaped_variable_accessors():
    """variables.py:452-536 outside a request context: get_variable (store, override, default; lists copied), set_variable only
    inside allow_set_variable(), reset_variable(s), and copy_variables() seeing the overrides (round-5 advisor: scripts written
    against the reference's module)"""
    from reina_model_amd import variables as V
    assert V.get_variable('hospital_beds') == 2600
    assert V.get_variable('hospital_beds', var_store={'hospital_beds': 12}) == 12
    ivs = V.get_variable('interventions')
    ivs.append('x')
    assert 'x' not in V.get_variable('interventions')
    with pytest.raises(Exception, match='outside of request context'):
        V.set_variable('hospital_beds', 10)
    with V.allow_set_variable():
        V.set_variable('hospital_beds', 10)
        with pytest.raises(AssertionError):
            V.set_variable('hospital_beds', 'ten')
        with pytest.raises(AssertionError):
            V.set_variable('no_such_variable', 1)
    try:
        assert V.get_variable('hospital_beds') == 10 and V.copy_variables()['hospital_beds'] == 10
        assert V.copy_variables(override_set='hus')['hospital_beds'] == 2600
        V.reset_variable('hospital_beds')
        assert V.get_variable('hospital_beds') == 2600
        with V.allow_set_variable():
            V.set_variable('icu_units', 1)
    finally:
        V.reset_variables()
    assert V.get_variable('icu_units') == 300
