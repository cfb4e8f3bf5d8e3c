"""Drop-in for the reference's `cythonsim.model` module (cythonsim/main.pyx), MI355X-native.

Exports the names `calc/simulation.py` uses -- `Context`, `DISEASE_PARAMS`, `SEVERITY_TO_STR`,
`SimulationFailed`, `__file__` -- with the same constructor signature, `add_intervention`,
`generate_state`, `iterate`, `apply_intervention`, `get_population_stats`,
`get_date_for_today` protocol (main.pyx:1759-1960, 2011-2018).  Agent state lives in HBM as SoA
PyTorch-ROCm tensors; every day step is a handful of hand-written HIP kernels behind the C ABI
of include/reina_hip.h.  Host-level work the reference also does in Python stays here:
intervention dispatch, the contact-table rebuild, weekly-import bookkeeping.

Differences a caller can observe (DESIGN.md "Parity tiers"):
  * agents are stored sorted by age (the reference shuffles agent ids; ids are never exported);
  * random decisions are Philox-keyed per (agent, day, purpose) instead of one sequential PCG64
    stream, so a given seed yields a different but statistically equivalent trajectory;
  * `iterate()` only enqueues GPU work; problems surface at the next `generate_state()` /
    `synchronize()` (the reference raises at the end of `iterate()`, main.pyx:2017-2018).
    `Context(..., strict=True)` (or `ctx.strict = True`) restores the reference's behaviour: every
    `iterate()` waits for its day and raises `SimulationFailed` on the day of the problem.
"""
import ctypes
from datetime import date, timedelta

import numpy as np

from . import engine as _eng
from .contacts import ContactMatrix, PLACES

# main.pyx:777-785
DISEASE_PARAMS = (
    'p_susceptibility', 'p_symptomatic', 'p_severe', 'p_critical',
    'p_fatal', 'p_hospital_death_no_beds', 'p_icu_death_no_beds',
    'p_death_outside_hospital', 'p_asymptomatic_infection',
    'infectiousness_multiplier', 'mean_incubation_duration',
    'mean_duration_from_onset_to_death', 'mean_duration_from_onset_to_recovery',
    'ratio_of_duration_before_hospitalisation', 'ratio_of_duration_in_ward',
    'p_mask_protects_wearer', 'p_mask_protects_others', 'variants',
)
# main.pyx:101-108
SEVERITY_TO_STR = {0: 'ASYMPTOMATIC', 1: 'MILD', 2: 'SEVERE', 3: 'CRITICAL', 4: 'FATAL'}
STR_TO_SEVERITY = {v: k for k, v in SEVERITY_TO_STR.items()}
# main.pyx:110-121 (+ this engine's capacity overflows, include/reina_hip.h)
PROBLEM_TO_STR = {
    0: 'No problemos', 1: 'Too many infectees', 2: 'Too many contacts',
    3: 'Hospital accounting failure', 4: 'Negative number of contacts', 5: 'Malloc failure',
    6: 'Other failure', 7: 'Wrong state', 8: 'Contact probability failure', 9: 'Infectees mismatch',
    100: 'Work list overflow', 101: 'Candidate list overflow', 102: 'Testing queue overflow',
    103: 'Hospital event list overflow', 104: 'Day counter overflow', 105: 'Device synchronisation timeout',
    106: 'Cross-shard exchange segment overflow', 107: 'Infectee list pool overflow',
}
# main.pyx:660-682: infectiousness by day relative to symptom onset (Luca et al. 2020)
INFECTIOUSNESS_OVER_TIME = (
    (-10, 0.00183), (-9, 0.00280), (-8, 0.00446), (-7, 0.00742), (-6, 0.01291), (-5, 0.02350),
    (-4, 0.04419), (-3, 0.08247), (-2, 0.14018), (-1, 0.19032), (0, 0.18539), (1, 0.13091),
    (2, 0.07538), (3, 0.04018), (4, 0.02144), (5, 0.01185), (6, 0.00686), (7, 0.00415),
    (8, 0.00262), (9, 0.00172), (10, 0.00117),
)
NO_TESTING, ALL_WITH_SYMPTOMS_CT, ALL_WITH_SYMPTOMS, ONLY_SEVERE_SYMPTOMS = 0, 1, 2, 3  # main.pyx:441-445

POP_ATTRS = ('susceptible', 'vaccinated', 'infected', 'all_infected', 'detected', 'all_detected',
             'in_icu', 'cum_icu', 'in_ward', 'dead', 'recovered', 'non_hospital_deaths',
             'new_infections')  # generate_state order, main.pyx:1819-1833


class SimulationFailed(Exception):
    pass


def _expand_lte(pairs, nr_ages):
    """ClassifiedValues + cv_get_greatest_lte (main.pyx:684-730) expanded to one float32 per age."""
    classes = [int(p[0]) for p in pairs]
    values = np.asarray([p[1] for p in pairs], dtype=np.float64).astype(np.float32)
    if classes[0] > 0:
        raise ValueError('first age class must be 0 (the reference reads out of bounds otherwise)')
    out = np.zeros(_eng.MAX_AGES, dtype=np.float32)
    for age in range(nr_ages):
        idx = len(classes) - 1
        for i, c in enumerate(classes):
            if c > age:
                idx = i - 1
                break
        out[age] = values[idx]
    return out


def _cv_div(a, b):
    # main.pyx:808-817, Python double division before the float32 store
    assert [x[0] for x in a] == [x[0] for x in b]
    return [(x[0], x[1] / y[1]) for x, y in zip(a, b)]


_disease_cache = {}


def build_disease_struct(disease_params, nr_ages, imported_infection_ages):
    """Memoised on the parameter VALUES: ensemble members share one scenario."""
    key = (repr(sorted(disease_params.items(), key=lambda kv: kv[0])), nr_ages, repr(imported_infection_ages))
    hit = _disease_cache.get(key)
    if hit is None:
        if len(_disease_cache) > 8:
            _disease_cache.clear()
        hit = _disease_cache[key] = _build_disease_struct(disease_params, nr_ages, imported_infection_ages)
    return hit   # (Disease struct, variant names): treated as read-only by every user


def _build_disease_struct(disease_params, nr_ages, imported_infection_ages):
    """Disease.__init__ / variant_init (main.pyx:820-881) -> reina_disease_t."""
    d = _eng.Disease()
    variants = [dict(disease_params)]
    names = ['wild-type']
    for v in disease_params['variants']:
        p = dict(disease_params)
        p.update(v)
        variants.append(p)
        names.append(v['name'])
    if len(variants) > _eng.MAX_VARIANTS:
        raise ValueError('at most %d variants' % _eng.MAX_VARIANTS)
    scalar_fields = ('infectiousness_multiplier', 'p_asymptomatic_infection', 'p_hospital_death_no_beds',
                     'p_icu_death_no_beds', 'mean_incubation_duration', 'mean_duration_from_onset_to_death',
                     'mean_duration_from_onset_to_recovery', 'ratio_of_duration_before_hospitalisation',
                     'ratio_of_duration_in_ward', 'p_mask_protects_others', 'p_mask_protects_wearer')
    for vi, p in enumerate(variants):
        for f in scalar_fields:
            getattr(d, f)[vi] = float(np.float32(p[f]))
        for k, (day, val) in enumerate(INFECTIOUSNESS_OVER_TIME):
            d.infectiousness_over_time[vi][k] = float(np.float32(val))
        sus = _expand_lte(p['p_susceptibility'], nr_ages)
        for a in range(nr_ages):
            d.p_susceptibility[vi][a] = float(sus[a])
    base = variants[0]
    sym = [tuple(x) for x in base['p_symptomatic']]
    sev = [tuple(x) for x in base['p_severe']]
    cri = [tuple(x) for x in base['p_critical']]
    fat = [tuple(x) for x in base['p_fatal']]
    for field, pairs in (('p_symptomatic', sym), ('p_severe_given_symptomatic', _cv_div(sev, sym)),
                         ('p_critical_given_severe', _cv_div(cri, sev)),
                         ('p_fatal_given_critical', _cv_div(fat, cri)),
                         ('p_death_outside_hospital', base['p_death_outside_hospital'])):
        arr = _expand_lte(pairs, nr_ages)
        tgt = getattr(d, field)
        for a in range(nr_ages):
            tgt[a] = float(arr[a])
    # imported_infection_ages -> cumulative float32 weights (main.pyx:1376-1384)
    wsum = sum([x[1] for x in imported_infection_ages])
    total = 0
    n = len(imported_infection_ages)
    if n > _eng.MAX_IMPORT_CLASSES:
        raise ValueError('too many import age classes')
    d.n_import_classes = n
    for k, (age, weight) in enumerate(imported_infection_ages):
        weight = weight / wsum
        d.import_class_min_age[k] = int(age)
        nxt = imported_infection_ages[k + 1][0] if k + 1 < n else nr_ages
        d.import_class_max_age[k] = min(int(nxt) - 1, nr_ages - 1)
        d.import_class_cum[k] = float(np.float32(weight + total))
        total += weight
    return d, names


_static_pack_cache = {}


def _static_pack(tables):
    """(ranges, range id per entry, meta words): depend only on the entry keys, which never change
    for a given ContactMatrix; cached by the identity of its key arrays."""
    ident = _static_pack_cache.get('ident')   # the very same key arrays as last time (native builder): no hashing
    if ident is not None and ident[0] is tables.place and ident[1] is tables.cmin and ident[2] is tables.cmax:
        return ident[3]
    key = (tables.place.tobytes(), tables.cmin.tobytes(), tables.cmax.tobytes())
    hit = _static_pack_cache.get(key)
    if hit is None:
        pairs = list(zip(tables.cmin.tolist(), tables.cmax.tolist()))
        ranges = sorted(set(pairs))
        if len(ranges) > _eng.MAX_RANGES:
            raise ValueError('more than %d distinct contact age ranges' % _eng.MAX_RANGES)
        range_id = {r: k for k, r in enumerate(ranges)}
        rid = np.asarray([range_id[r] for r in pairs], dtype=np.uint32)
        m_all = (tables.place.astype(np.uint32) | (tables.cmin.astype(np.uint32) << 8)
                 | (tables.cmax.astype(np.uint32) << 16) | (rid << 24))
        hit = (ranges, m_all, {})   # {}: the padded meta array per (ages, entries per age), built on first use
        _static_pack_cache.clear()
        _static_pack_cache[key] = hit
    _static_pack_cache['ident'] = (tables.place, tables.cmin, tables.cmax, hit)
    return hit


def pack_contact_tables(tables, nr_ages):
    """ContactTables (contacts.py) -> the fixed-shape arrays of reina_contact_tables_t."""
    E = _eng.MAX_ENTRIES
    ranges, m_all, padded = _static_pack(tables)
    packed = getattr(tables, 'packed', None)
    if packed is not None:   # built by the library's host-side builder, thresholds included
        nrc, thr = packed
        c = int(tables.count[0])
        hit = padded.get((nr_ages, c))
        if hit is None:
            count = np.zeros(_eng.MAX_AGES, dtype=np.int32)
            meta = np.zeros((_eng.MAX_AGES, E), dtype=np.uint32)
            count[:nr_ages] = c
            meta[:nr_ages, :c] = m_all.reshape(nr_ages, c)
            hit = padded[(nr_ages, c)] = (count, meta)
        return nrc, hit[0], thr, hit[1], ranges
    nrc = np.zeros(_eng.MAX_AGES, dtype=np.float32)
    nrc[:nr_ages] = tables.nr_contacts_by_age.astype(np.float32)
    count = np.zeros(_eng.MAX_AGES, dtype=np.int32)
    thr = np.full((_eng.MAX_AGES, E), 0xFFFFFFFF, dtype=np.uint32)
    meta = np.zeros((_eng.MAX_AGES, E), dtype=np.uint32)
    with np.errstate(invalid='ignore'):
        t_all = np.clip(np.floor(np.nan_to_num(tables.cum_p, nan=0.0) * 4294967296.0), 0, 4294967295.0)
    t_all = t_all.astype(np.uint64).astype(np.uint32)
    cnt = tables.count[:nr_ages].astype(np.int64)
    if cnt.max() > E:
        raise ValueError('more than %d contact entries for an age' % E)
    count[:nr_ages] = cnt
    if np.all(cnt == cnt[0]) and np.array_equal(tables.offset[:nr_ages], np.arange(nr_ages) * cnt[0]):
        c = int(cnt[0])
        thr[:nr_ages, :c] = t_all.reshape(nr_ages, c)
        meta[:nr_ages, :c] = m_all.reshape(nr_ages, c)
    else:
        for a in range(nr_ages):
            o, c = int(tables.offset[a]), int(tables.count[a])
            thr[a, :c] = t_all[o:o + c]
            meta[a, :c] = m_all[o:o + c]
    return nrc, count, thr, meta, ranges


class Context:
    """MI355X-native agent engine with the reference `Context` protocol (main.pyx:1746-2101)."""

    def __init__(self, population_params, healthcare_params, disease_params, start_date,
                 random_seed=4321, device='cuda:0', engine_factory=None, comm=None, strict=False):
        """`comm` (sharding.TorchComm or compatible: .rank, .world, .all_reduce_sum/max) makes this
        Context one shard of a population split over comm.world engine instances; population,
        beds, ICU units and import / vaccination quotas given here are the GLOBAL ones."""
        from .sharding import split_count, split_population
        self.comm = comm
        self.strict = bool(strict)   # iterate() raises on the day of a problem (main.pyx:2017-2018)
        # testing aid: take the begin / all-reduce / end path even with a single shard
        self.always_collective = bool(comm is not None and getattr(comm, 'always_collective', False))
        self._direct = getattr(comm, 'direct', None) if comm is not None else None
        self.shard_rank = comm.rank if comm is not None else 0
        self.n_shards = comm.world if comm is not None else 1
        self._split = (lambda x: x) if self.n_shards == 1 else (lambda x: split_count(x, self.shard_rank, self.n_shards))
        # cross-shard infector links: 'exact' (global ids, contact / feedback / tracing records exchanged
        # through all-to-all segments -- the true infector as in the reference, main.pyx:219-233) or 'mirror' (stand-in
        # infectors, one all-reduce per day; sharding.py).  The comm object says which; exact unless told otherwise.
        from .sharding import DEFAULT_ATTRIBUTION
        self.attribution = getattr(comm, 'attribution', DEFAULT_ATTRIBUTION) if self.n_shards > 1 else 'none'
        if self.attribution not in ('exact', 'mirror', 'none'):
            raise ValueError("comm.attribution must be 'exact' or 'mirror'")
        # (shards stepped together in one process -- sharding.InProcessComm, `members` -- are exchanged by their driver)
        if self.attribution == 'exact' and not (hasattr(comm, 'all_to_all') or hasattr(comm, 'members') or getattr(self._direct, 'a2a_ptr', None)):
            # (a comm object written for the one all-reduce of rounds 1-4 has no `attribution` and would be switched to exact
            # attribution silently, to fail with AttributeError on the first stepped day: round-5 advisor)
            raise TypeError("exact attribution (sharding.DEFAULT_ATTRIBUTION; SURVEY 8 f-4) exchanges records: the comm object "
                            "must provide all_to_all(send, recv), or set comm.attribution = 'mirror' for one all-reduce a day")
        population_params = dict(population_params)
        ipc = population_params.pop('initial_population_condition', None)

        ages = population_params['age_structure']
        if hasattr(ages, 'items') and hasattr(ages, 'index'):
            nr_ages = int(ages.index.max()) + 1
            age_counts = np.zeros(nr_ages, dtype=np.int64)
            for a, c in ages.items():
                age_counts[int(a)] = int(c)
        else:
            age_counts = np.asarray(ages, dtype=np.int64).copy()
            nr_ages = len(age_counts)
        if nr_ages > _eng.MAX_AGES:
            raise ValueError('at most %d ages' % _eng.MAX_AGES)
        self.global_age_counts = age_counts.copy()
        if self.n_shards > 1:
            if self.n_shards > _eng.MAX_SHARDS:
                raise ValueError('at most %d shards' % _eng.MAX_SHARDS)
            age_counts = split_population(age_counts, self.shard_rank, self.n_shards)
        total = int(age_counts.sum())
        if total >= 2 ** 31:
            raise ValueError('a single engine instance holds < 2^31 agents; shard the population')
        self.nr_ages = nr_ages
        self.total_people = total
        self.age_counts = age_counts
        self.age_start = np.zeros(_eng.MAX_AGES + 1, dtype=np.int64)
        self.age_start[1:nr_ages + 1] = np.cumsum(age_counts)
        self.age_start[nr_ages + 1:] = total

        self.age_group_labels = list(population_params['age_groups']['labels'])
        self.age_group_indices = np.asarray(population_params['age_groups']['age_indices'], dtype=np.int64)

        disease, self.variant_names = build_disease_struct(
            disease_params, nr_ages, population_params['imported_infection_ages'])
        self._disease = disease
        self._seed = int(random_seed) & 0xFFFFFFFFFFFFFFFF
        self._sample_calls = 0
        self.nr_variants = len(self.variant_names)

        cfg = _eng.Config()
        cfg.n_agents = total
        cfg.nr_ages = nr_ages
        cfg.nr_variants = self.nr_variants
        cfg.seed = int(random_seed) & 0xFFFFFFFFFFFFFFFF
        cfg.max_work_items = total + 1024
        # candidate records: one region per scanning wave (total) + the region k_remote fills with one
        # record per incoming cross-shard attempt (a few % of the shard per day at an epidemic peak) + an
        # overflow region of the same size (at most 2^20 records) for waves with more hits than agents
        cfg.max_candidates = total + 2 * (max(1024 * 1024, total // 2) if self.n_shards > 1 else 1024 * 1024)
        cfg.max_queue = total + 64
        cfg.max_hosp_events = _eng.default_max_hosp_events(total)   # bed / ICU events of one day (clamped to what the walk's buckets hold)
        cfg.n_shards = self.n_shards
        cfg.shard_rank = self.shard_rank
        if self.n_shards > 1:
            # the shards exchange per-bucket maps of the day's bed / ICU events: the same number of buckets on every shard
            # (that of the largest shard: every age's count divided by the shards, rounded up)
            largest = int(sum(-(-int(c) // self.n_shards) for c in self.global_age_counts))
            cfg.hosp_ranges = _eng.hosp_ranges(largest)
        slots = 64
        while slots < total // 1024 and slots < (1 << 20):
            slots <<= 1
        cfg.mirror_slots = slots
        for a in range(_eng.MAX_AGES + 1):
            cfg.age_start[a] = int(self.age_start[a])
        if self.attribution == 'exact':
            if total > _eng.GID_INDEX_MASK:
                raise ValueError('exact cross-shard attribution holds fewer than 2^27 agents per shard')
            cfg.exact_attribution = 1
            # records per peer shard and exchange.  An exchange moves whole segments whatever they hold (the host cannot know the
            # counts without waiting for the GPU), so the capacity is what the exchange costs: 2.4 x the need of the default
            # scenario's peak day (cross-shard contacts that pass the whole transmission test: 0.65 % of a shard's agents, spread
            # over the peers -- DESIGN section 6).  A day with more fails loudly (problem 106); comm.xchg_cap overrides.
            # (the same on every shard -- the segments are exchanged whole --: from the largest shard's size)
            cap = getattr(comm, 'xchg_cap', None)
            largest = int(sum(-(-int(c) // self.n_shards) for c in self.global_age_counts))
            # (a small population gets relatively more: 16 384 records are 128 KB -- a heavy outbreak among 30 000 agents on two
            # shards put 2100 records a day into a segment of 2048, found by the randomised soak)
            cfg.xchg_cap = int(cap) if cap else max(16384, largest // (64 * self.n_shards))
            cfg.pool_cap = max(4096, total // 16)
            # every shard's age_start: a source draws its target on the other shard and needs its age
            tab = np.zeros((self.n_shards, _eng.MAX_AGES + 1), dtype=np.int32)
            for r in range(self.n_shards):
                cnt = split_population(self.global_age_counts, r, self.n_shards)
                tab[r, 1:nr_ages + 1] = np.cumsum(cnt)
                tab[r, nr_ages + 1:] = int(cnt.sum())
            self._shard_age_start = np.ascontiguousarray(tab)
            cfg.shard_age_start = self._shard_age_start.ctypes.data
        if engine_factory is None:
            self.engine = _eng.hip_engine(cfg, disease, device)
        else:
            self.engine = engine_factory(cfg, disease)
        self.beds = int(healthcare_params['hospital_beds'])
        self.icu_units = int(healthcare_params['icu_units'])
        self.engine.init_state(self._split(self.beds), self._split(self.icu_units))

        self.contact_matrix = ContactMatrix(population_params['contacts_per_day'], nr_ages)
        # table rebuilds on mobility changes go through the library's host-side builder
        self.contact_matrix.native_build = self.engine.f['build_contact_tables']
        self.contact_matrix.pack_ages, self.contact_matrix.pack_entries = _eng.MAX_AGES, _eng.MAX_ENTRIES
        self._upload_tables()

        # HealthcareSystem host-side settings (main.pyx:461-472)
        self.testing_mode = NO_TESTING
        self.p_detected_anyway = np.float32(0)
        self.p_successful_tracing = np.float32(1.0)
        self.vaccinations = []  # dicts(min_age, max_age, nr_daily, slot)
        # Population weekly imports (main.pyx:1366-1369)
        self.weekly_infections_amount = 0
        self.weekly_infections_leftover = [0.0] * (self.nr_variants + 1)
        self.weekly_infections_shares = [0.0] * self.nr_variants
        self.weekly_infections_shares[0] = 1.0

        self.start_date = start_date
        self.day = 0
        self.interventions = []
        self._pending_imports = []
        self._pending_beds = 0
        self._pending_icu = 0
        self._keep = []
        self._iv_index = None
        self._iv_version = 0
        # sharded with a communicator of our own: the engine queues the pressure all-reduce itself
        self._in_stream = self._direct is not None and (self.n_shards > 1 or self.always_collective)
        if self._in_stream:
            self.engine.set_collective(self._direct.fn_ptr, self._direct.comm_ptr)
            if self.attribution == 'exact':
                self.engine.set_alltoall(self._direct.a2a_ptr, self._direct.comm_ptr)
        # main.pyx:1780-1781: the initial condition is applied last, before any intervention exists
        if ipc is not None and ipc.has_initial_state():
            self._set_initial_state(ipc)

    def _set_initial_state(self, ipc):
        """Population.set_initial_state (main.pyx:1452-1516) on the engine; a sharded population
        applies each shard's share of every number."""
        # An ICU-fated agent takes a bed first and hands it back when it moves on (hc.to_icu), so it is refused one iff the
        # hospital has NO beds -- and then the reference does not construct: person_hospitalize leaves the agent dead or
        # recovered, person_transfer_to_icu follows, and Population.transfer_to_icu / release_from_hospital assert
        # state == HOSPITALIZED (AssertionError out of Context.__init__, main.pyx:1781 -> :1495 -> :350 -> :1603; recorded
        # with the real reference in the build container).  Same answer here, for every shard alike (global numbers).
        # The reference walks range(were_incubating()) over boundaries that add up to MORE than that when fewer people
        # recovered than are incubating (recovered_without_illness() = were_incubating - were_ill = incubating, so the
        # boundaries end at 2 * incubating + ill + dead + in_icu + in_ward): the walk then stops short and the LAST
        # categories -- in ward, in ICU, ... -- lose slots (main.pyx:1456-1463, calc/datasets.py:120-134).  Otherwise the
        # slots behind the last boundary recovered on their own.  The slots each category keeps are worked out for the
        # WHOLE population first and divided among the shards afterwards.
        M = int(ipc.were_incubating())
        widths = [int(ipc.incubating), int(ipc.recovered_without_illness()), int(ipc.ill), int(ipc.dead), int(ipc.in_icu),
                  int(ipc.in_ward)]
        kept, lo = [], 0
        for w in widths:
            kept.append(max(0, min(lo + w, M) - lo))
            lo += w
        rest = max(0, M - lo)
        if kept[4] > 0 and int(self.beds) == 0:
            raise AssertionError('initial population condition: an agent bound for ICU was refused a hospital bed')
        ic = _eng.InitialState()
        sp = self._split
        (ic.incubating, ic.recovered_without_illness, ic.ill, ic.dead, ic.in_icu, ic.in_ward) = [sp(k) for k in kept]
        ic.were_incubating = (ic.incubating + ic.recovered_without_illness + ic.ill + ic.dead + ic.in_icu + ic.in_ward
                              + sp(rest))
        ic.confirmed_cases = int(ipc.confirmed_cases)
        ic.confirmed_first = self.shard_rank
        ic.confirmed_stride = self.n_shards
        self.engine.set_initial_state(ic)

    # ------------------------------------------------------------------ host helpers
    def _packed_tables(self):
        t = self.contact_matrix.tables
        nrc, count, thr, meta, ranges = pack_contact_tables(t, self.nr_ages)
        mask = np.zeros((_eng.MAX_AGES, 8), dtype=np.float32)
        mask[:self.nr_ages, :6] = self.contact_matrix.mask_probabilities.astype(np.float32)
        return nrc, count, thr, meta, mask, ranges

    def _upload_tables(self):
        self.engine.upload_contact_tables(*self._packed_tables())

    def get_date_for_today(self):
        d = date.fromisoformat(self.start_date)
        return (d + timedelta(days=self.day)).isoformat()

    @staticmethod
    def _iv_date(iv):
        """An intervention's date as a datetime.date.  The reference compares iv.date with today's ISO string
        (main.pyx:2014), so a date written any other way silently never applies there; here it is an error, raised
        when the intervention is added (round-3 advisor finding: it used to surface from the first day of the run)."""
        ds = str(iv.date)
        try:
            return date.fromisoformat(ds)
        except ValueError:
            raise ValueError('intervention %r: date %r is not an ISO date (YYYY-MM-DD)' % (getattr(iv, 'type', iv), ds)) from None

    def add_intervention(self, iv):
        self._iv_date(iv)
        self.interventions.append(iv)
        self._iv_version += 1   # (the by-day index of _build_day is rebuilt)

    def find_variant(self, variant_str):
        if variant_str is None:
            return 0
        for idx, vn in enumerate(self.variant_names):
            if variant_str == vn:
                return idx
        raise Exception('Variant %s not found' % variant_str)

    # main.pyx:1880-1960
    def apply_intervention(self, iv):
        params = iv.get_param_values()
        t = iv.type
        if t == 'test-all-with-symptoms':
            self.testing_mode = ALL_WITH_SYMPTOMS
        elif t == 'test-only-severe-symptoms':
            self.testing_mode = ONLY_SEVERE_SYMPTOMS
            self.p_detected_anyway = np.float32(params['mild_detection_rate'] / 100.0)
        elif t == 'test-with-contact-tracing':
            self.testing_mode = ALL_WITH_SYMPTOMS_CT
            self.p_successful_tracing = np.float32(params['efficiency'] / 100.0)
        elif t == 'build-new-icu-units':
            self._pending_icu += int(params['units'])
        elif t == 'build-new-hospital-beds':
            self._pending_beds += int(params['beds'])
        elif t == 'import-infections':
            # (the testing mode of THIS moment decides whether the imported agents keep an infectee list: interventions
            # of one date are applied in list order, main.pyx:2013-2015)
            self._pending_imports.append((int(params['amount']), self.find_variant(params.get('variant')), 1, self.testing_mode))
        elif t == 'import-infections-weekly':
            shares = [0] * len(self.variant_names)
            for pn in params.keys():
                if not pn.startswith('variant_'):
                    continue
                vid = self.find_variant(pn.replace('variant_', ''))
                share = params[pn]
                shares[vid] = share / 100 if share else 0
            shares[0] = 1 - sum(shares)
            self.weekly_infections_amount = int(params['weekly_amount'])
            self.weekly_infections_shares = shares
        elif t == 'limit-mobility':
            reduction = (100 - params['reduction']) / 100.0
            place = params.get('place')
            if place is not None:
                place = PLACES.index(place)
            self.contact_matrix.set_mobility_factor(reduction, place=place, min_age=params.get('min_age'),
                                                    max_age=params.get('max_age'))
        elif t == 'wear-masks':
            p = params['share_of_contacts'] / 100.0
            place = params.get('place')
            if place is not None:
                place = PLACES.index(place)
            self.contact_matrix.set_mask_probability(p, place=place, min_age=params.get('min_age'),
                                                     max_age=params.get('max_age'))
        elif t == 'vaccinate':
            nr = params['weekly_vaccinations'] / 7
            mn, mx = params.get('min_age'), params.get('max_age')
            for v in self.vaccinations:
                if v['min_age'] == mn and v['max_age'] == mx:
                    break
            else:
                if len(self.vaccinations) >= _eng.MAX_VACCINATIONS:
                    raise Exception('too many vaccination programmes')
                v = dict(min_age=mn, max_age=mx, slot=len(self.vaccinations))
                self.vaccinations.append(v)
            v['nr_daily'] = nr
        else:
            raise Exception()

    def _build_day(self, history_ptr=None):
        """Host part of iterate(): interventions dated today, init_day bookkeeping -> reina_day_t.
        Returns (Day, tables_changed)."""
        # (this function is host time per day, and a short run is host-bound -- DESIGN section 5: the interventions are indexed
        # by day number, a population that is not sharded splits nothing, and without a weekly import flow its float32
        # leftovers stay what they are: below 1, so no import either)
        ivs = self.interventions
        stamp = (len(ivs), self._iv_version, id(ivs[0]) if ivs else 0, id(ivs[-1]) if ivs else 0)
        if self._iv_index is None or self._iv_index[0] != stamp:
            self._date0 = date.fromisoformat(self.start_date)
            by_day = {}
            for iv in ivs:  # list order is kept within a date (main.pyx:2013-2015)
                by_day.setdefault((self._iv_date(iv) - self._date0).days, []).append(iv)
            self._iv_index = (stamp, by_day)
        if self.day >= _eng.MAX_DAYS:
            raise SimulationFailed('Day counter overflow: the engine simulates at most %d days' % _eng.MAX_DAYS)
        for iv in self._iv_index[1].get(self.day, ()):
            self.apply_intervention(iv)
        changed = self.contact_matrix.init_day()
        # Population.infect_people_daily (main.pyx:1671-1685): float32 leftover arithmetic
        weekly = []
        for vid in range(self.nr_variants if self.weekly_infections_amount else 0):
            leftover = np.float32(self.weekly_infections_leftover[vid])
            leftover = np.float32(float(leftover) + self.weekly_infections_amount / 7.0 * self.weekly_infections_shares[vid])
            amount_today = int(leftover)
            if amount_today:
                weekly.append((amount_today, vid, 0, self.testing_mode))
                leftover = np.float32(leftover - np.float32(amount_today))
            assert leftover >= 0
            self.weekly_infections_leftover[vid] = float(leftover)
        d = _eng.Day()
        d.day = self.day
        d.testing_mode = self.testing_mode
        d.p_detected_anyway = float(self.p_detected_anyway)
        d.p_successful_tracing = float(self.p_successful_tracing)
        # quotas are global; a shard takes its 1/G share (new capacity: of the running totals)
        d.add_beds = self._split(self.beds + self._pending_beds) - self._split(self.beds)
        d.add_icu_units = self._split(self.icu_units + self._pending_icu) - self._split(self.icu_units)
        self.beds += self._pending_beds
        self.icu_units += self._pending_icu
        self._pending_beds = 0
        self._pending_icu = 0
        batches = self._pending_imports + weekly
        self._pending_imports = []
        if len(batches) > _eng.MAX_IMPORT_BATCHES:
            raise Exception('more than %d import batches in one day' % _eng.MAX_IMPORT_BATCHES)
        d.n_import_batches = len(batches)
        for k, (count, variant, pre, mode) in enumerate(batches):
            d.import_batches[k].count = self._split(count)
            d.import_batches[k].variant = variant
            d.import_batches[k].pre_init = pre
            d.import_batches[k].testing_mode = mode
        nv = 0
        pop_max_age = self.nr_ages - 1
        for v in self.vaccinations:
            if not v['nr_daily']:
                continue
            mn = 0 if v['min_age'] is None else v['min_age']
            mx = pop_max_age if v['max_age'] is None else v['max_age']
            d.vaccinations[nv].nr = self._split(int(v['nr_daily']))
            d.vaccinations[nv].idx_start = int(self.age_start[mn])
            d.vaccinations[nv].idx_end = int(self.age_start[mx + 1]) if mx < pop_max_age else self.total_people
            d.vaccinations[nv].slot = v['slot']
            nv += 1
        d.n_vaccinations = nv
        d.history_row = history_ptr
        return d, changed

    # ------------------------------------------------------------------ day stepping
    # main.pyx:2011-2018
    def _step(self, d):
        if self._in_stream or (self.n_shards == 1 and not self.always_collective):
            self.engine.step_day(d)
        else:
            # the day phase by phase, the collectives each phase asks for in between: one all-reduce (the cross-shard
            # pressure block with the bed / ICU event maps), and under exact attribution the record exchanges
            for ph in range(_eng.PH_NR):
                need = self.engine.step_phase(d, ph)
                if ph == _eng.PH_MAIN and self.always_collective:
                    need |= _eng.X_ALLREDUCE   # (a single shard asks for no exchange: the world-1 tests exercise the call anyway)
                if need & _eng.X_ALLREDUCE:
                    self.comm.all_reduce_sum(self.engine.tensors['pressure'])
                if need & _eng.X_ALLTOALL:
                    self.comm.all_to_all(self.engine.tensors['xsend'], self.engine.tensors['xrecv'])

    def iterate(self):
        d, changed = self._build_day()
        if changed:
            self._upload_tables()
        self._step(d)
        self.day += 1
        if self.strict:
            # the reference checks `problem` at the end of iterate() (main.pyx:2017-2018): wait for the day
            self._raise_on_problem(self._read_counters_global())
        if self.n_shards == 1 and not self.always_collective:
            self.engine.prefetch_counters()   # the next generate_state() finds them on the host

    def make_plan(self, days):
        """Host part of `days` consecutive days, done once: the intervention schedule turned into
        day descriptors, cut into stretches of unchanged contact tables.  A plan does not depend
        on the random seed, so one plan can drive every member of a Monte-Carlo ensemble
        (reina_model_amd/ensemble.py).  Advances this Context's host-side state by `days`."""
        segments = []   # (packed tables or None, ctypes Day array, n)
        start_day = self.day
        pending = []
        tables = None
        mobility = []
        for _ in range(days):
            mobility.append(float(self.contact_matrix.mobility_factor))
            d, changed = self._build_day(None)
            if changed:
                if pending:
                    segments.append((tables, (_eng.Day * len(pending))(*pending), len(pending)))
                    pending = []
                tables = self._packed_tables()
            pending.append(d)
            self.day += 1
        if pending:
            segments.append((tables, (_eng.Day * len(pending))(*pending), len(pending)))
        return dict(segments=segments, days=days, mobility_history=mobility, start_day=start_day)

    def run_plan(self, plan, record_history=True):
        """Execute a plan made by make_plan (of this Context or of another one with the same
        scenario).  Returns history[days, COUNTER_WORDS] like run().  The scenario's host-side state
        (intervention cursor, contact matrix) advances only in the Context that MADE the plan:
        continue a replayed simulation with further plans of that same planner."""
        days = plan['days']
        a = self.engine.alloc
        hist = self._history_buffer(days) if record_history else None
        base = a.ptr(hist) if record_history else None
        done = 0
        for tables, arr, n in plan['segments']:
            if tables is not None:
                self.engine.upload_contact_tables(*tables)
            ptr = base + 4 * _eng.COUNTER_WORDS * done if record_history else None
            self.engine.run_day_array(arr, n, ptr)
            done += n
        self.mobility_history = plan['mobility_history']
        self.day = plan['start_day'] + days
        if record_history:
            return self._history_to_host(hist, days)
        return None

    def _history_buffer(self, days):
        """`days` history rows (each written whole by its day's opening launch: no memset); they come back together with the
        counters after the last day in one library call (engine.read_history)"""
        return self.engine.alloc.empty(max(days, 1) * _eng.COUNTER_WORDS, np.int32)

    def _history_to_host(self, hist, days):
        out = self.engine.read_history(hist, days)
        self._raise_on_problem(out[days])
        return out[:days]

    def _run_streamed(self, days, record_history):
        """run() for an unsharded population, or a sharded one whose engine queues the per-day
        all-reduce itself (reina_set_collective): day descriptors are built on the host and handed to
        the library in growing chunks (1, 2, 4, ... 64 days), so the GPU works on the first days while the host is still
        turning the intervention schedule into the later ones (table uploads are queued copies from
        pinned staging, they do not drain the stream either)."""
        a = self.engine.alloc
        single = self.n_shards == 1 and not self.always_collective
        hist = None
        if record_history:
            hist = self._history_buffer(days) if single else a.zeros(days * _eng.COUNTER_WORDS, np.int32)
        base = a.ptr(hist) if record_history else None
        row = 4 * _eng.COUNTER_WORDS
        self.mobility_history = []
        pending, issued, chunk = [], 0, 1   # (1, 2, 4, ... 64 days per call: day 0 runs on the GPU while day 1 is being planned)

        def flush():
            nonlocal pending, issued, chunk
            if pending:
                arr = (_eng.Day * len(pending))(*pending)
                self.engine.run_day_array(arr, len(pending), base + row * issued if record_history else None)
                issued += len(pending)
                pending = []
                chunk = min(chunk * 2, 64)

        for _ in range(days):
            self.mobility_history.append(float(self.contact_matrix.mobility_factor))
            d, changed = self._build_day(None)
            if changed:
                flush()
                self.engine.upload_contact_tables(*self._packed_tables())
            pending.append(d)
            self.day += 1
            if len(pending) >= chunk:
                flush()
        flush()
        if record_history:
            if single:
                out = self._history_to_host(hist, days)
            else:   # sharded: rows are summed over the shards when exported
                out = self._reduce_counter_rows(hist, days)
                self._raise_on_problem(self._read_counters_global())
            return out
        return None

    def run(self, days, record_history=True):
        """Run `days` consecutive days with one library call per stretch of unchanged contact
        tables (the loop of calc/simulation.py:194-270 without per-day host round trips).
        Returns history[days, COUNTER_WORDS] (row d = counters BEFORE day d ran) as a host array,
        or None; `self.mobility_history[d]` is the mobility factor generate_state() would have
        reported on that day."""
        if self._in_stream or (self.n_shards == 1 and not self.always_collective):
            return self._run_streamed(days, record_history)
        a = self.engine.alloc
        hist = a.zeros(days * _eng.COUNTER_WORDS, np.int32) if record_history else None
        base = a.ptr(hist) if record_history else 0
        self.mobility_history = []
        for k in range(days):
            ptr = base + 4 * _eng.COUNTER_WORDS * k if record_history else None
            self.mobility_history.append(float(self.contact_matrix.mobility_factor))
            d, changed = self._build_day(ptr)
            if changed:
                self._upload_tables()
            self._step(d)
            self.day += 1
        if record_history:
            out = self._reduce_counter_rows(hist, days)
            self._raise_on_problem(self._read_counters_global())
            return out
        return None

    def synchronize(self):
        self._raise_on_problem(self._read_counters_global())

    def exchange_fill(self):
        """exact attribution: (the most records any exchange segment of THIS shard has held so far, the segments' capacity) -- how
        close the run came to problem 106; (0, 0) for a population that exchanges no records"""
        if self.attribution != 'exact':
            return 0, 0
        ctl = self.engine.alloc.to_host(self.engine.tensors['control'])
        return int(ctl[_eng.L_XCHG_PEAK]), int(self.engine.config.xchg_cap)

    def _raise_on_problem(self, counters):
        problem = int(counters[_eng.C_NR * _eng.MAX_AGES + _eng.S_PROBLEM])
        if problem != 0:
            raise SimulationFailed(PROBLEM_TO_STR.get(problem, 'Problem %d' % problem))

    # ---- sharded state export: counters are additive over shards; problem / day are not
    def _reduce_counter_rows(self, buf, rows):
        base = _eng.C_NR * _eng.MAX_AGES
        local = self.engine.alloc.to_host(buf).reshape(rows, _eng.COUNTER_WORDS).copy()
        problem = np.ascontiguousarray(local[:, base + _eng.S_PROBLEM])
        day = local[:, base + _eng.S_DAY].copy()
        self.comm.all_reduce_sum(local)
        self.comm.all_reduce_max(problem)
        local[:, base + _eng.S_PROBLEM] = problem
        local[:, base + _eng.S_DAY] = day
        return local

    def _read_counters_global(self):
        c = self.engine.read_counters()
        if self.n_shards == 1 and not self.always_collective:
            return c
        return self._reduce_counter_rows_host(c)

    def _reduce_counter_rows_host(self, c):
        base = _eng.C_NR * _eng.MAX_AGES
        local = np.array(c, dtype=np.int32).reshape(1, -1)
        problem = np.ascontiguousarray(local[:, base + _eng.S_PROBLEM])
        day = local[:, base + _eng.S_DAY].copy()
        self.comm.all_reduce_sum(local)
        self.comm.all_reduce_max(problem)
        local[:, base + _eng.S_PROBLEM] = problem
        local[:, base + _eng.S_DAY] = day
        return local[0]

    # ------------------------------------------------------------------ state export
    def state_from_counters(self, counters, mobility_factor=None):
        """Context.generate_state (main.pyx:1813-1857) from one counter block."""
        A = _eng.MAX_AGES
        sc = counters[_eng.C_NR * A:]
        total_infections, total_infectors = int(sc[_eng.S_TOTAL_INFECTIONS]), int(sc[_eng.S_TOTAL_INFECTORS])
        r = total_infections / total_infectors if total_infectors > 5 else 0
        mf = self.contact_matrix.mobility_factor if mobility_factor is None else mobility_factor
        s = dict(
            available_icu_units=int(sc[_eng.S_AVAILABLE_ICU]),
            available_hospital_beds=int(sc[_eng.S_AVAILABLE_BEDS]),
            total_icu_units=int(sc[_eng.S_ICU_UNITS]),
            r=r,
            exposed_per_day=int(sc[_eng.S_EXPOSED_PER_DAY]),
            ct_cases_per_day=int(sc[_eng.S_CT_CASES_PER_DAY]),
            mobility_limitation=1 - float(mf),
        )
        ngroups = len(self.age_group_labels)
        for attr in POP_ATTRS:
            ci = _eng.C_NAMES.index(attr)
            per_age = counters[ci * A: ci * A + self.nr_ages]
            s[attr] = np.bincount(self.age_group_indices[:self.nr_ages], weights=per_age,
                                  minlength=ngroups).astype(np.int32)
        s['infected_by_variant'] = {self.variant_names[i]: int(sc[_eng.S_INFECTED_BY_VARIANT + i])
                                    for i in range(self.nr_variants)}
        s['daily_contacts'] = {PLACES[i]: int(sc[_eng.S_DAILY_CONTACTS + i]) for i in range(6)}
        return s

    def generate_state(self):
        counters = self._read_counters_global()
        self._raise_on_problem(counters)
        return self.state_from_counters(counters)

    # main.pyx:2047-2101
    SAMPLE_KINDS = ('contacts_per_day', 'symptom_severity', 'incubation_period', 'illness_period',
                    'hospitalization_period', 'icu_period', 'onset_to_removed_period')

    def sample(self, what, age, severity=None, sample_size=10000):
        """10 000 draws of one per-agent quantity with the engine's samplers (host-side; each call
        advances a private stream like the reference advances its RandomPool)."""
        if what == 'infectiousness':
            # day -> infectiousness over days -100..99 (0 outside the 21-day profile, main.pyx:660-682)
            days = list(range(-100, 100))
            iot = dict(INFECTIOUSNESS_OVER_TIME)
            vals = [float(np.float32(iot.get(d, 0.0))) for d in days]
            return np.rec.fromarrays((days, vals), names=('day', 'val'))
        if what not in self.SAMPLE_KINDS:
            raise Exception('unknown sample type. supported: %s' % ', '.join(self.SAMPLE_KINDS))
        sev = -1 if severity is None else STR_TO_SEVERITY[severity]
        nrc = np.float32(self.contact_matrix.tables.nr_contacts_by_age[age])
        self._sample_calls += 1
        seed = (self._seed * 0x9E3779B97F4A7C15 + self._sample_calls) & 0xFFFFFFFFFFFFFFFF
        return self.engine.sample(self._disease, seed, self.SAMPLE_KINDS.index(what), age, sev, nrc, sample_size)

    # main.pyx:1859-1866
    def get_population_stats(self, what):
        if what not in ('dead', 'all_infected', 'all_detected'):
            raise Exception()
        counters = self._read_counters_global()
        ci = _eng.C_NAMES.index(what)
        return counters[ci * _eng.MAX_AGES: ci * _eng.MAX_AGES + self.nr_ages].copy()

    def per_age_counters(self):
        counters = self._read_counters_global()
        return {n: counters[i * _eng.MAX_AGES: i * _eng.MAX_AGES + self.nr_ages].copy()
                for i, n in enumerate(_eng.C_NAMES)}
