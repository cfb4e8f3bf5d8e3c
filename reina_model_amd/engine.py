"""ctypes binding of the engine C ABI (include/reina_hip.h) + state allocation.

`HipEngine` is the product path: it loads `csrc/libreina_hip.so` (hand-written HIP kernels for
gfx950), allocates the per-agent SoA state as PyTorch-ROCm tensors in HBM and hands their device
pointers to the library.  There is NO CPU fallback: if the shared library or a GPU is missing,
construction raises.

`Engine` itself is ABI-generic (library handle + symbol prefix + allocator) so the test-suite can
drive another implementation of the same ABI -- the CPU checker under oracle/ -- through the
identical host code.  Nothing in this package imports or references that checker.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# (REINA_HIP_LIB: a diagnostic build of the same sources, e.g. with in-kernel stamps -- tools/ only)
HIP_LIB_PATH = os.environ.get('REINA_HIP_LIB') or os.path.join(HERE, 'csrc', 'libreina_hip.so')

MAX_AGES = 128
MAX_VARIANTS = 4
MAX_ENTRIES = 96
NR_PLACES = 6
IOT_LEN = 21
MAX_IMPORT_CLASSES = 16
MAX_IMPORT_BATCHES = 16
MAX_VACCINATIONS = 16
MAX_HOSP_EVENTS = 16384
MAX_SCAN_WAVES = 8192
MAX_SHARDS = 16
MAX_RANGES = 32
PRESSURE_WORDS = MAX_SHARDS * MAX_RANGES * MAX_VARIANTS

C_NAMES = ('infected', 'detected', 'all_detected', 'all_infected', 'in_ward', 'hospitalized',
           'in_icu', 'cum_icu', 'dead', 'susceptible', 'recovered', 'vaccinated',
           'non_hospital_deaths', 'new_infections')
C_NR = len(C_NAMES)
S_AVAILABLE_BEDS, S_AVAILABLE_ICU, S_BEDS, S_ICU_UNITS, S_TOTAL_INFECTIONS, S_TOTAL_INFECTORS, \
    S_EXPOSED_PER_DAY, S_CT_CASES_PER_DAY, S_PROBLEM, S_DAY, S_UNABLE_TO_IMPORT, S_QUEUE_LEN = range(12)
S_DAILY_CONTACTS = 16
S_INFECTED_BY_VARIANT = 24
S_NR = 32
COUNTER_WORDS = C_NR * MAX_AGES + S_NR
L_NR = 48 + MAX_AGES   # REINA_L_NR (48 named words and cursors + the detections-by-age side block)
L_POOL = 26        # control word (exact attribution): nodes of infectee_pool handed out
L_XCHG_PEAK = 27   # control word (exact attribution): the fullest any exchange segment has been (against Config.xchg_cap)
L_HOSP_PEAK = 12   # control word: bed / ICU event count of the busiest day on which the events' order mattered
MAX_DAYS = 4096    # reina_day_t.day < MAX_DAYS (include/reina_hip.h: REINA_MAX_DAYS)
ABI_VERSION = 7   # (round 6: 6 = REINA_PK_SMALL_DAY, the one-launch day of a small population; 7 = exchange segments carry a trailer, exact attribution's MAIN phase asks for ONE collective)  reina_abi_version(): struct layouts of include/reina_hip.h (round 3: 32-byte cold record + inline infectee slots instead of seven per-agent arrays; round 4: the two per-agent bit planes; round 5: exact cross-shard attribution -- exchange buffers, infectee pool, reina_step_phase)
INLINE_INFECTEES = 8   # REINA_INLINE_INFECTEES
COLD_WORDS = 8         # sizeof(reina_cold_t) / 4: claim (2 words), infector, n_infected, onset_days, vacc_day, first_infectee, next_sibling
COLD_FIELDS = dict(infector=2, n_infected=3, onset_days=4, vacc_day=5, first_infectee=6, next_sibling=7)   # word of each 32-bit field
PROFILE_KINDS = ('k_open', 'k_test_trace1', 'k_vaccinate', 'k_day', 'k_hospital', 'k_hosp_sort', 'k_hosp_walk', 'k_remote', 'k_hosp_install', 'k_xchg', 'collective', 'k_small_day')
# exact cross-shard attribution (include/reina_hip.h): global ids = [shard : 4][index : 27]; the phases of a day and the
# collectives reina_step_phase asks for
GID_SHIFT = 27
GID_INDEX_MASK = (1 << GID_SHIFT) - 1
PH_OPEN, PH_TRACE, PH_MAIN, PH_END, PH_FEEDBACK, PH_NR = range(6)
X_ALLREDUCE, X_ALLTOALL = 1, 2

ABI_FUNCTIONS = ('create', 'destroy', 'bind_buffers', 'init_state', 'set_initial_state', 'upload_contact_tables',
                 'step_day', 'step_day_begin', 'step_day_end', 'step_phase', 'set_collective', 'set_alltoall', 'run_days', 'run_days_hist', 'sample', 'read_counters', 'read_history', 'profile_enable', 'profile_read',
                 'profile_read_kernels',
                 'group_create', 'group_destroy', 'group_upload_contact_tables', 'group_run_days',
                 'build_contact_tables', 'test_prims', 'last_error', 'abi_version')
TEST_PRIMS = dict(philox4=(0, 6, 4), philox2=(1, 3, 2), normal=(2, 1, 1), expf=(3, 1, 1), logf=(4, 1, 1), gamma=(5, 8, 1),
                  count_draw=(6, 4, 1))   # name -> (REINA_TP_*, words in, words out)


class Config(ctypes.Structure):
    _fields_ = [('n_agents', ctypes.c_uint32), ('nr_ages', ctypes.c_uint32),
                ('nr_variants', ctypes.c_uint32), ('max_hosp_events', ctypes.c_uint32),
                ('seed', ctypes.c_uint64),
                ('max_work_items', ctypes.c_uint32), ('max_candidates', ctypes.c_uint32),
                ('max_queue', ctypes.c_uint32), ('n_shards', ctypes.c_uint32),
                ('shard_rank', ctypes.c_uint32), ('mirror_slots', ctypes.c_uint32), ('hosp_ranges', ctypes.c_uint32),
                ('age_start', ctypes.c_int32 * (MAX_AGES + 1)),
                ('exact_attribution', ctypes.c_uint32), ('xchg_cap', ctypes.c_uint32), ('pool_cap', ctypes.c_uint32),
                ('reserved_', ctypes.c_uint32), ('shard_age_start', ctypes.c_void_p)]


_FV = ctypes.c_float * MAX_VARIANTS
_FA = ctypes.c_float * MAX_AGES


class Disease(ctypes.Structure):
    _fields_ = [(n, _FV) for n in (
        'infectiousness_multiplier', 'p_asymptomatic_infection', 'p_hospital_death_no_beds',
        'p_icu_death_no_beds', 'mean_incubation_duration', 'mean_duration_from_onset_to_death',
        'mean_duration_from_onset_to_recovery', 'ratio_of_duration_before_hospitalisation',
        'ratio_of_duration_in_ward', 'p_mask_protects_others', 'p_mask_protects_wearer')] + [
        ('infectiousness_over_time', (ctypes.c_float * (IOT_LEN + 3)) * MAX_VARIANTS),
        ('p_susceptibility', _FA * MAX_VARIANTS),
        ('p_symptomatic', _FA), ('p_severe_given_symptomatic', _FA),
        ('p_critical_given_severe', _FA), ('p_fatal_given_critical', _FA),
        ('p_death_outside_hospital', _FA),
        ('n_import_classes', ctypes.c_uint32),
        ('import_class_min_age', ctypes.c_int32 * MAX_IMPORT_CLASSES),
        ('import_class_max_age', ctypes.c_int32 * MAX_IMPORT_CLASSES),
        ('import_class_cum', ctypes.c_float * MAX_IMPORT_CLASSES)]


class ContactTablesABI(ctypes.Structure):
    _fields_ = [('nr_contacts_by_age', ctypes.c_void_p), ('count', ctypes.c_void_p),
                ('threshold', ctypes.c_void_p), ('meta', ctypes.c_void_p),
                ('mask_p', ctypes.c_void_p), ('n_ranges', ctypes.c_uint32),
                ('range_min', ctypes.c_int32 * MAX_RANGES), ('range_max', ctypes.c_int32 * MAX_RANGES)]


BUFFER_FIELDS = ('hot', 'cold', 'infectees', 'counters', 'control', 'work_items', 'candidates',
                 'queue0', 'queue1', 'level1', 'hosp_events', 'pressure', 'mirror', 'mirror_meta', 'work_counts', 'scan_lists',
                 'active_bits', 'infected_bits', 'xsend', 'xrecv', 'infectee_pool')


class Buffers(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in BUFFER_FIELDS]


class InitialState(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in (
        'incubating', 'recovered_without_illness', 'ill', 'dead', 'in_icu', 'in_ward', 'were_incubating',
        'confirmed_cases', 'confirmed_first', 'confirmed_stride')]


class ImportBatch(ctypes.Structure):
    _fields_ = [('count', ctypes.c_uint32), ('variant', ctypes.c_uint32),
                ('pre_init', ctypes.c_uint32), ('testing_mode', ctypes.c_uint32)]


class Vaccination(ctypes.Structure):
    _fields_ = [('nr', ctypes.c_uint32), ('idx_start', ctypes.c_uint32),
                ('idx_end', ctypes.c_uint32), ('slot', ctypes.c_uint32)]


class Day(ctypes.Structure):
    _fields_ = [('day', ctypes.c_uint32), ('testing_mode', ctypes.c_uint32),
                ('p_detected_anyway', ctypes.c_float), ('p_successful_tracing', ctypes.c_float),
                ('add_beds', ctypes.c_int32), ('add_icu_units', ctypes.c_int32),
                ('n_import_batches', ctypes.c_uint32), ('n_vaccinations', ctypes.c_uint32),
                ('import_batches', ImportBatch * MAX_IMPORT_BATCHES),
                ('vaccinations', Vaccination * MAX_VACCINATIONS),
                ('history_row', ctypes.c_void_p)]


HOSP_MAX_RANGES = 1024         # include/reina_hip.h: REINA_HOSP_MAX_RANGES
HOSP_MAX_BUCKET_KEYS = 4096    # keys of one priority bucket the ordered event walk holds (k_hospital.inc: HOSP_P_THREADS * HOSP_P_E)


def hosp_ranges(n_agents):
    """include/reina_hip.h: REINA_HOSP_RANGES (priority buckets of the day's bed / ICU events)"""
    r = 16
    while r < HOSP_MAX_RANGES and r * 65536 < n_agents:
        r <<= 1
    return r


def hosp_bucket_cap(n_agents, max_hosp_events, ranges=0):
    """include/reina_hip.h: REINA_HOSP_BUCKET_CAP(_R); `ranges`: reina_config_t.hosp_ranges when given"""
    return 2 * (max(MAX_HOSP_EVENTS, max_hosp_events) // (ranges or hosp_ranges(n_agents))) + 64


def default_max_hosp_events(n_agents):
    """Bed / ICU events one day may hold: one agent in 128, but no more than the event walk's buckets can take (a bucket
    holds at most HOSP_MAX_BUCKET_KEYS keys, sized for twice its mean share + 64; reina_create refuses more).  From about
    2.6e8 agents per engine instance on, the day's capacity therefore stays at 2 064 384 events (0.8 % of the agents at
    2.6e8, 0.1 % at 2e9) and a day with more fails loudly (problem 103) -- shard the population before that."""
    per_bucket = (HOSP_MAX_BUCKET_KEYS - 64) // 2
    return max(MAX_HOSP_EVENTS, min(n_agents // 128, per_bucket * hosp_ranges(n_agents)))


def hosp_event_words(n_agents, max_hosp_events, ranges=0):
    """include/reina_hip.h: REINA_HOSP_EVENT_WORDS(_R) (64-bit words of buffers.hosp_events)"""
    r = ranges or hosp_ranges(n_agents)
    return r // 2 + 2 * r + r * hosp_bucket_cap(n_agents, max_hosp_events, r)


def bits_words(n_agents):
    """include/reina_hip.h: REINA_BITS_WORDS -- uint32 words of a per-agent bit plane (buffers.active_bits / infected_bits)"""
    return ((n_agents + 511) // 512 + 1) * 16


def exchange_words(n_shards, ranges):
    """include/reina_hip.h: REINA_EXCHANGE_WORDS -- int32 words of buffers.pressure = what a sharded population all-reduces
    once per day: the infection-pressure block + every shard's table of bed / ICU event maps"""
    return PRESSURE_WORDS + (n_shards * 2 * ranges if n_shards > 1 else 0)


def xchg_words(n_shards, xchg_cap, ranges):
    """include/reina_hip.h: REINA_XCHG_WORDS -- 64-bit words of buffers.xsend / xrecv (exact attribution): per peer shard a
    count word, xchg_cap records and the trailer (two capacity words + the maps of the sender's `ranges` event buckets: what rides
    in the mid-day exchange instead of an all-reduce, ABI 7)"""
    return n_shards * (xchg_cap + 1 + 2 + ranges)


def bind_abi(lib, prefix):
    """Resolve and type every ABI entry point; raises AttributeError if one is missing."""
    f = {}
    for name in ABI_FUNCTIONS:
        f[name] = getattr(lib, prefix + name)
    vp = ctypes.c_void_p
    f['create'].argtypes = [ctypes.POINTER(Config), ctypes.POINTER(Disease), ctypes.POINTER(vp)]
    f['destroy'].argtypes = [vp]
    f['bind_buffers'].argtypes = [vp, ctypes.POINTER(Buffers)]
    f['init_state'].argtypes = [vp, ctypes.c_int32, ctypes.c_int32, vp]
    f['set_initial_state'].argtypes = [vp, ctypes.POINTER(InitialState), vp]
    f['upload_contact_tables'].argtypes = [vp, ctypes.POINTER(ContactTablesABI), vp]
    f['step_day'].argtypes = [vp, ctypes.POINTER(Day), vp]
    f['step_day_begin'].argtypes = [vp, ctypes.POINTER(Day), vp]
    f['step_day_end'].argtypes = [vp, ctypes.POINTER(Day), vp]
    f['step_phase'].argtypes = [vp, ctypes.POINTER(Day), ctypes.c_int, vp]
    f['set_collective'].argtypes = [vp, vp, vp]
    f['set_alltoall'].argtypes = [vp, vp, vp]
    f['run_days'].argtypes = [vp, ctypes.POINTER(Day), ctypes.c_uint32, vp]
    f['run_days_hist'].argtypes = [vp, ctypes.POINTER(Day), ctypes.c_uint32, vp, vp]
    f['read_counters'].argtypes = [vp, vp, vp]
    f['read_history'].argtypes = [vp, vp, ctypes.c_uint32, vp, vp]
    f['group_create'].argtypes = [ctypes.POINTER(vp), ctypes.c_uint32, ctypes.POINTER(vp)]
    f['group_destroy'].argtypes = [vp]
    f['group_upload_contact_tables'].argtypes = [vp, ctypes.POINTER(ContactTablesABI), vp]
    f['group_run_days'].argtypes = [vp, ctypes.POINTER(Day), ctypes.c_uint32, ctypes.POINTER(vp), vp]
    f['sample'].argtypes = [ctypes.POINTER(Disease), ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                            ctypes.c_float, ctypes.c_int, vp]
    f['build_contact_tables'].argtypes = [vp, vp, vp, ctypes.c_uint32, vp, ctypes.c_uint32, vp, vp, ctypes.c_uint32,
                                          ctypes.c_uint32, vp, vp, vp, vp, ctypes.c_uint32]
    f['test_prims'].argtypes = [ctypes.c_int, vp, ctypes.c_uint32, vp]
    f['profile_enable'].argtypes = [vp, ctypes.c_int]
    f['profile_read'].argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64),
                                  ctypes.POINTER(ctypes.c_double)]
    f['profile_read_kernels'].argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
    f['last_error'].restype = ctypes.c_char_p
    f['last_error'].argtypes = []
    f['abi_version'].argtypes = []
    for name in ABI_FUNCTIONS:
        if name != 'last_error':
            f[name].restype = ctypes.c_int
    got = f['abi_version']()
    if got != ABI_VERSION:
        raise EngineError('%sabi_version() = %d, this binding is written for %d (stale library? run python -m reina_model_amd.build)'
                          % (prefix, got, ABI_VERSION))
    return f


class EngineError(RuntimeError):
    pass


class NumpyAllocator:
    """Host-memory allocator (used by the test-suite to drive a CPU implementation of the ABI)."""
    device = 'cpu'

    def zeros(self, n, dtype):
        return np.zeros(n, dtype=dtype)

    def empty(self, n, dtype):
        return np.zeros(n, dtype=dtype)

    def copy_into(self, dst, offset, src):
        dst[offset:offset + len(src)] = src

    def ptr(self, arr):
        return arr.ctypes.data

    def stream(self):
        return None

    def to_host(self, arr):
        return np.array(arr, copy=True)


# short read-backs go through a small pinned block kept for the purpose -- one per host THREAD: the threaded ensemble
# (ensemble.run_ensemble(batched=False)) reads members' histories back from several threads, each on its own stream,
# and a block shared by all of them could be overwritten between one thread's synchronize() and its copy-out
import threading
_PIN_SMALL = threading.local()


class TorchAllocator:
    """HBM allocator: state lives in PyTorch-ROCm tensors, the library sees raw device pointers."""

    def __init__(self, device='cuda:0'):
        import torch
        if not torch.cuda.is_available():
            raise EngineError('no GPU visible to PyTorch-ROCm: the HIP engine has no CPU fallback')
        self.torch = torch
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self._np2t = {np.uint32: torch.int32, np.int32: torch.int32, np.float32: torch.float32,
                      np.uint64: torch.int64, np.int64: torch.int64}

    def zeros(self, n, dtype):
        return self.torch.zeros(int(n), dtype=self._np2t[dtype], device=self.device)

    def empty(self, n, dtype):
        """a buffer every word of which the caller's launches will write (a run's history rows): no memset launch"""
        return self.torch.empty(int(n), dtype=self._np2t[dtype], device=self.device)

    def copy_into(self, dst, offset, src):
        """device-to-device copy queued on the day stream (the final counter block behind a run's history rows)"""
        dst[offset:offset + src.numel()].copy_(src, non_blocking=True)

    def ptr(self, t):
        return t.data_ptr()

    def stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def to_host(self, t):
        # histories (a few MB per run, tens of MB per engine group) come back through pinned memory: the
        # pageable path runs at a fraction of the link rate and is part of every run()'s wall time.
        # (PyTorch's caching host allocator keeps the pinned block for the next run.)
        nbytes = t.numel() * t.element_size()
        if (1 << 18) <= nbytes <= (1 << 29):
            # (round 6: the read-back WAITS for the stream first and asks for its page-locked block afterwards: requested while an
            # engine group's kernels were still running, the 325 MB block of its history cost 22 ms more -- 106 ms against 83 for a
            # 128-member year, tools/ens_first_run2.py; the copy below waits for the stream anyway)
            self.torch.cuda.current_stream(self.device).synchronize()
            h = self.torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            self.torch.cuda.current_stream(self.device).synchronize()
            return h.numpy()
        if nbytes < (1 << 18) and t.dim() == 1:
            # short histories (a 20-day window is 146 KB) through ONE pinned block kept for the purpose: the pageable copy
            # of t.cpu() took 62 us of such a window's 820
            blocks = getattr(_PIN_SMALL, 'blocks', None)   # (per thread, not per Context: allocating pinned memory costs more than the copy)
            if blocks is None:
                blocks = _PIN_SMALL.blocks = {}
            key = (t.dtype, str(self.device))
            pin = blocks.get(key)
            if pin is None:
                pin = blocks[key] = self.torch.empty((1 << 18) // t.element_size(), dtype=t.dtype, pin_memory=True)
            v = pin[:t.numel()]
            v.copy_(t, non_blocking=True)
            self.torch.cuda.current_stream(self.device).synchronize()
            return v.numpy().copy()
        return t.cpu().numpy()


class Engine:
    """One engine instance = one shard of agents on one device."""

    def __init__(self, lib, prefix, allocator, config, disease):
        self.f = bind_abi(lib, prefix)
        self.alloc = allocator
        self.config = config
        self._h = ctypes.c_void_p()
        self._check(self.f['create'](ctypes.byref(config), ctypes.byref(disease), ctypes.byref(self._h)), 'create')
        n = config.n_agents
        a = allocator
        self.tensors = dict(
            hot=a.zeros(n, np.uint32),
            # everything touched only at events, one 32-byte record per agent (include/reina_hip.h: reina_cold_t) ...
            cold=a.zeros(COLD_WORDS * n, np.int32),
            # ... and the agent's first INLINE_INFECTEES infectees side by side (contact tracing reads them in one access)
            infectees=a.zeros(INLINE_INFECTEES * n, np.int32),
            counters=a.zeros(COUNTER_WORDS, np.int32),
            control=a.zeros(L_NR, np.int32),
            work_items=a.zeros(4 * config.max_work_items, np.uint32),
            candidates=a.zeros(4 * config.max_candidates, np.uint32),
            queue0=a.zeros(config.max_queue, np.uint32), queue1=a.zeros(config.max_queue, np.uint32),
            level1=a.zeros(config.max_queue, np.uint32),
            # the day's bed / ICU events by priority range: [bucket counts][bucket aggregates][keys]
            hosp_events=a.zeros(hosp_event_words(n, config.max_hosp_events, config.hosp_ranges), np.uint64),
            pressure=a.zeros(exchange_words(config.n_shards, config.hosp_ranges or hosp_ranges(n)), np.int32),
            mirror=a.zeros(MAX_RANGES * MAX_VARIANTS * config.mirror_slots if config.n_shards > 1 else 64 + 4 * 8192, np.uint64),   # (unsharded: scratch of the diagnostic builds, one row per wave of k_day)
            mirror_meta=a.zeros(2 * MAX_RANGES * MAX_VARIANTS, np.uint32),
            work_counts=a.zeros(5 * MAX_SCAN_WAVES, np.uint32),
            scan_lists=a.zeros(4 * config.max_work_items, np.uint32),
            # one bit per agent each: the hot word's ACTIVE flag again (what k_day streams on a sparse day) / ever infected
            # (what a contact looks its target up in)
            active_bits=a.zeros(bits_words(n), np.uint32), infected_bits=a.zeros(bits_words(n), np.uint32),
            # exact cross-shard attribution: the records bound for / received from the other shards, and the overflow nodes
            # of the infectee lists (otherwise placeholders: the library wants non-null pointers)
            xsend=a.zeros(xchg_words(config.n_shards, config.xchg_cap, config.hosp_ranges or hosp_ranges(n)) if config.exact_attribution else 2, np.uint64),
            xrecv=a.zeros(xchg_words(config.n_shards, config.xchg_cap, config.hosp_ranges or hosp_ranges(n)) if config.exact_attribution else 2, np.uint64),
            infectee_pool=a.zeros(2 * config.pool_cap if config.exact_attribution else 2, np.uint32),
        )
        bufs = Buffers(**{k: a.ptr(v) for k, v in self.tensors.items()})
        self._check(self.f['bind_buffers'](self._h, ctypes.byref(bufs)), 'bind_buffers')
        # the 32-bit fields of the cold record by name: strided views of `cold` (what the parity tests compare)
        rec = self.tensors['cold'].reshape(n, COLD_WORDS)
        for name, word in COLD_FIELDS.items():
            self.tensors[name] = rec[:, word]
        self._keep = []

    def _check(self, rc, what):
        if what != 'read_counters':
            self._prefetched = False      # anything launched since makes a prefetched counter block stale
        if rc != 0:
            msg = self.f['last_error']()
            raise EngineError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))

    def close(self):
        if self._h:
            self.f['destroy'](self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init_state(self, beds, icu_units):
        self._check(self.f['init_state'](self._h, int(beds), int(icu_units), self.alloc.stream()), 'init_state')

    @staticmethod
    def _tables_abi(nrc, count, threshold, meta, mask_p, ranges):
        arrs = [np.ascontiguousarray(nrc, dtype=np.float32), np.ascontiguousarray(count, dtype=np.int32),
                np.ascontiguousarray(threshold, dtype=np.uint32), np.ascontiguousarray(meta, dtype=np.uint32),
                np.ascontiguousarray(mask_p, dtype=np.float32)]
        t = ContactTablesABI(*[x.ctypes.data for x in arrs])
        t.n_ranges = len(ranges)
        for k, (lo, hi) in enumerate(ranges):
            t.range_min[k] = int(lo)
            t.range_max[k] = int(hi)
        return t, arrs

    def set_initial_state(self, ic):
        self._check(self.f['set_initial_state'](self._h, ctypes.byref(ic), self.alloc.stream()), 'set_initial_state')

    def upload_contact_tables(self, nrc, count, threshold, meta, mask_p, ranges):
        t, _keep = self._tables_abi(nrc, count, threshold, meta, mask_p, ranges)
        self._check(self.f['upload_contact_tables'](self._h, ctypes.byref(t), self.alloc.stream()), 'upload_contact_tables')

    def step_day(self, day):
        self._check(self.f['step_day'](self._h, ctypes.byref(day), self.alloc.stream()), 'step_day')

    def step_phase(self, day, phase):
        """one phase of a day (include/reina_hip.h: reina_step_phase); returns the collectives that must follow (X_*)"""
        rc = self.f['step_phase'](self._h, ctypes.byref(day), int(phase), self.alloc.stream())
        if rc < 0:
            self._check(rc, 'step_phase')
        self._prefetched = False
        return rc

    def set_alltoall(self, fn_ptr, comm_ptr):
        """in-stream exchange of exact attribution: address of an ncclAllToAll-compatible function + its communicator"""
        self._check(self.f['set_alltoall'](self._h, fn_ptr, comm_ptr), 'set_alltoall')

    def set_collective(self, fn_ptr, comm_ptr):
        """in-stream pressure all-reduce: address of an ncclAllReduce-compatible function + its communicator"""
        self._check(self.f['set_collective'](self._h, fn_ptr, comm_ptr), 'set_collective')

    def step_day_begin(self, day):
        self._check(self.f['step_day_begin'](self._h, ctypes.byref(day), self.alloc.stream()), 'step_day_begin')

    def step_day_end(self, day):
        self._check(self.f['step_day_end'](self._h, ctypes.byref(day), self.alloc.stream()), 'step_day_end')

    def run_days(self, days):
        arr = (Day * len(days))(*days)
        self._check(self.f['run_days'](self._h, arr, len(days), self.alloc.stream()), 'run_days')

    def run_day_array(self, arr, n, history_ptr):
        """`arr` is a ctypes (Day * n) array that may be shared between engines."""
        self._check(self.f['run_days_hist'](self._h, arr, n, history_ptr, self.alloc.stream()), 'run_days_hist')

    def prefetch_counters(self):
        """Queue the copy of the counter block behind what has been launched so far (pinned memory, no
        wait).  The reference's loop alternates iterate() and generate_state(): with the copy already in
        flight, the day runs on the GPU while the host is still turning the previous state into its
        dictionaries, and generate_state() finds the block waiting instead of draining the stream."""
        t = getattr(self.alloc, 'torch', None)
        if t is None:
            return
        if getattr(self, '_pf_buf', None) is None:
            self._pf_buf = t.empty(COUNTER_WORDS, dtype=t.int32, pin_memory=True)
            self._pf_ev = t.cuda.Event()
        self._pf_buf.copy_(self.tensors['counters'], non_blocking=True)
        self._pf_ev.record(t.cuda.current_stream(self.alloc.device))
        self._prefetched = True

    def read_counters(self):
        if getattr(self, '_prefetched', False):
            self._prefetched = False
            self._pf_ev.synchronize()
            return self._pf_buf.numpy().copy()
        out = np.zeros(COUNTER_WORDS, dtype=np.int32)
        self._check(self.f['read_counters'](self._h, out.ctypes.data, self.alloc.stream()), 'read_counters')
        return out

    def read_history(self, hist, rows):
        """the `rows` history rows in `hist` (device) and, behind them, the counter block as it stands: one library call -- two
        copies into page-locked memory and one wait (torch's slicing, copy_ and synchronize around the same bytes took 68 us of
        a 20-day window's 740)"""
        n = (rows + 1) * COUNTER_WORDS
        t = getattr(self.alloc, 'torch', None)
        if t is None:
            out = np.zeros(n, dtype=np.int32)
            ptr = out.ctypes.data
        else:
            pin = t.empty(n, dtype=t.int32, pin_memory=True)   # (the caching host allocator hands the block of the last run back)
            out = pin.numpy()
            ptr = pin.data_ptr()
        self._check(self.f['read_history'](self._h, self.alloc.ptr(hist), int(rows), ptr, self.alloc.stream()), 'read_history')
        return out.reshape(rows + 1, COUNTER_WORDS)

    def sample(self, disease, seed, what, age, severity, nrc, n):
        out = np.zeros(n, dtype=np.int32)
        self._check(self.f['sample'](ctypes.byref(disease), int(seed) & 0xFFFFFFFFFFFFFFFF, int(what), int(age),
                                     int(severity), float(nrc), int(n), out.ctypes.data), 'sample')
        return out

    def profile_enable(self, on=True):
        """on: False/0 off, True/1 every day, k > 1 every k-th day (day % k == 0); -k: stride k and only k_day is timed"""
        self._check(self.f['profile_enable'](self._h, int(on)), 'profile_enable')

    def profile_read(self):
        a, b, c = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_double()
        self._check(self.f['profile_read'](self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), 'profile_read')
        return dict(scan_ms_total=a.value, scan_launches=b.value, all_ms_total=c.value)

    def profile_read_kernels(self):
        """{kernel: (summed ms, timed launches)} since the last read, HIP events on the launch stream"""
        n = len(PROFILE_KINDS)
        ms, cnt = (ctypes.c_double * n)(), (ctypes.c_uint64 * n)()
        self._check(self.f['profile_read_kernels'](self._h, ms, cnt), 'profile_read_kernels')
        return {k: (ms[i], int(cnt[i])) for i, k in enumerate(PROFILE_KINDS)}


class EngineGroup:
    """K unsharded engines of the same population (different seeds) stepped with one launch per
    phase for all of them (include/reina_hip.h: reina_group_*).  Members stay usable on their own."""

    def __init__(self, engines):
        self.engines = list(engines)
        e0 = self.engines[0]
        self.f = e0.f
        self.alloc = e0.alloc
        hs = (ctypes.c_void_p * len(self.engines))(*[e._h.value for e in self.engines])
        self._h = ctypes.c_void_p()
        e0._check(self.f['group_create'](hs, len(self.engines), ctypes.byref(self._h)), 'group_create')

    def close(self):
        if self._h:
            self.f['group_destroy'](self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_contact_tables(self, nrc, count, threshold, meta, mask_p, ranges):
        t, _keep = Engine._tables_abi(nrc, count, threshold, meta, mask_p, ranges)
        self.engines[0]._check(self.f['group_upload_contact_tables'](self._h, ctypes.byref(t), self.alloc.stream()),
                               'group_upload_contact_tables')

    def run_day_array(self, arr, n, history_ptrs):
        """history_ptrs: one device pointer per member (row k of member m at ptr[m] + k rows) or None."""
        hp = None
        if history_ptrs is not None:
            hp = (ctypes.c_void_p * len(self.engines))(*[int(p) for p in history_ptrs])
        for e in self.engines:
            e._prefetched = False
        self.engines[0]._check(self.f['group_run_days'](self._h, arr, n, hp, self.alloc.stream()), 'group_run_days')


_hip_lib = None


def load_hip_library():
    """dlopen csrc/libreina_hip.so; loud failure when it has not been built (see
    __graft_entry__.build / reina_model_amd/build.py)."""
    global _hip_lib
    if _hip_lib is None:
        # PyTorch-ROCm ships its own libamdhip64 (same SONAME as /opt/rocm's).  Import torch first
        # so the process has ONE HIP runtime: our library's DT_NEEDED libamdhip64.so.7 then binds
        # to the copy torch already loaded (two runtimes in one process cannot both open the GPU).
        import torch  # noqa: F401
        if not os.path.exists(HIP_LIB_PATH):
            raise EngineError('HIP extension missing: %s (run `python -m reina_model_amd.build`); '
                              'there is no CPU fallback' % HIP_LIB_PATH)
        _hip_lib = ctypes.CDLL(HIP_LIB_PATH)
    return _hip_lib


def test_prims(f, name, records):
    """TEST HOOK (include/reina_hip.h: reina_test_prims): evaluates primitive `name` for every row of `records` (uint32
    words) with the bound library `f` (bind_abi) -- on the device for the HIP library, one lane per record."""
    what, n_in, n_out = TEST_PRIMS[name]
    rec = np.ascontiguousarray(np.asarray(records, dtype=np.uint32).reshape(-1, n_in))
    out = np.zeros((len(rec), n_out), dtype=np.uint32)
    rc = f['test_prims'](what, rec.ctypes.data, len(rec), out.ctypes.data)
    if rc != 0:
        msg = f['last_error']()
        raise EngineError('test_prims failed (%d): %s' % (rc, msg.decode() if msg else ''))
    return out


def hip_engine(config, disease, device='cuda:0'):
    alloc = TorchAllocator(device)
    return Engine(load_hip_library(), 'reina_', alloc, config, disease)
