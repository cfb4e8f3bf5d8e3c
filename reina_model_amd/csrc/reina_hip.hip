// reina_hip.hip -- MI355X (gfx950 / CDNA4) agent engine behind the C ABI of include/reina_hip.h.
//
// One simulated day (the reference's Context.iterate, cythonsim/main.pyx:2011-2018) is a short
// sequence of kernels over SoA agent state resident in HBM:
//
//   k_open          first launch: roles by arrival ticket -- history snapshot, new beds, imports (claim rounds),
//                   daily zeroing | weekly imports | test queue -> detect, contact tracing (main.pyx:1652-1699,495-558)
//   k_vaccinate     days with a programme: oldest-first cursor scan                            (:560-583)
//   k_day           every agent's 4-byte hot word streamed once: R bookkeeping, state machine (:1968-1992,395-438),
//                   and in the same waves the contact sampling of the infectious agents found: LDS-staged
//                   contact tables, target gather, Bernoulli transmission, atomicMin winner claim
//                   (:1290-1304,1525-1573,908-934)
//   k_hosp_install  workgroup 0: bed / ICU admission in priority order (:321-367,617-651) beside workgroups 1..:
//                   winners become INCUBATION (severity + incubation draw, :209-235), symptom onsets, bookkeeping
//
// The path is HBM-bound integer / RNG work: no MFMA.  Wave64 ballots compact rare events,
// per-lane Philox keys every decision by (agent, day, purpose) so results do not depend on launch
// geometry.  Integer results are bit-identical to oracle/reina_par.c by construction (same
// primitives header, FP contraction off).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

// gfx950 (MI355X / CDNA4) only: k_day's hand-written ISA (global_load_dwordx4 into hand-reserved VGPRs, global_load_lds_dword through
// M0, manual vmcnt accounting) and reina_prims.h's v_bitop3 Philox are written for this target and checked on it alone
// (tests/test_abi.py disassembles the code object); another --offload-arch must not build silently
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libreina_hip is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

#include <climits>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/reina_hip.h"
#include "reina_prims.h"
#include "reina_sample.h"
#include "reina_contacts.h"

#define CNT_IDX(c, age) ((c) * REINA_MAX_AGES + (age))
#define SC_IDX(s) (REINA_C_NR * REINA_MAX_AGES + (s))

enum { EV_HOSPITALIZE = 0, EV_TO_ICU = 1, EV_RELEASE_WARD = 2, EV_RELEASE_ICU = 3 };

// The kernels, in the order of a day (each part is included exactly once, here):
#include "k_common.inc"
#include "k_open.inc"
#include "k_testing.inc"
#include "k_scan.inc"
#include "k_hospital.inc"
#include "k_initial.inc"
#include "k_contacts.inc"
#include "k_remote.inc"
#include "k_install.inc"
#include "k_small.inc"
#define SMALL_MAX_DAYS 64u   // days of one k_small_days launch (the records' device buffer)

// ---------------------------------------------------------------------------------------------
// host side

static int grid_for(uint32_t n_items, int threads, int max_blocks) {
    long blocks = ((long)n_items + threads - 1) / threads;
    if (blocks < 1) blocks = 1;
    if (blocks > max_blocks) blocks = max_blocks;
    return (int)blocks;
}

// HIP-event timing of the day's kernels (reina_profile_enable): start/stop timestamps ride on the kernel's
// own dispatch packet (hipExtLaunchKernelGGL), no extra stream commands.  On a profiled day ONE kind of kernel
// is timed (the kinds take turns), so the cost of timestamped dispatches is spread thinly over the run.
// (the events only carry timestamps, read after a synchronisation: a DEVICE-scope release where they are recorded -- the default, a
// system-scope release, writes the caches back behind every timestamped kernel)
#define REINA_TIMING_EVENT_FLAGS (hipEventReleaseToDevice)
static bool take_event_pair(reina_engine *e, size_t *a, size_t *b) {
    while (e->ev_used + 2 > e->ev_pool.size()) {
        if (e->ev_pool.size() >= reina_engine::MAX_EVENTS) return false;   // pool exhausted: this launch goes untimed
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, REINA_TIMING_EVENT_FLAGS) != hipSuccess || !ev) return false;
        e->ev_pool.push_back(ev);
    }
    *a = e->ev_used++;
    *b = e->ev_used++;
    return true;
}

// which kind of kernel carries timestamps on `day` (-1: none)
static int profiled_kind(const reina_engine *e, uint32_t day) {
    if (!e->profile) return -1;
    const uint32_t stride = e->profile_stride;
    if (stride <= 1) return REINA_PK_NR;        // every kernel, every day
    const uint32_t ph = day % stride;
    // the four kernels of every day at evenly spaced phases; the occasional ones ride with k_open's phase
    if (ph == 0) return REINA_PK_DAY;
    if (e->profile_day_only) return -1;   // (enable < 0: only the dominant kernel carries timestamps -- a third of the instrument's cost)
    if (ph == stride / 4) return REINA_PK_OPEN;
    if (ph == stride / 2) return REINA_PK_HOSPITAL;
    if (ph == stride / 2 + stride / 4) return REINA_PK_INSTALL;
    return -1;
}
static bool kind_timed(int today, int kind) {
    if (today < 0) return false;
    if (today == REINA_PK_NR || today == kind) return true;
    // the one launch of a small population's day (k_small_day) is timed on k_day's days and on the event walk's: twice per stride
    if (kind == REINA_PK_SMALL_DAY) return today == REINA_PK_DAY || today == REINA_PK_HOSPITAL;
    // kernels that do not run every day are timed on k_open's days
    if (today == REINA_PK_HOSPITAL && (kind == REINA_PK_REMOTE || kind == REINA_PK_HOSP_SORT || kind == REINA_PK_HOSP_WALK || kind == REINA_PK_XCHG || kind == REINA_PK_COLLECTIVE)) return true;
    return today == REINA_PK_OPEN && (kind == REINA_PK_TRACE1 || kind == REINA_PK_VACCINATE);
}

#define LAUNCH_TIMED(e, today, kind, kernel, grid, block, lds, stream, ...)                                              \
    do {                                                                                                                 \
        size_t ev_a_, ev_b_;                                                                                             \
        if (kind_timed(today, kind) && take_event_pair(e, &ev_a_, &ev_b_)) {                                             \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, (e)->ev_pool[ev_a_], (e)->ev_pool[ev_b_], 0, __VA_ARGS__); \
            (e)->kpairs[kind].emplace_back(ev_a_, ev_b_);                                                                \
        } else {                                                                                                         \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                           \
        }                                                                                                                \
    } while (0)

// a kernel of the day: the GROUP instantiation for an engine group (members found in `refs`), else the single engine's
// MemberRef by value (k_common.inc: DAY_KERNEL)
#define LAUNCH_DAY(e, today, kind, kernel, grid, block, lds, stream, ...)                                                \
    do {                                                                                                                 \
        if (K > 1) LAUNCH_TIMED(e, today, kind, (kernel<true, false>), grid, block, lds, stream, refs, (e)->h_ref, __VA_ARGS__);  \
        else if ((e)->exact) LAUNCH_TIMED(e, today, kind, (kernel<false, true>), grid, block, lds, stream, refs, (e)->h_ref, __VA_ARGS__); \
        else LAUNCH_TIMED(e, today, kind, (kernel<false, false>), grid, block, lds, stream, refs, (e)->h_ref, __VA_ARGS__); \
    } while (0)

static void resolve_profile(reina_engine *e) {
    for (int k = 0; k < REINA_PK_NR; k++) {
        size_t idx = 0;
        for (auto &p : e->kpairs[k]) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, e->ev_pool[p.first], e->ev_pool[p.second]) == hipSuccess) {
                e->k_ms[k] += ms;
                // (a k_small_days launch runs a stretch of days: the kind counts DAYS, so that ms / count is a day's time like the others')
                e->k_launches[k] += k == REINA_PK_SMALL_DAY && idx < e->small_pair_days.size() ? e->small_pair_days[idx] : 1u;
            }
            idx++;
        }
        if (k == REINA_PK_SMALL_DAY) e->small_pair_days.clear();
        e->kpairs[k].clear();
    }
    e->ev_used = 0;
}

// frees what reina_create / reina_group_create had acquired when a later step fails
static void free_engine(reina_engine *e) {
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    for (size_t k = 0; k < e->stage.size(); k++) {
        (void)hipEventSynchronize(e->stage_ev[k]);
        (void)hipEventDestroy(e->stage_ev[k]);
        (void)hipHostFree(e->stage[k]);
    }
    if (e->d_params) (void)hipFree(e->d_params);
    if (e->d_tables) (void)hipFree(e->d_tables);
    if (e->d_ref) (void)hipFree(e->d_ref);
    if (e->d_bar) (void)hipFree(e->d_bar);
    if (e->d_days) (void)hipFree(e->d_days);
    for (size_t k = 0; k < e->days_stage.size(); k++) {
        (void)hipEventSynchronize(e->days_stage_ev[k]);
        (void)hipEventDestroy(e->days_stage_ev[k]);
        (void)hipHostFree(e->days_stage[k]);
    }
    if (e->counted_live) g_live_engines--;
    delete e;
}
#define HIP_CHECK_OR(x, cleanup)                                                             \
    do {                                                                                     \
        hipError_t _e = (x);                                                                 \
        if (_e != hipSuccess) {                                                              \
            g_last_error = std::string(#x) + ": " + hipGetErrorString(_e);                   \
            cleanup;                                                                         \
            return REINA_E_HIP;                                                              \
        }                                                                                    \
    } while (0)

// One row of contact-count thresholds + its guide table (reina_contacts.h: 105 erfc + log in double, 4 us a row, 80 us for the
// 20 distinct rows of a table -- host time that a short run cannot hide behind the GPU: the 20-day window of the round
// driver's bench contains one table change, and it cost 5 us per step).  A row depends on one float only, and the same
// values come back all the time -- a table change that closes schools leaves the other ages' rows as they were; every seed
// of an ensemble, every scenario run of a serving process rebuilds the same tables on the same dates -- so rows are kept,
// process-wide, keyed by the float's bits (at most 4096 rows, 1.5 MB; beyond that rows are computed and not kept).
struct CountRow { uint32_t thr[REINA_COUNT_WORDS]; uint8_t guide[256]; };
static void count_row_for(float nr_contacts, uint32_t *thr, uint8_t *guide) {
    static std::mutex mu;
    static std::unordered_map<uint32_t, CountRow> rows;
    uint32_t key;
    std::memcpy(&key, &nr_contacts, 4);
    // (REINA_COUNT_ROW_CACHE=0: every row computed afresh -- what a table change costs a process that has not seen the
    // scenario's mobility values before; bench.py reports that figure beside the warm one)
    const char *cache_env = std::getenv("REINA_COUNT_ROW_CACHE");
    const bool use_cache = !(cache_env && cache_env[0] == '0');
    if (use_cache) {
        std::lock_guard<std::mutex> g(mu);
        auto it = rows.find(key);
        if (it != rows.end()) {
            std::memcpy(thr, it->second.thr, sizeof(it->second.thr));
            std::memcpy(guide, it->second.guide, 256);
            return;
        }
    }
    CountRow row;
    rc_count_thresholds(nr_contacts, row.thr);
    int idx = 0;
    for (uint32_t b = 0; b < 256; b++) {
        while (idx < REINA_COUNT_FULL && row.thr[idx] <= (b << 24)) idx++;   // thresholds are non-decreasing
        row.guide[b] = (uint8_t)idx;
    }
    std::memcpy(thr, row.thr, sizeof(row.thr));
    std::memcpy(guide, row.guide, 256);
    if (!use_cache) return;
    std::lock_guard<std::mutex> g(mu);
    if (rows.size() < 4096) rows.emplace(key, row);
}

extern "C" {

int reina_abi_version(void) { return 7; }

#ifdef REINA_ABLATE
int reina_debug_ablate(uint32_t bits) {   // diagnostic builds only (tools/ablate_day.py)
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ablate_dev), &bits, sizeof(bits)) == hipSuccess ? 0 : -1;
}
#endif

int reina_build_contact_tables(const double *base, const int32_t *row_page, const int32_t *row_place, uint32_t n_rows,
                               const double *mobility, uint32_t n_mobility, const int32_t *rows_mat,
                               const int32_t *sorted_mat, uint32_t n_ages, uint32_t n_entries, double *totals_out,
                               double *cum_out, float *nrc_out, uint32_t *thr_out, uint32_t thr_stride) {
    const int rc = reina_build_contact_tables_impl(base, row_page, row_place, n_rows, mobility, n_mobility, rows_mat, sorted_mat,
                                                   n_ages, n_entries, totals_out, cum_out, nrc_out, thr_out, thr_stride);
    return rc == 0 ? REINA_OK : REINA_E_INVALID;
}

// test hook: one primitive of reina_prims.h evaluated on the device, one lane per record
__global__ __launch_bounds__(256) void k_test_prims(int what, const uint32_t *in, uint32_t n, uint32_t n_in, uint32_t n_out, uint32_t *out) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        uint32_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[4] = {0, 0, 0, 0};
        for (uint32_t j = 0; j < n_in; j++) a[j] = in[(size_t)k * n_in + j];
        rp_test_prim(what, a, o);
        for (uint32_t j = 0; j < n_out; j++) out[(size_t)k * n_out + j] = o[j];
    }
}

int reina_test_prims(int what, const uint32_t *in_host, uint32_t n, uint32_t *out_host) {
    uint32_t n_in = 0, n_out = 0;
    rp_test_prim_words(what, &n_in, &n_out);
    if (!n_in || !in_host || !out_host) return REINA_E_INVALID;
    if (n == 0) return REINA_OK;
    uint32_t *d_in = nullptr, *d_out = nullptr;
    HIP_CHECK(hipMalloc(&d_in, (size_t)n * n_in * 4));
    HIP_CHECK_OR(hipMalloc(&d_out, (size_t)n * n_out * 4), (void)hipFree(d_in));
#define TP_CLEAN { (void)hipFree(d_in); (void)hipFree(d_out); }
    HIP_CHECK_OR(hipMemcpy(d_in, in_host, (size_t)n * n_in * 4, hipMemcpyHostToDevice), TP_CLEAN);
    hipLaunchKernelGGL(k_test_prims, dim3(grid_for(n, 256, 4096)), dim3(256), 0, nullptr, what, d_in, n, n_in, n_out, d_out);
    HIP_CHECK_OR(hipGetLastError(), TP_CLEAN);
    HIP_CHECK_OR(hipMemcpy(out_host, d_out, (size_t)n * n_out * 4, hipMemcpyDeviceToHost), TP_CLEAN);
    TP_CLEAN;
#undef TP_CLEAN
    return REINA_OK;
}

int reina_sample(const reina_disease_t *disease, uint64_t seed, int what, int age, int severity,
                 float nr_contacts_of_age, int n, int32_t *out) {
    return reina_sample_impl(disease, seed, what, age, severity, nr_contacts_of_age, n, out);
}
const char *reina_last_error(void) { return g_last_error.c_str(); }

int reina_create(const reina_config_t *cfg, const reina_disease_t *disease, reina_engine_t **out) {
    if (!cfg || !disease || !out) return REINA_E_INVALID;
    if (cfg->nr_ages == 0 || cfg->nr_ages > REINA_MAX_AGES || cfg->nr_variants == 0 ||
        cfg->nr_variants > REINA_MAX_VARIANTS) {
        g_last_error = "nr_ages / nr_variants out of range";
        return REINA_E_INVALID;
    }
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    if (ndev == 0) {
        g_last_error = "no HIP device";
        return REINA_E_HIP;
    }
    for (uint32_t a = 0; a < cfg->nr_ages; a++)
        if (cfg->age_start[a] < 0 || cfg->age_start[a] > cfg->age_start[a + 1] || (uint32_t)cfg->age_start[a + 1] > cfg->n_agents) {
            g_last_error = "age_start must be non-decreasing and end at n_agents";
            return REINA_E_INVALID;
        }
    reina_engine *e = new reina_engine();
    e->cfg = *cfg;
    {   // compute units of the current device: grids of one-workgroup-per-CU kernels are sized to it
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 1)
            e->n_cus = (uint32_t)cus;
        if (const char *w = std::getenv("REINA_VACC_ONE_WG")) e->vacc_one_wg = std::atoi(w) != 0;   // (the single-workgroup vaccination pass at every size: the tests compare)
        if (const char *w = std::getenv("REINA_NO_PLACE_GROUPS")) e->no_place_groups = std::atoi(w) != 0;
        if (const char *w = std::getenv("REINA_LDS_ROWS_CAP")) {
            const int v = std::atoi(w);
            if (v >= 1) e->lds_rows_cap = (uint32_t)v;
        }
        if (const char *w = std::getenv("REINA_WALK_DIV")) {
            const int v = std::atoi(w);
            if (v >= 16 && v <= 1024) e->walk_div = (uint32_t)v;
        }
        if (const char *w = std::getenv("REINA_IMPORT_WGS")) {
            const int v = std::atoi(w);
            if (v >= 1 && v <= 16) e->import_wgs = (uint32_t)v;
        }
        // k_day streams the ACTIVE bit plane instead of the hot words when the population is large enough for that form to pay
        // (day_is_sparse).  REINA_DAY_MODE = dense | sparse forces one form whatever the size (the tests run every scenario in
        // both), alternate switches between them day by day; REINA_DAY_FLAGS are k_day's measurement switches
        if (const char *w = std::getenv("REINA_DAY_MODE")) {
            if (!std::strcmp(w, "dense")) e->day_mode = 1;
            else if (!std::strcmp(w, "sparse")) e->day_mode = 2;
            else if (!std::strcmp(w, "alternate")) e->day_mode = 3;
        }
        if (const char *w = std::getenv("REINA_DAY_FLAGS")) e->day_flags |= (uint32_t)std::atoi(w);
        if (const char *w = std::getenv("REINA_EXPORT")) e->export_by_copies = !std::strcmp(w, "memcpy");
        if (const char *w = std::getenv("REINA_OPEN_TICKETS")) e->open_tickets = std::atoi(w) != 0;   // (the tests' handle on the ticket path of a single engine)
        if (const char *w = std::getenv("REINA_IMPORTS_IN_OPEN")) e->imports_in_open = std::atoi(w) != 0;   // (the round-3 placement, for comparison)
        // stretches of days of a small unsharded population as ONE launch (k_small.inc): REINA_FUSED_DAY=1 switches it on (off by
        // default: measured slower than the three launches a day; the tests run scenario families both ways), REINA_FUSED_WGS its
        // workgroups (measurement handle; 8..64)
        if (const char *w = std::getenv("REINA_FUSED_DAY")) e->fused_day = std::atoi(w) != 0;
        if (const char *w = std::getenv("REINA_FUSED_WGS")) {
            const int v = std::atoi(w);
            if (v >= 8 && v <= 64) e->small_wgs = (uint32_t)v;
        }
    }
    std::memset(&e->h_params, 0, sizeof(DevParams));
    std::memset(&e->h_tables, 0, sizeof(Tables));
    e->h_params.dis = *disease;
    std::memcpy(e->h_params.age_start, cfg->age_start, sizeof(cfg->age_start));
    e->h_params.n_agents = cfg->n_agents;
    e->h_params.nr_ages = cfg->nr_ages;
    e->h_params.nr_variants = cfg->nr_variants;
    e->cfg.n_shards = cfg->n_shards ? cfg->n_shards : 1;
    if (e->cfg.n_shards > REINA_MAX_SHARDS || e->cfg.shard_rank >= e->cfg.n_shards) {
        g_last_error = "n_shards / shard_rank out of range";
        delete e;
        return REINA_E_INVALID;
    }
    const uint64_t shard_seed = rp_shard_seed(cfg->seed, e->cfg.shard_rank);
    e->h_params.k0 = (uint32_t)shard_seed;
    e->h_params.k1 = (uint32_t)(shard_seed >> 32);
    for (uint32_t v = 0; v < REINA_MAX_VARIANTS; v++) {
        uint32_t m = 0;
        for (int k = 0; k < REINA_IOT_LEN; k++)
            if (disease->infectiousness_over_time[v][k] != 0.0f) m |= 1u << k;
        e->h_params.iot_mask[v] = m;
    }
    e->h_params.n_shards = e->cfg.n_shards;
    e->h_params.shard_rank = e->cfg.shard_rank;
    e->h_params.mirror_slots = cfg->mirror_slots ? cfg->mirror_slots : 64;
    if (e->h_params.mirror_slots & (e->h_params.mirror_slots - 1)) {
        g_last_error = "mirror_slots must be a power of two";
        delete e;
        return REINA_E_INVALID;
    }
    for (uint32_t v = 0; v < cfg->nr_variants; v++) {
        float m = 0.0f;
        for (uint32_t a = 0; a < cfg->nr_ages; a++)
            if (disease->p_susceptibility[v][a] > m) m = disease->p_susceptibility[v][a];
        e->h_params.psus_max[v] = m;
    }
    // exact cross-shard attribution (include/reina_hip.h): global ids in the link fields, records exchanged between the shards
    e->exact = cfg->exact_attribution != 0 && e->cfg.n_shards > 1;
    e->h_params.gid_mask = 0xFFFFFFFFu;
    if (e->exact) {
        if (!cfg->shard_age_start || cfg->xchg_cap == 0 || cfg->pool_cap == 0 || cfg->n_agents > RP_GID_INDEX_MASK) {
            g_last_error = "exact attribution needs shard_age_start, xchg_cap > 0, pool_cap > 0 and fewer than 2^27 agents per shard";
            delete e;
            return REINA_E_INVALID;
        }
        e->h_params.exact = 1u;
        e->h_params.gid_base = e->cfg.shard_rank << RP_GID_SHIFT;
        e->h_params.gid_mask = RP_GID_INDEX_MASK;
        e->h_params.xchg_cap = cfg->xchg_cap;
        e->h_params.pool_cap = cfg->pool_cap;
        std::memcpy(e->h_params.shard_age_start, cfg->shard_age_start, sizeof(int32_t) * (size_t)e->cfg.n_shards * (REINA_MAX_AGES + 1));
        for (uint32_t sh = 0; sh < e->cfg.n_shards; sh++) {
            const uint64_t ss = rp_shard_seed(cfg->seed, sh);
            e->h_params.shard_k0[sh] = (uint32_t)ss;
            e->h_params.shard_k1[sh] = (uint32_t)(ss >> 32);
        }
    }
    e->cfg.shard_age_start = nullptr;   // (the caller's array is not kept)
    e->h_params.max_work_items = cfg->max_work_items;
    e->h_params.max_candidates = cfg->max_candidates;
    {   // the space above the per-wave regions: half (at most 2^20 records) for records that overflow a wave's
        // region, the rest for the cross-shard candidates of k_remote
        const uint32_t extra = cfg->max_candidates > cfg->max_work_items ? cfg->max_candidates - cfg->max_work_items : 0u;
        uint32_t ovf = extra / 2;
        if (ovf > (1u << 20)) ovf = 1u << 20;
        e->h_params.cand_ovf_cap = ovf;
        e->h_params.cand_ovf_base = cfg->max_candidates - ovf;
    }
    e->h_params.max_queue = cfg->max_queue;
    e->h_params.max_hosp_events = cfg->max_hosp_events > REINA_MAX_HOSP_EVENTS ? cfg->max_hosp_events : REINA_MAX_HOSP_EVENTS;
    e->h_params.hosp_ranges = cfg->hosp_ranges ? cfg->hosp_ranges : REINA_HOSP_RANGES(cfg->n_agents);
    e->h_params.xchg_stride = (uint32_t)REINA_XCHG_SEG_WORDS(e->h_params.xchg_cap, e->h_params.hosp_ranges);   // (exact attribution: a segment = count + records + trailer)
    if (e->h_params.hosp_ranges < 16 || e->h_params.hosp_ranges > REINA_HOSP_MAX_RANGES || (e->h_params.hosp_ranges & (e->h_params.hosp_ranges - 1))) {
        g_last_error = "hosp_ranges must be a power of two in [16, REINA_HOSP_MAX_RANGES]";
        delete e;
        return REINA_E_INVALID;
    }
    e->h_params.hosp_range_bits = 0;
    while ((1u << e->h_params.hosp_range_bits) < e->h_params.hosp_ranges) e->h_params.hosp_range_bits++;
    e->h_params.hosp_bucket_cap = REINA_HOSP_BUCKET_CAP_R(e->h_params.hosp_ranges, cfg->max_hosp_events);
    // (a sharded population always takes the bucket-per-wave walk: its shards exchange per-bucket maps)
    e->h_params.hosp_parallel = (cfg->n_agents > REINA_HOSP_SMALL_AGENTS || e->cfg.n_shards > 1) ? 1u : 0u;
    e->exchange_words = (uint32_t)REINA_EXCHANGE_WORDS(e->cfg.n_shards, e->h_params.hosp_ranges);
    if (e->h_params.hosp_parallel && e->h_params.hosp_bucket_cap > REINA_HOSP_MAX_BUCKET_KEYS) {
        g_last_error = "max_hosp_events too large for this population: a bucket of the event walk holds at most 4096 keys "
                       "(pass at most REINA_HOSP_MAX_EVENTS_FOR(n_agents), include/reina_hip.h)";
        delete e;
        return REINA_E_INVALID;
    }
    HIP_CHECK_OR(hipMalloc(&e->d_params, sizeof(DevParams)), free_engine(e));
    HIP_CHECK_OR(hipMalloc(&e->d_tables, sizeof(Tables)), free_engine(e));
    HIP_CHECK_OR(hipMalloc(&e->d_ref, sizeof(MemberRef)), free_engine(e));
    HIP_CHECK_OR(hipMemcpy(e->d_params, &e->h_params, sizeof(DevParams), hipMemcpyHostToDevice), free_engine(e));
    HIP_CHECK_OR(hipMemcpy(e->d_tables, &e->h_tables, sizeof(Tables), hipMemcpyHostToDevice), free_engine(e));
    // (every instantiation of every day kernel that takes dynamic LDS -- single engine, group member, shard under exact attribution --,
    // sized for the larger of its launch shapes)
    const size_t walk_lds = (size_t)HOSP_P_THREADS * HOSP_P_E * 8, small_lds = (size_t)REINA_MAX_HOSP_EVENTS * 8;
    const int day_lds = (int)day_shared_bytes(REINA_LDS_ROWS, REINA_LDS_CROWS, REINA_MAX_SHARDS), inst_lds = (int)(walk_lds > small_lds ? walk_lds : small_lds);
#define SET_LDS(kernel, bytes)                                                                                                                       \
    HIP_CHECK_OR(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes), free_engine(e)); \
    HIP_CHECK_OR(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes), free_engine(e));  \
    HIP_CHECK_OR(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes), free_engine(e))
#define SET_LDS_DAY(SP)                                                                                                                              \
    HIP_CHECK_OR(hipFuncSetAttribute(reinterpret_cast<const void *>(k_day<false, false, SP>), hipFuncAttributeMaxDynamicSharedMemorySize, day_lds), free_engine(e)); \
    HIP_CHECK_OR(hipFuncSetAttribute(reinterpret_cast<const void *>(k_day<true, false, SP>), hipFuncAttributeMaxDynamicSharedMemorySize, day_lds), free_engine(e));  \
    HIP_CHECK_OR(hipFuncSetAttribute(reinterpret_cast<const void *>(k_day<false, true, SP>), hipFuncAttributeMaxDynamicSharedMemorySize, day_lds), free_engine(e))
    SET_LDS_DAY(false);
    SET_LDS_DAY(true);
#undef SET_LDS_DAY
    {
        // k_day addresses v104..v127 by hand (k_contacts.inc): the code object must allocate all 128 registers to every instantiation.
        // tests/test_abi.py checks that in the disassembly of the build it runs on; this is the check a box without the tests makes
        // (round-5 verdict: a compiler that stopped honouring the clobber list would hand the kernel fewer registers than it names)
        const void *day_kernels[6] = {reinterpret_cast<const void *>(k_day<false, false, false>), reinterpret_cast<const void *>(k_day<false, false, true>),
                                      reinterpret_cast<const void *>(k_day<true, false, false>), reinterpret_cast<const void *>(k_day<true, false, true>),
                                      reinterpret_cast<const void *>(k_day<false, true, false>), reinterpret_cast<const void *>(k_day<false, true, true>)};
        for (const void *k : day_kernels) {
            hipFuncAttributes fa;
            HIP_CHECK_OR(hipFuncGetAttributes(&fa, k), free_engine(e));
            if (fa.numRegs != 128 || fa.localSizeBytes != 0) {
                g_last_error = "k_day was built with " + std::to_string(fa.numRegs) + " VGPRs / " + std::to_string(fa.localSizeBytes) +
                               " bytes of scratch: it names v104..v127 by hand and needs 128 / 0 (rebuild with the ROCm release the sources were written for)";
                free_engine(e);
                return REINA_E_HIP;
            }
        }
    }
    SET_LDS(k_hosp_presort, (int)walk_lds);
    SET_LDS(k_hosp_install, inst_lds);
#undef SET_LDS
    {
        // the fused day of a small population: its LDS block (the stream's image or the hospital role's keys + maps, whichever is
        // larger) beside the static LDS of the three phases' functions must fit one compute unit; its barrier counter
        hipFuncAttributes fa;
        HIP_CHECK_OR(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_small_days)), free_engine(e));
        e->small_static_lds = fa.sharedSizeBytes;
        const size_t most = 160u * 1024u > fa.sharedSizeBytes ? 160u * 1024u - fa.sharedSizeBytes : 0u;
        HIP_CHECK_OR(hipFuncSetAttribute(reinterpret_cast<const void *>(k_small_days), hipFuncAttributeMaxDynamicSharedMemorySize, (int)most), free_engine(e));
        HIP_CHECK_OR(hipMalloc(&e->d_bar, 256), free_engine(e));
        HIP_CHECK_OR(hipMemset(e->d_bar, 0, 256), free_engine(e));
        HIP_CHECK_OR(hipMalloc(&e->d_days, sizeof(SmallDayRec) * SMALL_MAX_DAYS), free_engine(e));
    }
    e->counted_live = true;
    g_live_engines++;
    *out = e;
    return REINA_OK;
}

int reina_destroy(reina_engine_t *e) {
    if (!e) return REINA_E_INVALID;
    free_engine(e);   // teardown: nothing useful can be done about a failing free
    return REINA_OK;
}

int reina_bind_buffers(reina_engine_t *e, const reina_buffers_t *b) {
    if (!e || !b) return REINA_E_INVALID;
    const void *const *p = reinterpret_cast<const void *const *>(b);
    for (size_t k = 0; k < sizeof(reina_buffers_t) / sizeof(void *); k++)
        if (!p[k]) {
            g_last_error = "null buffer pointer";
            return REINA_E_INVALID;
        }
    e->buf = *b;
    e->bound = true;
    MemberRef r;
    r.P = e->d_params;
    r.T = e->d_tables;
    r.B = e->buf;
    r.history_base = nullptr;

    HIP_CHECK(hipMemcpy(e->d_ref, &r, sizeof(MemberRef), hipMemcpyHostToDevice));
    e->h_ref = r;
    return REINA_OK;
}

int reina_init_state(reina_engine_t *e, int32_t beds, int32_t icu, void *stream) {
    if (!e || !e->bound) return REINA_E_NOT_BOUND;
    hipStream_t s = (hipStream_t)stream;
    e->init_beds = beds;
    hipLaunchKernelGGL(k_init, dim3(grid_for(e->cfg.n_agents, 256, 4096), 1), dim3(256), 0, s, e->d_ref, beds, icu);
    HIP_CHECK(hipGetLastError());
    return REINA_OK;
}

int reina_set_initial_state(reina_engine_t *e, const reina_initial_state_t *ic, void *stream) {
    if (!e || !ic) return REINA_E_INVALID;
    if (!e->bound) return REINA_E_NOT_BOUND;
    // slot j of the walk over [0, were_incubating) is bound for ICU iff it lies behind the incubating / recovered / ill / dead
    // slots: a walk that stops short of them (fewer recovered than incubating people, main.pyx:1456-1463) constructs
    const uint64_t first_icu_slot = (uint64_t)ic->incubating + ic->recovered_without_illness + ic->ill + ic->dead;
    if (e->cfg.n_shards <= 1 && ic->in_icu > 0 && (uint64_t)ic->were_incubating > first_icu_slot && e->init_beds == 0) {
        // (the reference raises AssertionError out of Context.__init__: an agent bound for ICU is refused a bed and
        // Population.transfer_to_icu asserts state == HOSPITALIZED, main.pyx:1495 -> :350 -> :1603; a shard cannot tell --
        // its own share of the beds may be 0 while the population has some -- and leaves the check to its caller)
        g_last_error = "initial population condition: people in ICU but a hospital without beds (the reference refuses it)";
        return REINA_E_INVALID;
    }
    hipLaunchKernelGGL(k_initial_state, dim3(1, 1), dim3(PRO_THREADS), 0, (hipStream_t)stream, e->d_ref, *ic);
    HIP_CHECK(hipGetLastError());
    return REINA_OK;
}

int reina_upload_contact_tables(reina_engine_t *e, const reina_contact_tables_t *t, void *stream) {
    if (!e || !t) return REINA_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t A = e->cfg.nr_ages;
    if (t->n_ranges > REINA_MAX_RANGES) {
        g_last_error = "more than REINA_MAX_RANGES contact ranges";
        return REINA_E_INVALID;
    }
    for (uint32_t a = 0; a < A; a++)
        if (t->count[a] < 0 || t->count[a] > REINA_MAX_ENTRIES) {
            g_last_error = "contact entries per age must be in [0, REINA_MAX_ENTRIES]";
            return REINA_E_INVALID;
        }
    if (e->cfg.n_shards > 1 && t->n_ranges >= REINA_MAX_RANGES) {
        g_last_error = "a sharded engine takes at most REINA_MAX_RANGES - 1 contact ranges (the last range's pressure words carry free capacity)";
        return REINA_E_INVALID;
    }
    std::memcpy(e->h_params.nrc, t->nr_contacts_by_age, sizeof(float) * A);
    std::memcpy(e->h_params.tcount, t->count, sizeof(int32_t) * A);
    std::memcpy(e->h_params.mask_p, t->mask_p, sizeof(float) * A * 8);
    e->h_params.n_ranges = t->n_ranges;
    std::memcpy(e->h_params.range_min, t->range_min, sizeof(t->range_min));
    std::memcpy(e->h_params.range_max, t->range_max, sizeof(t->range_max));
    {   // distinct contact rows (entry count, thresholds, meta words): ages of one class of the matrix share a row
        Tables &T = e->h_tables;
        uint32_t n_rows = 0;
        T.grouped = 1;
        for (uint32_t a = 0; a < A; a++) {
            const uint32_t *thr_a = t->threshold + (size_t)a * REINA_MAX_ENTRIES, *meta_a = t->meta + (size_t)a * REINA_MAX_ENTRIES;
            const size_t used = sizeof(uint32_t) * (size_t)t->count[a];
            uint32_t r = 0;
            for (; r < n_rows; r++)
                if (T.rcount[r] == t->count[a] && std::memcmp(T.thr[r], thr_a, used) == 0 && std::memcmp(T.meta[r], meta_a, used) == 0) break;
            if (r == n_rows) {
                std::memcpy(T.thr[r], thr_a, sizeof(uint32_t) * REINA_MAX_ENTRIES);
                std::memcpy(T.meta[r], meta_a, sizeof(uint32_t) * REINA_MAX_ENTRIES);
                T.rcount[r] = t->count[a];
                const int cnt = t->count[a];
                int idx = 0;
                for (uint32_t b = 0; b < 256; b++) {
                    const uint32_t floor_ = b << 24;
                    while (idx < cnt - 1 && T.thr[r][idx] <= floor_) idx++;   // thresholds are non-decreasing
                    T.guide[r][b] = (uint8_t)(cnt > 0 ? idx : 0);
                }
                // place groups (k_common.inc: Tables::grp)
                uint32_t *G = T.grp[r];
                for (int q = 0; q < 5; q++) G[q] = 0xFFFFFFFFu;
                G[5] = G[6] = G[7] = 0;
                uint32_t groups = 0, last_place = 0;
                for (int en = 0; en < cnt; en++) {
                    const uint32_t place = T.meta[r][en] & 0xFFu;
                    if (en == 0 || place != last_place) {
                        if (en > 0 && groups <= 5) G[groups - 1] = T.thr[r][en - 1];
                        if (groups < 6) G[5] |= (place * 5u) << (5u * groups);
                        groups++;
                        last_place = place;
                    }
                    if (place >= REINA_NR_PLACES) groups = 99;   // (cannot be packed; never with the places of the model)
                }
                for (uint32_t q = groups; q < 6 && groups > 0; q++) G[5] |= (last_place * 5u) << (5u * q);
                G[6] = groups;
                if (groups > 6) T.grouped = 0;
                n_rows++;
            }
            T.row_of_age[a] = (uint8_t)r;
        }
        T.n_rows = n_rows;
        if (e->no_place_groups) T.grouped = 0;
        // contact-count thresholds: one row per distinct nr_contacts_by_age value
        uint32_t n_crows = 0;
        float crow_value[REINA_MAX_AGES];
        for (uint32_t a = 0; a < A; a++) {
            uint32_t r = 0;
            for (; r < n_crows; r++)
                if (std::memcmp(&crow_value[r], &t->nr_contacts_by_age[a], sizeof(float)) == 0) break;
            if (r == n_crows) {
                crow_value[r] = t->nr_contacts_by_age[a];
                count_row_for(crow_value[r], T.cthr[r], T.cguide[r]);
                n_crows++;
            }
            T.crow_of_age[a] = (uint8_t)r;
        }
        T.n_crows = n_crows;
        // coarse index -> age map (the age of a sampled target: one table read + 0-1 steps instead of a search)
        uint32_t shift = 0;
        while (((uint64_t)e->cfg.n_agents >> shift) >= REINA_AGE_BLOCKS) shift++;
        T.age_shift = shift;
        uint32_t a = 0;
        for (uint32_t b = 0; b < REINA_AGE_BLOCKS; b++) {
            const uint64_t i = (uint64_t)b << shift;
            while (a + 1 < A && (uint64_t)e->cfg.age_start[a + 1] <= i) a++;
            T.age_block[b] = (uint8_t)a;
        }
    }
    e->h_tables.uniform_meta = 1;
    for (uint32_t r = 1; r < e->h_tables.n_rows && e->h_tables.uniform_meta; r++)
        if (e->h_tables.rcount[r] != e->h_tables.rcount[0] ||
            std::memcmp(e->h_tables.meta[r], e->h_tables.meta[0], sizeof(uint32_t) * (size_t)e->h_tables.rcount[0]) != 0)
            e->h_tables.uniform_meta = 0;
    size_t slot = e->stage.size();
    for (size_t k = 0; k < e->stage.size(); k++)
        if (hipEventQuery(e->stage_ev[k]) == hipSuccess) {
            slot = k;
            break;
        }
    if (slot == e->stage.size()) {
        if (e->stage.size() < reina_engine::MAX_STAGES) {
            reina_engine::Stage *st = nullptr;
            hipEvent_t ev;
            HIP_CHECK(hipHostMalloc((void **)&st, sizeof(reina_engine::Stage), hipHostMallocDefault));
            HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e->stage.push_back(st);
            e->stage_ev.push_back(ev);
        } else {
            slot = 0;
            HIP_CHECK(hipEventSynchronize(e->stage_ev[0]));
        }
    }
    std::memcpy(&e->stage[slot]->p, &e->h_params, sizeof(DevParams));
    UploadSegs segs;
    {
        const Tables &T = e->h_tables;
        const uint32_t nr = T.n_rows ? T.n_rows : 1u, nc = T.n_crows ? T.n_crows : 1u;
        const size_t seg[8][2] = {
            {offsetof(Tables, thr), sizeof(T.thr[0]) * nr},
            {offsetof(Tables, meta), sizeof(T.meta[0]) * nr},
            {offsetof(Tables, guide), sizeof(T.guide[0]) * nr},
            {offsetof(Tables, grp), sizeof(T.grp[0]) * nr},
            {offsetof(Tables, rcount), offsetof(Tables, cthr) - offsetof(Tables, rcount)},   // rcount, row_of_age, n_rows, uniform_meta, grouped
            {offsetof(Tables, cthr), sizeof(T.cthr[0]) * nc},
            {offsetof(Tables, cguide), sizeof(T.cguide[0]) * nc},
            {offsetof(Tables, crow_of_age), sizeof(Tables) - offsetof(Tables, crow_of_age)},   // crow_of_age, n_crows, age_shift, age_block
        };
        segs.n = 8;
        for (int q = 0; q < 8; q++) {
            std::memcpy(reinterpret_cast<char *>(&e->stage[slot]->t) + seg[q][0], reinterpret_cast<const char *>(&T) + seg[q][0], seg[q][1]);
            segs.off[q] = (uint32_t)(seg[q][0] / 4);
            segs.words[q] = (uint32_t)(seg[q][1] / 4);
        }
    }
    // The transfer is a KERNEL that reads the pinned host slot directly (zero-copy): an ordinary
    // dispatch in the day stream.  hipMemcpyAsync of the 98 KB table was measured to block the host
    // for 7-8 ms once per run when >1000 dispatches were queued ahead of it (ROCm 7.2).
    void *dsrc = nullptr;
    HIP_CHECK(hipHostGetDevicePointer(&dsrc, e->stage[slot], 0));
    static_assert(sizeof(DevParams) % 4 == 0 && sizeof(Tables) % 4 == 0 && offsetof(reina_engine::Stage, t) % 4 == 0 &&
                  offsetof(Tables, meta) % 4 == 0 && offsetof(Tables, guide) % 4 == 0 && offsetof(Tables, grp) % 4 == 0 && offsetof(Tables, rcount) % 4 == 0 &&
                  offsetof(Tables, cthr) % 4 == 0 && offsetof(Tables, cguide) % 4 == 0 && offsetof(Tables, crow_of_age) % 4 == 0, "word copies");
    const uint32_t *src_w = reinterpret_cast<const uint32_t *>(dsrc);
    hipLaunchKernelGGL(k_upload, dim3(64), dim3(256), 0, s, reinterpret_cast<uint32_t *>(e->d_params), src_w,
                       (uint32_t)(sizeof(DevParams) / 4), reinterpret_cast<uint32_t *>(e->d_tables),
                       src_w + offsetof(reina_engine::Stage, t) / 4, segs);
    HIP_CHECK(hipEventRecord(e->stage_ev[slot], s));
    return REINA_OK;
}

// k_day's workgroups for one engine instance: one 512-agent tile per wave (small populations) up to one
// workgroup of 16 waves per CU, whose waves then walk several tiles; members of a group share the chip.
// Every later kernel takes the resulting wave count as a parameter (the per-wave slices of the lists).
static uint32_t day_blocks_for(uint32_t n_agents, uint32_t K, uint32_t n_cus) {
    const uint32_t tiles = ((n_agents >> 2) + 127u) / 128u;
    uint32_t b = (tiles + DAY_WAVES - 1) / DAY_WAVES;
    if (b < 1) b = 1;
    uint32_t cap = K > 1 ? (n_cus / K > 0 ? n_cus / K : 1u) : n_cus;
    if (cap > REINA_MAX_SCAN_WAVES / DAY_WAVES) cap = REINA_MAX_SCAN_WAVES / DAY_WAVES;
    if (b > cap) b = cap;
    return b;
}

// One day's launches for K engine instances at once (K = 1: a single engine; K > 1: a group of
// identically configured engines, one launch per phase for all of them, member = blockIdx.y).
// `e` is the representative engine: geometry, scenario flags, optional second stream.
// The day comes in the PHASES of include/reina_hip.h (reina_step_phase): between them a sharded population exchanges.
struct DayGeom {   // what the launches of a day's first phases agree on (a function of the engine and the day alone)
    int weekly_own;            // k_open's import roles (k_testing.inc: open_role)
    uint32_t stream_imports;   // workgroups of k_day's launch that place the weekly imports beside the stream
    uint32_t day_blocks;       // streaming workgroups of k_day
    bool xtrace;               // exact attribution on a contact-tracing day: tracing level by level, an exchange behind each
};
static int day_geometry(reina_engine_t *e, uint32_t K, const reina_day_t &dp, DayGeom *g) {
    const uint32_t N = e->cfg.n_agents;
    if (dp.day >= REINA_MAX_DAYS) {
        // claim / mirror keys carry the day in 12 bits (rp_order_key) and claim[] is never cleared: from day
        // 4096 on, stale claims would beat today's
        g_last_error = "day >= REINA_MAX_DAYS (4096): the winner-selection keys carry the day in 12 bits";
        return REINA_E_INVALID;
    }
    if (dp.n_import_batches > REINA_MAX_IMPORT_BATCHES || dp.n_vaccinations > REINA_MAX_VACCINATIONS) {
        g_last_error = "n_import_batches / n_vaccinations out of range";
        return REINA_E_INVALID;
    }
    if (dp.testing_mode != RT_NO_TESTING) e->testing_ever = true;
    uint32_t n_pre = 0, n_post = 0;
    for (uint32_t b = 0; b < dp.n_import_batches; b++)
        (dp.import_batches[b].pre_init ? n_pre : n_post) += dp.import_batches[b].count;
    // weekly imports on a day without intervention imports (which would share their claim keys) have workgroups of their
    // own: one per wave of imports, at most 16 (k_open.inc, imports_any); members of a group take one each (they wait on each
    // other, and a group's launch is not resident as a whole)
    // (> 0: so many workgroups for the weekly imports; <= 0: the opening workgroup places the day's imports, with -weekly_own
    // helpers: open_role, k_testing.inc)
    int weekly_own = n_pre == 0 && n_post > 0;
    if (K == 1) {
        const uint32_t most = n_pre > n_post ? n_pre : n_post;
        int w = (int)((most + 511u) / 512u);
        if (w > (int)e->import_wgs) w = (int)e->import_wgs;
        if (weekly_own) weekly_own = w;
        else if (n_pre > 0 && w > 1) weekly_own = -(w - 1);
    }
    uint32_t day_blocks = day_blocks_for(N, K, e->n_cus);
    // weekly imports on a day without intervention imports are placed in k_day's launch, beside the stream, by workgroups
    // of their own behind the streaming ones (k_open.inc: deferred placement); all of them fit the chip together.
    // (Their records go to the shared candidate overflow list: only when the day's imports fit it with room to spare for real
    // overflow records -- a caller of the C ABI may have sized max_candidates tightly -- else the opening launch places them)
    uint32_t stream_imports = 0;
    if (weekly_own > 0 && dp.day + 1u < REINA_MAX_DAYS && !e->imports_in_open && (uint64_t)n_post * 2u <= e->h_params.cand_ovf_cap) {
        stream_imports = K == 1 ? (uint32_t)weekly_own : 1u;
        const uint32_t cap = K > 1 ? (e->n_cus / K > 0 ? e->n_cus / K : 1u) : e->n_cus;
        if (day_blocks + stream_imports > cap && cap > stream_imports) day_blocks = cap - stream_imports;
        weekly_own = OPEN_WEEKLY_IN_STREAM;
    }
    g->weekly_own = weekly_own;
    g->stream_imports = stream_imports;
    g->day_blocks = day_blocks;
    g->xtrace = e->exact && dp.testing_mode == RT_ALL_WITH_SYMPTOMS_CT;
    return REINA_OK;
}

// REINA_PH_OPEN: the day's opening -- roles (0 opens the day, 1 weekly imports, 2.. the test queue) by arrival ticket
static int launch_day_open(reina_engine_t *e, const MemberRef *refs, uint32_t K, const reina_day_t &dp, uint32_t hist_slot, hipStream_t s) {
    DayGeom geo;
    if (int rc = day_geometry(e, K, dp, &geo)) return rc;
    const uint32_t N = e->cfg.n_agents;
    const int today = profiled_kind(e, dp.day);
    const int weekly_own = geo.weekly_own;
    // (groups: the members share the chip, so each gets proportionally fewer workgroups per phase)
    int tg = grid_for(N / 64 + 1, PRO_THREADS, 64);
    if (K > 1 && tg > (int)(e->n_cus / K)) tg = e->n_cus / K > 0 ? (int)(e->n_cus / K) : 1;
    const bool ct = dp.testing_mode == RT_ALL_WITH_SYMPTOMS_CT;
    const int helpers = weekly_own == OPEN_WEEKLY_IN_STREAM ? 0 : weekly_own > 0 ? weekly_own : -weekly_own;
    const int g0 = 1 + (helpers > 1 ? helpers : 1), g = g0 + tg;
    // roles by arrival ticket unless the launch is resident as a whole -- and, measured, for small populations too
    // (HUS year: k_open 10.4 us a day with tickets, 11.2 by block number; 10^8 agents: 14.6 against 14.1)
    const int open_tickets = (K > 1 || g > (int)e->n_cus || e->open_tickets || N <= 8000000u) ? 1 : 0;
    if (!e->testing_ever) {
        LAUNCH_DAY(e, today, REINA_PK_OPEN, k_open, dim3(g0, K), dim3(PRO_THREADS), 0, s, dp, hist_slot, weekly_own, 0, open_tickets);
    } else if (ct && N <= 8000000u && !geo.xtrace) {
        LAUNCH_DAY(e, today, REINA_PK_OPEN, k_open, dim3(g, K), dim3(PRO_THREADS), 0, s, dp, hist_slot, weekly_own, 3, open_tickets);  // detects + traces, both levels
    } else if (ct) {
        // detects + traces level 0; level 1 by a launch of its own (exact attribution: behind the exchange of the level-0 requests)
        LAUNCH_DAY(e, today, REINA_PK_OPEN, k_open, dim3(g, K), dim3(PRO_THREADS), 0, s, dp, hist_slot, weekly_own, 2, open_tickets);
        if (!geo.xtrace) LAUNCH_DAY(e, today, REINA_PK_TRACE1, k_test_trace1, dim3(grid_for(N / 64 + 1, 256, 256), K), dim3(256), 0, s, dp);
    } else {
        LAUNCH_DAY(e, today, REINA_PK_OPEN, k_open, dim3(g, K), dim3(PRO_THREADS), 0, s, dp, hist_slot, weekly_own, 1, open_tickets);
    }
    HIP_CHECK(hipGetLastError());
    return REINA_OK;
}

// workgroups of a launch that takes in the records of an exchange (k_xtrace, k_feedback)
static int xchg_grid(const reina_engine_t *e) { return grid_for(e->cfg.xchg_cap * (e->cfg.n_shards - 1u) / 4u + 1u, 256, 128); }

// REINA_PH_TRACE (exact attribution, contact-tracing days): the level-0 requests of the other shards, then level 1
static int launch_day_trace(reina_engine_t *e, const MemberRef *refs, uint32_t K, const reina_day_t &dp, hipStream_t s) {
    DayGeom geo;
    if (int rc = day_geometry(e, K, dp, &geo)) return rc;
    if (!geo.xtrace) return REINA_OK;
    const int today = profiled_kind(e, dp.day);
    LAUNCH_DAY(e, today, REINA_PK_XCHG, k_xtrace, dim3(xchg_grid(e), K), dim3(256), 0, s, dp, 0);
    LAUNCH_DAY(e, today, REINA_PK_TRACE1, k_test_trace1, dim3(grid_for(e->cfg.n_agents / 64 + 1, 256, 256), K), dim3(256), 0, s, dp);
    HIP_CHECK(hipGetLastError());
    return REINA_OK;
}

// REINA_PH_MAIN: (the level-1 requests of the other shards,) vaccination, the stream with the contact sampling, and a sharded
// population's event maps for the all-reduce
static int launch_day_main(reina_engine_t *e, const MemberRef *refs, uint32_t K, const reina_day_t &dp, hipStream_t s) {
    DayGeom geo;
    if (int rc = day_geometry(e, K, dp, &geo)) return rc;
    const int today = profiled_kind(e, dp.day);
    if (geo.xtrace) LAUNCH_DAY(e, today, REINA_PK_XCHG, k_xtrace, dim3(xchg_grid(e), K), dim3(256), 0, s, dp, 1);
    const uint32_t day_blocks = geo.day_blocks, stream_imports = geo.stream_imports;
    uint32_t lds_rows = K > 1 ? e->group_lds_rows : e->h_tables.n_rows;   // (a member stages min(its own rows, lds_rows))
    if (lds_rows > REINA_LDS_ROWS) lds_rows = REINA_LDS_ROWS;
    if (e->lds_rows_cap && lds_rows > e->lds_rows_cap) lds_rows = e->lds_rows_cap;
    uint32_t lds_crows = K > 1 ? e->group_lds_crows : e->h_tables.n_crows;
    if (lds_crows > REINA_LDS_CROWS) lds_crows = REINA_LDS_CROWS;
    if (e->lds_rows_cap && lds_crows > e->lds_rows_cap) lds_crows = e->lds_rows_cap;
    // a vaccination programme: its pass over the agents comes after the test queue and before the stream
    // (HealthcareSystem.iterate, main.pyx:514-558)
    if (dp.n_vaccinations) {
        // a day on which some programme's number exceeds a step of 16 x 1024 agents: a chain of workgroups, one step each, with room
        // for a third more agents than the largest number (k_open.inc: pro_vaccinate_chain; the programmes one after the other, the
        // launch agreeing on each one's end); otherwise one workgroup.  (the numbers as the kernel clips them: a programme whose window
        // holds fewer agents never needs the chain.)  Members of a group chain too while the whole launch stays resident.
        uint32_t vg = 1, most = 0, need[REINA_MAX_VACCINATIONS], sum_need = 0;
        bool disjoint = dp.n_vaccinations > 1;
        for (uint32_t k = 0; k < dp.n_vaccinations; k++) {
            const reina_vaccination_t &vk = dp.vaccinations[k];
            const uint32_t window = vk.idx_end > vk.idx_start ? vk.idx_end - vk.idx_start : 0u;
            const uint32_t nrk = vk.nr < window ? vk.nr : window;
            if (nrk > most) most = nrk;
            need[k] = (nrk + nrk / 3u) / (VACC_CHUNKS * PRO_THREADS) + 1u;
            sum_need += need[k];
            for (uint32_t j = 0; j < k; j++)
                if (vk.idx_start < dp.vaccinations[j].idx_end && dp.vaccinations[j].idx_start < vk.idx_end) disjoint = false;
        }
        VaccGeom geo;
        std::memset(&geo, 0, sizeof(geo));
        // (the published words live in the head of buffers.scan_lists: 16 programmes x 512 + 16 arrival words, 64 bits each)
        if (most > VACC_CHUNKS * PRO_THREADS && e->cfg.max_work_items >= 8192 + 16 && !e->vacc_one_wg) {
            uint32_t cap = K > 1 ? e->n_cus / K : e->n_cus;
            if (cap > 256u) cap = 256u;
            if (disjoint && sum_need <= cap) {
                // age tiers side by side (windows pairwise disjoint: the order of the programmes does not matter): every
                // programme on workgroups of its own, all at once
                geo.parallel = 1u;
                for (uint32_t k = 0; k < dp.n_vaccinations; k++) geo.g[k] = (uint16_t)need[k];
                vg = sum_need;
            } else {
                vg = (most + most / 3u) / (VACC_CHUNKS * PRO_THREADS) + 1u;
                if (vg > cap) vg = cap;
            }
            if (vg < 1u) vg = 1u;
        }
        // The tag is drawn from ONE process-wide counter (round-5 advisor): the tagged words live in each member's own buffers and
        // are never cleared, so a per-engine counter could come back to a value those buffers already hold -- a member stepped in
        // a group (tag = the representative's count) and later alone or under another representative.  A process-wide value is
        // never reused on any buffers (2^32 - 1 vaccination launches per process before it wraps).
        static std::atomic<uint32_t> vacc_seq_next{0};
        do e->vacc_seq = ++vacc_seq_next; while (e->vacc_seq == 0u);
        LAUNCH_DAY(e, today, REINA_PK_VACCINATE, k_vaccinate, dim3(vg, K), dim3(PRO_THREADS), 0, s, dp, e->vacc_seq, geo);
    }
    {
        // the form of the stream (k_contacts.inc): sparse from DAY_SPARSE_MIN_TILES tiles per wave on -- every day of such a population:
        // a threshold on yesterday's active agents (rounds 4-5, decided in the kernel) never chose the dense form where the sparse
        // one was possible
        const uint32_t tiles = ((e->cfg.n_agents >> 2) + 127u) / 128u;
        bool sparse = e->day_mode == 2 || (e->day_mode == 0 && tiles >= DAY_SPARSE_MIN_TILES * day_blocks * DAY_WAVES);
        if (e->day_mode == 3) sparse = (dp.day & 1u) == 0u;
        const dim3 grid(day_blocks + stream_imports, K), block(DAY_THREADS);
        const size_t lds = day_shared_bytes(lds_rows, lds_crows, e->cfg.n_shards);
#define K_DAY(G, X, SP) LAUNCH_TIMED(e, today, REINA_PK_DAY, (k_day<G, X, SP>), grid, block, lds, s, refs, (e)->h_ref, dp, lds_rows, lds_crows, e->day_flags, stream_imports)
        if (K > 1) { if (sparse) K_DAY(true, false, true); else K_DAY(true, false, false); }
        else if (e->exact) { if (sparse) K_DAY(false, true, true); else K_DAY(false, true, false); }
        else { if (sparse) K_DAY(false, false, true); else K_DAY(false, false, false); }
#undef K_DAY
    }
    e->cur_scan_waves = day_blocks * DAY_WAVES;   // (the day's later launches walk the per-wave slices)
    if (e->cfg.n_shards > 1) {
        // a sharded population: its event buckets sorted and their maps written to the exchange block BEFORE the all-reduce
        const uint32_t n_walk = (e->h_params.hosp_ranges + 15u) / 16u;
        LAUNCH_DAY(e, today, REINA_PK_HOSP_SORT, k_hosp_presort, dim3(n_walk, K), dim3(HOSP_THREADS), (size_t)HOSP_P_THREADS * HOSP_P_E * 8, s, dp);
    }
    HIP_CHECK(hipGetLastError());
    return REINA_OK;
}

static int launch_day_begin(reina_engine_t *e, const MemberRef *refs, uint32_t K, const reina_day_t &dp,
                            uint32_t hist_slot, hipStream_t s) {
    if (int rc = launch_day_open(e, refs, K, dp, hist_slot, s)) return rc;
    if (int rc = launch_day_trace(e, refs, K, dp, s)) return rc;
    return launch_day_main(e, refs, K, dp, s);
}

// REINA_PH_FEEDBACK (exact attribution): the sources of the day's cross-shard infections take their infectees
static int launch_day_feedback(reina_engine_t *e, const MemberRef *refs, uint32_t K, const reina_day_t &dp, hipStream_t s) {
    if (!e->exact) return REINA_OK;
    const int today = profiled_kind(e, dp.day);
    LAUNCH_DAY(e, today, REINA_PK_XCHG, k_feedback, dim3(xchg_grid(e), K), dim3(256), 0, s, dp);
    HIP_CHECK(hipGetLastError());
    return REINA_OK;
}

static int launch_day_end(reina_engine_t *e, const MemberRef *refs, uint32_t K, const reina_day_t &dp, hipStream_t s) {
    const uint32_t N = e->cfg.n_agents;
    const int today = profiled_kind(e, dp.day);
    const bool sharded = e->cfg.n_shards > 1;
    if (sharded)   // (also takes this shard's share of the pooled free beds / ICU units: the walk below starts from it)
        LAUNCH_DAY(e, today, REINA_PK_REMOTE, k_remote, dim3(grid_for(N / 256 + 1, 256, 256), K), dim3(256), 0, s, dp);
    {
        const uint32_t scan_tiles = ((N >> 2) + 127u) / 128u;
        const uint32_t scan_waves = e->cur_scan_waves;
        // installing workgroups: one wave per unit of a quiet day (3 lists of scan_waves / 8 units + the event buckets), so
        // that no wave walks two dependent chains one after the other; at most one workgroup per CU
        int ig = grid_for(N / 64 + 1, HOSP_THREADS, 128) * 2;
        {
            const uint32_t units = 3u * ((scan_waves + 7u) / 8u) + (e->h_params.hosp_ranges + 7u) / 8u;
            const int want = (int)((units + HOSP_THREADS / 64 - 1) / (HOSP_THREADS / 64));
            if (want > ig) ig = want;
            if (ig > (int)e->n_cus) ig = (int)e->n_cus;
        }
        // (groups: 128 workgroups for all members together -- every workgroup pays its prologue and its histogram flush)
        // (a group of 128 HUS members, measured in round 4: ONE installing workgroup per member beside the one for the events --
        // 256 workgroups, one per CU, all resident together -- 67 us a launch against 72.5 with two, 70 with three)
        if (K > 1 && ig > (int)(128 / K)) ig = (int)(128 / K) >= 2 ? ((int)(128 / K) & ~1) : 1;
        const bool par = e->h_params.hosp_parallel != 0;
        if (par) {
            // a large population: the launch's first workgroups are the walkers of a day on which the events' order matters
            // (one wave per priority bucket: R / 16 workgroups; on any other day they install too), the others install.
            // One workgroup of 1024 threads per CU: all of them resident together, so the walk runs beside the installs.
            const uint32_t n_walk = (e->h_params.hosp_ranges + e->walk_div - 1u) / e->walk_div;
            int rest = (int)e->n_cus - (int)n_walk;
            if (K > 1) rest = (int)(e->n_cus / K) - (int)n_walk;
            if (rest > ig) rest = ig;
            if (rest < 2) rest = 2;
            LAUNCH_DAY(e, today, REINA_PK_INSTALL, k_hosp_install, dim3(n_walk + rest, K), dim3(HOSP_THREADS),
                         (size_t)HOSP_P_THREADS * HOSP_P_E * 8, s, dp, scan_waves, scan_tiles, HI_INSTALL | HI_EVENTS, n_walk);
        } else {
            // workgroup 0 walks the bed / ICU events of a day on which order matters, beside the installs
            if (ig < 2 && K == 1) ig = 2;
            LAUNCH_DAY(e, today, REINA_PK_INSTALL, k_hosp_install, dim3(ig + 1, K), dim3(HOSP_THREADS), (size_t)REINA_MAX_HOSP_EVENTS * 8, s, dp, scan_waves, scan_tiles, HI_HOSP_WG | HI_INSTALL | HI_EVENTS, 0u);
        }
    }
    HIP_CHECK(hipGetLastError());
    return REINA_OK;
}

// A stretch of days of a small unsharded population as ONE launch (k_small.inc).  Every workgroup of it must be resident at once
// (they meet at three barriers a day): at most one per compute unit by their LDS, `small_wgs` of them, and only while the process
// holds so few engines that all of theirs fit the chip together (other engines' launches may run beside this one on streams of
// their own).  A day qualifies when nothing in it needs another launch shape: no vaccination programme, at most one workgroup's
// worth of weekly imports, import helpers and one test-queue role within the launch's workgroups.
static bool small_days_engine_ok(const reina_engine_t *e) {
    if (!e->fused_day || e->cfg.n_shards != 1 || e->exact || e->coll_fn || e->h_params.hosp_parallel != 0u) return false;
    if (e->cfg.n_agents > REINA_HOSP_SMALL_AGENTS) return false;
    if (e->day_mode != 0 || e->open_tickets || e->imports_in_open) return false;      // (the tests' handles on the three-launch forms)
    if ((uint32_t)g_live_engines.load() * e->small_wgs > e->n_cus) return false;
    return SMALL_TAIL_BYTES + e->small_static_lds <= 160u * 1024u;
}
// the record of one day, or false: the day takes the three launches
static bool small_day_record(reina_engine_t *e, const reina_day_t &dp, SmallDayRec *rec) {
    if (dp.n_vaccinations != 0u) return false;
    DayGeom geo;
    if (day_geometry(e, 1, dp, &geo) != REINA_OK) return false;
    const uint32_t W = e->small_wgs;
    const uint32_t helpers = geo.weekly_own == OPEN_WEEKLY_IN_STREAM ? 0u : geo.weekly_own > 0 ? (uint32_t)geo.weekly_own : (uint32_t)-geo.weekly_own;
    if (W < 1u + (helpers > 1u ? helpers : 1u) + 1u || geo.stream_imports > 1u) return false;
    std::memset(rec, 0, sizeof(*rec));
    rec->dp = dp;
    rec->weekly_own = geo.weekly_own;
    const bool ct = dp.testing_mode == RT_ALL_WITH_SYMPTOMS_CT;
    rec->mode = !e->testing_ever ? 0 : ct ? 3 : 1;   // (launch_day_open: a population this small walks level 1 in the opening)
    rec->stream_imports = geo.stream_imports;
    return true;
}
static int launch_small_days(reina_engine_t *e, const SmallDayRec *recs, uint32_t n, hipStream_t s) {
    // the records travel like the tables: a pinned slot the host fills, a copy kernel in the day stream (reina_upload_contact_tables)
    size_t slot = e->days_stage.size();
    for (size_t k = 0; k < e->days_stage.size(); k++)
        if (hipEventQuery(e->days_stage_ev[k]) == hipSuccess) {
            slot = k;
            break;
        }
    if (slot == e->days_stage.size()) {
        if (e->days_stage.size() < reina_engine::MAX_STAGES) {
            SmallDayRec *st = nullptr;
            hipEvent_t ev;
            HIP_CHECK(hipHostMalloc((void **)&st, sizeof(SmallDayRec) * SMALL_MAX_DAYS, hipHostMallocDefault));
            HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e->days_stage.push_back(st);
            e->days_stage_ev.push_back(ev);
        } else {
            slot = 0;
            HIP_CHECK(hipEventSynchronize(e->days_stage_ev[0]));
        }
    }
    std::memcpy(e->days_stage[slot], recs, sizeof(SmallDayRec) * n);
    void *dsrc = nullptr;
    HIP_CHECK(hipHostGetDevicePointer(&dsrc, e->days_stage[slot], 0));
    UploadSegs none;
    std::memset(&none, 0, sizeof(none));
    const uint32_t words = (uint32_t)(sizeof(SmallDayRec) * n / 4);
    hipLaunchKernelGGL(k_upload, dim3(words / 256u + 1u < 64u ? words / 256u + 1u : 64u), dim3(256), 0, s, reinterpret_cast<uint32_t *>(e->d_days),
                       reinterpret_cast<const uint32_t *>(dsrc), words, (uint32_t *)nullptr, (const uint32_t *)nullptr, none);
    HIP_CHECK(hipEventRecord(e->days_stage_ev[slot], s));
    const uint32_t N = e->cfg.n_agents, W = e->small_wgs;
    SmallGeom g;
    std::memset(&g, 0, sizeof(g));
    g.n_days = n;
    g.lds_rows = e->h_tables.n_rows > REINA_LDS_ROWS ? REINA_LDS_ROWS : e->h_tables.n_rows;
    g.lds_crows = e->h_tables.n_crows > REINA_LDS_CROWS ? REINA_LDS_CROWS : e->h_tables.n_crows;
    if (e->lds_rows_cap && g.lds_rows > e->lds_rows_cap) g.lds_rows = e->lds_rows_cap;
    if (e->lds_rows_cap && g.lds_crows > e->lds_rows_cap) g.lds_crows = e->lds_rows_cap;
    g.day_flags = e->day_flags;
    g.scan_tiles = ((N >> 2) + 127u) / 128u;
    g.bar_base = e->bar_epoch;
    g.bar = e->d_bar;
    g.days = e->d_days;
    e->bar_epoch += SMALL_BARRIERS_PER_DAY * n;
    e->cur_scan_waves = (W - 1u) * DAY_WAVES;
    // (timed as a whole on the stretches that begin on one of the two profiled phases; the kind's "launches" count DAYS: its mean is per day)
    const int today = profiled_kind(e, recs[0].dp.day);
    size_t ev_a, ev_b;
    if (kind_timed(today, REINA_PK_SMALL_DAY) && take_event_pair(e, &ev_a, &ev_b)) {
        hipExtLaunchKernelGGL(k_small_days, dim3(W), dim3(HOSP_THREADS), SMALL_TAIL_BYTES, s, e->ev_pool[ev_a], e->ev_pool[ev_b], 0, e->h_ref, g);
        e->kpairs[REINA_PK_SMALL_DAY].emplace_back(ev_a, ev_b);
        e->small_pair_days.push_back(n);
    } else {
        hipLaunchKernelGGL(k_small_days, dim3(W), dim3(HOSP_THREADS), SMALL_TAIL_BYTES, s, e->h_ref, g);
    }
    HIP_CHECK(hipGetLastError());
    return REINA_OK;
}
// days[0 .. n): stretches of qualifying days (at least two: a single day is faster as three launches) as one launch each, the rest day by day
static int run_days_small(reina_engine_t *e, const reina_day_t *days, uint32_t n_days, int32_t *history_base, hipStream_t s) {
    std::vector<SmallDayRec> recs;
    recs.reserve(SMALL_MAX_DAYS);
    uint32_t k = 0;
    while (k < n_days) {
        recs.clear();
        uint32_t m = 0;
        while (k + m < n_days && m < SMALL_MAX_DAYS) {
            reina_day_t d = days[k + m];
            if (history_base) d.history_row = history_base + (size_t)(k + m) * REINA_COUNTER_WORDS;
            SmallDayRec r;
            if (!small_day_record(e, d, &r)) break;
            recs.push_back(r);
            m++;
        }
        if (m >= 2u) {
            if (int rc = launch_small_days(e, recs.data(), m, s)) return rc;
            k += m;
            continue;
        }
        // (one qualifying day between two that do not, or none: the launches)
        const uint32_t take = m ? m : 1u;
        for (uint32_t q = 0; q < take; q++) {
            reina_day_t d = days[k + q];
            if (history_base) d.history_row = history_base + (size_t)(k + q) * REINA_COUNTER_WORDS;
            for (int ph = 0; ph < REINA_PH_NR; ph++) {
                const int need = reina_step_phase(e, &d, ph, s);
                if (need < 0) return need;
            }
        }
        k += take;
    }
    return REINA_OK;
}

int reina_step_phase(reina_engine_t *e, const reina_day_t *day, int phase, void *stream) {
    if (!e || !day) return REINA_E_INVALID;
    if (!e->bound) return REINA_E_NOT_BOUND;
    hipStream_t s = (hipStream_t)stream;
    const bool xtrace = e->exact && day->testing_mode == RT_ALL_WITH_SYMPTOMS_CT;
    int rc;
    switch (phase) {
    case REINA_PH_OPEN:
        rc = launch_day_open(e, e->d_ref, 1, *day, 0, s);
        return rc ? rc : (xtrace ? REINA_X_ALLTOALL : 0);
    case REINA_PH_TRACE:
        rc = launch_day_trace(e, e->d_ref, 1, *day, s);
        return rc ? rc : (xtrace ? REINA_X_ALLTOALL : 0);
    case REINA_PH_MAIN:
        rc = launch_day_main(e, e->d_ref, 1, *day, s);
        // (a single shard with a collective set exchanges too: the one-GPU box exercises the call)
        // exact attribution (round 6, ABI 7): what the all-reduce carries -- the shards' capacity words and event maps -- rides in the
        // trailers of the contact records' segments: ONE collective here, not two
        if (rc) return rc;
        if (e->exact) return REINA_X_ALLTOALL;
        return (e->cfg.n_shards > 1 || e->coll_fn) ? REINA_X_ALLREDUCE : 0;
    case REINA_PH_END:
        rc = launch_day_end(e, e->d_ref, 1, *day, s);
        return rc ? rc : (e->exact ? REINA_X_ALLTOALL : 0);
    case REINA_PH_FEEDBACK:
        return launch_day_feedback(e, e->d_ref, 1, *day, s);
    }
    g_last_error = "phase out of range";
    return REINA_E_INVALID;
}

// the two halves of a day around its one all-reduce (a population without exact attribution)
int reina_step_day_begin(reina_engine_t *e, const reina_day_t *day, void *stream) {
    if (!e || !day) return REINA_E_INVALID;
    if (!e->bound) return REINA_E_NOT_BOUND;
    if (e->exact) {
        g_last_error = "exact attribution: a day has more than one exchange -- step it by reina_step_phase (or reina_step_day with reina_set_alltoall)";
        return REINA_E_INVALID;
    }
    return launch_day_begin(e, e->d_ref, 1, *day, 0, (hipStream_t)stream);
}

int reina_step_day_end(reina_engine_t *e, const reina_day_t *day, void *stream) {
    if (!e || !day) return REINA_E_INVALID;
    if (!e->bound) return REINA_E_NOT_BOUND;
    if (e->exact) {
        g_last_error = "exact attribution: a day has more than one exchange -- step it by reina_step_phase (or reina_step_day with reina_set_alltoall)";
        return REINA_E_INVALID;
    }
    return launch_day_end(e, e->d_ref, 1, *day, (hipStream_t)stream);
}

int reina_set_collective(reina_engine_t *e, reina_allreduce_fn allreduce, void *comm) {
    if (!e) return REINA_E_INVALID;
    e->coll_fn = allreduce;
    e->coll_comm = comm;
    return REINA_OK;
}

int reina_set_alltoall(reina_engine_t *e, reina_alltoall_fn alltoall, void *comm) {
    if (!e) return REINA_E_INVALID;
    e->a2a_fn = alltoall;
    e->a2a_comm = comm;
    return REINA_OK;
}

int reina_step_day(reina_engine_t *e, const reina_day_t *day, void *stream) {
    for (int ph = 0; ph < REINA_PH_NR; ph++) {
        const int need = reina_step_phase(e, day, ph, stream);
        if (need < 0) return need;
        // the exchanges of a sharded population, queued on the day stream itself (on the profiled days of the event-walk kind an
        // event pair is recorded around them: `collective` of reina_profile_read_kernels -- what RCCL costs a day, apart from the kernels)
        size_t ev_a = 0, ev_b = 0;
        const bool timed = need > 0 && (e->coll_fn || e->a2a_fn) && kind_timed(profiled_kind(e, day->day), REINA_PK_COLLECTIVE) &&
                           take_event_pair(e, &ev_a, &ev_b) && hipEventRecord(e->ev_pool[ev_a], (hipStream_t)stream) == hipSuccess;
        if ((need & REINA_X_ALLREDUCE) && e->coll_fn) {
            const int r = e->coll_fn(e->buf.pressure, e->buf.pressure, e->exchange_words, 2 /* ncclInt32 */, 0 /* ncclSum */, e->coll_comm, stream);
            if (r != 0) {
                g_last_error = "collective failed with code " + std::to_string(r);
                return REINA_E_HIP;
            }
        }
        if (need & REINA_X_ALLTOALL) {
            if (!e->a2a_fn) {
                g_last_error = "exact attribution: no all-to-all set (reina_set_alltoall) -- step the day by reina_step_phase and exchange buffers.xsend / xrecv yourself";
                return REINA_E_NOT_BOUND;
            }
            const int r = e->a2a_fn(e->buf.xsend, e->buf.xrecv, (size_t)e->h_params.xchg_stride, 4 /* ncclInt64 */, e->a2a_comm, stream);
            if (r != 0) {
                g_last_error = "all-to-all failed with code " + std::to_string(r);
                return REINA_E_HIP;
            }
        }
        if (timed && hipEventRecord(e->ev_pool[ev_b], (hipStream_t)stream) == hipSuccess) e->kpairs[REINA_PK_COLLECTIVE].emplace_back(ev_a, ev_b);
    }
    return REINA_OK;
}

int reina_run_days(reina_engine_t *e, const reina_day_t *days, uint32_t n_days, void *stream) {
    if (e && days && e->bound && n_days >= 2u && small_days_engine_ok(e)) return run_days_small(e, days, n_days, nullptr, (hipStream_t)stream);
    for (uint32_t k = 0; k < n_days; k++) {
        int rc = reina_step_day(e, &days[k], stream);
        if (rc) return rc;
    }
    return REINA_OK;
}

int reina_run_days_hist(reina_engine_t *e, const reina_day_t *days, uint32_t n_days, int32_t *history_base, void *stream) {
    if (e && days && e->bound && n_days >= 2u && small_days_engine_ok(e)) return run_days_small(e, days, n_days, history_base, (hipStream_t)stream);
    for (uint32_t k = 0; k < n_days; k++) {
        reina_day_t d = days[k];
        d.history_row = history_base ? history_base + (size_t)k * REINA_COUNTER_WORDS : nullptr;
        int rc = reina_step_day(e, &d, stream);
        if (rc) return rc;
    }
    return REINA_OK;
}

struct reina_group {
    std::vector<reina_engine_t *> members;
    MemberRef *d_refs = nullptr;
    std::vector<MemberRef> h_refs;
};

int reina_group_create(reina_engine_t **engines, uint32_t n, reina_group_t **out) {
    if (!engines || !out || n == 0 || n > 65535) return REINA_E_INVALID;
    for (uint32_t k = 0; k < n; k++) {
        reina_engine_t *m = engines[k];
        if (!m || !m->bound) return REINA_E_NOT_BOUND;
        if (m->cfg.n_agents != engines[0]->cfg.n_agents || m->cfg.nr_ages != engines[0]->cfg.nr_ages ||
            m->cfg.nr_variants != engines[0]->cfg.nr_variants || m->cfg.n_shards != 1 ||
            std::memcmp(m->cfg.age_start, engines[0]->cfg.age_start, sizeof(m->cfg.age_start)) != 0) {
            g_last_error = "group members must be unsharded engines of the same population";
            return REINA_E_INVALID;
        }
    }
    reina_group *g = new reina_group();
    g->members.assign(engines, engines + n);
    g->h_refs.resize(n);
    for (uint32_t k = 0; k < n; k++) {
        g->h_refs[k].P = engines[k]->d_params;
        g->h_refs[k].T = engines[k]->d_tables;
        g->h_refs[k].B = engines[k]->buf;
        g->h_refs[k].history_base = nullptr;

    }
    HIP_CHECK_OR(hipMalloc(&g->d_refs, sizeof(MemberRef) * n), delete g);
    HIP_CHECK_OR(hipMemcpy(g->d_refs, g->h_refs.data(), sizeof(MemberRef) * n, hipMemcpyHostToDevice),   // (table broadcasts may precede the first run)
                 { (void)hipFree(g->d_refs); delete g; });
    *out = g;
    return REINA_OK;
}

int reina_group_destroy(reina_group_t *g) {
    if (!g) return REINA_E_INVALID;
    (void)hipFree(g->d_refs);
    delete g;
    return REINA_OK;
}

// the table-dependent parts of DevParams: [nrc, iot_mask) and [n_ranges, end)
#define DP_TAB0_BEGIN offsetof(DevParams, nrc)
#define DP_TAB0_END offsetof(DevParams, iot_mask)
#define DP_TAB1_BEGIN offsetof(DevParams, n_ranges)
static_assert(DP_TAB0_BEGIN % 4 == 0 && DP_TAB0_END % 4 == 0 && DP_TAB1_BEGIN % 4 == 0, "word copies");

// member 0's freshly uploaded tables, device to device, into every other member of the group
__global__ __launch_bounds__(256) void k_group_tables(const MemberRef *refs) {
    const MemberRef &src = refs[0], &dst = refs[blockIdx.y + 1];
    const uint32_t stride = gridDim.x * blockDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
    {
        const uint32_t *s_ = reinterpret_cast<const uint32_t *>(src.T);
        uint32_t *d_ = reinterpret_cast<uint32_t *>(const_cast<Tables *>(dst.T));
        for (uint32_t k = first; k < sizeof(Tables) / 4; k += stride) d_[k] = s_[k];
    }
    const uint32_t *sp = reinterpret_cast<const uint32_t *>(src.P);
    uint32_t *dp_ = reinterpret_cast<uint32_t *>(const_cast<DevParams *>(dst.P));
    for (uint32_t k = DP_TAB0_BEGIN / 4 + first; k < DP_TAB0_END / 4; k += stride) dp_[k] = sp[k];
    for (uint32_t k = DP_TAB1_BEGIN / 4 + first; k < sizeof(DevParams) / 4; k += stride) dp_[k] = sp[k];
}

int reina_group_upload_contact_tables(reina_group_t *g, const reina_contact_tables_t *t, void *stream) {
    if (!g) return REINA_E_INVALID;
    // the members sample from identical tables: one transfer from the host (member 0), one device-to-device
    // broadcast to the others -- two launches whatever the size of the group
    reina_engine_t *e0 = g->members[0];
    int rc = reina_upload_contact_tables(e0, t, stream);
    if (rc) return rc;
    const uint32_t K = (uint32_t)g->members.size();
    if (K == 1) return REINA_OK;
    for (uint32_t k = 1; k < K; k++) {   // host mirrors follow
        reina_engine_t *m = g->members[k];
        std::memcpy(reinterpret_cast<char *>(&m->h_params) + DP_TAB0_BEGIN, reinterpret_cast<const char *>(&e0->h_params) + DP_TAB0_BEGIN,
                    DP_TAB0_END - DP_TAB0_BEGIN);
        std::memcpy(reinterpret_cast<char *>(&m->h_params) + DP_TAB1_BEGIN, reinterpret_cast<const char *>(&e0->h_params) + DP_TAB1_BEGIN,
                    sizeof(DevParams) - DP_TAB1_BEGIN);
        std::memcpy(&m->h_tables, &e0->h_tables, sizeof(Tables));
    }
    hipLaunchKernelGGL(k_group_tables, dim3(32, K - 1), dim3(256), 0, (hipStream_t)stream, g->d_refs);
    HIP_CHECK(hipGetLastError());
    return REINA_OK;
}

int reina_group_run_days(reina_group_t *g, const reina_day_t *days, uint32_t n_days, int32_t *const *history_bases,
                         void *stream) {
    if (!g || !days) return REINA_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t K = (uint32_t)g->members.size();
    for (uint32_t k = 0; k < K; k++) g->h_refs[k].history_base = history_bases ? history_bases[k] : nullptr;
    HIP_CHECK(hipMemcpyAsync(g->d_refs, g->h_refs.data(), sizeof(MemberRef) * K, hipMemcpyHostToDevice, s));
    reina_engine_t *e0 = g->members[0];
    bool tested = false;  // the test-queue kernels run for all members once any member ever tested
    e0->group_lds_rows = e0->group_lds_crows = 0;
    for (auto m : g->members) {
        if (m->h_tables.n_rows > e0->group_lds_rows) e0->group_lds_rows = m->h_tables.n_rows;
        if (m->h_tables.n_crows > e0->group_lds_crows) e0->group_lds_crows = m->h_tables.n_crows;
    }
    for (auto m : g->members) tested = tested || m->testing_ever;
    for (uint32_t d = 0; d < n_days; d++) tested = tested || days[d].testing_mode != RT_NO_TESTING;
    e0->testing_ever = e0->testing_ever || (tested && n_days == 0);
    // (a group of ONE member launches the single-engine kernels, which take the member by value: its history base for this run)
    const MemberRef own_ref = e0->h_ref;
    if (K == 1) e0->h_ref.history_base = g->h_refs[0].history_base;
    for (uint32_t d = 0; d < n_days; d++) {
        reina_day_t dp = days[d];
        dp.history_row = nullptr;
        int rc = launch_day_begin(e0, g->d_refs, K, dp, d, s);
        if (rc == REINA_OK) rc = launch_day_end(e0, g->d_refs, K, dp, s);
        if (rc) {
            e0->h_ref = own_ref;
            return rc;
        }
    }
    e0->h_ref = own_ref;
    for (auto m : g->members) {
        m->testing_ever = m->testing_ever || e0->testing_ever;
        m->cur_scan_waves = e0->cur_scan_waves;
    }
    return REINA_OK;
}

int reina_read_counters(reina_engine_t *e, int32_t *out_host, void *stream) {
    if (!e || !out_host) return REINA_E_INVALID;
    if (!e->bound) return REINA_E_NOT_BOUND;
    hipStream_t s = (hipStream_t)stream;
    HIP_CHECK(hipMemcpyAsync(out_host, e->buf.counters, sizeof(int32_t) * REINA_COUNTER_WORDS, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return REINA_OK;
}

int reina_read_history(reina_engine_t *e, const int32_t *history_dev, uint32_t n_rows, int32_t *out_host, void *stream) {
    if (!e || !out_host || (n_rows && !history_dev)) return REINA_E_INVALID;
    if (!e->bound) return REINA_E_NOT_BOUND;
    hipStream_t s = (hipStream_t)stream;
    const size_t row = sizeof(int32_t) * REINA_COUNTER_WORDS;
    static_assert(sizeof(int32_t) * REINA_COUNTER_WORDS % 16 == 0, "rows are copied 16 bytes at a time");
    // a page-locked destination (what engine.py passes) is written by a kernel, over the link: one launch instead of the copy
    // engine's two set-ups (REINA_EXPORT=memcpy: the copies, for comparison)
    void *export_dev = nullptr;
    {
        hipPointerAttribute_t at;
        std::memset(&at, 0, sizeof(at));
        if (!e->export_by_copies && hipPointerGetAttributes(&at, out_host) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer &&
            (reinterpret_cast<uintptr_t>(at.devicePointer) & 15u) == 0u)
            export_dev = at.devicePointer;
        else
            (void)hipGetLastError();   // (pageable memory is not an error here: it takes the copies below)
    }
    // (k_export reads 16 bytes a lane: the history rows AND the counter block -- a C-ABI caller may have bound either at any
    // 4-byte offset into a block of its own; anything not 16-byte aligned takes the copies below)
    if (export_dev && (n_rows == 0 || (reinterpret_cast<uintptr_t>(history_dev) & 15u) == 0u) &&
        (reinterpret_cast<uintptr_t>(e->buf.counters) & 15u) == 0u) {
        const uint32_t n_hist4 = (uint32_t)(row * n_rows / 16), n_cnt4 = (uint32_t)(row / 16);
        const uint32_t blocks = (n_hist4 + n_cnt4 + 255u) / 256u;
        hipLaunchKernelGGL(k_export, dim3(blocks < 64u ? blocks : 64u), dim3(256), 0, s, reinterpret_cast<const uint4 *>(history_dev), n_hist4,
                           reinterpret_cast<const uint4 *>(e->buf.counters), n_cnt4, reinterpret_cast<uint4 *>(export_dev));
        HIP_CHECK(hipGetLastError());
    } else {
        if (n_rows) HIP_CHECK(hipMemcpyAsync(out_host, history_dev, row * n_rows, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(out_host + (size_t)n_rows * REINA_COUNTER_WORDS, e->buf.counters, row, hipMemcpyDeviceToHost, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));
    return REINA_OK;
}

int reina_profile_enable(reina_engine_t *e, int enable) {
    if (!e) return REINA_E_INVALID;
    e->profile = enable != 0;
    e->profile_day_only = enable < 0;      // -k: stride k, and only k_day (the stream + contact sampling) is timed
    if (enable < 0) enable = -enable;
    e->profile_stride = enable > 1 ? (uint32_t)(enable < 4 ? 4 : enable) : 1u;   // (four kinds take turns: stride >= 4)
    // create timing events up front: hipEventCreate inside a timed region costs microseconds each
    while (e->profile && e->ev_pool.size() < 1024) {
        hipEvent_t ev;
        HIP_CHECK(hipEventCreateWithFlags(&ev, REINA_TIMING_EVENT_FLAGS));
        e->ev_pool.push_back(ev);
    }
    return REINA_OK;
}

int reina_profile_read_kernels(reina_engine_t *e, double *ms_total, uint64_t *launches) {
    if (!e) return REINA_E_INVALID;
    HIP_CHECK(hipDeviceSynchronize());
    resolve_profile(e);
    for (int k = 0; k < REINA_PK_NR; k++) {
        if (ms_total) ms_total[k] = e->k_ms[k];
        if (launches) launches[k] = e->k_launches[k];
        e->k_ms[k] = 0;
        e->k_launches[k] = 0;
    }
    return REINA_OK;
}

int reina_profile_read(reina_engine_t *e, double *scan_ms_total, uint64_t *scan_launches, double *all_ms_total) {
    double ms[REINA_PK_NR];
    uint64_t n[REINA_PK_NR];
    const int rc = reina_profile_read_kernels(e, ms, n);
    if (rc) return rc;
    if (scan_ms_total) *scan_ms_total = ms[REINA_PK_DAY];
    if (scan_launches) *scan_launches = n[REINA_PK_DAY];
    if (all_ms_total) {   // every timed launch of every kind
        *all_ms_total = 0;
        for (int k = 0; k < REINA_PK_NR; k++) *all_ms_total += ms[k];
    }
    return REINA_OK;
}

}  // extern "C"
