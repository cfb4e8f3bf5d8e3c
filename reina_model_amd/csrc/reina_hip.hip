// reina_hip.hip -- MI355X (gfx950 / CDNA4) agent engine behind the C ABI of include/reina_hip.h.
//
// One simulated day (the reference's Context.iterate, cythonsim/main.pyx:2011-2018) is a short
// sequence of kernels over SoA agent state resident in HBM:
//
//   k_prologue  1 workgroup   history snapshot, new beds, imports (claim rounds), daily zeroing,
//                             vaccination cursor scan                      (main.pyx:1652-1699,560-583)
//   k_test_*    grid          test queue -> detect, contact tracing level 0 / level 1 (:495-558)
//   k_scan      grid, stream  every agent's 4-byte hot word: R bookkeeping, state machine,
//                             contact COUNT draw, hospital events, work items  (:1968-1992,395-438)
//   k_hospital  1 workgroup   bed / ICU admission in priority order           (:321-367,617-651)
//   k_contacts  grid          one lane per sampled contact: LDS-staged contact tables, target
//                             gather, Bernoulli transmission, atomicMin winner claim (:1290-1304,
//                             1525-1573,908-934)
//   k_install   grid          winners become INCUBATION (severity + incubation draw) (:209-235)
//
// The path is HBM-bound integer / RNG work: no MFMA.  Wave64 ballots compact rare events,
// per-lane Philox keys every decision by (agent, day, purpose) so results do not depend on launch
// geometry.  Integer results are bit-identical to oracle/reina_par.c by construction (same
// primitives header, FP contraction off).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <climits>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/reina_hip.h"
#include "reina_prims.h"
#include "reina_sample.h"

#define CNT_IDX(c, age) ((c) * REINA_MAX_AGES + (age))
#define SC_IDX(s) (REINA_C_NR * REINA_MAX_AGES + (s))

enum { EV_HOSPITALIZE = 0, EV_TO_ICU = 1, EV_RELEASE_WARD = 2, EV_RELEASE_ICU = 3 };

// ---------------------------------------------------------------------------------------------
// device-side parameter block (one per engine, lives in HBM, read through the scalar/L1 caches)
struct DevParams {
    reina_disease_t dis;
    int32_t age_start[REINA_MAX_AGES + 1];
    uint32_t n_agents, nr_ages, nr_variants;
    uint32_t k0, k1;
    uint32_t max_work_items, max_candidates, max_queue;
    // contact tables
    float nrc[REINA_MAX_AGES];
    int32_t tcount[REINA_MAX_AGES];
    float mask_p[REINA_MAX_AGES][8];
    // sharding
    uint32_t iot_mask[REINA_MAX_VARIANTS];   // bit (day+10) set when infectiousness_over_time[v][day+10] != 0
    uint32_t n_shards, shard_rank, mirror_slots;
    float psus_max[REINA_MAX_VARIANTS];
    uint32_t n_ranges;
    int32_t range_min[REINA_MAX_RANGES], range_max[REINA_MAX_RANGES];
};

struct Tables {  // bigger tables staged into LDS by k_contacts
    uint32_t thr[REINA_MAX_AGES][REINA_MAX_ENTRIES];
    uint32_t meta[REINA_MAX_AGES][REINA_MAX_ENTRIES];
};

// what a kernel needs to know about one engine instance.  Every kernel takes an array of these and
// works on element blockIdx.y: a single engine launches with grid.y = 1, a Monte-Carlo group of K
// engines with grid.y = K (one launch per phase for the whole ensemble).
struct MemberRef {
    const DevParams *P;
    const Tables *T;
    reina_buffers_t B;
    int32_t *history_base;  // group runs: row k of this member's history is history_base + k * COUNTER_WORDS
};

static thread_local std::string g_last_error;

struct reina_engine {
    reina_config_t cfg;
    reina_buffers_t buf;
    bool bound = false;
    DevParams *d_params = nullptr;
    Tables *d_tables = nullptr;
    MemberRef *d_ref = nullptr;   // {d_params, d_tables, buf, no history base} for single-engine launches
    DevParams h_params;
    Tables h_tables;
    // pinned staging ring for table uploads: the copies are queued behind the days already issued
    // without stalling the host (a pageable source would drain the stream first)
    // (a slot is reused once its copy has left it; while the host runs far ahead of the GPU new
    // slots are added instead of waiting, up to MAX_STAGES)
    struct Stage { DevParams p; Tables t; };
    static constexpr size_t MAX_STAGES = 32;
    std::vector<Stage *> stage;           // hipHostMalloc'd, one per slot
    std::vector<hipEvent_t> stage_ev;
    bool testing_ever = false;
    int uniform_meta = 0;
    // (running independent kernels of a day on a second stream was measured on MI355X / ROCm 7.2:
    // the cross-stream event waits cost more than the overlap wins back -- HUS 0.108 -> 0.127 ms/day,
    // 50 M agents 0.315 -> 0.311 -- so independent phases share ONE launch instead: k_hosp_contacts)
    // profiling
    bool profile = false;
    uint32_t profile_stride = 1;  // time the scan launch of every profile_stride-th day
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<std::pair<size_t, size_t>> scan_pairs, day_pairs;
    size_t ev_day0 = 0;
    double scan_ms = 0, all_ms = 0;
    uint64_t scan_launches = 0;
};

#define HIP_CHECK(x)                                                                         \
    do {                                                                                     \
        hipError_t _e = (x);                                                                 \
        if (_e != hipSuccess) {                                                              \
            g_last_error = std::string(#x) + ": " + hipGetErrorString(_e);                   \
            return REINA_E_HIP;                                                              \
        }                                                                                    \
    } while (0)

// ---------------------------------------------------------------------------------------------
// small device helpers

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// One slot per calling lane from a global counter, one atomic per wave (ballot + popcount).
__device__ __forceinline__ uint32_t wave_alloc(int32_t *ctr) {
    uint64_t m = __ballot(1);
    uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
    uint32_t base = 0;
    if (rank == 0) base = (uint32_t)atomicAdd(ctr, (int)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    return base + rank;
}

__device__ __forceinline__ void set_problem(int32_t *counters, int p) {
    atomicCAS(&counters[SC_IDX(REINA_S_PROBLEM)], 0, p);
}

// age of sorted agent index i; `as` = age_start (LDS or global), search within [lo, hi]
__device__ __forceinline__ int age_of(const int32_t *as, uint32_t i, int lo, int hi) {
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if ((uint32_t)as[mid] <= i)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ uint32_t ld_hot(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_claim(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Disease.get_symptom_severity (main.pyx:1042-1091) in float32; both FATAL branches are
// DEATH_OUTSIDE_HOSPITAL (quirk Q2); variant-0 tables (quirk Q3)
__device__ int severity_of(const reina_disease_t &d, int age, float val, float vmod, int *pod_outside) {
    float syc = d.p_symptomatic[age];
    *pod_outside = 0;
    if (val >= syc) return RV_ASYMPTOMATIC;
    syc *= vmod;
    float dohc = d.p_death_outside_hospital[age];
    if (dohc != 0.0f) {
        if (val < dohc * syc) {
            *pod_outside = 1;
            return RV_FATAL;
        }
        val = (val - dohc) / (1.0f - dohc);
    }
    float sc = d.p_severe_given_symptomatic[age];
    float cc = d.p_critical_given_severe[age];
    float fc = d.p_fatal_given_critical[age];
    if (val < fc * cc * sc * syc) {
        *pod_outside = 1;
        return RV_FATAL;
    }
    if (val < cc * sc * syc) return RV_CRITICAL;
    if (val < sc * syc) return RV_SEVERE;
    return RV_MILD;
}

__device__ __forceinline__ uint32_t clamp_days(int32_t *counters, int d) {
    if (d < 0) d = 0;
    if (d > 255) {
        set_problem(counters, REINA_PROBLEM_DAYS_OVERFLOW);
        d = 255;
    }
    return (uint32_t)d;
}

// person_infect (main.pyx:209-235) + Population.infect (:1576-1582).  `expect` is the susceptible
// word the caller saw; the CAS makes duplicate winner records install once.
// `src_word`: the source's hot word if the caller already holds it (saves a dependent load), else 0
// with src_known = false.
__device__ bool install_infection(const DevParams *P, const reina_buffers_t &B, const int32_t *age_start, uint32_t t, uint32_t expect,
                                  uint32_t day, uint32_t variant, int32_t src, int fresh,
                                  uint32_t testing_mode, int32_t *new_by_age, int32_t *new_by_variant,
                                  bool src_known = false, uint32_t src_word = 0) {
    int age = age_of(age_start, t, 0, (int)P->nr_ages - 1);
    rp_u4 r = rp_philox(P->k0, P->k1, t, day, RP_P_INFECT, 0);
    float val = rp_uniform24(r.v[0]);
    float vmod = 1.0f;
    if ((expect & RH_VACCINATED) && ((int)day - B.vacc_day[t] > 14)) vmod = 0.1f;
    int pod = 0;
    int sev = severity_of(P->dis, age, val, vmod, &pod);
    float g = rp_gamma_mu_cv(P->dis.mean_incubation_duration[0], 0.86f, P->k0, P->k1, t, day, RP_P_INFECT, 1);
    uint32_t dl = clamp_days(B.counters, rp_round_to_int(g));
    uint32_t nw = RS_INCUBATION | ((uint32_t)sev << 3) | (variant << 8) | (pod ? RH_POD_OUTSIDE : 0u) |
                  (fresh ? RH_FRESH : 0u) | (expect & RH_VACCINATED) |
                  (testing_mode == RT_ALL_WITH_SYMPTOMS_CT ? RH_HASLIST : 0u) | (dl << 16);
    if (atomicCAS(&B.hot[t], expect, nw) != expect) return false;
    atomicAnd(&B.sus_bits[t >> 5], ~(1u << (t & 31u)));
    if (src >= 0) {
        B.infector[t] = src;
        if (!src_known) src_word = ld_hot(&B.hot[src]);
        // the count and the list head are bumped side by side (two independent round trips); a
        // 65th infectee fails the whole simulation (TOO_MANY_INFECTEES), so its link does not matter
        const int old = atomicAdd(&B.n_infected[src], 1);
        if (src_word & RH_HASLIST) {
            B.next_sibling[t] = atomicExch(&B.first_infectee[src], (int32_t)t);
            if (old >= 64) set_problem(B.counters, 1 /* TOO_MANY_INFECTEES */);
        }
    }
    atomicAdd(&new_by_age[age], 1);          // workgroup-local (LDS) histograms,
    atomicAdd(&new_by_variant[variant], 1);  // flushed once per workgroup by flush_new_infections
    return true;
}

// Population.infect counters (main.pyx:1576-1582) for a workgroup's worth of new infections
__device__ void flush_new_infections(const reina_buffers_t &B, int32_t *new_by_age, int32_t *new_by_variant,
                                     int nthreads) {
    __syncthreads();
    for (int k = threadIdx.x; k < REINA_MAX_AGES; k += nthreads) {
        int32_t v = new_by_age[k];
        if (v) {
            atomicAdd(&B.counters[CNT_IDX(REINA_C_SUSCEPTIBLE, k)], -v);
            atomicAdd(&B.counters[CNT_IDX(REINA_C_INFECTED, k)], v);
            atomicAdd(&B.counters[CNT_IDX(REINA_C_ALL_INFECTED, k)], v);
            atomicAdd(&B.counters[CNT_IDX(REINA_C_NEW_INFECTIONS, k)], v);
            new_by_age[k] = 0;
        }
    }
    if (threadIdx.x < REINA_MAX_VARIANTS) {
        int32_t v = new_by_variant[threadIdx.x];
        if (v) atomicAdd(&B.counters[SC_IDX(REINA_S_INFECTED_BY_VARIANT + threadIdx.x)], v);
        new_by_variant[threadIdx.x] = 0;
    }
    __syncthreads();
}

// The test-queue and weekly-import workgroups share the launch of the day's opening workgroup
// (k_open): before they add to the counters they wait here until it has taken the history snapshot
// and zeroed the daily counters.  The wait
// is one-directional (on a workgroup with a lower index of the same launch, which the dispatcher
// starts first), and bounded: after 20 ms the day is flagged failed instead of hanging the device.
__device__ __forceinline__ void wait_day_open(const reina_buffers_t &B, uint32_t day) {
    if (threadIdx.x == 0) {
        const uint64_t t0 = wall_clock64();
        while (__hip_atomic_load(&B.control[REINA_L_DAY_OPEN], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (int32_t)day + 1) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > 2000000ull) {  // 100 MHz ticks
                set_problem(B.counters, REINA_PROBLEM_SYNC_TIMEOUT);
                break;
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// k_init: _create_agents / _init_stats (main.pyx:1389-1450)
__global__ void k_init(const MemberRef *M_, int32_t beds, int32_t icu) {
    const MemberRef &mref_ = M_[blockIdx.y];
    const DevParams *P = mref_.P;
    const reina_buffers_t B = mref_.B;  // by value: pointers live in SGPRs, never re-read after stores
    uint32_t N = P->n_agents;
    uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
        B.hot[i] = 0;
        B.infector[i] = -1;
        B.n_infected[i] = 0;
        B.onset_days[i] = 0.0f;
        B.vacc_day[i] = -1;
        B.first_infectee[i] = -1;
        B.next_sibling[i] = -1;
        B.claim[i] = ~0ull;
    }
    if (P->n_shards > 1) {
        const size_t nm = (size_t)REINA_MAX_RANGES * REINA_MAX_VARIANTS * P->mirror_slots;
        for (size_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nm; k += stride) B.mirror[k] = ~0ull;
    }
    const uint32_t nwords = (N + 31u) / 32u + 1u;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nwords; k += stride) {
        uint32_t lo = k * 32u;
        uint32_t bits = 0;
        if (lo + 32u <= N) bits = 0xFFFFFFFFu;
        else if (lo < N) bits = (1u << (N - lo)) - 1u;
        B.sus_bits[k] = bits;
    }
    if (blockIdx.x == 0) {
        for (uint32_t k = threadIdx.x; k < REINA_COUNTER_WORDS; k += blockDim.x) {
            int32_t v = 0;
            if (k >= CNT_IDX(REINA_C_SUSCEPTIBLE, 0) && k < CNT_IDX(REINA_C_SUSCEPTIBLE, 0) + P->nr_ages) {
                uint32_t a = k - CNT_IDX(REINA_C_SUSCEPTIBLE, 0);
                v = P->age_start[a + 1] - P->age_start[a];
            }
            if (k == SC_IDX(REINA_S_AVAILABLE_BEDS) || k == SC_IDX(REINA_S_BEDS)) v = beds;
            if (k == SC_IDX(REINA_S_AVAILABLE_ICU) || k == SC_IDX(REINA_S_ICU_UNITS)) v = icu;
            B.counters[k] = v;
        }
        for (uint32_t k = threadIdx.x; k < REINA_L_NR; k += blockDim.x)
            B.control[k] = (k >= REINA_L_VACC_CURSOR && k < REINA_L_VACC_CURSOR + REINA_MAX_VACCINATIONS) ? INT_MIN : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// k_prologue: single workgroup of 1024 threads.
#define PRO_THREADS 1024
#define PRO_MAX_IMPORTS 16384

// Population.infect_people / get_import_infection_person (main.pyx:1632-1665), parallel form.
// Each import owns up to 10 tries (draws keyed by import number and try).  In a round every
// unplaced import walks its remaining tries to the first one that hits a never-infected agent and
// proposes it (atomicMin claim); a target proposed by several imports goes to the lowest import
// number, the others go on with their next try in the next round.  Usually one round.
__device__ __forceinline__ bool import_target(const DevParams *P, const int32_t *s_age_start, const reina_day_t &dp,
                                              uint32_t j, uint32_t k, uint32_t *t_out) {
    const reina_disease_t &d = P->dis;
    rp_u4 r = rp_philox(P->k0, P->k1, j, dp.day, RP_P_IMPORT, k);
    float p = rp_uniform24(r.v[0]);
    uint32_t c = d.n_import_classes - 1;
    for (uint32_t q = 0; q < d.n_import_classes; q++)
        if (p <= d.import_class_cum[q]) {
            c = q;
            break;
        }
    uint32_t start = (uint32_t)s_age_start[d.import_class_min_age[c]];
    uint32_t end = (uint32_t)s_age_start[d.import_class_max_age[c] + 1];
    if (end <= start) return false;
    *t_out = start + r.v[1] % (end - start);
    return true;
}

__device__ void pro_imports(const DevParams *P, const reina_buffers_t &B, const reina_day_t &dp, int pre_init,
                            uint32_t *import_base, uint8_t *placed, uint32_t *s_unplaced,
                            int32_t *new_by_age, int32_t *new_by_variant, const int32_t *s_age_start,
                            bool wait_for_open = false) {
    uint32_t total = 0;
    for (uint32_t b = 0; b < dp.n_import_batches; b++)
        if ((int)dp.import_batches[b].pre_init == pre_init) total += dp.import_batches[b].count;
    if (total == 0) return;
    if (total > PRO_MAX_IMPORTS) {
        if (threadIdx.x == 0) set_problem(B.counters, REINA_PROBLEM_WORK_OVERFLOW);
        total = PRO_MAX_IMPORTS;
    }
    // placed[j]: next try (0..10), 255 = placed; bit 7 of (try | 0x80) marks "proposed this round"
    for (uint32_t j = threadIdx.x; j < total; j += PRO_THREADS) placed[j] = 0;
    __syncthreads();
    const uint32_t base = *import_base;
    for (uint32_t round = 0; round < 10; round++) {
        int proposals = 0;
        for (uint32_t j = threadIdx.x; j < total; j += PRO_THREADS) {
            if (placed[j] == 255) continue;
            uint32_t k = placed[j], t = 0;
            bool found = false;
            for (; k < 10; k++) {
                if (import_target(P, s_age_start, dp, base + j, k, &t) && RH_STATE(ld_hot(&B.hot[t])) == RS_SUSCEPTIBLE) {
                    found = true;
                    break;
                }
            }
            if (found) {
                atomicMin((unsigned long long *)&B.claim[t], (unsigned long long)rp_order_key(dp.day, 0xFFFFFu - round, j));
                placed[j] = (uint8_t)(0x80u | k);  // proposed try k
                proposals++;
            } else {
                placed[j] = 10;
            }
        }
        if (__syncthreads_or(proposals) == 0) break;
        for (uint32_t j = threadIdx.x; j < total; j += PRO_THREADS) {
            const uint8_t st = placed[j];
            if (st == 255 || !(st & 0x80u)) continue;
            const uint32_t k = st & 0x7Fu;
            uint32_t t = 0;
            import_target(P, s_age_start, dp, base + j, k, &t);
            placed[j] = (uint8_t)(k + 1);
            if (ld_claim(&B.claim[t]) == rp_order_key(dp.day, 0xFFFFFu - round, j)) {
                // variant of import j: walk the batches of this phase
                uint32_t variant = 0, acc = 0;
                for (uint32_t b = 0; b < dp.n_import_batches; b++) {
                    if ((int)dp.import_batches[b].pre_init != pre_init) continue;
                    acc += dp.import_batches[b].count;
                    if (j < acc) {
                        variant = dp.import_batches[b].variant;
                        break;
                    }
                }
                uint32_t w = ld_hot(&B.hot[t]);
                install_infection(P, B, s_age_start, t, w, dp.day, variant, -1, 1, dp.testing_mode, new_by_age, new_by_variant);
                placed[j] = 255;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *s_unplaced = 0;
    if (wait_for_open) wait_day_open(B, dp.day);   // agents are placed; the counters follow the daily zeroing
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t j = threadIdx.x; j < total; j += PRO_THREADS)
        if (placed[j] != 255) mine++;
    if (mine) atomicAdd(s_unplaced, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (*s_unplaced) atomicAdd(&B.counters[SC_IDX(REINA_S_UNABLE_TO_IMPORT)], (int)*s_unplaced);
        *import_base += total;
    }
    flush_new_infections(B, new_by_age, new_by_variant, PRO_THREADS);
}

// HealthcareSystem.vaccinate_people (main.pyx:560-583): oldest first from a persistent cursor.
__device__ void pro_vaccinate(const DevParams *P, const reina_buffers_t &B, const reina_day_t &dp,
                              uint32_t *s_wave_cnt, int32_t *s_scalar) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t k = 0; k < dp.n_vaccinations; k++) {
        const reina_vaccination_t v = dp.vaccinations[k];
        int32_t c = B.control[REINA_L_VACC_CURSOR + v.slot];
        if (c == INT_MIN) c = (int32_t)v.idx_end - 1;
        uint32_t nr = v.nr, done = 0;
        if (nr > v.idx_end - v.idx_start) nr = v.idx_end - v.idx_start;
        while (done < nr && c >= (int32_t)v.idx_start) {
            int32_t i = c - tid;
            bool in_range = i >= (int32_t)v.idx_start;
            uint32_t w = 0;
            bool elig = false;
            if (in_range) {
                w = ld_hot(&B.hot[i]);
                elig = !(RH_STATE(w) == RS_DEAD || (w & (RH_VACCINATED | RH_DETECTED)));
            }
            uint64_t m = __ballot(elig);
            uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) s_wave_cnt[wave] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (int wv = 0; wv < PRO_THREADS / 64; wv++) {
                uint32_t n = s_wave_cnt[wv];
                if (wv < wave) before += n;
                total += n;
            }
            uint32_t pos = done + before + rank;  // 0-based order among eligible, oldest first
            if (elig && pos < nr) {
                atomicOr(&B.hot[i], RH_VACCINATED);
                B.vacc_day[i] = (int32_t)dp.day;
                atomicAdd(&B.counters[CNT_IDX(REINA_C_VACCINATED, age_of(P->age_start, (uint32_t)i, 0, (int)P->nr_ages - 1))], 1);
                if (pos == nr - 1) *s_scalar = i - 1;  // the sequential loop stops right after this one
            }
            __syncthreads();
            if (done + total >= nr) {
                c = *s_scalar;
                done = nr;
            } else {
                done += total;
                c -= PRO_THREADS;
            }
            __syncthreads();
        }
        if (c < (int32_t)v.idx_start - 1) c = (int32_t)v.idx_start - 1;
        if (tid == 0) B.control[REINA_L_VACC_CURSOR + v.slot] = c;
        __syncthreads();
    }
}

__device__ __forceinline__ void prologue_block(const MemberRef *M_, const reina_day_t &dp, uint32_t hist_slot, int weekly_elsewhere) {
    const MemberRef &mref_ = M_[blockIdx.y];
    const DevParams *P = mref_.P;
    const reina_buffers_t B = mref_.B;  // by value: pointers live in SGPRs, never re-read after stores
    // (dp itself stays untouched: writing a field of the by-value kernel argument would spill the whole struct)
    int32_t *const history_row =
        mref_.history_base ? mref_.history_base + (size_t)hist_slot * REINA_COUNTER_WORDS : dp.history_row;
    __shared__ uint8_t placed[PRO_MAX_IMPORTS];
    __shared__ uint32_t s_wave_cnt[PRO_THREADS / 64];
    __shared__ uint32_t s_unplaced;
    __shared__ int32_t s_scalar;
    __shared__ uint32_t s_import_base;
    __shared__ int32_t new_by_age[REINA_MAX_AGES];
    __shared__ int32_t new_by_variant[REINA_MAX_VARIANTS];
    __shared__ int32_t s_age_start[REINA_MAX_AGES + 1];
    const int tid = threadIdx.x;
#ifdef REINA_OPEN_STAMPS
    const uint64_t ps_t = wall_clock64();
#endif
    if (tid <= REINA_MAX_AGES) s_age_start[tid] = P->age_start[tid];
    if (tid < REINA_MAX_AGES) new_by_age[tid] = 0;
    if (tid < REINA_MAX_VARIANTS) new_by_variant[tid] = 0;
    // (the scalars thread 0 rewrites below are requested now, beside the snapshot loads)
    int32_t pre_beds = 0, pre_abeds = 0, pre_icu = 0, pre_aicu = 0, pre_qlen = 0;
    if (tid == 0) {
        pre_beds = B.counters[SC_IDX(REINA_S_BEDS)];
        pre_abeds = B.counters[SC_IDX(REINA_S_AVAILABLE_BEDS)];
        pre_icu = B.counters[SC_IDX(REINA_S_ICU_UNITS)];
        pre_aicu = B.counters[SC_IDX(REINA_S_AVAILABLE_ICU)];
        pre_qlen = B.control[(dp.day & 1) ? REINA_L_QUEUE1 : REINA_L_QUEUE0];
    }
    // generate_state() is taken BEFORE iterate() (calc/simulation.py:195 vs :270)
    if (history_row)
        for (int k = tid; k < REINA_COUNTER_WORDS; k += PRO_THREADS) history_row[k] = B.counters[k];
    __syncthreads();
    if (tid == 0) {
        s_import_base = 0;
        B.counters[SC_IDX(REINA_S_DAY)] = (int32_t)dp.day + 1;
        B.counters[SC_IDX(REINA_S_BEDS)] = pre_beds + dp.add_beds;
        B.counters[SC_IDX(REINA_S_AVAILABLE_BEDS)] = pre_abeds + dp.add_beds;
        B.counters[SC_IDX(REINA_S_ICU_UNITS)] = pre_icu + dp.add_icu_units;
        B.counters[SC_IDX(REINA_S_AVAILABLE_ICU)] = pre_aicu + dp.add_icu_units;
    }
    __syncthreads();
    pro_imports(P, B, dp, 1, &s_import_base, placed, &s_unplaced, new_by_age, new_by_variant, s_age_start);
    // Population.init_day (main.pyx:1687-1699) + Context._iterate zeroing (:1998-2000)
    __syncthreads();
    for (int k = tid; k < (int)P->nr_ages; k += PRO_THREADS) {
        B.counters[CNT_IDX(REINA_C_NEW_INFECTIONS, k)] = 0;
        B.counters[CNT_IDX(REINA_C_DETECTED, k)] = 0;
    }
    for (int k = tid; k < REINA_PRESSURE_WORDS; k += PRO_THREADS) B.pressure[k] = 0;
    if (tid < REINA_NR_PLACES) B.counters[SC_IDX(REINA_S_DAILY_CONTACTS + tid)] = 0;
    if (tid < REINA_MAX_VARIANTS) B.counters[SC_IDX(REINA_S_INFECTED_BY_VARIANT + tid)] = 0;
    if (tid == 0) {
        B.counters[SC_IDX(REINA_S_TOTAL_INFECTORS)] = 0;
        B.counters[SC_IDX(REINA_S_TOTAL_INFECTIONS)] = 0;
        B.counters[SC_IDX(REINA_S_EXPOSED_PER_DAY)] = 0;
        B.control[REINA_L_WORK] = 0;
        B.control[REINA_L_CAND] = 0;
        B.control[REINA_L_HOSP] = 0;
        B.control[REINA_L_CONTACTS] = 0;
        B.control[REINA_L_HOSP_ADMIT] = 0;
        B.control[REINA_L_ICU_ADMIT] = 0;
        // HealthcareSystem.iterate: ct_cases_per_day = len(queue) (main.pyx:518-519)
        B.counters[SC_IDX(REINA_S_CT_CASES_PER_DAY)] = pre_qlen;
    }
    // the day is open: the test-queue workgroups of the same launch (k_open) may start; what follows
    // (weekly imports) only touches never-infected agents and the infection counters
    __threadfence();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&B.control[REINA_L_DAY_OPEN], (int32_t)dp.day + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#ifdef REINA_OPEN_STAMPS
    if (tid == 0) atomicAdd((unsigned long long *)&B.mirror[5], (unsigned long long)(wall_clock64() - ps_t));
#endif
    // weekly imports (Population.infect_people_daily, main.pyx:1671-1685) run after init_day's
    // zeroing; vaccination follows the test-queue pass in the reference (main.pyx:547-558) and is
    // launched from k_vaccinate after both.
    __syncthreads();
    if (!weekly_elsewhere)
        pro_imports(P, B, dp, 0, &s_import_base, placed, &s_unplaced, new_by_age, new_by_variant, s_age_start);
#ifdef REINA_OPEN_STAMPS
    if (tid == 0) atomicAdd((unsigned long long *)&B.mirror[6], (unsigned long long)(wall_clock64() - ps_t));
#endif
}

// The weekly imports in a workgroup of their own (days without intervention imports, whose claim
// keys they would share): agents are drawn, claimed and infected from the first instruction of the
// launch; only the counter updates wait for the opening workgroup's daily zeroing.
__device__ __forceinline__ void weekly_imports_block(const MemberRef *M_, const reina_day_t &dp) {
    const MemberRef &mref_ = M_[blockIdx.y];
    const DevParams *P = mref_.P;
    const reina_buffers_t B = mref_.B;
    __shared__ uint8_t placed[PRO_MAX_IMPORTS];
    __shared__ uint32_t s_unplaced;
    __shared__ uint32_t s_import_base;
    __shared__ int32_t new_by_age[REINA_MAX_AGES];
    __shared__ int32_t new_by_variant[REINA_MAX_VARIANTS];
    __shared__ int32_t s_age_start[REINA_MAX_AGES + 1];
    const int tid = threadIdx.x;
    if (tid <= REINA_MAX_AGES) s_age_start[tid] = P->age_start[tid];
    if (tid < REINA_MAX_AGES) new_by_age[tid] = 0;
    if (tid < REINA_MAX_VARIANTS) new_by_variant[tid] = 0;
    if (tid == 0) s_import_base = 0;
    __syncthreads();
    pro_imports(P, B, dp, 0, &s_import_base, placed, &s_unplaced, new_by_age, new_by_variant, s_age_start, true);
}

__global__ __launch_bounds__(PRO_THREADS) void k_vaccinate(const MemberRef *M_, reina_day_t dp) {
    const MemberRef &mref_ = M_[blockIdx.y];
    const DevParams *P = mref_.P;
    const reina_buffers_t B = mref_.B;  // by value: pointers live in SGPRs, never re-read after stores
    __shared__ uint32_t s_wave_cnt[PRO_THREADS / 64];
    __shared__ int32_t s_scalar;
    pro_vaccinate(P, B, dp, s_wave_cnt, &s_scalar);
}

// ---------------------------------------------------------------------------------------------
// testing queue + contact tracing (HealthcareSystem.iterate main.pyx:514-545,
// perform_contact_tracing :495-512, queue_for_testing :474-488)

__device__ __forceinline__ void queue_append(const DevParams *P, const reina_buffers_t &B, int which, uint32_t idx) {
    uint32_t pos = wave_alloc(&B.control[which ? REINA_L_QUEUE1 : REINA_L_QUEUE0]);
    if (pos >= P->max_queue) {
        set_problem(B.counters, REINA_PROBLEM_QUEUE_OVERFLOW);
        return;
    }
    (which ? B.queue1 : B.queue0)[pos] = idx;
}

// Q1: every queued test is positive (quirk Q8): clear QUEUED, set DETECTED
__device__ __forceinline__ void test_detect_block(const MemberRef *M_, const reina_day_t &dp, uint32_t bx, uint32_t nbx) {
    const MemberRef &mref_ = M_[blockIdx.y];
    const DevParams *P = mref_.P;
    const reina_buffers_t B = mref_.B;  // by value: pointers live in SGPRs, never re-read after stores
    __shared__ int32_t s_det[REINA_MAX_AGES];
    __shared__ int32_t s_age_start[REINA_MAX_AGES + 1];
    if (B.control[(dp.day & 1) ? REINA_L_QUEUE1 : REINA_L_QUEUE0] <= (int)(bx * blockDim.x)) return;
    if (threadIdx.x <= REINA_MAX_AGES) s_age_start[threadIdx.x] = P->age_start[threadIdx.x];
    if (threadIdx.x < REINA_MAX_AGES) s_det[threadIdx.x] = 0;
    __syncthreads();
    const int cur = dp.day & 1;
    const uint32_t *q = cur ? B.queue1 : B.queue0;
    const int n = B.control[cur ? REINA_L_QUEUE1 : REINA_L_QUEUE0];
    for (int k = bx * blockDim.x + threadIdx.x; k < n; k += nbx * blockDim.x) {
        uint32_t i = q[k];
        uint32_t w = B.hot[i];
        if (w & RH_DETECTED) set_problem(B.counters, 7 /* WRONG_STATE */);
        B.hot[i] = (w & ~RH_QUEUED) | RH_DETECTED;
        atomicAdd(&s_det[age_of(s_age_start, i, 0, (int)P->nr_ages - 1)], 1);
    }
    // hot words, queues and lists are free to touch from the first instruction of the launch; the
    // detection COUNTERS wait for the opening workgroup's snapshot + daily zeroing (normally long done)
    wait_day_open(B, dp.day);
    if (threadIdx.x < REINA_MAX_AGES && s_det[threadIdx.x]) {
        atomicAdd(&B.counters[CNT_IDX(REINA_C_DETECTED, threadIdx.x)], s_det[threadIdx.x]);
        atomicAdd(&B.counters[CNT_IDX(REINA_C_ALL_DETECTED, threadIdx.x)], s_det[threadIdx.x]);
    }
}

// the tracing success roll is keyed by (candidate, tracer): the SET of queued agents is order-free
__device__ __forceinline__ bool try_queue(const DevParams *P, const reina_buffers_t &B, uint32_t cand,
                                          uint32_t tracer, const reina_day_t &dp) {
    uint32_t w = ld_hot(&B.hot[cand]);
    if (RH_STATE(w) == RS_DEAD || (w & (RH_DETECTED | RH_QUEUED))) return false;
    rp_u4 r = rp_philox(P->k0, P->k1, cand, dp.day, RP_P_TRACE, tracer);
    if (!rp_chance(dp.p_successful_tracing, r.v[0])) return false;
    uint32_t old = atomicOr(&B.hot[cand], RH_QUEUED);
    return !(old & RH_QUEUED);
}

// level 0 (from the detected queue, accepted candidates also go to the level-1 list) and
// level 1 (from the level-1 list, no further recursion)
// Level 0 also performs the detection of its queue entry (k_test_detect's job) in the same pass:
// every member of today's queue carries QUEUED until its single store replaces it with DETECTED,
// so a tracer can never re-queue another member, whichever of the two runs first.
#ifdef REINA_OPEN_STAMPS
#define OSTAMP(k) do { if (threadIdx.x == 0) { uint64_t t_ = wall_clock64(); atomicAdd((unsigned long long *)&B.mirror[k], (unsigned long long)(t_ - os_t)); os_t = t_; } } while (0)
#else
#define OSTAMP(k) do { } while (0)
#endif
#define TRACE_STAGE 4096
// successes of a workgroup are staged in LDS and appended to the global lists with ONE allocation
// per list at the end (a returning global atomic per success would sit in every lane's dependent
// chain); overflow of the stage falls back to direct appends
struct TraceStage {
    uint32_t n;
    uint32_t base_q, base_l;
    uint32_t item[TRACE_STAGE];
};
template <int LEVEL>
__device__ __forceinline__ void trace_accept(const DevParams *P, const reina_buffers_t &B, TraceStage &S, int nxt, uint32_t cand) {
    const uint32_t pos = atomicAdd(&S.n, 1u);
    if (pos < TRACE_STAGE) {
        S.item[pos] = cand;
        return;
    }
    queue_append(P, B, nxt, cand);
    if (LEVEL == 0) {
        uint32_t p1 = wave_alloc(&B.control[REINA_L_LEVEL1]);
        if (p1 < P->max_queue) B.level1[p1] = cand; else set_problem(B.counters, REINA_PROBLEM_QUEUE_OVERFLOW);
    }
}

template <int LEVEL, bool FOLD = false>
__device__ __forceinline__ void test_trace_block(const MemberRef *M_, const reina_day_t &dp, uint32_t bx, uint32_t nbx) {
    const MemberRef &mref_ = M_[blockIdx.y];
    const DevParams *P = mref_.P;
    const reina_buffers_t B = mref_.B;  // by value: pointers live in SGPRs, never re-read after stores
    __shared__ int32_t s_det[REINA_MAX_AGES];
    __shared__ int32_t s_age_start[REINA_MAX_AGES + 1];
    __shared__ TraceStage S;
    const int cur = dp.day & 1, nxt = cur ^ 1;
    const uint32_t *src = LEVEL == 0 ? (cur ? B.queue1 : B.queue0) : B.level1;
    const int n = LEVEL == 0 ? B.control[cur ? REINA_L_QUEUE1 : REINA_L_QUEUE0] : B.control[REINA_L_LEVEL1];
    if (n <= (int)(bx * blockDim.x)) return;
#ifdef REINA_OPEN_STAMPS
    uint64_t os_t = wall_clock64();
#endif
    if (LEVEL == 0) {
        if (threadIdx.x <= REINA_MAX_AGES) s_age_start[threadIdx.x] = P->age_start[threadIdx.x];
        if (threadIdx.x < REINA_MAX_AGES) s_det[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) S.n = 0;
    __syncthreads();
    for (int k = bx * blockDim.x + threadIdx.x; k < n; k += nbx * blockDim.x) {
        const uint32_t i = src[k];
        // everything that depends only on i is requested together
        const int32_t inf = B.infector[i];
        int32_t c = B.first_infectee[i];
        uint32_t wi;
        if (LEVEL == 0) {
            const uint32_t w0 = B.hot[i];
            if (w0 & RH_DETECTED) set_problem(B.counters, 7 /* WRONG_STATE */);
            wi = (w0 & ~RH_QUEUED) | RH_DETECTED;
            B.hot[i] = wi;
            atomicAdd(&s_det[age_of(s_age_start, i, 0, (int)P->nr_ages - 1)], 1);
        } else {
            wi = ld_hot(&B.hot[i]);
        }
        if (inf >= 0 && try_queue(P, B, (uint32_t)inf, i, dp)) trace_accept<LEVEL>(P, B, S, nxt, (uint32_t)inf);
        if (wi & RH_HASLIST) {
            while (c >= 0) {
                const int32_t next = B.next_sibling[c];   // requested beside the candidate's hot word
                if (try_queue(P, B, (uint32_t)c, i, dp)) trace_accept<LEVEL>(P, B, S, nxt, (uint32_t)c);
                c = next;
            }
        }
    }
    __syncthreads();
    {   // one allocation per list for the whole workgroup, then coalesced copies out of LDS
        const uint32_t cnt = S.n < TRACE_STAGE ? S.n : TRACE_STAGE;
        if (threadIdx.x == 0 && cnt) {
            S.base_q = (uint32_t)atomicAdd(&B.control[nxt ? REINA_L_QUEUE1 : REINA_L_QUEUE0], (int)cnt);
            if (LEVEL == 0) S.base_l = (uint32_t)atomicAdd(&B.control[REINA_L_LEVEL1], (int)cnt);
        }
        __syncthreads();
        if (cnt) {
            uint32_t *q = nxt ? B.queue1 : B.queue0;
            if (S.base_q + cnt > P->max_queue || (LEVEL == 0 && S.base_l + cnt > P->max_queue)) {
                if (threadIdx.x == 0) set_problem(B.counters, REINA_PROBLEM_QUEUE_OVERFLOW);
            } else {
                for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) {
                    const uint32_t x = S.item[k];
                    q[S.base_q + k] = x;
                    if (LEVEL == 0) B.level1[S.base_l + k] = x;
                }
            }
        }
    }
    OSTAMP(LEVEL == 0 ? 1 : 4);
    if (LEVEL == 0) {
        wait_day_open(B, dp.day);   // (includes a workgroup barrier) counters only after the snapshot + zeroing
        OSTAMP(2);
        if (threadIdx.x < REINA_MAX_AGES && s_det[threadIdx.x]) {
            atomicAdd(&B.counters[CNT_IDX(REINA_C_DETECTED, threadIdx.x)], s_det[threadIdx.x]);
            atomicAdd(&B.counters[CNT_IDX(REINA_C_ALL_DETECTED, threadIdx.x)], s_det[threadIdx.x]);
        }
    }
    if (LEVEL == 0 && FOLD) {
        // small populations: the level-0 workgroup that finishes last walks the level-1 list itself
        // (a few hundred entries) instead of a launch of its own
        __shared__ int s_last;
        uint32_t busy = ((uint32_t)n + blockDim.x - 1) / blockDim.x;   // workgroups that had queue entries
        if (busy > nbx) busy = nbx;
        if (busy == 1) {
            __syncthreads();   // the only one: its own list writes are visible to itself
            OSTAMP(3);
            test_trace_block<1>(M_, dp, 0, 1);
            return;
        }
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            const int done = atomicAdd(&B.control[REINA_L_TRACE_DONE], 1) + 1;
            s_last = done == (int)busy;
            if (s_last) {
                B.control[REINA_L_TRACE_DONE] = 0;
                __atomic_thread_fence(__ATOMIC_ACQUIRE);   // see the other workgroups' list entries and flags
            }
        }
        __syncthreads();
        OSTAMP(3);
        if (s_last) test_trace_block<1>(M_, dp, 0, 1);
    }
}

__global__ __launch_bounds__(256) void k_test_trace1(const MemberRef *M_, reina_day_t dp) {
    test_trace_block<1>(M_, dp, blockIdx.x, gridDim.x);
}

// k_open: the first launch of a day.  Workgroup 0 opens the day (prologue_block: snapshot, beds,
// intervention imports, daily zeroing), workgroup 1 places the weekly imports; workgroups 2.. work off the test
// queue (MODE 1: detection only, MODE 2: detection + level-0 contact tracing, MODE 3: the same with
// level 1 folded in) as soon as workgroup 0
// signals that the bookkeeping part is done -- the import placement that follows it touches only
// never-infected agents and infection counters, the test queue only infected agents and detection
// counters, so the two run side by side inside one launch.
template <int MODE>
__global__ __launch_bounds__(PRO_THREADS) void k_open(const MemberRef *M_, reina_day_t dp, uint32_t hist_slot, int weekly_own) {
    if (blockIdx.x == 0) {
        prologue_block(M_, dp, hist_slot, weekly_own);
    } else if (blockIdx.x == 1) {
        if (weekly_own) weekly_imports_block(M_, dp);
    } else if (MODE == 1) {
        test_detect_block(M_, dp, blockIdx.x - 2, gridDim.x - 2);
    } else if (MODE == 2) {
        test_trace_block<0>(M_, dp, blockIdx.x - 2, gridDim.x - 2);
    } else if (MODE == 3) {
        test_trace_block<0, true>(M_, dp, blockIdx.x - 2, gridDim.x - 2);
    }
}

// ---------------------------------------------------------------------------------------------
// k_scan: Context._process_person + person_advance for every agent (main.pyx:1968-1992,395-438)
//
// A pure streaming kernel: each lane loads 16 B (4 hot words) per 1-KiB wave load, two loads in
// flight per wave; a per-lane need-mask picks the words that are infected or removed-but-uncounted
// (everything else costs three integer ops).  The integer part of the state machine (countdown,
// recover / die at home, R marking) runs in place and changed words are stored back where they
// were loaded.  Everything that needs random numbers, gathers or counters is only RECORDED here,
// as (agent, word|kind) pairs appended with wave ballots to four per-wave slices -- no atomics,
// no LDS -- and executed densely by the kernels that follow:
//   exposure candidates  -> k_contacts (contact COUNT draw, then the contacts themselves)
//   symptom onsets       -> k_install  (gamma draw of the illness course, testing decision)
//   hospital events      -> k_hospital (bed / ICU requests and releases in priority order)
//   bookkeeping          -> k_install  (R statistics, recovered / died-at-home counters)
#define SCAN_THREADS 256
#define SCAN_WAVES (SCAN_THREADS / 64)
enum { SL_INFECTED = 0, SL_RECOVERED, SL_DEAD, SL_NHD, SL_NR };
enum { LIST_EXP = 0, LIST_ILL = 1, LIST_EV = 2, LIST_BOOK = 3, LIST_CAND = 4 };
enum { EVX_COUNT_R = 4, EVX_RECOVERED_HOME = 5, EVX_DIED_HOME = 6 };

// entries of scan wave `sw` start here in every list (a wave's slice is as large as the number
// of agents it scans: tiles sw, sw + W, ... of 512 agents)
__host__ __device__ __forceinline__ uint32_t scan_slice_base(uint32_t sw, uint32_t scan_waves, uint32_t scan_tiles) {
    const uint32_t tq = scan_tiles / scan_waves, tr = scan_tiles % scan_waves;
    return 512u * (sw * tq + (sw < tr ? sw : tr));
}

// person_become_ill (main.pyx:284-291, 989-1014) + seek_testing (:595-615); `w` already carries
// days_left == 0 from the countdown
// person_become_ill's durations (main.pyx:284-288,989-1014): onset->removed gamma, illness days
__device__ __forceinline__ uint32_t onset_word(const DevParams *P, const reina_buffers_t &B, uint32_t i, uint32_t w, uint32_t day) {
    const reina_disease_t &d = P->dis;
    int v = RH_VARIANT(w), sev = RH_SEV(w);
    float mu = sev == RV_FATAL ? d.mean_duration_from_onset_to_death[v] : d.mean_duration_from_onset_to_recovery[v];
    float od = rp_gamma_mu_cv(mu, 0.45f, P->k0, P->k1, i, day, RP_P_ONSET, 1);
    B.onset_days[i] = od;
    float f = od;
    if (sev >= RV_SEVERE) f *= d.ratio_of_duration_before_hospitalisation[v];
    w = RH_SET_STATE(w, RS_ILLNESS);
    w = RH_SET_DAYS_LEFT(w, clamp_days(B.counters, rp_round_to_int(f)));
    w = RH_SET_DOI(w, 0);
    return w;
}

__device__ void become_ill(const DevParams *P, const reina_buffers_t &B, const reina_day_t &dp, uint32_t i, uint32_t w) {
    const int sev = RH_SEV(w);
    w = onset_word(P, B, i, w, dp.day);
    if (sev != RV_ASYMPTOMATIC && !(w & RH_DETECTED)) {
        int q = 0;
        if (dp.testing_mode == RT_ALL_WITH_SYMPTOMS || dp.testing_mode == RT_ALL_WITH_SYMPTOMS_CT) {
            q = 1;
        } else if (dp.testing_mode == RT_ONLY_SEVERE_SYMPTOMS) {
            if (sev >= RV_SEVERE)
                q = 1;
            else
                q = rp_chance(dp.p_detected_anyway, rp_philox(P->k0, P->k1, i, dp.day, RP_P_ONSET, 0).v[3]);
        }
        if (q && !(w & RH_QUEUED)) {
            w |= RH_QUEUED;
            queue_append(P, B, (dp.day & 1) ^ 1, i);
        }
    }
    B.hot[i] = w;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

struct ScanLists {
    uint2 *l[4];
    uint32_t n[4];
};

__device__ __forceinline__ void list_push(ScanLists &L, int which, bool pred, uint32_t x, uint32_t y) {
    const uint64_t m = __ballot(pred);
    if (pred) L.l[which][L.n[which] + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull))] = make_uint2(x, y);
    L.n[which] += (uint32_t)__popcll(m);
}

// the in-place part for one hot word; returns the word to store (`w` itself when nothing changes
// or when the final word is written later by become_ill)
__device__ __forceinline__ uint32_t scan_word(const DevParams *P, ScanLists &L, bool valid, uint32_t i, uint32_t w) {
    const uint32_t st = RH_STATE(w);
    bool p_exp = false, p_ill = false, p_ev = false, p_bk = false;
    uint32_t kind = 0, nw = w;
    if (valid) {
        if (st >= RS_RECOVERED) {
            // removed, not yet counted into R (main.pyx:1969-1972): mark; n_infected is gathered later
            nw = w | RH_INCLUDED;
            p_bk = true;
            kind = EVX_COUNT_R;
        } else if (st == RS_INCUBATION && (w & RH_FRESH)) {
            nw = w & ~RH_FRESH;  // infected earlier today: waits (main.pyx:402)
        } else {
            const int v = RH_VARIANT(w), sev = RH_SEV(w);
            uint32_t dl = RH_DAYS_LEFT(w);
            if (st <= RS_ILLNESS) {
                if (!(w & RH_DETECTED)) {
                    const int dayrel = st == RS_INCUBATION ? -(int)dl : (int)RH_DOI(w);
                    p_exp = dayrel >= -10 && dayrel <= 10 && ((P->iot_mask[v] >> (dayrel + 10)) & 1u);
                }
                if (st == RS_INCUBATION) {
                    if (dl > 0) dl--;
                    nw = RH_SET_DAYS_LEFT(w, dl);
                    p_ill = dl == 0;  // become_ill stores the final word
                } else {
                    uint32_t doi = RH_DOI(w);
                    if (doi < 255) doi++;
                    if (dl > 0) dl--;
                    nw = RH_SET_DOI(RH_SET_DAYS_LEFT(w, dl), doi);
                    if (dl == 0) {
                        if (sev == RV_FATAL && (w & RH_POD_OUTSIDE)) {
                            p_bk = true;
                            kind = EVX_DIED_HOME;
                            nw = RH_SET_STATE(nw, RS_DEAD) & ~RH_HASLIST;
                        } else if (sev >= RV_SEVERE) {
                            p_ev = true;
                            kind = EV_HOSPITALIZE;
                        } else {
                            p_bk = true;
                            kind = EVX_RECOVERED_HOME;
                            nw = RH_SET_STATE(nw, RS_RECOVERED) & ~RH_HASLIST;
                        }
                    }
                }
            } else {
                if (dl > 0) dl--;
                nw = RH_SET_DAYS_LEFT(w, dl);
                if (dl == 0) {
                    p_ev = true;
                    kind = st == RS_HOSPITALIZED ? (sev >= RV_CRITICAL ? EV_TO_ICU : EV_RELEASE_WARD) : EV_RELEASE_ICU;
                }
            }
        }
    }
    if (__any(p_exp)) list_push(L, LIST_EXP, p_exp, i, w);
    if (__any(p_ill)) list_push(L, LIST_ILL, p_ill, i, nw);
    if (__any(p_ev)) list_push(L, LIST_EV, p_ev, i, kind);
    if (__any(p_bk)) list_push(L, LIST_BOOK, p_bk, i, kind);
    return p_ill ? w : nw;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan(const MemberRef *M_, reina_day_t dp) {
    const MemberRef &mref_ = M_[blockIdx.y];
    const DevParams *P = mref_.P;
    const reina_buffers_t B = mref_.B;  // by value: pointers live in SGPRs, never re-read after stores
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t N = P->n_agents;
    const uint32_t n4 = N >> 2;
    const uint4 *hot4 = reinterpret_cast<const uint4 *>(B.hot);
    // each wave walks tiles of 128 uint4 (two 1-KiB loads in flight per wave)
    const uint32_t tiles = (n4 + 127u) / 128u;
    const uint32_t wave_global = blockIdx.x * SCAN_WAVES + wave, waves_total = gridDim.x * SCAN_WAVES;
    ScanLists L;
    {
        const uint32_t cap = P->max_work_items, base = scan_slice_base(wave_global, waves_total, tiles);
        L.l[LIST_EXP] = reinterpret_cast<uint2 *>(B.work_items) + base;
        L.l[LIST_ILL] = reinterpret_cast<uint2 *>(B.work_items) + cap + base;
        L.l[LIST_EV] = reinterpret_cast<uint2 *>(B.scan_lists) + base;
        L.l[LIST_BOOK] = reinterpret_cast<uint2 *>(B.scan_lists) + cap + base;
        L.n[0] = L.n[1] = L.n[2] = L.n[3] = 0;
    }
    // software pipeline: the next tile's two 1-KiB loads are in flight while this tile is worked on
    uint4 na_ = make_uint4(0, 0, 0, 0), nb_ = make_uint4(0, 0, 0, 0);
    {
        const uint32_t q0 = wave_global * 128u + lane, q1 = q0 + 64u;
        if (wave_global < tiles) {
            if (q0 < n4) na_ = hot4[q0];
            if (q1 < n4) nb_ = hot4[q1];
        }
    }
    for (uint32_t t = wave_global; t < tiles; t += waves_total) {
        const uint32_t q0 = t * 128u + lane, q1 = q0 + 64u;
        const uint4 a = na_, b = nb_;
        {
            const uint32_t tn = t + waves_total;
            const uint32_t p0 = tn * 128u + lane, p1 = p0 + 64u;
            na_ = make_uint4(0, 0, 0, 0);
            nb_ = make_uint4(0, 0, 0, 0);
            if (tn + 1 < tiles) {          // interior tile: no bounds checks
                na_ = hot4[p0];
                nb_ = hot4[p1];
            } else if (tn < tiles) {
                if (p0 < n4) na_ = hot4[p0];
                if (p1 < n4) nb_ = hot4[p1];
            }
        }
        // which of this lane's 8 words need the state machine today: infected (state 1..4) or
        // removed but not yet counted into R.  With x = state | counted-bit, that is 1 <= x <= 6
        // (counted removed agents have x = 0x405 / 0x406, susceptible ones 0).
        uint32_t mask = 0;
#define NEED_BIT(wd, k) mask |= ((((wd) & 0x407u) - 1u) < 6u ? 1u : 0u) << (k);
        NEED_BIT(a.x, 0) NEED_BIT(a.y, 1) NEED_BIT(a.z, 2) NEED_BIT(a.w, 3)
        NEED_BIT(b.x, 4) NEED_BIT(b.y, 5) NEED_BIT(b.z, 6) NEED_BIT(b.w, 7)
#undef NEED_BIT
        // every round each lane takes its next needy word: rounds = max needy words per lane
        // (1-2 at a few % prevalence) instead of one pass per word position
        while (__any(mask != 0u)) {
            const bool valid = mask != 0u;
            const int k = valid ? (int)__ffs(mask) - 1 : 0;
            mask &= mask - 1u;
            uint32_t w = a.x;
            w = k == 1 ? a.y : w;
            w = k == 2 ? a.z : w;
            w = k == 3 ? a.w : w;
            w = k == 4 ? b.x : w;
            w = k == 5 ? b.y : w;
            w = k == 6 ? b.z : w;
            w = k == 7 ? b.w : w;
            const uint32_t i = k < 4 ? 4u * q0 + (uint32_t)k : 4u * q1 + (uint32_t)(k - 4);
            const uint32_t nw = scan_word(P, L, valid, i, w);
            // a deferred (onset) word keeps its old value here and is written by become_ill
            if (valid && nw != w) B.hot[i] = nw;
        }
    }
    if (wave_global == 0) {  // ragged tail: N not a multiple of 4
        uint32_t i = (n4 << 2) + lane;
        uint32_t w = 0;
        bool in = lane < (int)(N & 3u);
        if (in) w = B.hot[i];
        const bool need = in && (((w & 0x407u) - 1u) < 6u);
        uint32_t nw = scan_word(P, L, need, i, w);
        if (need && nw != w) B.hot[i] = nw;
    }
    if (lane < 4) {
        const uint32_t c = lane == 0 ? L.n[0] : lane == 1 ? L.n[1] : lane == 2 ? L.n[2] : L.n[3];
        B.work_counts[lane * REINA_MAX_SCAN_WAVES + wave_global] = c;
    }
}

// ---------------------------------------------------------------------------------------------
// k_hospital: person_hospitalize / transfer_to_icu / release_from_hospital (main.pyx:321-367) +
// HealthcareSystem bed accounting (:617-651). One workgroup; when capacity can bind, events are
// bitonic-sorted by (priority, agent) in LDS and the saturating bed/ICU walk is replayed in order.
#define HOSP_THREADS 1024

__device__ __forceinline__ int dies_in_hospital(const DevParams *P, uint32_t i, uint32_t day, int sev, int v, int care) {
    if (sev == RV_FATAL) return 1;
    float p = 0.0f;
    if (sev == RV_CRITICAL) {
        if (care) return 0;
        p = P->dis.p_icu_death_no_beds[v];
    } else if (sev == RV_SEVERE) {
        if (care) return 0;
        p = P->dis.p_hospital_death_no_beds[v];
    }
    return rp_chance(p, rp_philox(P->k0, P->k1, i, day, RP_P_HOSPITAL, 0).v[0]);
}

// saturating-counter walk as a scan: every event acts on the free-bed count x as
// f(x) = max(x + a, m) (admission: a=-1, m=0; release: a=+1, m=-inf); such maps compose to the
// same form, (a1,m1) then (a2,m2) = (a1+a2, max(m1+a2, m2)), so "free beds just before event k"
// is a prefix composition -- computed by a workgroup scan instead of a serial loop.
struct SatFn { int a, m; };
#define SAT_NEG (-(1 << 29))
__device__ __forceinline__ SatFn sat_then(SatFn f, SatFn g) {  // f first, then g
    SatFn r;
    r.a = f.a + g.a;
    int t = f.m + g.a;
    if (t < SAT_NEG) t = SAT_NEG;
    r.m = t > g.m ? t : g.m;
    return r;
}
__device__ __forceinline__ int sat_apply(SatFn f, int x) {
    int y = x + f.a;
    return y > f.m ? y : f.m;
}
__device__ __forceinline__ SatFn bed_fn(int type) {
    SatFn f;
    f.m = SAT_NEG;
    f.a = 0;
    if (type == EV_HOSPITALIZE) { f.a = -1; f.m = 0; }
    else if (type == EV_TO_ICU || type == EV_RELEASE_WARD) f.a = 1;
    return f;
}
__device__ __forceinline__ SatFn icu_fn(int type) {
    SatFn f;
    f.m = SAT_NEG;
    f.a = 0;
    if (type == EV_TO_ICU) { f.a = -1; f.m = 0; }
    else if (type == EV_RELEASE_ICU) f.a = 1;
    return f;
}

enum { HL_INFECTED = 0, HL_DETECTED, HL_ALL_DETECTED, HL_HOSPITALIZED, HL_IN_WARD, HL_IN_ICU, HL_CUM_ICU,
       HL_DEAD, HL_NHD, HL_RECOVERED, HL_NR };

// Bitonic sort of n (power of two, <= E * 1024) 64-bit keys in LDS by the 1024-thread workgroup.
// Thread t keeps elements t*E .. t*E+E-1 in registers: compare-exchange strides below E stay in
// registers, strides below 64*E are wave shuffles, only the strides of 64*E and above go through
// LDS (element r of thread t at [r][t], conflict-free) -- 10 LDS round trips for 16384 keys
// instead of 105.
template <int E, int STRD>
__device__ __forceinline__ void hosp_sort_reg(uint64_t (&x)[E], int size, int tid) {
#pragma unroll
    for (int r = 0; r < E; r++) {
        if ((r & STRD) == 0 && (r | STRD) < E) {
            const bool up = (((tid * E) + r) & size) == 0;
            const uint64_t a = x[r], b = x[r | STRD];
            const bool sw = (a > b) == up;
            x[r] = sw ? b : a;
            x[r | STRD] = sw ? a : b;
        }
    }
}

template <int E>
__device__ __forceinline__ void hosp_sort(uint64_t *ev, int n, int tid) {
    uint64_t x[E];
    const int T = n / E > 0 ? n / E : 1;   // threads that hold keys; the others idle through the barriers
    const bool active = tid < T;            // (events past n in the same LDS array must stay untouched)
#pragma unroll
    for (int r = 0; r < E; r++) x[r] = (tid * E + r < n) ? ev[tid * E + r] : ~0ull;
    __syncthreads();
    for (int size = 2; size <= n; size <<= 1) {
        for (int strd = size >> 1; strd > 0; strd >>= 1) {
            if (strd >= 64 * E) {
                if (active) {
#pragma unroll
                    for (int r = 0; r < E; r++) ev[r * T + tid] = x[r];
                }
                __syncthreads();
                const int m = strd / E;
                const bool take_min = ((tid & m) == 0) == (((tid * E) & size) == 0);
                if (active) {
#pragma unroll
                    for (int r = 0; r < E; r++) {
                        const uint64_t o = ev[r * T + (tid ^ m)];
                        x[r] = take_min ? (o < x[r] ? o : x[r]) : (o > x[r] ? o : x[r]);
                    }
                }
                __syncthreads();
            } else if (strd >= E) {
                const int m = strd / E;   // partner lane = lane ^ m, m in 1..32
                const bool take_min = ((tid & m) == 0) == (((tid * E) & size) == 0);
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const uint64_t o = __shfl_xor((unsigned long long)x[r], m);
                    x[r] = take_min ? (o < x[r] ? o : x[r]) : (o > x[r] ? o : x[r]);
                }
            } else {
                switch (strd) {
                    case 1: hosp_sort_reg<E, 1>(x, size, tid); break;
                    case 2: hosp_sort_reg<E, 2>(x, size, tid); break;
                    case 4: hosp_sort_reg<E, 4>(x, size, tid); break;
                    default: hosp_sort_reg<E, 8>(x, size, tid); break;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < E; r++)
        if (tid * E + r < n) ev[tid * E + r] = x[r];
    __syncthreads();
}

// Larger event sets: one counting pass over the top 11 priority bits (2048 buckets, two per thread)
// scatters the keys through a global scratch array, then every thread insertion-sorts its own two
// adjacent buckets (16 keys on average for 16384 events) -- O(n) instead of a single-CU O(n log^2 n)
// network.  `cnt` / `cur` are 2048 ints each (the SatFn arrays, not yet in use at this point).
__device__ __forceinline__ void hosp_bucket_sort(uint64_t *ev, int R, int tid, uint64_t *scratch, int *cnt, int *cur,
                                                 int (*s_wsum)[HOSP_THREADS / 64]) {
    cnt[2 * tid] = 0;
    cnt[2 * tid + 1] = 0;
    __syncthreads();
    for (int k = tid; k < R; k += HOSP_THREADS) atomicAdd(&cnt[(int)(ev[k] >> 43) & 2047], 1);
    __syncthreads();
    const int c0 = cnt[2 * tid], c1 = cnt[2 * tid + 1], sum = c0 + c1;
    const int lane = tid & 63, wv = tid >> 6;
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wsum[0][wv] = inc;
    __syncthreads();
    int base0 = inc - sum;
    for (int w2 = 0; w2 < wv; w2++) base0 += s_wsum[0][w2];
    cur[2 * tid] = base0;
    cur[2 * tid + 1] = base0 + c0;
    __syncthreads();
    for (int k = tid; k < R; k += HOSP_THREADS) {
        const uint64_t e = ev[k];
        scratch[atomicAdd(&cur[(int)(e >> 43) & 2047], 1)] = e;
    }
    __syncthreads();
    for (int k = tid; k < R; k += HOSP_THREADS) ev[k] = ld_claim(&scratch[k]);
    __syncthreads();
    for (int i = base0 + 1; i < base0 + sum; i++) {
        const uint64_t key = ev[i];
        int j = i - 1;
        while (j >= base0 && ev[j] > key) {
            ev[j + 1] = ev[j];
            j--;
        }
        ev[j + 1] = key;
    }
    __syncthreads();
}

#ifdef REINA_HOSP_STAMPS
// diagnostic: per-phase time of the event walk accumulated in buffers.mirror[0..7] (100 MHz ticks)
#define HSTAMP(k) do { if (threadIdx.x == 0) { uint64_t t_ = wall_clock64(); atomicAdd((unsigned long long *)&B.mirror[k], (unsigned long long)(t_ - hs_t)); hs_t = t_; } } while (0)
#else
#define HSTAMP(k) do { } while (0)
#endif
__device__ __forceinline__ void hospital_block(const MemberRef *M_, const reina_day_t &dp,
                                               uint32_t scan_waves, uint32_t scan_tiles) {
    const MemberRef &mref_ = M_[blockIdx.y];
    const DevParams *P = mref_.P;
    const reina_buffers_t B = mref_.B;  // by value: pointers live in SGPRs, never re-read after stores
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *ev = reinterpret_cast<uint64_t *>(smem);               // [M2]
#ifdef REINA_HOSP_STAMPS
    uint64_t hs_t = wall_clock64();
#endif
    __shared__ int s_b, s_c;
    __shared__ SatFn s_fb[HOSP_THREADS], s_fc[HOSP_THREADS];
    __shared__ int32_t s_cnt[HL_NR][REINA_MAX_AGES];
    __shared__ int32_t s_age_start[REINA_MAX_AGES + 1];
    const int tid = threadIdx.x;
    __shared__ int s_wsum[4][HOSP_THREADS / 64];
    __shared__ int s_tot[4];
    const uint2 *l_ev = reinterpret_cast<const uint2 *>(B.scan_lists);
    // pass 1: count this thread's events by type (its <= 8 scan slices)
    const uint32_t per_w = (scan_waves + HOSP_THREADS - 1) / HOSP_THREADS;  // <= 8
    uint32_t cnts[8];
    int mine[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        const uint32_t sw = tid * per_w + k;
        cnts[k] = (k < per_w && sw < scan_waves) ? B.work_counts[LIST_EV * REINA_MAX_SCAN_WAVES + sw] : 0u;
    }
    // the first record of every non-empty slice is requested at once (most slices hold 0 or 1 events);
    // both passes below use these registers and only go back to memory for a slice's later records
    // (registers for 4 slices: all of them up to 4096 scanning waves, i.e. populations up to 2 M agents)
    uint2 first[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        first[k] = make_uint2(0, 0);
        if (cnts[k]) first[k] = l_ev[scan_slice_base(tid * per_w + k, scan_waves, scan_tiles)];
    }
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        if (cnts[k] == 0) continue;
        if (k < 4) {
            const uint32_t ty = first[k].y & 3u;
            mine[0] += ty == 0;
            mine[1] += ty == 1;
            mine[2] += ty == 2;
            mine[3] += ty == 3;
            if (cnts[k] == 1) continue;
        }
        const uint32_t base = scan_slice_base(tid * per_w + k, scan_waves, scan_tiles);
        for (uint32_t j = k < 4 ? 1 : 0; j < cnts[k]; j++) {
            const uint32_t ty = l_ev[base + j].y & 3u;
            mine[0] += ty == 0;
            mine[1] += ty == 1;
            mine[2] += ty == 2;
            mine[3] += ty == 3;
        }
    }
    // exclusive prefix per type over the workgroup (wave shuffles + 16 wave totals)
    int excl[4];
    {
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int ty = 0; ty < 4; ty++) {
            int inc = mine[ty];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                int o = __shfl_up(inc, off);
                if (lane >= off) inc += o;
            }
            if (lane == 63) s_wsum[ty][wv] = inc;
            excl[ty] = inc - mine[ty];
        }
        __syncthreads();
#pragma unroll
        for (int ty = 0; ty < 4; ty++) {
            int before = 0, total = 0;
            for (int w2 = 0; w2 < HOSP_THREADS / 64; w2++) {
                int t2 = s_wsum[ty][w2];
                if (w2 < wv) before += t2;
                total += t2;
            }
            excl[ty] += before;
            if (tid == 0) s_tot[ty] = total;
        }
        __syncthreads();
    HSTAMP(1);
    }
    const int nH = s_tot[EV_HOSPITALIZE], nT = s_tot[EV_TO_ICU], nW = s_tot[EV_RELEASE_WARD], nI = s_tot[EV_RELEASE_ICU];
    int M = nH + nT + nW + nI;
    if (M > REINA_MAX_HOSP_EVENTS) {
        if (tid == 0) set_problem(B.counters, REINA_PROBLEM_HOSPITAL_OVERFLOW);
        return;
    }
    if (M == 0) return;
    const int b0 = B.counters[SC_IDX(REINA_S_AVAILABLE_BEDS)], c0 = B.counters[SC_IDX(REINA_S_AVAILABLE_ICU)];
    // Only events that touch a resource that can run out today need their order: beds are touched
    // by HOSPITALIZE / TO_ICU / RELEASE_WARD, ICU units by TO_ICU / RELEASE_ICU.  Events are laid out
    // by type so that the order-relevant ones form a prefix [0, R) of the LDS array.
    const bool beds_bind = b0 < nH, icu_bind = c0 < nT;
    const bool ordered = beds_bind || icu_bind;
    int off_ty[4];
    int R;
    if (icu_bind && !beds_bind) {        // [T][I][H][W]
        off_ty[EV_TO_ICU] = 0; off_ty[EV_RELEASE_ICU] = nT; off_ty[EV_HOSPITALIZE] = nT + nI; off_ty[EV_RELEASE_WARD] = nT + nI + nH;
        R = nT + nI;
    } else {                             // [T][H][W][I]
        off_ty[EV_TO_ICU] = 0; off_ty[EV_HOSPITALIZE] = nT; off_ty[EV_RELEASE_WARD] = nT + nH; off_ty[EV_RELEASE_ICU] = nT + nH + nW;
        R = (beds_bind && !icu_bind) ? nT + nH + nW : M;
    }
    if (!ordered) R = 0;
    int M2 = 1;  // bitonic size: the relevant prefix padded to a power of two
    while (M2 < R) M2 <<= 1;
    for (int k = tid; k < HL_NR * REINA_MAX_AGES; k += HOSP_THREADS) (&s_cnt[0][0])[k] = 0;
    if (tid <= REINA_MAX_AGES) s_age_start[tid] = P->age_start[tid];
    {   // pass 2: place. event word = [priority:20][agent:32][type:2]; priority = Philox(agent, day)
        int pos[4] = {off_ty[0] + excl[0], off_ty[1] + excl[1], off_ty[2] + excl[2], off_ty[3] + excl[3]};
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            if (cnts[k] == 0) continue;
            const uint32_t base = scan_slice_base(tid * per_w + k, scan_waves, scan_tiles);
            for (uint32_t j = 0; j < cnts[k]; j++) {
                const uint2 r = (k < 4 && j == 0) ? first[k] : l_ev[base + j];
                const uint64_t prio = rp_priority20(P->k0, P->k1, r.x, dp.day);
                const uint32_t ty = r.y & 3u;
                const int p2 = ty == 0 ? pos[0]++ : ty == 1 ? pos[1]++ : ty == 2 ? pos[2]++ : pos[3]++;
                ev[p2] = (prio << 34) | ((uint64_t)r.x << 2) | (uint64_t)ty;
            }
        }
    }
    __syncthreads();
    HSTAMP(2);
    if (ordered) {
        if (M2 <= HOSP_THREADS) {
            // sort the prefix [0, M2): entries past R (other types, or past M) compare as +infinity
            for (int k = tid; k < M2; k += HOSP_THREADS)
                if (k >= R && k < M) ev[k] |= 1ull << 62;
            for (int k = M + tid; k < M2; k += HOSP_THREADS) ev[k] = ~0ull;
            __syncthreads();
            hosp_sort<1>(ev, M2, tid);
            for (int k = tid; k < M2 && k < M; k += HOSP_THREADS) ev[k] &= ~(1ull << 62);
            __syncthreads();
        } else {
            hosp_bucket_sort(ev, R, tid, B.hosp_events, reinterpret_cast<int *>(s_fb), reinterpret_cast<int *>(s_fc), s_wsum);
        }
    HSTAMP(3);
        // chunked scan: thread t owns events [t*per, (t+1)*per)
        const int per = (R + HOSP_THREADS - 1) / HOSP_THREADS;
        const int lo = min(R, tid * per), hi = min(R, lo + per);
        SatFn fb, fc;
        fb.a = fc.a = 0;
        fb.m = fc.m = SAT_NEG;
        for (int k = lo; k < hi; k++) {
            int type = (int)(ev[k] & 3);
            fb = sat_then(fb, bed_fn(type));
            fc = sat_then(fc, icu_fn(type));
        }
        s_fb[tid] = fb;
        s_fc[tid] = fc;
        __syncthreads();
        // inclusive Hillis-Steele scan of the per-thread maps (composition is associative)
        for (int off = 1; off < HOSP_THREADS; off <<= 1) {
            SatFn pb = fb, pc = fc;
            if (tid >= off) {
                pb = sat_then(s_fb[tid - off], fb);
                pc = sat_then(s_fc[tid - off], fc);
            }
            __syncthreads();
            fb = pb;
            fc = pc;
            s_fb[tid] = fb;
            s_fc[tid] = fc;
            __syncthreads();
        }
        int b = b0, c = c0;
        if (tid > 0) {
            b = sat_apply(s_fb[tid - 1], b0);
            c = sat_apply(s_fc[tid - 1], c0);
        }
        for (int k = lo; k < hi; k++) {
            int type = (int)(ev[k] & 3);
            uint8_t ok = 1;
            if (type == EV_HOSPITALIZE) {
                if (b == 0) ok = 0; else b--;
            } else if (type == EV_TO_ICU) {
                b++;
                if (c == 0) ok = 0; else c--;
            } else if (type == EV_RELEASE_WARD) {
                b++;
            } else {
                c++;
            }
            if (!ok) ev[k] |= 1ull << 63;
        }
        if (tid == HOSP_THREADS - 1) {
            int fb_ = sat_apply(s_fb[HOSP_THREADS - 1], b0), fc_ = sat_apply(s_fc[HOSP_THREADS - 1], c0);
            // events outside the ordered prefix touch a resource that cannot run out: plain sums
            if (icu_bind && !beds_bind) fb_ += nW - nH;
            if (beds_bind && !icu_bind) fc_ += nI;
            s_b = fb_;
            s_c = fc_;
        }
    } else {
        if (tid == 0) {
            s_b = b0;
            s_c = c0;
        }
        __syncthreads();
        int db = 0, dc = 0;
        for (int k = tid; k < M; k += HOSP_THREADS) {
            int type = (int)(ev[k] & 3);
            if (type == EV_HOSPITALIZE) db--;
            else if (type == EV_TO_ICU) { db++; dc--; }
            else if (type == EV_RELEASE_WARD) db++;
            else dc++;
        }
        if (db) atomicAdd(&s_b, db);
        if (dc) atomicAdd(&s_c, dc);
    }
    __syncthreads();
    HSTAMP(4);
    const reina_disease_t &d = P->dis;
    for (int k = tid; k < M; k += HOSP_THREADS) {
        uint64_t e = ev[k];
        int type = (int)(e & 3);
        const bool granted = !(e >> 63);
        uint32_t i = (uint32_t)((e >> 2) & 0xFFFFFFFFu);
        uint32_t w = B.hot[i];
        int age = age_of(s_age_start, i, 0, (int)P->nr_ages - 1), v = RH_VARIANT(w), sev = RH_SEV(w);
        float od = B.onset_days[i];
        int died = -1;  // -1 stays in care, 0 recovers, 1 dies
        if (type == EV_HOSPITALIZE) {
            if (!(w & RH_DETECTED)) {
                w |= RH_DETECTED;
                atomicAdd(&s_cnt[HL_DETECTED][age], 1);
                atomicAdd(&s_cnt[HL_ALL_DETECTED][age], 1);
            }
            if (!granted) {
                died = dies_in_hospital(P, i, dp.day, sev, v, 0);
            } else {
                float f;
                if (sev == RV_SEVERE)
                    f = od * (1.0f - d.ratio_of_duration_before_hospitalisation[v]);
                else
                    f = od * d.ratio_of_duration_in_ward[v];
                w = RH_SET_DAYS_LEFT(RH_SET_STATE(w, RS_HOSPITALIZED), clamp_days(B.counters, rp_round_to_int(f)));
                atomicAdd(&s_cnt[HL_HOSPITALIZED][age], 1);
                atomicAdd(&s_cnt[HL_IN_WARD][age], 1);
            }
        } else if (type == EV_TO_ICU) {
            if (!granted && dies_in_hospital(P, i, dp.day, sev, v, 0)) {
                atomicAdd(&s_cnt[HL_IN_WARD][age], -1);
                atomicAdd(&s_cnt[HL_HOSPITALIZED][age], -1);
                died = 1;
            } else {
                float f = 1.0f - d.ratio_of_duration_in_ward[v] - d.ratio_of_duration_before_hospitalisation[v];
                f *= od;
                w = RH_SET_DAYS_LEFT(RH_SET_STATE(w, RS_IN_ICU), clamp_days(B.counters, rp_round_to_int(f)));
                atomicAdd(&s_cnt[HL_IN_WARD][age], -1);
                atomicAdd(&s_cnt[HL_IN_ICU][age], 1);
                atomicAdd(&s_cnt[HL_CUM_ICU][age], 1);
            }
        } else if (type == EV_RELEASE_WARD) {
            atomicAdd(&s_cnt[HL_IN_WARD][age], -1);
            atomicAdd(&s_cnt[HL_HOSPITALIZED][age], -1);
            died = dies_in_hospital(P, i, dp.day, sev, v, 1);
        } else {
            atomicAdd(&s_cnt[HL_IN_ICU][age], -1);
            atomicAdd(&s_cnt[HL_HOSPITALIZED][age], -1);
            died = dies_in_hospital(P, i, dp.day, sev, v, 1);
        }
        if (died == 1) {
            atomicAdd(&s_cnt[HL_INFECTED][age], -1);
            atomicAdd(&s_cnt[HL_DEAD][age], 1);
            if (w & RH_POD_OUTSIDE) atomicAdd(&s_cnt[HL_NHD][age], 1);
            w = RH_SET_STATE(w, RS_DEAD) & ~RH_HASLIST;
        } else if (died == 0) {
            atomicAdd(&s_cnt[HL_INFECTED][age], -1);
            atomicAdd(&s_cnt[HL_RECOVERED][age], 1);
            w = RH_SET_STATE(w, RS_RECOVERED) & ~RH_HASLIST;
        }
        B.hot[i] = w;
    }
    __syncthreads();
    HSTAMP(5);
    for (int k = tid; k < HL_NR * REINA_MAX_AGES; k += HOSP_THREADS) {
        int32_t v = (&s_cnt[0][0])[k];
        if (v) {
            const int map[HL_NR] = {REINA_C_INFECTED, REINA_C_DETECTED, REINA_C_ALL_DETECTED, REINA_C_HOSPITALIZED,
                                    REINA_C_IN_WARD, REINA_C_IN_ICU, REINA_C_CUM_ICU, REINA_C_DEAD,
                                    REINA_C_NON_HOSPITAL_DEATHS, REINA_C_RECOVERED};
            atomicAdd(&B.counters[CNT_IDX(map[k / REINA_MAX_AGES], k % REINA_MAX_AGES)], v);
        }
    }
    if (tid == 0) {
        B.counters[SC_IDX(REINA_S_AVAILABLE_BEDS)] = s_b;
        B.counters[SC_IDX(REINA_S_AVAILABLE_ICU)] = s_c;
    }
    HSTAMP(6);
}

// ---------------------------------------------------------------------------------------------
// k_initial_state: Population.set_initial_state (main.pyx:1452-1516), parallel form (see
// include/reina_hip.h: reina_set_initial_state and oracle/reina_par.c: par_set_initial_state).
// One workgroup; slots in chunks of PRO_MAX_IMPORTS; per chunk the propose / claim / resolve
// rounds of the import placement, then the slot's fate applied by the winning lane.
__global__ __launch_bounds__(PRO_THREADS) void k_initial_state(const MemberRef *M_, reina_initial_state_t ic) {
    const MemberRef &mref_ = M_[blockIdx.y];
    const DevParams *P = mref_.P;
    const reina_buffers_t B = mref_.B;
    __shared__ uint8_t placed[PRO_MAX_IMPORTS];
    __shared__ int32_t s_cnt[HL_NR][REINA_MAX_AGES];
    __shared__ int32_t new_by_age[REINA_MAX_AGES];
    __shared__ int32_t new_by_variant[REINA_MAX_VARIANTS];
    __shared__ int32_t s_age_start[REINA_MAX_AGES + 1];
    __shared__ int32_t s_beds_used, s_icu_used, s_unplaced;
    const int tid = threadIdx.x;
    if (tid <= REINA_MAX_AGES) s_age_start[tid] = P->age_start[tid];
    if (tid < REINA_MAX_AGES) new_by_age[tid] = 0;
    if (tid < REINA_MAX_VARIANTS) new_by_variant[tid] = 0;
    for (int k = tid; k < HL_NR * REINA_MAX_AGES; k += PRO_THREADS) (&s_cnt[0][0])[k] = 0;
    if (tid == 0) s_beds_used = s_icu_used = s_unplaced = 0;
    __syncthreads();
    const reina_disease_t &d = P->dis;
    const uint32_t N = P->n_agents, M = ic.were_incubating;
    const uint32_t i_inc = ic.incubating, i_rec = i_inc + ic.recovered_without_illness, i_ill = i_rec + ic.ill,
                   i_dead = i_ill + ic.dead, i_icu = i_dead + ic.in_icu, i_ward = i_icu + ic.in_ward;
    const int beds0 = B.counters[SC_IDX(REINA_S_AVAILABLE_BEDS)], icu0 = B.counters[SC_IDX(REINA_S_AVAILABLE_ICU)];
    for (uint32_t c0 = 0; c0 < M; c0 += PRO_MAX_IMPORTS) {
        const uint32_t total = M - c0 < PRO_MAX_IMPORTS ? M - c0 : PRO_MAX_IMPORTS;
        for (uint32_t j = tid; j < total; j += PRO_THREADS) placed[j] = 0;
        __syncthreads();
        for (uint32_t round = 0; round < 10; round++) {
            int proposals = 0;
            for (uint32_t j = tid; j < total; j += PRO_THREADS) {
                if (placed[j] == 255) continue;
                uint32_t k = placed[j], t = 0;
                bool found = false;
                for (; k < 10; k++) {
                    t = rp_philox(P->k0, P->k1, c0 + j, RP_INIT_DAY, RP_P_INITIAL, k).v[0] % N;
                    if (RH_STATE(ld_hot(&B.hot[t])) == RS_SUSCEPTIBLE) {
                        found = true;
                        break;
                    }
                }
                if (found) {
                    atomicMin((unsigned long long *)&B.claim[t], (unsigned long long)rp_order_key(0, 0xFFFFFu - round, c0 + j));
                    placed[j] = (uint8_t)(0x80u | k);
                    proposals++;
                } else {
                    placed[j] = 10;
                }
            }
            if (__syncthreads_or(proposals) == 0) break;
            for (uint32_t j = tid; j < total; j += PRO_THREADS) {
                const uint8_t st = placed[j];
                if (st == 255 || !(st & 0x80u)) continue;
                const uint32_t k = st & 0x7Fu, slot = c0 + j;
                const uint32_t t = rp_philox(P->k0, P->k1, slot, RP_INIT_DAY, RP_P_INITIAL, k).v[0] % N;
                placed[j] = (uint8_t)(k + 1);
                if (ld_claim(&B.claim[t]) != rp_order_key(0, 0xFFFFFu - round, slot)) continue;
                placed[j] = 255;
                uint32_t w = ld_hot(&B.hot[t]);
                if (!install_infection(P, B, s_age_start, t, w, RP_INIT_DAY, 0, -1, slot < i_inc, RT_NO_TESTING, new_by_age, new_by_variant))
                    continue;
                if (slot < i_inc) continue;
                w = ld_hot(&B.hot[t]);
                const int age = age_of(s_age_start, t, 0, (int)P->nr_ages - 1);
                int died = -1;  // -1 keeps the state set below, 0 recovers, 1 dies
                if (slot < i_rec) {
                    died = 0;
                } else {
                    w = onset_word(P, B, t, w, RP_INIT_DAY);
                    const int v = RH_VARIANT(w), sev = RH_SEV(w);
                    const float od = B.onset_days[t];
                    if (slot < i_ill) {
                    } else if (slot < i_dead) {
                        died = 1;
                    } else if (slot < i_ward) {
                        const bool to_icu = slot < i_icu;
                        w |= RH_DETECTED;
                        atomicAdd(&s_cnt[HL_DETECTED][age], 1);
                        atomicAdd(&s_cnt[HL_ALL_DETECTED][age], 1);
                        const bool bed = to_icu ? beds0 > 0 : (int)(slot - i_icu) < beds0;
                        if (!bed) {
                            died = dies_in_hospital(P, t, RP_INIT_DAY, sev, v, 0);
                        } else if (!to_icu) {
                            atomicAdd(&s_beds_used, 1);
                            float f = sev == RV_SEVERE ? od * (1.0f - d.ratio_of_duration_before_hospitalisation[v])
                                                       : od * d.ratio_of_duration_in_ward[v];
                            w = RH_SET_DAYS_LEFT(RH_SET_STATE(w, RS_HOSPITALIZED), clamp_days(B.counters, rp_round_to_int(f)));
                            atomicAdd(&s_cnt[HL_HOSPITALIZED][age], 1);
                            atomicAdd(&s_cnt[HL_IN_WARD][age], 1);
                        } else {
                            const bool unit = (int)(slot - i_dead) < icu0;
                            if (unit) atomicAdd(&s_icu_used, 1);
                            if (!unit && dies_in_hospital(P, t, RP_INIT_DAY, sev, v, 0)) {
                                died = 1;
                            } else {
                                float f = 1.0f - d.ratio_of_duration_in_ward[v] - d.ratio_of_duration_before_hospitalisation[v];
                                f *= od;
                                w = RH_SET_DAYS_LEFT(RH_SET_STATE(w, RS_IN_ICU), clamp_days(B.counters, rp_round_to_int(f)));
                                atomicAdd(&s_cnt[HL_HOSPITALIZED][age], 1);
                                atomicAdd(&s_cnt[HL_IN_ICU][age], 1);
                                atomicAdd(&s_cnt[HL_CUM_ICU][age], 1);
                            }
                        }
                    } else {
                        died = 0;
                    }
                }
                if (died == 1) {
                    atomicAdd(&s_cnt[HL_INFECTED][age], -1);
                    atomicAdd(&s_cnt[HL_DEAD][age], 1);
                    if (w & RH_POD_OUTSIDE) atomicAdd(&s_cnt[HL_NHD][age], 1);
                    w = RH_SET_STATE(w, RS_DEAD) & ~RH_HASLIST;
                } else if (died == 0) {
                    atomicAdd(&s_cnt[HL_INFECTED][age], -1);
                    atomicAdd(&s_cnt[HL_RECOVERED][age], 1);
                    w = RH_SET_STATE(w, RS_RECOVERED) & ~RH_HASLIST;
                }
                B.hot[t] = w;
            }
            __syncthreads();
        }
        int mine = 0;
        for (uint32_t j = tid; j < total; j += PRO_THREADS)
            if (placed[j] != 255) mine++;
        if (mine) atomicAdd(&s_unplaced, mine);
        __syncthreads();
    }
    flush_new_infections(B, new_by_age, new_by_variant, PRO_THREADS);
    // per-age counters; all_detected[0..99] restarts from the confirmed cases (main.pyx:1503-1516)
    const uint32_t stride = ic.confirmed_stride ? ic.confirmed_stride : 1u;
    for (int k = tid; k < HL_NR * REINA_MAX_AGES; k += PRO_THREADS) {
        const int what = k / REINA_MAX_AGES, age = k % REINA_MAX_AGES;
        const int map[HL_NR] = {REINA_C_INFECTED, REINA_C_DETECTED, REINA_C_ALL_DETECTED, REINA_C_HOSPITALIZED,
                                REINA_C_IN_WARD, REINA_C_IN_ICU, REINA_C_CUM_ICU, REINA_C_DEAD,
                                REINA_C_NON_HOSPITAL_DEATHS, REINA_C_RECOVERED};
        int32_t v = (&s_cnt[0][0])[k];
        if (what == HL_ALL_DETECTED && age < 100 && age < (int)P->nr_ages) {
            // confirmed cases first, first+stride, ... < confirmed_cases with index % 100 == age
            int32_t n = 0;
            for (uint32_t i = ic.confirmed_first; i < ic.confirmed_cases; i += stride)
                if ((int)(i % 100u) == age) n++;
            B.counters[CNT_IDX(REINA_C_ALL_DETECTED, age)] = n;
        } else if (v) {
            B.counters[CNT_IDX(map[what], age)] += v;
        }
    }
    if (tid == 0) {
        B.counters[SC_IDX(REINA_S_AVAILABLE_BEDS)] = beds0 - s_beds_used;
        B.counters[SC_IDX(REINA_S_AVAILABLE_ICU)] = icu0 - s_icu_used;
        if (s_unplaced) B.counters[SC_IDX(REINA_S_UNABLE_TO_IMPORT)] += s_unplaced;
    }
}

// ---------------------------------------------------------------------------------------------
// k_contacts: get_one_contact / get_person_from_age_range / person_expose / did_infect
// (main.pyx:1290-1304,1525-1535,238-244,908-934), one lane per sampled contact.
// Each wave takes 64 work items, prefix-sums their contact counts across lanes and spreads the
// contacts evenly over its lanes.  Contact tables are staged in LDS once per workgroup.
#define CON_THREADS 1024
#define CON_WAVES (CON_THREADS / 64)
#define CAND_CHUNK 128

struct ConShared {
    uint32_t meta_row[REINA_MAX_ENTRIES];   // shared (place, range) pattern when every age has the same
    float mask_p[REINA_MAX_AGES][8];
    float p_sus[REINA_MAX_VARIANTS][REINA_MAX_AGES];
    int32_t age_start[REINA_MAX_AGES + 1];
    int32_t tcount[REINA_MAX_AGES];
    uint32_t pre[CON_WAVES][64];        // inclusive prefix of contact counts per wave batch
    uint4 item[CON_WAVES][64];          // the wave's current 64 work items
    int32_t daily[REINA_NR_PLACES];
    int32_t n_contacts;
    float iot[REINA_MAX_VARIANTS][REINA_IOT_LEN + 3];
    float nrc[REINA_MAX_AGES];
    // dynamic tail: uint32_t thr[nr_ages][REINA_MAX_ENTRIES]; then, when sharded,
    // int32_t pressure[REINA_PRESSURE_WORDS] (this workgroup's outgoing cross-shard pressure)
};
static size_t con_shared_bytes(uint32_t nr_ages, uint32_t n_shards) {
    return sizeof(ConShared) + (size_t)nr_ages * REINA_MAX_ENTRIES * 4 + (n_shards > 1 ? REINA_PRESSURE_WORDS * 4 : 0);
}

__device__ __forceinline__ void contacts_block(const MemberRef *M_, const reina_day_t &dp, uint32_t scan_waves,
                                               uint32_t scan_tiles, int uniform_meta, uint32_t bx, uint32_t nbx) {
    const MemberRef &mref_ = M_[blockIdx.y];
    const DevParams *P = mref_.P;
    const reina_buffers_t B = mref_.B;  // by value: pointers live in SGPRs, never re-read after stores
    const Tables *T = mref_.T;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    ConShared &S = *reinterpret_cast<ConShared *>(smem_raw);
    uint32_t (*S_thr)[REINA_MAX_ENTRIES] = reinterpret_cast<uint32_t (*)[REINA_MAX_ENTRIES]>(smem_raw + sizeof(ConShared));
    int32_t *S_pressure = reinterpret_cast<int32_t *>(smem_raw + sizeof(ConShared) + (size_t)P->nr_ages * REINA_MAX_ENTRIES * 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (bx * CON_WAVES >= scan_waves) return;  // no slice for this workgroup
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(&T->thr[0][0]);
        uint4 *dst = reinterpret_cast<uint4 *>(&S_thr[0][0]);
        const int n16 = (int)(P->nr_ages * REINA_MAX_ENTRIES * 4 / 16);
        for (int k = tid; k < n16; k += CON_THREADS) dst[k] = src[k];
        if (tid < REINA_MAX_ENTRIES) S.meta_row[tid] = T->meta[0][tid];
        for (int k = tid; k < REINA_MAX_AGES * 8; k += CON_THREADS) (&S.mask_p[0][0])[k] = (&P->mask_p[0][0])[k];
        for (int k = tid; k < REINA_MAX_VARIANTS * REINA_MAX_AGES; k += CON_THREADS) (&S.p_sus[0][0])[k] = (&P->dis.p_susceptibility[0][0])[k];
        for (int k = tid; k <= REINA_MAX_AGES; k += CON_THREADS) S.age_start[k] = P->age_start[k];
        for (int k = tid; k < REINA_MAX_AGES; k += CON_THREADS) S.tcount[k] = P->tcount[k];
        if (tid < REINA_NR_PLACES) S.daily[tid] = 0;
        if (tid == 0) S.n_contacts = 0;
        for (int k = tid; k < REINA_MAX_VARIANTS * (REINA_IOT_LEN + 3); k += CON_THREADS)
            (&S.iot[0][0])[k] = (&P->dis.infectiousness_over_time[0][0])[k];
        for (int k = tid; k < REINA_MAX_AGES; k += CON_THREADS) S.nrc[k] = P->nrc[k];
        if (P->n_shards > 1)
            for (int k = tid; k < REINA_PRESSURE_WORDS; k += CON_THREADS) S_pressure[k] = 0;
    }
    __syncthreads();
    const reina_disease_t &d = P->dis;
    const uint32_t n_shards = P->n_shards, shard_rank = P->shard_rank;
    const uint2 *items = reinterpret_cast<const uint2 *>(B.work_items);
    const uint32_t total_waves = nbx * CON_WAVES;
    uint32_t wave_contacts = 0;
    for (uint32_t sw = bx * CON_WAVES + wave; sw < scan_waves; sw += total_waves) {
        const uint32_t slice_base = scan_slice_base(sw, scan_waves, scan_tiles);
        // successful attempts of this slice's sources go to the slice's own candidate region
        const uint32_t slice_cap = (sw + 1 < scan_waves ? scan_slice_base(sw + 1, scan_waves, scan_tiles) : P->max_work_items) - slice_base;
        uint32_t n_cand = 0;
        const uint32_t W = B.work_counts[LIST_EXP * REINA_MAX_SCAN_WAVES + sw];
        for (uint32_t b0 = 0; b0 < W; b0 += 64) {
            const uint32_t idx = b0 + lane;
            uint4 it = make_uint4(0, 0, 0, 0);
            uint32_t nr = 0;
            if (idx < W) {
                // person_expose_others -> get_exposed_people -> get_nr_contacts (main.pyx:247-281,
                // 936-955,1308-1320): the scan recorded (agent, start-of-day word) of every
                // infectious, undetected agent; draw its contact COUNT here, 64 agents per wave
                const uint2 ex = items[slice_base + idx];
                const uint32_t i = ex.x, w = ex.y;
                const uint32_t st = RH_STATE(w);
                const int v = RH_VARIANT(w), sev = RH_SEV(w);
                const int dayrel = st == RS_INCUBATION ? -(int)RH_DAYS_LEFT(w) : (int)RH_DOI(w);
                const float inf = S.iot[v][dayrel + 10];
                const int age = age_of(S.age_start, i, 0, (int)P->nr_ages - 1);
                float factor = 1.0f;
                int limit = 100;
                if (st == RS_ILLNESS && sev != RV_ASYMPTOMATIC) {
                    factor = 0.5f;
                    limit = 5;
                }
                float z = rp_normal_from_u32(rp_philox(P->k0, P->k1, i, dp.day, RP_P_NRCONTACTS, 0).v[0]);
                float f = rp_expf(0.5f * z) * S.nrc[age];
                f *= factor;
                if (f < 1.0f) f = 1.0f;
                int n = (int)f - 1;
                if (n > limit) n = limit;
                nr = (uint32_t)n;
                float src_inf = inf;
                if (sev == RV_ASYMPTOMATIC) src_inf *= d.p_asymptomatic_infection[v];
                it = make_uint4(i, nr | ((uint32_t)v << 8) | ((uint32_t)age << 16), rp_f2u(src_inf), 0u);
            }
            // inclusive prefix sum of nr across the wave
            uint32_t inc = nr;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t o = __shfl_up(inc, off);
                if (lane >= off) inc += o;
            }
            S.pre[wave][lane] = inc;
            S.item[wave][lane] = it;
            const uint32_t total = __shfl(inc, 63);
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): wave-private LDS rows written
            __builtin_amdgcn_wave_barrier();
            wave_contacts += total;
            for (uint32_t j0 = 0; j0 < total; j0 += 64) {
                const uint32_t j = j0 + lane;
                const bool act = j < total;
                bool hit = false;
                uint4 cand = make_uint4(0xFFFFFFFFu, 0, 0, 0);
                int place = -1;
                if (act) {
                    // owner item: first lane whose inclusive prefix exceeds j
                    int lo = 0, hi = 63;
                    while (lo < hi) {
                        int mid = (lo + hi) >> 1;
                        if (S.pre[wave][mid] > j) hi = mid; else lo = mid + 1;
                    }
                    const int owner = lo;
                    const uint32_t c = j - (owner ? S.pre[wave][owner - 1] : 0u);
                    const uint4 own = S.item[wave][owner];
                    const uint32_t src = own.x;
                    const float src_inf = rp_u2f(own.z);
                    const int v = (int)((own.y >> 8) & 0xFFu), row = (int)(own.y >> 16);
                    rp_u4 r = rp_philox(P->k0, P->k1, src, dp.day, RP_P_CONTACT, c);
                    // first entry with r0 < threshold (thresholds are non-decreasing); none -> last entry
                    const int cnt = S.tcount[row];
                    int l2 = 0, h2 = cnt - 1;
                    while (l2 < h2) {
                        int mid = (l2 + h2) >> 1;
                        if (r.v[0] < S_thr[row][mid]) h2 = mid; else l2 = mid + 1;
                    }
                    const uint32_t m = uniform_meta ? S.meta_row[l2] : T->meta[row][l2];
                    place = (int)(m & 0xFFu);
                    const int cmin = (int)((m >> 8) & 0xFFu), cmax = (int)((m >> 16) & 0xFFu);
                    const uint32_t start = (uint32_t)S.age_start[cmin], end = (uint32_t)S.age_start[cmax + 1];
                    // uniform member of the range over the WHOLE population: uniform shard, then
                    // uniform agent of that shard (every shard holds 1/G of every age)
                    // (an unsharded population skips the two integer divisions: x % 1 = 0, x / 1 = x)
                    const uint32_t dest = n_shards > 1 ? r.v[1] % n_shards : 0u;
                    if (dest != shard_rank) {
                        // source-side part of did_infect; the destination applies p_sus / psus_max
                        float qv = src_inf * P->psus_max[v] * d.infectiousness_multiplier[v];
                        bool pass = rp_chance(qv, r.v[2]) != 0;
                        if (pass) {
                            const float mp = S.mask_p[row][place];
                            if (mp != 0.0f) {
                                float a = mp * d.p_mask_protects_others[v];
                                float b = mp * d.p_mask_protects_wearer[v];
                                float pm = a + b - a * b;
                                if (rp_chance(pm, r.v[3])) pass = false;
                            }
                        }
                        if (pass) {
                            atomicAdd(&S_pressure[(dest * REINA_MAX_RANGES + (m >> 24)) * REINA_MAX_VARIANTS + (uint32_t)v], 1);
                            // mirror table: smallest (tie-break, src) per slot, tagged with today
                            rp_u4 hm = rp_philox(P->k0, P->k1, src, dp.day, RP_P_MIRROR, c);
                            const uint32_t MS = P->mirror_slots;
                            uint64_t *slot = B.mirror + ((size_t)((m >> 24) * REINA_MAX_VARIANTS + (uint32_t)v)) * MS + (hm.v[0] & (MS - 1));
                            atomicMin((unsigned long long *)slot, (unsigned long long)rp_order_key(dp.day, hm.v[1] >> 12, src));
                        }
                    } else if (end > start) {
                        const uint32_t t = start + (n_shards > 1 ? r.v[1] / n_shards : r.v[1]) % (end - start);
                        // 1 bit per agent: the whole table (N/8 bytes) stays in L2 / Infinity Cache
                        if ((B.sus_bits[t >> 5] >> (t & 31u)) & 1u) {
                            const int age_t = age_of(S.age_start, t, cmin, cmax);
                            float p = src_inf * S.p_sus[v][age_t] * d.infectiousness_multiplier[v];
                            if (rp_chance(p, r.v[2])) {
                                hit = true;
                                const float mp = S.mask_p[row][place];
                                if (mp != 0.0f) {
                                    float a = mp * d.p_mask_protects_others[v];
                                    float b = mp * d.p_mask_protects_wearer[v];
                                    float pm = a + b - a * b;
                                    if (rp_chance(pm, r.v[3])) hit = false;
                                }
                                if (hit) {
                                    const uint32_t prio = rp_priority20(P->k0, P->k1, src, dp.day);
                                    atomicMin((unsigned long long *)&B.claim[t], (unsigned long long)rp_order_key(dp.day, prio, src));
                                    cand = make_uint4(t, src, (uint32_t)v, prio);
                                }
                            }
                        }
                    }
                }
                // daily_contacts[place] (main.pyx:1571): one LDS atomic per place per wave step
#pragma unroll
                for (int pl = 0; pl < REINA_NR_PLACES; pl++) {
                    uint64_t pm_ = __ballot(place == pl);
                    if (pm_ && lane == 0) atomicAdd(&S.daily[pl], (int)__popcll(pm_));
                }
                // candidate records: ballot slots in the slice's region, no atomics
                const uint64_t hm = __ballot(hit);
                if (hm) {
                    if (hit) {
                        const uint32_t pos = n_cand + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
                        if (pos >= slice_cap)
                            set_problem(B.counters, REINA_PROBLEM_CANDIDATE_OVERFLOW);
                        else
                            reinterpret_cast<uint4 *>(B.candidates)[slice_base + pos] = cand;
                    }
                    n_cand += (uint32_t)__popcll(hm);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) B.work_counts[LIST_CAND * REINA_MAX_SCAN_WAVES + sw] = n_cand < slice_cap ? n_cand : slice_cap;
    }
    if (lane == 0 && wave_contacts) atomicAdd(&S.n_contacts, (int)wave_contacts);
    __syncthreads();
    if (tid < REINA_NR_PLACES && S.daily[tid]) atomicAdd(&B.counters[SC_IDX(REINA_S_DAILY_CONTACTS + tid)], S.daily[tid]);
    if (tid == 0 && S.n_contacts) {
        atomicAdd(&B.control[REINA_L_CONTACTS], S.n_contacts);
        atomicAdd(&B.counters[SC_IDX(REINA_S_EXPOSED_PER_DAY)], S.n_contacts);  // = sum of the counts drawn
    }
    if (n_shards > 1)
        for (int k = tid; k < REINA_PRESSURE_WORDS; k += CON_THREADS)
            if (S_pressure[k]) atomicAdd(&B.pressure[k], S_pressure[k]);
}

// ---------------------------------------------------------------------------------------------
// k_hosp_contacts: the day's bed / ICU events and its contact sampling are independent (the events
// rewrite hot words of agents leaving ILLNESS / HOSPITALIZED / IN_ICU and the bed counters; contacts
// read the scan's exposure records, the susceptible bitmap and the claim words), so they share one
// launch: workgroup 0 walks the events while workgroups 1.. sample contacts.  Both are 1024 threads;
// VGPR use already limits either to one workgroup per CU, so the larger LDS request costs nothing.
static_assert(HOSP_THREADS == CON_THREADS, "fused launch");
__global__ __launch_bounds__(CON_THREADS) void k_hosp_contacts(const MemberRef *M_, reina_day_t dp, uint32_t scan_waves,
                                                               uint32_t scan_tiles, int uniform_meta) {
    if (blockIdx.x == 0)
        hospital_block(M_, dp, scan_waves, scan_tiles);
    else
        contacts_block(M_, dp, scan_waves, scan_tiles, uniform_meta, blockIdx.x - 1, gridDim.x - 1);
}

// ---------------------------------------------------------------------------------------------
// k_remote: realise the cross-shard infection pressure aimed at this shard (buffers.pressure has
// been summed over all shards by the caller).  Attempt k of cell (range, variant) picks a uniform
// local agent of the range and applies the target-side part of did_infect, p_sus(age)/psus_max;
// survivors compete for the target exactly like local contacts (atomicMin claim + candidate).
__global__ __launch_bounds__(256) void k_remote(const MemberRef *M_, reina_day_t dp) {
    const MemberRef &mref_ = M_[blockIdx.y];
    const DevParams *P = mref_.P;
    const reina_buffers_t B = mref_.B;  // by value: pointers live in SGPRs, never re-read after stores
    __shared__ int32_t s_age_start[REINA_MAX_AGES + 1];
    __shared__ uint32_t s_pre[REINA_MAX_RANGES * REINA_MAX_VARIANTS + 1];  // exclusive prefix of cell counts
    const int tid = threadIdx.x;
    const uint32_t V = P->nr_variants, cells = P->n_ranges * V;
    if (tid <= REINA_MAX_AGES) s_age_start[tid] = P->age_start[tid];
    if (tid == 0) {
        uint32_t acc = 0;
        for (uint32_t c = 0; c < cells; c++) {
            s_pre[c] = acc;
            uint32_t rg = c / V, v = c % V;
            int n = B.pressure[(P->shard_rank * REINA_MAX_RANGES + rg) * REINA_MAX_VARIANTS + v];
            acc += n > 0 ? (uint32_t)n : 0u;
        }
        s_pre[cells] = acc;
    }
    __syncthreads();
    uint32_t total = s_pre[cells];
    const uint32_t room = P->max_candidates - P->max_work_items;
    if (total > room) {
        if (tid == 0 && blockIdx.x == 0) set_problem(B.counters, REINA_PROBLEM_CANDIDATE_OVERFLOW);
        total = room;
    }
    if (tid == 0 && blockIdx.x == 0) B.control[REINA_L_CAND] = (int)total;  // records in the remote region
    uint4 *rcand = reinterpret_cast<uint4 *>(B.candidates) + P->max_work_items;
    const reina_disease_t &d = P->dis;
    for (uint32_t idx = blockIdx.x * blockDim.x + tid; idx < total; idx += gridDim.x * blockDim.x) {
        rcand[idx] = make_uint4(0xFFFFFFFFu, 0, 0, 0);  // hole unless the attempt succeeds below
        uint32_t lo = 0, hi = cells - 1;  // cell with s_pre[cell] <= idx < s_pre[cell+1]
        while (lo < hi) {
            uint32_t mid = (lo + hi + 1) >> 1;
            if (s_pre[mid] <= idx) lo = mid; else hi = mid - 1;
        }
        const uint32_t rg = lo / V, v = lo % V, k = idx - s_pre[lo];
        const uint32_t start = (uint32_t)s_age_start[P->range_min[rg]], end = (uint32_t)s_age_start[P->range_max[rg] + 1];
        if (end <= start) continue;
        rp_u4 r = rp_philox(P->k0, P->k1, k, dp.day, RP_P_REMOTE, rg | (v << 8));
        const uint32_t t = start + r.v[0] % (end - start);
        if (!((B.sus_bits[t >> 5] >> (t & 31u)) & 1u)) continue;
        const int age_t = age_of(s_age_start, t, P->range_min[rg], P->range_max[rg]);
        const float p = d.p_susceptibility[v][age_t] / P->psus_max[v];
        if (!rp_chance(p, r.v[1])) continue;
        const uint32_t prio = r.v[2] >> 12;
        // mirror attribution: first slot at/after a hashed start that holds an entry of today;
        // own cell first, then the other ranges of the variant, then the other variants
        uint32_t src = RP_REMOTE_SRC | idx;
        {
            const uint32_t MS = P->mirror_slots;
            const uint32_t probes = MS < RP_MIRROR_PROBES ? MS : RP_MIRROR_PROBES;
            bool found = false;
            for (uint32_t dv = 0; dv < V && !found; dv++)
                for (uint32_t dr = 0; dr < P->n_ranges && !found; dr++) {
                    const uint32_t cell = ((rg + dr) % P->n_ranges) * REINA_MAX_VARIANTS + (v + dv) % V;
                    const uint64_t *tab = B.mirror + (size_t)cell * MS;
                    for (uint32_t j = 0; j < probes; j++) {
                        uint64_t ent = tab[(r.v[3] + j) & (MS - 1)];
                        if ((ent >> 52) == ((4095u - dp.day) & 0xFFFu)) {
                            src = (uint32_t)ent;
                            found = true;
                            break;
                        }
                    }
                }
        }
        atomicMin((unsigned long long *)&B.claim[t], (unsigned long long)rp_order_key(dp.day, prio, src));
        rcand[idx] = make_uint4(t, src, v, prio);
    }
}

// ---------------------------------------------------------------------------------------------
// k_install: the attempt whose source holds the smallest (priority, id) key per target wins
// (the reference: first source in rotated scan order, main.pyx:1982-1992) and infects it.
__global__ __launch_bounds__(256) void k_install(const MemberRef *M_, reina_day_t dp,
                                                 uint32_t scan_waves, uint32_t scan_tiles) {
    const MemberRef &mref_ = M_[blockIdx.y];
    const DevParams *P = mref_.P;
    const reina_buffers_t B = mref_.B;  // by value: pointers live in SGPRs, never re-read after stores
    __shared__ int32_t new_by_age[REINA_MAX_AGES];
    __shared__ int32_t new_by_variant[REINA_MAX_VARIANTS];
    __shared__ int32_t s_age_start[REINA_MAX_AGES + 1];
    __shared__ int32_t s_cnt[SL_NR][REINA_MAX_AGES];
    __shared__ int32_t s_infectors, s_infections;
    for (int k = threadIdx.x; k <= REINA_MAX_AGES; k += blockDim.x) s_age_start[k] = P->age_start[k];
    for (int k = threadIdx.x; k < SL_NR * REINA_MAX_AGES; k += blockDim.x) (&s_cnt[0][0])[k] = 0;
    if (threadIdx.x == 0) {
        s_infectors = 0;
        s_infections = 0;
    }
    if (threadIdx.x < REINA_MAX_AGES) new_by_age[threadIdx.x] = 0;
    if (threadIdx.x < REINA_MAX_VARIANTS) new_by_variant[threadIdx.x] = 0;
    __syncthreads();
    // even workgroups install the winning candidates, odd ones walk the scan's deferred lists:
    // two latency-bound jobs side by side instead of one after the other
    const uint32_t half = gridDim.x >> 1;                 // grid is even (host)
    const bool do_cand = (blockIdx.x & 1u) == 0u;
    const uint32_t blk = blockIdx.x >> 1;
    const uint4 *cand = reinterpret_cast<const uint4 *>(B.candidates);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_g = (blk * blockDim.x + threadIdx.x) >> 6, waves_t = (half * blockDim.x) >> 6;
    if (do_cand) {
        // per-slice candidate regions written by k_contacts: 8 slices at a time, 64 records per step
        for (uint32_t sw0 = wave_g * 8u; sw0 < scan_waves; sw0 += waves_t * 8u) {
            uint32_t c_n[8], base[8], tot = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t sw = sw0 + (uint32_t)k;
                const bool in = sw < scan_waves;
                c_n[k] = in ? B.work_counts[LIST_CAND * REINA_MAX_SCAN_WAVES + sw] : 0u;
                base[k] = scan_slice_base(in ? sw : 0u, scan_waves, scan_tiles);
                tot += c_n[k];
            }
            for (uint32_t j0 = 0; j0 < tot; j0 += 64u) {
                uint32_t j = j0 + lane;
                if (j >= tot) continue;
                uint32_t bsel = base[0];
                bool done = false;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (!done) {
                        if (j < c_n[k]) {
                            bsel = base[k];
                            done = true;
                        } else {
                            j -= c_n[k];
                        }
                    }
                }
                const uint4 cd = cand[bsel + j];
                // claim word, target word and source word are requested together
                const uint64_t cl = ld_claim(&B.claim[cd.x]);
                const uint32_t w = ld_hot(&B.hot[cd.x]);
                const uint32_t ws = ld_hot(&B.hot[cd.y]);
                if (cl != rp_order_key(dp.day, cd.w, cd.y)) continue;
                if (RH_STATE(w) != RS_SUSCEPTIBLE) continue;  // duplicate record of the same winner
                install_infection(P, B, s_age_start, cd.x, w, dp.day, cd.z, (int32_t)cd.y, 0, dp.testing_mode, new_by_age, new_by_variant,
                                  true, ws);
            }
        }
        // candidates realised from cross-shard pressure (k_remote), indexed above the slice regions
        const int C = P->n_shards > 1 ? B.control[REINA_L_CAND] : 0;
        const uint4 *rcand = cand + P->max_work_items;
        for (int k = blk * blockDim.x + threadIdx.x; k < C; k += half * blockDim.x) {
            const uint4 cd = rcand[k];
            if (cd.x == 0xFFFFFFFFu) continue;  // attempt that did not get through
            if (B.claim[cd.x] != rp_order_key(dp.day, cd.w, cd.y)) continue;
            uint32_t w = ld_hot(&B.hot[cd.x]);
            if (RH_STATE(w) != RS_SUSCEPTIBLE) continue;
            const int32_t src = (cd.y & RP_REMOTE_SRC) ? -1 : (int32_t)cd.y;
            install_infection(P, B, s_age_start, cd.x, w, dp.day, cd.z, src, 0, dp.testing_mode, new_by_age, new_by_variant);
        }
    }
    // the scan's deferred work.  Each wave of this kernel takes up to 8 scanning-wave slices at a
    // time, loads their counts together and walks the concatenation 64 records at a time, so lanes
    // stay dense even though a single slice holds only a handful of records.
    const uint32_t cap = P->max_work_items;
    const uint2 *l_ill = reinterpret_cast<const uint2 *>(B.work_items) + cap;
    const uint2 *l_book = reinterpret_cast<const uint2 *>(B.scan_lists) + cap;
    for (uint32_t sw0 = do_cand ? scan_waves : wave_g * 8u; sw0 < scan_waves; sw0 += waves_t * 8u) {
        uint32_t c_ill[8], c_bk[8], base[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t sw = sw0 + (uint32_t)k;
            const bool in = sw < scan_waves;
            c_ill[k] = in ? B.work_counts[LIST_ILL * REINA_MAX_SCAN_WAVES + sw] : 0u;
            c_bk[k] = in ? B.work_counts[LIST_BOOK * REINA_MAX_SCAN_WAVES + sw] : 0u;
            base[k] = scan_slice_base(in ? sw : 0u, scan_waves, scan_tiles);
        }
        uint32_t t_ill = 0, t_bk = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            t_ill += c_ill[k];
            t_bk += c_bk[k];
        }
        // symptom onsets (person_become_ill)
        for (uint32_t j0 = 0; j0 < t_ill; j0 += 64u) {
            uint32_t j = j0 + lane;
            if (j < t_ill) {
                uint32_t bsel = base[0];
                bool done = false;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (!done) {
                        if (j < c_ill[k]) {
                            bsel = base[k];
                            done = true;
                        } else {
                            j -= c_ill[k];
                        }
                    }
                }
                const uint2 r = l_ill[bsel + j];
                become_ill(P, B, dp, r.x, r.y);
            }
        }
        // bookkeeping: R statistics (main.pyx:1969-1972) and the per-age counters of agents who
        // recovered or died at home today (Population.recover / die, main.pyx:1584-1623)
        for (uint32_t j0 = 0; j0 < t_bk; j0 += 64u) {
            uint32_t j = j0 + lane;
            int ni = 0, cr = 0;
            if (j < t_bk) {
                uint32_t bsel = base[0];
                bool done = false;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (!done) {
                        if (j < c_bk[k]) {
                            bsel = base[k];
                            done = true;
                        } else {
                            j -= c_bk[k];
                        }
                    }
                }
                const uint2 r = l_book[bsel + j];
                if (r.y == EVX_COUNT_R) {
                    cr = 1;
                    ni = B.n_infected[r.x];
                } else {
                    const int age = age_of(s_age_start, r.x, 0, (int)P->nr_ages - 1);
                    atomicAdd(&s_cnt[SL_INFECTED][age], -1);
                    if (r.y == EVX_RECOVERED_HOME) {
                        atomicAdd(&s_cnt[SL_RECOVERED][age], 1);
                    } else {
                        atomicAdd(&s_cnt[SL_DEAD][age], 1);
                        atomicAdd(&s_cnt[SL_NHD][age], 1);
                    }
                }
            }
            const int tc = wave_sum(cr), tn = wave_sum(ni);
            if (lane == 0 && tc) {
                atomicAdd(&s_infectors, tc);
                if (tn) atomicAdd(&s_infections, tn);
            }
        }
    }
    flush_new_infections(B, new_by_age, new_by_variant, (int)blockDim.x);
    for (int k = threadIdx.x; k < SL_NR * REINA_MAX_AGES; k += blockDim.x) {
        int32_t v = (&s_cnt[0][0])[k];
        if (v) {
            const int map[SL_NR] = {REINA_C_INFECTED, REINA_C_RECOVERED, REINA_C_DEAD, REINA_C_NON_HOSPITAL_DEATHS};
            atomicAdd(&B.counters[CNT_IDX(map[k / REINA_MAX_AGES], k % REINA_MAX_AGES)], v);
        }
    }
    if (threadIdx.x == 0) {
        if (s_infectors) atomicAdd(&B.counters[SC_IDX(REINA_S_TOTAL_INFECTORS)], s_infectors);
        if (s_infections) atomicAdd(&B.counters[SC_IDX(REINA_S_TOTAL_INFECTIONS)], s_infections);
        // today's test queue has been processed (k_test_*): empty it for the day after tomorrow
        if (blockIdx.x == 0) {
            B.control[(dp.day & 1) ? REINA_L_QUEUE1 : REINA_L_QUEUE0] = 0;
            B.control[REINA_L_LEVEL1] = 0;   // consumed by k_test_trace1; tomorrow's level-0 pass appends from its first instruction
        }
    }
}

// table upload: two word copies out of a pinned host buffer (see reina_upload_contact_tables)
__global__ __launch_bounds__(256) void k_upload(uint32_t *d0, const uint32_t *s0, uint32_t n0, uint32_t *d1, const uint32_t *s1, uint32_t n1) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n0; k += stride) d0[k] = s0[k];
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n1; k += stride) d1[k] = s1[k];
}

// ---------------------------------------------------------------------------------------------
// host side

static int grid_for(uint32_t n_items, int threads, int max_blocks) {
    long blocks = ((long)n_items + threads - 1) / threads;
    if (blocks < 1) blocks = 1;
    if (blocks > max_blocks) blocks = max_blocks;
    return (int)blocks;
}

static size_t take_event(reina_engine *e) {
    if (e->ev_used == e->ev_pool.size()) {
        hipEvent_t ev;
        hipEventCreate(&ev);
        e->ev_pool.push_back(ev);
    }
    return e->ev_used++;
}

static void resolve_profile(reina_engine *e) {
    for (auto &p : e->scan_pairs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e->ev_pool[p.first], e->ev_pool[p.second]) == hipSuccess) {
            e->scan_ms += ms;
            e->scan_launches++;
        }
    }
    for (auto &p : e->day_pairs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e->ev_pool[p.first], e->ev_pool[p.second]) == hipSuccess) e->all_ms += ms;
    }
    e->scan_pairs.clear();
    e->day_pairs.clear();
    e->ev_used = 0;
}

extern "C" {

int reina_abi_version(void) { return 1; }

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
    reina_engine *e = new reina_engine();
    e->cfg = *cfg;
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
    e->h_params.max_work_items = cfg->max_work_items;
    e->h_params.max_candidates = cfg->max_candidates;
    e->h_params.max_queue = cfg->max_queue;
    HIP_CHECK(hipMalloc(&e->d_params, sizeof(DevParams)));
    HIP_CHECK(hipMalloc(&e->d_tables, sizeof(Tables)));
    HIP_CHECK(hipMalloc(&e->d_ref, sizeof(MemberRef)));
    HIP_CHECK(hipMemcpy(e->d_params, &e->h_params, sizeof(DevParams), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(e->d_tables, &e->h_tables, sizeof(Tables), hipMemcpyHostToDevice));
    {
        size_t lds = con_shared_bytes(REINA_MAX_AGES, REINA_MAX_SHARDS);
        if (lds < (size_t)REINA_MAX_HOSP_EVENTS * 8) lds = (size_t)REINA_MAX_HOSP_EVENTS * 8;
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_hosp_contacts),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    *out = e;
    return REINA_OK;
}

int reina_destroy(reina_engine_t *e) {
    if (!e) return REINA_E_INVALID;
    for (auto ev : e->ev_pool) hipEventDestroy(ev);
    for (size_t k = 0; k < e->stage.size(); k++) {
        hipEventSynchronize(e->stage_ev[k]);
        hipEventDestroy(e->stage_ev[k]);
        hipHostFree(e->stage[k]);
    }
    hipFree(e->d_params);
    hipFree(e->d_tables);
    hipFree(e->d_ref);
    delete e;
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
    return REINA_OK;
}

int reina_init_state(reina_engine_t *e, int32_t beds, int32_t icu, void *stream) {
    if (!e || !e->bound) return REINA_E_NOT_BOUND;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_init, dim3(grid_for(e->cfg.n_agents, 256, 4096), 1), dim3(256), 0, s, e->d_ref, beds, icu);
    HIP_CHECK(hipGetLastError());
    return REINA_OK;
}

int reina_set_initial_state(reina_engine_t *e, const reina_initial_state_t *ic, void *stream) {
    if (!e || !ic) return REINA_E_INVALID;
    if (!e->bound) return REINA_E_NOT_BOUND;
    hipLaunchKernelGGL(k_initial_state, dim3(1, 1), dim3(PRO_THREADS), 0, (hipStream_t)stream, e->d_ref, *ic);
    HIP_CHECK(hipGetLastError());
    return REINA_OK;
}

int reina_upload_contact_tables(reina_engine_t *e, const reina_contact_tables_t *t, void *stream) {
    if (!e || !t) return REINA_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t A = e->cfg.nr_ages;
    std::memcpy(e->h_params.nrc, t->nr_contacts_by_age, sizeof(float) * A);
    std::memcpy(e->h_params.tcount, t->count, sizeof(int32_t) * A);
    std::memcpy(e->h_params.mask_p, t->mask_p, sizeof(float) * A * 8);
    e->h_params.n_ranges = t->n_ranges;
    std::memcpy(e->h_params.range_min, t->range_min, sizeof(t->range_min));
    std::memcpy(e->h_params.range_max, t->range_max, sizeof(t->range_max));
    std::memcpy(e->h_tables.thr, t->threshold, sizeof(uint32_t) * A * REINA_MAX_ENTRIES);
    std::memcpy(e->h_tables.meta, t->meta, sizeof(uint32_t) * A * REINA_MAX_ENTRIES);
    e->uniform_meta = 1;  // every participant age lists the same (place, contact range) sequence?
    for (uint32_t a = 1; a < A && e->uniform_meta; a++)
        if (t->count[a] != t->count[0] ||
            std::memcmp(e->h_tables.meta[a], e->h_tables.meta[0], sizeof(uint32_t) * (size_t)t->count[0]) != 0)
            e->uniform_meta = 0;
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
    std::memcpy(&e->stage[slot]->t, &e->h_tables, sizeof(Tables));
    // The transfer is a KERNEL that reads the pinned host slot directly (zero-copy): an ordinary
    // dispatch in the day stream.  hipMemcpyAsync of the 98 KB table was measured to block the host
    // for 7-8 ms once per run when >1000 dispatches were queued ahead of it (ROCm 7.2).
    void *dsrc = nullptr;
    HIP_CHECK(hipHostGetDevicePointer(&dsrc, e->stage[slot], 0));
    static_assert(sizeof(DevParams) % 4 == 0 && sizeof(Tables) % 4 == 0 && offsetof(reina_engine::Stage, t) % 4 == 0, "word copies");
    const uint32_t *src_w = reinterpret_cast<const uint32_t *>(dsrc);
    hipLaunchKernelGGL(k_upload, dim3(64), dim3(256), 0, s, reinterpret_cast<uint32_t *>(e->d_params), src_w,
                       (uint32_t)(sizeof(DevParams) / 4), reinterpret_cast<uint32_t *>(e->d_tables),
                       src_w + offsetof(reina_engine::Stage, t) / 4, (uint32_t)(sizeof(Tables) / 4));
    HIP_CHECK(hipEventRecord(e->stage_ev[slot], s));
    return REINA_OK;
}

// One day's launches for K engine instances at once (K = 1: a single engine; K > 1: a group of
// identically configured engines, one launch per phase for all of them, member = blockIdx.y).
// `e` is the representative engine: geometry, scenario flags, optional second stream.
static int launch_day_begin(reina_engine_t *e, const MemberRef *refs, uint32_t K, const reina_day_t &dp,
                            uint32_t hist_slot, hipStream_t s) {
    const uint32_t N = e->cfg.n_agents;
    if (dp.testing_mode != RT_NO_TESTING) e->testing_ever = true;
    uint32_t n_pre = 0, n_post = 0;
    for (uint32_t b = 0; b < dp.n_import_batches; b++)
        (dp.import_batches[b].pre_init ? n_pre : n_post) += dp.import_batches[b].count;
    const int weekly_own = n_pre == 0 && n_post > 0;   // (intervention imports share the claim keys: same workgroup then)
    if (!e->testing_ever) {
        hipLaunchKernelGGL(k_open<0>, dim3(2, K), dim3(PRO_THREADS), 0, s, refs, dp, hist_slot, weekly_own);
    } else {
        // (groups: the members share the chip, so each gets proportionally fewer workgroups per phase)
        int tg = grid_for(N / 64 + 1, PRO_THREADS, 64);
        if (K > 1 && tg > (int)(256 / K)) tg = 256 / K > 0 ? (int)(256 / K) : 1;
        const int g = 2 + tg;
        if (dp.testing_mode == RT_ALL_WITH_SYMPTOMS_CT && N <= 8000000u) {
            hipLaunchKernelGGL(k_open<3>, dim3(g, K), dim3(PRO_THREADS), 0, s, refs, dp, hist_slot, weekly_own);  // detects + traces, both levels
        } else if (dp.testing_mode == RT_ALL_WITH_SYMPTOMS_CT) {
            hipLaunchKernelGGL(k_open<2>, dim3(g, K), dim3(PRO_THREADS), 0, s, refs, dp, hist_slot, weekly_own);  // detects + traces level 0
            hipLaunchKernelGGL(k_test_trace1, dim3(grid_for(N / 64 + 1, 256, 256), K), dim3(256), 0, s, refs, dp);
        } else {
            hipLaunchKernelGGL(k_open<1>, dim3(g, K), dim3(PRO_THREADS), 0, s, refs, dp, hist_slot, weekly_own);
        }
    }
    if (dp.n_vaccinations) hipLaunchKernelGGL(k_vaccinate, dim3(1, K), dim3(PRO_THREADS), 0, s, refs, dp);
    // scan geometry: tiles of 512 agents, as many waves as tiles (small populations) up to 8192
    const uint32_t scan_tiles = ((N >> 2) + 127u) / 128u;
    uint32_t scan_blocks = (scan_tiles + SCAN_WAVES - 1) / SCAN_WAVES;  // one 512-agent tile per wave until the grid cap
    if (scan_blocks < 1) scan_blocks = 1;
    if (scan_blocks > REINA_MAX_SCAN_WAVES / SCAN_WAVES) scan_blocks = REINA_MAX_SCAN_WAVES / SCAN_WAVES;
    const uint32_t scan_waves = scan_blocks * SCAN_WAVES;
    if (e->profile && K == 1 && dp.day % e->profile_stride == 0) {
        // start/stop timestamps ride on the kernel's own dispatch packet: no extra stream commands
        const size_t ev_s0 = take_event(e), ev_s1 = take_event(e);
        hipExtLaunchKernelGGL(k_scan, dim3(scan_blocks, K), dim3(SCAN_THREADS), 0, s, e->ev_pool[ev_s0], e->ev_pool[ev_s1], 0,
                              refs, dp);
        e->scan_pairs.emplace_back(ev_s0, ev_s1);
    } else {
        hipLaunchKernelGGL(k_scan, dim3(scan_blocks, K), dim3(SCAN_THREADS), 0, s, refs, dp);
    }
    {   // bed / ICU events (workgroup 0, latency-bound) beside the contact sampling (workgroups 1..)
        uint32_t con_blocks = (scan_waves + CON_WAVES - 1) / CON_WAVES;
        if (con_blocks > 255) con_blocks = 255;  // with the event workgroup: one resident wave of workgroups on 256 CUs
        if (K > 1) {  // a group: 256 workgroups for all members together, each staging its tables once for more slices
            const uint32_t per = 256u / K > 1u ? 256u / K - 1u : 1u;
            if (con_blocks > per) con_blocks = per;
        }
        size_t lds = con_shared_bytes(e->cfg.nr_ages, e->cfg.n_shards);
        if (lds < (size_t)REINA_MAX_HOSP_EVENTS * 8) lds = (size_t)REINA_MAX_HOSP_EVENTS * 8;
        hipLaunchKernelGGL(k_hosp_contacts, dim3(con_blocks + 1, K), dim3(CON_THREADS), lds, s, refs, dp, scan_waves, scan_tiles,
                           e->uniform_meta);
    }
    HIP_CHECK(hipGetLastError());
    return REINA_OK;
}

static int launch_day_end(reina_engine_t *e, const MemberRef *refs, uint32_t K, const reina_day_t &dp, hipStream_t s) {
    const uint32_t N = e->cfg.n_agents;
    if (e->cfg.n_shards > 1)
        hipLaunchKernelGGL(k_remote, dim3(grid_for(N / 256 + 1, 256, 256), K), dim3(256), 0, s, refs, dp);
    {
        const uint32_t scan_tiles = ((N >> 2) + 127u) / 128u;
        uint32_t scan_blocks = (scan_tiles + SCAN_WAVES - 1) / SCAN_WAVES;
        if (scan_blocks < 1) scan_blocks = 1;
        if (scan_blocks > REINA_MAX_SCAN_WAVES / SCAN_WAVES) scan_blocks = REINA_MAX_SCAN_WAVES / SCAN_WAVES;
        int ig = grid_for(N / 64 + 1, 256, 512) * 2;  // even: candidates / deferred lists
        if (K > 1 && ig > (int)(4096 / K)) ig = (int)(4096 / K) >= 2 ? ((int)(4096 / K) & ~1) : 2;
        hipLaunchKernelGGL(k_install, dim3(ig, K), dim3(256), 0, s, refs, dp, scan_blocks * SCAN_WAVES, scan_tiles);
    }
    HIP_CHECK(hipGetLastError());
    return REINA_OK;
}

int reina_step_day_begin(reina_engine_t *e, const reina_day_t *day, void *stream) {
    if (!e || !day) return REINA_E_INVALID;
    if (!e->bound) return REINA_E_NOT_BOUND;
    return launch_day_begin(e, e->d_ref, 1, *day, 0, (hipStream_t)stream);
}

int reina_step_day_end(reina_engine_t *e, const reina_day_t *day, void *stream) {
    if (!e || !day) return REINA_E_INVALID;
    if (!e->bound) return REINA_E_NOT_BOUND;
    return launch_day_end(e, e->d_ref, 1, *day, (hipStream_t)stream);
}

int reina_step_day(reina_engine_t *e, const reina_day_t *day, void *stream) {
    int rc = reina_step_day_begin(e, day, stream);
    if (rc) return rc;
    return reina_step_day_end(e, day, stream);
}

int reina_run_days(reina_engine_t *e, const reina_day_t *days, uint32_t n_days, void *stream) {
    for (uint32_t k = 0; k < n_days; k++) {
        int rc = reina_step_day(e, &days[k], stream);
        if (rc) return rc;
    }
    return REINA_OK;
}

int reina_run_days_hist(reina_engine_t *e, const reina_day_t *days, uint32_t n_days, int32_t *history_base, void *stream) {
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
    HIP_CHECK(hipMalloc(&g->d_refs, sizeof(MemberRef) * n));
    *out = g;
    return REINA_OK;
}

int reina_group_destroy(reina_group_t *g) {
    if (!g) return REINA_E_INVALID;
    hipFree(g->d_refs);
    delete g;
    return REINA_OK;
}

int reina_group_upload_contact_tables(reina_group_t *g, const reina_contact_tables_t *t, void *stream) {
    if (!g) return REINA_E_INVALID;
    for (auto m : g->members) {
        int rc = reina_upload_contact_tables(m, t, stream);
        if (rc) return rc;
    }
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
    for (auto m : g->members) tested = tested || m->testing_ever;
    for (uint32_t d = 0; d < n_days; d++) tested = tested || days[d].testing_mode != RT_NO_TESTING;
    e0->testing_ever = e0->testing_ever || (tested && n_days == 0);
    for (uint32_t d = 0; d < n_days; d++) {
        reina_day_t dp = days[d];
        dp.history_row = nullptr;
        int rc = launch_day_begin(e0, g->d_refs, K, dp, d, s);
        if (rc) return rc;
        rc = launch_day_end(e0, g->d_refs, K, dp, s);
        if (rc) return rc;
    }
    for (auto m : g->members) m->testing_ever = m->testing_ever || e0->testing_ever;
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

int reina_profile_enable(reina_engine_t *e, int enable) {
    if (!e) return REINA_E_INVALID;
    e->profile = enable != 0;
    e->profile_stride = enable > 1 ? (uint32_t)enable : 1u;
    // create the timing events up front: hipEventCreate inside a timed region costs microseconds each
    while (e->profile && e->ev_pool.size() < 1024) {
        hipEvent_t ev;
        HIP_CHECK(hipEventCreate(&ev));
        e->ev_pool.push_back(ev);
    }
    return REINA_OK;
}

int reina_profile_read(reina_engine_t *e, double *scan_ms_total, uint64_t *scan_launches, double *all_ms_total) {
    if (!e) return REINA_E_INVALID;
    HIP_CHECK(hipDeviceSynchronize());
    resolve_profile(e);
    if (scan_ms_total) *scan_ms_total = e->scan_ms;
    if (scan_launches) *scan_launches = e->scan_launches;
    if (all_ms_total) *all_ms_total = e->all_ms;
    e->scan_ms = 0;
    e->all_ms = 0;
    e->scan_launches = 0;
    return REINA_OK;
}

}  // extern "C"
