/* include/reina_hip.h -- C ABI of the MI355X-native REINA agent engine (libreina_hip.so).
 *
 * The reference has no C-level plugin interface: its boundary is the Python extension module
 * `cythonsim.model` (cythonsim/main.pyx, imported at calc/simulation.py:14).  This header is the
 * FFI a Python `Context` with the reference's method set binds instead of the Cython class
 * (reina_model_amd/model.py; binding shown in INTEGRATION.md).  Each entry point names the
 * reference code it replaces.
 *
 * Conventions: `extern "C"`, plain structs / pointers / sizes, no torch or C++ types.  Every
 * function returns 0 on success or a negative REINA_E_* code.  "dev" pointers are device (HBM)
 * addresses owned by the caller (PyTorch-ROCm tensors); the library never frees them.  All work
 * is enqueued on the `stream` passed in (a hipStream_t as void*); nothing here synchronises
 * except reina_read_* which copy to host.
 */
#ifndef REINA_HIP_H
#define REINA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REINA_MAX_AGES 128      /* A: ages 0..A-1 (reference: 101, main.pyx:1355) */
#define REINA_MAX_VARIANTS 4    /* wild-type + 3 (reference default: 2, variables.py:413-435) */
#define REINA_MAX_ENTRIES 96    /* contact entries per participant age (reference: 6 places x 15 ranges = 90) */
#define REINA_NR_PLACES 6       /* main.pyx:64-74 */
#define REINA_IOT_LEN 21        /* infectiousness profile day -10..10, main.pyx:660-682 */
#define REINA_MAX_IMPORT_CLASSES 16
#define REINA_MAX_IMPORT_BATCHES 16
#define REINA_MAX_VACCINATIONS 16
#define REINA_MAX_HOSP_EVENTS 16384
#define REINA_MAX_SCAN_WAVES 8192
#define REINA_MAX_SHARDS 16     /* ranks an agent population can be sharded over */
#define REINA_MAX_RANGES 32     /* distinct contact age ranges (reference: 15) */
#define REINA_MAX_DAYS 4096     /* reina_day_t.day < 4096: winner-selection keys carry the day in 12 bits and are never
                                   cleared (the reference's default scenario runs 565 days, variables.py:233) */
#define REINA_PRESSURE_WORDS (REINA_MAX_SHARDS * REINA_MAX_RANGES * REINA_MAX_VARIANTS)
/* The block a sharded population all-reduces once per day (buffers.pressure) = the REINA_PRESSURE_WORDS above followed by
 * every shard's BED / ICU EVENT MAPS: shard s writes, for each of its R priority buckets of the day's bed / ICU events, the
 * composed saturating map of that bucket's events (free beds, free units before -> after; 64 bits, two int32 words:
 * csrc/reina_prims.h rp_sat_pack) into segment s and zeros elsewhere, so that the sum is the table of all shards' maps.
 * With it every shard walks its own events against ONE pool of beds and ICU units, in the global order (bucket, shard,
 * priority, agent): exact pooled capacity (the reference hands beds out of one counter,
 * main.pyx:617-651).  R must be the same on every shard: reina_config_t.hosp_ranges. */
#define REINA_EXCHANGE_WORDS(n_shards, hosp_ranges) \
    ((size_t)REINA_PRESSURE_WORDS + ((n_shards) > 1 ? (size_t)(n_shards) * 2u * (size_t)(hosp_ranges) : 0u))
#define REINA_MIRROR_CELLS (REINA_MAX_RANGES * REINA_MAX_VARIANTS)
/* sharded populations: the four pressure words of contact range REINA_MAX_RANGES - 1 (never a real range: a
 * sharded engine accepts at most REINA_MAX_RANGES - 1) carry each shard's free beds / free ICU units at day open and
 * its demand of the day (admission / ICU-transfer requests) through the same all-reduce: their sums are the ONE pool
 * the day's events of all shards are walked against, and decide -- for all shards alike -- whether a resource can run
 * out today (whether the events' order matters) */
#define REINA_PRESSURE_FREE_BEDS(rank) (((rank) * REINA_MAX_RANGES + (REINA_MAX_RANGES - 1)) * REINA_MAX_VARIANTS)
#define REINA_PRESSURE_FREE_ICU(rank) (REINA_PRESSURE_FREE_BEDS(rank) + 1)
#define REINA_PRESSURE_DEMAND_BEDS(rank) (REINA_PRESSURE_FREE_BEDS(rank) + 2)
#define REINA_PRESSURE_DEMAND_ICU(rank) (REINA_PRESSURE_FREE_BEDS(rank) + 3)

/* error codes */
#define REINA_OK 0
#define REINA_E_INVALID (-1)
#define REINA_E_HIP (-2)
#define REINA_E_NOT_BOUND (-3)

/* problem codes reported by the engine (reina_read_counters scalar REINA_S_PROBLEM); 1..9 are the
 * reference's SimulationProblem values (main.pyx:51-61), >= 100 are capacity overflows of this
 * engine's work lists. */
#define REINA_PROBLEM_WORK_OVERFLOW 100
#define REINA_PROBLEM_CANDIDATE_OVERFLOW 101
#define REINA_PROBLEM_QUEUE_OVERFLOW 102
#define REINA_PROBLEM_HOSPITAL_OVERFLOW 103
#define REINA_PROBLEM_DAYS_OVERFLOW 104
#define REINA_PROBLEM_SYNC_TIMEOUT 105   /* a workgroup gave up waiting (2^22 polls, seconds) for the day's opening bookkeeping */
#define REINA_PROBLEM_EXCHANGE_OVERFLOW 106   /* exact attribution: more records for one peer shard than an exchange segment holds (reina_config_t.xchg_cap) */
#define REINA_PROBLEM_INFECTEE_POOL_OVERFLOW 107   /* exact attribution: more infectees beyond the inline slots than reina_config_t.pool_cap */

/* EXACT CROSS-SHARD ATTRIBUTION (the reference records the TRUE infector and appends to the infector's
 * infectee array at infection time, main.pyx:219-233, and contact tracing walks exactly those links, :495-512).
 * With reina_config_t.exact_attribution every link field of a sharded population -- reina_cold_t.infector, the inline
 * infectee slots, the overflow list -- holds a GLOBAL id (shard, index), and the shards exchange 8-byte records through
 * fixed-capacity all-to-all segments (buffers.xsend / xrecv) instead of the pressure counts:
 *   contact records   (mid-day, with the all-reduce): a cross-shard contact that passed did_infect's whole test at the source
 *                     (the source knows the target's age: every shard's age_start table is global knowledge) -> the target's
 *                     shard, which claims the target with the source's own key;
 *   feedback records  (after the day's last launch): (source, infectee) of every cross-shard infection -> the source's
 *                     shard, so that its infection count and infectee list are true before the next morning;
 *   tracing requests  (contact-tracing days, one exchange per level): (candidate, tracer) -> the candidate's shard, which
 *                     rolls (the roll is keyed by candidate and tracer), queues, and expands level 1.
 * A global id is non-negative as an int32 (-1 stays "none"): 4 bits of shard above 27 bits of index, so an engine instance
 * holds fewer than 2^27 agents in this mode. */
#define REINA_GID_SHIFT 27
#define REINA_GID_INDEX_MASK ((1u << REINA_GID_SHIFT) - 1u)
#define REINA_GID(shard, index) ((int32_t)(((uint32_t)(shard) << REINA_GID_SHIFT) | (uint32_t)(index)))
/* an exchange buffer: n_shards segments of REINA_XCHG_SEG_WORDS 64-bit words; segment d of xsend is bound for shard d, segment s
 * of xrecv came from shard s; word 0 of a segment = its record count (may exceed xchg_cap after an overflow: readers clamp),
 * then xchg_cap record slots: [gid : 31 << 32][flags : 5 << 27][index at the receiving shard : 27] (csrc/reina_prims.h: rp_xrec),
 * then -- round 6, ABI 7 -- the TRAILER: what a population under mirror attribution all-reduces for its one pool of beds and ICU
 * units rides in the mid-day exchange of the contact records instead (one collective less a day): word 0 = the sender's free
 * beds (low half) and free ICU units (high half) at day open, word 1 = its demand of the day (admission / ICU-transfer requests,
 * same halves), words 2 .. 2 + R - 1 = the packed maps of its R = hosp_ranges bed / ICU event buckets (the sender's segment of
 * REINA_EXCHANGE_WORDS' table).  The sender fills the trailer of every peer's segment before the mid-day exchange
 * (k_hosp_presort), the receiver copies the trailers into its own buffers.pressure behind it (k_remote): every kernel
 * downstream finds the block as an all-reduce would have left it. */
#define REINA_XCHG_TRAILER_WORDS(hosp_ranges) (2u + (size_t)(hosp_ranges))
#define REINA_XCHG_SEG_WORDS(xchg_cap, hosp_ranges) ((size_t)(xchg_cap) + 1u + REINA_XCHG_TRAILER_WORDS(hosp_ranges))
#define REINA_XCHG_WORDS(n_shards, xchg_cap, hosp_ranges) ((size_t)(n_shards) * REINA_XCHG_SEG_WORDS(xchg_cap, hosp_ranges))

/* per-age counter arrays, Population stats main.pyx:1335-1341 */
enum {
    REINA_C_INFECTED = 0, REINA_C_DETECTED, REINA_C_ALL_DETECTED, REINA_C_ALL_INFECTED,
    REINA_C_IN_WARD, REINA_C_HOSPITALIZED, REINA_C_IN_ICU, REINA_C_CUM_ICU, REINA_C_DEAD,
    REINA_C_SUSCEPTIBLE, REINA_C_RECOVERED, REINA_C_VACCINATED, REINA_C_NON_HOSPITAL_DEATHS,
    REINA_C_NEW_INFECTIONS, REINA_C_NR
};

/* scalar slots following the per-age arrays (Context / HealthcareSystem scalars,
 * main.pyx:452-453,1756; daily_contacts :1341; infected_by_variant :1338) */
enum {
    /* (a SHARD's available_* are its contributions to the one pool of the population: admissions are decided against the
     * pooled count while every shard adds only its own events' changes, so a single shard's value may be negative or
     * exceed its REINA_S_BEDS -- only the sum over the shards is the hospital's free capacity; model.Context sums them) */
    REINA_S_AVAILABLE_BEDS = 0, REINA_S_AVAILABLE_ICU, REINA_S_BEDS, REINA_S_ICU_UNITS,
    REINA_S_TOTAL_INFECTIONS, REINA_S_TOTAL_INFECTORS, REINA_S_EXPOSED_PER_DAY,
    REINA_S_CT_CASES_PER_DAY, REINA_S_PROBLEM, REINA_S_DAY, REINA_S_UNABLE_TO_IMPORT,
    REINA_S_QUEUE_LEN,
    REINA_S_DAILY_CONTACTS = 16,                        /* [6] */
    REINA_S_INFECTED_BY_VARIANT = 24,                   /* [4] */
    REINA_S_NR = 32
};
#define REINA_COUNTER_WORDS (REINA_C_NR * REINA_MAX_AGES + REINA_S_NR)

/* control block (device, int32): list lengths and cursors the kernels hand to each other */
enum {
    REINA_L_WORK = 0, REINA_L_CAND, REINA_L_QUEUE0, REINA_L_QUEUE1, REINA_L_LEVEL1, REINA_L_HOSP,
    REINA_L_CONTACTS, REINA_L_HOSP_ADMIT, REINA_L_ICU_ADMIT,
    REINA_L_DAY_OPEN,                                   /* day + 1 once that day's snapshot / zeroing is done */
    REINA_L_TRACE_DONE,                                 /* level-0 tracing workgroups finished today (folded level 1) */
    REINA_L_CAND_OVF,                                   /* candidate records that did not fit their slice's region today */
    REINA_L_HOSP_PEAK,                                  /* busiest day so far on which a bed or ICU unit could run out (the events' order
                                                           mattered): its bed / ICU event count */
    REINA_L_OPEN_TICKET,                                /* arrival tickets of the day-opening launch: 0 opens the day, 1 places the weekly
                                                           imports, 2.. work off the test queue; reset by the launch's last arrival */
    REINA_L_BEDS_OPEN, REINA_L_ICU_OPEN,                /* free beds / ICU units when the day opened (after new capacity was added) */
    REINA_L_EV_HOSPITALIZE, REINA_L_EV_TO_ICU,          /* the day's admission / ICU-transfer requests (counted by the stream): with the two words
                                                           above they say whether a resource can run out today, i.e. whether event ORDER matters */
    REINA_L_WALK_TICKET,                                /* a large population's ordered event walk: the next priority bucket to be handed out
                                                           (zeroed by the day's opening) */
    REINA_L_SORT_TICKET,                                /* the same for a sharded population's pre-sort of its event buckets */
    REINA_L_IMPORT_SYNC,                                /* [4] arrival words of the workgroups that share a day's weekly imports (zeroed by the
                                                           day's last launch) */
    REINA_L_ACTIVE = 24,                                /* [2] agents the daily stream found something to do for (infected, or removed and not
                                                           yet counted into R), word [day & 1] (a diagnostic: rounds 4-5 chose the form of
                                                           k_day's stream by yesterday's count; the launch now does, by population size) */
    REINA_L_POOL = 26,                                  /* exact attribution: nodes of buffers.infectee_pool handed out so far */
    REINA_L_PRESORT_DONE = 28,                          /* exact attribution: workgroups of k_hosp_presort that have finished today (the last one fills
                                                           the exchange segments' trailers and resets the word) */
    REINA_L_XCHG_PEAK = 27,                             /* exact attribution: the most records any exchange segment of this shard has held so far
                                                           (against reina_config_t.xchg_cap: how close a run came to problem 106) */
    REINA_L_VACC_CURSOR = 32,                           /* [REINA_MAX_VACCINATIONS] */
    REINA_L_DET_SIDE = 48,                              /* [REINA_MAX_AGES] the day's detections by age from the test queue (the day's opening
                                                           launch), folded into the counters by the day's last launch */
    REINA_L_NR = 48 + REINA_MAX_AGES
};

/* The day's bed / ICU events are kept in buckets by priority range (buffers.hosp_events, 64-bit words):
 * R = REINA_HOSP_RANGES(n_agents) buckets of REINA_HOSP_BUCKET_CAP(n_agents, max_hosp_events) keys, preceded by
 * R / 2 words of bucket counts and 2 R words for the buckets' published saturating maps (the first R are used).  A
 * population of at most REINA_HOSP_SMALL_AGENTS agents has its events walked by one workgroup (at most
 * REINA_MAX_HOSP_EVENTS - 1024 a day); a larger one by one wave (or, for a bucket of more than 256 keys, one workgroup)
 * per bucket inside the day's last launch.
 * LIMIT: a bucket holds at most REINA_HOSP_MAX_BUCKET_KEYS keys, i.e. REINA_HOSP_BUCKET_CAP <= that: reina_create
 * refuses a larger max_hosp_events.  With the usual max_hosp_events = n_agents / 128 that is reached at about 2.64e8
 * agents per engine instance; above it pass REINA_HOSP_MAX_EVENTS_FOR(n_agents) (2 064 384 events a day: a day with
 * more fails loudly, problem 103) or shard the population (reina_model_amd.engine.default_max_hosp_events does). */
#define REINA_HOSP_MAX_RANGES 1024
#define REINA_HOSP_MAX_BUCKET_KEYS 4096
#define REINA_HOSP_SMALL_AGENTS (128u * REINA_MAX_HOSP_EVENTS)
static inline uint32_t REINA_HOSP_RANGES(uint32_t n_agents) {
    uint32_t r = 16;
    while (r < REINA_HOSP_MAX_RANGES && (uint64_t)r * 65536u < n_agents) r <<= 1;
    return r;
}
static inline uint32_t REINA_HOSP_BUCKET_CAP_R(uint32_t ranges, uint32_t max_hosp_events) {
    const uint32_t cap = max_hosp_events > REINA_MAX_HOSP_EVENTS ? max_hosp_events : REINA_MAX_HOSP_EVENTS;
    return 2u * (cap / ranges) + 64u;   /* twice the mean at the day capacity: > 10 sigma */
}
static inline uint32_t REINA_HOSP_BUCKET_CAP(uint32_t n_agents, uint32_t max_hosp_events) {
    return REINA_HOSP_BUCKET_CAP_R(REINA_HOSP_RANGES(n_agents), max_hosp_events);
}
static inline uint32_t REINA_HOSP_MAX_EVENTS_FOR(uint32_t n_agents) {   /* the largest max_hosp_events reina_create accepts */
    return ((REINA_HOSP_MAX_BUCKET_KEYS - 64u) / 2u) * REINA_HOSP_RANGES(n_agents);
}
static inline size_t REINA_HOSP_EVENT_WORDS_R(uint32_t ranges, uint32_t max_hosp_events) {   /* (reina_config_t.hosp_ranges given) */
    const size_t r = ranges;
    return r / 2 + 2 * r + r * (size_t)REINA_HOSP_BUCKET_CAP_R(ranges, max_hosp_events);
}
static inline size_t REINA_HOSP_EVENT_WORDS(uint32_t n_agents, uint32_t max_hosp_events) {
    return REINA_HOSP_EVENT_WORDS_R(REINA_HOSP_RANGES(n_agents), max_hosp_events);
}

typedef struct {
    uint32_t n_agents;        /* agents of this engine instance, sorted by age */
    uint32_t nr_ages;         /* A */
    uint32_t nr_variants;     /* V */
    uint32_t max_hosp_events; /* bed / ICU events one day may hold (0 or less than REINA_MAX_HOSP_EVENTS = that number):
                                 sizes the event buckets; more than REINA_HOSP_BUCKET_CAP keys in one bucket, or more
                                 than one workgroup's walk holds in a small population, fail the run (problem 103) */
    uint64_t seed;            /* Philox key (random_seed of Context, main.pyx:1759); same on all shards,
                                 the engine mixes the rank in */
    uint32_t max_work_items;  /* capacity of work_items (records) */
    uint32_t max_candidates;  /* capacity of candidates (records) */
    uint32_t max_queue;       /* capacity of each testing queue */
    uint32_t n_shards;        /* G >= 1: the population is split over G engine instances (ranks) */
    uint32_t shard_rank;      /* this instance's rank in [0, G) */
    uint32_t mirror_slots;    /* power of two: slots per (range, variant) cell of buffers.mirror */
    uint32_t hosp_ranges;     /* 0: REINA_HOSP_RANGES(n_agents); a sharded population passes the SAME power of two (16 ..
                                 REINA_HOSP_MAX_RANGES) on every shard, e.g. REINA_HOSP_RANGES of its largest shard: the
                                 shards exchange per-bucket maps of the day's bed / ICU events */
    int32_t age_start[REINA_MAX_AGES + 1]; /* first agent index of each age; [A] = n_agents
                                              (Population.age_start, main.pyx:1332,1442) */
    uint32_t exact_attribution; /* sharded populations: 1 = exact cross-shard attribution (above), 0 = mirror attribution
                                   (stand-in infectors, one all-reduce per day; reina_model_amd/sharding.py) */
    uint32_t xchg_cap;          /* exact attribution: records per peer segment of buffers.xsend / xrecv */
    uint32_t pool_cap;          /* exact attribution: nodes of buffers.infectee_pool */
    uint32_t reserved_;
    const int32_t *shard_age_start; /* exact attribution: host array [n_shards][REINA_MAX_AGES + 1], every shard's age_start
                                       (read by reina_create only) */
} reina_config_t;

/* Disease parameters, all float32 like the reference's `cdef float` fields (main.pyx:787-806).
 * Age-classed step functions (ClassifiedValues + cv_get_greatest_lte, main.pyx:684-730) are
 * expanded per age by the host. Conditional probabilities as produced by variant_init :834-843. */
typedef struct {
    float infectiousness_multiplier[REINA_MAX_VARIANTS];
    float p_asymptomatic_infection[REINA_MAX_VARIANTS];
    float p_hospital_death_no_beds[REINA_MAX_VARIANTS];
    float p_icu_death_no_beds[REINA_MAX_VARIANTS];
    float mean_incubation_duration[REINA_MAX_VARIANTS];
    float mean_duration_from_onset_to_death[REINA_MAX_VARIANTS];
    float mean_duration_from_onset_to_recovery[REINA_MAX_VARIANTS];
    float ratio_of_duration_before_hospitalisation[REINA_MAX_VARIANTS];
    float ratio_of_duration_in_ward[REINA_MAX_VARIANTS];
    float p_mask_protects_others[REINA_MAX_VARIANTS];
    float p_mask_protects_wearer[REINA_MAX_VARIANTS];
    float infectiousness_over_time[REINA_MAX_VARIANTS][REINA_IOT_LEN + 3]; /* day -10..10 */
    float p_susceptibility[REINA_MAX_VARIANTS][REINA_MAX_AGES];
    /* severity is always drawn with the target's pre-infection variant, i.e. 0 (quirk Q3,
     * main.pyx:211-212 vs :224-225) */
    float p_symptomatic[REINA_MAX_AGES];
    float p_severe_given_symptomatic[REINA_MAX_AGES];
    float p_critical_given_severe[REINA_MAX_AGES];
    float p_fatal_given_critical[REINA_MAX_AGES];
    float p_death_outside_hospital[REINA_MAX_AGES];
    /* imported infection age classes (main.pyx:1376-1384,1632-1650) */
    uint32_t n_import_classes;
    int32_t import_class_min_age[REINA_MAX_IMPORT_CLASSES];
    int32_t import_class_max_age[REINA_MAX_IMPORT_CLASSES];
    float import_class_cum[REINA_MAX_IMPORT_CLASSES];
} reina_disease_t;

/* Contact sampling tables for one rebuild (ContactMatrix.generate_contact_probabilities,
 * main.pyx:1184-1235), host arrays:
 *   nr_contacts_by_age[A]          float32(total contacts/day)
 *   count[A]                       entries per age (<= REINA_MAX_ENTRIES)
 *   threshold[A][REINA_MAX_ENTRIES] uint32 floor(cum_p * 2^32) (saturated), padded with 0xFFFFFFFF
 *   meta[A][REINA_MAX_ENTRIES]      place | cmin << 8 | cmax << 16 | range_id << 24
 *   mask_p[A][8]                    float32 mask probability by (participant age, place)
 *   range_min/max[n_ranges]         the distinct contact age ranges, indexed by range_id */
typedef struct {
    const float *nr_contacts_by_age;
    const int32_t *count;
    const uint32_t *threshold;
    const uint32_t *meta;
    const float *mask_p;
    uint32_t n_ranges;
    int32_t range_min[REINA_MAX_RANGES];
    int32_t range_max[REINA_MAX_RANGES];
} reina_contact_tables_t;

/* Everything about an agent that is touched only at EVENTS (infection, onset, hospital, tracing), one 32-byte record =
 * one HBM sector per agent (the reference keeps the same fields in its Person struct, main.pyx:132-144).  Round 2 kept
 * them in seven arrays: installing one infection touched five sectors of the target and two of the source. */
typedef struct {
    uint64_t claim;           /* winner-selection key of the day's exposures, init 0xFF..FF; [4095-day:12][priority:20][id:32] */
    int32_t infector;         /* Person.infector, -1 = none */
    int32_t n_infected;       /* Person.other_people_infected */
    float onset_days;         /* Person.days_from_onset_to_removed */
    int32_t vacc_day;         /* Person.day_of_vaccination, -1 = never */
    int32_t first_infectee;   /* infectees beyond the REINA_INLINE_INFECTEES inline slots: head of a linked list, -1 = empty
                                 (exact attribution: a node of buffers.infectee_pool) */
    int32_t next_sibling;     /* ... and its link: next such infectee of the same infector, -1 = end (exact attribution: unused,
                                 the links are in the pool) */
} reina_cold_t;
/* Person.infectees (main.pyx:128,231: 64 ids per person, kept while contact tracing is on): the first
 * REINA_INLINE_INFECTEES of an agent's infectees sit side by side in buffers.infectees (slot = their rank among the
 * agent's infections, -1 = empty), so contact tracing reads them in one access; the rest go to the linked list above. */
#define REINA_INLINE_INFECTEES 8

/* Per-agent state and work lists: device pointers owned by the caller. */
typedef struct {
    uint32_t *hot;            /* [N] packed hot word, layout in reina_prims.h */
    reina_cold_t *cold;       /* [N] event-time fields */
    int32_t *infectees;       /* [N * REINA_INLINE_INFECTEES] inline infectee slots */
    int32_t *counters;        /* [REINA_COUNTER_WORDS] */
    int32_t *control;         /* [REINA_L_NR] */
    uint32_t *work_items;     /* [max_work_items * 4]: the second half holds the per-slice lists of symptom onsets
                                 (agent, word); max_work_items >= n_agents + 1024 */
    uint32_t *candidates;     /* [max_candidates * 4] (target, src, variant, prio); target 0xFFFFFFFF = hole.
                                 [0, max_work_items): per-slice regions; above: candidates realised from
                                 cross-shard pressure. max_candidates >= max_work_items + expected remote */
    uint32_t *queue0;         /* [max_queue] testing queue, even days */
    uint32_t *queue1;         /* [max_queue] testing queue, odd days */
    uint32_t *level1;         /* [max_queue] contact-tracing level-1 work list */
    uint64_t *hosp_events;    /* [REINA_HOSP_EVENT_WORDS(n_agents, max_hosp_events)]: the day's bed / ICU events by priority range */
    int32_t *pressure;        /* [REINA_EXCHANGE_WORDS(n_shards, hosp_ranges)] cross-shard infection pressure of the day:
                                 [dest shard][contact range][variant] = transmissible contacts aimed at
                                 agents of another shard. Filled by reina_step_day_begin, summed over
                                 shards by the caller (one all-reduce), consumed by reina_step_day_end */
    uint64_t *mirror;         /* [REINA_MAX_RANGES * REINA_MAX_VARIANTS * mirror_slots] sharded runs only:
                                 a day-tagged hash sample of this shard's OUTGOING cross-shard attempts,
                                 from which an incoming infection takes a local stand-in infector
                                 ("mirror attribution", reina_model_amd/sharding.py) */
    uint32_t *mirror_meta;    /* [2 * REINA_MIRROR_CELLS] sharded runs only: per cell the number of slots in use
                                 today (a power of two, sized from yesterday's traffic so the table stays
                                 dense and a lookup takes a handful of probes) and day + 1 of its last entry */
    uint32_t *work_counts;    /* [5 * REINA_MAX_SCAN_WAVES] entries per scanning-wave slice of the per-slice
                                 lists (no global append counters): exposure candidates, symptom onsets,
                                 hospital events, bookkeeping (written by the scan) and infection
                                 candidates (written by the contact kernel for the slice's sources) */
    uint32_t *scan_lists;     /* [4 * max_work_items] two lists of (agent, kind) pairs written by the
                                 scan: hospital events, then bookkeeping (R statistics, home
                                 recoveries / deaths). The other two lists live in work_items. */
    uint32_t *active_bits;    /* [REINA_BITS_WORDS(n_agents)] one bit per agent (bit i & 31 of word i >> 5): the hot word's ACTIVE
                                 flag again -- "the daily stream has something to do for this agent" (infected, or removed and
                                 not yet counted into R).  On a day with few such agents k_day streams these 1/32 of the hot
                                 words' bytes and fetches only the words of the agents whose bit is set (Context._iterate_people
                                 skips everybody else at once too: `if not person.is_infected: return`, main.pyx:1974-1975) */
    uint32_t *infected_bits;  /* [REINA_BITS_WORDS(n_agents)] one bit per agent: has ever been infected, i.e. is not SUSCEPTIBLE
                                 (person_expose's test, main.pyx:239).  A contact that can transmit looks its target up here --
                                 a table of N / 8 bytes that stays in the 256 MB Infinity Cache -- instead of gathering the
                                 target's hot word from HBM */
    uint64_t *xsend;          /* [REINA_XCHG_WORDS(n_shards, xchg_cap, hosp_ranges)] exact attribution: records bound for the other shards */
    uint64_t *xrecv;          /* [REINA_XCHG_WORDS(n_shards, xchg_cap, hosp_ranges)] ... and what the all-to-all brought from them */
    uint32_t *infectee_pool;  /* [2 * pool_cap] exact attribution: (infectee gid, next node or -1) of the infectees beyond an
                                 agent's inline slots -- an infectee may live on another shard, so the list cannot be threaded
                                 through the infectees' own records.  (Without exact attribution: any non-null pointers.) */
} reina_buffers_t;
/* words of a per-agent bit plane: whole 512-agent tiles of 16 words (k_day's wave tiles), and one spare tile */
#define REINA_BITS_WORDS(n_agents) ((((size_t)(n_agents) + 511u) / 512u + 1u) * 16u)

/* `pre_init` = 1 for batches that come from an `import-infections` intervention: the reference
 * applies those BEFORE Population.init_day zeroes new_infections / infected_by_variant
 * (main.pyx:2013-2016 then :1687-1699), so they do not show up in those daily counters; weekly
 * imports (infect_people_daily, :1671-1685) run after the zeroing and do. */
/* `testing_mode`: the mode in force when the batch was imported.  Interventions of one date are applied in list order
 * (main.pyx:2013-2015) and import-infections infects at once, so a `test-with-contact-tracing` LATER in the list of the
 * same date does not give these imports an infectee list (person_infect, main.pyx:226-232), and one EARLIER does even
 * if the day ends in another mode; weekly batches run after all interventions: the day's mode. */
typedef struct { uint32_t count; uint32_t variant; uint32_t pre_init; uint32_t testing_mode; } reina_import_batch_t;
/* vaccinate `nr` agents per day among sorted indices [idx_start, idx_end), oldest first
 * (HealthcareSystem.vaccinate_people main.pyx:560-583); `slot` keeps the device-side cursor */
typedef struct { uint32_t nr; uint32_t idx_start; uint32_t idx_end; uint32_t slot; } reina_vaccination_t;

/* Everything the host decides for one day (Context.iterate main.pyx:2011-2016 +
 * apply_intervention :1880-1960 + infect_people_daily :1671-1685). */
typedef struct {
    uint32_t day;
    uint32_t testing_mode;            /* TestingMode main.pyx:441-445 */
    float p_detected_anyway;
    float p_successful_tracing;
    int32_t add_beds;                 /* build-new-hospital-beds */
    int32_t add_icu_units;            /* build-new-icu-units */
    uint32_t n_import_batches;
    uint32_t n_vaccinations;
    reina_import_batch_t import_batches[REINA_MAX_IMPORT_BATCHES];
    reina_vaccination_t vaccinations[REINA_MAX_VACCINATIONS];
    int32_t *history_row;             /* dev, optional: counters are copied here BEFORE the day runs
                                         (= generate_state() taken before iterate(), simulation.py:195) */
} reina_day_t;

/* InitialPopulationCondition of the reference (calc/datasets.py:106-134) in the order
 * Population.set_initial_state walks it (main.pyx:1452-1516): slot i of [0, were_incubating) is
 * infected, then by rank: [0, incubating) stays incubating; the next recovered_without_illness
 * recover at once; everyone after falls ill, of whom the next `ill` stay ill at home, the next `dead`
 * die, the next `in_icu` are hospitalised and moved to ICU, the next `in_ward` hospitalised, the
 * rest recover.  Then all_detected[0..99] is reset and confirmed_cases are spread round-robin over
 * those ages.  (A sharded population gives every shard its share of each number.) */
typedef struct {
    uint32_t incubating, recovered_without_illness, ill, dead, in_icu, in_ward;
    uint32_t were_incubating;   /* total slots */
    uint32_t confirmed_cases;
    uint32_t confirmed_first;   /* shards: this shard owns confirmed cases first, first+stride, ... */
    uint32_t confirmed_stride;  /* 1 for an unsharded population */
} reina_initial_state_t;

typedef struct reina_engine reina_engine_t;
typedef struct reina_group reina_group_t;

/* replaces Context.__init__ / Population.__init__ / Disease.__init__ (main.pyx:1759-1781,
 * 1354-1450, 868-881) */
int reina_create(const reina_config_t *cfg, const reina_disease_t *disease, reina_engine_t **out);
int reina_destroy(reina_engine_t *e);
/* attach caller-owned state; reina_init_state fills it like _create_agents/_init_stats
 * (main.pyx:1389-1450): all susceptible, counters = age histogram, beds/ICU free */
int reina_bind_buffers(reina_engine_t *e, const reina_buffers_t *buffers);
int reina_init_state(reina_engine_t *e, int32_t hospital_beds, int32_t icu_units, void *stream);
/* replaces Population.set_initial_state (main.pyx:1452-1516); call once, right after
 * reina_init_state and before the first day.  Parallel form: every slot draws one uniform agent, with replacement
 * like the reference; an agent drawn by several slots is visited by them in slot order (each visit moves the counters,
 * the last one decides what the agent is); beds and ICU units are granted in slot order.  Precondition (the caller's to
 * check, for the whole population when the engine is a shard): if `in_icu` > 0 the hospital has at least one bed -- the
 * reference raises AssertionError out of Context.__init__ otherwise (main.pyx:1495 -> :350 -> :1603), and so does
 * reina_model_amd.model.Context. */
int reina_set_initial_state(reina_engine_t *e, const reina_initial_state_t *ic, void *stream);
/* replaces ContactMatrix.generate_contact_probabilities upload (main.pyx:1184-1235) */
int reina_upload_contact_tables(reina_engine_t *e, const reina_contact_tables_t *t, void *stream);
/* replaces Context.iterate() for one day (main.pyx:2011-2018) */
int reina_step_day(reina_engine_t *e, const reina_day_t *day, void *stream);
/* In-stream collective for a sharded population: `allreduce` has the signature of RCCL's
 * ncclAllReduce (sendbuff, recvbuff, count, datatype, op, comm, stream) and is called by
 * reina_step_day / reina_run_days* between the two halves of every day as
 * allreduce(pressure, pressure, REINA_EXCHANGE_WORDS(n_shards, hosp_ranges), 2 = ncclInt32, 0 = ncclSum, comm, stream), i.e.
 * on the day stream itself: no host round trip, no second stream.  The library does not link RCCL;
 * the caller hands in the function and its communicator (NULL, NULL switches it off). */
typedef int (*reina_allreduce_fn)(const void *sendbuff, void *recvbuff, size_t count, int datatype, int op,
                                  void *comm, void *stream);
int reina_set_collective(reina_engine_t *e, reina_allreduce_fn allreduce, void *comm);
/* exact attribution: `alltoall` has the signature of RCCL's ncclAllToAll (sendbuff, recvbuff, count per peer, datatype, comm,
 * stream) and is called as alltoall(xsend, xrecv, REINA_XCHG_SEG_WORDS(xchg_cap, hosp_ranges), 4 = ncclInt64, comm, stream) on the day stream wherever
 * reina_step_phase asks for REINA_X_ALLTOALL. */
typedef int (*reina_alltoall_fn)(const void *sendbuff, void *recvbuff, size_t count, int datatype, void *comm, void *stream);
int reina_set_alltoall(reina_engine_t *e, reina_alltoall_fn alltoall, void *comm);
/* One day as the phases between which a sharded population exchanges: reina_step_day == the phases in order with the
 * collectives they return queued in between.  Returns < 0 on error, else the collectives the caller must run before the next
 * phase: REINA_X_ALLREDUCE = sum buffers.pressure over the shards (REINA_EXCHANGE_WORDS int32), REINA_X_ALLTOALL = segment d of
 * every shard's buffers.xsend -> segment (sender's rank) of shard d's buffers.xrecv.
 *   REINA_PH_OPEN      the day's opening: snapshot, imports, test queue, level-0 tracing   -> ALLTOALL (exact, tracing days)
 *   REINA_PH_TRACE     exact, tracing days: the received level-0 requests, level-1 tracing -> ALLTOALL (exact, tracing days)
 *   REINA_PH_MAIN      (the received level-1 requests,) vaccination, the stream + contacts  -> ALLREDUCE (mirror attribution) or ALLTOALL
 *                      (exact: the segments' trailers carry what the all-reduce would -- ABI 7)
 *   REINA_PH_END       cross-shard contacts claim, bed / ICU events, installs                -> ALLTOALL (exact)
 *   REINA_PH_FEEDBACK  exact: the sources of cross-shard infections take their infectees
 * reina_step_day_begin == OPEN + TRACE + MAIN and reina_step_day_end == END + FEEDBACK for a population WITHOUT exact attribution. */
enum { REINA_PH_OPEN = 0, REINA_PH_TRACE, REINA_PH_MAIN, REINA_PH_END, REINA_PH_FEEDBACK, REINA_PH_NR };
#define REINA_X_ALLREDUCE 1
#define REINA_X_ALLTOALL 2
int reina_step_phase(reina_engine_t *e, const reina_day_t *day, int phase, void *stream);
/* the same day in two halves for a sharded population: `begin` runs everything up to and including
 * contact sampling and leaves this shard's outgoing pressure (and its free capacity / demand words) in
 * buffers.pressure; the caller sums `pressure` over all shards (ncclAllReduce / torch.distributed.all_reduce,
 * the ONLY per-day collective); `end` takes this shard's share of the pooled beds / ICU units, realises the
 * pressure aimed at this shard, walks the bed / ICU events and installs the day's infections.
 * reina_step_day == begin + end (with n_shards == 1 nothing is exchanged). */
int reina_step_day_begin(reina_engine_t *e, const reina_day_t *day, void *stream);
int reina_step_day_end(reina_engine_t *e, const reina_day_t *day, void *stream);
/* runs `n_days` consecutive days from an array of day descriptors (the loop of
 * calc/simulation.py:194-270 without the per-day host round trip) */
int reina_run_days(reina_engine_t *e, const reina_day_t *days, uint32_t n_days, void *stream);
/* the same with the history rows laid out by the library: day k snapshots its counters to
 * history_base + k * REINA_COUNTER_WORDS (dev pointer, may be NULL); days[k].history_row is
 * ignored, so ONE descriptor array can drive many engine instances (Monte-Carlo ensembles) */
int reina_run_days_hist(reina_engine_t *e, const reina_day_t *days, uint32_t n_days, int32_t *history_base,
                        void *stream);
/* Monte-Carlo ensembles (the reference: calc/simulation.py:349-385, a process pool over seeds):
 * a group of identically configured, unsharded engines that differ in their seed is stepped with
 * ONE launch per phase for all members (member = blockIdx.y), so small populations still fill the
 * chip.  history_bases: host array of n device pointers (or NULL); member m's row of day k is
 * history_bases[m] + k * REINA_COUNTER_WORDS. */
int reina_group_create(reina_engine_t **engines, uint32_t n, reina_group_t **out);
int reina_group_destroy(reina_group_t *g);
int reina_group_upload_contact_tables(reina_group_t *g, const reina_contact_tables_t *t, void *stream);
int reina_group_run_days(reina_group_t *g, const reina_day_t *days, uint32_t n_days, int32_t *const *history_bases,
                         void *stream);
/* replaces Context.generate_state's reads (main.pyx:1813-1857): copies the counter block to host
 * (synchronises `stream`) */
int reina_read_counters(reina_engine_t *e, int32_t *out_host, void *stream);
/* the end of a run (calc/simulation.py:194-290: the rows the loop collected, then the final state): copies the `n_rows`
 * history rows at `history_dev` (as written by reina_run_days_hist) and, behind them as row n_rows, the counter block as it
 * stands, to `out_host` [(n_rows + 1) * REINA_COUNTER_WORDS] -- page-locked memory makes it one DMA --, and synchronises
 * `stream`: one call, two copies, one wait */
int reina_read_history(reina_engine_t *e, const int32_t *history_dev, uint32_t n_rows, int32_t *out_host, void *stream);
/* timing hooks for bench.py: HIP events on the launch stream around the day's kernels (start / stop
 * timestamps of the kernel's own dispatch packet).
 * enable: 0 off; 1 every kernel of every day; k >= 4: one KIND of kernel per profiled day, the kinds taking
 * turns -- k_day (the stream + contact sampling) on days with day % k == 0, k_open (+ the occasional kernels:
 * level-1 tracing, vaccination) at k/4, the event-walk launches (a sharded population's, a large population's
 * k_hosp_sort / k_hosp_walk) and the cross-shard realisation at k/2, k_hosp_install at 3k/4 -- so that the cost of timestamped dispatches (a few
 * microseconds each, which matters when a whole day takes 40) stays small; k <= -4: stride -k, and ONLY k_day is timed (the
 * dominant kernel of a short window: a third of the timestamped dispatches).
 * reina_profile_read_kernels: summed milliseconds and launch counts per kind since the last read, arrays of
 * REINA_PK_NR; synchronises the device.  reina_profile_read: the k_day pair of those numbers and the sum
 * over all kinds. */
enum {
    REINA_PK_OPEN = 0, REINA_PK_TRACE1, REINA_PK_VACCINATE, REINA_PK_DAY, REINA_PK_HOSPITAL, REINA_PK_HOSP_SORT,
    REINA_PK_HOSP_WALK, REINA_PK_REMOTE, REINA_PK_INSTALL, REINA_PK_XCHG /* exact attribution: the kernels that take in exchanged records */,
    REINA_PK_COLLECTIVE /* the in-stream collectives of a sharded day (all-reduce, all-to-alls), event pairs recorded around them */,
    REINA_PK_SMALL_DAY /* round 6, ABI 6: the one launch of a small unsharded population's whole day (k_small_day) */, REINA_PK_NR
};
int reina_profile_enable(reina_engine_t *e, int enable);
int reina_profile_read_kernels(reina_engine_t *e, double *ms_total, uint64_t *launches);
int reina_profile_read(reina_engine_t *e, double *scan_ms_total, uint64_t *scan_launches,
                       double *all_ms_total);
/* replaces Context.sample(what, age, severity) (main.pyx:2047-2101): n draws of one per-agent
 * quantity with the engine's samplers; host-only, needs no engine and no GPU. `what`: 0
 * contacts_per_day, 1 symptom_severity, 2 incubation_period, 3 illness_period,
 * 4 hospitalization_period, 5 icu_period, 6 onset_to_removed_period; severity < 0 = None */
int reina_sample(const reina_disease_t *disease, uint64_t seed, int what, int age, int severity,
                 float nr_contacts_of_age, int n, int32_t *out);
/* replaces ContactMatrix.generate_contact_probabilities (main.pyx:1184-1235; pandas in the reference, run
 * by init_day :1285-1288 whenever a mobility limitation changes) for a matrix whose ages all have E
 * entries, and packs the thresholds of reina_contact_tables_t; host-only, needs no engine and no GPU.
 * base / row_page / row_place: the long-form contact rows (contacts per day, participant age, place);
 * mobility: n_mobility x (place or -1, min_age, max_age, factor) as doubles, applied in order;
 * rows_mat / sorted_mat [A*E]: each age's rows in summation order / in table order.
 * Out: totals [A] (nr_contacts_by_age), cum [A*E] (cumulative probabilities), and, if not NULL,
 * nrc [A] (float) and thr [A*thr_stride] (uint32 thresholds, entries past E untouched). */
int reina_build_contact_tables(const double *base, const int32_t *row_page, const int32_t *row_place, uint32_t n_rows,
                               const double *mobility, uint32_t n_mobility, const int32_t *rows_mat,
                               const int32_t *sorted_mat, uint32_t n_ages, uint32_t n_entries, double *totals_out,
                               double *cum_out, float *nrc_out, uint32_t *thr_out, uint32_t thr_stride);
/* TEST HOOK, no reference counterpart: evaluates one numeric primitive of the day step (csrc/reina_prims.h) for n input
 * records ON THE DEVICE, one lane per record, so that the device build of Philox4x32 / Philox2x32 / inverse normal / exp /
 * log / gamma / the contact-count draw can be checked against published known answers and against the host build bit for
 * bit (tests/test_prims_gpu.py).  Host pointers, synchronous.  Words per record (in -> out): REINA_TP_PHILOX4 (k0, k1, c0,
 * c1, c2, c3) -> 4; REINA_TP_PHILOX2 (key, c0, c1) -> 2; REINA_TP_NORMAL (32-bit draw) -> float bits; REINA_TP_EXPF /
 * REINA_TP_LOGF (float bits) -> float bits; REINA_TP_GAMMA (mu bits, cv bits, k0, k1, who, day, purpose, first block) ->
 * float bits; REINA_TP_COUNT_DRAW (k0, k1, who, day) -> the 32-bit word the contact count is inverted from. */
enum { REINA_TP_PHILOX4 = 0, REINA_TP_PHILOX2, REINA_TP_NORMAL, REINA_TP_EXPF, REINA_TP_LOGF, REINA_TP_GAMMA, REINA_TP_COUNT_DRAW,
       REINA_TP_NR };
int reina_test_prims(int what, const uint32_t *in_host, uint32_t n, uint32_t *out_host);
const char *reina_last_error(void);
int reina_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* REINA_HIP_H */
