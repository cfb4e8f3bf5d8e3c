// reina_prims.h -- numeric primitives of the parallel day step.
//
// Compiled by hipcc (device code, gfx950) AND by gcc (oracle/reina_par.c, the CPU checker).
// Everything here is built from IEEE-754 basic operations only (+ - * / sqrt, integer ops) with
// FP contraction disabled on both compilers, so a value computed on a CDNA4 lane and on a host
// core is bit-identical by construction: no libm, no native exp/log, no fma.
//
// The parallel engine cannot consume the reference's single PCG64 stream in scan order
// (cythonsim/simrandom.pyx:13-55 is inherently sequential), so every random decision is drawn
// from Philox4x32-10 keyed by (seed) and counted by (who, day, purpose, block): the value a
// decision sees is independent of thread scheduling, shard layout and launch geometry.
#ifndef REINA_PRIMS_H
#define REINA_PRIMS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RP_HD __host__ __device__ __forceinline__
#else
#define RP_HD static inline
#endif

// ------------------------------------------------------------------ bit casts
RP_HD float rp_u2f(uint32_t u) {
    union { uint32_t u; float f; } c;
    c.u = u;
    return c.f;
}
RP_HD uint32_t rp_f2u(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    return c.u;
}

// ------------------------------------------------------------------ Philox4x32-10
typedef struct { uint32_t v[4]; } rp_u4;

#define RP_PHILOX_M0 0xD2511F53u
#define RP_PHILOX_M1 0xCD9E8D57u
#define RP_PHILOX_W0 0x9E3779B9u
#define RP_PHILOX_W1 0xBB67AE85u

// a ^ b ^ c: on the GPU one instruction (gfx950: v_bitop3_b32, truth table 0x96) -- the compiler keeps two v_xor_b32 when one
// operand is a round key in a scalar register, a third of a Philox2x32 round's vector instructions
#if defined(__HIP_DEVICE_COMPILE__)
#define RP_XOR3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)
#else
#define RP_XOR3(a, b, c) ((a) ^ (b) ^ (c))
#endif
RP_HD rp_u4 rp_philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)RP_PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)RP_PHILOX_M1 * c2;
        uint32_t n0 = RP_XOR3((uint32_t)(p1 >> 32), c1, k0);
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = RP_XOR3((uint32_t)(p0 >> 32), c3, k1);
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += RP_PHILOX_W0;
        k1 += RP_PHILOX_W1;
    }
    rp_u4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// ------------------------------------------------------------------ Philox2x32-10
// (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 philox2x32_R(10, ..):
// 64-bit counter (c0, c1), one 32-bit key, one 32 x 32 -> 64 multiply per round instead of two.)  The contact draws
// are nine tenths of all random numbers of a busy day and need two words each until a contact survives the thinning
// (place / age range, transmission), two more after (target, mask): two words per block is their natural size.
typedef struct { uint32_t v[2]; } rp_u2;
#define RP_PHILOX2_M 0xD256D193u
RP_HD rp_u2 rp_philox2(uint32_t key, uint32_t c0, uint32_t c1) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < 10; r++) {
        uint64_t p = (uint64_t)RP_PHILOX2_M * c0;
        c0 = RP_XOR3((uint32_t)(p >> 32), c1, key);
        c1 = (uint32_t)p;
        key += RP_PHILOX_W0;
    }
    rp_u2 o;
    o.v[0] = c0; o.v[1] = c1;
    return o;
}
// contact draws of source `src` on `day`: key from the engine's Philox key, counter (src, day | contact # << 12 | half << 20);
// half 0 -> (place / age-range draw, transmission draw), half 1 -> (shard + target draw, mask draw)
RP_HD uint32_t rp_contact_key(uint32_t k0, uint32_t k1) { return k0 ^ (k1 * 0x9E3779B9u + 0x7F4A7C15u); }
RP_HD uint32_t rp_contact_ctr(uint32_t day, uint32_t c, uint32_t half) { return (day & 0xFFFu) | (c << 12) | (half << 20); }

// the draw behind an infectious agent's contact COUNT of the day (inverted through the count thresholds,
// reina_contacts.h: rc_count_thresholds): its own key, counter (agent, day)
// (never 0xFFFFFFFF: that value is the thresholds' "never" -- a count row of an age without contacts is all 0xFFFFFFFF, and
// the searches test r >= threshold -- so the one draw in 2^32 that would pass it is folded onto its neighbour)
// (keyed: with rp_contact_key(k0, k1) already at hand)
RP_HD uint32_t rp_count_draw_keyed(uint32_t contact_key, uint32_t who, uint32_t day) {
    const uint32_t r = rp_philox2(contact_key ^ 0x6A09E667u, who, day).v[0];
    return r == 0xFFFFFFFFu ? 0xFFFFFFFEu : r;
}
RP_HD uint32_t rp_count_draw(uint32_t k0, uint32_t k1, uint32_t who, uint32_t day) {
    return rp_count_draw_keyed(rp_contact_key(k0, k1), who, day);
}

// Purposes (counter word c2 low byte). The sub-index (contact number, import try, tracer id)
// goes in c3 or the upper bits of c2; `who` (agent / event id) in c0, day in c1.
enum {
    RP_P_NRCONTACTS = 1,  // (unused since the contact count is drawn by inversion: rp_count_draw)
    RP_P_CONTACT = 2,     // (unused since the contact draws moved to Philox2x32: rp_contact_key / rp_contact_ctr)
    RP_P_INFECT = 3,      // (target, day, c3 = block): severity, incubation gamma
    RP_P_ONSET = 4,       // (agent, day, c3 = block): onset->removed gamma; block 0 v[3] = "tested anyway"
    RP_P_HOSPITAL = 5,    // (agent, day): no-bed / no-ICU death roll
    RP_P_TRACE = 6,       // (candidate, day, c3 = tracer): contact-tracing success
    RP_P_IMPORT = 7,      // (import #, day, c3 = try): age class, target
    RP_P_PRIORITY = 8,    // (agent, day): order key for scarce resources / winner selection
    RP_P_REMOTE = 9,      // (attempt #, day, c3 = range | variant << 8): realisation of cross-shard pressure
    RP_P_MIRROR = 10,     // (agent, day, c3 = contact #): slot + tie-break of an outgoing attempt in the mirror table
    RP_P_INITIAL = 11     // (slot, RP_INIT_DAY, c3 = try): agent of an initial-condition slot
};

// uniform in [0,1) with 24 random bits: exact in float
RP_HD float rp_uniform24(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// RandomPool.chance semantics (simrandom.pyx:32-39): p==1 -> true, p==0 -> false without
// looking at the draw, else u < p.
RP_HD int rp_chance(float p, uint32_t r) {
    if (p == 1.0f) return 1;
    if (p == 0.0f) return 0;
    return rp_uniform24(r) < p;
}
// The same decision as one integer comparison: rp_chance(p, r) == ((r >> 8) < rp_chance_threshold(p)) for every p and r.
// p <= 0, -0 or NaN: never (threshold 0); p >= 1: always (2^24, above every 24-bit draw); between: u x 2^-24 < p is
// u < p x 2^24 (a power-of-two scaling, exact), i.e. u < ceil(p x 2^24) for an integer u.  For a probability that is tested
// against many draws (k_day: the thinning bound of a source, once per contact).  tests/test_prims.py checks the identity.
RP_HD uint32_t rp_chance_threshold(float p) {
    if (!(p > 0.0f)) return 0u;
    if (p >= 1.0f) return 1u << 24;
    const float s = p * 16777216.0f;
    const uint32_t f = (uint32_t)s;
    return (float)f < s ? f + 1u : f;
}

// ------------------------------------------------------------------ exp / log (float, reproducible)
// expf for |x| <= ~80. n = round(x*log2e); r = x - n*ln2 (two-term Cody-Waite); degree-6 poly.
RP_HD float rp_expf(float x) {
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    float t = x * 1.44269504088896341f;
    // round to nearest via floor(t + 0.5)
    float tf = t + 0.5f;
    int n = (int)tf;
    if ((float)n > tf) n -= 1;  // floor for negatives
    float fn = (float)n;
    float r = x - fn * 0.693359375f;          // ln2_hi (exact in 10 bits)
    r = r - fn * -2.12194440e-4f;             // ln2_lo
    // e^r ~ 1 + r + r^2/2 + ... (|r| <= 0.3466)
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    float r2 = r * r;
    float e = p * r2 + r + 1.0f;
    // scale by 2^n (n in [-126,127] after the clamps above)
    return e * rp_u2f((uint32_t)(n + 127) << 23);
}

// logf for finite x > 0 (normal numbers). m in [sqrt(1/2), sqrt(2)); log m = 2 atanh((m-1)/(m+1)).
RP_HD float rp_logf(float x) {
    uint32_t ux = rp_f2u(x);
    int e = (int)(ux >> 23) - 127;
    uint32_t mant = (ux & 0x007FFFFFu) | 0x3F800000u;
    float m = rp_u2f(mant);
    if (m > 1.41421356f) {
        m = m * 0.5f;
        e += 1;
    }
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    float p = 0.11111111f;          // 1/9
    p = p * z + 0.14285714f;        // 1/7
    p = p * z + 0.2f;               // 1/5
    p = p * z + 0.33333333f;        // 1/3
    p = p * z + 1.0f;
    float lm = 2.0f * s * p;
    float fe = (float)e;
    return (fe * 0.693359375f + lm) + fe * -2.12194440e-4f;
}

// ------------------------------------------------------------------ standard normal from one u32
// Inverse CDF by M. J. Wichura's algorithm AS 241, routine PPND7 (Appl. Statist. 37, 1988: "about seven decimal digits"),
// the single-precision member of the pair -- rational functions with small positive coefficients that evaluate well in float.
// u is the centre of one of 2^32 equal cells of (0,1): never 0 or 1; a tail is computed on the exact cell of its own side, so
// the two tails mirror each other bit for bit.  (Rounds 1-2 used P. J. Acklam's approximation, whose large alternating
// coefficients needed double arithmetic: 650 SIMD cycles per wave call against 270 -- the draw sits inside the gamma draws of
// every infection and every symptom onset, a quarter of the day's last launch at an epidemic peak.)
RP_HD float rp_normal_from_u32(uint32_t r) {
    // p = (r + 0.5) / 2^32 computed in two exact steps: hi 24 bits + low 8 bits
    const float p = ((float)(r >> 8) + ((float)(r & 0xFFu) + 0.5f) * (1.0f / 256.0f)) * (1.0f / 16777216.0f);
    const float q = p - 0.5f;
    if (q >= -0.425f && q <= 0.425f) {
        const float t = 0.180625f - q * q;
        return q * (((5.9109374720e+01f * t + 1.5929113202e+02f) * t + 5.0434271938e+01f) * t + 3.3871327179e+00f) /
               (((6.7187563600e+01f * t + 7.8757757664e+01f) * t + 1.7895169469e+01f) * t + 1.0f);
    }
    const int upper = q > 0.0f;
    const uint32_t rc = upper ? 0xFFFFFFFFu - r : r;   // the cell counted from the nearer end
    const float pc = ((float)(rc >> 8) + ((float)(rc & 0xFFu) + 0.5f) * (1.0f / 256.0f)) * (1.0f / 16777216.0f);
    float s = sqrtf(-rp_logf(pc));
    float x;
    if (s <= 5.0f) {
        s = s - 1.6f;
        x = (((1.7023821103e-01f * s + 1.3067284816e+00f) * s + 2.7568153900e+00f) * s + 1.4234372777e+00f) /
            ((1.2021132975e-01f * s + 7.3700164250e-01f) * s + 1.0f);
    } else {
        s = s - 5.0f;
        x = (((1.7337203997e-02f * s + 4.2868294337e-01f) * s + 3.0812263860e+00f) * s + 6.6579051150e+00f) /
            ((1.2258202635e-02f * s + 2.4197894225e-01f) * s + 1.0f);
    }
    return upper ? x : -x;
}

// ------------------------------------------------------------------ gamma (Marsaglia-Tsang, shape >= 1)
// RandomPool.gamma(mu, cv) (simrandom.pyx:46-55): sigma = cv*mu, theta = sigma^2/mu,
// kappa = mu/theta; returns theta * Gamma(kappa).  Same construction numpy's random_gamma_f uses
// (Marsaglia & Tsang 2000), on Philox blocks: block k (k = first_block, first_block+1, ...)
// supplies two (normal, uniform) attempts.  Bounded at 32 blocks (P(reject 64x) < 1e-80).
RP_HD float rp_gamma_mu_cv(float mu, float cv, uint32_t k0, uint32_t k1, uint32_t who, uint32_t day,
                           uint32_t purpose, uint32_t first_block) {
    float sigma = cv * mu;
    float theta = (sigma * sigma) / mu;
    float kappa = mu / theta;
    float b = kappa - 1.0f / 3.0f;
    float c = 1.0f / sqrtf(9.0f * b);
    for (uint32_t blk = 0; blk < 32u; blk++) {
        rp_u4 r = rp_philox(k0, k1, who, day, purpose, first_block + blk);
        for (int h = 0; h < 2; h++) {
            float X = rp_normal_from_u32(r.v[2 * h]);
            float V = 1.0f + c * X;
            if (V <= 0.0f) continue;
            V = V * V * V;
            float U = rp_uniform24(r.v[2 * h + 1]);
            float X2 = X * X;
            if (U < 1.0f - 0.0331f * X2 * X2) return theta * (b * V);
            if (U > 0.0f && rp_logf(U) < 0.5f * X2 + b * (1.0f - V + rp_logf(V))) return theta * (b * V);
        }
    }
    return theta * b;  // unreachable in practice
}

// round_to_int of the reference (main.pyx:773-774): (int)(f + 0.5)
RP_HD int rp_round_to_int(float f) { return (int)(f + 0.5f); }

// ------------------------------------------------------------------ hot word (4 B per agent)
//  bits  0-2  state            (main.pyx:41-48)
//  bits  3-5  symptom severity (main.pyx:33-38)
//  bit   6    was_detected
//  bit   7    queued_for_testing
//  bits  8-9  variant_idx
//  bit   10   included_in_totals
//  bit   11   place_of_death == DEATH_OUTSIDE_HOSPITAL
//  bit   12   (free)
//  bit   13   vaccinated (day_of_vaccination >= 0)
//  bit   14   has infectee list (infected while contact tracing was on, main.pyx:227-233)
//  bit   15   active: needs the day's state machine (infected, or removed but not yet counted into R): set when
//             the agent is infected, cleared when its removal has been counted -- the streaming pass tests this bit only
//  bits 16-23 days_left, as the absolute day (mod 256) whose scan finds it at 0 (see RH_DAYS_LEFT below)
//  bits 24-31 ILLNESS and later: day_of_illness, as the absolute day (mod 256) whose scan finds it at 0;
//             INCUBATION: the day of infection (mod 256) -- an agent infected before today's scan (an import) sits that
//             scan out (day_of_infection == today, main.pyx:402): the scan compares this field with today
#define RH_STATE(w) ((w) & 7u)
#define RH_SEV(w) (((w) >> 3) & 7u)
#define RH_DETECTED 0x40u
#define RH_QUEUED 0x80u
#define RH_VARIANT(w) (((w) >> 8) & 3u)
#define RH_INCLUDED 0x400u
#define RH_POD_OUTSIDE 0x800u
#define RH_VACCINATED 0x2000u
#define RH_HASLIST 0x4000u
#define RH_ACTIVE 0x8000u
#define RH_SET_STATE(w, s) (((w) & ~7u) | (uint32_t)(s))
// days_left / day_of_illness are kept as ABSOLUTE days (mod 256), so the word of an agent that is only
// waiting does not change from one day to the next (no daily write-back): bits 16-23 = the day whose
// scan finds days_left == 0 before its own decrement, bits 24-31 = the day whose scan finds
// day_of_illness == 0.  A countdown never sits at 0 for a second scan (every state leaves on the day
// it gets there: person_advance main.pyx:395-438), so the difference mod 256 is the countdown itself,
// 0..255 as in the counting form.
#define RH_MAX_DAYS 255
// the countdown as the scan of `day` finds it (before that scan's decrement)
#define RH_DAYS_LEFT(w, day) ((uint32_t)((((w) >> 16) - (day)) & 0xFFu))
#define RH_DOI(w, day) ((uint32_t)(((day) - ((w) >> 24)) & 0xFFu))
// written by code that runs on `day` (after that day's scan, before it for a FRESH import, RP_INIT_DAY
// = -1 while the initial condition is applied): the NEXT day's scan finds days_left == d / day_of_illness == 0
#define RH_DAYS_FIELD(d, day) ((((uint32_t)(day) + 1u + (uint32_t)(d)) & 0xFFu) << 16)
#define RH_SET_DAYS_LEFT(w, d, day) (((w) & ~0x00FF0000u) | RH_DAYS_FIELD(d, day))
#define RH_SET_DOI0(w, day) (((w) & 0x00FFFFFFu) | ((((uint32_t)(day) + 1u) & 0xFFu) << 24))
// INCUBATION: infected on `day` (an agent of the initial condition that stays incubating: counted as infected on day 0,
// whose scan it sits out)
#define RH_INFECTED_ON(day) (((uint32_t)(day) & 0xFFu) << 24)
#define RH_INFECTED_TODAY(w, day) (RH_STATE(w) == RS_INCUBATION && ((w) >> 24) == ((uint32_t)(day) & 0xFFu))

enum { RS_SUSCEPTIBLE = 0, RS_INCUBATION, RS_ILLNESS, RS_HOSPITALIZED, RS_IN_ICU, RS_RECOVERED, RS_DEAD };
enum { RV_ASYMPTOMATIC = 0, RV_MILD, RV_SEVERE, RV_CRITICAL, RV_FATAL };
// Disease.get_hospitalization_days / get_icu_days (main.pyx:1016-1039), before rounding: the ward / ICU stay of an agent
// of severity `sev` whose onset-to-removal time is `od`.  ASYMPTOMATIC and MILD agents get 0 -- the day loop never
// hospitalises them, but Population.set_initial_state (main.pyx:1452-1516) puts agents of ANY severity into ward and
// ICU, and those leave again on the first day.
RP_HD float rp_ward_stay(int sev, float od, float ratio_before_hospitalisation, float ratio_in_ward) {
    if (sev == RV_SEVERE) return od * (1.0f - ratio_before_hospitalisation);
    if (sev == RV_FATAL || sev == RV_CRITICAL) return od * ratio_in_ward;
    return 0.0f;
}
RP_HD float rp_icu_stay(int sev, float od, float ratio_before_hospitalisation, float ratio_in_ward) {
    if (sev == RV_FATAL || sev == RV_CRITICAL) {
        float f = 1.0f - ratio_in_ward - ratio_before_hospitalisation;
        f *= od;
        return f;
    }
    return 0.0f;
}
enum { RT_NO_TESTING = 0, RT_ALL_WITH_SYMPTOMS_CT, RT_ALL_WITH_SYMPTOMS, RT_ONLY_SEVERE_SYMPTOMS };

// "day" of every draw made while the initial population condition is applied (before day 0)
#define RP_INIT_DAY 0xFFFFFFFFu

// claim / ordering key: smaller wins. [4095-day : 12][priority : 20][id : 32]
RP_HD uint64_t rp_order_key(uint32_t day, uint32_t prio20, uint32_t id) {
    return ((uint64_t)((4095u - day) & 0xFFFu) << 52) | ((uint64_t)(prio20 & 0xFFFFFu) << 32) | id;
}
RP_HD uint32_t rp_priority20(uint32_t k0, uint32_t k1, uint32_t who, uint32_t day) {
    return rp_philox(k0, k1, who, day, RP_P_PRIORITY, 0).v[0] >> 12;
}

// Sharded populations: every shard derives its own Philox key from the common seed.
RP_HD uint64_t rp_shard_seed(uint64_t seed, uint32_t rank) {
    return seed + (uint64_t)rank * 0x9E3779B97F4A7C15ull;
}
// ------------------------------------------------------------------ saturating maps of the bed / ICU walk
// Every bed / ICU event acts on a free count x as f(x) = max(x + a, m): an admission (a = -1, m = 0), a release (a = +1,
// m = "never binds").  Such maps compose to the same form -- (a1, m1) then (a2, m2) = (a1 + a2, max(m1 + a2, m2)) -- so
// the free count in front of any event of an ORDERED walk is a prefix composition (k_hospital.inc), and a whole bucket of
// events is ONE map: what a sharded population's shards exchange (include/reina_hip.h: REINA_EXCHANGE_WORDS).
typedef struct { int a, m; } rp_sat_t;
#define RP_SAT_NEG (-(1 << 29))
RP_HD rp_sat_t rp_sat_id(void) {
    rp_sat_t f;
    f.a = 0;
    f.m = RP_SAT_NEG;
    return f;
}
RP_HD rp_sat_t rp_sat_then(rp_sat_t f, rp_sat_t g) {   // f first, then g
    rp_sat_t r;
    r.a = f.a + g.a;
    int t = f.m + g.a;
    if (t < RP_SAT_NEG) t = RP_SAT_NEG;
    r.m = t > g.m ? t : g.m;
    return r;
}
RP_HD int rp_sat_apply(rp_sat_t f, int x) {
    int y = x + f.a;
    return y > f.m ? y : f.m;
}
// the bed map and the ICU map of a bucket in 57 bits: a and m of either lie in [-4096, 4096] (a bucket holds at most 4096
// keys), m may also be "never binds" (<= RP_SAT_NEG / 2); bit 63 = present (a zero word reads as "no events": identity)
RP_HD uint64_t rp_sat_pack(rp_sat_t fb, rp_sat_t fc) {
    const uint64_t ba = (uint64_t)(fb.a + 8192) & 0x3FFFu, ca = (uint64_t)(fc.a + 8192) & 0x3FFFu;
    const uint64_t bm = fb.m <= RP_SAT_NEG / 2 ? 0x3FFFull : ((uint64_t)(fb.m + 8192) & 0x3FFFu);
    const uint64_t cm = fc.m <= RP_SAT_NEG / 2 ? 0x3FFFull : ((uint64_t)(fc.m + 8192) & 0x3FFFu);
    return (1ull << 63) | ba | (bm << 14) | (ca << 28) | (cm << 42);
}
RP_HD void rp_sat_unpack(uint64_t v, rp_sat_t *fb, rp_sat_t *fc) {
    if (!(v >> 63)) {
        *fb = rp_sat_id();
        *fc = rp_sat_id();
        return;
    }
    const uint32_t bm = (uint32_t)(v >> 14) & 0x3FFFu, cm = (uint32_t)(v >> 42) & 0x3FFFu;
    fb->a = (int)((uint32_t)v & 0x3FFFu) - 8192;
    fb->m = bm == 0x3FFFu ? RP_SAT_NEG : (int)bm - 8192;
    fc->a = (int)((uint32_t)(v >> 28) & 0x3FFFu) - 8192;
    fc->m = cm == 0x3FFFu ? RP_SAT_NEG : (int)cm - 8192;
}

// ------------------------------------------------------------------ exact cross-shard attribution (include/reina_hip.h)
// A global id: [shard : 4][index : 27], non-negative as an int32.  `gid_base` = shard << 27 of the engine's own shard when its
// population keeps global ids (exact attribution), 0 otherwise; `gid_mask` = the index bits then, all ones otherwise -- so the
// same three expressions serve a population that keeps plain local indices (unsharded, or mirror attribution).
#define RP_GID_SHIFT 27
#define RP_GID_INDEX_MASK ((1u << RP_GID_SHIFT) - 1u)
RP_HD uint32_t rp_gid_is_local(uint32_t gid, uint32_t gid_base, uint32_t gid_mask) { return (gid & ~gid_mask) == gid_base; }
RP_HD uint32_t rp_gid_shard(uint32_t gid) { return gid >> RP_GID_SHIFT; }
// One 8-byte record of an exchange segment: `index` at the RECEIVING shard (27 bits), 5 bits of flags, a global id (31 bits):
//   contact record   (target, flags = variant | source-keeps-a-list << 2, source gid)   source's shard -> target's
//   feedback record  (source, flags = source-keeps-a-list << 2, infectee gid)            target's shard -> source's
//   tracing request  (candidate, flags = 0, tracer gid)                                   tracer's shard -> candidate's
RP_HD uint64_t rp_xrec(uint32_t index, uint32_t flags, uint32_t gid) {
    return ((uint64_t)gid << 32) | ((uint64_t)(flags & 31u) << RP_GID_SHIFT) | (uint64_t)(index & RP_GID_INDEX_MASK);
}
RP_HD uint32_t rp_xrec_index(uint64_t r) { return (uint32_t)r & RP_GID_INDEX_MASK; }
RP_HD uint32_t rp_xrec_flags(uint64_t r) { return ((uint32_t)r >> RP_GID_SHIFT) & 31u; }
RP_HD uint32_t rp_xrec_gid(uint64_t r) { return (uint32_t)(r >> 32); }

// candidate "source id" of an infection realised from cross-shard pressure (no local infector)
#define RP_REMOTE_SRC 0x80000000u
#define RP_MIRROR_PROBES 16u      // probes per cell of the mirror table (tables are kept >= ~40 % full)
#define RP_MIRROR_MIN_SLOTS 8u    // smallest table: fully scanned by one lookup
#define RP_MIRROR_STALE_DAYS 14u   // a stand-in infector may be an agent that aimed at another shard up to this many days ago

// ------------------------------------------------------------------ test hook (reina_test_prims / par_test_prims)
// One record of one primitive, the same code path on a CDNA4 lane and on a host core: what the device-side known-answer
// tests evaluate (include/reina_hip.h: REINA_TP_*).  Words per record: rp_test_prim_words.
RP_HD void rp_test_prim_words(int what, uint32_t *n_in, uint32_t *n_out) {
    const uint32_t in_w[7] = {6, 3, 1, 1, 1, 8, 4}, out_w[7] = {4, 2, 1, 1, 1, 1, 1};
    *n_in = (what >= 0 && what < 7) ? in_w[what] : 0u;
    *n_out = (what >= 0 && what < 7) ? out_w[what] : 0u;
}
RP_HD void rp_test_prim(int what, const uint32_t *in, uint32_t *out) {
    if (what == 0) {          // Philox4x32-10: (k0, k1, c0, c1, c2, c3) -> 4 words
        const rp_u4 r = rp_philox(in[0], in[1], in[2], in[3], in[4], in[5]);
        out[0] = r.v[0]; out[1] = r.v[1]; out[2] = r.v[2]; out[3] = r.v[3];
    } else if (what == 1) {   // Philox2x32-10: (key, c0, c1) -> 2 words
        const rp_u2 r = rp_philox2(in[0], in[1], in[2]);
        out[0] = r.v[0]; out[1] = r.v[1];
    } else if (what == 2) {   // inverse normal of a 32-bit draw -> float bits
        out[0] = rp_f2u(rp_normal_from_u32(in[0]));
    } else if (what == 3) {
        out[0] = rp_f2u(rp_expf(rp_u2f(in[0])));
    } else if (what == 4) {
        out[0] = rp_f2u(rp_logf(rp_u2f(in[0])));
    } else if (what == 5) {   // gamma(mu, cv): (mu bits, cv bits, k0, k1, who, day, purpose, first block) -> float bits
        out[0] = rp_f2u(rp_gamma_mu_cv(rp_u2f(in[0]), rp_u2f(in[1]), in[2], in[3], in[4], in[5], in[6], in[7]));
    } else if (what == 6) {   // the draw behind the contact count: (k0, k1, who, day) -> 32-bit word
        out[0] = rp_count_draw(in[0], in[1], in[2], in[3]);
    }
}

#endif  // REINA_PRIMS_H
