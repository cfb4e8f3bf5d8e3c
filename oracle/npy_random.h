/* oracle/npy_random.h -- TEST INFRASTRUCTURE (oracle), never linked into the product.
 *
 * CPU restatement of the random-number arithmetic the reference's RandomPool sits on
 * (cythonsim/simrandom.pyx:13-55).  That arithmetic is third-party: numpy's PCG64 bit
 * generator and `random_lognormal` / `random_gamma_f` from numpy/random/src/distributions
 * (requirements.txt:113 pins numpy==1.20.1; this image has 2.2.6; the algorithms below are the
 * published ones: PCG XSL-RR 128/64, 256-layer ziggurat normal, Marsaglia-Tsang gamma).
 * Pinned by tests/golden/rng_kat.npz (recorded at the RandomPool boundary in this container).
 */
#ifndef ORACLE_NPY_RANDOM_H
#define ORACLE_NPY_RANDOM_H

#include <math.h>
#include <stdint.h>

#include "npy_ziggurat_tables.h"

typedef unsigned __int128 u128;

typedef struct {
    u128 state, inc;
    int has_uint32;
    uint32_t uinteger;
} npy_pcg64;

#define PCG_MULT_128 ((((u128)0x2360ED051FC65DA4ULL) << 64) | (u128)0x4385DF649FCCF645ULL)

static inline void pcg64_init(npy_pcg64 *r, uint64_t st_hi, uint64_t st_lo, uint64_t inc_hi,
                              uint64_t inc_lo) {
    r->state = (((u128)st_hi) << 64) | st_lo;
    r->inc = (((u128)inc_hi) << 64) | inc_lo;
    r->has_uint32 = 0;
    r->uinteger = 0;
}

/* step, then XSL-RR output of the NEW state */
static inline uint64_t pcg64_next64(npy_pcg64 *r) {
    r->state = r->state * PCG_MULT_128 + r->inc;
    uint64_t hi = (uint64_t)(r->state >> 64), lo = (uint64_t)r->state;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((-rot) & 63));
}

/* simrandom.pyx:28-30 getint(): low half first, high half buffered for the next call */
static inline uint32_t pcg64_next32(npy_pcg64 *r) {
    if (r->has_uint32) {
        r->has_uint32 = 0;
        return r->uinteger;
    }
    uint64_t n = pcg64_next64(r);
    r->has_uint32 = 1;
    r->uinteger = (uint32_t)(n >> 32);
    return (uint32_t)n;
}

/* simrandom.pyx:24-26 get(): 53-bit double, does not touch the uint32 buffer */
static inline double pcg64_next_double(npy_pcg64 *r) {
    return (double)(pcg64_next64(r) >> 11) * (1.0 / 9007199254740992.0);
}

static inline float pcg64_next_float(npy_pcg64 *r) {
    return (float)(pcg64_next32(r) >> 8) * (1.0f / 16777216.0f);
}

#define ZIG_NOR_R 3.6541528853610087963519472518
#define ZIG_NOR_INV_R 0.27366123732975827203338247596
#define ZIG_NOR_R_F 3.6541528853610087963519472518f
#define ZIG_NOR_INV_R_F 0.27366123732975827203338247596f

static inline double npy_standard_normal(npy_pcg64 *g) {
    for (;;) {
        uint64_t r = pcg64_next64(g);
        int idx = (int)(r & 0xff);
        r >>= 8;
        int sign = (int)(r & 1);
        uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
        double x = (double)rabs * wi_double[idx];
        if (sign) x = -x;
        if (rabs < ki_double[idx]) return x;
        if (idx == 0) {
            for (;;) {
                double xx = -ZIG_NOR_INV_R * log1p(-pcg64_next_double(g));
                double yy = -log1p(-pcg64_next_double(g));
                if (yy + yy > xx * xx)
                    return ((rabs >> 8) & 1) ? -(ZIG_NOR_R + xx) : ZIG_NOR_R + xx;
            }
        } else {
            if (((fi_double[idx - 1] - fi_double[idx]) * pcg64_next_double(g) + fi_double[idx]) <
                exp(-0.5 * x * x))
                return x;
        }
    }
}

static inline float npy_standard_normal_f(npy_pcg64 *g) {
    for (;;) {
        uint32_t r = pcg64_next32(g);
        int idx = (int)(r & 0xff);
        int sign = (int)((r >> 8) & 1);
        uint32_t rabs = (r >> 9) & 0x0007fffff;
        float x = (float)rabs * wi_float[idx];
        if (sign) x = -x;
        if (rabs < ki_float[idx]) return x;
        if (idx == 0) {
            for (;;) {
                float xx = -ZIG_NOR_INV_R_F * log1pf(-pcg64_next_float(g));
                float yy = -log1pf(-pcg64_next_float(g));
                if (yy + yy > xx * xx)
                    return ((rabs >> 8) & 1) ? -(ZIG_NOR_R_F + xx) : ZIG_NOR_R_F + xx;
            }
        } else {
            if (((fi_float[idx - 1] - fi_float[idx]) * pcg64_next_float(g) + fi_float[idx]) <
                exp(-0.5 * x * x))
                return x;
        }
    }
}

/* shape >= 1 branch only is reachable from the simulator (kappa = 1/cv^2 = 1.35 or 4.94);
 * the other branches are restated for completeness of `random_standard_gamma_f`. */
static inline float npy_standard_exponential_f_inv(npy_pcg64 *g) {
    return -log1pf(-pcg64_next_float(g));
}

static inline float npy_standard_gamma_f(npy_pcg64 *g, float shape) {
    float b, c, U, V, X, Y;
    if (shape == 1.0f) {
        /* numpy uses the ziggurat exponential here; unreachable from the simulator */
        return npy_standard_exponential_f_inv(g);
    } else if (shape == 0.0f) {
        return 0.0f;
    } else if (shape < 1.0f) {
        for (;;) {
            U = pcg64_next_float(g);
            V = npy_standard_exponential_f_inv(g);
            if (U <= 1.0f - shape) {
                X = powf(U, 1.0f / shape);
                if (X <= V) return X;
            } else {
                Y = -logf((1.0f - U) / shape);
                X = powf(1.0f - shape + shape * Y, 1.0f / shape);
                if (X <= (V + Y)) return X;
            }
        }
    } else {
        b = shape - 1.0f / 3.0f;
        c = 1.0f / sqrtf(9.0f * b);
        for (;;) {
            do {
                X = npy_standard_normal_f(g);
                V = 1.0f + c * X;
            } while (V <= 0.0f);
            V = V * V * V;
            U = pcg64_next_float(g);
            if (U < 1.0f - 0.0331f * (X * X) * (X * X)) return (b * V);
            if (logf(U) < 0.5f * X * X + b * (1.0f - V + logf(V))) return (b * V);
        }
    }
}

static inline float npy_gamma_f(npy_pcg64 *g, float shape, float scale) {
    return scale * npy_standard_gamma_f(g, shape);
}

static inline double npy_lognormal(npy_pcg64 *g, double mean, double sigma) {
    return exp(mean + sigma * npy_standard_normal(g));
}

/* ---- RandomPool (simrandom.pyx:13-55) ---- */
static inline double rp_get(npy_pcg64 *g) { return pcg64_next_double(g); }
static inline uint32_t rp_getint(npy_pcg64 *g) { return pcg64_next32(g); }
static inline int rp_chance(npy_pcg64 *g, double p) {
    if (p == 1.0) return 1;
    if (p == 0) return 0;
    return pcg64_next_double(g) < p;
}
static inline double rp_lognormal(npy_pcg64 *g, double mean, double sigma) {
    return npy_lognormal(g, mean, sigma);
}
static inline float rp_gamma(npy_pcg64 *g, float mu, float cv) {
    float sigma = cv * mu;
    float theta = (sigma * sigma) / mu;
    float kappa = mu / theta;
    return npy_gamma_f(g, kappa, theta);
}

#endif
