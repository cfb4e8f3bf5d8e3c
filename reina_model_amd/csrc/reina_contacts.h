// Host-side contact-table build, shared by the HIP library (reina_build_contact_tables) and the CPU
// oracle library (par_build_contact_tables).  Plain C, no GPU.
//
// ContactMatrix.generate_contact_probabilities (cythonsim/main.pyx:1184-1235), which the reference
// runs in pandas whenever a mobility limitation changes (init_day :1285-1288), for a matrix whose
// participant ages all have the same number E of (place, contact-range) entries -- the shape of the
// reference's own contact file -- plus the packing into the thresholds of reina_contact_tables_t.
// Same arithmetic in the same order as the numpy form in contacts.py (double throughout; row sums
// Kahan-compensated like pandas' groupby-sum; running sums of quotients), so the tables are
// bit-identical; the Python form stays as the definition and the two are compared in the tests.
#ifndef REINA_CONTACTS_H
#define REINA_CONTACTS_H
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

// ------------------------------------------------------------------ the contact COUNT of an infectious agent
// get_nr_contacts (main.pyx:1308-1320): f = lognormal(0, 0.5) * nr_contacts_by_age[age] * factor; f = max(f, 1);
// nr = min(int(f) - 1, limit) -- (factor, limit) = (1, 100), or (0.5, 5) for an agent with symptoms.  The parallel engines
// draw it by INVERSION from one 32-bit word r (cell centre u = (r + 0.5) / 2^32 of the uniform): nr > k iff
// exp(z / 2) * c >= k + 2 iff u >= Phi(2 ln((k + 2) / c)), c = nr_contacts_by_age * factor -- so a row of `limit`
// thresholds thr[k] = the smallest r with nr > k turns the draw into a table search (no inverse normal, no exp on
// the device: they were 650 + 130 SIMD cycles per 64 agents, a fifth of k_day on a peak day).  Built on the host in
// double whenever the contact tables are (re)built; the HIP library and the CPU oracle call this same function.
#define REINA_COUNT_FULL 100   // limit of an agent without symptoms
#define REINA_COUNT_ILL 5      // limit of an agent with symptoms (factor 0.5)
#define REINA_COUNT_WORDS 112  // one row: [0, 100) full class, [100, 105) ill class, padding to a multiple of 16 bytes
static inline void rc_count_thresholds(float nr_contacts_of_age, uint32_t *row /* [REINA_COUNT_WORDS] */) {
    for (int cls = 0; cls < 2; cls++) {
        const double c = (double)nr_contacts_of_age * (cls ? 0.5 : 1.0);
        const int limit = cls ? REINA_COUNT_ILL : REINA_COUNT_FULL;
        uint32_t *thr = row + (cls ? REINA_COUNT_FULL : 0);
        for (int k = 0; k < limit; k++) {
            double t = 4294967296.0;
            if (c > 0.0) {
                const double x = 2.0 * log((double)(k + 2) / c);             // z from which exp(z / 2) * c >= k + 2
                t = ceil(4294967296.0 * (0.5 * erfc(-x / sqrt(2.0))) - 0.5);   // the first cell whose centre lies at or above Phi(x)
            }
            thr[k] = t <= 0.0 ? 0u : t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
        }
    }
    for (int k = REINA_COUNT_FULL + REINA_COUNT_ILL; k < REINA_COUNT_WORDS; k++) row[k] = 0xFFFFFFFFu;
}
// the count a draw gives with such a row (plain form: the oracle, Context.sample; the kernels start from a guide table)
static inline int rc_count_from_draw(const uint32_t *row, int ill, uint32_t r) {
    const uint32_t *thr = row + (ill ? REINA_COUNT_FULL : 0);
    const int limit = ill ? REINA_COUNT_ILL : REINA_COUNT_FULL;
    int n = 0;
    if (r == 0xFFFFFFFFu) r = 0xFFFFFFFEu;   // (0xFFFFFFFF is the thresholds' "never": rp_count_draw never returns it either)
    while (n < limit && r >= thr[n]) n++;
    return n;
}

static int reina_build_contact_tables_impl(const double *base, const int32_t *row_page, const int32_t *row_place,
                                           uint32_t n_rows, const double *mobility, uint32_t n_mobility,
                                           const int32_t *rows_mat, const int32_t *sorted_mat, uint32_t A, uint32_t E,
                                           double *totals_out, double *cum_out, float *nrc_out, uint32_t *thr_out,
                                           uint32_t thr_stride) {
    if (A == 0 || E == 0 || E > thr_stride) return -1;
    double *c = (double *)malloc((size_t)n_rows * sizeof(double));
    if (!c) return -2;
    memcpy(c, base, (size_t)n_rows * sizeof(double));
    // mobility factors in list order: (place or -1 = every place, min_age, max_age, factor).  Rows of one
    // (participant age, place) see the same factors in the same order: that product sequence is worked out
    // once per pair and then applied row by row (the multiplications stay sequential per row, as in the
    // reference: c *= f1; c *= f2; ...)
    if (n_mobility) {
        int32_t n_places = 1, max_page = 0;
        for (uint32_t r = 0; r < n_rows; r++) {
            if (row_place[r] + 1 > n_places) n_places = row_place[r] + 1;
            if (row_page[r] > max_page) max_page = row_page[r];
        }
        const size_t pairs = (size_t)(max_page + 1) * (size_t)n_places;
        uint16_t *cnt = (uint16_t *)calloc(pairs, sizeof(uint16_t));
        double *seq = (double *)malloc(pairs * n_mobility * sizeof(double));
        if (!cnt || !seq || n_mobility > 65535u) {
            free(cnt);
            free(seq);
            free(c);
            return -2;
        }
        for (uint32_t m = 0; m < n_mobility; m++) {
            const double *mf = mobility + 4u * m;
            const double factor = mf[3];
            if (factor == 1.0) continue;
            const int place = (int)mf[0];
            int lo = (int)mf[1], hi = (int)mf[2];
            if (lo < 0) lo = 0;
            if (hi > max_page) hi = max_page;
            for (int a = lo; a <= hi; a++)
                for (int pl = place < 0 ? 0 : place; pl < (place < 0 ? n_places : place + 1) && pl < n_places; pl++) {
                    const size_t q = (size_t)a * (size_t)n_places + (size_t)pl;
                    seq[q * n_mobility + cnt[q]++] = factor;
                }
        }
        for (uint32_t r = 0; r < n_rows; r++) {
            if (row_page[r] < 0 || row_place[r] < 0) continue;
            const size_t q = (size_t)row_page[r] * (size_t)n_places + (size_t)row_place[r];
            const double *fs = seq + q * n_mobility;
            double x = c[r];
            for (uint32_t k = 0; k < cnt[q]; k++) x *= fs[k];
            c[r] = x;
        }
        free(cnt);
        free(seq);
    }
    // four ages at a time: each age is its own dependent chain (Kahan sum, running sum); four chains
    // overlap in the pipeline while every index / output stream stays sequential
    for (uint32_t a0 = 0; a0 < A; a0 += 4) {
        const uint32_t na = A - a0 < 4u ? A - a0 : 4u;
        double sumx[4] = {0.0, 0.0, 0.0, 0.0}, comp[4] = {0.0, 0.0, 0.0, 0.0}, run[4] = {0.0, 0.0, 0.0, 0.0};
        for (uint32_t k = 0; k < E; k++)
            for (uint32_t j = 0; j < na; j++) {   // Kahan, in the rows' file order
                const double y = c[rows_mat[(size_t)(a0 + j) * E + k]] - comp[j];
                const double t = sumx[j] + y;
                comp[j] = t - sumx[j] - y;
                sumx[j] = t;
            }
        for (uint32_t j = 0; j < na; j++) {
            totals_out[a0 + j] = sumx[j];
            if (nrc_out) nrc_out[a0 + j] = (float)sumx[j];
        }
        for (uint32_t k = 0; k < E; k++)
            for (uint32_t j = 0; j < na; j++) {
                const size_t at = (size_t)(a0 + j) * E + k;
                const double q = c[sorted_mat[at]] / sumx[j];
                run[j] = k == 0 ? q : run[j] + q;
                cum_out[at] = run[j];
                if (thr_out) {
                    // floor(nan_to_num(run, nan=0) * 2^32) clipped to [0, 2^32 - 1]: NaN and negatives give 0,
                    // truncation is floor for the rest
                    const double y = run[j] * 4294967296.0;
                    uint32_t t = 0;
                    if (y >= 4294967295.0) t = 4294967295u;
                    else if (y >= 0.0) t = (uint32_t)y;
                    thr_out[(size_t)(a0 + j) * thr_stride + k] = t;
                }
            }
    }
    free(c);
    return 0;
}
#endif
