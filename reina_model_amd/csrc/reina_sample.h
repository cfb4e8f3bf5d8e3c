// reina_sample.h -- Context.sample() of the reference (cythonsim/main.pyx:2047-2101): draw n values
// of one per-agent quantity with the engine's own samplers (Philox + reina_prims.h), on the host.
// Used by the parameter graphs of the reference's UI (calc/simulation.py:293-346); not a hot path.
#ifndef REINA_SAMPLE_H
#define REINA_SAMPLE_H

#include "../../include/reina_hip.h"
#include "reina_prims.h"
#include "reina_contacts.h"

enum {
    REINA_SAMPLE_CONTACTS_PER_DAY = 0, REINA_SAMPLE_SYMPTOM_SEVERITY, REINA_SAMPLE_INCUBATION_PERIOD,
    REINA_SAMPLE_ILLNESS_PERIOD, REINA_SAMPLE_HOSPITALIZATION_PERIOD, REINA_SAMPLE_ICU_PERIOD,
    REINA_SAMPLE_ONSET_TO_REMOVED_PERIOD
};

static inline int rs_severity(const reina_disease_t *d, int age, float val) {
    // Disease.get_symptom_severity (main.pyx:1042-1091), unvaccinated
    float syc = d->p_symptomatic[age];
    if (val >= syc) return RV_ASYMPTOMATIC;
    float dohc = d->p_death_outside_hospital[age];
    if (dohc != 0.0f) {
        if (val < dohc * syc) return RV_FATAL;
        val = (val - dohc) / (1.0f - dohc);
    }
    float sc = d->p_severe_given_symptomatic[age], cc = d->p_critical_given_severe[age], fc = d->p_fatal_given_critical[age];
    if (val < fc * cc * sc * syc) return RV_FATAL;
    if (val < cc * sc * syc) return RV_CRITICAL;
    if (val < sc * syc) return RV_SEVERE;
    return RV_MILD;
}

// severity < 0 means None (-> MILD, main.pyx:2054-2057). Variant 0 (the sampled person is a copy of
// people[0] with variant_idx 0). `nrc` = nr_contacts_by_age[age] of the current contact tables.
static inline int reina_sample_impl(const reina_disease_t *d, uint64_t seed, int what, int age, int severity,
                                    float nrc, int n, int32_t *out) {
    if (!d || !out || age < 0 || age >= REINA_MAX_AGES || what < 0 || what > REINA_SAMPLE_ONSET_TO_REMOVED_PERIOD)
        return REINA_E_INVALID;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int sev = severity >= 0 ? severity : RV_MILD;
    const uint32_t day = 0xFFFFFF00u + (uint32_t)what;  // a "day" no simulation reaches
    uint32_t count_row[REINA_COUNT_WORDS];
    if (what == REINA_SAMPLE_CONTACTS_PER_DAY) rc_count_thresholds(nrc, count_row);
    for (int i = 0; i < n; i++) {
        const uint32_t who = (uint32_t)i;
        if (what == REINA_SAMPLE_CONTACTS_PER_DAY) {
            out[i] = rc_count_from_draw(count_row, 0, rp_count_draw(k0, k1, who, day));   // the engine's own count sampler
        } else if (what == REINA_SAMPLE_SYMPTOM_SEVERITY) {
            out[i] = rs_severity(d, age, rp_uniform24(rp_philox(k0, k1, who, day, RP_P_INFECT, 0).v[0]));
        } else if (what == REINA_SAMPLE_INCUBATION_PERIOD) {
            out[i] = rp_round_to_int(rp_gamma_mu_cv(d->mean_incubation_duration[0], 0.86f, k0, k1, who, day, RP_P_INFECT, 1));
        } else {
            float mu = sev == RV_FATAL ? d->mean_duration_from_onset_to_death[0] : d->mean_duration_from_onset_to_recovery[0];
            float od = rp_gamma_mu_cv(mu, 0.45f, k0, k1, who, day, RP_P_ONSET, 1);
            float f;
            if (what == REINA_SAMPLE_ILLNESS_PERIOD) {
                f = od;
                if (sev >= RV_SEVERE) f *= d->ratio_of_duration_before_hospitalisation[0];
            } else if (what == REINA_SAMPLE_HOSPITALIZATION_PERIOD) {
                if (sev == RV_SEVERE) f = od * (1.0f - d->ratio_of_duration_before_hospitalisation[0]);
                else if (sev >= RV_CRITICAL) f = od * d->ratio_of_duration_in_ward[0];
                else f = 0.0f;
            } else if (what == REINA_SAMPLE_ICU_PERIOD) {
                if (sev >= RV_CRITICAL) {
                    f = 1.0f - d->ratio_of_duration_in_ward[0] - d->ratio_of_duration_before_hospitalisation[0];
                    f *= od;
                } else {
                    f = 0.0f;
                }
            } else {
                f = od;
            }
            out[i] = rp_round_to_int(f);
        }
    }
    return REINA_OK;
}

#endif
