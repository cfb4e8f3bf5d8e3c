/* oracle/reina_seq.c -- TEST INFRASTRUCTURE (oracle "A"), never linked into the product.
 *
 * Sequential CPU restatement of the reference's per-day agent update loop
 * (cythonsim/main.pyx + cythonsim/simrandom.pyx), one PCG64 stream consumed in the reference's
 * draw order.  Gate: per-day state histograms bit-exact against the
 * golden vectors recorded from the real cythonsim in this container (tests/golden/*.npz).
 *
 * Split with the Python driver (oracle/seq_oracle.py): everything the reference does in
 * Python/pandas at host level (intervention dispatch main.pyx:1880-1960, contact-table build
 * :1184-1235) stays in Python and pushes plain arrays in here; everything that consumes the
 * RNG stream or touches agents lives here.
 *
 * Every function cites the reference lines it follows.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "npy_random.h"

/* main.pyx:33-48 */
enum { ASYMPTOMATIC = 0, MILD, SEVERE, CRITICAL, FATAL };
enum { SUSCEPTIBLE = 0, INCUBATION, ILLNESS, HOSPITALIZED, IN_ICU, RECOVERED, DEAD };
/* main.pyx:51-61 */
enum {
    NO_PROBLEMOS = 0, TOO_MANY_INFECTEES, TOO_MANY_CONTACTS, HOSPITAL_ACCOUNTING_FAILURE,
    NEGATIVE_CONTACTS, MALLOC_FAILURE, OTHER_FAILURE, WRONG_STATE, CONTACT_PROBABILITY_FAILURE,
    INFECTEES_MISMATCH
};
/* main.pyx:441-445 */
enum { NO_TESTING = 0, ALL_WITH_SYMPTOMS_CT, ALL_WITH_SYMPTOMS, ONLY_SEVERE_SYMPTOMS };
enum { DEATH_IN_HOSPITAL = 0, DEATH_OUTSIDE_HOSPITAL };

#define NR_CONTACT_PLACES 6
#define MAX_INFECTEES 64
#define MAX_CONTACTS 128
#define MAX_CLASSES 16
#define MAX_VARIANTS 8
#define N_IOT 21

/* main.pyx:132-144 (`int16` is typedef'd to C int, :30) */
typedef struct {
    int32_t idx, infector;
    uint8_t age, has_immunity, is_infected, was_detected, queued_for_testing, symptom_severity,
        place_of_death, state, included_in_totals, variant_idx;
    int day_of_infection, days_left, other_people_infected, other_people_exposed_today,
        day_of_illness;
    int max_contacts_per_day;
    int day_of_vaccination;
    float days_from_onset_to_removed;
    uint8_t nr_infectees;
    int32_t *infectees;
} Person;

typedef struct {
    int32_t person_idx;
    float mask_p;
    int place;
} Contact;

/* main.pyx:684-730 */
typedef struct {
    int classes[MAX_CLASSES + N_IOT];
    float values[MAX_CLASSES + N_IOT];
    int num_classes, min_class, max_class;
} ClassifiedValues;

/* main.pyx:787-806 */
typedef struct {
    float p_icu_death_no_beds, p_hospital_death_no_beds;
    float mean_incubation_duration, mean_duration_from_onset_to_death,
        mean_duration_from_onset_to_recovery;
    float ratio_of_duration_before_hospitalisation, ratio_of_duration_in_ward;
    float infectiousness_multiplier, p_asymptomatic_infection;
    ClassifiedValues p_susceptibility, p_symptomatic, p_severe, p_critical, p_fatal,
        p_death_outside_hospital, infectiousness_over_time;
    float p_mask_protects_wearer, p_mask_protects_others;
} Variant;

/* main.pyx:1094-1103 */
typedef struct {
    int place, contact_age_min, contact_age_max;
    double cum_p;
    float mask_p;
} ContactProbability;

typedef struct {
    double nr_daily;
    int min_age, max_age; /* -1 = None */
} Vaccination;

enum {
    C_INFECTED = 0, C_DETECTED, C_ALL_DETECTED, C_ALL_INFECTED, C_IN_WARD, C_HOSPITALIZED,
    C_IN_ICU, C_CUM_ICU, C_DEAD, C_SUSCEPTIBLE, C_RECOVERED, C_VACCINATED,
    C_NON_HOSPITAL_DEATHS, C_NEW_INFECTIONS, C_NR
};

typedef struct {
    /* Population (main.pyx:1323-1352) */
    Person *people;
    int total_people, nr_ages;
    int32_t *people_sorted_by_age, *age_start;
    int *cnt[C_NR];
    int infected_by_variant[MAX_VARIANTS];
    int daily_contacts[NR_CONTACT_PLACES];
    ClassifiedValues imported_infection_ages;
    int weekly_infections_amount;
    double weekly_infections_leftover[MAX_VARIANTS + 1]; /* Python floats holding C-float values */
    double weekly_infections_shares[MAX_VARIANTS];
    int limit_mass_gatherings;
    /* ContactMatrix (main.pyx:1119-1129) */
    double *nr_contacts_by_age;
    ContactProbability *cp;
    int *cp_offset, *cp_count;
    /* Disease */
    Variant variants[MAX_VARIANTS];
    int nr_variants;
    /* HealthcareSystem (main.pyx:451-472) */
    int beds, icu_units, available_beds, available_icu_units, ct_cases_per_day;
    float p_detected_anyway, p_successful_tracing;
    int testing_mode;
    int *queue;
    int queue_len, queue_cap;
    Vaccination vaccinations[64];
    int nr_vaccinations;
    /* Context (main.pyx:1746-1757) */
    npy_pcg64 rng;
    int problem, day;
    int total_infections, total_infectors, exposed_per_day;
    long n_unable_to_import;
} Sim;

static void set_problem(Sim *s, int problem) { s->problem = problem; }

/* ---- ClassifiedValues: main.pyx:690-730 ---- */
static void cv_init(ClassifiedValues *cv, int n, const int *classes, const double *values) {
    cv->num_classes = n;
    cv->min_class = 0x7fffffff;
    cv->max_class = 0;
    for (int i = 0; i < n; i++) {
        cv->classes[i] = classes[i];
        cv->values[i] = (float)values[i];
        if (classes[i] < cv->min_class) cv->min_class = classes[i];
        if (classes[i] > cv->max_class) cv->max_class = classes[i];
    }
}

/* main.pyx:710-718 */
static float cv_get(const ClassifiedValues *cv, int kls, float dflt) {
    if (kls < cv->min_class || kls > cv->max_class) return dflt;
    for (int i = 0; i < cv->num_classes; i++)
        if (cv->classes[i] == kls) return cv->values[i];
    return dflt;
}

/* main.pyx:721-730: loop variable keeps its last value when the loop runs out */
static float cv_get_greatest_lte(const ClassifiedValues *cv, int kls) {
    int idx = 0;
    for (int i = 0; i < cv->num_classes; i++) {
        idx = i;
        if (cv->classes[i] > kls) {
            idx = i - 1;
            break;
        }
    }
    return cv->values[idx];
}

/* main.pyx:773-774 */
static inline int round_to_int(float f) { return (int)(f + 0.5f); }

/* ---- Population counters: main.pyx:1576-1630 ---- */
static void pop_infect(Sim *s, Person *p) {
    int age = p->age;
    s->cnt[C_SUSCEPTIBLE][age] -= 1;
    s->cnt[C_INFECTED][age] += 1;
    s->cnt[C_ALL_INFECTED][age] += 1;
    s->cnt[C_NEW_INFECTIONS][age] += 1;
    s->infected_by_variant[p->variant_idx] += 1;
}
static void pop_recover(Sim *s, Person *p) {
    s->cnt[C_INFECTED][p->age] -= 1;
    s->cnt[C_RECOVERED][p->age] += 1;
}
static void pop_detect(Sim *s, Person *p) {
    s->cnt[C_DETECTED][p->age] += 1;
    s->cnt[C_ALL_DETECTED][p->age] += 1;
}
static void pop_hospitalize(Sim *s, Person *p) {
    s->cnt[C_HOSPITALIZED][p->age] += 1;
    s->cnt[C_IN_WARD][p->age] += 1;
}
static void pop_transfer_to_icu(Sim *s, Person *p) {
    s->cnt[C_IN_WARD][p->age] -= 1;
    s->cnt[C_IN_ICU][p->age] += 1;
    s->cnt[C_CUM_ICU][p->age] += 1;
}
static void pop_release_from_hospital(Sim *s, Person *p) {
    if (p->state == IN_ICU)
        s->cnt[C_IN_ICU][p->age] -= 1;
    else
        s->cnt[C_IN_WARD][p->age] -= 1;
    s->cnt[C_HOSPITALIZED][p->age] -= 1;
}
static void pop_die(Sim *s, Person *p) {
    s->cnt[C_INFECTED][p->age] -= 1;
    s->cnt[C_DEAD][p->age] += 1;
    if (p->place_of_death == DEATH_OUTSIDE_HOSPITAL) s->cnt[C_NON_HOSPITAL_DEATHS][p->age] += 1;
}

/* ---- Disease: main.pyx:895-1091 ---- */
static float get_source_infectiousness(Sim *s, const Person *src) {
    int day;
    if (src->state == INCUBATION)
        day = -src->days_left;
    else if (src->state == ILLNESS)
        day = src->day_of_illness;
    else
        return 0;
    return cv_get(&s->variants[src->variant_idx].infectiousness_over_time, day, 0);
}

/* main.pyx:908-934 */
static int did_infect(Sim *s, Person *person, Person *source, float mask_p) {
    float source_infectiousness = get_source_infectiousness(s, source);
    Variant *variant = &s->variants[source->variant_idx];
    float p_susceptibility = cv_get_greatest_lte(&variant->p_susceptibility, person->age);
    float p, a, b;
    if (source->symptom_severity == ASYMPTOMATIC)
        source_infectiousness *= variant->p_asymptomatic_infection;
    p = source_infectiousness * p_susceptibility * variant->infectiousness_multiplier;
    if (!rp_chance(&s->rng, p)) return 0;
    if (mask_p) {
        a = mask_p * variant->p_mask_protects_others;
        b = mask_p * variant->p_mask_protects_wearer;
        p = a + b - a * b;
        if (rp_chance(&s->rng, p)) return 0;
    }
    return 1;
}

/* main.pyx:957-974 */
static int dies_in_hospital(Sim *s, Person *person, int care_available) {
    Variant *variant = &s->variants[person->variant_idx];
    float chance = 0;
    if (person->symptom_severity == FATAL) {
        return 1;
    } else if (person->symptom_severity == CRITICAL) {
        if (care_available) return 0;
        chance = variant->p_icu_death_no_beds;
    } else if (person->symptom_severity == SEVERE) {
        if (care_available) return 0;
        chance = variant->p_hospital_death_no_beds;
    }
    return rp_chance(&s->rng, chance);
}

/* main.pyx:977-986 */
static int get_incubation_days(Sim *s, Person *person) {
    Variant *variant = &s->variants[person->variant_idx];
    float f = rp_gamma(&s->rng, variant->mean_incubation_duration, 0.86f);
    return round_to_int(f);
}

/* main.pyx:989-1001 */
static float get_days_from_onset_to_removed(Sim *s, Person *person) {
    Variant *variant = &s->variants[person->variant_idx];
    float mu, cv = 0.45f;
    if (person->symptom_severity == FATAL)
        mu = variant->mean_duration_from_onset_to_death;
    else
        mu = variant->mean_duration_from_onset_to_recovery;
    return rp_gamma(&s->rng, mu, cv);
}

/* main.pyx:1004-1014 */
static int get_illness_days(Sim *s, Person *person) {
    Variant *variant = &s->variants[person->variant_idx];
    float f = person->days_from_onset_to_removed;
    if (person->symptom_severity != ASYMPTOMATIC && person->symptom_severity != MILD)
        f *= variant->ratio_of_duration_before_hospitalisation;
    return round_to_int(f);
}

/* main.pyx:1016-1027 */
static int get_hospitalization_days(Sim *s, Person *person) {
    Variant *variant = &s->variants[person->variant_idx];
    float f;
    if (person->symptom_severity == SEVERE)
        f = person->days_from_onset_to_removed * (1 - variant->ratio_of_duration_before_hospitalisation);
    else if (person->symptom_severity == FATAL || person->symptom_severity == CRITICAL)
        f = person->days_from_onset_to_removed * variant->ratio_of_duration_in_ward;
    else
        f = 0;
    return round_to_int(f);
}

/* main.pyx:1029-1039 */
static int get_icu_days(Sim *s, Person *person) {
    Variant *variant = &s->variants[person->variant_idx];
    float f;
    if (person->symptom_severity == FATAL || person->symptom_severity == CRITICAL) {
        f = 1 - variant->ratio_of_duration_in_ward - variant->ratio_of_duration_before_hospitalisation;
        f *= person->days_from_onset_to_removed;
    } else {
        f = 0;
    }
    return round_to_int(f);
}

/* main.pyx:1042-1091 (both FATAL branches carry the same condition: quirk Q2) */
static int get_symptom_severity(Sim *s, Person *person) {
    Variant *variant = &s->variants[person->variant_idx];
    float syc, sc, cc, fc, dohc, val, vmod;
    int days;
    val = (float)rp_get(&s->rng);
    vmod = 1.0f;
    if (person->day_of_vaccination >= 0) {
        days = s->day - person->day_of_vaccination;
        if (days > 14) vmod *= (float)(1 - 0.90);
    }
    syc = cv_get_greatest_lte(&variant->p_symptomatic, person->age);
    if (val >= syc) return ASYMPTOMATIC;
    syc *= vmod;
    dohc = cv_get_greatest_lte(&variant->p_death_outside_hospital, person->age);
    if (dohc) {
        if (val < dohc * syc) {
            person->place_of_death = DEATH_OUTSIDE_HOSPITAL;
            return FATAL;
        }
        val = (val - dohc) / (1 - dohc);
    }
    sc = cv_get_greatest_lte(&variant->p_severe, person->age);
    cc = cv_get_greatest_lte(&variant->p_critical, person->age);
    fc = cv_get_greatest_lte(&variant->p_fatal, person->age);
    if (val < fc * cc * sc * syc) {
        person->place_of_death = DEATH_OUTSIDE_HOSPITAL;
        return FATAL;
    }
    if (val < cc * sc * syc) return CRITICAL;
    if (val < sc * syc) return SEVERE;
    return MILD;
}

/* ---- ContactMatrix sampling: main.pyx:1290-1320 ---- */
static ContactProbability *get_one_contact(Sim *s, Person *person) {
    double p = rp_get(&s->rng);
    ContactProbability *base = s->cp + s->cp_offset[person->age];
    int n = s->cp_count[person->age];
    for (int i = 0; i < n; i++)
        if (p < base[i].cum_p) return &base[i];
    s->problem = CONTACT_PROBABILITY_FAILURE;
    return NULL;
}

static int get_nr_contacts(Sim *s, Person *person, float factor, int limit) {
    float f = (float)(rp_lognormal(&s->rng, 0, 0.5) * s->nr_contacts_by_age[person->age]);
    f *= factor;
    if (f < 1) f = 1;
    int nr = (int)f - 1;
    if (nr > limit) nr = limit;
    return nr;
}

/* main.pyx:1525-1535 */
static int get_person_from_age_range(Sim *s, int min_age, int max_age) {
    int idx_start = s->age_start[min_age], idx_end;
    if (max_age < s->nr_ages - 1)
        idx_end = s->age_start[max_age + 1];
    else
        idx_end = s->total_people;
    return s->people_sorted_by_age[idx_start + (int)(rp_getint(&s->rng) % (uint32_t)(idx_end - idx_start))];
}

/* main.pyx:1539-1573 */
static int get_contacts(Sim *s, Person *person, Contact *contacts, float factor, int limit) {
    if (s->limit_mass_gatherings && s->limit_mass_gatherings < limit) limit = s->limit_mass_gatherings;
    int nr = get_nr_contacts(s, person, factor, limit);
    if (nr > MAX_CONTACTS) {
        s->problem = TOO_MANY_CONTACTS;
        return 0;
    }
    for (int i = 0; i < nr; i++) {
        ContactProbability *cp = get_one_contact(s, person);
        if (cp == NULL) continue;
        Contact *c = contacts + i;
        c->person_idx = get_person_from_age_range(s, cp->contact_age_min, cp->contact_age_max);
        c->place = cp->place;
        c->mask_p = cp->mask_p;
        s->daily_contacts[c->place] += 1;
    }
    return nr;
}

/* main.pyx:936-955 */
static int get_exposed_people(Sim *s, Person *person, Contact *contacts) {
    if (person->was_detected) return 0;
    if (!get_source_infectiousness(s, person)) return 0;
    if (person->state == INCUBATION) {
        return get_contacts(s, person, contacts, 1.0f, 100);
    } else if (person->state == ILLNESS) {
        if (person->symptom_severity == ASYMPTOMATIC) return get_contacts(s, person, contacts, 1.0f, 100);
        return get_contacts(s, person, contacts, 0.5f, 5);
    }
    return 0;
}

/* ---- Person transitions: main.pyx:209-438 ---- */
static void person_infect(Sim *s, Person *self, Person *source, int variant_idx) {
    self->state = INCUBATION;
    self->symptom_severity = (uint8_t)get_symptom_severity(s, self);
    self->days_left = get_incubation_days(s, self);
    self->is_infected = 1;
    self->day_of_infection = s->day;
    if (source != NULL) {
        self->infector = source->idx;
        if (source->infectees != NULL) {
            if (source->nr_infectees >= MAX_INFECTEES) {
                set_problem(s, TOO_MANY_INFECTEES);
                return;
            }
            source->infectees[source->nr_infectees] = self->idx;
            source->nr_infectees += 1;
        }
        variant_idx = source->variant_idx;
    }
    self->variant_idx = (uint8_t)variant_idx;
    if (s->testing_mode == ALL_WITH_SYMPTOMS_CT) {
        if (self->infectees != NULL) {
            set_problem(s, INFECTEES_MISMATCH);
        } else {
            self->infectees = (int32_t *)malloc(sizeof(int32_t) * MAX_INFECTEES);
            if (self->infectees == NULL) set_problem(s, MALLOC_FAILURE);
        }
    }
    pop_infect(s, self);
}

static int person_expose(Sim *s, Person *self, Person *source, float mask_p) {
    if (self->is_infected || self->has_immunity) return 0;
    if (did_infect(s, self, source, mask_p)) {
        person_infect(s, self, source, -1);
        return 1;
    }
    return 0;
}

static void person_expose_others(Sim *s, Person *self) {
    Contact contacts[MAX_CONTACTS];
    int nr_contacts = get_exposed_people(s, self, contacts);
    self->other_people_exposed_today = nr_contacts;
    if (nr_contacts == 0) return;
    if (nr_contacts < 0) {
        set_problem(s, NEGATIVE_CONTACTS);
        return;
    }
    if (nr_contacts > self->max_contacts_per_day) self->max_contacts_per_day = nr_contacts;
    int32_t *infectees = self->infectees;
    for (int i = 0; i < nr_contacts; i++) {
        int exposee_idx = contacts[i].person_idx;
        Person *target = &s->people[exposee_idx];
        if (person_expose(s, target, self, contacts[i].mask_p)) {
            if (infectees != NULL) {
                if (self->other_people_infected >= MAX_INFECTEES) {
                    set_problem(s, TOO_MANY_INFECTEES);
                    break;
                }
                infectees[self->other_people_infected] = exposee_idx;
            }
            self->other_people_infected += 1;
        }
    }
}

static void person_detect(Sim *s, Person *self) {
    if (self->was_detected) set_problem(s, WRONG_STATE);
    self->was_detected = 1;
    pop_detect(s, self);
}

/* HealthcareSystem.queue_for_testing main.pyx:474-488 */
static int queue_for_testing(Sim *s, int person_idx, float p_success) {
    Person *p = s->people + person_idx;
    if (p->state == DEAD || p->was_detected || p->queued_for_testing) return 0;
    if (!rp_chance(&s->rng, p_success)) return 0;
    p->queued_for_testing = 1;
    if (s->queue_len == s->queue_cap) {
        s->queue_cap = s->queue_cap ? s->queue_cap * 2 : 1024;
        s->queue = (int *)realloc(s->queue, sizeof(int) * s->queue_cap);
    }
    s->queue[s->queue_len++] = person_idx;
    return 1;
}

/* main.pyx:595-615 */
static void seek_testing(Sim *s, Person *person) {
    int q = 0;
    if (s->testing_mode == ALL_WITH_SYMPTOMS || s->testing_mode == ALL_WITH_SYMPTOMS_CT) {
        q = 1;
    } else if (s->testing_mode == ONLY_SEVERE_SYMPTOMS) {
        if (person->symptom_severity == SEVERE || person->symptom_severity == CRITICAL ||
            person->symptom_severity == FATAL)
            q = 1;
        else if (rp_chance(&s->rng, s->p_detected_anyway))
            q = 1;
    }
    if (q) queue_for_testing(s, person->idx, 1);
}

static void person_become_ill(Sim *s, Person *self) {
    self->state = ILLNESS;
    self->days_from_onset_to_removed = get_days_from_onset_to_removed(s, self);
    self->days_left = get_illness_days(s, self);
    if (self->symptom_severity != ASYMPTOMATIC)
        if (!self->was_detected) seek_testing(s, self);
}

static void person_become_removed(Sim *s, Person *self) {
    (void)s;
    self->is_infected = 0;
    self->has_immunity = 1;
    if (self->infectees != NULL) {
        free(self->infectees);
        self->infectees = NULL;
    }
}

static void person_recover(Sim *s, Person *self) {
    self->state = RECOVERED;
    person_become_removed(s, self);
    pop_recover(s, self);
}

static void person_die(Sim *s, Person *self) {
    self->state = DEAD;
    pop_die(s, self);
    person_become_removed(s, self);
}

/* HealthcareSystem.hospitalize/release/to_icu/release_from_icu main.pyx:617-651 */
static int hc_hospitalize(Sim *s) {
    if (s->available_beds == 0) return 0;
    s->available_beds -= 1;
    return 1;
}
static int hc_to_icu(Sim *s) {
    s->available_beds += 1;
    if (s->available_icu_units == 0) return 0;
    s->available_icu_units -= 1;
    return 1;
}

static void person_hospitalize(Sim *s, Person *self) {
    if (!self->was_detected) person_detect(s, self);
    if (!hc_hospitalize(s)) {
        if (dies_in_hospital(s, self, 0))
            person_die(s, self);
        else
            person_recover(s, self);
        return;
    }
    self->days_left = get_hospitalization_days(s, self);
    self->state = HOSPITALIZED;
    pop_hospitalize(s, self);
}

static void person_transfer_to_icu(Sim *s, Person *self) {
    if (!hc_to_icu(s)) {
        if (dies_in_hospital(s, self, 0)) {
            pop_release_from_hospital(s, self);
            person_die(s, self);
            return;
        }
    }
    self->days_left = get_icu_days(s, self);
    pop_transfer_to_icu(s, self);
    self->state = IN_ICU;
}

static void person_release_from_hospital(Sim *s, Person *self) {
    int death;
    pop_release_from_hospital(s, self);
    if (self->state == IN_ICU) {
        death = dies_in_hospital(s, self, 1);
        s->available_icu_units += 1;
    } else {
        death = dies_in_hospital(s, self, 1);
        s->available_beds += 1;
    }
    if (death)
        person_die(s, self);
    else
        person_recover(s, self);
}

/* main.pyx:395-438 */
static void person_advance(Sim *s, Person *self) {
    self->other_people_exposed_today = 0;
    if (self->state == INCUBATION) {
        if (self->day_of_infection == s->day) return;
        person_expose_others(s, self);
        if (self->days_left > 0) self->days_left -= 1;
        if (self->days_left == 0) person_become_ill(s, self);
    } else if (self->state == ILLNESS) {
        person_expose_others(s, self);
        self->day_of_illness += 1;
        if (self->days_left > 0) self->days_left -= 1;
        if (self->days_left == 0) {
            if (self->symptom_severity == FATAL && self->place_of_death == DEATH_OUTSIDE_HOSPITAL)
                person_die(s, self);
            else if (self->symptom_severity == SEVERE || self->symptom_severity == CRITICAL ||
                     self->symptom_severity == FATAL)
                person_hospitalize(s, self);
            else
                person_recover(s, self);
        }
    } else if (self->state == HOSPITALIZED) {
        if (self->days_left > 0) self->days_left -= 1;
        if (self->days_left == 0) {
            if (self->symptom_severity == CRITICAL || self->symptom_severity == FATAL)
                person_transfer_to_icu(s, self);
            else
                person_release_from_hospital(s, self);
        }
    } else if (self->state == IN_ICU) {
        if (self->days_left > 0) self->days_left -= 1;
        if (self->days_left == 0) person_release_from_hospital(s, self);
    }
}

/* main.pyx:377-392 */
static int person_vaccinate(Sim *s, Person *self) {
    if (self->state == DEAD || self->day_of_vaccination >= 0) return 0;
    if (self->was_detected) return 0;
    self->day_of_vaccination = s->day;
    s->cnt[C_VACCINATED][self->age] += 1;
    return 1;
}

/* main.pyx:495-512 */
static void perform_contact_tracing(Sim *s, int person_idx, int level) {
    Person *p = s->people + person_idx;
    if (level > 1) return;
    if (p->infector >= 0) {
        if (queue_for_testing(s, p->infector, s->p_successful_tracing))
            perform_contact_tracing(s, p->infector, level + 1);
    }
    if (p->infectees != NULL) {
        for (int i = 0; i < p->nr_infectees; i++) {
            int infectee_idx = p->infectees[i];
            if (queue_for_testing(s, infectee_idx, s->p_successful_tracing))
                perform_contact_tracing(s, infectee_idx, level + 1);
        }
    }
}

/* main.pyx:560-583 */
static void vaccinate_people(Sim *s, int nr_to_vaccinate, int min_age, int max_age) {
    int idx_start = s->age_start[min_age], idx_end, idx, vaccinated = 0;
    if (max_age < s->nr_ages - 1)
        idx_end = s->age_start[max_age + 1];
    else
        idx_end = s->total_people;
    idx = idx_end - 1;
    if (nr_to_vaccinate > idx_end - idx_start) nr_to_vaccinate = idx_end - idx_start;
    while (vaccinated < nr_to_vaccinate && idx >= idx_start) {
        Person *person = s->people + s->people_sorted_by_age[idx];
        idx -= 1;
        if (!person_vaccinate(s, person)) continue;
        vaccinated += 1;
    }
}

/* HealthcareSystem.iterate main.pyx:514-558 */
static void hc_iterate(Sim *s) {
    int *queue = s->queue;
    int n = s->queue_len;
    s->ct_cases_per_day = n;
    s->queue = NULL;
    s->queue_len = 0;
    s->queue_cap = 0;
    for (int k = 0; k < n; k++) {
        int idx = queue[k];
        Person *person = s->people + idx;
        person->queued_for_testing = 0;
        /* both guards have compile-time-empty bodies (TESTING_TRACE False): every test is positive */
        person_detect(s, person);
        if (s->testing_mode == ALL_WITH_SYMPTOMS_CT) perform_contact_tracing(s, idx, 0);
    }
    free(queue);
    int pop_max_age = s->nr_ages - 1;
    for (int k = 0; k < s->nr_vaccinations; k++) {
        Vaccination *v = &s->vaccinations[k];
        if (!v->nr_daily) continue;
        int min_age = v->min_age < 0 ? 0 : v->min_age;
        int max_age = v->max_age < 0 ? pop_max_age : v->max_age;
        vaccinate_people(s, (int)v->nr_daily, min_age, max_age);
    }
}

/* ---- imports: main.pyx:1632-1685 ---- */
static int get_import_infection_person(Sim *s) {
    ClassifiedValues *ages = &s->imported_infection_ages;
    int idx = 0, age = 0, max_age, min_age;
    float cumprob, p;
    p = (float)rp_get(&s->rng);
    for (int i = 0; i < ages->num_classes; i++) {
        idx = i;
        age = ages->classes[i];
        cumprob = ages->values[i];
        if (p <= cumprob) break;
    }
    min_age = age;
    if (idx == ages->num_classes)
        max_age = s->nr_ages;
    else
        max_age = ages->classes[idx + 1] - 1; /* reads one past the end for the last class (Q10) */
    return get_person_from_age_range(s, min_age, max_age);
}

void seq_infect_people(Sim *s, int count, int variant) {
    for (int i = 0; i < count; i++) {
        Person *person = NULL;
        int found = 0;
        for (int x = 0; x < 10; x++) {
            int person_idx = get_import_infection_person(s);
            person = &s->people[person_idx];
            if (person->state == SUSCEPTIBLE) {
                found = 1;
                break;
            }
        }
        if (!found) {
            s->n_unable_to_import++;
            continue;
        }
        person_infect(s, person, NULL, variant);
    }
}

static void infect_people_daily(Sim *s) {
    for (int vid = 0; vid < s->nr_variants; vid++) {
        float leftover = (float)s->weekly_infections_leftover[vid];
        /* `leftover += <C double> * <Python float>`: add in double, store back as C float */
        leftover = (float)((double)leftover + s->weekly_infections_amount / 7.0 * s->weekly_infections_shares[vid]);
        int amount_today = (int)leftover;
        if (amount_today) {
            seq_infect_people(s, amount_today, vid);
            leftover -= amount_today;
        }
        s->weekly_infections_leftover[vid] = leftover;
    }
}

/* Population.init_day main.pyx:1687-1699 (contact-table rebuild is done by the Python driver) */
static void pop_init_day(Sim *s) {
    for (int i = 0; i < NR_CONTACT_PLACES; i++) s->daily_contacts[i] = 0;
    for (int i = 0; i < s->nr_ages; i++) {
        s->cnt[C_NEW_INFECTIONS][i] = 0;
        s->cnt[C_DETECTED][i] = 0;
    }
    for (int i = 0; i < s->nr_variants; i++) s->infected_by_variant[i] = 0;
    infect_people_daily(s);
}

/* main.pyx:1968-1992 */
static inline void process_person(Sim *s, Person *person) {
    if ((person->state == RECOVERED || person->state == DEAD) && !person->included_in_totals) {
        s->total_infectors += 1;
        s->total_infections += person->other_people_infected;
        person->included_in_totals = 1;
    }
    if (!person->is_infected) return;
    person_advance(s, person);
    s->exposed_per_day += person->other_people_exposed_today;
}

static void iterate_people(Sim *s) {
    int total = s->total_people;
    int start_idx = (int)(rp_getint(&s->rng) % (uint32_t)total);
    for (int i = 0; i < total; i++) {
        int person_idx = (start_idx + i) % total;
        process_person(s, s->people + person_idx);
    }
}

/* Population.set_initial_state main.pyx:1452-1516 (+ get_random_person :1518-1521).  The caller
 * passes the InitialPopulationCondition numbers (calc/datasets.py:106-134); persons are drawn
 * WITH replacement and without any state check, exactly like the reference. */
/* returns 1 where the reference raises AssertionError out of Context.__init__: an ICU-fated agent who was refused a bed
 * (person_hospitalize left it dead or recovered) goes on to person_transfer_to_icu, whose Population.transfer_to_icu /
 * release_from_hospital assert state == HOSPITALIZED (main.pyx:1495 -> :350 -> :1603, :1613) -- i.e. an initial condition
 * with people in ICU and a hospital without beds cannot be constructed */
int seq_set_initial_state(Sim *s, int incubating, int recovered_without_illness, int ill, int dead,
                          int in_icu, int in_ward, int were_incubating, int confirmed_cases) {
    int i_incubating = incubating;
    int i_recovered_without_symptoms = i_incubating + recovered_without_illness;
    int i_ill_at_home = i_recovered_without_symptoms + ill;
    int i_dead = i_ill_at_home + dead;
    int i_in_icu = i_dead + in_icu;
    int i_in_ward = i_in_icu + in_ward;
    for (int i = 0; i < were_incubating; i++) {
        Person *person = s->people + (int)(rp_getint(&s->rng) % (uint32_t)s->total_people);
        person_infect(s, person, NULL, 0);
        if (i < i_incubating) continue;
        if (i < i_recovered_without_symptoms) {
            person_recover(s, person);
            continue;
        }
        person_become_ill(s, person);
        if (i < i_ill_at_home) continue;
        if (i < i_dead) {
            person_die(s, person);
            continue;
        }
        if (i < i_in_icu) {
            person_hospitalize(s, person);
            if (person->state != HOSPITALIZED) return 1;   /* the assertion of the reference */
            person_transfer_to_icu(s, person);
            continue;
        }
        if (i < i_in_ward) {
            person_hospitalize(s, person);
            continue;
        }
        person_recover(s, person);
    }
    for (int age = 0; age < 100 && age < s->nr_ages; age++) s->cnt[C_ALL_DETECTED][age] = 0;
    for (int i = 0; i < confirmed_cases; i++) {
        int age = (100 + i) % 100;
        if (age < s->nr_ages) s->cnt[C_ALL_DETECTED][age] += 1;
    }
    return 0;
}

/* Context._iterate main.pyx:1994-2009; returns the problem code (iterate() raises on != 0) */
int seq_iterate(Sim *s) {
    pop_init_day(s);
    s->total_infectors = 0;
    s->total_infections = 0;
    s->exposed_per_day = 0;
    hc_iterate(s);
    iterate_people(s);
    s->day += 1;
    return s->problem;
}

/* ---------------- construction / setters ---------------- */

/* Variant parameter block handed over by the driver: scalar floats + class tables as doubles
 * (main.pyx:820-850 converts to C float at assignment; conditional probabilities are divided in
 * Python double precision first, :808-843). */
typedef struct {
    double p_hospital_death_no_beds, p_icu_death_no_beds, infectiousness_multiplier,
        p_asymptomatic_infection, mean_incubation_duration, mean_duration_from_onset_to_death,
        mean_duration_from_onset_to_recovery, ratio_of_duration_in_ward,
        ratio_of_duration_before_hospitalisation, p_mask_protects_others, p_mask_protects_wearer;
    int n_sus, n_sym, n_sev, n_cri, n_fat, n_doh, n_iot;
    int c_sus[MAX_CLASSES], c_sym[MAX_CLASSES], c_sev[MAX_CLASSES], c_cri[MAX_CLASSES],
        c_fat[MAX_CLASSES], c_doh[MAX_CLASSES], c_iot[N_IOT];
    double v_sus[MAX_CLASSES], v_sym[MAX_CLASSES], v_sev[MAX_CLASSES], v_cri[MAX_CLASSES],
        v_fat[MAX_CLASSES], v_doh[MAX_CLASSES], v_iot[N_IOT];
} SeqVariantParams;

static void variant_init(Variant *v, const SeqVariantParams *p) {
    v->p_hospital_death_no_beds = (float)p->p_hospital_death_no_beds;
    v->p_icu_death_no_beds = (float)p->p_icu_death_no_beds;
    v->infectiousness_multiplier = (float)p->infectiousness_multiplier;
    v->p_asymptomatic_infection = (float)p->p_asymptomatic_infection;
    v->mean_incubation_duration = (float)p->mean_incubation_duration;
    v->mean_duration_from_onset_to_death = (float)p->mean_duration_from_onset_to_death;
    v->mean_duration_from_onset_to_recovery = (float)p->mean_duration_from_onset_to_recovery;
    v->ratio_of_duration_in_ward = (float)p->ratio_of_duration_in_ward;
    v->ratio_of_duration_before_hospitalisation = (float)p->ratio_of_duration_before_hospitalisation;
    v->p_mask_protects_others = (float)p->p_mask_protects_others;
    v->p_mask_protects_wearer = (float)p->p_mask_protects_wearer;
    cv_init(&v->p_susceptibility, p->n_sus, p->c_sus, p->v_sus);
    cv_init(&v->p_symptomatic, p->n_sym, p->c_sym, p->v_sym);
    cv_init(&v->p_severe, p->n_sev, p->c_sev, p->v_sev);
    cv_init(&v->p_critical, p->n_cri, p->c_cri, p->v_cri);
    cv_init(&v->p_fatal, p->n_fat, p->c_fat, p->v_fat);
    cv_init(&v->p_death_outside_hospital, p->n_doh, p->c_doh, p->v_doh);
    cv_init(&v->infectiousness_over_time, p->n_iot, p->c_iot, p->v_iot);
}

/* Population.__init__/_create_agents main.pyx:1354-1450. `perm` is the legacy
 * np.random.seed(seed); np.random.shuffle(arange(N)) permutation, computed by the driver. */
Sim *seq_create(int nr_ages, const int32_t *age_counts, const int32_t *perm,
                const uint64_t *pcg_state /* state_hi, state_lo, inc_hi, inc_lo */,
                int nr_variants, const SeqVariantParams *vparams, int n_import_classes,
                const int *import_classes, const double *import_cum, int beds, int icu_units) {
    Sim *s = (Sim *)calloc(1, sizeof(Sim));
    s->nr_ages = nr_ages;
    long total = 0;
    for (int a = 0; a < nr_ages; a++) total += age_counts[a];
    s->total_people = (int)total;
    s->people = (Person *)calloc((size_t)total, sizeof(Person));
    s->people_sorted_by_age = (int32_t *)malloc(sizeof(int32_t) * (size_t)total);
    s->age_start = (int32_t *)malloc(sizeof(int32_t) * (size_t)nr_ages);
    for (int c = 0; c < C_NR; c++) s->cnt[c] = (int *)calloc((size_t)nr_ages, sizeof(int));
    int idx = 0;
    for (int a = 0; a < nr_ages; a++) {
        s->age_start[a] = idx;
        s->cnt[C_SUSCEPTIBLE][a] = age_counts[a];
        for (int i = 0; i < age_counts[a]; i++) {
            int person_idx = perm[idx];
            Person *p = s->people + person_idx;
            /* person_init main.pyx:153-160 */
            p->idx = person_idx;
            p->age = (uint8_t)a;
            p->symptom_severity = ASYMPTOMATIC;
            p->state = SUSCEPTIBLE;
            p->infector = -1;
            p->infectees = NULL;
            p->day_of_vaccination = -1;
            s->people_sorted_by_age[idx] = person_idx;
            idx++;
        }
    }
    s->nr_variants = nr_variants;
    for (int v = 0; v < nr_variants; v++) variant_init(&s->variants[v], &vparams[v]);
    s->weekly_infections_amount = 0;
    s->weekly_infections_shares[0] = 1.0;
    cv_init(&s->imported_infection_ages, n_import_classes, import_classes, import_cum);
    s->nr_contacts_by_age = (double *)calloc((size_t)nr_ages, sizeof(double));
    s->cp_offset = (int *)calloc((size_t)nr_ages, sizeof(int));
    s->cp_count = (int *)calloc((size_t)nr_ages, sizeof(int));
    s->beds = s->available_beds = beds;
    s->icu_units = s->available_icu_units = icu_units;
    s->testing_mode = NO_TESTING;
    s->p_detected_anyway = 0;
    s->p_successful_tracing = 1.0f;
    pcg64_init(&s->rng, pcg_state[0], pcg_state[1], pcg_state[2], pcg_state[3]);
    s->problem = NO_PROBLEMOS;
    return s;
}

void seq_destroy(Sim *s) {
    for (int i = 0; i < s->total_people; i++) free(s->people[i].infectees);
    free(s->people);
    free(s->people_sorted_by_age);
    free(s->age_start);
    for (int c = 0; c < C_NR; c++) free(s->cnt[c]);
    free(s->nr_contacts_by_age);
    free(s->cp);
    free(s->cp_offset);
    free(s->cp_count);
    free(s->queue);
    free(s);
}

/* tables produced by the host-side builder (generate_contact_probabilities main.pyx:1184-1235) */
void seq_set_contact_tables(Sim *s, const double *nr_contacts_by_age, const int32_t *offsets,
                            const int32_t *counts, int n_entries, const int32_t *place,
                            const int32_t *cmin, const int32_t *cmax, const double *cum_p,
                            const float *mask_p) {
    free(s->cp);
    s->cp = (ContactProbability *)malloc(sizeof(ContactProbability) * (size_t)n_entries);
    for (int a = 0; a < s->nr_ages; a++) {
        s->nr_contacts_by_age[a] = nr_contacts_by_age[a];
        s->cp_offset[a] = offsets[a];
        s->cp_count[a] = counts[a];
    }
    for (int i = 0; i < n_entries; i++) {
        s->cp[i].place = place[i];
        s->cp[i].contact_age_min = cmin[i];
        s->cp[i].contact_age_max = cmax[i];
        s->cp[i].cum_p = cum_p[i];
        s->cp[i].mask_p = mask_p[i];
    }
}

/* HealthcareSystem.set_testing_mode main.pyx:623-628 */
void seq_set_testing_mode(Sim *s, int mode, double p) {
    s->testing_mode = mode;
    if (mode == ALL_WITH_SYMPTOMS_CT)
        s->p_successful_tracing = (float)p;
    else if (mode == ONLY_SEVERE_SYMPTOMS)
        s->p_detected_anyway = (float)p;
}
void seq_add_beds(Sim *s, int n) { s->beds += n; s->available_beds += n; }
void seq_add_icu_units(Sim *s, int n) { s->icu_units += n; s->available_icu_units += n; }
/* Population.infect_weekly main.pyx:1667-1669 */
void seq_infect_weekly(Sim *s, int amount, const double *shares) {
    s->weekly_infections_amount = amount;
    for (int v = 0; v < s->nr_variants; v++) s->weekly_infections_shares[v] = shares[v];
}
/* HealthcareSystem.start_vaccinating main.pyx:585-593 */
void seq_start_vaccinating(Sim *s, double nr_daily, int min_age, int max_age) {
    int k;
    for (k = 0; k < s->nr_vaccinations; k++)
        if (s->vaccinations[k].min_age == min_age && s->vaccinations[k].max_age == max_age) break;
    if (k == s->nr_vaccinations) {
        s->vaccinations[k].min_age = min_age;
        s->vaccinations[k].max_age = max_age;
        s->nr_vaccinations++;
    }
    s->vaccinations[k].nr_daily = nr_daily;
}

/* ---- state export: Context.generate_state main.pyx:1813-1857 (grouping done by the driver) ---- */
void seq_get_counters(Sim *s, int32_t *out /* [C_NR][nr_ages] */) {
    for (int c = 0; c < C_NR; c++)
        for (int a = 0; a < s->nr_ages; a++) out[c * s->nr_ages + a] = s->cnt[c][a];
}
/* scalars: available_icu, available_beds, icu_units, beds, total_infections, total_infectors,
 * exposed_per_day, ct_cases_per_day, day, problem, queue_len, unable_to_import */
void seq_get_scalars(Sim *s, int64_t *out) {
    out[0] = s->available_icu_units;
    out[1] = s->available_beds;
    out[2] = s->icu_units;
    out[3] = s->beds;
    out[4] = s->total_infections;
    out[5] = s->total_infectors;
    out[6] = s->exposed_per_day;
    out[7] = s->ct_cases_per_day;
    out[8] = s->day;
    out[9] = s->problem;
    out[10] = s->queue_len;
    out[11] = s->n_unable_to_import;
}
void seq_get_daily(Sim *s, int32_t *daily_contacts, int32_t *infected_by_variant) {
    for (int i = 0; i < NR_CONTACT_PLACES; i++) daily_contacts[i] = s->daily_contacts[i];
    for (int i = 0; i < s->nr_variants; i++) infected_by_variant[i] = s->infected_by_variant[i];
}

/* Context.sample main.pyx:2047-2101: works on a COPY of people[0] with age/severity overridden;
 * advances the context's RNG stream. `what`: 0 contacts_per_day, 1 symptom_severity,
 * 2 incubation_period, 3 illness_period, 4 hospitalization_period, 5 icu_period,
 * 6 onset_to_removed_period. severity < 0 means None (-> MILD). */
void seq_sample(Sim *s, int what, int age, int severity, int n, int32_t *out) {
    Person p = s->people[0];
    Contact contacts[MAX_CONTACTS];
    p.age = (uint8_t)age;
    p.symptom_severity = (uint8_t)(severity >= 0 ? severity : MILD);
    for (int i = 0; i < n; i++) {
        switch (what) {
        case 0: out[i] = get_contacts(s, &p, contacts, 1.0f, 100); break;
        case 1: out[i] = get_symptom_severity(s, &p); break;
        case 2: out[i] = get_incubation_days(s, &p); break;
        case 3:
            p.days_from_onset_to_removed = get_days_from_onset_to_removed(s, &p);
            out[i] = get_illness_days(s, &p);
            break;
        case 4:
            p.days_from_onset_to_removed = get_days_from_onset_to_removed(s, &p);
            out[i] = get_hospitalization_days(s, &p);
            break;
        case 5:
            p.days_from_onset_to_removed = get_days_from_onset_to_removed(s, &p);
            out[i] = get_icu_days(s, &p);
            break;
        case 6: out[i] = round_to_int(get_days_from_onset_to_removed(s, &p)); break;
        }
    }
}

/* RandomPool known-answer hook (tests/golden/rng_kat.npz): pattern chars d,u,l,g */
void seq_rng_pattern(const uint64_t *pcg_state, const char *pattern, int n, double a, double b,
                     double *out) {
    npy_pcg64 g;
    pcg64_init(&g, pcg_state[0], pcg_state[1], pcg_state[2], pcg_state[3]);
    for (int i = 0; i < n; i++) {
        switch (pattern[i]) {
        case 'd': out[i] = rp_get(&g); break;
        case 'u': out[i] = (double)rp_getint(&g); break;
        case 'l': out[i] = rp_lognormal(&g, a, b); break;
        case 'g': out[i] = (double)rp_gamma(&g, (float)a, (float)b); break;
        default: out[i] = -1; break;
        }
    }
}

int seq_sizeof_variant_params(void) { return (int)sizeof(SeqVariantParams); }
