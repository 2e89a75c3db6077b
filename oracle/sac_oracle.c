/*
 * oracle/sac_oracle.c -- CPU oracle for the SAC "free" sampling hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see sac_oracle.h).  Plain C restatement of
 * /root/reference/sac/free/sac_free.jl; every function cites the reference
 * lines it follows ("free:NNN" = sac/free/sac_free.jl line NNN).
 *
 * Build with -ffp-contract=off: the Julia loops carry no fused multiply-adds
 * (only the exp line is @fastmath), so a*b+c must round twice here too.
 */
#include "ora_common.h"

void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    philox4x32_10_impl(ctr, key, out);
}

double ora_log_u01(double u) { return log_u01_impl(u); }

double ora_chi2(const double *gbar, const double *g, const double *sinv, int n_tau)
{
    return chi2_impl(gbar, g, sinv, n_tau);
}

/* ------------------------------------------------------------------------ */
/* The sampler state (free:5-76, sampler fields only)                       */
/* ------------------------------------------------------------------------ */

struct ora_free {
    int par, symm, n_tau, n_omega;
    int64_t n_grid, wi0, wim;
    const double *K, *G, *sigma_inv;
    int64_t *wi;            /* omega_i_array (absolute grid indices) */
    double *A;              /* A_array */
    double *Gbar, *Gbar_new;
    double chi2, chi2_min;
    double accept_rates[6];
    double windows[4];
    double *spec;           /* sampled_spec[n_grid] */
    double *spec_tmp;       /* n_omega scratch for the reference histogram semantics */
    double sampled_chi2;
    int hist_mode;
    draws_t draws;
    ora_move_rec_t *log;
    int64_t log_cap, log_len;
    uint64_t counters[4];
};

static const double *col(const ora_free_t *s, int64_t w)
{
    return s->K + (w - s->wi0) * (int64_t)s->n_tau;   /* K_array[:, w - wi0 + 1] */
}

static void log_move(ora_free_t *s, int move, int evaluated, int accepted, double chi2_new)
{
    s->counters[0]++;
    if (evaluated) s->counters[1]++;
    if (s->log && s->log_len < s->log_cap) {
        ora_move_rec_t *r = &s->log[s->log_len];
        r->chi2_new = evaluated ? chi2_new : 0.0;
        r->flags = (evaluated ? 1 : 0) | (accepted ? 2 : 0);
        r->move = move;
    }
    s->log_len++;
}

/* free:310-318 (and the identical commit blocks of the other moves) */
static void commit(ora_free_t *s, double chi2_new)
{
    memcpy(s->Gbar, s->Gbar_new, sizeof(double) * (size_t)s->n_tau);
    s->chi2 = chi2_new;
    if (chi2_new < s->chi2_min) s->chi2_min = chi2_new;
}

/* free:274-324 */
static double single_w_move(ora_free_t *s, double inv2theta)
{
    int64_t accept = 0;
    const double window = s->windows[0];
    const int nt = s->n_tau;
    for (int64_t i = 0; i < s->n_omega; ++i) {
        prop_t p; prop_begin(&p, &s->draws, ORA_MOVE_SINGLE, (uint32_t)i);
        int64_t k = draw_R(&p, 0, s->n_omega);                  /* free:284 */
        int64_t dw = draw_shift(draw_U(&p, 1), window);         /* free:285 */
        int64_t w1 = (draw_U(&p, 3) < 0.5) ? s->wi[k] + dw : s->wi[k] - dw;  /* free:287-291 */
        if (w1 < s->wi0 || w1 > s->wim) {                       /* free:293-297 */
            log_move(s, ORA_MOVE_SINGLE, 0, 0, 0.0);
            continue;
        }
        const double A = s->A[k];
        const double *kn = col(s, w1), *ko = col(s, s->wi[k]);
        for (int j = 0; j < nt; ++j)                            /* free:300-303 */
            s->Gbar_new[j] = s->Gbar[j] + (A * (kn[j] - ko[j]));
        double chi2_new = ora_chi2(s->Gbar_new, s->G, s->sigma_inv, nt);   /* free:305 */
        int acc = metropolis(s->chi2, chi2_new, inv2theta, draw_U(&p, 5)); /* free:307-309 */
        if (acc) {
            s->wi[k] = w1;
            commit(s, chi2_new);
            accept += 1;
            s->counters[2] += 1; s->counters[3] += 1;
        }
        log_move(s, ORA_MOVE_SINGLE, 1, acc, chi2_new);
    }
    return (double)accept / (double)s->n_omega;                 /* free:323 */
}

/* free:328-397 */
static double double_w_move(ora_free_t *s, double inv2theta)
{
    int64_t accept = 0;
    const double window = s->windows[1];
    const int nt = s->n_tau;
    const int64_t n_it = s->n_omega / 2;
    for (int64_t i = 0; i < n_it; ++i) {
        prop_t p; prop_begin(&p, &s->draws, ORA_MOVE_DOUBLE, (uint32_t)i);
        int64_t k1 = draw_R(&p, 0, s->n_omega);
        int64_t k2 = draw_other(&p, 1, 0, s->n_omega, k1);
        if (k2 < 0) { log_move(s, ORA_MOVE_DOUBLE, 0, 0, 0.0); continue; }
        int64_t dw = draw_shift(draw_U(&p, 2), window);         /* free:343 */
        int64_t w1, w2;
        if (draw_U(&p, 4) < 0.5) { w1 = s->wi[k1] + dw; w2 = s->wi[k2] - dw; }  /* free:345-351 */
        else                     { w1 = s->wi[k1] - dw; w2 = s->wi[k2] + dw; }
        if (w1 < s->wi0 || w2 < s->wi0 || w1 > s->wim || w2 > s->wim) {   /* free:353-361 */
            log_move(s, ORA_MOVE_DOUBLE, 0, 0, 0.0);
            continue;
        }
        const double A1 = s->A[k1], A2 = s->A[k2];
        const double *kn1 = col(s, w1), *ko1 = col(s, s->wi[k1]);
        const double *kn2 = col(s, w2), *ko2 = col(s, s->wi[k2]);
        for (int j = 0; j < nt; ++j)                            /* free:366-370 */
            s->Gbar_new[j] = (s->Gbar[j] + A1 * (kn1[j] - ko1[j])) + A2 * (kn2[j] - ko2[j]);
        double chi2_new = ora_chi2(s->Gbar_new, s->G, s->sigma_inv, nt);
        int acc = metropolis(s->chi2, chi2_new, inv2theta, draw_U(&p, 6));
        if (acc) {
            s->wi[k1] = w1; s->wi[k2] = w2;
            commit(s, chi2_new);
            accept += 2;                                        /* free:389 */
            s->counters[2] += 1; s->counters[3] += 2;
        }
        log_move(s, ORA_MOVE_DOUBLE, 1, acc, chi2_new);
    }
    return (double)accept / (double)n_it;                       /* free:393 */
}

/* free:401-475 */
static double triple_w_move(ora_free_t *s, double inv2theta)
{
    int64_t accept = 0;
    const int nt = s->n_tau;
    const int64_t n_it = s->n_omega / 3;
    for (int64_t i = 0; i < n_it; ++i) {
        prop_t p; prop_begin(&p, &s->draws, ORA_MOVE_TRIPLE, (uint32_t)i);
        int64_t k1 = draw_R(&p, 0, s->n_omega);
        int64_t k2 = draw_other(&p, 1, 0, s->n_omega, k1);      /* free:413-416 */
        if (k2 < 0) { log_move(s, ORA_MOVE_TRIPLE, 0, 0, 0.0); continue; }
        int64_t k3 = draw_R(&p, 2, s->n_omega);                 /* free:417-420 */
        while (k3 == k1 || k3 == k2) {
            k3 = draw_R_retry(&p, &p.retry, 1, s->n_omega);
            if (k3 < 0) break;
        }
        if (k3 < 0) { log_move(s, ORA_MOVE_TRIPLE, 0, 0, 0.0); continue; }
        /* Julia Int64 `div` truncates toward zero, as C does            free:424 */
        int64_t dw = (s->wi[k2] + s->wi[k3] - (2 * s->wi[k1])) / 3;
        int64_t w1 = s->wi[k1] + 2 * dw;
        int64_t w2 = s->wi[k2] - dw;
        int64_t w3 = s->wi[k3] - dw;
        if (w1 < s->wi0 || w2 < s->wi0 || w3 < s->wi0 ||
            w1 > s->wim || w2 > s->wim || w3 > s->wim) {        /* free:430-442 */
            log_move(s, ORA_MOVE_TRIPLE, 0, 0, 0.0);
            continue;
        }
        const double A1 = s->A[k1], A2 = s->A[k2], A3 = s->A[k3];
        const double *kn1 = col(s, w1), *ko1 = col(s, s->wi[k1]);
        const double *kn2 = col(s, w2), *ko2 = col(s, s->wi[k2]);
        const double *kn3 = col(s, w3), *ko3 = col(s, s->wi[k3]);
        for (int j = 0; j < nt; ++j)                            /* free:447-452 */
            s->Gbar_new[j] = ((s->Gbar[j] + A1 * (kn1[j] - ko1[j]))
                                          + A2 * (kn2[j] - ko2[j]))
                                          + A3 * (kn3[j] - ko3[j]);
        double chi2_new = ora_chi2(s->Gbar_new, s->G, s->sigma_inv, nt);
        int acc = metropolis(s->chi2, chi2_new, inv2theta, draw_U(&p, 3));
        if (acc) {
            s->wi[k1] = w1; s->wi[k2] = w2; s->wi[k3] = w3;
            commit(s, chi2_new);
            accept += 1;                                        /* free:469 */
            s->counters[2] += 1; s->counters[3] += 3;
        }
        log_move(s, ORA_MOVE_TRIPLE, 1, acc, chi2_new);
    }
    return (double)accept / (double)n_it;
}

/* A_array ./= sum(A_array) (x2 if symm)   free:563-567, 721-725 */
static void renormalise(ora_free_t *s)
{
    double tot = strided_sum(s->A, s->n_omega);
    if (s->symm == 1) tot = 2 * tot;
    for (int64_t i = 0; i < s->n_omega; ++i) s->A[i] = s->A[i] / tot;
}

/* free:483-570 */
static double A_w_move(ora_free_t *s, double inv2theta)
{
    int64_t accept = 0;
    const double window = s->windows[3];
    const int nt = s->n_tau;
    const int64_t n_it = s->n_omega / 2;
    for (int64_t i = 0; i < n_it; ++i) {
        prop_t p; prop_begin(&p, &s->draws, ORA_MOVE_AW, (uint32_t)i);
        int64_t k1 = draw_R(&p, 0, s->n_omega);
        int64_t k2 = draw_other(&p, 1, 0, s->n_omega, k1);
        if (k2 < 0) { log_move(s, ORA_MOVE_AW, 0, 0, 0.0); continue; }
        int64_t dw = draw_shift(draw_U(&p, 2), window);         /* free:499 */
        int64_t w1, w2;
        if (draw_U(&p, 4) < 0.5) { w1 = s->wi[k1] + dw; w2 = s->wi[k2] - dw; }
        else                     { w1 = s->wi[k1] - dw; w2 = s->wi[k2] + dw; }
        if (w1 < s->wi0 || w2 < s->wi0 || w1 > s->wim || w2 > s->wim) {
            log_move(s, ORA_MOVE_AW, 0, 0, 0.0);
            continue;
        }
        const double A1 = s->A[k1], A2 = s->A[k2];
        /* free:522  (dw is the unsigned magnitude; integer denominator converted once) */
        double dA = ((double)dw * (A1 - A2)) / (double)(2 * dw + s->wi[k1] - s->wi[k2]);
        double A1p = A1 + dA, A2p = A2 - dA;
        if (A1p < 0 || A2p < 0) {                               /* free:528-532 */
            log_move(s, ORA_MOVE_AW, 0, 0, 0.0);
            continue;
        }
        const double *kn1 = col(s, w1), *ko1 = col(s, s->wi[k1]);
        const double *kn2 = col(s, w2), *ko2 = col(s, s->wi[k2]);
        for (int j = 0; j < nt; ++j)                            /* free:535-541 */
            s->Gbar_new[j] = (((s->Gbar[j] + (A1p * kn1[j])) + (A2p * kn2[j]))
                                           - (A1 * ko1[j])) - (A2 * ko2[j]);
        double chi2_new = ora_chi2(s->Gbar_new, s->G, s->sigma_inv, nt);
        int acc = metropolis(s->chi2, chi2_new, inv2theta, draw_U(&p, 6));
        if (acc) {
            s->wi[k1] = w1; s->wi[k2] = w2;
            s->A[k1] = A1p; s->A[k2] = A2p;
            commit(s, chi2_new);
            accept += 2;                                        /* free:560 */
            s->counters[2] += 1; s->counters[3] += 2;
        }
        log_move(s, ORA_MOVE_AW, 1, acc, chi2_new);
    }
    renormalise(s);
    return (double)accept / (double)n_it;
}

/* free:674-728 */
static double double_A_move(ora_free_t *s, double inv2theta)
{
    int64_t accept = 0;
    const int nt = s->n_tau;
    for (int64_t i = 0; i < s->n_omega; ++i) {
        prop_t p; prop_begin(&p, &s->draws, ORA_MOVE_DOUBLE_A, (uint32_t)i);
        int64_t k1 = draw_R(&p, 0, s->n_omega);
        /* free:683-686: second delta must sit at a different frequency */
        int64_t k2 = k1;
        int first = 1;
        while (s->wi[k1] == s->wi[k2]) {
            k2 = first ? draw_R(&p, 1, s->n_omega) : draw_R_retry(&p, &p.retry, 0, s->n_omega);
            first = 0;
            if (k2 < 0) break;
        }
        if (k2 < 0) { log_move(s, ORA_MOVE_DOUBLE_A, 0, 0, 0.0); continue; }
        double m0 = s->A[k1] + s->A[k2];                        /* free:688 */
        double r = draw_U(&p, 2);
        double dA1 = (r * m0) - s->A[k1];                       /* free:692-693 */
        double dA2 = ((1 - r) * m0) - s->A[k2];
        const double *c1 = col(s, s->wi[k1]), *c2 = col(s, s->wi[k2]);
        for (int j = 0; j < nt; ++j)                            /* free:696-700 */
            s->Gbar_new[j] = (s->Gbar[j] + (dA1 * c1[j])) + (dA2 * c2[j]);
        double chi2_new = ora_chi2(s->Gbar_new, s->G, s->sigma_inv, nt);
        int acc = metropolis(s->chi2, chi2_new, inv2theta, draw_U(&p, 4));
        if (acc) {
            s->A[k1] += dA1; s->A[k2] += dA2;                   /* free:708-709 */
            commit(s, chi2_new);
            accept += 1;
            s->counters[2] += 1; s->counters[3] += 2;
        }
        log_move(s, ORA_MOVE_DOUBLE_A, 1, acc, chi2_new);
    }
    renormalise(s);
    return (double)accept / (double)s->n_omega;
}

/* free:825-861 */
void ora_free_run_updates(ora_free_t *s, double theta)
{
    const double inv2theta = 1.0 / (2.0 * theta);
    s->accept_rates[0] += single_w_move(s, inv2theta);
    s->accept_rates[1] += double_w_move(s, inv2theta);
    s->accept_rates[2] += triple_w_move(s, inv2theta);
    if (s->par == 2) {
        s->accept_rates[3] += A_w_move(s, inv2theta);
        s->accept_rates[4] += double_A_move(s, inv2theta);
    }
    s->draws.sweep++;
}

/* free:866-899.  The calc_Gbar/calc_chi2 at free:876-877 assign to unused locals:
 * Gbar is NOT refreshed, so neither is it here. */
void ora_free_adjust_windows(ora_free_t *s, int64_t steps, double theta)
{
    const int64_t per = steps / 10;
    for (int j = 0; j < 10; ++j) {
        for (int k = 0; k < 6; ++k) s->accept_rates[k] = 0;
        for (int64_t i = 0; i < per; ++i) ora_free_run_updates(s, theta);
        for (int k = 0; k < 6; ++k) s->accept_rates[k] /= (double)per;
        for (int k = 0; k < 4; ++k) {                           /* free:885-895 */
            double a = s->accept_rates[k];
            if (a > 0.8)       s->windows[k] *= 2;
            else if (a < 0.2)  s->windows[k] /= 2;
            else if (a > 0.55) s->windows[k] *= 1.2;
            else if (a < 0.45) s->windows[k] /= 1.2;
        }
    }
}

/* free:903-927 */
void ora_free_sample(ora_free_t *s, int64_t steps, double theta)
{
    memset(s->spec, 0, sizeof(double) * (size_t)s->n_grid);
    s->sampled_chi2 = 0;
    for (int k = 0; k < 6; ++k) s->accept_rates[k] = 0;
    for (int64_t j = 0; j < steps; ++j) {
        ora_free_run_updates(s, theta);
        if (s->hist_mode == ORA_HIST_REFERENCE) {
            /* free:918  spec[idx] .+= A  ==  spec[idx] .= spec[idx] .+ A : the right-hand
             * side is gathered first, then scattered; duplicates in idx do not accumulate,
             * the last delta at a frequency wins. */
            for (int64_t i = 0; i < s->n_omega; ++i)
                s->spec_tmp[i] = s->spec[s->wi[i] - s->wi0] + s->A[i];
            for (int64_t i = 0; i < s->n_omega; ++i)
                s->spec[s->wi[i] - s->wi0] = s->spec_tmp[i];
        } else {
            for (int64_t i = 0; i < s->n_omega; ++i)
                s->spec[s->wi[i] - s->wi0] += s->A[i];
        }
        s->sampled_chi2 += s->chi2;                             /* free:919 */
    }
    for (int64_t i = 0; i < s->n_grid; ++i) s->spec[i] /= (double)steps;   /* free:924-926 */
    s->sampled_chi2 /= (double)steps;
    for (int k = 0; k < 6; ++k) s->accept_rates[k] /= (double)steps;
}

/* ------------------------------------------------------------------------ */
/* Plumbing                                                                 */
/* ------------------------------------------------------------------------ */

ora_free_t *ora_free_create(int par, int symm, int n_tau, int n_omega,
                            int64_t n_grid, int64_t wi0, int64_t wim,
                            const double *K, const double *G, const double *sigma_inv)
{
    if (n_tau < 1 || n_tau > 128 || n_omega < 3 || n_grid < 1) return NULL;
    ora_free_t *s = (ora_free_t *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->par = par; s->symm = symm; s->n_tau = n_tau; s->n_omega = n_omega;
    s->n_grid = n_grid; s->wi0 = wi0; s->wim = wim;
    s->K = K;
    double *g = (double *)malloc(sizeof(double) * 2 * (size_t)n_tau);
    memcpy(g, G, sizeof(double) * (size_t)n_tau);
    memcpy(g + n_tau, sigma_inv, sizeof(double) * (size_t)n_tau);
    s->G = g; s->sigma_inv = g + n_tau;
    s->wi = (int64_t *)calloc((size_t)n_omega, sizeof(int64_t));
    s->A = (double *)calloc((size_t)n_omega, sizeof(double));
    s->Gbar = (double *)calloc((size_t)n_tau, sizeof(double));
    s->Gbar_new = (double *)calloc((size_t)n_tau, sizeof(double));
    s->spec = (double *)calloc((size_t)n_grid, sizeof(double));
    s->spec_tmp = (double *)calloc((size_t)n_omega, sizeof(double));
    s->draws.mode = ORA_DRAW_PHILOX;
    return s;
}

void ora_free_destroy(ora_free_t *s)
{
    if (!s) return;
    free((void *)s->G); free(s->wi); free(s->A); free(s->Gbar); free(s->Gbar_new);
    free(s->spec); free(s->spec_tmp); free(s);
}

void ora_free_set_state(ora_free_t *s, const int64_t *wi, const double *A, const double *windows)
{
    if (wi) memcpy(s->wi, wi, sizeof(int64_t) * (size_t)s->n_omega);
    if (A) memcpy(s->A, A, sizeof(double) * (size_t)s->n_omega);
    if (windows) memcpy(s->windows, windows, sizeof(double) * 4);
}

void ora_free_get_state(const ora_free_t *s, int64_t *wi, double *A, double *windows)
{
    if (wi) memcpy(wi, s->wi, sizeof(int64_t) * (size_t)s->n_omega);
    if (A) memcpy(A, s->A, sizeof(double) * (size_t)s->n_omega);
    if (windows) memcpy(windows, s->windows, sizeof(double) * 4);
}

/* free:243-246 calc_Gbar: Gbar = K[:, idx] * A.  Julia hands this to BLAS gemv whose
 * summation order is unspecified; here: per tau, sequential over the deltas. */
void ora_free_init_gbar(ora_free_t *s)
{
    for (int j = 0; j < s->n_tau; ++j) {
        double acc = 0.0;
        for (int64_t i = 0; i < s->n_omega; ++i)
            acc = acc + col(s, s->wi[i])[j] * s->A[i];
        s->Gbar[j] = acc;
    }
    s->chi2 = ora_chi2(s->Gbar, s->G, s->sigma_inv, s->n_tau);
    s->chi2_min = s->chi2;
}

void ora_free_set_gbar(ora_free_t *s, const double *gbar, double chi2, double chi2_min)
{
    memcpy(s->Gbar, gbar, sizeof(double) * (size_t)s->n_tau);
    s->chi2 = chi2; s->chi2_min = chi2_min;
}

void ora_free_get_gbar(const ora_free_t *s, double *gbar, double *chi2, double *chi2_min)
{
    if (gbar) memcpy(gbar, s->Gbar, sizeof(double) * (size_t)s->n_tau);
    if (chi2) *chi2 = s->chi2;
    if (chi2_min) *chi2_min = s->chi2_min;
}

void ora_free_use_philox(ora_free_t *s, uint64_t seed, uint32_t chain, uint64_t sweep0)
{
    s->draws.mode = ORA_DRAW_PHILOX;
    s->draws.seed_lo = (uint32_t)seed; s->draws.seed_hi = (uint32_t)(seed >> 32);
    s->draws.chain = chain; s->draws.sweep = sweep0;
}

void ora_free_use_trace(ora_free_t *s, const double *trace, int64_t len)
{
    s->draws.mode = ORA_DRAW_TRACE;
    s->draws.trace = trace; s->draws.trace_len = len; s->draws.trace_pos = 0;
}

int64_t ora_free_trace_pos(const ora_free_t *s) { return s->draws.trace_pos; }
uint64_t ora_free_sweep_counter(const ora_free_t *s) { return s->draws.sweep; }

void ora_free_set_log(ora_free_t *s, ora_move_rec_t *log, int64_t cap)
{
    s->log = log; s->log_cap = cap; s->log_len = 0;
}

int64_t ora_free_log_len(const ora_free_t *s) { return s->log_len; }
void ora_free_set_hist_mode(ora_free_t *s, int mode) { s->hist_mode = mode; }

void ora_free_get_accept_rates(const ora_free_t *s, double out[6])
{
    memcpy(out, s->accept_rates, sizeof(double) * 6);
}

double ora_free_sampled_chi2(const ora_free_t *s) { return s->sampled_chi2; }

void ora_free_get_spec(const ora_free_t *s, double *out)
{
    memcpy(out, s->spec, sizeof(double) * (size_t)s->n_grid);
}

void ora_free_get_counters(const ora_free_t *s, uint64_t out[4])
{
    memcpy(out, s->counters, sizeof(uint64_t) * 4);
}
