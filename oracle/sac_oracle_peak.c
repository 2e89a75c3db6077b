/*
 * oracle/sac_oracle_peak.c -- CPU oracle for the SAC "peak" parameterisation.
 *
 * TEST INFRASTRUCTURE ONLY (see sac_oracle.h).  Plain-C restatement of
 * /root/reference/sac/peak/sac_peak.jl ("peak:NNN" = line NNN of that file): the delta-peak
 * + continuum sampler with four delta groups (peak+, continuum+, peak-, continuum-), two kernel
 * tables Kp/Kn on the half-offset grid, the mutual ordering constraint peak < continuum with
 * lazily refreshed bounds, and the +/- weight-transfer moves.
 * Build with -ffp-contract=off.
 */
#include "ora_common.h"

enum { PK_SINGLE = 11, PK_DOUBLE = 13, PK_TRIPLE = 15, PK_PMOVE = 17,
       PK_CONT_X = 19, PK_PEAK_X = 20, PK_AW_X = 21 };

struct ora_peak {
    int n_tau, n_omega, n_p, n_tot;
    int symm, bosonic, fix_edge, indiv_update;
    int64_t wi0, wim, n_grid;
    const double *Kp, *Kn, *G, *sigma_inv;
    int64_t *wi;
    double *A;
    double *Gbar, *Gbar_new;
    double chi2, chi2_min;
    double wi_pp, wi_np;        /* Float64 fields of the reference (peak:14,16) */
    int64_t wi_pc, wi_nc;
    double accept_rates[11], windows[11];
    double *pspec[2], *cspec[2];    /* [pn][n_grid] */
    double sampled_chi2;
    double edge_res[3][2];
    draws_t draws;
    ora_move_rec_t *log;
    int64_t log_cap, log_len;
    uint64_t counters[5];
};

/* group index ranges, 0-based half-open (peak:166-169) */
static int64_t g_first(const ora_peak_t *s, int grp)
{
    switch (grp) {
    case 0: return 0;                                   /* peak_p */
    case 1: return s->n_p;                              /* cont_p */
    case 2: return s->n_p + s->n_omega;                 /* peak_n */
    default: return 2 * s->n_p + s->n_omega;            /* cont_n */
    }
}
static int64_t g_count(const ora_peak_t *s, int grp) { return (grp & 1) ? s->n_omega : s->n_p; }

static const double *colp(const ora_peak_t *s, int64_t w) { return s->Kp + w * (int64_t)s->n_tau; }
static const double *coln(const ora_peak_t *s, int64_t w) { return s->Kn + w * (int64_t)s->n_tau; }

static void plog(ora_peak_t *s, int move, int evaluated, int accepted, double chi2_new)
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

static void pcommit(ora_peak_t *s, double chi2_new, int n_deltas, int n_cols)
{
    memcpy(s->Gbar, s->Gbar_new, sizeof(double) * (size_t)s->n_tau);
    s->chi2 = chi2_new;
    if (chi2_new < s->chi2_min) s->chi2_min = chi2_new;
    s->counters[2] += 1;
    s->counters[3] += (uint64_t)n_deltas;
    (void)n_cols;
}

static int64_t gmin(const ora_peak_t *s, int grp)
{
    int64_t f = g_first(s, grp), m = s->wi[f];
    for (int64_t i = 1; i < g_count(s, grp); ++i) if (s->wi[f + i] < m) m = s->wi[f + i];
    return m;
}
static int64_t gmax(const ora_peak_t *s, int grp)      /* peak_bound, peak:141-147 */
{
    int64_t f = g_first(s, grp), m = s->wi[f];
    for (int64_t i = 1; i < g_count(s, grp); ++i) if (s->wi[f + i] > m) m = s->wi[f + i];
    return m;
}

/* peak:347-353 calc_Gbar: four matrix-vector products added left to right.  Julia hands each to
 * BLAS (order unspecified); here: per tau, sequential over the group's deltas.  Groups on the
 * unsampled negative axis have zero amplitude and are skipped (adds exact zeros). */
static void p_calc_gbar(ora_peak_t *s)
{
    const int sample_neg = !(s->symm == 1 || s->bosonic);
    for (int j = 0; j < s->n_tau; ++j) {
        double tot = 0.0;
        for (int grp = 0; grp < 4; ++grp) {
            if (grp >= 2 && !sample_neg) continue;
            double acc = 0.0;
            const int64_t f = g_first(s, grp);
            for (int64_t i = 0; i < g_count(s, grp); ++i) {
                const double *c = (grp < 2) ? colp(s, s->wi[f + i]) : coln(s, s->wi[f + i]);
                acc = acc + c[j] * s->A[f + i];
            }
            tot = (grp == 0) ? acc : tot + acc;
        }
        s->Gbar[j] = tot;
    }
}

/* peak:382-447 / 450-532 / 535-625: continuum frequency moves on axis pn (1 or 2) */
static double cont_move(ora_peak_t *s, double inv2theta, int pn, int ncol)
{
    const int grp = (pn == 1) ? 1 : 3;
    const int64_t first = g_first(s, grp), n = s->n_omega;
    const double wb = (pn == 1) ? s->wi_pp : s->wi_np;            /* peak:390,394 */
    const int64_t lo = (int64_t)floor(wb);                        /* new <= floor(w_b): continue */
    const double window = s->windows[(ncol == 1 ? 0 : 1) + (pn - 1) * 7];
    const int move = (ncol == 1 ? PK_SINGLE : ncol == 2 ? PK_DOUBLE : PK_TRIPLE) + (pn - 1);
    const int64_t n_it = (ncol == 1) ? n : (ncol == 2) ? n / 2 : n / 3;
    const int nt = s->n_tau;
    int64_t accept = 0;
    for (int64_t i = 0; i < n_it; ++i) {
        prop_t p; prop_begin(&p, &s->draws, (uint32_t)move, (uint32_t)i);
        int64_t k[3], wn[3];
        k[0] = first + draw_R(&p, 0, n);
        if (ncol >= 2) {
            int64_t r = draw_other(&p, 1, 0, n, k[0] - first);
            if (r < 0) { plog(s, move, 0, 0, 0.0); continue; }
            k[1] = first + r;
        }
        if (ncol == 3) {
            int64_t r = draw_R(&p, 2, n);
            while (first + r == k[0] || first + r == k[1]) {
                r = draw_R_retry(&p, &p.retry, 1, n);
                if (r < 0) break;
            }
            if (r < 0) { plog(s, move, 0, 0, 0.0); continue; }
            k[2] = first + r;
        }
        if (ncol == 1) {
            int64_t dw = draw_shift(draw_U(&p, 1), window);
            wn[0] = (draw_U(&p, 3) < 0.5) ? s->wi[k[0]] + dw : s->wi[k[0]] - dw;
        } else if (ncol == 2) {
            int64_t dw = draw_shift(draw_U(&p, 2), window);
            if (draw_U(&p, 4) < 0.5) { wn[0] = s->wi[k[0]] + dw; wn[1] = s->wi[k[1]] - dw; }
            else                     { wn[0] = s->wi[k[0]] - dw; wn[1] = s->wi[k[1]] + dw; }
        } else {
            int64_t dw = (s->wi[k[1]] + s->wi[k[2]] - (2 * s->wi[k[0]])) / 3;     /* peak:566 */
            wn[0] = s->wi[k[0]] + 2 * dw; wn[1] = s->wi[k[1]] - dw; wn[2] = s->wi[k[2]] - dw;
        }
        int ok = 1;
        for (int q = 0; q < ncol; ++q) if (wn[q] <= lo || wn[q] > s->wim) ok = 0;   /* peak:409-413 */
        if (!ok) { plog(s, move, 0, 0, 0.0); continue; }
        for (int j = 0; j < nt; ++j) {                            /* peak:416-419, 497-501, 589-594 */
            double g = s->Gbar[j];
            for (int q = 0; q < ncol; ++q) {
                const double *kn = (pn == 1) ? colp(s, wn[q]) : coln(s, wn[q]);
                const double *ko = (pn == 1) ? colp(s, s->wi[k[q]]) : coln(s, s->wi[k[q]]);
                g = g + s->A[k[q]] * (kn[j] - ko[j]);
            }
            s->Gbar_new[j] = g;
        }
        double chi2_new = chi2_impl(s->Gbar_new, s->G, s->sigma_inv, nt);
        int acc = metropolis(s->chi2, chi2_new, inv2theta,
                             draw_U(&p, ncol == 1 ? 5 : ncol == 2 ? 6 : 3));
        s->counters[4] += 2 * (uint64_t)ncol;
        if (acc) {
            for (int q = 0; q < ncol; ++q) s->wi[k[q]] = wn[q];
            pcommit(s, chi2_new, ncol, 2 * ncol);
            accept += (ncol == 2) ? 2 : 1;                        /* peak:434, 518, 611 */
        }
        plog(s, move, 1, acc, chi2_new);
    }
    if (pn == 1) s->wi_pc = gmin(s, 1); else s->wi_nc = gmin(s, 3);   /* peak:440-444 */
    return (double)accept / (double)n_it;
}

/* peak:628-700 single_w_pmove */
static double peak_move(ora_peak_t *s, double inv2theta, int pn)
{
    const int grp = (pn == 1) ? 0 : 2;
    const int64_t first = g_first(s, grp), n = s->n_p;
    const int64_t wb = (pn == 1) ? s->wi_pc : s->wi_nc;           /* peak:636,641 */
    const double window = s->windows[pn == 1 ? 3 : 4];
    const int move = PK_PMOVE + (pn - 1);
    const int64_t n_it = 10 * n;
    const int nt = s->n_tau;
    int64_t accept = 0;
    for (int64_t i = 0; i < n_it; ++i) {
        prop_t p; prop_begin(&p, &s->draws, (uint32_t)move, (uint32_t)i);
        int64_t k = first + draw_R(&p, 0, n);
        int64_t dw = draw_shift(draw_U(&p, 1), window);
        int64_t w1 = (draw_U(&p, 3) < 0.5) ? s->wi[k] + dw : s->wi[k] - dw;
        if (w1 < s->wi0 || w1 > wb) { plog(s, move, 0, 0, 0.0); continue; }   /* peak:659-663 */
        const double A = s->A[k];
        const double *kn = (pn == 1) ? colp(s, w1) : coln(s, w1);
        const double *ko = (pn == 1) ? colp(s, s->wi[k]) : coln(s, s->wi[k]);
        for (int j = 0; j < nt; ++j) s->Gbar_new[j] = s->Gbar[j] + (A * (kn[j] - ko[j]));
        double chi2_new = chi2_impl(s->Gbar_new, s->G, s->sigma_inv, nt);
        int acc = metropolis(s->chi2, chi2_new, inv2theta, draw_U(&p, 5));
        s->counters[4] += 2;
        if (acc) {
            s->wi[k] = w1;
            pcommit(s, chi2_new, 1, 2);
            accept += 1;
        }
        plog(s, move, 1, acc, chi2_new);
    }
    if (pn == 1) s->wi_pp = (double)gmax(s, 0); else s->wi_np = (double)gmax(s, 2);   /* peak:693-697 */
    return (double)accept / (double)n_it;
}

/* dA = dw (A_n - A_p) / (2 dw + w_p + w_n), dw signed   (peak:745, 831, 939, 947) */
static double transfer_dA(int64_t dw, double A_p, double A_n, int64_t w_p, int64_t w_n)
{
    return ((double)dw * (A_n - A_p)) / (double)(2 * dw + w_p - (-w_n));
}

/* peak:706-792 cont_Aw_transfer (which = 0) and peak:798-880 peak_Aw_transfer (which = 1) */
static double pair_transfer(ora_peak_t *s, double inv2theta, int which)
{
    const int gp = which ? 0 : 1, gn = which ? 2 : 3;
    const int64_t n_it = which ? 10 : s->n_omega / 2;
    const double window = s->windows[which ? 6 : 5];
    const int move = which ? PK_PEAK_X : PK_CONT_X;
    const int nt = s->n_tau;
    double accept = 0.;
    for (int64_t i = 0; i < n_it; ++i) {
        prop_t p; prop_begin(&p, &s->draws, (uint32_t)move, (uint32_t)i);
        int64_t kp = g_first(s, gp) + draw_R(&p, 0, g_count(s, gp));
        int64_t kn_ = g_first(s, gn) + draw_R(&p, 1, g_count(s, gn));
        int64_t w_p = s->wi[kp], w_n = s->wi[kn_];
        int64_t dw = draw_shift(draw_U(&p, 2), window);           /* peak:724, 810 */
        if (draw_U(&p, 4) < 0.5) dw = -dw;
        int64_t w_pp = w_p + dw, w_np = w_n - dw;
        int ok;
        if (!which) {   /* peak:734-740 (the n-side test also uses wi_pp, as the reference does) */
            ok = !(w_pp > s->wim || w_pp < s->wi0 || (double)w_pp < s->wi_pp) &&
                 !(w_np > s->wim || w_np < s->wi0 || (double)w_np < s->wi_pp);
        } else {        /* peak:820-825 */
            ok = !(w_pp < s->wi0 || w_pp > s->wi_pc || w_pp > s->wim) &&
                 !(w_np < s->wi0 || w_np > s->wi_nc || w_np > s->wim);
        }
        if (!ok) { plog(s, move, 0, 0, 0.0); continue; }
        const double A_p = s->A[kp], A_n = s->A[kn_];
        const double dA = transfer_dA(dw, A_p, A_n, w_p, w_n);
        const double A_pp = A_p + dA, A_np = A_n - dA;
        if (A_pp < 0 || A_np < 0) { plog(s, move, 0, 0, 0.0); continue; }    /* peak:751-755 */
        const double *c1 = colp(s, w_pp), *c2 = coln(s, w_np), *c3 = colp(s, w_p), *c4 = coln(s, w_n);
        for (int j = 0; j < nt; ++j)                              /* peak:758-764 */
            s->Gbar_new[j] = (((s->Gbar[j] + (A_pp * c1[j])) + (A_np * c2[j])) - (A_p * c3[j])) - (A_n * c4[j]);
        double chi2_new = chi2_impl(s->Gbar_new, s->G, s->sigma_inv, nt);
        int acc = metropolis(s->chi2, chi2_new, inv2theta, draw_U(&p, 6));
        s->counters[4] += 4;
        if (acc) {
            s->wi[kp] = w_pp; s->wi[kn_] = w_np;
            s->A[kp] = A_pp; s->A[kn_] = A_np;
            pcommit(s, chi2_new, 2, 4);
            accept += 1.;
        }
        plog(s, move, 1, acc, chi2_new);
    }
    if (!which) { s->wi_pc = gmin(s, 1); s->wi_nc = gmin(s, 3); }             /* peak:787-788 */
    else { s->wi_pp = (double)gmax(s, 0); s->wi_np = (double)gmax(s, 2); }    /* peak:876-877 */
    return accept / (double)n_it;
}

/* peak:882-996 Aw_transfer: continuum pair and peak pair move together */
static double full_transfer(ora_peak_t *s, double inv2theta)
{
    const int64_t n_it = s->n_omega / 2;
    const double cw = s->windows[10], pw = s->windows[10];        /* peak:885-886 */
    const int nt = s->n_tau;
    double accept = 0.;
    for (int64_t i = 0; i < n_it; ++i) {
        prop_t p; prop_begin4(&p, &s->draws, PK_AW_X, (uint32_t)i);
        int64_t k_pc = g_first(s, 1) + draw_R(&p, 0, s->n_omega);
        int64_t k_nc = g_first(s, 3) + draw_R(&p, 1, s->n_omega);
        int64_t w_pc = s->wi[k_pc], w_nc = s->wi[k_nc];
        int64_t k_pp = g_first(s, 0) + draw_R(&p, 2, s->n_p);
        int64_t k_np = g_first(s, 2) + draw_R(&p, 3, s->n_p);
        int64_t w_pp = s->wi[k_pp], w_np = s->wi[k_np];
        int64_t dc = draw_shift(draw_U(&p, 4), cw);               /* peak:903-904 */
        int64_t dp = draw_shift(draw_U(&p, 6), pw);
        if (draw_U(&p, 8) < 0.5) dc = -dc;                        /* peak:906-912 */
        if (draw_U(&p, 10) < 0.5) dp = -dp;
        int64_t n_pc = w_pc + dc, n_nc = w_nc - dc;
        int ok = !(n_pc > s->wim || (double)n_pc < s->wi_pp || n_pc < s->wi0) &&     /* peak:918-923 */
                 !(n_nc > s->wim || (double)n_nc < s->wi_np || n_nc < s->wi0);
        int64_t n_pp = w_pp + dp, n_np = w_np - dp;
        ok = ok && !(n_pp > s->wi_pc || n_pp > s->wim || n_pp < s->wi0) &&           /* peak:929-934 */
                   !(n_np > s->wi_nc || n_np > s->wim || n_np < s->wi0);
        if (!ok) { plog(s, PK_AW_X, 0, 0, 0.0); continue; }
        const double A_pc = s->A[k_pc], A_nc = s->A[k_nc];
        double dA = transfer_dA(dc, A_pc, A_nc, w_pc, w_nc);      /* peak:939 */
        const double A_pc2 = A_pc + dA, A_nc2 = A_nc - dA;
        const double A_pp = s->A[k_pp], A_np = s->A[k_np];
        dA = transfer_dA(dp, A_pp, A_np, w_pp, w_np);             /* peak:947 */
        const double A_pp2 = A_pp + dA, A_np2 = A_np - dA;
        if (A_pc2 < 0 || A_nc2 < 0 || A_pp2 < 0 || A_np2 < 0) { plog(s, PK_AW_X, 0, 0, 0.0); continue; }
        const double *a1 = colp(s, n_pc), *a2 = coln(s, n_nc), *a3 = colp(s, w_pc), *a4 = coln(s, w_nc);
        const double *b1 = colp(s, n_pp), *b2 = coln(s, n_np), *b3 = colp(s, w_pp), *b4 = coln(s, w_np);
        for (int j = 0; j < nt; ++j)                              /* peak:957-967 */
            s->Gbar_new[j] = (((((((s->Gbar[j] + (A_pc2 * a1[j])) + (A_nc2 * a2[j])) - (A_pc * a3[j]))
                                 - (A_nc * a4[j])) + (A_pp2 * b1[j])) + (A_np2 * b2[j])) - (A_pp * b3[j]))
                             - (A_np * b4[j]);
        double chi2_new = chi2_impl(s->Gbar_new, s->G, s->sigma_inv, nt);
        int acc = metropolis(s->chi2, chi2_new, inv2theta, draw_U(&p, 12));
        s->counters[4] += 8;
        if (acc) {
            s->wi[k_pc] = n_pc; s->wi[k_nc] = n_nc; s->A[k_pc] = A_pc2; s->A[k_nc] = A_nc2;
            s->wi[k_pp] = n_pp; s->wi[k_np] = n_np; s->A[k_pp] = A_pp2; s->A[k_np] = A_np2;
            pcommit(s, chi2_new, 4, 8);
            accept += 1.;
        }
        plog(s, PK_AW_X, 1, acc, chi2_new);
    }
    return accept / (double)n_it;       /* bounds are not refreshed here (peak:995) */
}

/* peak:1006-1059 */
void ora_peak_run_updates(ora_peak_t *s, double theta, int transfer)
{
    const double inv2theta = 1.0 / (2.0 * theta);
    int sample_neg = 1;
    if (s->symm == 1 || s->bosonic) { sample_neg = 0; transfer = 0; }
    if (transfer) {
        if (s->indiv_update) {
            s->accept_rates[5] += pair_transfer(s, inv2theta, 0);
            s->accept_rates[6] += pair_transfer(s, inv2theta, 1);
        }
        s->accept_rates[10] += full_transfer(s, inv2theta);
    }
    for (int pn = 1; pn <= (sample_neg ? 2 : 1); ++pn) {
        s->accept_rates[0 + (pn - 1) * 7] += cont_move(s, inv2theta, pn, 1);
        s->accept_rates[1 + (pn - 1) * 7] += cont_move(s, inv2theta, pn, 2);
        s->accept_rates[2 + (pn - 1) * 7] += cont_move(s, inv2theta, pn, 3);
    }
    if (s->fix_edge != 1) {
        s->accept_rates[3] += peak_move(s, inv2theta, 1);
        if (sample_neg) s->accept_rates[4] += peak_move(s, inv2theta, 2);
    }
    s->draws.sweep++;
}

static void refresh(ora_peak_t *s, int bounds)
{
    p_calc_gbar(s);                                               /* peak:1072-1073, 1120-1121 */
    s->chi2 = chi2_impl(s->Gbar, s->G, s->sigma_inv, s->n_tau);
    if (bounds) { s->wi_pp = (double)gmax(s, 0); s->wi_np = (double)gmax(s, 2); }
}

/* peak:1063-1096 */
void ora_peak_adjust_windows(ora_peak_t *s, int64_t steps, double theta)
{
    const int64_t per = steps / 10;
    for (int j = 0; j < 10; ++j) {
        for (int k = 0; k < 11; ++k) s->accept_rates[k] = 0;
        for (int64_t i = 0; i < per; ++i) {
            refresh(s, 1);
            ora_peak_run_updates(s, theta, 1);
        }
        for (int k = 0; k < 11; ++k) s->accept_rates[k] /= (double)per;
        for (int k = 0; k < 11; ++k) {                            /* peak:1081-1091 */
            if (s->accept_rates[k] > 0.55) s->windows[k] *= 1.2;
            else if (s->accept_rates[k] < 0.45) s->windows[k] /= 1.2;
        }
    }
}

/* peak:1100-1166 */
void ora_peak_sample(ora_peak_t *s, int64_t steps, double theta, int bin)
{
    int transfer;
    if (!bin) {
        for (int pn = 0; pn < 2; ++pn) {
            memset(s->pspec[pn], 0, sizeof(double) * (size_t)s->n_grid);
            memset(s->cspec[pn], 0, sizeof(double) * (size_t)s->n_grid);
        }
        transfer = 1;
    } else {
        transfer = 0;
    }
    s->sampled_chi2 = 0;
    for (int k = 0; k < 11; ++k) s->accept_rates[k] = 0;
    s->edge_res[0][0] = s->edge_res[0][1] = 0;
    s->wi_pp = (double)gmax(s, 0); s->wi_np = (double)gmax(s, 2);     /* peak:1115-1116 */
    for (int64_t j = 0; j < steps; ++j) {
        refresh(s, 0);
        ora_peak_run_updates(s, theta, transfer);
        for (int grp = 0; grp < 4; ++grp) {                       /* peak:1126-1140 */
            double *h = (grp & 1) ? s->cspec[grp >> 1] : s->pspec[grp >> 1];
            const int64_t f = g_first(s, grp);
            for (int64_t i = 0; i < g_count(s, grp); ++i) h[s->wi[f + i]] += s->A[f + i];
        }
        s->wi_pp = (double)gmax(s, 0); s->wi_np = (double)gmax(s, 2);     /* peak:1142-1143 */
        s->sampled_chi2 += s->chi2;
        s->edge_res[0][0] += s->wi_pp;
        s->edge_res[0][1] += s->wi_np;
    }
    if (!bin)
        for (int pn = 0; pn < 2; ++pn)
            for (int64_t i = 0; i < s->n_grid; ++i) {
                s->pspec[pn][i] /= (double)steps;
                s->cspec[pn][i] /= (double)steps;
            }
    s->sampled_chi2 /= (double)steps;
    for (int k = 0; k < 11; ++k) s->accept_rates[k] /= (double)steps;
    for (int pn = 0; pn < 2; ++pn) {                              /* peak:1159-1161 */
        s->edge_res[0][pn] /= (double)steps;
        s->edge_res[1][pn] += s->edge_res[0][pn];
        s->edge_res[2][pn] += s->edge_res[0][pn] * s->edge_res[0][pn];
    }
    if (s->symm == 1 || s->bosonic)
        for (int r = 0; r < 3; ++r) s->edge_res[r][1] = s->edge_res[r][0];    /* peak:1163-1165 */
}

/* ---- plumbing ---------------------------------------------------------------------------- */

ora_peak_t *ora_peak_create(int n_tau, int n_omega, int n_p, int64_t wi0, int64_t wim, int symm,
                            int bosonic, int fix_edge, const double *Kp, const double *Kn,
                            const double *G, const double *sigma_inv)
{
    if (n_tau < 1 || n_tau > 128 || n_omega < 3 || n_p < 1 || wim < 1) return NULL;
    ora_peak_t *s = (ora_peak_t *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->n_tau = n_tau; s->n_omega = n_omega; s->n_p = n_p; s->n_tot = 2 * (n_omega + n_p);
    s->symm = symm; s->bosonic = bosonic; s->fix_edge = fix_edge;
    s->wi0 = wi0; s->wim = wim; s->n_grid = wim + 1;
    s->Kp = Kp; s->Kn = Kn;
    double *g = (double *)malloc(sizeof(double) * 2 * (size_t)n_tau);
    memcpy(g, G, sizeof(double) * (size_t)n_tau);
    memcpy(g + n_tau, sigma_inv, sizeof(double) * (size_t)n_tau);
    s->G = g; s->sigma_inv = g + n_tau;
    s->wi = (int64_t *)calloc((size_t)s->n_tot, sizeof(int64_t));
    s->A = (double *)calloc((size_t)s->n_tot, sizeof(double));
    s->Gbar = (double *)calloc((size_t)n_tau, sizeof(double));
    s->Gbar_new = (double *)calloc((size_t)n_tau, sizeof(double));
    for (int pn = 0; pn < 2; ++pn) {
        s->pspec[pn] = (double *)calloc((size_t)s->n_grid, sizeof(double));
        s->cspec[pn] = (double *)calloc((size_t)s->n_grid, sizeof(double));
    }
    s->draws.mode = ORA_DRAW_PHILOX;
    return s;
}

void ora_peak_destroy(ora_peak_t *s)
{
    if (!s) return;
    free((void *)s->G); free(s->wi); free(s->A); free(s->Gbar); free(s->Gbar_new);
    for (int pn = 0; pn < 2; ++pn) { free(s->pspec[pn]); free(s->cspec[pn]); }
    free(s);
}

void ora_peak_set_state(ora_peak_t *s, const int64_t *wi, const double *A, const double *windows)
{
    if (wi) memcpy(s->wi, wi, sizeof(int64_t) * (size_t)s->n_tot);
    if (A) memcpy(s->A, A, sizeof(double) * (size_t)s->n_tot);
    if (windows) memcpy(s->windows, windows, sizeof(double) * 11);
}

void ora_peak_get_state(const ora_peak_t *s, int64_t *wi, double *A, double *windows)
{
    if (wi) memcpy(wi, s->wi, sizeof(int64_t) * (size_t)s->n_tot);
    if (A) memcpy(A, s->A, sizeof(double) * (size_t)s->n_tot);
    if (windows) memcpy(windows, s->windows, sizeof(double) * 11);
}

/* peak:1503-1506 plus the bounds initialize! leaves (peak:190-191, 202-203, 282) */
void ora_peak_init(ora_peak_t *s)
{
    p_calc_gbar(s);
    s->chi2 = chi2_impl(s->Gbar, s->G, s->sigma_inv, s->n_tau);
    s->chi2_min = s->chi2;
    s->wi_pp = (double)gmax(s, 0); s->wi_np = (double)gmax(s, 2);
    s->wi_pc = gmin(s, 1); s->wi_nc = gmin(s, 3);
}

void ora_peak_set_bounds(ora_peak_t *s, const double *b4)
{
    s->wi_pp = b4[0]; s->wi_pc = (int64_t)b4[1]; s->wi_np = b4[2]; s->wi_nc = (int64_t)b4[3];
}

void ora_peak_get_bounds(const ora_peak_t *s, double *b4)
{
    b4[0] = s->wi_pp; b4[1] = (double)s->wi_pc; b4[2] = s->wi_np; b4[3] = (double)s->wi_nc;
}

void ora_peak_get_gbar(const ora_peak_t *s, double *gbar, double *chi2, double *chi2_min)
{
    if (gbar) memcpy(gbar, s->Gbar, sizeof(double) * (size_t)s->n_tau);
    if (chi2) *chi2 = s->chi2;
    if (chi2_min) *chi2_min = s->chi2_min;
}

void ora_peak_use_philox(ora_peak_t *s, uint64_t seed, uint32_t chain, uint64_t sweep0)
{
    s->draws.mode = ORA_DRAW_PHILOX;
    s->draws.seed_lo = (uint32_t)seed; s->draws.seed_hi = (uint32_t)(seed >> 32);
    s->draws.chain = chain; s->draws.sweep = sweep0;
}

void ora_peak_set_indiv_update(ora_peak_t *s, int on) { s->indiv_update = on; }
void ora_peak_set_log(ora_peak_t *s, ora_move_rec_t *log, int64_t cap) { s->log = log; s->log_cap = cap; s->log_len = 0; }
int64_t ora_peak_log_len(const ora_peak_t *s) { return s->log_len; }
uint64_t ora_peak_sweep_counter(const ora_peak_t *s) { return s->draws.sweep; }
void ora_peak_get_accept_rates(const ora_peak_t *s, double out[11]) { memcpy(out, s->accept_rates, sizeof(double) * 11); }
double ora_peak_sampled_chi2(const ora_peak_t *s) { return s->sampled_chi2; }
void ora_peak_get_edge_res(const ora_peak_t *s, double out[6]) { memcpy(out, s->edge_res, sizeof(double) * 6); }
void ora_peak_set_edge_res(ora_peak_t *s, const double in[6]) { memcpy(s->edge_res, in, sizeof(double) * 6); }
void ora_peak_zero_spec(ora_peak_t *s)
{
    for (int pn = 0; pn < 2; ++pn) {
        memset(s->pspec[pn], 0, sizeof(double) * (size_t)s->n_grid);
        memset(s->cspec[pn], 0, sizeof(double) * (size_t)s->n_grid);
    }
}
/* which: 0 peak+, 1 cont+, 2 peak-, 3 cont- */
void ora_peak_get_spec(const ora_peak_t *s, int which, double *out)
{
    const double *h = (which & 1) ? s->cspec[which >> 1] : s->pspec[which >> 1];
    memcpy(out, h, sizeof(double) * (size_t)s->n_grid);
}
void ora_peak_get_counters(const ora_peak_t *s, uint64_t out[5]) { memcpy(out, s->counters, sizeof(uint64_t) * 5); }
