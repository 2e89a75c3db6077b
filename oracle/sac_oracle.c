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
#include "sac_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* Random draws                                                             */
/* ------------------------------------------------------------------------ */

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

/* Philox4x32-10 (Salmon et al., SC'11).  Checked against the Random123
 * known-answer vectors in tests/test_oracle_rng.py. */
void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct {
    int mode;
    /* Philox fixed-slot stream: block b of iteration `iter` of move `move`
     * in sweep `sweep` of chain `chain`:
     *   ctr = { iter, sweep[31:0], sweep[47:32] | move<<16 | b<<24, chain }
     *   key = { seed[31:0], seed[63:32] }                                   */
    uint32_t seed_lo, seed_hi, chain;
    uint64_t sweep;
    /* explicit trace, consumed in the reference's rand() call order */
    const double *trace;
    int64_t trace_len, trace_pos;
} draws_t;

/* one proposal's view of the draw source */
typedef struct {
    draws_t *d;
    uint32_t move, iter;
    uint32_t w[8];      /* primary words (blocks 0 and 1) */
    uint32_t retry;     /* next retry block (starts at 2) */
} prop_t;

static void philox_block(const draws_t *d, uint32_t move, uint32_t iter, uint32_t b, uint32_t out[4])
{
    uint32_t ctr[4], key[2];
    ctr[0] = iter;
    ctr[1] = (uint32_t)d->sweep;
    ctr[2] = ((uint32_t)(d->sweep >> 32) & 0xFFFFu) | (move << 16) | (b << 24);
    ctr[3] = d->chain;
    key[0] = d->seed_lo; key[1] = d->seed_hi;
    ora_philox4x32_10(ctr, key, out);
}

static void prop_begin(prop_t *p, draws_t *d, uint32_t move, uint32_t iter)
{
    p->d = d; p->move = move; p->iter = iter; p->retry = 2;
    if (d->mode == ORA_DRAW_PHILOX) {
        philox_block(d, move, iter, 0, p->w);
        philox_block(d, move, iter, 1, p->w + 4);
    }
}

static double trace_next(draws_t *d)
{
    if (d->trace_pos >= d->trace_len) { d->trace_pos++; return 0.0; }
    return d->trace[d->trace_pos++];
}

/* U = rand() in [0,1): 53 bits from two words */
static double words_to_u01(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

/* rand(1:n) - 1 from one word (multiply-shift) */
static int64_t word_to_index(uint32_t w, int64_t n)
{
    return (int64_t)(((uint64_t)w * (uint64_t)n) >> 32);
}

/* U draw occupying primary words slot, slot+1 */
static double draw_U(prop_t *p, int slot)
{
    if (p->d->mode == ORA_DRAW_TRACE) return trace_next(p->d);
    return words_to_u01(p->w[slot], p->w[slot + 1]);
}

/* 0-based rand(1:n) from primary word `slot` (trace: 1-based integer-valued double) */
static int64_t draw_R(prop_t *p, int slot, int64_t n)
{
    if (p->d->mode == ORA_DRAW_TRACE) return (int64_t)trace_next(p->d) - 1;
    return word_to_index(p->w[slot], n);
}

/* retry draw of a rejection loop: word `sel` of the next retry block.
 * Returns -1 when the retry blocks are exhausted (Philox mode only). */
static int64_t draw_R_retry(prop_t *p, uint32_t *retry, int sel, int64_t n)
{
    if (p->d->mode == ORA_DRAW_TRACE) return (int64_t)trace_next(p->d) - 1;
    if (*retry > 255) return -1;
    uint32_t blk[4];
    philox_block(p->d, p->move, p->iter, *retry, blk);
    (*retry)++;
    return word_to_index(blk[sel], n);
}

/* ------------------------------------------------------------------------ */
/* Arithmetic shared by spec with the HIP kernels                           */
/* ------------------------------------------------------------------------ */

/* ln(u) for a draw u in [0,1); ln(0) = -inf.  Reduction u = 2^k (1+f) with
 * sqrt(1/2) < 1+f < sqrt(2), s = f/(2+f), even polynomial in s (the classical
 * fdlibm scheme, general branch only); basic IEEE operations in a fixed order, so
 * the HIP kernels' own implementation of the same sequence agrees bit for bit.
 * Max error < 1 ulp (tests/test_oracle_rng.py). */
double ora_log_u01(double u)
{
    if (!(u > 0.0)) return -INFINITY;
    uint64_t bits;
    memcpy(&bits, &u, 8);
    uint32_t hx = (uint32_t)(bits >> 32);
    const uint32_t lx = (uint32_t)bits;
    int k = (int)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    const uint32_t i = (hx + 0x95f64u) & 0x100000u;
    const uint64_t mb = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | lx;
    double m;
    memcpy(&m, &mb, 8);
    k += (int)(i >> 20);
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double R = t2 + t1;
    const double hfsq = (0.5 * f) * f;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

/* Canonical 64-slot pairwise tree: slot l = t[l] (+ t[l+64]); missing slots are +0.
 * Levels pair (l, l^1), (l, l^2), ... (l, l^32): the wave64 xor butterfly.   */
static double tree_sum64(const double *t, int n)
{
    double s[64];
    for (int l = 0; l < 64; ++l) {
        double v = (l < n) ? t[l] : 0.0;
        if (l + 64 < n) v = v + t[l + 64];
        s[l] = v;
    }
    for (int w = 1; w < 64; w <<= 1)
        for (int i = 0; i < 64; i += 2 * w)
            s[i] = s[i] + s[i + w];
    return s[0];
}

/* free:249-263  calc_chi2 / calc_chi2_new:  sum_j ((Gbar_j - G_j) * sigma_inv_j)^2 */
double ora_chi2(const double *gbar, const double *g, const double *sinv, int n_tau)
{
    double t[128];
    for (int j = 0; j < n_tau; ++j) {
        double d = (gbar[j] - g[j]) * sinv[j];
        t[j] = d * d;
    }
    return tree_sum64(t, n_tau);
}

/* Canonical sum of n values (used for sum(A_array), free:563-567, 721-725):
 * 64 strided sequential partials (lane l sums elements l, l+64, ...), then the tree. */
static double strided_sum(const double *a, int64_t n)
{
    double part[64];
    for (int l = 0; l < 64; ++l) {
        double v = 0.0;
        for (int64_t i = l; i < n; i += 64) v = v + a[i];
        part[l] = v;
    }
    return tree_sum64(part, 64);
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

/* Metropolis test.  free:307-309: P = exp((chi2 - chi2_new)/(2 theta)); rand() <= P.
 * Restated in the log domain: u <= exp(x)  <=>  ln u <= x  (x NaN: false, as in Julia). */
static int metropolis(double chi2, double chi2_new, double inv2theta, double u)
{
    double x = (chi2 - chi2_new) * inv2theta;
    return x >= ora_log_u01(u);
}

/* free:310-318 (and the identical commit blocks of the other moves) */
static void commit(ora_free_t *s, double chi2_new)
{
    memcpy(s->Gbar, s->Gbar_new, sizeof(double) * (size_t)s->n_tau);
    s->chi2 = chi2_new;
    if (chi2_new < s->chi2_min) s->chi2_min = chi2_new;
}

/* delta = 1 + Int64(floor(rand()*window))   free:285 */
static int64_t draw_shift(double u, double window)
{
    double f = floor(u * window);
    if (!(f < 4.0e18)) return INT64_MAX / 4;   /* far out of any grid: proposal will `continue` */
    return 1 + (int64_t)f;
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

/* second index of a pair: `while k2 == k1` rejection loop (free:338-341) */
static int64_t draw_other(prop_t *p, int slot, int sel, int64_t n, int64_t k1)
{
    int64_t k2 = draw_R(p, slot, n);
    while (k2 == k1) {
        k2 = draw_R_retry(p, &p->retry, sel, n);
        if (k2 < 0) return -1;
    }
    return k2;
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
