/*
 * oracle/ora_common.h -- draw source and arithmetic shared by the oracle's samplers
 * (sac_oracle.c: free, sac_oracle_peak.c: peak).  TEST INFRASTRUCTURE ONLY.
 */
#ifndef ORA_COMMON_H
#define ORA_COMMON_H
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
static inline void philox4x32_10_impl(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
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
    uint32_t w[16];     /* primary words (blocks 0 and 1; 0..3 for the four-delta transfer move) */
    uint32_t retry;     /* next retry block (starts at 2) */
} prop_t;

static inline void philox_block(const draws_t *d, uint32_t move, uint32_t iter, uint32_t b, uint32_t out[4])
{
    uint32_t ctr[4], key[2];
    ctr[0] = iter;
    ctr[1] = (uint32_t)d->sweep;
    ctr[2] = ((uint32_t)(d->sweep >> 32) & 0xFFFFu) | (move << 16) | (b << 24);
    ctr[3] = d->chain;
    key[0] = d->seed_lo; key[1] = d->seed_hi;
    philox4x32_10_impl(ctr, key, out);
}

/* four primary blocks (16 words), used by peak's A_w_transfer; no rejection loops there */
static inline void prop_begin4(prop_t *p, draws_t *d, uint32_t move, uint32_t iter)
{
    p->d = d; p->move = move; p->iter = iter; p->retry = 4;
    if (d->mode == ORA_DRAW_PHILOX)
        for (uint32_t b = 0; b < 4; ++b) philox_block(d, move, iter, b, p->w + 4 * b);
}

static inline void prop_begin(prop_t *p, draws_t *d, uint32_t move, uint32_t iter)
{
    p->d = d; p->move = move; p->iter = iter; p->retry = 2;
    if (d->mode == ORA_DRAW_PHILOX) {
        philox_block(d, move, iter, 0, p->w);
        philox_block(d, move, iter, 1, p->w + 4);
    }
}

static inline double trace_next(draws_t *d)
{
    if (d->trace_pos >= d->trace_len) { d->trace_pos++; return 0.0; }
    return d->trace[d->trace_pos++];
}

/* U = rand() in [0,1): 53 bits from two words */
static inline double words_to_u01(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

/* rand(1:n) - 1 from one word (multiply-shift) */
static inline int64_t word_to_index(uint32_t w, int64_t n)
{
    return (int64_t)(((uint64_t)w * (uint64_t)n) >> 32);
}

/* U draw occupying primary words slot, slot+1 */
static inline double draw_U(prop_t *p, int slot)
{
    if (p->d->mode == ORA_DRAW_TRACE) return trace_next(p->d);
    return words_to_u01(p->w[slot], p->w[slot + 1]);
}

/* 0-based rand(1:n) from primary word `slot` (trace: 1-based integer-valued double) */
static inline int64_t draw_R(prop_t *p, int slot, int64_t n)
{
    if (p->d->mode == ORA_DRAW_TRACE) return (int64_t)trace_next(p->d) - 1;
    return word_to_index(p->w[slot], n);
}

/* retry draw of a rejection loop: word `sel` of the next retry block.
 * Returns -1 when the retry blocks are exhausted (Philox mode only). */
static inline int64_t draw_R_retry(prop_t *p, uint32_t *retry, int sel, int64_t n)
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
static inline double log_u01_impl(double u)
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
static inline double tree_sum64(const double *t, int n)
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
static inline double chi2_impl(const double *gbar, const double *g, const double *sinv, int n_tau)
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
static inline double strided_sum(const double *a, int64_t n)
{
    double part[64];
    for (int l = 0; l < 64; ++l) {
        double v = 0.0;
        for (int64_t i = l; i < n; i += 64) v = v + a[i];
        part[l] = v;
    }
    return tree_sum64(part, 64);
}

/* Metropolis test.  free:307-309: P = exp((chi2 - chi2_new)/(2 theta)); rand() <= P.
 * Restated in the log domain: u <= exp(x)  <=>  ln u <= x  (x NaN: false, as in Julia). */
static inline int metropolis(double chi2, double chi2_new, double inv2theta, double u)
{
    double x = (chi2 - chi2_new) * inv2theta;
    return x >= log_u01_impl(u);
}

/* delta = 1 + Int64(floor(rand()*window))   free:285 */
static inline int64_t draw_shift(double u, double window)
{
    double f = floor(u * window);
    if (!(f < 4.0e18)) return INT64_MAX / 4;   /* far out of any grid: proposal will `continue` */
    return 1 + (int64_t)f;
}

/* second index of a pair: `while k2 == k1` rejection loop (free:338-341) */
static inline int64_t draw_other(prop_t *p, int slot, int sel, int64_t n, int64_t k1)
{
    int64_t k2 = draw_R(p, slot, n);
    while (k2 == k1) {
        k2 = draw_R_retry(p, &p->retry, sel, n);
        if (k2 < 0) return -1;
    }
    return k2;
}


#endif
