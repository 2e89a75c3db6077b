/*
 * oracle/sac_oracle_edge.c -- CPU oracle for the `edge` parameterisation.
 *
 * TEST INFRASTRUCTURE ONLY (see sac_oracle.h).  Plain-C restatement of the Monte-Carlo part of
 * gabes135/sac  sac/edge/sac_edge.jl ("edge:NNN" = line NNN of that file): the Taylor lookup
 * calc_Gbar!/calc_Gbar_new! (edge:591-636), check_order (edge:721-784), the five moves
 * (edge:793-1252), run_updates / measure / record_bin / run_bins / adjust_D / bin_averages
 * (edge:1260-1449) and the invariant checker check_w (edge:658-718, reported instead of exit()).
 *
 * Arithmetic shared by specification with the HIP kernels (DESIGN.md "edge"):
 *   - no FMA contraction; K_new = (K + dK*d) + d2K*d^2, Gbar_new += A_k*(K_new - K_curr)
 *   - chi^2 by the canonical 64-slot tree (ora_common.h)
 *   - every reduction the Julia text leaves to `sum`/BLAS (sum(d_array) edge:982, sum(temp_A)
 *     edge:1110, K_curr*dA edge:1113-1115, K_curr*A edge:615) runs sequentially in index order
 *   - accept test rand() < exp(x) evaluated as ln(U) < x (own ln, ora_common.h)
 *   - log/exp of the amplitude profile (edge:1092-1098) by the fixed-sequence sac_log/sac_exp below
 *   - draws: Philox fixed slots, move ids 41..51 (layout at ed_block below)
 * Parity status: no per-move golden vectors exist in the reference ("parity unpinned at move
 * granularity"); the deterministic part (Taylor lookup + chi^2 of the 50x50 initial scan) is pinned by
 * the reference's logged (j_opt, k_opt), tests/test_oracle_edge.py.
 */
#include "ora_common.h"

#include <stdio.h>

/* Julia index k = 1 .. N+1 is kept (slot 0 unused); side rl = 0 (R_L = 1), 1 (R_L = 2) */
struct ora_edge {
    ora_edge_cfg_t cfg;
    int N, NE, nt, stride;
    const double *K[2], *dK[2], *d2K[2];       /* borrowed: [n_grid][n_tau] each */
    double *G, *sinv;
    double *w[2], *A[2], *dw[2], *mw[2], *tempA[2];
    int64_t *nm[2];
    double *Kcurr[2], *Knew[2];               /* [stride][n_tau] */
    double *ar[2][2];                         /* accept_rates[m][k][rl] -> ar[rl][m][k] */
    double amp_ar[3][2];
    double n0[2], eps0[2], dn0[2], deps0[2];
    double wb0[2], wbf[2];
    double *Gbar, *Gbar_new, *tmp_d;
    double chi2, chi2_min;
    double chi2_res[3], edge_res[3][2], n0_res[3][2], wn0_res[3][2];
    double *spec[2][2];                       /* sampled_spec[:, which, rl] -> spec[rl][which] */
    double *w_avg[2], *A_avg[2];
    int pass;
    draws_t draws;
    ora_move_rec_t *log;
    int64_t log_cap, log_len;
    uint64_t counters[5];                     /* proposals, evaluated, accepted, -, Taylor lookups */
    int64_t violations;                       /* check_w failures (edge:658-718) */
};

enum { ED_SINGLE = 41, ED_MULTI = 43, ED_MULTI_RETRY = 45, ED_AMP = 47, ED_CONT = 49, ED_CONT_AMP = 50,
       ED_CONT_AMP_RETRY = 51 };

/* ---- fixed-sequence log / exp (shared spec with the HIP kernels) ------------------------------- */

/* ln(x), x > 0 normal: the same reduction and polynomial as log_u01_impl (fdlibm general branch) */
static double sac_log(double x) { return log_u01_impl(x); }

/* exp(x) for |x| < 700: fdlibm's e_exp.c scheme, IEEE operations in this fixed order */
static double sac_exp(double x)
{
    static const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                        invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                        P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    double hi = x, lo = 0.0;
    int k = 0;
    if (fabs(x) > 0.34657359027997264) {               /* 0.5 ln2 */
        k = (int)(invln2 * x + ((x < 0.0) ? -0.5 : 0.5));
        const double t = (double)k;
        hi = x - t * ln2HI;
        lo = t * ln2LO;
    }
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y;
    if (k == 0) return 1.0 - ((r * c) / (c - 2.0) - r);
    y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    uint64_t b;                                          /* exact scaling by 2^k (result stays normal here) */
    memcpy(&b, &y, 8);
    b += (uint64_t)((int64_t)k << 52);
    memcpy(&y, &b, 8);
    return y;
}

double ora_sac_log(double x) { return sac_log(x); }
double ora_sac_exp(double x) { return sac_exp(x); }

/* ---- draws ------------------------------------------------------------------------------------- */
/* block b of iteration `iter` of move `move`; word layouts:
 *   ED_SINGLE+rl   b0: w0 k, (w1,w2) U shift;             b1: (w0,w1) U accept
 *   ED_MULTI+rl    b0: w0 k, (w1,w2) U accept;            b(1+u/2): U_u of multi_1 at words 2(u%2), +1
 *                                                         b(1+p): pair p of multi_k: w0 i, w1 j, (w2,w3) U
 *   ED_MULTI_RETRY+rl  b = running retry counter of the iteration: w0 = next j
 *   ED_AMP+rl      b0: (w0,w1) U n0, (w2,w3) U eps;       b1: (w0,w1) U accept
 *   ED_CONT        b0: w0 k, (w1,w2) U shift;             b1: (w0,w1) U accept
 *   ED_CONT_AMP    b0: w0 j, w1 k, (w2,w3) r;             b1: (w0,w1) U accept;  ED_CONT_AMP_RETRY: w0 = next k */
static void ed_block(ora_edge_t *s, uint32_t move, uint32_t iter, uint32_t b, uint32_t out[4])
{
    philox_block(&s->draws, move, iter, b, out);
}

static void elog(ora_edge_t *s, int move, int evaluated, int accepted, double chi2_new)
{
    s->counters[0] += 1;
    s->counters[1] += (uint64_t)evaluated;
    s->counters[2] += (uint64_t)accepted;
    if (s->log && s->log_len < s->log_cap) {
        ora_move_rec_t *r = &s->log[s->log_len];
        r->chi2_new = evaluated ? chi2_new : 0.0;
        r->flags = (evaluated ? 1 : 0) | (accepted ? 2 : 0);
        r->move = move;
    }
    if (s->log) s->log_len++;
}

/* ln U < (chi2 - chi2_new)/(2 theta)   <=>   rand() < P   (edge:863-864) */
static int accept_lt(double chi2, double chi2_new, double inv2theta, double u)
{
    const double x = (chi2 - chi2_new) * inv2theta;
    return log_u01_impl(u) < x;
}

/* ---- Taylor lookup ----------------------------------------------------------------------------- */

/* K(w) = K[i] + dK[i] d + d2K[i] d^2, i = floor(w/dw), d = w - i dw   (edge:595-606, 626-633) */
static void taylor_col(const ora_edge_t *s, int rl, double wk, double *out)
{
    const int64_t wi = (int64_t)floor(wk / s->cfg.dw);
    const double d = wk - ((double)wi * s->cfg.dw);
    const double d2 = d * d;
    int64_t col = wi - s->cfg.wi0;
    if (col < 0) col = 0;                                /* the reference would throw a BoundsError */
    if (col >= s->cfg.n_grid) col = s->cfg.n_grid - 1;
    const double *k0 = s->K[rl] + col * s->nt, *k1 = s->dK[rl] + col * s->nt, *k2 = s->d2K[rl] + col * s->nt;
    for (int t = 0; t < s->nt; ++t) out[t] = (k0[t] + (k1[t] * d)) + (k2[t] * d2);
}

/* edge:625-636 */
static void calc_gbar_new(ora_edge_t *s, int k, double wk, int rl)
{
    double *kn = s->Knew[rl] + (size_t)k * s->nt;
    const double *kc = s->Kcurr[rl] + (size_t)k * s->nt;
    taylor_col(s, rl, wk, kn);
    const double Ak = s->A[rl][k];
    for (int t = 0; t < s->nt; ++t) s->Gbar_new[t] = s->Gbar_new[t] + Ak * (kn[t] - kc[t]);
    s->counters[4] += 1;
}

/* edge:591-622: update = 1 recomputes K_curr from the tables; Gbar = K_curr * A, columns in order */
static void calc_gbar(ora_edge_t *s, int update)
{
    if (update)
        for (int rl = 0; rl < s->cfg.n_rl; ++rl)
            for (int k = 1; k <= s->N + 1; ++k) taylor_col(s, rl, s->w[rl][k], s->Kcurr[rl] + (size_t)k * s->nt);
    for (int t = 0; t < s->nt; ++t) {
        double acc = 0.0;
        for (int k = 1; k <= s->N + 1; ++k) acc = acc + s->Kcurr[0][(size_t)k * s->nt + t] * s->A[0][k];
        s->Gbar[t] = acc;
    }
    if (s->cfg.n_rl == 2)                                /* edge:618-620 */
        for (int t = 0; t < s->nt; ++t) {
            double acc = 0.0;
            for (int k = 1; k <= s->N + 1; ++k) acc = acc + s->Kcurr[1][(size_t)k * s->nt + t] * s->A[1][k];
            s->Gbar[t] = s->Gbar[t] + acc;
        }
}

/* edge:721-784 */
static int check_order(const ora_edge_t *s, int k, double wk, int rl)
{
    const double *w = s->w[rl];
    const int NE = s->NE;
    double d1, d2, d3, d4;
    if (k == 2) {
        d1 = w[3] - wk;
        if (d1 < s->cfg.d0) return 0;
        if (NE > 2) { d2 = w[4] - w[3]; if (d2 < d1) return 0; }
    } else if (k == 3) {
        d1 = wk - w[2];
        if (d1 < s->cfg.d0) return 0;
        d2 = -1;
        if (NE > 2) { d2 = w[4] - wk; if (d2 < d1) return 0; }
        if (NE > 3) { d3 = w[5] - w[4]; if (d3 < d2) return 0; }
    } else if (k == NE) {
        d1 = w[k - 1] - w[k - 2];
        d2 = wk - w[k - 1];
        if (d2 < d1) return 0;
        d3 = w[k + 1] - wk;
        if (d3 < d2) return 0;
    } else if (k == NE + 1) {
        d1 = w[k - 1] - w[k - 2];
        d2 = wk - w[k - 1];
        if (d2 < d1) return 0;
    } else {
        d1 = w[k - 1] - w[k - 2];
        d2 = wk - w[k - 1];
        if (d2 < d1) return 0;
        d3 = w[k + 1] - wk;
        if (d3 < d2) return 0;
        d4 = w[k + 2] - w[k + 1];
        if (d4 < d3) return 0;
    }
    return 1;
}

static double arr_max(const double *a, int lo, int hi)
{
    double m = a[lo];
    for (int i = lo + 1; i <= hi; ++i) if (a[i] > m) m = a[i];
    return m;
}

static double arr_min(const double *a, int lo, int hi)
{
    double m = a[lo];
    for (int i = lo + 1; i <= hi; ++i) if (a[i] < m) m = a[i];
    return m;
}

/* edge:800-813 (first = 2) and edge:907-918 (first = 1: the dummy slot takes part, as in the reference) */
static void edge_bounds(ora_edge_t *s, int rl, int first)
{
    if (s->cfg.double_in) {
        s->wb0[rl] = -arr_max(s->w[1 - rl], first, s->N + 1);
        s->wbf[rl] = -s->w[1 - rl][2];
    } else {
        s->wb0[rl] = s->cfg.w_m;
        s->wbf[rl] = s->cfg.w_m;
    }
    if (s->cfg.A_c > 0 && rl == 0 && s->N > s->NE) {
        const double m = arr_min(s->w[0], s->NE + 2, s->N + 1);
        s->wb0[rl] = (m < s->wb0[rl]) ? m : s->wb0[rl];
    }
}

static void commit_chi2(ora_edge_t *s, double chi2_new)
{
    s->chi2 = chi2_new;
    if (chi2_new < s->chi2_min) s->chi2_min = chi2_new;
}

/* ---- moves ------------------------------------------------------------------------------------- */

/* edge:793-885 */
static void single_edge_move(ora_edge_t *s, double inv2theta, int rl)
{
    if (s->cfg.A_c == 1) return;
    edge_bounds(s, rl, 2);
    const int NE = s->NE, nt = s->nt, move = ED_SINGLE + rl;
    const int k_0 = (s->cfg.fix_edge == 1) ? 3 : 2;
    double *w = s->w[rl];
    for (int i = 1; i <= NE; ++i) {
        uint32_t b0[4], b1[4];
        ed_block(s, (uint32_t)move, (uint32_t)i, 0, b0);
        int k;
        if (i % 4 == 0 && s->cfg.fix_edge == 0) k = 2;
        else k = k_0 + (int)word_to_index(b0[0], NE + 2 - k_0);
        const double wk = w[k] + (s->dw[rl][k] * (words_to_u01(b0[1], b0[2]) - 0.5));
        if (k == 2 && wk > s->wb0[rl]) { elog(s, move, 0, 0, 0.0); continue; }
        if (k < NE + 1) {
            if (wk < w[k - 1] || wk > w[k + 1]) { elog(s, move, 0, 0, 0.0); continue; }
        } else {
            if (wk < w[k - 1] || wk > s->cfg.w_m || wk > s->wbf[rl]) { elog(s, move, 0, 0, 0.0); continue; }
        }
        s->pass = check_order(s, k, wk, rl);
        if (s->pass != 1) { elog(s, move, 0, 0, 0.0); continue; }
        memcpy(s->Gbar_new, s->Gbar, sizeof(double) * (size_t)nt);
        calc_gbar_new(s, k, wk, rl);
        const double chi2_new = chi2_impl(s->Gbar_new, s->G, s->sinv, nt);
        ed_block(s, (uint32_t)move, (uint32_t)i, 1, b1);
        const int acc = accept_lt(s->chi2, chi2_new, inv2theta, words_to_u01(b1[0], b1[1]));
        if (acc) {
            memcpy(s->Gbar, s->Gbar_new, sizeof(double) * (size_t)nt);
            memcpy(s->Kcurr[rl] + (size_t)k * nt, s->Knew[rl] + (size_t)k * nt, sizeof(double) * (size_t)nt);
            w[k] = wk;
            commit_chi2(s, chi2_new);
            s->ar[rl][0][k] += 1;
        }
        elog(s, move, 1, acc, chi2_new);
    }
}

static void sort_asc(double *a, int n)
{
    for (int i = 1; i < n; ++i) {
        const double v = a[i];
        int j = i - 1;
        while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; --j; }
        a[j + 1] = v;
    }
}

/* edge:973-999 */
static void multi_1(ora_edge_t *s, int n, int rl, uint32_t move, uint32_t iter)
{
    s->pass = 0;
    const double *w = s->w[rl];
    double *mw = s->mw[rl], *d = s->tmp_d;
    const double d1 = w[n + 2] - w[n + 1];
    uint32_t blk[4];
    for (int i = 2; i <= n; ++i) {
        const int u = i - 2;
        if ((u & 1) == 0) ed_block(s, move, iter, (uint32_t)(1 + u / 2), blk);
        d[i - 2] = s->cfg.d0 + ((d1 - s->cfg.d0) * words_to_u01(blk[2 * (u & 1)], blk[2 * (u & 1) + 1]));
    }
    double sum = 0.0;
    for (int i = 0; i < n - 1; ++i) sum = sum + d[i];
    const double w1 = w[n + 1] - sum;
    if (w1 > s->cfg.w_floor[rl] && w1 < s->wb0[rl]) {
        if (n > 2) sort_asc(d, n - 1);
        mw[2] = w1;
        for (int i = 2; i <= n; ++i) mw[i + 1] = mw[i] + d[i - 2];
        s->pass = 1;
    }
}

/* edge:1005-1052 */
static void multi_k(ora_edge_t *s, int k, int n, int rl, uint32_t move, uint32_t iter)
{
    s->pass = 0;
    const double *w = s->w[rl];
    double *mw = s->mw[rl], *d = s->tmp_d;
    for (int i = 0; i < n; ++i) d[i] = w[k + i + 1] - w[k + i];
    const double dk_minus1 = w[k] - w[k - 1];
    const double dk_plusn = w[k + n + 1] - w[k + n];
    uint32_t retry = 0;
    for (int p = 0; p < n / 2; ++p) {
        uint32_t blk[4];
        ed_block(s, move, iter, (uint32_t)(1 + p), blk);
        const int i = (int)word_to_index(blk[0], n);
        int j = (int)word_to_index(blk[1], n);
        while (j == i) {
            if (retry > 255) return;                    /* 2^-256-ish: give the proposal up */
            uint32_t rb[4];
            ed_block(s, move + (ED_MULTI_RETRY - ED_MULTI), iter, retry++, rb);
            j = (int)word_to_index(rb[0], n);
        }
        const double di = d[i], dj = d[j];
        const double a = di + dj - dk_plusn, b = di + dj - dk_minus1;
        const double d_min = (dk_minus1 > a) ? dk_minus1 : a;      /* max */
        const double d_max = (dk_plusn < b) ? dk_plusn : b;        /* min */
        d[i] = d_min + (d_max - d_min) * words_to_u01(blk[2], blk[3]);
        d[j] = di + dj - d[i];
    }
    sort_asc(d, n);
    mw[k] = w[k];
    for (int i = k + 1; i <= k + n - 1; ++i) mw[i] = mw[i - 1] + d[i - k - 1];
    if (mw[k + n - 1] < s->wbf[rl]) s->pass = 1;
}

/* edge:889-968 */
static void multi_edge_move(ora_edge_t *s, double inv2theta, int rl)
{
    if (s->cfg.A_c == 1) return;
    const int NE = s->NE, nt = s->nt;
    if (NE < 4) return;
    edge_bounds(s, rl, 1);
    const uint32_t move = (uint32_t)(ED_MULTI + rl);
    for (int i = 0; i <= NE - 1; ++i) {
        uint32_t b0[4];
        ed_block(s, move, (uint32_t)i, 0, b0);
        int k, n;
        if (i % 4 == 0 && s->cfg.fix_edge == 0) {
            k = 2;
            n = (int)s->nm[rl][2];
            multi_1(s, n, rl, move, (uint32_t)i);
        } else {
            k = 3 + (int)word_to_index(b0[0], NE - 3);
            n = (int)s->nm[rl][k];
            multi_k(s, k, n, rl, move, (uint32_t)i);
        }
        if (s->pass != 1) { elog(s, (int)move, 0, 0, 0.0); continue; }
        memcpy(s->Gbar_new, s->Gbar, sizeof(double) * (size_t)nt);
        for (int j = 1; j <= n; ++j) calc_gbar_new(s, k + j - 1, s->mw[rl][k + j - 1], rl);
        const double chi2_new = chi2_impl(s->Gbar_new, s->G, s->sinv, nt);
        const int acc = accept_lt(s->chi2, chi2_new, inv2theta, words_to_u01(b0[1], b0[2]));
        if (acc) {
            memcpy(s->Gbar, s->Gbar_new, sizeof(double) * (size_t)nt);
            for (int j = k; j <= k + n - 1; ++j) {
                memcpy(s->Kcurr[rl] + (size_t)j * nt, s->Knew[rl] + (size_t)j * nt, sizeof(double) * (size_t)nt);
                s->w[rl][j] = s->mw[rl][j];
            }
            commit_chi2(s, chi2_new);
            s->ar[rl][1][k] += 1;
        }
        elog(s, (int)move, 1, acc, chi2_new);
    }
}

/* amplitude profile of the edge (edge:357-366, 1091-1110) into out[2 .. NE+1], normalised to f1 */
static void edge_profile(const ora_edge_t *s, int rl, double n1, double eps_prod, double *out)
{
    const double c = s->cfg.c[rl];
    for (int i = 1; i <= s->NE; ++i) {
        const double x = sac_log((double)i / n1);
        const double cx = c * x;
        const double root = sqrt((cx * cx) + eps_prod);
        const double ln_Ai = (c < 0) ? (cx + root) : (cx - root);
        out[i + 1] = sac_exp(ln_Ai / 2);
    }
    double sum = 0.0;
    for (int i = 2; i <= s->NE + 1; ++i) sum = sum + out[i];
    const double f1 = (rl == 0) ? (s->cfg.A_r / (1 + s->cfg.A_c)) : ((1 - s->cfg.A_r) / (1 + s->cfg.A_c));
    const double scale = f1 / sum;
    for (int i = 2; i <= s->NE + 1; ++i) out[i] = out[i] * scale;
}

/* edge:1057-1142 */
static void amplitude_edge_move(ora_edge_t *s, double inv2theta, int rl)
{
    if (s->cfg.A_c == 1) return;
    if (s->NE < 3) return;
    const int nt = s->nt, move = ED_AMP + rl;
    uint32_t b0[4], b1[4];
    ed_block(s, (uint32_t)move, 0, 0, b0);
    const double c = s->cfg.c[rl];
    const double eps = (M_LN2 * c) * (M_LN2 * c);
    double n1 = s->n0[rl], e1 = s->eps0[rl];
    n1 += (s->dn0[rl] * (words_to_u01(b0[0], b0[1]) - 0.5));
    if (n1 < 5 || n1 > (s->NE - 5)) { elog(s, move, 0, 0, 0.0); return; }
    e1 += s->deps0[rl] * (words_to_u01(b0[2], b0[3]) - 0.5);
    if (e1 < 0 || e1 > 1) { elog(s, move, 0, 0, 0.0); return; }
    edge_profile(s, rl, n1, e1 * eps, s->tempA[rl]);
    for (int t = 0; t < nt; ++t) {                       /* edge:1113-1115 */
        double acc = 0.0;
        for (int k = 2; k <= s->NE + 1; ++k)
            acc = acc + s->Kcurr[rl][(size_t)k * nt + t] * (s->tempA[rl][k] - s->A[rl][k]);
        s->Gbar_new[t] = s->Gbar[t] + acc;
    }
    const double chi2_new = chi2_impl(s->Gbar_new, s->G, s->sinv, nt);
    ed_block(s, (uint32_t)move, 0, 1, b1);
    const int acc = accept_lt(s->chi2, chi2_new, inv2theta, words_to_u01(b1[0], b1[1]));
    if (acc) {
        s->n0[rl] = n1;
        s->eps0[rl] = e1;
        for (int k = 2; k <= s->NE + 1; ++k) s->A[rl][k] = s->tempA[rl][k];
        memcpy(s->Gbar, s->Gbar_new, sizeof(double) * (size_t)nt);
        commit_chi2(s, chi2_new);
        s->amp_ar[0][rl] += 1;
    }
    elog(s, move, 1, acc, chi2_new);
}

/* edge:1146-1189 */
static void single_cont_move(ora_edge_t *s, double inv2theta)
{
    if (s->NE == s->N || s->cfg.A_c == 0) return;
    const int NE = s->NE, N = s->N, nt = s->nt;
    double *w = s->w[0];
    const double wc_min = w[2], wc_max = w[NE + 1];
    int it = 0;
    for (int i = NE + 2; i <= N + 1; ++i, ++it) {
        uint32_t b0[4], b1[4];
        ed_block(s, ED_CONT, (uint32_t)it, 0, b0);
        const int k = NE + 2 + (int)word_to_index(b0[0], N - NE);
        const double wk = w[k] + (s->dw[0][k] * (words_to_u01(b0[1], b0[2]) - 0.5));
        if (wk < wc_min || wk > wc_max) { elog(s, ED_CONT, 0, 0, 0.0); continue; }
        memcpy(s->Gbar_new, s->Gbar, sizeof(double) * (size_t)nt);
        calc_gbar_new(s, k, wk, 0);
        const double chi2_new = chi2_impl(s->Gbar_new, s->G, s->sinv, nt);
        ed_block(s, ED_CONT, (uint32_t)it, 1, b1);
        const int acc = accept_lt(s->chi2, chi2_new, inv2theta, words_to_u01(b1[0], b1[1]));
        if (acc) {
            memcpy(s->Gbar, s->Gbar_new, sizeof(double) * (size_t)nt);
            memcpy(s->Kcurr[0] + (size_t)k * nt, s->Knew[0] + (size_t)k * nt, sizeof(double) * (size_t)nt);
            w[k] = wk;
            commit_chi2(s, chi2_new);
            s->ar[0][0][k] += 1;
        }
        elog(s, ED_CONT, 1, acc, chi2_new);
    }
}

/* edge:1194-1252.  j, k = rand((N_e+2):N_w): the last continuum delta is never drawn (as in the
 * reference); with fewer than two candidates the reference's `while k == j` would not end: skipped. */
static void amplitude_cont_move(ora_edge_t *s, double inv2theta)
{
    if (s->NE == s->N || s->cfg.A_c == 0) return;
    const int NE = s->NE, N = s->N, nt = s->nt;
    const int ncand = N - NE - 1;
    if (ncand < 2) return;
    int64_t n_accept = 0;
    for (int i = 1; i <= N; ++i) {
        uint32_t b0[4], b1[4];
        ed_block(s, ED_CONT_AMP, (uint32_t)i, 0, b0);
        const int j = NE + 2 + (int)word_to_index(b0[0], ncand);
        int k = NE + 2 + (int)word_to_index(b0[1], ncand);
        uint32_t retry = 0;
        int give_up = 0;
        while (k == j) {
            if (retry > 255) { give_up = 1; break; }
            uint32_t rb[4];
            ed_block(s, ED_CONT_AMP_RETRY, (uint32_t)i, retry++, rb);
            k = NE + 2 + (int)word_to_index(rb[0], ncand);
        }
        if (give_up) { elog(s, ED_CONT_AMP, 0, 0, 0.0); continue; }
        const double A_j = s->A[0][j], A_k = s->A[0][k];
        const double m0 = A_j + A_k;
        const double r = words_to_u01(b0[2], b0[3]);
        const double dA_j = (r * m0) - A_j;
        const double dA_k = ((1 - r) * m0) - A_k;
        const double *kj = s->Kcurr[0] + (size_t)j * nt, *kk = s->Kcurr[0] + (size_t)k * nt;
        for (int t = 0; t < nt; ++t) s->Gbar_new[t] = (s->Gbar[t] + dA_j * kj[t]) + dA_k * kk[t];
        const double chi2_new = chi2_impl(s->Gbar_new, s->G, s->sinv, nt);
        ed_block(s, ED_CONT_AMP, (uint32_t)i, 1, b1);
        const int acc = accept_lt(s->chi2, chi2_new, inv2theta, words_to_u01(b1[0], b1[1]));
        if (acc) {
            s->A[0][j] += dA_j;
            s->A[0][k] += dA_k;
            memcpy(s->Gbar, s->Gbar_new, sizeof(double) * (size_t)nt);
            commit_chi2(s, chi2_new);
            n_accept += 1;
        }
        elog(s, ED_CONT_AMP, 1, acc, chi2_new);
    }
    s->amp_ar[1][0] += (double)n_accept / (double)(N - NE);
}

/* ---- stage ------------------------------------------------------------------------------------- */

/* edge:1305-1333 */
static void measure(ora_edge_t *s)
{
    s->chi2_res[0] += s->chi2;
    for (int rl = 0; rl < s->cfg.n_rl; ++rl) {
        s->n0_res[0][rl] += s->n0[rl] / s->NE;
        int idx = (int)floor(s->n0[rl]);
        if (idx < 1) idx = 1;
        if (idx > s->N + 1) idx = s->N + 1;
        s->wn0_res[0][rl] += s->w[rl][idx];
        for (int k = 1; k <= s->N + 1; ++k) {
            s->A_avg[rl][k] += s->A[rl][k];
            s->w_avg[rl][k] += s->w[rl][k];
        }
        s->edge_res[0][rl] += s->w[rl][2];
        for (int i = 1; i <= s->N; ++i) {
            const int64_t wi = (int64_t)floor(s->w[rl][i + 1] / s->cfg.dw) - s->cfg.wi0;
            if (wi < 0 || wi >= s->cfg.n_grid) continue;            /* reference: BoundsError */
            s->spec[rl][(i <= s->NE) ? 0 : 1][wi] += s->A[rl][i + 1];
        }
    }
}

/* edge:1337-1353 */
static void record_bin(ora_edge_t *s, int64_t steps)
{
    s->chi2_res[0] /= (double)steps;
    for (int rl = 0; rl < 2; ++rl) {
        s->n0_res[0][rl] /= (double)steps;
        s->wn0_res[0][rl] /= (double)steps;
        s->edge_res[0][rl] /= (double)steps;
    }
    s->chi2_res[1] += s->chi2_res[0];
    s->chi2_res[2] += s->chi2_res[0] * s->chi2_res[0];
    for (int rl = 0; rl < 2; ++rl) {
        s->n0_res[1][rl] += s->n0_res[0][rl];
        s->wn0_res[1][rl] += s->wn0_res[0][rl];
        s->edge_res[1][rl] += s->edge_res[0][rl];
        s->n0_res[2][rl] += s->n0_res[0][rl] * s->n0_res[0][rl];
        s->wn0_res[2][rl] += s->wn0_res[0][rl] * s->wn0_res[0][rl];
        s->edge_res[2][rl] += s->edge_res[0][rl] * s->edge_res[0][rl];
    }
}

/* edge:658-718, reported through the violation counter instead of exit() */
static void check_w(ora_edge_t *s, int rl)
{
    const double *w = s->w[rl];
    int bad = 0;
    if (w[2] < s->cfg.w_floor[rl]) bad = 1;
    if (arr_min(w, 1, s->N + 1) < s->cfg.w_floor[rl] || arr_max(w, 1, s->N + 1) > s->cfg.w_m) bad = 1;
    if (s->NE < s->N && rl == 0 && w[2] > arr_min(w, s->NE + 2, s->N + 1)) bad = 1;
    if (w[3] - w[2] < s->cfg.d0) bad = 1;
    for (int i = 2; i <= s->NE - 1; ++i) {
        const double v1 = w[i + 1] - w[i], v2 = w[i + 2] - w[i + 1];
        if (v2 < v1 && v1 - v2 > 1e-10) bad = 1;
    }
    s->violations += bad;
}

/* edge:1260-1302 */
void ora_edge_run_updates(ora_edge_t *s, int64_t steps, double theta)
{
    const double inv2theta = 1.0 / (2.0 * theta);
    s->chi2_res[0] = 0;
    for (int rl = 0; rl < 2; ++rl) { s->edge_res[0][rl] = 0; s->n0_res[0][rl] = 0; s->wn0_res[0][rl] = 0; }
    for (int rl = 0; rl < 2; ++rl)
        for (int m = 0; m < 2; ++m) memset(s->ar[rl][m], 0, sizeof(double) * (size_t)s->stride);
    calc_gbar(s, 0);
    for (int64_t i = 0; i < steps; ++i) {
        for (int rl = 0; rl < s->cfg.n_rl; ++rl) {
            single_edge_move(s, inv2theta, rl);
            multi_edge_move(s, inv2theta, rl);
            amplitude_edge_move(s, inv2theta, rl);
        }
        if (s->cfg.A_c > 0) {
            single_cont_move(s, inv2theta);
            amplitude_cont_move(s, inv2theta);
        }
        measure(s);
        s->draws.sweep += 1;
    }
    record_bin(s, steps);
    for (int rl = 0; rl < s->cfg.n_rl; ++rl) check_w(s, rl);
}

/* edge:1384-1432 */
static void adjust_delta(ora_edge_t *s, int64_t steps, int rl)
{
    const int N = s->N, NE = s->NE;
    for (int k = 1; k <= N + 1; ++k) { s->ar[rl][0][k] /= (double)steps; s->ar[rl][1][k] /= (double)steps; }
    for (int i = 1; i <= N; ++i) {
        if (s->ar[rl][0][i + 1] > 0.55) s->dw[rl][i + 1] *= 1.25;
        else if (s->ar[rl][0][i + 1] < 0.45) s->dw[rl][i + 1] /= 1.2;
    }
    for (int i = 0; i <= NE - 2; ++i) {
        if (s->ar[rl][1][i + 1] > 0.55) s->nm[rl][i + 1] = (int64_t)floor((double)s->nm[rl][i + 1] * 1.25) + 1;
        else if (s->ar[rl][1][i + 1] < 0.45) {
            const int64_t v = (int64_t)floor((double)s->nm[rl][i + 1] / 1.2);
            s->nm[rl][i + 1] = (v > 1) ? v : 1;
        }
    }
    if (NE > 3)
        for (int i = 2; i <= NE - 1; ++i)
            if (s->nm[rl][i] > NE - i) s->nm[rl][i] = NE - i;
    s->amp_ar[0][rl] /= (double)steps;
    s->amp_ar[1][rl] /= (double)steps;
    if (s->amp_ar[0][rl] > 0.55) { s->dn0[rl] *= 1.25; s->deps0[rl] *= 1.25; }
    else if (s->amp_ar[0][rl] < 0.45) { s->dn0[rl] /= 1.20; s->deps0[rl] /= 1.20; }
}

/* edge:1436-1449 */
static void bin_averages(ora_edge_t *s, int64_t bins)
{
    const double b = (double)bins;
    for (int r = 0; r < 3; ++r) {
        s->chi2_res[r] /= b;
        for (int rl = 0; rl < 2; ++rl) { s->n0_res[r][rl] /= b; s->wn0_res[r][rl] /= b; s->edge_res[r][rl] /= b; }
    }
    const double den = sqrt((double)(bins - 1));
    s->chi2_res[2] = sqrt(fabs(s->chi2_res[2] - s->chi2_res[1] * s->chi2_res[1])) / den;
    for (int rl = 0; rl < s->cfg.n_rl; ++rl) {
        s->n0_res[2][rl] = sqrt(fabs(s->n0_res[2][rl] - s->n0_res[1][rl] * s->n0_res[1][rl])) / den;
        s->wn0_res[2][rl] = sqrt(fabs(s->wn0_res[2][rl] - s->wn0_res[1][rl] * s->wn0_res[1][rl])) / den;
        s->edge_res[2][rl] = sqrt(fabs(s->edge_res[2][rl] - s->edge_res[1][rl] * s->edge_res[1][rl])) / den;
    }
}

/* edge:1357-1379 */
void ora_edge_run_bins(ora_edge_t *s, int64_t steps, int64_t bins, double theta)
{
    memset(s->chi2_res, 0, sizeof(s->chi2_res));
    memset(s->edge_res, 0, sizeof(s->edge_res));
    memset(s->n0_res, 0, sizeof(s->n0_res));
    memset(s->wn0_res, 0, sizeof(s->wn0_res));
    for (int rl = 0; rl < 2; ++rl) {
        for (int wch = 0; wch < 2; ++wch) memset(s->spec[rl][wch], 0, sizeof(double) * (size_t)s->cfg.n_grid);
        memset(s->w_avg[rl], 0, sizeof(double) * (size_t)s->stride);
        memset(s->A_avg[rl], 0, sizeof(double) * (size_t)s->stride);
    }
    for (int64_t i = 0; i < bins; ++i) {
        ora_edge_run_updates(s, steps, theta);
        for (int rl = 0; rl < s->cfg.n_rl; ++rl) adjust_delta(s, steps, rl);
    }
    bin_averages(s, bins);
}

/* ---- plumbing ---------------------------------------------------------------------------------- */

ora_edge_t *ora_edge_create(const ora_edge_cfg_t *cfg, const double *const *K, const double *const *dK,
                            const double *const *d2K, const double *G, const double *sigma_inv)
{
    if (!cfg || cfg->n_tau < 1 || cfg->n_tau > 128 || cfg->n_e < 1 || cfg->n_omega < cfg->n_e ||
        cfg->n_rl < 1 || cfg->n_rl > 2 || cfg->n_grid < 2)
        return NULL;
    ora_edge_t *s = (ora_edge_t *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->cfg = *cfg;
    s->N = cfg->n_omega; s->NE = cfg->n_e; s->nt = cfg->n_tau;
    s->stride = s->N + 3;
    const size_t st = (size_t)s->stride, nt = (size_t)s->nt;
    s->G = (double *)calloc(nt, 8); s->sinv = (double *)calloc(nt, 8);
    memcpy(s->G, G, nt * 8); memcpy(s->sinv, sigma_inv, nt * 8);
    s->Gbar = (double *)calloc(nt, 8); s->Gbar_new = (double *)calloc(nt, 8);
    s->tmp_d = (double *)calloc(st, 8);
    for (int rl = 0; rl < 2; ++rl) {
        s->K[rl] = (rl < cfg->n_rl) ? K[rl] : NULL;
        s->dK[rl] = (rl < cfg->n_rl) ? dK[rl] : NULL;
        s->d2K[rl] = (rl < cfg->n_rl) ? d2K[rl] : NULL;
        s->w[rl] = (double *)calloc(st, 8); s->A[rl] = (double *)calloc(st, 8);
        s->dw[rl] = (double *)calloc(st, 8); s->mw[rl] = (double *)calloc(st, 8);
        s->tempA[rl] = (double *)calloc(st, 8); s->nm[rl] = (int64_t *)calloc(st, 8);
        s->Kcurr[rl] = (double *)calloc(st * nt, 8); s->Knew[rl] = (double *)calloc(st * nt, 8);
        s->ar[rl][0] = (double *)calloc(st, 8); s->ar[rl][1] = (double *)calloc(st, 8);
        s->w_avg[rl] = (double *)calloc(st, 8); s->A_avg[rl] = (double *)calloc(st, 8);
        for (int wch = 0; wch < 2; ++wch) s->spec[rl][wch] = (double *)calloc((size_t)cfg->n_grid, 8);
        s->dn0[rl] = 2.0; s->deps0[rl] = 0.1;
    }
    s->chi2_min = 1.0e20;
    return s;
}

void ora_edge_destroy(ora_edge_t *s)
{
    if (!s) return;
    free(s->G); free(s->sinv); free(s->Gbar); free(s->Gbar_new); free(s->tmp_d);
    for (int rl = 0; rl < 2; ++rl) {
        free(s->w[rl]); free(s->A[rl]); free(s->dw[rl]); free(s->mw[rl]); free(s->tempA[rl]); free(s->nm[rl]);
        free(s->Kcurr[rl]); free(s->Knew[rl]); free(s->ar[rl][0]); free(s->ar[rl][1]);
        free(s->w_avg[rl]); free(s->A_avg[rl]);
        for (int wch = 0; wch < 2; ++wch) free(s->spec[rl][wch]);
    }
    free(s);
}

/* arrays in the reference's shape [N_w+1, 2] (column-major: side after side); scal = n0[2], eps0[2],
 * dn0[2], deps0[2]; any pointer may be NULL */
void ora_edge_set_state(ora_edge_t *s, const double *w, const double *A, const double *dw, const int64_t *nm,
                        const double *scal)
{
    const int n1 = s->N + 1;
    for (int rl = 0; rl < 2; ++rl)
        for (int k = 1; k <= n1; ++k) {
            if (w) s->w[rl][k] = w[rl * n1 + k - 1];
            if (A) { s->A[rl][k] = A[rl * n1 + k - 1]; s->tempA[rl][k] = s->A[rl][k]; }
            if (dw) s->dw[rl][k] = dw[rl * n1 + k - 1];
            if (nm) s->nm[rl][k] = nm[rl * n1 + k - 1];
        }
    if (scal)
        for (int rl = 0; rl < 2; ++rl) {
            s->n0[rl] = scal[rl]; s->eps0[rl] = scal[2 + rl]; s->dn0[rl] = scal[4 + rl]; s->deps0[rl] = scal[6 + rl];
        }
}

void ora_edge_get_state(const ora_edge_t *s, double *w, double *A, double *dw, int64_t *nm, double *scal)
{
    const int n1 = s->N + 1;
    for (int rl = 0; rl < 2; ++rl)
        for (int k = 1; k <= n1; ++k) {
            if (w) w[rl * n1 + k - 1] = s->w[rl][k];
            if (A) A[rl * n1 + k - 1] = s->A[rl][k];
            if (dw) dw[rl * n1 + k - 1] = s->dw[rl][k];
            if (nm) nm[rl * n1 + k - 1] = s->nm[rl][k];
        }
    if (scal)
        for (int rl = 0; rl < 2; ++rl) {
            scal[rl] = s->n0[rl]; scal[2 + rl] = s->eps0[rl]; scal[4 + rl] = s->dn0[rl]; scal[6 + rl] = s->deps0[rl];
        }
}

/* calc_Gbar!(self, 1); chi2 = calc_chi2; chi2_min = chi2   (edge:559-561, 1728-1730) */
void ora_edge_init_gbar(ora_edge_t *s)
{
    calc_gbar(s, 1);
    s->chi2 = chi2_impl(s->Gbar, s->G, s->sinv, s->nt);
    s->chi2_min = s->chi2;
}

/* set_amplitudes!-style profile for given n0, eps0 (edge:357-366, 386): fills out[N_e] */
void ora_edge_profile(ora_edge_t *s, int rl, double n0, double eps0, double *out)
{
    const double c = s->cfg.c[rl];
    double *tmp = (double *)calloc((size_t)s->stride, 8);
    edge_profile(s, rl, n0, ((M_LN2 * c) * (M_LN2 * c)) * eps0, tmp);
    memcpy(out, tmp + 2, sizeof(double) * (size_t)s->NE);
    free(tmp);
}

void ora_edge_get_gbar(const ora_edge_t *s, double *gbar, double *chi2, double *chi2_min)
{
    if (gbar) memcpy(gbar, s->Gbar, sizeof(double) * (size_t)s->nt);
    if (chi2) *chi2 = s->chi2;
    if (chi2_min) *chi2_min = s->chi2_min;
}

void ora_edge_use_philox(ora_edge_t *s, uint64_t seed, uint32_t chain, uint64_t sweep0)
{
    s->draws.mode = ORA_DRAW_PHILOX;
    s->draws.seed_lo = (uint32_t)seed; s->draws.seed_hi = (uint32_t)(seed >> 32);
    s->draws.chain = chain; s->draws.sweep = sweep0;
}

void ora_edge_set_log(ora_edge_t *s, ora_move_rec_t *log, int64_t cap) { s->log = log; s->log_cap = cap; s->log_len = 0; }
int64_t ora_edge_log_len(const ora_edge_t *s) { return s->log_len; }
uint64_t ora_edge_sweep_counter(const ora_edge_t *s) { return s->draws.sweep; }
int64_t ora_edge_violations(const ora_edge_t *s) { return s->violations; }
void ora_edge_get_counters(const ora_edge_t *s, uint64_t out[5]) { memcpy(out, s->counters, sizeof(uint64_t) * 5); }

/* res[21] = chi2_res[3], edge_res[3][2], n0_res[3][2], wn0_res[3][2]; amp_ar[6] = amplitude_accept_rates[3][2] */
void ora_edge_get_results(const ora_edge_t *s, double *res, double *amp_ar)
{
    if (res) {
        memcpy(res, s->chi2_res, 24);
        memcpy(res + 3, s->edge_res, 48);
        memcpy(res + 9, s->n0_res, 48);
        memcpy(res + 15, s->wn0_res, 48);
    }
    if (amp_ar) memcpy(amp_ar, s->amp_ar, 48);
}

/* accept_rates[m, :, rl] for m = 0 (single), 1 (multi): N_w+1 doubles (Julia k = 1..N_w+1) */
void ora_edge_get_accept_rates(const ora_edge_t *s, int m, int rl, double *out)
{
    memcpy(out, s->ar[rl][m] + 1, sizeof(double) * (size_t)(s->N + 1));
}

void ora_edge_get_spec(const ora_edge_t *s, int which, int rl, double *out)
{
    memcpy(out, s->spec[rl][which], sizeof(double) * (size_t)s->cfg.n_grid);
}

/* w_avg / A_avg [N_w+1, 2] */
void ora_edge_get_avg(const ora_edge_t *s, double *w_avg, double *A_avg)
{
    const int n1 = s->N + 1;
    for (int rl = 0; rl < 2; ++rl)
        for (int k = 1; k <= n1; ++k) {
            if (w_avg) w_avg[rl * n1 + k - 1] = s->w_avg[rl][k];
            if (A_avg) A_avg[rl * n1 + k - 1] = s->A_avg[rl][k];
        }
}
