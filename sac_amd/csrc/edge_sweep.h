// Device side of the `edge` parameterisation (gfx950 only): one wavefront = one Markov chain of
// sac/edge/sac_edge.jl ("edge:NNN" = line NNN of that file), lane = tau index.
//
// The chain's small arrays (positions, amplitudes, windows, cluster sizes, accept counters) live in
// LDS; Gbar, G, sigma^-1 in registers; K_curr / K_new (one column per delta, edge:59-60) in global
// memory, where lane t only ever touches element t of a column.  Control flow depends only on
// wave-uniform values (LDS reads of the same address, Philox blocks computed redundantly by every
// lane), so the wave never diverges.  Arithmetic order is the one fixed in DESIGN.md "edge" and
// implemented independently by the CPU oracle.
#pragma once
#include "free_sweep.h"
#include "wave_ops.h"

namespace sacdev {

enum : int { EDGE_BIN_START = 1, EDGE_BIN_END = 2 };
// LDS scalar block: per side n0, eps0, dn0, deps0; amplitude_accept_rates[3][2]; bin accumulators
enum : int { SS_N0 = 0, SS_EPS0 = 2, SS_DN0 = 4, SS_DEPS0 = 6, SS_AMP_AR = 8, SS_CHI2_ACC = 14, SS_EDGE_ACC = 15,
             SS_N0_ACC = 17, SS_WN0_ACC = 19, SS_CNT = 24 /* 4 x u64 */, SS_CHI2_MIN = 28, SS_WB0 = 30, SS_WBF = 32,
#ifdef EDGE_PROF
             SS_PROF = 36, SS_COUNT = 36 + 16 };
#else
             SS_COUNT = 36 };
#endif
enum : int { ED_SINGLE = 41, ED_MULTI = 43, ED_MULTI_RETRY = 45, ED_AMP = 47, ED_CONT = 49, ED_CONT_AMP = 50,
             ED_CONT_AMP_RETRY = 51 };

struct EdgeScalars {
    double chi2, chi2_min;
    double n0[2], eps0[2], dn0[2], deps0[2];
    double amp_ar[3][2];                 // amplitude_accept_rates (edge:69)
    double chi2_res[3], edge_res[3][2], n0_res[3][2], wn0_res[3][2];   // edge:78-81
    unsigned long long sweep;
    unsigned long long counters[5];      // proposals, evaluated, accepted, -, Taylor lookups
    unsigned long long violations;       // check_w failures (edge:658-718)
};

struct EdgeParams {
    const double *tab[2];       // per side: [n_grid][3][ntp]  (K, dK, d2K interleaved per column)
    const double *G, *sinv;     // [128] zero padded
    double *w, *A, *dw, *mw;    // [n_chains][2][S], Julia index k = 1..N+1 kept (slot 0 unused)
    int *nm;                    // [n_chains][2][S]
    double *arc;                // [n_chains][2 side][2 move][S] accept_rates
    double *kcurr, *knew;       // [n_chains][2][S][ntp]
    double *gbar;               // [n_chains][128]
    double *wavg, *aavg;        // [n_chains][2][S]
    double *hist;               // [hist_copies][2 side][2 which][n_grid]
    EdgeScalars *sc;
    MoveRec *log;
    unsigned long long *log_len;
    long long log_cap;
    int log_chain;
    int n_chains, n_tau, ntp, NE, N, S, n_rl, double_in, fix_edge, hist_copies, hist_exclusive;
    long long n_grid, wi0;
    double A_c, A_r, d0, w_floor[2], w_m, dwg, inv_dwg, c[2];
    uint32_t seed_lo, seed_hi, chain_offset;
    int n_sweeps, flags;
    int prio_shift;             // WavePriority clock shift for this launch (host: prio_shift_for)
    double steps_in_bin, inv2theta;
};

__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const double o = __shfl_xor(v, m, 64); v = (o < v) ? o : v; }
    return v;
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const double o = __shfl_xor(v, m, 64); v = (o > v) ? o : v; }
    return v;
}

#ifdef EDGE_PROF   // development aid: cycles per section of chain 0, printed at the end of the launch
#define EDGE_TICK(i) do { const unsigned long long now_ = clock64(); if (lane == 0) reinterpret_cast<unsigned long long *>(ss + SS_PROF)[i] += now_ - tprev; tprev = now_; } while (0)
#else
#define EDGE_TICK(i) do { } while (0)
#endif

template <int NTL>
struct EdgeChain {
#ifdef EDGE_PROF
    unsigned long long tprev;
    __device__ __forceinline__ void tick(int i) { EDGE_TICK(i); }
#else
    __device__ __forceinline__ void tick(int) {}
#endif
    const EdgeParams *p;
    WavePriority prio;          // rotated per group of iterations of the move loops (wave_ops.h)
    int lane, chain;
    uint32_t chain_id, sweep_lo, sweep_hi;
    // LDS (per side arrays of S entries)
    double *w[2], *A[2], *dwn[2], *mw[2], *arc[2][2], *tmpd, *tmpd2, *tempA;
    int *nm[2], *bins;
    // global, this chain
    double *kcurr[2], *knew[2];
    uint32_t off[NTL];          // byte offset of this lane's tau element(s) inside a column
    double gbar[NTL], G[NTL], sinv[NTL];
    double chi2;                // chi2_min: ss[SS_CHI2_MIN]
    double *ss;                 // LDS: rarely used wave-uniform scalars (indices SS_*), kept out of the VGPRs
    double inv2theta;
    // this sweep's counts, wave-uniform: proposals | evaluated << 21 | accepted << 42 in one word (separate
    // variables end up in an indexed stack slot), flushed to ss[SS_CNT..] by flush_counts() every sweep
    unsigned long long cnt;
    uint32_t n_cols;
    MoveRec *log;
    long long log_cap;
    unsigned long long log_pos;

    __device__ __forceinline__ u32x4 block(uint32_t move, uint32_t iter, uint32_t b) const
    {
        return philox4x32_10(iter, sweep_lo, sweep_hi | (move << 16) | (b << 24), chain_id, p->seed_lo, p->seed_hi);
    }

    // a condition every lane evaluates alike, as a scalar (branches on it are scalar branches)
    static __device__ __forceinline__ bool uni(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }

    template <bool LOG>
    __device__ __forceinline__ void rec(int move, bool evaluated, bool accepted, double chi2_new)
    {
        cnt += 1ull + ((unsigned long long)(evaluated ? 1 : 0) << 21) + ((unsigned long long)(accepted ? 1 : 0) << 42);
        if constexpr (LOG) {
            if (log != nullptr) {
                if ((long long)log_pos < log_cap && lane == 0) {
                    MoveRec r;
                    r.chi2_new = evaluated ? chi2_new : 0.0;
                    r.flags = (evaluated ? 1 : 0) | (accepted ? 2 : 0);
                    r.move = move;
                    log[log_pos] = r;
                }
                log_pos += 1;
            }
        }
    }

    __device__ __forceinline__ double chi2_of(const double (&g)[NTL]) const
    {
        double term = 0.0;
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            const double d = (g[t] - G[t]) * sinv[t];
            term = (t == 0) ? d * d : term + d * d;
        }
        return wave_sum_uniform(term);
    }

    // floor(w / dw) with the value the correctly rounded fp64 division gives (edge:626), |result| < 2^31.
    // The reciprocal product decides unless it lies within a few ulp of an integer; then the division does.
    __device__ __forceinline__ int grid_index(double wk) const
    {
        const double q = wk * p->inv_dwg;
        const double f = __builtin_floor(q);
        const double r = q - f;
        const double m = __builtin_fabs(q) * 1e-15 + 1e-300;
        if (r > m && r < 1.0 - m) return (int)f;
        return (int)__builtin_floor(wk / p->dwg);
    }

    // K(w) by the reference's lookup (edge:626-633): this lane's tau element(s)
    __device__ __forceinline__ void taylor(int rl, double wk, double (&out)[NTL])
    {
        const int wi = grid_index(wk);
        const double d = wk - ((double)wi * p->dwg);
        const double d2 = d * d;
        long long col = (long long)wi - p->wi0;
        col = col < 0 ? 0 : (col >= p->n_grid ? p->n_grid - 1 : col);
        const char *b = reinterpret_cast<const char *>(p->tab[rl]) + (size_t)col * 3u * (size_t)p->ntp * 8u;
        const size_t cb = (size_t)p->ntp * 8u;
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            const double k0 = *reinterpret_cast<const double *>(b + off[t]);
            const double k1 = *reinterpret_cast<const double *>(b + cb + off[t]);
            const double k2 = *reinterpret_cast<const double *>(b + 2 * cb + off[t]);
            out[t] = (k0 + (k1 * d)) + (k2 * d2);
        }
        n_cols += 1;
    }

    // the three table reads of a lookup, arithmetic left to the caller (lets several lookups overlap)
    __device__ __forceinline__ void taylor_fetch(int rl, double wk, double (&raw)[3][NTL], double &d, double &d2) const
    {
        const int wi = grid_index(wk);
        d = wk - ((double)wi * p->dwg);
        d2 = d * d;
        long long col = (long long)wi - p->wi0;
        col = col < 0 ? 0 : (col >= p->n_grid ? p->n_grid - 1 : col);
        const char *b = reinterpret_cast<const char *>(p->tab[rl]) + (size_t)col * 3u * (size_t)p->ntp * 8u;
        const size_t cb = (size_t)p->ntp * 8u;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int t = 0; t < NTL; ++t) raw[q][t] = *reinterpret_cast<const double *>(b + q * cb + off[t]);
    }

    __device__ __forceinline__ double col_get(const double *base, int k, int t) const
    {
        return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + (size_t)k * p->ntp * 8u + off[t]);
    }
    __device__ __forceinline__ void col_put(double *base, int k, int t, double v) const
    {
        *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + (size_t)k * p->ntp * 8u + off[t]) = v;
    }

    // acc[t] = acc[t] + K[k][t] * f[k] for k = lo..hi in index order (multiply, then add), a batch of columns in flight
    __device__ __forceinline__ void cols_dot(const double *K, const double *f, int lo, int hi, double (&acc)[NTL]) const
    {
        constexpr int B = (NTL == 1) ? 8 : 4;
        for (int k0 = lo; k0 <= hi; k0 += B) {
            double col[B][NTL], fv[B];
#pragma unroll
            for (int u = 0; u < B; ++u) {
                const int k = (k0 + u <= hi) ? k0 + u : hi;
                fv[u] = f[k];
#pragma unroll
                for (int t = 0; t < NTL; ++t) col[u][t] = col_get(K, k, t);
            }
#pragma unroll
            for (int u = 0; u < B; ++u) {
                if (k0 + u <= hi) {
#pragma unroll
                    for (int t = 0; t < NTL; ++t) acc[t] = acc[t] + col[u][t] * fv[u];
                }
            }
        }
    }

    // a[lo] + a[lo+1] + ... in index order, starting from 0.0; LDS reads eight at a time
    __device__ __forceinline__ double lds_sum_seq(const double *a, int lo, int hi) const
    {
        double sum = 0.0;
        for (int i0 = lo; i0 <= hi; i0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a[(i0 + u <= hi) ? i0 + u : hi];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u <= hi) sum = sum + v[u];
        }
        return sum;
    }

    // ln U < (chi2 - chi2_new)/(2 theta)  <=>  rand() < P  (edge:863-864)
    __device__ __forceinline__ bool accept_lt(double chi2_new, double u) const
    {
        return log_u01(u) < (chi2 - chi2_new) * inv2theta;
    }

    __device__ __forceinline__ void commit_chi2(double chi2_new)
    {
        chi2 = chi2_new;
        if (lane == 0 && chi2_new < ss[SS_CHI2_MIN]) ss[SS_CHI2_MIN] = chi2_new;
    }

    // once per sweep (a sweep has far fewer than 2^21 proposals)
    __device__ __forceinline__ void flush_counts()
    {
        if (lane == 0) {
            unsigned long long *cu = reinterpret_cast<unsigned long long *>(ss + SS_CNT);
            cu[0] += cnt & 0x1FFFFFull;
            cu[1] += (cnt >> 21) & 0x1FFFFFull;
            cu[2] += cnt >> 42;
            cu[3] += n_cols;
        }
        cnt = 0;
        n_cols = 0;
    }

    __device__ __forceinline__ double arr_max(const double *a, int lo, int hi) const
    {
        double m = a[lo];
        for (int i = lo + lane; i <= hi; i += 64) m = (a[i] > m) ? a[i] : m;
        return wave_max(m);
    }
    __device__ __forceinline__ double arr_min(const double *a, int lo, int hi) const
    {
        double m = a[lo];
        for (int i = lo + lane; i <= hi; i += 64) m = (a[i] < m) ? a[i] : m;
        return wave_min(m);
    }

    // edge:800-813 (first = 2), edge:907-918 (first = 1); results in ss[SS_WB0 + rl], ss[SS_WBF + rl]
    template <int RL>
    __device__ __forceinline__ void edge_bounds(int first)
    {
        constexpr int rl = RL;
        double b0, bf;
        if (p->double_in) {
            b0 = -arr_max(w[1 - rl], first, p->N + 1);
            bf = -w[1 - rl][2];
        } else {
            b0 = p->w_m;
            bf = p->w_m;
        }
        if (p->A_c > 0 && rl == 0 && p->N > p->NE) {
            const double m = arr_min(w[0], p->NE + 2, p->N + 1);
            b0 = (m < b0) ? m : b0;
        }
        __syncthreads();
        if (lane == 0) { ss[SS_WB0 + rl] = b0; ss[SS_WBF + rl] = bf; }
        __syncthreads();
    }

    // edge:721-784 on the 5-point neighbourhood x[k-2..k+2] already in registers (xm2, xm1, xp1, xp2)
    __device__ __forceinline__ bool order_ok(int k, double wk, double xm2, double xm1, double xp1, double xp2) const
    {
        const int NE = p->NE;
        const double d0 = p->d0;
        if (k == 2) {
            const double d1 = xp1 - wk;
            if (d1 < d0) return false;
            if (NE > 2) { const double d2 = xp2 - xp1; if (d2 < d1) return false; }
            return true;
        }
        if (k == 3) {
            const double d1 = wk - xm1;
            if (d1 < d0) return false;
            double d2 = -1;
            if (NE > 2) { d2 = xp1 - wk; if (d2 < d1) return false; }
            if (NE > 3) { const double d3 = xp2 - xp1; if (d3 < d2) return false; }
            return true;
        }
        const double d1 = xm1 - xm2;
        const double d2 = wk - xm1;
        if (d2 < d1) return false;
        if (k == NE + 1) return true;
        const double d3 = xp1 - wk;
        if (d3 < d2) return false;
        if (k == NE) return true;
        const double d4 = xp2 - xp1;
        return !(d4 < d3);
    }

    // one-delta proposal shared by single_edge_move and single_cont_move: evaluate, accept, commit
    template <bool LOG, int RL>
    __device__ __forceinline__ void try_single(int move, int k, double wk, double Ak, double ln_u)
    {
        constexpr int rl = RL;
        double kn[NTL], gn[NTL];
        taylor(rl, wk, kn);
#pragma unroll
        for (int t = 0; t < NTL; ++t) gn[t] = gbar[t] + Ak * (kn[t] - col_get(kcurr[rl], k, t));
        const double chi2_new = chi2_of(gn);
        const bool acc = uni(ln_u < (chi2 - chi2_new) * inv2theta);
        if (acc) {
#pragma unroll
            for (int t = 0; t < NTL; ++t) { gbar[t] = gn[t]; col_put(kcurr[rl], k, t, kn[t]); }
            if (lane == 0) { w[rl][k] = wk; arc[rl][0][k] += 1.0; }
            commit_chi2(chi2_new);
            __syncthreads();
        }
        rec<LOG>(move, true, acc, chi2_new);
    }

    // edge:793-885
    template <bool LOG, int RL>
    __device__ __forceinline__ void single_edge_move()
    {
        constexpr int rl = RL;
        if (p->A_c == 1) return;
        edge_bounds<RL>(2);
        const int NE = p->NE, move = ED_SINGLE + rl;
        const int k_0 = (p->fix_edge == 1) ? 3 : 2;
        const double *x = w[rl];
        // draws of 64 iterations at a time, one iteration per lane (Philox blocks 0 and 1, U - 1/2, ln U);
        // the sequential loop below reads them with v_readlane
        for (int base = 1; base <= NE; base += 64) {
            prio.rotate();
            const int cnt = (NE - base + 1 < 64) ? (NE - base + 1) : 64;
            const u32x4 b0 = block((uint32_t)move, (uint32_t)(base + lane), 0);
            const u32x4 b1 = block((uint32_t)move, (uint32_t)(base + lane), 1);
            const int kk = k_0 + index_from_word(b0.x, (uint32_t)(NE + 2 - k_0));
            const double ush = u01(b0.y, b0.z) - 0.5;
            const double lnu = log_u01(u01(b1.x, b1.y));
            for (int j = 0; j < cnt; ++j) {
                const int i = base + j;
                int k = __builtin_amdgcn_readlane(kk, j);
                if (i % 4 == 0 && p->fix_edge == 0) k = 2;
                // one LDS round trip: the 5-point neighbourhood of x[k], its window and amplitude
                const int kp2 = (k + 2 < p->S) ? k + 2 : p->S - 1;
                const double xk = x[k], xm2 = x[k - 2], xm1 = x[k - 1], xp1 = x[k + 1], xp2 = x[kp2];
                const double dwk = dwn[rl][k], Ak = A[rl][k], b0 = ss[SS_WB0 + rl], bf = ss[SS_WBF + rl];
                const double wk = xk + (dwk * readlane_f64(ush, j));
                bool ok = true;
                if (k == 2 && wk > b0) ok = false;
                if (k < NE + 1) { if (wk < xm1 || wk > xp1) ok = false; }
                else { if (wk < xm1 || wk > p->w_m || wk > bf) ok = false; }
                if (ok) ok = order_ok(k, wk, xm2, xm1, xp1, xp2);
                ok = uni(ok);
                if (!ok) { rec<LOG>(move, false, false, 0.0); continue; }
                try_single<LOG, RL>(move, k, wk, Ak, readlane_f64(lnu, j));
            }
        }
    }

    // ascending sort of tmpd[0..n): rank sort through tmpd2 (values only; ties keep index order)
    __device__ __forceinline__ void sort_asc(int n)
    {
        __syncthreads();
        for (int e = lane; e < n; e += 64) {
            const double v = tmpd[e];
            int r = 0;
            for (int j = 0; j < n; ++j) {
                const double o = tmpd[j];
                r += (o < v || (o == v && j < e)) ? 1 : 0;
            }
            tmpd2[r] = v;
        }
        __syncthreads();
        for (int e = lane; e < n; e += 64) tmpd[e] = tmpd2[e];
        __syncthreads();
    }

    // edge:973-999
    template <int RL>
    __device__ __forceinline__ int multi_1(int n, uint32_t move, uint32_t iter)
    {
        constexpr int rl = RL;
        const double *x = w[rl];
        double *m = mw[rl];
        const double d1 = x[n + 2] - x[n + 1];
        const double d0 = p->d0;
        __syncthreads();
        for (int u = lane; u < n - 1; u += 64) {          // U_u: block 1 + u/2, words 2(u%2), +1
            const u32x4 b = block(move, iter, (uint32_t)(1 + u / 2));
            const double U = (u & 1) ? u01(b.z, b.w) : u01(b.x, b.y);
            tmpd[u] = d0 + ((d1 - d0) * U);
        }
        __syncthreads();
        double sum = 0.0;
        for (int i = 0; i < n - 1; ++i) sum = sum + tmpd[i];
        const double w1 = x[n + 1] - sum;
        if (!(w1 > p->w_floor[rl] && w1 < ss[SS_WB0 + rl])) return 0;
        if (n > 2) sort_asc(n - 1);
        double cur = w1;                                   // running position, identical on every lane
        if (lane == 0) m[2] = w1;
        for (int i = 2; i <= n; ++i) {
            cur = cur + tmpd[i - 2];
            if (lane == 0) m[i + 1] = cur;
        }
        __syncthreads();
        return 1;
    }

    // edge:1005-1052
    template <int RL>
    __device__ __forceinline__ int multi_k(int k, int n, uint32_t move, uint32_t iter)
    {
        constexpr int rl = RL;
        const double *x = w[rl];
        double *m = mw[rl];
        __syncthreads();
        for (int i = lane; i < n; i += 64) tmpd[i] = x[k + i + 1] - x[k + i];
        __syncthreads();
        const double dk_minus1 = x[k] - x[k - 1];
        const double dk_plusn = x[k + n + 1] - x[k + n];
        uint32_t retry = 0;
        const int npair = n / 2;
        const u32x4 pb0 = block(move, iter, (uint32_t)(1 + lane));          // pair q = lane
        const double pu0 = u01(pb0.z, pb0.w);
        for (int q = 0; q < npair; ++q) {
            const int ql = q & 63;
            uint32_t wx, wy;
            double U;
            if (q < 64) {
                wx = (uint32_t)__builtin_amdgcn_readlane((int)pb0.x, ql);
                wy = (uint32_t)__builtin_amdgcn_readlane((int)pb0.y, ql);
                U = readlane_f64(pu0, ql);
            } else {                                                        // clusters of more than 128 deltas: rare
                const u32x4 b = block(move, iter, (uint32_t)(1 + q));
                wx = b.x; wy = b.y; U = u01(b.z, b.w);
            }
            const int i = index_from_word(wx, (uint32_t)n);
            int j = index_from_word(wy, (uint32_t)n);
            while (j == i) {
                if (retry > 255) return 0;
                const u32x4 rb = block(move + (ED_MULTI_RETRY - ED_MULTI), iter, retry++);
                j = index_from_word(rb.x, (uint32_t)n);
            }
            const double di = tmpd[i], dj = tmpd[j];
            const double a = di + dj - dk_plusn, bb = di + dj - dk_minus1;
            const double d_min = (dk_minus1 > a) ? dk_minus1 : a;
            const double d_max = (dk_plusn < bb) ? dk_plusn : bb;
            const double ni = d_min + (d_max - d_min) * U;
            const double nj = di + dj - ni;
            __syncthreads();
            if (lane == 0) { tmpd[i] = ni; tmpd[j] = nj; }
            __syncthreads();
        }
        sort_asc(n);
        double cur = x[k];
        if (lane == 0) m[k] = cur;
        for (int i = k + 1; i <= k + n - 1; ++i) {
            cur = cur + tmpd[i - k - 1];
            if (lane == 0) m[i] = cur;
        }
        __syncthreads();
        return (m[k + n - 1] < ss[SS_WBF + rl]) ? 1 : 0;
    }

    // ---- clusters of n <= 64 deltas live in registers: lane q holds the q-th spacing / position --------------

    __device__ __forceinline__ double lane_scatter_f64(double v, int dst_lane) const
    {
        const uint64_t b = __builtin_bit_cast(uint64_t, v);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_permute(dst_lane << 2, (int)(uint32_t)b);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_permute(dst_lane << 2, (int)(uint32_t)(b >> 32));
        return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
    }

    // v of lane src_byte / 4 (per lane)
    __device__ __forceinline__ double bpermute_lane_f64(int src_byte, double v) const
    {
        const uint64_t b = __builtin_bit_cast(uint64_t, v);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src_byte, (int)(uint32_t)b);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src_byte, (int)(uint32_t)(b >> 32));
        return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
    }

    // a = lane (l & 31)'s value, b = lane (l & 31) + 32's value, in every lane l (v_permlane32_swap x, x: [h0 h0], [h1 h1])
    __device__ __forceinline__ void halves_f64(double v, double &a, double &b) const
    {
        const uint64_t bits = __builtin_bit_cast(uint64_t, v);
        const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
        const auto slo = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto shi = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        a = __builtin_bit_cast(double, ((uint64_t)shi[0] << 32) | slo[0]);
        b = __builtin_bit_cast(double, ((uint64_t)shi[1] << 32) | slo[1]);
    }

    // ascending rank sort of the values of lanes 0..n-1 (ties keep lane order, as sort_asc); other lanes keep theirs
    __device__ __forceinline__ double lane_sort_asc(double v, int n) const
    {
        int r = 0;
        for (int j = 0; j < n; ++j) {
            const double o = readlane_f64(v, j);
            r += (o < v || (o == v && j < lane)) ? 1 : 0;
        }
        return lane_scatter_f64(v, (lane < n) ? r : lane);
    }

    // running positions: lane q gets ((start + t_0) + t_1) ... + t_{q-1}; returns the last one (lane n-1's).
    // Systolic: every step each lane takes its left neighbour's value (DPP wave_shr:1) plus that neighbour's t, so lane q is
    // final after q steps and holds exactly the sequential sum.  Two lane shifts and one add per step on the vector ALU -- the
    // v_readlane form cost twice the instructions and a scalar-register round trip in every step of the dependent chain.
    __device__ __forceinline__ double lane_running(double start, double t, int n, double &mine) const
    {
        double tprev = dpp_f64<0x138>(t);                 // wave_shr:1: lane q <- t_{q-1}, lane 0 <- +0
        tprev = (lane == 0) ? start : tprev;              // lane 0: 0 + start = start in every step
        double cur = start;
        for (int i = 1; i < n; ++i) cur = dpp_f64<0x138>(cur) + tprev;
        if (n <= 1) cur = start;
        mine = cur;
        return readlane_f64(cur, n - 1);
    }

    // multi_1 (edge:973-999), 2 <= n <= 64 positions; lane q returns the new w[2 + q] in mine
    template <int RL>
    __device__ __forceinline__ int multi_1_reg(int n, uint32_t move, uint32_t iter, double &mine)
    {
        constexpr int rl = RL;
        const double *x = w[rl];
        const double xa = x[n + 1], xb = x[n + 2];
        const double d1 = xb - xa;
        const double d0 = p->d0;
        const u32x4 b = block(move, iter, (uint32_t)(1 + lane / 2));          // U_u, u = lane
        const double U = (lane & 1) ? u01(b.z, b.w) : u01(b.x, b.y);
        double t = d0 + ((d1 - d0) * U);
        double part;
        const double sum = lane_running(0.0, t, n, part);   // ((t_0 + t_1) + ...) + t_{n-2}, in index order (0 + t_0 = t_0)
        const double w1 = xa - sum;
        if (!uni(w1 > p->w_floor[rl] && w1 < ss[SS_WB0 + rl])) return 0;
        if (n > 2) t = lane_sort_asc(t, n - 1);
        lane_running(w1, t, n, mine);
        if (lane < n) mw[rl][2 + lane] = mine;
        return 1;
    }

    // multi_k (edge:1005-1052), 1 <= n <= 64; lane q returns the new w[k + q] in mine
    template <int RL>
    __device__ __forceinline__ int multi_k_reg(int k, int n, uint32_t move, uint32_t iter, const u32x4 &pair0, const u32x4 &retry0,
                                               int jj, double &mine)
    {
        constexpr int rl = RL;
        const double *x = w[rl];
        const int q0 = (lane < n) ? lane : n - 1;
        const double xa = x[k + q0], xb = x[k + q0 + 1];
        const double xkm1 = x[k - 1], xkn1 = x[k + n + 1];
        double t = xb - xa;                                                  // spacing q of the cluster
        const double xk = readlane_f64(xa, 0);
        const double dk_minus1 = xk - xkm1;
        const double dk_plusn = xkn1 - readlane_f64(xb, n - 1);
        uint32_t retry = 0;
        const int npair = n / 2;
        // pair 0's block and the first retry block of this iteration were drawn by lane jj of the caller's group
        // (pair0, retry0); clusters of 4 and more draw pairs 1.. here, pair q on lane q.  The pair's two indices and its
        // uniform are decoded by the drawing lane (lane-parallel, off the sequential path) and read from there.
        int ip = index_from_word(pair0.x, (uint32_t)n), jp = index_from_word(pair0.y, (uint32_t)n);
        double Up = u01(pair0.z, pair0.w);
        int src = jj;
        for (int q = 0; q < npair; ++q) {
            if (q == 1) {
                const u32x4 pb = block(move, iter, (uint32_t)(1 + lane));
                ip = index_from_word(pb.x, (uint32_t)n);
                jp = index_from_word(pb.y, (uint32_t)n);
                Up = u01(pb.z, pb.w);
                src = 1;
            }
            const int i = __builtin_amdgcn_readlane(ip, src);
            int j = __builtin_amdgcn_readlane(jp, src);
            const double U = readlane_f64(Up, src);
            if (q >= 1) src += 1;
            while (j == i) {
                if (retry > 255) return 0;
                uint32_t rx;
                if (retry == 0) rx = (uint32_t)__builtin_amdgcn_readlane((int)retry0.x, jj);
                else rx = (uint32_t)__builtin_amdgcn_readfirstlane((int)block(move + (ED_MULTI_RETRY - ED_MULTI), iter, retry).x);
                retry += 1;
                j = index_from_word(rx, (uint32_t)n);
            }
            const double di = readlane_f64(t, i), dj = readlane_f64(t, j);
            const double a = di + dj - dk_plusn, bb = di + dj - dk_minus1;
            const double d_min = (dk_minus1 > a) ? dk_minus1 : a;
            const double d_max = (dk_plusn < bb) ? dk_plusn : bb;
            const double ni = d_min + (d_max - d_min) * U;
            const double nj = di + dj - ni;
            t = (lane == i) ? ni : ((lane == j) ? nj : t);
        }
        EDGE_TICK(14);
        if (n > 1) t = lane_sort_asc(t, n);
        EDGE_TICK(15);
        const double last = lane_running(xk, t, n, mine);
        if (lane < n) mw[rl][k + lane] = mine;
        return uni(last < ss[SS_WBF + rl]) ? 1 : 0;
    }

    // evaluate and accept/reject the cluster k..k+n-1 at the positions held by lanes 0..n-1 (edge:936-966).
    // Lane q decodes column q (grid index, Taylor offsets, table offset, amplitude); the columns are then
    // accumulated in index order, CH table rows in flight at a time.
    template <bool LOG, int RL>
    __device__ __forceinline__ void cluster_try(uint32_t move, int k, int n, double mine, double ln_u)
    {
        constexpr int rl = RL;
        constexpr int CH = (NTL == 1) ? 4 : 2;
        const int q0 = (lane < n) ? lane : n - 1;
        const int wi = grid_index(mine);
        const double dq = mine - ((double)wi * p->dwg);
        const double dq2 = dq * dq;
        long long col = (long long)wi - p->wi0;
        col = col < 0 ? 0 : (col >= p->n_grid ? p->n_grid - 1 : col);
        const uint32_t cb = (uint32_t)p->ntp * 8u;
        const uint64_t boff = (uint64_t)col * (uint64_t)(3u * cb);
        const double Aq = A[rl][k + q0];
        const char *tabb = reinterpret_cast<const char *>(p->tab[rl]);
        const char *kc = reinterpret_cast<const char *>(kcurr[rl]) + (size_t)k * cb;
        char *kw = reinterpret_cast<char *>(knew[rl]) + (size_t)k * cb;
        double gn[NTL], raw[CH][3][NTL], old[CH][NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) gn[t] = gbar[t];
        auto fetch = [&](int u, int j) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)boff, j);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(boff >> 32), j);
            const char *b = tabb + (((uint64_t)hi << 32) | lo);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < NTL; ++t) raw[u][r][t] = *reinterpret_cast<const double *>(b + r * cb + off[t]);
#pragma unroll
            for (int t = 0; t < NTL; ++t) old[u][t] = *reinterpret_cast<const double *>(kc + (uint32_t)j * cb + off[t]);
        };
        EDGE_TICK(5);
        // batches of CH columns: all loads of a batch are issued (unconditionally, index clamped) before its
        // first column is consumed
        const bool small = (n <= CH);               // one batch: the new columns stay in registers, K_new is not used
        for (int j0 = 0; j0 < n; j0 += CH) {
#pragma unroll
            for (int u = 0; u < CH; ++u) fetch(u, (j0 + u < n) ? j0 + u : n - 1);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int j = j0 + u;
                if (j < n) {
                    const double d = readlane_f64(dq, j), d2 = readlane_f64(dq2, j), Ak = readlane_f64(Aq, j);
#pragma unroll
                    for (int t = 0; t < NTL; ++t) {
                        const double kn = (raw[u][0][t] + (raw[u][1][t] * d)) + (raw[u][2][t] * d2);
                        gn[t] = gn[t] + Ak * (kn - old[u][t]);
                        raw[u][0][t] = kn;                               // kept for the commit of a small cluster
                        if (!small) *reinterpret_cast<double *>(kw + (uint32_t)j * cb + off[t]) = kn;
                    }
                }
            }
        }
        n_cols += (uint32_t)n;
        EDGE_TICK(6);
        const double chi2_new = chi2_of(gn);
        const bool acc = uni(ln_u < (chi2 - chi2_new) * inv2theta);
        EDGE_TICK(7);
        if (acc) {
#pragma unroll
            for (int t = 0; t < NTL; ++t) gbar[t] = gn[t];
            if (small) {
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    if (u < n) {
#pragma unroll
                        for (int t = 0; t < NTL; ++t)
                            *reinterpret_cast<double *>(const_cast<char *>(kc) + (uint32_t)u * cb + off[t]) = raw[u][0][t];
                    }
                }
            } else for (int j0 = 0; j0 < n; j0 += 4) {          // K_curr columns <- K_new, four in flight
                double cp[4][NTL];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = (j0 + u < n) ? j0 + u : n - 1;
#pragma unroll
                    for (int t = 0; t < NTL; ++t) cp[u][t] = *reinterpret_cast<const double *>(kw + (uint32_t)j * cb + off[t]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (j0 + u < n) {
#pragma unroll
                        for (int t = 0; t < NTL; ++t)
                            *reinterpret_cast<double *>(const_cast<char *>(kc) + (uint32_t)(j0 + u) * cb + off[t]) = cp[u][t];
                    }
                }
            }
            if (lane < n) w[rl][k + lane] = mine;
            if (lane == 0) arc[rl][1][k] += 1.0;
            commit_chi2(chi2_new);
            __syncthreads();
        }
        rec<LOG>((int)move, true, acc, chi2_new);
    }

    // cluster_try for N_tau <= 32 (one tau slot per lane, half the wave idle in cluster_try): TWO columns per wave
    // instruction -- lanes 0..31 work on column j, lanes 32..63 on column j + 1 of a pair.  A lane takes its column's decode
    // (table offset, Taylor offset, amplitude) from the decoding lane by ds_bpermute instead of one v_readlane per value and
    // column, the three table rows and the K_curr column of both columns are one load instruction each, and the increments
    // A (K_new - K_curr) of the pair reach every lane by v_permlane32_swap, lower half's first: Gbar_new is accumulated in
    // index order exactly as in cluster_try.  (Lanes 32..63 follow the same sums; their sigma^-1 is zero.)
    template <bool LOG, int RL>
    __device__ __forceinline__ void cluster_try_pairs(uint32_t move, int k, int n, double mine, double ln_u)
    {
        static_assert(NTL == 1, "one tau slot per lane");
        constexpr int rl = RL;
        constexpr int CP = 4;                          // column pairs in flight
        const int wi = grid_index(mine);
        const double dq = mine - ((double)wi * p->dwg);
        long long col = (long long)wi - p->wi0;
        col = col < 0 ? 0 : (col >= p->n_grid ? p->n_grid - 1 : col);
        const uint32_t cb = (uint32_t)p->ntp * 8u;
        const uint64_t boff = (uint64_t)col * (uint64_t)(3u * cb);
        const int q0 = (lane < n) ? lane : n - 1;
        const double Aq = A[rl][k + q0];
        const char *tabb = reinterpret_cast<const char *>(p->tab[rl]);
        const char *kc = reinterpret_cast<const char *>(kcurr[rl]) + (size_t)k * cb;
        char *kw = reinterpret_cast<char *>(knew[rl]) + (size_t)k * cb;
        const int half = lane >> 5;
        const int tl = lane & 31;
        const uint32_t off2 = (uint32_t)((tl < p->ntp) ? tl : p->ntp - 1) * 8u;
        double gn[1] = {gbar[0]};
        double raw[CP][3], old[CP], dd[CP], aa[CP];
        EDGE_TICK(5);
        const bool small = (n <= 2 * CP);              // one batch: the new columns stay in registers, K_new is not used
        for (int j0 = 0; j0 < n; j0 += 2 * CP) {
#pragma unroll
            for (int u = 0; u < CP; ++u) {
                const int jc = j0 + 2 * u + half;
                const int src = ((jc < n) ? jc : n - 1) << 2;
                const uint32_t blo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)boff);
                const uint32_t bhi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(boff >> 32));
                dd[u] = bpermute_lane_f64(src, dq);
                aa[u] = bpermute_lane_f64(src, Aq);
                const char *b = tabb + (((uint64_t)bhi << 32) | blo);
#pragma unroll
                for (int r = 0; r < 3; ++r) raw[u][r] = *reinterpret_cast<const double *>(b + r * cb + off2);
                old[u] = *reinterpret_cast<const double *>(kc + (uint32_t)(src >> 2) * cb + off2);
            }
#pragma unroll
            for (int u = 0; u < CP; ++u) {
                const int ja = j0 + 2 * u;             // the pair's first column (lower half), ja + 1 the upper half's
                if (ja < n) {
                    const double d = dd[u];
                    const double kn = (raw[u][0] + (raw[u][1] * d)) + (raw[u][2] * (d * d));
                    const double inc = aa[u] * (kn - old[u]);
                    double inc_a, inc_b;
                    halves_f64(inc, inc_a, inc_b);     // lower / upper half's value in every lane
                    gn[0] = gn[0] + inc_a;
                    if (ja + 1 < n) gn[0] = gn[0] + inc_b;
                    raw[u][0] = kn;                    // kept for the commit of a small cluster
                    if (!small && ja + half < n) *reinterpret_cast<double *>(kw + (uint32_t)(ja + half) * cb + off2) = kn;
                }
            }
        }
        n_cols += (uint32_t)n;
        EDGE_TICK(6);
        const double chi2_new = chi2_of(gn);
        const bool acc = uni(ln_u < (chi2 - chi2_new) * inv2theta);
        EDGE_TICK(7);
        if (acc) {
            gbar[0] = gn[0];
            if (small) {
#pragma unroll
                for (int u = 0; u < CP; ++u) {
                    const int jc = 2 * u + half;
                    if (jc < n) *reinterpret_cast<double *>(const_cast<char *>(kc) + (uint32_t)jc * cb + off2) = raw[u][0];
                }
            } else for (int j0 = 0; j0 < n; j0 += 8) {          // K_curr columns <- K_new, four pairs in flight
                double cp[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jc = j0 + 2 * u + half;
                    cp[u] = *reinterpret_cast<const double *>(kw + (uint32_t)((jc < n) ? jc : n - 1) * cb + off2);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jc = j0 + 2 * u + half;
                    if (jc < n) *reinterpret_cast<double *>(const_cast<char *>(kc) + (uint32_t)jc * cb + off2) = cp[u];
                }
            }
            if (lane < n) w[rl][k + lane] = mine;
            if (lane == 0) arc[rl][1][k] += 1.0;
            commit_chi2(chi2_new);
            __syncthreads();
        }
        rec<LOG>((int)move, true, acc, chi2_new);
    }

    // edge:889-968
    template <bool LOG, int RL>
    __device__ __forceinline__ void multi_edge_move()
    {
        constexpr int rl = RL;
        if (p->A_c == 1) return;
        const int NE = p->NE;
        if (NE < 4) return;
        edge_bounds<RL>(1);
        const uint32_t move = (uint32_t)(ED_MULTI + rl);
        for (int base = 0; base <= NE - 1; base += 64) {
          const int cnt = (NE - base < 64) ? (NE - base) : 64;
          const u32x4 b0 = block(move, (uint32_t)(base + lane), 0);      // one iteration per lane
          const int kk = 3 + index_from_word(b0.x, (uint32_t)(NE - 3));
          const double lnu = log_u01(u01(b0.y, b0.z));
          const u32x4 pair0 = block(move, (uint32_t)(base + lane), 1);   // multi_k: pair 0 and first retry of iteration base + lane
          const u32x4 retry0 = block(move + (ED_MULTI_RETRY - ED_MULTI), (uint32_t)(base + lane), 0);
          for (int jj = 0; jj < cnt; ++jj) {
            const int i = base + jj;
            if ((jj & 15) == 0) prio.rotate();
            int k, n, pass;
            EDGE_TICK(1);
            const bool first = (i % 4 == 0 && p->fix_edge == 0);
            k = first ? 2 : __builtin_amdgcn_readlane(kk, jj);
            n = __builtin_amdgcn_readfirstlane(nm[rl][k]);
            if (n <= 64) {                                   // the usual case: cluster in registers
                double mine;
                EDGE_TICK(2);
                pass = first ? multi_1_reg<RL>(n, move, (uint32_t)i, mine) : multi_k_reg<RL>(k, n, move, (uint32_t)i, pair0, retry0, jj, mine);
                EDGE_TICK(first ? 3 : 4);
                if (pass != 1) { rec<LOG>((int)move, false, false, 0.0); continue; }
                if constexpr (NTL == 1) {
                if (p->ntp <= 32) cluster_try_pairs<LOG, RL>(move, k, n, mine, readlane_f64(lnu, jj));
                else cluster_try<LOG, RL>(move, k, n, mine, readlane_f64(lnu, jj));
            } else {
                cluster_try<LOG, RL>(move, k, n, mine, readlane_f64(lnu, jj));
            }
                EDGE_TICK(8);
                continue;
            }
            pass = first ? multi_1<RL>(n, move, (uint32_t)i) : multi_k<RL>(k, n, move, (uint32_t)i);
            if (pass != 1) { rec<LOG>((int)move, false, false, 0.0); continue; }
            double gn[NTL];
#pragma unroll
            for (int t = 0; t < NTL; ++t) gn[t] = gbar[t];
            // the cluster's columns: fetched CH at a time (independent gathers in flight together),
            // accumulated in index order (edge:938-940)
            constexpr int CH = 2;
            for (int j0 = 0; j0 < n; j0 += CH) {
                double raw[CH][3][NTL], old[CH][NTL], dd[CH], dd2[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int j = (j0 + u < n) ? j0 + u : n - 1;
                    taylor_fetch(rl, mw[rl][k + j], raw[u], dd[u], dd2[u]);
#pragma unroll
                    for (int t = 0; t < NTL; ++t) old[u][t] = col_get(kcurr[rl], k + j, t);
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    if (j0 + u < n) {
                        const int j = j0 + u;
                        const double Ak = A[rl][k + j];
#pragma unroll
                        for (int t = 0; t < NTL; ++t) {
                            const double kn = (raw[u][0][t] + (raw[u][1][t] * dd[u])) + (raw[u][2][t] * dd2[u]);
                            gn[t] = gn[t] + Ak * (kn - old[u][t]);
                            col_put(knew[rl], k + j, t, kn);
                        }
                        n_cols += 1;
                    }
                }
            }
            const double chi2_new = chi2_of(gn);
            const bool acc = uni(readlane_f64(lnu, jj) < (chi2 - chi2_new) * inv2theta);
            if (acc) {
#pragma unroll
                for (int t = 0; t < NTL; ++t) gbar[t] = gn[t];
                for (int j = k; j <= k + n - 1; ++j) {
#pragma unroll
                    for (int t = 0; t < NTL; ++t) col_put(kcurr[rl], j, t, col_get(knew[rl], j, t));
                }
                __syncthreads();
                for (int j = k + lane; j <= k + n - 1; j += 64) w[rl][j] = mw[rl][j];
                if (lane == 0) arc[rl][1][k] += 1.0;
                commit_chi2(chi2_new);
                __syncthreads();
            }
            rec<LOG>((int)move, true, acc, chi2_new);
          }
        }
    }

    // amplitude profile of the edge into tempA[2..NE+1], normalised to f1 (edge:1091-1110)
    template <int RL>
    __device__ __forceinline__ void edge_profile(double n1, double eps_prod)
    {
        constexpr int rl = RL;
        const double c = p->c[rl];
        const int NE = p->NE;
        __syncthreads();
        for (int i = 1 + lane; i <= NE; i += 64) {
            const double x = log_u01((double)i / n1);
            const double cx = c * x;
            const double root = __builtin_sqrt((cx * cx) + eps_prod);
            const double ln_Ai = (c < 0) ? (cx + root) : (cx - root);
            tempA[i + 1] = sac_exp(ln_Ai / 2);
        }
        __syncthreads();
        const double sum = lds_sum_seq(tempA, 2, NE + 1);
        const double f1 = (rl == 0) ? (p->A_r / (1 + p->A_c)) : ((1 - p->A_r) / (1 + p->A_c));
        const double scale = f1 / sum;
        __syncthreads();
        for (int i = 2 + lane; i <= NE + 1; i += 64) tempA[i] = tempA[i] * scale;
        __syncthreads();
    }

    // edge:1057-1142
    template <bool LOG, int RL>
    __device__ __forceinline__ void amplitude_edge_move()
    {
        constexpr int rl = RL;
        if (p->A_c == 1) return;
        const int NE = p->NE, move = ED_AMP + rl;
        if (NE < 3) return;
        const u32x4 b0 = block((uint32_t)move, 0, 0);
        const double c = p->c[rl];
        const double eps = (0.69314718055994530942 * c) * (0.69314718055994530942 * c);
        double n1 = ss[SS_N0 + rl], e1 = ss[SS_EPS0 + rl];
        n1 += (ss[SS_DN0 + rl] * (u01(b0.x, b0.y) - 0.5));
        if (n1 < 5 || n1 > (NE - 5)) { rec<LOG>(move, false, false, 0.0); return; }
        e1 += ss[SS_DEPS0 + rl] * (u01(b0.z, b0.w) - 0.5);
        if (e1 < 0 || e1 > 1) { rec<LOG>(move, false, false, 0.0); return; }
        edge_profile<RL>(n1, e1 * eps);
        double gn[NTL], acc_[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc_[t] = 0.0;
        for (int k = 2 + lane; k <= NE + 1; k += 64) tmpd[k] = tempA[k] - A[rl][k];
        __syncthreads();
        cols_dot(kcurr[rl], tmpd, 2, NE + 1, acc_);             // edge:1113-1115
#pragma unroll
        for (int t = 0; t < NTL; ++t) gn[t] = gbar[t] + acc_[t];
        const double chi2_new = chi2_of(gn);
        const u32x4 b1 = block((uint32_t)move, 0, 1);
        const bool acc = uni(accept_lt(chi2_new, u01(b1.x, b1.y)));
        if (acc) {
            if (lane == 0) { ss[SS_N0 + rl] = n1; ss[SS_EPS0 + rl] = e1; ss[SS_AMP_AR + rl] += 1; }
            __syncthreads();
            for (int k = 2 + lane; k <= NE + 1; k += 64) A[rl][k] = tempA[k];
#pragma unroll
            for (int t = 0; t < NTL; ++t) gbar[t] = gn[t];
            commit_chi2(chi2_new);
            __syncthreads();
        }
        rec<LOG>(move, true, acc, chi2_new);
    }

    // edge:1146-1189
    template <bool LOG>
    __device__ __forceinline__ void single_cont_move()
    {
        const int NE = p->NE, N = p->N;
        if (NE == N || p->A_c == 0) return;
        const double *x = w[0];
        const double wc_min = x[2], wc_max = x[NE + 1];
        for (int base = 0; base < N - NE; base += 64) {
            prio.rotate();
            const int cnt = (N - NE - base < 64) ? (N - NE - base) : 64;
            const u32x4 b0 = block(ED_CONT, (uint32_t)(base + lane), 0);
            const u32x4 b1 = block(ED_CONT, (uint32_t)(base + lane), 1);
            const int kk = NE + 2 + index_from_word(b0.x, (uint32_t)(N - NE));
            const double ush = u01(b0.y, b0.z) - 0.5;
            const double lnu = log_u01(u01(b1.x, b1.y));
            for (int j = 0; j < cnt; ++j) {
                const int k = __builtin_amdgcn_readlane(kk, j);
                const double wk = x[k] + (dwn[0][k] * readlane_f64(ush, j));
                if (wk < wc_min || wk > wc_max) { rec<LOG>(ED_CONT, false, false, 0.0); continue; }
                try_single<LOG, 0>(ED_CONT, k, wk, A[0][k], readlane_f64(lnu, j));
            }
        }
    }

    // edge:1194-1252
    template <bool LOG>
    __device__ __forceinline__ void amplitude_cont_move()
    {
        const int NE = p->NE, N = p->N;
        if (NE == N || p->A_c == 0) return;
        const int ncand = N - NE - 1;
        if (ncand < 2) return;
        int n_accept = 0;
        for (int i = 1; i <= N; ++i) {
            const u32x4 b0 = block(ED_CONT_AMP, (uint32_t)i, 0);
            const int j = NE + 2 + index_from_word(b0.x, (uint32_t)ncand);
            int k = NE + 2 + index_from_word(b0.y, (uint32_t)ncand);
            uint32_t retry = 0;
            bool give_up = false;
            while (k == j) {
                if (retry > 255) { give_up = true; break; }
                const u32x4 rb = block(ED_CONT_AMP_RETRY, (uint32_t)i, retry++);
                k = NE + 2 + index_from_word(rb.x, (uint32_t)ncand);
            }
            if (give_up) { rec<LOG>(ED_CONT_AMP, false, false, 0.0); continue; }
            const double A_j = A[0][j], A_k = A[0][k];
            const double m0 = A_j + A_k;
            const double r = u01(b0.z, b0.w);
            const double dA_j = (r * m0) - A_j;
            const double dA_k = ((1 - r) * m0) - A_k;
            double gn[NTL];
#pragma unroll
            for (int t = 0; t < NTL; ++t)
                gn[t] = (gbar[t] + dA_j * col_get(kcurr[0], j, t)) + dA_k * col_get(kcurr[0], k, t);
            const double chi2_new = chi2_of(gn);
            const u32x4 b1 = block(ED_CONT_AMP, (uint32_t)i, 1);
            const bool acc = uni(accept_lt(chi2_new, u01(b1.x, b1.y)));
            if (acc) {
                __syncthreads();
                if (lane == 0) { A[0][j] = A_j + dA_j; A[0][k] = A_k + dA_k; }
#pragma unroll
                for (int t = 0; t < NTL; ++t) gbar[t] = gn[t];
                commit_chi2(chi2_new);
                n_accept += 1;
                __syncthreads();
            }
            rec<LOG>(ED_CONT_AMP, true, acc, chi2_new);
        }
        __syncthreads();
        if (lane == 0) ss[SS_AMP_AR + 2] += (double)n_accept / (double)(N - NE);
        __syncthreads();
    }
};

}  // namespace sacdev
