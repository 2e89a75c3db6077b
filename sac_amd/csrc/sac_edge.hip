// libsac_hip.so, edge parameterisation: stage kernel + C ABI (sac_edge_* in include/sac_hip.h).
// "edge:NNN" = /root/reference/sac/edge/sac_edge.jl line NNN.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "../../include/sac_hip.h"
#include "edge_sweep.h"
#include "table_build.h"
#include "cv_score.h"
#include "comm.h"
#include "devbuf.h"

using namespace sacdev;

int sac_internal_fail(int code, const char *msg);   // sac_hip.hip

namespace {

int efail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return sac_internal_fail(code, buf);
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return efail(SAC_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                         __FILE__, __LINE__);                                              \
    } while (0)

// LDS bytes of one chain: per sampled side 6 double arrays, 2 shared double arrays, 3 int arrays of S entries,
// and the scalar block
constexpr size_t edge_lds_bytes(int S, int n_rl) { return (size_t)S * ((6u * (unsigned)n_rl + 2u) * 8u + 3u * 4u) + SS_COUNT * 8u + 64u; }

}  // namespace

// One launch = n_sweeps sweeps of run_updates (edge:1260-1302) for every chain; by flag the
// start-of-bin work (zero the bin accumulators and accept counters, Gbar = K_curr A) and the
// end-of-bin work (record_bin, check_w, adjust_D).  One wave = one chain.
#ifndef EDGE_WAVES
#define EDGE_WAVES 4
#endif
template <int NTL, bool LOG>
__global__ void __launch_bounds__(64, EDGE_WAVES) edge_stage_kernel(const EdgeParams p)
{
    extern __shared__ __align__(16) unsigned char s_dyn[];
    const int chain = (int)blockIdx.x;
    const int lane = lane_id();
    const int S = p.S, N = p.N, NE = p.NE;
    // The hot scalars become values of their own: left as lanes of the wide kernel-argument loads they are
    // spilled and reloaded a whole tuple at a time inside the move loops.
    EdgeParams hp = p;
#define SAC_OWN(x) asm volatile("" : "+s"(x))
    SAC_OWN(hp.tab[0]); SAC_OWN(hp.tab[1]); SAC_OWN(hp.ntp); SAC_OWN(hp.NE); SAC_OWN(hp.N); SAC_OWN(hp.S);
    SAC_OWN(hp.double_in); SAC_OWN(hp.fix_edge); SAC_OWN(hp.n_grid); SAC_OWN(hp.wi0); SAC_OWN(hp.A_c); SAC_OWN(hp.A_r);
    SAC_OWN(hp.d0); SAC_OWN(hp.w_floor[0]); SAC_OWN(hp.w_floor[1]); SAC_OWN(hp.w_m); SAC_OWN(hp.dwg); SAC_OWN(hp.inv_dwg);
    SAC_OWN(hp.c[0]); SAC_OWN(hp.c[1]); SAC_OWN(hp.seed_lo); SAC_OWN(hp.seed_hi); SAC_OWN(hp.inv2theta);
#undef SAC_OWN
    EdgeChain<NTL> c;
    c.p = &hp;
    c.lane = lane;
    c.prio.init(p.prio_shift);
    c.chain = chain;
    {
        double *d = reinterpret_cast<double *>(s_dyn);
        c.ss = d; d += SS_COUNT;
        _Pragma("unroll") for (int rl = 0; rl < 2; ++rl) {
            if (rl < p.n_rl) {
                c.w[rl] = d; d += S;
                c.A[rl] = d; d += S;
                c.dwn[rl] = d; d += S;
                c.mw[rl] = d; d += S;
                c.arc[rl][0] = d; d += S;
                c.arc[rl][1] = d; d += S;
            } else {                                       // single-sided modes: side 1 is never touched
                c.w[rl] = c.w[0]; c.A[rl] = c.A[0]; c.dwn[rl] = c.dwn[0]; c.mw[rl] = c.mw[0];
                c.arc[rl][0] = c.arc[0][0]; c.arc[rl][1] = c.arc[0][1];
            }
        }
        c.tmpd = d; d += S;
        c.tmpd2 = d; d += S;
        c.tempA = c.tmpd2;                                 // amplitude profile and cluster sort never overlap
        int *ip = reinterpret_cast<int *>(d);
        c.nm[0] = ip; ip += S;
        c.nm[1] = (p.n_rl == 2) ? ip : c.nm[0];
        ip += S;
        c.bins = ip;
    }
    const size_t cs = (size_t)chain * 2u * (size_t)S;
    _Pragma("unroll") for (int rl = 0; rl < 2; ++rl) {
        const size_t o = cs + (size_t)rl * S;
        if (rl < p.n_rl) for (int k = lane; k < S; k += 64) {
            c.w[rl][k] = p.w[o + k];
            c.A[rl][k] = p.A[o + k];
            c.dwn[rl][k] = p.dw[o + k];
            c.mw[rl][k] = p.mw[o + k];
            c.nm[rl][k] = p.nm[o + k];
            c.arc[rl][0][k] = p.arc[(cs + (size_t)rl * S) * 2u + k];
            c.arc[rl][1][k] = p.arc[(cs + (size_t)rl * S) * 2u + S + k];
        }
        c.kcurr[rl] = p.kcurr + (cs + (size_t)rl * S) * (size_t)p.ntp;
        c.knew[rl] = p.knew + (cs + (size_t)rl * S) * (size_t)p.ntp;
    }
    double *gbar_g = p.gbar + (size_t)chain * SAC_MAX_N_TAU;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        c.off[t] = (uint32_t)((j < p.ntp) ? j : p.ntp - 1) * 8u;
        c.gbar[t] = gbar_g[j];
        c.G[t] = p.G[j];
        c.sinv[t] = p.sinv[j];
    }
    EdgeScalars *scp = p.sc + chain;
    c.chi2 = scp->chi2;
    if (lane == 0) {
        c.ss[SS_CHI2_MIN] = scp->chi2_min;
        for (int q = 0; q < 4; ++q) reinterpret_cast<unsigned long long *>(c.ss + SS_CNT)[q] = 0ull;
        for (int rl = 0; rl < 2; ++rl) {
            c.ss[SS_WB0 + rl] = p.w_m; c.ss[SS_WBF + rl] = p.w_m;
            c.ss[SS_N0 + rl] = scp->n0[rl]; c.ss[SS_EPS0 + rl] = scp->eps0[rl];
            c.ss[SS_DN0 + rl] = scp->dn0[rl]; c.ss[SS_DEPS0 + rl] = scp->deps0[rl];
            for (int m = 0; m < 3; ++m) c.ss[SS_AMP_AR + 2 * m + rl] = scp->amp_ar[m][rl];
            c.ss[SS_EDGE_ACC + rl] = scp->edge_res[0][rl];
            c.ss[SS_N0_ACC + rl] = scp->n0_res[0][rl];
            c.ss[SS_WN0_ACC + rl] = scp->wn0_res[0][rl];
        }
        c.ss[SS_CHI2_ACC] = scp->chi2_res[0];
    }
    unsigned long long sweep = scp->sweep;
    c.inv2theta = p.inv2theta;
    c.chain_id = p.chain_offset + (uint32_t)chain;
    c.cnt = 0;
    c.n_cols = 0;
#ifdef EDGE_PROF
    if (lane == 0) for (int q = 0; q < 16; ++q) reinterpret_cast<unsigned long long *>(c.ss + SS_PROF)[q] = 0ull;
    c.tprev = clock64();
#endif
    c.log = nullptr;
    c.log_cap = 0;
    c.log_pos = 0;
    if constexpr (LOG) {
        if (chain == p.log_chain) { c.log = p.log; c.log_cap = p.log_cap; c.log_pos = *p.log_len; }
    }
    __syncthreads();

    if (p.flags & EDGE_BIN_START) {                       // edge:1263-1270
        if (lane == 0) {
            c.ss[SS_CHI2_ACC] = 0.0;
            for (int q = SS_EDGE_ACC; q < SS_EDGE_ACC + 6; ++q) c.ss[q] = 0.0;
        }
        _Pragma("unroll") for (int rl = 0; rl < 2; ++rl) {
            if (rl < p.n_rl) for (int k = lane; k < S; k += 64) { c.arc[rl][0][k] = 0.0; c.arc[rl][1][k] = 0.0; }
        }
        __syncthreads();
        double acc[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[t] = 0.0;
        c.cols_dot(c.kcurr[0], c.A[0], 1, N + 1, acc);    // calc_Gbar!(self, 0), edge:615
#pragma unroll
        for (int t = 0; t < NTL; ++t) c.gbar[t] = acc[t];
        if (p.n_rl == 2) {                                // edge:618-620
#pragma unroll
            for (int t = 0; t < NTL; ++t) acc[t] = 0.0;
            c.cols_dot(c.kcurr[1], c.A[1], 1, N + 1, acc);
#pragma unroll
            for (int t = 0; t < NTL; ++t) c.gbar[t] = c.gbar[t] + acc[t];
        }
    }

    double *hist = p.hist + (size_t)(chain % p.hist_copies) * 4u * (size_t)p.n_grid;
    for (int s = 0; s < p.n_sweeps; ++s) {
        c.sweep_lo = (uint32_t)sweep;
        c.sweep_hi = (uint32_t)(sweep >> 32) & 0xFFFFu;
        c.tick(0);
        c.template single_edge_move<LOG, 0>();            // edge:1273-1281
        c.tick(10);
        c.template multi_edge_move<LOG, 0>();
        c.tick(11);
        c.template amplitude_edge_move<LOG, 0>();
        c.tick(12);
        if (p.n_rl == 2) {
            c.template single_edge_move<LOG, 1>();
            c.template multi_edge_move<LOG, 1>();
            c.template amplitude_edge_move<LOG, 1>();
        }
        if (p.A_c > 0) {                                  // edge:1283-1287
            c.template single_cont_move<LOG>();
            c.template amplitude_cont_move<LOG>();
        }
        // measure (edge:1305-1333)
        __syncthreads();
        if (lane == 0) c.ss[SS_CHI2_ACC] += c.chi2;
        _Pragma("unroll") for (int rl = 0; rl < 2; ++rl) if (rl < p.n_rl) {
            if (lane == 0) {
                const double n0v = c.ss[SS_N0 + rl];
                c.ss[SS_N0_ACC + rl] += n0v / NE;
                int idx = (int)__builtin_floor(n0v);
                idx = idx < 1 ? 1 : (idx > N + 1 ? N + 1 : idx);
                c.ss[SS_WN0_ACC + rl] += c.w[rl][idx];
                c.ss[SS_EDGE_ACC + rl] += c.w[rl][2];
            }
            double *wavg = p.wavg + cs + (size_t)rl * S, *aavg = p.aavg + cs + (size_t)rl * S;
            for (int k = 1 + lane; k <= N + 1; k += 64) {
                aavg[k] += c.A[rl][k];
                wavg[k] += c.w[rl][k];
            }
            // histograms: bin of delta i (1..N) at Julia index i+1; -1 = outside the grid
            for (int i = 1 + lane; i <= N; i += 64) {
                const long long wi = (long long)c.grid_index(c.w[rl][i + 1]) - p.wi0;
                c.bins[i] = (wi < 0 || wi >= p.n_grid) ? -1 : (int)wi;
            }
            __syncthreads();
            for (int i = 1 + lane; i <= N; i += 64) {
                const int b = c.bins[i];
                if (b < 0) continue;
                const int lo = (i <= NE) ? 1 : NE + 1, hi = (i <= NE) ? NE : N;
                double *dst = hist + ((size_t)rl * 2u + (i <= NE ? 0u : 1u)) * (size_t)p.n_grid + b;
                if (p.hist_exclusive) {
                    // the first delta of a bin adds the bin's deltas in index order (= the reference's loop order)
                    bool head = true;
                    for (int j = lo; j < i; ++j) head = head && (c.bins[j] != b);
                    if (!head) continue;
                    double v = *dst;
                    for (int j = i; j <= hi; ++j)
                        if (c.bins[j] == b) v = v + c.A[rl][j + 1];
                    *dst = v;
                } else {
                    unsafeAtomicAdd(dst, c.A[rl][i + 1]);
                }
            }
            __syncthreads();
        }
        c.flush_counts();
        c.tick(13);
        sweep += 1;
    }

#ifdef EDGE_PROF
    if (chain == 0 && lane == 0 && p.n_sweeps > 1) {
        const unsigned long long *pr = reinterpret_cast<const unsigned long long *>(c.ss + SS_PROF);
        printf("prof sweeps=%d: start %llu | single %llu multi %llu amp %llu measure %llu | multi: loop %llu disp %llu m1 %llu mk(rest) %llu mk-pairs %llu mk-sort %llu decode %llu gather %llu chi2 %llu commit %llu\n",
               p.n_sweeps, pr[0], pr[10], pr[11], pr[12], pr[13], pr[1], pr[2], pr[3], pr[4], pr[14], pr[15], pr[5], pr[6], pr[7], pr[8]);
    }
#endif
    unsigned long long viol = 0;
    if (p.flags & EDGE_BIN_END) {
        const double steps = p.steps_in_bin;
        // record_bin (edge:1337-1353)
        __syncthreads();
        if (lane == 0) {
            const double c1 = c.ss[SS_CHI2_ACC] / steps;
            c.ss[SS_CHI2_ACC] = c1;
            scp->chi2_res[1] += c1;
            scp->chi2_res[2] += c1 * c1;
            for (int rl = 0; rl < 2; ++rl) {
                const double a = c.ss[SS_N0_ACC + rl] / steps, b = c.ss[SS_WN0_ACC + rl] / steps, e = c.ss[SS_EDGE_ACC + rl] / steps;
                c.ss[SS_N0_ACC + rl] = a; c.ss[SS_WN0_ACC + rl] = b; c.ss[SS_EDGE_ACC + rl] = e;
                scp->n0_res[1][rl] += a;
                scp->wn0_res[1][rl] += b;
                scp->edge_res[1][rl] += e;
                scp->n0_res[2][rl] += a * a;
                scp->wn0_res[2][rl] += b * b;
                scp->edge_res[2][rl] += e * e;
            }
        }
        // check_w (edge:658-718), counted instead of exit()
        _Pragma("unroll") for (int rl = 0; rl < 2; ++rl) if (rl < p.n_rl) {
            const double *x = c.w[rl];
            bool bad = false;
            if (x[2] < p.w_floor[rl]) bad = true;
            if (c.arr_min(x, 1, N + 1) < p.w_floor[rl] || c.arr_max(x, 1, N + 1) > p.w_m) bad = true;
            if (NE < N && rl == 0 && x[2] > c.arr_min(x, NE + 2, N + 1)) bad = true;
            if (x[3] - x[2] < p.d0) bad = true;
            double worst = 0.0;
            for (int i = 2 + lane; i <= NE - 1; i += 64) {
                const double v1 = x[i + 1] - x[i], v2 = x[i + 2] - x[i + 1];
                if (v2 < v1 && v1 - v2 > worst) worst = v1 - v2;
            }
            if (wave_max(worst) > 1e-10) bad = true;
            viol += bad ? 1 : 0;
        }
        // adjust_D (edge:1384-1432)
        __syncthreads();
        _Pragma("unroll") for (int rl = 0; rl < 2; ++rl) if (rl < p.n_rl) {
            for (int k = 1 + lane; k <= N + 1; k += 64) {
                c.arc[rl][0][k] /= steps;
                c.arc[rl][1][k] /= steps;
            }
            __syncthreads();
            for (int i = 1 + lane; i <= N; i += 64) {
                if (c.arc[rl][0][i + 1] > 0.55) c.dwn[rl][i + 1] *= 1.25;
                else if (c.arc[rl][0][i + 1] < 0.45) c.dwn[rl][i + 1] /= 1.2;
            }
            for (int i = lane; i <= NE - 2; i += 64) {
                if (c.arc[rl][1][i + 1] > 0.55)
                    c.nm[rl][i + 1] = (int)__builtin_floor((double)c.nm[rl][i + 1] * 1.25) + 1;
                else if (c.arc[rl][1][i + 1] < 0.45) {
                    const int v = (int)__builtin_floor((double)c.nm[rl][i + 1] / 1.2);
                    c.nm[rl][i + 1] = v > 1 ? v : 1;
                }
            }
            __syncthreads();
            if (NE > 3)
                for (int i = 2 + lane; i <= NE - 1; i += 64)
                    if (c.nm[rl][i] > NE - i) c.nm[rl][i] = NE - i;
            if (lane == 0) {
                c.ss[SS_AMP_AR + rl] /= steps;
                c.ss[SS_AMP_AR + 2 + rl] /= steps;
                if (c.ss[SS_AMP_AR + rl] > 0.55) { c.ss[SS_DN0 + rl] *= 1.25; c.ss[SS_DEPS0 + rl] *= 1.25; }
                else if (c.ss[SS_AMP_AR + rl] < 0.45) { c.ss[SS_DN0 + rl] /= 1.20; c.ss[SS_DEPS0 + rl] /= 1.20; }
            }
        }
        __syncthreads();
    }

    // write back
    __syncthreads();
    _Pragma("unroll") for (int rl = 0; rl < 2; ++rl) {
        const size_t o = cs + (size_t)rl * S;
        if (rl < p.n_rl) for (int k = lane; k < S; k += 64) {
            p.w[o + k] = c.w[rl][k];
            p.A[o + k] = c.A[rl][k];
            p.dw[o + k] = c.dwn[rl][k];
            p.mw[o + k] = c.mw[rl][k];
            p.nm[o + k] = c.nm[rl][k];
            p.arc[o * 2u + k] = c.arc[rl][0][k];
            p.arc[o * 2u + S + k] = c.arc[rl][1][k];
        }
    }
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        if (j < p.n_tau) gbar_g[j] = c.gbar[t];
    }
    if (lane == 0) {
        scp->chi2 = c.chi2;
        scp->chi2_min = c.ss[SS_CHI2_MIN];
        for (int rl = 0; rl < 2; ++rl) {
            scp->n0[rl] = c.ss[SS_N0 + rl]; scp->eps0[rl] = c.ss[SS_EPS0 + rl];
            scp->dn0[rl] = c.ss[SS_DN0 + rl]; scp->deps0[rl] = c.ss[SS_DEPS0 + rl];
            for (int m = 0; m < 3; ++m) scp->amp_ar[m][rl] = c.ss[SS_AMP_AR + 2 * m + rl];
            scp->edge_res[0][rl] = c.ss[SS_EDGE_ACC + rl];
            scp->n0_res[0][rl] = c.ss[SS_N0_ACC + rl];
            scp->wn0_res[0][rl] = c.ss[SS_WN0_ACC + rl];
        }
        scp->chi2_res[0] = c.ss[SS_CHI2_ACC];
        scp->sweep = sweep;
        const unsigned long long *cu = reinterpret_cast<const unsigned long long *>(c.ss + SS_CNT);
        scp->counters[0] += cu[0];
        scp->counters[1] += cu[1];
        scp->counters[2] += cu[2];
        scp->counters[4] += cu[3];
        scp->violations += viol;
        if constexpr (LOG) {
            if (c.log != nullptr) *p.log_len = c.log_pos;
        }
    }
}

// calc_Gbar!(self, 1); chi2 = calc_chi2; chi2_min = chi2  (edge:559-561, 1728-1730)
template <int NTL>
__global__ void __launch_bounds__(64) edge_init_kernel(const EdgeParams p)
{
    const int chain = (int)blockIdx.x, lane = lane_id();
    const int S = p.S, N = p.N;
    EdgeChain<NTL> c;
    c.p = &p;
    c.lane = lane;
    c.prio.init(p.prio_shift);
    c.n_cols = 0;
    const size_t cs = (size_t)chain * 2u * (size_t)S;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        c.off[t] = (uint32_t)((j < p.ntp) ? j : p.ntp - 1) * 8u;
        c.G[t] = p.G[j];
        c.sinv[t] = p.sinv[j];
        c.gbar[t] = 0.0;
    }
    for (int rl = 0; rl < p.n_rl; ++rl) {
        double *kc = p.kcurr + (cs + (size_t)rl * S) * (size_t)p.ntp;
        const double *w = p.w + cs + (size_t)rl * S, *A = p.A + cs + (size_t)rl * S;
        double acc[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[t] = 0.0;
        for (int k = 1; k <= N + 1; ++k) {
            double kn[NTL];
            c.taylor(rl, w[k], kn);
            const double Ak = A[k];
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                c.col_put(kc, k, t, kn[t]);
                acc[t] = acc[t] + kn[t] * Ak;
            }
        }
#pragma unroll
        for (int t = 0; t < NTL; ++t) c.gbar[t] = (rl == 0) ? acc[t] : c.gbar[t] + acc[t];
    }
    const double chi2 = c.chi2_of(c.gbar);
    double *gbar_g = p.gbar + (size_t)chain * SAC_MAX_N_TAU;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        gbar_g[j] = (j < p.n_tau) ? c.gbar[t] : 0.0;
    }
    if (lane == 0) {
        p.sc[chain].chi2 = chi2;
        p.sc[chain].chi2_min = chi2;
    }
}

// ---- per-stage merge over ranks (comm.h) --------------------------------------------------------------------------
// dst[i] = sum over `rows` rows of src[r * row_len + i], rows in index order (the histogram replicas)
__global__ void edge_sum_rows_kernel(const double *src, int rows, long long row_len, double *dst)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_len) return;
    double s = 0.0;
    for (int r = 0; r < rows; ++r) s = s + src[(size_t)r * (size_t)row_len + (size_t)i];
    dst[i] = s;
}

// dst[g * n1 + k - 1] = sum over chains of src[c * chain_stride + g * S + k], k = 1 .. n1 (per-delta arrays keep Julia's index)
__global__ void edge_sum_chains_kernel(const double *src, int n_chains, long long chain_stride, int groups, int S, int n1, double *dst)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= groups * n1) return;
    const int g = i / n1, k = i % n1 + 1;
    double s = 0.0;
    for (int c = 0; c < n_chains; ++c) s = s + src[(size_t)c * (size_t)chain_stride + (size_t)g * S + k];
    dst[i] = s;
}

// dst[t] = sum over chains of Gbar[c][t], chains in index order
__global__ void edge_gbar_sum_kernel(const double *gbar, int n_chains, double *dst)
{
    const int t = (int)threadIdx.x;
    if (t >= SAC_MAX_N_TAU) return;
    double s = 0.0;
    for (int c = 0; c < n_chains; ++c) s = s + gbar[(size_t)c * SAC_MAX_N_TAU + t];
    dst[t] = s;
}

// dst[g * n1 + k - 1] = sum over chains of (double) src[c * chain_stride + g * S + k]: the cluster sizes n_multi (edge:73)
__global__ void edge_sum_chains_int_kernel(const int *src, int n_chains, long long chain_stride, int groups, int S, int n1, double *dst)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= groups * n1) return;
    const int g = i / n1, k = i % n1 + 1;
    double s = 0.0;
    for (int c = 0; c < n_chains; ++c) s = s + (double)src[(size_t)c * (size_t)chain_stride + (size_t)g * S + k];
    dst[i] = s;
}

// dst[0], dst[1] = sum over chains of the lowest / highest continuum frequency of side 0 (what write_res prints, edge:1454-1460)
__global__ void edge_cont_extent_kernel(const double *w, int n_chains, int S, int NE, int N, double *dst)
{
    const int t = (int)threadIdx.x;
    if (t >= 2) return;
    double s = 0.0;
    for (int c = 0; c < n_chains; ++c) {
        const double *x = w + (size_t)c * 2 * (size_t)S;
        double m = x[NE + 2];
        for (int k = NE + 3; k <= N + 1; ++k) m = (t == 0) ? (x[k] < m ? x[k] : m) : (x[k] > m ? x[k] : m);
        s = s + m;
    }
    dst[t] = s;
}

// first passages of anneal(chi2_target) (edge:1680) per chain: marks the chains whose stage average <chi2> (chi2_res[1] / bins)
// lies below their own target; pass[c] keeps the 1-based index of the first such stage
__global__ void edge_passage_kernel(const EdgeScalars *sc, const double *targets, int *pass, int n_chains, double bins, int stage)
{
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= n_chains) return;
    if (pass[c] == 0 && sc[c].chi2_res[1] / bins < targets[c]) pass[c] = stage;
}

__global__ void edge_targets_kernel(const EdgeScalars *sc, double a, double *targets, int *pass, int n_chains)
{
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= n_chains) return;
    const double m = sc[c].chi2_min;
    targets[c] = m + a * sqrt(2 * m);                               // edge:1718
    pass[c] = 0;
}

constexpr int EDGE_SLOTS = 46;  // chi2_res[3], edge_res[6], n0_res[6], wn0_res[6], amp_ar[6], n0[2], eps0[2], dn0[2], deps0[2], chi2,
                                // sum chi2_min, proposals, evaluated, accepted, lookups, violations, chains, sums of the lowest
                                // and of the highest continuum frequency (43, 44), chains below their target (45)

// one thread per slot: bin_averages (edge:1436-1449) of every chain, summed over the chains in index order
__global__ void edge_pack_kernel(const EdgeScalars *sc, const int *pass, int n_chains, double bins, int n_rl, double *out_sum, double *out_min)
{
    const int t = (int)threadIdx.x;
    if (t >= EDGE_SLOTS) return;
    const double den = sqrt(bins - 1.0);
    double s = 0.0, mn = sc[0].chi2_min;
    for (int c = 0; c < n_chains; ++c) {
        const EdgeScalars &x = sc[c];
        double v = 0.0;
        if (t < 3) {
            v = x.chi2_res[t] / bins;
            if (t == 2) { const double m = x.chi2_res[1] / bins; v = sqrt(fabs(v - m * m)) / den; }
        } else if (t < 21) {
            const int a = (t - 3) / 6, q = ((t - 3) % 6) / 2, rl = (t - 3) % 2;
            const double (*res)[2] = a == 0 ? x.edge_res : a == 1 ? x.n0_res : x.wn0_res;
            v = res[q][rl] / bins;
            if (q == 2 && rl < n_rl) { const double m = res[1][rl] / bins; v = sqrt(fabs(v - m * m)) / den; }
        } else if (t < 27) v = x.amp_ar[(t - 21) / 2][(t - 21) % 2];
        else if (t < 29) v = x.n0[t - 27];
        else if (t < 31) v = x.eps0[t - 29];
        else if (t < 33) v = x.dn0[t - 31];
        else if (t < 35) v = x.deps0[t - 33];
        else if (t == 35) v = x.chi2;
        else if (t == 36) { v = x.chi2_min; if (x.chi2_min < mn) mn = x.chi2_min; }
        else if (t == 37) v = (double)x.counters[0];
        else if (t == 38) v = (double)x.counters[1];
        else if (t == 39) v = (double)x.counters[2];
        else if (t == 40) v = (double)x.counters[4];
        else if (t == 41) v = (double)x.violations;
        else if (t == 42) v = 1.0;
        else if (t == 45) v = (pass != nullptr && pass[c] != 0) ? 1.0 : 0.0;     // chains below their own target so far
        s = s + v;
    }
    out_sum[t] = s;
    if (t == 36) out_min[0] = mn;
}

// =========================================================================================
// Host side
// =========================================================================================

struct sac_edge {
    sac_edge_config_t cfg{};
    int ntp = 0, S = 0, hist_copies = 1;
    bool hist_exclusive = false;
    int sweeps_per_launch = 64;
    double *dTab[2] = {nullptr, nullptr};
    double *dG = nullptr, *dSinv = nullptr;
    double *dW = nullptr, *dA = nullptr, *dDw = nullptr, *dMw = nullptr, *dArc = nullptr;
    int *dNm = nullptr;
    double *dKcurr = nullptr, *dKnew = nullptr, *dGbar = nullptr, *dWavg = nullptr, *dAavg = nullptr, *dHist = nullptr;
    EdgeScalars *dSc = nullptr;
    MoveRec *dLog = nullptr;
    unsigned long long *dLogLen = nullptr;
    long long log_cap = 0;
    int log_chain = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool problem = false, state_set = false, gbar_set = false;
    bool table_built[2] = {false, false};   // sac_edge_build_tables
    int64_t last_steps = 0, last_bins = 0;
    double last_stage_ms = 0.0;
    double ms_per_sweep = 0.0;     // measured stage time per sweep (priority rotation period)
    saccomm::Comm *comm = nullptr; // sac_edge_comm_init: per-stage merge over ranks
    double *dGbarSum = nullptr;    // [128] sac_edge_get_gbar_mean without a communicator
    double *dMeans = nullptr;      // [4 n1 + 2] sac_edge_get_move_means without a communicator
    double *dTargets = nullptr;    // [n_chains] per-chain chi2 targets of the second anneal (sac_edge_set_*targets*)
    int *dPass = nullptr;          // [n_chains] 1-based stage index of each chain's first passage below its target, 0 = none
    bool targets_on = false;
    int stages_since_targets = 0;
    // layout of the fused merge buffer: [hist 4 n_grid | w_avg 2 n1 | A_avg 2 n1 | accept rates 4 n1 | sum of Gbar 128 |
    //                                    windows 2 n1 | cluster sizes 2 n1 | EDGE_SLOTS]
    size_t off_wavg() const { return 4 * (size_t)cfg.n_grid; }
    size_t off_aavg() const { return off_wavg() + 2 * (size_t)(cfg.n_omega + 1); }
    size_t off_arc() const { return off_aavg() + 2 * (size_t)(cfg.n_omega + 1); }
    size_t off_gbar() const { return off_arc() + 4 * (size_t)(cfg.n_omega + 1); }
    size_t off_dw() const { return off_gbar() + SAC_MAX_N_TAU; }
    size_t off_nm() const { return off_dw() + 2 * (size_t)(cfg.n_omega + 1); }
    size_t n_payload() const { return off_nm() + 2 * (size_t)(cfg.n_omega + 1); }
};

static int use_device(sac_edge *h)
{
    HIP_TRY(hipSetDevice(h->cfg.device));
    return SAC_OK;
}

static void fill_params(const sac_edge *h, EdgeParams &p)
{
    std::memset(&p, 0, sizeof(p));
    const sac_edge_config_t &c = h->cfg;
    p.tab[0] = h->dTab[0]; p.tab[1] = h->dTab[1];
    p.G = h->dG; p.sinv = h->dSinv;
    p.w = h->dW; p.A = h->dA; p.dw = h->dDw; p.mw = h->dMw; p.nm = h->dNm; p.arc = h->dArc;
    p.kcurr = h->dKcurr; p.knew = h->dKnew; p.gbar = h->dGbar; p.wavg = h->dWavg; p.aavg = h->dAavg;
    p.hist = h->dHist; p.sc = h->dSc;
    p.log = h->dLog; p.log_len = h->dLogLen; p.log_cap = h->log_cap; p.log_chain = h->log_chain;
    p.n_chains = c.n_chains; p.n_tau = c.n_tau; p.ntp = h->ntp; p.NE = c.n_e; p.N = c.n_omega; p.S = h->S;
    p.n_rl = c.n_rl; p.double_in = c.double_in; p.fix_edge = c.fix_edge;
    p.hist_copies = h->hist_copies; p.hist_exclusive = h->hist_exclusive ? 1 : 0;
    p.n_grid = c.n_grid; p.wi0 = c.wi_0;
    p.A_c = c.A_c; p.A_r = c.A_r; p.d0 = c.d0; p.w_floor[0] = c.w_floor[0]; p.w_floor[1] = c.w_floor[1];
    p.w_m = c.w_m; p.dwg = c.d_omega; p.inv_dwg = 1.0 / c.d_omega; p.c[0] = c.c[0]; p.c[1] = c.c[1];
    p.seed_lo = (uint32_t)c.seed; p.seed_hi = (uint32_t)(c.seed >> 32); p.chain_offset = c.chain_offset;
}

extern "C" int sac_edge_create(const sac_edge_config_t *cfg, sac_edge_t **out)
{
    if (!cfg || !out) return efail(SAC_E_INVALID, "sac_edge_create: null argument");
    *out = nullptr;
    if (cfg->n_tau < 1 || cfg->n_tau > SAC_MAX_N_TAU)
        return efail(SAC_E_INVALID, "sac_edge_create: n_tau=%d outside 1..%d", cfg->n_tau, SAC_MAX_N_TAU);
    if (cfg->n_e < 4 || cfg->n_e > SAC_MAX_N_EDGE)
        return efail(SAC_E_INVALID, "sac_edge_create: n_e=%d outside 4..%d", cfg->n_e, SAC_MAX_N_EDGE);
    if (cfg->n_omega < cfg->n_e || cfg->n_omega > 2 * SAC_MAX_N_EDGE)
        return efail(SAC_E_INVALID, "sac_edge_create: n_omega=%d outside n_e..%d", cfg->n_omega, 2 * SAC_MAX_N_EDGE);
    if (cfg->n_rl < 1 || cfg->n_rl > 2) return efail(SAC_E_INVALID, "sac_edge_create: n_rl=%d", cfg->n_rl);
    if (cfg->double_in && cfg->n_rl != 2) return efail(SAC_E_INVALID, "sac_edge_create: double_in needs n_rl = 2");
    if (cfg->n_chains < 1) return efail(SAC_E_INVALID, "sac_edge_create: n_chains=%d", cfg->n_chains);
    if (cfg->n_grid < 2 || cfg->n_grid > (1ll << 30))
        return efail(SAC_E_INVALID, "sac_edge_create: n_grid=%lld outside 2..2^30", (long long)cfg->n_grid);
    if (!(cfg->d_omega > 0)) return efail(SAC_E_INVALID, "sac_edge_create: d_omega must be positive");
    if (cfg->A_c < 0 || cfg->A_c >= 1) return efail(SAC_E_INVALID, "sac_edge_create: A_c=%g outside [0,1)", cfg->A_c);
    if (cfg->A_c > 0 && cfg->n_omega == cfg->n_e)
        return efail(SAC_E_INVALID, "sac_edge_create: A_c > 0 needs continuum deltas (n_omega > n_e)");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return efail(SAC_E_INVALID, "sac_edge_create: device %d of %d", cfg->device, ndev);
    sac_edge *h = new (std::nothrow) sac_edge();
    if (!h) return efail(SAC_E_NOMEM, "sac_edge_create: out of host memory");
    h->cfg = *cfg;
    h->ntp = (cfg->n_tau + 7) / 8 * 8;
    h->S = (cfg->n_omega + 3 + 7) / 8 * 8;
    if (cfg->sweeps_per_launch > 0) h->sweeps_per_launch = cfg->sweeps_per_launch;
    h->hist_copies = cfg->hist_copies > 0 ? cfg->hist_copies : 1;
    if (h->hist_copies > cfg->n_chains) h->hist_copies = cfg->n_chains;
    h->hist_exclusive = (h->hist_copies == cfg->n_chains);
    int rc = use_device(h);
    if (rc) { delete h; return rc; }
    const size_t nc = (size_t)cfg->n_chains, S = (size_t)h->S;
#define EALLOC(ptr, bytes)                                                           \
    do {                                                                             \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes));                         \
        if (e_ == hipSuccess) e_ = hipMemset((ptr), 0, (bytes));                     \
        if (e_ != hipSuccess) {                                                      \
            sac_edge_destroy(h);                                                     \
            return efail(SAC_E_HIP, "sac_edge_create: allocating %zu bytes: %s", (size_t)(bytes), hipGetErrorString(e_)); \
        }                                                                            \
    } while (0)
    EALLOC(h->dG, SAC_MAX_N_TAU * sizeof(double));
    EALLOC(h->dSinv, SAC_MAX_N_TAU * sizeof(double));
    EALLOC(h->dW, nc * 2 * S * sizeof(double));
    EALLOC(h->dA, nc * 2 * S * sizeof(double));
    EALLOC(h->dDw, nc * 2 * S * sizeof(double));
    EALLOC(h->dMw, nc * 2 * S * sizeof(double));
    EALLOC(h->dNm, nc * 2 * S * sizeof(int));
    EALLOC(h->dArc, nc * 4 * S * sizeof(double));
    EALLOC(h->dKcurr, nc * 2 * S * (size_t)h->ntp * sizeof(double));
    EALLOC(h->dKnew, nc * 2 * S * (size_t)h->ntp * sizeof(double));
    EALLOC(h->dGbar, nc * SAC_MAX_N_TAU * sizeof(double));
    EALLOC(h->dWavg, nc * 2 * S * sizeof(double));
    EALLOC(h->dAavg, nc * 2 * S * sizeof(double));
    EALLOC(h->dHist, (size_t)h->hist_copies * 4 * (size_t)cfg->n_grid * sizeof(double));
    EALLOC(h->dSc, nc * sizeof(EdgeScalars));
    EALLOC(h->dLogLen, sizeof(unsigned long long));
    EALLOC(h->dGbarSum, SAC_MAX_N_TAU * sizeof(double));
    EALLOC(h->dMeans, (4 * ((size_t)cfg->n_omega + 1) + 2) * sizeof(double));
    EALLOC(h->dTargets, nc * sizeof(double));
    EALLOC(h->dPass, nc * sizeof(int));
#undef EALLOC
    if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess ||
        hipEventCreate(&h->ev1) != hipSuccess) {
        sac_edge_destroy(h);
        return efail(SAC_E_HIP, "sac_edge_create: stream/event creation failed");
    }
    *out = h;
    return SAC_OK;
}

extern "C" int sac_edge_destroy(sac_edge_t *h)
{
    if (!h) return SAC_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *ptrs[] = {h->dTab[0], h->dTab[1], h->dG, h->dSinv, h->dW, h->dA, h->dDw, h->dMw, h->dNm, h->dArc, h->dKcurr,
                    h->dKnew, h->dGbar, h->dWavg, h->dAavg, h->dHist, h->dSc, h->dLog, h->dLogLen, h->dGbarSum, h->dMeans, h->dTargets, h->dPass};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    saccomm::destroy(h->comm);
    h->comm = nullptr;
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return SAC_OK;
}

extern "C" int sac_edge_upload_problem(sac_edge_t *h, const double *const *K, const double *const *dK,
                                       const double *const *d2K, const double *G, const double *sigma_inv)
{
    if (!h || !K || !dK || !d2K || !G || !sigma_inv) return efail(SAC_E_INVALID, "sac_edge_upload_problem: null argument");
    int rc;
    if ((rc = use_device(h))) return rc;
    const size_t ng = (size_t)h->cfg.n_grid, nt = (size_t)h->cfg.n_tau, ntp = (size_t)h->ntp;
    const size_t CH = 16384;                              // columns per staging chunk
    std::vector<double> stage(CH * 3 * ntp);
    for (int rl = 0; rl < h->cfg.n_rl; ++rl) {
        if (!K[rl] || !dK[rl] || !d2K[rl]) return efail(SAC_E_INVALID, "sac_edge_upload_problem: null table for side %d", rl);
        if (!h->dTab[rl]) HIP_TRY(hipMalloc((void **)&h->dTab[rl], ng * 3 * ntp * sizeof(double)));
        for (size_t a = 0; a < ng; a += CH) {
            const size_t n = (ng - a < CH) ? ng - a : CH;
            std::fill(stage.begin(), stage.begin() + n * 3 * ntp, 0.0);
            const double *src[3] = {K[rl], dK[rl], d2K[rl]};
            for (size_t i = 0; i < n; ++i)
                for (int q = 0; q < 3; ++q)
                    std::memcpy(&stage[(i * 3 + q) * ntp], src[q] + (a + i) * nt, nt * sizeof(double));
            HIP_TRY(copy_sync(h->stream, h->dTab[rl] + a * 3 * ntp, stage.data(), n * 3 * ntp * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    double pad[SAC_MAX_N_TAU];
    std::memset(pad, 0, sizeof(pad));
    std::memcpy(pad, G, nt * sizeof(double));
    HIP_TRY(copy_sync(h->stream, h->dG, pad, sizeof(pad), hipMemcpyHostToDevice));
    std::memset(pad, 0, sizeof(pad));
    std::memcpy(pad, sigma_inv, nt * sizeof(double));
    HIP_TRY(copy_sync(h->stream, h->dSinv, pad, sizeof(pad), hipMemcpyHostToDevice));
    h->problem = true;
    return SAC_OK;
}

static int upload_data(sac_edge *h, const double *G, const double *sigma_inv)
{
    const size_t nt = (size_t)h->cfg.n_tau;
    double pad[SAC_MAX_N_TAU];
    std::memset(pad, 0, sizeof(pad));
    std::memcpy(pad, G, nt * sizeof(double));
    HIP_TRY(copy_sync(h->stream, h->dG, pad, sizeof(pad), hipMemcpyHostToDevice));
    std::memset(pad, 0, sizeof(pad));
    std::memcpy(pad, sigma_inv, nt * sizeof(double));
    HIP_TRY(copy_sync(h->stream, h->dSinv, pad, sizeof(pad), hipMemcpyHostToDevice));
    return SAC_OK;
}

extern "C" int sac_edge_upload_data(sac_edge_t *h, const double *G, const double *sigma_inv)
{
    if (!h || !G || !sigma_inv) return efail(SAC_E_INVALID, "sac_edge_upload_data: null argument");
    for (int rl = 0; rl < h->cfg.n_rl; ++rl)
        if (!h->table_built[rl]) return efail(SAC_E_STATE, "sac_edge_upload_data: build the tables of side %d first", rl);
    int rc;
    if ((rc = use_device(h))) return rc;
    if ((rc = upload_data(h, G, sigma_inv))) return rc;
    h->problem = true;
    return SAC_OK;
}

extern "C" int sac_edge_build_tables(sac_edge_t *h, int32_t side, const sac_table_spec_t *spec, const double *tau, const double *U)
{
    if (!h || !spec || !tau || !U) return efail(SAC_E_INVALID, "sac_edge_build_tables: null argument");
    if (side < 0 || side >= h->cfg.n_rl) return efail(SAC_E_INVALID, "sac_edge_build_tables: side %d outside 0..%d", side, h->cfg.n_rl - 1);
    if (spec->kernel_type < 0 || spec->kernel_type > 2) return efail(SAC_E_INVALID, "sac_edge_build_tables: Invalid Kernel type.");   // edge:191
    if (!(spec->d_omega > 0) || !std::isfinite(spec->beta)) return efail(SAC_E_INVALID, "sac_edge_build_tables: d_omega=%g beta=%g", spec->d_omega, spec->beta);
    if (spec->i_first != h->cfg.wi_0) return efail(SAC_E_INVALID, "sac_edge_build_tables: i_first=%lld, the handle's wi_0 is %lld", (long long)spec->i_first, (long long)h->cfg.wi_0);
    int rc;
    if ((rc = use_device(h))) return rc;
    const size_t ng = (size_t)h->cfg.n_grid, nt = (size_t)h->cfg.n_tau, ntp = (size_t)h->ntp;
    if (!h->dTab[side]) HIP_TRY(hipMalloc((void **)&h->dTab[side], ng * 3 * ntp * sizeof(double)));
    double *d_tau = nullptr, *d_U = nullptr;
    HIP_TRY(hipMalloc(&d_tau, nt * sizeof(double)));
    if (hipMalloc(&d_U, nt * nt * sizeof(double)) != hipSuccess) { (void)hipFree(d_tau); return efail(SAC_E_HIP, "sac_edge_build_tables: out of device memory"); }
    hipError_t e = copy_sync(h->stream, d_tau, tau, nt * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = copy_sync(h->stream, d_U, U, nt * nt * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        TableSpec sp;
        sp.kernel_type = spec->kernel_type; sp.n_tau = (int)nt; sp.mirror = spec->mirror; sp.negate = spec->negate;
        sp.i_first = spec->i_first; sp.offset = spec->offset; sp.d_omega = spec->d_omega; sp.beta = spec->beta;
        sp.n_grid = (long long)ng; sp.ld = 3 * (int)ntp; sp.n_cols = (int)ntp;
        e = launch_table_build(h->stream, sp, d_tau, d_U, h->dTab[side]);
        if (e == hipSuccess) {
            const double dw = spec->d_omega;
            const long long idx_m = h->cfg.wi_0 + (long long)ng - 1;          // wi_m: the reference's index wi_m + 1, 0-based
            const long long total = (long long)ng * (long long)ntp;
            hipLaunchKernelGGL(edge_stencil_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream,
                               h->dTab[side], (long long)ng, (int)ntp, idx_m, dw, 2 * dw, 2 * (dw * dw));
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    (void)hipFree(d_tau);
    (void)hipFree(d_U);
    if (e != hipSuccess) return efail(SAC_E_HIP, "sac_edge_build_tables: %s", hipGetErrorString(e));
    h->table_built[side] = true;
    h->gbar_set = false;
    return SAC_OK;
}

extern "C" int sac_edge_get_tables(sac_edge_t *h, int32_t side, double *K, double *dK, double *d2K)
{
    if (!h) return efail(SAC_E_INVALID, "sac_edge_get_tables: null handle");
    if (side < 0 || side >= h->cfg.n_rl || !h->dTab[side]) return efail(SAC_E_INVALID, "sac_edge_get_tables: side %d has no table", side);
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t ng = (size_t)h->cfg.n_grid, nt = (size_t)h->cfg.n_tau, ntp = (size_t)h->ntp;
    double *dst[3] = {K, dK, d2K};
    for (int q = 0; q < 3; ++q)
        if (dst[q])
            HIP_TRY(hipMemcpy2D(dst[q], nt * sizeof(double), h->dTab[side] + (size_t)q * ntp, 3 * ntp * sizeof(double),
                                nt * sizeof(double), ng, hipMemcpyDeviceToHost));
    return SAC_OK;
}

// arrays in the reference's shape [N_w+1, 2] (column-major: side after side)
extern "C" int sac_edge_set_state(sac_edge_t *h, int32_t chain, const double *w, const double *A, const double *dw,
                                  const int64_t *n_multi, const double *scal8)
{
    if (!h) return efail(SAC_E_INVALID, "sac_edge_set_state: null handle");
    const int nc = h->cfg.n_chains, n1 = h->cfg.n_omega + 1, NE = h->cfg.n_e, S = h->S;
    if (chain < -1 || chain >= nc) return efail(SAC_E_INVALID, "sac_edge_set_state: chain %d outside -1..%d", chain, nc - 1);
    if (n_multi)
        for (int rl = 0; rl < h->cfg.n_rl; ++rl)
            for (int k = 2; k <= NE - 1; ++k) {
                const int64_t v = n_multi[rl * n1 + k - 1];
                if (v < 0 || v > NE - k)
                    return efail(SAC_E_INVALID, "sac_edge_set_state: n_multi[%d, side %d] = %lld outside 0..%d", k, rl + 1,
                                 (long long)v, NE - k);
            }
    if (w)
        for (int i = 0; i < 2 * n1; ++i)
            if (!std::isfinite(w[i])) return efail(SAC_E_INVALID, "sac_edge_set_state: non-finite frequency");
    int rc;
    if ((rc = use_device(h))) return rc;
    if (h->comm) h->comm->merged = false;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int first = chain < 0 ? 0 : chain, count = chain < 0 ? nc : 1;
    auto put_d = [&](double *dst, const double *src) -> int {
        std::vector<double> one(2 * (size_t)S, 0.0);
        for (int rl = 0; rl < 2; ++rl)
            for (int k = 1; k <= n1; ++k) one[(size_t)rl * S + k] = src[rl * n1 + k - 1];
        std::vector<double> all((size_t)count * one.size());
        for (int c = 0; c < count; ++c) std::memcpy(all.data() + (size_t)c * one.size(), one.data(), one.size() * sizeof(double));
        HIP_TRY(copy_sync(h->stream, dst + (size_t)first * one.size(), all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice));
        return SAC_OK;
    };
    if (w && (rc = put_d(h->dW, w))) return rc;
    if (A && (rc = put_d(h->dA, A))) return rc;
    if (dw && (rc = put_d(h->dDw, dw))) return rc;
    if (n_multi) {
        std::vector<int> one(2 * (size_t)S, 0);
        for (int rl = 0; rl < 2; ++rl)
            for (int k = 1; k <= n1; ++k) one[(size_t)rl * S + k] = (int)n_multi[rl * n1 + k - 1];
        std::vector<int> all((size_t)count * one.size());
        for (int c = 0; c < count; ++c) std::memcpy(all.data() + (size_t)c * one.size(), one.data(), one.size() * sizeof(int));
        HIP_TRY(copy_sync(h->stream, h->dNm + (size_t)first * one.size(), all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    if (scal8) {
        std::vector<EdgeScalars> sc((size_t)count);
        HIP_TRY(copy_sync(h->stream, sc.data(), h->dSc + first, sc.size() * sizeof(EdgeScalars), hipMemcpyDeviceToHost));
        for (auto &s : sc)
            for (int rl = 0; rl < 2; ++rl) {
                s.n0[rl] = scal8[rl]; s.eps0[rl] = scal8[2 + rl]; s.dn0[rl] = scal8[4 + rl]; s.deps0[rl] = scal8[6 + rl];
            }
        HIP_TRY(copy_sync(h->stream, h->dSc + first, sc.data(), sc.size() * sizeof(EdgeScalars), hipMemcpyHostToDevice));
    }
    if (w && A) h->state_set = true;
    if (w || A) h->gbar_set = false;
    return SAC_OK;
}

extern "C" int sac_edge_get_state(sac_edge_t *h, int32_t chain, double *w, double *A, double *dw, int64_t *n_multi,
                                  double *scal8)
{
    if (!h) return efail(SAC_E_INVALID, "sac_edge_get_state: null handle");
    const int nc = h->cfg.n_chains, n1 = h->cfg.n_omega + 1, S = h->S;
    if (chain < 0 || chain >= nc) return efail(SAC_E_INVALID, "sac_edge_get_state: chain %d outside 0..%d", chain, nc - 1);
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<double> one(2 * (size_t)S);
    auto get_d = [&](const double *src, double *dst) -> int {
        HIP_TRY(copy_sync(h->stream, one.data(), src + (size_t)chain * one.size(), one.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int rl = 0; rl < 2; ++rl)
            for (int k = 1; k <= n1; ++k) dst[rl * n1 + k - 1] = one[(size_t)rl * S + k];
        return SAC_OK;
    };
    if (w && (rc = get_d(h->dW, w))) return rc;
    if (A && (rc = get_d(h->dA, A))) return rc;
    if (dw && (rc = get_d(h->dDw, dw))) return rc;
    if (n_multi) {
        std::vector<int> oi(2 * (size_t)S);
        HIP_TRY(copy_sync(h->stream, oi.data(), h->dNm + (size_t)chain * oi.size(), oi.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int rl = 0; rl < 2; ++rl)
            for (int k = 1; k <= n1; ++k) n_multi[rl * n1 + k - 1] = oi[(size_t)rl * S + k];
    }
    if (scal8) {
        EdgeScalars s;
        HIP_TRY(copy_sync(h->stream, &s, h->dSc + chain, sizeof(s), hipMemcpyDeviceToHost));
        for (int rl = 0; rl < 2; ++rl) {
            scal8[rl] = s.n0[rl]; scal8[2 + rl] = s.eps0[rl]; scal8[4 + rl] = s.dn0[rl]; scal8[6 + rl] = s.deps0[rl];
        }
    }
    return SAC_OK;
}

extern "C" int sac_edge_init_gbar(sac_edge_t *h)
{
    if (!h) return efail(SAC_E_INVALID, "sac_edge_init_gbar: null handle");
    if (!h->problem || !h->state_set) return efail(SAC_E_STATE, "sac_edge_init_gbar: upload the problem and set the state first");
    int rc;
    if ((rc = use_device(h))) return rc;
    if (h->comm) h->comm->merged = false;
    EdgeParams p;
    fill_params(h, p);
    const dim3 grid((unsigned)h->cfg.n_chains), block(64);
    if (h->cfg.n_tau <= 64) hipLaunchKernelGGL(edge_init_kernel<1>, grid, block, 0, h->stream, p);
    else hipLaunchKernelGGL(edge_init_kernel<2>, grid, block, 0, h->stream, p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->gbar_set = true;
    return SAC_OK;
}

static int launch_stage(sac_edge *h, EdgeParams &p)
{
    const bool log = (h->log_chain >= 0 && h->dLog != nullptr);
    const dim3 grid((unsigned)h->cfg.n_chains), block(64);
    const size_t lds = edge_lds_bytes(h->S, h->cfg.n_rl);
    const int variant = (h->cfg.n_tau <= 64 ? 0 : 2) | (log ? 1 : 0);
#define SAC_ELAUNCH(NTL_, LOG_)                                                                          \
    do {                                                                                                 \
        if (lds > 48 * 1024)                                                                             \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&edge_stage_kernel<NTL_, LOG_>),  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
        hipLaunchKernelGGL((edge_stage_kernel<NTL_, LOG_>), grid, block, lds, h->stream, p);             \
    } while (0)
    switch (variant) {
    case 0: SAC_ELAUNCH(1, false); break;
    case 1: SAC_ELAUNCH(1, true); break;
    case 2: SAC_ELAUNCH(2, false); break;
    default: SAC_ELAUNCH(2, true); break;
    }
#undef SAC_ELAUNCH
    HIP_TRY(hipGetLastError());
    return SAC_OK;
}

// The per-stage merge over ranks (include/sac_hip.h "multi-GPU merge"): histogram replicas, per-delta averages and accept
// rates summed over this rank's chains into the payload, the bin averages packed behind them, ONE all-reduce(sum) + ONE
// all-reduce(min), merged scalars read back.
static int edge_comm_merge(sac_edge *h)
{
    if (!h->comm) return SAC_OK;
    saccomm::Comm *c = h->comm;
    const int nc = h->cfg.n_chains, n1 = h->cfg.n_omega + 1, S = h->S;
    const long long nh = 4ll * (long long)h->cfg.n_grid;
    hipLaunchKernelGGL(edge_sum_rows_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, h->stream,
                       (const double *)h->dHist, h->hist_copies, nh, c->dSum);
    hipLaunchKernelGGL(edge_sum_chains_kernel, dim3((unsigned)((2 * n1 + 255) / 256)), dim3(256), 0, h->stream,
                       (const double *)h->dWavg, nc, 2ll * S, 2, S, n1, c->dSum + h->off_wavg());
    hipLaunchKernelGGL(edge_sum_chains_kernel, dim3((unsigned)((2 * n1 + 255) / 256)), dim3(256), 0, h->stream,
                       (const double *)h->dAavg, nc, 2ll * S, 2, S, n1, c->dSum + h->off_aavg());
    hipLaunchKernelGGL(edge_sum_chains_kernel, dim3((unsigned)((4 * n1 + 255) / 256)), dim3(256), 0, h->stream,
                       (const double *)h->dArc, nc, 4ll * S, 4, S, n1, c->dSum + h->off_arc());
    hipLaunchKernelGGL(edge_gbar_sum_kernel, dim3(1), dim3(SAC_MAX_N_TAU), 0, h->stream, (const double *)h->dGbar, nc,
                       c->dSum + h->off_gbar());
    hipLaunchKernelGGL(edge_sum_chains_kernel, dim3((unsigned)((2 * n1 + 255) / 256)), dim3(256), 0, h->stream,
                       (const double *)h->dDw, nc, 2ll * S, 2, S, n1, c->dSum + h->off_dw());
    hipLaunchKernelGGL(edge_sum_chains_int_kernel, dim3((unsigned)((2 * n1 + 255) / 256)), dim3(256), 0, h->stream,
                       (const int *)h->dNm, nc, 2ll * S, 2, S, n1, c->dSum + h->off_nm());
    hipLaunchKernelGGL(edge_pack_kernel, dim3(1), dim3(64), 0, h->stream, (const EdgeScalars *)h->dSc,
                       (const int *)(h->targets_on ? h->dPass : nullptr), nc, (double)h->last_bins, h->cfg.n_rl,
                       c->dSum + h->n_payload(), c->dMin);
    if (h->cfg.n_omega > h->cfg.n_e)      // after the pack kernel in stream order: it writes 0 into slots 43, 44
        hipLaunchKernelGGL(edge_cont_extent_kernel, dim3(1), dim3(64), 0, h->stream, (const double *)h->dW, nc, S, h->cfg.n_e,
                           h->cfg.n_omega, c->dSum + h->n_payload() + 43);
    HIP_TRY(hipGetLastError());
    char msg[256];
    if (saccomm::merge(c, h->n_payload(), true, h->stream, msg, sizeof(msg))) return efail(SAC_E_HIP, "merge over ranks: %s", msg);
    return SAC_OK;
}

// run_bins(self, steps, bins, theta)  edge:1357-1379 (bin_averages is applied by sac_edge_get_results)
extern "C" int sac_edge_run_bins(sac_edge_t *h, int64_t steps, int64_t bins, double theta)
{
    if (!h) return efail(SAC_E_INVALID, "sac_edge_run_bins: null handle");
    if (!h->problem || !h->state_set || !h->gbar_set)
        return efail(SAC_E_STATE, "sac_edge_run_bins: problem, state and Gbar must be set first");
    if (steps < 1 || bins < 2) return efail(SAC_E_INVALID, "sac_edge_run_bins: steps=%lld bins=%lld (>= 1, >= 2)", (long long)steps, (long long)bins);
    if (!(theta > 0) || !std::isfinite(theta)) return efail(SAC_E_INVALID, "sac_edge_run_bins: theta=%g", theta);
    int rc;
    if ((rc = use_device(h))) return rc;
    if (h->comm) h->comm->merged = false;
    const size_t nc = (size_t)h->cfg.n_chains, S = (size_t)h->S;
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    // edge:1359-1365: zero the results, spectra and averages
    {
        HIP_TRY(hipStreamSynchronize(h->stream));
        std::vector<EdgeScalars> sc(nc);
        HIP_TRY(copy_sync(h->stream, sc.data(), h->dSc, nc * sizeof(EdgeScalars), hipMemcpyDeviceToHost));
        for (auto &s : sc) {
            std::memset(s.chi2_res, 0, sizeof(s.chi2_res));
            std::memset(s.edge_res, 0, sizeof(s.edge_res));
            std::memset(s.n0_res, 0, sizeof(s.n0_res));
            std::memset(s.wn0_res, 0, sizeof(s.wn0_res));
        }
        HIP_TRY(copy_sync(h->stream, h->dSc, sc.data(), nc * sizeof(EdgeScalars), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemsetAsync(h->dHist, 0, (size_t)h->hist_copies * 4 * (size_t)h->cfg.n_grid * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(h->dWavg, 0, nc * 2 * S * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(h->dAavg, 0, nc * 2 * S * sizeof(double), h->stream));
    EdgeParams p;
    fill_params(h, p);
    p.inv2theta = 1.0 / (2.0 * theta);
    p.steps_in_bin = (double)steps;
    for (int64_t b = 0; b < bins; ++b) {
        int64_t done = 0;
        while (done < steps) {
            int64_t chunk = steps - done;
            if (chunk > h->sweeps_per_launch) chunk = h->sweeps_per_launch;
            p.n_sweeps = (int)chunk;
            p.flags = (done == 0 ? EDGE_BIN_START : 0) | (done + chunk >= steps ? EDGE_BIN_END : 0);
            p.prio_shift = prio_shift_for(h->ms_per_sweep * (double)chunk);
            if ((rc = launch_stage(h, p))) return rc;
            done += chunk;
        }
    }
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->last_stage_ms = ms;
    if (steps * bins > 0 && ms > 0.f) h->ms_per_sweep = (double)ms / (double)(steps * bins);
    h->last_steps = steps;
    h->last_bins = bins;
    if (h->targets_on) {                                     // first passages below the chains' own targets (edge:1680)
        h->stages_since_targets += 1;
        hipLaunchKernelGGL(edge_passage_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, h->stream,
                           (const EdgeScalars *)h->dSc, (const double *)h->dTargets, h->dPass, (int)nc, (double)bins,
                           h->stages_since_targets);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return edge_comm_merge(h);
}

// bin_averages (edge:1436-1449) of one chain's accumulators
static void bin_averages(const EdgeScalars &s, int64_t bins, int n_rl, sac_edge_results_t &r)
{
    const double b = (double)bins, den = std::sqrt((double)(bins - 1));
    for (int q = 0; q < 3; ++q) {
        r.chi2_res[q] = s.chi2_res[q] / b;
        for (int rl = 0; rl < 2; ++rl) {
            r.edge_res[q][rl] = s.edge_res[q][rl] / b;
            r.n0_res[q][rl] = s.n0_res[q][rl] / b;
            r.wn0_res[q][rl] = s.wn0_res[q][rl] / b;
        }
    }
    r.chi2_res[2] = std::sqrt(std::fabs(r.chi2_res[2] - r.chi2_res[1] * r.chi2_res[1])) / den;
    for (int rl = 0; rl < n_rl; ++rl) {
        r.n0_res[2][rl] = std::sqrt(std::fabs(r.n0_res[2][rl] - r.n0_res[1][rl] * r.n0_res[1][rl])) / den;
        r.wn0_res[2][rl] = std::sqrt(std::fabs(r.wn0_res[2][rl] - r.wn0_res[1][rl] * r.wn0_res[1][rl])) / den;
        r.edge_res[2][rl] = std::sqrt(std::fabs(r.edge_res[2][rl] - r.edge_res[1][rl] * r.edge_res[1][rl])) / den;
    }
}

// chain = -1: mean over chains of every entry (chi2_min: minimum; counters, violations: sums)
extern "C" int sac_edge_get_results(sac_edge_t *h, int32_t chain, sac_edge_results_t *out)
{
    if (!h || !out) return efail(SAC_E_INVALID, "sac_edge_get_results: null argument");
    const int nc = h->cfg.n_chains;
    if (chain < -1 || chain >= nc) return efail(SAC_E_INVALID, "sac_edge_get_results: chain %d outside -1..%d", chain, nc - 1);
    if (h->last_bins < 2) return efail(SAC_E_STATE, "sac_edge_get_results: no run_bins stage has run");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (chain < 0 && h->comm && h->comm->merged) {                   // the values over all ranks (edge_comm_merge)
        const double *t = h->comm->hTail;
        const double n = t[42];
        std::memset(out, 0, sizeof(*out));
        for (int q = 0; q < 3; ++q) {
            out->chi2_res[q] = t[q] / n;
            for (int rl = 0; rl < 2; ++rl) {
                out->edge_res[q][rl] = t[3 + q * 2 + rl] / n;
                out->n0_res[q][rl] = t[9 + q * 2 + rl] / n;
                out->wn0_res[q][rl] = t[15 + q * 2 + rl] / n;
                out->amp_accept_rates[q][rl] = t[21 + q * 2 + rl] / n;
            }
        }
        for (int rl = 0; rl < 2; ++rl) {
            out->n0[rl] = t[27 + rl] / n; out->eps0[rl] = t[29 + rl] / n;
            out->dn0[rl] = t[31 + rl] / n; out->deps0[rl] = t[33 + rl] / n;
        }
        out->chi2 = t[35] / n;
        out->chi2_min_mean = t[36] / n;
        out->chi2_min = t[EDGE_SLOTS];
        out->proposals = (uint64_t)t[37]; out->evaluated = (uint64_t)t[38]; out->accepted = (uint64_t)t[39];
        out->lookups = (uint64_t)t[40]; out->violations = (uint64_t)t[41];
        out->n_chains_total = (int64_t)n;
        out->chains_below_target = (int64_t)t[45];
        EdgeScalars last;
        HIP_TRY(copy_sync(h->stream, &last, h->dSc + (nc - 1), sizeof(last), hipMemcpyDeviceToHost));
        out->sweeps = last.sweep;
        out->last_stage_ms = h->last_stage_ms;
        return SAC_OK;
    }
    const int first = chain < 0 ? 0 : chain, count = chain < 0 ? nc : 1;
    std::vector<EdgeScalars> sc((size_t)count);
    HIP_TRY(copy_sync(h->stream, sc.data(), h->dSc + first, sc.size() * sizeof(EdgeScalars), hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof(*out));
    out->chi2_min = 1e300;
    for (const auto &s : sc) {
        sac_edge_results_t r;
        std::memset(&r, 0, sizeof(r));
        bin_averages(s, h->last_bins, h->cfg.n_rl, r);
        for (int q = 0; q < 3; ++q) {
            out->chi2_res[q] += r.chi2_res[q] / count;
            for (int rl = 0; rl < 2; ++rl) {
                out->edge_res[q][rl] += r.edge_res[q][rl] / count;
                out->n0_res[q][rl] += r.n0_res[q][rl] / count;
                out->wn0_res[q][rl] += r.wn0_res[q][rl] / count;
                out->amp_accept_rates[q][rl] += s.amp_ar[q][rl] / count;
            }
        }
        for (int rl = 0; rl < 2; ++rl) {
            out->n0[rl] += s.n0[rl] / count; out->eps0[rl] += s.eps0[rl] / count;
            out->dn0[rl] += s.dn0[rl] / count; out->deps0[rl] += s.deps0[rl] / count;
        }
        out->chi2 += s.chi2 / count;
        out->chi2_min_mean += s.chi2_min / count;
        if (s.chi2_min < out->chi2_min) out->chi2_min = s.chi2_min;
        out->sweeps = s.sweep;
        out->proposals += s.counters[0]; out->evaluated += s.counters[1]; out->accepted += s.counters[2];
        out->lookups += s.counters[4]; out->violations += s.violations;
    }
    out->n_chains_total = count;
    if (h->targets_on) {
        std::vector<int> pass((size_t)count);
        HIP_TRY(copy_sync(h->stream, pass.data(), h->dPass + first, pass.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int v : pass) out->chains_below_target += (v != 0) ? 1 : 0;
    }
    out->last_stage_ms = h->last_stage_ms;
    return SAC_OK;
}

// accept_rates[m, :, side] (edge:68) after the last adjust_D: N_w+1 doubles; chain = -1: chain mean
extern "C" int sac_edge_get_accept_rates(sac_edge_t *h, int32_t chain, int32_t m, int32_t side, double *out)
{
    if (!h || !out) return efail(SAC_E_INVALID, "sac_edge_get_accept_rates: null argument");
    const int nc = h->cfg.n_chains, n1 = h->cfg.n_omega + 1, S = h->S;
    if (chain < -1 || chain >= nc || m < 0 || m > 1 || side < 0 || side > 1)
        return efail(SAC_E_INVALID, "sac_edge_get_accept_rates: bad chain/m/side");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (chain < 0 && h->comm && h->comm->merged) {                   // mean over all ranks' chains (edge_comm_merge)
        HIP_TRY(copy_sync(h->stream, out, h->comm->dSum + h->off_arc() + ((size_t)side * 2 + (size_t)m) * (size_t)n1, (size_t)n1 * sizeof(double),
                          hipMemcpyDeviceToHost));
        const double n = h->comm->hTail[42];
        for (int k = 0; k < n1; ++k) out[k] = out[k] / n;
        return SAC_OK;
    }
    const int first = chain < 0 ? 0 : chain, count = chain < 0 ? nc : 1;
    std::vector<double> buf((size_t)count * 4 * S);
    HIP_TRY(copy_sync(h->stream, buf.data(), h->dArc + (size_t)first * 4 * S, buf.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int k = 1; k <= n1; ++k) out[k - 1] = 0.0;
    for (int c = 0; c < count; ++c)
        for (int k = 1; k <= n1; ++k) out[k - 1] += buf[(size_t)c * 4 * S + ((size_t)side * 2 + m) * S + k] / count;
    return SAC_OK;
}

// sampled_spec[:, which, side] (edge:75): accumulated (not normalised) sums, averaged over chains
extern "C" int sac_edge_get_histogram(sac_edge_t *h, int32_t which, int32_t side, double *out)
{
    if (!h || !out) return efail(SAC_E_INVALID, "sac_edge_get_histogram: null argument");
    if (which < 0 || which > 1 || side < 0 || side > 1) return efail(SAC_E_INVALID, "sac_edge_get_histogram: which/side");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t ng = (size_t)h->cfg.n_grid;
    if (h->comm && h->comm->merged) {                                // mean over all ranks' chains (edge_comm_merge)
        HIP_TRY(copy_sync(h->stream, out, h->comm->dSum + ((size_t)side * 2 + (size_t)which) * ng, ng * sizeof(double), hipMemcpyDeviceToHost));
        const double n = h->comm->hTail[42];
        if (n > 1)
            for (size_t i = 0; i < ng; ++i) out[i] = out[i] / n;
        return SAC_OK;
    }
    std::vector<double> tmp(ng);
    for (size_t i = 0; i < ng; ++i) out[i] = 0.0;
    for (int c = 0; c < h->hist_copies; ++c) {
        HIP_TRY(copy_sync(h->stream, tmp.data(), h->dHist + ((size_t)c * 4 + (size_t)side * 2 + which) * ng, ng * sizeof(double),
                          hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ng; ++i) out[i] = out[i] + tmp[i];
    }
    const double nc = (double)h->cfg.n_chains;
    if (nc > 1)
        for (size_t i = 0; i < ng; ++i) out[i] = out[i] / nc;
    return SAC_OK;
}

// w_avg / A_avg (edge:83-84) [N_w+1, 2], accumulated sums; chain = -1: chain mean
extern "C" int sac_edge_get_avg(sac_edge_t *h, int32_t chain, double *w_avg, double *A_avg)
{
    if (!h) return efail(SAC_E_INVALID, "sac_edge_get_avg: null handle");
    const int nc = h->cfg.n_chains, n1 = h->cfg.n_omega + 1, S = h->S;
    if (chain < -1 || chain >= nc) return efail(SAC_E_INVALID, "sac_edge_get_avg: chain %d", chain);
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (chain < 0 && h->comm && h->comm->merged) {                   // mean over all ranks' chains (edge_comm_merge)
        const double n = h->comm->hTail[42];
        for (int pass = 0; pass < 2; ++pass) {
            double *dst = pass == 0 ? w_avg : A_avg;
            if (!dst) continue;
            HIP_TRY(copy_sync(h->stream, dst, h->comm->dSum + (pass == 0 ? h->off_wavg() : h->off_aavg()), (size_t)(2 * n1) * sizeof(double),
                              hipMemcpyDeviceToHost));
            for (int i = 0; i < 2 * n1; ++i) dst[i] = dst[i] / n;
        }
        return SAC_OK;
    }
    const int first = chain < 0 ? 0 : chain, count = chain < 0 ? nc : 1;
    std::vector<double> buf((size_t)count * 2 * S);
    for (int pass = 0; pass < 2; ++pass) {
        double *dst = pass == 0 ? w_avg : A_avg;
        if (!dst) continue;
        HIP_TRY(copy_sync(h->stream, buf.data(), (pass == 0 ? h->dWavg : h->dAavg) + (size_t)first * 2 * S, buf.size() * sizeof(double),
                          hipMemcpyDeviceToHost));
        for (int i = 0; i < 2 * n1; ++i) dst[i] = 0.0;
        for (int c = 0; c < count; ++c)
            for (int rl = 0; rl < 2; ++rl)
                for (int k = 1; k <= n1; ++k) dst[rl * n1 + k - 1] += buf[(size_t)c * 2 * S + (size_t)rl * S + k] / count;
    }
    return SAC_OK;
}

// Explicit proposal stream for one chain (parity / debug entry): new positions of n consecutive deltas
// k .. k+n-1 of one side (n = 1: single_edge_move / single_cont_move, edge:856-880, 1176-1186; n > 1: multi_edge_move,
// edge:938-964) and the accept draw.  Evaluation is the sweep kernel's: Taylor lookup per delta (edge:626-633), Gbar_new
// accumulated in index order, chi2 by the canonical tree, accept when u < P (edge:864), commit of omega, K_curr, Gbar, chi2.
struct EdgeReplayProp { int side, k, n, reserved; long long w_offset; double u; };

template <int NTL>
__global__ void __launch_bounds__(64) edge_replay_kernel(const EdgeParams p, int chain, const EdgeReplayProp *props, long long n_props,
                                                         const double *w_new, double *chi2_out, int *acc_out)
{
    const int lane = lane_id();
    const size_t cb = (size_t)p.ntp * 8u;
    uint32_t off[NTL];
    double gbar[NTL], G[NTL], sinv[NTL];
    double *gbar_g = p.gbar + (size_t)chain * SAC_MAX_N_TAU;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        off[t] = (uint32_t)((j < p.ntp) ? j : p.ntp - 1) * 8u;
        gbar[t] = gbar_g[j];
        G[t] = p.G[j];
        sinv[t] = p.sinv[j];
    }
    EdgeScalars *sc = p.sc + chain;
    double chi2 = sc->chi2, chi2_min = sc->chi2_min;
    unsigned long long lookups = 0;
    for (long long i = 0; i < n_props; ++i) {
        const EdgeReplayProp pr = props[i];
        const size_t base = ((size_t)chain * 2u + (size_t)pr.side) * (size_t)p.S;
        double *w = p.w + base;
        const double *A = p.A + base;
        char *kc = reinterpret_cast<char *>(p.kcurr) + base * cb;
        char *kw = reinterpret_cast<char *>(p.knew) + base * cb;
        const char *tab = reinterpret_cast<const char *>(p.tab[pr.side]);
        double gn[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) gn[t] = gbar[t];
        for (int j = 0; j < pr.n; ++j) {
            const double wk = w_new[pr.w_offset + j];
            // floor(w / dw) by the correctly rounded division (edge:626)
            const int wi = (int)__builtin_floor(wk / p.dwg);
            const double d = wk - ((double)wi * p.dwg);
            const double d2 = d * d;
            long long col = (long long)wi - p.wi0;
            col = col < 0 ? 0 : (col >= p.n_grid ? p.n_grid - 1 : col);
            const char *b = tab + (size_t)col * 3u * cb;
            const double Ak = A[pr.k + j];
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                const double k0 = *reinterpret_cast<const double *>(b + off[t]);
                const double k1 = *reinterpret_cast<const double *>(b + cb + off[t]);
                const double k2 = *reinterpret_cast<const double *>(b + 2 * cb + off[t]);
                const double kn = (k0 + (k1 * d)) + (k2 * d2);
                const double old = *reinterpret_cast<const double *>(kc + (size_t)(pr.k + j) * cb + off[t]);
                gn[t] = gn[t] + Ak * (kn - old);
                *reinterpret_cast<double *>(kw + (size_t)(pr.k + j) * cb + off[t]) = kn;
            }
        }
        lookups += (unsigned long long)pr.n;
        double term = 0.0;
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            const double e = (gn[t] - G[t]) * sinv[t];
            term = (t == 0) ? e * e : term + e * e;
        }
        const double chi2_new = wave_sum_uniform(term);
        const bool acc = log_u01(pr.u) < (chi2 - chi2_new) * p.inv2theta;      // rand() < P
        if (acc) {
#pragma unroll
            for (int t = 0; t < NTL; ++t) gbar[t] = gn[t];
            for (int j = 0; j < pr.n; ++j) {
#pragma unroll
                for (int t = 0; t < NTL; ++t)
                    *reinterpret_cast<double *>(kc + (size_t)(pr.k + j) * cb + off[t]) =
                        *reinterpret_cast<const double *>(kw + (size_t)(pr.k + j) * cb + off[t]);
            }
            if (lane < pr.n) w[pr.k + lane] = w_new[pr.w_offset + lane];
            chi2 = chi2_new;
            if (chi2_new < chi2_min) chi2_min = chi2_new;
        }
        __syncthreads();
        if (lane == 0) { chi2_out[i] = chi2_new; acc_out[i] = acc ? 1 : 0; }
    }
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        if (j < p.n_tau) gbar_g[j] = gbar[t];
    }
    if (lane == 0) { sc->chi2 = chi2; sc->chi2_min = chi2_min; sc->counters[4] += lookups; }
}

extern "C" int sac_edge_replay(sac_edge_t *h, int32_t chain, double theta, const sac_edge_proposal_t *props, int64_t n,
                               const double *w_new, int64_t n_w, double *chi2_new, int32_t *accepted)
{
    if (!h || (n > 0 && (!props || !w_new))) return efail(SAC_E_INVALID, "sac_edge_replay: null argument");
    if (chain < 0 || chain >= h->cfg.n_chains) return efail(SAC_E_INVALID, "sac_edge_replay: chain %d outside 0..%d", chain, h->cfg.n_chains - 1);
    if (!h->problem || !h->state_set || !h->gbar_set)
        return efail(SAC_E_STATE, "sac_edge_replay: needs the problem, the state and sac_edge_init_gbar first");
    if (!(theta > 0.0) || !std::isfinite(theta)) return efail(SAC_E_INVALID, "sac_edge_replay: theta=%g", theta);
    static_assert(sizeof(EdgeReplayProp) == sizeof(sac_edge_proposal_t), "proposal layout");
    const double w_lo = (double)h->cfg.wi_0 * h->cfg.d_omega, w_hi = (double)(h->cfg.wi_0 + h->cfg.n_grid - 1) * h->cfg.d_omega;
    for (int64_t i = 0; i < n; ++i) {
        const sac_edge_proposal_t &q = props[i];
        if (q.side < 0 || q.side >= h->cfg.n_rl) return efail(SAC_E_INVALID, "sac_edge_replay: proposal %lld: side %d", (long long)i, q.side);
        if (q.n < 1 || q.n > 64 || q.k < 2 || q.k + q.n - 1 > h->cfg.n_omega + 1)
            return efail(SAC_E_INVALID, "sac_edge_replay: proposal %lld: deltas %d..%d outside 2..%d (at most 64 at a time)", (long long)i,
                         q.k, q.k + q.n - 1, h->cfg.n_omega + 1);
        if (q.w_offset < 0 || q.w_offset + q.n > n_w)
            return efail(SAC_E_INVALID, "sac_edge_replay: proposal %lld: positions %lld.. outside w_new[%lld]", (long long)i,
                         (long long)q.w_offset, (long long)n_w);
        for (int j = 0; j < q.n; ++j) {
            const double x = w_new[q.w_offset + j];
            if (!(x >= w_lo && x <= w_hi))
                return efail(SAC_E_INVALID, "sac_edge_replay: proposal %lld: frequency %g outside the kernel grid [%g, %g]", (long long)i, x, w_lo, w_hi);
        }
    }
    if (n == 0) return SAC_OK;
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    DevBuf b_dP, b_dC, b_dAcc, b_dW;      // released on every exit path
    EdgeReplayProp *dP = nullptr;
    double *dW = nullptr, *dC = nullptr;
    int *dAcc = nullptr;
    HIP_TRY(b_dP.alloc((size_t)n * sizeof(EdgeReplayProp)));
    dP = b_dP.as<std::remove_pointer_t<decltype(dP)>>();
    HIP_TRY(b_dW.alloc((size_t)n_w * sizeof(double)));
    dW = b_dW.as<std::remove_pointer_t<decltype(dW)>>();
    HIP_TRY(b_dC.alloc((size_t)n * sizeof(double)));
    dC = b_dC.as<std::remove_pointer_t<decltype(dC)>>();
    HIP_TRY(b_dAcc.alloc((size_t)n * sizeof(int)));
    dAcc = b_dAcc.as<std::remove_pointer_t<decltype(dAcc)>>();
    HIP_TRY(copy_sync(h->stream, dP, props, (size_t)n * sizeof(EdgeReplayProp), hipMemcpyHostToDevice));
    HIP_TRY(copy_sync(h->stream, dW, w_new, (size_t)n_w * sizeof(double), hipMemcpyHostToDevice));
    EdgeParams p;
    fill_params(h, p);
    p.inv2theta = 1.0 / (2.0 * theta);
    if (h->cfg.n_tau <= 64) hipLaunchKernelGGL(edge_replay_kernel<1>, dim3(1), dim3(64), 0, h->stream, p, (int)chain, dP, (long long)n, dW, dC, dAcc);
    else hipLaunchKernelGGL(edge_replay_kernel<2>, dim3(1), dim3(64), 0, h->stream, p, (int)chain, dP, (long long)n, dW, dC, dAcc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (chi2_new) HIP_TRY(copy_sync(h->stream, chi2_new, dC, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (accepted) HIP_TRY(copy_sync(h->stream, accepted, dAcc, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    return SAC_OK;
}

extern "C" int sac_edge_validation_chi2(sac_edge_t *h, const double *U_this, const double *U_other, const double *G_other,
                                        const double *sigma_inv_other, const double *gbar_mean, double *chi2_mean, double *chi2_chain)
{
    if (!h || !U_this || !U_other || !G_other || !sigma_inv_other)
        return efail(SAC_E_INVALID, "sac_edge_validation_chi2: null argument");
    if (!h->gbar_set) return efail(SAC_E_STATE, "sac_edge_validation_chi2: no Gbar yet (sac_edge_init_gbar)");
    const size_t nt = (size_t)h->cfg.n_tau;
    for (size_t j = 0; j < nt; ++j)
        if (!std::isfinite(G_other[j]) || !std::isfinite(sigma_inv_other[j]))
            return efail(SAC_E_INVALID, "sac_edge_validation_chi2: non-finite G or sigma_inv at tau index %zu", j);
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(cv_score_run(h->stream, h->dGbar, SAC_MAX_N_TAU, h->cfg.n_chains, h->cfg.n_tau, U_this, U_other, G_other,
                         sigma_inv_other, gbar_mean, chi2_mean, chi2_chain));
    return SAC_OK;
}

extern "C" int sac_edge_get_gbar(sac_edge_t *h, int32_t chain, double *gbar, double *chi2, double *chi2_min)
{
    if (!h) return efail(SAC_E_INVALID, "sac_edge_get_gbar: null handle");
    if (chain < 0 || chain >= h->cfg.n_chains) return efail(SAC_E_INVALID, "sac_edge_get_gbar: chain %d", chain);
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (gbar)
        HIP_TRY(copy_sync(h->stream, gbar, h->dGbar + (size_t)chain * SAC_MAX_N_TAU, (size_t)h->cfg.n_tau * sizeof(double), hipMemcpyDeviceToHost));
    EdgeScalars s;
    HIP_TRY(copy_sync(h->stream, &s, h->dSc + chain, sizeof(s), hipMemcpyDeviceToHost));
    if (chi2) *chi2 = s.chi2;
    if (chi2_min) *chi2_min = s.chi2_min;
    return SAC_OK;
}

extern "C" int sac_edge_comm_init(sac_edge_t *h, const void *id128, int32_t rank, int32_t world)
{
    if (!h || !id128) return efail(SAC_E_INVALID, "sac_edge_comm_init: null argument");
    if (h->comm) return efail(SAC_E_STATE, "sac_edge_comm_init: the handle has a communicator already");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    char msg[320];
    h->comm = saccomm::create(id128, rank, world, h->n_payload(), EDGE_SLOTS, 1, msg, sizeof(msg));
    if (!h->comm) return efail(SAC_E_HIP, "sac_edge_comm_init: %s", msg);
    return SAC_OK;
}

extern "C" int sac_edge_comm_info(sac_edge_t *h, int32_t *rank, int32_t *world, int64_t *merges, int64_t *allreduce_calls)
{
    if (!h) return efail(SAC_E_INVALID, "sac_edge_comm_info: null handle");
    if (rank) *rank = h->comm ? h->comm->rank : 0;
    if (world) *world = h->comm ? h->comm->world : 1;
    if (merges) *merges = h->comm ? h->comm->merges : 0;
    if (allreduce_calls) *allreduce_calls = h->comm ? h->comm->allreduce_calls : 0;
    return SAC_OK;
}

extern "C" int sac_edge_set_targets_from_chi2_min(sac_edge_t *h, double a_criterion)
{
    if (!h || !std::isfinite(a_criterion)) return efail(SAC_E_INVALID, "sac_edge_set_targets_from_chi2_min: bad argument");
    if (!h->gbar_set) return efail(SAC_E_STATE, "sac_edge_set_targets_from_chi2_min: no chi2_min yet (sac_edge_init_gbar first)");
    int rc;
    if ((rc = use_device(h))) return rc;
    const int nc = h->cfg.n_chains;
    hipLaunchKernelGGL(edge_targets_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, h->stream,
                       (const EdgeScalars *)h->dSc, a_criterion, h->dTargets, h->dPass, nc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->targets_on = true;
    h->stages_since_targets = 0;
    if (h->comm) h->comm->merged = false;
    return SAC_OK;
}

extern "C" int sac_edge_set_chain_targets(sac_edge_t *h, const double *targets)
{
    if (!h) return efail(SAC_E_INVALID, "sac_edge_set_chain_targets: null handle");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t nc = (size_t)h->cfg.n_chains;
    HIP_TRY(hipMemset(h->dPass, 0, nc * sizeof(int)));
    h->targets_on = (targets != nullptr);
    h->stages_since_targets = 0;
    if (targets) HIP_TRY(copy_sync(h->stream, h->dTargets, targets, nc * sizeof(double), hipMemcpyHostToDevice));
    if (h->comm) h->comm->merged = false;
    return SAC_OK;
}

extern "C" int sac_edge_get_first_passages(sac_edge_t *h, int32_t *stage)
{
    if (!h || !stage) return efail(SAC_E_INVALID, "sac_edge_get_first_passages: null argument");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    static_assert(sizeof(int) == sizeof(int32_t), "int is 32 bits");
    HIP_TRY(copy_sync(h->stream, stage, h->dPass, (size_t)h->cfg.n_chains * sizeof(int), hipMemcpyDeviceToHost));
    return SAC_OK;
}

extern "C" int sac_edge_get_move_means(sac_edge_t *h, double *dw_mean, double *n_multi_mean, double *cont_extent)
{
    if (!h) return efail(SAC_E_INVALID, "sac_edge_get_move_means: null handle");
    int rc;
    if ((rc = use_device(h))) return rc;
    const int nc = h->cfg.n_chains, n1 = h->cfg.n_omega + 1, S = h->S;
    const bool cont = h->cfg.n_omega > h->cfg.n_e;
    std::vector<double> buf(4 * (size_t)n1 + 2, 0.0);
    double n = (double)nc;
    if (h->comm && h->comm->merged) {                                // summed over all ranks' chains (edge_comm_merge)
        HIP_TRY(copy_sync(h->stream, buf.data(), h->comm->dSum + h->off_dw(), 4 * (size_t)n1 * sizeof(double), hipMemcpyDeviceToHost));
        buf[4 * (size_t)n1] = h->comm->hTail[43];
        buf[4 * (size_t)n1 + 1] = h->comm->hTail[44];
        n = h->comm->hTail[42];
    } else {
        hipLaunchKernelGGL(edge_sum_chains_kernel, dim3((unsigned)((2 * n1 + 255) / 256)), dim3(256), 0, h->stream,
                           (const double *)h->dDw, nc, 2ll * S, 2, S, n1, h->dMeans);
        hipLaunchKernelGGL(edge_sum_chains_int_kernel, dim3((unsigned)((2 * n1 + 255) / 256)), dim3(256), 0, h->stream,
                           (const int *)h->dNm, nc, 2ll * S, 2, S, n1, h->dMeans + 2 * n1);
        HIP_TRY(hipMemsetAsync(h->dMeans + 4 * n1, 0, 2 * sizeof(double), h->stream));
        if (cont)
            hipLaunchKernelGGL(edge_cont_extent_kernel, dim3(1), dim3(64), 0, h->stream, (const double *)h->dW, nc, S, h->cfg.n_e,
                               h->cfg.n_omega, h->dMeans + 4 * n1);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(buf.data(), h->dMeans, buf.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    for (int i = 0; i < 2 * n1; ++i) {
        if (dw_mean) dw_mean[i] = buf[(size_t)i] / n;
        if (n_multi_mean) n_multi_mean[i] = buf[2 * (size_t)n1 + (size_t)i] / n;
    }
    if (cont_extent) { cont_extent[0] = buf[4 * (size_t)n1] / n; cont_extent[1] = buf[4 * (size_t)n1 + 1] / n; }
    return SAC_OK;
}

extern "C" int sac_edge_get_gbar_mean(sac_edge_t *h, double *gbar_mean)
{
    if (!h || !gbar_mean) return efail(SAC_E_INVALID, "sac_edge_get_gbar_mean: null argument");
    int rc;
    if ((rc = use_device(h))) return rc;
    const size_t nt = (size_t)h->cfg.n_tau;
    double n = (double)h->cfg.n_chains;
    if (h->comm && h->comm->merged) {                                // summed over all ranks' chains (edge_comm_merge)
        HIP_TRY(copy_sync(h->stream, gbar_mean, h->comm->dSum + h->off_gbar(), nt * sizeof(double), hipMemcpyDeviceToHost));
        n = h->comm->hTail[42];
    } else {
        hipLaunchKernelGGL(edge_gbar_sum_kernel, dim3(1), dim3(SAC_MAX_N_TAU), 0, h->stream, (const double *)h->dGbar,
                           h->cfg.n_chains, h->dGbarSum);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(gbar_mean, h->dGbarSum, nt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    for (size_t t = 0; t < nt; ++t) gbar_mean[t] = gbar_mean[t] / n;
    return SAC_OK;
}

extern "C" int sac_edge_set_move_log(sac_edge_t *h, int32_t chain, int64_t capacity)
{
    if (!h) return efail(SAC_E_INVALID, "sac_edge_set_move_log: null handle");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->dLog) { HIP_TRY(hipFree(h->dLog)); h->dLog = nullptr; }
    h->log_chain = -1;
    h->log_cap = 0;
    if (chain < 0 || capacity <= 0) return SAC_OK;
    if (chain >= h->cfg.n_chains) return efail(SAC_E_INVALID, "sac_edge_set_move_log: chain %d", chain);
    HIP_TRY(hipMalloc((void **)&h->dLog, (size_t)capacity * sizeof(MoveRec)));
    HIP_TRY(hipMemset(h->dLogLen, 0, sizeof(unsigned long long)));
    h->log_chain = chain;
    h->log_cap = capacity;
    return SAC_OK;
}

extern "C" int sac_edge_get_move_log(sac_edge_t *h, sac_move_rec_t *out, int64_t capacity, int64_t *n_recorded)
{
    if (!h || !n_recorded) return efail(SAC_E_INVALID, "sac_edge_get_move_log: null argument");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    unsigned long long n = 0;
    if (h->dLog) HIP_TRY(copy_sync(h->stream, &n, h->dLogLen, sizeof(n), hipMemcpyDeviceToHost));
    *n_recorded = (int64_t)n;
    long long m = (long long)n < h->log_cap ? (long long)n : h->log_cap;
    if (m > capacity) m = capacity;
    if (out && m > 0) {
        static_assert(sizeof(MoveRec) == sizeof(sac_move_rec_t), "move record layout");
        HIP_TRY(copy_sync(h->stream, out, h->dLog, (size_t)m * sizeof(MoveRec), hipMemcpyDeviceToHost));
    }
    return SAC_OK;
}
