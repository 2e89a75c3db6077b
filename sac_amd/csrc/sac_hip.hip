// libsac_hip.so -- kernels and C ABI (include/sac_hip.h) of the MI355X SAC sampling engine.
// gfx950 only; build: see __graft_entry__.build() / Makefile (hipcc --offload-arch=gfx950
// -ffp-contract=off -munsafe-fp-atomics).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "../../include/sac_hip.h"
#include "free_sweep.h"
#include "hist_sort.h"
#include "free_sweep_packed.h"
#include "free_sweep_spec.h"
#include "peak_sweep.h"
#include "table_build.h"
#include "cv_score.h"
#include "comm.h"
#include "devbuf.h"

using namespace sacdev;

// =========================================================================================
// Kernels
// =========================================================================================

// One launch = n_sweeps sweeps (run_updates, free:825-861) of every chain, plus, by flag, the
// per-sweep sampling work (free:918-919) and the end-of-round window retune (free:883-895).
// PAR2 (amplitude updates, free:839-855): the amplitudes are read and written in place in global memory
// (16 KB per chain, L2 resident), the positions stay in LDS.
template <int NTL, bool LOG, bool UA, bool PAR2>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, 4)
free_stage_kernel(const FreeParams p)
{
    constexpr int WPB = WAVES_PER_BLOCK;
    __shared__ uint32_t s_wi[WPB][MAX_N_OMEGA];
    __shared__ double s_sc[WPB][12];   // windows[4], accept accumulators[6], sum chi2
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chain = (int)blockIdx.x * WPB + wave;
    if (chain >= p.n_chains) return;
    const int lane = lane_id();

    Chain<NTL> c;
    c.Kb = reinterpret_cast<const char *>(p.K);
    c.col_bytes = (uint32_t)p.ntp * 8u;
    c.tab_bytes = (uint32_t)((unsigned long long)p.n_grid * (unsigned long long)p.ntp * 8ull);
    c.wi = s_wi[wave];
    c.lane = lane;
    c.prio.init(p.prio_shift);
    c.n_omega = p.n_omega;
    c.gmax = p.gmax;
    c.k_first = 0; c.k_count = p.n_omega; c.lo = 0; c.hi = p.gmax;
    c.inv2theta = p.inv2theta;
    c.key0 = p.seed_lo;
    c.key1 = p.seed_hi;
    c.chain_id = p.chain_offset + (uint32_t)chain;
    double *A_g = p.A + (size_t)chain * (size_t)p.wi_stride;
    c.A = A_g;
    c.A_uniform = p.A_uniform;
    c.n_eval = c.n_acc = c.n_accmoves = c.n_cols = 0;
    double *gbar_g = p.gbar + (size_t)chain * SAC_MAX_N_TAU;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        c.off[t] = (uint32_t)((j < p.ntp) ? j : p.ntp - 1) * 8u;
        c.gbar[t] = gbar_g[j];
        c.G[t] = p.G[j];
        c.sinv[t] = p.sinv[j];
    }
    uint32_t *wi_g = p.wi + (size_t)chain * (size_t)p.wi_stride;
    for (int i = lane; i < p.n_omega; i += 64) c.wi[i] = wi_g[i];

    ChainScalars *scp = p.sc + chain;
    c.chi2 = scp->chi2;
    c.chi2_min = scp->chi2_min;
    // rarely touched per-chain scalars stay in LDS, not in registers
    double *win = s_sc[wave], *acc = s_sc[wave] + 4, *sum_chi2 = s_sc[wave] + 10;
    const bool zero = (p.flags & STAGE_ZERO_ACC) != 0;
    if (lane < 4) win[lane] = scp->windows[lane];
    if (lane < 6) acc[lane] = zero ? 0.0 : scp->acc[lane];
    if (lane == 0) *sum_chi2 = zero ? 0.0 : scp->sum_chi2;
    unsigned long long sweep = scp->sweep;
    c.log = nullptr;
    c.log_cap = 0;
    c.log_pos = 0;
    if constexpr (LOG) {
        if (chain == p.log_chain) {
            c.log = p.log;
            c.log_cap = p.log_cap;
            c.log_pos = *p.log_len;
        }
    }
    double *hist = p.hist + (size_t)(chain % p.hist_copies) * (size_t)p.n_grid;

    const int n = p.n_omega;
    for (int s = 0; s < p.n_sweeps; ++s) {
        c.sweep_lo = (uint32_t)sweep;
        c.sweep_hi = (uint32_t)(sweep >> 32) & 0xFFFFu;
        int a;
        // prefetch depth per move type: two tau slots per lane double the ring's registers
        constexpr int D1 = (NTL == 1) ? 8 : 4, D2 = (NTL == 1) ? 4 : 2, D3 = (NTL == 1) ? 4 : 2;
        a = move_loop<1, D1, NTL, LOG, UA>(c, 1, n, win[0]);        // free:829-830
        if (lane == 0) acc[0] += (double)a / (double)n;
        a = move_loop<2, D2, NTL, LOG, UA>(c, 2, n / 2, win[1]);    // free:832-833
        if (lane == 0) acc[1] += (double)(2 * a) / (double)(n / 2);
        a = move_loop<3, D3, NTL, LOG, UA>(c, 3, n / 3, 0.0);       // free:835-836
        if (lane == 0) acc[2] += (double)a / (double)(n / 3);
        if constexpr (PAR2) {
            double *A = A_g;
            a = aw_move_loop<(NTL == 1) ? 4 : 2, NTL, LOG>(c, A, n / 2, win[3]);   // free:851-852
            if (lane == 0) acc[3] += (double)(2 * a) / (double)(n / 2);
            renormalise(A, n, p.symm, lane);                        // free:563-567
            a = da_move_loop<(NTL == 1) ? 8 : 4, NTL, LOG>(c, A, n);               // free:854-855
            if (lane == 0) acc[4] += (double)a / (double)n;
            renormalise(A, n, p.symm, lane);                        // free:721-725
        }
        if (p.flags & STAGE_SAMPLE) {
            if (p.hist_mode == SAC_HIST_REFERENCE) {
                if (p.n_grid < (1ll << 21))
                    hist_reference<uint32_t, 32>(c.wi, n, c.A, c.A_uniform, UA ? 1 : 0, hist,
                                                 p.hist_exclusive, lane);
                else
                    hist_reference<uint64_t, 32>(c.wi, n, c.A, c.A_uniform, UA ? 1 : 0, hist,
                                                 p.hist_exclusive, lane);
            } else {
                hist_accumulate(c.wi, n, c.A, c.A_uniform, UA ? 1 : 0, hist, lane);
            }
            if (lane == 0) *sum_chi2 += c.chi2;                      // free:919
        }
        sweep += 1;
    }

    if ((p.flags & STAGE_RETUNE) && lane == 0) {                     // free:883-895
        for (int k = 0; k < 6; ++k) acc[k] = acc[k] / p.retune_div;
        for (int k = 0; k < 4; ++k) {
            const double r = acc[k];
            if (r > 0.8) win[k] *= 2;
            else if (r < 0.2) win[k] /= 2;
            else if (r > 0.55) win[k] *= 1.2;
            else if (r < 0.45) win[k] /= 1.2;
        }
    }

    for (int i = lane; i < p.n_omega; i += 64) wi_g[i] = c.wi[i];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        if (j < p.n_tau) gbar_g[j] = c.gbar[t];
    }
    if (lane == 0) {
        scp->chi2 = c.chi2;
        scp->chi2_min = c.chi2_min;
        for (int k = 0; k < 4; ++k) scp->windows[k] = win[k];
        for (int k = 0; k < 6; ++k) scp->acc[k] = acc[k];
        scp->sum_chi2 = *sum_chi2;
        scp->sweep = sweep;
        scp->counters[0] += (unsigned long long)p.n_sweeps *
                            (unsigned long long)(n + n / 2 + n / 3 + (PAR2 ? n / 2 + n : 0));
        scp->counters[1] += c.n_eval;
        scp->counters[2] += c.n_acc;
        scp->counters[3] += c.n_accmoves;
        scp->counters[4] += c.n_cols;
        if constexpr (LOG) {
            if (c.log != nullptr) *p.log_len = c.log_pos;
        }
    }
}

// Gbar = K[:, idx] * A per chain, summed sequentially over the deltas (no fma), then chi^2 in
// the canonical tree order; chi2_min = chi2 (free:243-255, 1213-1216).
template <int NTL>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) init_gbar_kernel(const FreeParams p)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chain = (int)blockIdx.x * WAVES_PER_BLOCK + wave;
    if (chain >= p.n_chains) return;
    const int lane = lane_id();
    const uint32_t *wi = p.wi + (size_t)chain * (size_t)p.wi_stride;
    const double *A = p.A + (size_t)chain * (size_t)p.wi_stride;
    double acc[NTL], G[NTL], sinv[NTL];
    int idx[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        idx[t] = (j < p.ntp) ? j : p.ntp - 1;
        acc[t] = 0.0;
        G[t] = p.G[j];
        sinv[t] = p.sinv[j];
    }
    for (int i = 0; i < p.n_omega; ++i) {
        const double a = A[i];
        const double *col = p.K + (size_t)wi[i] * (size_t)p.ntp;
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[t] = acc[t] + col[idx[t]] * a;
    }
    double term = 0.0;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        if (j >= p.n_tau) acc[t] = 0.0;
        const double e = (acc[t] - G[t]) * sinv[t];
        const double e2 = e * e;
        term = (t == 0) ? e2 : (term + e2);
    }
    const double chi2 = wave_sum_uniform(term);
    double *gbar_g = p.gbar + (size_t)chain * SAC_MAX_N_TAU;
#pragma unroll
    for (int t = 0; t < NTL; ++t) gbar_g[lane + 64 * t] = acc[t];
    if (lane == 0) {
        p.sc[chain].chi2 = chi2;
        p.sc[chain].chi2_min = chi2;
    }
}

// ---- peak parameterisation -------------------------------------------------------------------
// One launch = n_sweeps sweeps of sac_peak.jl's run_updates (peak:1006-1059) with the per-sweep
// from-scratch refresh of Gbar/chi2 (peak:1072-1073, 1120-1121), by flag the sampling work
// (peak:1126-1147) and the window retune (peak:1079-1091).  XFER: the amplitudes change (weight
// transfer between the axes); they are read and written in place in global memory (16 KB per chain at
// 2 x 1001 deltas, L2 resident), so that LDS holds the positions only and 16 chains stay resident per CU.

#ifdef PEAK_PROF   // development aid (as EDGE_PROF): cycles per section of chain 0's sweep, printed at the end of the launch
#define PEAK_NSC 42
#define PEAK_TICK(i) do { const unsigned long long now_ = clock64(); if (lane == 0) reinterpret_cast<unsigned long long *>(s_sc_w + 26)[i] += now_ - tprev; tprev = now_; } while (0)
#else
#define PEAK_NSC 26
#define PEAK_TICK(i) do { } while (0)
#endif
__host__ __device__ constexpr size_t peak_lds_bytes_per_wave(int wi_stride, bool xfer)
{
    (void)xfer;                     // the transfer variants keep their amplitudes in global memory (L2)
    return (size_t)wi_stride * 4u + (size_t)PEAK_NSC * 8u;
}

template <int NTL, bool LOG, bool XFER>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, 4)
peak_stage_kernel(const PeakParams pp)
{
    constexpr int WPB = WAVES_PER_BLOCK;
    const FreeParams &p = pp.f;
    // dynamic LDS, per wave: [amplitudes f64 x wi_stride (XFER)] [26 f64: windows[11], accept
    // accumulators[11], sum chi2, edge sums[2]] [positions u32 x wi_stride]; sized by peak_lds_bytes()
    extern __shared__ __align__(16) unsigned char s_dyn[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char *s_mine = s_dyn + (size_t)wave * peak_lds_bytes_per_wave(p.wi_stride, XFER);
    double *s_sc_w = reinterpret_cast<double *>(s_mine);
    uint32_t *s_wi_w = reinterpret_cast<uint32_t *>(s_sc_w + PEAK_NSC);
    const int chain = (int)blockIdx.x * WPB + wave;
    if (chain >= p.n_chains) return;
    const int lane = lane_id();
#ifdef PEAK_PROF
    if (lane < 16) reinterpret_cast<unsigned long long *>(s_sc_w + 26)[lane] = 0ull;
    unsigned long long tprev = clock64();
#endif
    const int n_p = pp.n_p, n = pp.n_cont;
    const int n_used = pp.sample_neg ? 2 * (n_p + n) : (n_p + n);
    const char *Kp = reinterpret_cast<const char *>(p.K), *Kn = reinterpret_cast<const char *>(pp.Kn);

    Chain<NTL> c;
    c.Kb = Kp;
    c.col_bytes = (uint32_t)p.ntp * 8u;
    c.tab_bytes = (uint32_t)((unsigned long long)p.n_grid * (unsigned long long)p.ntp * 8ull);
    c.wi = s_wi_w;
    c.lane = lane;
    c.prio.init(p.prio_shift);
    c.n_omega = n;
    c.gmax = pp.wim;
    c.inv2theta = p.inv2theta;
    c.key0 = p.seed_lo;
    c.key1 = p.seed_hi;
    c.chain_id = p.chain_offset + (uint32_t)chain;
    double *A_g = p.A + (size_t)chain * (size_t)p.wi_stride;
    double *A = A_g;      // XFER: read and written in place; a chain's 16 KB stay in L2 between its transfer moves
    c.A = A;
    c.A_uniform = 0.0;
    c.n_eval = c.n_acc = c.n_accmoves = c.n_cols = 0;
    unsigned long long n_prop = 0;
    double *gbar_g = p.gbar + (size_t)chain * SAC_MAX_N_TAU;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        c.off[t] = (uint32_t)((j < p.ntp) ? j : p.ntp - 1) * 8u;
        c.gbar[t] = gbar_g[j];
        c.G[t] = p.G[j];
        c.sinv[t] = p.sinv[j];
    }
    uint32_t *wi_g = p.wi + (size_t)chain * (size_t)p.wi_stride;
    for (int i = lane; i < n_used; i += 64) c.wi[i] = wi_g[i];

    PeakScalars *scp = pp.psc + chain;
    c.chi2 = scp->chi2;
    c.chi2_min = scp->chi2_min;
    double *win = s_sc_w, *acc = s_sc_w + 11, *sum_chi2 = s_sc_w + 22, *edge = s_sc_w + 23;
    const bool zero = (p.flags & STAGE_ZERO_ACC) != 0;
    if (lane < 11) { win[lane] = scp->windows[lane]; acc[lane] = zero ? 0.0 : scp->acc[lane]; }
    if (lane == 0) *sum_chi2 = zero ? 0.0 : scp->sum_chi2;
    if (lane < 2) edge[lane] = zero ? 0.0 : scp->edge_sum[lane];
    unsigned long long sweep = scp->sweep;
    int b_pp = scp->bounds[0], b_pc = scp->bounds[1], b_np = scp->bounds[2], b_nc = scp->bounds[3];
    c.log = nullptr;
    c.log_cap = 0;
    c.log_pos = 0;
    if constexpr (LOG) {
        if (chain == p.log_chain) { c.log = p.log; c.log_cap = p.log_cap; c.log_pos = *p.log_len; }
    }
    double *hist = p.hist + (size_t)(chain % p.hist_copies) * 4 * (size_t)p.n_grid;
    const int f_pp = 0, f_cp = n_p, f_pn = n_p + n, f_cn = 2 * n_p + n;
    const int wi0 = pp.wi0, wim = pp.wim;
    auto imax = [](int a, int b) { return a > b ? a : b; };
    auto imin = [](int a, int b) { return a < b ? a : b; };

    if (p.flags & STAGE_BOUNDS_AT_START) {                           // peak:1115-1116
        b_pp = group_max(c.wi, f_pp, n_p, lane);
        if (pp.sample_neg) b_np = group_max(c.wi, f_pn, n_p, lane);
    }
#ifndef PEAK_D1
#define PEAK_D1 8
#endif
    constexpr int D1 = (NTL == 1) ? PEAK_D1 : 4, D2 = (NTL == 1) ? 4 : 2, D3 = (NTL == 1) ? 4 : 2;
    for (int s = 0; s < p.n_sweeps; ++s) {
        c.sweep_lo = (uint32_t)sweep;
        c.sweep_hi = (uint32_t)(sweep >> 32) & 0xFFFFu;
        PEAK_TICK(9);
        peak_refresh_gbar<NTL>(c, Kp, Kn, A, n_p, n, pp.sample_neg);
        PEAK_TICK(0);
        if (pp.refresh_bounds) {                                     // peak:1074-1075
            b_pp = group_max(c.wi, f_pp, n_p, lane);
            if (pp.sample_neg) b_np = group_max(c.wi, f_pn, n_p, lane);
        }
        int a;
        if constexpr (XFER) {
            if (pp.transfer && pp.sample_neg) {                      // peak:1018-1034
                if (pp.indiv_update) {
                    XferSpec<1> sp;
                    sp.move_id = 19;                                 // cont_Aw_transfer, peak:706-792
                    sp.first[0] = f_cp; sp.count[0] = n; sp.first[1] = f_cn; sp.count[1] = n;
                    sp.lo[0] = imax(wi0, b_pp); sp.hi[0] = wim;
                    sp.lo[1] = imax(wi0, b_pp); sp.hi[1] = wim;     // sic: peak:738 tests wi_pp on the n side
                    sp.window[0] = win[5];
                    a = xfer_loop<1, (NTL == 1) ? 4 : 2, NTL, LOG>(c, Kp, Kn, A, sp, n / 2);
                    n_prop += n / 2;
                    if (lane == 0) acc[5] += (double)a / (double)(n / 2);
                    PEAK_TICK(1);
                    b_pc = group_min(c.wi, f_cp, n, lane);           // peak:787-788
                    b_nc = group_min(c.wi, f_cn, n, lane);
                    sp.move_id = 20;                                 // peak_Aw_transfer, peak:798-880
                    sp.first[0] = f_pp; sp.count[0] = n_p; sp.first[1] = f_pn; sp.count[1] = n_p;
                    sp.lo[0] = wi0; sp.hi[0] = imin(b_pc, wim);
                    sp.lo[1] = wi0; sp.hi[1] = imin(b_nc, wim);
                    sp.window[0] = win[6];
                    a = xfer_loop<1, (NTL == 1) ? 4 : 2, NTL, LOG>(c, Kp, Kn, A, sp, 10);
                    n_prop += 10;
                    if (lane == 0) acc[6] += (double)a / 10.0;
                    PEAK_TICK(2);
                    b_pp = group_max(c.wi, f_pp, n_p, lane);         // peak:876-877
                    b_np = group_max(c.wi, f_pn, n_p, lane);
                }
                PEAK_TICK(9);
                XferSpec<2> sp;                                      // Aw_transfer, peak:882-996
                sp.move_id = 21;
                sp.first[0] = f_cp; sp.count[0] = n; sp.first[1] = f_cn; sp.count[1] = n;
                sp.first[2] = f_pp; sp.count[2] = n_p; sp.first[3] = f_pn; sp.count[3] = n_p;
                sp.lo[0] = imax(wi0, b_pp); sp.hi[0] = wim;          // peak:918
                sp.lo[1] = imax(wi0, b_np); sp.hi[1] = wim;          // peak:921
                sp.lo[2] = wi0; sp.hi[2] = imin(b_pc, wim);          // peak:929
                sp.lo[3] = wi0; sp.hi[3] = imin(b_nc, wim);          // peak:932
                sp.window[0] = win[10]; sp.window[1] = win[10];
                a = xfer_loop<2, 2, NTL, LOG>(c, Kp, Kn, A, sp, n / 2);
                n_prop += n / 2;
                if (lane == 0) acc[10] += (double)a / (double)(n / 2);
                PEAK_TICK(3);
            }
        }
        for (int pn = 1; pn <= (pp.sample_neg ? 2 : 1); ++pn) {      // peak:1037-1047
            const int o7 = (pn - 1) * 7;
            c.Kb = (pn == 1) ? Kp : Kn;
            c.k_first = (pn == 1) ? f_cp : f_cn;
            c.k_count = n;
            c.lo = ((pn == 1) ? b_pp : b_np) + 1;                    // new <= floor(w_b): continue (peak:409)
            c.hi = wim;
            PEAK_TICK(9);
            a = move_loop_live<1, D1, NTL, LOG, false>(c, 11 + pn - 1, n, win[0 + o7]);
            if (lane == 0) acc[0 + o7] += (double)a / (double)n;
            PEAK_TICK(4);
            a = move_loop_live<2, D2, NTL, LOG, false>(c, 13 + pn - 1, n / 2, win[1 + o7]);
            if (lane == 0) acc[1 + o7] += (double)(2 * a) / (double)(n / 2);
            PEAK_TICK(5);
            a = move_loop_live<3, D3, NTL, LOG, false>(c, 15 + pn - 1, n / 3, 0.0);
            if (lane == 0) acc[2 + o7] += (double)a / (double)(n / 3);
            PEAK_TICK(6);
            n_prop += n + n / 2 + n / 3;
            const int m = group_min(c.wi, c.k_first, n, lane);       // peak:440-444
            if (pn == 1) b_pc = m; else b_nc = m;
        }
        if (pp.fix_edge != 1) {                                      // peak:1048-1056
            for (int pn = 1; pn <= (pp.sample_neg ? 2 : 1); ++pn) {
                c.Kb = (pn == 1) ? Kp : Kn;
                c.k_first = (pn == 1) ? f_pp : f_pn;
                c.k_count = n_p;
                c.lo = wi0;                                          // peak:659-663
                c.hi = (pn == 1) ? b_pc : b_nc;
                a = move_loop_live<1, D1, NTL, LOG, false>(c, 17 + pn - 1, 10 * n_p, win[pn == 1 ? 3 : 4]);
                if (lane == 0) acc[pn == 1 ? 3 : 4] += (double)a / (double)(10 * n_p);
                n_prop += 10 * n_p;
                const int m = group_max(c.wi, c.k_first, n_p, lane); // peak:693-697
                if (pn == 1) b_pp = m; else b_np = m;
            }
        }
        PEAK_TICK(7);
        if (p.flags & STAGE_SAMPLE) {
            for (int i = lane; i < n_used; i += 64) {                // peak:1126-1140
                const int which = (i >= f_cn) ? 3 : (i >= f_pn) ? 2 : (i >= f_cp) ? 1 : 0;
                unsafeAtomicAdd(&hist[(size_t)which * (size_t)p.n_grid + c.wi[i]], A[i]);
            }
            b_pp = group_max(c.wi, f_pp, n_p, lane);                 // peak:1142-1143
            if (pp.sample_neg) b_np = group_max(c.wi, f_pn, n_p, lane);
            if (lane == 0) {
                *sum_chi2 += c.chi2;                                 // peak:1145-1147
                edge[0] += (double)b_pp;
                edge[1] += (double)b_np;
            }
        }
        PEAK_TICK(8);
        sweep += 1;
    }
#ifdef PEAK_PROF
    if (chain == 0 && lane == 0 && p.n_sweeps > 1) {
        const unsigned long long *pr = reinterpret_cast<const unsigned long long *>(s_sc_w + 26);
        printf("peak prof sweeps=%d cycles: refresh_gbar %llu | cont_xfer %llu peak_xfer %llu Aw_transfer %llu | cont single %llu double %llu triple %llu | "
               "peak moves %llu | sample %llu | other %llu\n", p.n_sweeps, pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7], pr[8], pr[9]);
    }
#endif

    if ((p.flags & STAGE_RETUNE) && lane == 0) {                     // peak:1079-1091
        for (int k = 0; k < 11; ++k) acc[k] = acc[k] / p.retune_div;
        for (int k = 0; k < 11; ++k) {
            if (acc[k] > 0.55) win[k] *= 1.2;
            else if (acc[k] < 0.45) win[k] /= 1.2;
        }
    }
    for (int i = lane; i < n_used; i += 64) wi_g[i] = c.wi[i];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        if (j < p.n_tau) gbar_g[j] = c.gbar[t];
    }
    if (lane == 0) {
        scp->chi2 = c.chi2;
        scp->chi2_min = c.chi2_min;
        for (int k = 0; k < 11; ++k) { scp->windows[k] = win[k]; scp->acc[k] = acc[k]; }
        scp->sum_chi2 = *sum_chi2;
        scp->edge_sum[0] = edge[0];
        scp->edge_sum[1] = edge[1];
        scp->sweep = sweep;
        scp->bounds[0] = b_pp; scp->bounds[1] = b_pc; scp->bounds[2] = b_np; scp->bounds[3] = b_nc;
        scp->counters[0] += n_prop;
        scp->counters[1] += c.n_eval;
        scp->counters[2] += c.n_acc;
        scp->counters[3] += c.n_accmoves;
        scp->counters[4] += c.n_cols;
        if constexpr (LOG) {
            if (c.log != nullptr) *p.log_len = c.log_pos;
        }
    }
}

// peak:1503-1506 and the bounds initialize! leaves (peak:190-191, 202-203)
template <int NTL>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) peak_init_kernel(const PeakParams pp)
{
    const FreeParams &p = pp.f;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chain = (int)blockIdx.x * WAVES_PER_BLOCK + wave;
    if (chain >= p.n_chains) return;
    const int lane = lane_id();
    const int n_p = pp.n_p, n = pp.n_cont;
    Chain<NTL> c;
    c.Kb = reinterpret_cast<const char *>(p.K);
    c.col_bytes = (uint32_t)p.ntp * 8u;
    c.tab_bytes = (uint32_t)((unsigned long long)p.n_grid * (unsigned long long)p.ntp * 8ull);
    c.wi = p.wi + (size_t)chain * (size_t)p.wi_stride;   // read in place (one pass)
    c.lane = lane;
    c.prio.init(p.prio_shift);
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        c.off[t] = (uint32_t)((j < p.ntp) ? j : p.ntp - 1) * 8u;
        c.G[t] = p.G[j];
        c.sinv[t] = p.sinv[j];
    }
    const double *A = p.A + (size_t)chain * (size_t)p.wi_stride;
    peak_refresh_gbar<NTL>(c, c.Kb, reinterpret_cast<const char *>(pp.Kn), A, n_p, n, pp.sample_neg);
    double *gbar_g = p.gbar + (size_t)chain * SAC_MAX_N_TAU;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        gbar_g[j] = (j < p.n_tau) ? c.gbar[t] : 0.0;
    }
    const int b_pp = group_max(c.wi, 0, n_p, lane), b_pc = group_min(c.wi, n_p, n, lane);
    int b_np = 0, b_nc = 0;
    if (pp.sample_neg) {
        b_np = group_max(c.wi, n_p + n, n_p, lane);
        b_nc = group_min(c.wi, 2 * n_p + n, n, lane);
    }
    if (lane == 0) {
        PeakScalars *scp = pp.psc + chain;
        scp->chi2 = c.chi2;
        scp->chi2_min = c.chi2;
        scp->bounds[0] = b_pp; scp->bounds[1] = b_pc; scp->bounds[2] = b_np; scp->bounds[3] = b_nc;
    }
}

// Explicit proposal stream for one chain of the free parameterisation (parity / debug entry,
// SURVEY 8b-7): every proposal names its move (1, 2 or 3 deltas), the delta indices, their new
// positions and the accept uniform, in the order a foreign sampler drew them.  Evaluation is the
// sweep kernels' (free:296-303, 354-363, 437-449: Gbar_new, chi2_new, Metropolis, commit).
struct ReplayProp { int move; int k[3]; long long w_new[3]; double u; };

template <int NTL>
__global__ void __launch_bounds__(64) replay_kernel(const FreeParams p, int chain, const ReplayProp *props, long long n,
                                                    double *chi2_out, int *acc_out)
{
    const int lane = lane_id();
    uint32_t *wi = p.wi + (size_t)chain * (size_t)p.wi_stride;
    const double *A = p.A + (size_t)chain * (size_t)p.wi_stride;
    double *gbar_g = p.gbar + (size_t)chain * SAC_MAX_N_TAU;
    const char *Kb = reinterpret_cast<const char *>(p.K);
    const size_t col_bytes = (size_t)p.ntp * 8u;
    uint32_t off[NTL];
    double gbar[NTL], G[NTL], sinv[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        off[t] = (uint32_t)((j < p.ntp) ? j : p.ntp - 1) * 8u;
        gbar[t] = gbar_g[j];
        G[t] = p.G[j];
        sinv[t] = p.sinv[j];
    }
    double chi2 = p.sc[chain].chi2, chi2_min = p.sc[chain].chi2_min;
    for (long long i = 0; i < n; ++i) {
        const ReplayProp pr = props[i];
        double gn[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) gn[t] = gbar[t];
        for (int e = 0; e < pr.move; ++e) {
            const double a = p.uniform_A ? p.A_uniform : A[pr.k[e]];
            const size_t cn = (size_t)pr.w_new[e] * col_bytes, co = (size_t)wi[pr.k[e]] * col_bytes;
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                const double kn = *reinterpret_cast<const double *>(Kb + cn + off[t]);
                const double ko = *reinterpret_cast<const double *>(Kb + co + off[t]);
                gn[t] = gn[t] + a * (kn - ko);
            }
        }
        double term = 0.0;
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            const double d = (gn[t] - G[t]) * sinv[t];
            term = (t == 0) ? d * d : term + d * d;
        }
        const double chi2_new = wave_sum_uniform(term);
        const bool acc = metropolis(chi2, chi2_new, p.inv2theta, log_u01(pr.u));
        if (acc) {
#pragma unroll
            for (int t = 0; t < NTL; ++t) gbar[t] = gn[t];
            if (lane == 0)
                for (int e = 0; e < pr.move; ++e) wi[pr.k[e]] = (uint32_t)pr.w_new[e];
            chi2 = chi2_new;
            if (chi2_new < chi2_min) chi2_min = chi2_new;
            __syncthreads();
        }
        if (lane == 0) { chi2_out[i] = chi2_new; acc_out[i] = acc ? 1 : 0; }
    }
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        if (j < p.n_tau) gbar_g[j] = gbar[t];
    }
    if (lane == 0) { p.sc[chain].chi2 = chi2; p.sc[chain].chi2_min = chi2_min; }
}

// The same for the peak parameterisation: a proposal touches up to four deltas of the chain's state array (peak+, cont+,
// peak-, cont-; the index selects the axis and with it the table Kp / Kn).  transfer = 0: frequency moves, amplitudes
// stay (peak:415-423, 486-494, 573-585, 666-673: Gbar + A (K_new - K_old) per delta); transfer = 1: weight transfer
// between the axes, pairs (k[0], k[1]) [and (k[2], k[3])] = (+ axis, - axis) with new amplitudes (peak:758-764, 957-967:
// (((Gbar + A'_p K'_p) + A'_n K'_n) - A_p K_p) - A_n K_n per pair).
struct PeakReplayProp { int n; int transfer; int k[4]; long long w_new[4]; double A_new[4]; double u; };

template <int NTL>
__global__ void __launch_bounds__(64) peak_replay_kernel(const PeakParams pp, int chain, const PeakReplayProp *props, long long n,
                                                         double *chi2_out, int *acc_out)
{
    const FreeParams &p = pp.f;
    const int lane = lane_id();
    uint32_t *wi = p.wi + (size_t)chain * (size_t)p.wi_stride;
    double *A = p.A + (size_t)chain * (size_t)p.wi_stride;
    double *gbar_g = p.gbar + (size_t)chain * SAC_MAX_N_TAU;
    const char *Kb[2] = {reinterpret_cast<const char *>(p.K), reinterpret_cast<const char *>(pp.Kn)};
    const size_t col_bytes = (size_t)p.ntp * 8u;
    const int n_axis = pp.n_p + pp.n_cont;
    uint32_t off[NTL];
    double gbar[NTL], G[NTL], sinv[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        off[t] = (uint32_t)((j < p.ntp) ? j : p.ntp - 1) * 8u;
        gbar[t] = gbar_g[j];
        G[t] = p.G[j];
        sinv[t] = p.sinv[j];
    }
    double chi2 = pp.psc[chain].chi2, chi2_min = pp.psc[chain].chi2_min;
    for (long long i = 0; i < n; ++i) {
        const PeakReplayProp pr = props[i];
        double gn[NTL], fn[4][NTL], fo[4][NTL], a_old[4];
#pragma unroll
        for (int t = 0; t < NTL; ++t) gn[t] = gbar[t];
        for (int e = 0; e < pr.n; ++e) {
            const char *tab = Kb[pr.k[e] >= n_axis ? 1 : 0];
            a_old[e] = A[pr.k[e]];
            const size_t cn = (size_t)pr.w_new[e] * col_bytes, co = (size_t)wi[pr.k[e]] * col_bytes;
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                fn[e][t] = *reinterpret_cast<const double *>(tab + cn + off[t]);
                fo[e][t] = *reinterpret_cast<const double *>(tab + co + off[t]);
            }
        }
        if (!pr.transfer) {
            for (int e = 0; e < pr.n; ++e)
#pragma unroll
                for (int t = 0; t < NTL; ++t) gn[t] = gn[t] + a_old[e] * (fn[e][t] - fo[e][t]);
        } else {
            for (int q = 0; q < pr.n; q += 2)
#pragma unroll
                for (int t = 0; t < NTL; ++t)
                    gn[t] = (((gn[t] + (pr.A_new[q] * fn[q][t])) + (pr.A_new[q + 1] * fn[q + 1][t])) - (a_old[q] * fo[q][t]))
                            - (a_old[q + 1] * fo[q + 1][t]);
        }
        double term = 0.0;
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            const double d = (gn[t] - G[t]) * sinv[t];
            term = (t == 0) ? d * d : term + d * d;
        }
        const double chi2_new = wave_sum_uniform(term);
        const bool acc = metropolis(chi2, chi2_new, p.inv2theta, log_u01(pr.u));
        if (acc) {
#pragma unroll
            for (int t = 0; t < NTL; ++t) gbar[t] = gn[t];
            if (lane == 0)
                for (int e = 0; e < pr.n; ++e) {
                    wi[pr.k[e]] = (uint32_t)pr.w_new[e];
                    if (pr.transfer) A[pr.k[e]] = pr.A_new[e];
                }
            chi2 = chi2_new;
            if (chi2_new < chi2_min) chi2_min = chi2_new;
            __syncthreads();
        }
        if (lane == 0) { chi2_out[i] = chi2_new; acc_out[i] = acc ? 1 : 0; }
    }
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int j = lane + 64 * t;
        if (j < p.n_tau) gbar_g[j] = gbar[t];
    }
    if (lane == 0) { pp.psc[chain].chi2 = chi2; pp.psc[chain].chi2_min = chi2_min; }
}

// chi^2 of a given Gbar (sac_set_gbar consistency helper is host side; this kernel sums replicas)
__global__ void hist_sum_kernel(const double *hist, long long n_grid, int copies, double *dst)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_grid) return;
    double s = 0.0;
    for (int c = 0; c < copies; ++c) s = s + hist[(size_t)c * (size_t)n_grid + (size_t)i];
    dst[i] = s;
}

// dst[t] = sum over chains of Gbar[c][t], chains in index order (sac_get_gbar_mean; the merge payload)
__global__ void gbar_sum_kernel(const double *gbar, int n_chains, double *dst)
{
    const int t = (int)threadIdx.x;
    if (t >= SAC_MAX_N_TAU) return;
    double s = 0.0;
    for (int c = 0; c < n_chains; ++c) s = s + gbar[(size_t)c * SAC_MAX_N_TAU + t];
    dst[t] = s;
}

// peak:1159-1165 per chain: mean edge of the sample stage, running sums of the means and of their squares
__global__ void peak_edge_rows_kernel(const PeakScalars *sc, double *rows, int n_chains, double steps, int sample_neg)
{
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= n_chains) return;
    double *e = rows + (size_t)c * 6;
    for (int pn = 0; pn < 2; ++pn) {
        e[0 + pn] = sc[c].edge_sum[pn] / steps;
        e[2 + pn] += e[0 + pn];
        e[4 + pn] += e[0 + pn] * e[0 + pn];
    }
    if (!sample_neg) { e[1] = e[0]; e[3] = e[2]; e[5] = e[4]; }
}

// ---- per-stage merge over ranks (comm.h): this rank's sums over its chains, one thread per scalar slot, chains in
// index order (the order sac_get_stats uses on the host, so one rank alone reproduces it bit for bit) -------------------
constexpr int FREE_SLOTS = 20;      // <chi2>, rates[6], windows[4], sum chi2_min, counters[5], sweeps, chains, spare
constexpr int PEAK_SLOTS = 46;      // <chi2>, rates[11], windows[11], edge_res[6], bounds[4], weights[4], sum chi2_min,
                                    // counters[5], sweeps, chains, spare

__global__ void free_pack_kernel(const ChainScalars *sc, int n_chains, double steps, int sampled, double *out_sum, double *out_min)
{
    const int t = (int)threadIdx.x;
    if (t >= FREE_SLOTS) return;
    double s = 0.0, mn = sc[0].chi2_min;
    for (int c = 0; c < n_chains; ++c) {
        const ChainScalars &x = sc[c];
        double v = 0.0;
        if (t == 0) v = sampled ? x.sum_chi2 / steps : 0.0;
        else if (t <= 6) v = sampled ? x.acc[t - 1] / steps : x.acc[t - 1];
        else if (t <= 10) v = x.windows[t - 7];
        else if (t == 11) { v = x.chi2_min; if (x.chi2_min < mn) mn = x.chi2_min; }
        else if (t <= 16) v = (double)x.counters[t - 12];
        else if (t == 17) v = (double)x.sweep;
        else if (t == 18) v = 1.0;
        s = s + v;
    }
    out_sum[t] = s;
    if (t == 11) out_min[0] = mn;
}

__global__ void peak_pack_kernel(const PeakScalars *sc, const double *edge_rows, const double *weights, int n_chains, double steps,
                                 int sampled, double *out_sum, double *out_min)
{
    const int t = (int)threadIdx.x;
    if (t >= PEAK_SLOTS) return;
    double s = 0.0, mn = sc[0].chi2_min;
    for (int c = 0; c < n_chains; ++c) {
        const PeakScalars &x = sc[c];
        double v = 0.0;
        if (t == 0) v = sampled ? x.sum_chi2 / steps : 0.0;
        else if (t <= 11) v = sampled ? x.acc[t - 1] / steps : x.acc[t - 1];
        else if (t <= 22) v = x.windows[t - 12];
        else if (t <= 28) v = edge_rows[(size_t)c * 6 + (t - 23)];
        else if (t <= 32) v = (double)x.bounds[t - 29];
        else if (t <= 36) v = weights[(size_t)c * 4 + (t - 33)];
        else if (t == 37) { v = x.chi2_min; if (x.chi2_min < mn) mn = x.chi2_min; }
        else if (t <= 42) v = (double)x.counters[t - 38];
        else if (t == 43) v = (double)x.sweep;
        else if (t == 44) v = 1.0;
        s = s + v;
    }
    out_sum[t] = s;
    if (t == 37) out_min[0] = mn;
}

// =========================================================================================
// Host side
// =========================================================================================

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// for the library's other translation units (sac_edge.hip): same thread-local message buffer
int sac_internal_fail(int code, const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(SAC_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                              \
    } while (0)

// ---- scratch of sac_build_table_noalloc ---------------------------------------------------------
// Staging area for tau[n_tau] and U[n_tau][n_tau] (what sac_build_table puts into two hipMalloc'ed buffers): one
// file-scope device array per GPU, 132 KB, held until the process ends.  Host code never uses the array's name as
// an address: sac_create resolves it once per device (hipGetSymbolAddress with that device current) and every
// later use goes through the resolved pointer.
constexpr size_t SCRATCH_DOUBLES = (size_t)SAC_MAX_N_TAU + (size_t)SAC_MAX_N_TAU * SAC_MAX_N_TAU;
static __device__ __align__(256) double scratch[SCRATCH_DOUBLES];

constexpr int MAX_DEVICES = 64;
static std::mutex g_scratch_mutex;
static double *g_scratch_ptr[MAX_DEVICES] = {};
static std::mutex g_scratch_use[MAX_DEVICES];   // one user of a GPU's scratch array at a time (all handles on it share the array)

// The calling thread's current device must be `device`.
static int resolve_scratch(int device, double **out)
{
    if (device < 0 || device >= MAX_DEVICES) return SAC_E_INVALID;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        double *ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) return SAC_E_HIP;
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return SAC_OK;
}

struct sac_handle {
    sac_config_t cfg{};
    int ntp = 0;
    int wi_stride = 0;
    int hist_copies = 1;
    bool hist_exclusive = false;
    int sweeps_per_launch = 100;
    int chains_per_wave = 0;       // sac_set_chains_per_wave: 0 = the library chooses
    int last_chains_per_wave = 1;  // what the last sweep launch used
    int proposals_per_step = 0;    // sac_set_proposals_per_step: 0 = the library chooses
    int last_proposals_per_step = 1;
    double *dK = nullptr, *dG = nullptr, *dSinv = nullptr, *dA = nullptr, *dGbar = nullptr;
    double *dHist = nullptr;
    uint32_t *dWi = nullptr;
    ChainScalars *dSc = nullptr;
    MoveRec *dLog = nullptr;
    unsigned long long *dLogLen = nullptr;
    long long log_cap = 0;
    int log_chain = -1;
    double *scratch = nullptr;     // this GPU's copy of the file-scope `scratch` array (resolved by sac_create)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kev;   // reusable per-launch event pairs
    bool problem = false, state_set = false, gbar_set = false;
    bool table_built[2] = {false, false};   // sac_build_table
    bool uniform_A = false;
    double A_uniform = 0.0;
    // layout of one chain's state as the caller sees it
    bool peak = false;
    int n_state = 0;               // positions / amplitudes per chain
    int n_windows = 4;
    int64_t pos_offset = 0;        // caller position - stored position
    int hist_arrays = 1;
    // peak only
    double *dKn = nullptr;
    PeakScalars *dPsc = nullptr;
    double *dWeights = nullptr;    // [n_chains][4] group weight sums
    bool sample_neg = false, indiv_update = false, binned = false;
    std::vector<double> edge_rows; // [n_chains][3][2]: peak:74 edge_res per chain (host mirror of dEdgeRows)
    double *dEdgeRows = nullptr;   // the same on the device (peak_edge_rows_kernel)
    double *dGbarSum = nullptr;    // [128] sac_get_gbar_mean without a communicator
    saccomm::Comm *comm = nullptr; // sac_comm_init: per-stage merge over ranks
    int last_stage = 0;            // 0 none, 1 adjust_windows, 2 sample
    int64_t last_steps = 0;
    double last_stage_ms = 0.0, last_kernel_ms = 0.0;
    double ms_per_sweep = 0.0;     // measured kernel time per sweep of the last stage (priority rotation period)
    int64_t last_launches = 0;
};

extern "C" const char *sac_last_error(void) { return g_err; }

#ifndef SAC_BUILD_ID
#define SAC_BUILD_ID "unknown"
#endif
extern "C" const char *sac_build_id(void) { return SAC_BUILD_ID; }

// host code changed the chains: the merged view of the last stage (comm_merge) no longer describes them
static void drop_merged(sac_handle *h)
{
    if (h->comm) h->comm->merged = false;
}

static int use_device(sac_handle *h)
{
    HIP_TRY(hipSetDevice(h->cfg.device));
    return SAC_OK;
}

extern "C" int sac_create(const sac_config_t *cfg, sac_handle_t **out)
{
    if (!cfg || !out) return fail(SAC_E_INVALID, "sac_create: null argument");
    *out = nullptr;
    if (cfg->kind != SAC_KIND_FREE && cfg->kind != SAC_KIND_PEAK)
        return fail(SAC_E_INVALID, "sac_create: unsupported kind %d", cfg->kind);
    const bool peak = (cfg->kind == SAC_KIND_PEAK);
    if (!peak && (cfg->par < 1 || cfg->par > 3))
        return fail(SAC_E_INVALID, "sac_create: par=%d (1, 2 or 3)", cfg->par);
    const bool sample_neg = peak && !(cfg->symm == 1 || cfg->bosonic);
    if (peak) {
        if (cfg->n_peak < 1) return fail(SAC_E_INVALID, "sac_create: n_peak=%d", cfg->n_peak);
        const long long used = (sample_neg ? 2ll : 1ll) * ((long long)cfg->n_omega + cfg->n_peak);
        if (used > SAC_MAX_PEAK_STATE)
            return fail(SAC_E_INVALID, "sac_create: %lld sampled deltas per chain exceed %d", used, SAC_MAX_PEAK_STATE);
        if (cfg->wi_0 < 0 || cfg->n_grid != cfg->wi_m + 1)
            return fail(SAC_E_INVALID, "sac_create: peak needs wi_0 >= 0 and n_grid == wi_m + 1");
    }
    if (cfg->n_tau < 1 || cfg->n_tau > SAC_MAX_N_TAU)
        return fail(SAC_E_INVALID, "sac_create: n_tau=%d outside 1..%d", cfg->n_tau, SAC_MAX_N_TAU);
    if (cfg->n_omega < 3 || cfg->n_omega > SAC_MAX_N_OMEGA)
        return fail(SAC_E_INVALID, "sac_create: n_omega=%d outside 3..%d", cfg->n_omega, SAC_MAX_N_OMEGA);
    if (cfg->n_chains < 1) return fail(SAC_E_INVALID, "sac_create: n_chains=%d", cfg->n_chains);
    if (cfg->n_grid < 1 || cfg->n_grid > (1ll << 30))
        return fail(SAC_E_INVALID, "sac_create: n_grid=%lld outside 1..2^30", (long long)cfg->n_grid);
    if ((unsigned long long)cfg->n_grid * (unsigned long long)((cfg->n_tau + 7) / 8 * 8) * 8ull >= (1ull << 32))
        return fail(SAC_E_INVALID, "sac_create: a table of %lld columns x %d time points reaches 4 GiB (columns are addressed "
                    "with 32-bit byte offsets)", (long long)cfg->n_grid, cfg->n_tau);
    if (cfg->wi_m < cfg->wi_0 || cfg->wi_m - cfg->wi_0 + 1 > cfg->n_grid)
        return fail(SAC_E_INVALID, "sac_create: grid bounds [%lld,%lld] do not fit n_grid=%lld",
                    (long long)cfg->wi_0, (long long)cfg->wi_m, (long long)cfg->n_grid);
    if (cfg->hist_mode != SAC_HIST_REFERENCE && cfg->hist_mode != SAC_HIST_ACCUMULATE)
        return fail(SAC_E_INVALID, "sac_create: hist_mode=%d", cfg->hist_mode);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(SAC_E_INVALID, "sac_create: device %d of %d", cfg->device, ndev);

    sac_handle *h = new (std::nothrow) sac_handle();
    if (!h) return fail(SAC_E_NOMEM, "sac_create: out of host memory");
    h->cfg = *cfg;
    h->ntp = (cfg->n_tau + 7) / 8 * 8;
    h->peak = peak;
    h->sample_neg = sample_neg;
    h->n_state = peak ? 2 * (cfg->n_omega + cfg->n_peak) : cfg->n_omega;
    h->n_windows = peak ? 11 : 4;
    h->pos_offset = peak ? 0 : cfg->wi_0;
    h->hist_arrays = peak ? 4 : 1;
    h->wi_stride = (h->n_state + 63) / 64 * 64;
    int copies = cfg->hist_copies;
    if (copies <= 0) copies = cfg->n_chains < 16 ? cfg->n_chains : 16;
    if (copies > cfg->n_chains) copies = cfg->n_chains;
    h->hist_copies = copies;
    h->hist_exclusive = (copies == cfg->n_chains);
    if (cfg->sweeps_per_launch > 0) h->sweeps_per_launch = cfg->sweeps_per_launch;

    int rc = use_device(h);
    if (rc) { delete h; return rc; }
    if ((rc = resolve_scratch(cfg->device, &h->scratch))) {
        delete h;
        return fail(rc, "sac_create: hipGetSymbolAddress(scratch) failed on device %d", cfg->device);
    }
    const size_t nc = (size_t)cfg->n_chains;
#define ALLOC(ptr, bytes)                                                           \
    do {                                                                            \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes));                        \
        if (e_ != hipSuccess) {                                                     \
            sac_destroy(h);                                                         \
            return fail(SAC_E_NOMEM, "hipMalloc(%zu bytes) for %s failed: %s",      \
                        (size_t)(bytes), #ptr, hipGetErrorString(e_));              \
        }                                                                           \
        hipMemset((ptr), 0, (bytes));                                               \
    } while (0)
    ALLOC(h->dK, (size_t)cfg->n_grid * h->ntp * sizeof(double));
    ALLOC(h->dG, SAC_MAX_N_TAU * sizeof(double));
    ALLOC(h->dSinv, SAC_MAX_N_TAU * sizeof(double));
    ALLOC(h->dWi, nc * h->wi_stride * sizeof(uint32_t));
    ALLOC(h->dA, nc * h->wi_stride * sizeof(double));
    ALLOC(h->dGbar, nc * SAC_MAX_N_TAU * sizeof(double));
    ALLOC(h->dSc, nc * sizeof(ChainScalars));
    ALLOC(h->dHist, (size_t)copies * (size_t)h->hist_arrays * (size_t)cfg->n_grid * sizeof(double));
    if (peak) {
        ALLOC(h->dKn, (size_t)cfg->n_grid * h->ntp * sizeof(double));
        ALLOC(h->dPsc, nc * sizeof(PeakScalars));
        ALLOC(h->dWeights, nc * 4 * sizeof(double));
        h->edge_rows.assign(nc * 6, 0.0);
        ALLOC(h->dEdgeRows, nc * 6 * sizeof(double));
    }
    ALLOC(h->dLogLen, sizeof(unsigned long long));
    ALLOC(h->dGbarSum, SAC_MAX_N_TAU * sizeof(double));
#undef ALLOC
    if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess ||
        hipEventCreate(&h->ev1) != hipSuccess) {
        sac_destroy(h);
        return fail(SAC_E_HIP, "sac_create: stream/event creation failed");
    }
    HIP_TRY(hipDeviceSynchronize());
    *out = h;
    return SAC_OK;
}

extern "C" int sac_destroy(sac_handle_t *h)
{
    if (!h) return SAC_OK;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    hipFree(h->dK); hipFree(h->dG); hipFree(h->dSinv); hipFree(h->dWi); hipFree(h->dA);
    hipFree(h->dGbar); hipFree(h->dSc); hipFree(h->dHist); hipFree(h->dLog); hipFree(h->dLogLen);
    hipFree(h->dKn); hipFree(h->dPsc); hipFree(h->dWeights); hipFree(h->dEdgeRows); hipFree(h->dGbarSum);
    saccomm::destroy(h->comm);
    h->comm = nullptr;
    for (auto &pr : h->kev) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return SAC_OK;
}

extern "C" int sac_upload_problem(sac_handle_t *h, const double *K, const double *G, const double *sigma_inv)
{
    if (!h || !K || !G || !sigma_inv) return fail(SAC_E_INVALID, "sac_upload_problem: null argument");
    int rc = use_device(h);
    if (rc) return rc;
    const size_t nt = (size_t)h->cfg.n_tau, ng = (size_t)h->cfg.n_grid;
    for (size_t j = 0; j < nt; ++j)
        if (!std::isfinite(G[j]) || !std::isfinite(sigma_inv[j]))
            return fail(SAC_E_INVALID, "sac_upload_problem: non-finite G or sigma_inv at tau index %zu", j);
    if ((size_t)h->ntp == nt) {
        HIP_TRY(copy_sync(h->stream, h->dK, K, ng * nt * sizeof(double), hipMemcpyHostToDevice));
    } else {
        HIP_TRY(hipMemcpy2D(h->dK, (size_t)h->ntp * sizeof(double), K, nt * sizeof(double),
                            nt * sizeof(double), ng, hipMemcpyHostToDevice));
    }
    HIP_TRY(copy_sync(h->stream, h->dG, G, nt * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(copy_sync(h->stream, h->dSinv, sigma_inv, nt * sizeof(double), hipMemcpyHostToDevice));
    h->problem = true;
    h->gbar_set = false;            // Gbar / chi2 of the old problem: sac_init_gbar (or sac_set_gbar) must follow
    return SAC_OK;
}

extern "C" int sac_upload_data(sac_handle_t *h, const double *G, const double *sigma_inv)
{
    if (!h || !G || !sigma_inv) return fail(SAC_E_INVALID, "sac_upload_data: null argument");
    if (!h->table_built[0] || (h->peak && h->sample_neg && !h->table_built[1]))
        return fail(SAC_E_STATE, "sac_upload_data: build the table(s) first (sac_build_table)");
    int rc = use_device(h);
    if (rc) return rc;
    const size_t nt = (size_t)h->cfg.n_tau;
    for (size_t j = 0; j < nt; ++j)
        if (!std::isfinite(G[j]) || !std::isfinite(sigma_inv[j]))
            return fail(SAC_E_INVALID, "sac_upload_data: non-finite G or sigma_inv at tau index %zu", j);
    HIP_TRY(copy_sync(h->stream, h->dG, G, nt * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(copy_sync(h->stream, h->dSinv, sigma_inv, nt * sizeof(double), hipMemcpyHostToDevice));
    h->problem = true;
    h->gbar_set = false;
    return SAC_OK;
}

extern "C" int sac_build_table(sac_handle_t *h, int32_t which, const sac_table_spec_t *spec, const double *tau, const double *U)
{
    if (!h || !spec || !tau || !U) return fail(SAC_E_INVALID, "sac_build_table: null argument");
    if (which < 0 || which > 1 || (which == 1 && !h->dKn))
        return fail(SAC_E_INVALID, "sac_build_table: table %d does not exist on this handle", which);
    if (spec->kernel_type < 0 || spec->kernel_type > 2)
        return fail(SAC_E_INVALID, "sac_build_table: Invalid Kernel type.");           // free:195
    if (!(spec->d_omega > 0) || !std::isfinite(spec->beta) || !std::isfinite(spec->offset))
        return fail(SAC_E_INVALID, "sac_build_table: d_omega=%g beta=%g offset=%g", spec->d_omega, spec->beta, spec->offset);
    int rc = use_device(h);
    if (rc) return rc;
    const size_t nt = (size_t)h->cfg.n_tau;
    for (size_t j = 0; j < nt; ++j)
        if (!std::isfinite(tau[j])) return fail(SAC_E_INVALID, "sac_build_table: non-finite tau[%zu]", j);
    double *d_tau = nullptr, *d_U = nullptr;
    HIP_TRY(hipMalloc(&d_tau, nt * sizeof(double)));
    if (hipMalloc(&d_U, nt * nt * sizeof(double)) != hipSuccess) { (void)hipFree(d_tau); return fail(SAC_E_HIP, "sac_build_table: out of device memory"); }
    hipError_t e = copy_sync(h->stream, d_tau, tau, nt * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = copy_sync(h->stream, d_U, U, nt * nt * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        TableSpec sp;
        sp.kernel_type = spec->kernel_type; sp.n_tau = (int)nt; sp.mirror = spec->mirror; sp.negate = spec->negate;
        sp.i_first = spec->i_first; sp.offset = spec->offset; sp.d_omega = spec->d_omega; sp.beta = spec->beta;
        sp.n_grid = h->cfg.n_grid; sp.ld = h->ntp; sp.n_cols = h->ntp;
        e = launch_table_build(h->stream, sp, d_tau, d_U, which == 0 ? h->dK : h->dKn);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    (void)hipFree(d_tau);
    (void)hipFree(d_U);
    if (e != hipSuccess) return fail(SAC_E_HIP, "sac_build_table: %s", hipGetErrorString(e));
    h->table_built[which] = true;
    h->gbar_set = false;
    return SAC_OK;
}

// sac_build_table for callers that cannot allocate: tau and U are staged in this GPU's `scratch` array instead of
// two hipMalloc'ed buffers.  No hipMalloc / hipFree / allocating library call on this path.  The scratch array is
// shared by every handle on the GPU and keeps the last call's contents: calls that overlap on one GPU race.
extern "C" int sac_build_table_noalloc(sac_handle_t *h, int32_t which, const sac_table_spec_t *spec, const double *tau,
                                       const double *U)
{
    if (!h || !spec || !tau || !U) return fail(SAC_E_INVALID, "sac_build_table_noalloc: null argument");
    if (which < 0 || which > 1 || (which == 1 && !h->dKn))
        return fail(SAC_E_INVALID, "sac_build_table_noalloc: table %d does not exist on this handle", which);
    if (spec->kernel_type < 0 || spec->kernel_type > 2)
        return fail(SAC_E_INVALID, "sac_build_table_noalloc: Invalid Kernel type.");   // free:195
    if (!(spec->d_omega > 0) || !std::isfinite(spec->beta) || !std::isfinite(spec->offset))
        return fail(SAC_E_INVALID, "sac_build_table_noalloc: d_omega=%g beta=%g offset=%g", spec->d_omega, spec->beta, spec->offset);
    const size_t nt = (size_t)h->cfg.n_tau;
    if (nt > (size_t)SAC_MAX_N_TAU || nt + nt * nt > SCRATCH_DOUBLES)
        return fail(SAC_E_INVALID, "sac_build_table_noalloc: n_tau=%zu exceeds the scratch array (n_tau <= %d, %zu bytes)", nt,
                    SAC_MAX_N_TAU, SCRATCH_DOUBLES * sizeof(double));
    if (!h->scratch) return fail(SAC_E_STATE, "sac_build_table_noalloc: scratch array not resolved on device %d", h->cfg.device);
    int rc = use_device(h);
    if (rc) return rc;
    for (size_t j = 0; j < nt; ++j)
        if (!std::isfinite(tau[j])) return fail(SAC_E_INVALID, "sac_build_table_noalloc: non-finite tau[%zu]", j);
    double *d_tau = h->scratch, *d_U = h->scratch + SAC_MAX_N_TAU;
    HIP_TRY(hipStreamSynchronize(h->stream));
    // held from the first copy into the array to the end of the kernel that reads it: two host threads building tables
    // on one GPU (through any handles) take turns instead of overwriting each other's tau / U
    std::lock_guard<std::mutex> scratch_user(g_scratch_use[h->cfg.device]);
    HIP_TRY(copy_sync(h->stream, d_tau, tau, nt * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(copy_sync(h->stream, d_U, U, nt * nt * sizeof(double), hipMemcpyHostToDevice));
    TableSpec sp;
    sp.kernel_type = spec->kernel_type; sp.n_tau = (int)nt; sp.mirror = spec->mirror; sp.negate = spec->negate;
    sp.i_first = spec->i_first; sp.offset = spec->offset; sp.d_omega = spec->d_omega; sp.beta = spec->beta;
    sp.n_grid = h->cfg.n_grid; sp.ld = h->ntp; sp.n_cols = h->ntp;
    HIP_TRY(launch_table_build(h->stream, sp, d_tau, d_U, which == 0 ? h->dK : h->dKn));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->table_built[which] = true;
    h->gbar_set = false;
    return SAC_OK;
}

extern "C" int sac_get_table(sac_handle_t *h, int32_t which, double *out)
{
    if (!h || !out) return fail(SAC_E_INVALID, "sac_get_table: null argument");
    if (which < 0 || which > 1 || (which == 1 && !h->dKn))
        return fail(SAC_E_INVALID, "sac_get_table: table %d does not exist on this handle", which);
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t nt = (size_t)h->cfg.n_tau, ng = (size_t)h->cfg.n_grid;
    HIP_TRY(hipMemcpy2D(out, nt * sizeof(double), which == 0 ? h->dK : h->dKn, (size_t)h->ntp * sizeof(double),
                        nt * sizeof(double), ng, hipMemcpyDeviceToHost));
    return SAC_OK;
}

static int chain_range(sac_handle *h, int32_t chain, int *first, int *count, const char *who)
{
    if (chain == -1) { *first = 0; *count = h->cfg.n_chains; return SAC_OK; }
    if (chain < 0 || chain >= h->cfg.n_chains)
        return fail(SAC_E_INVALID, "%s: chain %d outside 0..%d", who, chain, h->cfg.n_chains - 1);
    *first = chain; *count = 1;
    return SAC_OK;
}

extern "C" int sac_set_state(sac_handle_t *h, int32_t chain, const int64_t *wi, const double *A,
                             const double *windows)
{
    if (!h) return fail(SAC_E_INVALID, "sac_set_state: null handle");
    int first, count, rc;
    if ((rc = chain_range(h, chain, &first, &count, "sac_set_state"))) return rc;
    if ((rc = use_device(h))) return rc;
    drop_merged(h);
    const int n = h->n_state;
    if (wi) {
        std::vector<uint32_t> rel((size_t)h->wi_stride, 0u);
        const int64_t lowest = h->peak ? 0 : h->cfg.wi_0;
        for (int i = 0; i < n; ++i) {
            if (wi[i] < lowest || wi[i] > h->cfg.wi_m)
                return fail(SAC_E_INVALID, "sac_set_state: omega_i[%d]=%lld outside [%lld,%lld]", i,
                            (long long)wi[i], (long long)lowest, (long long)h->cfg.wi_m);
            rel[i] = (uint32_t)(wi[i] - h->pos_offset);
        }
        std::vector<uint32_t> all((size_t)count * (size_t)h->wi_stride);
        for (int c = 0; c < count; ++c)
            std::memcpy(all.data() + (size_t)c * h->wi_stride, rel.data(), (size_t)h->wi_stride * sizeof(uint32_t));
        HIP_TRY(copy_sync(h->stream, h->dWi + (size_t)first * h->wi_stride, all.data(), all.size() * sizeof(uint32_t),
                          hipMemcpyHostToDevice));
        h->state_set = true;
        h->gbar_set = false;        // the kernels never rebuild Gbar from the positions (free:876 assigns a dead local)
    }
    if (A) {
        bool uni = true;
        for (int i = 0; i < n; ++i) {
            if (!std::isfinite(A[i])) return fail(SAC_E_INVALID, "sac_set_state: non-finite A[%d]", i);
            uni = uni && (A[i] == A[0]);
        }
        std::vector<double> pad((size_t)h->wi_stride, 0.0);
        std::memcpy(pad.data(), A, (size_t)n * sizeof(double));
        std::vector<double> all((size_t)count * (size_t)h->wi_stride);
        for (int c = 0; c < count; ++c)
            std::memcpy(all.data() + (size_t)c * h->wi_stride, pad.data(), (size_t)h->wi_stride * sizeof(double));
        HIP_TRY(copy_sync(h->stream, h->dA + (size_t)first * h->wi_stride, all.data(), all.size() * sizeof(double),
                          hipMemcpyHostToDevice));
        if (chain == -1 || h->cfg.n_chains == 1) {
            h->uniform_A = uni;
            h->A_uniform = A[0];
        } else if (!(h->uniform_A && uni && A[0] == h->A_uniform)) {
            h->uniform_A = false;
        }
        h->gbar_set = false;
    }
    if (windows && h->peak) {
        std::vector<PeakScalars> sc((size_t)count);
        HIP_TRY(copy_sync(h->stream, sc.data(), h->dPsc + first, (size_t)count * sizeof(PeakScalars), hipMemcpyDeviceToHost));
        for (auto &s : sc)
            for (int k = 0; k < 11; ++k) s.windows[k] = windows[k];
        HIP_TRY(copy_sync(h->stream, h->dPsc + first, sc.data(), (size_t)count * sizeof(PeakScalars), hipMemcpyHostToDevice));
    } else if (windows) {
        std::vector<ChainScalars> sc((size_t)count);
        HIP_TRY(copy_sync(h->stream, sc.data(), h->dSc + first, (size_t)count * sizeof(ChainScalars), hipMemcpyDeviceToHost));
        for (auto &s : sc)
            for (int k = 0; k < 4; ++k) s.windows[k] = windows[k];
        HIP_TRY(copy_sync(h->stream, h->dSc + first, sc.data(), (size_t)count * sizeof(ChainScalars), hipMemcpyHostToDevice));
    }
    return SAC_OK;
}

extern "C" int sac_get_state(sac_handle_t *h, int32_t chain, int64_t *wi, double *A, double *windows)
{
    if (!h) return fail(SAC_E_INVALID, "sac_get_state: null handle");
    if (chain < 0 || chain >= h->cfg.n_chains)
        return fail(SAC_E_INVALID, "sac_get_state: chain %d outside 0..%d", chain, h->cfg.n_chains - 1);
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int n = h->n_state;
    if (wi) {
        std::vector<uint32_t> rel((size_t)n);
        HIP_TRY(copy_sync(h->stream, rel.data(), h->dWi + (size_t)chain * h->wi_stride, (size_t)n * sizeof(uint32_t),
                          hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) wi[i] = (int64_t)rel[i] + h->pos_offset;
    }
    if (A)
        HIP_TRY(copy_sync(h->stream, A, h->dA + (size_t)chain * h->wi_stride, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (windows && h->peak) {
        PeakScalars s;
        HIP_TRY(copy_sync(h->stream, &s, h->dPsc + chain, sizeof(s), hipMemcpyDeviceToHost));
        for (int k = 0; k < 11; ++k) windows[k] = s.windows[k];
    } else if (windows) {
        ChainScalars s;
        HIP_TRY(copy_sync(h->stream, &s, h->dSc + chain, sizeof(s), hipMemcpyDeviceToHost));
        for (int k = 0; k < 4; ++k) windows[k] = s.windows[k];
    }
    return SAC_OK;
}

static void fill_params(sac_handle *h, FreeParams &p)
{
    std::memset(&p, 0, sizeof(p));
    p.K = h->dK; p.G = h->dG; p.sinv = h->dSinv;
    p.wi = h->dWi; p.A = h->dA; p.gbar = h->dGbar; p.sc = h->dSc; p.hist = h->dHist;
    p.log = h->dLog; p.log_len = h->dLogLen; p.log_cap = h->log_cap; p.log_chain = h->log_chain;
    p.n_grid = h->cfg.n_grid;
    p.A_uniform = h->A_uniform;
    p.uniform_A = h->uniform_A ? 1 : 0;
    p.ntp = h->ntp; p.n_tau = h->cfg.n_tau; p.n_omega = h->cfg.n_omega; p.wi_stride = h->wi_stride;
    p.gmax = (int)(h->cfg.wi_m - h->cfg.wi_0);
    p.n_chains = h->cfg.n_chains;
    p.chain_offset = h->cfg.chain_offset;
    p.seed_lo = (uint32_t)h->cfg.seed; p.seed_hi = (uint32_t)(h->cfg.seed >> 32);
    p.hist_copies = h->hist_copies; p.hist_exclusive = h->hist_exclusive ? 1 : 0;
    p.hist_mode = h->cfg.hist_mode;
    p.par = h->cfg.par;
    p.symm = h->cfg.symm;
}

static unsigned grid_blocks(const sac_handle *h)
{
    return (unsigned)((h->cfg.n_chains + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
}

static int comm_merge(sac_handle *h);
__global__ void peak_weights_kernel(const double *A, int stride, int n_chains, int n_p, int n, double *out);
static int peak_init(sac_handle *h);
static int peak_adjust_windows(sac_handle *h, int64_t steps, double theta);
static int peak_sample(sac_handle *h, int64_t steps, double theta, bool bin);
static int peak_chain_stats(sac_handle *h, int32_t chain, double *out6, double *accept_rates);

extern "C" int sac_init_gbar(sac_handle_t *h)
{
    if (!h) return fail(SAC_E_INVALID, "sac_init_gbar: null handle");
    if (!h->problem || !h->state_set)
        return fail(SAC_E_STATE, "sac_init_gbar: upload the problem and set the state first");
    int rc;
    if ((rc = use_device(h))) return rc;
    if (h->peak) return peak_init(h);
    FreeParams p;
    fill_params(h, p);
    if (h->cfg.n_tau <= 64)
        hipLaunchKernelGGL(init_gbar_kernel<1>, dim3(grid_blocks(h)), dim3(64 * WAVES_PER_BLOCK), 0, h->stream, p);
    else
        hipLaunchKernelGGL(init_gbar_kernel<2>, dim3(grid_blocks(h)), dim3(64 * WAVES_PER_BLOCK), 0, h->stream, p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->gbar_set = true;
    h->last_stage = 0;
    return comm_merge(h);
}

extern "C" int sac_get_gbar(sac_handle_t *h, int32_t chain, double *gbar, double *chi2, double *chi2_min)
{
    if (!h) return fail(SAC_E_INVALID, "sac_get_gbar: null handle");
    if (chain < 0 || chain >= h->cfg.n_chains)
        return fail(SAC_E_INVALID, "sac_get_gbar: chain %d outside 0..%d", chain, h->cfg.n_chains - 1);
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (gbar)
        HIP_TRY(copy_sync(h->stream, gbar, h->dGbar + (size_t)chain * SAC_MAX_N_TAU, (size_t)h->cfg.n_tau * sizeof(double),
                          hipMemcpyDeviceToHost));
    if (h->peak) {
        PeakScalars s;
        HIP_TRY(copy_sync(h->stream, &s, h->dPsc + chain, sizeof(s), hipMemcpyDeviceToHost));
        if (chi2) *chi2 = s.chi2;
        if (chi2_min) *chi2_min = s.chi2_min;
        return SAC_OK;
    }
    ChainScalars s;
    HIP_TRY(copy_sync(h->stream, &s, h->dSc + chain, sizeof(s), hipMemcpyDeviceToHost));
    if (chi2) *chi2 = s.chi2;
    if (chi2_min) *chi2_min = s.chi2_min;
    return SAC_OK;
}

extern "C" int sac_get_gbar_mean(sac_handle_t *h, double *gbar_mean)
{
    if (!h || !gbar_mean) return fail(SAC_E_INVALID, "sac_get_gbar_mean: null argument");
    int rc;
    if ((rc = use_device(h))) return rc;
    const size_t nt = (size_t)h->cfg.n_tau;
    double n = (double)h->cfg.n_chains;
    if (h->comm && h->comm->merged && h->last_stage == 2) {           // summed over all ranks' chains (comm_merge)
        const size_t n_hist = (size_t)h->cfg.n_grid * (size_t)h->hist_arrays;
        HIP_TRY(copy_sync(h->stream, gbar_mean, h->comm->dSum + n_hist, nt * sizeof(double), hipMemcpyDeviceToHost));
        n = h->comm->hTail[h->peak ? 44 : 18];
    } else {
        hipLaunchKernelGGL(gbar_sum_kernel, dim3(1), dim3(SAC_MAX_N_TAU), 0, h->stream, (const double *)h->dGbar, h->cfg.n_chains,
                           h->dGbarSum);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(gbar_mean, h->dGbarSum, nt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    for (size_t t = 0; t < nt; ++t) gbar_mean[t] = gbar_mean[t] / n;
    return SAC_OK;
}

extern "C" int sac_set_gbar(sac_handle_t *h, int32_t chain, const double *gbar, double chi2, double chi2_min)
{
    if (!h || !gbar) return fail(SAC_E_INVALID, "sac_set_gbar: null argument");
    if (h->peak) return fail(SAC_E_INVALID, "sac_set_gbar: peak refreshes Gbar itself every sweep (peak:1072)");
    int first, count, rc;
    if ((rc = chain_range(h, chain, &first, &count, "sac_set_gbar"))) return rc;
    if ((rc = use_device(h))) return rc;
    drop_merged(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<double> pad(SAC_MAX_N_TAU, 0.0);
    std::memcpy(pad.data(), gbar, (size_t)h->cfg.n_tau * sizeof(double));
    std::vector<ChainScalars> sc((size_t)count);
    HIP_TRY(copy_sync(h->stream, sc.data(), h->dSc + first, (size_t)count * sizeof(ChainScalars), hipMemcpyDeviceToHost));
    for (int c = 0; c < count; ++c) {
        HIP_TRY(copy_sync(h->stream, h->dGbar + (size_t)(first + c) * SAC_MAX_N_TAU, pad.data(),
                          SAC_MAX_N_TAU * sizeof(double), hipMemcpyHostToDevice));
        sc[c].chi2 = chi2;
        sc[c].chi2_min = chi2_min;
    }
    HIP_TRY(copy_sync(h->stream, h->dSc + first, sc.data(), (size_t)count * sizeof(ChainScalars), hipMemcpyHostToDevice));
    h->gbar_set = true;
    return SAC_OK;
}

// Chains per wave of the next free sweep launch: the caller's choice (sac_set_chains_per_wave) when the shape allows
// it, else the library's.  Several chains share a wave only with equal fixed amplitudes (par = 1), a
// chain count that fills whole waves, N_tau <= 64 and <= 4 slots x the chain's lanes, and positions that fit the CU's LDS.
static int packed_chains_per_wave(const sac_handle *h)
{
    auto fits = [&](int c) {
        if (c == 1) return true;
        if (h->cfg.n_chains % c != 0) return false;
        if (h->cfg.n_tau > 64 || 4 * (64 / c) < h->cfg.n_tau) return false;
        return (size_t)WAVES_PER_BLOCK * packed_lds_bytes_per_wave(c, h->wi_stride) <= 160u * 1024u;
    };
    int want = h->chains_per_wave;
    // automatic: two chains per wave once N_tau <= 32 leaves half of every wave idle and there are enough chains to
    // fill the chip's 4096 wave slots that way (measured, profiles/r02_packed_scan.jsonl: 1.49 x at N_tau = 22,
    // 1.47 x at N_tau = 32 with 8192 chains; 0.96 x with 4096 chains, and 0.93 x for 32 < N_tau <= 64 where the
    // packed kernel holds two waves per SIMD)
    if (want == 0) want = (h->cfg.n_tau <= 32 && h->cfg.n_chains >= 6144) ? 2 : 1;
    while (want > 1 && !fits(want)) want /= 2;
    return want;
}

// Proposals of one chain evaluated per step of the next free sweep launch (free_sweep_spec.h): the caller's choice
// (sac_set_proposals_per_step) when the shape allows it -- equal fixed amplitudes, N_tau <= 64 and <= 4 slots x the lanes
// of a group -- else 1 (free_stage_kernel).  Automatic: runs that leave the chip mostly empty (at most one wave per SIMD),
// where the latency of one chain is what counts; with more chains the one-proposal kernels have the higher throughput.
static int spec_proposals_per_step(const sac_handle *h)
{
    auto fits = [&](int c) { return c == 1 || (h->cfg.n_tau <= 64 && 4 * (64 / c) >= h->cfg.n_tau); };
    int want = h->proposals_per_step;
    if (want == 0) want = (h->cfg.n_chains <= 1024 && h->cfg.n_tau <= 64) ? 2 : 1;
    while (want > 1 && !fits(want)) want /= 2;
    return want;
}

// Launches `n_sweeps` sweeps in chunks; flags_first apply to the first chunk only (zeroing),
// flags_last to the last chunk only (retune), flags_all to every chunk.
static int run_sweeps(sac_handle *h, int64_t n_sweeps, double theta, int flags_all, int flags_first,
                      int flags_last, double retune_div)
{
    FreeParams p;
    fill_params(h, p);
    p.inv2theta = 1.0 / (2.0 * theta);
    p.retune_div = retune_div;
    const bool log = (h->log_chain >= 0 && h->dLog != nullptr);
    int64_t done = 0;
    bool first = true;
    do {
        int64_t chunk = n_sweeps - done;
        if (chunk > h->sweeps_per_launch) chunk = h->sweeps_per_launch;
        const bool last = (done + chunk >= n_sweeps);
        p.n_sweeps = (int)chunk;
        p.flags = flags_all | (first ? flags_first : 0) | (last ? flags_last : 0);
        p.prio_shift = prio_shift_for(h->ms_per_sweep * (double)chunk);
        const size_t li = (size_t)h->last_launches;
        if (h->kev.size() <= li) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            h->kev.emplace_back(a, b);
        }
        HIP_TRY(hipEventRecord(h->kev[li].first, h->stream));
        const bool par2 = (h->cfg.par == 2);
        const bool ua = h->uniform_A && !par2;
        const int wpb = WAVES_PER_BLOCK;
        const int cpw = ua ? packed_chains_per_wave(h) : 1;
        h->last_chains_per_wave = cpw;
        if (cpw > 1) {
            h->last_proposals_per_step = 1;      // a packed handle does not speculate
            // several chains per wave (free_sweep_packed.h): same chains, other mapping onto the wave
            const int lpc = 64 / cpw;
            const int T = (lpc >= h->cfg.n_tau) ? 1 : (2 * lpc >= h->cfg.n_tau) ? 2 : 4;
            const int waves = h->cfg.n_chains / cpw;
            const dim3 pgrid((unsigned)((waves + wpb - 1) / wpb)), pblock(64 * wpb);
            const size_t lds = (size_t)wpb * packed_lds_bytes_per_wave(cpw, h->wi_stride);
#define SAC_PK1(C_, T_, L_)                                                                                     \
    do {                                                                                                        \
        if (lds > 48 * 1024)                                                                                    \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&free_stage_packed_kernel<C_, T_, L_>),  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                 \
        hipLaunchKernelGGL((free_stage_packed_kernel<C_, T_, L_>), pgrid, pblock, lds, h->stream, p);           \
    } while (0)
#define SAC_PK(C_, T_)                                                                                          \
    do {                                                                                                        \
        if (log) SAC_PK1(C_, T_, true);                                                                         \
        else SAC_PK1(C_, T_, false);                                                                            \
    } while (0)
            switch (cpw * 8 + T) {
            case 2 * 8 + 1: SAC_PK(2, 1); break;
            case 2 * 8 + 2: SAC_PK(2, 2); break;
            case 4 * 8 + 1: SAC_PK(4, 1); break;
            case 4 * 8 + 2: SAC_PK(4, 2); break;
            default: SAC_PK(4, 4); break;
            }
#undef SAC_PK
#undef SAC_PK1
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(h->kev[li].second, h->stream));
            h->last_launches += 1;
            done += chunk;
            first = false;
            continue;
        }
        const int pps = ua ? spec_proposals_per_step(h) : 1;
        h->last_proposals_per_step = pps;
        if (pps > 1) {
            // few chains: one chain per wave, several of its proposals per step (free_sweep_spec.h)
            const int lpc = 64 / pps;
            const int T = (lpc >= h->cfg.n_tau) ? 1 : (2 * lpc >= h->cfg.n_tau) ? 2 : 4;
            const dim3 sgrid((unsigned)h->cfg.n_chains), sblock(64);
            const size_t lds = spec_lds_bytes(h->wi_stride);
#define SAC_SP(C_, T_)                                                                                          \
    do {                                                                                                        \
        if (log) hipLaunchKernelGGL((free_stage_spec_kernel<C_, T_, true>), sgrid, sblock, lds, h->stream, p);  \
        else hipLaunchKernelGGL((free_stage_spec_kernel<C_, T_, false>), sgrid, sblock, lds, h->stream, p);     \
    } while (0)
            switch (pps * 8 + T) {
            case 2 * 8 + 1: SAC_SP(2, 1); break;
            case 2 * 8 + 2: SAC_SP(2, 2); break;
            case 4 * 8 + 1: SAC_SP(4, 1); break;
            case 4 * 8 + 2: SAC_SP(4, 2); break;
            default: SAC_SP(4, 4); break;
            }
#undef SAC_SP
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(h->kev[li].second, h->stream));
            h->last_launches += 1;
            done += chunk;
            first = false;
            continue;
        }
        const dim3 grid((unsigned)((h->cfg.n_chains + wpb - 1) / wpb)), block(64 * wpb);
        const int variant = (par2 ? 8 : 0) | (h->cfg.n_tau <= 64 ? 0 : 4) | (log ? 2 : 0) | (ua ? 1 : 0);
#define SAC_LAUNCH(NTL_, LOG_, UA_, P2_) \
    hipLaunchKernelGGL((free_stage_kernel<NTL_, LOG_, UA_, P2_>), grid, block, 0, h->stream, p)
        switch (variant) {
        case 0: SAC_LAUNCH(1, false, false, false); break;
        case 1: SAC_LAUNCH(1, false, true, false); break;
        case 2: SAC_LAUNCH(1, true, false, false); break;
        case 3: SAC_LAUNCH(1, true, true, false); break;
        case 4: SAC_LAUNCH(2, false, false, false); break;
        case 5: SAC_LAUNCH(2, false, true, false); break;
        case 6: SAC_LAUNCH(2, true, false, false); break;
        case 7: SAC_LAUNCH(2, true, true, false); break;
        case 8: SAC_LAUNCH(1, false, false, true); break;
        case 10: SAC_LAUNCH(1, true, false, true); break;
        case 12: SAC_LAUNCH(2, false, false, true); break;
        default: SAC_LAUNCH(2, true, false, true); break;
        }
#undef SAC_LAUNCH
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(h->kev[li].second, h->stream));
        h->last_launches += 1;
        done += chunk;
        first = false;
    } while (done < n_sweeps);
    return SAC_OK;
}

// The per-stage merge over ranks (include/sac_hip.h "multi-GPU merge"): this rank's histogram replicas summed into the
// payload, its scalar sums packed behind them, ONE all-reduce(sum) + ONE all-reduce(min), merged scalars read back.
static int comm_merge(sac_handle *h)
{
    if (!h->comm) return SAC_OK;
    saccomm::Comm *c = h->comm;
    const int nc = h->cfg.n_chains;
    const long long n_hist = (long long)h->cfg.n_grid * h->hist_arrays, n_pay = n_hist + SAC_MAX_N_TAU;   // [hist | sum of Gbar]
    const bool sampled = (h->last_stage == 2);
    const double steps = (double)h->last_steps;
    if (sampled) {
        hipLaunchKernelGGL(hist_sum_kernel, dim3((unsigned)((n_hist + 255) / 256)), dim3(256), 0, h->stream,
                           (const double *)h->dHist, n_hist, h->hist_copies, c->dSum);
        hipLaunchKernelGGL(gbar_sum_kernel, dim3(1), dim3(SAC_MAX_N_TAU), 0, h->stream, (const double *)h->dGbar, nc, c->dSum + n_hist);
        HIP_TRY(hipGetLastError());
    }
    if (h->peak) {
        hipLaunchKernelGGL(peak_weights_kernel, dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, h->stream,
                           (const double *)h->dA, h->wi_stride, nc, h->cfg.n_peak, h->cfg.n_omega, h->dWeights);
        hipLaunchKernelGGL(peak_pack_kernel, dim3(1), dim3(64), 0, h->stream, (const PeakScalars *)h->dPsc,
                           (const double *)h->dEdgeRows, (const double *)h->dWeights, nc, steps, sampled ? 1 : 0,
                           c->dSum + n_pay, c->dMin);
    } else {
        hipLaunchKernelGGL(free_pack_kernel, dim3(1), dim3(64), 0, h->stream, (const ChainScalars *)h->dSc, nc, steps,
                           sampled ? 1 : 0, c->dSum + n_pay, c->dMin);
    }
    HIP_TRY(hipGetLastError());
    char msg[256];
    if (saccomm::merge(c, (size_t)n_pay, sampled, h->stream, msg, sizeof(msg)))
        return fail(SAC_E_HIP, "merge over ranks: %s", msg);
    return SAC_OK;
}

static int stage_begin(sac_handle *h, const char *who, int64_t steps, double theta)
{
    if (!h) return fail(SAC_E_INVALID, "%s: null handle", who);
    if (!h->problem || !h->state_set || !h->gbar_set)
        return fail(SAC_E_STATE, "%s: needs sac_upload_problem, sac_set_state and sac_init_gbar first", who);
    if (steps < 0) return fail(SAC_E_INVALID, "%s: steps=%lld", who, (long long)steps);
    if (!(theta > 0.0) || !std::isfinite(theta)) return fail(SAC_E_INVALID, "%s: theta=%g", who, theta);
    int rc;
    if ((rc = use_device(h))) return rc;
    h->last_launches = 0;
    drop_merged(h);
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    return SAC_OK;
}

static int stage_end(sac_handle *h, int kind, int64_t steps)
{
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->last_stage_ms = ms;
    double kms = 0.0;
    for (int64_t i = 0; i < h->last_launches; ++i) {
        float k = 0.f;
        HIP_TRY(hipEventElapsedTime(&k, h->kev[(size_t)i].first, h->kev[(size_t)i].second));
        kms += k;
    }
    h->last_kernel_ms = kms;
    if (steps > 0 && kms > 0.0) h->ms_per_sweep = kms / (double)steps;
    h->last_stage = kind;
    h->last_steps = steps;
    return SAC_OK;
}

extern "C" int sac_set_chains_per_wave(sac_handle_t *h, int32_t chains_per_wave)
{
    if (!h) return fail(SAC_E_INVALID, "sac_set_chains_per_wave: null handle");
    if (h->peak) return fail(SAC_E_INVALID, "sac_set_chains_per_wave: free parameterisation only");
    if (chains_per_wave != 0 && chains_per_wave != 1 && chains_per_wave != 2 && chains_per_wave != 4)
        return fail(SAC_E_INVALID, "sac_set_chains_per_wave: %d (0 = automatic, 1, 2 or 4)", chains_per_wave);
    h->chains_per_wave = chains_per_wave;
    return SAC_OK;
}

extern "C" int sac_get_chains_per_wave(sac_handle_t *h, int32_t *last_used)
{
    if (!h || !last_used) return fail(SAC_E_INVALID, "sac_get_chains_per_wave: null argument");
    *last_used = h->last_chains_per_wave;
    return SAC_OK;
}

extern "C" int sac_set_proposals_per_step(sac_handle_t *h, int32_t proposals_per_step)
{
    if (!h) return fail(SAC_E_INVALID, "sac_set_proposals_per_step: null handle");
    if (h->peak) return fail(SAC_E_INVALID, "sac_set_proposals_per_step: free parameterisation only");
    if (proposals_per_step != 0 && proposals_per_step != 1 && proposals_per_step != 2 && proposals_per_step != 4)
        return fail(SAC_E_INVALID, "sac_set_proposals_per_step: %d (0 = automatic, 1, 2 or 4)", proposals_per_step);
    h->proposals_per_step = proposals_per_step;
    return SAC_OK;
}

extern "C" int sac_get_proposals_per_step(sac_handle_t *h, int32_t *last_used)
{
    if (!h || !last_used) return fail(SAC_E_INVALID, "sac_get_proposals_per_step: null argument");
    *last_used = h->last_proposals_per_step;
    return SAC_OK;
}

// ---- peak host side --------------------------------------------------------------------------

// per-chain sums of A over the four groups (peak:1302-1311), lane-strided partials + tree
__global__ void peak_weights_kernel(const double *A, int stride, int n_chains, int n_p, int n, double *out)
{
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (wave >= n_chains) return;
    const int lane = lane_id();
    const double *a = A + (size_t)wave * (size_t)stride;
    const int first[4] = {0, n_p, n_p + n, 2 * n_p + n}, count[4] = {n_p, n, n_p, n};
    for (int g = 0; g < 4; ++g) {
        double part = 0.0;
        for (int i = lane; i < count[g]; i += 64) part = part + a[first[g] + i];
        const double tot = wave_sum_uniform(part);
        if (lane == 0) out[(size_t)wave * 4 + g] = tot;
    }
}

static void fill_peak_params(sac_handle *h, PeakParams &pp)
{
    fill_params(h, pp.f);
    pp.f.gmax = (int)h->cfg.wi_m;
    pp.Kn = h->dKn;
    pp.psc = h->dPsc;
    pp.n_p = h->cfg.n_peak;
    pp.n_cont = h->cfg.n_omega;
    pp.wi0 = (int)h->cfg.wi_0;
    pp.wim = (int)h->cfg.wi_m;
    pp.sample_neg = h->sample_neg ? 1 : 0;
    pp.fix_edge = h->cfg.fix_edge;
    pp.indiv_update = h->indiv_update ? 1 : 0;
    pp.transfer = 1;
    pp.refresh_bounds = 0;
}

extern "C" int sac_upload_problem_peak(sac_handle_t *h, const double *Kp, const double *Kn, const double *G,
                                       const double *sigma_inv)
{
    if (!h || !h->peak) return fail(SAC_E_INVALID, "sac_upload_problem_peak: not a SAC_KIND_PEAK handle");
    if (h->sample_neg && !Kn) return fail(SAC_E_INVALID, "sac_upload_problem_peak: Kn is required when both axes are sampled");
    int rc = sac_upload_problem(h, Kp, G, sigma_inv);
    if (rc) return rc;
    if (Kn) {
        const size_t nt = (size_t)h->cfg.n_tau, ng = (size_t)h->cfg.n_grid;
        if ((size_t)h->ntp == nt) {
            HIP_TRY(copy_sync(h->stream, h->dKn, Kn, ng * nt * sizeof(double), hipMemcpyHostToDevice));
        } else {
            HIP_TRY(hipMemcpy2D(h->dKn, (size_t)h->ntp * sizeof(double), Kn, nt * sizeof(double),
                                nt * sizeof(double), ng, hipMemcpyHostToDevice));
        }
    }
    return SAC_OK;
}

static int peak_init(sac_handle *h)
{
    PeakParams pp;
    fill_peak_params(h, pp);
    const dim3 grid(grid_blocks(h)), block(64 * WAVES_PER_BLOCK);
    if (h->cfg.n_tau <= 64) hipLaunchKernelGGL(peak_init_kernel<1>, grid, block, 0, h->stream, pp);
    else hipLaunchKernelGGL(peak_init_kernel<2>, grid, block, 0, h->stream, pp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->gbar_set = true;
    h->last_stage = 0;
    return comm_merge(h);
}

static int run_peak_sweeps(sac_handle *h, int64_t n_sweeps, double theta, int flags_all, int flags_first,
                           int flags_last, double retune_div, bool transfer, bool refresh_bounds)
{
    PeakParams pp;
    fill_peak_params(h, pp);
    pp.f.inv2theta = 1.0 / (2.0 * theta);
    pp.f.retune_div = retune_div;
    pp.transfer = transfer ? 1 : 0;
    pp.refresh_bounds = refresh_bounds ? 1 : 0;
    const bool log = (h->log_chain >= 0 && h->dLog != nullptr);
    int64_t done = 0;
    bool first = true;
    do {
        int64_t chunk = n_sweeps - done;
        if (chunk > h->sweeps_per_launch) chunk = h->sweeps_per_launch;
        const bool last = (done + chunk >= n_sweeps);
        pp.f.n_sweeps = (int)chunk;
        pp.f.flags = flags_all | (first ? flags_first : 0) | (last ? flags_last : 0);
        pp.f.prio_shift = prio_shift_for(h->ms_per_sweep * (double)chunk);
        const size_t li = (size_t)h->last_launches;
        if (h->kev.size() <= li) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            h->kev.emplace_back(a, b);
        }
        HIP_TRY(hipEventRecord(h->kev[li].first, h->stream));
        const bool xfer = h->sample_neg;
        const int wpb = WAVES_PER_BLOCK;
        const dim3 grid((unsigned)((h->cfg.n_chains + wpb - 1) / wpb)), block(64 * wpb);
        const int variant = (xfer ? 4 : 0) | (h->cfg.n_tau <= 64 ? 0 : 2) | (log ? 1 : 0);
        const size_t lds = (size_t)wpb * peak_lds_bytes_per_wave(h->wi_stride, xfer);
#define SAC_PLAUNCH(NTL_, LOG_, X_)                                                                      \
    do {                                                                                                 \
        if (lds > 48 * 1024)                                                                             \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&peak_stage_kernel<NTL_, LOG_, X_>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
        hipLaunchKernelGGL((peak_stage_kernel<NTL_, LOG_, X_>), grid, block, lds, h->stream, pp);        \
    } while (0)
        switch (variant) {
        case 0: SAC_PLAUNCH(1, false, false); break;
        case 1: SAC_PLAUNCH(1, true, false); break;
        case 2: SAC_PLAUNCH(2, false, false); break;
        case 3: SAC_PLAUNCH(2, true, false); break;
        case 4: SAC_PLAUNCH(1, false, true); break;
        case 5: SAC_PLAUNCH(1, true, true); break;
        case 6: SAC_PLAUNCH(2, false, true); break;
        default: SAC_PLAUNCH(2, true, true); break;
        }
#undef SAC_PLAUNCH
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(h->kev[li].second, h->stream));
        h->last_launches += 1;
        done += chunk;
        first = false;
    } while (done < n_sweeps);
    return SAC_OK;
}

static int peak_adjust_windows(sac_handle *h, int64_t steps, double theta)
{
    int rc;
    const int64_t per = steps / 10;                                  // peak:1070
    for (int round = 0; round < 10; ++round)                         // peak:1066-1093
        if ((rc = run_peak_sweeps(h, per, theta, 0, STAGE_ZERO_ACC, STAGE_RETUNE, (double)per, true, true)))
            return rc;
    if ((rc = stage_end(h, 1, steps))) return rc;
    return comm_merge(h);
}

static int peak_sample(sac_handle *h, int64_t steps, double theta, bool bin)
{
    int rc;
    if (!bin) {                                                      // peak:1103-1106
        HIP_TRY(hipMemsetAsync(h->dHist, 0, (size_t)h->hist_copies * 4 * (size_t)h->cfg.n_grid * sizeof(double),
                               h->stream));
        h->binned = false;
    } else {
        h->binned = true;
    }
    if ((rc = run_peak_sweeps(h, steps, theta, STAGE_SAMPLE, STAGE_ZERO_ACC | STAGE_BOUNDS_AT_START, 0, 1.0,
                              !bin, false)))
        return rc;
    if ((rc = stage_end(h, 2, steps))) return rc;
    // peak:1159-1165: per-chain edge means, and the running sums of means and squares
    const int nc = h->cfg.n_chains;
    hipLaunchKernelGGL(peak_edge_rows_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, h->stream,
                       (const PeakScalars *)h->dPsc, h->dEdgeRows, nc, (double)steps, h->sample_neg ? 1 : 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h->edge_rows.data(), h->dEdgeRows, (size_t)nc * 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return comm_merge(h);
}

extern "C" int sac_sample_bin(sac_handle_t *h, int64_t steps, double theta)
{
    int rc;
    if (!h || !h->peak) return fail(SAC_E_INVALID, "sac_sample_bin: not a SAC_KIND_PEAK handle");
    if ((rc = stage_begin(h, "sac_sample_bin", steps, theta))) return rc;
    return peak_sample(h, steps, theta, true);
}

extern "C" int sac_set_indiv_update(sac_handle_t *h, int32_t on)
{
    if (!h || !h->peak) return fail(SAC_E_INVALID, "sac_set_indiv_update: not a SAC_KIND_PEAK handle");
    h->indiv_update = (on != 0);
    return SAC_OK;
}

extern "C" int sac_zero_accumulators(sac_handle_t *h)
{
    if (!h || !h->peak) return fail(SAC_E_INVALID, "sac_zero_accumulators: not a SAC_KIND_PEAK handle");
    int rc;
    if ((rc = use_device(h))) return rc;
    drop_merged(h);
    HIP_TRY(hipMemsetAsync(h->dHist, 0, (size_t)h->hist_copies * 4 * (size_t)h->cfg.n_grid * sizeof(double), h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::fill(h->edge_rows.begin(), h->edge_rows.end(), 0.0);
    HIP_TRY(hipMemset(h->dEdgeRows, 0, h->edge_rows.size() * sizeof(double)));
    h->binned = true;
    return SAC_OK;
}

static int peak_chain_stats(sac_handle *h, int32_t chain, double *out6, double *accept_rates)
{
    if (chain < 0 || chain >= h->cfg.n_chains)
        return fail(SAC_E_INVALID, "sac_get_chain_stats: chain %d outside 0..%d", chain, h->cfg.n_chains - 1);
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    PeakScalars s;
    HIP_TRY(copy_sync(h->stream, &s, h->dPsc + chain, sizeof(s), hipMemcpyDeviceToHost));
    const double steps = (double)h->last_steps;
    if (out6) {
        out6[0] = (h->last_stage == 2) ? s.sum_chi2 / steps : 0.0;
        out6[1] = s.chi2;
        out6[2] = s.chi2_min;
        out6[3] = (double)s.sweep;
        out6[4] = h->edge_rows[(size_t)chain * 6 + 0];
        out6[5] = h->edge_rows[(size_t)chain * 6 + 1];
    }
    if (accept_rates)       // 11 entries for peak
        for (int k = 0; k < 11; ++k) accept_rates[k] = (h->last_stage == 2) ? s.acc[k] / steps : s.acc[k];
    return SAC_OK;
}

extern "C" int sac_get_bounds(sac_handle_t *h, int32_t chain, double *out4)
{
    if (!h || !h->peak || !out4) return fail(SAC_E_INVALID, "sac_get_bounds: not a SAC_KIND_PEAK handle");
    if (chain < 0 || chain >= h->cfg.n_chains)
        return fail(SAC_E_INVALID, "sac_get_bounds: chain %d outside 0..%d", chain, h->cfg.n_chains - 1);
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    PeakScalars s;
    HIP_TRY(copy_sync(h->stream, &s, h->dPsc + chain, sizeof(s), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; ++k) out4[k] = (double)s.bounds[k];
    return SAC_OK;
}

extern "C" int sac_get_peak_stats(sac_handle_t *h, sac_peak_stats_t *out)
{
    if (!h || !h->peak || !out) return fail(SAC_E_INVALID, "sac_get_peak_stats: not a SAC_KIND_PEAK handle");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->comm && h->comm->merged) {                                // the values over all ranks (comm_merge)
        const double *t = h->comm->hTail;
        const double n = t[44];
        std::memset(out, 0, sizeof(*out));
        out->sampled_chi2 = t[0] / n;
        for (int k = 0; k < 11; ++k) { out->accept_rates[k] = t[1 + k] / n; out->windows[k] = t[12 + k] / n; }
        for (int k = 0; k < 6; ++k) (&out->edge_res[0][0])[k] = t[23 + k] / n;
        for (int k = 0; k < 4; ++k) { out->bounds[k] = t[29 + k] / n; out->weights[k] = t[33 + k] / n; }
        out->chi2_min_mean = t[37] / n;
        out->chi2_min = t[PEAK_SLOTS];
        out->proposals = (uint64_t)t[38]; out->evaluated = (uint64_t)t[39]; out->accepted = (uint64_t)t[40];
        out->accepted_moves = (uint64_t)t[41]; out->columns_gathered = (uint64_t)t[42]; out->sweeps = (uint64_t)t[43];
        out->n_chains_total = (int64_t)n;
        out->last_stage_ms = h->last_stage_ms;
        out->last_kernel_ms = h->last_kernel_ms;
        out->last_kernel_launches = h->last_launches;
        return SAC_OK;
    }
    const int nc = h->cfg.n_chains;
    std::vector<PeakScalars> sc((size_t)nc);
    HIP_TRY(copy_sync(h->stream, sc.data(), h->dPsc, (size_t)nc * sizeof(PeakScalars), hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(peak_weights_kernel, dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, h->stream,
                       (const double *)h->dA, h->wi_stride, nc, h->cfg.n_peak, h->cfg.n_omega, h->dWeights);
    HIP_TRY(hipGetLastError());
    std::vector<double> wts((size_t)nc * 4);
    HIP_TRY(hipMemcpyAsync(wts.data(), h->dWeights, wts.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::memset(out, 0, sizeof(*out));
    const double steps = (double)h->last_steps, dn = (double)nc;
    double cmin = sc[0].chi2_min;
    for (int c = 0; c < nc; ++c) {
        const PeakScalars &s = sc[(size_t)c];
        if (h->last_stage == 2) out->sampled_chi2 += s.sum_chi2 / steps;
        for (int k = 0; k < 11; ++k) {
            out->accept_rates[k] += (h->last_stage == 2) ? s.acc[k] / steps : s.acc[k];
            out->windows[k] += s.windows[k];
        }
        for (int k = 0; k < 4; ++k) { out->bounds[k] += (double)s.bounds[k]; out->weights[k] += wts[(size_t)c * 4 + k]; }
        for (int k = 0; k < 6; ++k) (&out->edge_res[0][0])[k] += h->edge_rows[(size_t)c * 6 + k];
        if (s.chi2_min < cmin) cmin = s.chi2_min;
        out->chi2_min_mean += s.chi2_min;
        out->proposals += s.counters[0];
        out->evaluated += s.counters[1];
        out->accepted += s.counters[2];
        out->accepted_moves += s.counters[3];
        out->columns_gathered += s.counters[4];
        out->sweeps += s.sweep;
    }
    out->sampled_chi2 /= dn;
    out->chi2_min_mean /= dn;
    out->n_chains_total = nc;
    out->chi2_min = cmin;
    for (int k = 0; k < 11; ++k) { out->accept_rates[k] /= dn; out->windows[k] /= dn; }
    for (int k = 0; k < 4; ++k) { out->bounds[k] /= dn; out->weights[k] /= dn; }
    for (int k = 0; k < 6; ++k) (&out->edge_res[0][0])[k] /= dn;
    out->last_stage_ms = h->last_stage_ms;
    out->last_kernel_ms = h->last_kernel_ms;
    out->last_kernel_launches = h->last_launches;
    return SAC_OK;
}

extern "C" int sac_get_histogram_peak(sac_handle_t *h, int32_t which, double *out)
{
    if (!h || !h->peak || !out) return fail(SAC_E_INVALID, "sac_get_histogram_peak: not a SAC_KIND_PEAK handle");
    if (which < 0 || which > 3) return fail(SAC_E_INVALID, "sac_get_histogram_peak: which=%d", which);
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t ng = (size_t)h->cfg.n_grid;
    if (h->comm && h->comm->merged && h->last_stage == 2) {           // summed over all ranks' chains (comm_merge)
        HIP_TRY(copy_sync(h->stream, out, h->comm->dSum + (size_t)which * ng, ng * sizeof(double), hipMemcpyDeviceToHost));
        const double n = h->comm->hTail[44], div = h->binned ? 1.0 : (double)h->last_steps;
        for (size_t i = 0; i < ng; ++i) out[i] = out[i] / div / n;
        return SAC_OK;
    }
    std::vector<double> tmp(ng);
    std::fill(out, out + ng, 0.0);
    for (int c = 0; c < h->hist_copies; ++c) {
        HIP_TRY(copy_sync(h->stream, tmp.data(), h->dHist + ((size_t)c * 4 + (size_t)which) * ng, ng * sizeof(double),
                          hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ng; ++i) out[i] = (c == 0) ? tmp[i] : out[i] + tmp[i];
    }
    const double nc = (double)h->cfg.n_chains;
    const double div = h->binned ? 1.0 : (double)h->last_steps;      // peak:1151-1154
    for (size_t i = 0; i < ng; ++i) out[i] = out[i] / div / nc;
    return SAC_OK;
}

extern "C" int sac_adjust_windows(sac_handle_t *h, int64_t steps, double theta)
{
    int rc;
    if ((rc = stage_begin(h, "sac_adjust_windows", steps, theta))) return rc;
    if (h->peak) return peak_adjust_windows(h, steps, theta);
    const int64_t per = steps / 10;                                  // free:873
    for (int round = 0; round < 10; ++round) {                       // free:869
        // free:871 zero the rates, free:873-880 `per` sweeps, free:883-895 retune.
        // per == 0 still runs the (empty) launch so that 0/0 = NaN leaves the windows alone.
        if ((rc = run_sweeps(h, per, theta, 0, STAGE_ZERO_ACC, STAGE_RETUNE, (double)per))) return rc;
    }
    if ((rc = stage_end(h, 1, steps))) return rc;
    return comm_merge(h);
}

extern "C" int sac_sample(sac_handle_t *h, int64_t steps, double theta)
{
    int rc;
    if ((rc = stage_begin(h, "sac_sample", steps, theta))) return rc;
    if (h->peak) return peak_sample(h, steps, theta, false);
    HIP_TRY(hipMemsetAsync(h->dHist, 0, (size_t)h->hist_copies * (size_t)h->cfg.n_grid * sizeof(double),
                           h->stream));                             // free:906
    if ((rc = run_sweeps(h, steps, theta, STAGE_SAMPLE, STAGE_ZERO_ACC, 0, 1.0))) return rc;
    if ((rc = stage_end(h, 2, steps))) return rc;
    return comm_merge(h);
}

extern "C" int sac_get_stats(sac_handle_t *h, sac_stats_t *out)
{
    if (!h || !out) return fail(SAC_E_INVALID, "sac_get_stats: null argument");
    if (h->peak) return fail(SAC_E_INVALID, "sac_get_stats: use sac_get_peak_stats for SAC_KIND_PEAK");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->comm && h->comm->merged) {                                // the values over all ranks (comm_merge)
        const double *t = h->comm->hTail;
        const double n = t[18];
        std::memset(out, 0, sizeof(*out));
        out->sampled_chi2 = t[0] / n;
        for (int k = 0; k < 6; ++k) out->accept_rates[k] = t[1 + k] / n;
        for (int k = 0; k < 4; ++k) out->windows[k] = t[7 + k] / n;
        out->chi2_min_mean = t[11] / n;
        out->chi2_min = t[FREE_SLOTS];
        out->proposals = (uint64_t)t[12]; out->evaluated = (uint64_t)t[13]; out->accepted = (uint64_t)t[14];
        out->accepted_moves = (uint64_t)t[15]; out->columns_gathered = (uint64_t)t[16]; out->sweeps = (uint64_t)t[17];
        out->n_chains_total = (int64_t)n;
        out->last_stage_ms = h->last_stage_ms;
        out->last_kernel_ms = h->last_kernel_ms;
        out->last_kernel_launches = h->last_launches;
        return SAC_OK;
    }
    const int nc = h->cfg.n_chains;
    std::vector<ChainScalars> sc((size_t)nc);
    HIP_TRY(copy_sync(h->stream, sc.data(), h->dSc, (size_t)nc * sizeof(ChainScalars), hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof(*out));
    const double steps = (double)h->last_steps;
    double s_chi2 = 0.0, cmm = 0.0, ar[6] = {0, 0, 0, 0, 0, 0}, win[4] = {0, 0, 0, 0};
    double cmin = sc[0].chi2_min;
    for (int c = 0; c < nc; ++c) {
        const ChainScalars &s = sc[(size_t)c];
        if (h->last_stage == 2) {                                    // free:925-926
            s_chi2 += s.sum_chi2 / steps;
            for (int k = 0; k < 6; ++k) ar[k] += s.acc[k] / steps;
        } else {
            for (int k = 0; k < 6; ++k) ar[k] += s.acc[k];
        }
        for (int k = 0; k < 4; ++k) win[k] += s.windows[k];
        if (s.chi2_min < cmin) cmin = s.chi2_min;
        cmm += s.chi2_min;
        out->proposals += s.counters[0];
        out->evaluated += s.counters[1];
        out->accepted += s.counters[2];
        out->accepted_moves += s.counters[3];
        out->columns_gathered += s.counters[4];
        out->sweeps += s.sweep;
    }
    out->sampled_chi2 = s_chi2 / (double)nc;
    out->chi2_min_mean = cmm / (double)nc;
    out->n_chains_total = nc;
    out->chi2_min = cmin;
    for (int k = 0; k < 6; ++k) out->accept_rates[k] = ar[k] / (double)nc;
    for (int k = 0; k < 4; ++k) out->windows[k] = win[k] / (double)nc;
    out->last_stage_ms = h->last_stage_ms;
    out->last_kernel_ms = h->last_kernel_ms;
    out->last_kernel_launches = h->last_launches;
    return SAC_OK;
}

extern "C" int sac_get_chain_stats(sac_handle_t *h, int32_t chain, double *out6, double *accept_rates)
{
    if (!h) return fail(SAC_E_INVALID, "sac_get_chain_stats: null handle");
    if (h->peak) return peak_chain_stats(h, chain, out6, accept_rates);
    if (chain < 0 || chain >= h->cfg.n_chains)
        return fail(SAC_E_INVALID, "sac_get_chain_stats: chain %d outside 0..%d", chain, h->cfg.n_chains - 1);
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    ChainScalars s;
    HIP_TRY(copy_sync(h->stream, &s, h->dSc + chain, sizeof(s), hipMemcpyDeviceToHost));
    const double steps = (double)h->last_steps;
    if (out6) {
        out6[0] = (h->last_stage == 2) ? s.sum_chi2 / steps : 0.0;
        out6[1] = s.chi2;
        out6[2] = s.chi2_min;
        out6[3] = (double)s.sweep;
        out6[4] = out6[5] = 0.0;
    }
    if (accept_rates)
        for (int k = 0; k < 6; ++k) accept_rates[k] = (h->last_stage == 2) ? s.acc[k] / steps : s.acc[k];
    return SAC_OK;
}

extern "C" int sac_replay(sac_handle_t *h, int32_t chain, double theta, const sac_proposal_t *props, int64_t n,
                          double *chi2_new, int32_t *accepted)
{
    if (!h || (n > 0 && !props)) return fail(SAC_E_INVALID, "sac_replay: null argument");
    if (h->peak || h->cfg.par == 2) return fail(SAC_E_INVALID, "sac_replay: free parameterisation with par = 1 or 3 only");
    if (chain < 0 || chain >= h->cfg.n_chains) return fail(SAC_E_INVALID, "sac_replay: chain %d outside 0..%d", chain, h->cfg.n_chains - 1);
    int rc;
    if ((rc = stage_begin(h, "sac_replay", n, theta))) return rc;
    static_assert(sizeof(ReplayProp) == sizeof(sac_proposal_t), "proposal layout");
    std::vector<ReplayProp> rel((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const sac_proposal_t &q = props[i];
        if (q.move < 1 || q.move > 3) return fail(SAC_E_INVALID, "sac_replay: proposal %lld: move %d", (long long)i, q.move);
        rel[(size_t)i].move = q.move;
        rel[(size_t)i].u = q.u;
        for (int e = 0; e < 3; ++e) {
            rel[(size_t)i].k[e] = 0;
            rel[(size_t)i].w_new[e] = 0;
            if (e >= q.move) continue;
            if (q.k[e] < 0 || q.k[e] >= h->cfg.n_omega)
                return fail(SAC_E_INVALID, "sac_replay: proposal %lld: delta index %d outside 0..%d", (long long)i, q.k[e], h->cfg.n_omega - 1);
            if (q.w_new[e] < h->cfg.wi_0 || q.w_new[e] > h->cfg.wi_m)      // the reference `continue`s before evaluating these
                return fail(SAC_E_INVALID, "sac_replay: proposal %lld: position %lld outside [%lld, %lld]", (long long)i,
                            (long long)q.w_new[e], (long long)h->cfg.wi_0, (long long)h->cfg.wi_m);
            rel[(size_t)i].k[e] = q.k[e];
            rel[(size_t)i].w_new[e] = q.w_new[e] - h->pos_offset;
        }
        for (int e = 1; e < q.move; ++e)
            for (int f = 0; f < e; ++f)
                if (q.k[e] == q.k[f]) return fail(SAC_E_INVALID, "sac_replay: proposal %lld repeats delta %d", (long long)i, q.k[e]);
    }
    if (n == 0) return stage_end(h, 0, 0);
    DevBuf b_dP, b_dC, b_dAcc, b_dW;      // released on every exit path
    ReplayProp *dP = nullptr;
    double *dC = nullptr;
    int *dAcc = nullptr;
    HIP_TRY(b_dP.alloc((size_t)n * sizeof(ReplayProp)));
    dP = b_dP.as<std::remove_pointer_t<decltype(dP)>>();
    HIP_TRY(b_dC.alloc((size_t)n * sizeof(double)));
    dC = b_dC.as<std::remove_pointer_t<decltype(dC)>>();
    HIP_TRY(b_dAcc.alloc((size_t)n * sizeof(int)));
    dAcc = b_dAcc.as<std::remove_pointer_t<decltype(dAcc)>>();
    HIP_TRY(copy_sync(h->stream, dP, rel.data(), (size_t)n * sizeof(ReplayProp), hipMemcpyHostToDevice));
    FreeParams p;
    fill_params(h, p);
    p.inv2theta = 1.0 / (2.0 * theta);
    if (h->cfg.n_tau <= 64) hipLaunchKernelGGL(replay_kernel<1>, dim3(1), dim3(64), 0, h->stream, p, (int)chain, dP, (long long)n, dC, dAcc);
    else hipLaunchKernelGGL(replay_kernel<2>, dim3(1), dim3(64), 0, h->stream, p, (int)chain, dP, (long long)n, dC, dAcc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (chi2_new) HIP_TRY(copy_sync(h->stream, chi2_new, dC, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (accepted) HIP_TRY(copy_sync(h->stream, accepted, dAcc, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    return stage_end(h, 0, 0);
}

extern "C" int sac_replay_peak(sac_handle_t *h, int32_t chain, double theta, const sac_peak_proposal_t *props, int64_t n,
                               double *chi2_new, int32_t *accepted)
{
    if (!h || (n > 0 && !props)) return fail(SAC_E_INVALID, "sac_replay_peak: null argument");
    if (!h->peak) return fail(SAC_E_INVALID, "sac_replay_peak: not a SAC_KIND_PEAK handle");
    if (chain < 0 || chain >= h->cfg.n_chains) return fail(SAC_E_INVALID, "sac_replay_peak: chain %d outside 0..%d", chain, h->cfg.n_chains - 1);
    int rc;
    if ((rc = stage_begin(h, "sac_replay_peak", n, theta))) return rc;
    static_assert(sizeof(PeakReplayProp) == sizeof(sac_peak_proposal_t), "proposal layout");
    const int n_axis = h->cfg.n_omega + h->cfg.n_peak;
    const int n_used = h->sample_neg ? 2 * n_axis : n_axis;
    for (int64_t i = 0; i < n; ++i) {
        const sac_peak_proposal_t &q = props[i];
        if (q.n < 1 || q.n > 4 || (q.transfer && q.n != 2 && q.n != 4) || (!q.transfer && q.n > 3))
            return fail(SAC_E_INVALID, "sac_replay_peak: proposal %lld: %d deltas, transfer=%d", (long long)i, q.n, q.transfer);
        for (int e = 0; e < q.n; ++e) {
            if (q.k[e] < 0 || q.k[e] >= n_used)
                return fail(SAC_E_INVALID, "sac_replay_peak: proposal %lld: delta index %d outside 0..%d", (long long)i, q.k[e], n_used - 1);
            if (q.w_new[e] < 0 || q.w_new[e] > h->cfg.wi_m)
                return fail(SAC_E_INVALID, "sac_replay_peak: proposal %lld: position %lld outside [0, %lld]", (long long)i,
                            (long long)q.w_new[e], (long long)h->cfg.wi_m);
            if (q.transfer && !(std::isfinite(q.A_new[e]) && q.A_new[e] >= 0))    // the reference rejects A' < 0 before evaluating (peak:952)
                return fail(SAC_E_INVALID, "sac_replay_peak: proposal %lld: amplitude %g", (long long)i, q.A_new[e]);
            if (q.transfer && ((q.k[e] >= n_axis) != ((e & 1) == 1)))
                return fail(SAC_E_INVALID, "sac_replay_peak: proposal %lld: a transfer pairs a + axis delta with a - axis delta", (long long)i);
            for (int f = 0; f < e; ++f)
                if (q.k[e] == q.k[f]) return fail(SAC_E_INVALID, "sac_replay_peak: proposal %lld repeats delta %d", (long long)i, q.k[e]);
        }
    }
    if (n == 0) return stage_end(h, 0, 0);
    DevBuf b_dP, b_dC, b_dAcc, b_dW;      // released on every exit path
    PeakReplayProp *dP = nullptr;
    double *dC = nullptr;
    int *dAcc = nullptr;
    HIP_TRY(b_dP.alloc((size_t)n * sizeof(PeakReplayProp)));
    dP = b_dP.as<std::remove_pointer_t<decltype(dP)>>();
    HIP_TRY(b_dC.alloc((size_t)n * sizeof(double)));
    dC = b_dC.as<std::remove_pointer_t<decltype(dC)>>();
    HIP_TRY(b_dAcc.alloc((size_t)n * sizeof(int)));
    dAcc = b_dAcc.as<std::remove_pointer_t<decltype(dAcc)>>();
    HIP_TRY(copy_sync(h->stream, dP, props, (size_t)n * sizeof(PeakReplayProp), hipMemcpyHostToDevice));
    PeakParams pp;
    fill_peak_params(h, pp);
    pp.f.inv2theta = 1.0 / (2.0 * theta);
    if (h->cfg.n_tau <= 64) hipLaunchKernelGGL(peak_replay_kernel<1>, dim3(1), dim3(64), 0, h->stream, pp, (int)chain, dP, (long long)n, dC, dAcc);
    else hipLaunchKernelGGL(peak_replay_kernel<2>, dim3(1), dim3(64), 0, h->stream, pp, (int)chain, dP, (long long)n, dC, dAcc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (chi2_new) HIP_TRY(copy_sync(h->stream, chi2_new, dC, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (accepted) HIP_TRY(copy_sync(h->stream, accepted, dAcc, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    return stage_end(h, 0, 0);
}

extern "C" int sac_validation_chi2(sac_handle_t *h, const double *U_this, const double *U_other, const double *G_other,
                                   const double *sigma_inv_other, const double *gbar_mean, double *chi2_mean, double *chi2_chain)
{
    if (!h || !U_this || !U_other || !G_other || !sigma_inv_other)
        return fail(SAC_E_INVALID, "sac_validation_chi2: null argument");
    if (!h->gbar_set) return fail(SAC_E_STATE, "sac_validation_chi2: no Gbar yet (sac_init_gbar)");
    const size_t nt = (size_t)h->cfg.n_tau;
    for (size_t j = 0; j < nt; ++j)
        if (!std::isfinite(G_other[j]) || !std::isfinite(sigma_inv_other[j]))
            return fail(SAC_E_INVALID, "sac_validation_chi2: non-finite G or sigma_inv at tau index %zu", j);
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(cv_score_run(h->stream, h->dGbar, SAC_MAX_N_TAU, h->cfg.n_chains, h->cfg.n_tau, U_this, U_other, G_other,
                         sigma_inv_other, gbar_mean, chi2_mean, chi2_chain));
    return SAC_OK;
}

extern "C" int sac_get_chain_chi2(sac_handle_t *h, double *chi2, double *chi2_min)
{
    if (!h) return fail(SAC_E_INVALID, "sac_get_chain_chi2: null handle");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t nc = (size_t)h->cfg.n_chains;
    if (h->peak) {
        std::vector<PeakScalars> v(nc);
        HIP_TRY(copy_sync(h->stream, v.data(), h->dPsc, nc * sizeof(PeakScalars), hipMemcpyDeviceToHost));
        for (size_t c = 0; c < nc; ++c) {
            if (chi2) chi2[c] = v[c].chi2;
            if (chi2_min) chi2_min[c] = v[c].chi2_min;
        }
    } else {
        std::vector<ChainScalars> v(nc);
        HIP_TRY(copy_sync(h->stream, v.data(), h->dSc, nc * sizeof(ChainScalars), hipMemcpyDeviceToHost));
        for (size_t c = 0; c < nc; ++c) {
            if (chi2) chi2[c] = v[c].chi2;
            if (chi2_min) chi2_min[c] = v[c].chi2_min;
        }
    }
    return SAC_OK;
}

extern "C" int sac_copy_chain(sac_handle_t *h, int32_t src, int32_t dst)
{
    if (!h) return fail(SAC_E_INVALID, "sac_copy_chain: null handle");
    const int nc = h->cfg.n_chains;
    if (src < 0 || src >= nc || dst < 0 || dst >= nc)
        return fail(SAC_E_INVALID, "sac_copy_chain: chains %d -> %d outside 0..%d", src, dst, nc - 1);
    if (!h->state_set || !h->gbar_set) return fail(SAC_E_STATE, "sac_copy_chain: state / Gbar not set");
    if (src == dst) return SAC_OK;
    int rc;
    if ((rc = use_device(h))) return rc;
    drop_merged(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t ws = (size_t)h->wi_stride;
    HIP_TRY(copy_sync(h->stream, h->dWi + dst * ws, h->dWi + src * ws, ws * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    HIP_TRY(copy_sync(h->stream, h->dA + dst * ws, h->dA + src * ws, ws * sizeof(double), hipMemcpyDeviceToDevice));
    HIP_TRY(copy_sync(h->stream, h->dGbar + (size_t)dst * SAC_MAX_N_TAU, h->dGbar + (size_t)src * SAC_MAX_N_TAU,
                      SAC_MAX_N_TAU * sizeof(double), hipMemcpyDeviceToDevice));
    // scalars: everything but the destination's own sweep counter (its Philox stream) and counters
    if (h->peak) {
        PeakScalars a, b;
        HIP_TRY(copy_sync(h->stream, &a, h->dPsc + src, sizeof(a), hipMemcpyDeviceToHost));
        HIP_TRY(copy_sync(h->stream, &b, h->dPsc + dst, sizeof(b), hipMemcpyDeviceToHost));
        const unsigned long long sweep = b.sweep;
        unsigned long long cnt[5];
        std::memcpy(cnt, b.counters, sizeof(cnt));
        b = a;
        b.sweep = sweep;
        std::memcpy(b.counters, cnt, sizeof(cnt));
        HIP_TRY(copy_sync(h->stream, h->dPsc + dst, &b, sizeof(b), hipMemcpyHostToDevice));
    } else {
        ChainScalars a, b;
        HIP_TRY(copy_sync(h->stream, &a, h->dSc + src, sizeof(a), hipMemcpyDeviceToHost));
        HIP_TRY(copy_sync(h->stream, &b, h->dSc + dst, sizeof(b), hipMemcpyDeviceToHost));
        const unsigned long long sweep = b.sweep;
        unsigned long long cnt[5];
        std::memcpy(cnt, b.counters, sizeof(cnt));
        b = a;
        b.sweep = sweep;
        std::memcpy(b.counters, cnt, sizeof(cnt));
        HIP_TRY(copy_sync(h->stream, h->dSc + dst, &b, sizeof(b), hipMemcpyHostToDevice));
    }
    return SAC_OK;
}

extern "C" int sac_get_histogram(sac_handle_t *h, double *out)
{
    if (!h || !out) return fail(SAC_E_INVALID, "sac_get_histogram: null argument");
    if (h->peak) return fail(SAC_E_INVALID, "sac_get_histogram: use sac_get_histogram_peak for SAC_KIND_PEAK");
    if (h->last_stage != 2) return fail(SAC_E_STATE, "sac_get_histogram: no sample stage has run");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t ng = (size_t)h->cfg.n_grid;
    if (h->comm && h->comm->merged) {                                // summed over all ranks' chains (comm_merge)
        HIP_TRY(copy_sync(h->stream, out, h->comm->dSum, ng * sizeof(double), hipMemcpyDeviceToHost));
        const double steps = (double)h->last_steps, n = h->comm->hTail[18];
        for (size_t i = 0; i < ng; ++i) out[i] = out[i] / steps / n;
        return SAC_OK;
    }
    std::vector<double> tmp(ng);
    HIP_TRY(copy_sync(h->stream, out, h->dHist, ng * sizeof(double), hipMemcpyDeviceToHost));
    for (int c = 1; c < h->hist_copies; ++c) {
        HIP_TRY(copy_sync(h->stream, tmp.data(), h->dHist + (size_t)c * ng, ng * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ng; ++i) out[i] = out[i] + tmp[i];
    }
    const double steps = (double)h->last_steps, nc = (double)h->cfg.n_chains;
    for (size_t i = 0; i < ng; ++i) out[i] = out[i] / steps / nc;   // free:924, then chain mean
    return SAC_OK;
}

extern "C" int sac_get_histogram_sum_device(sac_handle_t *h, void *dst_device, double *steps_times_chains)
{
    if (!h || !dst_device) return fail(SAC_E_INVALID, "sac_get_histogram_sum_device: null argument");
    if (h->peak) return fail(SAC_E_INVALID, "sac_get_histogram_sum_device: not available for SAC_KIND_PEAK yet");
    if (h->last_stage != 2) return fail(SAC_E_STATE, "sac_get_histogram_sum_device: no sample stage has run");
    int rc;
    if ((rc = use_device(h))) return rc;
    // a kernel writes through dst_device: refuse anything that is not device memory of this handle's GPU
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, dst_device) != hipSuccess) {
        (void)hipGetLastError();
        return fail(SAC_E_INVALID, "sac_get_histogram_sum_device: dst_device is not a device pointer (pageable host memory?)");
    }
    if (attr.type != hipMemoryTypeDevice || attr.device != h->cfg.device)
        return fail(SAC_E_INVALID, "sac_get_histogram_sum_device: dst_device must be device memory on GPU %d (memory type %d on "
                    "device %d)", h->cfg.device, (int)attr.type, attr.device);
    const long long ng = h->cfg.n_grid;
    hipLaunchKernelGGL(hist_sum_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, h->stream,
                       (const double *)h->dHist, ng, h->hist_copies, (double *)dst_device);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (steps_times_chains) *steps_times_chains = (double)h->last_steps * (double)h->cfg.n_chains;
    return SAC_OK;
}

extern "C" int sac_comm_unique_id(void *id128)
{
    if (!id128) return fail(SAC_E_INVALID, "sac_comm_unique_id: null argument");
    saccomm::Api &a = saccomm::api();
    if (!a.lib) return fail(SAC_E_STATE, "sac_comm_unique_id: RCCL is not available: %s", a.why);
    saccomm::UniqueId id;
    const int rc = a.GetUniqueId(&id);
    if (rc != 0) return fail(SAC_E_HIP, "ncclGetUniqueId: %s", a.GetErrorString(rc));
    static_assert(sizeof(id) == SAC_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof(id));
    return SAC_OK;
}

extern "C" int sac_comm_init(sac_handle_t *h, const void *id128, int32_t rank, int32_t world)
{
    if (!h || !id128) return fail(SAC_E_INVALID, "sac_comm_init: null argument");
    if (h->comm) return fail(SAC_E_STATE, "sac_comm_init: the handle has a communicator already");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    char msg[320];
    h->comm = saccomm::create(id128, rank, world, (size_t)h->cfg.n_grid * (size_t)h->hist_arrays + SAC_MAX_N_TAU,
                              h->peak ? PEAK_SLOTS : FREE_SLOTS, 1, msg, sizeof(msg));
    if (!h->comm) return fail(SAC_E_HIP, "sac_comm_init: %s", msg);
    return SAC_OK;
}

extern "C" int sac_comm_info(sac_handle_t *h, int32_t *rank, int32_t *world, int64_t *merges, int64_t *allreduce_calls)
{
    if (!h) return fail(SAC_E_INVALID, "sac_comm_info: null handle");
    if (rank) *rank = h->comm ? h->comm->rank : 0;
    if (world) *world = h->comm ? h->comm->world : 1;
    if (merges) *merges = h->comm ? h->comm->merges : 0;
    if (allreduce_calls) *allreduce_calls = h->comm ? h->comm->allreduce_calls : 0;
    return SAC_OK;
}

extern "C" int sac_set_move_log(sac_handle_t *h, int32_t chain, int64_t capacity)
{
    if (!h) return fail(SAC_E_INVALID, "sac_set_move_log: null handle");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->dLog) { hipFree(h->dLog); h->dLog = nullptr; }
    h->log_cap = 0;
    h->log_chain = -1;
    HIP_TRY(hipMemset(h->dLogLen, 0, sizeof(unsigned long long)));
    if (chain < 0 || capacity <= 0) return SAC_OK;
    if (chain >= h->cfg.n_chains)
        return fail(SAC_E_INVALID, "sac_set_move_log: chain %d outside 0..%d", chain, h->cfg.n_chains - 1);
    if (hipMalloc((void **)&h->dLog, (size_t)capacity * sizeof(MoveRec)) != hipSuccess)
        return fail(SAC_E_NOMEM, "sac_set_move_log: hipMalloc of %lld records failed", (long long)capacity);
    HIP_TRY(hipMemset(h->dLog, 0, (size_t)capacity * sizeof(MoveRec)));
    h->log_cap = capacity;
    h->log_chain = chain;
    return SAC_OK;
}

extern "C" int sac_get_move_log(sac_handle_t *h, sac_move_rec_t *out, int64_t capacity, int64_t *n_recorded)
{
    if (!h || !n_recorded) return fail(SAC_E_INVALID, "sac_get_move_log: null argument");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    unsigned long long n = 0;
    HIP_TRY(copy_sync(h->stream, &n, h->dLogLen, sizeof(n), hipMemcpyDeviceToHost));
    *n_recorded = (int64_t)n;
    int64_t m = (int64_t)n;
    if (m > h->log_cap) m = h->log_cap;
    if (m > capacity) m = capacity;
    static_assert(sizeof(MoveRec) == sizeof(sac_move_rec_t), "move record layout");
    if (out && m > 0) HIP_TRY(copy_sync(h->stream, out, h->dLog, (size_t)m * sizeof(MoveRec), hipMemcpyDeviceToHost));
    return SAC_OK;
}
