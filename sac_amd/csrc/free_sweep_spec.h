// sac/free fixed-Theta sweep for runs with FEW chains: one Markov chain per wavefront, C consecutive proposals of that
// chain evaluated per step (gfx950).
//
// An annealing run of the reference is ONE chain, and its Theta ladder is sequential: with a few dozen chains the chip is
// empty and what counts is how fast a single chain advances.  A lone wave of free_stage_kernel needs ~800 cycles per
// proposal -- the dependent path gather -> AXPY -> six tree levels -> compare -> branch -> commit -- whatever the SIMD
// does besides.  Here the wave is cut into C lane groups as in free_sweep_packed.h (LPC = 64 / C lanes, T tau slots per
// lane), but the groups work for the SAME chain: group g evaluates proposal j0 + g of the step against the chain's
// current Gbar, all groups at once, in one instruction stream.  The chain stays the sequential one:
//
//   * the proposals of a step are resolved in order; the first accepted one commits (its group's Gbar_new goes to the
//     other groups by row / half lane swaps, chi2 to scalar registers, the positions to LDS), and the proposals behind it in the
//     step are evaluated AGAIN against the new state -- from the columns they already hold in registers, unless the
//     accepted move touched one of their deltas (then they are decoded again and their columns refetched, the stale
//     rule of free_sweep.h);
//   * a step therefore always consumes C proposals (static ring of prefetched columns, as in the other kernels) and
//     costs 1 + (number of accepts among its first C - 1 proposals) evaluation passes: 1.5 passes per 2 proposals, 2.5
//     per 4 at the tuned acceptance of one half, against 2 and 4 dependent paths of the one-proposal kernel;
//   * draws, word layout, arithmetic and summation order are those of free_sweep.h / the oracle: same chain, bit for
//     bit, including every record of the move log (tests/test_gpu_free_spec.py).
//
// Equal fixed amplitudes (par = 1) and N_tau <= 64 only, like the packed kernel; everything else runs free_stage_kernel.
#pragma once
#include "free_sweep_packed.h"

namespace sacdev {

// one 64-bit value of every lane from lane `src_byte / 4`
__device__ __forceinline__ double bpermute_f64(int src_byte, double v)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src_byte, (int)(uint32_t)b);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src_byte, (int)(uint32_t)(b >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// v of lane group `g` (wave-uniform) in every group, lane q from lane q: row / half swaps on the vector ALU (an LDS
// permute would put ~100 cycles on the path between two evaluation passes)
template <int C>
__device__ __forceinline__ double group_broadcast(double v, int g)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
    if constexpr (C == 4) {
        // v_permlane16_swap a, b with a = b = x: a' = [r0 r0 r2 r2], b' = [r1 r1 r3 r3]
        const auto slo = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto shi = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        lo = (g & 1) ? slo[1] : slo[0];
        hi = (g & 1) ? shi[1] : shi[0];
    }
    // v_permlane32_swap a, b with a = b = x: a' = [h0 h0], b' = [h1 h1]
    const auto slo = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto shi = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const bool up = (C == 4) ? ((g & 2) != 0) : (g != 0);
    lo = up ? slo[1] : slo[0];
    hi = up ? shi[1] : shi[0];
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// One move loop of a sweep of the wave's chain: n_it proposals of NCOL simultaneous delta shifts, C per step.
// Returns the number of accepted proposals.  `c` is used as PackedChain with every lane working for the same chain
// (c.wi = the chain's positions for all lanes; c.g / c.q = lane group and lane within it; counters in slot 0).
template <int NCOL, int DEPTH, int C, int T, bool LOG>
__device__ __forceinline__ int spec_move_loop(PackedChain<C, T> &c, const int MOVE, int n_it, double window)
{
    constexpr int LPC = 64 / C;
    int accepted = 0;
    uint32_t evaluated = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(c.Kb), 0, (int)c.tab_bytes, 0x00020000);

    for (int base = 0; base < n_it; base += 64) {
        c.prio.rotate();
        const int cnt = (n_it - base < 64) ? (n_it - base) : 64;
        const int nsteps = (cnt + C - 1) / C;
        const unsigned long long in_group = (cnt == 64) ? ~0ull : ((1ull << cnt) - 1ull);
        PDecoded<NCOL> d;                        // lane l holds proposal base + l
        decode_packed<NCOL, C, T>(c, MOVE, (uint32_t)(base + (c.lane < cnt ? c.lane : cnt - 1)), window, d);
        unsigned long long inb_bits = __builtin_amdgcn_ballot_w64(d.inb != 0) & in_group;
        unsigned long long stale = 0ull;         // proposals decoded again while their columns were in flight

        TauVec<T> kn[DEPTH][NCOL], ko[DEPTH][NCOL];
        double lnu_r[DEPTH];                     // ln u of the proposal this lane's group evaluates in that step
        uint32_t nx_wo[NCOL], nx_we[NCOL];       // the next step to request, fetched one step ahead of its loads
        double nx_lnu;

        auto src_of = [&](int s) {               // ds_bpermute address of the lane holding this group's proposal of step s
            int j = s * C + c.g;
            j = (j < cnt) ? j : cnt - 1;
            return j * 4;
        };
        auto start = [&](int s) {
            const int src = src_of(s);
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                nx_wo[q] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, d.wold[q]);
                nx_we[q] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, d.weff[q]);
            }
            nx_lnu = bpermute_f64(src, d.lnu);
        };
        auto issue = [&](auto U) {
            constexpr int u = decltype(U)::value;
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                kn[u][q] = load_col_packed<T>(rs, nx_we[q], c.col_bytes, c.lane_off);
                ko[u][q] = load_col_packed<T>(rs, nx_wo[q], c.col_bytes, c.lane_off);
            }
            lnu_r[u] = nx_lnu;
        };
        auto resolve = [&](auto U, int s) {
            constexpr int u = decltype(U)::value;
            const int j0 = s * C;
            const unsigned long long step_bits = (((1ull << C) - 1ull) << j0) & in_group;
            int g = 0;                           // first group of the step not resolved yet
            double dl[NCOL][T];
            bool have_dl = false;
            auto record = [&](int gg, int ev, int ac, double chi2_new) {
                if constexpr (LOG) {
                    if (c.log != nullptr) {
                        if (c.lane == 0 && (long long)c.log_pos < c.log_cap) {
                            MoveRec r;
                            r.chi2_new = ev ? chi2_new : 0.0;
                            r.flags = ev | (ac << 1);
                            r.move = MOVE;
                            c.log[c.log_pos] = r;
                        }
                        c.log_pos += 1;
                    }
                }
            };
            while (true) {
                const unsigned long long pend = step_bits & (~0ull << (j0 + g));
                if ((inb_bits & pend) == 0ull) {     // nothing left to evaluate: the rest of the step is out of bounds
                    for (int gg = g; gg < C; ++gg)
                        if ((step_bits >> (j0 + gg)) & 1ull) record(gg, 0, 0, 0.0);
                    stale &= ~step_bits;
                    return;
                }
                // A (K[:, new] - K[:, old]) per moved delta, once per step and formed inside each branch (a join of the
                // prefetched and the refetched columns would make the compiler drain every outstanding load)
                if ((stale & pend) != 0ull) {        // rare: a delta of a pending proposal moved meanwhile: its columns again
                    stale &= ~pend;
                    const int src = src_of(s);
                    TauVec<T> fn[NCOL], fo[NCOL];
#pragma unroll
                    for (int q = 0; q < NCOL; ++q) {
                        const uint32_t wo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, d.wold[q]);
                        const uint32_t we = (uint32_t)__builtin_amdgcn_ds_bpermute(src, d.weff[q]);
                        fn[q] = load_col_packed<T>(rs, we, c.col_bytes, c.lane_off);
                        fo[q] = load_col_packed<T>(rs, wo, c.col_bytes, c.lane_off);
                    }
#pragma unroll
                    for (int q = 0; q < NCOL; ++q)
#pragma unroll
                        for (int t = 0; t < T; ++t) dl[q][t] = c.A_uniform * (fn[q].v[t] - fo[q].v[t]);
                    have_dl = true;
                } else if (!have_dl) {
#pragma unroll
                    for (int q = 0; q < NCOL; ++q)
#pragma unroll
                        for (int t = 0; t < T; ++t) dl[q][t] = c.A_uniform * (kn[u][q].v[t] - ko[u][q].v[t]);
                    have_dl = true;
                }
                // one pass: every group evaluates its proposal against the current Gbar (free:301-302, 259-261)
                double gnew[T], e2[T];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    double gg_ = c.gbar[t];
#pragma unroll
                    for (int q = 0; q < NCOL; ++q) gg_ = gg_ + dl[q][t];
                    gnew[t] = gg_;
                    const double e = (gg_ - c.G[t]) * c.sinv[t];
                    e2[t] = e * e;
                }
                double term;
                if constexpr (T == 1) term = e2[0];
                else if constexpr (T == 2) term = e2[0] + e2[1];
                else term = (e2[0] + e2[1]) + (e2[2] + e2[3]);
                const double sum = chain_sum<C>(term);               // the group's chi2_new, in every lane of the group
                const bool ok = metropolis(c.chi2, sum, c.inv2theta, lnu_r[u]);
                const unsigned long long okb = __builtin_amdgcn_ballot_w64(ok);
                // in order: out-of-bounds proposals and rejections are final, the first acceptance commits
                int acc_g = -1;
                for (int gg = g; gg < C; ++gg) {
                    if (!((step_bits >> (j0 + gg)) & 1ull)) break;
                    const int ev = (int)((inb_bits >> (j0 + gg)) & 1ull);
                    const int ac = ev & (int)((okb >> (gg * LPC)) & 1ull);
                    evaluated += (uint32_t)ev;
                    if constexpr (LOG) record(gg, ev, ac, ev ? readlane_f64(sum, gg * LPC) : 0.0);
                    if (ac) { acc_g = gg; break; }
                }
                if (acc_g < 0) { stale &= ~step_bits; return; }
                // commit (free:310-318)
                const int j = j0 + acc_g;
#pragma unroll
                for (int t = 0; t < T; ++t) c.gbar[t] = group_broadcast<C>(gnew[t], acc_g);
                const double chi2_new = readlane_f64(sum, acc_g * LPC);
                c.chi2 = chi2_new;
                c.chi2_min = __builtin_fmin(chi2_new, c.chi2_min);
                int s_k[NCOL];
#pragma unroll
                for (int q = 0; q < NCOL; ++q) {
                    s_k[q] = __builtin_amdgcn_readlane(d.k[q], j);
                    const int s_new = __builtin_amdgcn_readlane(d.wnew[q], j);
                    if (c.lane == 0) c.wi[s_k[q]] = (uint32_t)s_new;
                }
                accepted += 1;
                // other lanes read these positions below: see free_sweep_packed.h (store-to-load forwarding across lanes)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                bool same = false;
#pragma unroll
                for (int a = 0; a < NCOL; ++a)
#pragma unroll
                    for (int b = 0; b < NCOL; ++b) same = same || (d.k[a] == s_k[b]);
                const unsigned long long hits = __builtin_amdgcn_ballot_w64(same) & ((~0ull << j) << 1) & in_group;
                if (hits != 0ull) {
                    // the steps up to s + DEPTH have their columns (or column indices) requested already
                    const int lim = (s + DEPTH + 1) * C;
                    const unsigned long long inflight = (lim >= 64) ? ~0ull : ((1ull << lim) - 1ull);
                    stale |= hits & inflight;
                    if ((hits >> c.lane) & 1ull) {
#pragma unroll
                        for (int q = 0; q < NCOL; ++q) d.wold[q] = (int)c.wi[d.k[q]];
                        compute_new_packed<NCOL>(d, 0, c.gmax);
                    }
                    inb_bits = __builtin_amdgcn_ballot_w64(d.inb != 0) & in_group;
                }
                g = acc_g + 1;
                if (g >= C || !((step_bits >> (j0 + g)) & 1ull)) { stale &= ~step_bits; return; }
            }
        };

        [&]<int... Us>(std::integer_sequence<int, Us...>) {
            ((start(Us), issue(std::integral_constant<int, Us>{}), __builtin_amdgcn_sched_barrier(0)), ...);
        }(std::make_integer_sequence<int, DEPTH>{});
        start(DEPTH);

        for (int s0 = 0; s0 < nsteps; s0 += DEPTH) {
            [&]<int... Us>(std::integer_sequence<int, Us...>) {
                (((s0 + Us < nsteps) ? (resolve(std::integral_constant<int, Us>{}, s0 + Us), 0) : 0,
                  issue(std::integral_constant<int, Us>{}), start(s0 + Us + DEPTH + 1)), ...);
            }(std::make_integer_sequence<int, DEPTH>{});
        }
    }
    c.n_eval[0] += evaluated;
    c.n_cols[0] += evaluated * (uint32_t)(2 * NCOL);
    c.n_acc[0] += (uint32_t)accepted;
    c.n_accmoves[0] += (uint32_t)accepted * (uint32_t)NCOL;
    return accepted;
}

// LDS of one chain: positions, then 12 doubles (windows[4], accept accumulators[6], sum chi2, pad)
__host__ __device__ constexpr size_t spec_lds_bytes(int wi_stride) { return (size_t)wi_stride * 4u + 12u * 8u; }

// The stage kernel of free_stage_kernel<.., UA = true, PAR2 = false> for one chain per 64-thread block, C proposals per step.
template <int C, int T, bool LOG>
__global__ void __launch_bounds__(64) free_stage_spec_kernel(const FreeParams p)
{
    constexpr int LPC = 64 / C;
    extern __shared__ __align__(16) unsigned char s_spec[];
    const int chain = (int)blockIdx.x;
    if (chain >= p.n_chains) return;
    const int lane = lane_id();
    uint32_t *s_wi = reinterpret_cast<uint32_t *>(s_spec);
    double *s_sc = reinterpret_cast<double *>(s_spec + (size_t)p.wi_stride * 4u);

    PackedChain<C, T> c;
    c.Kb = reinterpret_cast<const char *>(p.K);
    c.col_bytes = (uint32_t)p.ntp * 8u;
    c.tab_bytes = (uint32_t)((unsigned long long)p.n_grid * (unsigned long long)p.ntp * 8ull);
    c.lane = lane;
    c.g = lane / LPC;
    c.q = lane % LPC;
    c.wi = s_wi;
    c.dummy = reinterpret_cast<uint32_t *>(s_sc + 11);
    c.prio.init(p.prio_shift);
    c.n_omega = p.n_omega;
    c.gmax = p.gmax;
    c.inv2theta = p.inv2theta;
    c.A_uniform = p.A_uniform;
    c.key0 = p.seed_lo;
    c.key1 = p.seed_hi;
    c.chain_id = p.chain_offset + (uint32_t)chain;
#pragma unroll
    for (int g = 0; g < C; ++g) c.n_eval[g] = c.n_acc[g] = c.n_accmoves[g] = c.n_cols[g] = 0;
    const int slot0 = T * c.q;
    c.lane_off = (uint32_t)((slot0 + T <= p.ntp) ? slot0 : p.ntp - T) * 8u;
    double *gbar_g = p.gbar + (size_t)chain * SAC_MAX_N_TAU;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int j = slot0 + t;
        const bool in = (j < SAC_MAX_N_TAU);
        c.gbar[t] = in ? gbar_g[in ? j : 0] : 0.0;
        c.G[t] = in ? p.G[in ? j : 0] : 0.0;
        c.sinv[t] = in ? p.sinv[in ? j : 0] : 0.0;
    }
    const uint32_t *wi_g = p.wi + (size_t)chain * (size_t)p.wi_stride;
    for (int i = lane; i < p.n_omega; i += 64) s_wi[i] = wi_g[i];
    ChainScalars *scp = p.sc + chain;
    c.chi2 = scp->chi2;
    c.chi2_min = scp->chi2_min;
    double *win = s_sc, *acc = win + 4, *sum_chi2 = win + 10;
    const bool zero = (p.flags & STAGE_ZERO_ACC) != 0;
    if (lane < 4) win[lane] = scp->windows[lane];
    if (lane < 6) acc[lane] = zero ? 0.0 : scp->acc[lane];
    if (lane == 0) *sum_chi2 = zero ? 0.0 : scp->sum_chi2;
    unsigned long long sweep = scp->sweep;
    c.log = nullptr;
    c.log_cap = 0;
    c.log_pos = 0;
    c.log_g = -1;
    if constexpr (LOG) {
        if (p.log_chain == chain) {
            c.log = p.log;
            c.log_cap = p.log_cap;
            c.log_pos = *p.log_len;
            c.log_g = 0;
        }
    }
    __syncthreads();

    const int n = p.n_omega;
    constexpr int D1 = (T <= 2) ? 8 : 4, D2 = (T <= 2) ? 4 : 2, D3 = (T <= 2) ? 4 : 2;
    for (int s = 0; s < p.n_sweeps; ++s) {
        c.sweep_lo = (uint32_t)sweep;
        c.sweep_hi = (uint32_t)(sweep >> 32) & 0xFFFFu;
        int a = spec_move_loop<1, D1, C, T, LOG>(c, 1, n, win[0]);                 // free:829-830
        if (lane == 0) acc[0] += (double)a / (double)n;
        a = spec_move_loop<2, D2, C, T, LOG>(c, 2, n / 2, win[1]);                 // free:832-833
        if (lane == 0) acc[1] += (double)(2 * a) / (double)(n / 2);
        a = spec_move_loop<3, D3, C, T, LOG>(c, 3, n / 3, 0.0);                    // free:835-836
        if (lane == 0) acc[2] += (double)a / (double)(n / 3);
        if (p.flags & STAGE_SAMPLE) {
            double *hist = p.hist + (size_t)(chain % p.hist_copies) * (size_t)p.n_grid;   // free:918
            if (p.hist_mode == SAC_HIST_REFERENCE) {
                if (p.n_grid < (1ll << 21))
                    hist_reference<uint32_t, 32, 200 + 8 * C + T>(s_wi, n, nullptr, p.A_uniform, 1, hist, p.hist_exclusive, lane);
                else
                    hist_reference<uint64_t, 32, 200 + 8 * C + T>(s_wi, n, nullptr, p.A_uniform, 1, hist, p.hist_exclusive, lane);
            } else {
                hist_accumulate(s_wi, n, nullptr, p.A_uniform, 1, hist, lane);
            }
            if (lane == 0) *sum_chi2 += c.chi2;                                    // free:919
        }
        sweep += 1;
    }

    if ((p.flags & STAGE_RETUNE) && lane == 0) {                                   // free:883-895
        for (int k = 0; k < 6; ++k) acc[k] = acc[k] / p.retune_div;
        for (int k = 0; k < 4; ++k) {
            const double r = acc[k];
            if (r > 0.8) win[k] *= 2;
            else if (r < 0.2) win[k] /= 2;
            else if (r > 0.55) win[k] *= 1.2;
            else if (r < 0.45) win[k] /= 1.2;
        }
    }
    __syncthreads();

    uint32_t *wi_o = p.wi + (size_t)chain * (size_t)p.wi_stride;
    for (int i = lane; i < p.n_omega; i += 64) wi_o[i] = s_wi[i];
    if (c.g == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int j = slot0 + t;
            if (j < p.n_tau) gbar_g[j] = c.gbar[t];
        }
    }
    if (lane == 0) {
        scp->chi2 = c.chi2;
        scp->chi2_min = c.chi2_min;
        for (int k = 0; k < 4; ++k) scp->windows[k] = win[k];
        for (int k = 0; k < 6; ++k) scp->acc[k] = acc[k];
        scp->sum_chi2 = *sum_chi2;
        scp->sweep = sweep;
        scp->counters[0] += (unsigned long long)p.n_sweeps * (unsigned long long)(n + n / 2 + n / 3);
        scp->counters[1] += c.n_eval[0];
        scp->counters[2] += c.n_acc[0];
        scp->counters[3] += c.n_accmoves[0];
        scp->counters[4] += c.n_cols[0];
        if constexpr (LOG) {
            if (c.log != nullptr) *p.log_len = c.log_pos;
        }
    }
}

}  // namespace sacdev
