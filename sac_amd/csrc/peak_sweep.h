// sac/peak fixed-Theta sweep pieces that the free sweep does not have (gfx950).
//
// Reference behaviour: /root/reference/sac/peak/sac_peak.jl ("peak:NNN").
//   four delta groups in one position/amplitude array (peak:166-169): peak+, cont+, peak-, cont-
//   two kernel tables Kp / Kn on the half-offset grid (peak:248-276)
//   continuum and peak frequency moves = the free move loop on an index sub-range with bounds
//     (peak:382-700; bounds are refreshed only AFTER a loop, peak:440-444, 693-697)
//   weight-transfer moves between the axes (peak:706-996): this file
//   Gbar / chi2 refreshed from scratch before every sweep (peak:1072-1073, 1120-1121): this file
#pragma once
#include "free_sweep.h"

namespace sacdev {

struct PeakScalars {
    double chi2, chi2_min;
    double windows[11];
    double acc[11];             // accept_rates accumulators (peak:62)
    double sum_chi2;            // sampled_chi2 accumulator (peak:1145)
    double edge_sum[2];         // edge_res[1, :] accumulators (peak:1146-1147)
    unsigned long long sweep;
    unsigned long long counters[5];
    int bounds[4];              // wi_pp, wi_pc, wi_np, wi_nc (peak:14-17)
};

struct PeakParams {
    FreeParams f;               // K = Kp; shared fields (state arrays, histogram base, Philox key ...)
    const double *Kn;
    PeakScalars *psc;
    int n_p, n_cont;            // N_p, N_omega
    int wi0, wim;
    int sample_neg;             // both axes sampled (not symm, not bosonic)
    int fix_edge, indiv_update;
    int transfer;               // run_updates' `transfer` argument (peak:1006)
    int refresh_bounds;         // adjust_windows refreshes wi_pp/wi_np every sweep (peak:1074-1075)
};

// wave-uniform min / max of wi[first .. first+count)
__device__ __forceinline__ int group_min(const uint32_t *wi, int first, int count, int lane)
{
    int m = 0x7fffffff;
    for (int i = lane; i < count; i += 64) { const int v = (int)wi[first + i]; m = v < m ? v : m; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(m, o, 64); m = v < m ? v : m; }
    return __builtin_amdgcn_readfirstlane(m);
}
__device__ __forceinline__ int group_max(const uint32_t *wi, int first, int count, int lane)
{
    int m = -1;
    for (int i = lane; i < count; i += 64) { const int v = (int)wi[first + i]; m = v > m ? v : m; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(m, o, 64); m = v > m ? v : m; }
    return __builtin_amdgcn_readfirstlane(m);
}

// peak:347-353 calc_Gbar and peak:357-363 calc_chi2: per group sequentially over its deltas
// (mul, then add), the four group sums added left to right; chi2 by the canonical tree.
template <int NTL>
__device__ __forceinline__ void peak_refresh_gbar(Chain<NTL> &c, const char *Kp, const char *Kn,
                                                  const double *A, int n_p, int n_cont, int sample_neg)
{
    double tot[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) tot[t] = 0.0;
    const int first[4] = {0, n_p, n_p + n_cont, 2 * n_p + n_cont};
    const int count[4] = {n_p, n_cont, n_p, n_cont};
    const char *keep = c.Kb;
    for (int grp = 0; grp < (sample_neg ? 4 : 2); ++grp) {
        c.Kb = (grp < 2) ? Kp : Kn;
        double acc[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[t] = 0.0;
        const int f = first[grp], n = count[grp];
        int i = 0;
        constexpr int B = 8;                         // columns in flight (16 was measured: slower, at full occupancy and for a lone chain)
        for (; i + B <= n; i += B) {
            double col[B][NTL], a[B];
#pragma unroll
            for (int u = 0; u < B; ++u) {
                const int w = __builtin_amdgcn_readfirstlane((int)c.wi[f + i + u]);
                load_col<NTL>(c, w, col[u]);
                a[u] = A[f + i + u];
            }
#pragma unroll
            for (int u = 0; u < B; ++u)
#pragma unroll
                for (int t = 0; t < NTL; ++t) acc[t] = acc[t] + col[u][t] * a[u];
        }
        for (; i < n; ++i) {
            double col[NTL];
            const int w = __builtin_amdgcn_readfirstlane((int)c.wi[f + i]);
            load_col<NTL>(c, w, col);
            const double a = A[f + i];
#pragma unroll
            for (int t = 0; t < NTL; ++t) acc[t] = acc[t] + col[t] * a;
        }
#pragma unroll
        for (int t = 0; t < NTL; ++t) tot[t] = (grp == 0) ? acc[t] : (tot[t] + acc[t]);
    }
    c.Kb = keep;
    double term = 0.0;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        c.gbar[t] = tot[t];
        const double e = (tot[t] - c.G[t]) * c.sinv[t];
        const double e2 = e * e;
        term = (t == 0) ? e2 : (term + e2);
    }
    c.chi2 = wave_sum_uniform(term);
}

// ---- weight transfer between the axes -------------------------------------------------------
// NPAIR = 1: cont_Aw_transfer (peak:706-792) or peak_Aw_transfer (peak:798-880)
// NPAIR = 2: Aw_transfer (peak:882-996): continuum pair (shift dc) and peak pair (shift dp) together.
// Element e = 2*pair + side; side 0 = positive axis (Kp), side 1 = negative axis (Kn).

template <int NPAIR>
struct XferSpec {
    int first[2 * NPAIR], count[2 * NPAIR];     // index range each element is drawn from
    int lo[2 * NPAIR], hi[2 * NPAIR];           // allowed new positions (stale bounds, as the reference)
    double window[NPAIR];
    int move_id;
};

template <int NPAIR>
struct DecodedX {
    int k[2 * NPAIR], wold[2 * NPAIR], wnew[2 * NPAIR];
    int sd[NPAIR];
    int inb;
    double lnu, A[2 * NPAIR], Ap[2 * NPAIR];
};

template <int NPAIR>
__device__ __forceinline__ void xfer_compute(DecodedX<NPAIR> &d, const XferSpec<NPAIR> &sp)
{
    bool ok = true;
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) {
        const int p = 2 * pr, n = 2 * pr + 1;
        d.wnew[p] = d.wold[p] + d.sd[pr];                           // peak:730-731, 915-916, 925-926
        d.wnew[n] = d.wold[n] - d.sd[pr];
        ok = ok && d.wnew[p] >= sp.lo[p] && d.wnew[p] <= sp.hi[p] && d.wnew[n] >= sp.lo[n] && d.wnew[n] <= sp.hi[n];
        // peak:745: dA = dw (A_n - A_p) / (2 dw + w_p + w_n), dw signed
        const double dA = ((double)d.sd[pr] * (d.A[n] - d.A[p])) / (double)(2 * d.sd[pr] + d.wold[p] + d.wold[n]);
        d.Ap[p] = d.A[p] + dA;
        d.Ap[n] = d.A[n] - dA;
    }
#pragma unroll
    for (int e = 0; e < 2 * NPAIR; ++e)
        if (d.Ap[e] < 0) ok = false;                                // peak:751-755, 952-954
    d.inb = ok ? 1 : 0;
}

// signed shift of the transfer moves: magnitude as free:285, NEGATIVE when the sign draw is < 1/2
// (peak:726-728 `if rand() < 0.5: dw *= -1`)
__device__ __forceinline__ int xfer_shift(double u_mag, double u_sign, double window)
{
    const double f = __builtin_floor(u_mag * window);
    const int mag = (f < 1.0e9) ? 1 + (int)f : (1 << 30);
    return (u_sign < 0.5) ? -mag : mag;
}

template <int NPAIR, int DEPTH, int NTL, bool LOG>
__device__ __forceinline__ int xfer_loop(Chain<NTL> &c, const char *Kp, const char *Kn, double *A,
                                         const XferSpec<NPAIR> &sp, int n_it)
{
    constexpr int NE = 2 * NPAIR;
    int accepted = 0;
    const char *keep = c.Kb;
    for (int base = 0; base < n_it; base += 64) {
        c.prio.rotate();
        const int cnt = (n_it - base < 64) ? (n_it - base) : 64;
        DecodedX<NPAIR> d;
        {
            const uint32_t iter = (uint32_t)(base + (c.lane < cnt ? c.lane : cnt - 1));
            const uint32_t c2 = c.sweep_hi | ((uint32_t)sp.move_id << 16);
            const u32x4 b0 = philox4x32_10(iter, c.sweep_lo, c2, c.chain_id, c.key0, c.key1);
            const u32x4 b1 = philox4x32_10(iter, c.sweep_lo, c2 | (1u << 24), c.chain_id, c.key0, c.key1);
            if constexpr (NPAIR == 1) {
                d.k[0] = sp.first[0] + index_from_word(b0.x, (uint32_t)sp.count[0]);
                d.k[1] = sp.first[1] + index_from_word(b0.y, (uint32_t)sp.count[1]);
                d.sd[0] = xfer_shift(u01(b0.z, b0.w), u01(b1.x, b1.y), sp.window[0]);
                d.lnu = log_u01(u01(b1.z, b1.w));
            } else {
                const u32x4 b2 = philox4x32_10(iter, c.sweep_lo, c2 | (2u << 24), c.chain_id, c.key0, c.key1);
                const u32x4 b3 = philox4x32_10(iter, c.sweep_lo, c2 | (3u << 24), c.chain_id, c.key0, c.key1);
                d.k[0] = sp.first[0] + index_from_word(b0.x, (uint32_t)sp.count[0]);   // peak:891-898
                d.k[1] = sp.first[1] + index_from_word(b0.y, (uint32_t)sp.count[1]);
                d.k[2] = sp.first[2] + index_from_word(b0.z, (uint32_t)sp.count[2]);
                d.k[3] = sp.first[3] + index_from_word(b0.w, (uint32_t)sp.count[3]);
                d.sd[0] = xfer_shift(u01(b1.x, b1.y), u01(b2.x, b2.y), sp.window[0]);  // peak:903-912
                d.sd[1] = xfer_shift(u01(b1.z, b1.w), u01(b2.z, b2.w), sp.window[1]);
                d.lnu = log_u01(u01(b3.x, b3.y));
            }
#pragma unroll
            for (int e = 0; e < NE; ++e) { d.wold[e] = (int)c.wi[d.k[e]]; d.A[e] = A[d.k[e]]; }
            xfer_compute<NPAIR>(d, sp);
        }
        // only proposals that reach the evaluation enter the prefetch ring (see move_loop_live in free_sweep.h)
        const unsigned long long in_group = (cnt == 64) ? ~0ull : ((1ull << cnt) - 1ull);
        unsigned long long pend = __builtin_amdgcn_ballot_w64(d.inb != 0) & in_group;
        unsigned long long inring = 0ull, stale = 0ull;
        int last_issued = -1, hold = 0, log_next = 0;
        double kn[DEPTH][NE][NTL], ko[DEPTH][NE][NTL];
        int is_j[DEPTH], is_old[DEPTH][NE], is_new[DEPTH][NE];

        auto fetch = [&](int e, int col, double (&out)[NTL]) {
            c.Kb = (e & 1) ? Kn : Kp;
            load_col<NTL>(c, col, out);
        };
        auto log_skipped = [&](int upto) {
            if constexpr (LOG) {
                if (c.log != nullptr) {
                    for (int s_ = log_next; s_ < upto; ++s_) {
                        if (c.lane == 0 && (long long)c.log_pos < c.log_cap) {
                            MoveRec r;
                            r.chi2_new = 0.0;
                            r.flags = 0;
                            r.move = sp.move_id;
                            c.log[c.log_pos] = r;
                        }
                        c.log_pos += 1;
                    }
                }
                log_next = upto;
            }
        };
        auto issue = [&](auto U) {
            constexpr int u = decltype(U)::value;
            int j = 64;
            if (hold == 0 && pend != 0ull) {
                j = (int)__builtin_ctzll(pend);
                pend &= pend - 1ull;
                inring |= 1ull << j;
                last_issued = j;
            }
            hold = 0;
            is_j[u] = j;
            const bool live = j < 64;
            const int jj = live ? j : 0;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int o = __builtin_amdgcn_readlane(d.wold[e], jj);
                const int nw = __builtin_amdgcn_readlane(d.wnew[e], jj);
                is_old[u][e] = o;
                is_new[u][e] = nw;
                fetch(e, live ? nw : 0, kn[u][e]);        // empty slot: parked on column 0
                fetch(e, live ? o : 0, ko[u][e]);
            }
        };
        auto resolve = [&](auto U) {
            constexpr int u = decltype(U)::value;
            const int j = is_j[u];
            if (j >= 64) return;
            inring &= ~(1ull << j);
            log_skipped(j);
            int inb = 1;
            int s_old[NE], s_new[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) { s_old[e] = is_old[u][e]; s_new[e] = is_new[u][e]; }
            const bool was_stale = (stale != 0ull) && (((stale >> j) & 1ull) != 0ull);   // in-flight proposals only
            if (was_stale) {
                stale &= ~(1ull << j);
                inb = __builtin_amdgcn_readlane(d.inb, j);
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    s_old[e] = __builtin_amdgcn_readlane(d.wold[e], j);
                    s_new[e] = __builtin_amdgcn_readlane(d.wnew[e], j);
                }
            }
            int acc = 0;
            double chi2_new = 0.0;
            if (inb) {
                double sA[NE], sAp[NE];
#pragma unroll
                for (int e = 0; e < NE; ++e) { sA[e] = readlane_f64(d.A[e], j); sAp[e] = readlane_f64(d.Ap[e], j); }
                double gnew[NTL], term = 0.0;
                // Gbar_new and the chi2 terms are formed inside each branch: a join of the prefetched and the refetched
                // columns would make the compiler wait for EVERY outstanding load (vmcnt(0)) at each proposal, i.e. drain
                // the prefetch ring (as in free_sweep.h)
                auto form = [&](const double (&fn)[NE][NTL], const double (&fo)[NE][NTL]) {
#pragma unroll
                    for (int t = 0; t < NTL; ++t) {                   // peak:758-764, 957-967
                        double g = c.gbar[t];
#pragma unroll
                        for (int pr = 0; pr < NPAIR; ++pr) {
                            const int p = 2 * pr, n = 2 * pr + 1;
                            g = (((g + (sAp[p] * fn[p][t])) + (sAp[n] * fn[n][t])) - (sA[p] * fo[p][t])) - (sA[n] * fo[n][t]);
                        }
                        gnew[t] = g;
                        const double e_ = (g - c.G[t]) * c.sinv[t];
                        const double e2 = e_ * e_;
                        term = (t == 0) ? e2 : (term + e2);
                    }
                };
                // (the two distinct asm comments keep the optimiser from sinking the identical arithmetic of the branches
                // into their join, which brings the full wait back)
                if (!was_stale) {
                    form(kn[u], ko[u]);
                    asm volatile("; xfer: prefetched columns" : "+v"(term));
                } else {
                    double fn[NE][NTL], fo[NE][NTL];
#pragma unroll
                    for (int e = 0; e < NE; ++e) { fetch(e, s_new[e], fn[e]); fetch(e, s_old[e], fo[e]); }
                    form(fn, fo);
                    asm volatile("; xfer: refetched columns" : "+v"(term));
                }
                chi2_new = wave_sum_uniform(term);
                acc = lane63_true(metropolis(c.chi2, chi2_new, c.inv2theta, readlane_f64(d.lnu, j))) ? 1 : 0;
                c.n_eval += 1;
                c.n_cols += 2 * NE;
                if (acc) {
#pragma unroll
                    for (int t = 0; t < NTL; ++t) c.gbar[t] = gnew[t];
                    c.chi2 = chi2_new;
                    if (chi2_new < c.chi2_min) c.chi2_min = chi2_new;
                    int s_k[NE];
#pragma unroll
                    for (int e = 0; e < NE; ++e) s_k[e] = __builtin_amdgcn_readlane(d.k[e], j);
                    if (c.lane == 0) {
#pragma unroll
                        for (int e = 0; e < NE; ++e) { c.wi[s_k[e]] = (uint32_t)s_new[e]; A[s_k[e]] = sAp[e]; }
                    }
                    // A lives in global memory: later lanes of this wave re-read what lane 0 just stored
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    accepted += 1;
                    c.n_acc += 1;
                    c.n_accmoves += NE;
                    bool hit = false;
#pragma unroll
                    for (int e = 0; e < NE; ++e) hit = hit || (d.k[e] == s_k[e]);   // groups are disjoint
                    const unsigned long long above = (~0ull << j) << 1;
                    const unsigned long long hits = __builtin_amdgcn_ballot_w64(hit) & above & in_group;
                    if (hits != 0ull) {
                        if ((hits >> c.lane) & 1ull) {
#pragma unroll
                            for (int e = 0; e < NE; ++e) { d.wold[e] = (int)c.wi[d.k[e]]; d.A[e] = A[d.k[e]]; }
                            xfer_compute<NPAIR>(d, sp);
                        }
                        stale |= hits & inring;                               // columns in flight: refetched when resolved
                        const unsigned long long now = __builtin_amdgcn_ballot_w64(d.inb != 0) & above & in_group;
                        const unsigned long long fresh = now & ~inring;
                        const unsigned long long arrivals = fresh & ~pend;
                        if (arrivals != 0ull && (int)__builtin_ctzll(arrivals) < last_issued) {
                            pend = now;                                       // start the ring again behind proposal j (rare)
                            inring = 0ull;
                            stale = 0ull;
                            last_issued = j;
                            hold = 1;
#pragma unroll
                            for (int v = 0; v < DEPTH; ++v) is_j[v] = 64;
                        } else {
                            pend = fresh;
                        }
                    }
                }
            }
            if constexpr (LOG) {
                if (c.log != nullptr) {
                    if (c.lane == 0 && (long long)c.log_pos < c.log_cap) {
                        MoveRec r;
                        r.chi2_new = inb ? chi2_new : 0.0;
                        r.flags = (inb ? 1 : 0) | (acc ? 2 : 0);
                        r.move = sp.move_id;
                        c.log[c.log_pos] = r;
                    }
                    c.log_pos += 1;
                }
                log_next = j + 1;
            }
        };
        [&]<int... Us>(std::integer_sequence<int, Us...>) {
            (issue(std::integral_constant<int, Us>{}), ...);
        }(std::make_integer_sequence<int, DEPTH>{});
        while (true) {
            [&]<int... Us>(std::integer_sequence<int, Us...>) {
                ((resolve(std::integral_constant<int, Us>{}), issue(std::integral_constant<int, Us>{})), ...);
            }(std::make_integer_sequence<int, DEPTH>{});
            if (pend == 0ull && inring == 0ull) break;
        }
        log_skipped(cnt);
    }
    c.Kb = keep;
    return accepted;
}

}  // namespace sacdev
