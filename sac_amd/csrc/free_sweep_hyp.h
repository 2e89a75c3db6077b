// sac/free and sac/peak move loops for runs with FEW chains: one Markov chain per wavefront, lane = tau as in free_sweep.h,
// but the proposals are resolved two at a time as a HYPOTHESIS PAIR (gfx950).
//
// A lone wave of move_loop needs ~750 cycles per proposal although it issues only ~70 instructions for it: the path
// Gbar + A dK -> residual -> six dependent tree levels -> compare -> scalar branch is one long dependency chain and the SIMD
// has nothing else to issue.  Lane-group speculation (free_sweep_spec.h) fills idle LANES and is bounded by the acceptance
// rate the windows are tuned to (one half): proposal j+1 evaluated "as if j were rejected" is thrown away every second time.
// Here idle ISSUE SLOTS are filled instead, and nothing is thrown away: for two consecutive proposals a, b the wave forms
//     Gbar_a   = Gbar   + inc_a                  chi2_a
//     Gbar_b|r = Gbar   + inc_b                  chi2_b if a is rejected
//     Gbar_b|a = Gbar_a + inc_b                  chi2_b if a is accepted
// as three independent instruction streams (three trees interleaved by the scheduler, each in the canonical order), takes
// the three Metropolis tests in lane 63 -- b's test after an accepted a compares against chi2_a, which is what the
// sequential chain's chi2 would be -- and commits a, then b under the branch a decided.  Every step advances the chain
// by two proposals at the latency of about one.  The pair is formed only when both proposals reach the evaluation, their
// columns in the ring are current and they move different deltas; every other case goes through the one-proposal path of
// move_loop, so the chain is exactly the sequential one: draws, arithmetic, summation order, move log (tests/test_gpu_free_hyp.py).
#pragma once
#include "free_sweep.h"

namespace sacdev {

template <int NCOL, int DEPTH, int NTL, bool LOG, bool UA>
__device__ __forceinline__ int move_loop_hyp(Chain<NTL> &c, const int MOVE, int n_it, double window)
{
    static_assert(DEPTH % 2 == 0 && DEPTH >= 2, "pairs of ring slots");
    int accepted = 0, n_eval = 0;
    for (int base = 0; base < n_it; base += 64) {
        c.prio.rotate();
        const int cnt = (n_it - base < 64) ? (n_it - base) : 64;
        Decoded<NCOL, UA> d;
        decode<NCOL, NTL, UA>(c, MOVE, (uint32_t)(base + (c.lane < cnt ? c.lane : cnt - 1)), window, d);
        // bit l: proposal l moves a delta that proposal l-1 moves too.  The indices come from the draws alone, so this
        // never changes within the group.
        unsigned long long conflict;
        {
            bool cf = false;
#pragma unroll
            for (int b = 0; b < NCOL; ++b) {
                const int kprev = __builtin_amdgcn_update_dpp(-1, d.k[b], 0x138, 0xF, 0xF, false);   // wave_shr:1, lane 0 keeps -1
#pragma unroll
                for (int a = 0; a < NCOL; ++a) cf = cf || (d.k[a] == kprev);
            }
            conflict = __builtin_amdgcn_ballot_w64(cf);
        }

        double kn[DEPTH][NCOL][NTL], ko[DEPTH][NCOL][NTL];
        int is_inb[DEPTH], is_old[DEPTH][NCOL], is_new[DEPTH][NCOL];
        unsigned long long stale = 0ull;

        auto issue = [&](auto U, int j) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            const int jj = (j < cnt) ? j : cnt - 1;
            const int inb = __builtin_amdgcn_readlane(d.inb, jj);
            is_inb[u] = inb;
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                const int o = __builtin_amdgcn_readlane(d.wold[q], jj);
                const int nw = inb ? __builtin_amdgcn_readlane(d.wnew[q], jj) : o;
                is_old[u][q] = o;
                is_new[u][q] = nw;
                if constexpr (UA) {
                    load_col<NTL>(c, nw, kn[u][q]);
                    load_col<NTL>(c, o, ko[u][q]);
                } else {
                    const bool live = inb && (j < cnt);
                    load_col<NTL>(c, live ? nw : 0, kn[u][q]);
                    load_col<NTL>(c, live ? o : 0, ko[u][q]);
                }
            }
        };
        auto record = [&](int ev, int ac, double chi2_new) __attribute__((always_inline)) {
            if constexpr (LOG) {
                if (c.log != nullptr) {
                    if (c.lane == 0 && (long long)c.log_pos < c.log_cap) {
                        MoveRec r;
                        r.chi2_new = ev ? chi2_new : 0.0;
                        r.flags = (ev ? 1 : 0) | (ac ? 2 : 0);
                        r.move = MOVE;
                        c.log[c.log_pos] = r;
                    }
                    c.log_pos += 1;
                }
            }
        };
        // free:310-318 for proposal j; `inflight` = the proposals behind j whose columns are requested already, as a
        // mask relative to j
        auto commit = [&](int j, const double (&gnew)[NTL], double chi2_new, const int (&s_new)[NCOL],
                          unsigned long long inflight) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NTL; ++t) c.gbar[t] = gnew[t];
            c.chi2 = chi2_new;
            if (chi2_new < c.chi2_min) c.chi2_min = chi2_new;
            int s_k[NCOL];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) s_k[q] = __builtin_amdgcn_readlane(d.k[q], j);
            if (c.lane == 0) {
#pragma unroll
                for (int q = 0; q < NCOL; ++q) c.wi[s_k[q]] = (uint32_t)s_new[q];
            }
            accepted += 1;
            bool same = false;
#pragma unroll
            for (int a = 0; a < NCOL; ++a)
#pragma unroll
                for (int b = 0; b < NCOL; ++b) same = same || (d.k[a] == s_k[b]);
            const unsigned long long hits = __builtin_amdgcn_ballot_w64(same) & ((~0ull << j) << 1);
            if (hits != 0ull) {
                stale |= hits & (inflight << j);
                if ((hits >> c.lane) & 1ull) {
#pragma unroll
                    for (int q = 0; q < NCOL; ++q) d.wold[q] = (int)c.wi[d.k[q]];
                    compute_new<NCOL, UA>(d, c.lo, c.hi);
                }
            }
        };
        // the one-proposal path of move_loop (free_sweep.h)
        auto single = [&](auto U, int j, unsigned long long inflight) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            int inb = is_inb[u];
            int s_old[NCOL], s_new[NCOL];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) { s_old[q] = is_old[u][q]; s_new[q] = is_new[u][q]; }
            const bool was_stale = (stale != 0ull) && (((stale >> j) & 1ull) != 0ull);
            if (was_stale) {
                stale &= ~(1ull << j);
                inb = __builtin_amdgcn_readlane(d.inb, j);
#pragma unroll
                for (int q = 0; q < NCOL; ++q) {
                    s_old[q] = __builtin_amdgcn_readlane(d.wold[q], j);
                    s_new[q] = __builtin_amdgcn_readlane(d.wnew[q], j);
                }
            }
            int acc = 0;
            double chi2_new = 0.0;
            if (inb) {
                double s_A[NCOL];
#pragma unroll
                for (int q = 0; q < NCOL; ++q) {
                    if constexpr (UA) s_A[q] = c.A_uniform;
                    else s_A[q] = readlane_f64(d.A[q], j);
                }
                double inc[NCOL][NTL];
                if (!was_stale) {
                    increments<NCOL, NTL>(s_A, kn[u], ko[u], inc);
                } else {
                    double fn[NCOL][NTL], fo[NCOL][NTL];
#pragma unroll
                    for (int q = 0; q < NCOL; ++q) {
                        load_col<NTL>(c, s_new[q], fn[q]);
                        load_col<NTL>(c, s_old[q], fo[q]);
                    }
                    increments<NCOL, NTL>(s_A, fn, fo, inc);
                }
                double gnew[NTL], term = 0.0;
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    double g = c.gbar[t];
#pragma unroll
                    for (int q = 0; q < NCOL; ++q) g = g + inc[q][t];
                    gnew[t] = g;
                    const double e = (g - c.G[t]) * c.sinv[t];
                    const double e2 = e * e;
                    term = (t == 0) ? e2 : (term + e2);
                }
                const double sum63 = wave_sum_lane63(term);
                const double ln_u = readlane_f64(d.lnu, j);
                acc = lane63_true(metropolis(c.chi2, sum63, c.inv2theta, ln_u)) ? 1 : 0;
                n_eval += 1;
                if (LOG || acc) chi2_new = readlane_f64(sum63, 63);
                if (acc) commit(j, gnew, chi2_new, s_new, inflight);
            }
            record(inb, acc, chi2_new);
        };
        // proposals j (ring slot UA_) and j + 1 (slot UB_)
        auto pair = [&](auto UA_, auto UB_, int j) __attribute__((always_inline)) {
            constexpr int ua = decltype(UA_)::value, ub = decltype(UB_)::value;
            const bool both = (j + 1 < cnt);
            const bool fast = both && is_inb[ua] && is_inb[ub] && (((conflict >> (j + 1)) & 1ull) == 0ull) &&
                              (((stale >> j) & 3ull) == 0ull);
            if (!fast) {
                single(UA_, j, (1ull << DEPTH) - 2ull);                   // in flight behind j: j+1 .. j+DEPTH-1
                if (both) single(UB_, j + 1, (1ull << (DEPTH - 1)) - 2ull);   // behind j+1: j+2 .. j+DEPTH-1
                return;
            }
            double A_a[NCOL], A_b[NCOL];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                if constexpr (UA) { A_a[q] = c.A_uniform; A_b[q] = c.A_uniform; }
                else { A_a[q] = readlane_f64(d.A[q], j); A_b[q] = readlane_f64(d.A[q], j + 1); }
            }
            double inc_a[NCOL][NTL], inc_b[NCOL][NTL];
            increments<NCOL, NTL>(A_a, kn[ua], ko[ua], inc_a);
            increments<NCOL, NTL>(A_b, kn[ub], ko[ub], inc_b);
            double ga[NTL], gr[NTL], gb[NTL], ta = 0.0, tr = 0.0, tb = 0.0;
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                double x = c.gbar[t], y = c.gbar[t];
#pragma unroll
                for (int q = 0; q < NCOL; ++q) { x = x + inc_a[q][t]; y = y + inc_b[q][t]; }
                double z = x;
#pragma unroll
                for (int q = 0; q < NCOL; ++q) z = z + inc_b[q][t];
                ga[t] = x; gr[t] = y; gb[t] = z;
                const double ea = (x - c.G[t]) * c.sinv[t], er = (y - c.G[t]) * c.sinv[t], eb = (z - c.G[t]) * c.sinv[t];
                const double ea2 = ea * ea, er2 = er * er, eb2 = eb * eb;
                ta = (t == 0) ? ea2 : (ta + ea2);
                tr = (t == 0) ? er2 : (tr + er2);
                tb = (t == 0) ? eb2 : (tb + eb2);
            }
            double Sa = ta, Sr = tr, Sb = tb;
            wave_sum_lane63_x3(Sa, Sr, Sb);
            const double lnu_a = readlane_f64(d.lnu, j), lnu_b = readlane_f64(d.lnu, j + 1);
            // lane 63 holds the three sums; after an accepted a the chain's chi2 is Sa (the value commit() reads back)
            const bool acc_a = lane63_true(metropolis(c.chi2, Sa, c.inv2theta, lnu_a));
            const bool ok_r = lane63_true(metropolis(c.chi2, Sr, c.inv2theta, lnu_b));
            const bool ok_b = lane63_true(metropolis(Sa, Sb, c.inv2theta, lnu_b));
            const bool acc_b = acc_a ? ok_b : ok_r;
            n_eval += 2;
            double chi2_a = 0.0, chi2_b = 0.0;
            if (LOG || acc_a) chi2_a = readlane_f64(Sa, 63);
            if (acc_a) commit(j, ga, chi2_a, is_new[ua], (1ull << DEPTH) - 2ull);
            record(1, acc_a ? 1 : 0, chi2_a);
            if (acc_a) {
                if (LOG || acc_b) chi2_b = readlane_f64(Sb, 63);
                if (acc_b) commit(j + 1, gb, chi2_b, is_new[ub], (1ull << (DEPTH - 1)) - 2ull);
            } else {
                if (LOG || acc_b) chi2_b = readlane_f64(Sr, 63);
                if (acc_b) commit(j + 1, gr, chi2_b, is_new[ub], (1ull << (DEPTH - 1)) - 2ull);
            }
            record(1, acc_b ? 1 : 0, chi2_b);
        };

        [&]<int... Us>(std::integer_sequence<int, Us...>) {
            (issue(std::integral_constant<int, Us>{}, Us), ...);
        }(std::make_integer_sequence<int, DEPTH>{});

        for (int j0 = 0; j0 < cnt; j0 += DEPTH) {
            [&]<int... Ps>(std::integer_sequence<int, Ps...>) {
                (((j0 + 2 * Ps < cnt)
                      ? (pair(std::integral_constant<int, 2 * Ps>{}, std::integral_constant<int, 2 * Ps + 1>{}, j0 + 2 * Ps), 0)
                      : 0,
                  issue(std::integral_constant<int, 2 * Ps>{}, j0 + 2 * Ps + DEPTH),
                  issue(std::integral_constant<int, 2 * Ps + 1>{}, j0 + 2 * Ps + 1 + DEPTH)), ...);
            }(std::make_integer_sequence<int, DEPTH / 2>{});
        }
    }
    c.n_eval += (uint32_t)n_eval;
    c.n_cols += (uint32_t)n_eval * (uint32_t)(2 * NCOL);
    c.n_acc += (uint32_t)accepted;
    c.n_accmoves += (uint32_t)accepted * (uint32_t)NCOL;
    return accepted;
}

}  // namespace sacdev
