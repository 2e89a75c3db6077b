// sac/free fixed-Theta sweep with SEVERAL Markov chains per wavefront (gfx950).
//
// Same reference behaviour and the same arithmetic specification as free_sweep.h (single_w_move free:274-324,
// double_w_move 328-397, triple_w_move 401-475): a chain run here is, move for move and bit for bit, the chain the
// one-chain-per-wave kernel and the CPU oracle produce.  What changes is the mapping onto the wave:
//
//   C chains per wave, LPC = 64 / C lanes per chain, T tau slots per lane (slot = T * q + t for lane q of the chain):
//     (C, T) = (2, 2)  32 < N_tau <= 64      (2, 1) or (4, 2)  N_tau <= 32      (4, 4)  32 < N_tau <= 64
//   (N_tau > 64 stays with one chain per wave: its canonical order adds slots l and l + 64 first.)
//   One step of the resolve loop evaluates ONE proposal of EVERY chain of the wave: the element-wise work, the DPP
//   reduction tree and all wave-uniform bookkeeping (scalar branches, ring management, priority rotation) are paid once
//   per C proposals, and the tree loses log2(C) of its six levels (its first log2(T) levels are plain adds inside a
//   lane).  The one-chain kernel is bound by vector-ALU issue (profiles/r01_g_pmc_summary.txt: 46 VALU instructions
//   per proposal, 18 of them the tree); this is the lever that amortises them.
//
//   The canonical chi^2 order is unchanged: slot l pairs with l^1, l^2, ... l^32 (wave_ops.h).  With T slots per lane
//   the first levels are (t0 + t1) [+ (t2 + t3)] inside the lane, the following ones quad_perm / row_half_mirror /
//   row_mirror inside a 16-lane row, and for C = 2 one exchange of neighbouring rows (v_permlane16_swap_b32).  Every
//   lane of a chain ends up with the chain's sum, so the Metropolis test is taken lane-parallel against each lane's
//   own ln u and the decision of proposal j of chain g is bit g * LPC + j of the compare mask.
//
//   Proposals are decoded lane-parallel as before -- lane q of chain g decodes proposal base + q of its chain -- and
//   the columns of step j + DEPTH are in flight while step j resolves; a lane's column index comes from the decoding
//   lane by ds_bpermute (no VALU, off the critical path) and becomes a per-lane byte offset of one buffer load.
#pragma once
#include "free_sweep.h"
#include "hist_sort.h"

namespace sacdev {

template <int T> struct TauVec { double v[T]; };

// `col` (per lane: the column of this lane's chain) of the K table, T consecutive tau slots starting at lane_off
template <int T>
__device__ __forceinline__ TauVec<T> load_col_packed(__amdgpu_buffer_rsrc_t rs, uint32_t col, uint32_t col_bytes, uint32_t lane_off)
{
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int voff = (int)(col * col_bytes + lane_off);
    TauVec<T> out;
    if constexpr (T == 1) {
        const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
        out.v[0] = __builtin_bit_cast(double, ((uint64_t)a.y << 32) | a.x);
    } else if constexpr (T == 2) {
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
        out.v[0] = __builtin_bit_cast(double, ((uint64_t)a.y << 32) | a.x);
        out.v[1] = __builtin_bit_cast(double, ((uint64_t)a.w << 32) | a.z);
    } else {
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16, 0, 0);
        out.v[0] = __builtin_bit_cast(double, ((uint64_t)a.y << 32) | a.x);
        out.v[1] = __builtin_bit_cast(double, ((uint64_t)a.w << 32) | a.z);
        out.v[2] = __builtin_bit_cast(double, ((uint64_t)b.y << 32) | b.x);
        out.v[3] = __builtin_bit_cast(double, ((uint64_t)b.w << 32) | b.z);
    }
    return out;
}

// x + (x of the neighbouring 16-lane row): rows 0 <-> 1 and 2 <-> 3, result in every lane (both rows add the same two
// operands, fp add is commutative: identical bits)
__device__ __forceinline__ double row_pair_sum(double x)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, x);
    const uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
    // v_permlane16_swap_b32 a, b: odd rows of a <-> even rows of b.  With a = b = x: a' = [r0 r0 r2 r2], b' = [r1 r1 r3 r3]
    const auto slo = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto shi = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double a = __builtin_bit_cast(double, ((uint64_t)shi[0] << 32) | slo[0]);
    const double c = __builtin_bit_cast(double, ((uint64_t)shi[1] << 32) | slo[1]);
    return a + c;
}

// Sum over the LPC = 64 / C lanes of each chain in the canonical order, valid in every lane of the chain.
template <int C>
__device__ __forceinline__ double chain_sum(double v)
{
    static_assert(C == 2 || C == 4, "chains per wave");
    v = v + dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v = v + dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v = v + dpp_f64<0x141>(v);   // row_half_mirror
    v = v + dpp_f64<0x140>(v);   // row_mirror
    if constexpr (C == 2) v = row_pair_sum(v);
    return v;
}

// mask of the lanes of the chains whose bit is set in `bits` (bit g * LPC of `bits` = chain g)
template <int C>
__device__ __forceinline__ unsigned long long expand_chain_bits(unsigned long long bits)
{
    if constexpr (C == 2) {
        const uint32_t lo = (bits & 1ull) ? 0xFFFFFFFFu : 0u, hi = ((bits >> 32) & 1ull) ? 0xFFFFFFFFu : 0u;
        return ((unsigned long long)hi << 32) | lo;
    } else {
        unsigned long long m = bits & 0x0001000100010001ull;
        return (m << 16) - m;        // 0x0001 -> 0xFFFF in each 16-bit field
    }
}

// a where the lane's bit of `mask` is set, else b: the 64-bit scalar IS the lane mask of one v_cndmask per dword
// (the compiler's own ((mask >> lane) & 1) costs three vector instructions before the select)
__device__ __forceinline__ uint32_t select_by_mask(unsigned long long mask, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}
__device__ __forceinline__ double select_by_mask(unsigned long long mask, double a, double b)
{
    const uint64_t ab = __builtin_bit_cast(uint64_t, a), bb = __builtin_bit_cast(uint64_t, b);
    const uint32_t lo = select_by_mask(mask, (uint32_t)ab, (uint32_t)bb);
    const uint32_t hi = select_by_mask(mask, (uint32_t)(ab >> 32), (uint32_t)(bb >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

typedef __attribute__((address_space(3))) uint32_t lds_u32;     // a 32-bit LDS address

template <int C, int T>
struct PackedChain {
    WavePriority prio;
    const char *Kb;
    uint32_t col_bytes, tab_bytes;
    uint32_t *wi;               // LDS: this lane's chain's positions
    uint32_t *dummy;            // LDS: a word of this lane's chain nobody reads (target of the commits not taken)
    int lane, g, q;             // chain of the wave this lane works for, lane within the chain
    int n_omega, gmax;
    uint32_t lane_off;          // byte offset of this lane's first tau slot in a column (clamped into the column)
    double gbar[T], G[T], sinv[T];
    double chi2, chi2_min;      // per lane, equal in all lanes of a chain
    double inv2theta, A_uniform;
    uint32_t key0, key1, chain_id, sweep_lo, sweep_hi;      // chain_id, sweep: per lane (its chain's)
    uint32_t n_eval[C], n_acc[C], n_accmoves[C], n_cols[C];   // per chain of the wave, this launch (wave-uniform)
    // move log of one chain of the wave (log_g < 0: none)
    MoveRec *log;
    long long log_cap;
    unsigned long long log_pos;
    int log_g;
};

template <int NCOL>
struct PDecoded {
    int k[NCOL], wold[NCOL], wnew[NCOL], weff[NCOL];    // weff = the column requested as "new": wnew, or wold when not evaluated
    int sd, inb, dead;
    double lnu;
};

template <int NCOL>
__device__ __forceinline__ void compute_new_packed(PDecoded<NCOL> &d, int lo, int hi)
{
    if constexpr (NCOL == 1) {
        d.wnew[0] = d.wold[0] + d.sd;                               // free:287-291
    } else if constexpr (NCOL == 2) {
        d.wnew[0] = d.wold[0] + d.sd;                               // free:345-351
        d.wnew[1] = d.wold[1] - d.sd;
    } else {
        const int dw = (d.wold[1] + d.wold[2] - 2 * d.wold[0]) / 3; // free:424 (truncating div)
        d.wnew[0] = d.wold[0] + 2 * dw;
        d.wnew[1] = d.wold[1] - dw;
        d.wnew[2] = d.wold[2] - dw;
    }
    bool ok = !d.dead;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) ok = ok && (d.wnew[c] >= lo) && (d.wnew[c] <= hi);  // free:293-297
    d.inb = ok ? 1 : 0;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) d.weff[c] = ok ? d.wnew[c] : d.wold[c];
}

// the word layout of free_sweep.h's decode(), with the lane's own chain id / sweep counter / window
template <int NCOL, int C, int T>
__device__ __forceinline__ void decode_packed(const PackedChain<C, T> &c, int move, uint32_t iter, double window, PDecoded<NCOL> &d)
{
    const uint32_t c2 = c.sweep_hi | ((uint32_t)move << 16);
    const u32x4 b0 = philox4x32_10(iter, c.sweep_lo, c2, c.chain_id, c.key0, c.key1);
    const u32x4 b1 = philox4x32_10(iter, c.sweep_lo, c2 | (1u << 24), c.chain_id, c.key0, c.key1);
    const uint32_t n = (uint32_t)c.n_omega;
    uint32_t retry = 2;
    d.dead = 0;
    d.sd = 0;
    d.k[0] = index_from_word(b0.x, n);
    if constexpr (NCOL >= 2) {
        int k2 = index_from_word(b0.y, n);
        while (k2 == d.k[0]) {                                      // free:339-341
            if (retry > 255) { d.dead = 1; k2 = (d.k[0] + 1) % (int)n; break; }
            const u32x4 br = philox4x32_10(iter, c.sweep_lo, c2 | (retry << 24), c.chain_id, c.key0, c.key1);
            ++retry;
            k2 = index_from_word(br.x, n);
        }
        d.k[1] = k2;
    }
    if constexpr (NCOL == 3) {
        int k3 = index_from_word(b0.z, n);
        while (k3 == d.k[0] || k3 == d.k[1]) {                      // free:418-420
            if (retry > 255) { d.dead = 1; k3 = (d.k[0] + d.k[1] + 1) % (int)n; break; }
            const u32x4 br = philox4x32_10(iter, c.sweep_lo, c2 | (retry << 24), c.chain_id, c.key0, c.key1);
            ++retry;
            k3 = index_from_word(br.y, n);
        }
        d.k[2] = k3;
    }
    if constexpr (NCOL == 1) {
        d.sd = shift_from_draw(u01(b0.y, b0.z), u01(b0.w, b1.x), window);
        d.lnu = log_u01(u01(b1.y, b1.z));
    } else if constexpr (NCOL == 2) {
        d.sd = shift_from_draw(u01(b0.z, b0.w), u01(b1.x, b1.y), window);
        d.lnu = log_u01(u01(b1.z, b1.w));
    } else {
        d.lnu = log_u01(u01(b0.w, b1.x));
    }
#pragma unroll
    for (int q = 0; q < NCOL; ++q) d.wold[q] = (int)c.wi[d.k[q]];
    compute_new_packed<NCOL>(d, 0, c.gmax);
}

// One move loop of a sweep for the C chains of the wave: n_it proposals of NCOL simultaneous delta shifts per chain.
// accepted[g] receives the number of accepted proposals of chain g.
template <int NCOL, int DEPTH, int C, int T, bool LOG>
__device__ __forceinline__ void packed_move_loop(PackedChain<C, T> &c, const int MOVE, int n_it, double window, int (&accepted)[C])
{
    constexpr int LPC = 64 / C;
    constexpr unsigned long long ONE_PER_CHAIN = (C == 2) ? 0x0000000100000001ull : 0x0001000100010001ull;
    constexpr unsigned long long CHAIN_LANES = (C == 2) ? 0xFFFFFFFFull : 0xFFFFull;
    uint32_t evaluated[C];
#pragma unroll
    for (int g = 0; g < C; ++g) { accepted[g] = 0; evaluated[g] = 0; }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(c.Kb), 0, (int)c.tab_bytes, 0x00020000);
    const int bperm_base = (c.g * LPC) * 4;          // ds_bpermute byte address of lane 0 of this lane's chain

    for (int base = 0; base < n_it; base += LPC) {
        c.prio.rotate();
        const int cnt = (n_it - base < LPC) ? (n_it - base) : LPC;
        PDecoded<NCOL> d;
        decode_packed<NCOL, C, T>(c, MOVE, (uint32_t)(base + (c.q < cnt ? c.q : cnt - 1)), window, d);
        unsigned long long inb_bits = __builtin_amdgcn_ballot_w64(d.inb != 0);
        unsigned long long stale = 0ull;         // proposals re-decoded while their columns were in flight
        unsigned long long eval_all = 0ull, acc_all = 0ull;   // this group's evaluated / accepted proposals (one bit each)

        TauVec<T> kn[DEPTH][NCOL], ko[DEPTH][NCOL];
        // column indices of the next step to request, fetched from the decoding lanes one step ahead of the loads
        // that use them (ds_bpermute latency stays off the resolve path)
        uint32_t nx_wo[NCOL], nx_we[NCOL];

        auto start = [&](int j) {                // what step j will request, as its lanes stand now
            const int jj = (j < cnt) ? j : cnt - 1;
            const int src = bperm_base + jj * 4;
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                nx_wo[q] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, d.wold[q]);
                nx_we[q] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, d.weff[q]);
            }
        };
        auto issue = [&](auto U) {               // requests the columns of the step `start` looked at last
            constexpr int u = decltype(U)::value;
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                kn[u][q] = load_col_packed<T>(rs, nx_we[q], c.col_bytes, c.lane_off);
                ko[u][q] = load_col_packed<T>(rs, nx_wo[q], c.col_bytes, c.lane_off);
            }
        };
        auto fetch = [&](int j, TauVec<T> (&fn)[NCOL], TauVec<T> (&fo)[NCOL]) {   // slow path: this step's columns, now
            const int src = bperm_base + j * 4;
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                const uint32_t wo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, d.wold[q]);
                const uint32_t we = (uint32_t)__builtin_amdgcn_ds_bpermute(src, d.weff[q]);
                fn[q] = load_col_packed<T>(rs, we, c.col_bytes, c.lane_off);
                fo[q] = load_col_packed<T>(rs, wo, c.col_bytes, c.lane_off);
            }
        };
        auto resolve = [&](auto U, int j) {
            constexpr int u = decltype(U)::value;
            const unsigned long long pbits = ONE_PER_CHAIN << j;          // this step's proposal lanes, one per chain
            const unsigned long long evalbits = inb_bits & pbits;
            auto record = [&](unsigned long long accb, double chi2_new) {
                if constexpr (LOG) {
                    if (c.log_g >= 0) {
                        const int ll = c.log_g * LPC + j;
                        const int ev = (int)((evalbits >> ll) & 1ull), ac = (int)((accb >> ll) & 1ull);
                        if (c.lane == ll && (long long)c.log_pos < c.log_cap) {
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
            if (evalbits == 0ull) { stale &= ~pbits; record(0ull, 0.0); return; }
            // A (K[:, new] - K[:, old]) per moved delta (free:301-302, 367-369, 448-451), formed inside each branch so that
            // the prefetched columns are waited for with a partial vmcnt (a join of both column sources would make
            // the compiler drain every outstanding load)
            double dl[NCOL][T];
            if ((stale & pbits) != 0ull) {       // rare: a delta of one of this step's proposals moved meanwhile
                stale &= ~pbits;
                TauVec<T> fn[NCOL], fo[NCOL];
                fetch(j, fn, fo);
#pragma unroll
                for (int q = 0; q < NCOL; ++q)
#pragma unroll
                    for (int t = 0; t < T; ++t) dl[q][t] = c.A_uniform * (fn[q].v[t] - fo[q].v[t]);
            } else {
#pragma unroll
                for (int q = 0; q < NCOL; ++q)
#pragma unroll
                    for (int t = 0; t < T; ++t) dl[q][t] = c.A_uniform * (kn[u][q].v[t] - ko[u][q].v[t]);
            }
            double gnew[T], e2[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                double g = c.gbar[t];
#pragma unroll
                for (int q = 0; q < NCOL; ++q) g = g + dl[q][t];
                gnew[t] = g;
                const double e = (g - c.G[t]) * c.sinv[t];            // free:259-261
                e2[t] = e * e;
            }
            double term;
            if constexpr (T == 1) term = e2[0];
            else if constexpr (T == 2) term = e2[0] + e2[1];
            else term = (e2[0] + e2[1]) + (e2[2] + e2[3]);
            const double sum = chain_sum<C>(term);
            // every lane of a chain holds its chain's chi2_new; the proposal's own lane holds its ln u
            const bool ok = metropolis(c.chi2, sum, c.inv2theta, d.lnu);
            const unsigned long long accbits = __builtin_amdgcn_ballot_w64(ok) & evalbits;
            record(accbits, sum);
            eval_all |= evalbits;
            acc_all |= accbits;
            // free:310-318 for the accepting chains, without a branch: every lane keeps or replaces its chain's state
            const unsigned long long lanes = expand_chain_bits<C>(accbits >> j);
#pragma unroll
            for (int t = 0; t < T; ++t) c.gbar[t] = select_by_mask(lanes, gnew[t], c.gbar[t]);
            c.chi2 = select_by_mask(lanes, sum, c.chi2);
            c.chi2_min = __builtin_fmin(c.chi2, c.chi2_min);
            // the accepting proposals' own lanes store the new positions, every other lane writes its chain's spare word
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                const uint32_t a_pos = (uint32_t)(uintptr_t)(lds_u32 *)&c.wi[d.k[q]];
                const uint32_t a_spare = (uint32_t)(uintptr_t)(lds_u32 *)c.dummy;
                lds_u32 *dst = (lds_u32 *)(uintptr_t)select_by_mask(accbits, a_pos, a_spare);
                *dst = (uint32_t)d.wnew[q];
            }
            if (accbits != 0ull) {
                // other lanes of the chain re-read these positions below: without the fence the compiler forwards a
                // lane's own last value of wi[k] instead of loading what the committing lane stored
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // later proposals of an accepting chain that use a moved delta: decode again
                int mk[NCOL];
#pragma unroll
                for (int b = 0; b < NCOL; ++b) {
                    mk[b] = __builtin_amdgcn_readlane(d.k[b], j);
#pragma unroll
                    for (int g = 1; g < C; ++g) {
                        const int s = __builtin_amdgcn_readlane(d.k[b], g * LPC + j);
                        mk[b] = (c.g == g) ? s : mk[b];
                    }
                }
                bool same = false;
#pragma unroll
                for (int a = 0; a < NCOL; ++a)
#pragma unroll
                    for (int b = 0; b < NCOL; ++b) same = same || (d.k[a] == mk[b]);
                // lanes above j of the accepting chains only
                const unsigned long long later = (((CHAIN_LANES << j) << 1) & CHAIN_LANES) * ONE_PER_CHAIN;
                const unsigned long long hits = __builtin_amdgcn_ballot_w64(same) & lanes & later;
                if (hits != 0ull) {
                    // steps j+1 .. j+DEPTH have their columns (or column indices) in flight already: refetched when
                    // they are resolved
                    const unsigned long long inflight = ((((2ull << DEPTH) - 2ull) << j) & CHAIN_LANES) * ONE_PER_CHAIN;
                    stale |= hits & inflight;
                    if ((hits >> c.lane) & 1ull) {
#pragma unroll
                        for (int q = 0; q < NCOL; ++q) d.wold[q] = (int)c.wi[d.k[q]];
                        compute_new_packed<NCOL>(d, 0, c.gmax);
                    }
                    inb_bits = __builtin_amdgcn_ballot_w64(d.inb != 0);
                }
            }
        };

        // prologue: steps 0 .. DEPTH-1 requested, step DEPTH looked at
        // (scheduling barriers: left alone, the compiler requests the ring slots in an arbitrary order and the first
        // resolve steps then wait for every outstanding load instead of the oldest)
        [&]<int... Us>(std::integer_sequence<int, Us...>) {
            ((start(Us), issue(std::integral_constant<int, Us>{}), __builtin_amdgcn_sched_barrier(0)), ...);
        }(std::make_integer_sequence<int, DEPTH>{});
        start(DEPTH);

        for (int j0 = 0; j0 < cnt; j0 += DEPTH) {
            [&]<int... Us>(std::integer_sequence<int, Us...>) {
                // the next request is issued unconditionally (clamped into the group) so that the compiler can count
                // outstanding loads exactly (partial vmcnt waits)
                (((j0 + Us < cnt) ? (resolve(std::integral_constant<int, Us>{}, j0 + Us), 0) : 0,
                  issue(std::integral_constant<int, Us>{}), start(j0 + Us + DEPTH + 1)), ...);
            }(std::make_integer_sequence<int, DEPTH>{});
        }
#pragma unroll
        for (int g = 0; g < C; ++g) {
            evaluated[g] += (uint32_t)__builtin_popcountll(eval_all & (CHAIN_LANES << (g * LPC)));
            accepted[g] += __builtin_popcountll(acc_all & (CHAIN_LANES << (g * LPC)));
        }
    }
#pragma unroll
    for (int g = 0; g < C; ++g) {
        c.n_eval[g] += evaluated[g];
        c.n_cols[g] += evaluated[g] * (uint32_t)(2 * NCOL);
        c.n_acc[g] += (uint32_t)accepted[g];
        c.n_accmoves[g] += (uint32_t)accepted[g] * (uint32_t)NCOL;
    }
}

// LDS of one wave: C chains' positions, then per chain 12 doubles (windows[4], accept accumulators[6], sum chi2, pad)
__host__ __device__ constexpr size_t packed_lds_bytes_per_wave(int chains_per_wave, int wi_stride)
{
    return (size_t)chains_per_wave * ((size_t)wi_stride * 4u + 12u * 8u);
}

// The stage kernel of free_stage_kernel<.., UA = true, PAR2 = false> (sac_hip.hip) for C chains per wave: same flags,
// same per-chain state in global memory, same results.  Wave w of the grid runs chains C*w .. C*w+C-1 (the host
// launches it only when n_chains is a multiple of C); lanes g*LPC .. g*LPC+LPC-1 work for chain g.
template <int C, int T, bool LOG>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, (C == 2 && T == 1) ? 4 : 2)
free_stage_packed_kernel(const FreeParams p)
{
    constexpr int WPB = WAVES_PER_BLOCK;
    constexpr int LPC = 64 / C;
    extern __shared__ __align__(16) unsigned char s_packed[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chain0 = ((int)blockIdx.x * WPB + wave) * C;
    if (chain0 + C > p.n_chains) return;
    const int lane = lane_id();
    unsigned char *s_mine = s_packed + (size_t)wave * packed_lds_bytes_per_wave(C, p.wi_stride);
    uint32_t *s_wi = reinterpret_cast<uint32_t *>(s_mine);                                   // [C][wi_stride]
    double *s_sc = reinterpret_cast<double *>(s_mine + (size_t)C * (size_t)p.wi_stride * 4u);  // [C][12]

    PackedChain<C, T> c;
    c.Kb = reinterpret_cast<const char *>(p.K);
    c.col_bytes = (uint32_t)p.ntp * 8u;
    c.tab_bytes = (uint32_t)((unsigned long long)p.n_grid * (unsigned long long)p.ntp * 8ull);
    c.lane = lane;
    c.g = lane / LPC;
    c.q = lane % LPC;
    c.wi = s_wi + (size_t)c.g * (size_t)p.wi_stride;
    c.dummy = reinterpret_cast<uint32_t *>(s_sc + 12 * c.g + 11);
    c.prio.init(p.prio_shift);
    c.n_omega = p.n_omega;
    c.gmax = p.gmax;
    c.inv2theta = p.inv2theta;
    c.A_uniform = p.A_uniform;
    c.key0 = p.seed_lo;
    c.key1 = p.seed_hi;
    const int my_chain = chain0 + c.g;
    c.chain_id = p.chain_offset + (uint32_t)my_chain;
#pragma unroll
    for (int g = 0; g < C; ++g) c.n_eval[g] = c.n_acc[g] = c.n_accmoves[g] = c.n_cols[g] = 0;
    // tau slots T*q .. T*q+T-1 of this lane's chain; slots past the padded column read inside it and meet G = sinv = 0
    const int slot0 = T * c.q;
    c.lane_off = (uint32_t)((slot0 + T <= p.ntp) ? slot0 : p.ntp - T) * 8u;
    double *gbar_g = p.gbar + (size_t)my_chain * SAC_MAX_N_TAU;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int j = slot0 + t;
        const bool in = (j < SAC_MAX_N_TAU);
        c.gbar[t] = in ? gbar_g[in ? j : 0] : 0.0;
        c.G[t] = in ? p.G[in ? j : 0] : 0.0;
        c.sinv[t] = in ? p.sinv[in ? j : 0] : 0.0;
    }
    for (int g = 0; g < C; ++g) {
        const uint32_t *wi_g = p.wi + (size_t)(chain0 + g) * (size_t)p.wi_stride;
        for (int i = lane; i < p.n_omega; i += 64) s_wi[(size_t)g * p.wi_stride + i] = wi_g[i];
    }
    ChainScalars *scp = p.sc + my_chain;
    c.chi2 = scp->chi2;
    c.chi2_min = scp->chi2_min;
    double *win = s_sc + 12 * c.g, *acc = win + 4, *sum_chi2 = win + 10;      // this lane's chain's
    const bool zero = (p.flags & STAGE_ZERO_ACC) != 0;
    if (c.q < 4) win[c.q] = scp->windows[c.q];
    if (c.q < 6) acc[c.q] = zero ? 0.0 : scp->acc[c.q];
    if (c.q == 0) *sum_chi2 = zero ? 0.0 : scp->sum_chi2;
    unsigned long long sweep = scp->sweep;     // per lane: its chain's sweep counter
    c.log = nullptr;
    c.log_cap = 0;
    c.log_pos = 0;
    c.log_g = -1;
    if constexpr (LOG) {
        if (p.log_chain >= chain0 && p.log_chain < chain0 + C) {
            c.log = p.log;
            c.log_cap = p.log_cap;
            c.log_pos = *p.log_len;
            c.log_g = p.log_chain - chain0;
        }
    }

    const int n = p.n_omega;
    constexpr int D1 = (T <= 2) ? 8 : 4, D2 = (T <= 2) ? 4 : 2, D3 = (T <= 2) ? 4 : 2;
    for (int s = 0; s < p.n_sweeps; ++s) {
        c.sweep_lo = (uint32_t)sweep;
        c.sweep_hi = (uint32_t)(sweep >> 32) & 0xFFFFu;
        int a[C];
        packed_move_loop<1, D1, C, T, LOG>(c, 1, n, win[0], a);                    // free:829-830
#pragma unroll
        for (int g = 0; g < C; ++g) if (lane == g * LPC) acc[0] += (double)a[g] / (double)n;
        packed_move_loop<2, D2, C, T, LOG>(c, 2, n / 2, win[1], a);                // free:832-833
#pragma unroll
        for (int g = 0; g < C; ++g) if (lane == g * LPC) acc[1] += (double)(2 * a[g]) / (double)(n / 2);
        packed_move_loop<3, D3, C, T, LOG>(c, 3, n / 3, 0.0, a);                   // free:835-836
#pragma unroll
        for (int g = 0; g < C; ++g) if (lane == g * LPC) acc[2] += (double)a[g] / (double)(n / 3);
        if (p.flags & STAGE_SAMPLE) {
            for (int g = 0; g < C; ++g) {                                     // free:918, one chain at a time on the whole wave
                double *hist = p.hist + (size_t)((chain0 + g) % p.hist_copies) * (size_t)p.n_grid;
                const uint32_t *wg = s_wi + (size_t)g * (size_t)p.wi_stride;
                if (p.hist_mode == SAC_HIST_REFERENCE) {
                    if (p.n_grid < (1ll << 21))
                        hist_reference<uint32_t, 32, 100 + 8 * C + T>(wg, n, nullptr, p.A_uniform, 1, hist, p.hist_exclusive, lane);
                    else
                        hist_reference<uint64_t, 32, 100 + 8 * C + T>(wg, n, nullptr, p.A_uniform, 1, hist, p.hist_exclusive, lane);
                } else {
                    hist_accumulate(wg, n, nullptr, p.A_uniform, 1, hist, lane);
                }
            }
            if (c.q == 0) *sum_chi2 += c.chi2;                                // free:919
        }
        sweep += 1;
    }

    if ((p.flags & STAGE_RETUNE) && c.q == 0) {                               // free:883-895
        for (int k = 0; k < 6; ++k) acc[k] = acc[k] / p.retune_div;
        for (int k = 0; k < 4; ++k) {
            const double r = acc[k];
            if (r > 0.8) win[k] *= 2;
            else if (r < 0.2) win[k] /= 2;
            else if (r > 0.55) win[k] *= 1.2;
            else if (r < 0.45) win[k] /= 1.2;
        }
    }

    for (int g = 0; g < C; ++g) {
        uint32_t *wi_g = p.wi + (size_t)(chain0 + g) * (size_t)p.wi_stride;
        for (int i = lane; i < p.n_omega; i += 64) wi_g[i] = s_wi[(size_t)g * p.wi_stride + i];
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int j = slot0 + t;
        if (j < p.n_tau) gbar_g[j] = c.gbar[t];
    }
    if (c.q == 0) {
        scp->chi2 = c.chi2;
        scp->chi2_min = c.chi2_min;
        for (int k = 0; k < 4; ++k) scp->windows[k] = win[k];
        for (int k = 0; k < 6; ++k) scp->acc[k] = acc[k];
        scp->sum_chi2 = *sum_chi2;
        scp->sweep = sweep;
        uint32_t ne = 0, na = 0, nm = 0, ncl = 0;
#pragma unroll
        for (int g = 0; g < C; ++g) {
            const bool me = (c.g == g);
            ne = me ? c.n_eval[g] : ne; na = me ? c.n_acc[g] : na; nm = me ? c.n_accmoves[g] : nm; ncl = me ? c.n_cols[g] : ncl;
        }
        scp->counters[0] += (unsigned long long)p.n_sweeps * (unsigned long long)(n + n / 2 + n / 3);
        scp->counters[1] += ne;
        scp->counters[2] += na;
        scp->counters[3] += nm;
        scp->counters[4] += ncl;
    }
    if constexpr (LOG) {
        if (c.log_g >= 0 && lane == 0) *p.log_len = c.log_pos;
    }
}

}  // namespace sacdev
