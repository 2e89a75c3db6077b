// sac/free fixed-Theta sweep on one wavefront per Markov chain (gfx950).
//
// Reference behaviour: /root/reference/sac/free/sac_free.jl ("free:NNN")
//   single_w_move 274-324, double_w_move 328-397, triple_w_move 401-475, run_updates 825-861,
//   adjust_windows 866-899 (window retune), sample 903-927 (histogram, <chi^2>).
//
// Mapping onto the wave:
//   lane            = tau index (two per lane when 64 < N_tau <= 128); Gbar, G, sigma_inv live in
//                     registers, chi^2 is one DPP tree reduction (wave_ops.h)
//   LDS             = the chain's delta positions (relative grid index, uint32)
//   proposal decode = lane-parallel: lane j draws and decodes proposal base+j of a group of 64
//                     (Philox blocks, index, shift, bounds) with no serial dependency
//   proposal resolve= sequential over the group (the Markov chain IS sequential); the K columns
//                     of proposal j+DEPTH are in flight while proposal j is resolved.  An accepted
//                     move that touches a delta used by a later, already decoded proposal re-decodes
//                     that lane, and a prefetched column pair that no longer matches is re-fetched,
//                     so the chain is exactly the sequential one.
#pragma once
#include "wave_ops.h"

namespace sacdev {

struct ChainScalars {
    double chi2, chi2_min;
    double windows[4];
    double acc[6];              // accept_rates accumulators (free:56)
    double sum_chi2;            // sampled_chi2 accumulator (free:919)
    unsigned long long sweep;   // sweeps since creation (Philox counter word)
    unsigned long long counters[5];   // proposals, evaluated, accepted, accepted delta-moves,
                                      // K columns gathered by evaluated proposals
};

struct MoveRec { double chi2_new; int flags; int move; };

enum : int { STAGE_ZERO_ACC = 1, STAGE_SAMPLE = 2, STAGE_RETUNE = 4, STAGE_BOUNDS_AT_START = 8 };

struct FreeParams {
    const double *K;            // [n_grid][ntp], tau contiguous, zero padded
    const double *G;            // [128] zero padded
    const double *sinv;         // [128] zero padded
    uint32_t *wi;               // [n_chains][wi_stride] relative positions
    double *A;                  // [n_chains][wi_stride] amplitudes (read when !uniform_A; par=2 writes them back)
    double *gbar;               // [n_chains][128]
    ChainScalars *sc;           // [n_chains]
    double *hist;               // [hist_copies][n_grid]
    MoveRec *log;
    unsigned long long *log_len;
    long long log_cap;
    long long n_grid;
    double A_uniform;
    double inv2theta;
    double retune_div;
    int ntp, n_tau, n_omega, wi_stride;
    int gmax;                   // wi_m - wi_0: largest relative position
    int n_chains;
    uint32_t chain_offset, seed_lo, seed_hi;
    int uniform_A;
    int hist_copies, hist_exclusive, hist_mode;
    int n_sweeps, flags, par, symm;
    int log_chain;
    int prio_shift;             // WavePriority clock shift for this launch (host: prio_shift_for)
};

constexpr int WAVES_PER_BLOCK = 4;
constexpr int MAX_N_OMEGA = 2048;

// Per-wave state that lives in registers for the whole launch.
template <int NTL>
struct Chain {
    WavePriority prio;          // rotated once per group of 64 proposals (wave_ops.h)
    const char *Kb;             // K table, byte addressed
    uint32_t col_bytes;         // bytes per (padded) column
    uint32_t tab_bytes;         // bytes of one table (< 4 GiB): num_records of the buffer descriptor
    uint32_t *wi;               // LDS
    int lane, n_omega, gmax;
    // what the current move loop draws from and where a delta may land (set per loop):
    // index = k_first + rand(0..k_count-1); a new position p is allowed iff lo <= p <= hi
    int k_first, k_count, lo, hi;
    uint32_t off[NTL];          // byte offset of this lane's tau slot(s), clamped into the column
    double gbar[NTL], G[NTL], sinv[NTL];
    double chi2, chi2_min, inv2theta;
    uint32_t key0, key1, chain_id, sweep_lo, sweep_hi;
    uint32_t n_eval, n_acc, n_accmoves, n_cols;   // per launch (fits: <= 2^31 proposals per launch)
    // amplitudes
    const double *A;            // this chain's amplitude array (global), used when !UA
    double A_uniform;
    // move log
    MoveRec *log;
    long long log_cap;
    unsigned long long log_pos;
};

// Column `col` of the K table for this lane's tau slot(s).  The 64-bit column base is formed in
// scalar registers (s_mul_i32 / s_mul_hi_u32; col is wave-uniform), so each load is one
// global_load_dwordx2 with an SGPR base and a 32-bit lane offset and costs no vector ALU work.
template <int NTL>
__device__ __forceinline__ void load_col(const Chain<NTL> &c, int col, double (&out)[NTL])
{
    // buffer load = table descriptor (SGPRs) + scalar byte offset of the column (one s_mul_i32; col is wave-uniform)
    // + the lane's 32-bit offset: no address arithmetic on the vector ALU.  Tables are below 4 GiB (sac_create checks).
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(c.Kb), 0, (int)c.tab_bytes, 0x00020000);
    const int soff = col * (int)c.col_bytes;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)c.off[t], soff, 0);
        out[t] = __builtin_bit_cast(double, ((uint64_t)v.y << 32) | v.x);
    }
}

// One decoded proposal per lane.  NCOL deltas move: k = index, wold/wnew = relative positions.
template <int NCOL, bool UA>
struct Decoded {
    int k[NCOL], wold[NCOL], wnew[NCOL];
    int sd;         // signed shift (single, pair)
    int inb;        // proposal reaches the chi^2 evaluation
    int dead;       // retry blocks exhausted: never evaluated
    double lnu;     // ln of the Metropolis draw (free:309), prepared off the sequential path
    double A[UA ? 1 : NCOL];    // per-delta amplitudes (unused when all amplitudes are equal)
};

template <int NCOL, bool UA>
__device__ __forceinline__ void compute_new(Decoded<NCOL, UA> &d, int lo, int hi)
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
}

// signed shift from the window draw: delta = 1 + floor(U * window)  (free:285)
__device__ __forceinline__ int shift_from_draw(double u_mag, double u_sign, double window)
{
    const double f = __builtin_floor(u_mag * window);
    const int mag = (f < 1.0e9) ? 1 + (int)f : (1 << 30);   // >= 2^30 is outside any grid
    return (u_sign < 0.5) ? mag : -mag;
}

template <int NCOL, int NTL, bool UA>
__device__ __forceinline__ void decode(const Chain<NTL> &c, int move, uint32_t iter, double window,
                                       Decoded<NCOL, UA> &d)
{
    const uint32_t c2 = c.sweep_hi | ((uint32_t)move << 16);
    const u32x4 b0 = philox4x32_10(iter, c.sweep_lo, c2, c.chain_id, c.key0, c.key1);
    const u32x4 b1 = philox4x32_10(iter, c.sweep_lo, c2 | (1u << 24), c.chain_id, c.key0, c.key1);
    const uint32_t n = (uint32_t)c.k_count;
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
    for (int q = 0; q < NCOL; ++q) {
        d.k[q] += c.k_first;
        d.wold[q] = (int)c.wi[d.k[q]];
        if constexpr (!UA) d.A[q] = c.A[d.k[q]];
    }
    compute_new<NCOL, UA>(d, c.lo, c.hi);
}

// A_c * (K[:, new] - K[:, old]) for every moved delta (free:301-302, 367-369, 448-451)
template <int NCOL, int NTL>
__device__ __forceinline__ void increments(const double (&A)[NCOL], const double (&kn)[NCOL][NTL],
                                           const double (&ko)[NCOL][NTL], double (&inc)[NCOL][NTL])
{
#pragma unroll
    for (int q = 0; q < NCOL; ++q)
#pragma unroll
        for (int t = 0; t < NTL; ++t) inc[q][t] = A[q] * (kn[q][t] - ko[q][t]);
}

// One move loop of a sweep: n_it proposals of NCOL simultaneous delta shifts.
// Returns the number of accepted proposals.
template <int NCOL, int DEPTH, int NTL, bool LOG, bool UA>
__device__ __forceinline__ int move_loop(Chain<NTL> &c, const int MOVE, int n_it, double window)
{
    int accepted = 0, n_eval = 0;
    for (int base = 0; base < n_it; base += 64) {
        c.prio.rotate();
        const int cnt = (n_it - base < 64) ? (n_it - base) : 64;
        Decoded<NCOL, UA> d;
        decode<NCOL, NTL, UA>(c, MOVE, (uint32_t)(base + (c.lane < cnt ? c.lane : cnt - 1)), window, d);

        // prefetch ring: columns of proposals j .. j+DEPTH-1, and what was requested for each
        double kn[DEPTH][NCOL][NTL], ko[DEPTH][NCOL][NTL];
        int is_inb[DEPTH], is_old[DEPTH][NCOL], is_new[DEPTH][NCOL];
        // lanes (= proposals of this group) re-decoded after their columns were requested
        unsigned long long stale = 0ull;

        auto issue = [&](auto U, int j) {
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
                    // A proposal that is out of bounds is never evaluated (free:296), but its two loads must still be issued
                    // (the waits count outstanding loads statically): they are parked on column 0, which stays in the CU's
                    // vector cache, instead of fetching the delta's old column -- a random 512-B row -- from HBM for nothing.
                    // Matters where many proposals fall outside the bounds (the two-axis peak runs: 59 %).
                    const bool live = inb && (j < cnt);
                    load_col<NTL>(c, live ? nw : 0, kn[u][q]);
                    load_col<NTL>(c, live ? o : 0, ko[u][q]);
                }
            }
        };
        auto resolve = [&](auto U, int j) {
            constexpr int u = decltype(U)::value;
            int inb = is_inb[u];
            int s_old[NCOL], s_new[NCOL];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) { s_old[q] = is_old[u][q]; s_new[q] = is_new[u][q]; }
            // `stale` holds only proposals whose columns are in flight, so it is almost always zero
            const bool was_stale = (stale != 0ull) && (((stale >> j) & 1ull) != 0ull);
            if (was_stale) {                 // rare: a delta of this proposal moved meanwhile
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
                    const double e = (g - c.G[t]) * c.sinv[t];        // free:259-261
                    const double e2 = e * e;
                    term = (t == 0) ? e2 : (term + e2);
                }
                // the sum is complete in lane 63; the Metropolis test is taken there and read off the
                // compare mask, chi2_new itself comes back to scalar registers only when it is needed
                const double sum63 = wave_sum_lane63(term);
                const double ln_u = readlane_f64(d.lnu, j);
                acc = lane63_true(metropolis(c.chi2, sum63, c.inv2theta, ln_u)) ? 1 : 0;
                n_eval += 1;
                if (LOG || acc) chi2_new = readlane_f64(sum63, 63);
                if (acc) {                                            // free:310-318
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
                    // later proposals of this group that use a moved delta: decode again
                    bool same = false;
#pragma unroll
                    for (int a = 0; a < NCOL; ++a)
#pragma unroll
                        for (int b = 0; b < NCOL; ++b) same = same || (d.k[a] == s_k[b]);
                    // lanes above j only (scalar mask: the common case costs one vector compare)
                    const unsigned long long hits = __builtin_amdgcn_ballot_w64(same) & ((~0ull << j) << 1);
                    if (hits != 0ull) {
                        // proposals j+1 .. j+DEPTH-1 have their columns in flight already: those are refetched when
                        // they are resolved; later ones read the new values when their columns are requested
                        stale |= hits & ((((1ull << DEPTH) - 2ull)) << j);
                        if ((hits >> c.lane) & 1ull) {
#pragma unroll
                            for (int q = 0; q < NCOL; ++q) d.wold[q] = (int)c.wi[d.k[q]];
                            compute_new<NCOL, UA>(d, c.lo, c.hi);
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
                        r.move = MOVE;
                        c.log[c.log_pos] = r;
                    }
                    c.log_pos += 1;
                }
            }
        };

        // prologue
        [&]<int... Us>(std::integer_sequence<int, Us...>) {
            (issue(std::integral_constant<int, Us>{}, Us), ...);
        }(std::make_integer_sequence<int, DEPTH>{});

        for (int j0 = 0; j0 < cnt; j0 += DEPTH) {
            [&]<int... Us>(std::integer_sequence<int, Us...>) {
                // the next request is issued unconditionally (clamped into the group) so that
                // the compiler can count outstanding loads exactly (partial vmcnt waits)
                (((j0 + Us < cnt) ? (resolve(std::integral_constant<int, Us>{}, j0 + Us), 0) : 0,
                  issue(std::integral_constant<int, Us>{}, j0 + Us + DEPTH)), ...);
            }(std::make_integer_sequence<int, DEPTH>{});
        }
    }
    c.n_eval += (uint32_t)n_eval;
    c.n_cols += (uint32_t)n_eval * (uint32_t)(2 * NCOL);
    c.n_acc += (uint32_t)accepted;
    c.n_accmoves += (uint32_t)accepted * (uint32_t)NCOL;
    return accepted;
}

// The same loop for runs in which many proposals never reach the evaluation (bounds: the peak program's continuum and
// peak moves, peak:382-700).  move_loop's ring holds the next DEPTH proposals whatever they are, so with a fraction f of
// them alive only f * DEPTH column pairs are really in flight and the wave waits out the memory latency (measured on the
// two-axis peak shape: 41 % alive, 4.9 TB/s = what 6.6 rows per wave in flight give by Little's law).  Here ONLY live
// proposals enter the ring: `pend` is the set of decoded proposals that are in bounds and not yet requested, consumed lowest
// index first; the others cost nothing but their log record.  The chain is the sequential chain as before:
//   * an accepted move re-decodes the later proposals that use a moved delta (as move_loop does);
//   * those already in the ring are marked stale and refetched when resolved (as move_loop does);
//   * those not yet requested are simply requested with their new columns, dropped if they left the bounds, added if they
//     came into bounds -- unless such an arrival lies in FRONT of a proposal that is already in the ring, which would
//     break the order: then the ring is started again from the proposal after the current one (rare).
template <int NCOL, int DEPTH, int NTL, bool LOG, bool UA>
__device__ __forceinline__ int move_loop_live(Chain<NTL> &c, const int MOVE, int n_it, double window)
{
    int accepted = 0, n_eval = 0;
    for (int base = 0; base < n_it; base += 64) {
        c.prio.rotate();
        const int cnt = (n_it - base < 64) ? (n_it - base) : 64;
        Decoded<NCOL, UA> d;
        decode<NCOL, NTL, UA>(c, MOVE, (uint32_t)(base + (c.lane < cnt ? c.lane : cnt - 1)), window, d);
        const unsigned long long in_group = (cnt == 64) ? ~0ull : ((1ull << cnt) - 1ull);
        unsigned long long pend = __builtin_amdgcn_ballot_w64(d.inb != 0) & in_group;
        unsigned long long inring = 0ull, stale = 0ull;
        int last_issued = -1;
        int hold = 0;                   // after a restart: the slot of the restarting proposal stays empty for one turn
        int log_next = 0;               // LOG: first proposal of the group without a record

        double kn[DEPTH][NCOL][NTL], ko[DEPTH][NCOL][NTL];
        int is_j[DEPTH], is_old[DEPTH][NCOL], is_new[DEPTH][NCOL];

        auto log_skipped = [&](int upto) {      // proposals log_next .. upto-1 never reached the evaluation
            if constexpr (LOG) {
                if (c.log != nullptr) {
                    for (int s_ = log_next; s_ < upto; ++s_) {
                        if (c.lane == 0 && (long long)c.log_pos < c.log_cap) {
                            MoveRec r;
                            r.chi2_new = 0.0;
                            r.flags = 0;
                            r.move = MOVE;
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
            for (int q = 0; q < NCOL; ++q) {
                const int o = __builtin_amdgcn_readlane(d.wold[q], jj);
                const int nw = __builtin_amdgcn_readlane(d.wnew[q], jj);
                is_old[u][q] = o;
                is_new[u][q] = nw;
                // an empty slot still issues its loads (the waits count outstanding loads statically): parked on column 0
                load_col<NTL>(c, live ? nw : 0, kn[u][q]);
                load_col<NTL>(c, live ? o : 0, ko[u][q]);
            }
        };
        auto resolve = [&](auto U) {
            constexpr int u = decltype(U)::value;
            const int j = is_j[u];
            if (j >= 64) return;
            inring &= ~(1ull << j);
            log_skipped(j);
            int inb = 1;
            int s_old[NCOL], s_new[NCOL];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) { s_old[q] = is_old[u][q]; s_new[q] = is_new[u][q]; }
            const bool was_stale = (stale != 0ull) && (((stale >> j) & 1ull) != 0ull);
            if (was_stale) {                 // a delta of this proposal moved while its columns were in flight
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
                    const double e = (g - c.G[t]) * c.sinv[t];        // free:259-261
                    const double e2 = e * e;
                    term = (t == 0) ? e2 : (term + e2);
                }
                const double sum63 = wave_sum_lane63(term);
                const double ln_u = readlane_f64(d.lnu, j);
                acc = lane63_true(metropolis(c.chi2, sum63, c.inv2theta, ln_u)) ? 1 : 0;
                n_eval += 1;
                if (LOG || acc) chi2_new = readlane_f64(sum63, 63);
                if (acc) {                                            // free:310-318
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
                    const unsigned long long above = (~0ull << j) << 1;
                    const unsigned long long hits = __builtin_amdgcn_ballot_w64(same) & above & in_group;
                    if (hits != 0ull) {
                        if ((hits >> c.lane) & 1ull) {
#pragma unroll
                            for (int q = 0; q < NCOL; ++q) d.wold[q] = (int)c.wi[d.k[q]];
                            compute_new<NCOL, UA>(d, c.lo, c.hi);
                        }
                        stale |= hits & inring;
                        const unsigned long long now = __builtin_amdgcn_ballot_w64(d.inb != 0) & above & in_group;
                        const unsigned long long fresh = now & ~inring;
                        const unsigned long long arrivals = fresh & ~pend;
                        if (arrivals != 0ull && (int)__builtin_ctzll(arrivals) < last_issued) {
                            // a proposal came into bounds in front of one whose columns are already requested: start again
                            pend = now;
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
                        r.move = MOVE;
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
    c.n_eval += (uint32_t)n_eval;
    c.n_cols += (uint32_t)n_eval * (uint32_t)(2 * NCOL);
    c.n_acc += (uint32_t)accepted;
    c.n_accmoves += (uint32_t)accepted * (uint32_t)NCOL;
    return accepted;
}

// ------------------------------------------------------------------------------------------
// par = 2 amplitude moves (free:483-570 A_w_move, free:674-728 double_A_move).  The chain's
// amplitudes live in LDS (c.A points there) because they change during the launch.
// ------------------------------------------------------------------------------------------

struct DecodedAW {      // A_w_move: pair shift + amplitude transfer that conserves the first moment
    int k[2], wold[2], wnew[2];
    int sd, inb, dead;
    double lnu, A[2], Ap[2];
};

__device__ __forceinline__ void aw_compute(DecodedAW &d, int gmax)
{
    d.wnew[0] = d.wold[0] + d.sd;                                   // free:501-507
    d.wnew[1] = d.wold[1] - d.sd;
    bool ok = !d.dead;
#pragma unroll
    for (int q = 0; q < 2; ++q) ok = ok && (d.wnew[q] >= 0) && (d.wnew[q] <= gmax);   // free:509-517
    const int mag = d.sd < 0 ? -d.sd : d.sd;
    // free:522  dA = dw (A1 - A2) / (2 dw + w1 - w2), dw the unsigned magnitude
    const double dA = ((double)mag * (d.A[0] - d.A[1])) / (double)(2 * mag + d.wold[0] - d.wold[1]);
    d.Ap[0] = d.A[0] + dA;
    d.Ap[1] = d.A[1] - dA;
    if (d.Ap[0] < 0 || d.Ap[1] < 0) ok = false;                     // free:528-532
    d.inb = ok ? 1 : 0;
}

template <int DEPTH, int NTL, bool LOG>
__device__ __forceinline__ int aw_move_loop(Chain<NTL> &c, double *A, int n_it, double window)
{
    constexpr int MOVE = 4;
    int accepted = 0;
    for (int base = 0; base < n_it; base += 64) {
        c.prio.rotate();
        const int cnt = (n_it - base < 64) ? (n_it - base) : 64;
        DecodedAW d;
        {
            const uint32_t iter = (uint32_t)(base + (c.lane < cnt ? c.lane : cnt - 1));
            const uint32_t c2 = c.sweep_hi | ((uint32_t)MOVE << 16);
            const u32x4 b0 = philox4x32_10(iter, c.sweep_lo, c2, c.chain_id, c.key0, c.key1);
            const u32x4 b1 = philox4x32_10(iter, c.sweep_lo, c2 | (1u << 24), c.chain_id, c.key0, c.key1);
            const uint32_t n = (uint32_t)c.n_omega;
            uint32_t retry = 2;
            d.dead = 0;
            d.k[0] = index_from_word(b0.x, n);
            int k2 = index_from_word(b0.y, n);
            while (k2 == d.k[0]) {                                  // free:495-497
                if (retry > 255) { d.dead = 1; k2 = (d.k[0] + 1) % (int)n; break; }
                const u32x4 br = philox4x32_10(iter, c.sweep_lo, c2 | (retry << 24), c.chain_id, c.key0, c.key1);
                ++retry;
                k2 = index_from_word(br.x, n);
            }
            d.k[1] = k2;
            d.sd = shift_from_draw(u01(b0.z, b0.w), u01(b1.x, b1.y), window);   // free:499-507
            d.lnu = log_u01(u01(b1.z, b1.w));
#pragma unroll
            for (int q = 0; q < 2; ++q) { d.wold[q] = (int)c.wi[d.k[q]]; d.A[q] = A[d.k[q]]; }
            aw_compute(d, c.gmax);
        }
        double kn[DEPTH][2][NTL], ko[DEPTH][2][NTL];
        int is_inb[DEPTH], is_old[DEPTH][2], is_new[DEPTH][2];
        unsigned long long stale = 0ull;

        auto issue = [&](auto U, int j) {
            constexpr int u = decltype(U)::value;
            const int jj = (j < cnt) ? j : cnt - 1;
            const int inb = __builtin_amdgcn_readlane(d.inb, jj);
            is_inb[u] = inb;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int o = __builtin_amdgcn_readlane(d.wold[q], jj);
                const int nw = inb ? __builtin_amdgcn_readlane(d.wnew[q], jj) : o;
                is_old[u][q] = o;
                is_new[u][q] = nw;
                load_col<NTL>(c, nw, kn[u][q]);
                load_col<NTL>(c, o, ko[u][q]);
            }
        };
        auto resolve = [&](auto U, int j) {
            constexpr int u = decltype(U)::value;
            int inb = is_inb[u];
            int s_old[2], s_new[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) { s_old[q] = is_old[u][q]; s_new[q] = is_new[u][q]; }
            const bool was_stale = (stale != 0ull) && (((stale >> j) & 1ull) != 0ull);   // in-flight proposals only
            if (was_stale) {
                stale &= ~(1ull << j);
                inb = __builtin_amdgcn_readlane(d.inb, j);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    s_old[q] = __builtin_amdgcn_readlane(d.wold[q], j);
                    s_new[q] = __builtin_amdgcn_readlane(d.wnew[q], j);
                }
            }
            int acc = 0;
            double chi2_new = 0.0;
            if (inb) {
                double sA[2], sAp[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) { sA[q] = readlane_f64(d.A[q], j); sAp[q] = readlane_f64(d.Ap[q], j); }
                double gnew[NTL], term = 0.0;
                // Gbar_new and the chi2 terms are formed inside each branch, and the two asm comments keep the optimiser from
                // sinking the identical arithmetic into the join: a join of the prefetched and the refetched columns makes the
                // compiler wait for EVERY outstanding load (vmcnt(0)) at each proposal, i.e. drain the prefetch ring
                auto form = [&](const double (&fn)[2][NTL], const double (&fo)[2][NTL]) {
#pragma unroll
                    for (int t = 0; t < NTL; ++t) {                   // free:535-541
                        const double g = (((c.gbar[t] + (sAp[0] * fn[0][t])) + (sAp[1] * fn[1][t]))
                                          - (sA[0] * fo[0][t])) - (sA[1] * fo[1][t]);
                        gnew[t] = g;
                        const double e = (g - c.G[t]) * c.sinv[t];
                        const double e2 = e * e;
                        term = (t == 0) ? e2 : (term + e2);
                    }
                };
                if (!was_stale) {
                    form(kn[u], ko[u]);
                    asm volatile("; A_w move: prefetched columns" : "+v"(term));
                } else {
                    double fn[2][NTL], fo[2][NTL];
#pragma unroll
                    for (int q = 0; q < 2; ++q) { load_col<NTL>(c, s_new[q], fn[q]); load_col<NTL>(c, s_old[q], fo[q]); }
                    form(fn, fo);
                    asm volatile("; A_w move: refetched columns" : "+v"(term));
                }
                chi2_new = wave_sum_uniform(term);
                acc = lane63_true(metropolis(c.chi2, chi2_new, c.inv2theta, readlane_f64(d.lnu, j))) ? 1 : 0;
                c.n_eval += 1;
                c.n_cols += 4;
                if (acc) {                                            // free:548-561
#pragma unroll
                    for (int t = 0; t < NTL; ++t) c.gbar[t] = gnew[t];
                    c.chi2 = chi2_new;
                    if (chi2_new < c.chi2_min) c.chi2_min = chi2_new;
                    int s_k[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) s_k[q] = __builtin_amdgcn_readlane(d.k[q], j);
                    if (c.lane == 0) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) { c.wi[s_k[q]] = (uint32_t)s_new[q]; A[s_k[q]] = sAp[q]; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // A may live in global memory: later lanes re-read it
                    accepted += 1;
                    c.n_acc += 1;
                    c.n_accmoves += 2;
                    bool hit = false;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) hit = hit || (d.k[a] == s_k[b]);
                    hit = hit && (c.lane > j);
                    const unsigned long long hits = __builtin_amdgcn_ballot_w64(hit);
                    if (hits != 0ull) {
                        stale |= hits & ((((1ull << DEPTH) - 2ull)) << j);   // columns in flight: j+1 .. j+DEPTH-1
                        if (hit) {
#pragma unroll
                            for (int q = 0; q < 2; ++q) { d.wold[q] = (int)c.wi[d.k[q]]; d.A[q] = A[d.k[q]]; }
                            aw_compute(d, c.gmax);
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
                        r.move = MOVE;
                        c.log[c.log_pos] = r;
                    }
                    c.log_pos += 1;
                }
            }
        };
        [&]<int... Us>(std::integer_sequence<int, Us...>) {
            (issue(std::integral_constant<int, Us>{}, Us), ...);
        }(std::make_integer_sequence<int, DEPTH>{});
        for (int j0 = 0; j0 < cnt; j0 += DEPTH) {
            [&]<int... Us>(std::integer_sequence<int, Us...>) {
                (((j0 + Us < cnt) ? (resolve(std::integral_constant<int, Us>{}, j0 + Us), 0) : 0,
                  issue(std::integral_constant<int, Us>{}, j0 + Us + DEPTH)), ...);
            }(std::make_integer_sequence<int, DEPTH>{});
        }
    }
    return accepted;
}

struct DecodedDA {      // double_A_move: re-split the weight of two deltas at different frequencies
    int k[2], w[2];
    int dead;
    double r, lnu, dA[2], A[2];     // A: the two amplitudes as decoded (kept current by the re-decode on a hit)
};

__device__ __forceinline__ void da_compute(DecodedDA &d, const double *A)
{
    const double A1 = A[d.k[0]], A2 = A[d.k[1]];
    d.A[0] = A1;
    d.A[1] = A2;
    const double m0 = A1 + A2;                                      // free:688
    d.dA[0] = (d.r * m0) - A1;                                      // free:692-693
    d.dA[1] = ((1 - d.r) * m0) - A2;
}

template <int DEPTH, int NTL, bool LOG>
__device__ __forceinline__ int da_move_loop(Chain<NTL> &c, double *A, int n_it)
{
    constexpr int MOVE = 5;
    int accepted = 0;
    for (int base = 0; base < n_it; base += 64) {
        c.prio.rotate();
        const int cnt = (n_it - base < 64) ? (n_it - base) : 64;
        DecodedDA d;
        {
            const uint32_t iter = (uint32_t)(base + (c.lane < cnt ? c.lane : cnt - 1));
            const uint32_t c2 = c.sweep_hi | ((uint32_t)MOVE << 16);
            const u32x4 b0 = philox4x32_10(iter, c.sweep_lo, c2, c.chain_id, c.key0, c.key1);
            const u32x4 b1 = philox4x32_10(iter, c.sweep_lo, c2 | (1u << 24), c.chain_id, c.key0, c.key1);
            const uint32_t n = (uint32_t)c.n_omega;
            uint32_t retry = 2;
            d.dead = 0;
            d.k[0] = index_from_word(b0.x, n);
            d.w[0] = (int)c.wi[d.k[0]];
            int k2 = index_from_word(b0.y, n);
            int w2 = (int)c.wi[k2];
            while (w2 == d.w[0]) {                                  // free:684-686 (by frequency)
                if (retry > 255) { d.dead = 1; break; }
                const u32x4 br = philox4x32_10(iter, c.sweep_lo, c2 | (retry << 24), c.chain_id, c.key0, c.key1);
                ++retry;
                k2 = index_from_word(br.x, n);
                w2 = (int)c.wi[k2];
            }
            d.k[1] = k2;
            d.w[1] = w2;
            d.r = u01(b0.z, b0.w);
            d.lnu = log_u01(u01(b1.x, b1.y));
            da_compute(d, A);
        }
        // frequencies do not change in this loop: the prefetched columns never go stale
        double k1[DEPTH][NTL], k2c[DEPTH][NTL];
        auto issue = [&](auto U, int j) {
            constexpr int u = decltype(U)::value;
            const int jj = (j < cnt) ? j : cnt - 1;
            load_col<NTL>(c, __builtin_amdgcn_readlane(d.w[0], jj), k1[u]);
            load_col<NTL>(c, __builtin_amdgcn_readlane(d.w[1], jj), k2c[u]);
        };
        auto resolve = [&](auto U, int j) {
            constexpr int u = decltype(U)::value;
            const int live = !__builtin_amdgcn_readlane(d.dead, j);
            int acc = 0;
            double chi2_new = 0.0;
            if (live) {
                const double dA1 = readlane_f64(d.dA[0], j), dA2 = readlane_f64(d.dA[1], j);
                double gnew[NTL], term = 0.0;
#pragma unroll
                for (int t = 0; t < NTL; ++t) {                       // free:696-700
                    const double g = (c.gbar[t] + (dA1 * k1[u][t])) + (dA2 * k2c[u][t]);
                    gnew[t] = g;
                    const double e = (g - c.G[t]) * c.sinv[t];
                    const double e2 = e * e;
                    term = (t == 0) ? e2 : (term + e2);
                }
                chi2_new = wave_sum_uniform(term);
                acc = lane63_true(metropolis(c.chi2, chi2_new, c.inv2theta, readlane_f64(d.lnu, j))) ? 1 : 0;
                c.n_eval += 1;
                c.n_cols += 2;
                if (acc) {                                            // free:707-718
#pragma unroll
                    for (int t = 0; t < NTL; ++t) c.gbar[t] = gnew[t];
                    c.chi2 = chi2_new;
                    if (chi2_new < c.chi2_min) c.chi2_min = chi2_new;
                    const int sk1 = __builtin_amdgcn_readlane(d.k[0], j), sk2 = __builtin_amdgcn_readlane(d.k[1], j);
                    const double nA1 = readlane_f64(d.A[0], j) + dA1, nA2 = readlane_f64(d.A[1], j) + dA2;   // free:708-709 (+=)
                    if (c.lane == 0) {
                        A[sk1] = nA1;
                        A[sk2] = nA2;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // A may live in global memory: later lanes re-read it
                    accepted += 1;
                    c.n_acc += 1;
                    c.n_accmoves += 2;
                    const bool hit = (c.lane > j) && (d.k[0] == sk1 || d.k[0] == sk2 || d.k[1] == sk1 || d.k[1] == sk2);
                    if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
                        if (hit) da_compute(d, A);
                    }
                }
            }
            if constexpr (LOG) {
                if (c.log != nullptr) {
                    if (c.lane == 0 && (long long)c.log_pos < c.log_cap) {
                        MoveRec r;
                        r.chi2_new = live ? chi2_new : 0.0;
                        r.flags = (live ? 1 : 0) | (acc ? 2 : 0);
                        r.move = MOVE;
                        c.log[c.log_pos] = r;
                    }
                    c.log_pos += 1;
                }
            }
        };
        [&]<int... Us>(std::integer_sequence<int, Us...>) {
            (issue(std::integral_constant<int, Us>{}, Us), ...);
        }(std::make_integer_sequence<int, DEPTH>{});
        for (int j0 = 0; j0 < cnt; j0 += DEPTH) {
            [&]<int... Us>(std::integer_sequence<int, Us...>) {
                (((j0 + Us < cnt) ? (resolve(std::integral_constant<int, Us>{}, j0 + Us), 0) : 0,
                  issue(std::integral_constant<int, Us>{}, j0 + Us + DEPTH)), ...);
            }(std::make_integer_sequence<int, DEPTH>{});
        }
    }
    return accepted;
}

// A_array ./= sum(A_array) (x2 if symm), free:563-567, 721-725.  Sum order: lane l adds elements
// l, l+64, ... sequentially, then the canonical tree.
__device__ __forceinline__ void renormalise(double *A, int n, int symm, int lane)
{
    double part = 0.0;
    for (int i = lane; i < n; i += 64) part = part + A[i];
    double tot = wave_sum_uniform(part);
    if (symm == 1) tot = 2 * tot;
    for (int i = lane; i < n; i += 64) A[i] = A[i] / tot;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

}  // namespace sacdev
