/*
 * include/sac_hip.h -- C ABI of libsac_hip.so, the MI355X (gfx950) sampling engine for
 * Stochastic Analytic Continuation.
 *
 * The reference (gabes135/sac, pure Julia) has no FFI: this header IS the drop-in boundary,
 * cut where the reference's annealing routines call the fixed-Theta stages
 * (SURVEY.md section 8b).  Each entry point names the reference call it replaces;
 * "free:NNN" = /root/reference/sac/free/sac_free.jl line NNN.  INTEGRATION.md shows the
 * `ccall` stubs a maintainer adds to sac_free.jl.
 *
 * Conventions: every function returns 0 on success or a negative SAC_E_* code and never
 * exits or throws (replaces Julia `throw("Invalid Kernel type.")` free:195); the message of
 * the calling thread's last failure is sac_last_error().  The caller owns all host buffers; the
 * library copies in/out during the call.  A handle is bound to one GPU and is not thread safe.
 * Arrays use the reference's element types: positions Int64 (absolute grid index omega_i),
 * amplitudes/Gbar/K Float64.
 *
 * Semantic extension over the reference: n_chains independent Markov chains per handle
 * (one wavefront each).  Chain c draws from the Philox4x32-10 stream keyed
 * (seed, chain_offset + c); with n_chains = 1 the chain is, move for move, the one the CPU
 * oracle (oracle/sac_oracle.c) produces for the same seed.
 */
#ifndef SAC_HIP_H
#define SAC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAC_OK            0
#define SAC_E_INVALID    -1   /* bad argument / unsupported size */
#define SAC_E_STATE      -2   /* call order (e.g. sampling before the problem is uploaded) */
#define SAC_E_HIP        -3   /* HIP runtime error (message has the hipError string) */
#define SAC_E_NOMEM      -4

#define SAC_KIND_FREE     0   /* sac/free/sac_free.jl */
#define SAC_KIND_PEAK     1   /* sac/peak/sac_peak.jl ("peak:NNN" = line NNN of that file) */

#define SAC_HIST_REFERENCE  0 /* free:918 semantics: duplicate frequencies collapse, last delta wins */
#define SAC_HIST_ACCUMULATE 1 /* every delta adds its amplitude */

#define SAC_MAX_N_TAU   128
#define SAC_MAX_N_OMEGA 2048
#define SAC_MAX_PEAK_STATE 4224  /* peak: sampled deltas per chain, (1 or 2 axes) x (n_omega + n_peak) */

typedef struct sac_handle sac_handle_t;

typedef struct {
    int32_t  kind;          /* SAC_KIND_* */
    int32_t  par;           /* free:6-10: 1 equal amplitudes, 2 amplitude updates, 3 unequal fixed */
    int32_t  symm;          /* free:71 */
    int32_t  n_tau;         /* N_tau   (<= SAC_MAX_N_TAU) */
    int32_t  n_omega;       /* N_omega: deltas per chain (3 .. SAC_MAX_N_OMEGA) */
    int32_t  n_chains;      /* independent chains on this GPU */
    int64_t  n_grid;        /* N_omega_i: columns of the K table (free:149,153) */
    int64_t  wi_0;          /* omega_i_0 (free:140) */
    int64_t  wi_m;          /* omega_i_m (free:141) */
    uint64_t seed;          /* Philox key */
    uint32_t chain_offset;  /* global id of chain 0 (shards chains across GPUs/ranks) */
    int32_t  device;        /* HIP device ordinal */
    int32_t  hist_mode;     /* SAC_HIST_* */
    int32_t  hist_copies;   /* histogram replicas chains scatter into; 0 = auto.
                               hist_copies == n_chains gives every chain a private replica
                               (plain read-modify-write, bit-reproducible). */
    int32_t  sweeps_per_launch; /* 0 = default; sweeps a single kernel launch runs */
    /* SAC_KIND_PEAK only (0 otherwise).  n_omega is then N_omega, the continuum deltas per axis;
     * a chain holds 2 (N_omega + N_p) deltas ordered peak+, cont+, peak-, cont- (peak:166-169),
     * 11 windows (peak:63), and n_grid = wi_m + 1 columns per table (peak:248-249). */
    int32_t  n_peak;        /* N_p (peak:9) */
    int32_t  fix_edge;      /* peak:76 */
    int32_t  bosonic;       /* kernel_type == :bosonic: only the positive axis is sampled (peak:1010) */
} sac_config_t;

typedef struct {
    double   sampled_chi2;     /* <chi^2>: mean over chains of the per-chain sweep average (free:919,925) */
    double   chi2_min;         /* min over chains (free:315-317) */
    double   accept_rates[6];  /* mean over chains (free:56; slot 2 counts 2 per accepted pair, free:389) */
    double   windows[4];       /* mean over chains of update_windows (free:57) */
    uint64_t proposals;        /* since creation, all chains */
    uint64_t evaluated;        /* proposals that reached the chi^2 evaluation (in bounds) */
    uint64_t accepted;         /* accepted proposals */
    uint64_t accepted_moves;   /* accepted proposals weighted by deltas moved (1/2/3) */
    uint64_t sweeps;           /* chain-sweeps since creation */
    uint64_t columns_gathered; /* K columns read by evaluated proposals (2/4/6 per single/pair/triple) */
    double   last_stage_ms;    /* device time of the last adjust_windows / sample call */
    double   last_kernel_ms;   /* device time spent in the sweep kernel proper during it */
    int64_t  last_kernel_launches;
    double   chi2_min_mean;    /* mean over chains of each chain's OWN chi2_min: what ONE chain of the reference sees
                                  (free:315-317); the drivers' single-chain-equivalent for the Theta criteria (free:1087, 1249) */
    int64_t  n_chains_total;   /* chains behind these numbers: n_chains, or the sum over ranks after sac_comm_init */
} sac_stats_t;

typedef struct {               /* SAC_KIND_PEAK: what sac_peak.jl's CSV writers print (peak:1301-1333, 1386-1421) */
    double   sampled_chi2;     /* mean over chains */
    double   chi2_min;         /* min over chains */
    double   accept_rates[11]; /* peak:62, mean over chains */
    double   windows[11];      /* peak:63, mean over chains */
    double   edge_res[3][2];   /* peak:74,1159-1161: [0] mean edge of the last sample stage, [1] sum of
                                  those means and [2] of their squares since sac_zero_accumulators;
                                  columns: + axis, - axis; mean over chains */
    double   bounds[4];        /* wi_pp, wi_pc, wi_np, wi_nc (peak:14-17), mean over chains */
    double   weights[4];       /* sum of A over peak+, cont+, peak-, cont- (peak:1302-1311), mean over chains */
    uint64_t proposals, evaluated, accepted, accepted_moves, sweeps, columns_gathered;
    double   last_stage_ms, last_kernel_ms;
    int64_t  last_kernel_launches;
    double   chi2_min_mean;    /* mean over chains of each chain's own chi2_min (peak:1343, 1545 see ONE chain's) */
    int64_t  n_chains_total;   /* n_chains, or the sum over ranks after sac_comm_init */
} sac_peak_stats_t;

typedef struct {
    double  chi2_new;   /* chi^2 of the proposed configuration (0 when not evaluated) */
    int32_t flags;      /* bit0 evaluated, bit1 accepted */
    int32_t move;       /* 1 single, 2 pair, 3 triple, 4 A+omega, 5 pair-A */
} sac_move_rec_t;

const char *sac_last_error(void);
/* 16 hex digits: fingerprint of the sources this library was built from (Makefile).  No counterpart in the reference;
 * bench.py quotes rocprofv3 counter files (profiles/traffic.json) only for the build they were measured on. */
const char *sac_build_id(void);

/* SAC(...) constructor, sampler fields only (free:5-76, 1191-1199). */
int sac_create(const sac_config_t *cfg, sac_handle_t **out);
int sac_destroy(sac_handle_t *h);

/* Hands over what read_G! and initialize! built on the host (free:99-131, 199-216):
 * K[n_grid][n_tau] with tau contiguous (Julia's column-major K_array[n_tau, n_grid]), already in
 * the covariance eigenbasis, G = U^T G and sigma_inv (free:123,110). */
int sac_upload_problem(sac_handle_t *h, const double *K, const double *G, const double *sigma_inv);

/* omega_i_array / A_array / update_windows of one chain (chain = -1: all chains).  NULL = keep.
 * free:215, 1218, 1254.  Writing positions or amplitudes (and sac_upload_problem*, sac_upload_data,
 * sac_build_table) invalidates Gbar / chi2: the stages return SAC_E_STATE until sac_init_gbar or sac_set_gbar
 * has run again (the free kernels never rebuild Gbar from the positions, free:876). */
int sac_set_state(sac_handle_t *h, int32_t chain, const int64_t *wi, const double *A,
                  const double *windows);
int sac_get_state(sac_handle_t *h, int32_t chain, int64_t *wi, double *A, double *windows);

/* Gbar = calc_Gbar; chi2 = calc_chi2; chi2_min = chi2 for every chain (free:1213-1216). */
int sac_init_gbar(sac_handle_t *h);
int sac_get_gbar(sac_handle_t *h, int32_t chain, double *gbar, double *chi2, double *chi2_min);
int sac_set_gbar(sac_handle_t *h, int32_t chain, const double *gbar, double chi2, double chi2_min);
/* Gbar averaged over the handle's chains, n_tau doubles (the row cross_val.jl:105-108 dumps is cov * Gbar of ONE chain);
 * after a merged sample stage (sac_comm_init) the average over all ranks' chains. */
int sac_get_gbar_mean(sac_handle_t *h, double *gbar_mean);

/* adjust_windows(self, steps, theta)  free:866-899 */
int sac_adjust_windows(sac_handle_t *h, int64_t steps, double theta);
/* sample(self, steps, theta)          free:903-927 */
int sac_sample(sac_handle_t *h, int64_t steps, double theta);

/* self.sampled_chi2, chi2_min, accept_rates, update_windows after a stage (free:1064-1076). */
int sac_get_stats(sac_handle_t *h, sac_stats_t *out);
/* per-chain view: out6 = {sampled_chi2, chi2, chi2_min, sweeps, 0, 0}; accept_rates[6] */
int sac_get_chain_stats(sac_handle_t *h, int32_t chain, double *out6, double *accept_rates);

/* self.sampled_spec (free:59,918,924): mean over chains of the per-chain sweep average, n_grid doubles. */
int sac_get_histogram(sac_handle_t *h, double *out);
/* Same histogram as the un-normalised sum over this handle's chains, written to DEVICE memory
 * (e.g. a tensor handed to an RCCL all-reduce); *steps_times_chains receives the divisor.
 * dst_device must be device memory on the handle's GPU (checked: SAC_E_INVALID otherwise). */
int sac_get_histogram_sum_device(sac_handle_t *h, void *dst_device, double *steps_times_chains);

/* ---- K table built on the device ------------------------------------
 * Replaces the table loop of initialize! / init_kernal! (free:199-213, peak:248-276, edge:167-236):
 * K[i][t] = sum_j kernel(w_i, tau_j) * U[j][t], w_i = ((i_first + i) + offset) * d_omega, i = 0 .. n_grid-1
 * (elementwise kernel, then the basis change as one fp64 MFMA product).  Not bit-identical to a table the
 * host built (other exp, other summation order); sac_get_table returns exactly what the device uses. */
#define SAC_KERNEL_FINITE_T 0
#define SAC_KERNEL_ZERO_T   1
#define SAC_KERNEL_BOSONIC  2
typedef struct {
    int32_t kernel_type;    /* SAC_KERNEL_* (free:220-239) */
    int32_t mirror;         /* 1: K(w) + K(-w) (symm: free:206-208, peak:258-262, edge:223) */
    int32_t negate;         /* 1: evaluate at -w (negative axis: peak:270-272, edge's R side) */
    int32_t reserved;
    int64_t i_first;        /* grid index of row 0 (free: min(wi_0, 0); peak: 0; edge: wi_0) */
    double  offset;         /* 0, or 0.5 on peak's half-offset grid (peak:253) */
    double  d_omega, beta;
} sac_table_spec_t;
/* which = 0: K_array (free) / Kp_array (peak); 1: Kn_array (peak).  tau[n_tau]; U[n_tau][n_tau] row-major = the
 * reference's `cov` (eigenvectors in columns, free:107-108). */
int sac_build_table(sac_handle_t *h, int32_t which, const sac_table_spec_t *spec, const double *tau, const double *U);
/* No counterpart in the reference.  sac_build_table for callers that cannot allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc): same arguments, same table bit for bit, but tau and U are staged in a file-scope
 * device array of the library (132 KB per GPU, held until the process ends) instead of two hipMalloc'ed buffers; the
 * call makes no hipMalloc / hipFree.  It does use blocking copies and stream synchronisation, so it is NOT for code
 * under HIP graph capture.  Limit: n_tau <= SAC_MAX_N_TAU (SAC_E_INVALID beyond).  The array is shared by all handles
 * on a GPU; calls from several host threads of one process take turns (a per-GPU lock spans copy-in to kernel end). */
int sac_build_table_noalloc(sac_handle_t *h, int32_t which, const sac_table_spec_t *spec, const double *tau, const double *U);
/* out[n_grid][n_tau], tau contiguous: the table as the device holds it (uploaded or built). */
int sac_get_table(sac_handle_t *h, int32_t which, double *out);
/* G = U^T G and sigma_inv (free:123, 110) without a table: completes a problem whose table(s) sac_build_table made. */
int sac_upload_data(sac_handle_t *h, const double *G, const double *sigma_inv);

/* ---- SAC_KIND_PEAK ------------------------------------------------------------------------- */
/* Kp_array / Kn_array of initialize! (peak:248-276), each [wi_m + 1][n_tau], tau contiguous; Kn may
 * be NULL when only the positive axis is sampled (symm or bosonic). */
int sac_upload_problem_peak(sac_handle_t *h, const double *Kp, const double *Kn, const double *G,
                            const double *sigma_inv);
/* sample(self, steps, theta, true): a bin of the final sampling -- histograms keep accumulating and
 * the +/- weight transfer is off (peak:1100-1109). */
int sac_sample_bin(sac_handle_t *h, int64_t steps, double theta);
/* sac.indiv_update (peak:85, 1523) */
int sac_set_indiv_update(sac_handle_t *h, int32_t on);
/* edge_res .= 0; sampled_pspec .= 0; sampled_cspec .= 0 (peak:1370-1372) */
int sac_zero_accumulators(sac_handle_t *h);
int sac_get_peak_stats(sac_handle_t *h, sac_peak_stats_t *out);
/* sampled_pspec[:, pn] / sampled_cspec[:, pn] (peak:65-66): which = 0 peak+, 1 cont+, 2 peak-,
 * 3 cont-.  Mean over chains; divided by the steps of the last sample stage (peak:1152-1153), or,
 * after sac_sample_bin calls, the raw accumulated sums (the driver divides, peak:1380-1381). */
int sac_get_histogram_peak(sac_handle_t *h, int32_t which, double *out);
/* ωi_pp, ωi_pc, ωi_np, ωi_nc of one chain (peak:14-17) */
int sac_get_bounds(sac_handle_t *h, int32_t chain, double *out4);

/* Population control for n_chains > 1 (no counterpart in the reference, which runs one chain):
 * chain dst takes over the configuration of chain src (positions, amplitudes, Gbar, chi2, chi2_min,
 * windows, rate accumulators, bounds) and continues on its OWN random stream and sweep counter. */
int sac_copy_chain(sac_handle_t *h, int32_t src, int32_t dst);
/* chi2 and chi2_min of every chain of the handle (n_chains doubles each; either may be NULL). */
int sac_get_chain_chi2(sac_handle_t *h, double *chi2, double *chi2_min);

/* Scheduling knob (no counterpart in the reference, which runs one chain on one core): how many chains of a
 * SAC_KIND_FREE handle share one wavefront -- 1, 2 or 4; 0 = the library chooses.  Every chain's trajectory and
 * every result is the same, bit for bit, for every value; only the mapping onto the wave changes
 * (free_sweep_packed.h: the chi^2 tree and the wave-uniform bookkeeping are paid once per step for all chains of the
 * wave).  Shapes the packed kernel does not serve (par = 2 or 3, n_chains not a multiple, N_tau above
 * 4 slots per lane of the chain) fall back to fewer chains per wave; sac_get_chains_per_wave reports what the last
 * sweep launch used. */
int sac_set_chains_per_wave(sac_handle_t *h, int32_t chains_per_wave);
int sac_get_chains_per_wave(sac_handle_t *h, int32_t *last_used);

/* Scheduling knob for runs with few chains (no counterpart in the reference; its annealing run IS one chain, free:1027-1262,
 * and the latency of that chain is the run's wall-clock): how many consecutive proposals of ONE chain a wavefront evaluates
 * per step -- 1, 2 or 4; 0 = the library chooses (2 when the handle has at most 1024 chains, i.e. at most one wave per SIMD,
 * else 1).  The chain is the sequential one for every value, bit for bit (free_sweep_spec.h: proposals behind an accepted
 * one are evaluated again against the new state).  par = 1 (equal fixed amplitudes) and N_tau <= 64 only; other shapes use
 * one proposal per step.  A handle that packs several chains per wave (sac_set_chains_per_wave) does not speculate.
 * sac_get_proposals_per_step reports what the last sweep launch used. */
int sac_set_proposals_per_step(sac_handle_t *h, int32_t proposals_per_step);
int sac_get_proposals_per_step(sac_handle_t *h, int32_t *last_used);

/* Cross-validation scoring behind the boundary: what calc_cv.jl:29-58 computes from a GSAC.csv row
 * (cross_val.jl:105-108), taken directly from the Gbar the chains hold -- no file round trip:
 *   t = U_this Gbar (cv:106), y = U_other^T t (calc_cv.jl:37), chi2_v = sum_j ((y_j - G_other_j) sigma_inv_other_j)^2.
 * U_this / U_other: the `cov` of this handle's t.in and of the OTHER half's, [n_tau][n_tau] row-major; G_other already in
 * its eigenbasis (read_G!, free:123).  gbar_mean: NULL = the mean over this handle's chains, else n_tau doubles (e.g. the
 * mean over all ranks).  chi2_mean: chi2_v of that mean Gbar (the quantity calc_cv.jl scores, not yet divided by N_tau);
 * chi2_chain[n_chains] (may be NULL): chi2_v of every chain's own Gbar. */
int sac_validation_chi2(sac_handle_t *h, const double *U_this, const double *U_other, const double *G_other,
                        const double *sigma_inv_other, const double *gbar_mean, double *chi2_mean, double *chi2_chain);

/* Parity / debug entry (no counterpart in the reference): evaluates an explicit proposal stream
 * on one chain of the free parameterisation (par = 1 or 3) -- what the reference's move loops do once the
 * indices, the new positions (in bounds) and the accept uniform are drawn (free:296-320, 354-380, 437-470).
 * chi2_new[n] / accepted[n] may be NULL.  Accepted proposals are committed to the chain. */
typedef struct {
    int32_t move;        /* number of deltas moved: 1, 2 or 3 */
    int32_t k[3];        /* 0-based delta indices (distinct) */
    int64_t w_new[3];    /* new grid indices, same coordinates as sac_set_state */
    double  u;           /* the accept draw rand() */
} sac_proposal_t;
int sac_replay(sac_handle_t *h, int32_t chain, double theta, const sac_proposal_t *props, int64_t n,
               double *chi2_new, int32_t *accepted);

/* The same for SAC_KIND_PEAK (peak:382-996).  transfer = 0: a frequency move of n = 1..3 deltas, amplitudes stay
 * (Gbar + A (K_new - K_old) per delta, peak:415-423, 486-494, 573-585, 666-673); transfer = 1: a weight transfer, n = 2 or
 * 4, pairs (k[0], k[1]) [and (k[2], k[3])] = (+ axis delta, - axis delta) with their new amplitudes (peak:758-764, 957-967).
 * k = 0-based index into the chain's state array (peak+, cont+, peak-, cont-): it selects the axis and the table Kp / Kn.
 * Starts from the handle's stored Gbar / chi2 (sac_init_gbar rebuilds them from scratch as a sweep does, peak:1072). */
typedef struct {
    int32_t n;
    int32_t transfer;
    int32_t k[4];
    int64_t w_new[4];
    double  A_new[4];    /* transfer = 1 only */
    double  u;           /* the accept draw rand() (accept when u <= P, peak:428) */
} sac_peak_proposal_t;
int sac_replay_peak(sac_handle_t *h, int32_t chain, double theta, const sac_peak_proposal_t *props, int64_t n,
                    double *chi2_new, int32_t *accepted);

/* ---- multi-GPU merge behind the boundary (SURVEY section 8e, 8b-5 "already reduced over chains and GPUs") -------------
 * No counterpart in the reference, whose only fan-out is one Julia process per parameter point / replica with files as
 * the hand-off (sac/cross_validation/run_cv.sh:5-9, sac/qsub_files/edge/submit.sh:25-51).  One process per GPU holds a
 * block of the chains of one problem (sac_config_t.chain_offset = the global id of its first chain).  Once a
 * communicator is attached, EVERY stage entry point (sac_init_gbar, sac_adjust_windows, sac_sample, sac_sample_bin;
 * sac_edge_init_gbar, sac_edge_run_bins) ends with the merge of the ranks' partial sums -- exactly ONE
 * ncclAllReduce(sum) over the fused device buffer [histogram(s) | per-delta averages | scalar sums | chain count] and
 * ONE ncclAllReduce(min) for chi2_min, in place on device memory on the handle's stream -- and the reduced getters
 * (sac_get_stats, sac_get_peak_stats, sac_get_histogram, sac_get_histogram_peak; sac_edge_get_results / _accept_rates /
 * _histogram / _avg with chain = -1) return the values over ALL ranks, identical on every rank, so every rank takes the
 * same Theta-schedule decisions (free:1087, 1128, 1249-1252; peak:1343, 1545; edge:1680-1688, 1718).  The stage calls
 * are collective: every rank makes the same sequence of them.  Per-chain getters stay local.  Writing state from the
 * host (sac_set_state, sac_set_gbar, sac_copy_chain, ...) drops the merged view until the next stage call; the reduced
 * getters then answer for the local chains (n_chains_total tells which).
 * RCCL (librccl.so.1; SAC_RCCL_LIB overrides the name) is loaded when the first of these is called. */
#define SAC_COMM_ID_BYTES 128
/* ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to the other ranks (file, socket, MPI, torch broadcast) */
int sac_comm_unique_id(void *id128);
/* ncclCommInitRank on the handle's GPU (collective over all ranks) + the merge buffers; released by sac_destroy */
int sac_comm_init(sac_handle_t *h, const void *id128, int32_t rank, int32_t world);
/* rank, world, stages merged and all-reduce calls issued so far (2 per merged stage); any pointer may be NULL */
int sac_comm_info(sac_handle_t *h, int32_t *rank, int32_t *world, int64_t *merges, int64_t *allreduce_calls);

/* Debug/parity: record every proposal of one chain (up to capacity) during the following stages. */
int sac_set_move_log(sac_handle_t *h, int32_t chain, int64_t capacity);
int sac_get_move_log(sac_handle_t *h, sac_move_rec_t *out, int64_t capacity, int64_t *n_recorded);

/* ===========================================================================================
 * edge parameterisation (sac/edge/sac_edge.jl, "edge:NNN"): its own handle type.  The boundary is
 * run_bins(self, steps, bins, theta) (edge:1357), the one stage function the annealing routines call
 * (edge:1660, 1741, 1814, 2052), plus calc_Gbar!(self, 1) / calc_chi2 at (re)initialisation
 * (edge:559-561, 1728-1730).  State arrays have the reference's shape [N_w+1, 2] (column-major:
 * rightward side, then leftward side; row 1 is the reference's unused leading slot).
 * =========================================================================================== */
#define SAC_MAX_N_EDGE 256      /* N_e; N_w = N_e + N_c <= 2 * SAC_MAX_N_EDGE */

typedef struct sac_edge sac_edge_t;

typedef struct {
    int32_t  n_tau, n_e, n_omega;   /* N_tau, N_e, N_w (edge:7-8, 43) */
    int32_t  n_rl;                  /* 1: single_edge, double_edge_symm; 2: double_edge_in / _out (edge:111) */
    int32_t  double_in;             /* mode == :double_edge_in (edge:800, 907) */
    int32_t  fix_edge;              /* edge:115 */
    int32_t  n_chains, device;
    int64_t  wi_0, n_grid;          /* wi_0 and N_wi of the kernel tables (edge:169-172) */
    double   A_c, A_r, d0;          /* edge:9-11 */
    double   w_floor[2], w_m, d_omega;   /* edge:13-14, 20 */
    double   c[2];                  /* edge:23 */
    uint64_t seed;
    uint32_t chain_offset;
    int32_t  sweeps_per_launch;     /* 0 = default */
    int32_t  hist_copies;           /* 0 = 1; n_chains = one private histogram per chain (bit-reproducible) */
    int32_t  reserved;
} sac_edge_config_t;

typedef struct {
    double   chi2_res[3];           /* after bin_averages (edge:1436-1449): [1] mean, [2] sigma of the bin means */
    double   edge_res[3][2], n0_res[3][2], wn0_res[3][2];
    double   amp_accept_rates[3][2];/* amplitude_accept_rates (edge:69) */
    double   n0[2], eps0[2], dn0[2], deps0[2];
    double   chi2, chi2_min;
    uint64_t sweeps, proposals, evaluated, accepted, lookups, violations;
    double   last_stage_ms;
    double   chi2_min_mean;         /* chain = -1: mean over chains of each chain's own chi2_min (edge:1682, 1718 see ONE
                                       chain's); one chain: its chi2_min */
    int64_t  n_chains_total;        /* chains behind a chain = -1 result: n_chains, or the sum over ranks after
                                       sac_edge_comm_init */
    int64_t  chains_below_target;   /* chain = -1: chains whose stage average <chi2> has fallen below their own target in some
                                       stage since sac_edge_set_chain_targets (all ranks' after a merged stage); one chain: 0 / 1 */
} sac_edge_results_t;

int sac_edge_create(const sac_edge_config_t *cfg, sac_edge_t **out);
int sac_edge_destroy(sac_edge_t *h);
/* K_array, dK_array, d2K_array of init_kernal! (edge:167-242): per side a [N_tau, N_wi] column-major slice */
int sac_edge_upload_problem(sac_edge_t *h, const double *const *K, const double *const *dK,
                            const double *const *d2K, const double *G, const double *sigma_inv);
/* Row f1 for edge: K of side `side` (0 = L, 1 = R) built on the device as sac_build_table does (spec->i_first = wi_0,
 * spec->negate = 1 for the R side, spec->mirror = 1 for double_edge_symm), then dK and d2K by the reference's
 * stencils (edge:230-235, incl. the row the reference addresses as wi_m + 1).  Call once per sampled side, then
 * sac_edge_upload_data. */
int sac_edge_build_tables(sac_edge_t *h, int32_t side, const sac_table_spec_t *spec, const double *tau, const double *U);
int sac_edge_upload_data(sac_edge_t *h, const double *G, const double *sigma_inv);
/* K, dK, d2K of one side as the device holds them, each [n_grid][n_tau] (any may be NULL) */
int sac_edge_get_tables(sac_edge_t *h, int32_t side, double *K, double *dK, double *d2K);
/* omega_array, A_array, D_omega_array, n_multi (edge:51-53, 72-73); scal8 = n0[2], eps_0[2], Dn0[2], Deps_0[2]
 * (edge:24-27).  chain = -1: all chains.  Any pointer may be NULL (left unchanged).  Also what
 * anneal_and_sample's save/restore needs (edge:1671-1675, 1724-1726). */
int sac_edge_set_state(sac_edge_t *h, int32_t chain, const double *w, const double *A, const double *dw,
                       const int64_t *n_multi, const double *scal8);
int sac_edge_get_state(sac_edge_t *h, int32_t chain, double *w, double *A, double *dw, int64_t *n_multi,
                       double *scal8);
/* calc_Gbar!(self, 1); chi2 = calc_chi2(self); chi2_min = chi2 */
int sac_edge_init_gbar(sac_edge_t *h);
/* run_bins(self, steps, bins, theta)  edge:1357-1379 */
int sac_edge_run_bins(sac_edge_t *h, int64_t steps, int64_t bins, double theta);
/* chain = -1: mean over chains (chi2_min: minimum; counters: sums) */
int sac_edge_get_results(sac_edge_t *h, int32_t chain, sac_edge_results_t *out);
/* anneal(self, chi2_target) with many chains (edge:1646-1693, 1718).  The reference stops its second anneal the FIRST time
 * its one chain's stage average <chi2> falls below chi2_target = chi2_min + a sqrt(2 chi2_min) -- a first passage of a
 * noisy single-chain curve.  The many-chain equivalent is the median of the chains' own first passages: every chain
 * gets its own target (from its own chi2_min), the library marks after every run_bins stage the chains whose stage
 * average lies below their target (sticky), and sac_edge_get_results(-1).chains_below_target counts them over all chains
 * (and ranks): the stage at which the count reaches half the chains is the median first passage.
 *   sac_edge_set_targets_from_chi2_min: target_c = chi2_min_c + a sqrt(2 chi2_min_c) from the chains' current chi2_min
 *   sac_edge_set_chain_targets: explicit targets[n_chains]; NULL switches the bookkeeping off
 *   sac_edge_get_first_passages: per chain the 1-based index of the stage (counted from the set_* call) of its first
 *   passage, 0 = not yet.  Setting targets clears the marks. */
int sac_edge_set_targets_from_chi2_min(sac_edge_t *h, double a_criterion);
int sac_edge_set_chain_targets(sac_edge_t *h, const double *targets);
int sac_edge_get_first_passages(sac_edge_t *h, int32_t *stage);
/* accept_rates[m, :, side] after the last adjust_D (edge:1386-1387): N_w+1 doubles */
int sac_edge_get_accept_rates(sac_edge_t *h, int32_t chain, int32_t m, int32_t side, double *out);
/* sampled_spec[:, which, side] (edge:75, 1322-1331): accumulated sums, N_wi doubles, chain mean */
int sac_edge_get_histogram(sac_edge_t *h, int32_t which, int32_t side, double *out);
/* omega_avg, A_avg (edge:83-84, 1314-1315): [N_w+1, 2] accumulated sums */
int sac_edge_get_avg(sac_edge_t *h, int32_t chain, double *w_avg, double *A_avg);
int sac_edge_get_gbar(sac_edge_t *h, int32_t chain, double *gbar, double *chi2, double *chi2_min);
/* What write_res prints about the moves (edge:1454-1460, 1484-1506), averaged over the handle's chains (all ranks' after a
 * merged stage): D_omega_array and n_multi as [N_w+1, 2] means, cont_extent = {lowest, highest continuum frequency of
 * side 1} (0, 0 without continuum).  Any pointer may be NULL. */
int sac_edge_get_move_means(sac_edge_t *h, double *dw_mean, double *n_multi_mean, double *cont_extent);
/* Gbar averaged over the handle's chains (cv:494-497), or over all ranks' chains after a merged stage */
int sac_edge_get_gbar_mean(sac_edge_t *h, double *gbar_mean);
/* Parity / debug entry (no counterpart in the reference): an explicit stream of position proposals on one
 * chain -- new frequencies w_new[w_offset .. w_offset+n-1] for the n consecutive deltas k .. k+n-1 (Julia's 1-based index,
 * 2 .. N_w+1) of one side, and the accept draw.  n = 1: what single_edge_move / single_cont_move evaluate once index, shift
 * and order check are done (edge:856-880, 1176-1186); n > 1: multi_edge_move (edge:938-964).  Accepted when u < P (edge:864)
 * and committed (omega, K_curr, Gbar, chi2).  The ordering constraints (check_order) are the caller's business. */
typedef struct {
    int32_t side;       /* column of the state arrays */
    int32_t k;          /* first moved delta */
    int32_t n;          /* deltas moved, <= 64 */
    int32_t reserved;
    int64_t w_offset;   /* first new frequency in w_new[] */
    double  u;          /* the accept draw rand() */
} sac_edge_proposal_t;
int sac_edge_replay(sac_edge_t *h, int32_t chain, double theta, const sac_edge_proposal_t *props, int64_t n,
                    const double *w_new, int64_t n_w, double *chi2_new, int32_t *accepted);
/* sac_validation_chi2 for an edge handle (cv:494-497 + calc_cv.jl:29-58) */
int sac_edge_validation_chi2(sac_edge_t *h, const double *U_this, const double *U_other, const double *G_other,
                             const double *sigma_inv_other, const double *gbar_mean, double *chi2_mean, double *chi2_chain);
/* the multi-GPU merge for an edge handle: see sac_comm_init */
int sac_edge_comm_init(sac_edge_t *h, const void *id128, int32_t rank, int32_t world);
int sac_edge_comm_info(sac_edge_t *h, int32_t *rank, int32_t *world, int64_t *merges, int64_t *allreduce_calls);
int sac_edge_set_move_log(sac_edge_t *h, int32_t chain, int64_t capacity);
int sac_edge_get_move_log(sac_edge_t *h, sac_move_rec_t *out, int64_t capacity, int64_t *n_recorded);

#ifdef __cplusplus
}
#endif
#endif
