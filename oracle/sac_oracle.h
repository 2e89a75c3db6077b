/*
 * oracle/sac_oracle.h -- CPU oracle for the SAC sampling hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * Julia loops (gabes135/sac, sac/free/sac_free.jl).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or
 * call it; nothing under sac_amd/ (the product) includes or links it.
 *
 * Parity status: the t.in parser / eigenbasis chi^2 path is pinned by the
 * reference's own cross-validation artifacts (tests/golden/cv_*, see
 * tests/test_oracle_golden.py).  The per-move arithmetic has no golden vectors
 * in the reference (it has no tests and an unseeded RNG): "parity unpinned at
 * move granularity" -- it is pinned structurally (line-by-line restatement,
 * each function cites the lines it follows) and statistically (committed
 * anneal.csv curves of real reference runs, tests/golden/anneal_*).
 *
 * Deliberate, documented differences from the Julia text (DESIGN.md "Oracle"):
 *   - chi^2 is summed in ONE defined order (64-slot pairwise tree); Julia's
 *     @simd reduction order is CPU dependent (free:251,259).
 *   - the Metropolis test rand() <= exp(x), x = (chi2-chi2_new)/(2 theta), is
 *     evaluated as x >= ln(rand()) with x = (chi2-chi2_new) * (1/(2 theta)) and an
 *     own <1ulp ln; Julia uses @fastmath exp (free:307) of an unpinned Julia
 *     version, so its decisions at the rounding edge are not reproducible anyway.
 *   - random draws come from an injected source: Philox4x32-10 fixed-slot
 *     stream (shared spec with the HIP kernels) or an explicit trace consumed
 *     in the reference's call order.
 */
#ifndef SAC_ORACLE_H
#define SAC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORA_DRAW_PHILOX = 0, ORA_DRAW_TRACE = 1 };
enum { ORA_MOVE_SINGLE = 1, ORA_MOVE_DOUBLE = 2, ORA_MOVE_TRIPLE = 3,
       ORA_MOVE_AW = 4, ORA_MOVE_DOUBLE_A = 5 };
enum { ORA_HIST_REFERENCE = 0, ORA_HIST_ACCUMULATE = 1 };

typedef struct {
    double  chi2_new;   /* chi^2 of the proposed configuration (0 if not evaluated) */
    int32_t flags;      /* bit0: evaluated (in bounds), bit1: accepted */
    int32_t move;       /* ORA_MOVE_* */
} ora_move_rec_t;

typedef struct ora_free ora_free_t;

/* Philox4x32-10 block; exported for known-answer tests. */
void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* ln(u) as used in the accept test; exported for accuracy tests. */
double ora_log_u01(double u);
/* chi^2 = sum_j ((gbar_j - g_j) * sinv_j)^2 in the canonical tree order. */
double ora_chi2(const double *gbar, const double *g, const double *sinv, int n_tau);

/* K is borrowed (caller keeps it alive): n_grid columns of n_tau doubles. */
ora_free_t *ora_free_create(int par, int symm, int n_tau, int n_omega,
                            int64_t n_grid, int64_t wi0, int64_t wim,
                            const double *K, const double *G, const double *sigma_inv);
void ora_free_destroy(ora_free_t *s);

void ora_free_set_state(ora_free_t *s, const int64_t *wi, const double *A, const double *windows);
void ora_free_get_state(const ora_free_t *s, int64_t *wi, double *A, double *windows);
/* Gbar = calc_Gbar, chi2 = calc_chi2, chi2_min = chi2   (free:1213-1216) */
void ora_free_init_gbar(ora_free_t *s);
void ora_free_set_gbar(ora_free_t *s, const double *gbar, double chi2, double chi2_min);
void ora_free_get_gbar(const ora_free_t *s, double *gbar, double *chi2, double *chi2_min);

void ora_free_use_philox(ora_free_t *s, uint64_t seed, uint32_t chain, uint64_t sweep0);
void ora_free_use_trace(ora_free_t *s, const double *trace, int64_t len);
int64_t ora_free_trace_pos(const ora_free_t *s);
uint64_t ora_free_sweep_counter(const ora_free_t *s);
void ora_free_set_log(ora_free_t *s, ora_move_rec_t *log, int64_t cap);
int64_t ora_free_log_len(const ora_free_t *s);
void ora_free_set_hist_mode(ora_free_t *s, int mode);

void ora_free_run_updates(ora_free_t *s, double theta);                 /* free:825 */
void ora_free_adjust_windows(ora_free_t *s, int64_t steps, double theta); /* free:866 */
void ora_free_sample(ora_free_t *s, int64_t steps, double theta);       /* free:903 */

void ora_free_get_accept_rates(const ora_free_t *s, double out[6]);
double ora_free_sampled_chi2(const ora_free_t *s);
void ora_free_get_spec(const ora_free_t *s, double *out);  /* n_grid doubles */
/* counters since creation: [0] proposals, [1] evaluated, [2] accepted proposals,
 * [3] accepted delta-moves (accepted proposals weighted 1/2/3 deltas) */
void ora_free_get_counters(const ora_free_t *s, uint64_t out[4]);

/* ---- peak parameterisation (sac/peak/sac_peak.jl), see sac_oracle_peak.c --------------------- */
typedef struct ora_peak ora_peak_t;
ora_peak_t *ora_peak_create(int n_tau, int n_omega, int n_p, int64_t wi0, int64_t wim, int symm,
                            int bosonic, int fix_edge, const double *Kp, const double *Kn,
                            const double *G, const double *sigma_inv);
void ora_peak_destroy(ora_peak_t *s);
void ora_peak_set_state(ora_peak_t *s, const int64_t *wi, const double *A, const double *windows);
void ora_peak_get_state(const ora_peak_t *s, int64_t *wi, double *A, double *windows);
void ora_peak_init(ora_peak_t *s);
void ora_peak_set_bounds(ora_peak_t *s, const double *b4);
void ora_peak_get_bounds(const ora_peak_t *s, double *b4);
void ora_peak_get_gbar(const ora_peak_t *s, double *gbar, double *chi2, double *chi2_min);
void ora_peak_use_philox(ora_peak_t *s, uint64_t seed, uint32_t chain, uint64_t sweep0);
void ora_peak_set_indiv_update(ora_peak_t *s, int on);
void ora_peak_set_log(ora_peak_t *s, ora_move_rec_t *log, int64_t cap);
int64_t ora_peak_log_len(const ora_peak_t *s);
uint64_t ora_peak_sweep_counter(const ora_peak_t *s);
void ora_peak_run_updates(ora_peak_t *s, double theta, int transfer);          /* peak:1006 */
void ora_peak_adjust_windows(ora_peak_t *s, int64_t steps, double theta);      /* peak:1063 */
void ora_peak_sample(ora_peak_t *s, int64_t steps, double theta, int bin);     /* peak:1100 */
void ora_peak_get_accept_rates(const ora_peak_t *s, double out[11]);
double ora_peak_sampled_chi2(const ora_peak_t *s);
void ora_peak_get_edge_res(const ora_peak_t *s, double out[6]);
void ora_peak_set_edge_res(ora_peak_t *s, const double in[6]);
void ora_peak_zero_spec(ora_peak_t *s);
void ora_peak_get_spec(const ora_peak_t *s, int which, double *out);
void ora_peak_get_counters(const ora_peak_t *s, uint64_t out[5]);

/* ---- edge parameterisation (sac/edge/sac_edge.jl), see sac_oracle_edge.c ---------------------- */
typedef struct {
    int32_t n_tau, n_e, n_omega;   /* N_tau, N_e, N_w = N_e + N_c */
    int32_t n_rl;                  /* 1: single_edge / double_edge_symm, 2: double_edge_in / _out */
    int32_t double_in, fix_edge;
    int64_t wi0, n_grid;           /* wi_0 and N_wi of the kernel tables */
    double A_c, A_r, d0, w_floor[2], w_m, dw, c[2];
} ora_edge_cfg_t;
typedef struct ora_edge ora_edge_t;
double ora_sac_log(double x);
double ora_sac_exp(double x);
/* K, dK, d2K: n_rl borrowed tables each, [n_grid][n_tau] */
ora_edge_t *ora_edge_create(const ora_edge_cfg_t *cfg, const double *const *K, const double *const *dK,
                            const double *const *d2K, const double *G, const double *sigma_inv);
void ora_edge_destroy(ora_edge_t *s);
void ora_edge_set_state(ora_edge_t *s, const double *w, const double *A, const double *dw, const int64_t *nm,
                        const double *scal);
void ora_edge_get_state(const ora_edge_t *s, double *w, double *A, double *dw, int64_t *nm, double *scal);
void ora_edge_init_gbar(ora_edge_t *s);
void ora_edge_profile(ora_edge_t *s, int rl, double n0, double eps0, double *out);
void ora_edge_get_gbar(const ora_edge_t *s, double *gbar, double *chi2, double *chi2_min);
void ora_edge_use_philox(ora_edge_t *s, uint64_t seed, uint32_t chain, uint64_t sweep0);
void ora_edge_set_log(ora_edge_t *s, ora_move_rec_t *log, int64_t cap);
int64_t ora_edge_log_len(const ora_edge_t *s);
uint64_t ora_edge_sweep_counter(const ora_edge_t *s);
int64_t ora_edge_violations(const ora_edge_t *s);
void ora_edge_get_counters(const ora_edge_t *s, uint64_t out[5]);
void ora_edge_run_updates(ora_edge_t *s, int64_t steps, double theta);              /* edge:1260 */
void ora_edge_run_bins(ora_edge_t *s, int64_t steps, int64_t bins, double theta);   /* edge:1357 */
void ora_edge_get_results(const ora_edge_t *s, double *res21, double *amp_ar6);
void ora_edge_get_accept_rates(const ora_edge_t *s, int m, int rl, double *out);
void ora_edge_get_spec(const ora_edge_t *s, int which, int rl, double *out);
void ora_edge_get_avg(const ora_edge_t *s, double *w_avg, double *A_avg);

#ifdef __cplusplus
}
#endif
#endif
