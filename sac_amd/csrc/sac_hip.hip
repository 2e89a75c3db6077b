// libsac_hip.so -- kernels and C ABI (include/sac_hip.h) of the MI355X SAC sampling engine.
// gfx950 only; build: see __graft_entry__.build() / Makefile (hipcc --offload-arch=gfx950
// -ffp-contract=off -munsafe-fp-atomics).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "../../include/sac_hip.h"
#include "free_sweep.h"
#include "hist_sort.h"

using namespace sacdev;

// =========================================================================================
// Kernels
// =========================================================================================

// One launch = n_sweeps sweeps (run_updates, free:825-861) of every chain, plus, by flag, the
// per-sweep sampling work (free:918-919) and the end-of-round window retune (free:883-895).
// PAR2 (amplitude updates, free:839-855): the amplitudes join the positions in LDS, two chains per block.
template <int NTL, bool LOG, bool UA, bool PAR2>
__global__ void __launch_bounds__(64 * (PAR2 ? 2 : WAVES_PER_BLOCK), PAR2 ? 2 : 4)
free_stage_kernel(const FreeParams p)
{
    constexpr int WPB = PAR2 ? 2 : WAVES_PER_BLOCK;
    __shared__ uint32_t s_wi[WPB][MAX_N_OMEGA];
    __shared__ double s_A[PAR2 ? WPB : 1][PAR2 ? MAX_N_OMEGA : 1];
    __shared__ double s_sc[WPB][12];   // windows[4], accept accumulators[6], sum chi2
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chain = (int)blockIdx.x * WPB + wave;
    if (chain >= p.n_chains) return;
    const int lane = lane_id();

    Chain<NTL> c;
    c.Kb = reinterpret_cast<const char *>(p.K);
    c.col_bytes = (uint32_t)p.ntp * 8u;
    c.wi = s_wi[wave];
    c.lane = lane;
    c.n_omega = p.n_omega;
    c.gmax = p.gmax;
    c.inv2theta = p.inv2theta;
    c.key0 = p.seed_lo;
    c.key1 = p.seed_hi;
    c.chain_id = p.chain_offset + (uint32_t)chain;
    double *A_g = p.A + (size_t)chain * (size_t)p.wi_stride;
    c.A = A_g;
    if constexpr (PAR2) {
        for (int i = lane; i < p.n_omega; i += 64) s_A[wave][i] = A_g[i];
        c.A = s_A[wave];
    }
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
        a = move_loop<1, 1, D1, NTL, LOG, UA>(c, n, win[0]);        // free:829-830
        if (lane == 0) acc[0] += (double)a / (double)n;
#ifndef EXP_NO2
        a = move_loop<2, 2, D2, NTL, LOG, UA>(c, n / 2, win[1]);    // free:832-833
        if (lane == 0) acc[1] += (double)(2 * a) / (double)(n / 2);
#endif
#ifndef EXP_NO3
        a = move_loop<3, 3, D3, NTL, LOG, UA>(c, n / 3, 0.0);       // free:835-836
        if (lane == 0) acc[2] += (double)a / (double)(n / 3);
#endif
        if constexpr (PAR2) {
            double *A = s_A[wave];
            a = aw_move_loop<(NTL == 1) ? 4 : 2, NTL, LOG>(c, A, n / 2, win[3]);   // free:851-852
            if (lane == 0) acc[3] += (double)(2 * a) / (double)(n / 2);
            renormalise(A, n, p.symm, lane);                        // free:563-567
            a = da_move_loop<(NTL == 1) ? 8 : 4, NTL, LOG>(c, A, n);               // free:854-855
            if (lane == 0) acc[4] += (double)a / (double)n;
            renormalise(A, n, p.symm, lane);                        // free:721-725
        }
#ifndef EXP_NOHIST
        if (p.flags & STAGE_SAMPLE) {
            if (p.hist_mode == SAC_HIST_REFERENCE) {
                if (p.n_grid <= (1ll << 21))
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
#endif
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
    if constexpr (PAR2) {
        for (int i = lane; i < p.n_omega; i += 64) A_g[i] = s_A[wave][i];
    }
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

// chi^2 of a given Gbar (sac_set_gbar consistency helper is host side; this kernel sums replicas)
__global__ void hist_sum_kernel(const double *hist, long long n_grid, int copies, double *dst)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_grid) return;
    double s = 0.0;
    for (int c = 0; c < copies; ++c) s = s + hist[(size_t)c * (size_t)n_grid + (size_t)i];
    dst[i] = s;
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

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(SAC_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                              \
    } while (0)

struct sac_handle {
    sac_config_t cfg{};
    int ntp = 0;
    int wi_stride = 0;
    int hist_copies = 1;
    bool hist_exclusive = false;
    int sweeps_per_launch = 25;
    double *dK = nullptr, *dG = nullptr, *dSinv = nullptr, *dA = nullptr, *dGbar = nullptr;
    double *dHist = nullptr;
    uint32_t *dWi = nullptr;
    ChainScalars *dSc = nullptr;
    MoveRec *dLog = nullptr;
    unsigned long long *dLogLen = nullptr;
    long long log_cap = 0;
    int log_chain = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kev;   // reusable per-launch event pairs
    bool problem = false, state_set = false, gbar_set = false;
    bool uniform_A = false;
    double A_uniform = 0.0;
    int last_stage = 0;            // 0 none, 1 adjust_windows, 2 sample
    int64_t last_steps = 0;
    double last_stage_ms = 0.0, last_kernel_ms = 0.0;
    int64_t last_launches = 0;
};

extern "C" const char *sac_last_error(void) { return g_err; }

static int use_device(sac_handle *h)
{
    HIP_TRY(hipSetDevice(h->cfg.device));
    return SAC_OK;
}

extern "C" int sac_create(const sac_config_t *cfg, sac_handle_t **out)
{
    if (!cfg || !out) return fail(SAC_E_INVALID, "sac_create: null argument");
    *out = nullptr;
    if (cfg->kind != SAC_KIND_FREE) return fail(SAC_E_INVALID, "sac_create: unsupported kind %d", cfg->kind);
    if (cfg->par < 1 || cfg->par > 3)
        return fail(SAC_E_INVALID, "sac_create: par=%d (1, 2 or 3)", cfg->par);
    if (cfg->n_tau < 1 || cfg->n_tau > SAC_MAX_N_TAU)
        return fail(SAC_E_INVALID, "sac_create: n_tau=%d outside 1..%d", cfg->n_tau, SAC_MAX_N_TAU);
    if (cfg->n_omega < 3 || cfg->n_omega > SAC_MAX_N_OMEGA)
        return fail(SAC_E_INVALID, "sac_create: n_omega=%d outside 3..%d", cfg->n_omega, SAC_MAX_N_OMEGA);
    if (cfg->n_chains < 1) return fail(SAC_E_INVALID, "sac_create: n_chains=%d", cfg->n_chains);
    if (cfg->n_grid < 1 || cfg->n_grid > (1ll << 30))
        return fail(SAC_E_INVALID, "sac_create: n_grid=%lld outside 1..2^30", (long long)cfg->n_grid);
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
    h->wi_stride = (cfg->n_omega + 63) / 64 * 64;
    int copies = cfg->hist_copies;
    if (copies <= 0) copies = cfg->n_chains < 16 ? cfg->n_chains : 16;
    if (copies > cfg->n_chains) copies = cfg->n_chains;
    h->hist_copies = copies;
    h->hist_exclusive = (copies == cfg->n_chains);
    if (cfg->sweeps_per_launch > 0) h->sweeps_per_launch = cfg->sweeps_per_launch;

    int rc = use_device(h);
    if (rc) { delete h; return rc; }
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
    ALLOC(h->dHist, (size_t)copies * (size_t)cfg->n_grid * sizeof(double));
    ALLOC(h->dLogLen, sizeof(unsigned long long));
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
        HIP_TRY(hipMemcpy(h->dK, K, ng * nt * sizeof(double), hipMemcpyHostToDevice));
    } else {
        HIP_TRY(hipMemcpy2D(h->dK, (size_t)h->ntp * sizeof(double), K, nt * sizeof(double),
                            nt * sizeof(double), ng, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(h->dG, G, nt * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->dSinv, sigma_inv, nt * sizeof(double), hipMemcpyHostToDevice));
    h->problem = true;
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
    const int n = h->cfg.n_omega;
    if (wi) {
        std::vector<uint32_t> rel((size_t)h->wi_stride, 0u);
        for (int i = 0; i < n; ++i) {
            if (wi[i] < h->cfg.wi_0 || wi[i] > h->cfg.wi_m)
                return fail(SAC_E_INVALID, "sac_set_state: omega_i[%d]=%lld outside [%lld,%lld]", i,
                            (long long)wi[i], (long long)h->cfg.wi_0, (long long)h->cfg.wi_m);
            rel[i] = (uint32_t)(wi[i] - h->cfg.wi_0);
        }
        std::vector<uint32_t> all((size_t)count * (size_t)h->wi_stride);
        for (int c = 0; c < count; ++c)
            std::memcpy(all.data() + (size_t)c * h->wi_stride, rel.data(), (size_t)h->wi_stride * sizeof(uint32_t));
        HIP_TRY(hipMemcpy(h->dWi + (size_t)first * h->wi_stride, all.data(), all.size() * sizeof(uint32_t),
                          hipMemcpyHostToDevice));
        h->state_set = true;
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
        HIP_TRY(hipMemcpy(h->dA + (size_t)first * h->wi_stride, all.data(), all.size() * sizeof(double),
                          hipMemcpyHostToDevice));
        if (chain == -1 || h->cfg.n_chains == 1) {
            h->uniform_A = uni;
            h->A_uniform = A[0];
        } else if (!(h->uniform_A && uni && A[0] == h->A_uniform)) {
            h->uniform_A = false;
        }
    }
    if (windows) {
        std::vector<ChainScalars> sc((size_t)count);
        HIP_TRY(hipMemcpy(sc.data(), h->dSc + first, (size_t)count * sizeof(ChainScalars), hipMemcpyDeviceToHost));
        for (auto &s : sc)
            for (int k = 0; k < 4; ++k) s.windows[k] = windows[k];
        HIP_TRY(hipMemcpy(h->dSc + first, sc.data(), (size_t)count * sizeof(ChainScalars), hipMemcpyHostToDevice));
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
    const int n = h->cfg.n_omega;
    if (wi) {
        std::vector<uint32_t> rel((size_t)n);
        HIP_TRY(hipMemcpy(rel.data(), h->dWi + (size_t)chain * h->wi_stride, (size_t)n * sizeof(uint32_t),
                          hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) wi[i] = (int64_t)rel[i] + h->cfg.wi_0;
    }
    if (A)
        HIP_TRY(hipMemcpy(A, h->dA + (size_t)chain * h->wi_stride, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (windows) {
        ChainScalars s;
        HIP_TRY(hipMemcpy(&s, h->dSc + chain, sizeof(s), hipMemcpyDeviceToHost));
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

extern "C" int sac_init_gbar(sac_handle_t *h)
{
    if (!h) return fail(SAC_E_INVALID, "sac_init_gbar: null handle");
    if (!h->problem || !h->state_set)
        return fail(SAC_E_STATE, "sac_init_gbar: upload the problem and set the state first");
    int rc;
    if ((rc = use_device(h))) return rc;
    FreeParams p;
    fill_params(h, p);
    if (h->cfg.n_tau <= 64)
        hipLaunchKernelGGL(init_gbar_kernel<1>, dim3(grid_blocks(h)), dim3(64 * WAVES_PER_BLOCK), 0, h->stream, p);
    else
        hipLaunchKernelGGL(init_gbar_kernel<2>, dim3(grid_blocks(h)), dim3(64 * WAVES_PER_BLOCK), 0, h->stream, p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->gbar_set = true;
    return SAC_OK;
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
        HIP_TRY(hipMemcpy(gbar, h->dGbar + (size_t)chain * SAC_MAX_N_TAU, (size_t)h->cfg.n_tau * sizeof(double),
                          hipMemcpyDeviceToHost));
    ChainScalars s;
    HIP_TRY(hipMemcpy(&s, h->dSc + chain, sizeof(s), hipMemcpyDeviceToHost));
    if (chi2) *chi2 = s.chi2;
    if (chi2_min) *chi2_min = s.chi2_min;
    return SAC_OK;
}

extern "C" int sac_set_gbar(sac_handle_t *h, int32_t chain, const double *gbar, double chi2, double chi2_min)
{
    if (!h || !gbar) return fail(SAC_E_INVALID, "sac_set_gbar: null argument");
    int first, count, rc;
    if ((rc = chain_range(h, chain, &first, &count, "sac_set_gbar"))) return rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<double> pad(SAC_MAX_N_TAU, 0.0);
    std::memcpy(pad.data(), gbar, (size_t)h->cfg.n_tau * sizeof(double));
    std::vector<ChainScalars> sc((size_t)count);
    HIP_TRY(hipMemcpy(sc.data(), h->dSc + first, (size_t)count * sizeof(ChainScalars), hipMemcpyDeviceToHost));
    for (int c = 0; c < count; ++c) {
        HIP_TRY(hipMemcpy(h->dGbar + (size_t)(first + c) * SAC_MAX_N_TAU, pad.data(),
                          SAC_MAX_N_TAU * sizeof(double), hipMemcpyHostToDevice));
        sc[c].chi2 = chi2;
        sc[c].chi2_min = chi2_min;
    }
    HIP_TRY(hipMemcpy(h->dSc + first, sc.data(), (size_t)count * sizeof(ChainScalars), hipMemcpyHostToDevice));
    h->gbar_set = true;
    return SAC_OK;
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
        const int wpb = par2 ? 2 : WAVES_PER_BLOCK;
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
    h->last_stage = kind;
    h->last_steps = steps;
    return SAC_OK;
}

extern "C" int sac_adjust_windows(sac_handle_t *h, int64_t steps, double theta)
{
    int rc;
    if ((rc = stage_begin(h, "sac_adjust_windows", steps, theta))) return rc;
    const int64_t per = steps / 10;                                  // free:873
    for (int round = 0; round < 10; ++round) {                       // free:869
        // free:871 zero the rates, free:873-880 `per` sweeps, free:883-895 retune.
        // per == 0 still runs the (empty) launch so that 0/0 = NaN leaves the windows alone.
        if ((rc = run_sweeps(h, per, theta, 0, STAGE_ZERO_ACC, STAGE_RETUNE, (double)per))) return rc;
    }
    return stage_end(h, 1, steps);
}

extern "C" int sac_sample(sac_handle_t *h, int64_t steps, double theta)
{
    int rc;
    if ((rc = stage_begin(h, "sac_sample", steps, theta))) return rc;
    HIP_TRY(hipMemsetAsync(h->dHist, 0, (size_t)h->hist_copies * (size_t)h->cfg.n_grid * sizeof(double),
                           h->stream));                             // free:906
    if ((rc = run_sweeps(h, steps, theta, STAGE_SAMPLE, STAGE_ZERO_ACC, 0, 1.0))) return rc;
    return stage_end(h, 2, steps);
}

extern "C" int sac_get_stats(sac_handle_t *h, sac_stats_t *out)
{
    if (!h || !out) return fail(SAC_E_INVALID, "sac_get_stats: null argument");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int nc = h->cfg.n_chains;
    std::vector<ChainScalars> sc((size_t)nc);
    HIP_TRY(hipMemcpy(sc.data(), h->dSc, (size_t)nc * sizeof(ChainScalars), hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof(*out));
    const double steps = (double)h->last_steps;
    double s_chi2 = 0.0, ar[6] = {0, 0, 0, 0, 0, 0}, win[4] = {0, 0, 0, 0};
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
        out->proposals += s.counters[0];
        out->evaluated += s.counters[1];
        out->accepted += s.counters[2];
        out->accepted_moves += s.counters[3];
        out->columns_gathered += s.counters[4];
        out->sweeps += s.sweep;
    }
    out->sampled_chi2 = s_chi2 / (double)nc;
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
    if (chain < 0 || chain >= h->cfg.n_chains)
        return fail(SAC_E_INVALID, "sac_get_chain_stats: chain %d outside 0..%d", chain, h->cfg.n_chains - 1);
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    ChainScalars s;
    HIP_TRY(hipMemcpy(&s, h->dSc + chain, sizeof(s), hipMemcpyDeviceToHost));
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

extern "C" int sac_get_histogram(sac_handle_t *h, double *out)
{
    if (!h || !out) return fail(SAC_E_INVALID, "sac_get_histogram: null argument");
    if (h->last_stage != 2) return fail(SAC_E_STATE, "sac_get_histogram: no sample stage has run");
    int rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t ng = (size_t)h->cfg.n_grid;
    std::vector<double> tmp(ng);
    HIP_TRY(hipMemcpy(out, h->dHist, ng * sizeof(double), hipMemcpyDeviceToHost));
    for (int c = 1; c < h->hist_copies; ++c) {
        HIP_TRY(hipMemcpy(tmp.data(), h->dHist + (size_t)c * ng, ng * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ng; ++i) out[i] = out[i] + tmp[i];
    }
    const double steps = (double)h->last_steps, nc = (double)h->cfg.n_chains;
    for (size_t i = 0; i < ng; ++i) out[i] = out[i] / steps / nc;   // free:924, then chain mean
    return SAC_OK;
}

extern "C" int sac_get_histogram_sum_device(sac_handle_t *h, void *dst_device, double *steps_times_chains)
{
    if (!h || !dst_device) return fail(SAC_E_INVALID, "sac_get_histogram_sum_device: null argument");
    if (h->last_stage != 2) return fail(SAC_E_STATE, "sac_get_histogram_sum_device: no sample stage has run");
    int rc;
    if ((rc = use_device(h))) return rc;
    const long long ng = h->cfg.n_grid;
    hipLaunchKernelGGL(hist_sum_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, h->stream,
                       (const double *)h->dHist, ng, h->hist_copies, (double *)dst_device);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (steps_times_chains) *steps_times_chains = (double)h->last_steps * (double)h->cfg.n_chains;
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
    HIP_TRY(hipMemcpy(&n, h->dLogLen, sizeof(n), hipMemcpyDeviceToHost));
    *n_recorded = (int64_t)n;
    int64_t m = (int64_t)n;
    if (m > h->log_cap) m = h->log_cap;
    if (m > capacity) m = capacity;
    static_assert(sizeof(MoveRec) == sizeof(sac_move_rec_t), "move record layout");
    if (out && m > 0) HIP_TRY(hipMemcpy(out, h->dLog, (size_t)m * sizeof(MoveRec), hipMemcpyDeviceToHost));
    return SAC_OK;
}
