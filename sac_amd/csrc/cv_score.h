// Cross-validation scoring on the device: calc_cv.jl:29-58 applied to the Gbar the chains hold,
// without the GSAC.csv round trip of cross_val.jl:105-108.
//   t = U_this Gbar            (cv:106: the row GSAC.csv would hold, back in the tau basis)
//   y = U_other^T t            (calc_cv.jl:37: into the other half's covariance eigenbasis)
//   chi2_v = sum_j ((y_j - G_other_j) sigma_inv_other_j)^2      (calc_cv.jl:45-51), canonical tree order
// One wave per scored vector: wave 0 scores the mean over the chains (or a caller-provided mean), wave 1 + c chain c.
#pragma once
#include <cstring>
#include <vector>
#include "wave_ops.h"
#include "devbuf.h"

namespace sacdev {

struct CvScoreArgs {
    const double *gbar;         // [n_chains][gbar_stride]
    const double *mean_in;      // [128] or nullptr
    const double *Ut, *Uo;      // [n_tau][n_tau] row-major
    const double *Gv, *sinv;    // [128] zero padded
    double *out;                // [1 + n_chains]
    int n_chains, n_tau, gbar_stride;
};

template <int TAG>      // a template only so that both translation units of the library may hold it
__global__ void __launch_bounds__(64) cv_score_kernel(const CvScoreArgs a)
{
    __shared__ double s_g[128], s_t[128];
    const int lane = lane_id();
    const int which = (int)blockIdx.x;          // 0: mean, 1 + c: chain c
    for (int j = lane; j < 128; j += 64) {
        double g = 0.0;
        if (j < a.n_tau) {
            if (which > 0) g = a.gbar[(size_t)(which - 1) * a.gbar_stride + j];
            else if (a.mean_in != nullptr) g = a.mean_in[j];
            else {
                for (int c = 0; c < a.n_chains; ++c) g = g + a.gbar[(size_t)c * a.gbar_stride + j];
                g = g / (double)a.n_chains;
            }
        }
        s_g[j] = g;
    }
    __syncthreads();
    for (int j = lane; j < 128; j += 64) {      // t = U_this * Gbar, sequentially over k
        double t = 0.0;
        if (j < a.n_tau)
            for (int k = 0; k < a.n_tau; ++k) t = t + a.Ut[(size_t)j * a.n_tau + k] * s_g[k];
        s_t[j] = t;
    }
    __syncthreads();
    double term = 0.0;
    for (int j = lane; j < 128; j += 64) {      // y = U_other^T * t
        double e2 = 0.0;
        if (j < a.n_tau) {
            double y = 0.0;
            for (int k = 0; k < a.n_tau; ++k) y = y + a.Uo[(size_t)k * a.n_tau + j] * s_t[k];
            const double e = (y - a.Gv[j]) * a.sinv[j];
            e2 = e * e;
        }
        term = (j < 64) ? e2 : (term + e2);
    }
    const double chi2 = wave_sum_uniform(term);
    if (lane == 0) a.out[which] = chi2;
}

}  // namespace sacdev

// Host side, shared by both handle types: uploads the small operands, launches, reads back.  Returns a hipError_t.
static inline hipError_t cv_score_run(hipStream_t stream, const double *d_gbar, int gbar_stride, int n_chains, int n_tau,
                                      const double *U_this, const double *U_other, const double *G_other,
                                      const double *sinv_other, const double *gbar_mean, double *chi2_mean, double *chi2_chain)
{
    const size_t nt = (size_t)n_tau, mat = nt * nt;
    const size_t n_d = 2 * mat + 3 * 128 + 1 + (size_t)n_chains;
    double *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, n_d * sizeof(double));
    if (e != hipSuccess) return e;
    std::vector<double> host(2 * mat + 3 * 128, 0.0);
    std::memcpy(host.data(), U_this, mat * sizeof(double));
    std::memcpy(host.data() + mat, U_other, mat * sizeof(double));
    std::memcpy(host.data() + 2 * mat, G_other, nt * sizeof(double));
    std::memcpy(host.data() + 2 * mat + 128, sinv_other, nt * sizeof(double));
    if (gbar_mean) std::memcpy(host.data() + 2 * mat + 256, gbar_mean, nt * sizeof(double));
    e = copy_sync(stream, d, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        sacdev::CvScoreArgs a;
        a.gbar = d_gbar; a.mean_in = gbar_mean ? d + 2 * mat + 256 : nullptr;
        a.Ut = d; a.Uo = d + mat; a.Gv = d + 2 * mat; a.sinv = d + 2 * mat + 128;
        a.out = d + 2 * mat + 384;
        a.n_chains = n_chains; a.n_tau = n_tau; a.gbar_stride = gbar_stride;
        hipLaunchKernelGGL(sacdev::cv_score_kernel<0>, dim3((unsigned)(1 + n_chains)), dim3(64), 0, stream, a);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e == hipSuccess) {
            std::vector<double> out(1 + (size_t)n_chains);
            e = copy_sync(stream, out.data(), a.out, out.size() * sizeof(double), hipMemcpyDeviceToHost);
            if (e == hipSuccess) {
                if (chi2_mean) *chi2_mean = out[0];
                if (chi2_chain) std::memcpy(chi2_chain, out.data() + 1, (size_t)n_chains * sizeof(double));
            }
        }
    }
    (void)hipFree(d);
    return e;
}
