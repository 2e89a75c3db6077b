// K-table construction on the device, gfx950.
//
// What the reference does on the host once per run (free:199-213, peak:248-276, edge:167-236):
//     K_array[:, i] = transpose(cov) * kernel(omega_i, tau)          for every point of the fine grid
// i.e. an elementwise K(omega_i, tau_j) followed by the basis change into the covariance eigenbasis,
// one [n_grid x n_tau] . [n_tau x n_tau] fp64 product -- the one dense contraction of the code base, done
// here with v_mfma_f64_16x16x4_f64.  For edge the first and second derivative tables are the reference's
// finite-difference stencils of that K (edge:230-235).
//
// Not bit-identical to a host-built table (different exp, different summation order): the host path
// (sac_upload_problem*) stays the one used for move-for-move parity; a table built here can be read back
// (sac_get_table / sac_edge_get_tables) so that a checker runs on exactly the table the device uses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wave_ops.h"   // sac_exp

namespace sacdev {

struct TableSpec {
    int kernel_type;      // 0 finiteT, 1 zeroT, 2 bosonic (free:220-239)
    int n_tau;
    int mirror;           // K(w) + K(-w) (symm: free:206-208, peak:258-262, edge:223)
    int negate;           // evaluate at -w (negative axis: peak:270-272, edge R side)
    long long i_first;    // w_i = ((i_first + i) + offset) * d_omega
    double offset;        // 0, or 1/2 on peak's half-offset grid (peak:253)
    double d_omega, beta;
    long long n_grid;
    int ld;               // doubles between consecutive grid rows of `out`
    int n_cols;           // columns of a row to write (>= n_tau: the zero padding is written too)
};

// exp over the whole double range (sac_exp covers |x| < 700)
__device__ __forceinline__ double exp_full(double x)
{
    if (x > 709.0) return __builtin_inf();
    if (x < -745.2) return 0.0;
    if (x < -700.0) { const double h = sac_exp(0.5 * x); return h * h; }
    return sac_exp(x);
}

__device__ __forceinline__ double kernel_value(int type, double w, double t, double beta)
{
    if (type == 1) return exp_full(-w * t);                                   // zeroT
    if (type == 2)                                                            // bosonic (free:237-239)
        return (exp_full(-w * t) + exp_full(-w * (beta - t))) / (1.0 + exp_full(-beta * w));
    const double num = -w * t, den = -beta * w;                               // finiteT: e^{-wt} / (1 + e^{-beta w})
    double m = num > den ? num : den;
    m = m > 0.0 ? m : 0.0;
    return exp_full(num - m) / (exp_full(-m) + exp_full(den - m));
}

typedef double table_acc_t __attribute__((ext_vector_type(4)));

// One wave = 16 grid rows.  LDS: U zero padded to [KP][NP], KP = ceil4(n_tau), NP = ceil16(n_tau).
// v_mfma_f64_16x16x4_f64: A lane l = A[row l&15][k l>>4], B lane l = B[k l>>4][col l&15],
// D lane l, register r = D[row (l>>4) + 4r][col l&15].
template <int NTILES>
static __global__ void __launch_bounds__(256) table_build_kernel(const TableSpec sp, const double *__restrict__ tau,
                                                                 const double *__restrict__ U, double *__restrict__ out)
{
    extern __shared__ __align__(16) double s_U[];
    const int nt = sp.n_tau;
    const int KP = (nt + 3) & ~3, NP = NTILES * 16;
    for (int e = (int)threadIdx.x; e < KP * NP; e += (int)blockDim.x) {
        const int k = e / NP, c = e - k * NP;
        s_U[e] = (k < nt && c < nt) ? U[(size_t)k * nt + c] : 0.0;
    }
    __syncthreads();
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const long long row0 = ((long long)blockIdx.x * 4 + wave) * 16;
    if (row0 >= sp.n_grid) return;
    const int r = lane & 15, kq = lane >> 4;
    long long i = row0 + r;
    if (i >= sp.n_grid) i = sp.n_grid - 1;                                    // clamped rows are computed, not stored
    double w = ((double)(sp.i_first + i) + sp.offset) * sp.d_omega;
    if (sp.negate) w = -w;
    table_acc_t acc[NTILES];
#pragma unroll
    for (int n = 0; n < NTILES; ++n) acc[n] = table_acc_t{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < KP; k0 += 4) {
        const int k = k0 + kq;
        double a = 0.0;
        if (k < nt) {
            const double t = tau[k];
            a = kernel_value(sp.kernel_type, w, t, sp.beta);
            if (sp.mirror) a = a + kernel_value(sp.kernel_type, -w, t, sp.beta);
        }
#pragma unroll
        for (int n = 0; n < NTILES; ++n) {
            const double b = s_U[k * NP + n * 16 + r];
            acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[n], 0, 0, 0);
        }
    }
#pragma unroll
    for (int n = 0; n < NTILES; ++n) {
        const int col = n * 16 + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long row = row0 + kq + 4 * q;
            if (row < sp.n_grid && col < sp.n_cols) out[(size_t)row * (size_t)sp.ld + col] = acc[n][q];
        }
    }
}

// edge:230-235 on the interleaved edge layout [n_grid][3][ntp]: slot 0 = K (already built), 1 = dK, 2 = d2K.
// idx_m = the reference's 1-based index wi_m + 1 as a 0-based row (edge:235 addresses the array with the
// absolute wi_m whatever wi_0 is).
[[maybe_unused]] static __global__ void __launch_bounds__(256) edge_stencil_kernel(double *tab, long long n_grid, int ntp, long long idx_m,
                                                                   double dw, double two_dw, double two_dw2)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long i = e / ntp;
    const int t = (int)(e - i * ntp);
    if (i >= n_grid) return;
    const size_t ld = (size_t)3 * ntp;
    const double *K = tab;
    double dK = 0.0, d2K = 0.0;
    if (i >= 1 && i <= n_grid - 2) {
        const double kp = K[(size_t)(i + 1) * ld + t], k0 = K[(size_t)i * ld + t], km = K[(size_t)(i - 1) * ld + t];
        dK = (kp - km) / two_dw;
        d2K = ((kp - (2 * k0)) + km) / two_dw2;
    }
    if (i == 0) dK = (K[ld + t] - K[t]) / dw;                                  // edge:234
    if (i == idx_m && i >= 1) dK = (K[(size_t)i * ld + t] - K[(size_t)(i - 1) * ld + t]) / dw;   // edge:235
    tab[(size_t)i * ld + ntp + t] = dK;
    tab[(size_t)i * ld + 2 * (size_t)ntp + t] = d2K;
}

// Launches the build of one table on `stream`.  d_tau: n_tau doubles, d_U: [n_tau][n_tau] row-major (the reference's
// `cov`, eigenvectors in columns), both in device memory.  Returns hipSuccess or the launch error.
static inline hipError_t launch_table_build(hipStream_t stream, const TableSpec &sp, const double *d_tau, const double *d_U, double *out)
{
    const int ntiles = (sp.n_tau + 15) / 16;
    const int KP = (sp.n_tau + 3) & ~3, NP = ntiles * 16;
    const size_t lds = (size_t)KP * NP * sizeof(double);
    const dim3 grid((unsigned)((sp.n_grid + 63) / 64)), block(256);
#define SAC_TB(N_)                                                                                                   \
    do {                                                                                                             \
        if (lds > 48 * 1024) {                                                                                       \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&table_build_kernel<N_>),             \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
            if (e_ != hipSuccess) return e_;                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((table_build_kernel<N_>), grid, block, lds, stream, sp, d_tau, d_U, out);                 \
    } while (0)
    switch (ntiles) {
    case 1: SAC_TB(1); break;
    case 2: SAC_TB(2); break;
    case 3: SAC_TB(3); break;
    case 4: SAC_TB(4); break;
    case 5: SAC_TB(5); break;
    case 6: SAC_TB(6); break;
    case 7: SAC_TB(7); break;
    default: SAC_TB(8); break;
    }
#undef SAC_TB
    return hipGetLastError();
}

}  // namespace sacdev
