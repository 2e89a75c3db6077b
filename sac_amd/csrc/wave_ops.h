// Wave64 helpers for the SAC sweep kernels (gfx950 only).
//
// Arithmetic contracts shared BY SPECIFICATION with the CPU oracle (DESIGN.md "Arithmetic
// spec"); the oracle has its own independent C implementation of each:
//   * philox4x32_10        -- counter-based draws
//   * u01 / index_from_word
//   * exp_neg              -- exp(x), x < 0, fma Horner, < 1 ulp
//   * wave_sum_uniform     -- the canonical 64-slot pairwise tree
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sacdev {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// ---- cross-lane ------------------------------------------------------------------------

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// Sum of one double per lane in the canonical tree order: levels pair lane l with l^1, l^2,
// l^4 ... l^32 (fp add is commutative, so every lane of a pair holds the same bits).  Levels 1-2
// are quad permutes, 4 and 8 the half-row / row mirrors (a lane's mirror image lies in the other
// half, whose lanes all hold the same partial sum), 16 and 32 go through scalar registers.
// The result is wave-uniform.
__device__ __forceinline__ double wave_sum_uniform(double v)
{
    v = v + dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]  : l ^ 1
    v = v + dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]  : l ^ 2
    v = v + dpp_f64<0x141>(v);   // row_half_mirror      : other quad of the 8-group
    v = v + dpp_f64<0x140>(v);   // row_mirror           : other half of the 16-row
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16);
    const double r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return (r0 + r1) + (r2 + r3);
}

// ---- draws -----------------------------------------------------------------------------

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// U in [0,1): 53 bits from two words
__device__ __forceinline__ double u01(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// rand(1:n) - 1 from one word
__device__ __forceinline__ int index_from_word(uint32_t w, uint32_t n) { return (int)__umulhi(w, n); }

// ---- exp -------------------------------------------------------------------------------

// exp(x) for x < 0; 0 for x <= -708 or NaN.  Same operation sequence as the oracle's.
__device__ __forceinline__ double exp_neg(double x)
{
    if (!(x > -708.0)) return 0.0;
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821614599e-10;
    p = __builtin_fma(p, r, 2.0876756987868098979e-09);
    p = __builtin_fma(p, r, 2.5052108385441718775e-08);
    p = __builtin_fma(p, r, 2.7557319223985890653e-07);
    p = __builtin_fma(p, r, 2.7557319223985892511e-06);
    p = __builtin_fma(p, r, 2.4801587301587301566e-05);
    p = __builtin_fma(p, r, 1.9841269841269841253e-04);
    p = __builtin_fma(p, r, 1.3888888888888889419e-03);
    p = __builtin_fma(p, r, 8.3333333333333332177e-03);
    p = __builtin_fma(p, r, 4.1666666666666664354e-02);
    p = __builtin_fma(p, r, 1.6666666666666665741e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}

// Metropolis test  rand() <= exp((chi2 - chi2_new)/(2 theta))   (free:307-309)
__device__ __forceinline__ bool metropolis(double chi2, double chi2_new, double inv2theta, double u)
{
    const double x = (chi2 - chi2_new) * inv2theta;
    if (x != x) return false;
    if (x >= 0.0) return true;
    return u <= exp_neg(x);
}

}  // namespace sacdev
