// Wave64 helpers for the SAC sweep kernels (gfx950 only).
//
// Arithmetic contracts shared BY SPECIFICATION with the CPU oracle (DESIGN.md "Arithmetic
// spec"); the oracle has its own independent C implementation of each:
//   * philox4x32_10        -- counter-based draws
//   * u01 / index_from_word
//   * log_u01              -- ln of a draw, fdlibm-style, < 1 ulp
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

// v's image under a DPP lane permutation; every lane is written (bound_ctrl), no old value needed
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// row broadcast (lane 15 of the previous row / lane 31) into the rows of ROW_MASK; other rows get +0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_bcast_f64(double v)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// Sum of one double per lane in the canonical tree order: levels pair lane l with l^1, l^2,
// l^4 ... l^32 (fp add is commutative, so every lane of a pair holds the same bits).  Levels 1-2
// are quad permutes, 4 and 8 the half-row / row mirrors (a lane's mirror image lies in the other
// half, whose lanes all hold the same partial sum); then row sums R0..R3: rows 1 and 3 add their
// predecessor's (row_bcast15: R1+R0, R3+R2), row 3 adds lane 31's (R3+R2)+(R1+R0), and lane 63 is
// read into scalar registers.  The result is wave-uniform.
// All levels as in wave_sum_uniform, result valid in lane 63 only (no scalar read-back): for callers that can
// decide from lane 63 directly.  The two row_bcast levels write every row (lanes without a source read +0):
// rows 0 and 2 then hold values that are never used, row 3 / lane 63 holds ((R3+R2)+(R1+R0)).
__device__ __forceinline__ double wave_sum_lane63(double v)
{
    v = v + dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]  : l ^ 1
    v = v + dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]  : l ^ 2
    v = v + dpp_f64<0x141>(v);   // row_half_mirror      : other quad of the 8-group
    v = v + dpp_f64<0x140>(v);   // row_mirror           : other half of the 16-row
    v = v + dpp_f64<0x142>(v);   // row_bcast15: row r += lane 15 of row r-1  (R1+R0, R3+R2)
    v = v + dpp_f64<0x143>(v);   // row_bcast31: rows 2, 3 += lane 31         ((R3+R2)+(R1+R0))
    return v;
}

__device__ __forceinline__ double wave_sum_uniform(double v)
{
    return readlane_f64(wave_sum_lane63(v), 63);
}

// ---- fairness between the waves of a SIMD ------------------------------------------------
// The instruction arbiter serves the waves of a SIMD oldest first: of four resident chains the one in
// wave slot 0 ran 1.7 x as fast as the one in slot 3 (79 M against 134 M cycles per 25-sweep launch,
// identical from launch to launch), and a launch lasts as long as its slowest chain.  Each wave therefore
// walks through the four user priorities (s_setprio), starting from its slot number: every chain spends the
// same share of its time at each level and they finish together.
struct WavePriority {
    int slot, shift;
    unsigned long long t;       // clock read at the previous call (used one call later: never waited for)
    __device__ __forceinline__ void init(int shift_)
    {
        slot = (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 0xFu);   // HW_ID.wave_id
        shift = shift_;
        t = __builtin_amdgcn_s_memtime();
    }
    // The level follows the clock (one step every 2^shift cycles), so the waves of a SIMD change together and
    // hold four different levels at all times, whatever their progress.  Long steps run fastest (2^22 cycles gave
    // +5 % over 2^14) as long as a launch sees several full rotations: the host picks the shift from the
    // measured duration of the previous launches (prio_shift_for()).
    __device__ __forceinline__ void rotate()
    {
        const int level = (slot + (int)(uint32_t)(t >> shift)) & 3;
        t = __builtin_amdgcn_s_memtime();
        switch (level) {                               // s_setprio takes an immediate
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
    }
};

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

// ---- Metropolis ----------------------------------------------------------------------

// ln(u) for a draw u in [0,1) (multiples of 2^-53); ln(0) = -inf.  fdlibm-style reduction
// u = 2^k (1+f), s = f/(2+f), degree-14 even polynomial; basic IEEE operations only, in this
// fixed order, so the CPU oracle's own implementation of the same sequence agrees bit for bit.
__device__ __forceinline__ double log_u01(double u)
{
    if (!(u > 0.0)) return -__builtin_inf();
    const uint64_t bits = __builtin_bit_cast(uint64_t, u);
    uint32_t hx = (uint32_t)(bits >> 32);
    const uint32_t lx = (uint32_t)bits;
    int k = (int)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    const uint32_t i = (hx + 0x95f64u) & 0x100000u;
    const double m = __builtin_bit_cast(double, ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | lx);
    k += (int)(i >> 20);
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double R = t2 + t1;
    const double hfsq = (0.5 * f) * f;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// exp(x), |x| < 700: fdlibm e_exp.c scheme, IEEE operations in this fixed order (oracle: sac_exp)
__device__ __forceinline__ double sac_exp(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                 P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    double hi = x, lo = 0.0;
    int k = 0;
    if (__builtin_fabs(x) > 0.34657359027997264) {
        k = (int)(invln2 * x + ((x < 0.0) ? -0.5 : 0.5));
        const double t = (double)k;
        hi = x - t * ln2HI;
        lo = t * ln2LO;
    }
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((r * c) / (c - 2.0) - r);
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    uint64_t b = __builtin_bit_cast(uint64_t, y);
    b += (uint64_t)((int64_t)k << 52);
    return __builtin_bit_cast(double, b);
}

// Metropolis test.  The reference draws u = rand() and accepts when u <= exp((chi2-chi2_new)/(2 theta))
// (free:307-309); here, equivalently, when (chi2 - chi2_new) * (1/(2 theta)) >= ln u, with ln u
// prepared per proposal off the sequential path.  NaN never accepts.
__device__ __forceinline__ bool metropolis(double chi2, double chi2_new, double inv2theta, double ln_u)
{
    return (chi2 - chi2_new) * inv2theta >= ln_u;
}

// A condition that holds in lane 63 (every lane, when its inputs are wave-uniform), as a scalar: the compare mask's
// top bit is tested on the scalar unit and branches on it are scalar branches.
__device__ __forceinline__ bool lane63_true(bool cond)
{
    uint32_t hi = (uint32_t)(__builtin_amdgcn_ballot_w64(cond) >> 32);
    asm("" : "+s"(hi));
    return (hi >> 31) != 0u;
}

}  // namespace sacdev

// Host side: clock shift of the priority rotation for a launch expected to last `launch_ms` (0 = unknown):
// about eight full rotations (32 steps) per launch, a step between 2^12 and 2^22 cycles (2.4 GHz).
static inline int prio_shift_for(double launch_ms)
{
    if (!(launch_ms > 0.0)) return 17;
    const double step_cycles = launch_ms * 2.4e6 / 32.0;
    int sh = 12;
    while (sh < 22 && (double)(1ull << (sh + 1)) <= step_cycles) ++sh;
    return sh;
}
