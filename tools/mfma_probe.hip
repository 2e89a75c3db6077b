// Probe for the chi^2 reduction on the matrix pipe (gfx950): what v_mfma_f64_4x4x4f64 / v_mfma_f64_16x16x4f64 with a
// ones operand compute lane by lane (which lanes are summed, in which order -- needed bit for bit by the oracle), and
// what a 64-lane sum costs as (a) the 6-level DPP tree, (b) two 4x4x4 MFMAs + two DPP levels, (c) 4x4x4, 4x4x4, 16x16x4,
// with 4 waves per SIMD each running a dependent chain (the shape of the sweep kernels' resolve loop).
//   mfma_probe            prints the lane maps / orders found and the timings as JSON lines
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void sem_kernel(const double *a, const double *b, double *d4, double *d16)
{
    const int l = threadIdx.x;
    const double r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
    d4[l] = r;
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    const v4d q = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], z, 0, 0, 0);
    for (int k = 0; k < 4; ++k) d16[l * 4 + k] = q[k];
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

template <int MODE>
__device__ __forceinline__ double reduce(double v)
{
    if constexpr (MODE == 0) {
        v = v + dpp_f64<0xB1>(v);
        v = v + dpp_f64<0x4E>(v);
        v = v + dpp_f64<0x141>(v);
        v = v + dpp_f64<0x140>(v);
        v = v + dpp_f64<0x142>(v);
        v = v + dpp_f64<0x143>(v);
        return v;
    } else if constexpr (MODE == 1) {
        v = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
        v = v + dpp_f64<0x142>(v);
        v = v + dpp_f64<0x143>(v);
        return v;
    } else if constexpr (MODE == 2) {
        v = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
        const v4d z = {0.0, 0.0, 0.0, 0.0};
        const v4d q = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, z, 0, 0, 0);
        return q[0];
    } else if constexpr (MODE == 3) {      // 16x16x4 (fold the four 16-lane blocks), 3 adds, 16x16x4
        const v4d z = {0.0, 0.0, 0.0, 0.0};
        const v4d q = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, z, 0, 0, 0);
        const double u = (q[0] + q[1]) + (q[2] + q[3]);
        const v4d t = __builtin_amdgcn_mfma_f64_16x16x4f64(u, 1.0, z, 0, 0, 0);
        return t[0];
    } else {
        return v;                          // no reduction: the cost of the loop's other work
    }
}

// the dependent chain of a resolve step: EXTRA fp64 VALU operations, one 64-lane sum, a scalar decision on it
template <int MODE, int EXTRA>
__global__ void __launch_bounds__(256, 4) time_kernel(const double *in, double *out, int iters)
{
    const int l = threadIdx.x & 63;
    double g = in[l], acc = 0.0;
    const double G = in[64 + l], s = in[128 + l];
    for (int it = 0; it < iters; ++it) {
        double x = g;
#pragma unroll
        for (int e = 0; e < EXTRA; ++e) x = (x - G) * s;
        const double r = reduce<MODE>(x * x);
        if (__builtin_amdgcn_readlane((int)(__builtin_bit_cast(uint64_t, r) >> 32), 63) & 1) g = g + 1e-9;
        acc = acc + r;
    }
    if (acc == 123.456) out[blockIdx.x * 256 + threadIdx.x] = acc + g;
}

template <int MODE, int EXTRA>
static double time_it(const double *d_in, double *d_out, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    time_kernel<MODE, EXTRA><<<1024, 256>>>(d_in, d_out, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    time_kernel<MODE, EXTRA><<<1024, 256>>>(d_in, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    std::vector<double> a(64), ones(64, 1.0), d4(64), d16(256);
    srand(7);
    for (int l = 0; l < 64; ++l) {           // mixed magnitudes: every summation order rounds differently
        const double m = (double)(rand() % 1000003) / 1000003.0 + 0.5;
        a[l] = std::ldexp(m, (rand() % 40) - 20) * ((rand() & 1) ? 1.0 : -1.0);
    }
    double *da, *db, *d_4, *d_16;
    hipMalloc(&da, 64 * 8); hipMalloc(&db, 64 * 8); hipMalloc(&d_4, 64 * 8); hipMalloc(&d_16, 256 * 8);
    for (int pass = 0; pass < 2; ++pass) {   // pass 0: data in A, ones in B; pass 1: ones in A, data in B
        hipMemcpy(da, pass == 0 ? a.data() : ones.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(db, pass == 0 ? ones.data() : a.data(), 64 * 8, hipMemcpyHostToDevice);
        sem_kernel<<<1, 64>>>(da, db, d_4, d_16);
        hipMemcpy(d4.data(), d_4, 64 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(d16.data(), d_16, 256 * 8, hipMemcpyDeviceToHost);
        // 4x4x4: for every output lane find 4 source lanes (within its 16-lane block) and an order that reproduce it
        printf("pass %d (%s holds the data), v_mfma_f64_4x4x4f64:\n", pass, pass == 0 ? "A" : "B");
        for (int l = 0; l < 64; ++l) {
            const int blk = l & ~15;
            bool found = false;
            for (int s0 = 0; s0 < 16 && !found; ++s0) for (int s1 = 0; s1 < 16 && !found; ++s1)
            for (int s2 = 0; s2 < 16 && !found; ++s2) for (int s3 = 0; s3 < 16 && !found; ++s3) {
                if (s0 == s1 || s0 == s2 || s0 == s3 || s1 == s2 || s1 == s3 || s2 == s3) continue;
                const double x0 = a[blk + s0], x1 = a[blk + s1], x2 = a[blk + s2], x3 = a[blk + s3];
                const double seq = ((x0 + x1) + x2) + x3;
                const double pw = (x0 + x1) + (x2 + x3);
                if (seq == d4[l] && !(s0 > s1)) { printf("  lane %2d = ((L%d + L%d) + L%d) + L%d  (block-relative)%s\n", l, s0, s1, s2, s3, pw == d4[l] ? " [pairwise also fits]" : ""); found = true; }
                else if (pw == d4[l] && s0 < s1 && s2 < s3 && s0 < s2) { printf("  lane %2d = (L%d + L%d) + (L%d + L%d)  (block-relative)\n", l, s0, s1, s2, s3); found = true; }
            }
            if (!found) {
                // exact sum of some 4 lanes rounded once (fused / wide accumulation)?
                for (int s0 = 0; s0 < 16 && !found; ++s0) for (int s1 = s0 + 1; s1 < 16 && !found; ++s1)
                for (int s2 = s1 + 1; s2 < 16 && !found; ++s2) for (int s3 = s2 + 1; s3 < 16 && !found; ++s3) {
                    const long double e = (long double)a[blk + s0] + a[blk + s1] + a[blk + s2] + a[blk + s3];
                    if (std::fabs((double)e - d4[l]) <= 4e-16 * std::fabs(d4[l]) + 1e-300) { printf("  lane %2d ~ sum of L%d L%d L%d L%d, no simple order fits (got %.17g, long double %.17Lg)\n", l, s0, s1, s2, s3, d4[l], e); found = true; }
                }
            }
            if (!found) printf("  lane %2d: no 4-lane sum found (%.17g)\n", l, d4[l]);
        }
        printf("pass %d, v_mfma_f64_16x16x4f64 (4 results per lane):\n", pass);
        for (int l = 0; l < 64; l += 5) for (int r = 0; r < 4; ++r) {
            bool found = false;
            for (int i = 0; i < 16 && !found; ++i) {
                // candidates: source lanes i + 16 k, k = 0..3, in every order
                int p[4] = {0, 1, 2, 3};
                for (int perm = 0; perm < 24 && !found; ++perm) {
                    const double x0 = a[i + 16 * p[0]], x1 = a[i + 16 * p[1]], x2 = a[i + 16 * p[2]], x3 = a[i + 16 * p[3]];
                    if (((x0 + x1) + x2) + x3 == d16[l * 4 + r]) { printf("  lane %2d reg %d = ((L%d + L%d) + L%d) + L%d\n", l, r, i + 16 * p[0], i + 16 * p[1], i + 16 * p[2], i + 16 * p[3]); found = true; }
                    else if ((x0 + x1) + (x2 + x3) == d16[l * 4 + r]) { printf("  lane %2d reg %d = (L%d + L%d) + (L%d + L%d)\n", l, r, i + 16 * p[0], i + 16 * p[1], i + 16 * p[2], i + 16 * p[3]); found = true; }
                    // next permutation
                    int k = 2; while (k >= 0 && p[k] > p[k + 1]) --k;
                    if (k < 0) break;
                    int m = 3; while (p[m] < p[k]) --m;
                    int t = p[k]; p[k] = p[m]; p[m] = t;
                    for (int x = k + 1, y = 3; x < y; ++x, --y) { t = p[x]; p[x] = p[y]; p[y] = t; }
                }
            }
            if (!found) printf("  lane %2d reg %d: no fit (%.17g)\n", l, r, d16[l * 4 + r]);
        }
    }
    // timings
    std::vector<double> in(192);
    for (int l = 0; l < 64; ++l) { in[l] = 1.0 + l * 1e-3; in[64 + l] = 1.0; in[128 + l] = 0.999; }
    double *d_in, *d_out;
    hipMalloc(&d_in, 192 * 8);
    hipMalloc(&d_out, 1024 * 256 * 8);
    hipMemcpy(d_in, in.data(), 192 * 8, hipMemcpyHostToDevice);
    const int iters = 200000;
    const double ms_none = time_it<9, 4>(d_in, d_out, iters);
    const double ms_dpp = time_it<0, 4>(d_in, d_out, iters);
    const double ms_m1 = time_it<1, 4>(d_in, d_out, iters);
    const double ms_m2 = time_it<2, 4>(d_in, d_out, iters);
    const double ms_m3 = time_it<3, 4>(d_in, d_out, iters);
    printf("{\"iters\": %d, \"waves_per_simd\": 4, \"extra_valu\": 8, \"ms_no_reduce\": %.3f, \"ms_dpp_tree\": %.3f, \"ms_2x4x4x4_plus_2dpp\": %.3f, "
           "\"ms_2x4x4x4_plus_16x16x4\": %.3f, \"ms_16x16x4_3add_16x16x4\": %.3f}\n", iters, ms_none, ms_dpp, ms_m1, ms_m2, ms_m3);
    const double c = 2.4e6 / iters / 4.0;    // cycles per iteration per wave slot (4 waves share a SIMD)
    printf("{\"cycles_per_iter_per_simd_wave\": {\"none\": %.1f, \"dpp\": %.1f, \"m1\": %.1f, \"m2\": %.1f, \"m3\": %.1f}}\n",
           ms_none * c, ms_dpp * c, ms_m1 * c, ms_m2 * c, ms_m3 * c);
    const double ms_none16 = time_it<9, 12>(d_in, d_out, iters), ms_dpp16 = time_it<0, 12>(d_in, d_out, iters),
                 ms_m116 = time_it<1, 12>(d_in, d_out, iters), ms_m216 = time_it<2, 12>(d_in, d_out, iters);
    printf("{\"extra_valu\": 24, \"ms_no_reduce\": %.3f, \"ms_dpp_tree\": %.3f, \"ms_2x4x4x4_plus_2dpp\": %.3f, \"ms_2x4x4x4_plus_16x16x4\": %.3f}\n",
           ms_none16, ms_dpp16, ms_m116, ms_m216);
    return 0;
}
