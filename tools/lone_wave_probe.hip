// What one wavefront alone on a SIMD can issue (gfx950): cycles per instruction of dependent and independent fp64 chains,
// DPP tree levels, scalar/vector hand-overs -- the budget behind the per-chain latency of the sampling kernels.
//   hipcc -O3 --offload-arch=gfx950 -o tools/lone_wave_probe tools/lone_wave_probe.hip && tools/lone_wave_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 256

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

__global__ void probe(double *out, unsigned long long *cyc, unsigned long long *real, double seed)
{
    const int lane = threadIdx.x & 63;
    double a = seed + lane, b = seed * 2 + lane, c = seed * 3 + lane, d = seed * 0.5;
    unsigned long long t0, t1, r0, r1;
    int k = 0;
#define BEGIN() do { r0 = __builtin_readcyclecounter(); t0 = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define END()   do { __builtin_amdgcn_sched_barrier(0); t1 = clock64(); r1 = __builtin_readcyclecounter(); if (lane == 0) { cyc[k] = t1 - t0; real[k] = r1 - r0; } ++k; } while (0)
    // 0: dependent v_add_f64 chain
    BEGIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) { a = a + d; asm volatile("" : "+v"(a)); }
    END();
    // 1: three independent v_add_f64 chains, interleaved
    BEGIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) { a = a + d; b = b + d; c = c + d; asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }
    END();
    // 2: dependent tree levels (2 DPP movs + add), one chain
    BEGIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) { a = a + dpp_f64<0xB1>(a); asm volatile("" : "+v"(a)); }
    END();
    // 3: three tree chains interleaved
    BEGIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) {
        const double ta = dpp_f64<0xB1>(a), tb = dpp_f64<0xB1>(b), tc = dpp_f64<0xB1>(c);
        a = a + ta; b = b + tb; c = c + tc;
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
    }
    END();
    // 4: v_readlane -> scalar -> vector add (scalar/vector hand-over), dependent
    BEGIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) {
        const uint64_t bits = __builtin_bit_cast(uint64_t, a);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bits, 63);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(bits >> 32), 63);
        const double s = __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
        a = s + d;
        asm volatile("" : "+v"(a));
    }
    END();
    // 5: independent v_mov_b32 (pure issue rate of simple VALU)
    {
        int x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
        BEGIN();
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        }
        END();
        a += x0 + x1 + x2 + x3;
    }
    ++k;   // (6: unused)
    // 7: v_cmp -> vcc -> scalar test -> (untaken) branch, dependent on a vector value
    BEGIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(a > 1e300);
        if (m != 0ull) a = a * 0.5;
        a = a + d;
        asm volatile("" : "+v"(a));
    }
    END();
    // 8: independent v_mul_f64 (4 chains)
    {
        double e = seed + 4;
        BEGIN();
#pragma unroll
        for (int i = 0; i < REP; ++i) { a = a * d; b = b * d; c = c * d; e = e * d; asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(e)); }
        END();
        a += e;
    }
    // 9: independent v_mul_hi_u32 (4 chains): integer multiplies run at a quarter of the rate
    {
        uint32_t x0 = lane + 1, x1 = lane + 2, x2 = lane + 3, x3 = lane + 4;
        const uint32_t m = (uint32_t)seed + 0xD2511F52u;
        BEGIN();
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(m));
        }
        END();
        a += x0 + x1 + x2 + x3;
    }
    // 10: independent v_mad_u64_u32 (4 chains): the full 64-bit product in one instruction
    {
        uint32_t x0 = lane + 1, x1 = lane + 2, x2 = lane + 3, x3 = lane + 4;
        unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0;
        const uint32_t m = (uint32_t)seed + 0xD2511F52u;
        BEGIN();
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, 0\n v_mad_u64_u32 %1, vcc, %5, %8, 0\n v_mad_u64_u32 %2, vcc, %6, %8, 0\n v_mad_u64_u32 %3, vcc, %7, %8, 0"
                         : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(m) : "vcc");
            x0 = (uint32_t)(p0 >> 32); x1 = (uint32_t)(p1 >> 32); x2 = (uint32_t)(p2 >> 32); x3 = (uint32_t)(p3 >> 32);
        }
        END();
        a += (double)(p0 + p1 + p2 + p3);
    }
    // 11: independent v_mul_lo_u32 (4 chains)
    {
        uint32_t x0 = lane + 1, x1 = lane + 2, x2 = lane + 3, x3 = lane + 4;
        const uint32_t m = (uint32_t)seed + 0xD2511F52u;
        BEGIN();
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(m));
        }
        END();
        a += x0 + x1 + x2 + x3;
    }
    out[blockIdx.x * 64 + lane] = a + b + c;
}

int main()
{
    double *out; unsigned long long *cyc, *real;
    hipMalloc(&out, 64 * 8 * 1024); hipMalloc(&cyc, 16 * 8); hipMalloc(&real, 16 * 8);
    const char *names[] = {"dependent v_add_f64 (per add)", "3 independent v_add_f64 chains (per add)", "dependent tree level: 2 DPP mov + add (per level)",
                           "3 interleaved tree levels (per level)", "readlane x2 -> v_add_f64 with scalar operand (per round trip)",
                           "independent v_add_u32 (per instr)", "independent s_add_u32 (per instr)", "v_cmp + ballot test + scalar branch + add (per round)",
                           "4 independent v_mul_f64 chains (per mul)", "4 independent v_mul_hi_u32 chains (per instr)",
                           "4 independent v_mad_u64_u32 chains (per instr)", "4 independent v_mul_lo_u32 chains (per instr)"};
    const int per[] = {REP, 3 * REP, REP, 3 * REP, REP, 4 * REP, 4 * REP, REP, 4 * REP, 4 * REP, 4 * REP, 4 * REP};
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, real, 1.0);
        hipDeviceSynchronize();
    }
    unsigned long long h[16], r[16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(r, real, sizeof(r), hipMemcpyDeviceToHost);
    for (int k = 0; k < 12; ++k)
        printf("%-62s s_memtime %8.2f   s_memrealtime ticks %8.3f\n", names[k], (double)h[k] / per[k], (double)r[k] / per[k]);
    // the clock: a long dependent chain timed by the host
    return 0;
}
