// Measured ceiling for the sweep kernel's access pattern: one wavefront gathers random 512-B
// rows (8 B per lane, one coalesced wave load per row) of a table, DEPTH rows in flight, 16 waves
// per CU -- the same shape as free_stage_kernel's K-column gathers, with all arithmetic removed.
// Also the calibration run for FETCH_SIZE on this access width (known byte count).
//   gather_probe <table_MiB> <rows_per_wave> [waves]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

constexpr int DEPTH = 16;

__global__ void __launch_bounds__(256) gather_rows(const double *table, uint32_t n_rows, int iters, double *out)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t s = wave * 2654435761u + 12345u;
    double acc = 0.0;
    for (int it = 0; it < iters; it += DEPTH) {
        double v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            s = s * 1664525u + 1013904223u;                        // wave-uniform LCG
            const uint32_t row = __builtin_amdgcn_readfirstlane((uint32_t)(((uint64_t)s * n_rows) >> 32));
            v[d] = table[(size_t)row * 64 + lane];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
    if (acc == 123.456) out[wave] = acc;                           // keep the loads alive
}

int main(int argc, char **argv)
{
    const double mib = argc > 1 ? atof(argv[1]) : 49.0;
    const int iters = argc > 2 ? atoi(argv[2]) : 200000;
    const int waves = argc > 3 ? atoi(argv[3]) : 4096;
    const uint32_t n_rows = (uint32_t)(mib * 1048576.0 / 512.0);
    double *table, *out;
    hipMalloc(&table, (size_t)n_rows * 512);
    hipMalloc(&out, waves * sizeof(double));
    hipMemset(table, 0, (size_t)n_rows * 512);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int it = (iters / DEPTH) * DEPTH;
    gather_rows<<<waves / 4, 256>>>(table, n_rows, it / 10, out);    // warm
    hipDeviceSynchronize();
    hipEventRecord(e0);
    gather_rows<<<waves / 4, 256>>>(table, n_rows, it, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)waves * it * 512.0;
    printf("{\"table_MiB\": %.1f, \"rows\": %u, \"waves\": %d, \"rows_per_wave\": %d, \"bytes\": %.0f, "
           "\"ms\": %.3f, \"GBps\": %.1f}\n", mib, n_rows, waves, it, bytes, ms, bytes / ms / 1e6);
    return 0;
}
