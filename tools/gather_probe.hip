// Measured ceiling for the sweep kernel's access pattern: one wavefront gathers random 512-B
// rows (8 B per lane, one coalesced wave load per row) of a table, DEPTH rows in flight, 16 waves
// per CU -- the same shape as free_stage_kernel's K-column gathers, with all arithmetic removed.
// Also the calibration run for FETCH_SIZE on this access width (known byte count).
//   gather_probe <table_MiB> <rows_per_wave> [waves]
//   gather_probe --rows3 <table_MiB> <rows_per_wave> <rows_in_flight 1|2|4|8>
//       the edge sampler's access: random 1536-B rows (K, dK, d2K of one grid column: three 512-B wave loads), a given number of
//       rows in flight per wave and a full wait between batches -- one dependent batch per proposal, as edge_stage_kernel has it
//   gather_probe --positions <file> <rows_per_wave> [rows_in_flight 8|16|32] [x2|x4] [waves] [flat]
//       the sampler's OWN column distribution: <file> (tools/dump_positions.py) holds, for every chain of a tuned run at the
//       bench shape, its delta positions and its tuned single-move window; wave c then gathers what chain c's single moves
//       gather -- the column of a random delta of the chain and the column a window shift away (free:283-302) -- with the
//       chain's positions in LDS as in the sampler.  This is the gather ceiling for the concentrated, cache-resident access
//       pattern of BASELINE configs[1] (a uniform-random probe is not: the sampler beats it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>

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

// random 1536-B rows, INF of them in flight, then a full wait (the edge kernel's one dependent batch per proposal)
template <int INF>
__global__ void __launch_bounds__(64) gather_rows3(const double *table, uint32_t n_rows, int iters, double *out)
{
    const uint32_t wave = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t s = wave * 2654435761u + 12345u;
    double acc = 0.0;
    for (int it = 0; it < iters; it += INF) {
        double v[INF][3];
#pragma unroll
        for (int d = 0; d < INF; ++d) {
            s = s * 1664525u + 1013904223u;
            s ^= (uint32_t)(acc != 123.456 ? 0u : 1u);              // next rows depend on the batch before (a chain)
            const uint32_t row = __builtin_amdgcn_readfirstlane((uint32_t)(((uint64_t)s * n_rows) >> 32));
#pragma unroll
            for (int q = 0; q < 3; ++q) v[d][q] = table[(size_t)row * 192 + q * 64 + lane];
        }
#pragma unroll
        for (int d = 0; d < INF; ++d) acc += (v[d][0] + v[d][1]) + v[d][2];
    }
    if (acc == 123.456) out[wave] = acc;
}

template <int INF>
static int run_rows3(double mib, int iters, int waves)
{
    const uint32_t n_rows = (uint32_t)(mib * 1048576.0 / 1536.0);
    double *table, *out;
    if (hipMalloc(&table, (size_t)n_rows * 1536) != hipSuccess) { fprintf(stderr, "hipMalloc\n"); return 1; }
    hipMalloc(&out, waves * sizeof(double));
    hipMemset(table, 0, (size_t)n_rows * 1536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int it = (iters / INF) * INF;
    gather_rows3<INF><<<waves, 64>>>(table, n_rows, it / 10, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    gather_rows3<INF><<<waves, 64>>>(table, n_rows, it, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)waves * it * 1536.0;
    printf("{\"pattern\": \"random 1536-B rows, dependent batches\", \"table_MiB\": %.1f, \"rows\": %u, \"waves\": %d, \"rows_in_flight\": %d, "
           "\"rows_per_wave\": %d, \"bytes\": %.0f, \"ms\": %.3f, \"GBps\": %.1f, \"us_per_batch\": %.3f}\n", mib, n_rows, waves, INF, it, bytes, ms,
           bytes / ms / 1e6, ms * 1e3 / (it / INF));
    return 0;
}

// pairs (new, old) as chain `wave`'s single-delta proposals request them.  D rows in flight per wave; X4: one 16 B/lane
// wave load fetches BOTH rows of a pair (lanes 0-31 the new column, lanes 32-63 the old one), as the packed kernels do
template <int D, bool X4>
__global__ void __launch_bounds__(256) gather_pairs(const double *table, uint32_t n_rows, const uint32_t *pos, const double *window,
                                                    int n_omega, int n_chains, int iters, double *out)
{
    __shared__ uint32_t s_pos[4][2048];
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t chain = wave % (uint32_t)n_chains;
    const uint32_t wl = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < n_omega; i += 64) s_pos[wl][i] = pos[(size_t)chain * n_omega + i];
    const double w = window[chain];
    uint32_t s = wave * 2654435761u + 12345u;
    double acc = 0.0;
    constexpr int PAIRS = D / 2;
    for (int it = 0; it < iters; it += D) {
        double v[D];
#pragma unroll
        for (int d = 0; d < PAIRS; ++d) {
            s = s * 1664525u + 1013904223u;
            const uint32_t k = (uint32_t)(((uint64_t)s * (uint32_t)n_omega) >> 32);
            s = s * 1664525u + 1013904223u;
            const int shift = 1 + (int)((double)(s >> 8) * (1.0 / 16777216.0) * w);
            const int old = (int)s_pos[wl][k];
            int nw = (s & 1u) ? old + shift : old - shift;
            nw = nw < 0 ? 0 : (nw >= (int)n_rows ? (int)n_rows - 1 : nw);
            if constexpr (X4) {
                const uint32_t r = (lane < 32) ? (uint32_t)nw : (uint32_t)old;
                const double2 t = *reinterpret_cast<const double2 *>(table + (size_t)r * 64 + (lane & 31) * 2);
                v[2 * d] = t.x;
                v[2 * d + 1] = t.y;
            } else {
                const uint32_t r_old = __builtin_amdgcn_readfirstlane((uint32_t)old), r_new = __builtin_amdgcn_readfirstlane((uint32_t)nw);
                v[2 * d] = table[(size_t)r_new * 64 + lane];
                v[2 * d + 1] = table[(size_t)r_old * 64 + lane];
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) acc += v[d];
    }
    if (acc == 123.456) out[wave] = acc;
}

// --positions <file> <rows_per_wave> [rows_in_flight 8|16|32] [x2|x4] [waves] [flat]
//   flat: every chain's positions are replaced by uniformly random rows of the table (the deliberately flat-support case)
static int run_positions(const char *path, int iters, int depth, bool x4, int waves_arg, bool flat)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); return 1; }
    int32_t hdr[4];                                  // chains, n_omega, n_rows, reserved
    if (fread(hdr, sizeof(hdr), 1, f) != 1) return 1;
    const int chains = hdr[0], n_omega = hdr[1];
    const uint32_t n_rows = (uint32_t)hdr[2];
    const int waves = waves_arg > 0 ? waves_arg : chains;
    uint32_t *h_pos = (uint32_t *)malloc((size_t)chains * n_omega * 4);
    double *h_win = (double *)malloc((size_t)chains * 8);
    if (fread(h_pos, 4, (size_t)chains * n_omega, f) != (size_t)chains * n_omega) return 1;
    if (fread(h_win, 8, (size_t)chains, f) != (size_t)chains) return 1;
    fclose(f);
    if (n_omega > 2048 || waves % 4) { fprintf(stderr, "shape\n"); return 1; }
    if (flat) {
        uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < (size_t)chains * n_omega; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            h_pos[i] = (uint32_t)(((x >> 32) * (uint64_t)n_rows) >> 32);
        }
    }
    double *table, *out, *d_win;
    uint32_t *d_pos;
    if (hipMalloc(&table, (size_t)n_rows * 512) != hipSuccess) { fprintf(stderr, "hipMalloc\n"); return 1; }
    hipMalloc(&out, waves * sizeof(double));
    hipMalloc(&d_pos, (size_t)chains * n_omega * 4);
    hipMalloc(&d_win, (size_t)chains * 8);
    hipMemset(table, 0, (size_t)n_rows * 512);
    hipMemcpy(d_pos, h_pos, (size_t)chains * n_omega * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_win, h_win, (size_t)chains * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int it = (iters / depth) * depth;
    auto launch = [&](int n) {
#define GP(D, X) gather_pairs<D, X><<<waves / 4, 256>>>(table, n_rows, d_pos, d_win, n_omega, chains, n, out)
        if (depth == 8) { if (x4) GP(8, true); else GP(8, false); }
        else if (depth == 32) { if (x4) GP(32, true); else GP(32, false); }
        else { if (x4) GP(16, true); else GP(16, false); }
#undef GP
    };
    launch(it / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(it);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)waves * it * 512.0;
    double wmean = 0;
    for (int i = 0; i < chains; ++i) wmean += h_win[i] / chains;
    printf("{\"pattern\": \"%s\", \"table_MiB\": %.1f, \"rows\": %u, \"waves\": %d, \"rows_in_flight\": %d, "
           "\"load\": \"%s\", \"rows_per_wave\": %d, \"mean_window\": %.1f, \"bytes\": %.0f, \"ms\": %.3f, \"GBps\": %.1f}\n",
           flat ? "flat support: uniformly random positions + window shift" : "sampler positions + window shift",
           n_rows * 512.0 / 1048576.0, n_rows, waves, depth, x4 ? "16 B/lane, two rows per wave load" : "8 B/lane, one row per wave load",
           it, wmean, bytes, ms, bytes / ms / 1e6);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 4 && !strcmp(argv[1], "--rows3")) {
        const double mib = atof(argv[2]);
        const int iters = atoi(argv[3]), inf = atoi(argv[4]), waves = argc > 5 ? atoi(argv[5]) : 4096;
        if (inf == 1) return run_rows3<1>(mib, iters, waves);
        if (inf == 2) return run_rows3<2>(mib, iters, waves);
        if (inf == 4) return run_rows3<4>(mib, iters, waves);
        if (inf == 8) return run_rows3<8>(mib, iters, waves);
        fprintf(stderr, "rows_in_flight must be 1, 2, 4 or 8\n");
        return 1;
    }
    if (argc > 2 && !strcmp(argv[1], "--positions")) {
        const int iters = argc > 3 ? atoi(argv[3]) : 200000, depth = argc > 4 ? atoi(argv[4]) : 16;
        const bool x4 = argc > 5 && !strcmp(argv[5], "x4");
        const int waves = argc > 6 ? atoi(argv[6]) : 0;
        const bool flat = argc > 7 && !strcmp(argv[7], "flat");
        if (depth != 8 && depth != 16 && depth != 32) { fprintf(stderr, "rows_in_flight must be 8, 16 or 32\n"); return 1; }
        return run_positions(argv[2], iters, depth, x4, waves, flat);
    }
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
