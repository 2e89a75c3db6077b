#include <hip/hip_runtime.h>
#include <cstdio>
#include "../include/sac_hip.h"
#include "../sac_amd/csrc/free_sweep_packed.h"
using namespace sacdev;
__global__ void k(const double *in, double *o2, double *o4, double *o1) {
    int l = threadIdx.x;
    o2[l] = chain_sum<2>(in[l]);
    o4[l] = chain_sum<4>(in[l]);
    o1[l] = wave_sum_uniform(in[l]);
}
int main() {
    double h[64], r2[64], r4[64], r1[64], *d, *d2, *d4, *d1;
    for (int i = 0; i < 64; ++i) h[i] = 1.0 / (i + 3.7) + i * 1e-3;
    hipMalloc(&d, 512); hipMalloc(&d2, 512); hipMalloc(&d4, 512); hipMalloc(&d1, 512);
    hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, d2, d4, d1);
    hipMemcpy(r2, d2, 512, hipMemcpyDeviceToHost); hipMemcpy(r4, d4, 512, hipMemcpyDeviceToHost); hipMemcpy(r1, d1, 512, hipMemcpyDeviceToHost);
    auto tree = [&](int first, int n) { double v[64]; for (int i = 0; i < n; ++i) v[i] = h[first + i];
        for (int s = 1; s < n; s <<= 1) { double w[64]; for (int i = 0; i < n; ++i) w[i] = v[i] + v[i ^ s]; for (int i = 0; i < n; ++i) v[i] = w[i]; } return v[0]; };
    int bad = 0;
    for (int l = 0; l < 64; ++l) { double e = tree(l / 32 * 32, 32); if (r2[l] != e) { if (bad < 8) printf("C2 lane %d got %.17g want %.17g\n", l, r2[l], e); ++bad; } }
    for (int l = 0; l < 64; ++l) { double e = tree(l / 16 * 16, 16); if (r4[l] != e) { if (bad < 16) printf("C4 lane %d got %.17g want %.17g\n", l, r4[l], e); ++bad; } }
    if (r1[0] != tree(0, 64)) { printf("wave64 mismatch\n"); ++bad; }
    printf("bad=%d\n", bad);
    return bad != 0;
}
