// S(omega) histogram accumulation after a sweep (free:918), one wavefront per chain.
//
// Reference semantics (SAC_HIST_REFERENCE): `sampled_spec[idx] .+= A` gathers the right-hand
// side first and then scatters, so deltas that share a frequency do NOT accumulate -- the bin
// receives old + A of the LAST delta at that frequency.  To reproduce that without per-chain
// scratch memory the wave sorts its (position, index) pairs in registers (bitonic network,
// 64 lanes x EPL keys) and only the last element of every run of equal positions adds.
#pragma once
#include "wave_ops.h"

namespace sacdev {

template <typename KeyT> __device__ __forceinline__ KeyT kmin(KeyT a, KeyT b) { return a < b ? a : b; }
template <typename KeyT> __device__ __forceinline__ KeyT kmax(KeyT a, KeyT b) { return a < b ? b : a; }

template <typename KeyT>
__device__ __forceinline__ KeyT shfl_xor_key(KeyT v, int m)
{
    if constexpr (sizeof(KeyT) == 4) {
        return (KeyT)__shfl_xor((int)v, m, 64);
    } else {
        const int lo = __shfl_xor((int)(uint32_t)v, m, 64);
        const int hi = __shfl_xor((int)(uint32_t)(v >> 32), m, 64);
        return ((KeyT)(uint32_t)hi << 32) | (uint32_t)lo;
    }
}

// Ascending bitonic sort of 64*EPL keys; element e lives in lane e / EPL, register e % EPL.
template <typename KeyT, int EPL>
__device__ __forceinline__ void wave_bitonic_sort(KeyT (&key)[EPL], int lane)
{
    // phase A: sort inside each lane (k = 2 .. EPL); direction alternates with the register index
    // for k < EPL and with the lane's parity for k == EPL.
#pragma unroll
    for (int k = 2; k <= EPL; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int r = 0; r < EPL; ++r) {
                const int q = r ^ j;
                if (q > r) {
                    const KeyT lo = kmin(key[r], key[q]), hi = kmax(key[r], key[q]);
                    bool asc;
                    if (k < EPL) asc = ((r & k) == 0);
                    else asc = ((lane & 1) == 0);
                    key[r] = asc ? lo : hi;
                    key[q] = asc ? hi : lo;
                }
            }
        }
    }
    // phase B: merge across lanes (k = 2*EPL .. 64*EPL)
    bool flipped = false;
#pragma unroll 1
    for (int kl = 2; kl <= 64; kl <<= 1) {          // kl = k / EPL
        // lanes whose block is descending work on complemented keys, so every exchange is ascending
        const bool want_flip = (kl < 64) && ((lane & kl) != 0);
        if (want_flip != flipped) {
#pragma unroll
            for (int r = 0; r < EPL; ++r) key[r] = ~key[r];
            flipped = want_flip;
        }
#pragma unroll 1
        for (int m = kl >> 1; m > 0; m >>= 1) {     // partner lane = lane ^ m
            const bool lower = ((lane & m) == 0);
#pragma unroll
            for (int r = 0; r < EPL; ++r) {
                const KeyT other = shfl_xor_key<KeyT>(key[r], m);
                key[r] = lower ? kmin(key[r], other) : kmax(key[r], other);
            }
        }
#pragma unroll
        for (int j = EPL >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int r = 0; r < EPL; ++r) {
                const int q = r ^ j;
                if (q > r) {
                    const KeyT lo = kmin(key[r], key[q]), hi = kmax(key[r], key[q]);
                    key[r] = lo;
                    key[q] = hi;
                }
            }
        }
    }
    // the last phase (kl == 64) is ascending for every lane, so nothing is left complemented
}

// Adds the chain's deltas to one histogram replica.
//   wi[n]      relative positions (LDS)
//   A / A_uniform amplitudes
//   exclusive  the replica belongs to this chain alone: plain read-modify-write (bit exact);
//              otherwise fp64 hardware atomics.
// TAG only separates instantiations: an out-of-line function inherits the register budget of the kernels that call
// it, so kernels with different occupancy targets must not share one copy.
template <typename KeyT, int EPL, int TAG = 0>
__device__ __attribute__((noinline)) void hist_reference(const uint32_t *wi, int n, const double *A,
                                               double A_uniform, int uniform_A, double *hist,
                                               int exclusive, int lane)
{
    constexpr int KBITS = 11;                        // n <= 2048
    const KeyT PAD = ~(KeyT)0;
    KeyT key[EPL];
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const int i = r * 64 + lane;                 // conflict-free LDS read; any start order sorts
        key[r] = (i < n) ? (((KeyT)wi[i] << KBITS) | (KeyT)i) : PAD;
    }
    wave_bitonic_sort<KeyT, EPL>(key, lane);
    // first key of the next lane, to close this lane's last run
    KeyT nxt = shfl_xor_key<KeyT>(key[0], 0);        // placeholder, replaced below
    {
        if constexpr (sizeof(KeyT) == 4) {
            nxt = (KeyT)__shfl_down((int)key[0], 1, 64);
        } else {
            const int lo = __shfl_down((int)(uint32_t)key[0], 1, 64);
            const int hi = __shfl_down((int)(uint32_t)(key[0] >> 32), 1, 64);
            nxt = ((KeyT)(uint32_t)hi << 32) | (uint32_t)lo;
        }
        if (lane == 63) nxt = PAD;
    }
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const KeyT me = key[r];
        const KeyT nx = (r + 1 < EPL) ? key[r + 1 < EPL ? r + 1 : r] : nxt;
        const bool valid = (me != PAD);
        const bool last_of_run = (nx == PAD) || ((nx >> KBITS) != (me >> KBITS));
        if (valid && last_of_run) {
            const uint32_t pos = (uint32_t)(me >> KBITS);
            const int k = (int)(me & ((1u << KBITS) - 1u));
            const double a = uniform_A ? A_uniform : A[k];
            if (exclusive) hist[pos] = hist[pos] + a;
            else unsafeAtomicAdd(&hist[pos], a);
        }
    }
}

// SAC_HIST_ACCUMULATE: every delta adds (the semantics of peak:1126-1140 and edge:1322-1331).
__device__ __forceinline__ void hist_accumulate(const uint32_t *wi, int n, const double *A,
                                                double A_uniform, int uniform_A, double *hist, int lane)
{
    for (int i = lane; i < n; i += 64) {
        const double a = uniform_A ? A_uniform : A[i];
        unsafeAtomicAdd(&hist[wi[i]], a);
    }
}

}  // namespace sacdev
