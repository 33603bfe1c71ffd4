// Shared helpers for libccst_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ccst_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void ccst_set_error(const char* fmt, ...);

#define CCST_REQUIRE(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            ccst_set_error(__VA_ARGS__);         \
            return CCST_EINVAL;                  \
        }                                        \
    } while (0)

static inline int ccst_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ccst_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return CCST_OK;
}

// Compute units of the current device (256 on MI355X; the same for every device of a node, so asked once).
static inline int ccst_num_cus() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu = v;
    }
    return n_cu;
}

// Bijective XCD-aware remap of a linear workgroup id: blocks b and b+8 share an XCD
// (observed round-robin dispatch), so give each XCD a contiguous chunk of the tile
// space.  Speed only, never correctness (cdna_hip_programming.md T1).
__device__ __forceinline__ int ccst_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- per-tensor |max| slots (CCST_ABSMAX_WORDS unsigned words, include/ccst_hip.h) ------------------------------------------------
// A producer kernel leaves the largest |value| it wrote as raw fp32 bits (monotone as unsigned for non-negative floats; a NaN sorts above
// infinity) with one atomic max per wave, spread over 64 slots 256 B apart so that no address sees more than 1/64 of the atomics; the
// consumer's waves read the 64 slots with one gather and reduce them.  The caller zeroes the words before the producer runs.
constexpr int CCST_ABSMAX_SLOTS = 64, CCST_ABSMAX_STRIDE = CCST_ABSMAX_WORDS / CCST_ABSMAX_SLOTS;

__device__ __forceinline__ unsigned ccst_wave_umax(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)v, o, 64);
        v = v > t ? v : t;
    }
    return v;
}
// every lane passes the largest |value| it holds (>= 0, or NaN); `salt` spreads the waves of the grid over the slots
__device__ __forceinline__ void ccst_absmax_publish(unsigned* slots, float lane_max, unsigned salt) {
    const unsigned m = ccst_wave_umax(__float_as_uint(lane_max) & 0x7fffffffu);
    if ((threadIdx.x & 63) == 0 && m != 0u) {
        unsigned* const s = slots + (salt % CCST_ABSMAX_SLOTS) * CCST_ABSMAX_STRIDE;
        // A slot only grows, so a wave whose maximum is not above what the slot holds NOW (an agent-scope load: it goes to the coherent
        // level, as the atomics do) has nothing to add: after the first waves of a launch almost none issues the atomic.  With one
        // unconditional atomic per wave a launch of 16 k waves put 256 serialised atomics on every slot -- +1.7 ms on a ResNet50 step
        // whose 53 BatchNorm applies publish their maxima (measured: 3063 against 3331 images/s).
        if (m > __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            (void)__hip_atomic_fetch_max(s, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// the tensor's |max| bits, wave-uniform (every wave reads for itself: no LDS, no barrier)
__device__ __forceinline__ unsigned ccst_absmax_read(const unsigned* slots) {
    const unsigned v = slots[(threadIdx.x & 63) * CCST_ABSMAX_STRIDE];
    return (unsigned)__builtin_amdgcn_readfirstlane((int)ccst_wave_umax(v));
}
// k such that 2^k * max lies in [2^target, 2^(target+1)) for a normal max; clamped to +-126 so that 2^k is a normal float (the clamp is
// reached only by max = 0 or subnormal: k = 126 scales such a tensor up as far as one multiplication can, without overflow; an infinite
// or NaN max gives k = target - 128 and the non-finite elements stay non-finite through the split)
__host__ __device__ __forceinline__ int ccst_scale_exp(unsigned absmax_bits, int target) {
    const int e = (int)((absmax_bits >> 23) & 0xffu) - 127;
    const int k = target - e;
    return k < -126 ? -126 : (k > 126 ? 126 : k);
}
// targets of the half-piece (SPLIT) kernels: activations below 2^14, weights below 2^10 -- hi pieces far inside half's range (65504),
// lo pieces normal for every element within 2^-17 (2^-13 for weights) of the largest
constexpr int CCST_SPLIT_X_TARGET = 13, CCST_SPLIT_W_TARGET = 9;
