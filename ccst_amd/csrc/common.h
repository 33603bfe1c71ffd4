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
