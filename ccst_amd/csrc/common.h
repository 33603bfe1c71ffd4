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

// ---- per-tensor |max| words (CCST_ABSMAX_WORDS = 64 unsigned words = 256 bytes, include/ccst_hip.h) -------------------------------------
// A producer kernel leaves the largest |value| it wrote as raw fp32 bits (monotone as unsigned for non-negative floats; a NaN sorts above
// infinity): ONE conditional atomic max per WORKGROUP into word (workgroup id) % 64.  The consumer's waves read all 64 words with one
// coalesced 256-byte load and reduce them.  The caller zeroes the words before the producer runs.
//   Why this shape (round 4, measured on the AdaIN step with the operands kept real by a fixed scale: the first form -- 64 slots 256 B
//   apart, one unconditional-then-conditional atomic per WAVE, a dependent load at every wave's end, a 64-line gather at every wave's
//   start -- cost 6.5 % of the step, ~12 us per conv launch): the words are contiguous so that reading them is one request per wave;
//   one atomic per workgroup keeps the burst of a launch's first round (every workgroup sees zeros) at 4 per word; the word's current
//   value is PEEKED before the epilogue's stores and only compared after them, so that nobody waits for that load; the atomic itself
//   is fire-and-forget.  A word only grows, so a stale peek can only cause a superfluous atomic, never a missed one.
constexpr int CCST_ABSMAX_SLOTS = CCST_ABSMAX_WORDS;

// The largest value of the wave, in every lane (wave-uniform).  Six v_max_u32_dpp + a v_readlane: within a row of 16 lanes by row_shr
// 1, 2, 4 (lanes 4-15), 8 (lanes 8-15), across rows by row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 -- lane 63
// holds the maximum.  (The __shfl_xor form was six dependent ds_bpermute round trips through the LDS crossbar, ~0.3 us at the very end
// of every workgroup that publishes |max| words and at the start of every kernel that reads them.)
__device__ __forceinline__ unsigned ccst_wave_umax(unsigned v) {
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); v = v > t ? v : t;   // row_shr:1
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true); v = v > t ? v : t;   // row_shr:2
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, true); v = v > t ? v : t;   // row_shr:4, lanes 4-15 of a row
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, true); v = v > t ? v : t;   // row_shr:8, lanes 8-15
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true); v = v > t ? v : t;   // row_bcast:15 into rows 1, 3
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true); v = v > t ? v : t;   // row_bcast:31 into rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
constexpr unsigned CCST_NOT_PEEKED = 0xffffffffu;
// the current value of this workgroup's word (meaningful in thread 0 only); block = the workgroup's linear id
__device__ __forceinline__ unsigned ccst_absmax_peek(const unsigned* slots, unsigned block) {
    return threadIdx.x == 0 ? __hip_atomic_load(slots + block % CCST_ABSMAX_SLOTS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
}
// every lane passes the largest |value| it holds (>= 0, or NaN).  Called by ALL threads of the workgroup (it contains a barrier -- a raw
// s_barrier behind lgkmcnt(0) only: __syncthreads() would also wait for the epilogue's stores).
__device__ __forceinline__ void ccst_absmax_publish(unsigned* slots, float lane_max, unsigned block, unsigned peeked = CCST_NOT_PEEKED) {
    __shared__ unsigned ccst_amax_wave_[16];
    const unsigned m = ccst_wave_umax(__float_as_uint(lane_max) & 0x7fffffffu);
    if ((threadIdx.x & 63) == 0) ccst_amax_wave_[threadIdx.x >> 6] = m;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (threadIdx.x == 0) {
        unsigned mm = 0u;
        for (unsigned w = 0; w < (blockDim.x + 63u) / 64u; ++w) mm = mm > ccst_amax_wave_[w] ? mm : ccst_amax_wave_[w];
        unsigned* const s = slots + block % CCST_ABSMAX_SLOTS;
        if (mm != 0u) {
            const unsigned cur = peeked != CCST_NOT_PEEKED ? peeked : __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mm > cur) (void)__hip_atomic_fetch_max(s, mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// the two halves of ccst_absmax_read for kernels that want the load in flight behind other work: every lane loads its word early
// (ccst_absmax_load), the wave reduces it where the value is first needed (ccst_absmax_reduce)
__device__ __forceinline__ unsigned ccst_absmax_load(const unsigned* slots) {
    return slots[threadIdx.x & 63];
}
__device__ __forceinline__ unsigned ccst_absmax_reduce(unsigned lane_word) {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)ccst_wave_umax(lane_word));
}
// the tensor's |max| bits, wave-uniform (every wave reads for itself: one 256-byte load, no LDS, no barrier)
__device__ __forceinline__ unsigned ccst_absmax_read(const unsigned* slots) {
    const unsigned v = slots[threadIdx.x & 63];
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

// Four fp32 values scaled by s = 2^k -> two IEEE-half pieces each (hi = half(v s), lo = half(v s - hi): 22 significant bits), two per
// word.  8 vector instructions per four values: v_pk_mul_f32 + v_cvt_pk_f16_f32 per pair for hi, and ONE v_fma_mix{lo,hi}_f16 per value
// for lo -- fma(v, s, -hi) with hi read as the half it is, rounded to half.  (hipcc's own rendering of the same arithmetic re-derives hi
// per value, 14 instructions; the convert-back-and-subtract form takes 12.  Vector instructions are paid on top of the MFMA time.)
typedef unsigned ccst_u32x2 __attribute__((ext_vector_type(2)));
// ... of a pair that is already scaled: (hi word, lo word) = 1 v_cvt_pk_f16_f32 + 2 v_fma_mix (lo = half(q - hi), q * 1.0 - hi in one rounding)
__device__ __forceinline__ void ccst_split2_half(float q0, float q1, unsigned& hi, unsigned& lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const unsigned qh = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{q0, q1}, h2));
    unsigned ql;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(ql) : "v"(q0), "v"(qh));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(ql) : "v"(q1), "v"(qh));
    hi = qh;
    lo = ql;
}
__device__ __forceinline__ void ccst_split4_half(f32x4 v, float s, ccst_u32x2& hi, ccst_u32x2& lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f2 q = f2{v[2 * h], v[2 * h + 1]} * s;
        const unsigned qh = __builtin_bit_cast(unsigned, __builtin_convertvector(q, h2));
        unsigned ql;
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ql) : "v"(v[2 * h]), "v"(s), "v"(qh));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(ql) : "v"(v[2 * h + 1]), "v"(s), "v"(qh));
        hi[h] = qh;
        lo[h] = ql;
    }
}
