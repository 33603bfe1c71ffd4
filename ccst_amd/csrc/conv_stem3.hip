// First layer of the AdaIN encoder: Conv2d(3,3,1x1) folded into ReflectionPad2d(1) + Conv2d(3,64,3x3) + ReLU (net.py:39-42), from the
// NCHW image straight to the NHWC feature map.  27 multiply-adds per output: padding that to a generic implicit-GEMM tile (K = 48 in
// three k-steps, a separate pad / interleave pass over the image) took 118 + 11 us for a layer whose floor is its 403 MB of output.
//
//   * one wave = 32 consecutive pixels of an image row x 64 output channels, K = 27 padded to 32 = two k-steps of
//     v_mfma_f32_32x32x16_f16 with the WEIGHTS as the A operand (resident in 32 registers for the whole kernel) and the pixels' taps as
//     B, every fp32 product as three half-piece products (fp32 accumulate) -- round 4: 12 MFMAs of 8 passes per block where the fp32
//     MFMA (v_mfma_f32_32x32x2_f32, K = 36) took 36 of 16 passes: 50 us of MFMA time per launch, now 8, under a 403 MB write;
//   * k map: k-step 0 = taps 0..7 of channel lh (the lane half), k-step 1 = taps 0..7 of channel 2 (lh = 0) | tap 8 of channels 0..2
//     and five zeros (lh = 1): a B operand is ONE 4-byte load at (scalar row offset) + (per-lane column index + channel plane),
//     reflection applied to the three row offsets and three column indices once per block;
//   * scales: a lane is ONE pixel (column of B and of D), so every pixel is scaled by the power of two of ITS OWN largest tap (one
//     exchange between the lane halves) -- range-safe at any fp32 magnitude with no |max| pass over the image; the weights' scale is
//     found by the pack kernel and travels in the packed buffer's header;
//   * accumulators scaled back (v_ldexp), then a lane holds 4 consecutive channels of ONE pixel per register quad (256 B between
//     lanes): the block goes through 8 KB of wave-private LDS (XOR-swizzled 16-byte slots, no barrier) so that the global stores are
//     whole kilobytes of four pixels; bias, ReLU and max |y| on that side (a lane then holds the same four channels of eight pixels).
//   Measured (B=6, 512x512): 86-89 us against 115-118 for the fp32-MFMA form on the same box (a 403 MB fill: 61).  Timing experiments:
//   without the stores 50 us (63 % of the vector-issue slots busy, PMC: the splits, the scale and the transposition are what is left
//   of the arithmetic), without the MFMAs -7 us; four waves per SIMD need 128 registers and spill (158 us); with the weight fragments
//   in LDS instead (8 KB per eight-wave workgroup, re-read per block) they fit, and measure 92 us.
#include "common.h"

namespace {

struct Stem3Args {
    const float* x;      // [N,3,H,W]
    const float* wa;     // ccst_pack_stem3_weight_f32: [4] header (word 0 = the weights' scale exponent) | A fragments [2][2][2][64][4] | bias [64]
    float* y;            // [N,H,W,64]
    int N, H, W, relu;
    int blocksPerRow;
    int nblocks, blocksPerImage;
    unsigned long long mBpr, mH;   // ceil(2^40 / blocksPerRow), ceil(2^40 / H): divisions by multiplication (launcher checks the ranges)
    unsigned* ymax;      // nullptr, or zeroed |max| words [N][CCST_ABSMAX_WORDS] receiving max |y| PER IMAGE: the next layer's half-piece kernel scales by it
};

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2s __attribute__((ext_vector_type(2)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
constexpr int STEM_HDR = 4, STEM_FRAG = 2 * 2 * 2 * 64 * 4;      // floats

__device__ __forceinline__ int reflect_s(int i, int n) {
    i = (i < 0) ? -i : i;
    i = (i >= n) ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

// (channel, tap) of k index (lane half lh, element i) of k-step ks; channel 3 = a zero weight
__host__ __device__ __forceinline__ void stem3_k(int ks, int lh, int i, int& c, int& t) {
    if (ks == 0) {
        c = lh;
        t = i;
    } else if (lh == 0) {
        c = 2;
        t = i;
    } else {
        c = i < 3 ? i : 3;
        t = 8;
    }
}

#define STEM_WAVES 3
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(STEM_WAVES, STEM_WAVES))) void conv_stem3_kernel(const Stem3Args p) {
    __shared__ __attribute__((aligned(16))) char tr[4][32 * 256];             // per wave: 32 pixels x 64 channels
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    char* const my = tr[wave];

    // A operands [channel group nb][k-step][piece hi / lo]: row = output channel nb * 32 + li, k = 8 lh + i
    const int kw = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(p.wa));
    f16x8s wa[2][2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
                wa[nb][ks][pc] = __builtin_bit_cast(f16x8s, *reinterpret_cast<const f32x4*>(p.wa + STEM_HDR + (((nb * 2 + ks) * 2 + pc) * 64 + lane) * 4));
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.wa + STEM_HDR + STEM_FRAG + (lane & 15) * 4);      // the store side's four channels
    const unsigned HW4 = (unsigned)((long long)p.H * p.W) * 4u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, 0x7ffffffc, 0x00020000);

    // the 16 B operands of a block: all requested together, one block ahead of their MFMAs
    auto fetch = [&](int blk, float (&bv)[16]) {
        const int row = (int)(((unsigned long long)blk * p.mBpr) >> 40);       // (n, y)
        const int bx = blk - row * p.blocksPerRow;
        const int n = (int)(((unsigned long long)row * p.mH) >> 40), y = row - n * p.H;
        const int xc = min(bx * 32 + li, p.W - 1);
        unsigned rowo[3], colo[3];                                             // three row offsets (scalar), three column indices, reflected
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            rowo[d] = (unsigned)__builtin_amdgcn_readfirstlane((int)(((long long)n * 3 * p.H + reflect_s(y + d - 1, p.H)) * p.W * 4));
            colo[d] = (unsigned)reflect_s(xc + d - 1, p.W) * 4u;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)                                            // k-step 0: tap i of channel lh
            bv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, colo[i % 3] + (lh ? HW4 : 0u), rowo[i / 3], 0));
#pragma unroll
        for (int i = 0; i < 3; ++i) {                                          // k-step 1, i < 3: tap i of channel 2 | tap 8 of channel i
            const unsigned lhm = 0u - (unsigned)lh;                            // (bit select: a ?: here became a divergent branch around the load)
            const unsigned off = ((colo[2] + rowo[2] + (unsigned)i * HW4) & lhm) | ((colo[i] + rowo[0] + 2u * HW4) & ~lhm);
            bv[8 + i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, off, 0, 0));
        }
#pragma unroll
        for (int i = 3; i < 8; ++i)                                            // tap i of channel 2 (the other half's weights are zero)
            bv[8 + i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, colo[i % 3] + 2u * HW4, rowo[i / 3], 0));
    };
    const float floor_ = p.relu ? 0.f : -__builtin_inff();
    // grid = (workgroups per image, images): a workgroup walks the blocks of ONE image, so that its maximum is that image's (the |max|
    // words are per image) -- with nothing but arithmetic inside the pipelined loop (a publish behind a branch there put a vector-memory
    // operation on one path and with it a vmcnt(0) at the join: 86 -> 163 us)
    const int stride = (int)gridDim.x * 4;
    const int blk_end = ((int)blockIdx.y + 1) * p.blocksPerImage;
    int blk = (int)blockIdx.y * p.blocksPerImage + (int)blockIdx.x * 4 + wave;
    float amax = 0.f;         // largest |output| of this lane (lanes past the row's end hold copies of its last pixel)
    // one block from its 16 taps
    auto process = [&](int blk, const float (&bv)[16]) __attribute__((always_inline)) {
        const int row = (int)(((unsigned long long)blk * p.mBpr) >> 40);
        const int bx = blk - row * p.blocksPerRow;
        // this pixel's scale: the power of two that puts its largest tap below 2^14 (both lane halves hold taps of the same pixel)
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 2) m = fmaxf(fmaxf(m, fabsf(bv[i])), fabsf(bv[i + 1]));
        unsigned mb = __float_as_uint(m);
        {
            const unsigned o = (unsigned)__shfl_xor((int)mb, 32, 64);
            mb = mb > o ? mb : o;
        }
        const int kx = mb == 0u ? 0 : ccst_scale_exp(mb, CCST_SPLIT_X_TARGET);      // (an all-zero neighbourhood -- black borders -- needs no scale, and should not send its block down the v_ldexp path below)
        const float xs = __uint_as_float((unsigned)(127 + kx) << 23);
        f16x8s bhi[2], blo[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const f32x2s v = f32x2s{bv[8 * ks + 2 * h], bv[8 * ks + 2 * h + 1]} * xs;
                unsigned wh, wl;
                ccst_split2_half(v[0], v[1], wh, wl);
                const f16x2s ph = __builtin_bit_cast(f16x2s, wh), pl = __builtin_bit_cast(f16x2s, wl);
                bhi[ks][2 * h] = ph[0];
                bhi[ks][2 * h + 1] = ph[1];
                blo[ks][2 * h] = pl[0];
                blo[ks][2 * h + 1] = pl[1];
            }
        f32x16 acc[2] = {};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[nb][ks][1], bhi[ks], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[nb][ks][0], blo[ks], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[nb][ks][0], bhi[ks], acc[nb], 0, 0, 0);
            }
        // scaled back, then through LDS: slot (pixel li, 16-byte quad q = nb * 8 + 2 g + lh) at li * 256 + ((q ^ (li & 15)) << 4)
        const int kd = -(kx + kw);
        // (one exact multiplication by 2^kd where that is a normal float -- two values per v_pk_mul_f32 -- v_ldexp_f32 for the blocks
        //  with a pixel whose scale is beyond: |taps| outside ~[1e-34, 1e34])
        const int kdc = min(max(kd, -126), 126);
        const float sc = __uint_as_float((unsigned)(127 + kdc) << 23);
        if (__builtin_amdgcn_ballot_w64(kd != kdc) == 0) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 o = f32x4{acc[nb][4 * g], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]} * sc;
                    *reinterpret_cast<f32x4*>(my + li * 256 + (((nb * 8 + 2 * g + lh) ^ (li & 15)) << 4)) = o;
                }
        } else {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 o = {__builtin_ldexpf(acc[nb][4 * g], kd), __builtin_ldexpf(acc[nb][4 * g + 1], kd), __builtin_ldexpf(acc[nb][4 * g + 2], kd),
                                     __builtin_ldexpf(acc[nb][4 * g + 3], kd)};
                    *reinterpret_cast<f32x4*>(my + li * 256 + (((nb * 8 + 2 * g + lh) ^ (li & 15)) << 4)) = o;
                }
        }
        // (same wave wrote and reads: no barrier; the compiler orders the LDS accesses by lgkmcnt)
        const unsigned ybase = (unsigned)(row * p.W + bx * 32) * 256u;      // < 2^31 checked by the launcher
        const int npix = min(32, p.W - bx * 32);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            // the lanes past the row's end hold copies of its last pixel: they store those to the last pixel's address -- no predicate
            const int pix = r * 4 + (lane >> 4), q = lane & 15, pixs = min(pix, npix - 1);
            f32x4 o = *reinterpret_cast<const f32x4*>(my + pix * 256 + ((q ^ (pix & 15)) << 4)) + bias4;
#pragma unroll
            for (int i = 0; i < 4; ++i) asm("v_max_f32 %0, %1, %2" : "=v"(o[i]) : "v"(o[i]), "v"(floor_));   // (fmaxf would canonicalise first)
            amax = fmaxf(fmaxf(amax, fabsf(o[0])), fabsf(o[1]));
            amax = fmaxf(fmaxf(amax, fabsf(o[2])), fabsf(o[3]));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, o), yrs, ybase + (unsigned)(pixs * 256 + q * 16), 0, 2);      // (non-temporal: 403 MB that the L2 cannot hold; -7 us here, -8 us in conv1_2)
        }
    };
    // Two register sets, no copies: the taps of block i + 1 are requested before block i is computed.  (With one set refilled through
    // a second one and copied at the end of the iteration, hipcc waited for the loads it had just issued -- vmcnt(0) at the top of
    // every iteration: the whole memory latency once per block and wave, 118 -> 99 us was all the 16-bit MFMA bought until this.)
    float bva[16], bvb[16];
    // The prefetch is UNCONDITIONAL (past the end it re-reads the last block): behind a branch, the wait in front of the current
    // block's taps has to be right for the path that skipped the fetch too, and becomes vmcnt(0).
    const int last = blk_end - 1;
    if (blk < blk_end) fetch(blk, bva);
    while (blk < blk_end) {
        fetch(min(blk + stride, last), bvb);
        __builtin_amdgcn_sched_barrier(0);
        process(blk, bva);
        blk += stride;
        if (blk >= blk_end) break;
        fetch(min(blk + stride, last), bva);
        __builtin_amdgcn_sched_barrier(0);
        process(blk, bvb);
        blk += stride;
    }
    if (p.ymax != nullptr) ccst_absmax_publish(p.ymax + blockIdx.y * CCST_ABSMAX_WORDS, amax, blockIdx.x);
}

// OIHW [64,3,3,3] (+ bias [64]) -> header | A fragments | bias (Stem3Args::wa).  One workgroup: the weights' largest |value| gives their
// power-of-two scale.
__global__ __launch_bounds__(256) void pack_stem3_kernel(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ wa) {
    __shared__ unsigned wmaxs[4];
    const int tid = threadIdx.x;
    float m = 0.f;
    for (int i = tid; i < 64 * 27; i += 256) m = fmaxf(m, fabsf(w[i]));
    const unsigned mw = ccst_wave_umax(__float_as_uint(m));
    if ((tid & 63) == 0) wmaxs[tid >> 6] = mw;
    __syncthreads();
    const unsigned mb = max(max(wmaxs[0], wmaxs[1]), max(wmaxs[2], wmaxs[3]));
    const int kw = ccst_scale_exp(mb, CCST_SPLIT_W_TARGET);
    const float s = __uint_as_float((unsigned)(127 + kw) << 23);
    if (tid < STEM_HDR) reinterpret_cast<int*>(wa)[tid] = tid == 0 ? kw : 0;
    if (tid < 64) wa[STEM_HDR + STEM_FRAG + tid] = bias != nullptr ? bias[tid] : 0.f;
    for (int e = tid; e < 2 * 2 * 64; e += 256) {          // (channel group, k-step, lane): both pieces
        const int lane = e & 63, ks = (e >> 6) & 1, nb = e >> 7;
        const int co = nb * 32 + (lane & 31), lh = lane >> 5;
        f16x8s hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int c, t;
            stem3_k(ks, lh, i, c, t);
            const float v = c < 3 ? w[(co * 3 + c) * 9 + t] * s : 0.f;
            const _Float16 h = (_Float16)v;
            hi[i] = h;
            lo[i] = (_Float16)(v - (float)h);
        }
        f32x4* o = reinterpret_cast<f32x4*>(wa + STEM_HDR);
        o[((nb * 2 + ks) * 2 + 0) * 64 + lane] = __builtin_bit_cast(f32x4, hi);
        o[((nb * 2 + ks) * 2 + 1) * 64 + lane] = __builtin_bit_cast(f32x4, lo);
    }
}

}  // namespace

extern "C" int ccst_pack_stem3_weight_f32(const float* w_oihw, const float* bias, float* wa, int cout, void* stream) {
    CCST_REQUIRE(w_oihw && wa && cout == 64, "pack_stem3: a [64,3,3,3] weight");
    hipLaunchKernelGGL(pack_stem3_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w_oihw, bias, wa);
    return ccst_launch_status("pack_stem3");
}

// x NCHW [N,3,H,W] (contiguous), wa from ccst_pack_stem3_weight_f32 (18*2*64 floats), y NHWC [N,H,W,64]: reflection-padded 3x3 conv
// + bias (+ ReLU) -- net.py:39-42 with the 1x1 colour conv folded into the weight.
extern "C" int ccst_conv3x3_stem3_f32(const float* x_nchw, const float* wa, float* y_nhwc, int N, int H, int W, int relu, uint32_t* y_absmax,
                                      void* stream) {
    CCST_REQUIRE(x_nchw && wa && y_nhwc && N > 0 && H >= 2 && W >= 2, "conv3x3_stem3: bad args");
    CCST_REQUIRE((long long)N * 3 * H * W * 4 < 0x7fffffffLL && (long long)N * H * W * 64 * 4 < 0x7fffffffLL,
                 "conv3x3_stem3: tensors must be < 2^31 bytes");
    Stem3Args a;
    a.x = x_nchw; a.wa = wa; a.y = y_nhwc;
    a.N = N; a.H = H; a.W = W; a.relu = relu; a.ymax = y_absmax;
    a.blocksPerRow = (W + 31) / 32;
    const long long nblocks = (long long)N * H * a.blocksPerRow;
    CCST_REQUIRE(nblocks * a.blocksPerRow < (1LL << 40) && (long long)N * H * H < (1LL << 40), "conv3x3_stem3: too many rows");
    a.nblocks = (int)nblocks;
    a.mBpr = ((1ULL << 40) + a.blocksPerRow - 1) / a.blocksPerRow;
    a.mH = ((1ULL << 40) + H - 1) / H;
    a.blocksPerImage = H * a.blocksPerRow;
    CCST_REQUIRE(N <= 65535, "conv3x3_stem3: N must be <= 65535");
    const long long wgs = (a.blocksPerImage + 3) / 4;                                 // per image
    const long long resident = ((long long)ccst_num_cus() * STEM_WAVES + N - 1) / N;  // three workgroups (12 waves) per CU: persistent
    const unsigned gx = (unsigned)(wgs < resident ? wgs : (resident < 1 ? 1 : resident));
    hipLaunchKernelGGL(conv_stem3_kernel, dim3(gx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, a);
    return ccst_launch_status("conv3x3_stem3");
}
