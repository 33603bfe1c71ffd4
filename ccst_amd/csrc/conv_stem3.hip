// First layer of the AdaIN encoder: Conv2d(3,3,1x1) folded into ReflectionPad2d(1) + Conv2d(3,64,3x3) + ReLU (net.py:39-42), from the
// NCHW image straight to the NHWC feature map.  27 multiply-adds per output: padding that to a generic implicit-GEMM tile (K = 48 in
// three k-steps, a separate pad / interleave pass over the image) took 118 + 11 us for a layer whose floor is its 403 MB of output.
//
//   * one wave = 32 consecutive pixels of an image row x 64 output channels: v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand
//     (resident in 36 registers for the whole kernel) and the pixels' taps as B; K = 9 taps x 4 channels (the 4th is zero), so that
//     both k-halves of an MFMA read the same tap: a B operand is ONE 4-byte load at (scalar row offset) + (per-lane column index +
//     channel plane), reflection applied to the three row offsets and three column indices once per block;
//   * the bias rides in the one unused k slot (centre tap, channel 3: A = bias, B = 1); ReLU on the accumulators;
//   * a lane then holds 4 consecutive channels of ONE pixel per register quad (256 B between lanes): the block goes through 8 KB of
//     wave-private LDS (XOR-swizzled 16-byte slots, no barrier) so that the global stores are whole kilobytes of four pixels.
#include "common.h"

namespace {

struct Stem3Args {
    const float* x;      // [N,3,H,W]
    const float* wa;     // packed A operands [18][2][64]: (mfma j, channel group, lane)
    float* y;            // [N,H,W,64]
    int N, H, W, relu;
    int blocksPerRow;
    int nblocks;
    unsigned long long mBpr, mH;   // ceil(2^40 / blocksPerRow), ceil(2^40 / H): divisions by multiplication (launcher checks the ranges)
    unsigned* ymax;      // nullptr, or zeroed |max| words (CCST_ABSMAX_WORDS) receiving max |y|: the next layer's half-piece kernel scales by it
};

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int reflect_s(int i, int n) {
    i = (i < 0) ? -i : i;
    i = (i >= n) ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_stem3_kernel(const Stem3Args p) {
    __shared__ __attribute__((aligned(16))) char tr[4][32 * 256];             // per wave: 32 pixels x 64 channels
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    char* const my = tr[wave];

    // A operands: MFMA j (tap j >> 1, channel 2 (j & 1) + lh) x channel group nb, row = output channel nb * 32 + li
    float wa[18][2];
#pragma unroll
    for (int j = 0; j < 18; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) wa[j][nb] = p.wa[(j * 2 + nb) * 64 + lane];
    const long long HW = (long long)p.H * p.W;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, 0x7ffffffc, 0x00020000);
    // channel plane of this lane for the two MFMAs of a tap: channels lh and 2 + lh (channel 3 does not exist: its weight is 0, the
    // load re-reads channel 2)
    const unsigned cplane0 = (unsigned)(lh * HW) * 4u, cplane1 = (unsigned)(min(2 + lh, 2) * HW) * 4u;

    // the 18 B operands of a block: all requested together, one block ahead of their MFMAs
    auto fetch = [&](int blk, float (&bv)[18]) {
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
        for (int j = 0; j < 18; ++j) {
            const int t = j >> 1, dy = t / 3, dx = t - 3 * dy;
#ifdef ABLS_NO_LOAD
            bv[j] = __uint_as_float(colo[dx] + rowo[dy]);
#else
            bv[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, colo[dx] + ((j & 1) ? cplane1 : cplane0), rowo[dy], 0));
#endif
        }
    };
    const float floor_ = p.relu ? 0.f : -__builtin_inff();
    const int stride = (int)gridDim.x * 4;
    int blk = (int)blockIdx.x * 4 + wave;
    float bv[18], bvn[18];
    float amax = 0.f;         // largest |output| of this lane (lanes past the row's end hold copies of its last pixel)
    if (blk < p.nblocks) fetch(blk, bv);
    for (; blk < p.nblocks; blk += stride) {
        const bool more = blk + stride < p.nblocks;
        if (more) fetch(blk + stride, bvn);
        __builtin_amdgcn_sched_barrier(0);
        const int row = (int)(((unsigned long long)blk * p.mBpr) >> 40);
        const int bx = blk - row * p.blocksPerRow;
        // the bias rides in the unused k slot (centre tap, channel 3): its A operand holds the bias, its B operand is 1
        bv[9] = lh ? 1.f : bv[9];
        f32x16 acc[2] = {};
#pragma unroll
        for (int j = 0; j < 18; ++j)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
#ifdef ABLS_NO_MFMA
                acc[nb][j & 15] += wa[j][nb] * bv[j];
#else
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j][nb], bv[j], acc[nb], 0, 0, 0);
#endif
            }
        // ReLU, then through LDS: slot (pixel li, 16-byte quad q = nb * 8 + 2 g + lh) at li * 256 + ((q ^ (li & 15)) << 4)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o = {acc[nb][4 * g], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]};
#pragma unroll
                for (int i = 0; i < 4; ++i) asm("v_max_f32 %0, %1, %2" : "=v"(o[i]) : "v"(o[i]), "v"(floor_));   // (fmaxf would canonicalise first)
                amax = fmaxf(fmaxf(amax, fabsf(o[0])), fabsf(o[1]));        // (v_max3_f32 with |.| modifiers: 16 per block)
                amax = fmaxf(fmaxf(amax, fabsf(o[2])), fabsf(o[3]));
                *reinterpret_cast<f32x4*>(my + li * 256 + (((nb * 8 + 2 * g + lh) ^ (li & 15)) << 4)) = o;
            }
        // (same wave wrote and reads: no barrier; the compiler orders the LDS accesses by lgkmcnt)
        const unsigned ybase = (unsigned)(row * p.W + bx * 32) * 256u;      // < 2^31 checked by the launcher
        const int npix = min(32, p.W - bx * 32);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int pix = r * 4 + (lane >> 4), q = lane & 15;
            const f32x4 o = *reinterpret_cast<const f32x4*>(my + pix * 256 + ((q ^ (pix & 15)) << 4));
#ifdef ABLS_NO_STORE
            if (pix < npix && o[0] == 123.456f)
#else
            if (pix < npix)
#endif
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, o), yrs, ybase + (unsigned)(pix * 256 + q * 16), 0, 0);
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < 18; ++j) bv[j] = bvn[j];
        }
    }
    if (p.ymax != nullptr) ccst_absmax_publish(p.ymax, amax, blockIdx.x);
}

// OIHW [64,3,3,3] -> A operands [18 MFMAs][2 groups][64 lanes]: MFMA j = tap j >> 1, channel 2 (j & 1) + (lane >> 5) (zero for channel 3)
__global__ void pack_stem3_kernel(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ wa) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 18 * 2 * 64) return;
    const int lane = i & 63, nb = (i >> 6) & 1, j = i >> 7;
    const int co = nb * 32 + (lane & 31), c = 2 * (j & 1) + (lane >> 5), t = j >> 1;
    wa[i] = (c < 3) ? w[(co * 3 + c) * 9 + t] : (t == 4 && bias) ? bias[co] : 0.f;
}

}  // namespace

extern "C" int ccst_pack_stem3_weight_f32(const float* w_oihw, const float* bias, float* wa, int cout, void* stream) {
    CCST_REQUIRE(w_oihw && wa && cout == 64, "pack_stem3: a [64,3,3,3] weight");
    hipLaunchKernelGGL(pack_stem3_kernel, dim3(9), dim3(256), 0, (hipStream_t)stream, w_oihw, bias, wa);
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
    const long long wgs = (nblocks + 3) / 4;
    const long long resident = (long long)ccst_num_cus() * 3;                // three workgroups (12 waves) per CU: persistent
    const unsigned grid = (unsigned)(wgs < resident ? wgs : resident);
    hipLaunchKernelGGL(conv_stem3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return ccst_launch_status("conv3x3_stem3");
}
