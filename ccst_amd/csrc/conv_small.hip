// 3x3 stride-1 convolution with a tiny output-channel count (<= 4) writing NCHW: the image edge of
// the decoder (net.py:35, Conv2d(64,3,3x3)).  0.9 GFLOP/image against ~70 MB of traffic => HBM-bound
// (13 FLOP/B), so padding Cout to an MFMA tile would waste 10x the math; this is a direct VALU kernel:
//   workgroup = 16x32 output pixels (two vertically adjacent pixels per thread, lanes along x => 128-B NCHW row
//   stores), the input halo (18x34 pixels: 1.2x the tile, against 1.33x + poorer L2 reuse for the 8x32 tile this
//   replaced -- the kernel is bound by what it fetches: 770 MB measured against 403 MB algorithmic before) is staged
//   through LDS 16 channels at a time (80-B pixel pitch: conflict-free ds_read_b128); a thread reads 4 halo rows x 3
//   columns for its two outputs; weights are wave-uniform and come through scalar loads.
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int TH = 16, TW = 32, HH = TH + 2, HW_ = TW + 2, CKS = 16, PITCH = CKS + 4;

__device__ __forceinline__ int reflect_c(int i, int n) {
    i = (i < 0) ? -i : i;
    i = (i >= n) ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

template <int CO>
__global__ __launch_bounds__(256) void conv3x3_smallco_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y, int N, int H,
                                                              int W, int Cin, int reflect, int relu, int tilesX, int tilesY) {
    __shared__ __attribute__((aligned(16))) float halo[HH * HW_ * PITCH];
    const int tid = threadIdx.x;
    const int tx = tid & 31, ty = (tid >> 5) * 2;             // this thread's outputs: rows ty, ty + 1 of the tile
    int b = blockIdx.x;
    const int bx = b % tilesX;
    b /= tilesX;
    const int by = b % tilesY;
    const int n = b / tilesY;
    const int ox = bx * TW + tx, oy = by * TH + ty;

    // even / odd input-channel partial sums (packed fp32 FMAs); .x carries the bias
    f32x2 acc[2][CO];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[r][c] = f32x2{bias ? bias[c] : 0.f, 0.f};

    // halo load units: 612 pixels x 4 float4
    constexpr int UNITS = HH * HW_ * (CKS / 4);
    constexpr int UPT = (UNITS + 255) / 256;
    unsigned uoff[UPT];
    bool uok[UPT];
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
        const int u = min(tid + 256 * i, UNITS - 1);
        const int pix = u >> 2, part = u & 3;
        const int hy = pix / HW_, hx = pix - hy * HW_;
        int gy = by * TH + hy - 1, gx = bx * TW + hx - 1;
        bool ok = true;
        if (reflect) {
            gy = reflect_c(gy, H);
            gx = reflect_c(gx, W);
        } else {
            ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
            gy = min(max(gy, 0), H - 1);
            gx = min(max(gx, 0), W - 1);
        }
        uok[i] = ok;
        uoff[i] = (unsigned)(((n * H + gy) * W + gx) * Cin + part * 4);
    }

    const float* hp = &halo[(ty * HW_ + tx) * PITCH];
    // the next chunk's halo is fetched into registers while this one is consumed from LDS
    f32x4 stage[UPT];
#pragma unroll
    for (int i = 0; i < UPT; ++i)
        if (tid + 256 * i < UNITS) stage[i] = *reinterpret_cast<const f32x4*>(x + uoff[i]);
    for (int c0 = 0; c0 < Cin; c0 += CKS) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int u = tid + 256 * i;
            if (u < UNITS) *reinterpret_cast<f32x4*>(&halo[(u >> 2) * PITCH + (u & 3) * 4]) = uok[i] ? stage[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        if (c0 + CKS < Cin) {
#pragma unroll
            for (int i = 0; i < UPT; ++i)
                if (tid + 256 * i < UNITS) stage[i] = *reinterpret_cast<const f32x4*>(x + uoff[i] + c0 + CKS);
        }
#pragma unroll
        for (int hr = 0; hr < 4; ++hr) {            // halo row ty + hr feeds output row 0 with tap ky = hr and row 1 with ky = hr - 1
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* xp = hp + (hr * HW_ + kx) * PITCH;
                const float* w0 = w + ((min(hr, 2) * 3 + kx) * CO * Cin + c0);          // uniform -> scalar loads
                const float* w1 = w + ((max(hr - 1, 0) * 3 + kx) * CO * Cin + c0);
#pragma unroll
                for (int q = 0; q < CKS / 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xp + q * 4);
                    const f32x2 va = {v[0], v[1]}, vb = {v[2], v[3]};
#pragma unroll
                    for (int c = 0; c < CO; ++c) {
                        if (hr <= 2) {
                            const f32x4 k = *reinterpret_cast<const f32x4*>(w0 + c * Cin + q * 4);
                            acc[0][c] = __builtin_elementwise_fma(va, f32x2{k[0], k[1]}, acc[0][c]);
                            acc[0][c] = __builtin_elementwise_fma(vb, f32x2{k[2], k[3]}, acc[0][c]);
                        }
                        if (hr >= 1) {
                            const f32x4 k = *reinterpret_cast<const f32x4*>(w1 + c * Cin + q * 4);
                            acc[1][c] = __builtin_elementwise_fma(va, f32x2{k[0], k[1]}, acc[1][c]);
                            acc[1][c] = __builtin_elementwise_fma(vb, f32x2{k[2], k[3]}, acc[1][c]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (ox < W && oy + r < H) {
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                float v = acc[r][c][0] + acc[r][c][1];
                if (relu) v = fmaxf(v, 0.f);
                y[(((long long)n * CO + c) * H + oy + r) * W + ox] = v;
            }
        }
    }
}

}  // namespace

// x: NHWC [N,H,W,Cin] (Cin % 16 == 0);  w: [3][3][Cout][Cin];  y: NCHW [N,Cout,H,W];  Cout in 1..4.
extern "C" int ccst_conv3x3_smallco_f32(const float* x, const float* w_tap_co_ci, const float* bias, float* y, int N, int H,
                                        int W, int Cin, int Cout, int reflect, int relu, void* stream) {
    CCST_REQUIRE(x && w_tap_co_ci && y, "conv3x3_smallco: null pointer");
    CCST_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0 && Cout >= 1 && Cout <= 4, "conv3x3_smallco: bad shape");
    CCST_REQUIRE((long long)N * H * W * Cin < 0x7fffffffLL, "conv3x3_smallco: input must have < 2^31 elements");
    if (reflect) CCST_REQUIRE(H >= 2 && W >= 2, "conv3x3_smallco: reflection needs extent >= 2");
    const int tilesX = (W + TW - 1) / TW, tilesY = (H + TH - 1) / TH;
    const long long grid = (long long)N * tilesX * tilesY;
    CCST_REQUIRE(grid < 0x7fffffffLL, "conv3x3_smallco: grid too large");
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(CO)                                                                                                           \
    hipLaunchKernelGGL((conv3x3_smallco_kernel<CO>), dim3((unsigned)grid), dim3(256), 0, s, x, w_tap_co_ci, bias, y, N, H, W, Cin, \
                       reflect, relu, tilesX, tilesY)
    switch (Cout) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        default: LAUNCH(4); break;
    }
#undef LAUNCH
    return ccst_launch_status("conv3x3_smallco");
}
