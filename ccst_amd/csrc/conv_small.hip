// 3x3 stride-1 convolution with a tiny output-channel count (<= 4) writing NCHW: the image edge of
// the decoder (net.py:35, Conv2d(64,3,3x3)).  0.9 GFLOP/image against ~70 MB of traffic => HBM-bound
// (13 FLOP/B), so padding Cout to an MFMA tile would waste 10x the math; this is a direct VALU kernel:
//   workgroup = 8x32 output pixels (one pixel per thread, lanes along x => 128-B NCHW row stores),
//   the input halo (10x34 pixels) is staged through LDS 16 channels at a time (80-B pixel pitch:
//   conflict-free ds_read_b128), weights are wave-uniform and come through scalar loads.
#include "common.h"

namespace {

constexpr int TH = 8, TW = 32, HH = TH + 2, HW_ = TW + 2, CKS = 16, PITCH = CKS + 4;

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
    const int tx = tid & 31, ty = tid >> 5;
    int b = blockIdx.x;
    const int bx = b % tilesX;
    b /= tilesX;
    const int by = b % tilesY;
    const int n = b / tilesY;
    const int ox = bx * TW + tx, oy = by * TH + ty;

    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = bias ? bias[c] : 0.f;

    // halo load units: 340 pixels x 4 float4
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
    for (int c0 = 0; c0 < Cin; c0 += CKS) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int u = tid + 256 * i;
            if (u < UNITS) {
                f32x4 v = *reinterpret_cast<const f32x4*>(x + uoff[i] + c0);
                if (!uok[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(&halo[(u >> 2) * PITCH + (u & 3) * 4]) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* wp = w + ((ky * 3 + kx) * Cin + c0) * CO;     // uniform -> scalar loads
                const float* xp = hp + (ky * HW_ + kx) * PITCH;
#pragma unroll
                for (int q = 0; q < CKS / 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xp + q * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int c = 0; c < CO; ++c) acc[c] = fmaf(v[j], wp[(q * 4 + j) * CO + c], acc[c]);
                }
            }
        }
    }
    if (ox < W && oy < H) {
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            float v = acc[c];
            if (relu) v = fmaxf(v, 0.f);
            y[(((long long)n * CO + c) * H + oy) * W + ox] = v;
        }
    }
}

}  // namespace

// x: NHWC [N,H,W,Cin] (Cin % 16 == 0);  w: [3][3][Cin][Cout];  y: NCHW [N,Cout,H,W];  Cout in 1..4.
extern "C" int ccst_conv3x3_smallco_f32(const float* x, const float* w_tap_ci_co, const float* bias, float* y, int N, int H,
                                        int W, int Cin, int Cout, int reflect, int relu, void* stream) {
    CCST_REQUIRE(x && w_tap_ci_co && y, "conv3x3_smallco: null pointer");
    CCST_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0 && Cout >= 1 && Cout <= 4, "conv3x3_smallco: bad shape");
    CCST_REQUIRE((long long)N * H * W * Cin < 0x7fffffffLL, "conv3x3_smallco: input must have < 2^31 elements");
    if (reflect) CCST_REQUIRE(H >= 2 && W >= 2, "conv3x3_smallco: reflection needs extent >= 2");
    const int tilesX = (W + TW - 1) / TW, tilesY = (H + TH - 1) / TH;
    const long long grid = (long long)N * tilesX * tilesY;
    CCST_REQUIRE(grid < 0x7fffffffLL, "conv3x3_smallco: grid too large");
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(CO)                                                                                                           \
    hipLaunchKernelGGL((conv3x3_smallco_kernel<CO>), dim3((unsigned)grid), dim3(256), 0, s, x, w_tap_ci_co, bias, y, N, H, W, Cin, \
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
