// Stand-alone NHWC layer ops for the AdaIN networks (net.py:6-69) when a layer is NOT fused into a
// convolution (arbitrary slices of the nn.Sequential), plus NCHW<->NHWC transposes for API tensors.
// All HBM-bound; 16-B accesses along C.
#include "common.h"

namespace {

__device__ __forceinline__ int reflect1(int i, int n) {
    i = (i < 0) ? -i : i;
    return (i >= n) ? 2 * n - 2 - i : i;
}

// x[N,C,HW] -> y[N,HW,Cp] (Cp >= C, zero filled) through a 32x32 LDS tile.
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW, int Cp) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? x[((long long)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (p < HW && c < Cp) y[((long long)n * HW + p) * Cp + c] = tile[tx][r];
    }
}

// x[N,HW,Cs] (first C channels used) -> y[N,C,HW]
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW, int Cs) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (c < C && p < HW) ? x[((long long)n * HW + p) * Cs + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) y[((long long)n * C + c) * HW + p] = tile[tx][r];
    }
}

// mode 0: ReLU (same shape)            mode 1: ReflectionPad2d(pad)   [H,W] -> [H+2p, W+2p]
// mode 2: Upsample nearest x2          mode 3: MaxPool2d(2,2,ceil_mode=True)
template <int MODE>
__global__ void nhwc_map_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int N, int H, int W, int Ho, int Wo, int C4,
                                int pad) {
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long j = i / C4;
        const int ox = (int)(j % Wo);
        j /= Wo;
        const int oy = (int)(j % Ho);
        const int n = (int)(j / Ho);
        const f32x4* xb = x + (long long)n * H * W * C4 + c;
        f32x4 v;
        if (MODE == 0) {
            v = xb[((long long)oy * W + ox) * C4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        } else if (MODE == 1) {
            v = xb[((long long)reflect1(oy - pad, H) * W + reflect1(ox - pad, W)) * C4];
        } else if (MODE == 2) {
            v = xb[((long long)(oy >> 1) * W + (ox >> 1)) * C4];
        } else {
            const int y0 = 2 * oy, x0 = 2 * ox;
            v = xb[((long long)y0 * W + x0) * C4];
            if (x0 + 1 < W) {
                const f32x4 u = xb[((long long)y0 * W + x0 + 1) * C4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], u[k]);
            }
            if (y0 + 1 < H) {
                f32x4 u = xb[((long long)(y0 + 1) * W + x0) * C4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], u[k]);
                if (x0 + 1 < W) {
                    u = xb[((long long)(y0 + 1) * W + x0 + 1) * C4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], u[k]);
                }
            }
        }
        y[i] = v;
    }
}

// torchvision.utils.save_image's quantisation (x*255+0.5, clamp to [0,255], uint8, CHW->HWC) in one pass,
// so only a quarter of the bytes cross PCIe (CCST_OverallStyleTransfer.py:156,167).
__global__ void quantize_u8_hwc_kernel(const float* __restrict__ x, unsigned char* __restrict__ y, int C, int HW, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW;
        const int p = (int)(i - n * HW);
        for (int c = 0; c < C; ++c) {
            float v = __fadd_rn(__fmul_rn(x[(n * C + c) * HW + p], 255.f), 0.5f);     // mul(255).add_(0.5): two roundings, never an FMA
            v = fminf(fmaxf(v, 0.f), 255.f);
            y[i * C + c] = (unsigned char)v;
        }
    }
}

// transforms.Resize(size) applied to the stylised TENSOR before save_image (CCST_OverallStyleTransfer.py:154-157): on tensors
// torchvision (0.8 .. 0.16) calls torch.nn.functional.interpolate(mode='bilinear', align_corners=False) without antialiasing.
// Source index = scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out (ATen area_pixel_compute_source_index).
__global__ void resize_bilinear_planes_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int oh, int ow, float sy,
                                              float sx, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % ow);
        const long long t = i / ow;
        const int oy = (int)(t % oh);
        const long long plane = t / oh;
        float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* p = x + plane * (long long)H * W;
        const float a = p[(long long)y0 * W + x0], b = p[(long long)y0 * W + x1], c = p[(long long)y1 * W + x0], d = p[(long long)y1 * W + x1];
        y[i] = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * c + lx * d);
    }
}

int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// Largest |value| of a tensor as raw fp32 bits, max-accumulated into the caller's zeroed CCST_ABSMAX_WORDS words (common.h): the
// stand-alone producer of the operand-scale words the half-piece conv kernels read, for tensors whose producing kernel did not leave them
// (checkpoint weights at pack time, a feature map handed in from outside the plan).  One streaming read, 16 bytes per lane.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ slots) {
    float m = 0.f;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v[0])), fmaxf(fabsf(v[1]), fabsf(v[2]))), fabsf(v[3]));
        // (fmaxf drops a NaN operand: keep it visible -- a NaN sorts above infinity as unsigned bits and clamps the consumer's scale)
        if (v[0] != v[0] || v[1] != v[1] || v[2] != v[2] || v[3] != v[3]) m = __builtin_nanf("");
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
        const float v = x[(n4 << 2) + threadIdx.x];
        m = (v != v) ? v : fmaxf(m, fabsf(v));
    }
    ccst_absmax_publish(slots, m, blockIdx.x);
}

// The |max| words of n tensors in ONE launch (the pointwise conv weights of a ResNet after an optimiser step): table [n][2] int64 =
// (device pointer, element count), words [n][CCST_ABSMAX_WORDS] zeroed by the caller; blockIdx.y = tensor.
__global__ __launch_bounds__(256) void absmax_batch_kernel(const long long* __restrict__ table, unsigned* __restrict__ words) {
    const float* x = reinterpret_cast<const float*>(table[2 * blockIdx.y]);
    const long long n = table[2 * blockIdx.y + 1], n4 = n >> 2;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v[0])), fmaxf(fabsf(v[1]), fabsf(v[2]))), fabsf(v[3]));
        if (v[0] != v[0] || v[1] != v[1] || v[2] != v[2] || v[3] != v[3]) m = __builtin_nanf("");
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
        const float v = x[(n4 << 2) + threadIdx.x];
        m = (v != v) ? v : fmaxf(m, fabsf(v));
    }
    ccst_absmax_publish(words + (long long)blockIdx.y * CCST_ABSMAX_WORDS, m, blockIdx.x);
}

// The |max| words of the N images of a batch in one launch: x [N][per] contiguous, words [N][CCST_ABSMAX_WORDS] zeroed by the caller;
// blockIdx.y = image.  (The AdaIN-path kernels keep one set of words PER IMAGE: a sample's power-of-two scale -- and with it its bits --
// must not depend on its batch-mates, function.py:4-13.)  16-byte loads when an image's elements are a multiple of four.
__global__ __launch_bounds__(256) void absmax_samples_kernel(const float* __restrict__ x, long long per, unsigned* __restrict__ words) {
    const float* xb = x + (long long)blockIdx.y * per;
    float m = 0.f;
    if ((per & 3) == 0) {
        const long long n4 = per >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xb + i * 4);
            m = fmaxf(fmaxf(fmaxf(m, fabsf(v[0])), fmaxf(fabsf(v[1]), fabsf(v[2]))), fabsf(v[3]));
            if (v[0] != v[0] || v[1] != v[1] || v[2] != v[2] || v[3] != v[3]) m = __builtin_nanf("");
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long long)gridDim.x * 256) {
            const float v = xb[i];
            m = (v != v) ? v : fmaxf(m, fabsf(v));
        }
    }
    ccst_absmax_publish(words + (long long)blockIdx.y * CCST_ABSMAX_WORDS, m, blockIdx.x);
}

}  // namespace

extern "C" int ccst_absmax_f32(const float* x, int64_t n, uint32_t* absmax, void* stream) {
    CCST_REQUIRE(x && absmax && n > 0, "absmax: bad args");
    CCST_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "absmax: x must be 16-byte aligned");
    const long long blocks = (n / 4 + 255) / 256;
    const long long cap = (long long)ccst_num_cus() * 8;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks < cap ? blocks : cap))), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)n, absmax);
    return ccst_launch_status("absmax");
}

extern "C" int ccst_absmax_samples_f32(const float* x, int N, int64_t per_sample, uint32_t* absmax, void* stream) {
    CCST_REQUIRE(x && absmax && N > 0 && N <= 65535 && per_sample > 0, "absmax_samples: bad args");
    CCST_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "absmax_samples: x must be 16-byte aligned");
    const long long blocks = (per_sample / 4 + 255) / 256;
    const long long cap = ((long long)ccst_num_cus() * 8 + N - 1) / N;
    const long long gx = blocks < 1 ? 1 : (blocks < cap ? blocks : (cap < 1 ? 1 : cap));
    hipLaunchKernelGGL(absmax_samples_kernel, dim3((unsigned)gx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, x, (long long)per_sample, absmax);
    return ccst_launch_status("absmax_samples");
}

extern "C" int ccst_absmax_batch_f32(const int64_t* table, int n, uint32_t* absmax, void* stream) {
    CCST_REQUIRE(table && absmax && n > 0 && n <= 65535, "absmax_batch: bad args");
    hipLaunchKernelGGL(absmax_batch_kernel, dim3(16, (unsigned)n), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const long long*>(table), absmax);
    return ccst_launch_status("absmax_batch");
}


extern "C" int ccst_nchw_to_nhwc_f32(const float* x, float* y, int N, int C, int HW, int Cp, void* stream) {
    CCST_REQUIRE(x && y && N > 0 && C > 0 && HW > 0 && Cp >= C && N <= 65535, "nchw_to_nhwc: bad args");
    dim3 grid((HW + 31) / 32, (Cp + 31) / 32, N);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, HW, Cp);
    return ccst_launch_status("nchw_to_nhwc");
}

extern "C" int ccst_nhwc_to_nchw_f32(const float* x, float* y, int N, int C, int HW, int Cs, void* stream) {
    CCST_REQUIRE(x && y && N > 0 && C > 0 && HW > 0 && Cs >= C && N <= 65535, "nhwc_to_nchw: bad args");
    dim3 grid((HW + 31) / 32, (C + 31) / 32, N);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, HW, Cs);
    return ccst_launch_status("nhwc_to_nchw");
}

extern "C" int ccst_resize_bilinear_nchw_f32(const float* x, float* y, int planes, int H, int W, int oh, int ow, void* stream) {
    CCST_REQUIRE(x && y && planes > 0 && H > 0 && W > 0 && oh > 0 && ow > 0, "resize_bilinear: bad args");
    const long long total = (long long)planes * oh * ow;
    hipLaunchKernelGGL(resize_bilinear_planes_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, oh, ow,
                       (float)H / (float)oh, (float)W / (float)ow, total);
    return ccst_launch_status("resize_bilinear");
}

extern "C" int ccst_quantize_u8_hwc_f32(const float* x_nchw, uint8_t* y_nhwc, int N, int C, int HW, void* stream) {
    CCST_REQUIRE(x_nchw && y_nhwc && N > 0 && C > 0 && HW > 0, "quantize_u8: bad args");
    const long long total = (long long)N * HW;
    hipLaunchKernelGGL(quantize_u8_hwc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x_nchw, y_nhwc, C, HW, total);
    return ccst_launch_status("quantize_u8");
}

extern "C" int ccst_nhwc_layer_f32(int mode, const float* x, float* y, int N, int H, int W, int C, int pad, void* stream) {
    CCST_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "nhwc_layer: bad args (C %% 4 must be 0)");
    const int C4 = C / 4;
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) {
        const long long t = (long long)N * H * W * C4;
        hipLaunchKernelGGL(nhwc_map_kernel<0>, dim3(grid_for(t)), dim3(256), 0, s, (const f32x4*)x, (f32x4*)y, N, H, W, H, W, C4, 0);
    } else if (mode == 1) {
        CCST_REQUIRE(pad >= 0 && pad < H && pad < W, "reflection pad %d needs extent > pad", pad);
        const int Ho = H + 2 * pad, Wo = W + 2 * pad;
        const long long t = (long long)N * Ho * Wo * C4;
        hipLaunchKernelGGL(nhwc_map_kernel<1>, dim3(grid_for(t)), dim3(256), 0, s, (const f32x4*)x, (f32x4*)y, N, H, W, Ho, Wo, C4, pad);
    } else if (mode == 2) {
        const long long t = (long long)N * 4 * H * W * C4;
        hipLaunchKernelGGL(nhwc_map_kernel<2>, dim3(grid_for(t)), dim3(256), 0, s, (const f32x4*)x, (f32x4*)y, N, H, W, 2 * H, 2 * W, C4, 0);
    } else if (mode == 3) {
        const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
        const long long t = (long long)N * Ho * Wo * C4;
        hipLaunchKernelGGL(nhwc_map_kernel<3>, dim3(grid_for(t)), dim3(256), 0, s, (const f32x4*)x, (f32x4*)y, N, H, W, Ho, Wo, C4, 0);
    } else {
        ccst_set_error("nhwc_layer: unknown mode %d", mode);
        return CCST_EINVAL;
    }
    return ccst_launch_status("nhwc_layer");
}
