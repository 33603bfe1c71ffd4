// Input edge of both CLIs on the GPU: crop -> PIL-exact bilinear resize -> ToTensor -> Normalize -> flip.
//
// What it replaces, per image, on the reference's CPU DataLoader thread:
//   transforms.RandomResizedCrop((S,S),(min,max)) / Resize((S,S))  -> img.crop(box).resize((S,S), BILINEAR)      (PIL, uint8)
//   transforms.ToTensor()                                           -> uint8 HWC -> float CHW / 255
//   transforms.Normalize(mean, std)                                 -> (x - mean[c]) / std[c]
//   transforms.RandomHorizontalFlip(p)                              -> x.flip(-1)
// (data/data_helper.py:161-181, style_transfer/AdaIN/cjm_util/data_helper.py:46-49).  The host decodes the file and draws
// the crop rectangle / flip; only the decoded uint8 pixels cross PCIe (3.8x fewer bytes than the fp32 tensor at 227 -> 222).
//
// PIL's resize is an integer algorithm (libImaging/Resample.c, 8 bits per channel): per output coordinate a window of
// source pixels [xmin, xmin+xmax) with triangle-filter weights normalised in double precision and rounded to 22-bit fixed
// point, accumulated in int32 from 1<<21 and shifted back with a clip to [0,255]; two passes, horizontal first, with a uint8
// intermediate.  ccst_image_plan() builds those tables on the HOST (double arithmetic in the same order, no FMA contraction)
// and the kernel does the integer part, so the uint8 result is byte-identical to PIL's, and the float result bit-identical
// to ToTensor/Normalize (correctly rounded fp32 divide / subtract, no contraction).
#include "common.h"

#include <math.h>

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;      // libImaging/Resample.c

// One axis: bounds[out][2] = (first source index, tap count), coefs[out][ksize] fixed-point weights.
int axis_ksize(int in_size, int out_size) {
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;               // bilinear: filter support 1.0
    return (int)ceil(support) * 2 + 1;
}

void axis_tables(int in_size, int out_size, int ksize, int32_t* bounds, int32_t* coefs) {
#pragma clang fp contract(off)      // PIL's build rounds every product and sum separately
    const double scale = (double)in_size / out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    const double ss = 1.0 / filterscale;
    double w[1024];
    double* k = ksize <= 1024 ? w : new double[ksize];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            const double v = a < 1.0 ? 1.0 - a : 0.0;
            k[x] = v;
            ww += v;
        }
        int32_t* kk = coefs + (size_t)xx * ksize;
        for (int x = 0; x < xmax; ++x) {
            double v = k[x];
            if (ww != 0.0) v /= ww;
            kk[x] = v < 0 ? (int32_t)(-0.5 + v * (1 << PRECISION_BITS)) : (int32_t)(0.5 + v * (1 << PRECISION_BITS));
        }
        for (int x = xmax; x < ksize; ++x) kk[x] = 0;
        bounds[xx * 2] = xmin;
        bounds[xx * 2 + 1] = xmax;
    }
    if (k != w) delete[] k;
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// grid (ceil(out_w/256), out_h, n); thread = one output pixel, all three channels.
__global__ void __launch_bounds__(256) crop_resize_norm_kernel(const uint8_t* __restrict__ src, const CcstImageXform* __restrict__ xfs,
                                                               const int32_t* __restrict__ tables, float* __restrict__ dst, uint8_t* __restrict__ dst_u8,
                                                               int out_h, int out_w, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y, n = blockIdx.z;
    if (ox >= out_w) return;
    const CcstImageXform xf = xfs[n];
    const int bx = tables[xf.bounds_x + 2 * ox], nx = tables[xf.bounds_x + 2 * ox + 1];
    const int by = tables[xf.bounds_y + 2 * oy], ny = tables[xf.bounds_y + 2 * oy + 1];
    const int32_t* __restrict__ kx = tables + xf.coefs_x + (size_t)ox * xf.kx;
    const int32_t* __restrict__ ky = tables + xf.coefs_y + (size_t)oy * xf.ky;
    const uint8_t* __restrict__ base = src + xf.src_off + ((size_t)(xf.crop_i + by) * xf.src_w + xf.crop_j + bx) * 3;
    const size_t pitch = (size_t)xf.src_w * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int r = 0; r < ny; ++r) {
        const uint8_t* __restrict__ row = base + r * pitch;
        int h0 = 1 << (PRECISION_BITS - 1), h1 = h0, h2 = h0;
        for (int x = 0; x < nx; ++x) {
            const int k = kx[x];
            h0 += row[x * 3 + 0] * k;
            h1 += row[x * 3 + 1] * k;
            h2 += row[x * 3 + 2] * k;
        }
        const int k = ky[r];
        a0 += clip8(h0) * k;       // the horizontal pass's uint8 intermediate (ImagingResampleHorizontal_8bpc)
        a1 += clip8(h1) * k;
        a2 += clip8(h2) * k;
    }
    const int v0 = clip8(a0), v1 = clip8(a1), v2 = clip8(a2);
    const int wx = xf.flip ? out_w - 1 - ox : ox;
    if (dst_u8) {
        uint8_t* o = dst_u8 + (((size_t)n * out_h + oy) * out_w + wx) * 3;
        o[0] = (uint8_t)v0;
        o[1] = (uint8_t)v1;
        o[2] = (uint8_t)v2;
    }
    if (dst) {
        const size_t plane = (size_t)out_h * out_w;
        float* o = dst + (size_t)n * 3 * plane + (size_t)oy * out_w + wx;
        // ToTensor: uint8 -> float, .div(255); Normalize: .sub_(mean).div_(std) -- each step correctly rounded, nothing fused
        o[0] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v0, 255.0f), m0), s0);
        o[plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v1, 255.0f), m1), s1);
        o[2 * plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v2, 255.0f), m2), s2);
    }
}

}  // namespace

extern "C" int64_t ccst_image_plan(int n, CcstImageXform* xf, int out_h, int out_w, int32_t* tables, int64_t cap) {
    if (!xf || n <= 0 || out_h <= 0 || out_w <= 0) {
        ccst_set_error("image_plan: bad args");
        return CCST_EINVAL;
    }
    int64_t used = 0;
    for (int i = 0; i < n; ++i) {
        CcstImageXform& t = xf[i];
        if (t.crop_h <= 0 || t.crop_w <= 0 || t.crop_i < 0 || t.crop_j < 0 || t.src_w < t.crop_j + t.crop_w || t.src_off < 0) {
            ccst_set_error("image_plan: image %d has an empty or out-of-range crop", i);
            return CCST_EINVAL;
        }
        t.kx = axis_ksize(t.crop_w, out_w);
        t.ky = axis_ksize(t.crop_h, out_h);
        const int64_t need = 2LL * out_w + (int64_t)out_w * t.kx + 2LL * out_h + (int64_t)out_h * t.ky;
        if (tables) {
            if (used + need > cap) {
                ccst_set_error("image_plan: table buffer too small (%lld ints needed so far, %lld given)", (long long)(used + need), (long long)cap);
                return CCST_EWORKSPACE;
            }
            if (used + need > 0x7fffffffLL) {
                ccst_set_error("image_plan: tables exceed 2^31 ints");
                return CCST_EINVAL;
            }
            t.bounds_x = (int32_t)used;
            t.coefs_x = t.bounds_x + 2 * out_w;
            t.bounds_y = t.coefs_x + out_w * t.kx;
            t.coefs_y = t.bounds_y + 2 * out_h;
            axis_tables(t.crop_w, out_w, t.kx, tables + t.bounds_x, tables + t.coefs_x);
            axis_tables(t.crop_h, out_h, t.ky, tables + t.bounds_y, tables + t.coefs_y);
        }
        used += need;
    }
    return used;
}

extern "C" int ccst_crop_resize_norm_u8_f32(const uint8_t* src, const CcstImageXform* xf, const int32_t* tables, float* dst_nchw,
                                            uint8_t* dst_u8_nhwc, int n, int out_h, int out_w, const float* mean3, const float* std3,
                                            void* stream) {
    CCST_REQUIRE(src && xf && tables && (dst_nchw || dst_u8_nhwc), "crop_resize_norm: null pointer");
    CCST_REQUIRE(n > 0 && n <= 65535 && out_h > 0 && out_h <= 65535 && out_w > 0, "crop_resize_norm: bad sizes");
    float m[3] = {0.f, 0.f, 0.f}, s[3] = {1.f, 1.f, 1.f};
    if (mean3) for (int c = 0; c < 3; ++c) m[c] = mean3[c];      // HOST pointers (three floats each)
    if (std3) for (int c = 0; c < 3; ++c) s[c] = std3[c];
    dim3 grid((out_w + 255) / 256, out_h, n);
    hipLaunchKernelGGL(crop_resize_norm_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, xf, tables, dst_nchw, dst_u8_nhwc, out_h,
                       out_w, m[0], m[1], m[2], s[0], s[1], s[2]);
    return ccst_launch_status("crop_resize_norm");
}
