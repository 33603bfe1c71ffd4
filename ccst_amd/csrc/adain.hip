// AdaIN feature statistics and normalisation (HBM-bound; gfx950).
//
//   partials : per-(n, split, c) fp32 sum and sum of squares over a slice of the H*W plane
//              (each thread adds <= a few dozen values in fp32; wide 16-B loads).  For calc_mean_std / AdaIN
//              (SHIFT) the sums are of d = x - pivot, pivot = the slice's own first element of that channel:
//              the reference computes a two-pass variance (function.py:9), and raw fp32 sum(x^2) loses it
//              entirely once |mean| >> sigma (100 +- 0.01: ulp(1e4) = 1e-3 vs var = 1e-4) or the plane is
//              constant; shifted sums keep |d| ~ sigma, and a constant slice gives exactly 0.
//   finalize : per slice mean_k = pivot_k + sum d / n_k, M2_k = sum d^2 - (sum d)^2 / n_k, merged over the
//              slices in fp64 (Chan et al.) -> mean, sqrt(M2/(HW-1)+eps)   (calc_mean_std, function.py:4-13);
//              or raw per-channel totals over N too (calc_sum, mean_std...py:103-115: fp32 sums as the reference).
//   apply    : y = ((x-mu_c)/sigma_c)*sigma_s+mu_s, then y*alpha + x*(1-alpha)      (function.py:26-33,
//              CCST_OverallStyleTransfer.py:45), one read + one write of the tensor.
// layout 0 = NCHW planes (API tensors), 1 = NHWC (the pipeline's internal layout).
#include "common.h"

namespace {

constexpr int TPB = 256;

// ---- NHWC: x[n][p][c]; thread = one float4 channel group, PL pixel lanes per block -----------
template <bool SHIFT>
__global__ __launch_bounds__(TPB) void partials_nhwc_kernel(const float* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                            int S, int cgb, int PL) {
    __shared__ f32x4 red[2][TPB];
    const int n = blockIdx.z, split = blockIdx.x;
    const int t = threadIdx.x;
    const int cgl = t % cgb, pl = t / cgb;
    const int cg = blockIdx.y * cgb + cgl;
    const int per = (HW + S - 1) / S;
    const int p0 = split * per, p1 = min(HW, p0 + per);
    f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
    if (pl < PL && cg * 4 < C && p0 < p1) {
        const float* base = x + ((long long)n * HW) * C + cg * 4;
        f32x4 pv = {0, 0, 0, 0};
        if (SHIFT) pv = *reinterpret_cast<const f32x4*>(base + (long long)p0 * C);
        // four pixels in flight per thread: one load per iteration is a chain of HBM latencies (0.85 TB/s on the 268 MB relu4_1 batch
        // of stage 1); the partial sums stay in pixel order (s0 + s1) + (s2 + s3) per iteration -- fixed, so reproducible
        int p = p0 + pl;
        for (; p + 3 * PL < p1; p += 4 * PL) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(base + (long long)p * C) - pv;
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(base + (long long)(p + PL) * C) - pv;
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(base + (long long)(p + 2 * PL) * C) - pv;
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(base + (long long)(p + 3 * PL) * C) - pv;
            s += (v0 + v1) + (v2 + v3);
            q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
        }
        for (; p < p1; p += PL) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)p * C) - pv;
            s += v;
            q += v * v;
        }
    }
    red[0][t] = s;
    red[1][t] = q;
    __syncthreads();
    if (pl == 0 && cg * 4 < C) {
        for (int k = 1; k < PL; ++k) {
            s += red[0][t + k * cgb];
            q += red[1][t + k * cgb];
        }
        float* o = part + (((long long)n * S + split) * C + cg * 4) * 2;
        // layout [n][split][c][2] interleaved (sum, sq) per channel
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[2 * j] = s[j];
            o[2 * j + 1] = q[j];
        }
    }
}

// ---- NCHW: plane (n,c) contiguous; block = (plane, split) -----------------------------------
template <bool SHIFT>
__global__ __launch_bounds__(TPB) void partials_nchw_kernel(const float* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                            int S) {
    __shared__ float red[2][TPB / 64];
    const int plane = blockIdx.y, split = blockIdx.x;
    const int n = plane / C, c = plane - n * C;
    const int per = (HW + S - 1) / S;
    const int p0 = split * per, p1 = min(HW, p0 + per);
    const float* base = x + (long long)plane * HW;
    float s = 0.f, q = 0.f;
    const float pv = (SHIFT && p0 < p1) ? base[p0] : 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += TPB) {
        const float v = base[p] - pv;
        s += v;
        q += v * v;
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s;
        red[1][threadIdx.x >> 6] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0.f, tq = 0.f;
        for (int k = 0; k < TPB / 64; ++k) {
            ts += red[0][k];
            tq += red[1][k];
        }
        float* o = part + (((long long)n * S + split) * C + c) * 2;
        o[0] = ts;
        o[1] = tq;
    }
}

// mean/std per (n,c) from the SHIFTED split partials, merged in fp64.  A workgroup owns 16 consecutive (n,c) pairs;
// 16 split-lanes per pair each fold every 16th slice (coalesced: consecutive threads read consecutive channels of
// one split), then the lanes are folded in fixed order through LDS.  Two rounds: the plane mean (sum n_k * mean_k),
// then M2 = sum M2_k + n_k (mean_k - mean)^2.
__global__ __launch_bounds__(256) void finalize_mean_std_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                                float* __restrict__ mean, float* __restrict__ stdv, int N, int C, int S,
                                                                int HW, int layout, float eps) {
    __shared__ double red[16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + cl;
    const bool live = i < N * C;
    const int n = live ? i / C : 0, c = live ? i - n * C : 0;
    const int per = (HW + S - 1) / S;
    auto slice = [&](int k, double& cnt, double& mk, double& m2k) {
        const int p0 = k * per, p1 = min(HW, p0 + per);
        cnt = (double)max(0, p1 - p0);
        if (p1 <= p0) { mk = 0.0; m2k = 0.0; return; }
        const float2 o = *reinterpret_cast<const float2*>(part + (((long long)n * S + k) * C + c) * 2);
        const double pv = (double)(layout == 1 ? x[((long long)n * HW + p0) * C + c] : x[((long long)n * C + c) * HW + p0]);
        const double sd = (double)o.x, qd = (double)o.y;
        mk = pv + sd / cnt;
        m2k = qd - sd * sd / cnt;
        if (m2k < 0.0) m2k = 0.0;
    };
    double acc = 0.0;
    if (live)
        for (int k = sl; k < S; k += 16) {
            double cnt, mk, m2k;
            slice(k, cnt, mk, m2k);
            acc += cnt * mk;
        }
    red[sl][cl] = acc;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += red[k][cl];
    const double mu = tot / (double)HW;
    __syncthreads();
    acc = 0.0;
    if (live)
        for (int k = sl; k < S; k += 16) {
            double cnt, mk, m2k;
            slice(k, cnt, mk, m2k);
            acc += m2k + cnt * (mk - mu) * (mk - mu);
        }
    red[sl][cl] = acc;
    __syncthreads();
    if (sl == 0 && live) {
        double m2 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) m2 += red[k][cl];
        const double var = m2 / ((double)HW - 1.0);   // unbiased, function.py:9 (HW==1 -> NaN like the reference)
        mean[i] = (float)mu;
        stdv[i] = sqrtf((float)var + eps);
    }
}

// per-channel totals over n and splits (calc_sum): a workgroup owns 16 channels, 16 slice-lanes per channel each fold every 16th
// (n, split) partial in fp64 (consecutive threads read consecutive channels of one partial), then the lanes in fixed order
// CENTRED: the partials are (sum, M2 about the slab's own mean, count, 0) quadruples (the half-piece conv kernels' epilogues): the raw
// sum of squares of a slab is M2 + sum^2 / count, formed here in fp64
template <bool CENTRED>
__global__ __launch_bounds__(256) void finalize_chan_sums_kernel(const float* __restrict__ part, float* __restrict__ sum, float* __restrict__ sq,
                                                                 int N, int C, int S) {
    __shared__ double red[2][16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0, q = 0.0;
    if (CENTRED) {
        if (c < C)
            for (int k = sl; k < N * S; k += 16) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(part + ((long long)k * C + c) * 4);
                s += (double)o[0];
                q += (double)o[1] + (o[2] > 0.f ? (double)o[0] * (double)o[0] / (double)o[2] : 0.0);
            }
    } else if (c < C) {
        const int K = N * S;
        int k = sl;
        for (; k + 48 < K; k += 64) {            // four loads in flight (L2 latency), folded in index order
            const float2 o0 = *reinterpret_cast<const float2*>(part + ((long long)k * C + c) * 2);
            const float2 o1 = *reinterpret_cast<const float2*>(part + ((long long)(k + 16) * C + c) * 2);
            const float2 o2 = *reinterpret_cast<const float2*>(part + ((long long)(k + 32) * C + c) * 2);
            const float2 o3 = *reinterpret_cast<const float2*>(part + ((long long)(k + 48) * C + c) * 2);
            s += (double)o0.x; q += (double)o0.y;
            s += (double)o1.x; q += (double)o1.y;
            s += (double)o2.x; q += (double)o2.y;
            s += (double)o3.x; q += (double)o3.y;
        }
        for (; k < K; k += 16) {
            const float2 o = *reinterpret_cast<const float2*>(part + ((long long)k * C + c) * 2);
            s += (double)o.x;
            q += (double)o.y;
        }
    }
    red[0][sl][cl] = s;
    red[1][sl][cl] = q;
    __syncthreads();
    if (sl == 0 && c < C) {
        double ts = 0.0, tq = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            ts += red[0][k][cl];
            tq += red[1][k][cl];
        }
        sum[c] = (float)ts;
        sq[c] = (float)tq;
    }
}

__device__ __forceinline__ float adain_one(float x, float mu, float sd, float sm, float ss, float alpha, bool blend) {
    // evaluation order AND rounding of function.py:31-33 (sub, div, mul, add as four separately rounded tensor ops: no FMA)
    float t = __fadd_rn(__fmul_rn(__fdiv_rn(__fsub_rn(x, mu), sd), ss), sm);
    if (blend) t = __fadd_rn(__fmul_rn(t, alpha), __fmul_rn(x, 1.f - alpha)); // CCST_OverallStyleTransfer.py:45
    return t;
}

__global__ __launch_bounds__(TPB) void adain_apply_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               const float* __restrict__ mean, const float* __restrict__ stdv,
                                                               const float* __restrict__ smean, const float* __restrict__ sstd,
                                                               int style_per_n, float alpha, long long per_image4, int HW, int C,
                                                               unsigned* __restrict__ ymax) {
    // (grid: x = blocks over one image's quads, y = image -- the |max| words are per image)
    const int cg = C / 4;
    const bool blend = (alpha != 1.f);
    float amax = 0.f;
    const int n = blockIdx.y;
    for (long long j = (long long)blockIdx.x * TPB + threadIdx.x; j < per_image4; j += (long long)gridDim.x * TPB) {
        const long long i = (long long)n * per_image4 + j;
        const int c = (int)(j % cg) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + n * C + c);
        const f32x4 sd = *reinterpret_cast<const f32x4*>(stdv + n * C + c);
        const int so = (style_per_n ? n * C : 0) + c;
        const f32x4 sm = *reinterpret_cast<const f32x4*>(smean + so);
        const f32x4 ss = *reinterpret_cast<const f32x4*>(sstd + so);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = adain_one(v[j], mu[j], sd[j], sm[j], ss[j], alpha, blend);
        amax = fmaxf(fmaxf(fmaxf(amax, fabsf(o[0])), fmaxf(fabsf(o[1]), fabsf(o[2]))), fabsf(o[3]));
        *reinterpret_cast<f32x4*>(y + i * 4) = o;
    }
    if (ymax != nullptr) ccst_absmax_publish(ymax + n * CCST_ABSMAX_WORDS, amax, blockIdx.x);
}

__global__ __launch_bounds__(TPB) void adain_apply_nchw_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               const float* __restrict__ mean, const float* __restrict__ stdv,
                                                               const float* __restrict__ smean, const float* __restrict__ sstd,
                                                               int style_per_n, float alpha, int HW, int C, unsigned* __restrict__ ymax) {
    const int plane = blockIdx.y;
    const int c = plane % C;
    const float mu = mean[plane], sd = stdv[plane];
    const int so = style_per_n ? plane : c;
    const float sm = smean[so], ss = sstd[so];
    const bool blend = (alpha != 1.f);
    const float* xb = x + (long long)plane * HW;
    float* yb = y + (long long)plane * HW;
    float amax = 0.f;
    for (int p = blockIdx.x * TPB + threadIdx.x; p < HW; p += gridDim.x * TPB) {
        const float o = adain_one(xb[p], mu, sd, sm, ss, alpha, blend);
        amax = fmaxf(amax, fabsf(o));
        yb[p] = o;
    }
    if (ymax != nullptr) ccst_absmax_publish(ymax + (plane / C) * CCST_ABSMAX_WORDS, amax, blockIdx.y * gridDim.x + blockIdx.x);
}

// ---- single-pass AdaIN (NHWC, H*W <= 4096): statistics AND normalise with the tensor read once and written once ------------------
// One workgroup of 512 threads owns image n x 16 channels x ALL pixels: thread (pixel lane = t >> 2 of 128, channel quad = t & 3)
// keeps its <= 32 pixels x 4 channels in registers (<= 128 VGPRs), so the reference's own TWO-PASS statistics (mean, then the
// unbiased variance of x - mean, function.py:9-13) cost no second trip to memory: wavefront reductions over the 16 pixel lanes of a
// wave (xor-shuffles that keep the channel quad), then over the 8 waves through LDS in fixed order (fp64), twice.  The normalise
// (function.py:31-33, four separately rounded operations) and the alpha blend run on the registers; 64 contiguous bytes per pixel
// in and out.  Algorithmic traffic = the HBM roofline's 2 * N*C*H*W*4 bytes, in ONE launch.
#define FP_CQ 4                      // channel quads per workgroup (16 channels = 64 contiguous bytes per pixel)
#define FP_THREADS 512
constexpr int FP_T = FP_THREADS, FP_PL = FP_T / FP_CQ, FP_PPT = 4096 / FP_PL, FP_W = FP_T / 64;

__device__ __forceinline__ f32x4 quad_lane_sum(f32x4 v) {       // sum over the lanes of a wave that share lane % FP_CQ
#pragma unroll
    for (int o = FP_CQ; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += __shfl_xor(v[j], o, 64);
    return v;
}

__global__ __launch_bounds__(FP_T) void adain_fused_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                const float* __restrict__ smean, const float* __restrict__ sstd,
                                                                int style_per_n, float alpha, int HW, int C, float eps,
                                                                float* __restrict__ mean_out, float* __restrict__ std_out,
                                                                unsigned* __restrict__ ymax) {
    __shared__ double red[2][FP_W][FP_CQ][4];
    const int t = threadIdx.x, cq = t % FP_CQ, pl = t / FP_CQ, wave = t >> 6;
    const int n = blockIdx.y, c0 = blockIdx.x * (4 * FP_CQ) + cq * 4;
    const float* xb = x + ((long long)n * HW) * C + c0;
    float* yb = y + ((long long)n * HW) * C + c0;
    f32x4 v[FP_PPT];
#pragma unroll
    for (int i = 0; i < FP_PPT; ++i) {
        const int p = pl + i * FP_PL;
        v[i] = (p < HW) ? *reinterpret_cast<const f32x4*>(xb + (long long)p * C) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // pass 1: mean
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FP_PPT; ++i) s += v[i];
    s = quad_lane_sum(s);
    if ((t & 63) < FP_CQ)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[0][wave][cq][j] = (double)s[j];
    __syncthreads();
    f32x4 mu;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < FP_W; ++w) a += red[0][w][cq][j];
        mu[j] = (float)(a / (double)HW);
    }
    // pass 2: unbiased variance of x - mean (registers only)
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FP_PPT; ++i) {
        if (pl + i * FP_PL < HW) {
            const f32x4 d = v[i] - mu;
            q += d * d;
        }
    }
    q = quad_lane_sum(q);
    if ((t & 63) < FP_CQ)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[1][wave][cq][j] = (double)q[j];
    __syncthreads();
    f32x4 sd;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < FP_W; ++w) a += red[1][w][cq][j];
        sd[j] = sqrtf((float)(a / ((double)HW - 1.0)) + eps);
    }
    if (mean_out != nullptr && pl == 0) {
        *reinterpret_cast<f32x4*>(mean_out + n * C + c0) = mu;
        *reinterpret_cast<f32x4*>(std_out + n * C + c0) = sd;
    }
    const int so = (style_per_n ? n * C : 0) + c0;
    const f32x4 sm = *reinterpret_cast<const f32x4*>(smean + so), ss = *reinterpret_cast<const f32x4*>(sstd + so);
    const bool blend = (alpha != 1.f);
    float amax = 0.f;
    if (ymax != nullptr) ymax += n * CCST_ABSMAX_WORDS;            // per image
    const unsigned peeked = ymax != nullptr ? ccst_absmax_peek(ymax, blockIdx.x) : 0u;
#pragma unroll
    for (int i = 0; i < FP_PPT; ++i) {
        const int p = pl + i * FP_PL;
        if (p < HW) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = adain_one(v[i][j], mu[j], sd[j], sm[j], ss[j], alpha, blend);
            amax = fmaxf(fmaxf(fmaxf(amax, fabsf(o[0])), fmaxf(fabsf(o[1]), fabsf(o[2]))), fabsf(o[3]));
            *reinterpret_cast<f32x4*>(yb + (long long)p * C) = o;
        }
    }
    if (ymax != nullptr) ccst_absmax_publish(ymax, amax, blockIdx.x, peeked);
}

// The same fold for the FUSED AdaIN step (round 6): instead of streaming the tensor through (x - mu) / sigma * sigma_s + mu_s and the
// alpha blend, the step becomes the per-(image, channel) affine map  y = a x + b,
//     a = alpha sigma_s / sigma + (1 - alpha),     b = alpha (mu_s - mu sigma_s / sigma)        (in fp64, rounded once),
// which the decoder's first conv applies to x on its way into its input transform (ccst_conv3x3_f43_f32's in_scale / in_shift): the
// normalised tensor is never written or read -- the step's 100 MB pass is gone.  The conv needs the |max| words of the MAPPED tensor:
// the records' fourth float is the slab's largest |x|, so  max |a x + b| <= |a| max |x| + |b|  per channel -- or, for x >= 0 (x_nonneg: the
// producer applied ReLU), exactly max(|b|, |a max x + b|), the map being monotone on [0, max x] -- maximised over the image's channels into
// y_absmax[n] (an upper bound is a valid scale).  Centred records only (the half-piece conv kernels' epilogues).
__global__ __launch_bounds__(TPB) void tile_stats_affine_kernel(const float* __restrict__ part, int tpi, int HW, int C, float eps,
                                                                const float* __restrict__ smean, const float* __restrict__ sstd, int style_per_n,
                                                                float alpha, int x_nonneg, float* __restrict__ mean_out, float* __restrict__ std_out,
                                                                float* __restrict__ a_out, float* __restrict__ b_out, unsigned* __restrict__ ymax) {
    __shared__ double red[2][16][17];
    __shared__ float redm[16][17];
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + cl;                    // (image, channel); C % 16 == 0, so a workgroup stays inside one image
    const int n = i / C, c = i - n * C;
    double s = 0.0, q = 0.0, piv = 0.0;
    float vm = 0.f;
    const f32x4* pp = reinterpret_cast<const f32x4*>(part) + (long long)n * tpi * C + c;
    const f32x4 a0 = pp[0];
    piv = a0[2] > 0.f ? (double)(a0[0] / a0[2]) : 0.0;
    for (int k = kl; k < tpi; k += 16) {
        const f32x4 a = pp[(long long)k * C];
        if (a[2] > 0.f) {
            const double tk = (double)a[0] - (double)a[2] * piv;
            s += tk;
            q += (double)a[1] + tk * tk / (double)a[2];
            vm = fmaxf(vm, a[3]);
        }
    }
    red[0][kl][cl] = s;
    red[1][kl][cl] = q;
    redm[kl][cl] = vm;
    __syncthreads();
    float bound = 0.f;
    if (kl == 0) {
        s = red[0][0][cl];
        q = red[1][0][cl];
        vm = redm[0][cl];
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            s += red[0][k][cl];
            q += red[1][k][cl];
            vm = fmaxf(vm, redm[k][cl]);
        }
        const double m = s / (double)HW;
        const double var = fmax(q - s * m, 0.0) / ((double)HW - 1.0);
        const float mu = (float)(m + piv), sd = sqrtf((float)var + eps);          // the statistics as ccst_adain_tile_sums_f32 rounds them
        mean_out[i] = mu;
        std_out[i] = sd;
        const int so = style_per_n ? i : c;
        const double r = (double)sstd[so] / (double)sd;
        const float a = (float)((double)alpha * r + (1.0 - (double)alpha));
        const float b = (float)((double)alpha * ((double)smean[so] - (double)mu * r));
        a_out[i] = a;
        b_out[i] = b;
        bound = (x_nonneg ? fmaxf(fabsf(b), fabsf(__builtin_fmaf(a, vm, b))) : fabsf(a) * vm + fabsf(b)) * 1.0000005f;   // (rounded up: the words must bound every mapped value)
    }
    ccst_absmax_publish(ymax + n * CCST_ABSMAX_WORDS, bound, blockIdx.x);
}

int pick_splits(int N, int C, int HW, int layout) {
    // aim for ~2048 workgroups (8 per CU, four 16-byte loads in flight per thread), at least ~8 pixels (NHWC) / 1024 elements (NCHW) per split
    long long units = (layout == 1) ? (long long)N * ((C / 4 + 255) / 256) : (long long)N * C;
    long long s = 2048 / (units > 0 ? units : 1);
    const int minper = (layout == 1) ? 8 : 1024;
    const long long smax = (HW + minper - 1) / minper;
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int)s;
}

template <bool SHIFT>
int run_partials(const float* x, float* part, int N, int C, int HW, int layout, int S, hipStream_t st) {
    if (layout == 1) {
        const int cg = C / 4;
        const int cgb = cg < TPB ? cg : TPB;
        const int PL = TPB / cgb;
        dim3 grid(S, (cg + cgb - 1) / cgb, N);
        hipLaunchKernelGGL(partials_nhwc_kernel<SHIFT>, grid, dim3(TPB), 0, st, x, part, HW, C, S, cgb, PL);
    } else {
        dim3 grid(S, N * C);
        hipLaunchKernelGGL(partials_nchw_kernel<SHIFT>, grid, dim3(TPB), 0, st, x, part, HW, C, S);
    }
    return ccst_launch_status("stats partials");
}

int check_common(const void* x, int N, int C, int HW, int layout) {
    CCST_REQUIRE(x != nullptr, "stats: null tensor");
    CCST_REQUIRE(N > 0 && C > 0 && HW > 0, "stats: bad extents N=%d C=%d HW=%d", N, C, HW);
    CCST_REQUIRE(layout == 0 || layout == 1, "stats: layout must be 0 (NCHW) or 1 (NHWC)");
    if (layout == 1) CCST_REQUIRE(C % 4 == 0, "stats: NHWC path needs C %% 4 == 0 (C=%d)", C);
    CCST_REQUIRE(layout == 1 || (long long)N * C <= 65535, "stats: NCHW path supports at most 65535 planes");
    CCST_REQUIRE(N <= 65535, "stats: N too large");
    return CCST_OK;
}

// AdaIN where the producer of x has already left the statistics: per-(spatial tile, channel) records from the epilogue of the conv that
// wrote x (tiles n * tpi .. (n + 1) * tpi - 1 belong to image n, pixels outside the image excluded).  Two launches (round 5; one until
// round 4): tile_stats_fold_kernel folds the records ONCE -- a thread per (image, channel), fp64 -- into mu[N][C], sigma[N][C]; the
// streaming kernel then reads two floats per channel and streams the tensor once, every CU busy, no LDS, no barrier, no fp64.  (The
// one-launch form made every one of its 768 workgroups re-fold its image's records behind the loads it had issued: 27.6 us against
// 23.4 with the raw pairs -- the centred records are the right numerics, folding them in every pixel block was the price.)
//   records: CENTRED (the half-piece conv kernels): (S, M2 about the slab's own mean, count, 0) per (tile, channel); the fold keeps
//   s = sum of (x - pivot) and q = sum of (x - pivot)^2 about a pivot (the first slab's mean), assembled in fp64 from slab quantities
//   that carry no cancellation: the variance is sum M2_i + sum n_i (mean_i - mean)^2 up to fp64 rounding however large |mean| / sigma.
//   Raw (sum, sum of squares) pairs (a caller's own fp32 producer; round 3's F(4x4) kernel wrote them): mean = S / HW, unbiased variance = (Q - S mean) / (HW - 1) in fp64.
// (a workgroup = 16 channels of one image x 16 record lanes: lane kl folds records kl, kl + 16, ...; the lanes are then summed in
//  fixed order through LDS -- one memory round trip and one barrier deep, bitwise reproducible)
template <bool CENTRED>
__global__ __launch_bounds__(TPB) void tile_stats_fold_kernel(const float* __restrict__ part, int tpi, int HW, int C, int NC, float eps,
                                                              float* __restrict__ mean_out, float* __restrict__ std_out) {
    __shared__ double red[2][16][17];
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + cl;                    // (image, channel); C % 16 == 0, so a workgroup stays inside one image
    const int n = i / C, c = i - n * C;
    double s = 0.0, q = 0.0, piv = 0.0;
    if (CENTRED) {
        const f32x4* pp = reinterpret_cast<const f32x4*>(part) + (long long)n * tpi * C + c;
        const f32x4 a0 = pp[0];
        piv = a0[2] > 0.f ? (double)(a0[0] / a0[2]) : 0.0;          // (any pivot near the data will do: fp32 division)
        for (int k = kl; k < tpi; k += 16) {
            const f32x4 a = pp[(long long)k * C];
            if (a[2] > 0.f) {
                // sum of (x - pivot) = S - n pivot, exactly in fp64; sum of (x - pivot)^2 = M2 + (S - n pivot)^2 / n
                const double tk = (double)a[0] - (double)a[2] * piv;
                s += tk;
                q += (double)a[1] + tk * tk / (double)a[2];
            }
        }
    } else {
        const float* pp = part + ((long long)n * tpi * C + c) * 2;
        for (int k = kl; k < tpi; k += 16) {
            s += (double)pp[(long long)k * C * 2];
            q += (double)pp[(long long)k * C * 2 + 1];
        }
    }
    red[0][kl][cl] = s;
    red[1][kl][cl] = q;
    __syncthreads();
    if (kl != 0) return;
    s = red[0][0][cl];
    q = red[1][0][cl];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        s += red[0][k][cl];
        q += red[1][k][cl];
    }
    const double m = s / (double)HW;                                            // (CENTRED: of x - pivot)
    const double var = fmax(q - s * m, 0.0) / ((double)HW - 1.0);
    mean_out[i] = (float)(m + piv);
    std_out[i] = sqrtf((float)var + eps);
}

constexpr int TS_CQ = 16, TS_PL = TPB / TS_CQ, TS_PIX = 256;
__global__ __launch_bounds__(TPB) void adain_stream_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                const float* __restrict__ cmean, const float* __restrict__ cstd,
                                                                const float* __restrict__ smean, const float* __restrict__ sstd,
                                                                int style_per_n, float alpha, int HW, int C,
                                                                unsigned* __restrict__ ymax) {
    const int t = threadIdx.x, cq = t % TS_CQ, pl = t / TS_CQ;
    const int n = blockIdx.y, c0 = blockIdx.x * (4 * TS_CQ) + cq * 4;
    const float* xb = x + ((long long)n * HW) * C + c0;
    float* yb = y + ((long long)n * HW) * C + c0;
    const int p0 = blockIdx.z * TS_PIX;
    f32x4 v[TS_PIX / TS_PL];
#pragma unroll
    for (int i = 0; i < TS_PIX / TS_PL; ++i) {
        const int p = p0 + pl + i * TS_PL;
        if (p < HW) v[i] = *reinterpret_cast<const f32x4*>(xb + (long long)p * C);
    }
    const f32x4 mu = *reinterpret_cast<const f32x4*>(cmean + n * C + c0), sd = *reinterpret_cast<const f32x4*>(cstd + n * C + c0);
    const int so = (style_per_n ? n * C : 0) + c0;
    const f32x4 sm = *reinterpret_cast<const f32x4*>(smean + so), ss = *reinterpret_cast<const f32x4*>(sstd + so);
    const bool blend = (alpha != 1.f);
    float amax = 0.f;
    const unsigned bid_ = blockIdx.z * gridDim.x + blockIdx.x;
    if (ymax != nullptr) ymax += n * CCST_ABSMAX_WORDS;            // per image
    const unsigned peeked = ymax != nullptr ? ccst_absmax_peek(ymax, bid_) : 0u;       // (compared after the stores: nobody waits for it)
#pragma unroll
    for (int i = 0; i < TS_PIX / TS_PL; ++i) {
        const int p = p0 + pl + i * TS_PL;
        if (p < HW) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = adain_one(v[i][j], mu[j], sd[j], sm[j], ss[j], alpha, blend);
            amax = fmaxf(fmaxf(fmaxf(amax, fabsf(o[0])), fmaxf(fabsf(o[1]), fabsf(o[2]))), fabsf(o[3]));
            *reinterpret_cast<f32x4*>(yb + (long long)p * C) = o;
        }
    }
    if (ymax != nullptr) ccst_absmax_publish(ymax, amax, bid_, peeked);
}

// CCST_OverallStyleTransfer.py:36-45, the interpolation branch: feat = sum_k w_k * base[k] (accumulated in the reference's order, from
// zero, every product and sum rounded separately), then feat * alpha + content[0] * (1 - alpha).  Elementwise, so any layout.
__global__ __launch_bounds__(TPB) void interp_blend_kernel(const float* __restrict__ base, const float* __restrict__ content0,
                                                           const float* __restrict__ w, int K, long long elems, float alpha, float oma,
                                                           float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < elems; i += (long long)gridDim.x * TPB) {
        float feat = 0.f;
        for (int k = 0; k < K; ++k) feat = __fadd_rn(feat, __fmul_rn(w[k], base[(long long)k * elems + i]));
        out[i] = __fadd_rn(__fmul_rn(feat, alpha), __fmul_rn(content0[i], oma));
    }
}

}  // namespace

extern "C" int64_t ccst_stats_workspace_bytes(int N, int C, int HW) {
    (void)HW;
    // partials [N][S<=256][C][2] + mean[N*C] + std[N*C]
    return ((int64_t)N * 256 * C * 2 + 2LL * N * C) * 4;
}

extern "C" int ccst_calc_mean_std_f32(const float* x, float* mean, float* stdv, int N, int C, int HW, int layout, float eps,
                                      void* ws, int64_t ws_bytes, void* stream) {
    int rc = check_common(x, N, C, HW, layout);
    if (rc) return rc;
    CCST_REQUIRE(mean && stdv && ws, "calc_mean_std: null pointer");
    if (ws_bytes < ccst_stats_workspace_bytes(N, C, HW)) {
        ccst_set_error("calc_mean_std: workspace %lld < %lld", (long long)ws_bytes, (long long)ccst_stats_workspace_bytes(N, C, HW));
        return CCST_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    // per-plane statistics: the split count (= the summation order) is a function of the plane only, so a sample's result does
    // not depend on how many others share its batch (the batch-sliced AdaIN path, net.py, relies on it)
    const int S = pick_splits(1, C, HW, layout);
    float* part = (float*)ws;
    rc = run_partials<true>(x, part, N, C, HW, layout, S, st);
    if (rc) return rc;
    hipLaunchKernelGGL(finalize_mean_std_kernel, dim3((N * C + 15) / 16), dim3(256), 0, st, x, part, mean, stdv, N, C, S, HW, layout, eps);
    return ccst_launch_status("finalize_mean_std");
}

extern "C" int ccst_adain_f32(const float* x, const float* style_mean, const float* style_std, int style_per_n, float alpha,
                              float* y, int N, int C, int HW, int layout, float eps, void* ws, int64_t ws_bytes, uint32_t* y_absmax, void* stream) {
    int rc = check_common(x, N, C, HW, layout);
    if (rc) return rc;
    CCST_REQUIRE(style_mean && style_std && y && ws, "adain: null pointer");
    CCST_REQUIRE(alpha >= 0.f && alpha <= 1.f, "adain: alpha=%f outside [0,1]", (double)alpha);
    if (ws_bytes < ccst_stats_workspace_bytes(N, C, HW)) {
        ccst_set_error("adain: workspace too small");
        return CCST_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    float* mean = part + (int64_t)N * 256 * C * 2;
    float* stdv = mean + (int64_t)N * C;
    if (layout == 1 && C % (4 * FP_CQ) == 0 && HW >= 2 && HW <= FP_PL * FP_PPT && N <= 65535) {      // the metric's shape: one pass, one launch
        hipLaunchKernelGGL(adain_fused_nhwc_kernel, dim3(C / (4 * FP_CQ), N), dim3(FP_T), 0, st, x, y, style_mean, style_std, style_per_n, alpha, HW,
                           C, eps, mean, stdv, y_absmax);
        return ccst_launch_status("adain_fused");
    }
    rc = ccst_calc_mean_std_f32(x, mean, stdv, N, C, HW, layout, eps, ws, ws_bytes, stream);
    if (rc) return rc;
    if (layout == 1) {
        CCST_REQUIRE(N <= 65535, "adain: N must be <= 65535");
        const long long per4 = (long long)HW * (C / 4);
        const long long want = (per4 + TPB - 1) / TPB, cap = (4096 + N - 1) / N;
        hipLaunchKernelGGL(adain_apply_nhwc_kernel, dim3((unsigned)(want < cap ? want : cap), N), dim3(TPB), 0, st, x, y, mean, stdv, style_mean, style_std,
                           style_per_n, alpha, per4, HW, C, y_absmax);
    } else {
        const int gx = (HW + TPB - 1) / TPB < 64 ? (HW + TPB - 1) / TPB : 64;
        hipLaunchKernelGGL(adain_apply_nchw_kernel, dim3(gx, N * C), dim3(TPB), 0, st, x, y, mean, stdv, style_mean, style_std,
                           style_per_n, alpha, HW, C, y_absmax);
    }
    return ccst_launch_status("adain_apply");
}

// function.py:26-33 (+ the alpha blend) on an NHWC tensor x [N][HW][C] whose producer left per-tile channel sums: partials
// [N * tiles_per_image][C][2] = (sum, sum of squares) of x over the pixels of spatial tile t, tiles of image n contiguous
// -- or [..][C][4] centred records (ccst_conv3x3_f43_f32's / ccst_conv3x3_halo_split_f32's chan_sum_partials).  C a multiple of 64.
// mean_out / std_out: [N*C] each, REQUIRED: the content statistics, handed from the fold launch to the streaming launch.
extern "C" int ccst_adain_tile_sums_f32(const float* x, const float* partials, int partial_floats, int tiles_per_image, const float* style_mean,
                                        const float* style_std, int style_per_n, float alpha, float* y, int N, int C, int HW, float eps,
                                        float* mean_out, float* std_out, uint32_t* y_absmax, void* stream) {
    CCST_REQUIRE(x && partials && style_mean && style_std && y, "adain_tile_sums: null pointer");
    CCST_REQUIRE(mean_out && std_out, "adain_tile_sums: mean_out / std_out ([N*C] floats each) carry the folded statistics to the streaming kernel");
    CCST_REQUIRE(partial_floats == 2 || partial_floats == 4, "adain_tile_sums: partials are [..][C][2] (sum, sum^2) or [..][C][4] (sum, M2, count, 0)");
    CCST_REQUIRE(N > 0 && N <= 65535 && C > 0 && C % (4 * TS_CQ) == 0 && HW >= 2 && tiles_per_image > 0, "adain_tile_sums: bad shape (C %% 64 == 0, HW >= 2)");
    CCST_REQUIRE(alpha >= 0.f && alpha <= 1.f, "adain_tile_sums: alpha=%f outside [0,1]", (double)alpha);
    const int chunks = (HW + TS_PIX - 1) / TS_PIX;
    CCST_REQUIRE(chunks <= 65535, "adain_tile_sums: plane too large");
    hipStream_t st = (hipStream_t)stream;
    const int NC = N * C;
    if (partial_floats == 4)
        hipLaunchKernelGGL(tile_stats_fold_kernel<true>, dim3(NC / 16), dim3(TPB), 0, st, partials, tiles_per_image, HW, C, NC, eps, mean_out, std_out);
    else
        hipLaunchKernelGGL(tile_stats_fold_kernel<false>, dim3(NC / 16), dim3(TPB), 0, st, partials, tiles_per_image, HW, C, NC, eps, mean_out, std_out);
    hipLaunchKernelGGL(adain_stream_nhwc_kernel, dim3(C / (4 * TS_CQ), N, chunks), dim3(TPB), 0, st, x, y, mean_out, std_out, style_mean, style_std,
                       style_per_n, alpha, HW, C, y_absmax);
    return ccst_launch_status("adain_tile_sums");
}

// function.py:26-33 + the alpha blend as an affine map for the consumer to apply (see tile_stats_affine_kernel): partials = the centred
// records [N * tiles_per_image][C][4] (sum, M2, count, max |x|) of ccst_conv3x3_f43_f32's epilogue.  Outputs, [N*C] floats each: the
// content statistics (mean_out, std_out) and the map (a_out, b_out); y_absmax: zeroed per-image words [N][CCST_ABSMAX_WORDS] that receive
// a bound of max |a x + b|.  One launch of N*C/16 small workgroups; nothing touches the features.
extern "C" int ccst_adain_fold_affine_f32(const float* partials, int tiles_per_image, const float* style_mean, const float* style_std,
                                          int style_per_n, float alpha, int x_nonneg, int N, int C, int HW, float eps, float* mean_out,
                                          float* std_out, float* a_out, float* b_out, uint32_t* y_absmax, void* stream) {
    CCST_REQUIRE(partials && style_mean && style_std && mean_out && std_out && a_out && b_out && y_absmax, "adain_fold_affine: null pointer");
    CCST_REQUIRE(N > 0 && N <= 65535 && C > 0 && C % 16 == 0 && HW >= 2 && tiles_per_image > 0, "adain_fold_affine: bad shape (C %% 16 == 0, HW >= 2)");
    CCST_REQUIRE(alpha >= 0.f && alpha <= 1.f, "adain_fold_affine: alpha=%f outside [0,1]", (double)alpha);
    hipLaunchKernelGGL(tile_stats_affine_kernel, dim3(N * C / 16), dim3(TPB), 0, (hipStream_t)stream, partials, tiles_per_image, HW, C, eps,
                       style_mean, style_std, style_per_n, alpha, x_nonneg, mean_out, std_out, a_out, b_out, y_absmax);
    return ccst_launch_status("adain_fold_affine");
}

// style_transfer's interpolation branch (CCST_OverallStyleTransfer.py:36-45): base [K][elems] = the K stylised feature maps of one
// content image, content0 [elems] its encoder output, weights [K] (device), out [elems] = (sum_k w_k base[k]) * alpha + content0 *
// one_minus_alpha (the host's float(1 - alpha), as the reference's Python computes it).
extern "C" int ccst_interp_blend_f32(const float* base, const float* content0, const float* weights, int K, int64_t elems, float alpha,
                                     float one_minus_alpha, float* out, void* stream) {
    CCST_REQUIRE(base && content0 && weights && out && K > 0 && elems > 0, "interp_blend: bad args");
    CCST_REQUIRE(alpha >= 0.f && alpha <= 1.f, "interp_blend: alpha=%f outside [0,1]", (double)alpha);
    const long long blocks = (elems + TPB - 1) / TPB;
    hipLaunchKernelGGL(interp_blend_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(TPB), 0, (hipStream_t)stream, base, content0,
                       weights, K, (long long)elems, alpha, one_minus_alpha, out);
    return ccst_launch_status("interp_blend");
}

// Fold K per-tile records -- (sum, sum of squares) pairs [K][C][2] or the conv epilogues' centred quadruples [K][C][4] -- into the per-channel totals.
extern "C" int ccst_chan_sums_finalize_f32(const float* partials, int partial_floats, int K, int C, float* sum, float* sqsum, void* stream) {
    CCST_REQUIRE(partials && sum && sqsum && K > 0 && C > 0, "chan_sums_finalize: bad args");
    CCST_REQUIRE(partial_floats == 2 || partial_floats == 4, "chan_sums_finalize: partials are [K][C][2] (sum, sum^2) or [K][C][4] (sum, M2, count, 0)");
    if (partial_floats == 4) hipLaunchKernelGGL(finalize_chan_sums_kernel<true>, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, sum, sqsum, 1, C, K);
    else hipLaunchKernelGGL(finalize_chan_sums_kernel<false>, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, sum, sqsum, 1, C, K);
    return ccst_launch_status("finalize_chan_sums");
}

extern "C" int ccst_chan_sums_f32(const float* x, float* sum, float* sqsum, int N, int C, int HW, int layout, void* ws,
                                  int64_t ws_bytes, void* stream) {
    int rc = check_common(x, N, C, HW, layout);
    if (rc) return rc;
    CCST_REQUIRE(sum && sqsum && ws, "chan_sums: null pointer");
    if (ws_bytes < ccst_stats_workspace_bytes(N, C, HW)) {
        ccst_set_error("chan_sums: workspace too small");
        return CCST_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int S = pick_splits(N, C, HW, layout);
    float* part = (float*)ws;
    rc = run_partials<false>(x, part, N, C, HW, layout, S, st);
    if (rc) return rc;
    hipLaunchKernelGGL(finalize_chan_sums_kernel<false>, dim3((C + 15) / 16), dim3(256), 0, st, part, sum, sqsum, N, C, S);
    return ccst_launch_status("finalize_chan_sums");
}
