// ResNet training ops other than convolution (nets/resnet.py:132-191 + torchvision blocks;
// federated/fed_run.py:49-80).  All NHWC fp32, all HBM-bound: 16-B accesses along C, per-channel
// reductions as (split partials in fp32) -> (fixed-order fp64 combine), so results are bitwise
// reproducible run to run (no float atomics).
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int MAXS = 1024;  // max row-splits of a per-channel reduction

// ------------------------------------------------------------------------------------------
// per-channel partial reductions over rows of x[M][C].
// MODE 0: (sum x, sum x^2)                                   -> BN forward statistics
// MODE 1: (sum dy', sum dy' * xhat), dy' = relu ? dy*(y>0) : dy  -> BN backward
// thread = one float4 channel group; PL row lanes per block; grid (S, ceil(C/4/cgb)).
// ------------------------------------------------------------------------------------------
// relu mask: from the saved output y (y > 0) or, when y == nullptr (no residual was added), recomputed from x:
// z = (x-mean)*invstd*gamma + beta > 0 -- one tensor read less in both backward passes.
// MASK: 0 no ReLU, 1 mask from the saved output y, 2 mask recomputed from x, 3 mask from the forward's byte mask (bit j of
// byte i <-> element 4 i + j; written by bn_apply_kernel: 1/16 of a tensor read instead of a whole one)
template <int MODE, int MASK = 0>
__global__ __launch_bounds__(TPB) void chan_partials_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int relu, float* __restrict__ part,
                                                            long long M, int C, int S, int cgb, int PL,
                                                            const unsigned char* __restrict__ rmask = nullptr) {
    __shared__ f32x4 red[2][TPB];
    const int split = blockIdx.x, t = threadIdx.x;
    const int cgl = t % cgb, pl = t / cgb;
    const int cg = blockIdx.y * cgb + cgl;
    const long long per = (M + S - 1) / S;
    const long long r0 = split * per;
    const long long r1 = (r0 + per < M) ? r0 + per : M;
    f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (pl < PL && cg * 4 < C) {
        f32x4 mu = {0, 0, 0, 0}, is = {0, 0, 0, 0}, ga = {0, 0, 0, 0}, be = {0, 0, 0, 0};
        if (MODE == 1) {
            mu = *reinterpret_cast<const f32x4*>(mean + cg * 4);
            is = *reinterpret_cast<const f32x4*>(invstd + cg * 4);
            if (MASK == 2) {
                ga = *reinterpret_cast<const f32x4*>(gamma + cg * 4);
                be = *reinterpret_cast<const f32x4*>(beta + cg * 4);
            }
        }
        // two rows in flight per thread (the loop is a chain of HBM-latency loads otherwise); same summation order as one at a time
        auto fold = [&](const f32x4 v, f32x4 g, const f32x4 o_in, unsigned bits) {
            if (MODE == 0) {
                a += v;
                b += v * v;
            } else {
                const f32x4 xh = (v - mu) * is;
                if (MASK == 3) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = ((bits >> j) & 1u) ? g[j] : 0.f;
                } else if (MASK != 0) {
                    const f32x4 o = (MASK == 1) ? o_in : xh * ga + be;
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
                }
                a += g;
                b += g * xh;
            }
        };
        const f32x4 z4 = {0, 0, 0, 0};
        long long r = r0 + pl;
        for (; r + PL < r1; r += 2 * PL) {
            const long long o0 = r * C + cg * 4, o1 = (r + PL) * C + cg * 4;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + o0), v1 = *reinterpret_cast<const f32x4*>(x + o1);
            f32x4 g0 = z4, g1 = z4, y0 = z4, y1 = z4;
            unsigned b0 = 0, b1 = 0;
            if (MODE == 1) {
                g0 = *reinterpret_cast<const f32x4*>(dy + o0);
                g1 = *reinterpret_cast<const f32x4*>(dy + o1);
                if (MASK == 1) {
                    y0 = *reinterpret_cast<const f32x4*>(y + o0);
                    y1 = *reinterpret_cast<const f32x4*>(y + o1);
                }
                if (MASK == 3) {
                    b0 = rmask[o0 >> 2];
                    b1 = rmask[o1 >> 2];
                }
            }
            fold(v0, g0, y0, b0);
            fold(v1, g1, y1, b1);
        }
        if (r < r1) {
            const long long o0 = r * C + cg * 4;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + o0);
            f32x4 g0 = z4, y0 = z4;
            unsigned b0 = 0;
            if (MODE == 1) {
                g0 = *reinterpret_cast<const f32x4*>(dy + o0);
                if (MASK == 1) y0 = *reinterpret_cast<const f32x4*>(y + o0);
                if (MASK == 3) b0 = rmask[o0 >> 2];
            }
            fold(v0, g0, y0, b0);
        }
    }
    red[0][t] = a;
    red[1][t] = b;
    __syncthreads();
    if (pl == 0 && cg * 4 < C) {
        for (int k = 1; k < PL; ++k) {
            a += red[0][t + k * cgb];
            b += red[1][t + k * cgb];
        }
        float* o = part + ((long long)split * C + cg * 4) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[2 * j] = a[j];
            o[2 * j + 1] = b[j];
        }
    }
}

// Sum the S split partials of NCH channels with 256/NCH split-lanes each (fp64): every lane adds its partials (k = sl, sl + NL, ...)
// in order, the lanes of a channel are folded with a fixed xor tree inside the wave and the four waves through LDS -- a fixed
// summation order, so the result is bitwise reproducible.  Returns the two totals for channel c (valid on lanes with sl == 0).
// NCH = 16 for up to 512 splits, 4 above (the conv epilogue supplies thousands of row groups).
// These kernels are pure latency (one small workgroup per 4-16 channels on the critical path of every BatchNorm, ~100 per ResNet50
// step): all loads of a lane are issued eight at a time INCLUDING the tail (clamped index, the value dropped), and no lane walks
// an LDS column serially.
template <int NCH>
__device__ __forceinline__ void reduce_partials(const float* __restrict__ part, int C, int S, int c, int sl, double& s, double& q) {
    constexpr int NL = 256 / NCH;            // split-lanes per channel
    constexpr int LW = 64 / NCH;             // of which in one wave
    __shared__ double red[2][4][NCH];
    double a = 0.0, b = 0.0;
    if (c < C) {
        for (int k = sl; k < S; k += 8 * NL) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ku = k + u * NL;
                v[u] = *reinterpret_cast<const float2*>(part + ((long long)(ku < S ? ku : S - 1) * C + c) * 2);
                if (ku >= S) v[u] = float2{0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a += (double)v[u].x;
                b += (double)v[u].y;
            }
        }
    }
#pragma unroll
    for (int o = NCH; o < 64; o <<= 1) {     // lanes of one channel in this wave: lane = c % NCH + NCH * (sl % LW)
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((sl % LW) == 0) {
        red[0][wave][c % NCH] = a;
        red[1][wave][c % NCH] = b;
    }
    __syncthreads();
    s = 0.0;
    q = 0.0;
    if (sl == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s += red[0][k][c % NCH];
            q += red[1][k][c % NCH];
        }
    }
}

// BN forward finalize: batch mean / biased var, running-stat update (unbiased var), scale/shift.
template <int NCH>
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var, float momentum, float eps,
                                                              float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                              float* __restrict__ scale, float* __restrict__ shift, long long M, int C,
                                                              int S) {
    const int c = blockIdx.x * NCH + (threadIdx.x % NCH), sl = threadIdx.x / NCH;
    double s, q;
    reduce_partials<NCH>(part, C, S, c, sl, s, q);
    if (sl != 0 || c >= C) return;
    const double mu = s / (double)M;
    double var = q / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = (float)mu;
    save_invstd[c] = invstd;
    if (running_mean) {
        const double unb = (M > 1) ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mu * sc;
}

// y = x*scale + shift (+ residual) (ReLU).  scale/shift either precomputed (train) or derived from
// running stats on the fly (eval: scale==nullptr).
// Round 4: the grid-stride loop carried ONE 16-byte load per thread and iteration behind a 64-bit modulo for the channel index and two
// 16-byte scale / shift loads (2.85 TB/s over the ResNet50 step's 52 launches).  Now the stride (grid x 256 threads) is a multiple of
// the channel-group count, so a thread's four channels -- and its scale / shift -- never change, and four independent elements are
// loaded per iteration before any is used.
// Round 5: x (the conv output: its last use of the forward pass) and the residual (the block input: likewise) are read with
// NON-TEMPORAL loads -- what stays in the L2 / Infinity Cache is y, which the next conv reads (+1-2 % on the ResNet50 step; the same for
// x in the backward apply below).
__device__ __forceinline__ void bn_scale_shift(const float* scale, const float* shift, const float* gamma, const float* beta, const float* rmean,
                                               const float* rvar, float eps, int c, f32x4& sc, f32x4& sh) {
    if (scale) {
        sc = *reinterpret_cast<const f32x4*>(scale + c);
        sh = *reinterpret_cast<const f32x4*>(shift + c);
    } else {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
        const f32x4 m = *reinterpret_cast<const f32x4*>(rmean + c), v = *reinterpret_cast<const f32x4*>(rvar + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sc[j] = g[j] / sqrtf(v[j] + eps);
            sh[j] = b[j] - m[j] * sc[j];
        }
    }
}
constexpr int BN_UNROLL = 4;
__global__ __launch_bounds__(TPB) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ rmean,
                                                       const float* __restrict__ rvar, float eps, const float* __restrict__ residual,
                                                       int relu, float* __restrict__ y, long long total4, int C,
                                                       unsigned char* __restrict__ rmask = nullptr, unsigned* __restrict__ ymax = nullptr) {
    const int cg = C / 4;
    float amax = 0.f;          // ymax: the largest |y| this thread writes (the half-piece pointwise conv that reads y scales by it)
    const long long stride = (long long)gridDim.x * TPB;
    const long long i0 = (long long)blockIdx.x * TPB + threadIdx.x;
    auto finish = [&](f32x4 v, const f32x4 sc, const f32x4 sh, const f32x4 res, long long i) {
        v = v * sc + sh;
        if (residual) v += res;
        if (relu) {
            if (rmask) rmask[i] = (unsigned char)((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        amax = fmaxf(fmaxf(fmaxf(amax, fabsf(v[0])), fmaxf(fabsf(v[1]), fabsf(v[2]))), fabsf(v[3]));
        *reinterpret_cast<f32x4*>(y + i * 4) = v;
    };
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    if (stride % cg == 0) {            // (uniform) the launcher's grids: this thread's channels are fixed
        f32x4 sc, sh;
        bn_scale_shift(scale, shift, gamma, beta, rmean, rvar, eps, (int)(i0 % cg) * 4, sc, sh);
        long long i = i0;
        for (; i + (BN_UNROLL - 1) * stride < total4; i += BN_UNROLL * stride) {
            f32x4 v[BN_UNROLL], r[BN_UNROLL];
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + (i + u * stride) * 4));
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) r[u] = residual ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(residual + (i + u * stride) * 4)) : z4;
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) finish(v[u], sc, sh, r[u], i + u * stride);
        }
        for (; i < total4; i += stride)
            finish(*reinterpret_cast<const f32x4*>(x + i * 4), sc, sh, residual ? *reinterpret_cast<const f32x4*>(residual + i * 4) : z4, i);
    } else {
        for (long long i = i0; i < total4; i += stride) {
            f32x4 sc, sh;
            bn_scale_shift(scale, shift, gamma, beta, rmean, rvar, eps, (int)(i % cg) * 4, sc, sh);
            finish(*reinterpret_cast<const f32x4*>(x + i * 4), sc, sh, residual ? *reinterpret_cast<const f32x4*>(residual + i * 4) : z4, i);
        }
    }
    if (ymax != nullptr) ccst_absmax_publish(ymax, amax, blockIdx.x);
}
static int bn_grid(long long total4, int C, int per_thread);

template <int NCH>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ sums, int C, int S,
                                                              int accumulate) {
    const int c = blockIdx.x * NCH + (threadIdx.x % NCH), sl = threadIdx.x / NCH;
    double a, b;
    reduce_partials<NCH>(part, C, S, c, sl, a, b);
    if (sl != 0 || c >= C) return;
    sums[c] = (float)a;          // sum dy'
    sums[C + c] = (float)b;      // sum dy' * xhat
    if (accumulate) {
        dbeta[c] += (float)a;
        dgamma[c] += (float)b;
    } else {
        dbeta[c] = (float)a;
        dgamma[c] = (float)b;
    }
}

// dx = gamma*invstd*(dy' - mean(dy') - xhat*mean(dy'*xhat));  d_residual = dy'
// (as bn_apply_kernel since round 4: the stride is a multiple of the channel-group count, a thread's per-channel constants stay in
//  registers and two elements' loads are issued before either is used)
template <int MASK>
__global__ __launch_bounds__(TPB) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ sums, int relu,
                                                           float* __restrict__ dx, float* __restrict__ dres, long long total4, int C,
                                                           float invM, const unsigned char* __restrict__ rmask = nullptr,
                                                           unsigned* __restrict__ dxmax = nullptr) {
    const int cg = C / 4;
    float amax = 0.f;          // dxmax: the largest |dx| this thread writes (the half-piece weight-gradient / backward-data kernels scale by it)
    const long long stride = (long long)gridDim.x * TPB;
    const long long i0 = (long long)blockIdx.x * TPB + threadIdx.x;
    struct Chan {
        f32x4 mu, is, ga, be, k1, k2;           // k1 = mean(dy'), k2 = mean(dy' * xhat)
    };
    auto chan = [&](int c, Chan& k) {
        k.mu = *reinterpret_cast<const f32x4*>(mean + c);
        k.is = *reinterpret_cast<const f32x4*>(invstd + c);
        k.ga = *reinterpret_cast<const f32x4*>(gamma + c);
        k.be = MASK == 2 ? *reinterpret_cast<const f32x4*>(beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        k.k1 = *reinterpret_cast<const f32x4*>(sums + c) * invM;
        k.k2 = *reinterpret_cast<const f32x4*>(sums + C + c) * invM;
    };
    auto finish = [&](f32x4 g, const f32x4 v, const f32x4 o_in, unsigned bits, const Chan& k, long long i) {
        const f32x4 xh = (v - k.mu) * k.is;
        if (MASK == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = ((bits >> j) & 1u) ? g[j] : 0.f;
        } else if (MASK != 0) {
            const f32x4 o = (MASK == 1) ? o_in : xh * k.ga + k.be;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
        }
        if (dres) *reinterpret_cast<f32x4*>(dres + i * 4) = g;
        const f32x4 d = k.ga * k.is * (g - k.k1 - xh * k.k2);
        amax = fmaxf(fmaxf(fmaxf(amax, fabsf(d[0])), fmaxf(fabsf(d[1]), fabsf(d[2]))), fabsf(d[3]));
        *reinterpret_cast<f32x4*>(dx + i * 4) = d;
    };
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    auto one = [&](long long i, const Chan& k) {
        finish(*reinterpret_cast<const f32x4*>(dy + i * 4), *reinterpret_cast<const f32x4*>(x + i * 4),
               MASK == 1 ? *reinterpret_cast<const f32x4*>(y + i * 4) : z4, MASK == 3 ? (unsigned)rmask[i] : 0u, k, i);
    };
    if (stride % cg == 0) {            // (uniform) the launcher's grids
        Chan k;
        chan((int)(i0 % cg) * 4, k);
        long long i = i0;
        for (; i + stride < total4; i += 2 * stride) {
            const long long i1 = i + stride;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(dy + i * 4), g1 = *reinterpret_cast<const f32x4*>(dy + i1 * 4);
            const f32x4 v0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + i * 4)), v1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + i1 * 4));
            const f32x4 o0 = MASK == 1 ? *reinterpret_cast<const f32x4*>(y + i * 4) : z4, o1 = MASK == 1 ? *reinterpret_cast<const f32x4*>(y + i1 * 4) : z4;
            const unsigned b0 = MASK == 3 ? (unsigned)rmask[i] : 0u, b1 = MASK == 3 ? (unsigned)rmask[i1] : 0u;
            finish(g0, v0, o0, b0, k, i);
            finish(g1, v1, o1, b1, k, i1);
        }
        if (i < total4) one(i, k);
    } else {
        for (long long i = i0; i < total4; i += stride) {
            Chan k;
            chan((int)(i % cg) * 4, k);
            one(i, k);
        }
    }
    if (dxmax != nullptr) ccst_absmax_publish(dxmax, amax, blockIdx.x);
}
// grid of the BatchNorm apply kernels: enough workgroups for per_thread elements per thread, at most 2048, and such that grid x 256 is a
// multiple of the channel-group count (256 is a multiple of every C / 4 <= 256 of the ResNets; an even grid covers C / 4 = 512)
static int bn_grid(long long total4, int C, int per_thread) {
    const int cg = C / 4;
    long long g = (total4 + (long long)TPB * per_thread - 1) / ((long long)TPB * per_thread);
    g = g < 1 ? 1 : (g > 2048 ? 2048 : g);
    if (cg > TPB && cg % TPB == 0) {
        const long long m = cg / TPB;
        g = (g + m - 1) / m * m;
    }
    return (int)g;
}

// ------------------------------------------------------------------------------------------
// The stem's BatchNorm -> ReLU -> MaxPool2d(3, 2, 1) (nets/resnet.py:138-140) without the full-resolution tensors in between (64 x 111 x
// 111 x 64 floats = 201 MB each at B = 64): the forward pools relu(x * scale + shift) straight from the conv output (no normalised
// tensor is written or read back), the backward's two passes gather the pooled gradient through the saved window positions on the
// fly (no full-resolution gradient is written by a pooling backward and read twice).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void bn_relu_maxpool_fwd_kernel(const f32x4* __restrict__ x, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, f32x4* __restrict__ y,
                                                                  unsigned* __restrict__ idx, int N, int H, int W, int C4, int Ho, int Wo,
                                                                  unsigned* __restrict__ ymax) {
    const long long total = (long long)N * Ho * Wo * C4;
    float amax = 0.f;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const unsigned iu = (unsigned)i, j0 = iu / (unsigned)C4, j1 = j0 / (unsigned)Wo, nn = j1 / (unsigned)Ho;      // (total < 2^31: launcher)
        const int c = (int)(iu - j0 * (unsigned)C4), ox = (int)(j0 - j1 * (unsigned)Wo), oy = (int)(j1 - nn * (unsigned)Ho), n = (int)nn;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c * 4), sh = *reinterpret_cast<const f32x4*>(shift + c * 4);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    f32x4 v = x[(((long long)n * H + iy) * W + ix) * C4 + c];
                    v = v * sc + sh;                                  // as bn_apply_kernel
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = fmaxf(v[k], 0.f);
                        if (v[k] > best[k] || (v[k] != v[k])) {       // first maximum wins; NaN propagates like torch
                            best[k] = v[k];
                            bi[k] = ky * 3 + kx;
                        }
                    }
                }
            }
        }
        y[i] = best;
        amax = fmaxf(fmaxf(fmaxf(amax, fabsf(best[0])), fmaxf(fabsf(best[1]), fabsf(best[2]))), fabsf(best[3]));
        idx[i] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
    if (ymax != nullptr) ccst_absmax_publish(ymax, amax, blockIdx.x);
}

// gradient of the pooling INPUT pixel (n, iy, ix), channel group c, gathered from the pooled gradient (as maxpool3s2_bwd_kernel)
__device__ __forceinline__ f32x4 pool_grad_at(const f32x4* __restrict__ dy, const unsigned* __restrict__ idx, int n, int iy, int ix, int c,
                                              int C4, int Ho, int Wo) {
    f32x4 g = {0, 0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int ty = iy + 1 - ky;
        if (ty < 0 || (ty & 1)) continue;
        const int oy = ty >> 1;
        if (oy >= Ho) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int tx = ix + 1 - kx;
            if (tx < 0 || (tx & 1)) continue;
            const int ox = tx >> 1;
            if (ox >= Wo) continue;
            const long long o = (((long long)n * Ho + oy) * Wo + ox) * C4 + c;
            const unsigned id = idx[o];
            const f32x4 d = dy[o];
            const unsigned me = ky * 3 + kx;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (((id >> (8 * k)) & 0xffu) == me) g[k] += d[k];
        }
    }
    return g;
}

// BN-backward partial sums with dy = ReLU-masked pooled gradient gathered on the fly; layout / reduction as chan_partials_kernel
__global__ __launch_bounds__(TPB) void stem_bwd_partials_kernel(const float* __restrict__ x, const f32x4* __restrict__ dyp,
                                                                const unsigned* __restrict__ idx, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ part, long long M, int C,
                                                                int S, int cgb, int PL, int H, int W, int Ho, int Wo) {
    __shared__ f32x4 red[2][TPB];
    const int split = blockIdx.x, t = threadIdx.x;
    const int cgl = t % cgb, pl = t / cgb;
    const int cg = blockIdx.y * cgb + cgl;
    const long long per = (M + S - 1) / S;
    const long long r0 = split * per;
    const long long r1 = (r0 + per < M) ? r0 + per : M;
    f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (pl < PL && cg * 4 < C) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + cg * 4), is = *reinterpret_cast<const f32x4*>(invstd + cg * 4);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + cg * 4), be = *reinterpret_cast<const f32x4*>(beta + cg * 4);
        for (long long r = r0 + pl; r < r1; r += PL) {
            // (32-bit index arithmetic: M < 2^31 is checked by the launcher; the 64-bit divisions were most of this loop's instructions)
            const unsigned ru = (unsigned)r, q = ru / (unsigned)W, n = q / (unsigned)H;
            const int ix = (int)(ru - q * (unsigned)W), iy = (int)(q - n * (unsigned)H);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + cg * 4);
            f32x4 g = pool_grad_at(dyp, idx, (int)n, iy, ix, cg, C / 4, Ho, Wo);
            const f32x4 xh = (v - mu) * is;
            const f32x4 o = xh * ga + be;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
            a += g;
            b += g * xh;
        }
    }
    red[0][t] = a;
    red[1][t] = b;
    __syncthreads();
    if (pl == 0 && cg * 4 < C) {
        for (int k = 1; k < PL; ++k) {
            a += red[0][t + k * cgb];
            b += red[1][t + k * cgb];
        }
        float* o = part + ((long long)split * C + cg * 4) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[2 * j] = a[j];
            o[2 * j + 1] = b[j];
        }
    }
}

__global__ __launch_bounds__(TPB) void stem_bwd_apply_kernel(const float* __restrict__ x, const f32x4* __restrict__ dyp,
                                                             const unsigned* __restrict__ idx, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ sums,
                                                             float* __restrict__ dx, long long total4, int C, float invM, int H, int W, int Ho,
                                                             int Wo, unsigned* __restrict__ dxmax) {
    const int cg = C / 4;
    float amax = 0.f;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
        const unsigned iu = (unsigned)i, r = iu / (unsigned)cg, q = r / (unsigned)W, n = q / (unsigned)H;      // (total4 < 2^31: launcher)
        const int cq = (int)(iu - r * (unsigned)cg), c = cq * 4;
        const int ix = (int)(r - q * (unsigned)W), iy = (int)(q - n * (unsigned)H);
        f32x4 g = pool_grad_at(dyp, idx, (int)n, iy, ix, cq, cg, Ho, Wo);
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 xh = (v - mu) * is;
        const f32x4 o = xh * ga + *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + c), s2 = *reinterpret_cast<const f32x4*>(sums + C + c);
        const f32x4 d = ga * is * (g - s1 * invM - xh * (s2 * invM));
        amax = fmaxf(fmaxf(fmaxf(amax, fabsf(d[0])), fmaxf(fabsf(d[1]), fabsf(d[2]))), fabsf(d[3]));
        *reinterpret_cast<f32x4*>(dx + i * 4) = d;
    }
    if (dxmax != nullptr) ccst_absmax_publish(dxmax, amax, blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// MaxPool2d(3, stride 2, padding 1), NHWC.  idx = position (0..8) of the first maximum in the window.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void maxpool3s2_fwd_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y,
                                                             unsigned* __restrict__ idx, int N, int H, int W, int C4, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int c = (int)(i % C4);
        long long j = i / C4;
        const int ox = (int)(j % Wo);
        j /= Wo;
        const int oy = (int)(j % Ho);
        const int n = (int)(j / Ho);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    const f32x4 v = x[(((long long)n * H + iy) * W + ix) * C4 + c];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (v[k] > best[k] || (v[k] != v[k])) {   // first maximum wins; NaN propagates like torch
                            best[k] = v[k];
                            bi[k] = ky * 3 + kx;
                        }
                }
            }
        }
        y[i] = best;
        idx[i] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
}

__global__ __launch_bounds__(TPB) void maxpool3s2_bwd_kernel(const f32x4* __restrict__ dy, const unsigned* __restrict__ idx,
                                                             f32x4* __restrict__ dx, int N, int H, int W, int C4, int Ho, int Wo) {
    const long long total = (long long)N * H * W * C4;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int c = (int)(i % C4);
        long long j = i / C4;
        const int ix = (int)(j % W);
        j /= W;
        const int iy = (int)(j % H);
        const int n = (int)(j / H);
        f32x4 g = {0, 0, 0, 0};
        // windows (oy,ox) with 2*oy-1+ky == iy  =>  oy = (iy+1-ky)/2 for ky of matching parity
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + 1 - ky;
            if (ty < 0 || (ty & 1)) continue;
            const int oy = ty >> 1;
            if (oy >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + 1 - kx;
                if (tx < 0 || (tx & 1)) continue;
                const int ox = tx >> 1;
                if (ox >= Wo) continue;
                const long long o = (((long long)n * Ho + oy) * Wo + ox) * C4 + c;
                const unsigned id = idx[o];
                const f32x4 d = dy[o];
                const unsigned me = ky * 3 + kx;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (((id >> (8 * k)) & 0xffu) == me) g[k] += d[k];
            }
        }
        dx[i] = g;
    }
}

// AvgPool2d over the whole HW map + flatten: [N,HW,C] -> [N,C]
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int HW, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += x[((long long)n * HW + p) * C + c];
    y[i] = s / (float)HW;
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int HW, int C) {
    const long long total = (long long)N * HW * C;
    const float inv = 1.f / (float)HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int n = (int)(i / ((long long)HW * C));
        dx[i] = dy[n * C + c] * inv;
    }
}

// Linear: one wave per output element (n,o); K-strided dot product + wave reduction.
__global__ __launch_bounds__(TPB) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int N, int K, int O) {
    const int wave = (blockIdx.x * TPB + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= N * O) return;
    const int n = wave / O, o = wave - n * O;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s = fmaf(x[(long long)n * K + k], w[(long long)o * K + k], s);
    s = wave_sum(s);
    if (lane == 0) y[wave] = s + (b ? b[o] : 0.f);
}
// dx[n,k] = sum_o dy[n,o] w[o,k]
__global__ void linear_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int N, int K,
                                     int O) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * K) return;
    const int n = (int)(i / K), k = (int)(i - (long long)n * K);
    float s = 0.f;
    for (int o = 0; o < O; ++o) s = fmaf(dy[n * O + o], w[(long long)o * K + k], s);
    dx[i] = s;
}
// dw[o,k] (+)= sum_n dy[n,o] x[n,k] ;  db[o] (+)= sum_n dy[n,o]
__global__ void linear_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
                                     float* __restrict__ db, int N, int K, int O, int accumulate) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long long)O * K) {
        const int o = (int)(i / K), k = (int)(i - (long long)o * K);
        float s = 0.f;
        for (int n = 0; n < N; ++n) s = fmaf(dy[n * O + o], x[(long long)n * K + k], s);
        dw[i] = accumulate ? dw[i] + s : s;
    }
    if (db && i < O) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dy[n * O + (int)i];
        db[i] = accumulate ? db[i] + s : s;
    }
}

// CrossEntropyLoss(mean): one thread per row, then a single-block fixed-order reduction.
__global__ __launch_bounds__(TPB) void softmax_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                         float* __restrict__ loss, float* __restrict__ dlogits,
                                                         int* __restrict__ correct, int N, int O) {
    __shared__ float sl[TPB];
    __shared__ int sc[TPB];
    float lsum = 0.f;
    int csum = 0;
    for (int n = threadIdx.x; n < N; n += TPB) {
        const float* z = logits + (long long)n * O;
        float m = z[0];
        int am = 0;
        for (int o = 1; o < O; ++o)
            if (z[o] > m) {
                m = z[o];
                am = o;
            }
        float se = 0.f;
        for (int o = 0; o < O; ++o) se += expf(z[o] - m);
        const float lse = m + logf(se);
        const int lab = (int)labels[n];
        lsum += lse - z[lab];
        csum += (am == lab);
        if (dlogits) {
            const float invN = 1.f / (float)N;
            for (int o = 0; o < O; ++o) dlogits[(long long)n * O + o] = (expf(z[o] - lse) - (o == lab ? 1.f : 0.f)) * invN;
        }
    }
    sl[threadIdx.x] = lsum;
    sc[threadIdx.x] = csum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        int c = 0;
        for (int k = 0; k < TPB; ++k) {
            t += sl[k];
            c += sc[k];
        }
        loss[0] = t / (float)N;
        if (correct) correct[0] = c;
    }
}

__global__ __launch_bounds__(TPB) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float lr, long long n) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) {
        f32x4 a = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 b = reinterpret_cast<const f32x4*>(g)[i];
        a = a - b * lr;     // p.add_(g, alpha=-lr), fed_run.py:80
        reinterpret_cast<f32x4*>(p)[i] = a;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) p[i] -= lr * g[i];
}
// communication()'s fedavg branch (fed_run.py:400-414) over K flat arenas in ONE pass: server = sum_k w_k client_k accumulated from zero in
// client order with the arithmetic of the K sgd_kernel launches it replaces (t = t - (-w_k) c_k, one rounding sequence: bit-identical),
// and every client overwritten with it -- K reads + K + 1 writes per element instead of 3 K + K reads and 1 + 2 K writes.
constexpr int FEDAVG_MAXK = 16;
struct FedAvgArgs {
    float* client[FEDAVG_MAXK];
    float negw[FEDAVG_MAXK];
    float* server;
    int K;
};
__global__ __launch_bounds__(TPB) void fedavg_kernel(const FedAvgArgs a, long long n) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < a.K; ++k) t = t - reinterpret_cast<const f32x4*>(a.client[k])[i] * a.negw[k];
        reinterpret_cast<f32x4*>(a.server)[i] = t;
        for (int k = 0; k < a.K; ++k) reinterpret_cast<f32x4*>(a.client[k])[i] = t;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
        float t = 0.f;
        for (int k = 0; k < a.K; ++k) t -= a.negw[k] * a.client[k][i];
        a.server[i] = t;
        for (int k = 0; k < a.K; ++k) a.client[k][i] = t;
    }
}
__global__ __launch_bounds__(TPB) void scale_kernel(float* __restrict__ p, float s, long long n) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) {
        reinterpret_cast<f32x4*>(p)[i] = reinterpret_cast<f32x4*>(p)[i] * s;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) p[i] *= s;
}

// small helpers that keep ATen's own elementwise kernels out of the train step
__global__ __launch_bounds__(TPB) void fill_kernel(float* __restrict__ p, float v, long long n) {
    // head up to the first 16-byte boundary, body as float4, tail
    const long long head = min(n, (long long)(((16 - ((uintptr_t)p & 15)) & 15) >> 2));
    float* q = p + head;
    const long long n4 = (n - head) >> 2;
    const f32x4 vv = {v, v, v, v};
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) reinterpret_cast<f32x4*>(q)[i] = vv;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < head; i += (long long)gridDim.x * TPB) p[i] = v;
    for (long long i = head + (n4 << 2) + (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) p[i] = v;
}
__global__ void add_i64_kernel(long long* __restrict__ p, long long d, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += d;
}
__global__ __launch_bounds__(TPB) void mul_scalar_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ s, long long n) {
    const float sv = s[0];
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) y[i] = x[i] * sv;
}
// stem weight gradient: virtual-pixel form gv[co][kx*4+c][ky] -> OIHW g[co][c][ky][kx] += (nn_ops.StemConvFn)
__global__ void stem_grad_unfold_kernel(const float* __restrict__ gv, float* __restrict__ g, int cout, int kwp, int kh, int kw, int C, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = cout * C * kh * kw;
    if (i >= total) return;
    const int kx = i % kw, ky = (i / kw) % kh, c = (i / (kw * kh)) % C, co = i / (kw * kh * C);
    const float v = gv[((long long)co * kwp * 4 + kx * 4 + c) * kh + ky];
    g[i] = accumulate ? g[i] + v : v;
}

int grid_for(long long total) {
    long long g = (total + TPB - 1) / TPB;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

struct Split {
    int S, cgb, PL, gy;
};
Split pick_split(long long M, int C) {
    Split sp;
    const int cg = C / 4;
    sp.cgb = cg < TPB ? cg : TPB;
    sp.PL = TPB / sp.cgb;
    sp.gy = (cg + sp.cgb - 1) / sp.cgb;
    long long s = 2048 / sp.gy;                        // ~2048 workgroups (8 per CU: these loops are latency chains)
    const long long rows = 4LL * sp.PL > 16 ? 4LL * sp.PL : 16;     // >= 4 rows per row lane and >= 16 rows per split (the partials stay
    const long long smax = (M + rows - 1) / rows;                   //    below 1/8 of the tensor they summarise)
    if (s > smax) s = smax;
    if (s > MAXS) s = MAXS;
    if (s < 1) s = 1;
    sp.S = (int)s;
    return sp;
}

}  // namespace

extern "C" int64_t ccst_bn_workspace_bytes(int64_t M, int C) {
    (void)M;
    return ((int64_t)MAXS * C * 2 + 4LL * C) * 4;     // partials + scale/shift (fwd) or sums (bwd)
}

extern "C" int ccst_bn_train_fwd_mask_f32(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                          float momentum, float eps, const float* residual, int relu, float* y, uint8_t* relu_mask,
                                          float* save_mean, float* save_invstd, int64_t M, int C, const float* stats_in, int stats_groups,
                                          void* ws, int64_t ws_bytes, uint32_t* y_absmax, void* stream) {
    CCST_REQUIRE(!relu_mask || relu, "bn_train_fwd: a ReLU mask needs relu=1");
    CCST_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && ws, "bn_train_fwd: null pointer");
    CCST_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_train_fwd: need M>0 and C %% 4 == 0 (C=%d)", C);
    CCST_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_fwd: running stats must come together");
    if (ws_bytes < ccst_bn_workspace_bytes(M, C)) {
        ccst_set_error("bn_train_fwd: workspace too small");
        return CCST_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const Split sp = pick_split(M, C);
    float* part = (float*)ws;
    float* scale = part + (int64_t)MAXS * C * 2;
    float* shift = scale + C;
    const float* fin = part;
    int S = sp.S;
    if (stats_in != nullptr) {          // per-channel (sum, sum^2) partials already produced by the conv epilogue
        CCST_REQUIRE(stats_groups > 0, "bn_train_fwd: stats_groups must be positive");
        fin = stats_in;
        S = stats_groups;
    } else {
        hipLaunchKernelGGL(chan_partials_kernel<0>, dim3(sp.S, sp.gy), dim3(TPB), 0, st, x, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, 0, part, (long long)M, C, sp.S, sp.cgb, sp.PL);
    }
    if (S > 512)
        hipLaunchKernelGGL(bn_fwd_finalize_kernel<4>, dim3((C + 3) / 4), dim3(256), 0, st, fin, gamma, beta, running_mean, running_var,
                           momentum, eps, save_mean, save_invstd, scale, shift, (long long)M, C, S);
    else
        hipLaunchKernelGGL(bn_fwd_finalize_kernel<16>, dim3((C + 15) / 16), dim3(256), 0, st, fin, gamma, beta, running_mean,
                           running_var, momentum, eps, save_mean, save_invstd, scale, shift, (long long)M, C, S);
    const long long total4 = (long long)M * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_grid(total4, C, BN_UNROLL)), dim3(TPB), 0, st, x, scale, shift, nullptr, nullptr, nullptr, nullptr,
                       eps, residual, relu, y, total4, C, relu_mask, y_absmax);
    return ccst_launch_status("bn_train_fwd");
}

extern "C" int ccst_bn_eval_fwd_f32(const float* x, const float* gamma, const float* beta, const float* running_mean,
                                    const float* running_var, float eps, const float* residual, int relu, float* y, int64_t M, int C,
                                    uint32_t* y_absmax, void* stream) {
    CCST_REQUIRE(x && gamma && beta && running_mean && running_var && y, "bn_eval_fwd: null pointer");
    CCST_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_eval_fwd: need M>0 and C %% 4 == 0");
    const long long total4 = (long long)M * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_grid(total4, C, BN_UNROLL)), dim3(TPB), 0, (hipStream_t)stream, x, nullptr, nullptr, gamma, beta,
                       running_mean, running_var, eps, residual, relu, y, total4, C, nullptr, y_absmax);
    return ccst_launch_status("bn_eval_fwd");
}

extern "C" int ccst_bn_train_bwd_mask_f32(const float* dy, const float* x, const float* y, const uint8_t* relu_mask, const float* gamma,
                                          const float* beta, const float* save_mean, const float* save_invstd, int relu, float* dx,
                                          float* d_residual, float* dgamma, float* dbeta, int accumulate, int64_t M, int C, void* ws,
                                          int64_t ws_bytes, uint32_t* dx_absmax, void* stream) {
    CCST_REQUIRE(dy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta && ws, "bn_train_bwd: null pointer");
    CCST_REQUIRE(!relu || y || relu_mask || beta, "bn_train_bwd: relu=1 needs the forward's mask, the saved output y, or beta to recompute the mask from x");
    CCST_REQUIRE(!(relu && d_residual && !y && !relu_mask), "bn_train_bwd: with a residual the ReLU mask must come from the forward (mask or saved output y)");
    CCST_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_train_bwd: need M>0 and C %% 4 == 0");
    if (ws_bytes < ccst_bn_workspace_bytes(M, C)) {
        ccst_set_error("bn_train_bwd: workspace too small");
        return CCST_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const Split sp = pick_split(M, C);
    float* part = (float*)ws;
    float* sums = part + (int64_t)MAXS * C * 2;
    const int mask = !relu ? 0 : (relu_mask ? 3 : (y ? 1 : 2));
#define CCST_PARTIALS1(MK)                                                                                                          \
    hipLaunchKernelGGL((chan_partials_kernel<1, MK>), dim3(sp.S, sp.gy), dim3(TPB), 0, st, x, dy, y, save_mean, save_invstd, gamma, beta, \
                       relu, part, (long long)M, C, sp.S, sp.cgb, sp.PL, relu_mask)
    if (mask == 0) CCST_PARTIALS1(0);
    else if (mask == 1) CCST_PARTIALS1(1);
    else if (mask == 2) CCST_PARTIALS1(2);
    else CCST_PARTIALS1(3);
#undef CCST_PARTIALS1
    if (sp.S > 512)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<4>, dim3((C + 3) / 4), dim3(256), 0, st, part, dgamma, dbeta, sums, C, sp.S, accumulate);
    else
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<16>, dim3((C + 15) / 16), dim3(256), 0, st, part, dgamma, dbeta, sums, C, sp.S, accumulate);
    const long long total4 = (long long)M * (C / 4);
#define CCST_BWD_APPLY(MK)                                                                                                          \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<MK>), dim3(bn_grid(total4, C, 2)), dim3(TPB), 0, st, dy, x, y, gamma, beta, save_mean, save_invstd, \
                       sums, relu, dx, d_residual, total4, C, 1.f / (float)M, relu_mask, dx_absmax)
    if (mask == 0) CCST_BWD_APPLY(0);
    else if (mask == 1) CCST_BWD_APPLY(1);
    else if (mask == 2) CCST_BWD_APPLY(2);
    else CCST_BWD_APPLY(3);
#undef CCST_BWD_APPLY
    return ccst_launch_status("bn_train_bwd");
}

extern "C" int ccst_bn_train_bwd_partials_f32(const float* dy, const float* x, const float* gamma, const float* save_mean,
                                              const float* save_invstd, const float* partials, int groups, float* dx, float* dgamma,
                                              float* dbeta, int accumulate, int64_t M, int C, void* ws, int64_t ws_bytes,
                                              uint32_t* dx_absmax, void* stream) {
    CCST_REQUIRE(dy && x && gamma && save_mean && save_invstd && partials && dx && dgamma && dbeta && ws, "bn_train_bwd_partials: null pointer");
    CCST_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && groups > 0, "bn_train_bwd_partials: need M>0, groups>0 and C %% 4 == 0");
    if (ws_bytes < ccst_bn_workspace_bytes(M, C)) {
        ccst_set_error("bn_train_bwd_partials: workspace too small");
        return CCST_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* sums = (float*)ws + (int64_t)MAXS * C * 2;
    if (groups > 512)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<4>, dim3((C + 3) / 4), dim3(256), 0, st, partials, dgamma, dbeta, sums, C, groups, accumulate);
    else
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<16>, dim3((C + 15) / 16), dim3(256), 0, st, partials, dgamma, dbeta, sums, C, groups, accumulate);
    const long long total4 = (long long)M * (C / 4);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<0>), dim3(bn_grid(total4, C, 2)), dim3(TPB), 0, st, dy, x, nullptr, gamma, nullptr, save_mean, save_invstd,
                       sums, 0, dx, nullptr, total4, C, 1.f / (float)M, nullptr, dx_absmax);
    return ccst_launch_status("bn_train_bwd_partials");
}

extern "C" int ccst_bn_relu_maxpool_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* running_mean,
                                                  float* running_var, float momentum, float eps, float* y_pooled, uint32_t* idx,
                                                  float* save_mean, float* save_invstd, int N, int H, int W, int C, int Ho, int Wo,
                                                  const float* stats_in, int stats_groups, void* ws, int64_t ws_bytes, uint32_t* y_absmax,
                                                  void* stream) {
    CCST_REQUIRE(x && gamma && beta && y_pooled && idx && save_mean && save_invstd && ws, "bn_relu_maxpool_fwd: null pointer");
    CCST_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "bn_relu_maxpool_fwd: bad extents");
    CCST_REQUIRE(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1, "bn_relu_maxpool_fwd: Ho/Wo must be floor((H+2-3)/2)+1");
    CCST_REQUIRE((long long)N * H * W * (C / 4) < 0x7fffffffLL, "bn_relu_maxpool_fwd: the conv output must have < 2^31 channel quads (32-bit index arithmetic)");
    CCST_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_relu_maxpool_fwd: running stats must come together");
    const int64_t M = (int64_t)N * H * W;
    if (ws_bytes < ccst_bn_workspace_bytes(M, C)) {
        ccst_set_error("bn_relu_maxpool_fwd: workspace too small");
        return CCST_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const Split sp = pick_split(M, C);
    float* part = (float*)ws;
    float* scale = part + (int64_t)MAXS * C * 2;
    float* shift = scale + C;
    const float* fin = part;
    int S = sp.S;
    if (stats_in != nullptr) {
        CCST_REQUIRE(stats_groups > 0, "bn_relu_maxpool_fwd: stats_groups must be positive");
        fin = stats_in;
        S = stats_groups;
    } else {
        hipLaunchKernelGGL(chan_partials_kernel<0>, dim3(sp.S, sp.gy), dim3(TPB), 0, st, x, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, 0, part, (long long)M, C, sp.S, sp.cgb, sp.PL, nullptr);
    }
    if (S > 512)
        hipLaunchKernelGGL(bn_fwd_finalize_kernel<4>, dim3((C + 3) / 4), dim3(256), 0, st, fin, gamma, beta, running_mean, running_var,
                           momentum, eps, save_mean, save_invstd, scale, shift, (long long)M, C, S);
    else
        hipLaunchKernelGGL(bn_fwd_finalize_kernel<16>, dim3((C + 15) / 16), dim3(256), 0, st, fin, gamma, beta, running_mean,
                           running_var, momentum, eps, save_mean, save_invstd, scale, shift, (long long)M, C, S);
    const long long total = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel, dim3(grid_for(total)), dim3(TPB), 0, st, (const f32x4*)x, scale, shift, (f32x4*)y_pooled, idx,
                       N, H, W, C / 4, Ho, Wo, y_absmax);
    return ccst_launch_status("bn_relu_maxpool_fwd");
}

extern "C" int ccst_bn_relu_maxpool_train_bwd_f32(const float* dy_pooled, const uint32_t* idx, const float* x, const float* gamma,
                                                  const float* beta, const float* save_mean, const float* save_invstd, float* dx,
                                                  float* dgamma, float* dbeta, int accumulate, int N, int H, int W, int C, int Ho, int Wo,
                                                  void* ws, int64_t ws_bytes, uint32_t* dx_absmax, void* stream) {
    CCST_REQUIRE(dy_pooled && idx && x && gamma && beta && save_mean && save_invstd && dx && dgamma && dbeta && ws, "bn_relu_maxpool_bwd: null pointer");
    CCST_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1, "bn_relu_maxpool_bwd: bad extents");
    CCST_REQUIRE((long long)N * H * W * (C / 4) < 0x7fffffffLL, "bn_relu_maxpool_bwd: the conv output must have < 2^31 channel quads (32-bit index arithmetic)");
    const int64_t M = (int64_t)N * H * W;
    if (ws_bytes < ccst_bn_workspace_bytes(M, C)) {
        ccst_set_error("bn_relu_maxpool_bwd: workspace too small");
        return CCST_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const Split sp = pick_split(M, C);
    float* part = (float*)ws;
    float* sums = part + (int64_t)MAXS * C * 2;
    hipLaunchKernelGGL(stem_bwd_partials_kernel, dim3(sp.S, sp.gy), dim3(TPB), 0, st, x, (const f32x4*)dy_pooled, idx, save_mean, save_invstd,
                       gamma, beta, part, (long long)M, C, sp.S, sp.cgb, sp.PL, H, W, Ho, Wo);
    if (sp.S > 512)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<4>, dim3((C + 3) / 4), dim3(256), 0, st, part, dgamma, dbeta, sums, C, sp.S, accumulate);
    else
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<16>, dim3((C + 15) / 16), dim3(256), 0, st, part, dgamma, dbeta, sums, C, sp.S, accumulate);
    const long long total4 = (long long)M * (C / 4);
    hipLaunchKernelGGL(stem_bwd_apply_kernel, dim3(grid_for(total4)), dim3(TPB), 0, st, x, (const f32x4*)dy_pooled, idx, gamma, beta, save_mean,
                       save_invstd, sums, dx, total4, C, 1.f / (float)M, H, W, Ho, Wo, dx_absmax);
    return ccst_launch_status("bn_relu_maxpool_bwd");
}

extern "C" int ccst_maxpool3s2_fwd_f32(const float* x, float* y, uint32_t* idx, int N, int H, int W, int C, int Ho, int Wo,
                                       void* stream) {
    CCST_REQUIRE(x && y && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool_fwd: bad args");
    CCST_REQUIRE(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1, "maxpool_fwd: Ho/Wo must be floor((H+2-3)/2)+1");
    const long long total = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, (const f32x4*)x, (f32x4*)y, idx,
                       N, H, W, C / 4, Ho, Wo);
    return ccst_launch_status("maxpool_fwd");
}

extern "C" int ccst_maxpool3s2_bwd_f32(const float* dy, const uint32_t* idx, float* dx, int N, int H, int W, int C, int Ho, int Wo,
                                       void* stream) {
    CCST_REQUIRE(dy && idx && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool_bwd: bad args");
    const long long total = (long long)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, (const f32x4*)dy, idx,
                       (f32x4*)dx, N, H, W, C / 4, Ho, Wo);
    return ccst_launch_status("maxpool_bwd");
}

extern "C" int ccst_avgpool_fwd_f32(const float* x, float* y, int N, int HW, int C, void* stream) {
    CCST_REQUIRE(x && y && N > 0 && HW > 0 && C > 0, "avgpool_fwd: bad args");
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, N, HW, C);
    return ccst_launch_status("avgpool_fwd");
}
extern "C" int ccst_avgpool_bwd_f32(const float* dy, float* dx, int N, int HW, int C, void* stream) {
    CCST_REQUIRE(dy && dx && N > 0 && HW > 0 && C > 0, "avgpool_bwd: bad args");
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((long long)N * HW * C)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, HW, C);
    return ccst_launch_status("avgpool_bwd");
}

extern "C" int ccst_linear_fwd_f32(const float* x, const float* w, const float* b, float* y, int N, int K, int O, void* stream) {
    CCST_REQUIRE(x && w && y && N > 0 && K > 0 && O > 0, "linear_fwd: bad args");
    const long long waves = (long long)N * O;
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((unsigned)((waves * 64 + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)stream, x, w, b, y,
                       N, K, O);
    return ccst_launch_status("linear_fwd");
}
extern "C" int ccst_linear_bwd_f32(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int accumulate,
                                   int N, int K, int O, void* stream) {
    CCST_REQUIRE(x && w && dy && N > 0 && K > 0 && O > 0, "linear_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (dx) hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3((unsigned)(((long long)N * K + 255) / 256)), dim3(256), 0, st, dy, w, dx, N, K, O);
    if (dw) hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3((unsigned)(((long long)O * K + 255) / 256)), dim3(256), 0, st, dy, x, dw, db, N,
                               K, O, accumulate);
    return ccst_launch_status("linear_bwd");
}

extern "C" int ccst_softmax_ce_f32(const float* logits, const int64_t* labels, float* loss, float* dlogits, int32_t* correct, int N,
                                   int O, void* stream) {
    CCST_REQUIRE(logits && labels && loss && N > 0 && O > 0, "softmax_ce: bad args");
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream, logits, (const long long*)labels, loss, dlogits,
                       correct, N, O);
    return ccst_launch_status("softmax_ce");
}

extern "C" int ccst_sgd_f32(float* p, const float* g, float lr, int64_t n, void* stream) {
    CCST_REQUIRE(p && g && n > 0, "sgd: bad args");
    CCST_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0), "sgd: arenas must be 16-byte aligned");
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n / 4 + 1)), dim3(TPB), 0, (hipStream_t)stream, p, g, lr, (long long)n);
    return ccst_launch_status("sgd");
}
extern "C" int ccst_fedavg_f32(float* server, float* const* clients_host, const float* weights_host, int K, int64_t n, void* stream) {
    CCST_REQUIRE(server && clients_host && weights_host && n > 0, "fedavg: bad args");
    CCST_REQUIRE(K >= 1 && K <= FEDAVG_MAXK, "fedavg: 1..16 clients per launch");
    FedAvgArgs a;
    a.server = server;
    a.K = K;
    CCST_REQUIRE((uintptr_t)server % 16 == 0, "fedavg: arenas must be 16-byte aligned");
    for (int k = 0; k < K; ++k) {
        CCST_REQUIRE(clients_host[k] != nullptr && (uintptr_t)clients_host[k] % 16 == 0 && clients_host[k] != server, "fedavg: arenas must be distinct and 16-byte aligned");
        a.client[k] = clients_host[k];
        a.negw[k] = -weights_host[k];
    }
    hipLaunchKernelGGL(fedavg_kernel, dim3(grid_for(n / 4 + 1)), dim3(TPB), 0, (hipStream_t)stream, a, (long long)n);
    return ccst_launch_status("fedavg");
}
extern "C" int ccst_fill_f32(float* p, float value, int64_t n, void* stream) {
    CCST_REQUIRE(p && n > 0 && ((uintptr_t)p % 4 == 0), "fill: bad args");
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n / 4 + 1)), dim3(TPB), 0, (hipStream_t)stream, p, value, (long long)n);
    return ccst_launch_status("fill");
}
extern "C" int ccst_add_i64(int64_t* p, int64_t delta, int n, void* stream) {
    CCST_REQUIRE(p && n > 0, "add_i64: bad args");
    hipLaunchKernelGGL(add_i64_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (long long*)p, (long long)delta, n);
    return ccst_launch_status("add_i64");
}
extern "C" int ccst_mul_scalar_f32(float* y, const float* x, const float* s_dev, int64_t n, void* stream) {
    CCST_REQUIRE(y && x && s_dev && n > 0, "mul_scalar: bad args");
    hipLaunchKernelGGL(mul_scalar_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, y, x, s_dev, (long long)n);
    return ccst_launch_status("mul_scalar");
}
extern "C" int ccst_stem_grad_unfold_f32(const float* gv, float* g_oihw, int cout, int kwp, int kh, int kw, int C, int accumulate, void* stream) {
    CCST_REQUIRE(gv && g_oihw && cout > 0 && kwp >= kw && kh > 0 && kw > 0 && C > 0 && C <= 4, "stem_grad_unfold: bad args");
    const int total = cout * C * kh * kw;
    hipLaunchKernelGGL(stem_grad_unfold_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, gv, g_oihw, cout, kwp, kh, kw, C, accumulate);
    return ccst_launch_status("stem_grad_unfold");
}
extern "C" int ccst_scale_f32(float* p, float s, int64_t n, void* stream) {
    CCST_REQUIRE(p && n > 0, "scale: bad args");
    CCST_REQUIRE((uintptr_t)p % 16 == 0, "scale: arena must be 16-byte aligned");
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n / 4 + 1)), dim3(TPB), 0, (hipStream_t)stream, p, s, (long long)n);
    return ccst_launch_status("scale");
}
