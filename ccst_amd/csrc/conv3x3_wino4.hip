// 3x3 stride-1 "same" convolution as fused Winograd F(4x4, 3x3) on the fp32-input MFMA.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   per 4x4 output tile (6x6 transform domain), summed over input channels:
//   36 multiplies per 16 outputs and input channel = 2.25 per output, against 4 for F(2x2,3x3) (conv3x3_wino.hip) and 9 for the
//   direct form.  Interpolation points 0, +-1, +-2, inf (Lavin & Gray); fp32 throughout.  Over the whole 17-layer AdaIN path the
//   result differs from an fp64 evaluation by 1.2e-4 (F(2x2): 4e-5, direct fp32: 3e-5; measured on the CPU restatement), inside
//   the 1e-3 contract.
//
// Mapping: one workgroup = 16x32 output pixels = 4x8 Winograd tiles = the M dimension (32) of v_mfma_f32_32x32x2_f32, x 32
// output channels, TWELVE waves (768 threads, one workgroup per CU, 3 waves per SIMD):
//   * wave w owns transform row r = w >> 1 and the column triple q in {3*(w&1) .. 3*(w&1)+2}: three 32x32 accumulators
//     [tile][cout] (48 VGPRs) -- 36 positions / 12 waves;
//   * the raw input halo (18x34 pixels x 16 channels per k-step, reflection / zero padding and the nearest-x2 upsample applied
//     by the loader) is staged in LDS, double buffered; a lane (tile = lane & 31, channel half = lane >> 5) reads the <= 4 patch
//     rows its transform row needs for 5 of the 6 patch columns (8 channels: 2 x ds_read_b128 per pixel), forms row r of B^T d
//     with wave-uniform coefficients, then its three columns of (B^T d) B -- the A operands never exist in memory;
//   * LDS image: pixel columns in four planes by (x mod 4), so that a tile step (4 pixels) is one 80-byte pixel pitch (5 slots,
//     odd) and four rows shift by 8 slots: the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots;
//   * the transformed weights U[chunk][36 positions][k half][cout][8] stream L2 -> registers (no two waves share them);
//   * epilogue: each wave folds its three columns into the four output columns ((.)A, partial sums), the A^T(.) combination over
//     the twelve waves goes through LDS (two passes of two output columns in the freed halo buffers); bias, ReLU, optional 2x2
//     ceil-mode max-pool (a 4x4 tile holds four pooling windows), NHWC stores.
#include "common.h"

namespace {

struct Wino4Args {
    const float* x;
    const float* u;
    const float* bias;
    float* y;
    int N, H, W, Hs, Ws, Cin, Cout, CoutPad;      // H,W: conv (= output) extent; Hs,Ws: source extent (H/2,W/2 if ups)
    int reflect, ups, relu;
    long long ysN;
    int ysH, ysW;                                  // output strides (of the pooled tensor when POOL)
    int tilesX, tilesY, tilesN;
};

constexpr int NT4 = 768;                           // threads per workgroup
constexpr int CK4 = 16, PIT4 = CK4 + 4;            // channels per k-step; floats per pixel in LDS (80 B)
constexpr int TH4 = 16, TW4 = 32;                  // output pixels per workgroup: 4 x 8 tiles of 4 x 4
constexpr int HH4 = TH4 + 2, HW4 = TW4 + 2;        // halo 18 x 34
constexpr int PLANE4 = ((HW4 + 3) / 4) * PIT4;     // 9 pixel columns per plane
constexpr int ROWP4 = 4 * PLANE4 + 8;              // 728 floats = 182 slots: 4 rows shift by 8 slots (mod 16)
constexpr int HIMG4 = HH4 * ROWP4;                 // 13104 floats = 52.4 KB per buffer
constexpr int HUNITS4 = HH4 * HW4 * (CK4 / 4), HR4 = (HUNITS4 + NT4 - 1) / NT4;
static_assert(2 * 12 * 32 * 33 <= 2 * HIMG4, "epilogue exchange must fit in the halo buffers");

__device__ __forceinline__ int halo_addr4(int hy, int hx) { return hy * ROWP4 + (hx & 3) * PLANE4 + (hx >> 2) * PIT4; }

__device__ __forceinline__ int reflect4(int i, int n) {
    i = (i < 0) ? -i : i;
    i = (i >= n) ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

template <bool POOL, int HALF>
__device__ __forceinline__ void wino4_body(const Wino4Args& p, float* __restrict__ Hs0, float* __restrict__ Hs1) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);                    // wave 0..11
    const int wr = wv >> 1;                                                     // transform row r
    const int li = lane & 31, lh = lane >> 5;

    int bid = ccst_xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p.tilesN;
    bid /= p.tilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int co0 = tn * 32;
    const int oy0 = ty * TH4, ox0 = tx * TW4;

    // ---- halo load units of this thread ------------------------------------------------------------------
    unsigned hoff[HR4];
    int hdst[HR4];
    bool hok[HR4];
#pragma unroll
    for (int i = 0; i < HR4; ++i) {
        const int u = min(tid + NT4 * i, HUNITS4 - 1);
        const int pix = u >> 2, part = u & 3;
        const int hy = pix / HW4, hx = pix - hy * HW4;
        int gy = oy0 + hy - 1, gx = ox0 + hx - 1;
        bool ok = true;
        if (p.reflect) {
            gy = reflect4(gy, p.H);
            gx = reflect4(gx, p.W);
        } else {
            ok = (gy >= 0) & (gy < p.H) & (gx >= 0) & (gx < p.W);
            gy = min(max(gy, 0), p.H - 1);
            gx = min(max(gx, 0), p.W - 1);
        }
        gy >>= p.ups;
        gx >>= p.ups;
        hok[i] = ok;
        hoff[i] = (unsigned)(((n * p.Hs + gy) * p.Ws + gx) * p.Cin + part * 4);
        hdst[i] = halo_addr4(hy, hx) + part * 4;
    }

    // ---- A side: row r of B^T d = k0*d[i0] + k1*d[i1] + k2*d[i2] + k3*d[i3], wave-uniform --------------------
    //   r0: 4 d0 - 5 d2 + d4      r1: -4 d1 - 4 d2 + d3 + d4     r2: 4 d1 - 4 d2 - d3 + d4
    //   r3: -2 d1 - d2 + 2 d3 + d4    r4: 2 d1 - d2 - 2 d3 + d4      r5: 4 d1 - 5 d3 + d5
    int i0, i1, i2, i3;
    float k0, k1, k2, k3;
    if (wr == 0) { i0 = 0; i1 = 2; i2 = 4; i3 = 4; k0 = 4.f; k1 = -5.f; k2 = 1.f; k3 = 0.f; }
    else if (wr == 1) { i0 = 1; i1 = 2; i2 = 3; i3 = 4; k0 = -4.f; k1 = -4.f; k2 = 1.f; k3 = 1.f; }
    else if (wr == 2) { i0 = 1; i1 = 2; i2 = 3; i3 = 4; k0 = 4.f; k1 = -4.f; k2 = -1.f; k3 = 1.f; }
    else if (wr == 3) { i0 = 1; i1 = 2; i2 = 3; i3 = 4; k0 = -2.f; k1 = -1.f; k2 = 2.f; k3 = 1.f; }
    else if (wr == 4) { i0 = 1; i1 = 2; i2 = 3; i3 = 4; k0 = 2.f; k1 = -1.f; k2 = -2.f; k3 = 1.f; }
    else { i0 = 1; i1 = 3; i2 = 5; i3 = 5; k0 = 4.f; k1 = -5.f; k2 = 1.f; k3 = 0.f; }
    const bool four = (wr >= 1) && (wr <= 4);
    const int tyy = li >> 3, txx = li & 7;
    const int abase = halo_addr4(4 * tyy, 4 * txx) + lh * 8;     // patch (row a, column c): + a*ROWP4 + (c&3)*PLANE4 + (c>>2)*PIT4
    const int o0 = i0 * ROWP4, o1 = i1 * ROWP4, o2 = i2 * ROWP4, o3 = i3 * ROWP4;

    // ---- B side: U[chunk][pos = r*6+q][k half][cout_pad][8] ---------------------------------------------------
    const int nchunks = p.Cin / CK4;
    const long long uq = (long long)2 * p.CoutPad * 8;                          // floats per (chunk, position)
    const float* ub = p.u + ((long long)(wr * 6 + 3 * HALF)) * uq + ((long long)lh * p.CoutPad + co0 + li) * 8;
    const long long uchunk = 36 * uq;

    f32x16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    f32x4 rh[HR4];
    f32x4 bq[3][2];                 // [q][half of the lane's 8 k-values]
    auto load_b_half = [&](int c, int h) {
        const float* uc = ub + (long long)c * uchunk + h * 4;
#pragma unroll
        for (int q = 0; q < 3; ++q) bq[q][h] = *reinterpret_cast<const f32x4*>(uc + q * uq);
    };
    auto load_h = [&](int c) {
#pragma unroll
        for (int i = 0; i < HR4; ++i) rh[i] = *reinterpret_cast<const f32x4*>(p.x + hoff[i] + c * CK4);
    };
    auto store_h = [&](float* __restrict__ dst) {
#pragma unroll
        for (int i = 0; i < HR4; ++i) {
            if (tid + NT4 * i < HUNITS4) {
                f32x4 v = rh[i];
                if (!hok[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(dst + hdst[i]) = v;
            }
        }
    };

    // channels h*4 .. h*4+3 of the lane's 8: five patch columns -> row r of B^T d -> three columns of (B^T d) B -> 12 MFMAs
    auto half_step = [&](const float* __restrict__ hs, int h) {
        const float* ha = hs + abase + h * 4;
        f32x4 w[5];
#pragma unroll
        for (int cc = 0; cc < 5; ++cc) {
            const int c = cc + HALF;                                           // HALF 0: columns 0..4, HALF 1: columns 1..5
            const float* hp = ha + (c & 3) * PLANE4 + (c >> 2) * PIT4;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(hp + o0);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(hp + o1);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(hp + o2);
            f32x4 t = a0 * k0 + a1 * k1 + a2 * k2;
            if (four) t += *reinterpret_cast<const f32x4*>(hp + o3) * k3;
            w[cc] = t;
        }
        f32x4 v[3];
        if (HALF == 0) {        // w[] = W0..W4:  q0 = 4 W0 - 5 W2 + W4;  q1 = -4 (W1 + W2) + W3 + W4;  q2 = 4 (W1 - W2) - W3 + W4
            v[0] = w[0] * 4.f - w[2] * 5.f + w[4];
            v[1] = (w[3] + w[4]) - (w[1] + w[2]) * 4.f;
            v[2] = (w[1] - w[2]) * 4.f + (w[4] - w[3]);
        } else {                // w[] = W1..W5:  q3 = 2 (W3 - W1) + W4 - W2;  q4 = 2 (W1 - W3) + W4 - W2;  q5 = 4 W1 - 5 W3 + W5
            const f32x4 d31 = w[2] - w[0], d42 = w[3] - w[1];
            v[0] = d31 * 2.f + d42;
            v[1] = d42 - d31 * 2.f;
            v[2] = w[0] * 4.f - w[2] * 5.f + w[4];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q][j], bq[q][h][j], acc[q], 0, 0, 0);
    };

    // ---- prologue ------------------------------------------------------------------------------------------
    load_b_half(0, 0);
    load_b_half(0, 1);
    load_h(0);
    store_h(Hs0);
    __syncthreads();

    // ---- main loop: one 16-channel halo chunk per barrier ------------------------------------------------------
    for (int c = 0; c < nchunks; ++c) {
        const float* cur = (c & 1) ? Hs1 : Hs0;
        float* nxt = (c & 1) ? Hs0 : Hs1;
        const int cn = min(c + 1, nchunks - 1);
        load_h(cn);
        half_step(cur, 0);
        load_b_half(cn, 0);
        half_step(cur, 1);
        load_b_half(cn, 1);
        store_h(nxt);
        __syncthreads();
    }

    // ---- epilogue: (.)A partial sums locally, A^T(.) over the twelve waves through LDS -----------------------------
    // A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]
    float* ex = Hs0;                                          // [jj][12 waves][32 tiles][33] floats (Hs0 and Hs1 are contiguous)
    float yv[2][4][4];                                        // [tile of this lane][out row i][out col j]
    const int t0 = 4 * wv + 2 * lh;                           // waves 0..7 finalise tiles t0, t0+1 of channel co0+li
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        if (jp == 1) __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * jp + jj;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pv;
                if (HALF == 0) {
                    pv = (j == 0) ? acc[0][r] + acc[1][r] + acc[2][r] : (j == 2) ? acc[1][r] + acc[2][r] : acc[1][r] - acc[2][r];
                } else {
                    pv = (j == 0) ? acc[0][r] + acc[1][r] : (j == 1) ? 2.f * (acc[0][r] - acc[1][r])
                       : (j == 2) ? 4.f * (acc[0][r] + acc[1][r]) : 8.f * (acc[0][r] - acc[1][r]) + acc[2][r];
                }
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                ex[((jj * 12 + wv) * 32 + row) * 33 + li] = pv;
            }
        }
        __syncthreads();
        if (wv < 8) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    float s[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r)
                        s[r] = ex[((jj * 12 + 2 * r) * 32 + t0 + kk) * 33 + li] + ex[((jj * 12 + 2 * r + 1) * 32 + t0 + kk) * 33 + li];
                    const int j = 2 * jp + jj;
                    yv[kk][0][j] = s[0] + s[1] + s[2] + s[3] + s[4];
                    yv[kk][1][j] = (s[1] - s[2]) + 2.f * (s[3] - s[4]);
                    yv[kk][2][j] = (s[1] + s[2]) + 4.f * (s[3] + s[4]);
                    yv[kk][3][j] = (s[1] - s[2]) + 8.f * (s[3] - s[4]) + s[5];
                }
        }
    }
    if (wv >= 8) return;
    const int co = co0 + li;
    if (co >= p.Cout) return;
    const float bias = (p.bias != nullptr) ? p.bias[co] : 0.f;
    const bool relu = p.relu != 0;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int t = t0 + kk;
        const int oy = oy0 + 4 * (t >> 3), ox = ox0 + 4 * (t & 7);
        if (!POOL) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (oy + i < p.H && ox + j < p.W) {
                        float v = yv[kk][i][j] + bias;
                        if (relu) v = fmaxf(v, 0.f);
                        p.y[(long long)n * p.ysN + (long long)(oy + i) * p.ysH + (long long)(ox + j) * p.ysW + co] = v;
                    }
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int py = oy + 2 * a, px = ox + 2 * b;
                    if (py < p.H && px < p.W) {                // ceil mode: a window at the edge holds 1 or 2 valid pixels
                        float v = yv[kk][2 * a][2 * b];
                        if (px + 1 < p.W) v = fmaxf(v, yv[kk][2 * a][2 * b + 1]);
                        if (py + 1 < p.H) {
                            v = fmaxf(v, yv[kk][2 * a + 1][2 * b]);
                            if (px + 1 < p.W) v = fmaxf(v, yv[kk][2 * a + 1][2 * b + 1]);
                        }
                        v += bias;
                        if (relu) v = fmaxf(v, 0.f);
                        p.y[(long long)n * p.ysN + (long long)(py >> 1) * p.ysH + (long long)(px >> 1) * p.ysW + co] = v;
                    }
                }
        }
    }
}

template <bool POOL>
__global__ __launch_bounds__(NT4) void conv3x3_wino4_kernel(const Wino4Args p) {
    extern __shared__ __attribute__((aligned(16))) float wino4_lds[];          // 2 x 52.4 KB halo buffers; reused by the epilogue
    float* Hs0 = wino4_lds;
    float* Hs1 = wino4_lds + HIMG4;
    if ((threadIdx.x >> 6) & 1) wino4_body<POOL, 1>(p, Hs0, Hs1);              // wave-uniform: odd waves own columns 3..5
    else wino4_body<POOL, 0>(p, Hs0, Hs1);
}

// OIHW 3x3 -> U[chunk][pos = r*6+q][k half][cout_pad][8], U = G g G^T (accumulated in double), ci = chunk*16 + half*8 + j
__global__ void pack_weight_wino4_kernel(const float* __restrict__ w, float* __restrict__ u, int cout, int cin, int cin_pad, int cout_pad) {
    const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const long long total = (long long)(cin_pad / 16) * 36 * 2 * cout_pad * 8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        long long t = i >> 3;
        const int co = (int)(t % cout_pad);
        t /= cout_pad;
        const int half = (int)(t & 1);
        t >>= 1;
        const int pos = (int)(t % 36);
        const int chunk = (int)(t / 36);
        const int r = pos / 6, q = pos - 6 * r;
        const int ci = chunk * 16 + half * 8 + j;
        double val = 0.0;
        if (co < cout && ci < cin) {
            const float* g = w + ((long long)co * cin + ci) * 9;
            for (int cc = 0; cc < 3; ++cc) {
                const double gg = G[r][0] * (double)g[0 * 3 + cc] + G[r][1] * (double)g[1 * 3 + cc] + G[r][2] * (double)g[2 * 3 + cc];
                val += gg * G[q][cc];
            }
        }
        u[i] = (float)val;
    }
}

}  // namespace

extern "C" int64_t ccst_wino4_weight_floats(int cin, int cout_pad) { return (int64_t)((cin + 15) / 16) * 36 * 2 * cout_pad * 8; }

extern "C" int ccst_pack_conv_weight_wino4_f32(const float* w_oihw, float* u, int cout, int cin, int cout_pad, void* stream) {
    CCST_REQUIRE(w_oihw && u && cout > 0 && cin > 0, "pack_wino4: bad args");
    CCST_REQUIRE(cout_pad >= cout && cout_pad % 32 == 0, "pack_wino4: cout_pad must be a multiple of 32 >= cout");
    const int cin_pad = (cin + 15) / 16 * 16;
    const long long total = ccst_wino4_weight_floats(cin, cout_pad);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_weight_wino4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, u, cout, cin, cin_pad, cout_pad);
    return ccst_launch_status("pack_weight_wino4");
}

// x: NHWC source [N,Hs,Ws,Cin] (Hs = H/2 if CCST_CONV_UPS2), u: ccst_pack_conv_weight_wino4_f32 output, y: NHWC [N,H,W,Cout] or
// its 2x2 ceil-pooled form.  flags: CCST_CONV_RELU | POOL2 | UPS2 | REFLECT.
extern "C" int ccst_conv3x3_wino4_f32(const float* x, const float* u_packed, const float* bias, float* y, int N, int H, int W, int Cin,
                                      int Cout, int cout_pad, uint32_t flags, void* stream) {
    CCST_REQUIRE(x && u_packed && y, "conv3x3_wino4: null pointer");
    CCST_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0 && Cout > 0, "conv3x3_wino4: bad shape");
    CCST_REQUIRE(cout_pad >= Cout && cout_pad % 32 == 0, "conv3x3_wino4: cout_pad must be a multiple of 32 >= cout");
    CCST_REQUIRE(!(flags & ~(CCST_CONV_RELU | CCST_CONV_POOL2 | CCST_CONV_UPS2 | CCST_CONV_REFLECT)), "conv3x3_wino4: unsupported flag");
    const bool pool = (flags & CCST_CONV_POOL2) != 0, ups = (flags & CCST_CONV_UPS2) != 0;
    if (ups) CCST_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv3x3_wino4: upsampled extent must be even");
    if (flags & CCST_CONV_REFLECT) CCST_REQUIRE(H >= 2 && W >= 2, "conv3x3_wino4: reflection needs extent >= 2");
    Wino4Args a;
    a.x = x; a.u = u_packed; a.bias = bias; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Hs = ups ? H / 2 : H; a.Ws = ups ? W / 2 : W; a.Cin = Cin; a.Cout = Cout; a.CoutPad = cout_pad;
    a.reflect = (flags & CCST_CONV_REFLECT) ? 1 : 0; a.ups = ups ? 1 : 0; a.relu = (flags & CCST_CONV_RELU) ? 1 : 0;
    CCST_REQUIRE((long long)N * a.Hs * a.Ws * Cin < 0x7fffffffLL, "conv3x3_wino4: input must have < 2^31 elements");
    const int oh = pool ? (H + 1) / 2 : H, ow = pool ? (W + 1) / 2 : W;
    a.ysW = Cout; a.ysH = ow * Cout; a.ysN = (long long)oh * ow * Cout;
    a.tilesN = (Cout + 31) / 32;
    a.tilesY = (H + TH4 - 1) / TH4;
    a.tilesX = (W + TW4 - 1) / TW4;
    const long long grid = (long long)N * a.tilesY * a.tilesX * a.tilesN;
    if (grid <= 0 || grid > 0x7fffffffLL) {
        ccst_set_error("conv3x3_wino4: bad grid %lld", grid);
        return CCST_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)2 * HIMG4 * sizeof(float);                      // 104.8 KB: above the 64 KB default, one workgroup per CU
    static bool attr_set = false;                                              // immutable after the first call (idempotent)
    if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            ccst_set_error("conv3x3_wino4: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e1 != hipSuccess ? e1 : e2));
            return (int)(e1 != hipSuccess ? e1 : e2);
        }
        attr_set = true;
    }
    if (pool) hipLaunchKernelGGL(conv3x3_wino4_kernel<true>, dim3((unsigned)grid), dim3(NT4), lds, s, a);
    else hipLaunchKernelGGL(conv3x3_wino4_kernel<false>, dim3((unsigned)grid), dim3(NT4), lds, s, a);
    return ccst_launch_status("conv3x3_wino4");
}
