// 3x3 stride-1 convolution with at most THREE output channels writing NCHW -- the decoder's image edge (net.py:34-35,
// ReflectionPad2d + Conv2d(64, 3, 3x3)) -- as a 1x1 convolution to 9 * Cout "tap planes" followed by a shifted sum (round 4).
//
//     y[co][p] = b[co] + sum_tap z[tap * Cout + co][p + tap],      z[j][q] = sum_c x[q][c] w[tap(j)][co(j)][c]
//
// Why this form.  The layer is 0.9 GFLOP per image against 67 MB of input: HBM-bound at 13 FLOP/B, and padding Cout = 3 to an MFMA tile
// wastes 5-10x the math -- so the first kernel (a direct packed-fp32 VALU kernel, retired in round 6) measured 129 us for 403 MB (693 MB fetched: it
// stages the halo 16 channels at a time, so every 128-byte line of the input is fetched in two passes that L2 does not hold together).
// z has 27 rows for Cout = 3: padded to 32 it IS an MFMA shape with 16 % waste (v_mfma_f32_16x16x32_f16: M = 16 tap planes, N = 16
// pixels, K = 32 channels), z of a pixel needs no neighbours -- every pixel's 256-byte record is read ONCE, whole, straight into the B
// operand's registers -- and the 3x3 neighbourhood is nine shifted fp32 adds on the z planes in LDS.
//   * tile = 8 x 32 output pixels, four waves, FOUR PERSISTENT workgroups per CU (39 KB of LDS, <= 128 registers) walking the tiles of
//     their XCD's eighth of the image; z is computed for the tile's 10 x 34 halo (the ring's records are L2 hits of the neighbouring
//     tiles: 440 MB fetched for 403 MB of input, PMC) in 22 groups of 16 pixels, wave w takes groups w, w + 4, ... with two groups of
//     loads in flight ACROSS tiles: the next tile's first groups are requested before this tile's planes are summed;
//   * fp32 products as three half-piece products (x = hi + lo as IEEE half after a power-of-two scale from the tensor's |max| words,
//     common.h: range-safe at any fp32 magnitude), fp32 accumulate; the weights come pre-split in fragment order (27 x Cin: 8 KB);
//   * z planes in LDS [27 + 1][356] (the planes past 9 Cout of the padded M tile land in a plane nobody reads, so the stores carry no
//     predicate; every address of the loop is an instruction immediate), then each thread sums 9 taps x Cout for one pixel: lanes
//     along x, 128-byte NCHW row stores; the power-of-two scale comes off after the sum.
//   Measured (B=6, 512x512, 64 -> 3): 93-96 us against 133-137 for that VALU kernel on the same boxes; a plain linear read of the same
//   403 MB (absmax_kernel) takes 81 us there, a fill 60.  History / timing experiments: one workgroup per tile 98-101 us (2 / 3 / 4
//   groups of loads in flight 99 / 100 / 98; 16-row tiles at two workgroups per CU 115; without the MFMAs and the splits 94; without
//   the shifted sum 97; without both and with every load instruction 1 KB contiguous (wrong data) 90); persistent workgroups -5 us,
//   three groups in flight at three or four waves per SIMD +-0.
#include "common.h"
#include <stdint.h>

namespace {

typedef _Float16 f16x8z __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2z __attribute__((ext_vector_type(2)));
typedef float f32x2z __attribute__((ext_vector_type(2)));

#define Z_TH 8
#define Z_WGS 4                                         // workgroups per CU the launch bounds ask for
constexpr int Z_TW = 32, Z_HH = Z_TH + 2, Z_HW = Z_TW + 2;
constexpr int Z_PIX = Z_HH * Z_HW;                      // 340 halo pixels (612 for 16 rows)
constexpr int Z_GROUPS = (Z_PIX + 15) / 16;             // 22 groups of 16 pixels
constexpr int Z_GPW = (Z_GROUPS + 3) / 4;               // groups per wave (6)
#define Z_DEPTH 2                                       // groups of loads in flight per wave
constexpr int Z_PLANES = 27;
constexpr int Z_PITCH = Z_GROUPS * 16 + 4;              // words per z plane: the 16-pixel groups whole (the last one runs past the halo) + 4, so that the pitch is 4 modulo 8
static_assert(Z_PITCH % 8 == 4 && Z_PITCH - Z_PIX >= 16, "z plane pitch");

struct ZArgs {
    const float* x;
    const unsigned* xmax;
    const float* wp;          // ccst_pack_conv_weight_zform_f32
    const unsigned* wmax;
    const float* bias;
    float* y;
    int N, H, W, Cin, Cout;
    int reflect, relu;
    int tilesX, tilesY;
};

__device__ __forceinline__ int reflect_z(int i, int n) {
    i = (i < 0) ? -i : i;
    i = (i >= n) ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

// channel of k index (kg, i) of 32-channel k-step ks: a lane's eight k values are two runs of four channels 16 apart, so that one
// 16-byte load instruction covers 64 contiguous bytes of a pixel across the four lane groups (the K order is free: the weights are
// packed with the same map)
__host__ __device__ __forceinline__ int zform_channel(int ks, int kg, int i) { return ks * 32 + (i < 4 ? 4 * kg + i : 16 + 4 * kg + (i - 4)); }

// eight fp32 values scaled by s -> (hi, lo) half pieces
__device__ __forceinline__ void split8z(const f32x4 a, const f32x4 b, float s, f16x8z& hi, f16x8z& lo) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const f32x2z v = (h < 2 ? f32x2z{a[2 * h], a[2 * h + 1]} : f32x2z{b[2 * h - 4], b[2 * h - 3]}) * s;
        unsigned wh, wl;
        ccst_split2_half(v[0], v[1], wh, wl);
        const f16x2z ph = __builtin_bit_cast(f16x2z, wh), pl = __builtin_bit_cast(f16x2z, wl);
        hi[2 * h] = ph[0];
        hi[2 * h + 1] = ph[1];
        lo[2 * h] = pl[0];
        lo[2 * h + 1] = pl[1];
    }
}

template <int NKS, bool REFLECT>          // 32-channel k-steps: Cin = 32 NKS
__global__ __launch_bounds__(256, Z_WGS) void conv3x3_zform_kernel(const ZArgs p) {
    static_assert(Z_GPW % Z_DEPTH == 0, "the register ring keeps its phase from tile to tile");
    __shared__ float Z[(Z_PLANES + 1) * Z_PITCH];          // + a plane nobody reads
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pn = lane & 15, kg = lane >> 4;

    // PERSISTENT workgroups: a workgroup walks a sequence of tiles and its loads simply run on -- the first groups of the next tile
    // are in flight while this tile's planes are summed and stored (as separate workgroups, each tile paid its address arithmetic,
    // the first loads' latency and its shifted sum with nothing in flight: the tile walk alone read at 4.5 TB/s).  The tiles an XCD
    // runs side by side (hardware XCD = blockIdx % 8) are neighbours: each XCD owns a contiguous eighth of the tile list.
    const int ntiles = p.N * p.tilesY * p.tilesX;
    const bool by_xcd = ((int)gridDim.x & 7) == 0;
    const int chunk = by_xcd ? (ntiles + 7) >> 3 : ntiles;
    const int first = by_xcd ? ((int)blockIdx.x & 7) * chunk : 0;
    const int slot0 = by_xcd ? (int)blockIdx.x >> 3 : (int)blockIdx.x, per = by_xcd ? (int)gridDim.x >> 3 : (int)gridDim.x;
    auto tile_of = [&](int k) {                 // k-th tile of this workgroup, or -1
        const int idx = slot0 + k * per;
        return (idx < chunk && first + idx < ntiles) ? first + idx : -1;
    };

    unsigned wword = ccst_absmax_load(p.wmax);

    // a tile's pixel groups for this wave: float offsets of the lane's pixel (+ its first channel run), and whether the pixel is inside
    // the image (zero padding: an outside pixel has z = 0).  Waves 2 and 3 own five groups; their sixth is a copy of the tile's last
    // pixel, loaded (the ring keeps its phase) and not stored.
    struct Tile {
        int n, oy0, ox0;
        unsigned poff[Z_GPW], okbits;
    };
    auto offsets = [&](int t, Tile& T) {
        const int bx = t % p.tilesX, r = t / p.tilesX;
        const int by = r % p.tilesY;
        T.n = r / p.tilesY;
        T.oy0 = by * Z_TH;
        T.ox0 = bx * Z_TW;
        T.okbits = 0u;
        const bool inside = T.oy0 >= 1 && T.ox0 >= 1 && T.oy0 + Z_TH + 1 <= p.H && T.ox0 + Z_TW + 1 <= p.W;      // the whole halo is inside the image (uniform)
#pragma unroll
        for (int i = 0; i < Z_GPW; ++i) {
            const int P = min((wave + 4 * i) * 16 + pn, Z_PIX - 1);
            const int hy = P / Z_HW, hx = P - hy * Z_HW;
            int gy = T.oy0 + hy - 1, gx = T.ox0 + hx - 1;
            bool ok = true;
            if (!inside) {
                if (REFLECT) {
                    gy = reflect_z(gy, p.H);
                    gx = reflect_z(gx, p.W);
                } else {
                    ok = (gy >= 0) & (gy < p.H) & (gx >= 0) & (gx < p.W);
                    gy = min(max(gy, 0), p.H - 1);
                    gx = min(max(gx, 0), p.W - 1);
                }
            }
            T.okbits |= (ok ? 1u : 0u) << i;
            T.poff[i] = (((unsigned)T.n * (unsigned)p.H + (unsigned)gy) * (unsigned)p.W + (unsigned)gx) * (unsigned)p.Cin + 4u * kg;      // < 2^32 elements (checked by the launcher)
        }
    };
    f32x4 st[Z_DEPTH][2 * NKS];
    auto load_group = [&](int d, unsigned off) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int h = 0; h < 2; ++h) st[d][2 * ks + h] = *reinterpret_cast<const f32x4*>(p.x + off + ks * 32 + h * 16);
    };
    int t = tile_of(0);
    // (the per-XCD ranges are ceil(ntiles / 8) long, so the last XCD's workgroups can be left without a tile -- ntiles = 1025 on 256
    //  CUs: slots 122..127 of XCD 7; nothing below may run on tile -1.  Workgroup-uniform, before any barrier.)
    if (t < 0) return;
    Tile cur, nxt;
    offsets(t, cur);
#pragma unroll
    for (int d = 0; d < Z_DEPTH; ++d) load_group(d, cur.poff[d]);

    // the weight fragments (A operand: M = tap plane), [plane tile][k-step][piece]
    f16x8z wa[2][NKS][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
                wa[jt][ks][pc] = __builtin_bit_cast(f16x8z, *reinterpret_cast<const f32x4*>(p.wp + ((((jt * NKS + ks) * 2 + pc) * 64 + lane) * 4)));

    // x's scale is PER IMAGE (p.xmax: [N][CCST_ABSMAX_WORDS]): read when the walk enters an image (a workgroup's tiles are consecutive
    // tiles of its XCD's range: the image changes at most a couple of times per workgroup)
    const int kw = ccst_scale_exp(ccst_absmax_reduce(wword), CCST_SPLIT_W_TARGET);
    int scale_n = -1, kd = 0;
    float xs = 1.f;
    const int nplanes = 9 * p.Cout;
    // accumulator register r of tile jt of a lane: plane 16 jt + 4 kg + r of pixel 16 g + pn; the planes past 9 Cout go to a
    // plane of their own (never read), so that the stores below carry no predicate
    int zaddr[2][4];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = jt * 16 + 4 * kg + r;
            zaddr[jt][r] = min(j, j < nplanes ? j : Z_PLANES) * Z_PITCH + pn + wave * 16;      // + 64 i: an instruction immediate
        }
    float bias3[3];          // (loaded once: a load inside the tile loop would sit behind the prefetches in the in-order vmcnt queue)
#pragma unroll
    for (int co = 0; co < 3; ++co) bias3[co] = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
    constexpr int RPT = Z_TH / 8;          // output rows per thread
    const int tx = tid & 31, ty = (tid >> 5) * RPT;

    for (int k = 0; t >= 0; ++k) {
        if (cur.n != scale_n) {                 // (workgroup-uniform)
            scale_n = cur.n;
            const int kx = ccst_scale_exp(ccst_absmax_read(p.xmax + scale_n * CCST_ABSMAX_WORDS), CCST_SPLIT_X_TARGET);
            xs = __uint_as_float((unsigned)(127 + kx) << 23);
            kd = -(kx + kw);
        }
        const int tn = tile_of(k + 1);
        offsets(tn >= 0 ? tn : t, nxt);         // (past the end the prefetch re-reads this tile: unconditional loads, exact wait counts)
#pragma unroll
        for (int i = 0; i < Z_GPW; ++i) {
            const int d = i % Z_DEPTH;
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            const bool ok = REFLECT || ((cur.okbits >> i) & 1u);
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                f16x8z bhi, blo;
                split8z(ok ? st[d][2 * ks] : zero4, ok ? st[d][2 * ks + 1] : zero4, xs, bhi, blo);
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    acc[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[jt][ks][1], bhi, acc[jt], 0, 0, 0);
                    acc[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[jt][ks][0], blo, acc[jt], 0, 0, 0);
                    acc[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[jt][ks][0], bhi, acc[jt], 0, 0, 0);
                }
            }
            load_group(d, i + Z_DEPTH < Z_GPW ? cur.poff[i + Z_DEPTH] : nxt.poff[i + Z_DEPTH - Z_GPW]);
            // (still scaled by 2^(kx + kw): scaled back after the shifted sum)
            if (wave + 4 * i < Z_GROUPS) {          // wave-uniform
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Z[zaddr[jt][r] + i * 64] = acc[jt][r];
            }
        }
        __syncthreads();

        // ---- the shifted sum: thread -> column tx, rows ty .. of the tile -----------------------------------------------------------
        const int ox = cur.ox0 + tx;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int oy = cur.oy0 + ty + r;
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                if (co < p.Cout) {
                    float v = 0.f;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) v += Z[(tap * p.Cout + co) * Z_PITCH + (ty + r + tap / 3) * Z_HW + tx + tap % 3];
                    v = __builtin_ldexpf(v, kd) + bias3[co];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (ox < p.W && oy < p.H) p.y[(((long long)cur.n * p.Cout + co) * p.H + oy) * p.W + ox] = v;
                }
            }
        }
        __syncthreads();          // the planes are free for the next tile
        cur = nxt;
        t = tn;
    }
}

// w [3][3][Cout][Cin] fp32 -> the A-operand fragments [plane tile 2][k-step Cin / 32][piece 2][lane 64][8 halves], scaled by the power
// of two the weight's |max| words give
__global__ void pack_weight_zform_kernel(const float* __restrict__ w, const unsigned* __restrict__ wmax, float* __restrict__ out, int Cin,
                                         int Cout) {
    const int nks = Cin / 32;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * nks * 64) return;
    const int lane = t & 63, ks = (t >> 6) % nks, jt = (t >> 6) / nks;
    const int kw = ccst_scale_exp(ccst_absmax_read(wmax), CCST_SPLIT_W_TARGET);
    const float s = __uint_as_float((unsigned)(127 + kw) << 23);
    const int j = jt * 16 + (lane & 15), kg = lane >> 4;
    f16x8z hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = 0.f;
        if (j < 9 * Cout) v = w[((long long)(j / Cout) * Cout + (j % Cout)) * Cin + zform_channel(ks, kg, i)] * s;
        const _Float16 h = (_Float16)v;
        hi[i] = h;
        lo[i] = (_Float16)(v - (float)h);
    }
    f32x4* o = reinterpret_cast<f32x4*>(out);
    o[((jt * nks + ks) * 2 + 0) * 64 + lane] = __builtin_bit_cast(f32x4, hi);
    o[((jt * nks + ks) * 2 + 1) * 64 + lane] = __builtin_bit_cast(f32x4, lo);
}

}  // namespace

// Floats of the packed weight of ccst_conv3x3_zform_f32.
extern "C" int64_t ccst_conv3x3_zform_weight_floats(int Cin) { return Cin % 32 == 0 && Cin > 0 ? (int64_t)2 * (Cin / 32) * 2 * 64 * 4 : 0; }

// w_tap_co_ci: [3][3][Cout][Cin]; w_absmax: its |max| words (ccst_absmax_f32).
extern "C" int ccst_pack_conv_weight_zform_f32(const float* w_tap_co_ci, const unsigned* w_absmax, float* packed, int Cin, int Cout,
                                               void* stream) {
    CCST_REQUIRE(w_tap_co_ci && w_absmax && packed, "pack_conv_weight_zform: null pointer");
    CCST_REQUIRE((Cin == 32 || Cin == 64) && Cout >= 1 && Cout <= 3, "pack_conv_weight_zform: Cin must be 32 or 64, Cout 1..3");
    const int threads = 2 * (Cin / 32) * 64;
    hipLaunchKernelGGL(pack_weight_zform_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_tap_co_ci, w_absmax,
                       packed, Cin, Cout);
    return ccst_launch_status("pack_conv_weight_zform");
}

// x: NHWC [N,H,W,Cin], Cin 32 or 64, with its |max| words; y: NCHW [N,Cout,H,W], Cout 1..3.  Result contract:
// fp32 products to 2^-22 relative, fp32 accumulation.
extern "C" int ccst_conv3x3_zform_f32(const float* x, const unsigned* x_absmax, const float* w_packed, const unsigned* w_absmax,
                                      const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int reflect, int relu,
                                      void* stream) {
    CCST_REQUIRE(x && x_absmax && w_packed && w_absmax && y, "conv3x3_zform: null pointer");
    CCST_REQUIRE(N > 0 && H > 0 && W > 0 && (Cin == 32 || Cin == 64) && Cout >= 1 && Cout <= 3, "conv3x3_zform: bad shape");
    CCST_REQUIRE((long long)N * H * W * Cin < 0xffffffffLL, "conv3x3_zform: input must have < 2^32 elements");
    if (reflect) CCST_REQUIRE(H >= 2 && W >= 2, "conv3x3_zform: reflection needs extent >= 2");
    ZArgs a;
    a.x = x;
    a.xmax = x_absmax;
    a.wp = w_packed;
    a.wmax = w_absmax;
    a.bias = bias;
    a.y = y;
    a.N = N;
    a.H = H;
    a.W = W;
    a.Cin = Cin;
    a.Cout = Cout;
    a.reflect = reflect;
    a.relu = relu;
    a.tilesX = (W + Z_TW - 1) / Z_TW;
    a.tilesY = (H + Z_TH - 1) / Z_TH;
    const long long ntiles = (long long)N * a.tilesX * a.tilesY;
    CCST_REQUIRE(ntiles < 0x7fffffffLL, "conv3x3_zform: too many tiles");
    const long long resident = (long long)ccst_num_cus() * Z_WGS;               // persistent: Z_WGS workgroups per CU (a multiple of 8 on this chip)
    const long long grid = ntiles < resident ? ntiles : resident;
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 64 && reflect)
        hipLaunchKernelGGL((conv3x3_zform_kernel<2, true>), dim3((unsigned)grid), dim3(256), 0, s, a);
    else if (Cin == 64)
        hipLaunchKernelGGL((conv3x3_zform_kernel<2, false>), dim3((unsigned)grid), dim3(256), 0, s, a);
    else if (reflect)
        hipLaunchKernelGGL((conv3x3_zform_kernel<1, true>), dim3((unsigned)grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((conv3x3_zform_kernel<1, false>), dim3((unsigned)grid), dim3(256), 0, s, a);
    return ccst_launch_status("conv3x3_zform");
}
