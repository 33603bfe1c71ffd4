// 3x3 stride-1 "same" convolution as fused Winograd F(4x4, 3x3) on the fp32-input MFMA -- the WIDE form: 64 output channels per
// workgroup (conv3x3_wino4.hip: 32), so every transformed input value feeds TWO MFMAs and the input halo is fetched once per 64
// output channels.
//
// Why: on gfx950 fp32 vector instructions do not execute in the shadow of fp32 MFMAs (tools/micro/mfma_valu.hip), so the kernel's
// time is (MFMAs x 64 cycles) + (vector instructions x ~5.5 cycles) + stalls.  The 32-channel kernel spends 3.0-3.8 vector
// instructions per MFMA in its loop (each wave re-derives the row pass of its transform row) and ~730 per wave outside it.  Here:
//   * workgroup = 16x32 output pixels (4x8 Winograd tiles = M of v_mfma_f32_32x32x2_f32) x 64 output channels, FOUR waves, one per
//     SIMD with the whole register file (36 positions x 2 channel groups x 16 accumulator registers = 1152 of the CU's 2048);
//   * wave w owns the 3x3 block of transform positions rows 3*(w>>1).., columns 3*(w&1).. for BOTH 32-channel groups: 18
//     accumulators.  Row pairs (1,2) and (3,4) of B^T share their partial sums, so a 3-row block of one patch column costs 6 packed
//     instructions and the 3x3 block 5*6 + 3*6 = 48 per channel pair = 1.33 per MFMA;
//   * the loop is software pipelined over channel PAIRS (36 MFMAs = 9 positions x 2 channels x 2 groups): while pair p multiplies,
//     the wave reads pair p+1's 5x5 patch from the LDS halo (ds_read_b64), transforms it, refills each position's weight registers
//     right after their last use (L2 -> registers, one pair ahead) and moves the next chunk's halo global -> registers -> LDS;
//     ONE barrier per 16-channel chunk;
//   * halo staging with scalar row offsets: a thread owns one (pixel column, 4-channel part) and walks 9 rows whose byte offsets
//     live in SGPRs -- no per-unit address registers, ~30 instead of ~340 vector instructions of set-up per wave;
//   * the MFMA runs with the WEIGHTS as its A operand (M = 32 output channels) and the transformed input as B (N = 32 tiles): a
//     lane's accumulator registers 4g..4g+3 are then four CONSECUTIVE output channels of one tile, so the epilogue stores 16 bytes per
//     lane (a quarter of the store instructions; the epilogue of a store-issue-bound tile, not the bandwidth, is what costs);
//   * weights U[chunk][pos][channel pair][k half][cout/64][32][2 groups][2]: one 16-byte load per lane = both channel groups' operands
//     of one position and channel pair, a wave's load = 2 x 512 contiguous bytes (the [cout][8] layout of the 32-channel kernel gives
//     8 useful bytes per 32: the texture path, not the MFMA, set the pace -- measured: removing the loads alone was worth 16 %);
//   * epilogue: the 36 x 64 x 32 accumulators go through LDS once (147 KB, XOR-swizzled b128 slots, two passes of one channel group),
//     then every thread does the complete A^T(.)A of one tile for four channels in registers (100 packed operations per channel
//     pair) with the bias folded into position (1,1), whose two A^T coefficients are both 1 for every output.
// x, y and the flags: as conv3x3_wino4.hip; weights from ccst_pack_conv_weight_wino4w_f32 (cout_pad a multiple of 64).
#include "common.h"
#include <type_traits>

#define W4W_STORE_AUX 0
#define W4W_LOAD_AUX 0
namespace {

struct W4wArgs {
    const float* x;
    const float* u;
    const float* bias;
    float* y;
    float* sums;                                   // STATS: [N * tilesY * tilesX][Cout][2] per-tile (sum, sum of squares) of the output
    int N, H, W, Hs, Ws, Cin, Cout, CoutPad;      // H,W: conv (= output) extent; Hs,Ws: source extent (H/2,W/2 if ups)
    int reflect, ups, relu;
    long long ysN;
    int ysH, ysW;                                  // output strides (of the pooled tensor when POOL)
    int tilesX, tilesY, tilesN, ntiles;
    unsigned long long mN, mX, mY;                 // ceil(2^36 / tiles*): exact quotients for tile ids < 2^24 without a division
};

constexpr int NTW = 256;                           // threads per workgroup: four waves, one per SIMD
constexpr int CKW = 16, PITW = CKW + 4;            // channels per chunk; floats per pixel in LDS (80 B)
constexpr int THW = 16, TWW = 32;                  // output pixels per workgroup: 4 x 8 tiles of 4 x 4
constexpr int HHW = THW + 2, HWW = TWW + 2;        // halo 18 x 34
constexpr int PLANEW = ((HWW + 3) / 4) * PITW;     // 9 pixel columns per plane (x mod 4)
constexpr int ROWPW = 4 * PLANEW + 8;              // 728 floats per halo row
constexpr int HIMGW = HHW * ROWPW;                 // 13104 floats = 52.4 KB per buffer
constexpr int EXW_BYTES = 36 * 32 * 32 * 4;        // epilogue exchange: [position][channel][32 tiles] = 147456 B
static_assert(2 * HIMGW * 4 <= EXW_BYTES, "the halo buffers live inside the exchange area");
constexpr int MBW_BYTES = 4096 + 4 * 128;          // mailbox behind the exchange area: the NEXT tile's staging set-up (see wino4w_body)
constexpr int SRW_BYTES = 256 * 32;                // STATS: per-thread (sum[4], sq[4]) before the cross-tile reduction
constexpr unsigned OOBW = 0x40000000u;             // a byte offset beyond any image (images are < 2^30 bytes): the buffer load returns 0

__device__ __forceinline__ int reflectw(int i, int n) {
    i = (i < 0) ? -i : i;
    i = (i >= n) ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 bufw_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, W4W_LOAD_AUX));
}
__device__ __forceinline__ f32x2 bufw_load2(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0));
}
__device__ __forceinline__ f32x2 lo2w(f32x4 a) { return f32x2{a[0], a[1]}; }
__device__ __forceinline__ f32x2 hi2w(f32x4 a) { return f32x2{a[2], a[3]}; }
// packed fp32 in assembly (hipcc splits packed fp32 next to MFMAs back into scalar instructions; see conv3x3_wino4.hip)
typedef unsigned long long k64w;
__device__ __forceinline__ k64w splatw(float k) {
    const unsigned b = __builtin_amdgcn_readfirstlane(__float_as_uint(k));
    return ((k64w)b << 32) | b;
}
__device__ __forceinline__ f32x2 pkw_fma(f32x2 a, k64w k, f32x2 c) {          // a * k + c, k an SGPR pair
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pkw_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pkw_sub(f32x2 a, f32x2 b) {                   // a - b
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// Three outputs of the 1D input transform B^T from five consecutive inputs x0..x4.
//   FIRST  (rows / columns 0,1,2; x = d0..d4):  4 d0 - 5 d2 + d4;   (d4 - 4 d2) + (d3 - 4 d1);   (d4 - 4 d2) - (d3 - 4 d1)
//   !FIRST (rows / columns 3,4,5; x = d1..d5):  (d4 - d2) + 2 (d3 - d1);   (d4 - d2) - 2 (d3 - d1);   4 d1 - 5 d3 + d5
struct KW {
    k64w c4, cm5, cm4, c2, cm2;
};
template <bool FIRST>
__device__ __forceinline__ void bt3(const KW& K, f32x2 x0, f32x2 x1, f32x2 x2, f32x2 x3, f32x2 x4, f32x2& o0, f32x2& o1, f32x2& o2) {
    if (FIRST) {
        const f32x2 t = pkw_fma(x2, K.cm5, x4);
        const f32x2 s = pkw_fma(x2, K.cm4, x4);
        const f32x2 u = pkw_fma(x1, K.cm4, x3);
        o0 = pkw_fma(x0, K.c4, t);
        o1 = pkw_add(s, u);
        o2 = pkw_sub(s, u);
    } else {
        const f32x2 s = pkw_sub(x3, x1);
        const f32x2 t = pkw_sub(x2, x0);
        const f32x2 w = pkw_fma(x2, K.cm5, x4);
        o0 = pkw_fma(t, K.c2, s);
        o1 = pkw_fma(t, K.cm2, s);
        o2 = pkw_fma(x0, K.c4, w);
    }
}

// 1D output transform A^T: six values -> four.   A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]
struct KO {
    k64w c2, c4, c8;
};
__device__ __forceinline__ void at4(const KO& K, f32x2 m0, f32x2 m1, f32x2 m2, f32x2 m3, f32x2 m4, f32x2 m5, f32x2& y0, f32x2& y1, f32x2& y2,
                                    f32x2& y3) {
    const f32x2 s1 = pkw_add(m1, m2), d1 = pkw_sub(m1, m2), s2 = pkw_add(m3, m4), d2 = pkw_sub(m3, m4);
    y0 = pkw_add(pkw_add(m0, s1), s2);
    y1 = pkw_fma(d2, K.c2, d1);
    y2 = pkw_fma(s2, K.c4, s1);
    y3 = pkw_add(pkw_fma(d2, K.c8, d1), m5);
}

#define WW_SB __builtin_amdgcn_sched_barrier(0)

template <bool POOL, bool STATS, int RB, int CB>
__device__ __forceinline__ void wino4w_body(const W4wArgs& pk, float* __restrict__ lds) {
    float* const Hs0 = lds;
    float* const Hs1 = lds + HIMGW;
    constexpr int rq = RB;                                                       // row parity of this wave's halo rows (wave = 2 RB + CB)

    // tile id -> TileW (threads 0..127 walk the even halo rows, 128..255 the odd ones, nine each: thread = (pixel column hx = 0..31,
    // 4-channel part); pixel columns 32, 33 (8 units x 18 rows) are one extra unit on threads 0..143)
    // (plain locals, not a struct: an aggregate with an array and a buffer resource that is copied at the end of an iteration stays
    //  in scratch memory, its loads count as divergent, and every buffer load with a row offset becomes a waterfall loop)
    auto setup = [&](int tile, const W4wArgs& p, int tid, int& t_tn, int& t_n, int& t_co0, int& t_oy0, int& t_ox0, __amdgpu_buffer_rsrc_t& t_xrs,
                     unsigned (&t_srow)[9], unsigned& t_rowbad, unsigned& t_voff, unsigned& t_voffx, int& t_hdst, int& t_hdstx) {
        // (three runtime integer divisions cost ~1500 cycles of the tile's set-up: multiply by ceil(2^36 / d) instead, exact for
        //  ids < 2^24 and d <= 4096, both checked by the launcher)
        auto divmod = [](int a, int d, unsigned long long m, int& r) {
            const int q = (int)(((unsigned long long)(unsigned)a * m) >> 36);
            r = a - q * d;
            return q;
        };
        int bid = ccst_xcd_remap(tile, pk.ntiles), tx, ty;
        bid = divmod(bid, p.tilesN, p.mN, t_tn);
        bid = divmod(bid, p.tilesX, p.mX, tx);
        t_n = divmod(bid, p.tilesY, p.mY, ty);
        t_co0 = t_tn * 64;
        t_oy0 = ty * THW;
        t_ox0 = tx * TWW;
        const int oy0_ = t_oy0, ox0_ = t_ox0;
        t_xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + (long long)t_n * p.Hs * p.Ws * p.Cin, 0,
                                                  (int)((unsigned)p.Hs * p.Ws * p.Cin * 4u), 0x00020000);
        auto src_x = [&](int hx, bool& ok) {
            int gx = ox0_ + hx - 1;
            ok = true;
            if (p.reflect) {
                gx = reflectw(gx, p.W);
            } else {
                ok = (gx >= 0) & (gx < p.W);
                gx = min(max(gx, 0), p.W - 1);
            }
            return gx >> p.ups;
        };
        auto src_y = [&](int hy, bool& ok) {
            int gy = oy0_ + hy - 1;
            ok = true;
            if (p.reflect) {
                gy = reflectw(gy, p.H);
            } else {
                ok = (gy >= 0) & (gy < p.H);
                gy = min(max(gy, 0), p.H - 1);
            }
            return gy >> p.ups;
        };
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            bool ok;
            const int gy = src_y(rq + 2 * i, ok);
            t_srow[i] = (unsigned)__builtin_amdgcn_readfirstlane((gy * p.Ws * p.Cin) * 4);
            bad |= ok ? 0u : (1u << i);
        }
        t_rowbad = (unsigned)__builtin_amdgcn_readfirstlane((int)bad);
        {
            const int idx = tid & 127, hx = idx >> 2, part = idx & 3;
            bool ok;
            const int gx = src_x(hx, ok);
            t_voff = ok ? (unsigned)((gx * p.Cin + part * 4) * 4) : OOBW;
            t_hdst = rq * ROWPW + (hx & 3) * PLANEW + (hx >> 2) * PITW + part * 4;
        }
        t_voffx = OOBW;
        t_hdstx = 0;
        if (tid < 8 * HHW) {
            const int hy = tid >> 3, hx = 32 + ((tid >> 2) & 1), part = tid & 3;
            bool okx, oky;
            const int gx = src_x(hx, okx), gy = src_y(hy, oky);
            t_voffx = (okx & oky) ? (unsigned)(((gy * p.Ws + gx) * p.Cin + part * 4) * 4) : OOBW;
            t_hdstx = hy * ROWPW + (hx & 3) * PLANEW + (hx >> 2) * PITW + part * 4;
        }
    };
    // A tile's FIRST halo chunk is requested one tile early -- before the previous tile's epilogue, whose output stores would
    // otherwise sit in front of these loads in the memory pipeline (stamps: 3.6k cycles to issue 28 loads behind the store burst
    // + 3.6k waiting for them, of a 72k-cycle 64 -> 64 tile) -- into ten registers that live through that epilogue.
    f32x4 rp[10];
    auto prefetch = [&](__amdgpu_buffer_rsrc_t xr, const unsigned (&sr)[9], unsigned bad, unsigned vo, unsigned vox, bool has_x) {
#pragma unroll
        for (int i = 0; i < 9; ++i) rp[i] = bufw_load4(xr, ((bad >> i) & 1) ? OOBW : vo, (unsigned)__builtin_amdgcn_readfirstlane((int)sr[i]));
        rp[9] = bufw_load4(xr, vox, 0);       // every thread: without an extra unit the offset is out of range and nothing is fetched
        (void)has_x;                          // (a conditional load leaves a phi that is resolved by copies behind an s_waitcnt vmcnt(0),
                                              //  and the differing load counts of the two paths make every later wait a full drain)
    };
    int c_tn = 0, c_n = 0, c_co0 = 0, c_oy0 = 0, c_ox0 = 0, c_hdst = 0, c_hdstx = 0;
    unsigned c_srow[9], c_rowbad = 0, c_voff = OOBW, c_voffx = OOBW;
    __amdgpu_buffer_rsrc_t c_xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pk.x), 0, 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < 9; ++i) c_srow[i] = 0;
    {
        int tid0 = threadIdx.x;
        asm volatile("" : "+v"(tid0));
        if ((int)blockIdx.x < pk.ntiles) {
            setup(blockIdx.x, pk, tid0, c_tn, c_n, c_co0, c_oy0, c_ox0, c_xrs, c_srow, c_rowbad, c_voff, c_voffx, c_hdst, c_hdstx);
            prefetch(c_xrs, c_srow, c_rowbad, c_voff, c_voffx, tid0 < 8 * HHW);
        }
    }

    // Persistent workgroups: the grid is one workgroup per CU (a multiple of 8, so that tile ids t and t + 8 still share an XCD) and
    // each walks tiles blockIdx.x, + gridDim.x, ...  The output stores of a tile then drain while the next tile's loop runs; as one
    // tile per workgroup every CU finished at the same moment and the whole layer's output went to HBM in one burst that nothing
    // overlapped (removing the epilogue was worth 7-26 % per layer).
#pragma unroll 1
    for (int tile = blockIdx.x; tile < pk.ntiles; tile += gridDim.x) {
    // Everything below is per tile ON PURPOSE: the arguments and the thread id pass through an opaque zero each iteration, otherwise
    // the compiler hoists every tile-invariant address, descriptor and constant out of this loop and keeps ~400 scalar registers'
    // worth of them alive across the whole body (spilled to vector lanes, then the vector registers spill too).
    int zi = 0;
    float zf = 0.f;
    asm volatile("" : "+s"(zi), "+s"(zf));
    W4wArgs p = pk;
    p.H += zi; p.W += zi; p.Hs += zi; p.Ws += zi; p.Cin += zi; p.Cout += zi; p.CoutPad += zi; p.ysH += zi; p.ysW += zi; p.ysN += zi;
    p.tilesX += zi; p.tilesY += zi; p.tilesN += zi; p.reflect += zi; p.ups += zi; p.relu += zi;
    p.x += zi; p.u += zi; p.y += zi;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
#define WW_STAMP(k) do { } while (0)
    WW_STAMP(0);
    const int tn = c_tn, n = c_n, co0 = c_co0, oy0 = c_oy0, ox0 = c_ox0;
    const int tile_spatial = STATS ? (n * p.tilesY + oy0 / THW) * p.tilesX + ox0 / TWW : 0;
    (void)tile_spatial;
    const int nchunks_ = p.Cin / CKW;
    (void)nchunks_;
    const bool has_x = tid < 8 * HHW;
    // staging groups: G0 = rows 0..3, G1 = rows 4..7, G2 = row 8 + the extra unit
    // two register sets: a group is stored two pair-steps (~2 us) after its loads were issued -- they come from HBM
    f32x4 rh[2][4];
    auto load_g = [&](int g, int c, int rs) {
        const unsigned cs = (unsigned)c * (CKW * 4);
        if (g < 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = 4 * g + k;
                rh[rs][k] = bufw_load4(c_xrs, ((c_rowbad >> i) & 1) ? OOBW : c_voff, (unsigned)__builtin_amdgcn_readfirstlane((int)c_srow[i]) + cs);
            }
        } else {
            rh[rs][0] = bufw_load4(c_xrs, ((c_rowbad >> 8) & 1) ? OOBW : c_voff, (unsigned)__builtin_amdgcn_readfirstlane((int)c_srow[8]) + cs);
            rh[rs][1] = bufw_load4(c_xrs, c_voffx, cs);      // (unconditional, see prefetch)
        }
    };
    auto store_g = [&](int g, float* __restrict__ dst, int rs) {
        if (g < 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(dst + c_hdst + (4 * g + k) * 2 * ROWPW) = rh[rs][k];
        } else {
            *reinterpret_cast<f32x4*>(dst + c_hdst + 16 * ROWPW) = rh[rs][0];
            if (has_x) *reinterpret_cast<f32x4*>(dst + c_hdstx) = rh[rs][1];
        }
    };

    // ---- A side: the lane's tile (li) and channel half (lh); 5x5 patch rows RB.., columns CB.. -------------------------------
    const KW K{splatw(4.f + zf), splatw(-5.f + zf), splatw(-4.f + zf), splatw(2.f + zf), splatw(-2.f + zf)};
    const int tyy = li >> 3, txx = li & 7;
    const int abase = (4 * tyy + RB) * ROWPW + txx * PITW + lh * 8;              // + a*ROWPW + (c&3)*PLANEW + (c>>2)*PITW + 2*pair
    auto col_off = [](int cc) { const int c = cc + CB; return (c & 3) * PLANEW + (c >> 2) * PITW; };

    // ---- weights: U[chunk][pos][pair][k half][cout/64][32][2][2]: per-lane byte offset, (chunk, position, pair) in the scalar offset ----
    const int nchunks = p.Cin / CKW;
    const int last = nchunks - 1;
    const int ncb = p.CoutPad >> 6;
    const unsigned up_bytes = (unsigned)(2 * ncb * 128) * 4u;                   // bytes per (chunk, position, pair)
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, (int)(144u * up_bytes * (unsigned)nchunks),
                                                                         0x00020000);
    const unsigned uvoff = (unsigned)(((lh * ncb + tn) * 128 + li * 4) * 4);
    auto upos = [&](int j) { return (unsigned)(((3 * RB + j / 3) * 6 + 3 * CB + j % 3) * 4) * up_bytes; };

    f32x16 acc[9][2];              // written first by the C = 0 MFMAs of the very first pair-step (no 288-register clear)

    f32x2 v[9];                    // A operands of the current pair: [position][channel of the pair]
    f32x4 bq[2][9];                // weight operands: [pair-step parity][position][2 * channel group + channel of the pair]
    f32x2 w[3][5];                 // row pass of the next pair: [row of the block][patch column]
    f32x2 d[2][5];                 // raw patch column in flight: [slot][patch row]

    // (chunk c, pair pr) -> set pr & 1: refilled right after position j's MFMAs of the pair-step that used it, for the pair-step
    // AFTER the next one -- two pair-steps (~2 us) of lead.  Vector memory loads return in order: a weight load (L2) younger than a
    // halo load (HBM) cannot return before it, and with one pair-step of lead every halo load stalled the MFMAs behind it.
    auto load_b = [&](int j, int c, int pr) {
        bq[pr & 1][j] = bufw_load4(urs, uvoff, (unsigned)c * 144u * up_bytes + upos(j) + (unsigned)pr * up_bytes);
    };
    auto read_col = [&](const float* __restrict__ hs, int pr, int cc, int slot) {
        const float* hp = hs + abase + col_off(cc) + 2 * pr;
#pragma unroll
        for (int a = 0; a < 5; ++a) d[slot][a] = *reinterpret_cast<const f32x2*>(hp + a * ROWPW);
    };
    auto row_pass = [&](int cc, int slot) {
        bt3<RB == 0>(K, d[slot][0], d[slot][1], d[slot][2], d[slot][3], d[slot][4], w[0][cc], w[1][cc], w[2][cc]);
    };
    auto col_pass = [&](int rr) {
        bt3<CB == 0>(K, w[rr][0], w[rr][1], w[rr][2], w[rr][3], w[rr][4], v[3 * rr], v[3 * rr + 1], v[3 * rr + 2]);
    };
    // The 18 accumulators are 288 registers: 16 of them fill the 256 accumulation registers (AGPRs), position 8's two live in
    // ordinary VGPRs.  Written as assembly because the register class is not expressible through the builtin: left to itself the
    // allocator puts all 18 into AGPRs and then shuttles blocks between the two files inside the loop (496 v_accvgpr_* per chunk).
#define WW_M1(j, nb, k)                                                                                                              \
    do {                                                                                                                             \
        if ((j) < 8) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[j][nb]) : "v"(bq[ps][j][2 * (nb) + (k)]), "v"(v[j][k])); \
        else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j][nb]) : "v"(bq[ps][j][2 * (nb) + (k)]), "v"(v[j][k]));      \
    } while (0)
#define WW_M1I(j, nb, k)                                                                                                             \
    do {                                                                                                                             \
        if ((j) < 8) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=a"(acc[j][nb]) : "v"(bq[ps][j][2 * (nb) + (k)]), "v"(v[j][k])); \
        else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(acc[j][nb]) : "v"(bq[ps][j][2 * (nb) + (k)]), "v"(v[j][k]));     \
    } while (0)
#define WW_MFMA4(j)                                                                                                                  \
    do {                                                                                                                             \
        if (decltype(init)::value) { WW_M1I(j, 0, 0); WW_M1I(j, 1, 0); } else { WW_M1(j, 0, 0); WW_M1(j, 1, 0); }                     \
        WW_M1(j, 0, 1); WW_M1(j, 1, 1);                                                                                              \
    } while (0)

    // One pair-step (pair ps of its chunk): the 36 MFMAs of the current pair (operands v[], bq[ps & 1][]), position by position;
    // between the groups the LDS reads (one patch column ahead) + the transform of the NEXT pair (buffer hs, pair pn) and the
    // refill of the weight registers just used, for the pair-step after the next (chunk cb, pair pb).  The staging work of this
    // step (sg_store / sg_load: group or -1) sits in groups 2..4.
    // opts: bit 0 = refill the weight registers, bit 1 = read + transform the next pair (the tail of a tile's last chunk does neither:
    // loads still in flight when the epilogue starts are waited for there -- its first ds_reads reuse their destination registers)
    auto pair_step = [&](auto ps_t, const float* __restrict__ hs, int pn, int cb, int pb, int sg_store, float* __restrict__ sdst, int sg_load, int sc,
                         auto init, auto opts) {
        constexpr int ps = decltype(ps_t)::value & 1;
        constexpr bool LB = (decltype(opts)::value & 1) != 0, XF = (decltype(opts)::value & 2) != 0;
#define WW_GS(k) do { } while (0)
        WW_GS(0);
        if (XF) read_col(hs, pn, 0, 0);
        WW_SB;
        WW_MFMA4(0);
        if (LB) load_b(0, cb, pb);
        if (XF) read_col(hs, pn, 1, 1);
        WW_SB;
        WW_GS(1);
        WW_MFMA4(1);
        if (LB) load_b(1, cb, pb);
        if (XF) row_pass(0, 0);
        if (XF) read_col(hs, pn, 2, 0);
        WW_SB;
        WW_GS(2);
        WW_MFMA4(2);
        if (LB) load_b(2, cb, pb);
        if (XF) row_pass(1, 1);
        if (XF) read_col(hs, pn, 3, 1);
        if (sg_store >= 0) store_g(sg_store, sdst, sg_store == 1);
        WW_SB;
        WW_GS(3);
        WW_MFMA4(3);
        if (LB) load_b(3, cb, pb);
        if (XF) row_pass(2, 0);
        if (XF) read_col(hs, pn, 4, 0);
        if (sg_load >= 0) load_g(sg_load, sc, sg_load == 1);
        WW_SB;
        WW_GS(4);
        WW_MFMA4(4);
        if (LB) load_b(4, cb, pb);
        if (XF) row_pass(3, 1);
        WW_SB;
        WW_GS(5);
        WW_MFMA4(5);
        if (LB) load_b(5, cb, pb);
        if (XF) row_pass(4, 0);
        WW_SB;
        WW_GS(6);
        WW_MFMA4(6);
        if (LB) load_b(6, cb, pb);
        WW_SB;
        WW_GS(7);
        WW_MFMA4(7);
        if (LB) load_b(7, cb, pb);
        WW_SB;
        WW_GS(8);
        WW_MFMA4(8);
        if (LB) load_b(8, cb, pb);
        WW_SB;
        WW_GS(9);
        if (XF) {
            col_pass(0);
            col_pass(1);
            col_pass(2);
        }
        WW_SB;
        WW_GS(10);
    };

    // ---- prologue: chunk 0 -> LDS, weights of (chunk 0, pair 0), the transform of pair 0, group 0 of chunk 1 in flight ---------
    WW_STAMP(8);
    // the bias of both epilogue passes, loaded HERE: vector memory loads return in order, so inside the epilogue a load issued behind
    // the next tile's halo requests would wait for HBM, and one issued just before them stalls the epilogue for its own latency
    const int rquad_ = tid & 7;
    f32x4 biasv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (p.bias != nullptr) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = co0 + nb * 32 + 4 * rquad_ + k;
                biasv[nb][k] = (c < p.Cout) ? p.bias[c] : 0.f;
            }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) load_b(j, 0, 0);
#pragma unroll
    for (int j = 0; j < 9; ++j) load_b(j, 0, 1);
    WW_STAMP(9);
    // chunk 0 was requested a tile ago (rp[]); every wave is past the closing barrier, the exchange area is free
#pragma unroll
    for (int i = 0; i < 9; ++i) *reinterpret_cast<f32x4*>(Hs0 + c_hdst + i * 2 * ROWPW) = rp[i];
    if (has_x) *reinterpret_cast<f32x4*>(Hs0 + c_hdstx) = rp[9];
    WW_STAMP(10);
    load_g(0, min(1, last), 0);
    load_g(1, min(1, last), 1);
    // The NEXT tile's staging set-up, computed here -- where the wave waits for the barrier anyway and the kernel arguments are at
    // hand -- and parked in LDS behind the exchange area: in the epilogue, where the next tile's first halo chunk is requested, the
    // same set-up cost 3-4k cycles of serialised scalar-load latencies (the arguments do not survive the loop in registers).
    const int tile_next = tile + gridDim.x;
    const bool more = tile_next < pk.ntiles;                                    // (uniform)
    char* const mbox = reinterpret_cast<char*>(lds) + EXW_BYTES;
    constexpr int WVB = 4096 + (2 * RB + CB) * 128;
    if (more) {
        int t_tn, t_n, t_co0, t_oy0, t_ox0, t_hdst, t_hdstx;
        unsigned t_srow[9], t_rowbad, t_voff, t_voffx;
        __amdgpu_buffer_rsrc_t t_xrs;
        setup(tile_next, p, tid, t_tn, t_n, t_co0, t_oy0, t_ox0, t_xrs, t_srow, t_rowbad, t_voff, t_voffx, t_hdst, t_hdstx);
        *reinterpret_cast<u32x4*>(mbox + tid * 16) = u32x4{t_voff, t_voffx, (unsigned)t_hdst, (unsigned)t_hdstx};
        if (lane == 0) {
            const unsigned long long xb = (unsigned long long)(const_cast<float*>(p.x) + (long long)t_n * p.Hs * p.Ws * p.Cin);
            *reinterpret_cast<u32x4*>(mbox + WVB) = u32x4{t_srow[0], t_srow[1], t_srow[2], t_srow[3]};
            *reinterpret_cast<u32x4*>(mbox + WVB + 16) = u32x4{t_srow[4], t_srow[5], t_srow[6], t_srow[7]};
            *reinterpret_cast<u32x4*>(mbox + WVB + 32) = u32x4{t_srow[8], t_rowbad, (unsigned)xb, (unsigned)(xb >> 32)};
            *reinterpret_cast<u32x4*>(mbox + WVB + 48) = u32x4{(unsigned)t_tn, (unsigned)t_n, (unsigned)t_co0, (unsigned)t_oy0};
            *reinterpret_cast<u32x4*>(mbox + WVB + 64) = u32x4{(unsigned)t_ox0, (unsigned)p.Hs * p.Ws * p.Cin * 4u, 0u, 0u};
        }
    }
    __syncthreads();
    WW_STAMP(11);
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) {
        read_col(Hs0, 0, cc, 0);
        row_pass(cc, 0);
    }
    col_pass(0);
    col_pass(1);
    col_pass(2);

    // ---- main loop: four pair-steps per 16-channel chunk.  The halo of chunk c+1 goes to the other buffer during steps 0..2 (its
    // last readers left through the previous barrier), the barrier sits before step 3, whose transform already reads chunk c+1. ----
    typedef std::integral_constant<int, 3> FULL;
    auto chunk = [&](int c, auto first) {
        const float* cur = (c & 1) ? Hs1 : Hs0;
        float* nxt = (c & 1) ? Hs0 : Hs1;
        const int c1 = min(c + 1, last), c2 = min(c + 2, last);
        // halo of chunk c+1 -> the other buffer (its last readers left through the previous barrier): groups 0, 1 were requested two
        // pair-steps ago (sets 0, 1), group 2 is requested now (set 0, behind group 0's stores); groups 0, 1 of chunk c+2 in steps 2, 3
        pair_step(std::integral_constant<int, 0>{}, cur, 1, c, 2, 0, nxt, 2, c1, first, FULL{});
        pair_step(std::integral_constant<int, 1>{}, cur, 2, c, 3, 1, nxt, -1, 0, std::false_type{}, FULL{});
        pair_step(std::integral_constant<int, 2>{}, cur, 3, c1, 0, 2, nxt, 0, c2, std::false_type{}, FULL{});
        __syncthreads();
        WW_SB;
        pair_step(std::integral_constant<int, 3>{}, nxt, 0, c1, 1, -1, nullptr, 1, c2, std::false_type{}, FULL{});
    };
    // a tile's LAST chunk: no next chunk to stage, no weights beyond pair 3, nothing to transform in its last pair-step, no barrier
    // (the epilogue brings its own) -- and nothing left in flight when the epilogue starts
    auto chunk_last = [&](int c, auto first) {
        const float* cur = (c & 1) ? Hs1 : Hs0;
        pair_step(std::integral_constant<int, 0>{}, cur, 1, c, 2, -1, nullptr, -1, 0, first, FULL{});
        pair_step(std::integral_constant<int, 1>{}, cur, 2, c, 3, -1, nullptr, -1, 0, std::false_type{}, FULL{});
        pair_step(std::integral_constant<int, 2>{}, cur, 3, c, 0, -1, nullptr, -1, 0, std::false_type{}, std::integral_constant<int, 2>{});
        pair_step(std::integral_constant<int, 3>{}, cur, 0, c, 0, -1, nullptr, -1, 0, std::false_type{}, std::integral_constant<int, 0>{});
    };
    WW_STAMP(1);
    chunk(0, std::true_type{});         // its first pair-step starts the accumulators (C = 0); Cin >= 32: never the last chunk
    for (int c = 1; c < last; ++c) chunk(c, std::false_type{});
    chunk_last(last, std::false_type{});
    WW_STAMP(2);
    int n_tn = 0, n_n = 0, n_co0 = 0, n_oy0 = 0, n_ox0 = 0, n_hdst = 0, n_hdstx = 0;
    unsigned n_srow[9], n_rowbad = 0, n_voff = OOBW, n_voffx = OOBW;
    __amdgpu_buffer_rsrc_t n_xrs = c_xrs;
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (asm MFMAs: no automatic wait states between the last one and the first read of its result)
    __syncthreads();                // every wave is done with the halo buffers: the exchange area overlays them

    // ---- epilogue ---------------------------------------------------------------------------------------------------------------
    // exchange area ex[pos][tile 0..31][channel quad ^ (tile & 7)][4 channels]: a lane's accumulator registers 4g..4g+3 are
    // channels 8g + 4 lh .. +3 (of the group) of tile li = one 16-byte slot; the XOR spreads the 8 (16) lanes of a b128 access over
    // the banks.
    char* const ex = reinterpret_cast<char*>(lds);
    const int wbase = ((3 * RB) * 6 + 3 * CB) * 4096 + li * 128;                 // + (rr*6 + qq) * 4096 + ((2g + lh) ^ sw) * 16
    const int sw = li & 7;            // (LDS stores bank over 128 B = one 32-channel row: eight consecutive tiles need eight different slots)
    // reader: thread = (tile tid >> 3, channel quad tid & 7): consecutive lanes store consecutive 16 bytes of one pixel
    const int rt = tid >> 3, rquad = tid & 7;
    const int rbase = rt * 128 + ((rquad ^ (rt & 7)) << 4);
    const KO KOut{splatw(2.f + zf), splatw(4.f + zf), splatw(8.f + zf)};
    const bool interior = (oy0 + THW <= p.H) && (ox0 + TWW <= p.W);
    const bool relu = p.relu != 0;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + (long long)n * p.ysN, 0, (int)(p.ysN * 4), 0x00020000);
    const int oyb = oy0 + 4 * (rt >> 3), oxb = ox0 + 4 * (rt & 7);               // the reader's tile
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        if (nb == 1) { WW_STAMP(3); __syncthreads(); }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const f32x16 a = acc[j][nb];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 q4 = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                *reinterpret_cast<f32x4*>(ex + wbase + ((j / 3) * 6 + (j % 3)) * 4096 + (((2 * g + lh) ^ sw) << 4)) = q4;
            }
        }
        if (nb == 0) {              // the next tile's first halo chunk: ahead of this tile's output stores, half the accumulators retired
            WW_STAMP(12);
            if (more) {
                const u32x4 ml = *reinterpret_cast<const u32x4*>(mbox + tid * 16);
                const u32x4 m0 = *reinterpret_cast<const u32x4*>(mbox + WVB), m1 = *reinterpret_cast<const u32x4*>(mbox + WVB + 16);
                const u32x4 m2 = *reinterpret_cast<const u32x4*>(mbox + WVB + 32), m3 = *reinterpret_cast<const u32x4*>(mbox + WVB + 48);
                const u32x4 m4 = *reinterpret_cast<const u32x4*>(mbox + WVB + 64);
                auto uni = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
                n_voff = ml[0]; n_voffx = ml[1]; n_hdst = (int)ml[2]; n_hdstx = (int)ml[3];
#pragma unroll
                for (int i = 0; i < 4; ++i) { n_srow[i] = uni(m0[i]); n_srow[4 + i] = uni(m1[i]); }
                n_srow[8] = uni(m2[0]);
                n_rowbad = uni(m2[1]);
                n_tn = (int)uni(m3[0]); n_n = (int)uni(m3[1]); n_co0 = (int)uni(m3[2]); n_oy0 = (int)uni(m3[3]);
                n_ox0 = (int)uni(m4[0]);
                n_xrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)uni(m2[3]) << 32) | uni(m2[2])), 0,
                                                          (int)uni(m4[1]), 0x00020000);
                prefetch(n_xrs, n_srow, n_rowbad, n_voff, n_voffx, has_x);
            } else {                // (nothing carried over: otherwise the old registers would have to live through the loop)
#pragma unroll
                for (int i = 0; i < 10; ++i) rp[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 9; ++i) n_srow[i] = 0;
            }
            WW_STAMP(13);
        }
        __syncthreads();
        if (nb == 0) WW_STAMP(14);
        const int co = co0 + nb * 32 + 4 * rquad;                              // first of this thread's four channels
        const f32x4 bias4 = biasv[nb];
        f32x2 yo[2][4][4];                                                     // [channel pair][out row][out col]
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            f32x2 cq[6][4];                                                    // A^T over the rows, per transform column
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                f32x2 m[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(ex + rbase + (r * 6 + q) * 4096);
                    m[r] = hp ? hi2w(t4) : lo2w(t4);
                }
                if (q == 1) m[1] = pkw_add(m[1], hp ? hi2w(bias4) : lo2w(bias4));   // A^T[i][1] A^T[j][1] = 1 for every (i, j)
                at4(KOut, m[0], m[1], m[2], m[3], m[4], m[5], cq[q][0], cq[q][1], cq[q][2], cq[q][3]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                at4(KOut, cq[0][i], cq[1][i], cq[2][i], cq[3][i], cq[4][i], cq[5][i], yo[hp][i][0], yo[hp][i][1], yo[hp][i][2], yo[hp][i][3]);
        }
        if (nb == 0) WW_STAMP(15);
        if (STATS) {
            // Per-channel sum and sum of squares of this tile's OUTPUT (bias and ReLU applied, pixels outside the image excluded), the
            // stage-1 statistic (calc_sum, mean_std_computation_effcientMem.py:103-115) straight from the accumulators: per thread over
            // its 16 pixels, over the 32 tiles of the workgroup through LDS in fixed order (bitwise reproducible), one (sum, sq) pair
            // per (spatial tile, channel) to global; ccst_chan_sums_finalize_f32 folds the tiles in fp64.
            f32x4 ts = {0.f, 0.f, 0.f, 0.f}, tq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool in = interior || ((oyb + i < p.H) && (oxb + j < p.W));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float o = yo[k >> 1][i][j][k & 1];
                        o = relu ? fmaxf(o, 0.f) : o;
                        o = in ? o : 0.f;
                        ts[k] += o;
                        tq[k] = __builtin_fmaf(o, o, tq[k]);
                    }
                }
            char* const sred = reinterpret_cast<char*>(lds) + EXW_BYTES + MBW_BYTES;
            *reinterpret_cast<f32x4*>(sred + tid * 32) = ts;
            *reinterpret_cast<f32x4*>(sred + tid * 32 + 16) = tq;
            __syncthreads();
            if (tid < 64) {
                const int c = tid & 31, which = tid >> 5;
                float tot = 0.f;
#pragma unroll 8
                for (int t = 0; t < 32; ++t) tot += *reinterpret_cast<const float*>(sred + (t * 8 + (c >> 2)) * 32 + which * 16 + (c & 3) * 4);
                const int cg = co0 + nb * 32 + c;
                if (cg < p.Cout) p.sums[((long long)(tile_spatial) * p.Cout + cg) * 2 + which] = tot;
            }
        }
        if (co >= p.Cout) continue;
        // 16 bytes per lane and pixel through a buffer resource on image n: the per-lane byte offset carries (tile, channel quad), the
        // scalar offset the pixel (i, j) of the tile; a row below the image falls beyond the resource and is dropped by the hardware,
        // a column right of it gets an offset beyond it.  Channel counts that are not a multiple of 4 take the scalar stores.
        auto emit = [&](auto relu_t, auto edge_t) {
            constexpr bool RELU = decltype(relu_t)::value, EDGE = decltype(edge_t)::value;
            auto act = [](float o) { return RELU ? fmaxf(o, 0.f) : o; };
            const bool vec4 = !EDGE || (co + 3 < p.Cout && (p.Cout & 3) == 0);
            auto put = [&](f32x4 o, unsigned vo, unsigned so) {
                if (vec4) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrs, vo, so, W4W_STORE_AUX);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (co + k < p.Cout) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[k]), yrs, vo + 4u * k, so, 0);
                }
            };
            if (!POOL) {
                const unsigned vo = (unsigned)((oyb * p.ysH + oxb * p.ysW + co) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned vj = (EDGE && oxb + j >= p.W) ? 0x80000000u : vo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 o = {act(yo[0][i][j][0]), act(yo[0][i][j][1]), act(yo[1][i][j][0]), act(yo[1][i][j][1])};
                        put(o, vj, (unsigned)((i * p.ysH + j * p.ysW) * 4));
                    }
                }
            } else {
                const unsigned vo = (unsigned)(((oyb >> 1) * p.ysH + (oxb >> 1) * p.ysW + co) * 4);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int px = oxb + 2 * b;
                    const unsigned vb = (EDGE && px >= p.W) ? 0x80000000u : vo;
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        f32x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float o00 = yo[k >> 1][2 * a][2 * b][k & 1];
                            float o01 = yo[k >> 1][2 * a][2 * b + 1][k & 1], o10 = yo[k >> 1][2 * a + 1][2 * b][k & 1];
                            float o11 = yo[k >> 1][2 * a + 1][2 * b + 1][k & 1];
                            if (EDGE) {                                        // ceil mode: a window at the edge holds 1 or 2 valid pixels
                                const bool cx = px + 1 < p.W, cy = oyb + 2 * a + 1 < p.H;
                                o01 = cx ? o01 : o00;
                                o10 = cy ? o10 : o00;
                                o11 = (cx && cy) ? o11 : o00;
                            }
                            o[k] = act(fmaxf(fmaxf(o00, o01), fmaxf(o10, o11)));
                        }
                        put(o, vb, (unsigned)((a * p.ysH + b * p.ysW) * 4));
                    }
                }
            }
        };
        const bool fast = interior && (p.Cout & 3) == 0;
        if (fast) {
            if (relu) emit(std::true_type{}, std::false_type{});
            else emit(std::false_type{}, std::false_type{});
        } else {
            if (relu) emit(std::true_type{}, std::true_type{});
            else emit(std::false_type{}, std::true_type{});
        }
    }
    WW_STAMP(4);
    __syncthreads();                // the next tile's halo overwrites the exchange area
    WW_STAMP(5);
    c_tn = n_tn; c_n = n_n; c_co0 = n_co0; c_oy0 = n_oy0; c_ox0 = n_ox0; c_hdst = n_hdst; c_hdstx = n_hdstx;
    c_rowbad = n_rowbad; c_voff = n_voff; c_voffx = n_voffx; c_xrs = n_xrs;
#pragma unroll
    for (int i = 0; i < 9; ++i) c_srow[i] = n_srow[i];
    }
}

template <bool POOL, bool STATS>
__global__ __launch_bounds__(NTW) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_wino4w_kernel(const W4wArgs p) {
    extern __shared__ __attribute__((aligned(16))) float wino4w_lds[];         // 2 halo buffers / the epilogue exchange, mailbox, sums
    const int wv = threadIdx.x >> 6;                                           // wave-uniform dispatch on the position block
    if (wv == 0) wino4w_body<POOL, STATS, 0, 0>(p, wino4w_lds);
    else if (wv == 1) wino4w_body<POOL, STATS, 0, 1>(p, wino4w_lds);
    else if (wv == 2) wino4w_body<POOL, STATS, 1, 0>(p, wino4w_lds);
    else wino4w_body<POOL, STATS, 1, 1>(p, wino4w_lds);
}

// OIHW 3x3 -> U[chunk][pos = r*6+q][pair][k half][cout/64][32][group][2], U = G g G^T (accumulated in double);
// input channel = chunk*16 + half*8 + 2*pair + e, output channel = block*64 + group*32 + lane.  (cout, cin) are those of the conv the
// transform serves; bwd: that conv is the backward-data of a forward conv with weight w [cin][cout][3][3] -- channels swapped, taps
// reversed.  Grid-stride over blockIdx.x / gridDim.x.
__device__ __forceinline__ void pack_weight_wino4w_body(const float* __restrict__ w, float* __restrict__ u, int cout, int cin, int cin_pad,
                                                        int cout_pad, int bwd) {
    const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int ncb = cout_pad >> 6;
    const long long total = (long long)(cin_pad / 16) * 36 * 4 * 2 * ncb * 128;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 1), grp = (int)((i >> 1) & 1), lane = (int)((i >> 2) & 31);
        long long t = i >> 7;
        const int cb = (int)(t % ncb);
        t /= ncb;
        const int half = (int)(t & 1);
        t >>= 1;
        const int pair = (int)(t & 3);
        t >>= 2;
        const int pos = (int)(t % 36);
        const int chunk = (int)(t / 36);
        const int r = pos / 6, q = pos - 6 * r;
        const int ci = chunk * 16 + half * 8 + 2 * pair + e;
        const int co = cb * 64 + grp * 32 + lane;
        double val = 0.0;
        if (co < cout && ci < cin) {
            const float* g = bwd ? w + ((long long)ci * cout + co) * 9 : w + ((long long)co * cin + ci) * 9;
            for (int cc = 0; cc < 3; ++cc) {
                const int c0 = bwd ? 8 - cc : cc, st = bwd ? -3 : 3;          // tap (a, cc) of the served conv = tap (2 - a, 2 - cc) of w
                const double gg = G[r][0] * (double)g[c0] + G[r][1] * (double)g[c0 + st] + G[r][2] * (double)g[c0 + 2 * st];
                val += gg * G[q][cc];
            }
        }
        u[i] = (float)val;
    }
}

__global__ void pack_weight_wino4w_kernel(const float* __restrict__ w, float* __restrict__ u, int cout, int cin, int cin_pad, int cout_pad,
                                          int bwd) {
    pack_weight_wino4w_body(w, u, cout, cin, cin_pad, cout_pad, bwd);
}

}  // namespace

static int pack_wino4w_impl(const float* w_oihw, float* u, int cout, int cin, int cout_pad, int bwd, void* stream) {
    CCST_REQUIRE(w_oihw && u && cout > 0 && cin > 0, "pack_wino4w: bad args");
    CCST_REQUIRE(cout_pad >= cout && cout_pad % 64 == 0, "pack_wino4w: cout_pad must be a multiple of 64 >= cout");
    const int cin_pad = (cin + 15) / 16 * 16;
    const long long total = (long long)(cin_pad / 16) * 36 * 16 * cout_pad;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_weight_wino4w_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, u, cout, cin, cin_pad, cout_pad, bwd);
    return ccst_launch_status("pack_weight_wino4w");
}

// floats of a transformed weight: [Cin/16][36][2][cout_pad][8]
extern "C" int64_t ccst_wino4_weight_floats(int cin, int cout_pad) { return (int64_t)((cin + 15) / 16) * 36 * 2 * cout_pad * 8; }

extern "C" int ccst_pack_conv_weight_wino4w_f32(const float* w_oihw, float* u, int cout, int cin, int cout_pad, void* stream) {
    return pack_wino4w_impl(w_oihw, u, cout, cin, cout_pad, 0, stream);
}

extern "C" int ccst_wino4w_spatial_tiles(int N, int H, int W) { return N * ((H + THW - 1) / THW) * ((W + TWW - 1) / TWW); }

extern "C" int ccst_conv3x3_wino4w_f32(const float* x, const float* u_packed, const float* bias, float* y, int N, int H, int W, int Cin,
                                       int Cout, int cout_pad, uint32_t flags, float* chan_sum_partials, void* stream) {
    CCST_REQUIRE(x && u_packed && y, "conv3x3_wino4w: null pointer");
    CCST_REQUIRE(N > 0 && H > 0 && W > 0 && Cin >= 32 && Cin % 16 == 0 && Cout > 0, "conv3x3_wino4w: bad shape (Cin a multiple of 16, >= 32)");
    CCST_REQUIRE(cout_pad >= Cout && cout_pad % 64 == 0, "conv3x3_wino4w: cout_pad must be a multiple of 64 >= cout");
    CCST_REQUIRE(!(flags & ~(CCST_CONV_RELU | CCST_CONV_POOL2 | CCST_CONV_UPS2 | CCST_CONV_REFLECT)), "conv3x3_wino4w: unsupported flag");
    const bool pool = (flags & CCST_CONV_POOL2) != 0, ups = (flags & CCST_CONV_UPS2) != 0;
    if (ups) CCST_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv3x3_wino4w: upsampled extent must be even");
    if (flags & CCST_CONV_REFLECT) CCST_REQUIRE(H >= 2 && W >= 2, "conv3x3_wino4w: reflection needs extent >= 2");
    W4wArgs a;
    a.x = x; a.u = u_packed; a.bias = bias; a.y = y; a.sums = chan_sum_partials;
    a.N = N; a.H = H; a.W = W; a.Hs = ups ? H / 2 : H; a.Ws = ups ? W / 2 : W; a.Cin = Cin; a.Cout = Cout; a.CoutPad = cout_pad;
    a.reflect = (flags & CCST_CONV_REFLECT) ? 1 : 0; a.ups = ups ? 1 : 0; a.relu = (flags & CCST_CONV_RELU) ? 1 : 0;
    CCST_REQUIRE(!(chan_sum_partials && pool), "conv3x3_wino4w: channel sums are of the un-pooled output");
    CCST_REQUIRE((long long)a.Hs * a.Ws * Cin * 4 < (long long)OOBW, "conv3x3_wino4w: one image must be < 2^30 bytes");
    const int oh = pool ? (H + 1) / 2 : H, ow = pool ? (W + 1) / 2 : W;
    a.ysW = Cout; a.ysH = ow * Cout; a.ysN = (long long)oh * ow * Cout;
    CCST_REQUIRE(a.ysN * 4 < 0x7fffffffLL, "conv3x3_wino4w: one output image must be < 2^31 bytes");
    a.tilesN = (Cout + 63) / 64;
    a.tilesY = (H + THW - 1) / THW;
    a.tilesX = (W + TWW - 1) / TWW;
    const long long grid = (long long)N * a.tilesY * a.tilesX * a.tilesN;
    if (grid <= 0 || grid > 0x7fffffffLL) {
        ccst_set_error("conv3x3_wino4w: bad grid %lld", grid);
        return CCST_EINVAL;
    }
    a.ntiles = (int)grid;
    CCST_REQUIRE(grid < (1 << 24) && a.tilesN <= 4096 && a.tilesX <= 4096 && a.tilesY <= 4096, "conv3x3_wino4w: tile grid too large");
    a.mN = ((1ULL << 36) + a.tilesN - 1) / a.tilesN;
    a.mX = ((1ULL << 36) + a.tilesX - 1) / a.tilesX;
    a.mY = ((1ULL << 36) + a.tilesY - 1) / a.tilesY;
    hipStream_t s = (hipStream_t)stream;
    const int n_cu = ccst_num_cus() / 8 * 8 > 0 ? ccst_num_cus() / 8 * 8 : 8;   // a multiple of the 8 XCDs
    const long long wgs = grid < n_cu ? grid : n_cu;
    const bool stats = chan_sum_partials != nullptr;
    const size_t lds = (size_t)EXW_BYTES + MBW_BYTES + (stats ? SRW_BYTES : 0);   // 148.5 (156.5) KB of the CU's 160: one workgroup per CU
    // the opt-in above the 64 KB default is per device and idempotent: set it for the current device on every launch
    const void* kfn = pool ? reinterpret_cast<const void*>(&conv3x3_wino4w_kernel<true, false>)
                           : stats ? reinterpret_cast<const void*>(&conv3x3_wino4w_kernel<false, true>)
                                   : reinterpret_cast<const void*>(&conv3x3_wino4w_kernel<false, false>);
    hipError_t e1 = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e1 != hipSuccess) {
        ccst_set_error("conv3x3_wino4w: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e1));
        return (int)e1;
    }
    if (pool) hipLaunchKernelGGL((conv3x3_wino4w_kernel<true, false>), dim3((unsigned)wgs), dim3(NTW), lds, s, a);
    else if (stats) hipLaunchKernelGGL((conv3x3_wino4w_kernel<false, true>), dim3((unsigned)wgs), dim3(NTW), lds, s, a);
    else hipLaunchKernelGGL((conv3x3_wino4w_kernel<false, false>), dim3((unsigned)wgs), dim3(NTW), lds, s, a);
    return ccst_launch_status("conv3x3_wino4w");
}
