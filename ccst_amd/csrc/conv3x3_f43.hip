// 3x3 stride-1 "same" convolution as Winograd F(4,3) ALONG X ONLY, on half pieces on the 16-bit MFMA (round 5) -- the AdaIN encoder /
// decoder layers with Cout >= 128 (style_transfer/AdaIN/net.py:6-36,38-69).
//
// Why this form.  The F(2,3) kernel (conv3x3_f23.hip) runs the MFMA pipe under the board's power limit: what moves it is fewer executed
// MFMAs per output.  F(4,3) along x takes SIX transform positions per QUAD of output pixels where F(2,3) takes four per pair:
//     Y[y][4p + e] = sum_ky sum_q A[e][q] * ( V_q[y + ky][p] . U_q[ky] ),   V = B^T d (d = the six input pixels 4p-1 .. 4p+4),  U = G g
//     B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//     G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//     A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// i.e. 18 k-steps per 16-channel chunk and pixel quad instead of F(2,3)'s 24 (direct: 36): 1.5 executed MFMA FLOPs per algorithmic FLOP.
// The price is rounding (interpolation points +-2: ~3x F(2,3)'s error per layer, 1-2e-6 of max |y|; test) and six accumulator sets.
//
//   * workgroup = 8 rows x 32 pixels (= 64 GEMM rows: (row, pixel quad)) x 128 output channels, EIGHT waves = 2 POSITION GROUPS (q = 0..2,
//     q = 3..5) x 4 channel quarters: a wave holds 64 rows x 32 channels x 3 positions = 6 accumulators of 16 registers (all six
//     positions in one wave would be 192 registers) and runs 9 k-steps of 6 MFMAs per chunk (F(2,3): 12).  The two groups run the same
//     k-step schedule on different positions (the group is a template parameter of the main loop: every LDS offset is an immediate);
//   * the output transform needs all six positions: after the last chunk the groups exchange two partial sums per accumulator
//     element through LDS (the operand images are dead by then) -- group 0 finishes pixels 4p, 4p + 1, group 1 pixels 4p + 2, 4p + 3;
//   * the loader is the F(2,3) kernel's: raw fp32 halo pixels and pre-transformed, scaled, split weights by LDS-DMA with hand-counted
//     vmcnt; 480 threads transform one (halo row, quad, position, 4-channel part) item at each of four k-steps (3-6) per chunk: four 16-byte
//     LDS reads, c0 d_a + c1 d_b + c2 d_c + c3 d_d with the power-of-two scale folded into the coefficients, split, two 8-byte stores;
//   * weights: 18 slabs of 8 KB per chunk in the order the two groups consume them, a 9-stage LDS ring: at k-step t every wave refills
//     the two stages of k-step t (read one k-step earlier) with the slabs of k-steps t + 4 / t + 5.
//   * epilogue as in conv3x3_f23.hip: scale back, A^T, bias, ReLU, the 2x2 ceil max-pool, NHWC stores, max |y|, per-tile statistics.
#include "common.h"

namespace {

struct F43Args {
    const float* x;
    const float* u;
    const float* bias;
    float* y;
    const unsigned* xmax;
    const unsigned* wmax;
    unsigned* ymax;
    float* stats;        // nullptr, or [ccst_conv3x3_f23_tiles(N,H,W)][Cout][4] per-(8x32-pixel tile, position group) (sum, M2 about the slab's own mean, count, 0)
    int N, H, W, Hs, Ws, Cin, Cout, CoutPad;
    int reflect, ups, relu;
    long long ysN;
    int ysH, ysW;
    int tilesX, tilesY, tilesN;
};

typedef ccst_u32x2 u32x2g;
typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));
typedef float f32x2g __attribute__((ext_vector_type(2)));

constexpr int G_TH = 8, G_TW = 32, G_XQ = G_TW / 4, G_HH = G_TH + 2, G_BN = 128, G_NT = 512;
constexpr int G_QW = 16;                          // words per (quad, position): 16 channels hi (8 words) | 16 channels lo (8 words)
constexpr int G_XQW = 6 * G_QW + 4;               // 100 words = 25 sixteen-byte units per quad (9 modulo 16)
constexpr int G_ROWW = G_XQ * G_XQW + 16;         // 816 words = 204 units per halo row (12 modulo 16): the 16 lanes of a fragment read pass
                                                  // -- rows 0..3 x quads 0..3 -- land on 16 different 16-byte bank groups
constexpr int G_VW = G_HH * G_ROWW;               // words per V buffer (32.6 KB)
constexpr int G_RW = G_TW + 2;                    // raw halo pixels per row (34)
constexpr int G_RAW_PIECES = 22;                  // 1 KiB LDS-DMA pieces of the raw halo image: 10 x 34 pixels x 64 B = 21.25
constexpr int G_RING = 9;                         // weight stages of 8 KB (128 output channels x 64 B, XOR-swizzled parts)
constexpr int G_RAW0 = 2 * G_VW, G_WS0 = G_RAW0 + G_RAW_PIECES * 256;
constexpr int G_LDS_BYTES = (G_WS0 + G_RING * 2048) * 4;        // 161 536 B of the CU's 163 840: one workgroup per CU
// operand scale targets (common.h): a position is up to 10 x the largest pixel (|4| + |-5| + |1|); a transformed weight at most 1 x
constexpr int F43_X_TARGET = CCST_SPLIT_X_TARGET - 4, F43_W_TARGET = CCST_SPLIT_W_TARGET - 1;

__device__ __forceinline__ int reflect_g(int i, int n) {
    i = (i < 0) ? -i : i;
    i = (i >= n) ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

// One LDS-DMA piece (see conv3x3_f23.hip): 64 lanes x 16 bytes -> 1 KiB of LDS at the wave-uniform byte address lds_addr
__device__ __forceinline__ void glds16g(const void* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
template <int V>
struct GroupTag {
    static constexpr int value = V;
};
template <int N>
__device__ __forceinline__ void dma_barrier_g() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <bool POOL>
__global__ __launch_bounds__(G_NT, 2) void conv3x3_f43_kernel(const F43Args p) {
    extern __shared__ __attribute__((aligned(16))) float f43_lds[];
    float* const Vs = f43_lds;                     // [2][G_VW]             transformed halo, double buffered
    float* const Raw = f43_lds + G_RAW0;           // [G_RAW_PIECES * 256]  raw fp32 halo pixels of the NEXT chunk, 64 B per pixel
    float* const Ws = f43_lds + G_WS0;             // [G_RING][2048]        weight stages, 64 B per output channel, 16-byte parts XOR-swizzled
    const unsigned lds0 = (unsigned)(size_t)f43_lds;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;

    int bid = ccst_xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p.tilesN;
    bid /= p.tilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int co0 = tn * G_BN;
    const int oy0 = ty * G_TH, ox0 = tx * G_TW;

    unsigned xword = ccst_absmax_load(p.xmax), wword = ccst_absmax_load(p.wmax);
    const int nchunks = p.Cin / 16;

    const float* const ximg = p.x + (long long)n * p.Hs * p.Ws * p.Cin;
    // weight piece `wave` of a slab: lane -> row r = 16 wave + (lane >> 2), LDS slot k = lane & 3 holds part k ^ f(r), f(r) = (r >> 2) & 3
    unsigned wsrc;
    {
        const int r = wave * 16 + (lane >> 2), k = lane & 3;
        wsrc = (unsigned)(((co0 + r) * 16 + ((k ^ ((r >> 2) & 3)) * 4)) * 4);
    }
    // raw pieces g = wave + 8 i (i = 0..2) of the 10 x 34-pixel halo (pieces 22, 23 do not exist: waves 6, 7 fetch piece 21 a second time so
    // that every wave issues the same number of pieces -- the vmcnt table below counts on it)
    unsigned rsrc_[3];
    int rpiece[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        rpiece[i] = min(wave + 8 * i, G_RAW_PIECES - 1);
        const int P = min(rpiece[i] * 16 + (lane >> 2), G_HH * G_RW - 1);
        const int hy = P / G_RW, hx = P - hy * G_RW;
        int gy = oy0 + hy - 1, gx = ox0 + hx - 1;
        if (p.reflect) {
            gy = reflect_g(gy, p.H);
            gx = reflect_g(gx, p.W);
        } else {
            gy = min(max(gy, 0), p.H - 1);
            gx = min(max(gx, 0), p.W - 1);
        }
        rsrc_[i] = (unsigned)((((gy >> p.ups) * p.Ws + (gx >> p.ups)) * p.Cin + (lane & 3) * 4) * 4);
    }
    // transform items: 10 halo rows x 8 quads x 6 positions x 4 channel parts = 1920 = FOUR per thread for 480 threads: thread ->
    // (position q = (tid >> 2) % 6, part tid & 3), item i -> (row, quad) number (tid >> 2) / 6 + 20 i of the 80.  A position is
    // c0 d_a + c1 d_b + c2 d_c + c3 d_d of four of the quad's six raw pixels (the rows of B^T; q = 0, 5 have three terms: c3 = 0).
    int tsrc[4], tdst[4], td1, td2, td3;
    float tc0, tc1, tc2, tc3;
    unsigned tok = 0xffffu;          // zero padding: validity of (d_a .. d_d) of item i in bits 4 i .. 4 i + 3
    const bool titem = tid < 480;
    {
        const int u = tid >> 2, part = tid & 3, q = u % 6, v = u / 6;
        const int da = q == 0 ? 0 : 1, db = (q == 5) ? 3 : 2, dc = q == 0 ? 4 : (q == 5 ? 5 : 3), dd = q == 5 ? 5 : 4;
        tc0 = q == 0 ? 4.f : q == 1 ? -4.f : q == 2 ? 4.f : q == 3 ? -2.f : q == 4 ? 2.f : 4.f;
        tc1 = (q == 0 || q == 5) ? -5.f : (q == 1 || q == 2) ? -4.f : -1.f;
        tc2 = (q == 0 || q == 5) ? 1.f : q == 1 ? 1.f : q == 2 ? -1.f : q == 3 ? 2.f : -2.f;
        tc3 = (q == 0 || q == 5) ? 0.f : 1.f;
        td1 = (db - da) * 16;
        td2 = (dc - da) * 16;
        td3 = (dd - da) * 16;
        if (!p.reflect) tok = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int combo = min(v + 20 * i, 79), hy = combo >> 3, xq = combo & 7;
            tsrc[i] = (hy * G_RW + 4 * xq + da) * 16 + part * 4;
            tdst[i] = hy * G_ROWW + xq * G_XQW + q * G_QW + part * 2;
            if (!p.reflect) {
                const int gy = oy0 + hy - 1, gx = ox0 + 4 * xq - 1;
                const bool oky = (gy >= 0) & (gy < p.H);
                const int xs4[4] = {gx + da, gx + db, gx + dc, gx + dd};
#pragma unroll
                for (int j = 0; j < 4; ++j) tok |= ((oky & (xs4[j] >= 0) & (xs4[j] < p.W)) ? 1u : 0u) << (4 * i + j);
            }
        }
    }

    auto dma_w = [&](int c_, int m_) {          // slab m_ (>= 18: of the next chunk; clamped at the end) -> ring stage m_ % 9
        const int cc = min(c_ + m_ / 18, nchunks - 1), mm = m_ % 18;
        const float* wc = p.u + ((long long)mm * nchunks + cc) * p.CoutPad * 16;         // uniform
        glds16g(wc, wsrc, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)((G_WS0 + (m_ % G_RING) * 2048) * 4 + wave * 1024))));
    };
    auto dma_raw = [&](int c_, int i) {         // raw piece wave + 8 i of chunk c_
        const int cc = min(c_, nchunks - 1);
        glds16g(ximg + cc * 16, rsrc_[i], (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(G_RAW0 * 4 + rpiece[i] * 1024))));
    };
    // item i of this thread: raw -> position -> (hi, lo) -> V[buf]
    auto xform = [&](int buf, int i) {
        if (!titem) return;
        const float* r0 = &Raw[tsrc[i]];
        f32x4 da = *reinterpret_cast<const f32x4*>(r0);
        f32x4 db = *reinterpret_cast<const f32x4*>(r0 + td1);
        f32x4 dc = *reinterpret_cast<const f32x4*>(r0 + td2);
        f32x4 dd = *reinterpret_cast<const f32x4*>(r0 + td3);
        if (!p.reflect) {
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!((tok >> (4 * i)) & 1u)) da = z;
            if (!((tok >> (4 * i + 1)) & 1u)) db = z;
            if (!((tok >> (4 * i + 2)) & 1u)) dc = z;
            if (!((tok >> (4 * i + 3)) & 1u)) dd = z;
        }
        u32x2g hi, lo;
#pragma unroll
        for (int h = 0; h < 2; ++h) {          // the scale 2^kx rides in the coefficients (exact): three roundings, those of the fmas
            f32x2g pv = f32x2g{da[2 * h], da[2 * h + 1]} * tc0;
            pv = f32x2g{db[2 * h], db[2 * h + 1]} * tc1 + pv;
            pv = f32x2g{dc[2 * h], dc[2 * h + 1]} * tc2 + pv;
            pv = f32x2g{dd[2 * h], dd[2 * h + 1]} * tc3 + pv;
            unsigned wh, wl;
            ccst_split2_half(pv[0], pv[1], wh, wl);
            hi[h] = wh;
            lo[h] = wl;
        }
        float* o = &Vs[buf * G_VW + tdst[i]];
        *reinterpret_cast<u32x2g*>(o) = hi;
        *reinterpret_cast<u32x2g*>(o + 8) = lo;
    };

    f32x16 acc[3][2];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][mt][r] = 0.f;

    // fragment bases: GEMM row li of M tile mt is (row 4 mt + (li & 3), quad li >> 2); weight row wn * 32 + li, piece pc = parts
    // 2 pc + lh -> slot (2 pc + lh) ^ f(row)
    int aBase[2], bBase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) aBase[mt] = (4 * mt + (li & 3)) * G_ROWW + (li >> 2) * G_XQW + lh * 4;
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) bBase[pc] = (wn * 32 + li) * 16 + (((2 * pc + lh) ^ ((li >> 2) & 3)) * 4);

    struct Frags {
        f16x8g a[2][2], b[2];         // [piece][M tile], [piece]
    };

    // ---- prologue: raw pixels of chunk 0 and the weight slabs 0..8, transform chunk 0, then the raw pixels of chunk 1 ---------------
#pragma unroll
    for (int i = 0; i < 3; ++i) dma_raw(0, i);
#pragma unroll
    for (int m = 0; m < G_RING; ++m) dma_w(0, m);
    dma_barrier_g<0>();
    asm volatile("" : "+v"(xword), "+v"(wword));
    const int kx = ccst_scale_exp(ccst_absmax_reduce(xword), F43_X_TARGET);
    const int kw = ccst_scale_exp(ccst_absmax_reduce(wword), F43_W_TARGET);
    {
        const float xs = __uint_as_float((unsigned)(127 + kx) << 23);
        tc0 *= xs;
        tc1 *= xs;
        tc2 *= xs;
        tc3 *= xs;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) xform(0, i);
    dma_barrier_g<0>();
#pragma unroll
    for (int i = 0; i < 3; ++i) dma_raw(1, i);

    // The main loop of position group G_ (a compile-time constant: positions, ring stages and V offsets are instruction immediates).
    // k-step t = ky * 3 + j of a chunk: group G_ multiplies position q = 3 G_ + j of halo rows ky .. ky + 7 by slab m = 2 t + G_.
    auto run = [&](auto gtag) __attribute__((always_inline)) {
        constexpr int G_ = decltype(gtag)::value;
        auto read_frags = [&](Frags& f, int vbuf, int t_) {
            const float* vb = &Vs[vbuf * G_VW + (t_ / 3) * G_ROWW + (3 * G_ + t_ % 3) * G_QW];
            const float* bb = &Ws[((2 * t_ + G_) % G_RING) * 2048];
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) f.a[pc][mt] = __builtin_bit_cast(f16x8g, *reinterpret_cast<const f32x4*>(vb + aBase[mt] + 8 * pc));
                f.b[pc] = __builtin_bit_cast(f16x8g, *reinterpret_cast<const f32x4*>(bb + bBase[pc]));
            }
        };
        Frags cur, nxt;
        read_frags(cur, 0, 0);
        auto chunk = [&](const int c, const int PAR) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int j = t % 3;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[1][mt], cur.b[0], acc[j][mt], 0, 0, 0);   // a_lo b_hi
                __builtin_amdgcn_sched_barrier(0);
                {   // fragments of the NEXT k-step: its slabs landed before the previous barrier; V of this chunk, or of the next one -- complete
                    // since k-step 6's stores
                    const int tn_ = (t + 1) % 9;
                    read_frags(nxt, t == 8 ? PAR ^ 1 : PAR, tn_);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[0][mt], cur.b[1], acc[j][mt], 0, 0, 0);   // a_hi b_lo
                __builtin_amdgcn_sched_barrier(0);
                // staging.  The raw buffer holds chunk c + 1 (landed before barrier 1: its pieces are older than the second slab issued at
                // k-step 8; the prologue's: before barrier 2); it is transformed into V[PAR ^ 1] at k-steps 3-6, one item per thread and
                // k-step, and refilled with chunk c + 2 at k-steps 7 (two pieces) and 8.  Order inside a k-step: raw pieces, slab
                // 2 t + 9, slab 2 t + 10 -- the vmcnt table below counts on it.
                if (t == 7) {
                    dma_raw(c + 2, 0);
                    dma_raw(c + 2, 1);
                }
                if (t == 8) dma_raw(c + 2, 2);
                dma_w(c, 2 * t + 9);                   // (stages (2 t) % 9, (2 t + 1) % 9: this k-step's own, read during k-step t - 1)
                dma_w(c, 2 * t + 10);
                if (t >= 3 && t <= 6) xform(PAR ^ 1, t - 3);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[0][mt], cur.b[0], acc[j][mt], 0, 0, 0);   // a_hi b_hi
                __builtin_amdgcn_sched_barrier(0);
                // barrier t: the slabs of k-step t + 2 (2 t + 4, issued at k-step t - 3, and 2 t + 5, issued FIRST at k-step t - 2) must have
                // landed: the five slab pieces issued since, plus the raw pieces of k-steps t - 1 and t, may stay in flight
                switch (t) {
                    case 0: dma_barrier_g<6>(); break;
                    case 7: dma_barrier_g<7>(); break;
                    case 8: dma_barrier_g<8>(); break;
                    default: dma_barrier_g<5>(); break;
                }
                cur = nxt;
            }
        };
        int c = 0;
        for (; c + 1 < nchunks; c += 2) {
            chunk(c, 0);
            chunk(c + 1, 1);
        }
        if (c < nchunks) chunk(c, 0);
    };
    if (grp == 0) run(GroupTag<0>{});
    else run(GroupTag<1>{});
    // the clamped prefetches of the last k-steps must land, and every wave must be past its last fragment read, before the exchange
    // below overwrites the operand images
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- epilogue: scale back, the groups' halves of A^T, exchange, bias ---------------------------------------------------------
    //   group 0 (m0 m1 m2): a0 = m0 + m1 + m2, a1 = m1 - m2, a2 = m1 + m2;     group 1 (m3 m4 m5): b0 = m3 + m4, b1 = 2 (m3 - m4), b2 = 4 b0, b3 = 4 b1 + m5
    //   Y0 = a0 + b0, Y1 = a1 + b1 (finished by group 0: it receives b0, b1);   Y2 = a2 + b2, Y3 = a1 + b3 (group 1: receives a2, a1)
    const int co = co0 + wn * 32 + li;
    const float bias = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
    const int ks = -(kx + kw);
    float* const xch = f43_lds;                    // [8 waves][2 values][2 mt][4 register quads][64 lanes][4 floats] = 128 KB
    f32x16 mine[2][2];                             // [value][mt]: what this wave keeps; acc[0], acc[1] are overwritten with what it sends
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = __builtin_ldexpf(acc[0][mt][r], ks), m1 = __builtin_ldexpf(acc[1][mt][r], ks), m2 = __builtin_ldexpf(acc[2][mt][r], ks);
            if (grp == 0) {
                mine[0][mt][r] = (m0 + m1) + m2;       // a0
                mine[1][mt][r] = m1 - m2;              // a1
                acc[0][mt][r] = m1 + m2;               // a2 -> group 1's Y2
                acc[1][mt][r] = m1 - m2;               // a1 -> group 1's Y3
            } else {
                const float s = m0 + m1, d = m0 - m1;  // (m3 + m4), (m3 - m4)
                mine[0][mt][r] = 4.f * s;              // b2
                mine[1][mt][r] = 8.f * d + m2;         // b3
                acc[0][mt][r] = s;                     // b0 -> group 0's Y0
                acc[1][mt][r] = 2.f * d;               // b1 -> group 0's Y1
            }
        }
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                *reinterpret_cast<f32x4*>(&xch[((((wave * 2 + v) * 2 + mt) * 4 + rq) * 64 + lane) * 4]) =
                    f32x4{acc[v][mt][4 * rq], acc[v][mt][4 * rq + 1], acc[v][mt][4 * rq + 2], acc[v][mt][4 * rq + 3]};
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
        const int other = wave ^ 4;
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(&xch[((((other * 2 + v) * 2 + mt) * 4 + rq) * 64 + lane) * 4]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[v][mt][4 * rq + k] = (mine[v][mt][4 * rq + k] + t[k]) + bias;
                }
    }
    // acc[e][mt][r], e = 0, 1: pixel 4 quad + 2 grp + e of row 4 mt + (r & 3), quad = 2 (r >> 2) + lh
    const bool relu = p.relu != 0;
    float amax = 0.f;
    const unsigned peeked = p.ymax != nullptr ? ccst_absmax_peek(p.ymax, blockIdx.x) : 0u;
    const bool interior = (oy0 + G_TH <= p.H) && (ox0 + G_TW <= p.W) && (co0 + G_BN <= p.Cout);
    const bool cok = co < p.Cout;
    if (!POOL) {
        float* const tile = p.y + (long long)n * p.ysN + (long long)oy0 * p.ysH + (long long)(ox0 + 2 * grp) * p.ysW + co0 + wn * 32;
        const unsigned lane_off = (unsigned)(4 * lh * p.ysW + li);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 0x7fffffff, 0x00020000);
        float s1 = 0.f, cnt = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dy = 4 * mt + (r & 3);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float v = acc[e][mt][r];
                    if (relu) v = fmaxf(v, 0.f);
                    const int dx = 8 * (r >> 2) + e;               // + 4 lh (lane_off) + 2 grp (tile)
                    if (interior) {
                        amax = fmaxf(amax, fabsf(v));
                        s1 += v;
                        cnt += 1.f;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, lane_off * 4, (dy * p.ysH + dx * p.ysW) * 4, 0);
                    } else if (cok && oy0 + dy < p.H && ox0 + 2 * grp + dx + 4 * lh < p.W) {
                        amax = fmaxf(amax, fabsf(v));
                        s1 += v;
                        cnt += 1.f;
                        tile[(long long)dy * p.ysH + (long long)dx * p.ysW + lane_off] = v;
                    }
                }
            }
        }
        if (p.stats != nullptr) {
            // per-(tile, position group) channel statistics (sum, M2 about the slab's own mean, count): see conv3x3_f23.hip
            s1 += __shfl_xor(s1, 32, 64);
            cnt += __shfl_xor(cnt, 32, 64);
            const float mu = s1 / fmaxf(cnt, 1.f);
            float m2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dy = 4 * mt + (r & 3);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float v = acc[e][mt][r];
                        if (relu) v = fmaxf(v, 0.f);
                        const float dv = v - mu;
                        if (interior || (cok && oy0 + dy < p.H && ox0 + 2 * grp + 8 * (r >> 2) + e + 4 * lh < p.W)) m2 += dv * dv;
                    }
                }
            m2 += __shfl_xor(m2, 32, 64);
            if (lh == 0 && cok)
                *reinterpret_cast<f32x4*>(p.stats + ((long long)((((n * p.tilesY + ty) * p.tilesX + tx) * 2 + grp)) * p.Cout + co) * 4) = f32x4{s1, m2, cnt, 0.f};
        }
    } else {
        // a pooling window = rows (2 k, 2 k + 1) x pixels (4 quad + 2 grp, + 1) = four values of one lane: registers (r & 3) = 0, 1 | 2, 3
        const int Hp = (p.H + 1) >> 1, Wp = (p.W + 1) >> 1;
        const int py0 = oy0 >> 1, px0 = (ox0 >> 1) + grp;
        float* const tile = p.y + (long long)n * p.ysN + (long long)py0 * p.ysH + (long long)px0 * p.ysW + co0 + wn * 32;
        const unsigned lane_off = (unsigned)(2 * lh * p.ysW + li);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {                      // registers 2 g, 2 g + 1: rows 4 mt + 2 (g & 1), + 1 of quad 2 (g >> 1) + lh
                const int dyp = 2 * mt + (g & 1), xpu = 4 * (g >> 1);          // pooled row; pooled column 4 (g >> 1) + 2 lh + grp
                if (interior) {
                    float v = fmaxf(fmaxf(acc[0][mt][2 * g], acc[1][mt][2 * g]), fmaxf(acc[0][mt][2 * g + 1], acc[1][mt][2 * g + 1]));
                    if (relu) v = fmaxf(v, 0.f);
                    amax = fmaxf(amax, fabsf(v));
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, lane_off * 4, (dyp * p.ysH + xpu * p.ysW) * 4, 0);
                } else {
                    const int pyp = py0 + dyp, pxp = px0 + xpu + 2 * lh;
                    if (cok && pyp < Hp && pxp < Wp) {
                        const bool okx = (2 * pxp + 1 < p.W), oky = (2 * pyp + 1 < p.H);
                        float v = acc[0][mt][2 * g];
                        if (okx) v = fmaxf(v, acc[1][mt][2 * g]);
                        if (oky) v = fmaxf(v, acc[0][mt][2 * g + 1]);
                        if (okx && oky) v = fmaxf(v, acc[1][mt][2 * g + 1]);
                        if (relu) v = fmaxf(v, 0.f);
                        amax = fmaxf(amax, fabsf(v));
                        tile[(long long)dyp * p.ysH + (long long)xpu * p.ysW + lane_off] = v;
                    }
                }
            }
        }
    }
    if (p.ymax != nullptr) ccst_absmax_publish(p.ymax, amax, blockIdx.x, peeked);
}

// OIHW 3x3 -> [m = 2 t + group][Cin/16][cout_pad][16 words], t = ky * 3 + j, position q = 3 group + j: words 0..7 = the chunk's 16 input
// channels of U_q[ky] = (G g[ky][.])_q as half(u * 2^kw), two per word; words 8..15 = half(u * 2^kw - hi).  G g in double (packed once).
__global__ void pack_weight_f43_kernel(const float* __restrict__ w, unsigned* __restrict__ out, int cout, int cin, int cout_pad,
                                       const unsigned* __restrict__ wmax) {
    const int kw = ccst_scale_exp(ccst_absmax_read(wmax), F43_W_TARGET);
    const int nch = cin / 16;
    const long long total = 18LL * nch * cout_pad * 16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int word = (int)(i & 15);
        long long jj = i >> 4;
        const int co = (int)(jj % cout_pad);
        jj /= cout_pad;
        const int chunk = (int)(jj % nch), m = (int)(jj / nch);
        const int t = m >> 1, ky = t / 3, q = 3 * (m & 1) + t % 3;
        const int piece = word >> 3, k0 = chunk * 16 + 2 * (word & 7);
        unsigned r = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float u = 0.f;
            if (co < cout) {
                const float* g = w + ((long long)co * cin + k0 + e) * 9 + ky * 3;
                const double g0 = g[0], g1 = g[1], g2 = g[2];
                const double ud = q == 0 ? g0 * 0.25 : q == 1 ? -(g0 + g1 + g2) / 6.0 : q == 2 ? -(g0 - g1 + g2) / 6.0
                                : q == 3 ? g0 / 24.0 + g1 / 12.0 + g2 / 6.0 : q == 4 ? g0 / 24.0 - g1 / 12.0 + g2 / 6.0 : g2;
                u = (float)ud;
            }
            const float v = __builtin_ldexpf(u, kw);
            const _Float16 h = (_Float16)v;
            const _Float16 pq = piece ? (_Float16)(v - (float)h) : h;
            r |= (unsigned)__builtin_bit_cast(unsigned short, pq) << (16 * e);
        }
        out[i] = r;
    }
}

}  // namespace

// Transformed, scaled and split weights of ccst_conv3x3_f43_f32: 18 * cin * cout_pad floats' worth; cin a multiple of 16, cout_pad of 128.
extern "C" int ccst_pack_conv_weight_f43_f32(const float* w_oihw, float* u, int cout, int cin, int cout_pad, const uint32_t* w_absmax,
                                             void* stream) {
    CCST_REQUIRE(w_oihw && u && w_absmax && cout > 0 && cin > 0 && cin % 16 == 0, "pack_f43: bad args (cin a multiple of 16)");
    CCST_REQUIRE(cout_pad >= cout && cout_pad % 128 == 0, "pack_f43: cout_pad must be a multiple of 128 >= cout");
    const long long total = 18LL * cin * cout_pad;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weight_f43_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, reinterpret_cast<unsigned*>(u), cout, cin,
                       cout_pad, w_absmax);
    return ccst_launch_status("pack_weight_f43");
}

// Same tile (8 rows x 32 pixels x 128 channels), grid and statistics rows as ccst_conv3x3_f23_f32 (ccst_conv3x3_f23_workgroups / _tiles).
extern "C" int ccst_conv3x3_f43_f32(const float* x, const uint32_t* x_absmax, const float* u, const uint32_t* w_absmax, const float* bias,
                                    float* y, uint32_t* y_absmax, int N, int H, int W, int Cin, int Cout, int cout_pad, uint32_t flags,
                                    float* chan_sum_partials, void* stream) {
    CCST_REQUIRE(x && u && y && x_absmax && w_absmax, "conv3x3_f43: null pointer (the |max| words of x and w are required)");
    CCST_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0 && Cout > 0, "conv3x3_f43: bad shape");
    CCST_REQUIRE(cout_pad >= Cout && cout_pad % 128 == 0, "conv3x3_f43: cout_pad must be a multiple of 128 >= cout");
    const bool pool = (flags & CCST_CONV_POOL2) != 0, ups = (flags & CCST_CONV_UPS2) != 0;
    CCST_REQUIRE(!(chan_sum_partials && pool), "conv3x3_f43: channel sums are of the un-pooled output");
    if (ups) CCST_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv3x3_f43: upsampled extent must be even");
    if (flags & CCST_CONV_REFLECT) CCST_REQUIRE(H >= 2 && W >= 2, "conv3x3_f43: reflection needs extent >= 2");
    F43Args a;
    a.x = x; a.u = u; a.bias = bias; a.y = y; a.xmax = x_absmax; a.wmax = w_absmax; a.ymax = y_absmax; a.stats = chan_sum_partials;
    a.N = N; a.H = H; a.W = W; a.Hs = ups ? H / 2 : H; a.Ws = ups ? W / 2 : W; a.Cin = Cin; a.Cout = Cout; a.CoutPad = cout_pad;
    a.reflect = (flags & CCST_CONV_REFLECT) ? 1 : 0; a.ups = ups ? 1 : 0; a.relu = (flags & CCST_CONV_RELU) ? 1 : 0;
    CCST_REQUIRE((long long)N * a.Hs * a.Ws * Cin < 0x7fffffffLL, "conv3x3_f43: input must have < 2^31 elements");
    CCST_REQUIRE((long long)a.Hs * a.Ws * Cin < (1LL << 30), "conv3x3_f43: one image must have < 2^30 elements (32-bit byte offsets per image)");
    const int oh = pool ? (H + 1) / 2 : H, ow = pool ? (W + 1) / 2 : W;
    CCST_REQUIRE((long long)N * oh * ow * Cout < 0x7fffffffLL, "conv3x3_f43: output must have < 2^31 elements");
    a.ysW = Cout; a.ysH = ow * Cout; a.ysN = (long long)oh * ow * Cout;
    a.tilesN = (Cout + G_BN - 1) / G_BN;
    a.tilesY = (H + G_TH - 1) / G_TH;
    a.tilesX = (W + G_TW - 1) / G_TW;
    const long long grid = (long long)N * a.tilesY * a.tilesX * a.tilesN;
    CCST_REQUIRE(grid > 0 && grid <= 0x7fffffffLL, "conv3x3_f43: bad grid");
    const void* kfn = pool ? reinterpret_cast<const void*>(&conv3x3_f43_kernel<true>) : reinterpret_cast<const void*>(&conv3x3_f43_kernel<false>);
    hipError_t e1 = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES);
    if (e1 != hipSuccess) {
        ccst_set_error("conv3x3_f43: cannot reserve %d bytes of LDS: %s", G_LDS_BYTES, hipGetErrorString(e1));
        return (int)e1;
    }
    hipStream_t s = (hipStream_t)stream;
    if (pool) hipLaunchKernelGGL((conv3x3_f43_kernel<true>), dim3((unsigned)grid), dim3(G_NT), G_LDS_BYTES, s, a);
    else hipLaunchKernelGGL((conv3x3_f43_kernel<false>), dim3((unsigned)grid), dim3(G_NT), G_LDS_BYTES, s, a);
    return ccst_launch_status("conv3x3_f43");
}
