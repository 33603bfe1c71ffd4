// 3x3 stride-1 "same" convolution as Winograd F(2,3) ALONG X ONLY, on half pieces on the 16-bit MFMA (round 4) -- the AdaIN encoder /
// decoder layers with Cout >= 128 (style_transfer/AdaIN/net.py:6-36,38-69).
//
// Why this form.  The direct half-piece kernel (conv3x3_halo.hip, SPLIT) issues three 16-bit MFMAs per fp32 product and is bounded by
// the energy of those MFMAs (the chip runs it power-capped at 1.7-1.8 GHz): only fewer executed MFMAs per output move
// it.  2-D Winograd needs 16 (F(2x2)) or 36 (F(4x4)) accumulator sets per tile: at 16 registers per 32x32 tile a workgroup tile large
// enough to amortise the weight stream does not fit the register file.  The 1-D form keeps the direct kernel's whole structure -- halo
// rows in LDS, the ky taps as plain k-steps -- and replaces the three kx taps by FOUR transform positions per PAIR of output pixels:
//     Y[y][2p + e] = sum_ky sum_q A[e][q] * ( V_q[y + ky][p] . U_q[ky] ),   V = B^T d (d = the four input pixels 2p-1 .. 2p+2),
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  U = G g = [g0; (g0+g1+g2)/2; (g0-g1+g2)/2; g2],  A^T = [1 1 1 0; 0 1 -1 -1]
// i.e. 12 k-steps per 16-channel chunk and pixel pair instead of 18: 2.0 executed MFMA FLOPs per algorithmic FLOP instead of 3.0.
// The price: four accumulator sets (128 registers per wave for a 64 x 32 wave tile), a transformed halo image twice the size of the
// raw one (two positions per pixel), and F(2,3)'s rounding (one extra fp32 add on each side; per layer ~2e-6 of max |y|, test).
//
//   * workgroup = 8 rows x 32 pixels (= 128 GEMM rows: (row, pixel pair)) x 128 output channels, EIGHT waves (2 row halves x 4
//     channel quarters; two waves per SIMD from ONE workgroup -- the LDS image leaves room for one workgroup per CU), wave tile = 64
//     rows x 32 channels x 4 positions = 8 accumulators of 16 registers;
//   * the loader fetches the four raw pixels of a (halo row, pixel pair, 4-channel part) unit, forms the four positions with fp32
//     adds, scales by the tensor's power of two (per-tensor |max| words, common.h: range-safe at any fp32 magnitude), splits into
//     (hi, lo) IEEE-half pieces and writes V[row][pair][position][16 ch hi | 16 ch lo]; pair pitch 272 B and row pitch 4480 B put the
//     sixteen lanes of every ds_read_b128 pass on sixteen different 16-byte bank groups;
//   * weights pre-transformed, scaled and split at pack time ([ky * 4 + q][chunk][cout][16 ch hi | lo], 8 KB per k-step and
//     workgroup), staged through a 3-deep LDS ring two k-steps ahead exactly as in the direct kernel;
//   * epilogue: accumulators scaled back (v_ldexp), A^T applied in registers, bias, ReLU, the 2x2 ceil max-pool (a pooling window =
//     two rows of one pixel pair = four values of one lane), NHWC stores through a buffer resource, max |y| for the next layer.
#include "common.h"

namespace {

struct F23Args {
    const float* x;
    const float* u;
    const float* bias;
    float* y;
    const unsigned* xmax;
    const unsigned* wmax;
    unsigned* ymax;
    float* stats;        // nullptr, or [ccst_conv3x3_f23_tiles(N,H,W)][Cout][4] per-(8x32-pixel tile, wave row) (sum, M2 about the slab's own mean, count, 0) of the output
    int N, H, W, Hs, Ws, Cin, Cout, CoutPad;
    int reflect, ups, relu;
    long long ysN;
    int ysH, ysW;
    int tilesX, tilesY, tilesN;
};

typedef ccst_u32x2 u32x2f;
typedef _Float16 f16x8f __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2f __attribute__((ext_vector_type(2)));
typedef float f32x2f __attribute__((ext_vector_type(2)));

constexpr int F_TH = 8, F_TW = 32, F_XP = F_TW / 2, F_HH = F_TH + 2, F_BN = 128, F_NT = 512;
constexpr int F_QW = 16;                          // words per (pixel pair, position): 16 channels hi (8 words) | 16 channels lo (8 words)
constexpr int F_XPW = 4 * F_QW + 4;               // 68 words = 272 B per pixel pair: 17 sixteen-byte units (odd)
constexpr int F_ROWW = F_XP * F_XPW + 32;         // 1120 words = 4480 B per halo row: 280 units = 8 modulo 16
constexpr int F_VW = F_HH * F_ROWW;               // words per V buffer (44.8 KB)
constexpr int F_RW = F_TW + 2;                    // raw halo pixels per row (34)
constexpr int F_RAW_PIECES = 24;                  // 1 KiB LDS-DMA pieces of the raw halo image: 10 x 34 pixels x 64 B = 21.25, three per wave
constexpr int F_RING = 6;                         // weight stages of 8 KB (128 output channels x 64 B, un-padded, XOR-swizzled parts)
constexpr int F_LDS_BYTES = (2 * F_VW + F_RAW_PIECES * 256 + F_RING * 2048) * 4;      // 163 328 B of the CU's 163 840: one workgroup per CU
// operand scale targets (common.h): a position is a sum of two pixels, a transformed weight of up to three halves
constexpr int F23_X_TARGET = CCST_SPLIT_X_TARGET - 1, F23_W_TARGET = CCST_SPLIT_W_TARGET - 1;

__device__ __forceinline__ int reflect_f(int i, int n) {
    i = (i < 0) ? -i : i;
    i = (i >= n) ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

// four fp32 values scaled by s -> (hi, lo) half pieces, two per word (see split4h in conv3x3_halo.hip)
__device__ __forceinline__ void split4f(f32x4 v, float s, u32x2f& hi, u32x2f& lo) {
    ccst_u32x2 h2, l2;          // (round 5: 8 vector instructions per four values, common.h)
    ccst_split4_half(v, s, h2, l2);
    hi = h2;
    lo = l2;
}

// One LDS-DMA piece: 64 lanes x 16 bytes from (uniform base + per-lane byte offset) to 1 KiB of LDS at the wave-uniform byte address
// lds_addr (lane i lands at + 16 i).  hipcc neither counts it in its s_waitcnt bookkeeping nor waits for it: the kernel's barriers
// below carry hand-counted vmcnt values (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
// wait until all but the wave's N youngest vector-memory operations (= its LDS-DMA pieces) have landed and its own LDS accesses are
// done, then the workgroup barrier: what landed before it may be read by every wave after it
#define F23A_VM_EXTRA 0        // timing experiments only: let this many more pieces stay in flight than is safe
template <int N>
__device__ __forceinline__ void dma_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N == 0 ? 0 : N + F23A_VM_EXTRA) : "memory");
}

template <bool POOL>
__global__ __launch_bounds__(F_NT, 2) void conv3x3_f23_kernel(const F23Args p) {
    extern __shared__ __attribute__((aligned(16))) float f23_lds[];
    float* const Vs = f23_lds;                     // [2][F_VW]             transformed halo, double buffered
    float* const Raw = f23_lds + 2 * F_VW;         // [F_RAW_PIECES * 256]  raw fp32 halo pixels of the NEXT chunk, 64 B per pixel
    float* const Ws = Raw + F_RAW_PIECES * 256;    // [F_RING][2048]        weight stages, 64 B per output channel, 16-byte parts XOR-swizzled
    const unsigned lds0 = (unsigned)(size_t)f23_lds;                           // LDS byte address of the dynamic segment

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;

    int bid = ccst_xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p.tilesN;
    bid /= p.tilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int co0 = tn * F_BN;
    const int oy0 = ty * F_TH, ox0 = tx * F_TW;

    // the |max| words of x and w: loaded here, reduced after the prologue's first barrier -- their latency (the words were written by the
    // previous kernel's atomics: an L2 miss) hides behind the first pieces instead of standing in front of them
    unsigned xword = ccst_absmax_load(p.xmax), wword = ccst_absmax_load(p.wmax);
    const int nchunks = p.Cin / 16;

    // ---- staging: everything comes in by LDS-DMA (no staging registers, so the prefetch depth is LDS, not the register file) -------
    // Why: one workgroup of eight waves per CU is ONE barrier domain -- a wave that waits for a load stalls all eight at the next
    // barrier -- and 250 registers per wave left room for three k-steps of weights in flight (24 KB per CU): the first two versions of
    // this kernel ran at 60 % of their MFMA-only time, waiting on L2.  Now per k-step every wave issues ONE piece of the weight slab of
    // k-step t + 6 into a 6-stage ring (4 k-steps = 32 KB in flight across the barriers, counted vmcnt), and at k-steps 9-11 one piece
    // each of the raw pixels of chunk c + 2; the transform B^T d runs LDS -> registers -> LDS at k-steps 5-8.
    const float* const ximg = p.x + (long long)n * p.Hs * p.Ws * p.Cin;             // this image (uniform): per-lane offsets stay 32-bit
    // weight piece `wave` of a k-step's 8 KB slab: lane -> row r = 16 wave + (lane >> 2), LDS slot k = lane & 3 holds part k ^ f(r),
    // f(r) = (r >> 2) & 3 (the LDS image is lane-linear: the swizzle sits on the SOURCE address and on the fragment read)
    unsigned wsrc;
    {
        const int r = wave * 16 + (lane >> 2), k = lane & 3;
        wsrc = (unsigned)(((co0 + r) * 16 + ((k ^ ((r >> 2) & 3)) * 4)) * 4);
    }
    // raw pieces g = wave + 8 i (i = 0..2) of the 10 x 34-pixel halo: lane -> pixel P = 16 g + (lane >> 2) (clamped: pieces 21.25 .. 23
    // are padding), part lane & 3; reflection / zero padding (the value is zeroed at the transform) / nearest-x2 upsample on the address
    unsigned rsrc_[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int P = min((wave + 8 * i) * 16 + (lane >> 2), F_HH * F_RW - 1);
        const int hy = P / F_RW, hx = P - hy * F_RW;
        int gy = oy0 + hy - 1, gx = ox0 + hx - 1;
        if (p.reflect) {
            gy = reflect_f(gy, p.H);
            gx = reflect_f(gx, p.W);
        } else {
            gy = min(max(gy, 0), p.H - 1);
            gx = min(max(gx, 0), p.W - 1);
        }
        rsrc_[i] = (unsigned)((((gy >> p.ups) * p.Ws + (gx >> p.ups)) * p.Cin + (lane & 3) * 4) * 4);
    }
    // transform items: 10 halo rows x 16 pixel pairs x 4 channel parts x 4 positions = 2560 = FIVE per thread, one per k-step 4..8:
    // thread -> (pixel pair (tid >> 4) & 15, part (tid >> 2) & 3, position q = tid & 3), item i -> halo row (tid >> 8) + 2 i.  A position
    // is d_a + sgn d_b of two of the pair's four raw pixels (B^T rows: q0 = d0 - d2, q1 = d1 + d2, q2 = d2 - d1, q3 = d1 - d3): two 16-byte
    // LDS reads (a 16-lane pass covers three pixels x four parts: conflict-free), fma, split, two 8-byte stores (a pass covers the
    // pair's 256 bytes: conflict-free).  Every thread does the same work at every k-step (the first version did whole units: waves 0
    // and 1 two, the others one, at four k-steps -- the transform cost 13 % of the kernel, most of it the other waves' wait).
    int tsa, tsb, tdst;
    float tsgn;
    unsigned tok = 0x3ffu;          // zero padding: validity of (d_a, d_b) of item i in bits 2 i, 2 i + 1
    {
        const int xp = (tid >> 4) & 15, part = (tid >> 2) & 3, q = tid & 3, hy0 = tid >> 8;
        const int ja = q == 0 ? 0 : q == 2 ? 2 : 1, jb = q == 2 ? 1 : q == 3 ? 3 : 2;
        tsa = ((hy0 * F_RW + 2 * xp + ja) * 16 + part * 4);
        tsb = ((hy0 * F_RW + 2 * xp + jb) * 16 + part * 4);
        tdst = hy0 * F_ROWW + xp * F_XPW + q * F_QW + part * 2;
        tsgn = q == 1 ? 1.f : -1.f;
        if (!p.reflect) {
            tok = 0;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int gy = oy0 + hy0 + 2 * i - 1, gxa = ox0 + 2 * xp - 1 + ja, gxb = ox0 + 2 * xp - 1 + jb;
                const bool oky = (gy >= 0) & (gy < p.H);
                tok |= ((oky & (gxa >= 0) & (gxa < p.W)) ? 1u : 0u) << (2 * i);
                tok |= ((oky & (gxb >= 0) & (gxb < p.W)) ? 1u : 0u) << (2 * i + 1);
            }
        }
    }

    auto dma_w = [&](int c_, int s_) {          // weights of k-step s_ (>= 12: of the next chunk; clamped at the end) -> ring stage s_ % 6
        const int cc = min(c_ + s_ / 12, nchunks - 1), ss = s_ % 12;
        const float* wc = p.u + ((long long)ss * nchunks + cc) * p.CoutPad * 16;         // uniform
        glds16(wc, wsrc, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)((2 * F_VW + F_RAW_PIECES * 256 + (s_ % F_RING) * 2048) * 4 + wave * 1024))));
    };
    auto dma_raw = [&](int c_, int i) {         // raw piece wave + 8 i of chunk c_
        const int cc = min(c_, nchunks - 1);
        glds16(ximg + cc * 16, rsrc_[i], (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)((2 * F_VW) * 4 + (wave + 8 * i) * 1024))));
    };
    float xs = 1.f, xsb = 1.f;       // 2^kx and sgn 2^kx (set after the prologue's first barrier)
    // item i of this thread: raw -> position -> (hi, lo) -> V[buf]
    auto xform = [&](int buf, int i) {
        f32x4 da = *reinterpret_cast<const f32x4*>(&Raw[tsa + i * (2 * F_RW * 16)]);
        f32x4 db = *reinterpret_cast<const f32x4*>(&Raw[tsb + i * (2 * F_RW * 16)]);
        if (!p.reflect) {
            if (!((tok >> (2 * i)) & 1u)) da = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!((tok >> (2 * i + 1)) & 1u)) db = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        u32x2f hi, lo;
#pragma unroll
        for (int h = 0; h < 2; ++h) {          // (d_a + sgn d_b) * 2^kx as d_a * s + d_b * (sgn s): exact scaling, one rounding (the add's)
            const f32x2f pa = f32x2f{da[2 * h], da[2 * h + 1]} * xs;
            const f32x2f pv = f32x2f{db[2 * h], db[2 * h + 1]} * xsb + pa;
            unsigned wh, wl;
            ccst_split2_half(pv[0], pv[1], wh, wl);         // (1 + 2 vector instructions per pair; the convert-back-and-subtract form took 5)
            hi[h] = wh;
            lo[h] = wl;
        }
        float* o = &Vs[buf * F_VW + tdst + i * (2 * F_ROWW)];
        *reinterpret_cast<u32x2f*>(o) = hi;
        *reinterpret_cast<u32x2f*>(o + 8) = lo;
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][mt][r] = 0.f;

    // fragment bases: GEMM row li of M tile Tt = 2 wm + mt is (row 2 Tt + (li & 1), pixel pair li >> 1); weight row wn * 32 + li, piece
    // pc = parts 2 pc + lh -> slot (2 pc + lh) ^ f(row)
    int aBase[2], bBase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) aBase[mt] = (2 * (2 * wm + mt) + (li & 1)) * F_ROWW + (li >> 1) * F_XPW + lh * 4;
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) bBase[pc] = (wn * 32 + li) * 16 + (((2 * pc + lh) ^ ((li >> 2) & 3)) * 4);

    struct Frags {
        f16x8f a[2][2], b[2];         // [piece][M tile], [piece]
    };
    auto read_frags = [&](Frags& f, int vbuf, int ring, int s_) {
        const float* vb = &Vs[vbuf * F_VW + (s_ >> 2) * F_ROWW + (s_ & 3) * F_QW];
        const float* bb = &Ws[ring * 2048];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) f.a[pc][mt] = __builtin_bit_cast(f16x8f, *reinterpret_cast<const f32x4*>(vb + aBase[mt] + 8 * pc));
            f.b[pc] = __builtin_bit_cast(f16x8f, *reinterpret_cast<const f32x4*>(bb + bBase[pc]));
        }
    };

    // ---- prologue: raw pixels of chunk 0 and the weights of k-steps 0..5, transform chunk 0, then the raw pixels of chunk 1 ---------
#pragma unroll
    for (int i = 0; i < 3; ++i) dma_raw(0, i);
#pragma unroll
    for (int s = 0; s < F_RING; ++s) dma_w(0, s);
    dma_barrier<0>();
    asm volatile("" : "+v"(xword), "+v"(wword));      // (opaque here: hipcc otherwise schedules the reduction, and its wait, in front of the first pieces)
    const int kx = ccst_scale_exp(ccst_absmax_reduce(xword), F23_X_TARGET);
    const int kw = ccst_scale_exp(ccst_absmax_reduce(wword), F23_W_TARGET);
    xs = __uint_as_float((unsigned)(127 + kx) << 23);
    xsb = xs * tsgn;
#pragma unroll
    for (int i = 0; i < 5; ++i) xform(0, i);
    dma_barrier<0>();
#pragma unroll
    for (int i = 0; i < 3; ++i) dma_raw(1, i);
    Frags cur, nxt;
    read_frags(cur, 0, 0, 0);

    // one chunk; PAR = its parity = its V buffer (a compile-time constant: the chunks run in pairs, so every LDS offset of the loop is an
    // instruction immediate -- with a run-time parity the buffer base cost three vector adds per k-step)
    auto chunk = [&](const int c, const int PAR) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 12; ++s) {                 // k-step s = ky * 4 + q of chunk c; 12 is a multiple of the ring depth: static stages
            const int q = s & 3;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[q][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[1][mt], cur.b[0], acc[q][mt], 0, 0, 0);   // a_lo b_hi
            __builtin_amdgcn_sched_barrier(0);
            {   // fragments of the NEXT k-step: its weights landed before the previous barrier; V of this chunk, or of the next one -- complete
                // since k-step 8's stores
                const int sn = (s + 1) % 12;
                read_frags(nxt, s == 11 ? PAR ^ 1 : PAR, sn % F_RING, sn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[q][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[0][mt], cur.b[1], acc[q][mt], 0, 0, 0);   // a_hi b_lo
            __builtin_amdgcn_sched_barrier(0);
            // staging.  The raw buffer holds chunk c + 1 (landed before barrier 3: its pieces are older than the weights issued at k-step
            // 11); it is transformed into V[PAR ^ 1] at k-steps 4-8, one item per thread and k-step, and refilled with chunk c + 2 at
            // 9-11.  Order inside a k-step: raw piece first, weight piece second -- the vmcnt table below counts on it.
            if (s >= 9) dma_raw(c + 2, s - 9);
            dma_w(c, s + F_RING);                      // (stage s % 6: its fragments were read during k-step s - 1)
            if (s >= 4 && s <= 8) xform(PAR ^ 1, s - 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[q][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.a[0][mt], cur.b[0], acc[q][mt], 0, 0, 0);   // a_hi b_hi
            __builtin_amdgcn_sched_barrier(0);
            // barrier j: the weights of k-step j + 2 (issued at k-step j - 4) must have landed: all but the pieces issued since -- four
            // weight pieces plus the raw pieces of k-steps j - 3 .. j -- may stay in flight
            switch (s) {          // (folded once the k-step loop is unrolled; the asm operand must be a literal)
                case 0: case 11: dma_barrier<7>(); break;
                case 1: case 10: dma_barrier<6>(); break;
                case 2: case 9: dma_barrier<5>(); break;
                default: dma_barrier<4>(); break;
            }
            cur = nxt;
        }
    };
    {
        int c = 0;
        for (; c + 1 < nchunks; c += 2) {
            chunk(c, 0);
            chunk(c + 1, 1);
        }
        if (c < nchunks) chunk(c, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped prefetches of the last k-steps must land before this workgroup's LDS is released

    // ---- epilogue: scale back, A^T over the four positions, bias --------------------------------------------------------------
    const int co = co0 + wn * 32 + li;
    const float bias = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
    const int ks = -(kx + kw);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = __builtin_ldexpf(acc[0][mt][r], ks), m1 = __builtin_ldexpf(acc[1][mt][r], ks);
            const float m2 = __builtin_ldexpf(acc[2][mt][r], ks), m3 = __builtin_ldexpf(acc[3][mt][r], ks);
            acc[0][mt][r] = (m0 + m1 + m2) + bias;          // pixel 2 xp
            acc[1][mt][r] = (m1 - m2 - m3) + bias;          // pixel 2 xp + 1
        }
    const bool relu = p.relu != 0;
    float amax = 0.f;
    const unsigned peeked = p.ymax != nullptr ? ccst_absmax_peek(p.ymax, blockIdx.x) : 0u;      // (compared after the stores: nobody waits for it)
    const bool interior = (oy0 + F_TH <= p.H) && (ox0 + F_TW <= p.W) && (co0 + F_BN <= p.Cout);
    const bool cok = co < p.Cout;
    // accumulator register r of a lane is GEMM row (r & 3) + 8 (r >> 2) + 4 lh of its tile: row parity r & 1, pixel pair ((r & 3) >> 1) + 4 (r >> 2) + 2 lh
    if (!POOL) {
        float* const tile = p.y + (long long)n * p.ysN + (long long)oy0 * p.ysH + (long long)ox0 * p.ysW + co0 + wn * 32;
        const unsigned lane_off = (unsigned)(4 * lh * p.ysW + li);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 0x7fffffff, 0x00020000);
        float s1 = 0.f, cnt = 0.f;         // p.stats: this wave's 4 rows x 32 pixels of channel `co` (after bias / ReLU, pixels outside the image excluded)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int Tt = 2 * wm + mt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dy = 2 * Tt + (r & 1), xpu = ((r & 3) >> 1) + 4 * (r >> 2);        // + 2 lh pairs = 4 lh pixels (lane_off)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float v = acc[e][mt][r];
                    if (relu) v = fmaxf(v, 0.f);
                    const int dx = 2 * xpu + e;
                    if (interior) {
                        amax = fmaxf(amax, fabsf(v));
                        s1 += v;
                        cnt += 1.f;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, lane_off * 4, (dy * p.ysH + dx * p.ysW) * 4, 0);
                    } else if (cok && oy0 + dy < p.H && ox0 + dx + 4 * lh < p.W) {
                        amax = fmaxf(amax, fabsf(v));
                        s1 += v;
                        cnt += 1.f;
                        tile[(long long)dy * p.ysH + (long long)dx * p.ysW + lane_off] = v;
                    }
                }
            }
        }
        if (p.stats != nullptr) {
            // The per-tile channel statistics the AdaIN step and stage 1 take instead of a pass over the features: (sum, M2, count) with
            // M2 = the sum of squares ABOUT THE SLAB'S OWN MEAN -- a second pass over the values still in registers -- so that the
            // consumer's variance (Chan's merge, fp64) has no E[x^2] - mean^2 cancellation however large |mean| / sigma is (ADVICE r3:
            // raw fp32 sums of x^2 lose the variance once mean^2 >> var).
            s1 += __shfl_xor(s1, 32, 64);
            cnt += __shfl_xor(cnt, 32, 64);
            const float mu = s1 / fmaxf(cnt, 1.f);
            float m2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dy = 2 * (2 * wm + mt) + (r & 1), xpu = ((r & 3) >> 1) + 4 * (r >> 2);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float v = acc[e][mt][r];
                        if (relu) v = fmaxf(v, 0.f);
                        const float dv = v - mu;
                        if (interior || (cok && oy0 + dy < p.H && ox0 + 2 * xpu + e + 4 * lh < p.W)) m2 += dv * dv;
                    }
                }
            m2 += __shfl_xor(m2, 32, 64);
            if (lh == 0 && cok)
                *reinterpret_cast<f32x4*>(p.stats + ((long long)((((n * p.tilesY + ty) * p.tilesX + tx) * 2 + wm)) * p.Cout + co) * 4) = f32x4{s1, m2, cnt, 0.f};
        }
    } else {
        const int Hp = (p.H + 1) >> 1, Wp = (p.W + 1) >> 1;
        const int py0 = oy0 >> 1, px0 = ox0 >> 1;
        float* const tile = p.y + (long long)n * p.ysN + (long long)py0 * p.ysH + (long long)px0 * p.ysW + co0 + wn * 32;
        const unsigned lane_off = (unsigned)(2 * lh * p.ysW + li);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int dyp = 2 * wm + mt;                       // pooled row: the two image rows of M tile Tt
#pragma unroll
            for (int g = 0; g < 8; ++g) {                      // registers 2 g, 2 g + 1: rows 2 Tt, 2 Tt + 1 of pixel pair xpu + 2 lh
                const int xpu = (g & 1) + 4 * (g >> 1);
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

// OIHW 3x3 -> [ky * 4 + q][Cin/16][cout_pad][16 words]: words 0..7 = the chunk's 16 input channels of U_q[ky] = (G g[ky][.])_q as
// half(u * 2^kw), two per word; words 8..15 = half(u * 2^kw - hi).  G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1].
__global__ void pack_weight_f23_kernel(const float* __restrict__ w, unsigned* __restrict__ out, int cout, int cin, int cout_pad,
                                       const unsigned* __restrict__ wmax) {
    const int kw = ccst_scale_exp(ccst_absmax_read(wmax), F23_W_TARGET);
    const int nch = cin / 16;
    const long long total = 12LL * nch * cout_pad * 16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int word = (int)(i & 15);
        long long j = i >> 4;
        const int co = (int)(j % cout_pad);
        j /= cout_pad;
        const int chunk = (int)(j % nch), s = (int)(j / nch);
        const int ky = s >> 2, q = s & 3;
        const int piece = word >> 3, k0 = chunk * 16 + 2 * (word & 7);
        unsigned r = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float u = 0.f;
            if (co < cout) {
                const float* g = w + ((long long)co * cin + k0 + e) * 9 + ky * 3;
                u = q == 0 ? g[0] : q == 1 ? 0.5f * ((g[0] + g[2]) + g[1]) : q == 2 ? 0.5f * ((g[0] + g[2]) - g[1]) : g[2];
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

// Transformed, scaled and split weights of ccst_conv3x3_f23_f32: 12 * cin * cout_pad floats' worth; cin a multiple of 16, cout_pad of 128.
extern "C" int ccst_pack_conv_weight_f23_f32(const float* w_oihw, float* u, int cout, int cin, int cout_pad, const uint32_t* w_absmax,
                                             void* stream) {
    CCST_REQUIRE(w_oihw && u && w_absmax && cout > 0 && cin > 0 && cin % 16 == 0, "pack_f23: bad args (cin a multiple of 16)");
    CCST_REQUIRE(cout_pad >= cout && cout_pad % 128 == 0, "pack_f23: cout_pad must be a multiple of 128 >= cout");
    const long long total = 12LL * cin * cout_pad;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weight_f23_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, reinterpret_cast<unsigned*>(u), cout, cin,
                       cout_pad, w_absmax);
    return ccst_launch_status("pack_weight_f23");
}

// Workgroups the kernel launches for a layer: callers pick it where they fill whole rounds of the chip (one workgroup per CU).
extern "C" int ccst_conv3x3_f23_workgroups(int N, int H, int W, int Cout) {
    return N * ((H + F_TH - 1) / F_TH) * ((W + F_TW - 1) / F_TW) * ((Cout + F_BN - 1) / F_BN);
}

// Rows of chan_sum_partials: one per (image, 8x32-pixel tile, wave row), an image's rows contiguous.
extern "C" int ccst_conv3x3_f23_tiles(int N, int H, int W) { return N * ((H + F_TH - 1) / F_TH) * ((W + F_TW - 1) / F_TW) * 2; }

extern "C" int ccst_conv3x3_f23_f32(const float* x, const uint32_t* x_absmax, const float* u, const uint32_t* w_absmax, const float* bias,
                                    float* y, uint32_t* y_absmax, int N, int H, int W, int Cin, int Cout, int cout_pad, uint32_t flags,
                                    float* chan_sum_partials, void* stream) {
    CCST_REQUIRE(x && u && y && x_absmax && w_absmax, "conv3x3_f23: null pointer (the |max| words of x and w are required)");
    CCST_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0 && Cout > 0, "conv3x3_f23: bad shape");
    CCST_REQUIRE(cout_pad >= Cout && cout_pad % 128 == 0, "conv3x3_f23: cout_pad must be a multiple of 128 >= cout");
    const bool pool = (flags & CCST_CONV_POOL2) != 0, ups = (flags & CCST_CONV_UPS2) != 0;
    CCST_REQUIRE(!(chan_sum_partials && pool), "conv3x3_f23: channel sums are of the un-pooled output");
    if (ups) CCST_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv3x3_f23: upsampled extent must be even");
    if (flags & CCST_CONV_REFLECT) CCST_REQUIRE(H >= 2 && W >= 2, "conv3x3_f23: reflection needs extent >= 2");
    F23Args a;
    a.x = x; a.u = u; a.bias = bias; a.y = y; a.xmax = x_absmax; a.wmax = w_absmax; a.ymax = y_absmax; a.stats = chan_sum_partials;
    a.N = N; a.H = H; a.W = W; a.Hs = ups ? H / 2 : H; a.Ws = ups ? W / 2 : W; a.Cin = Cin; a.Cout = Cout; a.CoutPad = cout_pad;
    a.reflect = (flags & CCST_CONV_REFLECT) ? 1 : 0; a.ups = ups ? 1 : 0; a.relu = (flags & CCST_CONV_RELU) ? 1 : 0;
    CCST_REQUIRE((long long)N * a.Hs * a.Ws * Cin < 0x7fffffffLL, "conv3x3_f23: input must have < 2^31 elements");
    CCST_REQUIRE((long long)a.Hs * a.Ws * Cin < (1LL << 30), "conv3x3_f23: one image must have < 2^30 elements (32-bit byte offsets per image)");
    const int oh = pool ? (H + 1) / 2 : H, ow = pool ? (W + 1) / 2 : W;
    CCST_REQUIRE((long long)N * oh * ow * Cout < 0x7fffffffLL, "conv3x3_f23: output must have < 2^31 elements");
    a.ysW = Cout; a.ysH = ow * Cout; a.ysN = (long long)oh * ow * Cout;
    a.tilesN = (Cout + F_BN - 1) / F_BN;
    a.tilesY = (H + F_TH - 1) / F_TH;
    a.tilesX = (W + F_TW - 1) / F_TW;
    const long long grid = (long long)N * a.tilesY * a.tilesX * a.tilesN;
    CCST_REQUIRE(grid > 0 && grid <= 0x7fffffffLL, "conv3x3_f23: bad grid");
    const void* kfn = pool ? reinterpret_cast<const void*>(&conv3x3_f23_kernel<true>) : reinterpret_cast<const void*>(&conv3x3_f23_kernel<false>);
    // (the opt-in above the 64 KB default is per device and idempotent: set for the current device on every launch)
    hipError_t e1 = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_BYTES);
    if (e1 != hipSuccess) {
        ccst_set_error("conv3x3_f23: cannot reserve %d bytes of LDS: %s", F_LDS_BYTES, hipGetErrorString(e1));
        return (int)e1;
    }
    hipStream_t s = (hipStream_t)stream;
    if (pool) hipLaunchKernelGGL((conv3x3_f23_kernel<true>), dim3((unsigned)grid), dim3(F_NT), F_LDS_BYTES, s, a);
    else hipLaunchKernelGGL((conv3x3_f23_kernel<false>), dim3((unsigned)grid), dim3(F_NT), F_LDS_BYTES, s, a);
    return ccst_launch_status("conv3x3_f23");
}
