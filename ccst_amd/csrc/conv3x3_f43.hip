// 3x3 stride-1 "same" convolution as Winograd F(4,3) ALONG X ONLY, on half pieces on the 16-bit MFMA (round 5) -- the AdaIN encoder /
// decoder layers with Cout >= 128 (style_transfer/AdaIN/net.py:6-36,38-69).
//
// Why this form.  Round 4's F(2,3)-along-x kernel (retired in round 6; JOURNAL.md 3.11) ran the MFMA pipe under the board's power limit: what
// moves it is fewer executed MFMAs per output.  F(4,3) along x takes SIX transform positions per QUAD of output pixels where F(2,3) takes four per pair:
//     Y[y][4p + e] = sum_ky sum_q A[e][q] * ( V_q[y + ky][p] . U_q[ky] ),   V = B^T d (d = the six input pixels 4p-1 .. 4p+4),  U = G g
//     B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//     G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//     A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// i.e. 18 k-steps per 16-channel chunk and pixel quad instead of F(2,3)'s 24 (direct: 36): 1.5 executed MFMA FLOPs per algorithmic FLOP.
// The price is rounding (interpolation points +-2: ~3x F(2,3)'s error per layer, 1-5e-6 of max |y|; test) and six accumulator sets.
//
// Version 1 of this kernel kept the F(2,3) kernel's structure (weights through an LDS ring, a barrier per k-step) and was bound by LDS
// bandwidth: 755 KB per chunk through the CU's 128 B/clock port, overlapping badly with the MFMAs (timing ablation: everything but the
// MFMAs took 152 of the 217 us of a 256 -> 256 layer, the MFMAs alone 137).  Version 2 (four waves of 512 registers, 64 x 64 wave
// tiles) cut the traffic to a third but lost the second wave per SIMD that hides LDS latency: slower.  This version keeps eight waves:
//   * workgroup = 8 rows x 32 pixels (= 64 GEMM rows: (row, pixel quad)) x 128 output channels, EIGHT waves = 2 POSITION GROUPS (q = 0..2,
//     q = 3..5) x 4 channel quarters: a wave holds 64 rows x 32 channels x 3 positions = 6 accumulators and runs 9 k-steps of 6 MFMAs
//     per chunk; the group is a template parameter of the main loop (every LDS offset is an immediate);
//   * the WEIGHTS NEVER TOUCH LDS: every wave loads its own B fragments (32 channels x 16 k x hi | lo = 2 KB per k-step, packed in
//     fragment order) straight into a ring of THREE register sets, each refilled with the slab of three k-steps ahead right after its
//     last MFMA: no barrier for the weights (LDS: 755 -> 415 KB per chunk).  (Nine sets -- a whole chunk ahead -- were no faster and
//     cost the 40 registers the transform's prefetch now uses; the four-wave tile, which transforms between two chunks, keeps nine);
//   * LDS holds the transformed halo (double buffered) and the raw fp32 halo of the next chunk (LDS-DMA): TWO barriers per chunk;
//   * the transform works on PAIRS of positions that share their pixels -- (1,2) and (3,4): four 16-byte reads for two positions,
//     (0,5): six -- 960 items per chunk, two per thread; the scale 2^kx rides in the coefficients;
//   * the output transform needs all six positions: after the last chunk the groups exchange two partial sums per accumulator
//     element through LDS -- group 0 finishes pixels 4p, 4p + 1, group 1 pixels 4p + 2, 4p + 3;
//   * epilogue: scale back, A^T, bias, ReLU, the 2x2 ceil max-pool, NHWC stores, max |y|, per-tile statistics.
#include "common.h"

namespace {

struct F43Args {
    const float* x;
    const float* u;
    const float* bias;
    float* y;
    const unsigned* xmax;    // [N][CCST_ABSMAX_WORDS]: every image its own words (a sample's scale -- and bits -- do not depend on its batch-mates)
    const unsigned* wmax;
    unsigned* ymax;          // nullptr, or zeroed [N][CCST_ABSMAX_WORDS]
    float* stats;        // nullptr, or [ccst_conv3x3_f43_tiles(N,H,W)][Cout][4] per-(8x32-pixel tile, position group) (sum, M2 about the slab's own mean, count, max |y|)
    const float* aff_a;  // AFF: [N][Cin] per-(image, input channel) scale and shift applied to x ON ITS WAY INTO THE TRANSFORM: the conv of a x + b
    const float* aff_b;  //      (the AdaIN normalise + alpha blend folded into the decoder's first conv: no pass over the features)
    int N, H, W, Hs, Ws, Cin, Cout, CoutPad;
    int reflect, ups, relu;
    long long ysN;
    int ysH, ysW;
    int tilesX, tilesY, tilesN;
    int ubytes;
};

typedef ccst_u32x2 u32x2g;
typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));
typedef float f32x2g __attribute__((ext_vector_type(2)));

constexpr int G_TW = 32, G_XQ = G_TW / 4;
// the tile: 8 rows x 32 pixels x 128 output channels, eight waves (two position groups x four channel quarters), one workgroup per CU; or
// HALF (the layers with Cout <= 64): 8 rows x 32 pixels x 64 output channels, FOUR waves (two groups x two channel halves) and 64 KB of
// LDS: TWO workgroups per CU with barriers of their own -- one's prologue / epilogue runs under the other's k-loop (a four-chunk tile
// spends 40 % of its time there).  ONE V buffer: a chunk's nine k-steps, then the transform of the next chunk (the other workgroup has
// the MFMA pipe meanwhile).  (A 16-row x 64-channel tile of eight waves, one per CU, measured 5 % slower on those layers.)
template <bool HALF>
struct Tile {
    static constexpr int NT = HALF ? 256 : 512, WAVES = NT / 64;
    static constexpr int TH = 8, HH = TH + 2, BN = HALF ? 64 : 128;
    static constexpr int VW = HH * (G_XQ * (6 * 16 + 4) + 16);                  // words per V buffer (HH * G_ROWW)
    // raw halo image, 64 B per pixel, ONE PAD SLOT after every four pixels -- pixel hx sits in slot hx + (hx >> 2), a quad's six pixels
    // at slots 5 xq + {0, 1, 2, 3, 5, 6}: constant offsets, and the four consecutive quads of a 16-lane LDS pass 320 B apart = on four
    // different 64-byte bank groups
    static constexpr int RSLOTS = 42;
    static constexpr int RAW_PIECES = 27;                                       // 1 KiB LDS-DMA pieces (16 slots each): 10 x 42 slots = 26.25
    static constexpr int RAW_PER_WAVE = (RAW_PIECES + WAVES - 1) / WAVES;       // 4 / 7
    static constexpr int RAW0 = (HALF ? 1 : 2) * VW;
    static constexpr int OPERAND_BYTES = (RAW0 + RAW_PIECES * 256) * 4;         // 92 928 / 60 288 B
    static constexpr int XCH_BYTES = WAVES * 16384;                             // the epilogue's exchange
    static constexpr int LDS_BYTES = OPERAND_BYTES > XCH_BYTES ? OPERAND_BYTES : XCH_BYTES;
    static_assert(LDS_BYTES <= (HALF ? 81856 : 163840), "LDS layout");
};
constexpr int G_QW = 16;                          // words per (quad, position): 16 channels hi (8 words) | 16 channels lo (8 words)
constexpr int G_XQW = 6 * G_QW + 4;               // 100 words = 25 sixteen-byte units per quad (9 modulo 16)
constexpr int G_ROWW = G_XQ * G_XQW + 16;         // 816 words = 204 units per halo row (12 modulo 16): the 16 lanes of a fragment read pass
                                                  // -- rows 0..3 x quads 0..3 -- land on 16 different 16-byte bank groups
constexpr int G_RW = G_TW + 2;                    // raw halo pixels per row (34)
static_assert(Tile<false>::VW == 10 * G_ROWW, "V layout");
// operand scale targets (common.h): a position is up to 10 x the largest pixel (|4| + |-5| + |1|); a transformed weight at most 1 x
constexpr int F43_X_TARGET = CCST_SPLIT_X_TARGET - 4, F43_W_TARGET = CCST_SPLIT_W_TARGET - 1;

__device__ __forceinline__ int reflect_g(int i, int n) {
    i = (i < 0) ? -i : i;
    i = (i >= n) ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

// One LDS-DMA piece: 64 lanes x 16 bytes from (uniform base + per-lane byte offset) to 1 KiB of LDS at the wave-uniform byte address
// lds_addr (lane i lands at + 16 i).  hipcc neither counts it in its s_waitcnt bookkeeping nor waits for it: the loop's barrier at k-step
// 2 carries a hand-counted vmcnt (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16g(const void* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
// (compile-time integers handed to generic lambdas)
template <int V>
struct GroupTag {
    static constexpr int value = V;
};

// ZP: zero padding (the masks cost 16-24 vector instructions per transform item: the reflecting AdaIN layers run the kernel without them)
// NT: non-temporal output stores -- for outputs the caches cannot hold until the next layer reads them (>= 256 MB: the host decides)
// AFF: the conv of the per-(image, channel) affine map a x + b (p.aff_a, p.aff_b) of x -- applied to the TRANSFORMED values: B^T is linear and
//     with reflection padding every pixel of a quad is a real one, so  B^T (a d + b 1) = a B^T d + b B^T 1,  B^T 1 = (0, -6, 0, 0, 0, 0): one
//     multiplication per transformed value and one constant at position 1 -- half the vector work of mapping the 4-6 loaded pixels of an
//     item (measured: +16 us on the 512 -> 256 layer that way, the whole gain of the removed pass).  Reflecting layers only (zero padding
//     would have to keep b out of the border).  p.xmax are the words of the MAPPED tensor.  The 2 x Cin coefficients of the workgroup's
//     image sit in LDS behind the raw halo (inside the allocation the epilogue's exchange needs anyway).
template <bool POOL, bool ZP, bool HALF, bool NT, bool AFF>
__global__ __launch_bounds__(Tile<HALF>::NT, 2) void conv3x3_f43_kernel(const F43Args p) {
    typedef Tile<HALF> T;
    constexpr int G_TH = T::TH, G_HH = T::HH, G_BN = T::BN, G_VW = T::VW, G_RSLOTS = T::RSLOTS, G_RAW_PIECES = T::RAW_PIECES, G_RAW0 = T::RAW0,
                  NRAW = T::RAW_PER_WAVE;
    extern __shared__ __attribute__((aligned(16))) float f43_lds[];
    float* const Vs = f43_lds;                     // [2][G_VW]             transformed halo, double buffered
    float* const Raw = f43_lds + G_RAW0;           // [G_RAW_PIECES * 256]  raw fp32 halo pixels of the NEXT chunk
    float* const Aff = f43_lds + T::OPERAND_BYTES / 4;      // AFF: [2][Cin] scale | shift of this image
    const unsigned lds0 = (unsigned)(size_t)f43_lds;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = HALF ? wave >> 1 : wave >> 2, wn = HALF ? (wave & 1) : (wave & 3);      // position group, channel part
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

    unsigned xword = ccst_absmax_load(p.xmax + n * CCST_ABSMAX_WORDS), wword = ccst_absmax_load(p.wmax);
    unsigned* const ymax_n = p.ymax != nullptr ? p.ymax + n * CCST_ABSMAX_WORDS : nullptr;
    const int nchunks = p.Cin / 16;

    const float* const ximg = p.x + (long long)n * p.Hs * p.Ws * p.Cin;
    // raw pieces g = wave + 8 i (i = 0..3) of the 10 x 42-slot halo image (pieces 27..31 do not exist: those waves fetch piece 26 again so
    // that every wave issues the same number of pieces -- the vmcnt at k-step 2 counts on it); lane -> slot 16 g + (lane >> 2), part
    // lane & 3; reflection / zero padding (the value is zeroed at the transform) / nearest-x2 upsample on the address
    unsigned rsrc_[NRAW];
    int rpiece[NRAW];
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {
        rpiece[i] = min(wave + T::WAVES * i, G_RAW_PIECES - 1);
        const int S = min(rpiece[i] * 16 + (lane >> 2), G_HH * G_RSLOTS - 1);
        const int hy = S / G_RSLOTS, sx = S - hy * G_RSLOTS, hx = min(sx - sx / 5, G_RW - 1);
        int gy = oy0 + hy - 1, gx = ox0 + hx - 1;
        if (!ZP) {
            gy = reflect_g(gy, p.H);
            gx = reflect_g(gx, p.W);
        } else {
            gy = min(max(gy, 0), p.H - 1);
            gx = min(max(gx, 0), p.W - 1);
        }
        rsrc_[i] = (unsigned)((((gy >> p.ups) * p.Ws + (gx >> p.ups)) * p.Cin + (lane & 3) * 4) * 4);
    }
    auto dma_raw = [&](int c_, int i) {         // raw piece wave + 8 i of chunk c_ (clamped: the second-to-last chunk fetches the last once more)
        glds16g(ximg + min(c_, nchunks - 1) * 16, rsrc_[i], (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(G_RAW0 * 4 + rpiece[i] * 1024))));
    };

    // ---- transform items: 3 position pairs x 10 halo rows x 8 quads x 4 channel parts = 960 = (up to) TWO per thread.  Half-wave
    // 16 i + 2 wave + (lane >> 5) of item slot i takes (pair, row) task number h = that (30, 31: idle): pair = h / 10, row = h % 10;
    // inside a task lane -> (quad (lane >> 2) & 7, part lane & 3).  The pair is wave-uniform (the boundaries are even) and selects a code
    // path in which it is a compile-time constant; the row sits in the thread's base addresses: every LDS offset of the loop is an
    // instruction immediate (with run-time rows hipcc hoisted a vector add per access out of the loop and spilled 340 registers).
    //     slot 0: waves 0-4 pair 0 rows 2 w + lh; waves 5-7 pair 1 rows 2 w + lh - 10;   slot 1: waves 0-1 pair 1 rows 6 + 2 w + lh; waves 2-6 pair 2 rows 2 w + lh - 4
    const int part_t = lane & 3;
    const int xq_t = (lane >> 2) & 7;
    //   HALF (four waves: up to FOUR items per thread, between two chunks): half-wave 8 s + 2 wave + (lane >> 5) of item slot s takes (pair,
    //   row) task number h = that (30, 31: idle), pair = h / 10, row = h % 10.  The even part of the row is wave-uniform but not a
    //   constant: one scalar-plus-vector add per item and side, kept inside the loop (laundered: hoisted out they would cost registers).
    const int trow0 = HALF ? lh : 2 * wave + lh - (wave < 5 ? 0 : 10), trow1 = HALF ? lh : (wave < 2 ? 6 + 2 * wave + lh : min(2 * wave + lh - 4, 9));
    const int tsrcA = (trow0 * G_RSLOTS + 5 * xq_t) * 16 + part_t * 4, tsrcB = (trow1 * G_RSLOTS + 5 * xq_t) * 16 + part_t * 4;
    const int tdstA = trow0 * G_ROWW + xq_t * G_XQW + part_t * 2, tdstB = trow1 * G_ROWW + xq_t * G_XQW + part_t * 2;
    unsigned tokx = 0x3fu;       // zero padding: validity of the quad's six pixels;
    bool okyA = true, okyB = true;   // ... of the rows of item 0, of item 1 (HALF: computed per item)
    if (ZP) {
        tokx = 0;
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            const int gx = ox0 + 4 * xq_t - 1 + d;
            tokx |= (((gx >= 0) & (gx < p.W)) ? 1u : 0u) << d;
        }
        const int gyA = oy0 + trow0 - 1, gyB = oy0 + trow1 - 1;
        okyA = (gyA >= 0) & (gyA < p.H);
        okyB = (gyB >= 0) & (gyB < p.H);
    }
    float xs = 1.f;          // 2^kx (set in the prologue)
    int aff_c = 0;           // AFF: the chunk the transform is working on (its coefficients: Aff[16 aff_c + 4 part ..], Aff[Cin + ..])
    f32x4 fa = {1.f, 1.f, 1.f, 1.f}, fb = {0.f, 0.f, 0.f, 0.f};
    auto affine_coef = [&]() __attribute__((always_inline)) {          // once per transform item: the item's four channels of chunk aff_c
        if (!AFF) return;
        int co_ = aff_c * 16;
        asm volatile("" : "+s"(co_));                // (a scalar add per item, not a hoisted vector one)
        fa = *reinterpret_cast<const f32x4*>(Aff + co_ + part_t * 4);
        fb = *reinterpret_cast<const f32x4*>(Aff + p.Cin + co_ + part_t * 4);
    };
    static_assert(!(AFF && ZP), "the fused input affine is applied to transformed values: reflecting layers only");
    auto put = [&](float* o, f32x4 v) {          // four scaled fp32 values -> (hi, lo) half pieces -> V
        u32x2g hi, lo;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned a, b;
            ccst_split2_half(v[2 * h], v[2 * h + 1], a, b);
            hi[h] = a;
            lo[h] = b;
        }
        *reinterpret_cast<u32x2g*>(o) = hi;
        *reinterpret_cast<u32x2g*>(o + 8) = lo;
    };
    // one item: raw -> the two positions of pair PAIR -> (hi, lo) -> V.  (The scale rides in the coefficients: c * 2^kx is exact, every
    // position costs the roundings of its fused multiply-adds only.)
    auto xpair = [&](auto ptag, const float* r0, float* o, bool oky) __attribute__((always_inline)) {
        constexpr int PAIR = decltype(ptag)::value;
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        affine_coef();
        auto px = [&](int d) {                   // pixel d of the quad (0..5)
            f32x4 v = *reinterpret_cast<const f32x4*>(r0 + (d < 4 ? d : d + 1) * 16);
            if (ZP && !(oky && ((tokx >> d) & 1u))) v = z;
            return v;
        };
        if (PAIR < 2) {
            // positions (1, 2): c1 = -4, c2 = 4, c3 = 1;   (3, 4): c1 = -1, c2 = 2, c3 = 2:
            //   s = c1 d2 + d4,   V_a = -c2 d1 + (c3 d3 + s),   V_b = c2 d1 + (-c3 d3 + s)
            const float c1 = PAIR == 0 ? -4.f * xs : -xs, c2 = PAIR == 0 ? 4.f * xs : 2.f * xs, c3 = PAIR == 0 ? xs : 2.f * xs;
            const f32x4 d2 = px(2), d4 = px(4);
            const f32x4 sc = d2 * c1 + d4 * xs;
            const f32x4 d1 = px(1), d3 = px(3);
            f32x4 va = d1 * (-c2) + (d3 * c3 + sc), vb = d1 * c2 + (sc - d3 * c3);
            if (AFF) {
                va = PAIR == 0 ? va * fa + fb * (-6.f * xs) : va * fa;          // (position 1 carries b B^T 1)
                vb = vb * fa;
            }
            put(o + (PAIR == 0 ? 1 : 3) * G_QW, va);
            put(o + (PAIR == 0 ? 2 : 4) * G_QW, vb);
        } else {
            // positions 0: 4 d0 - 5 d2 + d4;   5: 4 d1 - 5 d3 + d5
            const float c4 = 4.f * xs, c5 = -5.f * xs;
            const f32x4 d0 = px(0), d2 = px(2), d4 = px(4);
            f32x4 v0 = d0 * c4 + (d2 * c5 + d4 * xs);
            if (AFF) v0 = v0 * fa;
            put(o, v0);
            const f32x4 d1 = px(1), d3 = px(3), d5 = px(5);
            f32x4 v5 = d1 * c4 + (d3 * c5 + d5 * xs);
            if (AFF) v5 = v5 * fa;
            put(o + 5 * G_QW, v5);
        }
    };
    // The same in two halves for the eight-wave loop: the item's pixels are requested one k-step before they are used (xr), so that the
    // LDS latency passes under that k-step's MFMAs instead of in front of the transform.
    f32x4 xr[6];
    auto xl = [&](auto ptag, const float* r0, bool oky) __attribute__((always_inline)) {
        constexpr int PAIR = decltype(ptag)::value;
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = (PAIR < 2 ? 1 : 0); d < (PAIR < 2 ? 5 : 6); ++d) {
            f32x4 v = *reinterpret_cast<const f32x4*>(r0 + (d < 4 ? d : d + 1) * 16);
            if (ZP && !(oky && ((tokx >> d) & 1u))) v = z;
            xr[d] = v;
        }
    };
    auto xc = [&](auto ptag, float* o) __attribute__((always_inline)) {
        constexpr int PAIR = decltype(ptag)::value;
        affine_coef();
        if (PAIR < 2) {
            const float c1 = PAIR == 0 ? -4.f * xs : -xs, c2 = PAIR == 0 ? 4.f * xs : 2.f * xs, c3 = PAIR == 0 ? xs : 2.f * xs;
            const f32x4 sc = xr[2] * c1 + xr[4] * xs;
            f32x4 va = xr[1] * (-c2) + (xr[3] * c3 + sc), vb = xr[1] * c2 + (sc - xr[3] * c3);
            if (AFF) {
                va = PAIR == 0 ? va * fa + fb * (-6.f * xs) : va * fa;
                vb = vb * fa;
            }
            put(o + (PAIR == 0 ? 1 : 3) * G_QW, va);
            put(o + (PAIR == 0 ? 2 : 4) * G_QW, vb);
        } else {
            const float c4 = 4.f * xs, c5 = -5.f * xs;
            f32x4 v0 = xr[0] * c4 + (xr[2] * c5 + xr[4] * xs), v5 = xr[1] * c4 + (xr[3] * c5 + xr[5] * xs);
            if (AFF) {
                v0 = v0 * fa;
                v5 = v5 * fa;
            }
            put(o, v0);
            put(o + 5 * G_QW, v5);
        }
    };
    auto xload = [&](int i) __attribute__((always_inline)) {
        if (i == 0) {
            if (wave < 5) xl(GroupTag<0>{}, &Raw[tsrcA], okyA);
            else xl(GroupTag<1>{}, &Raw[tsrcA], okyA);
        } else {
            if (wave < 2) xl(GroupTag<1>{}, &Raw[tsrcB], okyB);
            else if (wave < 7) xl(GroupTag<2>{}, &Raw[tsrcB], okyB);
        }
    };
    auto xcomp = [&](int dA, int dB, int i) __attribute__((always_inline)) {
        if (i == 0) {
            if (wave < 5) xc(GroupTag<0>{}, &Vs[dA]);
            else xc(GroupTag<1>{}, &Vs[dA]);
        } else {
            if (wave < 2) xc(GroupTag<1>{}, &Vs[dB]);
            else if (wave < 7) xc(GroupTag<2>{}, &Vs[dB]);
        }
    };
    // (dA, dB: the thread's destinations in the V buffer being filled -- tdstA / tdstB of buffer 0 or 1; the loop swaps them per chunk)
    auto xform = [&](int dA, int dB, int i) __attribute__((always_inline)) {
        if (HALF) {           // all four slots in one call (i unused)
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int u = 8 * sl + 2 * wave;
                if (u >= 30) continue;
                const int pair = u / 10;
                int re = u - 10 * pair;                      // wave-uniform even row; + lh in the thread's bases
                asm volatile("" : "+s"(re));
                const float* r0 = &Raw[tsrcA + re * (G_RSLOTS * 16)];
                float* o = &Vs[dA + re * G_ROWW];
                const int gy = oy0 + re + lh - 1;
                const bool oky = !ZP || ((gy >= 0) & (gy < p.H));
                if (sl <= 1 && pair == 0) xpair(GroupTag<0>{}, r0, o, oky);
                if ((sl == 1 || sl == 2) && pair == 1) xpair(GroupTag<1>{}, r0, o, oky);
                if (sl >= 2 && pair == 2) xpair(GroupTag<2>{}, r0, o, oky);
            }
            return;
        }
        if (i == 0) {
            if (wave < 5) xpair(GroupTag<0>{}, &Raw[tsrcA], &Vs[dA], okyA);
            else xpair(GroupTag<1>{}, &Raw[tsrcA], &Vs[dA], okyA);
        } else {
            if (wave < 2) xpair(GroupTag<1>{}, &Raw[tsrcB], &Vs[dB], okyB);
            else if (wave < 7) xpair(GroupTag<2>{}, &Raw[tsrcB], &Vs[dB], okyB);
        }
    };

    // ---- B fragments straight from memory: packed [m = 2 t + group][chunk][cout_pad / 32][piece][32 channels][16 k as 8 words] -- one
    // buffer load per (piece, 32-channel tile) reads 1 KiB contiguous; beyond the array (the prefetch of the chunk after the last) it returns 0
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.ubytes, 0x00020000);
    const int bvoff = li * 32 + lh * 16;
    const int wgroups = p.CoutPad >> 5, wg0 = (co0 >> 5) + wn;

    f32x16 acc[3][2];            // [position of the group][M tile]
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][mt][r] = 0.f;

    // A fragment base: GEMM row li of M tile mt is (row 4 mt + (li & 3), quad li >> 2)
    const int aBase = (li & 3) * G_ROWW + (li >> 2) * G_XQW + lh * 4;

    // ---- prologue: the raw pixels of chunk 0 first, the weights of chunk 0 behind them (their latencies overlap) ----------------------
    if (AFF) {          // this image's 2 x Cin coefficients -> LDS (complete before the first barrier: plain loads, the wave's own waits)
        const float* const sa = p.aff_a + (long long)n * p.Cin;
        const float* const sb = p.aff_b + (long long)n * p.Cin;
        for (int i = tid * 4; i < p.Cin; i += T::NT * 4) {
            *reinterpret_cast<f32x4*>(Aff + i) = *reinterpret_cast<const f32x4*>(sa + i);
            *reinterpret_cast<f32x4*>(Aff + p.Cin + i) = *reinterpret_cast<const f32x4*>(sb + i);
        }
    }
#pragma unroll
    for (int i = 0; i < NRAW; ++i) dma_raw(0, i);
    int kx = 0, kw = 0;

    // The main loop of position group G_ (a compile-time constant: positions are instruction immediates).  k-step t = ky * 3 + j of a
    // chunk: group G_ multiplies position q = 3 G_ + j of halo rows ky .. ky + 7 by slab m = 2 t + G_.  The two V buffers alternate per
    // chunk: the thread's bases into "this chunk's" and "the next chunk's" buffer are swapped at the end of a chunk (ONE loop body).
    auto run = [&](auto gtag) __attribute__((always_inline)) {
        constexpr int G_ = decltype(gtag)::value;
        constexpr int RING = HALF ? 9 : 3;      // register sets of the weight ring: k-step t's slab lives in set t % RING
        f16x8g bq[RING][2];          // [set][piece]
        const int tstride = 2 * nchunks * wgroups * 2048, cstride = wgroups * 2048;
        int sbase = (G_ * nchunks * wgroups + wg0) * 2048;          // slab (t = 0, group G_) of the chunk being LOADED
        auto load_b = [&](int t_) {
            const int soff = sbase + t_ * tstride;        // uniform
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) bq[t_ % RING][pc] = __builtin_bit_cast(f16x8g, __builtin_amdgcn_raw_buffer_load_b128(wrs, bvoff + pc * 1024, soff, 0));
        };
        f16x8g a[2][2];              // [piece][M tile]: ONE set -- the lo pieces of the next k-step are read as soon as this k-step's first
                                     // MFMA group (their only reader) is issued, the hi pieces after its last group
        auto read_a = [&](int pc, int vbase, int t_) {
            const float* vb = &Vs[vbase + (t_ / 3) * G_ROWW + (3 * G_ + t_ % 3) * G_QW + 8 * pc];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) a[pc][mt] = __builtin_bit_cast(f16x8g, *reinterpret_cast<const f32x4*>(vb + mt * 4 * G_ROWW));
        };
#pragma unroll
        for (int t = 0; t < RING; ++t) load_b(t);
        if (HALF) sbase += cstride;
        if (HALF) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");          // the raw pieces (older than the 18 weight loads) have landed
        asm volatile("" : "+v"(xword), "+v"(wword));
        kx = ccst_scale_exp(ccst_absmax_reduce(xword), F43_X_TARGET);
        kw = ccst_scale_exp(ccst_absmax_reduce(wword), F43_W_TARGET);
        xs = __uint_as_float((unsigned)(127 + kx) << 23);
#pragma unroll
        for (int i = 0; i < (HALF ? 1 : 2); ++i) xform(tdstA, tdstB, i);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (HALF) {
            // ---- four waves, ONE V buffer: the nine k-steps of a chunk (raw pieces of the next chunk requested at its first k-step, the
            // weights of k-step t of the next chunk behind k-step t), then -- every wave waits for its own pieces, barrier -- the transform
            // of the next chunk into the same buffer, barrier.  The CU's other workgroup has the MFMA pipe meanwhile.
            read_a(1, aBase, 0);
            read_a(0, aBase, 0);
            auto hchunk = [&](const int c, auto lasttag) __attribute__((always_inline)) {
                constexpr bool LAST = decltype(lasttag)::value != 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int j = t % 3;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][mt], bq[t][0], acc[j][mt], 0, 0, 0);   // a_lo b_hi
                    __builtin_amdgcn_sched_barrier(0);
                    if (t < 8) read_a(1, aBase, t + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][mt], bq[t][1], acc[j][mt], 0, 0, 0);   // a_hi b_lo
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][mt], bq[t][0], acc[j][mt], 0, 0, 0);   // a_hi b_hi
                    __builtin_amdgcn_sched_barrier(0);
                    if (t < 8) read_a(0, aBase, t + 1);
                    if (t == 0 && !LAST) {
#pragma unroll
                        for (int i = 0; i < NRAW; ++i) dma_raw(c + 1, i);
                    }
                    if (!LAST) load_b(t);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!LAST) {
                    // the raw pieces are older than the 18 weight loads of this chunk; every wave is past its last fragment read
                    asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    aff_c = c + 1;
                    xform(tdstA, tdstB, 0);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    read_a(1, aBase, 0);
                    read_a(0, aBase, 0);
                    sbase += cstride;
                }
            };
            for (int c = 0; c + 1 < nchunks; ++c) hchunk(c, GroupTag<0>{});
            hchunk(nchunks - 1, GroupTag<1>{});
            return;
        }
        if (nchunks > 1) {
#pragma unroll
            for (int i = 0; i < NRAW; ++i) dma_raw(1, i);
        }
        int aCur = aBase, aNxt = aBase + G_VW, dWn = tdstA + G_VW, d8n = tdstB + G_VW, dWc = tdstA, d8c = tdstB;
        read_a(1, aCur, 0);
        read_a(0, aCur, 0);

        // one chunk; LAST (compile time): nothing to prepare behind it -- no weight loads, no raw pieces, no transform
        auto chunk = [&](const int c, auto lasttag) __attribute__((always_inline)) {
            constexpr bool LAST = decltype(lasttag)::value != 0;
            aff_c = c + 1;              // (the transform items of this chunk's k-steps belong to the next chunk)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int j = t % 3;
                const int nbase = t == 8 ? aNxt : aCur, nt_ = (t + 1) % 9;          // the next k-step's V buffer (of the next chunk: complete since the barrier of k-step 6)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][mt], bq[t % RING][0], acc[j][mt], 0, 0, 0);   // a_lo b_hi
                __builtin_amdgcn_sched_barrier(0);
                read_a(1, nbase, nt_);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][mt], bq[t % RING][1], acc[j][mt], 0, 0, 0);   // a_hi b_lo
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][mt], bq[t % RING][0], acc[j][mt], 0, 0, 0);   // a_hi b_hi
                __builtin_amdgcn_sched_barrier(0);
                read_a(0, nbase, nt_);
                // staging.  The raw buffer holds chunk c + 1 (every wave waits for its own pieces, then the barrier, at k-step 2); it is
                // transformed into the other V buffer at k-steps 3, 4 (group 0) / 5, 6 (group 1), one item per thread and k-step, and after the barrier of k-step 6 refilled
                // with chunk c + 2 (k-steps 7, 8).  The weights of k-step t of the NEXT chunk replace this k-step's.
                if (t == 7 && !LAST) {
                    dma_raw(c + 2, 0);
                    dma_raw(c + 2, 1);
                }
                if (t == 8 && !LAST) {
                    dma_raw(c + 2, 2);
                    dma_raw(c + 2, 3);
                }
                if (t + RING < 9) load_b(t + RING);
                else if (!LAST) {
                    sbase += cstride;
                    load_b(t + RING - 9);
                    sbase -= cstride;
                }
                // (the two waves of a SIMD belong to different groups: staggered, one of them keeps the MFMA pipe busy while the other
                //  transforms; an item's pixels are requested one k-step ahead)
                if (G_ == 1 && t == 4 && !LAST) xload(0);
                if (t == 3 + 2 * G_ && !LAST) {
                    xcomp(dWn, d8n, 0);
                    xload(1);
                }
                if (t == 4 + 2 * G_ && !LAST) xcomp(dWn, d8n, 1);
                __builtin_amdgcn_sched_barrier(0);
                // k-step 2: the raw pieces (issued before the weights of k-step 8: the 8 loads issued since -- 6 after the prologue's --
                // may stay in flight) have landed, and every wave is done with the fragments of the previous chunk's V;  k-step 6: the
                // other V buffer is complete and the raw buffer is free
                if (t == 2) {
                    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    if (G_ == 0 && !LAST) xload(0);
                }
                if (t == 6) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            sbase += cstride;
            int tmp = aCur; aCur = aNxt; aNxt = tmp;
            tmp = dWc; dWc = dWn; dWn = tmp;
            tmp = d8c; d8c = d8n; d8n = tmp;
        };
        for (int c = 0; c + 1 < nchunks; ++c) chunk(c, GroupTag<0>{});
        chunk(nchunks - 1, GroupTag<1>{});
    };
    if (grp == 0) run(GroupTag<0>{});
    else run(GroupTag<1>{});
    // every wave must be past its last fragment read before the exchange below overwrites the operand images (nothing is in flight: the
    // last chunk fetches nothing)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- epilogue: the groups' halves of A^T on the scaled accumulators, exchange, scale back (one v_ldexp per OUTPUT: powers of two commute
    // with every rounding here), bias ---------------------------------------------------------
    //   group 0 (m0 m1 m2): a0 = m0 + m1 + m2, a1 = m1 - m2, a2 = m1 + m2;     group 1 (m3 m4 m5): b0 = m3 + m4, b1 = 2 (m3 - m4), b2 = 4 b0, b3 = 4 b1 + m5
    //   Y0 = a0 + b0, Y1 = a1 + b1 (finished by group 0: it receives b0, b1);   Y2 = a2 + b2, Y3 = a1 + b3 (group 1: receives a2, a1)
    // (in two phases over the accumulators, which stay where they are)
    const int ks = -(kx + kw);
    float* const xch = f43_lds;                    // [waves][2 mt][4 register quads][2 values][64 lanes][4 floats] = 128 KB
    // phase 1: what the partner wave (same channels, other group) needs of every accumulator element
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 s0, s1;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = 4 * rq + k;
                    const float m1 = acc[1][mt][r];
                    if (grp == 0) {
                        const float m2 = acc[2][mt][r];
                        s0[k] = m1 + m2;               // a2 -> group 1's Y2
                        s1[k] = m1 - m2;               // a1 -> group 1's Y3
                    } else {
                        const float m0 = acc[0][mt][r];
                        s0[k] = m0 + m1;               // b0 -> group 0's Y0
                        s1[k] = 2.f * (m0 - m1);       // b1 -> group 0's Y1
                    }
                }
                float* o = &xch[((((wave * 2 + mt) * 4 + rq) * 2) * 64 + lane) * 4];
                *reinterpret_cast<f32x4*>(o) = s0;
                *reinterpret_cast<f32x4*>(o + 256) = s1;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const bool relu = p.relu != 0;
    float amax = 0.f;
    const unsigned peeked = ymax_n != nullptr ? ccst_absmax_peek(ymax_n, blockIdx.x) : 0u;
    const bool interior = (oy0 + G_TH <= p.H) && (ox0 + G_TW <= p.W) && (co0 + G_BN <= p.Cout);
    const int other = wave ^ (HALF ? 2 : 4);
    // phase 2: fin[e][mt][r], e = 0, 1 = pixel 4 quad + 2 grp + e of row 4 mt + (r & 3), quad = 2 (r >> 2) + lh
    {
        const int co = co0 + wn * 32 + li;
        const bool cok = co < p.Cout;
        const float bias = (p.bias != nullptr && cok) ? p.bias[co] : 0.f;
        f32x16 fin[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const float* o = &xch[((((other * 2 + mt) * 4 + rq) * 2) * 64 + lane) * 4];
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(o), t1 = *reinterpret_cast<const f32x4*>(o + 256);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = 4 * rq + k;
                    const float m0 = acc[0][mt][r], m1 = acc[1][mt][r], m2 = acc[2][mt][r];
                    if (grp == 0) {
                        fin[0][mt][r] = __builtin_ldexpf(((m0 + m1) + m2) + t0[k], ks) + bias;                  // Y0 = a0 + b0
                        fin[1][mt][r] = __builtin_ldexpf((m1 - m2) + t1[k], ks) + bias;                         // Y1 = a1 + b1
                    } else {
                        fin[0][mt][r] = __builtin_ldexpf(4.f * (m0 + m1) + t0[k], ks) + bias;                   // Y2 = b2 + a2
                        fin[1][mt][r] = __builtin_ldexpf((8.f * (m0 - m1) + m2) + t1[k], ks) + bias;            // Y3 = b3 + a1
                    }
                }
            }
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
                        float v = fin[e][mt][r];
                        if (relu) v = fmaxf(v, 0.f);
                        const int dx = 8 * (r >> 2) + e;               // + 4 lh (lane_off) + 2 grp (tile)
                        if (interior) {
                            amax = fmaxf(amax, fabsf(v));
                            s1 += v;
                            cnt += 1.f;
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, lane_off * 4, (dy * p.ysH + dx * p.ysW) * 4, NT ? 2 : 0);
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
                // The per-(tile, position group) channel statistics the AdaIN step and stage 1 take instead of a pass over the features: (sum,
                // M2, count) with M2 = the sum of squares ABOUT THE SLAB'S OWN MEAN -- a second pass over the values still in registers -- so
                // that the consumer's variance (Chan's merge, fp64) has no E[x^2] - mean^2 cancellation however large |mean| / sigma is (raw
                // fp32 sums of x^2 lose the variance once mean^2 >> var)
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
                            float v = fin[e][mt][r];
                            if (relu) v = fmaxf(v, 0.f);
                            const float dv = v - mu;
                            if (interior || (cok && oy0 + dy < p.H && ox0 + 2 * grp + 8 * (r >> 2) + e + 4 * lh < p.W)) m2 += dv * dv;
                        }
                    }
                m2 += __shfl_xor(m2, 32, 64);
                // (... and the slab's largest |value|: a lane holds ONE channel, so its running maximum is the channel's -- what lets the AdaIN fold
                //  bound max |a x + b| per image without a pass over the tensor)
                const float cmax = fmaxf(amax, __shfl_xor(amax, 32, 64));
                if (lh == 0 && cok)
                    *reinterpret_cast<f32x4*>(p.stats + ((long long)((((n * p.tilesY + ty) * p.tilesX + tx) * 2 + grp)) * p.Cout + co) * 4) =
                        f32x4{s1, m2, cnt, cmax};
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
                        float v = fmaxf(fmaxf(fin[0][mt][2 * g], fin[1][mt][2 * g]), fmaxf(fin[0][mt][2 * g + 1], fin[1][mt][2 * g + 1]));
                        if (relu) v = fmaxf(v, 0.f);
                        amax = fmaxf(amax, fabsf(v));
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, lane_off * 4, (dyp * p.ysH + xpu * p.ysW) * 4, NT ? 2 : 0);
                    } else {
                        const int pyp = py0 + dyp, pxp = px0 + xpu + 2 * lh;
                        if (cok && pyp < Hp && pxp < Wp) {
                            const bool okx = (2 * pxp + 1 < p.W), oky = (2 * pyp + 1 < p.H);
                            float v = fin[0][mt][2 * g];
                            if (okx) v = fmaxf(v, fin[1][mt][2 * g]);
                            if (oky) v = fmaxf(v, fin[0][mt][2 * g + 1]);
                            if (okx && oky) v = fmaxf(v, fin[1][mt][2 * g + 1]);
                            if (relu) v = fmaxf(v, 0.f);
                            amax = fmaxf(amax, fabsf(v));
                            tile[(long long)dyp * p.ysH + (long long)xpu * p.ysW + lane_off] = v;
                        }
                    }
                }
            }
        }
    }
    if (ymax_n != nullptr) ccst_absmax_publish(ymax_n, amax, blockIdx.x, peeked);
}

// OIHW 3x3 -> [m = 2 t + group][Cin/16][cout_pad/32][piece][32 channels][8 words], t = ky * 3 + j, position q = 3 group + j: piece 0 = the
// chunk's 16 input channels of U_q[ky] = (G g[ky][.])_q as half(u * 2^kw), two per word; piece 1 = half(u * 2^kw - hi): the order in which a
// wave's B fragments are loaded (lane (channel, k half) -> 16 bytes, 1 KiB per load).  G g in double (the weights are packed once).
__global__ void pack_weight_f43_kernel(const float* __restrict__ w, unsigned* __restrict__ out, int cout, int cin, int cout_pad,
                                       const unsigned* __restrict__ wmax) {
    const int kw = ccst_scale_exp(ccst_absmax_read(wmax), F43_W_TARGET);
    const int nch = cin / 16;
    const long long total = 18LL * nch * cout_pad * 16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int word = (int)(i & 7);
        long long jj = i >> 3;
        const int row = (int)(jj & 31);
        jj >>= 5;
        const int piece = (int)(jj & 1);
        jj >>= 1;
        const int cg = (int)(jj % (cout_pad >> 5));
        jj /= (cout_pad >> 5);
        const int chunk = (int)(jj % nch), m = (int)(jj / nch);
        const int co = cg * 32 + row;
        const int t = m >> 1, ky = t / 3, q = 3 * (m & 1) + t % 3;
        const int k0 = chunk * 16 + 2 * word;
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
    CCST_REQUIRE(cout_pad >= cout && cout_pad % 64 == 0, "pack_f43: cout_pad must be a multiple of 64 >= cout");
    const long long total = 18LL * cin * cout_pad;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weight_f43_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, reinterpret_cast<unsigned*>(u), cout, cin,
                       cout_pad, w_absmax);
    return ccst_launch_status("pack_weight_f43");
}

// Rows of chan_sum_partials: one per (image, 8x32-pixel tile, position group), an image's rows contiguous.
extern "C" int ccst_conv3x3_f43_tiles(int N, int H, int W) { return N * ((H + 7) / 8) * ((W + G_TW - 1) / G_TW) * 2; }

// Workgroups the kernel launches for a layer (8 x 32 pixels x 128 channels each, one per CU; Cout <= 64: x 64 channels, two per CU).
extern "C" int ccst_conv3x3_f43_workgroups(int N, int H, int W, int Cout) {
    const int bn = Cout <= 64 ? 64 : 128;
    return N * ((H + 7) / 8) * ((W + G_TW - 1) / G_TW) * ((Cout + bn - 1) / bn);
}

template <bool POOL, bool ZP, bool HALF, bool NT, bool AFF = false>
static int launch_f43(F43Args& a, int N, int H, int W, int Cout, hipStream_t s) {
    typedef Tile<HALF> T;
    a.tilesN = (Cout + T::BN - 1) / T::BN;
    a.tilesY = (H + T::TH - 1) / T::TH;
    a.tilesX = (W + G_TW - 1) / G_TW;
    const long long grid = (long long)N * a.tilesY * a.tilesX * a.tilesN;
    CCST_REQUIRE(grid > 0 && grid <= 0x7fffffffLL, "conv3x3_f43: bad grid");
    void (*kern)(const F43Args) = conv3x3_f43_kernel<POOL, ZP, HALF, NT, AFF>;
    if (AFF) CCST_REQUIRE(T::OPERAND_BYTES + 8LL * a.Cin <= T::LDS_BYTES, "conv3x3_f43: too many input channels for the fused input affine (<= %d)",
                          (T::LDS_BYTES - T::OPERAND_BYTES) / 8);
    // (the opt-in above the 64 KB default is per device and idempotent: set for the current device on every launch)
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
    if (e1 != hipSuccess) {
        ccst_set_error("conv3x3_f43: cannot reserve %d bytes of LDS: %s", T::LDS_BYTES, hipGetErrorString(e1));
        return (int)e1;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(T::NT), T::LDS_BYTES, s, a);
    return ccst_launch_status("conv3x3_f43");
}

// Contract in include/ccst_hip.h; the tile is 8 rows x 32 pixels x 128 channels, or for Cout <= 64 (no statistics)
// 8 rows x 32 pixels x 64 channels, four waves, two workgroups per CU (ccst_conv3x3_f43_workgroups).
extern "C" int ccst_conv3x3_f43_f32(const float* x, const uint32_t* x_absmax, const float* u, const uint32_t* w_absmax, const float* bias,
                                    float* y, uint32_t* y_absmax, int N, int H, int W, int Cin, int Cout, int cout_pad, uint32_t flags,
                                    float* chan_sum_partials, const float* in_scale, const float* in_shift, void* stream) {
    CCST_REQUIRE(x && u && y && x_absmax && w_absmax, "conv3x3_f43: null pointer (the |max| words of x and w are required)");
    CCST_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv3x3_f43: in_scale and in_shift come together ([N][Cin] each)");
    const bool aff = in_scale != nullptr;
    CCST_REQUIRE(!aff || ((flags & CCST_CONV_REFLECT) && !(flags & (CCST_CONV_POOL2 | CCST_CONV_UPS2))),
                 "conv3x3_f43: the fused input affine goes with a plain reflection-padded conv (no zero padding, no pool, no upsample)");
    CCST_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0 && Cout > 0, "conv3x3_f43: bad shape");
    const bool half = Cout <= 64;
    CCST_REQUIRE(cout_pad >= Cout && cout_pad % (half ? 64 : 128) == 0, "conv3x3_f43: cout_pad must be a multiple of 128 (Cout <= 64: of 64) >= cout");
    const bool pool = (flags & CCST_CONV_POOL2) != 0, ups = (flags & CCST_CONV_UPS2) != 0;
    CCST_REQUIRE(!(chan_sum_partials && pool), "conv3x3_f43: channel sums are of the un-pooled output");
    if (ups) CCST_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv3x3_f43: upsampled extent must be even");
    if (flags & CCST_CONV_REFLECT) CCST_REQUIRE(H >= 2 && W >= 2, "conv3x3_f43: reflection needs extent >= 2");
    F43Args a;
    a.x = x; a.u = u; a.bias = bias; a.y = y; a.xmax = x_absmax; a.wmax = w_absmax; a.ymax = y_absmax; a.stats = chan_sum_partials;
    a.aff_a = in_scale; a.aff_b = in_shift;
    a.N = N; a.H = H; a.W = W; a.Hs = ups ? H / 2 : H; a.Ws = ups ? W / 2 : W; a.Cin = Cin; a.Cout = Cout; a.CoutPad = cout_pad;
    a.reflect = (flags & CCST_CONV_REFLECT) ? 1 : 0; a.ups = ups ? 1 : 0; a.relu = (flags & CCST_CONV_RELU) ? 1 : 0;
    CCST_REQUIRE((long long)N * a.Hs * a.Ws * Cin < 0x7fffffffLL, "conv3x3_f43: input must have < 2^31 elements");
    CCST_REQUIRE((long long)a.Hs * a.Ws * Cin < (1LL << 30), "conv3x3_f43: one image must have < 2^30 elements (32-bit byte offsets per image)");
    const int oh = pool ? (H + 1) / 2 : H, ow = pool ? (W + 1) / 2 : W;
    CCST_REQUIRE((long long)N * oh * ow * Cout < 0x7fffffffLL, "conv3x3_f43: output must have < 2^31 elements");
    a.ysW = Cout; a.ysH = ow * Cout; a.ysN = (long long)oh * ow * Cout;
    CCST_REQUIRE(18LL * Cin * cout_pad * 4 < 0x7fffffffLL, "conv3x3_f43: packed weights must be < 2^31 bytes");
    a.ubytes = (int)(18LL * Cin * cout_pad * 4);
    const bool zp = !a.reflect;
    hipStream_t st = (hipStream_t)stream;
    // outputs the Infinity Cache (256 MB) cannot hold leave with non-temporal stores (the bench's 512^2 x 64-channel map: 403 MB; the stem's
    // likewise): cached, they only push the next launch's weights and halos out (-7 % on the kernel that reads them, +1 % on the step);
    // smaller ones are read back from the caches, and non-temporal stores cost the step 2 % (measured at 100 / 200 / 400 MB thresholds)
    const long long nt_bytes = 256LL << 20;
    const bool nt = (long long)N * oh * ow * Cout * 4 >= nt_bytes;
    if (aff) {          // (never pooled; its output is small enough for the caches wherever the path uses it: ordinary stores)
        return half ? launch_f43<false, false, true, false, true>(a, N, H, W, Cout, st) : launch_f43<false, false, false, false, true>(a, N, H, W, Cout, st);
    }
#define F43_GO2(P_, Z_, H_) (nt ? launch_f43<P_, Z_, H_, true>(a, N, H, W, Cout, st) : launch_f43<P_, Z_, H_, false>(a, N, H, W, Cout, st))
#define F43_GO(P_, Z_) (half ? F43_GO2(P_, Z_, true) : F43_GO2(P_, Z_, false))
    return pool ? (zp ? F43_GO(true, true) : F43_GO(true, false)) : (zp ? F43_GO(false, true) : F43_GO(false, false));
#undef F43_GO
#undef F43_GO2
}
