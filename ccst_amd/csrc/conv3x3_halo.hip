// 3x3 stride-1 "same" convolution (reflection or zero padding) on the fp32-input MFMA with the input HALO
// staged in LDS -- the AdaIN encoder/decoder layers (net.py:6-69).
//
// conv_igemm.hip gathers an [M x 16] A tile per (tap, 16-channel chunk): every input element is loaded from
// L2 and written to LDS nine times.  Here a workgroup owns a spatial tile of TH x 16 output pixels of one
// image; per 16-channel chunk the (TH+2) x 18 halo is loaded ONCE (reflection / zero padding and the
// optional nearest-x2 upsample applied in that loader) and the nine taps read their A fragments from the
// same LDS image at shifted addresses.  A-side global loads, address arithmetic and ds_writes drop 9x
// (measured on the gather kernel: staging costs ~12 % of the time); the per-step work is the weight tile only.
//
//   GEMM roles, MFMA operand layout, packed weights [tap][ci/4][co][4], accumulators, bias/ReLU/fused
//   2x2 ceil max-pool epilogue: as conv_igemm.hip.  Row r of the M tile maps to the pixel
//   (py, px) = (2*(r>>5) + ((r&3)>>1), 2*((r&31)>>2) + (r&1)) so a pooling window sits in one lane's reg&3.
//   LDS: halo [2][(TH+2)*18][16+4] floats (80-B pixel pitch: conflict-free ds_read_b128) + W ring [3][16][BN].
//   Measured: 133.9 TF (128x128 tile, 256->256 @128x128, B=6) vs 129 TF for the gather kernel; staging the
//   weight tile by LDS-DMA instead of through registers measures the same (133.3 TF); fetching the weight
//   fragments straight into registers (LDS = halo only, one barrier per 9 taps) measures 106 TF -- the four
//   waves' duplicated fragment loads thrash the 32-KiB L1 (both variants kept under tools/micro/).  De-phasing the
//   ~3 co-resident workgroups (start offsets of 0.3/0.6 step, or static s_setprio 0/1/2) changes nothing
//   (133.4-133.8 TF): the residual MFMA idle time is not a lockstep effect.
//   What did matter is WHERE in a k-step the staging sits (133.5 -> 143 TF, table in DESIGN.md 3.0): nothing but
//   MFMAs right behind the barrier, loads / LDS writes in front of the last groups of the step.  A doubled main
//   loop shows the loop itself at 150 TF; the rest is ~28 us per launch of tile prologue / epilogue / dispatch.  A
//   persistent-workgroup version that requests the next tile's prologue ahead of the epilogue stores is slower
//   (137 TF, tools/micro/conv3x3_halo_persistent_variant.hip.txt): 64 stores overflow the 6-bit vmcnt, so the
//   prologue waits for the whole store drain, which the hardware dispatcher overlaps for free.
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

struct HaloArgs {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    int N, H, W, Hs, Ws, Cin, Cout, CoutPad;     // H,W: conv (= output) extent; Hs,Ws: source extent (H/2,W/2 if ups)
    int reflect, ups, relu;
    long long ysN;
    int ysH, ysW;                                 // output strides (of the pooled tensor when POOL)
    int tilesX, tilesY, tilesN;
    const unsigned* xmax;   // SPLIT: |max| words of x and of the OIHW weight (CCST_ABSMAX_WORDS each): the operands' power-of-two scales
    const unsigned* wmax;   //        are derived from them in the kernel, the accumulators scaled back before the epilogue
    unsigned* ymax;         // SPLIT: nullptr, or zeroed |max| words receiving max |y| of what this launch stores
    int wpn;                // words per image of xmax / ymax: CCST_ABSMAX_WORDS (the AdaIN entry: every image its own words, [N][64]) or 0
                            // (the ResNet train form: one set per tensor -- BatchNorm couples the samples of a batch anyway)
    float* stats;        // TRAIN: per-(spatial tile, wave row) (sum, sum^2) partials of the output, or nullptr; SPLIT: (sum, M2 about the slab's own mean, count, 0)
    int flip, accum;     // TRAIN: taps read in reverse order (backward-data); y += conv
    // TRAIN + SPLIT, backward-data whose result is the output gradient of a BatchNorm + ReLU that only this conv reads (bn1 -> conv2):
    // the epilogue recomputes that ReLU's mask from the BatchNorm's input, stores the MASKED gradient and leaves the BatchNorm
    // backward's per-channel partial sums (sum g, sum g xhat) per (tile, wave row) -- as conv1x1_stream_kernel's MASKED = 2 form
    const float *bn_x, *bn_mean, *bn_invstd, *bn_gamma, *bn_beta;
    float* bn_part;
};

// Positions (in groups of 4 MFMAs, 8 groups per k-step) of the staging inside a step; measured sweep in DESIGN.md 3.0.
#define LOAD_P 7       // global loads of the weights of step t+3 / a halo unit of the next chunk
#define STORE_P 6      // LDS writes of what was fetched during the previous step
#define READ1_P 3      // second 8-channel fragment pair of this step
#define PRE_P 5        // first fragment pair of the NEXT step (visible since the previous barrier: 3-deep weight ring)
constexpr int CKH = 16, PITCH = CKH + 4, HW_ = 18;
constexpr int BPITCH = 20;      // SPLIT: words per output-channel row of the weight image (16 channels hi | lo as half = 16 words, + 4 of pad)
typedef ccst_u32x2 u32x2h;
typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));      // the 16-bit pieces are IEEE half: 11 significant bits each
typedef _Float16 f16x2h __attribute__((ext_vector_type(2)));
typedef float f32x2h __attribute__((ext_vector_type(2)));

// four fp32 values, scaled by the tensor's power of two s (max |x| s < 2^14: nothing overflows half, and a tensor of small values is
// lifted out of half's subnormals) -> two half pieces each (hi = half(x s), lo = half(x s - hi): 22 significant bits for every element
// within 2^-17 of the tensor's largest, an absolute error of 2^-38 of that largest below), two per word.  Written on 2-vectors so that
// hipcc emits v_pk_mul_f32 / v_cvt_pk_f16_f32 / v_pk_fma_f32: 12 vector instructions per four values (the scalar form took 21).
__device__ __forceinline__ void split4h(f32x4 v, float s, u32x2h& hi, u32x2h& lo) {
    ccst_u32x2 h2, l2;          // (round 5: 8 vector instructions per four values, common.h)
    ccst_split4_half(v, s, h2, l2);
    hi = h2;
    lo = l2;
}

__device__ __forceinline__ int reflect_h(int i, int n) {
    i = (i < 0) ? -i : i;
    i = (i >= n) ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

// TRAIN = true: the ResNet-trunk form (zero padding, no bias/ReLU/pool) with what the train step needs from it --
// BatchNorm statistics of the output from the epilogue (as ccst_conv2d_igemm_stats_f32), the taps in reverse
// order (backward-data of a stride-1 3x3 conv = the same conv with flipped taps and the transposed weight), and
// y += conv (CCST_CONV_ACCUM).  A separate instantiation so the AdaIN kernels carry none of it.
// SPLIT = true (the AdaIN path's layers with Cin != Cout): the products on the 16-bit MFMA, each fp32 product as three products of
// IEEE-half pieces (x = hi + lo, hi = half(x), lo = half(x - hi): 22 significant bits; a_lo b_hi + a_hi b_lo + a_hi b_hi, fp32
// accumulate): v_mfma_f32_32x32x16_f16 does 16x the multiply-adds per cycle of v_mfma_f32_32x32x2_f32, so three products run at 5.3x the
// fp32 MFMA's rate (tools/micro/bf16x3.hip measures the bf16 twin), at 1e-6 of max |y| per layer -- bf16 pieces (16 bits) measured
// 5e-6 and failed the wide-dynamic-range path test at 1.8e-3.  The halo is split where it passes from registers to LDS: a pixel is
// [16 channels hi | 16 channels lo] as half -- the same 64 bytes (+ pad) as its fp32 form; the weights come pre-split (and scaled by a
// power of two into half's normal range, the accumulators scaled back) as rows [output channel][16 k hi | lo]; a k-step (tap, 16
// channels) is then 12 MFMAs of K = 16 per wave instead of 32 of K = 2.  Range: the activations are scaled too, by the power of two
// that puts the tensor's largest |value| (p.xmax, left by the producing kernel's epilogue) below 2^14 -- any finite fp32 input is safe.
template <int WM, int WN, int NT, bool POOL, bool TRAIN = false, bool SPLIT = false>
__global__ __launch_bounds__(256, (SPLIT && NT == 2) ? 2 : 3) void conv3x3_halo_kernel(const HaloArgs p) {
    static_assert(!(TRAIN && POOL), "the train form has no pooled epilogue");
    constexpr int MT = 2;
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    constexpr int TH = BM / 16, HH = TH + 2;
    constexpr int HPIX = HH * HW_;                         // halo pixels
    // row stride of the halo image in LDS, in words: SPLIT pads the 18 pixels x 20 words = 360 to 416 so that the sixteen lanes of a
    // ds_read_b128 pass (pixels (py, px), (py, px + 1), (py + 1, px), ..., px + 2 j) fall on 16 different 16-byte bank groups of the 64
    // banks: in 16-byte units a pixel is 5, so the lanes sit at 5 (a + 2 j) + b * row, and the row must be 8 modulo 16 (104 units) --
    // with 360 words (90 units) the row step and the two-pixel step are the same distance modulo 16 and every read is two-way
    // conflicted (PMC: 41 % of the LDS cycles of the first SPLIT version).  (Padding whole pixels, 18 -> 24, costs 480 words a row and
    // the 128x64 tile its third workgroup per CU.)
    constexpr int HROW = SPLIT ? 416 : HW_ * PITCH;
    constexpr int HUNITS = HPIX * (CKH / 4);               // float4 units per chunk
    constexpr int HR = (HUNITS + 255) / 256;               // halo units per thread per chunk
    constexpr int BUNITS = (CKH / 4) * BN;
    constexpr int BR = (BUNITS + 255) / 256;
    static_assert(HR <= 9, "halo load rounds must fit the 9 tap steps of a chunk");

    __shared__ __attribute__((aligned(16))) float Hs_[2][HH * HROW];
    __shared__ __attribute__((aligned(16))) float Bs[3][SPLIT ? BN * BPITCH : CKH * BN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    int bid = ccst_xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p.tilesN;
    bid /= p.tilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int co0 = tn * BN;
    const int oy0 = ty * TH, ox0 = tx * 16;

    // ---- halo load units of this thread (pixel coordinates are chunk-invariant) -------------------
    unsigned hoff[HR];
    bool hok[HR];
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        const int u = min(tid + 256 * i, HUNITS - 1);
        const int pix = u >> 2, part = u & 3;
        const int hy = pix / HW_, hx = pix - hy * HW_;
        int gy = oy0 + hy - 1, gx = ox0 + hx - 1;
        bool ok = true;
        if (p.reflect) {
            gy = reflect_h(gy, p.H);
            gx = reflect_h(gx, p.W);
        } else {
            ok = (gy >= 0) & (gy < p.H) & (gx >= 0) & (gx < p.W);
            gy = min(max(gy, 0), p.H - 1);
            gx = min(max(gx, 0), p.W - 1);
        }
        gy >>= p.ups;
        gx >>= p.ups;
        hok[i] = ok;
        hoff[i] = (unsigned)(((n * p.Hs + gy) * p.Ws + gx) * p.Cin + part * 4);
    }
    unsigned boff[BR];
#pragma unroll
    for (int b = 0; b < BR; ++b) {
        const int u = min(tid + 256 * b, BUNITS - 1);
        const int g = u / BN, col = u - g * BN;
        boff[b] = SPLIT ? (unsigned)((co0 + (u >> 2)) * 16 + (u & 3) * 4)          // pre-split rows [output channel][16 k hi | lo]: 16 words
                     : (unsigned)((g * p.CoutPad + co0 + col) * 4);
    }

    // SPLIT: operand scales from the tensors' |max| words (wave-uniform), 2^kx for the activations, 2^kw was applied when the weights were packed
    // (the words are LOADED here and reduced behind the prologue's first halo load: written by the previous kernel's atomics, they are an
    //  L2 miss, and reduced here that miss would stand in front of the prologue's own loads)
    int kx = 0, kw = 0;
    unsigned xword = 0u, wword = 0u;
    if (SPLIT) {
        xword = ccst_absmax_load(p.xmax + n * p.wpn);
        wword = ccst_absmax_load(p.wmax);
    }
    float xs = 1.f;

    f32x16 acc[MT][NT];
    float biasv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = co0 + wn * (32 * NT) + nt * 32 + li;
        biasv[nt] = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
        const float b = SPLIT ? 0.f : biasv[nt];      // SPLIT adds the bias after the accumulators are scaled back
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = b;
    }

    // A fragment base per M tile: pixel (py,px) of row li of tile Tt, channel slot lh
    int aBase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int Tt = wm * MT + mt;
        const int py = 2 * Tt + ((li & 3) >> 1), px = 2 * (li >> 2) + (li & 1);
        aBase[mt] = py * HROW + px * PITCH + lh * 4;
    }
    const float* bRd0 = &Bs[0][(lh * BN + wn * (32 * NT) + li) * 4];

    const int nchunks = p.Cin / CKH;
    f32x4 rh, rb[BR];
    f32x4 rh2[2], rb2[2][BR];      // SPLIT: two register sets, every global load two k-steps ahead of its LDS store

    auto load_b_to = [&](f32x4 (&r)[BR], int c_, int tap) {
        const int tw = (TRAIN && p.flip) ? 8 - tap : tap;
        const float* wc = SPLIT ? p.w + ((long long)tw * (p.Cin / CKH) + c_) * p.CoutPad * 16
                             : p.w + ((long long)tw * (p.Cin / 4) + c_ * (CKH / 4)) * p.CoutPad * 4;   // uniform
#pragma unroll
        for (int b = 0; b < BR; ++b) r[b] = *reinterpret_cast<const f32x4*>(wc + boff[b]);
    };
    auto store_b_from = [&](const f32x4 (&r)[BR], int buf) {
#pragma unroll
        for (int b = 0; b < BR; ++b) {
            const int u = tid + 256 * b;
            if (SPLIT) {           // unit u = (output channel u >> 2, 16-byte part u & 3) of a pre-split row
                if (BUNITS % 256 == 0 || u < BUNITS) *reinterpret_cast<f32x4*>(&Bs[buf][(u >> 2) * BPITCH + (u & 3) * 4]) = r[b];
                continue;
            }
            if (BUNITS % 256 == 0 || u < BUNITS) *reinterpret_cast<f32x4*>(&Bs[buf][u * 4]) = r[b];
        }
    };
    auto load_b = [&](int c_, int tap) { load_b_to(rb, c_, tap); };
    auto store_b = [&](int buf) { store_b_from(rb, buf); };
    // halo unit i of chunk c_: global -> register, register -> LDS
    auto load_h = [&](int c_, int i, unsigned off) { rh = *reinterpret_cast<const f32x4*>(p.x + off + c_ * CKH); (void)i; };
    auto store_h_from = [&](const f32x4& rsrc_, int buf, int i, bool ok) {
        const int u = tid + 256 * i;
        if (u < HUNITS) {
            f32x4 v = rsrc_;
            if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            const int pix_ = u >> 2, slot_ = SPLIT ? (pix_ / HW_) * HROW + (pix_ % HW_) * PITCH : pix_ * PITCH;
            if (SPLIT) {
                u32x2h hi, lo;
                split4h(v, xs, hi, lo);
                float* px_ = &Hs_[buf][slot_ + (u & 3) * 2];
                *reinterpret_cast<u32x2h*>(px_) = hi;
                *reinterpret_cast<u32x2h*>(px_ + 8) = lo;
            } else
            *reinterpret_cast<f32x4*>(&Hs_[buf][slot_ + (u & 3) * 4]) = v;
        }
    };
    auto store_h = [&](int buf, int i, bool ok) { store_h_from(rh, buf, i, ok); };
    f32x4 af[2][MT], bf[2][NT];
    auto read_frags = [&](int hbuf, int bbuf, int tapoff, int q) {
        const float* hb = &Hs_[hbuf][tapoff];
        const float* bRd = bRd0 + bbuf * (CKH * BN);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[q][mt] = *reinterpret_cast<const f32x4*>(hb + aBase[mt] + q * 8);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[q][nt] = *reinterpret_cast<const f32x4*>(bRd + (2 * q * BN + nt * 32) * 4);
    };
    auto mfma_frags = [&](int q, int s0, int s1) {
#pragma unroll
        for (int s = s0; s < s1; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][mt][s], bf[q][nt][s], acc[mt][nt], 0, 0, 0);
    };
    static_assert(CKH == 16, "the step body below is written for two 8-channel halves");
    // SPLIT fragments of one step: [piece][tile]; a lane's 8 consecutive channels (lh * 8 ..) = words lh * 4 .. of the piece
    struct FragsSplit {
        f16x8h a[2][MT], b[2][NT];
    };
    const float* bRdB0 = &Bs[0][(wn * (32 * NT) + li) * BPITCH + lh * 4];
    auto read_frags_split = [&](FragsSplit& f, int hbuf, int bbuf, int tapoff) {
        const float* hb = &Hs_[hbuf][tapoff];
        const float* bRd = bRdB0 + bbuf * (BN * BPITCH);
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) f.a[pc][mt] = __builtin_bit_cast(f16x8h, *reinterpret_cast<const f32x4*>(hb + aBase[mt] + 8 * pc));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) f.b[pc][nt] = __builtin_bit_cast(f16x8h, *reinterpret_cast<const f32x4*>(bRd + nt * 32 * BPITCH + 8 * pc));
        }
    };
    auto mfma_split = [&](const FragsSplit& f, int pa, int pb) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[pa][mt], f.b[pb][nt], acc[mt][nt], 0, 0, 0);
    };

    // ---- prologue: halo of chunk 0, weights of steps 0 and 1 in LDS, weights of step 2 in flight -----------------
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        load_h(0, i, hoff[i]);
        if (SPLIT && i == 0) {
            asm volatile("" : "+v"(xword), "+v"(wword));      // (opaque: keeps the reduction, and its wait, behind the load just issued)
            kx = ccst_scale_exp(ccst_absmax_reduce(xword), CCST_SPLIT_X_TARGET);
            kw = ccst_scale_exp(ccst_absmax_reduce(wword), CCST_SPLIT_W_TARGET);
            xs = __uint_as_float((unsigned)(127 + kx) << 23);
        }
        store_h(0, i, hok[i]);
    }
    // 3-deep weight ring: step t computes from Bs[t%3]; the registers fetched during step t-1 (weights of step
    // t+2) are stored at group STORE_P, the fetch for step t+3 is issued at LOAD_P, and the first fragment pair of
    // step t+1 -- already visible since the previous barrier -- is read before this step's barrier, so that a step
    // boundary is barrier -> MFMA with no LDS latency behind it.
    load_b(0, 0);
    store_b(0);
    load_b(0, 1);
    store_b(1);
    if (!SPLIT) load_b(0, 2);
    if (SPLIT) {
        // SPLIT: k-step t = 9 c + tap stores the weights of step t + 2 from the register set t & 1 (fetched during step t - 2) and
        // re-issues that set's loads for step t + 4; halo unit i is fetched at tap i and stored at tap i + 2.  With the fetch one step
        // ahead of its store (the fp32 form) the waits on the loads were 25 % of this kernel: an L2 hit takes about as long as one
        // k-step of 12 MFMAs x 2 waves (ablation: no weight loads -15 %, no halo loads -12 %, no barriers / LDS reads: nothing).
        static_assert(!SPLIT || HR <= 6, "halo unit i is stored at tap i + 2 <= 7");
        load_b_to(rb2[0], 0, 2);
        load_b_to(rb2[1], 0, 3);
        __syncthreads();
        FragsSplit cur, nxt;
        read_frags_split(cur, 0, 0, 0);
        // one chunk (9 taps); P = parity of 9 c, so that the register set of a step is a compile-time index
        auto chunk = [&](const int c, const int P) __attribute__((always_inline)) {
            const int cn = min(c + 1, nchunks - 1), cnn = min(c + 2, nchunks - 1);
            (void)cnn;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int q = (tap + P) & 1;
                mfma_split(cur, 1, 0);                                             // a_lo b_hi
                __builtin_amdgcn_sched_barrier(0);
                {   // first fragments of the NEXT step (visible since the previous barrier: 3-deep weight ring, halo of this chunk)
                    const int tp = (tap + 1) % 9;
                    read_frags_split(nxt, tap == 8 ? (c + 1) & 1 : c & 1, tp % 3, (tp / 3) * HROW + (tp % 3) * PITCH);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_split(cur, 0, 1);                                             // a_hi b_lo
                __builtin_amdgcn_sched_barrier(0);
                store_b_from(rb2[q], (tap + 2) % 3);
                if (tap >= 2 && tap <= HR + 1) store_h_from(rh2[tap & 1], (c + 1) & 1, tap - 2, hok[tap - 2]);
                if (tap + 4 < 9) load_b_to(rb2[q], c, tap + 4);
                else load_b_to(rb2[q], cn, tap + 4 - 9);
                if (tap < HR) rh2[tap & 1] = *reinterpret_cast<const f32x4*>(p.x + hoff[tap] + cn * CKH);
                __builtin_amdgcn_sched_barrier(0);
                mfma_split(cur, 0, 0);                                             // a_hi b_hi
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                cur = nxt;
            }
        };
        {   // (an if / else on the parity inside one loop made the compiler copy the accumulators at the join: 256 VGPRs + spills)
            int c = 0;
            for (; c + 1 < nchunks; c += 2) {
                chunk(c, 0);
                chunk(c + 1, 1);
            }
            if (c < nchunks) chunk(c, 0);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = __builtin_ldexpf(acc[mt][nt][r], -(kx + kw)) + biasv[nt];   // (exact scaling)
    } else {
    __syncthreads();
    read_frags(0, 0, 0, 0);
    for (int c = 0; c < nchunks; ++c) {
        const int cn = min(c + 1, nchunks - 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int tapoff = (tap / 3) * HROW + (tap % 3) * PITCH;
#pragma unroll
            for (int pos = 0; pos <= 8; ++pos) {
                if (pos == READ1_P) {
                    __builtin_amdgcn_sched_barrier(0);
                    read_frags(c & 1, tap % 3, tapoff, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (pos == STORE_P) {
                    __builtin_amdgcn_sched_barrier(0);
                    store_b((tap + 2) % 3);
                    if (tap >= 1 && tap <= HR) store_h((c + 1) & 1, tap - 1, hok[tap - 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (pos == LOAD_P) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap + 3 < 9) load_b(c, tap + 3);
                    else load_b(cn, tap + 3 - 9);
                    if (tap < HR) load_h(cn, tap, hoff[tap]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (pos == PRE_P) {
                    __builtin_amdgcn_sched_barrier(0);
                    const int tp = (tap + 1) % 9;
                    read_frags(tap == 8 ? (c + 1) & 1 : c & 1, tp % 3, (tp / 3) * HROW + (tp % 3) * PITCH, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (pos < 8) mfma_frags(pos >> 2, pos & 3, (pos & 3) + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    }
    }

    // ---- epilogue ----------------------------------------------------------------------------------------
    const bool relu = p.relu != 0;
    // Interior tiles (the common case) store through a buffer resource on the tile origin: the per-lane byte
    // offset is ONE VGPR for the whole epilogue and the row's (dy, dx) -- which depend only on mt and the register
    // index -- go into the scalar offset, so a store is `buffer_store v, v_off, s[rsrc], s_row offen offset:nt*128`
    // with no vector address arithmetic and no predicate.  Edge tiles keep the predicated pointer path.
    const int cw = wn * (32 * NT);                                            // uniform (wn is)
    float amax = 0.f;                                                         // SPLIT: largest |value| this lane stores (p.ymax)
    unsigned* const ymax_n = (SPLIT && p.ymax != nullptr) ? p.ymax + n * p.wpn : nullptr;
    const unsigned peeked = ymax_n != nullptr ? ccst_absmax_peek(ymax_n, blockIdx.x) : 0u;     // (compared after the stores)
    if (!POOL) {
        float* const tile = p.y + (long long)n * p.ysN + (long long)oy0 * p.ysH + (long long)ox0 * p.ysW + co0 + cw;
        const unsigned lane_off = (unsigned)(2 * lh * p.ysW + li);
        const bool interior = (oy0 + TH <= p.H) && (ox0 + 16 <= p.W) && (co0 + BN <= p.Cout);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 0x7fffffff, 0x00020000);
        if ((TRAIN || SPLIT) && p.stats != nullptr) {
            // this wave's 64 pixels x 32*NT channels -> per-channel statistics of the OUTPUT (after bias / ReLU); pixels outside the image
            // excluded.  TRAIN: raw (sum, sum^2) for the next BatchNorm.  SPLIT: (sum, M2, count, 0) with M2 about the slab's own mean (a
            // second pass over the registers): what the AdaIN step and stage 1 take instead of a pass over the features, free of the
            // E[x^2] - mean^2 cancellation.
            const int slab = ((n * p.tilesY + ty) * p.tilesX + tx) * WM + wm;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = co0 + cw + nt * 32 + li;
                if (!SPLIT || TRAIN) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int dy = 2 * (wm * MT + mt) + ((r & 3) >> 1), dx = 4 * (r >> 2) + (r & 1) + 2 * lh;
                            float v = (interior || (oy0 + dy < p.H && ox0 + dx < p.W)) ? acc[mt][nt][r] : 0.f;
                            if (relu) v = fmaxf(v, 0.f);
                            s1 += v;
                            s2 += v * v;
                        }
                    s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                    if (lh == 0 && co < p.Cout) {
                        float* o = p.stats + ((long long)slab * p.Cout + co) * 2;
                        o[0] = s1;
                        o[1] = s2;
                    }
                } else {
                    float s1 = 0.f, cnt = 0.f;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int dy = 2 * (wm * MT + mt) + ((r & 3) >> 1), dx = 4 * (r >> 2) + (r & 1) + 2 * lh;
                            const bool inside = interior || (oy0 + dy < p.H && ox0 + dx < p.W);
                            float v = acc[mt][nt][r];
                            if (relu) v = fmaxf(v, 0.f);
                            s1 += inside ? v : 0.f;
                            cnt += inside ? 1.f : 0.f;
                        }
                    s1 += __shfl_xor(s1, 32, 64);
                    cnt += __shfl_xor(cnt, 32, 64);
                    const float mu = s1 / fmaxf(cnt, 1.f);
                    float m2 = 0.f;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int dy = 2 * (wm * MT + mt) + ((r & 3) >> 1), dx = 4 * (r >> 2) + (r & 1) + 2 * lh;
                            float v = acc[mt][nt][r];
                            if (relu) v = fmaxf(v, 0.f);
                            const float dv = v - mu;
                            if (interior || (oy0 + dy < p.H && ox0 + dx < p.W)) m2 += dv * dv;
                        }
                    m2 += __shfl_xor(m2, 32, 64);
                    if (lh == 0 && co < p.Cout) *reinterpret_cast<f32x4*>(p.stats + ((long long)slab * p.Cout + co) * 4) = f32x4{s1, m2, cnt, 0.f};
                }
            }
        }
        if (TRAIN && SPLIT && p.bn_x != nullptr) {
            const float* const xtile = p.bn_x + (long long)n * p.ysN + (long long)oy0 * p.ysH + (long long)ox0 * p.ysW + co0 + cw;
            const int slab = ((n * p.tilesY + ty) * p.tilesX + tx) * WM + wm;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = co0 + cw + nt * 32 + li;
                const bool cok = co < p.Cout;
                const float mu = cok ? p.bn_mean[co] : 0.f, is = cok ? p.bn_invstd[co] : 0.f;
                const float ga = cok ? p.bn_gamma[co] : 0.f, be = cok ? p.bn_beta[co] : 0.f;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float xv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dy = 2 * (wm * MT + mt) + ((r & 3) >> 1), dx = 4 * (r >> 2) + (r & 1) + 2 * lh;
                        const bool inside = cok && (interior || (oy0 + dy < p.H && ox0 + dx < p.W));
                        xv[r] = inside ? xtile[(long long)dy * p.ysH + (long long)dx * p.ysW + nt * 32 + li] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dy = 2 * (wm * MT + mt) + ((r & 3) >> 1), dx = 4 * (r >> 2) + (r & 1) + 2 * lh;
                        const bool inside = cok && (interior || (oy0 + dy < p.H && ox0 + dx < p.W));
                        const float xh = (xv[r] - mu) * is;
                        const float g = (inside && xh * ga + be > 0.f) ? acc[mt][nt][r] : 0.f;
                        acc[mt][nt][r] = g;
                        s1 += g;
                        s2 += g * xh;
                    }
                }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lh == 0 && cok) {
                    float* o = p.bn_part + ((long long)slab * p.Cout + co) * 2;
                    o[0] = s1;
                    o[1] = s2;
                }
            }
        }
        if (TRAIN && p.accum && interior) {          // y += acc: the loads of the wave tile first, then adds + stores
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dy = 2 * (wm * MT + mt) + ((r & 3) >> 1), dx = 4 * (r >> 2) + (r & 1);
                    const int srow = (dy * p.ysH + dx * p.ysW) * 4;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt][r] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off * 4 + nt * 128, srow, 0));
                }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int Tt = wm * MT + mt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // row = (r&3) + 8*(r>>2) + 4*lh within the 32-row tile -> window w = 2*(r>>2)+lh, pos = r&3
                const int dy = 2 * Tt + ((r & 3) >> 1), dx = 4 * (r >> 2) + (r & 1);     // + 2*lh in x (lane_off)
                float* const rowp = tile + (long long)dy * p.ysH + (long long)dx * p.ysW;
                if (interior) {
                    const int srow = (dy * p.ysH + dx * p.ysW) * 4;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float v = acc[mt][nt][r];
                        if (relu) v = fmaxf(v, 0.f);
                        if (SPLIT) amax = fmaxf(amax, fabsf(v));
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, lane_off * 4 + nt * 128, srow, 0);
                    }
                } else if (oy0 + dy < p.H && ox0 + dx + 2 * lh < p.W) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float v = acc[mt][nt][r];
                        if (relu) v = fmaxf(v, 0.f);
                        if (co0 + cw + nt * 32 + li < p.Cout) {
                            if (TRAIN && p.accum) v += rowp[lane_off + nt * 32];
                            if (SPLIT) amax = fmaxf(amax, fabsf(v));
                            rowp[lane_off + nt * 32] = v;
                        }
                    }
                }
            }
        }
    } else {
        const int Hp = (p.H + 1) >> 1, Wp = (p.W + 1) >> 1;
        const int py0 = oy0 >> 1, px0 = ox0 >> 1;
        float* const tile = p.y + (long long)n * p.ysN + (long long)py0 * p.ysH + (long long)px0 * p.ysW + co0 + cw;
        const unsigned lane_off = (unsigned)(lh * p.ysW + li);
        const bool interior = (oy0 + TH <= p.H) && (ox0 + 16 <= p.W) && (co0 + BN <= p.Cout);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int dyp = wm * MT + mt;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float* const rowp = tile + (long long)dyp * p.ysH + (long long)(2 * g) * p.ysW;   // + lh in x (lane_off)
                if (interior) {
                    const int srow = (dyp * p.ysH + 2 * g * p.ysW) * 4;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float v = fmaxf(fmaxf(acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1]), fmaxf(acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]));
                        if (relu) v = fmaxf(v, 0.f);
                        if (SPLIT) amax = fmaxf(amax, fabsf(v));
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, lane_off * 4 + nt * 128, srow, 0);
                    }
                } else {
                    const int pyp = py0 + dyp, pxp = px0 + 2 * g + lh;
                    if (pyp < Hp && pxp < Wp) {
                        const bool okx = (2 * pxp + 1 < p.W), oky = (2 * pyp + 1 < p.H);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            float v = acc[mt][nt][4 * g];
                            if (okx) v = fmaxf(v, acc[mt][nt][4 * g + 1]);
                            if (oky) v = fmaxf(v, acc[mt][nt][4 * g + 2]);
                            if (okx && oky) v = fmaxf(v, acc[mt][nt][4 * g + 3]);
                            if (relu) v = fmaxf(v, 0.f);
                            if (co0 + cw + nt * 32 + li < p.Cout) {
                                if (SPLIT) amax = fmaxf(amax, fabsf(v));
                                rowp[lane_off + nt * 32] = v;
                            }
                        }
                    }
                }
            }
        }
    }
    if (ymax_n != nullptr) ccst_absmax_publish(ymax_n, amax, blockIdx.x, peeked);
}

template <int WM, int WN, int NT, bool POOL, bool TRAIN = false, bool SPLIT = false>
int launch_halo(HaloArgs& a, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 32 * NT * WN, TH = BM / 16;
    a.tilesN = (a.Cout + BN - 1) / BN;
    a.tilesY = (a.H + TH - 1) / TH;
    a.tilesX = (a.W + 15) / 16;
    const long long grid = (long long)a.N * a.tilesY * a.tilesX * a.tilesN;
    if (grid <= 0 || grid > 0x7fffffffLL) {
        ccst_set_error("conv3x3_halo: bad grid %lld", grid);
        return CCST_EINVAL;
    }
    hipLaunchKernelGGL((conv3x3_halo_kernel<WM, WN, NT, POOL, TRAIN, SPLIT>), dim3((unsigned)grid), dim3(256), 0, s, a);
    return ccst_launch_status("conv3x3_halo");
}

// OIHW 3x3 -> the pre-split weight image of the SPLIT kernels: [tap][Cin/16][cout_pad][16 words] with words 0..7 = the 16 input
// channels of the chunk as half(w * scale) (two per word, even channel in the low half), words 8..15 = half(w * scale - hi)
// transpose: the weight of the backward-data conv dY -> dX of the same layer -- rows = the forward conv's INPUT channels, k = its output
// channels (cout / cin name the GEMM's sides then: n = `cout` rows, k = `cin`); the taps stay in forward order (the kernel's flip
// reverses them).
__device__ __forceinline__ void pack_halo_split_words(const float* __restrict__ w, unsigned* __restrict__ out, int cout, int cin, int cout_pad,
                                                      const unsigned* __restrict__ wmax, int transpose) {
    const int kw = ccst_scale_exp(ccst_absmax_read(wmax), CCST_SPLIT_W_TARGET);      // the conv kernel derives the same exponent
    const int nch = cin / 16;
    const long long total = 9LL * nch * cout_pad * 16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int word = (int)(i & 15);
        long long j = i >> 4;
        const int co = (int)(j % cout_pad);
        j /= cout_pad;
        const int chunk = (int)(j % nch), tap = (int)(j / nch);
        const int piece = word >> 3, k0 = chunk * 16 + 2 * (word & 7);
        unsigned r = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const long long src = transpose ? ((long long)(k0 + e) * cout + co) * 9 + tap : ((long long)co * cin + k0 + e) * 9 + tap;
            const float v = (co < cout) ? __builtin_ldexpf(w[src], kw) : 0.f;
            const _Float16 h = (_Float16)v;
            const _Float16 q = piece ? (_Float16)(v - (float)h) : h;
            r |= (unsigned)__builtin_bit_cast(unsigned short, q) << (16 * e);
        }
        out[i] = r;
    }
}
__global__ void pack_weight_halo_split_kernel(const float* __restrict__ w, unsigned* __restrict__ out, int cout, int cin, int cout_pad,
                                              const unsigned* __restrict__ wmax, int transpose) {
    pack_halo_split_words(w, out, cout, cin, cout_pad, wmax, transpose);
}
// ... of a whole model in one launch: jobs[j] = {src, dst, n rows, k, cout_pad, |max| words, transpose, 0} (int64 each), blockIdx.y = job
__global__ void pack_weight_halo_split_batch_kernel(const long long* __restrict__ jobs) {
    const long long* jb = jobs + (long long)blockIdx.y * 8;
    pack_halo_split_words(reinterpret_cast<const float*>(jb[0]), reinterpret_cast<unsigned*>(jb[1]), (int)jb[2], (int)jb[3], (int)jb[4],
                          reinterpret_cast<const unsigned*>(jb[5]), (int)jb[6]);
}

}  // namespace

extern "C" int ccst_conv3x3_halo_narrow(int N, int H, int W, int Cout);

// Weights of ccst_conv3x3_halo_split_f32: 9 * cin * cout_pad floats (the same size as the fp32 packed form); cin a multiple of 16,
// cout_pad a multiple of 128.
extern "C" int ccst_pack_conv_weight_halo_split_f32(const float* w_oihw, float* out, int cout, int cin, int cout_pad, const uint32_t* w_absmax,
                                                    int transpose, void* stream) {
    if (transpose) {          // (the GEMM's sides: rows = the forward conv's input channels, k = its output channels)
        const int t = cout;
        cout = cin;
        cin = t;
    }
    CCST_REQUIRE(w_oihw && out && cout > 0 && cin > 0 && cin % 16 == 0, "pack_halo_split: bad args (the k side a multiple of 16)");
    CCST_REQUIRE(w_absmax, "pack_halo_split: the |max| words of the weight (ccst_absmax_f32)");
    CCST_REQUIRE(cout_pad >= cout && cout_pad % 128 == 0, "pack_halo_split: cout_pad must be a multiple of 128 >= cout");
    const long long total = 9LL * cin * cout_pad;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weight_halo_split_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, reinterpret_cast<unsigned*>(out), cout, cin,
                       cout_pad, w_absmax, transpose);
    return ccst_launch_status("pack_weight_halo_split");
}
extern "C" int ccst_pack_conv_weights_halo_split_batch_f32(const int64_t* jobs_device, int njobs, void* stream) {
    CCST_REQUIRE(jobs_device && njobs > 0, "pack_halo_split_batch: bad args");
    hipLaunchKernelGGL(pack_weight_halo_split_batch_kernel, dim3(64, njobs), dim3(256), 0, (hipStream_t)stream, (const long long*)jobs_device);
    return ccst_launch_status("pack_weight_halo_split_batch");
}

// x: NHWC source [N,Hs,Ws,Cin] (Hs = H/2 if CCST_CONV_UPS2), w: packed [9][Cin/4][cout_pad][4], y: NHWC
// [N,H,W,Cout] or its 2x2 ceil-pooled form.  flags: CCST_CONV_RELU | POOL2 | UPS2 | REFLECT.
static int halo_impl(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int cout_pad,
                     uint32_t flags, void* stream, bool bf, const uint32_t* xmax, const uint32_t* wmax, uint32_t* ymax, float* sums);

extern "C" int ccst_conv3x3_halo_f32(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W,
                                     int Cin, int Cout, int cout_pad, uint32_t flags, void* stream) {
    return halo_impl(x, w_packed, bias, y, N, H, W, Cin, Cout, cout_pad, flags, stream, false, nullptr, nullptr, nullptr, nullptr);
}

// The same convolution with every fp32 product as three products of IEEE-half pieces on the 16-bit MFMA (the SPLIT form of the kernel
// above; fp32 accumulation, ~1e-6 of max |y| per layer): w_split from ccst_pack_conv_weight_halo_split_f32; x_absmax / w_absmax: the
// |max| words of x and of the OIHW weight (the operand scales are derived from them on the device); y_absmax: NULL or zeroed words for max |y|.
// chan_sum_partials: NULL or [ccst_conv3x3_halo_split_tiles(N,H,W)][Cout][2], see include/ccst_hip.h.
extern "C" int ccst_conv3x3_halo_split_f32(const float* x, const uint32_t* x_absmax, const float* w_split, const uint32_t* w_absmax,
                                           const float* bias, float* y, uint32_t* y_absmax, int N, int H, int W, int Cin, int Cout, int cout_pad,
                                           uint32_t flags, float* chan_sum_partials, void* stream) {
    CCST_REQUIRE(x_absmax && w_absmax, "conv3x3_halo_split: the |max| words of x and w are required (ccst_absmax_f32 or a producer's y_absmax)");
    CCST_REQUIRE(!(chan_sum_partials && (flags & CCST_CONV_POOL2)), "conv3x3_halo_split: channel sums are of the un-pooled output");
    return halo_impl(x, w_split, bias, y, N, H, W, Cin, Cout, cout_pad, flags, stream, true, x_absmax, w_absmax, y_absmax, chan_sum_partials);
}

// Rows of chan_sum_partials: one per (image, 8x16-pixel tile, wave row), an image's rows contiguous.
extern "C" int ccst_conv3x3_halo_split_tiles(int N, int H, int W) { return N * ((H + 7) / 8) * ((W + 15) / 16) * 2; }

static int halo_impl(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int cout_pad,
                     uint32_t flags, void* stream, bool bf, const uint32_t* xmax, const uint32_t* wmax, uint32_t* ymax, float* sums) {
    CCST_REQUIRE(x && w_packed && y, "conv3x3_halo: null pointer");
    CCST_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0 && Cout > 0, "conv3x3_halo: bad shape");
    CCST_REQUIRE(cout_pad >= Cout && cout_pad % 128 == 0, "conv3x3_halo: cout_pad must be a multiple of 128 >= cout");
    const bool pool = (flags & CCST_CONV_POOL2) != 0, ups = (flags & CCST_CONV_UPS2) != 0;
    if (ups) CCST_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv3x3_halo: upsampled extent must be even");
    if (flags & CCST_CONV_REFLECT) CCST_REQUIRE(H >= 2 && W >= 2, "conv3x3_halo: reflection needs extent >= 2");
    HaloArgs a;
    a.x = x; a.w = w_packed; a.bias = bias; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Hs = ups ? H / 2 : H; a.Ws = ups ? W / 2 : W; a.Cin = Cin; a.Cout = Cout; a.CoutPad = cout_pad;
    a.reflect = (flags & CCST_CONV_REFLECT) ? 1 : 0; a.ups = ups ? 1 : 0; a.relu = (flags & CCST_CONV_RELU) ? 1 : 0;
    a.stats = sums; a.flip = 0; a.accum = 0; a.xmax = xmax; a.wmax = wmax; a.ymax = ymax; a.wpn = CCST_ABSMAX_WORDS;
    a.bn_x = a.bn_mean = a.bn_invstd = a.bn_gamma = a.bn_beta = nullptr; a.bn_part = nullptr;
    CCST_REQUIRE((long long)N * a.Hs * a.Ws * Cin < 0x7fffffffLL, "conv3x3_halo: input must have < 2^31 elements");
    const int oh = pool ? (H + 1) / 2 : H, ow = pool ? (W + 1) / 2 : W;
    a.ysW = Cout; a.ysH = ow * Cout; a.ysN = (long long)oh * ow * Cout;
    hipStream_t s = (hipStream_t)stream;
    // 128x128 tiles unless Cout <= 64 or the grid would leave CUs short of ~3 workgroups (768 slots):
    // same cost model as conv_igemm.hip's choose_tile (rounds x tile area / relative efficiency).
    bool narrow = Cout <= 64;
    if (!narrow) {
        const long long sp = (long long)N * ((H + 7) / 8) * ((W + 15) / 16);
        const double c222 = ceil(sp * ((Cout + 127) / 128) / 768.0) * 2.0 / 1.00;
        const double c221 = ceil(sp * ((Cout + 63) / 64) / 768.0) * 1.0 / 0.93;
        narrow = c221 < c222;
    }
    if (bf) {
        // Cout <= 64 (the 512 x 512 layers): 256 pixels (16 x 16) x 64 channels per workgroup, a wave 64 pixels x 64 channels -- the same 12
        // MFMAs per wave and k-step as the 128x128 tile for a halo of 1.27 instead of 1.41 pixels per output pixel (the 128x64 tile has 6
        // MFMAs per wave for the same halo conversion work: 3.2 other vector instructions per MFMA, 0.50 MFMA-busy)
        if (Cout <= 64 && sums == nullptr && H >= 16)
            return pool ? launch_halo<4, 1, 2, true, false, true>(a, s) : launch_halo<4, 1, 2, false, false, true>(a, s);
        if (narrow) return pool ? launch_halo<2, 2, 1, true, false, true>(a, s) : launch_halo<2, 2, 1, false, false, true>(a, s);
        return pool ? launch_halo<2, 2, 2, true, false, true>(a, s) : launch_halo<2, 2, 2, false, false, true>(a, s);
    }
    if (narrow) return pool ? launch_halo<2, 2, 1, true>(a, s) : launch_halo<2, 2, 1, false>(a, s);
    return pool ? launch_halo<2, 2, 2, true>(a, s) : launch_halo<2, 2, 2, false>(a, s);
}

// The ResNet-trunk form: zero padding, no bias / ReLU / pool; flags: CCST_CONV_FLIP (taps reversed: with the
// transposed packed weight this is the backward-data of the stride-1 conv) | CCST_CONV_ACCUM (y += conv); stats
// (may be NULL): [ccst_conv3x3_halo_stats_groups(N,H,W)][Cout][2] (sum, sum^2) partials of y for the next BatchNorm.
extern "C" int ccst_conv3x3_halo_train_f32(const float* x, const float* w_packed, float* y, float* stats, int N, int H, int W,
                                           int Cin, int Cout, int cout_pad, uint32_t flags, void* stream) {
    CCST_REQUIRE(x && w_packed && y, "conv3x3_halo_train: null pointer");
    CCST_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0 && Cout > 0, "conv3x3_halo_train: bad shape");
    CCST_REQUIRE(cout_pad >= Cout && cout_pad % 128 == 0, "conv3x3_halo_train: cout_pad must be a multiple of 128 >= cout");
    CCST_REQUIRE(!(flags & ~(CCST_CONV_FLIP | CCST_CONV_ACCUM)), "conv3x3_halo_train: only CCST_CONV_FLIP | CCST_CONV_ACCUM");
    CCST_REQUIRE(!(stats && (flags & CCST_CONV_ACCUM)), "conv3x3_halo_train: statistics are of the conv output, not of y += conv");
    CCST_REQUIRE((long long)N * H * W * Cin < 0x7fffffffLL, "conv3x3_halo_train: input must have < 2^31 elements");
    HaloArgs a;
    a.x = x; a.w = w_packed; a.bias = nullptr; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Hs = H; a.Ws = W; a.Cin = Cin; a.Cout = Cout; a.CoutPad = cout_pad;
    a.reflect = 0; a.ups = 0; a.relu = 0;
    a.stats = stats; a.flip = (flags & CCST_CONV_FLIP) ? 1 : 0; a.accum = (flags & CCST_CONV_ACCUM) ? 1 : 0;
    a.xmax = a.wmax = nullptr; a.ymax = nullptr; a.wpn = 0;
    a.bn_x = a.bn_mean = a.bn_invstd = a.bn_gamma = a.bn_beta = nullptr; a.bn_part = nullptr;
    a.ysW = Cout; a.ysH = W * Cout; a.ysN = (long long)H * W * Cout;
    hipStream_t s = (hipStream_t)stream;
    if (ccst_conv3x3_halo_narrow(N, H, W, Cout)) return launch_halo<2, 2, 1, false, true>(a, s);
    return launch_halo<2, 2, 2, false, true>(a, s);
}

// The ResNet-trunk form on half pieces (round 5): as ccst_conv3x3_halo_train_f32 with every fp32 product as three half-piece products
// on the 16-bit MFMA -- x scaled by its |max| words (an activation: left by the BatchNorm apply; a gradient: by the BatchNorm backward),
// w_split from ccst_pack_conv_weight_halo_split_f32 (transpose = 1 + CCST_CONV_FLIP: backward-data).  Same flags, same statistics.
extern "C" int ccst_conv3x3_halo_train_split_f32(const float* x, const uint32_t* x_absmax, const float* w_split, const uint32_t* w_absmax, float* y,
                                                 float* stats, int N, int H, int W, int Cin, int Cout, int cout_pad, uint32_t flags, const float* bn_x,
                                                 const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                                 float* bn_partials, void* stream) {
    CCST_REQUIRE((bn_x == nullptr) == (bn_partials == nullptr) && (bn_x == nullptr || (bn_mean && bn_invstd && bn_gamma && bn_beta)),
                 "conv3x3_halo_train_split: the BatchNorm + ReLU link needs its input, mean, invstd, gamma, beta and the partials buffer together");
    CCST_REQUIRE(!(bn_x && ((flags & CCST_CONV_ACCUM) || stats)), "conv3x3_halo_train_split: the BatchNorm + ReLU link goes with the plain backward-data form");
    CCST_REQUIRE(x && w_split && y && x_absmax && w_absmax, "conv3x3_halo_train_split: null pointer");
    CCST_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0 && Cout > 0, "conv3x3_halo_train_split: bad shape");
    CCST_REQUIRE(cout_pad >= Cout && cout_pad % 128 == 0, "conv3x3_halo_train_split: cout_pad must be a multiple of 128 >= cout");
    CCST_REQUIRE(!(flags & ~(CCST_CONV_FLIP | CCST_CONV_ACCUM)), "conv3x3_halo_train_split: only CCST_CONV_FLIP | CCST_CONV_ACCUM");
    CCST_REQUIRE(!(stats && (flags & CCST_CONV_ACCUM)), "conv3x3_halo_train_split: statistics are of the conv output, not of y += conv");
    CCST_REQUIRE((long long)N * H * W * Cin < 0x7fffffffLL, "conv3x3_halo_train_split: input must have < 2^31 elements");
    HaloArgs a;
    a.x = x; a.w = w_split; a.bias = nullptr; a.y = y;
    a.N = N; a.H = H; a.W = W; a.Hs = H; a.Ws = W; a.Cin = Cin; a.Cout = Cout; a.CoutPad = cout_pad;
    a.reflect = 0; a.ups = 0; a.relu = 0;
    a.stats = stats; a.flip = (flags & CCST_CONV_FLIP) ? 1 : 0; a.accum = (flags & CCST_CONV_ACCUM) ? 1 : 0;
    a.xmax = x_absmax; a.wmax = w_absmax; a.ymax = nullptr; a.wpn = 0;
    a.bn_x = bn_x; a.bn_mean = bn_mean; a.bn_invstd = bn_invstd; a.bn_gamma = bn_gamma; a.bn_beta = bn_beta; a.bn_part = bn_partials;
    a.ysW = Cout; a.ysH = W * Cout; a.ysN = (long long)H * W * Cout;
    hipStream_t s = (hipStream_t)stream;
    if (ccst_conv3x3_halo_narrow(N, H, W, Cout)) return launch_halo<2, 2, 1, false, true, true>(a, s);
    return launch_halo<2, 2, 2, false, true, true>(a, s);
}

// Row groups ccst_conv3x3_halo_train_f32 writes statistics for: two wave rows per 8x16-pixel tile.
extern "C" int ccst_conv3x3_halo_stats_groups(int N, int H, int W) { return N * ((H + 7) / 8) * ((W + 15) / 16) * 2; }

// 1 if the dispatcher above picks the 128x64 tile (bench.py names kernels with it).
extern "C" int ccst_conv3x3_halo_narrow(int N, int H, int W, int Cout) {
    if (Cout <= 64) return 1;
    const long long sp = (long long)N * ((H + 7) / 8) * ((W + 15) / 16);
    const double c222 = ceil(sp * ((Cout + 127) / 128) / 768.0) * 2.0 / 1.00;
    const double c221 = ceil(sp * ((Cout + 63) / 64) / 768.0) * 1.0 / 0.93;
    return c221 < c222 ? 1 : 0;
}
