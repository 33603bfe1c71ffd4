// Implicit-GEMM convolution for gfx950 on the fp32-input MFMA (v_mfma_f32_32x32x2_f32).
//
//   GEMM view:  D[m][co] = sum_{tap,ci} A[m][(tap,ci)] * W[(tap,ci)][co]
//     m  = output pixel (n,oy,ox)            -> BM = 64*WM rows per workgroup
//     co = output channel                    -> BN = 32*NT*WN columns per workgroup
//     k  = (tap, ci), stepped CK=16 input channels of one tap at a time
//   A is gathered on the fly from the NHWC activation (affine index map + zero / reflection
//   padding + optional nearest-x2 upsample on read): no im2col or padded tensor is materialised.
//
//   Workgroup = 4 waves (256 threads), wave tile = 64 x (32*NT): 2 x NT MFMA 32x32 tiles,
//   16 accumulator VGPRs each.  Per k-step the A tile [BM][16] and W tile [16][BN] are staged
//   global -> VGPR -> LDS (double buffered, one barrier per step); fragments are read with
//   ds_read_b128: lane (i=l&31, h=l>>5) takes channels 8q+4h .. 8q+4h+3 of row i, which feed four
//   consecutive MFMAs (k = h  <->  ci = 8q+4h+s); the packed weight layout [tap][ci/4][co][4]
//   gives the B fragment the same shape.  A rows are padded to 20 floats (80 B) so the 16-lane
//   groups of ds_read_b128 hit 16 distinct 16-B slots.
//
//   Measured alternatives (MI355X, 256->256 @128x128, B=6): LDS-DMA staging
//   (global_load_lds_dwordx4 + XOR-swizzled unpadded A image, no ds_write) 124.6 TF vs 129 TF for this
//   register-staged form; a 256x128 tile at 2 waves/SIMD 92-115 TF; static s_setprio / start stagger
//   between co-resident workgroups +-1 %.  A pure MFMA loop of the same shape sustains 155.7 TF.
//
//   Epilogue: bias is the initial accumulator; ReLU; either a strided store (NHWC / NCHW via the
//   y strides) or the fused MaxPool2d(2,2,ceil) where the M index is laid out so that the four
//   pixels of a pooling window sit in the four registers (reg&3) of one lane.
#include "common.h"
#include <type_traits>
#include <math.h>
#include <stdlib.h>

namespace {

typedef ccst_u32x2 u32x2s;
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));


struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    int N, Ho, Wo, Hi, Wi, Cin, Cout, CoutPad;
    int nky, nkx, ay, by, cy, ax, bx, cx;
    int tap_base, tap_sy, tap_sx;
    long long xsN;
    int xsH, xsW;
    long long y_off, ysN;
    int ysH, ysW, ysC;
    unsigned flags;
    int M;          // N*Ho*Wo
    int tilesN;     // column tiles
    int tilesY, tilesX;  // spatial tiles per image (pool mode)
    float invHW, invWo;  // reciprocals for the division-free pixel decode (valid while M < 2^22)
    int fastdiv;
    float* stats;        // optional [row groups][Cout][2]: per-64-row (sum, sum of squares) of the output (BN statistics)
    const unsigned char* rmask;   // pointwise streaming kernel, accumulate form: optional ReLU byte mask applied to the SUM (see the entry point)
    const float* bn_x;            // ... and, with it, the BatchNorm whose output gradient that sum is: its input, saved mean / invstd,
    const float* bn_mean;         //     and where to leave the per-32-row partial sums (sum g, sum g * xhat) of its backward
    const float* bn_invstd;
    const float* bn_gamma;        //     (gamma, beta: only where the ReLU mask is recomputed from bn_x)
    const float* bn_beta;
    float* bn_part;
    const unsigned* xmax;         // half-piece (BFP = 4) form: |max| words of x and of the OIHW weight (common.h): the operands' power-of-two
    const unsigned* wmax;         //     scales are derived from them in the kernel
};

// floor(m / d) for 0 <= m < 2^22 via one float multiply + correction (an integer division is ~40 VALU
// instructions; the row decode runs per thread row in the prologue and per accumulator row in the epilogue).
__device__ __forceinline__ int fdiv(int m, int d, float inv, int fast) {
    if (!fast) return m / d;
    int q = (int)((float)m * inv);
    const int r = m - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

constexpr unsigned CONV_DENSE_OUT = 0x80000000u;   // internal flag: output rows are m*ysW apart
constexpr unsigned CONV_DENSE_IN = 0x40000000u;    // internal flag: 1x1 stride-1 unpadded conv of a dense NHWC input: row m reads x + m*xsW
constexpr int CK_MIN = 16;        // smallest k-step (input channels of one tap per step)

__device__ __forceinline__ int reflect_idx(int i, int n) {
    // ReflectionPad2d: -1 -> 1, n -> n-2 (edge not repeated); clamp keeps overhanging
    // tile rows (results discarded) inside the tensor.
    i = (i < 0) ? -i : i;
    i = (i >= n) ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

// HALFP (round 5; the 64x64 tile with a 32-channel k-step only): the products on the 16-bit MFMA from two IEEE-half pieces per operand, as
// conv1x1_stream_kernel's HALFP form -- the gathered A rows are split where they pass from registers to LDS (rows [32 channels hi | 32
// channels lo], the fp32 row's 128 bytes), the weights arrive pre-split (ccst_pack_conv_weight_split_f32 with all taps), x scaled by
// its |max| words.  Used for BACKWARD-DATA only (the strided 3x3 layers' parity classes, the strided 1x1 downsample branches): the
// operand is a gradient with words at hand, and rounding there moves no ReLU mask.
template <int WM, int WN, int NT, bool POOL, int MT = 2, int CK = 16, bool HALFP = false>
__global__ __launch_bounds__(256, (MT == 1 ? 4 : (MT == 2 && CK == 16) ? 3 : 2)) void conv_igemm_kernel(const ConvArgs p) {
    static_assert(!HALFP || (CK == 32 && MT == 1 && NT == 1 && !POOL), "the half-piece form exists for the 64x64x32 step");
    constexpr int BM = 32 * MT * WM;
    constexpr int BN = 32 * NT * WN;
    constexpr int A_LD = CK + 4;                   // floats per LDS A row (+16 B: conflict-free ds_read_b128)
    constexpr int PPR = CK / 4;                    // float4 parts per A row
    constexpr int AR = BM * PPR / 256;             // A float4 units per thread per step
    constexpr int BUNITS = (CK / 4) * BN;          // W float4 units per step
    constexpr int BR = (BUNITS + 255) / 256;

    __shared__ __attribute__((aligned(16))) float As[2][BM * A_LD];
    constexpr int B_LD = A_LD;                     // HALFP: B rows are output channels, [32 k hi | 32 k lo] as half + pad, like the A rows
    constexpr int PWH = CK / 2;                    // HALFP: words per piece of a row
    __shared__ __attribute__((aligned(16))) float Bs[2][HALFP ? BN * B_LD : CK * BN];
    int kxs = 0, kws = 0;
    if (HALFP) {
        kxs = ccst_scale_exp(ccst_absmax_read(p.xmax), CCST_SPLIT_X_TARGET);
        kws = ccst_scale_exp(ccst_absmax_read(p.wmax), CCST_SPLIT_W_TARGET);
    }
    const float xsc = __uint_as_float((unsigned)(127 + kxs) << 23);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int bid = ccst_xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p.tilesN;
    const int tm = bid / p.tilesN;
    const int co0 = tn * BN;

    // ---- per-thread A rows: unit u = tid + 256*a -> row u / PPR, float4 part u % PPR ------
    // All offsets are 32-bit element offsets (host checks numel < 2^31).
    const int part = tid % PPR;
    int rowIy[AR], rowIx[AR];
    unsigned rowBase[AR];
    int pn = 0, pty = 0, ptx = 0;   // pool-mode tile coordinates
    if (POOL) {
        ptx = tm % p.tilesX;
        pty = (tm / p.tilesX) % p.tilesY;
        pn = tm / (p.tilesX * p.tilesY);
    }
#pragma unroll
    for (int a = 0; a < AR; ++a) {
        const int r = tid / PPR + (256 / PPR) * a;
        int n, oy, ox;
        if (POOL) {
            // row r -> M-tile T=r>>5, window w=(r&31)>>2, pos=r&3: py=2T+(pos>>1), px=2w+(pos&1)
            const int T = r >> 5, i = r & 31;
            n = pn;
            oy = pty * (BM / 16) + 2 * T + ((i & 3) >> 1);
            ox = ptx * 16 + 2 * (i >> 2) + (i & 1);
            oy = min(oy, p.Ho - 1);
            ox = min(ox, p.Wo - 1);
        } else if (p.flags & CONV_DENSE_IN) {
            // pointwise conv (two thirds of the ResNet50 trunk, forward and backward-data): no (n, oy, ox) decode, no bounds --
            // vector instructions are paid on top of the MFMA time, and these tiles have as few as 2 k-steps to amortise them over
            const int m = min(tm * BM + r, p.M - 1);
            rowIy[a] = 0;
            rowIx[a] = 0;
            rowBase[a] = (unsigned)(m * p.xsW + part * 4);
            continue;
        } else {
            int m = tm * BM + r;
            m = min(m, p.M - 1);
            n = fdiv(m, p.Ho * p.Wo, p.invHW, p.fastdiv);
            const int rem = m - n * (p.Ho * p.Wo);
            oy = fdiv(rem, p.Wo, p.invWo, p.fastdiv);
            ox = rem - oy * p.Wo;
        }
        rowIy[a] = oy * p.ay + p.cy;
        rowIx[a] = ox * p.ax + p.cx;
        rowBase[a] = (unsigned)(n * (int)p.xsN + part * 4);
    }
    const bool reflect = (p.flags & CCST_CONV_REFLECT) != 0;
    const int ups = (p.flags & CCST_CONV_UPS2) ? 1 : 0;

    const int nchunks = p.Cin / CK;
    const int T = p.nky * p.nkx * nchunks;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = co0 + wn * (32 * NT) + nt * 32 + li;
        const float b = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = b;
    }

    // ---- staging registers + step state --------------------------------------------------
    f32x4 ra[AR], rb[BR];
    unsigned aoff[AR];    // element offset of the row's pixel for the current tap (always in-bounds)
    bool aok[AR];         // false: zero padding (the loaded value is discarded at the LDS write)
    unsigned boff[BR];    // per-thread part of the packed-weight offset
#pragma unroll
    for (int b = 0; b < BR; ++b) {
        const int u = min(tid + 256 * b, BUNITS - 1);
        const int g = u / BN, col = u - g * BN;
        boff[b] = (unsigned)((g * p.CoutPad + co0 + col) * 4);
    }
    int ky = 0, kx = 0, c = 0;

    auto tap_offsets = [&](int ky_, int kx_) {
        if (p.flags & CONV_DENSE_IN) {
#pragma unroll
            for (int a = 0; a < AR; ++a) {
                aok[a] = true;
                aoff[a] = rowBase[a];
            }
            return;
        }
#pragma unroll
        for (int a = 0; a < AR; ++a) {
            int iy = rowIy[a] + ky_ * p.by, ix = rowIx[a] + kx_ * p.bx;
            bool ok = true;
            if (reflect) {
                iy = reflect_idx(iy, p.Hi);
                ix = reflect_idx(ix, p.Wi);
            } else {
                ok = (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
                iy = min(max(iy, 0), p.Hi - 1);
                ix = min(max(ix, 0), p.Wi - 1);
            }
            iy >>= ups;
            ix >>= ups;
            aok[a] = ok;
            aoff[a] = rowBase[a] + (unsigned)(iy * p.xsH + ix * p.xsW);
        }
    };
    auto load_step = [&](int ky_, int kx_, int c_) {
        const float* xc = p.x + c_ * CK;                                   // uniform
#pragma unroll
        for (int a = 0; a < AR; ++a) ra[a] = *reinterpret_cast<const f32x4*>(xc + aoff[a]);
        const int tap = p.tap_base + ky_ * p.tap_sy + kx_ * p.tap_sx;
        const float* wc = p.w + ((long long)tap * (p.Cin / 4) + c_ * (CK / 4)) * p.CoutPad * 4;   // uniform
#pragma unroll
        for (int b = 0; b < BR; ++b) rb[b] = *reinterpret_cast<const f32x4*>(wc + boff[b]);
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int a = 0; a < AR; ++a) {
            const int r = tid / PPR + (256 / PPR) * a;
            f32x4 v = ra[a];
            if (!aok[a]) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (HALFP) {
                u32x2s hi, lo;
                ccst_split4_half(v, xsc, hi, lo);
                *reinterpret_cast<u32x2s*>(&As[buf][r * A_LD + part * 2]) = hi;
                *reinterpret_cast<u32x2s*>(&As[buf][r * A_LD + PWH + part * 2]) = lo;
            } else {
                *reinterpret_cast<f32x4*>(&As[buf][r * A_LD + part * 4]) = v;
            }
        }
#pragma unroll
        for (int b = 0; b < BR; ++b) {
            const int u = tid + 256 * b;
            if (HALFP) {          // unit u = (k quad u / BN, column u % BN), pre-split: four hi halves | four lo halves
                float* row = &Bs[buf][(u % BN) * B_LD];
                *reinterpret_cast<u32x2s*>(row + (u / BN) * 2) = u32x2s{__float_as_uint(rb[b][0]), __float_as_uint(rb[b][1])};
                *reinterpret_cast<u32x2s*>(row + PWH + (u / BN) * 2) = u32x2s{__float_as_uint(rb[b][2]), __float_as_uint(rb[b][3])};
                continue;
            }
            if (BUNITS % 256 == 0 || u < BUNITS) *reinterpret_cast<f32x4*>(&Bs[buf][u * 4]) = rb[b];
        }
    };
    auto advance = [&]() {
        if (++c == nchunks) {
            c = 0;
            if (++kx == p.nkx) {
                kx = 0;
                ++ky;
            }
            tap_offsets(ky, kx);
        }
    };

    const float* aRd0 = &As[0][(wm * (32 * MT) + li) * A_LD + lh * 4];
    const float* bRd0 = &Bs[0][(lh * BN + wn * (32 * NT) + li) * 4];
    constexpr int NQ = CK / 8;        // fragment reads per step (8 channels each)
    constexpr int NG = 4 * NQ;        // MFMA groups per step: one k-pair of every (mt, nt) tile each
    f32x4 af[NQ][MT], bf[NQ][NT];
    auto read_frags = [&](int buf, int q) {
        const float* aRd = aRd0 + buf * (BM * A_LD);
        const float* bRd = bRd0 + buf * (CK * BN);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[q][mt] = *reinterpret_cast<const f32x4*>(aRd + mt * 32 * A_LD + q * 8);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[q][nt] = *reinterpret_cast<const f32x4*>(bRd + (2 * q * BN + nt * 32) * 4);
    };
    auto mfma_group = [&](int g) {
        const int q = g >> 2, s_ = g & 3;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][mt][s_], bf[q][nt][s_], acc[mt][nt], 0, 0, 0);
    };

    // One k-step = NG MFMA groups with the staging slotted LATE between them (same structure and the same
    // measurements as conv3x3_halo.hip): a step boundary is barrier -> ds_read -> MFMA only; the registers
    // fetched during the previous step (data of step t+1) go to the idle LDS buffers in front of the last two
    // groups, the fetch for step t+2 is issued in front of the last group, and each later 8-channel fragment
    // pair is read one group ahead of its first use.  STORE/LOAD are compile-time so that the peeled last two
    // steps stay branch-free (hipcc hoists conservative vmcnt waits above the MFMAs otherwise).
    // HALFP step: a lane's 8 consecutive channels of k-block kb (16 channels) = words 8 kb + 4 lh of the hi piece, + PWH for lo
    const float* aRdH = &As[0][(wm * 32 + li) * A_LD + lh * 4];
    const float* bRdH = &Bs[0][(wn * 32 + li) * B_LD + lh * 4];
    auto step_half = [&](int t, auto do_store, auto do_load) {
        const int buf = t & 1;
        const float* ar = aRdH + buf * (BM * A_LD);
        const float* br = bRdH + buf * (BN * B_LD);
        f16x8s af_[2][2], bf_[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                af_[kb][q] = __builtin_bit_cast(f16x8s, *reinterpret_cast<const f32x4*>(ar + q * PWH + 8 * kb));
                bf_[kb][q] = __builtin_bit_cast(f16x8s, *reinterpret_cast<const f32x4*>(br + q * PWH + 8 * kb));
            }
        auto block = [&](int kb) {      // a_lo b_hi + a_hi b_lo + a_hi b_hi, the smallest first
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[kb][1], bf_[kb][0], acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[kb][0], bf_[kb][1], acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[kb][0], bf_[kb][0], acc[0][0], 0, 0, 0);
        };
        block(0);
        if (decltype(do_store)::value) {
            __builtin_amdgcn_sched_barrier(0);
            store_step(buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (decltype(do_load)::value) {
            __builtin_amdgcn_sched_barrier(0);
            advance();
            load_step(ky, kx, c);
            __builtin_amdgcn_sched_barrier(0);
        }
        block(1);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](int t, auto do_store, auto do_load) {
        if (HALFP) {
            step_half(t, do_store, do_load);
            return;
        }
        const int buf = t & 1;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g == 0) read_frags(buf, 0);
            if (g == NG - 2 && decltype(do_store)::value) {
                __builtin_amdgcn_sched_barrier(0);
                store_step(buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g == NG - 1 && decltype(do_load)::value) {
                __builtin_amdgcn_sched_barrier(0);
                advance();
                load_step(ky, kx, c);
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((g & 3) == 3 && (g >> 2) + 1 < NQ) {
                __builtin_amdgcn_sched_barrier(0);
                read_frags(buf, (g >> 2) + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma_group(g);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;

    tap_offsets(0, 0);
    load_step(0, 0, 0);
    store_step(0);
    if (T > 1) {
        advance();
        load_step(ky, kx, c);      // step 1's data stays in registers until step 0 stores it
    }
    __syncthreads();
    for (int t = 0; t < T - 2; ++t) {
        step(t, Yes{}, Yes{});
        __syncthreads();
    }
    if (T > 1) {
        step(T - 2, Yes{}, No{});
        __syncthreads();
    }
    step(T - 1, No{}, No{});

    // ---- epilogue ------------------------------------------------------------------------
    if (HALFP) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = __builtin_ldexpf(acc[0][0][r], -(kxs + kws));      // (exact)
    }
    const bool relu = (p.flags & CCST_CONV_RELU) != 0;
    float* yb = p.y + p.y_off;
    if (!POOL && p.stats != nullptr) {
        // BatchNorm statistics for free: this wave's 64 rows x 32*NT columns -> per-column (sum, sum^2) partials,
        // slab index = tile row * WM + wave row.  Rows beyond M are excluded; the fp64 combine happens in the BN
        // finalize kernel, so the result does not depend on the tiling beyond fp32 partial-sum rounding.
        // (vector instructions are paid on top of the MFMA time on gfx950: the rows-beyond-M mask, three instructions per accumulator,
        //  is only applied to the last, partial row tile)
        const bool full_rows = (tm * BM + BM <= p.M);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float s1 = 0.f, s2 = 0.f;
            if (full_rows) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[mt][nt][r];
                        s1 += v;
                        s2 += v * v;
                    }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * (32 * MT) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float v = (tm * BM + row < p.M) ? acc[mt][nt][r] : 0.f;
                        s1 += v;
                        s2 += v * v;
                    }
            }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const int co = co0 + wn * (32 * NT) + nt * 32 + li;
            if (lh == 0 && co < p.Cout) {
                float* o = p.stats + ((long long)(tm * WM + wm) * p.Cout + co) * 2;
                o[0] = s1;
                o[1] = s2;
            }
        }
    }
    const bool dense = (p.flags & CONV_DENSE_OUT) != 0;       // &y[m] = y + m*ysW: no (n,oy,ox) decode
    const bool accum = (p.flags & CCST_CONV_ACCUM) != 0;      // y += conv (host guarantees: no ReLU, no pool)
    if (!POOL && dense && p.ysC == 1 && tm * BM + BM <= p.M && co0 + BN <= p.Cout) {
        // Full tile of a dense NHWC output (every ResNet layer, most of the time): store through a buffer resource
        // on this wave's tile origin -- one per-lane byte offset for the whole epilogue, the row in the scalar
        // offset, no predicates, no vector address arithmetic (as conv3x3_halo.hip).
        float* const tile = yb + (long long)(tm * BM + wm * (32 * MT)) * p.ysW + co0 + wn * (32 * NT);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 0x7fffffff, 0x00020000);
        const unsigned lane_off = (unsigned)(4 * lh * p.ysW + li) * 4u;
        if (accum) {      // y += acc: all loads of a wave tile first (64 in flight), then the adds and stores
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int srow = (mt * 32 + (r & 3) + 8 * (r >> 2)) * p.ysW * 4;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt][r] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off + nt * 128, srow, 0));
                }
        }
        if (relu) {       // one uniform branch, not a max + select per accumulator
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaxf(acc[mt][nt][r], 0.f);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = (mt * 32 + (r & 3) + 8 * (r >> 2)) * p.ysW * 4;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[mt][nt][r]), rsrc, lane_off + nt * 128, srow, 0);
            }
    } else if (!POOL) {
        const int HW = p.Ho * p.Wo;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (32 * MT) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = tm * BM + row;
                if (m < p.M) {
                    float* yrow;
                    if (dense) {
                        yrow = yb + (long long)m * p.ysW;
                    } else {
                        const int n = fdiv(m, HW, p.invHW, p.fastdiv);
                        const int rem = m - n * HW;
                        const int oy = fdiv(rem, p.Wo, p.invWo, p.fastdiv);
                        const int ox = rem - oy * p.Wo;
                        yrow = yb + (long long)n * p.ysN + (long long)oy * p.ysH + (long long)ox * p.ysW;
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int co = co0 + wn * (32 * NT) + nt * 32 + li;
                        float v = acc[mt][nt][r];
                        if (relu) v = fmaxf(v, 0.f);
                        if (co < p.Cout) {
                            if (accum) v += yrow[(long long)co * p.ysC];
                            yrow[(long long)co * p.ysC] = v;
                        }
                    }
                }
            }
        }
    } else {
        const int Hp = (p.Ho + 1) >> 1, Wp = (p.Wo + 1) >> 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int Tt = wm * MT + mt;
            const int pyp = pty * (BM / 32) + Tt;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int pxp = ptx * 8 + 2 * g + lh;
                if (pyp < Hp && pxp < Wp) {
                    const bool okx = (2 * pxp + 1 < p.Wo), oky = (2 * pyp + 1 < p.Ho);
                    float* yrow = yb + (long long)pn * p.ysN + (long long)pyp * p.ysH + (long long)pxp * p.ysW;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int co = co0 + wn * (32 * NT) + nt * 32 + li;
                        float v = acc[mt][nt][4 * g];
                        if (okx) v = fmaxf(v, acc[mt][nt][4 * g + 1]);
                        if (oky) v = fmaxf(v, acc[mt][nt][4 * g + 2]);
                        if (okx && oky) v = fmaxf(v, acc[mt][nt][4 * g + 3]);
                        if (relu) v = fmaxf(v, 0.f);
                        if (co < p.Cout) yrow[(long long)co * p.ysC] = v;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Pointwise (1x1, stride 1, dense NHWC in and out) convolution as a PERSISTENT streaming GEMM: D[M][N] = A[M][K] W[K][N].
// Two thirds of the ResNet50 trunk, forward and backward-data, with K = 64..2048: 2..64 k-steps per 64x64 tile.  In the kernel
// above a tile is a workgroup: its first loads (~2.5 us before the first MFMA) and its epilogue (~1.5 us of statistics and
// stores) are not covered by anything, because the workgroups that share a CU start together and stay in phase -- measured on the
// 6.6-GFLOP layers: 72-85 us, of which 42 us MFMA time and 30 us that adds to it instead of hiding behind it.  Here a workgroup
// walks a sequence of tiles and its loader simply runs on: the k-steps of all its tiles form ONE stream (global -> registers two
// steps ahead, registers -> LDS one step ahead), so the next tile's first operands are already in LDS / in flight while the
// current tile's epilogue runs, and workgroups drift out of phase.  Same 64x64x32 step as tile 1222 (one 32x32 MFMA tile per
// wave, 16 MFMAs per step, 4 workgroups per CU).
// ------------------------------------------------------------------------------------------------------------------------
// MASKED: 0 none, 1 the forward's byte mask (ACCUM), 2 recomputed from the BatchNorm's own input (bn(x) > 0; needs BSTATS)
// HALFP: the products on the 16-bit MFMA at fp32 accuracy -- every fp32 product as three products of IEEE-half pieces (v = hi + lo, 22
// significant bits, fp32 accumulation: a_lo b_hi + a_hi b_lo + a_hi b_hi; v_mfma_f32_32x32x16_f16 does 16x the multiply-adds per cycle
// of v_mfma_f32_32x32x2_f32).  LDS rows are [32 channels hi | 32 channels lo] as half = the same 128 bytes + pad as the fp32 row, both
// for the pixel rows of A and the output-channel rows of B ([col][k]: a lane's 8 consecutive k).  Round 5: the activations are split
// where they pass from registers to LDS by ccst_split4_half (8 vector instructions per four values; 14 before), and the WEIGHTS
// arrive pre-split from ccst_pack_conv_weight_split_f32 (a 16-byte unit of the packed layout = four k of one column as four hi
// halves | four lo halves, scaled by the power of two of the weight's |max| words): the B side of the loader is two 8-byte LDS stores
// and no arithmetic.  With both splits in the loader (round 3/4) a k-step carried ~60 vector instructions next to 6 MFMAs and only
// the training forward gained (1.43x); now it carries ~16, and the backward-data forms run on it too -- their operand is a
// gradient, scaled by the |max| words its producer (the BatchNorm backward's apply) leaves.
// (bf16 pieces -- 16 bits with two, 24 with three -- were measured in round 3 and retired: DESIGN / JOURNAL.)
template <bool STATS, bool ACCUM, int MASKED = 0, bool BSTATS = false, bool HALFP = false>
__global__ __launch_bounds__(256, 4) void conv1x1_stream_kernel(const ConvArgs p, int ntiles) {
    constexpr bool BF3 = HALFP;
    constexpr int NPIECE = 2;
    // HALFP: both operands scaled by powers of two from the tensors' |max| words (activations to < 2^14, weights to < 2^10): no finite
    // fp32 value overflows half, small tensors are lifted out of its subnormals; the accumulators are scaled back exactly (v_ldexp)
    int kxs = 0, kws = 0;
    if (HALFP) {
        kxs = ccst_scale_exp(ccst_absmax_read(p.xmax), CCST_SPLIT_X_TARGET);
        kws = ccst_scale_exp(ccst_absmax_read(p.wmax), CCST_SPLIT_W_TARGET);
    }
    const float xsc = __uint_as_float((unsigned)(127 + kxs) << 23);
    constexpr int BM = 64, BN = 64, CK = 32, PPR = CK / 4, AR = BM * PPR / 256, BR = (CK / 4) * BN / 256;
    constexpr int PW = CK / 2;                               // words per piece of a row (32 bf16)
    constexpr int A_LD = BF3 ? NPIECE * PW + 4 : CK + 4;     // BF: a row = [32 channels piece 0 | piece 1 | ...] + 4 words of pad
    constexpr int B_LD = BF3 ? A_LD : 0;                     // BF: B rows are output channels, laid out like the A rows
    __shared__ __attribute__((aligned(16))) float As[2][BM * A_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BF3 ? BN * A_LD : CK * BN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int T = p.Cin / CK;                                                   // k-steps per tile
    // Tiles of this workgroup: ONE column tile tn (the grid is a multiple of the column-tile count) and every strideM-th row tile
    // from tm0 on.  The workgroups an XCD runs side by side (consecutive ids after the remap) are the column tiles of the same few
    // row tiles, so they share A rows through its L2; and because tn never changes, the per-channel statistics of all of a
    // workgroup's tiles add up in registers: 2 * grid / tilesN partial slabs for the BatchNorm finalize instead of one per 32 rows
    // (6272 for the 56x56 layers at B = 64 -- the finalize kernels read 12.8 MB of partials there).
    // (Row classes are dealt to the XCDs in turn -- hardware XCD = blockIdx % 8 -- so that the classes that own one tile more than
    //  the others do not pile up on the first XCDs; the column tiles of a class stay on one XCD.)
    int wg;
    if ((int)gridDim.x % (8 * p.tilesN) == 0) {
        const int xcd = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3;
        wg = ((k / p.tilesN) * 8 + xcd) * p.tilesN + k % p.tilesN;
    } else {
        wg = ccst_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    }
    const int tnW = wg % p.tilesN, tm0 = wg / p.tilesN, strideM = (int)gridDim.x / p.tilesN, tilesM = ntiles / p.tilesN;
    const int mine = (tilesM - tm0 + strideM - 1) / strideM;                    // >= 1: the grid never exceeds the tile count
    const int G = mine * T;                                                     // k-steps of this workgroup

    // ---- loader state (runs up to two k-steps = possibly one tile ahead of the MFMAs) ----
    const int part = tid % PPR;
    unsigned aBase[AR], bBase[BR];
    int lt = 0, lc = 0;                                                         // loader's tile index / chunk
    auto loader_tile = [&](int i) {
        const int tn = tnW, tm = tm0 + i * strideM;
#pragma unroll
        for (int a = 0; a < AR; ++a) {
            const int m = min(tm * BM + tid / PPR + (256 / PPR) * a, p.M - 1);
            if (p.flags & CONV_DENSE_IN) {
                aBase[a] = (unsigned)(m * p.xsW + part * 4);
            } else {            // strided pointwise conv (the downsample branches): pixel (n, oy * ay, ox * ax), always inside the input
                const int n = fdiv(m, p.Ho * p.Wo, p.invHW, p.fastdiv);
                const int rem = m - n * (p.Ho * p.Wo);
                const int oy = fdiv(rem, p.Wo, p.invWo, p.fastdiv);
                const int ox = rem - oy * p.Wo;
                aBase[a] = (unsigned)(n * (int)p.xsN + oy * p.ay * p.xsH + ox * p.ax * p.xsW + part * 4);
            }
        }
#pragma unroll
        for (int b = 0; b < BR; ++b) {
            const int u = tid + 256 * b;
            bBase[b] = (unsigned)(((u / BN) * p.CoutPad + tn * BN + (u % BN)) * 4);
        }
    };
    f32x4 ra[AR], rb[BR];
    auto load_step = [&]() {
        const float* xc = p.x + lc * CK;                                                  // uniform
        const float* wc = p.w + (long long)lc * (CK / 4) * p.CoutPad * 4;                 // uniform
#pragma unroll
        for (int a = 0; a < AR; ++a) ra[a] = *reinterpret_cast<const f32x4*>(xc + aBase[a]);
#pragma unroll
        for (int b = 0; b < BR; ++b) rb[b] = *reinterpret_cast<const f32x4*>(wc + bBase[b]);
    };
    auto advance = [&]() {
        if (++lc == T) {
            lc = 0;
            loader_tile(++lt);
        }
    };
    auto store_step = [&](int buf) {
        if (HALFP) {
#pragma unroll
            for (int a = 0; a < AR; ++a) {          // row = pixel: [32 channels hi | 32 channels lo]
                u32x2s hi, lo;
                ccst_split4_half(ra[a], xsc, hi, lo);
                float* row = &As[buf][(tid / PPR + (256 / PPR) * a) * A_LD];
                *reinterpret_cast<u32x2s*>(row + part * 2) = hi;
                *reinterpret_cast<u32x2s*>(row + PW + part * 2) = lo;
            }
#pragma unroll
            for (int b = 0; b < BR; ++b) {          // unit u = (k quad u / BN, column u % BN), pre-split: row = output channel
                const int u = tid + 256 * b;
                float* row = &Bs[buf][(u % BN) * B_LD];
                *reinterpret_cast<u32x2s*>(row + (u / BN) * 2) = u32x2s{__float_as_uint(rb[b][0]), __float_as_uint(rb[b][1])};
                *reinterpret_cast<u32x2s*>(row + PW + (u / BN) * 2) = u32x2s{__float_as_uint(rb[b][2]), __float_as_uint(rb[b][3])};
            }
            return;
        }
#pragma unroll
        for (int a = 0; a < AR; ++a) *reinterpret_cast<f32x4*>(&As[buf][(tid / PPR + (256 / PPR) * a) * A_LD + part * 4]) = ra[a];
#pragma unroll
        for (int b = 0; b < BR; ++b) *reinterpret_cast<f32x4*>(&Bs[buf][(tid + 256 * b) * 4]) = rb[b];
    };

    const float* aRd0 = &As[0][(wm * 32 + li) * A_LD + lh * 4];
    const float* bRd0 = &Bs[0][(lh * BN + wn * 32 + li) * 4];
    constexpr int NQ = CK / 8, NG = 4 * NQ;
    f32x4 af[NQ], bf[NQ];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto read_frags = [&](int buf, int q) {
        af[q] = *reinterpret_cast<const f32x4*>(aRd0 + buf * (BM * A_LD) + q * 8);
        bf[q] = *reinterpret_cast<const f32x4*>(bRd0 + buf * (CK * BN) + 2 * q * BN * 4);
    };
    // BF3 fragments: a lane's 8 consecutive channels of k-block kb (16 channels): words 8 kb + 4 lh of the hi half, + 16 for lo
    const float* aRdB = &As[0][(wm * 32 + li) * A_LD + lh * 4];
    const float* bRdB = &Bs[0][(wn * 32 + li) * B_LD + lh * 4];
    auto step_bf3 = [&](int buf, auto do_store, auto do_load) {
        const float* ar = aRdB + buf * (BM * A_LD);
        const float* br = bRdB + buf * (BN * B_LD);
        constexpr int NPC = NPIECE;
        f16x8s af_[2][NPC], bf_[2][NPC];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int q = 0; q < NPC; ++q) {
                af_[kb][q] = __builtin_bit_cast(f16x8s, *reinterpret_cast<const f32x4*>(ar + q * PW + 8 * kb));
                bf_[kb][q] = __builtin_bit_cast(f16x8s, *reinterpret_cast<const f32x4*>(br + q * PW + 8 * kb));
            }
        // products of pieces (i, j) with i + j < NPC, the smallest first
        auto block = [&](int kb) {
#pragma unroll
            for (int sum = NPC - 1; sum >= 0; --sum)
#pragma unroll
                for (int i = sum; i >= 0; --i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[kb][i], bf_[kb][sum - i], acc, 0, 0, 0);
        };
        block(0);
        if (decltype(do_store)::value) {
            __builtin_amdgcn_sched_barrier(0);
            store_step(buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (decltype(do_load)::value) {
            __builtin_amdgcn_sched_barrier(0);
            advance();
            load_step();
            __builtin_amdgcn_sched_barrier(0);
        }
        block(1);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](int buf, auto do_store, auto do_load) {
        if (BF3) {
            step_bf3(buf, do_store, do_load);
            return;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g == 0) read_frags(buf, 0);
            if (g == NG - 2 && decltype(do_store)::value) {
                __builtin_amdgcn_sched_barrier(0);
                store_step(buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g == NG - 1 && decltype(do_load)::value) {
                __builtin_amdgcn_sched_barrier(0);
                advance();
                load_step();
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((g & 3) == 3 && (g >> 2) + 1 < NQ) {
                __builtin_amdgcn_sched_barrier(0);
                read_frags(buf, (g >> 2) + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g >> 2][g & 3], bf[g >> 2][g & 3], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- epilogue of the tile the MFMAs just finished: statistics, (accumulate,) stores through ONE buffer resource over the whole
    // output (rows beyond M are beyond its size: the hardware drops those stores and answers those loads with zeros) ----
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + p.y_off, 0, (int)((unsigned)p.M * (unsigned)p.ysW * 4u), 0x00020000);
    int ct = 0, ci = 0;                                                         // compute side: chunk in tile, tile index
    float rs1 = 0.f, rs2 = 0.f;                                                 // running per-channel sums over this workgroup's tiles
    const int col = tnW * BN + wn * 32 + li;
    auto finish_tile = [&]() {
        if (++ct < T) return;
        ct = 0;
        const int tm = tm0 + (ci++) * strideM;
        const int row0 = tm * BM + wm * 32;
        if (HALFP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = __builtin_ldexpf(acc[r], -(kxs + kws));      // (exact)
        }
        if (STATS) {
            float s1 = 0.f, s2 = 0.f;
            if (tm * BM + BM <= p.M) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s1 += acc[r];
                    s2 += acc[r] * acc[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = (row0 + (r & 3) + 8 * (r >> 2) + 4 * lh < p.M) ? acc[r] : 0.f;
                    s1 += v;
                    s2 += v * v;
                }
            }
            rs1 += s1;
            rs2 += s2;
        }
        const unsigned voff = (unsigned)(((row0 + 4 * lh) * p.ysW + col) * 4);
        if (ACCUM) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[r] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, voff, (unsigned)(((r & 3) + 8 * (r >> 2)) * p.ysW * 4), 0));
        }
        if (MASKED == 1) {  // the sum is the gradient of a ReLU output: zero where the forward's byte mask (bit j of byte i <-> element 4 i + j) says so
            const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.rmask), 0, (int)((unsigned)p.M * (unsigned)(p.ysW / 4)), 0x00020000);
            const unsigned moff = voff >> 4;
            unsigned char mb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) mb[r] = __builtin_amdgcn_raw_buffer_load_b8(mrs, moff, (unsigned)(((r & 3) + 8 * (r >> 2)) * (p.ysW / 4)), 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = ((mb[r] >> (li & 3)) & 1) ? acc[r] : 0.f;
        }
        if (BSTATS) {       // the (masked) result g is the output gradient of a BatchNorm: its backward's per-channel partial sums, from here
            const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bn_x), 0, (int)((unsigned)p.M * (unsigned)p.ysW * 4u), 0x00020000);
            const float mu = p.bn_mean[col], is = p.bn_invstd[col];
            float xv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, voff, (unsigned)(((r & 3) + 8 * (r >> 2)) * p.ysW * 4), 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = (xv[r] - mu) * is;                     // xhat
            if (MASKED == 2) {  // that BatchNorm ends in a ReLU without a residual: its mask is bn(x) > 0, and what is stored is the masked gradient
                const float ga = p.bn_gamma[col], be = p.bn_beta[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = (xv[r] * ga + be > 0.f) ? acc[r] : 0.f;
            }
            float s1 = 0.f, s2 = 0.f;
            const bool full = tm * BM + BM <= p.M;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float gq = (full || row0 + (r & 3) + 8 * (r >> 2) + 4 * lh < p.M) ? acc[r] : 0.f;
                s1 += gq;
                s2 += gq * xv[r];
            }
            rs1 += s1;
            rs2 += s2;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r]), yrs, voff, (unsigned)(((r & 3) + 8 * (r >> 2)) * p.ysW * 4), 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    };

    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    loader_tile(0);
    load_step();
    store_step(0);
    if (G > 1) {
        advance();
        load_step();
    }
    __syncthreads();
    for (int g = 0; g < G - 2; ++g) {
        step(g & 1, Yes{}, Yes{});
        finish_tile();
        __syncthreads();
    }
    if (G > 1) {
        step((G - 2) & 1, Yes{}, No{});
        finish_tile();
        __syncthreads();
    }
    step((G - 1) & 1, No{}, No{});
    finish_tile();
    if (STATS || BSTATS) {      // one slab per (row-tile class, wave row): [2 * strideM][Cout][2]
        rs1 += __shfl_xor(rs1, 32, 64);
        rs2 += __shfl_xor(rs2, 32, 64);
        if (lh == 0) {
            float* o = (STATS ? p.stats : p.bn_part) + ((long long)(tm0 * 2 + wm) * p.Cout + col) * 2;
            o[0] = rs1;
            o[1] = rs2;
        }
    }
}

// OIHW -> packed [tap][K/4][n_pad][4]
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int ntap,
                                   int transpose, int k_pad, int n_pad) {
    const long long total = (long long)ntap * k_pad * n_pad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k4 = (int)(i & 3);
        long long j = i >> 2;
        const int ncol = (int)(j % n_pad);
        j /= n_pad;
        const int kg = (int)(j % (k_pad / 4));
        const int tap = (int)(j / (k_pad / 4));
        const int k = kg * 4 + k4;
        const int ci = transpose ? ncol : k;
        const int co = transpose ? k : ncol;
        float v = 0.f;
        if (ci < cin && co < cout) v = w[((long long)co * cin + ci) * ntap + tap];
        out[i] = v;
    }
}

// The same packing for a whole model in ONE launch: jobs[j] = {src, dst, cout, cin, ntap, transpose, k_pad, n_pad}
// (int64 each, device resident); blockIdx.y = job, blockIdx.x strides over the job's elements.  After every
// optimiser step ResNet50 re-packs 105 weight tensors: 105 launches (1.7 ms of host time, ~0.5 ms of GPU time in
// 5-us kernels) become one.
__global__ void pack_weight_batch_kernel(const long long* __restrict__ jobs) {
    const long long* jb = jobs + (long long)blockIdx.y * 8;
    const float* __restrict__ w = reinterpret_cast<const float*>(jb[0]);
    float* __restrict__ out = reinterpret_cast<float*>(jb[1]);
    const int cout = (int)jb[2], cin = (int)jb[3], ntap = (int)jb[4], transpose = (int)jb[5], k_pad = (int)jb[6], n_pad = (int)jb[7];
    const long long total = (long long)ntap * k_pad * n_pad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k4 = (int)(i & 3);
        long long j = i >> 2;
        const int ncol = (int)(j % n_pad);
        j /= n_pad;
        const int kg = (int)(j % (k_pad / 4));
        const int tap = (int)(j / (k_pad / 4));
        const int k = kg * 4 + k4;
        const int ci = transpose ? ncol : k;
        const int co = transpose ? k : ncol;
        float v = 0.f;
        if (ci < cin && co < cout) v = w[((long long)co * cin + ci) * ntap + tap];
        out[i] = v;
    }
}

// A 1x1 weight in the pointwise kernel's packed layout [K/4][n_pad] of 16-byte units, PRE-SPLIT for its half-piece form: unit (kg, col)
// = the four k = 4 kg .. 4 kg + 3 of column col as four hi halves | four lo halves of w * 2^kw (kw from the weight's |max| words, the
// exponent the conv kernel derives from the same words: CCST_SPLIT_W_TARGET).  The same 16 bytes per unit as the fp32 layout.
__device__ __forceinline__ void pack_split_units(const float* __restrict__ w, const unsigned* __restrict__ wmax, float* __restrict__ out, int cout,
                                                 int cin, int ntap, int transpose, int k_pad, int n_pad) {
    const int kw = ccst_scale_exp(ccst_absmax_read(wmax), CCST_SPLIT_W_TARGET);
    const float wsc = __uint_as_float((unsigned)(127 + kw) << 23);
    const long long units = (long long)ntap * (k_pad / 4) * n_pad;          // [tap][K/4][n_pad], as ccst_pack_conv_weight_f32
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += (long long)gridDim.x * blockDim.x) {
        const int ncol = (int)(i % n_pad);
        const long long t = i / n_pad;
        const int kg = (int)(t % (k_pad / 4)), tap = (int)(t / (k_pad / 4));
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kg * 4 + j;
            const int ci = transpose ? ncol : k, co = transpose ? k : ncol;
            v[j] = (ci < cin && co < cout) ? w[((long long)co * cin + ci) * ntap + tap] : 0.f;
        }
        ccst_u32x2 hi, lo;
        ccst_split4_half(v, wsc, hi, lo);
        *reinterpret_cast<f32x4*>(out + i * 4) = f32x4{__uint_as_float(hi[0]), __uint_as_float(hi[1]), __uint_as_float(lo[0]), __uint_as_float(lo[1])};
    }
}
__global__ __launch_bounds__(256) void pack_weight_split_kernel(const float* __restrict__ w, const unsigned* __restrict__ wmax, float* __restrict__ out,
                                                                int cout, int cin, int ntap, int transpose, int k_pad, int n_pad) {
    pack_split_units(w, wmax, out, cout, cin, ntap, transpose, k_pad, n_pad);
}
// ... for a whole model in one launch: jobs[j] = {src, dst, cout, cin, |max| words, transpose + 2 * ntap, k_pad, n_pad} (int64 each),
// blockIdx.y = job
__global__ __launch_bounds__(256) void pack_weight_split_batch_kernel(const long long* __restrict__ jobs) {
    const long long* jb = jobs + (long long)blockIdx.y * 8;
    pack_split_units(reinterpret_cast<const float*>(jb[0]), reinterpret_cast<const unsigned*>(jb[4]), reinterpret_cast<float*>(jb[1]), (int)jb[2],
                     (int)jb[3], (int)(jb[5] >> 1), (int)(jb[5] & 1), (int)jb[6], (int)jb[7]);
}

// NCHW (C<=4) -> padded NHWC4
__global__ void nchw_to_nhwc4_pad_kernel(const float* __restrict__ x, f32x4* __restrict__ y, int N, int C, int H, int W,
                                         int pad, int Wp, int reflect) {
    const int Hp = H + 2 * pad;
    const long long total = (long long)N * Hp * Wp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int px = (int)(i % Wp);
        const long long j = i / Wp;
        const int py = (int)(j % Hp);
        const int n = (int)(j / Hp);
        int iy = py - pad, ix = px - pad;
        bool ok = px < W + 2 * pad;
        if (reflect) {
            iy = reflect_idx(iy, H);
            ix = reflect_idx(ix, W);
        } else {
            ok = ok && iy >= 0 && iy < H && ix >= 0 && ix < W;
        }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const float* src = x + ((long long)n * C * H + iy) * W + ix;
            v[0] = src[0];
            if (C > 1) v[1] = src[(long long)H * W];
            if (C > 2) v[2] = src[2LL * H * W];
            if (C > 3) v[3] = src[3LL * H * W];
        }
        y[i] = v;
    }
}

template <int WM, int WN, int NT, bool POOL, int MT = 2, int CK = 16, bool HALFP = false>
int launch_conv(ConvArgs& a, hipStream_t s) {
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    a.tilesN = (a.Cout + BN - 1) / BN;
    int tilesM;
    if (POOL) {
        a.tilesY = (a.Ho + BM / 16 - 1) / (BM / 16);
        a.tilesX = (a.Wo + 15) / 16;
        tilesM = a.N * a.tilesY * a.tilesX;
    } else {
        a.tilesY = a.tilesX = 0;
        tilesM = (a.M + BM - 1) / BM;
    }
    const long long grid = (long long)tilesM * a.tilesN;
    if (grid <= 0 || grid > 0x7fffffffLL) {
        ccst_set_error("conv: bad grid %lld", grid);
        return CCST_EINVAL;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, NT, POOL, MT, CK, HALFP>), dim3((unsigned)grid), dim3(256), 0, s, a);
    return ccst_launch_status("conv_igemm");
}

}  // namespace

// Tile choice.  128x128 (222) is the most efficient tile (measured 129 TF vs 116 for 256x64 and 114 for
// 128x64) but a launch only runs at that rate while every CU holds ~3 workgroups: with 768 resident
// slots a grid of 784 workgroups costs two rounds.  Cost model: rounds(grid / 768 slots) x tile area /
// relative tile efficiency; the cheapest candidate wins.  A 256x128 tile (MT=4, 2 waves/SIMD) and a 32-channel k-step (CK=32, 2
// workgroups/CU) were measured slower (92-115 / 112 TF) and are not dispatched.
// Small problems (the 14x14 and 7x7 ResNet stages at B=64: M = 12544 / 3136) cannot fill 768 slots even with
// 128x64 tiles; they run on 64x64 tiles (one MFMA tile per wave, code 1221) with a 32-channel k-step (1222) so a
// step still carries 16 MFMAs per wave: 4x the workgroups, 4 per CU.  Measured on those layers, forward +
// backward-data: 6.49 -> 6.09 ms per step (7x7 stages -10..14 %, 14x14 stages -1..3 %).
static int choose_tile(int M, int cout, int cin, int taps, bool pool) {
    int tile;
    const long long g221 = (((long long)M + 127) / 128) * ((cout + 63) / 64);
    if (cout <= 32 && !pool) tile = 411;
    else if (!pool && (g221 < 768 || M <= 65536 || cout <= 64)) {
        // 64x64 tiles (see below): under-filled grids, and -- measured per layer on the ResNet50 B=64 shapes, forward
        // and backward-data -- equal or faster than 128x64 / 128x128 everywhere up to 28x28 maps and on the Cout = 64
        // layers at 56x56 (-5 %); without a 32-channel k-step (Cin % 32 != 0, the stems) the larger tiles stay.
        tile = (cin % 32 == 0) ? 1222 : (g221 < 768 ? 1221 : 221);
    } else if (cout <= 64) tile = 221;     // 128x64: 8-10 % faster than 256x64 on the Cout=64 layers (pooled AdaIN form)
    else if (!pool && (long long)cin * taps < 1024) tile = 221;   // large maps, short K (ResNet 1x1 at 56x56): 128x64 beats 128x128 (-3..7 %)
    else {
        const double slots = 768.0;
        const long long g222 = (((long long)M + 127) / 128) * ((cout + 127) / 128);
        const double c222 = ceil(g222 / slots) * (128.0 * 128.0) / 1.00;
        const double c221 = ceil(g221 / slots) * (128.0 * 64.0) / 0.88;
        tile = (c221 < c222) ? 221 : 222;
    }
    return tile;
}

// The persistent pointwise kernel (conv1x1_stream_kernel) takes every 1x1 stride-1 convolution between dense NHWC tensors whose
// channel counts fit its 64x64x32 step.  CCST_CONV_STREAM=0 keeps them on the per-tile kernel (A/B).
// Workgroups of the streaming kernel for a problem: at most 4 per CU, a multiple of the column-tile count, never more than tiles.
// CCST_CONV_BF: how the streaming pointwise kernel forms its products.
//   4 (default): the TRAINING FORWARD (the calls with a statistics epilogue that come with the |max| words of both operands,
//      ccst_conv2d_igemm_stats_scaled_f32) on the 16-bit MFMA from two IEEE-half pieces per operand (22 bits), both operands scaled by
//      powers of two derived from the words on the device, accumulators scaled back exactly: the ten ResNet50 B=64 pointwise shapes
//      932 -> 653 us (tools/igemm_time.py), error 4e-7..1.3e-6 of max |y| (the fp32 MFMA: 2e-7..2e-6), every ResNet fixture and the
//      full-size gradient gate green, train step +3 %.  Range-safe: round 3's form (fixed 2^8 weight scale, unscaled activations)
//      overflowed for |w| >= 256 or |x| >= 65504.  Backward-data and the evaluation forward stay on the fp32 MFMA.
//   0: the fp32 MFMA everywhere.
//   (Round 3's experiments with two / three bf16 pieces, values 2 / 3, are retired: 16 bits failed the gradient gates, 24 bits gained
//   nothing -- DESIGN.md section 8 keeps the numbers; the kernel's BFP = 2 / 3 instantiations are no longer built.)
static int stream_bfp() {
    static const int bfp = [] { const char* e = getenv("CCST_CONV_BF"); const int v = e ? atoi(e) : 4; return v == 4 ? 4 : 0; }();
    return bfp;
}
static int stream_grid(int M, int cout) {
    const int tilesN = cout / 64, ntiles = ((M + 63) / 64) * tilesN;
    const int slots = 1024;
    const int cap = (slots / tilesN) * tilesN;
    return ntiles < cap ? ntiles : (cap > 0 ? cap : tilesN);
}
static bool stream_shape_ok(int M, int cout, int cin, int taps) {
    static const bool on = [] { const char* e = getenv("CCST_CONV_STREAM"); return !(e && atoi(e) == 0); }();
    return on && taps == 1 && cin % 32 == 0 && cout % 64 == 0 && M >= 64 && (long long)M * cout * 4 < 0xffffffffLL;
}

// Tile code (WM WN NT as decimal digits) the dispatcher picks for this problem; bench.py names kernels with it.
extern "C" int ccst_conv2d_igemm_tile(int M, int cout, int cin, int taps, int pool) { return choose_tile(M, cout, cin, taps, pool != 0); }

// Row groups of 64 output rows that ccst_conv2d_igemm_stats_f32 writes for a problem of M rows and cout columns.
extern "C" int ccst_conv2d_igemm_stats_groups(int M, int cout, int cin, int taps) {
    if (stream_shape_ok(M, cout, cin, taps)) return 2 * stream_grid(M, cout) / (cout / 64);      // see conv1x1_stream_kernel: per (row-tile class, wave row)
    const int tile = choose_tile(M, cout, cin, taps, false);
    const int WM = (tile / 100) % 10, BM = (tile >= 1000 ? 32 : 64) * WM;      // one slab per wave row
    return ((M + BM - 1) / BM) * WM;
}

static int conv_igemm_impl(const CcstConvDesc* d, const float* x, const float* w_packed, const float* bias, float* y,
                           float* stats, void* stream, const unsigned char* relu_mask = nullptr);

extern "C" int ccst_conv2d_igemm_f32(const CcstConvDesc* d, const float* x, const float* w_packed, const float* bias,
                                     float* y, void* stream) {
    return conv_igemm_impl(d, x, w_packed, bias, y, nullptr, stream);
}

extern "C" int ccst_conv2d_igemm_stats_f32(const CcstConvDesc* d, const float* x, const float* w_packed, const float* bias,
                                           float* y, float* stats, void* stream) {
    CCST_REQUIRE(stats != nullptr, "conv_stats: null stats buffer");
    CCST_REQUIRE(d && !(d->flags & (CCST_CONV_POOL2 | CCST_CONV_RELU)), "conv_stats: statistics are of the raw conv output (no ReLU / pool)");
    return conv_igemm_impl(d, x, w_packed, bias, y, stats, stream);
}

// 1 when this problem runs on the persistent pointwise kernel (the only one with the masked accumulate form).
extern "C" int ccst_conv2d_pointwise_ok(const CcstConvDesc* d) {
    if (!d || (d->flags & (CCST_CONV_POOL2 | CCST_CONV_RELU | CCST_CONV_UPS2 | CCST_CONV_REFLECT))) return 0;
    const bool dense_out = d->ysC == 1 && d->ysW == d->cout && d->ysH == (long long)d->wo * d->ysW && d->ysN == (long long)d->ho * d->wo * d->ysW;
    const bool dense_in = d->nky == 1 && d->nkx == 1 && d->ay == 1 && d->ax == 1 && d->cy == 0 && d->cx == 0 && d->hi == d->ho && d->wi == d->wo &&
                          d->xsH == (long long)d->wi * d->xsW && d->xsN == (long long)d->hi * d->wi * d->xsW;
    return (dense_in && dense_out && stream_shape_ok(d->n * d->ho * d->wo, d->cout, d->cin, 1)) ? 1 : 0;
}

namespace {
struct BnLink {
    const float *x, *mean, *invstd, *gamma, *beta;
    float* part;
};
}  // namespace
static int conv_igemm_impl(const CcstConvDesc* d, const float* x, const float* w_packed, const float* bias, float* y, float* stats,
                           void* stream, const unsigned char* relu_mask, const BnLink* bn, const uint32_t* xmax = nullptr,
                           const uint32_t* wmax = nullptr, bool gather_half = false);

// 1 when this problem runs on the persistent pointwise kernel at all (dense or strided 1x1 input, dense output, no bias / ReLU / pool):
// the shapes ccst_conv2d_pointwise_half_f32 takes.
static bool stream_problem(const CcstConvDesc* d) {
    if (!d || (d->flags & (CCST_CONV_POOL2 | CCST_CONV_RELU | CCST_CONV_UPS2 | CCST_CONV_REFLECT))) return false;
    const bool dense_out = d->ysC == 1 && d->ysW == d->cout && d->ysH == (long long)d->wo * d->ysW && d->ysN == (long long)d->ho * d->wo * d->ysW;
    const bool pw_in = d->nky == 1 && d->nkx == 1 && d->cy == 0 && d->cx == 0 && d->ay >= 1 && d->ax >= 1 &&
                       (long long)(d->ho - 1) * d->ay < d->hi && (long long)(d->wo - 1) * d->ax < d->wi;
    return pw_in && dense_out && stream_shape_ok(d->n * d->ho * d->wo, d->cout, d->cin, 1);
}
extern "C" int ccst_conv2d_stream_ok(const CcstConvDesc* d) { return stream_problem(d) ? 1 : 0; }

// Every form of the streaming pointwise kernel on half pieces (16-bit MFMA, fp32 accuracy; see conv1x1_stream_kernel): x scaled by its
// |max| words, the weight PRE-SPLIT by ccst_pack_conv_weight_split_f32 with the same w_absmax words.  The form follows from the
// arguments, as in the fp32 entries: stats (training forward), CCST_CONV_ACCUM (+ relu_mask (+ the BatchNorm link bn_x / bn_mean /
// bn_invstd / bn_partials)), the BatchNorm + ReLU backward link (bn_gamma / bn_beta too, no ACCUM), or plain.  Requires
// ccst_conv2d_stream_ok(d) (and ccst_conv2d_pointwise_ok(d) for the masked / linked forms).
extern "C" int ccst_conv2d_pointwise_half_f32(const CcstConvDesc* d, const float* x, const uint32_t* x_absmax, const float* w_split,
                                              const uint32_t* w_absmax, float* y, float* stats, const uint8_t* relu_mask, const float* bn_x,
                                              const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                              float* bn_partials, void* stream) {
    CCST_REQUIRE(x_absmax && w_absmax, "conv_pointwise_half: the |max| words of x and w");
    CCST_REQUIRE(stream_problem(d), "conv_pointwise_half: not a problem of the streaming pointwise kernel (ccst_conv2d_stream_ok)");
    CCST_REQUIRE((relu_mask == nullptr && bn_x == nullptr) || ccst_conv2d_pointwise_ok(d), "conv_pointwise_half: the masked / linked forms need a dense input");
    CCST_REQUIRE((bn_x != nullptr) == (bn_partials != nullptr) && (bn_x == nullptr || (bn_mean && bn_invstd)) && ((bn_gamma != nullptr) == (bn_beta != nullptr)),
                 "conv_pointwise_half: incomplete BatchNorm link");
    CCST_REQUIRE(stream_bfp() == 4, "conv_pointwise_half: CCST_CONV_BF=0 turns the half-piece kernels off");
    const BnLink bn = {bn_x, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_partials};
    return conv_igemm_impl(d, x, w_split, nullptr, y, stats, stream, relu_mask, bn_x ? &bn : nullptr, x_absmax, w_absmax);
}

extern "C" int ccst_pack_conv_weight_split_f32(const float* w_oihw, const uint32_t* w_absmax, float* packed, int cout, int cin, int ntap,
                                               int transpose, int k_pad, int n_pad, void* stream) {
    CCST_REQUIRE(w_oihw && w_absmax && packed && cout > 0 && cin > 0 && ntap > 0, "pack_conv_weight_split: bad args");
    CCST_REQUIRE(k_pad % 4 == 0 && k_pad >= (transpose ? cout : cin) && n_pad >= (transpose ? cin : cout), "pack_conv_weight_split: padding");
    const long long units = (long long)ntap * (k_pad / 4) * n_pad;
    hipLaunchKernelGGL(pack_weight_split_kernel, dim3((unsigned)((units + 255) / 256 < 2048 ? (units + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, w_absmax, packed, cout, cin, ntap, transpose, k_pad, n_pad);
    return ccst_launch_status("pack_weight_split");
}

// The gather GEMM on half pieces (conv_igemm_kernel's HALFP form: 64x64 tiles, 32-channel k-steps): any taps / strides / zero padding, dense
// or strided output, CCST_CONV_ACCUM -- the backward-data launches that are not problems of the streaming pointwise or the halo kernel
// (the parity classes of a stride-2 3x3 conv, the strided 1x1 downsample branches).  x scaled by its |max| words, w_split =
// ccst_pack_conv_weight_split_f32 of all taps with the same w_absmax words.  cin % 32 == 0; no bias / ReLU / pool / reflection / upsample.
extern "C" int ccst_conv2d_igemm_half_f32(const CcstConvDesc* d, const float* x, const uint32_t* x_absmax, const float* w_split,
                                          const uint32_t* w_absmax, float* y, void* stream) {
    CCST_REQUIRE(x_absmax && w_absmax, "conv_igemm_half: the |max| words of x and w");
    CCST_REQUIRE(d && d->cin % 32 == 0 && !(d->flags & (CCST_CONV_POOL2 | CCST_CONV_RELU | CCST_CONV_UPS2 | CCST_CONV_REFLECT)),
                 "conv_igemm_half: cin a multiple of 32, no ReLU / pool / upsample / reflection");
    CCST_REQUIRE(stream_bfp() == 4, "conv_igemm_half: CCST_CONV_BF=0 turns the half-piece kernels off");
    return conv_igemm_impl(d, x, w_split, nullptr, y, nullptr, stream, nullptr, nullptr, x_absmax, w_absmax, true);
}
extern "C" int ccst_pack_conv_weights_split_batch_f32(const int64_t* jobs_device, int njobs, void* stream) {
    CCST_REQUIRE(jobs_device && njobs > 0, "pack_conv_weights_split_batch: bad args");
    hipLaunchKernelGGL(pack_weight_split_batch_kernel, dim3(32, njobs), dim3(256), 0, (hipStream_t)stream, (const long long*)jobs_device);
    return ccst_launch_status("pack_weight_split_batch");
}

extern "C" int ccst_conv2d_igemm_accum_masked_f32(const CcstConvDesc* d, const float* x, const float* w_packed, float* y,
                                                  const uint8_t* relu_mask, const float* bn_x, const float* bn_mean,
                                                  const float* bn_invstd, float* bn_partials, void* stream) {
    CCST_REQUIRE(d && relu_mask && (d->flags & CCST_CONV_ACCUM), "conv_accum_masked: needs a mask and CCST_CONV_ACCUM");
    CCST_REQUIRE(ccst_conv2d_pointwise_ok(d), "conv_accum_masked: not a pointwise problem of the streaming kernel (ccst_conv2d_pointwise_ok)");
    CCST_REQUIRE((bn_x != nullptr) == (bn_partials != nullptr) && (bn_x == nullptr || (bn_mean && bn_invstd)),
                 "conv_accum_masked: the BatchNorm link needs its input, mean, invstd and the partials buffer together");
    const BnLink bn = {bn_x, bn_mean, bn_invstd, nullptr, nullptr, bn_partials};
    return conv_igemm_impl(d, x, w_packed, nullptr, y, nullptr, stream, relu_mask, bn_x ? &bn : nullptr);
}

extern "C" int ccst_conv2d_igemm_bn_relu_bwd_f32(const CcstConvDesc* d, const float* x, const float* w_packed, float* y, const float* bn_x,
                                                 const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                                 float* bn_partials, void* stream) {
    CCST_REQUIRE(d && !(d->flags & CCST_CONV_ACCUM), "conv_bn_relu_bwd: plain (non-accumulating) form");
    CCST_REQUIRE(ccst_conv2d_pointwise_ok(d), "conv_bn_relu_bwd: not a pointwise problem of the streaming kernel (ccst_conv2d_pointwise_ok)");
    CCST_REQUIRE(bn_x && bn_mean && bn_invstd && bn_gamma && bn_beta && bn_partials, "conv_bn_relu_bwd: null BatchNorm link");
    const BnLink bn = {bn_x, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_partials};
    return conv_igemm_impl(d, x, w_packed, nullptr, y, nullptr, stream, nullptr, &bn);
}

static int conv_igemm_impl(const CcstConvDesc* d, const float* x, const float* w_packed, const float* bias, float* y, float* stats,
                           void* stream, const unsigned char* relu_mask) {
    return conv_igemm_impl(d, x, w_packed, bias, y, stats, stream, relu_mask, nullptr);
}

static int conv_igemm_impl(const CcstConvDesc* d, const float* x, const float* w_packed, const float* bias, float* y, float* stats,
                           void* stream, const unsigned char* relu_mask, const BnLink* bn, const uint32_t* xmax, const uint32_t* wmax,
                           bool gather_half) {
    CCST_REQUIRE(d && x && w_packed && y, "conv: null pointer");
    CCST_REQUIRE(d->cin > 0 && d->cin % CK_MIN == 0, "conv: cin=%d must be a positive multiple of 16", d->cin);
    CCST_REQUIRE(d->cout > 0 && d->cout_pad >= d->cout && d->cout_pad % 128 == 0, "conv: cout=%d cout_pad=%d (need multiple of 128)",
                 d->cout, d->cout_pad);
    CCST_REQUIRE(d->n > 0 && d->ho > 0 && d->wo > 0 && d->hi > 0 && d->wi > 0 && d->nky > 0 && d->nkx > 0, "conv: bad extents");
    CCST_REQUIRE((long long)d->n * d->ho * d->wo < 0x7fffffffLL, "conv: M too large");
    CCST_REQUIRE((long long)d->n * d->xsN < 0x7fffffffLL, "conv: input tensor must have < 2^31 elements (32-bit offsets)");
    const bool pool = (d->flags & CCST_CONV_POOL2) != 0;
    if (d->flags & CCST_CONV_ACCUM)
        CCST_REQUIRE(!(d->flags & (CCST_CONV_POOL2 | CCST_CONV_RELU)) && stats == nullptr, "conv: CCST_CONV_ACCUM excludes ReLU / pool / statistics");
    if (d->flags & CCST_CONV_REFLECT) CCST_REQUIRE(d->hi >= 2 && d->wi >= 2, "conv: reflection needs extent >= 2");
    ConvArgs a;
    a.x = x; a.w = w_packed; a.bias = bias; a.y = y;
    a.N = d->n; a.Ho = d->ho; a.Wo = d->wo; a.Hi = d->hi; a.Wi = d->wi; a.Cin = d->cin; a.Cout = d->cout; a.CoutPad = d->cout_pad;
    a.nky = d->nky; a.nkx = d->nkx; a.ay = d->ay; a.by = d->by; a.cy = d->cy; a.ax = d->ax; a.bx = d->bx; a.cx = d->cx;
    a.tap_base = d->tap_base; a.tap_sy = d->tap_sy; a.tap_sx = d->tap_sx;
    a.xsN = d->xsN; a.xsH = d->xsH; a.xsW = d->xsW;
    a.y_off = d->y_off; a.ysN = d->ysN; a.ysH = d->ysH; a.ysW = d->ysW; a.ysC = d->ysC;
    a.flags = d->flags & 0xffu;
    if (!pool && d->ysH == (long long)d->wo * d->ysW && d->ysN == (long long)d->ho * d->wo * d->ysW) a.flags |= CONV_DENSE_OUT;
    if (!pool && d->nky == 1 && d->nkx == 1 && d->ay == 1 && d->ax == 1 && d->cy == 0 && d->cx == 0 && d->hi == d->ho && d->wi == d->wo &&
        !(d->flags & (CCST_CONV_UPS2 | CCST_CONV_REFLECT)) && d->xsH == (long long)d->wi * d->xsW && d->xsN == (long long)d->hi * d->wi * d->xsW)
        a.flags |= CONV_DENSE_IN;
    a.M = d->n * d->ho * d->wo;
    a.invHW = 1.0f / (float)(d->ho * d->wo);
    a.invWo = 1.0f / (float)d->wo;
    a.fastdiv = a.M < (1 << 22);
    a.stats = stats;
    a.rmask = relu_mask;
    a.bn_x = bn ? bn->x : nullptr; a.bn_mean = bn ? bn->mean : nullptr; a.bn_invstd = bn ? bn->invstd : nullptr; a.bn_part = bn ? bn->part : nullptr;
    a.bn_gamma = bn ? bn->gamma : nullptr; a.bn_beta = bn ? bn->beta : nullptr;
    a.xmax = xmax; a.wmax = wmax;
    hipStream_t s = (hipStream_t)stream;
    // pointwise input: dense, or a strided 1x1 without padding (the downsample branches: pixel (oy*ay, ox*ax) is always inside)
    const bool pw_in = (a.flags & CONV_DENSE_IN) ||
                       (d->nky == 1 && d->nkx == 1 && d->cy == 0 && d->cx == 0 && d->ay >= 1 && d->ax >= 1 && !(d->flags & (CCST_CONV_UPS2 | CCST_CONV_REFLECT)) &&
                        (long long)(d->ho - 1) * d->ay < d->hi && (long long)(d->wo - 1) * d->ax < d->wi);
    if (gather_half) return launch_conv<2, 2, 1, false, 1, 32, true>(a, s);      // (ccst_conv2d_igemm_half_f32: the caller asked for this kernel)
    if (pw_in && (a.flags & CONV_DENSE_OUT) && a.ysC == 1 && a.ysW == d->cout && bias == nullptr && !pool &&
        !(a.flags & CCST_CONV_RELU) && stream_shape_ok(a.M, d->cout, d->cin, d->nky * d->nkx)) {
        a.tilesN = d->cout / 64;
        const int ntiles = ((a.M + 63) / 64) * a.tilesN;
        const int grid = stream_grid(a.M, d->cout);                            // <= 4 workgroups per CU, a multiple of the column-tile count
        const bool acc = (a.flags & CCST_CONV_ACCUM) != 0;
        if (relu_mask) CCST_REQUIRE(acc && !stats, "conv: the ReLU mask goes with CCST_CONV_ACCUM (the sum is masked)");
        if (bn) CCST_REQUIRE(!stats && ((acc && relu_mask) || (!acc && bn->gamma)), "conv: BatchNorm link without its masked form");
        // half pieces only where the caller passed the |max| words of both operands -- without them nothing bounds the operands, and
        // the fp32 MFMA runs
        const bool halfp = xmax != nullptr && wmax != nullptr;      // (ccst_conv2d_pointwise_half_f32: w_packed is the pre-split layout)
#define CCST_STREAM(...)                                                                                                     \
    do {                                                                                                                     \
        if (halfp) hipLaunchKernelGGL((conv1x1_stream_kernel<__VA_ARGS__, true>), dim3(grid), dim3(256), 0, s, a, ntiles);   \
        else hipLaunchKernelGGL((conv1x1_stream_kernel<__VA_ARGS__, false>), dim3(grid), dim3(256), 0, s, a, ntiles);        \
    } while (0)
        if (stats) CCST_STREAM(true, false, 0, false);
        else if (acc && relu_mask && bn) CCST_STREAM(false, true, 1, true);
        else if (acc && relu_mask) CCST_STREAM(false, true, 1, false);
        else if (bn) CCST_STREAM(false, false, 2, true);
        else if (acc) CCST_STREAM(false, true, 0, false);
        else CCST_STREAM(false, false, 0, false);
#undef CCST_STREAM
        return ccst_launch_status("conv1x1_stream");
    }
    CCST_REQUIRE(relu_mask == nullptr && bn == nullptr, "conv: the masked / BatchNorm-linked forms exist only where ccst_conv2d_pointwise_ok() says so");
    CCST_REQUIRE(xmax == nullptr && wmax == nullptr, "conv: the half-piece form exists only on the streaming pointwise kernel (ccst_conv2d_stream_ok)");
    const int tile = choose_tile(a.M, d->cout, d->cin, d->nky * d->nkx, pool);
    // ccst_conv2d_igemm_stats_groups() counted the streaming kernel's slabs for every 1x1 problem of its shape class
    CCST_REQUIRE(!(stats && stream_shape_ok(a.M, d->cout, d->cin, d->nky * d->nkx)),
                 "conv_stats: a 1x1 problem with padding / a non-dense output has no statistics epilogue");
    if (pool) {
        if (tile == 412) return launch_conv<4, 1, 2, true>(a, s);
        if (tile == 221) return launch_conv<2, 2, 1, true>(a, s);
        return launch_conv<2, 2, 2, true>(a, s);
    }
    if (tile == 1221) return launch_conv<2, 2, 1, false, 1, 16>(a, s);      // 64x64 (Cin not a multiple of 32)
    if (tile == 1222) return launch_conv<2, 2, 1, false, 1, 32>(a, s);      // 64x64, 32-channel k-step
    if (tile == 411) return launch_conv<4, 1, 1, false>(a, s);
    if (tile == 412) return launch_conv<4, 1, 2, false>(a, s);
    if (tile == 221) return launch_conv<2, 2, 1, false>(a, s);
    return launch_conv<2, 2, 2, false>(a, s);
}

extern "C" int ccst_pack_conv_weight_f32(const float* w_oihw, float* packed, int cout, int cin, int kh, int kw, int transpose,
                                         int k_pad, int n_pad, void* stream) {
    CCST_REQUIRE(w_oihw && packed, "pack: null pointer");
    const int kdim = transpose ? cout : cin, ndim = transpose ? cin : cout;
    CCST_REQUIRE(k_pad >= kdim && k_pad % 16 == 0, "pack: k_pad=%d must be a multiple of 16 >= %d", k_pad, kdim);
    CCST_REQUIRE(n_pad >= ndim && n_pad % 128 == 0, "pack: n_pad=%d must be a multiple of 128 >= %d", n_pad, ndim);
    const long long total = (long long)kh * kw * k_pad * n_pad;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, cout, cin, kh * kw,
                       transpose, k_pad, n_pad);
    return ccst_launch_status("pack_weight");
}

extern "C" int ccst_pack_conv_weights_batch_f32(const int64_t* jobs_device, int njobs, void* stream) {
    CCST_REQUIRE(jobs_device && njobs > 0 && njobs <= 65535, "pack_batch: bad job table");
    hipLaunchKernelGGL(pack_weight_batch_kernel, dim3(64, njobs), dim3(256), 0, (hipStream_t)stream, (const long long*)jobs_device);
    return ccst_launch_status("pack_weight_batch");
}

extern "C" int ccst_nchw_to_nhwc4_pad_f32(const float* x, float* y, int N, int C, int H, int W, int pad, int Wp, int reflect,
                                          void* stream) {
    CCST_REQUIRE(x && y && N > 0 && C >= 1 && C <= 4 && H > 0 && W > 0 && pad >= 0 && Wp >= W + 2 * pad, "nchw_to_nhwc4: bad args");
    if (reflect) CCST_REQUIRE(H > pad && W > pad, "nchw_to_nhwc4: reflection pad %d needs extent > pad", pad);
    const long long total = (long long)N * (H + 2 * pad) * Wp;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nchw_to_nhwc4_pad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (f32x4*)y, N, C, H, W, pad, Wp,
                       reflect);
    return ccst_launch_status("nchw_to_nhwc4_pad");
}
