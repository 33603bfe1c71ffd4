// Convolution backward-weight on the fp32-input MFMA (gfx950).
//
//   dW[tap][ci][co] = sum_m X[pixel(m,tap)][ci] * dY[m][co]          (zero padding)
//   GEMM per tap: rows = ci, cols = co, reduction over the N*Ho*Wo output pixels m.
//   Workgroup = 4 waves (2x2), tile BI x BJ = (64*MI) x (64*NJ); the pixel range is split over
//   `splits` workgroups per tile; each writes an fp32 partial slab ws[split][tap][ci][co]; a second
//   kernel sums the slabs in fixed order (bitwise reproducible, no float atomics) and transposes into
//   the parameter's OIHW layout (optionally accumulating into the existing gradient).
//   Per step PK pixels are staged: X rows gathered with the forward conv's affine index map, dY rows
//   dense; LDS images are [pixel][channel] so the MFMA operands (lane = channel, k = pixel) are
//   conflict-free ds_read_b32.
#include "common.h"
#include <type_traits>

namespace {

struct BwdWArgs {
    const float* x;
    const float* dy;
    float* ws;
    int N, Ho, Wo, Hi, Wi, Cin, Cout;
    int nky, nkx, ay, by, cy, ax, bx, cx;
    long long xsN;
    int xsH, xsW;
    int M, splits, tilesI, tilesJ;
    float invHW, invWo;     // reciprocals for the division-free pixel decode (valid while M < 2^22)
    int fastdiv;
    const unsigned* xmax;   // (half-piece kernel) |max| words of x and of dy
    const unsigned* dmax;
    unsigned xbytes, dbytes; // (half-piece kernel) extents of x and dy in bytes: its buffer loads return zero beyond them
};

// floor(m / d) for 0 <= m < 2^22 via one float multiply + correction (an integer division costs ~40
// VALU instructions and this runs 4x per thread per k-step).
__device__ __forceinline__ int fdiv(int m, int d, float inv) {
    int q = (int)((float)m * inv);
    const int r = m - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// PW: pointwise (1x1, stride 1, no padding, dense NHWC input) -- the input pixel of row m IS pixel m, so the
// per-step (n, oy, ox) decode and bounds tests drop out (about half of the ResNet weight-gradient time).
// TP2: two taps share one 64-row ci tile (rows 0..31 = tap 2g, rows 32..63 = tap 2g+1) when Cin <= 32 -- the
// ResNet stem's virtual-pixel form (7 taps x 32 channels), which otherwise leaves half of every MFMA tile empty.
template <int MI, int NJ, int PK, bool PW, bool TP2 = false>
__global__ __launch_bounds__(256) void conv_bwd_weight_kernel(const BwdWArgs p) {
    static_assert(!TP2 || (MI == 1 && !PW), "tap packing is for the 64x64 tile of a multi-tap problem");
    constexpr int BI = 64 * MI, BJ = 64 * NJ;
    constexpr int LDI = BI + 4, LDJ = BJ + 4;            // +4 floats: rows land on different banks
    constexpr int XU = PK * BI / 4, DU = PK * BJ / 4;    // float4 units per step
    constexpr int XR = XU / 256, DR = DU / 256;
    static_assert(XU % 256 == 0 && DU % 256 == 0, "tile/thread mismatch");

    __shared__ __attribute__((aligned(16))) float Xs[2][PK * LDI];
    __shared__ __attribute__((aligned(16))) float Ds[2][PK * LDJ];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // Workgroup order (round 4): the workgroups of ONE pixel range -- its taps and its (ci, co) tiles, which read the same rows of x and
    // dy -- are consecutive, and an XCD owns a contiguous run of the list (hardware XCD = blockIdx % 8), so they run side by side
    // behind one L2.  With the split fastest (as before) the nine taps of a pixel range were thousands of workgroups apart and every
    // tap's pass over x missed L2: 819 MB fetched for 103 MB of operands on the 3x3 64 -> 64 layers.
    int b = ccst_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tj = b % p.tilesJ;
    b /= p.tilesJ;
    const int ti = b % p.tilesI;
    b /= p.tilesI;
    const int ngroups = (int)gridDim.x / (p.splits * p.tilesI * p.tilesJ);      // tap groups
    const int tapg = b % ngroups;
    const int split = b / ngroups;
    const int tap = TP2 ? 2 * tapg : tapg;
    const int ky = tap / p.nkx, kx = tap - ky * p.nkx;
    const int ci0 = ti * BI, co0 = tj * BJ;
    const int ntap = p.nky * p.nkx;

    // pixel range of this split, in steps of PK
    const int steps_total = (p.M + PK - 1) / PK;
    const int sps = (steps_total + p.splits - 1) / p.splits;
    const int s0 = split * sps;
    const int s1 = min(steps_total, s0 + sps);

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int c = 0; c < NJ; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    f32x4 rx[XR], rd[DR];
    const int HW = p.Ho * p.Wo;
    // (n, oy, ox) of the first pixel of the step being loaded: carried from step to step in scalar registers (a step advances by PK
    // pixels = PK / Wo rows + PK % Wo columns, one wrap each) instead of two divisions per thread and unit -- vector instructions are
    // paid on top of the MFMA time, and this loop had ~170 of them (16 quarter-rate v_mul_lo among them) per 32 MFMAs.
    int un = 0, uoy = 0, uox = 0;                       // valid when !PW
    const int stepRows = PK / p.Wo, stepCols = PK - (PK / p.Wo) * p.Wo;
    // t / d == (t * magic) >> 16 with magic = ceil(2^16 / d) while t * d < 2^16; here t < d + 32, so d <= 224 -- beyond that PK < d and
    // the quotient is 0 or 1 (a comparison)
    const unsigned magicW = (65536u + (unsigned)p.Wo - 1u) / (unsigned)p.Wo, magicH = (65536u + (unsigned)p.Ho - 1u) / (unsigned)p.Ho;
    const bool wideW = p.Wo > 224, wideH = p.Ho > 224;
    if (!PW) {
        const int m0 = min(s0 * PK, p.M - 1);
        un = m0 / HW;
        const int rem = m0 - un * HW;
        uoy = rem / p.Wo;
        uox = rem - uoy * p.Wo;
        un = __builtin_amdgcn_readfirstlane(un);
        uoy = __builtin_amdgcn_readfirstlane(uoy);
        uox = __builtin_amdgcn_readfirstlane(uox);
    }
    auto advance_pixels = [&]() {          // to the next step's first pixel (uniform)
        uox += stepCols;
        uoy += stepRows;
        if (uox >= p.Wo) {
            uox -= p.Wo;
            ++uoy;
        }
        while (uoy >= p.Ho) {               // (more than once only for maps narrower than a step)
            uoy -= p.Ho;
            ++un;
        }
    };

    auto load_step = [&](int st) {
        const int m0 = st * PK;
#pragma unroll
        for (int u = 0; u < XR; ++u) {
            const int unit = tid + 256 * u;
            const int px = unit / (BI / 4), cp = unit - px * (BI / 4);
            const int m = m0 + px;
            const int mc = min(m, p.M - 1);
            int ci = ci0 + cp * 4, kyu = ky, kxu = kx;
            bool tap_ok = true;
            if (TP2) {          // step-invariant per thread (cp is): the compiler hoists it out of the k-loop
                const int tapu = tap + (cp >> 3);
                ci = (cp & 7) * 4;
                kyu = tapu / p.nkx;
                kxu = tapu - kyu * p.nkx;
                tap_ok = tapu < ntap;
            }
            const int cic = min(ci, p.Cin - 4);
            bool ok;
            f32x4 v;
            if (PW) {
                ok = (m < p.M) & (ci < p.Cin);
                v = *reinterpret_cast<const f32x4*>(p.x + (long long)mc * p.xsW + cic);
            } else {
                // pixel m0 + px from the step's uniform (n, oy, ox): px < PK <= 32 columns further on, at most PK / Wo + 1 row wraps
                // and one image wrap (rows beyond M are masked by ok and may decode to anything inside the tensor)
                const unsigned t = (unsigned)(uox + px);
                const unsigned q = wideW ? (t >= (unsigned)p.Wo ? 1u : 0u) : (__umul24(t, magicW) >> 16);
                const int ox = (int)(t - __umul24(q, (unsigned)p.Wo));
                const unsigned ty = (unsigned)uoy + q;
                const unsigned r = wideH ? (ty >= (unsigned)p.Ho ? 1u : 0u) : (__umul24(ty, magicH) >> 16);
                const int oy = (int)(ty - __umul24(r, (unsigned)p.Ho));
                const int n = min(un + (int)r, p.N - 1);
                int iy = __mul24(oy, p.ay) + kyu * p.by + p.cy, ix = __mul24(ox, p.ax) + kxu * p.bx + p.cx;
                ok = (m < p.M) & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi) & (ci < p.Cin) & tap_ok;
                iy = min(max(iy, 0), p.Hi - 1);
                ix = min(max(ix, 0), p.Wi - 1);
                v = *reinterpret_cast<const f32x4*>(p.x + (long long)n * p.xsN + (unsigned)(__mul24(iy, p.xsH) + __mul24(ix, p.xsW) + cic));
            }
            if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            rx[u] = v;
        }
#pragma unroll
        for (int u = 0; u < DR; ++u) {
            const int unit = tid + 256 * u;
            const int px = unit / (BJ / 4), cp = unit - px * (BJ / 4);
            const int mc = min(m0 + px, p.M - 1);
            const int co = min(co0 + cp * 4, p.Cout - 4);
            rd[u] = *reinterpret_cast<const f32x4*>(p.dy + (long long)mc * p.Cout + co);
        }
        if (!PW) advance_pixels();
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int u = 0; u < XR; ++u) {
            const int unit = tid + 256 * u;
            const int px = unit / (BI / 4), cp = unit - px * (BI / 4);
            *reinterpret_cast<f32x4*>(&Xs[buf][px * LDI + cp * 4]) = rx[u];
        }
#pragma unroll
        for (int u = 0; u < DR; ++u) {
            const int unit = tid + 256 * u;
            const int px = unit / (BJ / 4), cp = unit - px * (BJ / 4);
            *reinterpret_cast<f32x4*>(&Ds[buf][px * LDJ + cp * 4]) = rd[u];
        }
    };
    constexpr int NG = PK / 2;            // MFMA groups per step (one pixel pair each)
    constexpr int NH = NG / 2;            // fragments are read in two halves
    float av[2][NH][MI], bv[2][NH][NJ];
    auto read_frags = [&](int buf, int h) {
        const float* xr = &Xs[buf][lh * LDI + wi * (32 * MI) + li];
        const float* dr = &Ds[buf][lh * LDJ + wj * (32 * NJ) + li];
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            const int k2 = h * NH + k;
#pragma unroll
            for (int a = 0; a < MI; ++a) av[h][k][a] = xr[2 * k2 * LDI + a * 32];
#pragma unroll
            for (int c = 0; c < NJ; ++c) bv[h][k][c] = dr[2 * k2 * LDJ + c * 32];
        }
    };
    auto mfma_group = [&](int g) {
        const int h = g / NH, k = g % NH;
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int c = 0; c < NJ; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[h][k][a], bv[h][k][c], acc[a][c], 0, 0, 0);
    };
    // Same step structure as conv_igemm.hip: barrier -> ds_read -> MFMA at the step boundary; the registers
    // fetched during the previous step go to the idle LDS buffers in front of the last two MFMA groups and the
    // fetch for step st+2 (with its index arithmetic) is issued in front of the last group.
    auto step = [&](int buf, int st_load, auto do_store, auto do_load) {
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
                load_step(st_load);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g == NH - 1) {
                __builtin_amdgcn_sched_barrier(0);
                read_frags(buf, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma_group(g);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;

    if (s0 < s1) {
        const int T = s1 - s0;
        load_step(s0);
        store_step(0);
        if (T > 1) load_step(s0 + 1);
        __syncthreads();
        for (int t = 0; t < T - 2; ++t) {
            step(t & 1, s0 + t + 2, Yes{}, Yes{});
            __syncthreads();
        }
        if (T > 1) {
            step((T - 2) & 1, 0, Yes{}, No{});
            __syncthreads();
        }
        step((T - 1) & 1, 0, No{}, No{});
    }

    // partial slab store: ws[((split*ntap + tap)*Cin + ci)*Cout + co]
    float* wsb = p.ws + ((long long)split * ntap + tap) * p.Cin * p.Cout;
    if (TP2) {      // this wave's 32 rows are the ci of tap (2g + wi)
        if (tap + wi >= ntap) return;
        wsb += (long long)wi * p.Cin * p.Cout;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int co = co0 + wj * 32 + li;
            if (ci < p.Cin && co < p.Cout) wsb[(long long)ci * p.Cout + co] = acc[0][0][r];
        }
        return;
    }
    if (ci0 + BI <= p.Cin && co0 + BJ <= p.Cout) {
        // full tile: buffer stores with the row in the scalar offset (as conv_igemm.hip's dense epilogue)
        float* const tile = wsb + (long long)(ci0 + wi * (32 * MI)) * p.Cout + co0 + wj * (32 * NJ);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 0x7fffffff, 0x00020000);
        const unsigned lane_off = (unsigned)(4 * lh * p.Cout + li) * 4u;
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = (a * 32 + (r & 3) + 8 * (r >> 2)) * p.Cout * 4;
#pragma unroll
                for (int c = 0; c < NJ; ++c)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[a][c][r]), rsrc, lane_off + c * 128, srow, 0);
            }
        return;
    }
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wi * (32 * MI) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (ci < p.Cin) {
#pragma unroll
                for (int c = 0; c < NJ; ++c) {
                    const int co = co0 + wj * (32 * NJ) + c * 32 + li;
                    if (co < p.Cout) wsb[(long long)ci * p.Cout + co] = acc[a][c][r];
                }
            }
        }
}


// ------------------------------------------------------------------------------------------------------------------------
// The same GEMM on the 16-bit MFMA (round 5): every fp32 product as three products of IEEE-half pieces (v = hi + lo, 22 significant
// bits, fp32 accumulation: x_lo dy_hi + x_hi dy_lo + x_hi dy_hi), v_mfma_f32_32x32x16_f16 -- 12 MFMAs of K = 16 per 64x64 wave tile
// and 16 pixels instead of 32 of K = 2, 5.3x the fp32 MFMA's rate.  The reduction runs over PIXELS, so an MFMA operand is eight
// consecutive pixels of one channel, while memory (NHWC) has the channels of one pixel side by side: the LDS images stay
// [pixel][channel] (what the loader's 16-byte loads deliver: four channels of a pixel -> two 8-byte stores of half pieces) and the
// operands are fetched with ds_read_b64_tr_b16, the transposing LDS read of gfx950 (a 16-lane group reads four pixel rows x 16 channels
// and each lane receives its channel's four pixels).  Rows are padded by 64 bytes: the four pixel rows a 32-lane half reads then
// fall on the four 64-byte quarters of the 256-byte bank row (conflict-free; 256-byte rows would be four-way).
// Range: x and dy are scaled by powers of two from their tensors' |max| words (x: left by the BatchNorm apply that produced it; dy:
// by the BatchNorm-backward apply, ccst_bn_train_bwd_*'s dx_absmax) to max < 2^14 as they are split; the slab is scaled back.  An
// element 2^-k below its tensor's maximum keeps min(22, 38 - k) bits, an absolute error of 2^-38 of that maximum at worst.
// Loads run two steps ahead in two register sets (a step of 16 pixels is ~0.3 us: one step would not cover an HBM round trip),
// unconditionally (past the range: clamped addresses, never stored), so that the compiler's vmcnt counts stay exact.
// ------------------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2w __attribute__((ext_vector_type(2)));
typedef float f32x2w __attribute__((ext_vector_type(2)));
typedef short s16x4w __attribute__((ext_vector_type(4)));
typedef short s16x8w __attribute__((ext_vector_type(8)));
typedef ccst_u32x2 u32x2w;

// eight consecutive pixels (k) of this lane's channel: two transposing reads four pixel rows apart.  EXEC must be all ones.
__device__ __forceinline__ f16x8w lds_tr8(const unsigned char* p, int row_bytes) {
    typedef __attribute__((address_space(3))) s16x4w* lp;
    const s16x4w a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)p);
    const s16x4w b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + 4 * row_bytes));
    return __builtin_bit_cast(f16x8w, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int MI, int NJ, int PK, bool PW, bool TP2 = false>
__global__ __launch_bounds__(256, 2) void conv_bwd_weight_split_kernel(const BwdWArgs p) {
    static_assert(!TP2 || (MI == 1 && !PW), "tap packing is for the 64x64 tile of a multi-tap problem");
    static_assert(PK % 16 == 0, "a step is whole K = 16 MFMAs");
    constexpr int BI = 64 * MI, BJ = 64 * NJ;
    constexpr int SI = 2 * BI + 64, SJ = 2 * BJ + 64;     // bytes per pixel row of a piece image (see the header comment)
    constexpr int XIMG = PK * SI, DIMG = PK * SJ;         // bytes per piece image
    constexpr int STAGE = 2 * XIMG + 2 * DIMG;            // x hi | x lo | dy hi | dy lo
    constexpr int XU = PK * BI / 4, DU = PK * BJ / 4;     // float4 units per step
    constexpr int XR = XU / 256, DR = DU / 256;
    static_assert(XU % 256 == 0 && DU % 256 == 0, "tile/thread mismatch");
    constexpr int NKS = PK / 16;

    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const unsigned xword = ccst_absmax_load(p.xmax), dword = ccst_absmax_load(p.dmax);

    // workgroup order: as conv_bwd_weight_kernel (the workgroups of one pixel range side by side behind one L2)
    int b = ccst_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tj = b % p.tilesJ;
    b /= p.tilesJ;
    const int ti = b % p.tilesI;
    b /= p.tilesI;
    const int ngroups = (int)gridDim.x / (p.splits * p.tilesI * p.tilesJ);
    const int tapg = b % ngroups;
    const int split = b / ngroups;
    const int tap = TP2 ? 2 * tapg : tapg;
    const int ky = tap / p.nkx, kx = tap - ky * p.nkx;
    const int ci0 = ti * BI, co0 = tj * BJ;
    const int ntap = p.nky * p.nkx;

    const int steps_total = (p.M + PK - 1) / PK;
    const int sps = (steps_total + p.splits - 1) / p.splits;
    const int s0 = split * sps;
    const int s1 = min(steps_total, s0 + sps);

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int c = 0; c < NJ; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    f32x4 rx[2][XR], rd[2][DR];
    // Loads are buffer loads: whatever is not data -- pixels past M, channels past Cin / Cout, taps that fall into the zero padding --
    // gets an offset beyond the resource's extent and comes back as ZERO from the memory pipeline: no select on loaded values (which
    // would wait for them), no clamping, and for the dense operands no address arithmetic beyond one add per unit and step.
    constexpr unsigned OOB = 0xfffffff0u;                // (>= any extent: the launcher keeps the tensors below 2^31 bytes)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dbytes, 0x00020000);
    unsigned xoffs[XR], xinc[XR], doffs[DR], dinc[DR];   // byte offsets of the NEXT step to load, and their advance per step
    int xtapy[XR], xtapx[XR], xch[XR];                   // (!PW) a unit's tap offsets (by*ky + cy, bx*kx + cx) and channel offset, or -1
#pragma unroll
    for (int u = 0; u < XR; ++u) {
        const int unit = tid + 256 * u;
        const int px = unit / (BI / 4), cp = unit - px * (BI / 4);
        int ci = ci0 + cp * 4, kyu = ky, kxu = kx;
        bool tap_ok = true;
        if (TP2) {
            const int tapu = tap + (cp >> 3);
            ci = (cp & 7) * 4;
            kyu = tapu / p.nkx;
            kxu = tapu - kyu * p.nkx;
            tap_ok = tapu < ntap;
        }
        const bool chan_ok = (ci < p.Cin) & tap_ok;
        xoffs[u] = chan_ok ? (unsigned)((s0 * PK + px) * p.xsW + ci) * 4u : OOB;       // (PW: xsW = Cin, pixel m IS input pixel m)
        xinc[u] = chan_ok ? (unsigned)(PK * p.xsW) * 4u : 0u;
        xtapy[u] = kyu * p.by + p.cy;
        xtapx[u] = kxu * p.bx + p.cx;
        xch[u] = chan_ok ? ci : -1;
    }
#pragma unroll
    for (int u = 0; u < DR; ++u) {
        const int unit = tid + 256 * u;
        const int px = unit / (BJ / 4), cp = unit - px * (BJ / 4);
        const int co = co0 + cp * 4;
        doffs[u] = co < p.Cout ? (unsigned)((s0 * PK + px) * p.Cout + co) * 4u : OOB;
        dinc[u] = co < p.Cout ? (unsigned)(PK * p.Cout) * 4u : 0u;
    }
    int un = 0, uoy = 0, uox = 0, um = s0 * PK;          // (!PW) (n, oy, ox) and index of the first pixel of the next step to load, wave-uniform
    const int HW = p.Ho * p.Wo;
    const int stepRows = PK / p.Wo, stepCols = PK - (PK / p.Wo) * p.Wo;
    const unsigned magicW = (65536u + (unsigned)p.Wo - 1u) / (unsigned)p.Wo, magicH = (65536u + (unsigned)p.Ho - 1u) / (unsigned)p.Ho;
    const bool wideW = p.Wo > 224, wideH = p.Ho > 224;
    if (!PW) {
        const int m0 = min(s0 * PK, p.M - 1);
        un = m0 / HW;
        const int rem = m0 - un * HW;
        uoy = rem / p.Wo;
        uox = rem - uoy * p.Wo;
        un = __builtin_amdgcn_readfirstlane(un);
        uoy = __builtin_amdgcn_readfirstlane(uoy);
        uox = __builtin_amdgcn_readfirstlane(uox);
    }
    auto advance_pixels = [&]() {
        um += PK;
        uox += stepCols;
        uoy += stepRows;
        if (uox >= p.Wo) {
            uox -= p.Wo;
            ++uoy;
        }
        while (uoy >= p.Ho) {
            uoy -= p.Ho;
            ++un;
        }
    };

    // loads of the next step into a register set (steps past this workgroup's range are loaded and never used; past the tensor: zeros)
    auto load_step = [&](f32x4 (&rxs)[XR], f32x4 (&rds)[DR]) {
#pragma unroll
        for (int u = 0; u < XR; ++u) {
            unsigned off;
            if (PW) {
                off = xoffs[u];
                xoffs[u] += xinc[u];
            } else {
                // pixel um + px from the step's uniform (n, oy, ox): px < PK columns further on, at most PK / Wo + 1 row wraps and one
                // image wrap
                const int unit = tid + 256 * u;
                const int px = unit / (BI / 4);
                const unsigned t = (unsigned)(uox + px);
                const unsigned q = wideW ? (t >= (unsigned)p.Wo ? 1u : 0u) : (__umul24(t, magicW) >> 16);
                const int ox = (int)(t - __umul24(q, (unsigned)p.Wo));
                const unsigned ty = (unsigned)uoy + q;
                const unsigned r = wideH ? (ty >= (unsigned)p.Ho ? 1u : 0u) : (__umul24(ty, magicH) >> 16);
                const int oy = (int)(ty - __umul24(r, (unsigned)p.Ho));
                const int n = un + (int)r;
                const int iy = __mul24(oy, p.ay) + xtapy[u], ix = __mul24(ox, p.ax) + xtapx[u];
                const bool ok = (um + px < p.M) & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi) & (xch[u] >= 0);
                off = ok ? ((unsigned)n * (unsigned)p.xsN + (unsigned)(__mul24(iy, p.xsH) + __mul24(ix, p.xsW) + xch[u])) * 4u : OOB;
            }
            rxs[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)off, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < DR; ++u) {
            rds[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, (int)doffs[u], 0, 0));
            doffs[u] += dinc[u];
        }
        if (!PW) advance_pixels();
    };

    const int kxs = ccst_scale_exp(ccst_absmax_reduce(xword), CCST_SPLIT_X_TARGET);
    const int kds = ccst_scale_exp(ccst_absmax_reduce(dword), CCST_SPLIT_X_TARGET);
    const float xsc = __uint_as_float((unsigned)(127 + kxs) << 23), dsc = __uint_as_float((unsigned)(127 + kds) << 23);

    auto store_step = [&](int buf, const f32x4 (&rxs)[XR], const f32x4 (&rds)[DR]) {
        unsigned char* const base = lds + buf * STAGE;
#pragma unroll
        for (int u = 0; u < XR; ++u) {
            const int unit = tid + 256 * u;
            const int px = unit / (BI / 4), cp = unit - px * (BI / 4);
            u32x2w hi, lo;
            ccst_split4_half(rxs[u], xsc, hi, lo);
            *reinterpret_cast<u32x2w*>(base + px * SI + cp * 8) = hi;
            *reinterpret_cast<u32x2w*>(base + XIMG + px * SI + cp * 8) = lo;
        }
#pragma unroll
        for (int u = 0; u < DR; ++u) {
            const int unit = tid + 256 * u;
            const int px = unit / (BJ / 4), cp = unit - px * (BJ / 4);
            u32x2w hi, lo;
            ccst_split4_half(rds[u], dsc, hi, lo);
            *reinterpret_cast<u32x2w*>(base + 2 * XIMG + px * SJ + cp * 8) = hi;
            *reinterpret_cast<u32x2w*>(base + 2 * XIMG + DIMG + px * SJ + cp * 8) = lo;
        }
    };

    // this lane's share of a transposing read: pixel row 8 lh + (lane & 15) / 4 of the 16-pixel k-step, channels 16 ((lane >> 4) & 1) +
    // 4 (lane & 3) .. + 3 of the wave's 32-channel block; it receives channel (lane & 31), pixels 8 lh .. 8 lh + 3 (+ 4 by the second read)
    const int trow = 8 * lh + ((lane & 15) >> 2), tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int xoff = trow * SI + 2 * (wi * (32 * MI) + tcol), doff = trow * SJ + 2 * (wj * (32 * NJ) + tcol);
    struct Frags {
        f16x8w a[2][MI], b[2][NJ];          // [piece][tile]
    };
    auto read_frags = [&](int buf, int ks, Frags& f) {
        const unsigned char* const base = lds + buf * STAGE;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
            for (int a = 0; a < MI; ++a) f.a[pc][a] = lds_tr8(base + pc * XIMG + ks * 16 * SI + xoff + a * 64, SI);
#pragma unroll
            for (int c = 0; c < NJ; ++c) f.b[pc][c] = lds_tr8(base + 2 * XIMG + pc * DIMG + ks * 16 * SJ + doff + c * 64, SJ);
        }
    };
    auto mfmas = [&](const Frags& f) {
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int c = 0; c < NJ; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[1][a], f.b[0][c], acc[a][c], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int c = 0; c < NJ; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[0][a], f.b[1][c], acc[a][c], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int c = 0; c < NJ; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[0][a], f.b[0][c], acc[a][c], 0, 0, 0);
    };
    // one step: the MFMAs of LDS stage `buf` (step t); registers of step t + 1 -> the other stage; loads of step t + 3 into the
    // register set just freed (the loader's own state says which step is next)
    auto step = [&](int buf, f32x4 (&rxs)[XR], f32x4 (&rds)[DR]) {
        Frags f[2];
        read_frags(buf, 0, f[0]);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + 1 < NKS) read_frags(buf, ks + 1, f[(ks + 1) & 1]);
            if (ks == NKS - 1) {
                store_step(buf ^ 1, rxs, rds);
                load_step(rxs, rds);
            }
            mfmas(f[ks & 1]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    if (s0 < s1) {
        const int T = s1 - s0;
        load_step(rx[0], rd[0]);
        load_step(rx[1], rd[1]);
        store_step(0, rx[0], rd[0]);
        load_step(rx[0], rd[0]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int t = 0;
        for (; t + 2 <= T; t += 2) {
            step(0, rx[1], rd[1]);
            step(1, rx[0], rd[0]);
        }
        if (t < T) step(0, rx[1], rd[1]);
    }

    // scale back (two exact multiplications: either power of two is a normal float, their product need not be)
    const float xin = __uint_as_float((unsigned)(127 - kxs) << 23), din = __uint_as_float((unsigned)(127 - kds) << 23);
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int c = 0; c < NJ; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = acc[a][c][r] * xin * din;

    // partial slab store: ws[((split*ntap + tap)*Cin + ci)*Cout + co]
    float* wsb = p.ws + ((long long)split * ntap + tap) * p.Cin * p.Cout;
    if (TP2) {
        if (tap + wi >= ntap) return;
        wsb += (long long)wi * p.Cin * p.Cout;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int co = co0 + wj * 32 + li;
            if (ci < p.Cin && co < p.Cout) wsb[(long long)ci * p.Cout + co] = acc[0][0][r];
        }
        return;
    }
    if (ci0 + BI <= p.Cin && co0 + BJ <= p.Cout) {
        float* const tile = wsb + (long long)(ci0 + wi * (32 * MI)) * p.Cout + co0 + wj * (32 * NJ);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 0x7fffffff, 0x00020000);
        const unsigned lane_off = (unsigned)(4 * lh * p.Cout + li) * 4u;
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = (a * 32 + (r & 3) + 8 * (r >> 2)) * p.Cout * 4;
#pragma unroll
                for (int c = 0; c < NJ; ++c)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[a][c][r]), rsrc, lane_off + c * 128, srow, 0);
            }
        return;
    }
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wi * (32 * MI) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (ci < p.Cin) {
#pragma unroll
                for (int c = 0; c < NJ; ++c) {
                    const int co = co0 + wj * (32 * NJ) + c * 32 + li;
                    if (co < p.Cout) wsb[(long long)ci * p.Cout + co] = acc[a][c][r];
                }
            }
        }
}

// dw[co][ci][tap] (+)= sum_s ws[s][tap][ci][co].  The slabs are the bulk of the traffic (splits x the weight tensor: 17-75 MB per
// ResNet50 layer at B=64) and this kernel is bound by reading them: a thread owns four consecutive co (one 16-byte load per slab) of
// one ci and sums the slabs l, l + NL, ... of its split-lane l with eight loads in flight; a workgroup = 32 co x (32 / NL) ci x NL
// lanes, the lanes folded in fixed order through LDS, the tile transposed so that the stores run along ci.  NL is chosen by the
// launcher so that there are >= 1024 workgroups where the tensor allows it (a 64x256 1x1 layer: 16 K elements, 256 slabs: NL = 32).
// Fixed summation order for a given (problem, NL) => bitwise reproducible.
template <int NL>
__global__ __launch_bounds__(256) void bwd_weight_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int ntap, int Cin,
                                                                int Cout, int splits, int accumulate) {
    constexpr int TCI = 32 / NL;                     // ci rows per workgroup
    __shared__ float part[NL][TCI][36];
    const int tap = blockIdx.z;
    const int ci0 = blockIdx.y * TCI, co0 = blockIdx.x * 32;
    const int q = threadIdx.x & 7, ty = (threadIdx.x >> 3) % TCI, sl = (threadIdx.x >> 3) / TCI;
    {
        const int ci = ci0 + ty, co = co0 + 4 * q;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (ci < Cin && co < Cout) {
            const long long stride = (long long)ntap * Cin * Cout;
            const float* p = ws + ((long long)tap * Cin + ci) * Cout + co;
            int k = sl;
            for (; k + 7 * NL < splits; k += 8 * NL) {
                f32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(p + (long long)(k + j * NL) * stride);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[j];
            }
            for (; k < splits; k += NL) s += *reinterpret_cast<const f32x4*>(p + (long long)k * stride);
        }
        *reinterpret_cast<f32x4*>(&part[sl][ty][4 * q]) = s;
    }
    __syncthreads();
    if (threadIdx.x < TCI * 32) {
        const int cil = threadIdx.x % TCI, col = threadIdx.x / TCI;
        const int ci = ci0 + cil, co = co0 + col;
        if (ci < Cin && co < Cout) {
            float s = part[0][cil][col];
#pragma unroll
            for (int l = 1; l < NL; ++l) s += part[l][cil][col];
            const long long o = ((long long)co * Cin + ci) * ntap + tap;
            dw[o] = accumulate ? dw[o] + s : s;
        }
    }
}

template <int MI, int NJ, int PK, bool PW, bool TP2 = false>
int launch(BwdWArgs& a, hipStream_t s) {
    a.tilesI = (a.Cin + 64 * MI - 1) / (64 * MI);
    a.tilesJ = (a.Cout + 64 * NJ - 1) / (64 * NJ);
    const int tapgroups = TP2 ? (a.nky * a.nkx + 1) / 2 : a.nky * a.nkx;
    const long long grid = (long long)tapgroups * a.tilesI * a.tilesJ * a.splits;
    if (grid <= 0 || grid > 0x7fffffffLL) {
        ccst_set_error("bwd_weight: bad grid");
        return CCST_EINVAL;
    }
    hipLaunchKernelGGL((conv_bwd_weight_kernel<MI, NJ, PK, PW, TP2>), dim3((unsigned)grid), dim3(256), 0, s, a);
    return ccst_launch_status("conv_bwd_weight");
}

template <int MI, int NJ, int PK, bool PW, bool TP2 = false>
int launch_split(BwdWArgs& a, hipStream_t s) {
    a.tilesI = (a.Cin + 64 * MI - 1) / (64 * MI);
    a.tilesJ = (a.Cout + 64 * NJ - 1) / (64 * NJ);
    const int tapgroups = TP2 ? (a.nky * a.nkx + 1) / 2 : a.nky * a.nkx;
    const long long grid = (long long)tapgroups * a.tilesI * a.tilesJ * a.splits;
    if (grid <= 0 || grid > 0x7fffffffLL) {
        ccst_set_error("bwd_weight_split: bad grid");
        return CCST_EINVAL;
    }
    hipLaunchKernelGGL((conv_bwd_weight_split_kernel<MI, NJ, PK, PW, TP2>), dim3((unsigned)grid), dim3(256), 0, s, a);
    return ccst_launch_status("conv_bwd_weight_split");
}

}  // namespace

// Tile (ci x co) per problem: 128x128 when both sides have >= 128 channels, else 64x64.  (64x128 / 128x64 tiles
// for ResNet layer1's 64 <-> 256 layers measured SLOWER, 55 vs 67 TF: half the tiles means twice the splits for
// the same number of workgroups, i.e. 12-step loops and twice the slab traffic.)
static void pick_tile(int cin, int cout, int* mi, int* nj) {
    *mi = *nj = (cin >= 128 && cout >= 128) ? 2 : 1;
}

// Suggested split count (also the number of slabs the workspace must hold).
extern "C" int ccst_conv2d_bwd_weight_splits(int M, int cin, int cout, int ntap) {
    int mi, nj;
    pick_tile(cin, cout, &mi, &nj);
    const int tapgroups = (mi == 1 && ntap > 1 && cin <= 32) ? (ntap + 1) / 2 : ntap;      // tap packing, see the kernel
    const long long tiles = (long long)tapgroups * ((cin + 64 * mi - 1) / (64 * mi)) * ((cout + 64 * nj - 1) / (64 * nj));
    long long s = (1024 + tiles - 1) / tiles;
    const int pk = (mi == 2 && nj == 2) ? 16 : 32;
    const long long smax = (M / pk) / 8 > 0 ? (M / pk) / 8 : 1;     // >= 8 steps per workgroup
    if (s > smax) s = smax;
    if (s > 512) s = 512;
    if (s < 1) s = 1;
    return (int)s;
}

// the half-piece kernel's tiles: 128 channels on a side that has them, else 64 (its loop is short enough that the 64 <-> 256 layers
// gain from the rectangular tiles the fp32 kernel measured slower with)
static void pick_tile_split(int cin, int cout, int* mi, int* nj) {
    *mi = cin >= 128 ? 2 : 1;
    *nj = cout >= 128 ? 2 : 1;
}
constexpr int SPLIT_PK = 16;      // pixels per step (32 -- two K = 16 MFMA steps per barrier, 80 KB of LDS -- measured 10-25 % slower on the pointwise shapes)

extern "C" int ccst_conv2d_bwd_weight_split_splits(int M, int cin, int cout, int ntap) {
    int mi, nj;
    pick_tile_split(cin, cout, &mi, &nj);
    const int tapgroups = (mi == 1 && ntap > 1 && cin <= 32) ? (ntap + 1) / 2 : ntap;
    const long long tiles = (long long)tapgroups * ((cin + 64 * mi - 1) / (64 * mi)) * ((cout + 64 * nj - 1) / (64 * nj));
    // two workgroups per CU, ONE round: measured over the 22 ResNet50 weight-gradient shapes at 256 / 512 / 1024 / 2048 workgroups,
    // 512 is the fastest or within 3 % of it on 19 (more workgroups = more slabs for the reduce and more prologues: -12 % at 1024)
    long long s = (512 + tiles - 1) / tiles;
    const long long smax = (M / SPLIT_PK) / 16 > 0 ? (M / SPLIT_PK) / 16 : 1;     // >= 16 steps per workgroup
    if (s > smax) s = smax;
    if (s > 512) s = 512;
    if (s < 1) s = 1;
    return (int)s;
}

static int bwd_weight_entry(const CcstConvDesc* d, const float* x, const uint32_t* x_absmax, const float* dy, const uint32_t* dy_absmax,
                            float* dw_oihw, int splits, int accumulate, void* ws, int64_t ws_bytes, void* stream);

extern "C" int ccst_conv2d_bwd_weight_split_f32(const CcstConvDesc* d, const float* x, const uint32_t* x_absmax, const float* dy,
                                                const uint32_t* dy_absmax, float* dw_oihw, int splits, int accumulate, void* ws,
                                                int64_t ws_bytes, void* stream) {
    CCST_REQUIRE(x_absmax && dy_absmax, "bwd_weight_split: the |max| words of x and dy are required");
    return bwd_weight_entry(d, x, x_absmax, dy, dy_absmax, dw_oihw, splits, accumulate, ws, ws_bytes, stream);
}

extern "C" int ccst_conv2d_bwd_weight_f32(const CcstConvDesc* d, const float* x, const float* dy, float* dw_oihw, int splits,
                                          int accumulate, void* ws, int64_t ws_bytes, void* stream) {
    return bwd_weight_entry(d, x, nullptr, dy, nullptr, dw_oihw, splits, accumulate, ws, ws_bytes, stream);
}

static int bwd_weight_entry(const CcstConvDesc* d, const float* x, const uint32_t* x_absmax, const float* dy, const uint32_t* dy_absmax,
                            float* dw_oihw, int splits, int accumulate, void* ws, int64_t ws_bytes, void* stream) {
    CCST_REQUIRE(d && x && dy && dw_oihw && ws, "bwd_weight: null pointer");
    CCST_REQUIRE(d->cin > 0 && d->cin % 4 == 0 && d->cout > 0 && d->cout % 4 == 0, "bwd_weight: cin/cout must be multiples of 4");
    CCST_REQUIRE(d->n > 0 && d->ho > 0 && d->wo > 0 && d->nky > 0 && d->nkx > 0 && splits >= 1, "bwd_weight: bad extents");
    CCST_REQUIRE((long long)d->n * d->ho * d->wo < 0x7fffffffLL, "bwd_weight: M too large");
    CCST_REQUIRE(d->xsH < (1 << 23) && d->xsW < (1 << 23) && d->hi < (1 << 15) && d->wi < (1 << 15) && d->ho < (1 << 15) && d->wo < (1 << 15) &&
                 (long long)d->hi * d->xsH < 0x7fffffffLL, "bwd_weight: extents beyond the 24-bit index arithmetic of the loader");
    const int ntap = d->nky * d->nkx;
    const long long need = (long long)splits * ntap * d->cin * d->cout * 4;
    if (ws_bytes < need) {
        ccst_set_error("bwd_weight: workspace %lld < %lld", (long long)ws_bytes, need);
        return CCST_EWORKSPACE;
    }
    BwdWArgs a;
    a.x = x; a.dy = dy; a.ws = (float*)ws;
    a.N = d->n; a.Ho = d->ho; a.Wo = d->wo; a.Hi = d->hi; a.Wi = d->wi; a.Cin = d->cin; a.Cout = d->cout;
    a.nky = d->nky; a.nkx = d->nkx; a.ay = d->ay; a.by = d->by; a.cy = d->cy; a.ax = d->ax; a.bx = d->bx; a.cx = d->cx;
    a.xsN = d->xsN; a.xsH = d->xsH; a.xsW = d->xsW;
    a.M = d->n * d->ho * d->wo;
    a.splits = splits;
    a.invHW = 1.0f / (float)(d->ho * d->wo);
    a.invWo = 1.0f / (float)d->wo;
    a.fastdiv = a.M < (1 << 22);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const bool pw = ntap == 1 && d->ay == 1 && d->ax == 1 && d->cy == 0 && d->cx == 0 && d->hi == d->ho && d->wi == d->wo &&
                    d->xsH == (long long)d->wi * d->xsW && d->xsN == (long long)d->hi * d->wi * d->xsW;
    int mi, nj;
    a.xmax = x_absmax;
    a.dmax = dy_absmax;
    if (x_absmax != nullptr) {          // half pieces on the 16-bit MFMA
        const long long xb = (long long)d->n * d->xsN * 4, db = (long long)a.M * d->cout * 4;
        CCST_REQUIRE(xb < 0x7fffffffLL && db < 0x7fffffffLL && d->xsN >= (long long)d->hi * d->xsH,
                     "bwd_weight_split: x and dy must be below 2^31 bytes each (32-bit buffer offsets)");
        a.xbytes = (unsigned)xb;
        a.dbytes = (unsigned)db;
        pick_tile_split(d->cin, d->cout, &mi, &nj);
        if (!pw && ntap > 1 && d->cin <= 32) rc = launch_split<1, 1, SPLIT_PK, false, true>(a, s);
        else if (mi == 2 && nj == 2) rc = pw ? launch_split<2, 2, SPLIT_PK, true>(a, s) : launch_split<2, 2, SPLIT_PK, false>(a, s);
        else if (mi == 2) rc = pw ? launch_split<2, 1, SPLIT_PK, true>(a, s) : launch_split<2, 1, SPLIT_PK, false>(a, s);
        else if (nj == 2) rc = pw ? launch_split<1, 2, SPLIT_PK, true>(a, s) : launch_split<1, 2, SPLIT_PK, false>(a, s);
        else rc = pw ? launch_split<1, 1, SPLIT_PK, true>(a, s) : launch_split<1, 1, SPLIT_PK, false>(a, s);
    } else {
    pick_tile(d->cin, d->cout, &mi, &nj);
    if (mi == 2 && nj == 2) rc = pw ? launch<2, 2, 16, true>(a, s) : launch<2, 2, 16, false>(a, s);
    else if (!pw && ntap > 1 && d->cin <= 32) rc = launch<1, 1, 32, false, true>(a, s);      // two taps per ci tile
    else rc = pw ? launch<1, 1, 32, true>(a, s) : launch<1, 1, 32, false>(a, s);
    }
    if (rc) return rc;
    // split-lanes of the reduce: the fewest (4, 8, 16, 32) that give >= 1024 workgroups, never more than there are slabs to share
    const long long cols = (long long)((d->cout + 31) / 32) * ntap;
    int nl = 4;
    while (nl < 32 && nl < splits && cols * ((d->cin + 32 / nl - 1) / (32 / nl)) < 1024) nl *= 2;
    dim3 grid((d->cout + 31) / 32, (d->cin + 32 / nl - 1) / (32 / nl), ntap);
#define CCST_REDUCE(NL_)                                                                                                       \
    hipLaunchKernelGGL(bwd_weight_reduce_kernel<NL_>, grid, dim3(256), 0, s, (const float*)ws, dw_oihw, ntap, d->cin, d->cout, splits, \
                       accumulate)
    if (nl == 4) CCST_REDUCE(4);
    else if (nl == 8) CCST_REDUCE(8);
    else if (nl == 16) CCST_REDUCE(16);
    else CCST_REDUCE(32);
#undef CCST_REDUCE
    return ccst_launch_status("bwd_weight_reduce");
}
