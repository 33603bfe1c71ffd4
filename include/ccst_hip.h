/*
 * ccst_hip.h -- C ABI of libccst_hip.so, the MI355X (gfx950) kernels behind the
 * CCST hot path.  Plain pointers and sizes only; no torch / C++ types.
 *
 * The reference (JeremyCJM/CCST) is pure Python and has no FFI of its own
 * (SURVEY.md 2.1): every entry point below replaces one of the *implicit ATen
 * ops* the reference's hot path dispatches, cited per function as
 * <file>:<line> relative to the reference root.  The Python host code in
 * ccst_amd/ binds these with ctypes (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's caching
 *     allocator in practice); the library allocates nothing and keeps no
 *     mutable global state; workspaces are passed explicitly;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL =
 *     the default stream); no call synchronises;
 *   - return 0 on success, a negative CCST_E* code for bad arguments, or a
 *     positive hipError_t if the launch failed; nothing throws;
 *   - activations are fp32 NHWC ("channels last") unless a flag says NCHW;
 *     conv weights are fp32 in the packed layout produced by
 *     ccst_pack_conv_weight_f32 from the checkpoint's OIHW tensor.
 */
#ifndef CCST_HIP_H
#define CCST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCST_OK 0
#define CCST_EINVAL (-1)   /* bad argument / unsupported shape */
#define CCST_EWORKSPACE (-2) /* workspace too small */

#define CCST_ABI_VERSION 2
int ccst_abi_version(void);
/* Human-readable text for the last non-zero return on this thread. */
const char* ccst_last_error(void);

/* ------------------------------------------------------------------------
 * Implicit-GEMM convolution on the fp32-input MFMA (v_mfma_f32_32x32x2_f32).
 *
 * One descriptor drives forward convs (any kernel size / stride / zero or
 * reflection padding), the stride-1 and per-parity stride-2 backward-data
 * convs, and the "virtual pixel" small-Cin stems, through affine index maps:
 *     iy = oy*ay + ky*by + cy        ix = ox*ax + kx*bx + cx
 *     weight tap = tap_base + ky*tap_sy + kx*tap_sx
 *     &y[n,oy,ox,co] = y + y_off + n*ysN + oy*ysH + ox*ysW + co*ysC
 *     &x[n,iy,ix,ci] = x + n*xsN + iy*xsH + ix*xsW + ci
 * Replaces: nn.ReflectionPad2d+nn.Conv2d(3x3)+nn.ReLU (+MaxPool2d ceil /
 * +Upsample nearest) of style_transfer/AdaIN/net.py:6-36,38-69, and the
 * bias-free Conv2d's of nets/resnet.py:136,160-161 + torchvision blocks
 * (forward and backward-data).
 * ------------------------------------------------------------------------ */
#define CCST_CONV_RELU      1u   /* ReLU epilogue                                   */
#define CCST_CONV_POOL2     2u   /* fused MaxPool2d(2,2,ceil_mode=True) epilogue:
                                    y is the POOLED tensor (ysH/ysW are its strides) */
#define CCST_CONV_UPS2      4u   /* input is read through a nearest x2 upsample:
                                    x is the SOURCE [N,Hi/2,Wi/2,Cin] tensor         */
#define CCST_CONV_REFLECT   8u   /* reflection padding (else zero padding)           */
#define CCST_CONV_FLIP     32u   /* ccst_conv3x3_halo_train_f32: taps in reverse order */
#define CCST_CONV_ACCUM    16u   /* y += conv(x) (no ReLU/pool): the gradient of a residual
                                    block input = identity-branch gradient already in y  */

typedef struct CcstConvDesc {
    int32_t n, ho, wo;          /* output pixel grid; GEMM M = n*ho*wo                */
    int32_t hi, wi;             /* logical input extent used for bounds / reflection  */
    int32_t cin, cout;          /* cin % 16 == 0; cout arbitrary                      */
    int32_t cout_pad;           /* packed-weight column count (multiple of 128)       */
    int32_t nky, nkx;           /* taps iterated                                      */
    int32_t ay, by, cy, ax, bx, cx;
    int32_t tap_base, tap_sy, tap_sx;
    int64_t xsN; int32_t xsH, xsW;
    int64_t y_off, ysN; int32_t ysH, ysW, ysC;
    uint32_t flags;
} CcstConvDesc;

int ccst_conv2d_igemm_f32(const CcstConvDesc* d, const float* x, const float* w_packed,
                          const float* bias /* may be NULL */, float* y, void* stream);
/* Same convolution (dense NHWC output, no ReLU / pool), additionally writing per-row-group (64 or 32 rows, one per
 * wave row of the dispatched tile) (sum, sum of squares) partials of the output, stats[groups][cout][2] with
 * groups = ccst_conv2d_igemm_stats_groups(M, cout, cin, taps): the
 * batch statistics the following BatchNorm2d needs (nets/resnet.py:138,162), so BN skips its own read pass. */
int ccst_conv2d_igemm_stats_f32(const CcstConvDesc* d, const float* x, const float* w_packed,
                                const float* bias, float* y, float* stats, void* stream);
int ccst_conv2d_igemm_stats_groups(int M, int cout, int cin /* padded, d->cin */, int taps /* nky*nkx */);
/* The streaming pointwise kernel on half pieces (round 5: every form, forward and backward-data): 1x1 convs of nets/resnet.py between
 * dense NHWC tensors (or a strided 1x1 read: the downsample branches) with every fp32 product formed from two IEEE-half pieces per
 * operand on the 16-bit MFMA (22 significant bits, fp32 accumulation).  x is scaled by the power of two of its |max| words (x_absmax:
 * CCST_ABSMAX_WORDS, below -- activations: left by the BatchNorm apply that produced them; gradients: by the BatchNorm backward's
 * dx_absmax) as it is split on its way into LDS; the weight arrives PRE-SPLIT and scaled from ccst_pack_conv_weight_split_f32 (same
 * [K/4][n_pad] units as ccst_pack_conv_weight_f32's 1x1 layout, each 16-byte unit = four hi halves | four lo halves; transpose = 1
 * for backward-data), packed with the SAME w_absmax words the call passes.  Safe at any finite fp32 magnitude, no host
 * synchronisation.  The form follows from the arguments exactly as in the fp32 entries: stats != NULL (training forward, BatchNorm
 * statistics epilogue: ccst_conv2d_igemm_stats_f32), CCST_CONV_ACCUM in d->flags with relu_mask and optionally the BatchNorm link
 * (ccst_conv2d_igemm_accum_masked_f32), the BatchNorm + ReLU link with bn_gamma / bn_beta (ccst_conv2d_igemm_bn_relu_bwd_f32), or
 * none of them (plain / y += conv).  Needs ccst_conv2d_stream_ok(d) -- and ccst_conv2d_pointwise_ok(d) for the masked / linked forms. */
/* The gather GEMM (ccst_conv2d_igemm_f32's kernel, 64x64 tiles) on half pieces, for the backward-data launches that are problems of
 * neither the streaming pointwise nor the halo kernel: the parity classes of a stride-2 3x3 conv, the strided 1x1 downsample branches
 * (any taps / strides / zero padding / strided output, CCST_CONV_ACCUM).  w_split = ccst_pack_conv_weight_split_f32 over all taps. */
int ccst_conv2d_igemm_half_f32(const CcstConvDesc* d, const float* x, const uint32_t* x_absmax, const float* w_split,
                               const uint32_t* w_absmax, float* y, void* stream);
int ccst_conv2d_stream_ok(const CcstConvDesc* d);
int ccst_conv2d_pointwise_half_f32(const CcstConvDesc* d, const float* x, const uint32_t* x_absmax, const float* w_split,
                                   const uint32_t* w_absmax, float* y, float* stats, const uint8_t* relu_mask, const float* bn_x,
                                   const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                   float* bn_partials, void* stream);
int ccst_pack_conv_weight_split_f32(const float* w_oihw, const uint32_t* w_absmax, float* packed, int cout, int cin, int ntap /* kh * kw */,
                                    int transpose, int k_pad, int n_pad, void* stream);
/* ... of n weights in one launch: jobs [n][8] int64 on the device = (w_oihw, packed, cout, cin, w_absmax, transpose + 2 * ntap, k_pad, n_pad). */
int ccst_pack_conv_weights_split_batch_f32(const int64_t* jobs_device, int njobs, void* stream);

/* 3x3 stride-1 "same" conv (reflection or zero padding) with the input halo staged once per 16-channel
 * chunk in LDS (A-side loads / LDS writes 9x fewer than the gather form): the AdaIN encoder/decoder
 * layers net.py:6-69.  x NHWC source [N,Hs,Ws,Cin] (Hs=H/2 with CCST_CONV_UPS2), w_packed as for
 * ccst_conv2d_igemm_f32, y dense NHWC [N,H,W,Cout] or its pooled form (CCST_CONV_POOL2).  flags: CCST_CONV_*. */
int ccst_conv3x3_halo_f32(const float* x, const float* w_packed, const float* bias, float* y, int N, int H, int W,
                          int Cin, int Cout, int cout_pad, uint32_t flags, void* stream);

int ccst_conv3x3_halo_narrow(int N, int H, int W, int Cout);   /* 1: the 128x64 tile is dispatched, 0: 128x128 */

/* |max| words: CCST_ABSMAX_WORDS = 64 uint32 (256 bytes) on the device, zeroed by the caller, into which a producing kernel
 * max-accumulates the raw fp32 bits of the largest |value| it wrote (one conditional atomic per workgroup into word id % 64) and from
 * which the half-piece kernels below derive their power-of-two operand scales ON THE DEVICE (one coalesced load per wave) -- no host
 * synchronisation anywhere.  ccst_absmax_f32 is the stand-alone producer (one pass over x) for tensors whose producer did not leave
 * the words.
 *   PER TENSOR for weights and for the ResNet train / eval kernels (BatchNorm couples the samples of a batch anyway);
 *   PER IMAGE -- [N][CCST_ABSMAX_WORDS], image n's words at + n * CCST_ABSMAX_WORDS -- for the activations of the AdaIN-path kernels
 *   (ccst_conv3x3_stem3_f32, ccst_conv3x3_f43_f32, ccst_conv3x3_halo_split_f32, ccst_conv3x3_zform_f32, ccst_adain_f32,
 *   ccst_adain_tile_sums_f32: every x_absmax / y_absmax of theirs), since ABI version 2: the reference is strictly per sample
 *   (function.py:4-13, net.py), so an image's scale -- and with it its bits -- must not depend on its batch-mates; ccst_absmax_samples_f32
 *   is their stand-alone producer (x [N][per_sample] contiguous). */
#define CCST_ABSMAX_WORDS 64
int ccst_absmax_f32(const float* x, int64_t n, uint32_t* absmax, void* stream);
int ccst_absmax_samples_f32(const float* x, int N, int64_t per_sample, uint32_t* absmax /* [N][CCST_ABSMAX_WORDS], zeroed */, void* stream);
/* ... of n tensors in one launch: table [n][2] int64 = (device pointer, 16-byte aligned; element count), absmax [n][CCST_ABSMAX_WORDS]
 * zeroed by the caller (the pointwise conv weights of a ResNet after each optimiser step, nets/resnet.py). */
int ccst_absmax_batch_f32(const int64_t* table, int n, uint32_t* absmax, void* stream);

/* The same convolution with every fp32 product computed as three products of 16-bit pieces on the half-precision MFMA (x = hi + lo,
 * hi = half(x), lo = half(x - hi): 22 significant bits; a b ~ a_lo b_hi + a_hi b_lo + a_hi b_hi, fp32 accumulation): 5.3x the fp32
 * MFMA's rate at about its accuracy.  Range-safe by construction: both operands are scaled by powers of two derived from their |max|
 * words (activations to < 2^14, weights to < 2^10: no finite fp32 input overflows half, and small tensors are lifted out of half's
 * subnormals), the accumulators are scaled back exactly (v_ldexp) and the bias is added after that.  The result is what the fp32
 * kernel gives to ~1e-6 of max |y| at ANY magnitude of x and w (tests/test_adain_gpu.py: 1e-30 .. 1e30).
 *   w_split from ccst_pack_conv_weight_halo_split_f32 (9 * cin * cout_pad floats' worth of [tap][cin/16][cout_pad][16 k hi | lo] rows)
 *   with w_absmax = the |max| words of w_oihw (ccst_absmax_f32), which the conv call takes too;
 *   x_absmax: the |max| words of x (any upper bound of max |x| is valid; from the producer's y_absmax or ccst_absmax_f32);
 *   y_absmax: NULL, or zeroed words receiving max |y| (of the stored, i.e. ReLU'd / pooled, output);
 * otherwise the x / y / flags contract of ccst_conv3x3_halo_f32. */
int ccst_pack_conv_weight_halo_split_f32(const float* w_oihw, float* w_split, int cout, int cin, int cout_pad, const uint32_t* w_absmax,
                                         int transpose /* 1: the backward-data weight (rows = cin, k = cout; cout_pad >= cin then) */, void* stream);
/* ... of n weights in one launch: jobs [n][8] int64 = (w_oihw, w_split, rows, k, cout_pad, w_absmax, transpose, 0), rows / k = the
 * GEMM's sides (cout, cin -- swapped when transpose). */
int ccst_pack_conv_weights_halo_split_batch_f32(const int64_t* jobs_device, int njobs, void* stream);
int ccst_conv3x3_halo_split_f32(const float* x, const uint32_t* x_absmax, const float* w_split, const uint32_t* w_absmax, const float* bias,
                                float* y, uint32_t* y_absmax, int N, int H, int W, int Cin, int Cout, int cout_pad, uint32_t flags,
                                float* chan_sum_partials, void* stream);
/* The same convolution (same x / y / flags / |max|-words contract as ccst_conv3x3_halo_split_f32) as Winograd F(4,3) along x on the half
 * pieces: SIX transform positions per QUAD of output pixels replace the three kx taps -- 18 instead of 36 k-steps per pixel quad, 1.5
 * instead of 3.0 executed 16-bit MFMA FLOPs per algorithmic FLOP (conv3x3_f43.hip); rounding 1-5e-6 of max |y| per layer.  u from
 * ccst_pack_conv_weight_f43_f32: 18 * cin * cout_pad floats' worth of [2 (ky * 3 + j) + group][cin/16][cout_pad][16 k hi | lo] rows of G g
 * (position q = 3 group + j), scaled by the power of two derived from w_absmax.  One 512-thread workgroup per CU covers 8 x 32 pixels x
 * 128 output channels (cout_pad a multiple of 128); layers with Cout <= 64 run an 8 x 32-pixel x 64-channel tile of four waves, two
 * workgroups per CU (cout_pad a multiple of 64); ccst_conv3x3_f43_workgroups tells a caller how many a layer launches.  One image must
 * have < 2^30 elements.  chan_sum_partials: NULL, or [ccst_conv3x3_f43_tiles(N,H,W)][Cout][4] (sum, M2, count, 0) of the un-pooled output
 * per (8x32-pixel tile, position group), an image's rows contiguous (see ccst_conv3x3_halo_split_tiles below).  Replaces
 * nn.Conv2d(.., (3, 3)) after ReflectionPad2d in style_transfer/AdaIN/net.py:6-36 (decoder) and :38-69 (vgg).
 * (Round 4's F(2,3) form, ccst_conv3x3_f23_f32, was retired in round 6 -- ABI version 2.) */
int ccst_pack_conv_weight_f43_f32(const float* w_oihw, float* u, int cout, int cin, int cout_pad, const uint32_t* w_absmax, void* stream);
int ccst_conv3x3_f43_f32(const float* x, const uint32_t* x_absmax, const float* u, const uint32_t* w_absmax, const float* bias, float* y,
                         uint32_t* y_absmax, int N, int H, int W, int Cin, int Cout, int cout_pad, uint32_t flags, float* chan_sum_partials,
                         const float* in_scale, const float* in_shift /* both NULL, or [N][Cin] each: the conv of in_scale * x + in_shift per
                            (image, input channel) -- applied inside the input transform (B^T is linear: one multiplication per transformed
                            value and one constant), x_absmax then being the words of the MAPPED tensor; CCST_CONV_REFLECT only, not with
                            POOL2 / UPS2.  How the AdaIN step (function.py:26-33 + the alpha blend)
                            rides in the decoder's first conv: ccst_adain_fold_affine_f32 below */,
                         void* stream);
int ccst_conv3x3_f43_workgroups(int N, int H, int W, int Cout);
int ccst_conv3x3_f43_tiles(int N, int H, int W);
/* chan_sum_partials (may be NULL; not with POOL2): [ccst_conv3x3_halo_split_tiles(N,H,W)][Cout][4] per-(8x16-pixel tile, wave row)
 * (sum, M2, count, 0) of the output after bias / ReLU, M2 = the sum of squares about the slab's OWN mean (no E[x^2] - mean^2
 * cancellation however large |mean| / sigma is), an image's rows contiguous -- the statistics ccst_adain_tile_sums_f32 and
 * ccst_chan_sums_finalize_f32 (partial_floats = 4) take instead of a pass over the tensor.  ccst_conv3x3_f43_f32 writes the same
 * quadruples (its own row count: ccst_conv3x3_f43_tiles). */
int ccst_conv3x3_halo_split_tiles(int N, int H, int W);
/* Fused Winograd F(2x2,3x3) (16 multiplies per 2x2 output tile and input channel instead of 36) for the ResNet trunk's 3x3 stride-1
 * layers (ccst_conv3x3_wino_train_f32 below): transformed weights from ccst_pack_conv_weight_wino_f32 (ccst_wino_weight_floats(cin,
 * cout_pad) floats, cout_pad a multiple of 32).  fp32 throughout; differs from the direct form by Winograd's rounding (~1e-6
 * relative).  (Its AdaIN entry -- bias / ReLU / pool / upsample flags -- was retired in round 5: no default path selected it.) */
int64_t ccst_wino_weight_floats(int cin, int cout_pad);
int ccst_pack_conv_weight_wino_f32(const float* w_oihw, float* u, int cout, int cin, int cout_pad, void* stream);

/* The AdaIN encoder's first layer (net.py:39-42: the 1x1 colour conv folded into ReflectionPad2d(1) + Conv2d(3,64,3x3) + ReLU) from the
 * contiguous NCHW image [N,3,H,W] to the NHWC map [N,H,W,64]; wa = ccst_pack_stem3_weight_f32(w [64,3,3,3], bias [64] or null)
 * (18*2*64 floats: a header with the weights' power-of-two scale, the half-piece fragments of the 16-bit MFMA's A operand and the bias;
 * fp32 products as three half-piece products, every pixel scaled by the power of two of its own largest tap: any finite fp32 image). */
int ccst_pack_stem3_weight_f32(const float* w_oihw, const float* bias, float* wa, int cout, void* stream);
int ccst_conv3x3_stem3_f32(const float* x_nchw, const float* wa, float* y_nhwc, int N, int H, int W, int relu,
                           uint32_t* y_absmax /* NULL or zeroed |max| words of y, see CCST_ABSMAX_WORDS */, void* stream);
/* calc_sum (mean_std_computation_effcientMem.py:103-115) without a pass over the tensor: folds the K per-tile records a conv epilogue
 * left (chan_sum_partials) into the [C] totals in fp64, fixed order (bitwise reproducible). */
int ccst_chan_sums_finalize_f32(const float* partials, int partial_floats /* 2: (sum, sum^2) pairs; 4: (sum, M2, count, 0), see
                                ccst_conv3x3_halo_split_f32 */, int K, int C, float* sum, float* sqsum, void* stream);
/* The Winograd kernel for the ResNet trunk's 3x3 stride-1 zero-padded bias-free convs (forward with the BatchNorm statistics
 * epilogue, backward-data with the weights from ccst_pack_conv_weight_wino_bwd_f32 and x = dY, optional y += with
 * CCST_CONV_ACCUM).  stats: NULL or [ccst_conv3x3_wino_stats_groups(N,H,W)][Cout][2]. */
int ccst_pack_conv_weight_wino_bwd_f32(const float* w_oihw, float* u, int cout, int cin, int cin_pad, void* stream);
int ccst_conv3x3_wino_train_f32(const float* x, const float* u_packed, float* y, float* stats, int N, int H, int W,
                                int Cin, int Cout, int cout_pad, uint32_t flags, void* stream);
int ccst_conv3x3_wino_stats_groups(int N, int H, int W);
/* many Winograd weight transforms in one launch: jobs_device [njobs][8] int64 {src OIHW, dst, n_out, n_in, n_in rounded up to 16,
 * n_out rounded up to 32, bwd, 0}, (n_out, n_in) = channels of the conv the transform serves. */
int ccst_pack_conv_weights_wino_batch_f32(const int64_t* jobs_device, int njobs, void* stream);

/* The same kernel for the ResNet trunk's 3x3 stride-1 zero-padded, bias-free convs (nets/resnet.py:160-161 via the
 * torchvision blocks), forward AND backward-data: flags = CCST_CONV_FLIP (reverse the taps; with the transposed packed
 * weight that is dX of the conv) | CCST_CONV_ACCUM (y += conv).  stats (NULL or [groups][Cout][2], groups =
 * ccst_conv3x3_halo_stats_groups(N,H,W)): (sum, sum of squares) partials of y for the following BatchNorm2d. */
int ccst_conv3x3_halo_train_f32(const float* x, const float* w_packed, float* y, float* stats, int N, int H, int W,
                                int Cin, int Cout, int cout_pad, uint32_t flags, void* stream);
int ccst_conv3x3_halo_stats_groups(int N, int H, int W);
/* ... on half pieces (16-bit MFMA, fp32 accuracy; as ccst_conv3x3_halo_split_f32): x scaled by its |max| words -- an activation's from
 * the BatchNorm apply that produced it, a gradient's from the BatchNorm backward (dx_absmax) --, w_split from
 * ccst_pack_conv_weight_halo_split_f32 (transpose = 1 together with CCST_CONV_FLIP: backward-data).  Same flags and statistics. */
int ccst_conv3x3_halo_train_split_f32(const float* x, const uint32_t* x_absmax, const float* w_split, const uint32_t* w_absmax, float* y,
                                      float* stats, int N, int H, int W, int Cin, int Cout, int cout_pad, uint32_t flags,
                                      const float* bn_x, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                      const float* bn_beta, float* bn_partials, void* stream);
/* bn_x .. bn_partials (all or none; plain backward-data only): y is the output gradient of a BatchNorm + ReLU with input bn_x
 * [N,H,W,Cout] that only this conv reads (bn1 -> conv2): the epilogue recomputes the ReLU mask (bn(bn_x) > 0), stores the MASKED
 * gradient and leaves that BatchNorm's backward partial sums (sum g, sum g*xhat) in bn_partials
 * [ccst_conv3x3_halo_stats_groups(N,H,W)][Cout][2] for ccst_bn_train_bwd_partials_f32 -- one pass over (x, g) less. */

/* The tile code (WM WN NT as decimal digits: 222 = 128x128, 221 = 128x64, 412 = 256x64, 411 = 256x32; 1221 / 1222 =
 * 64x64 with a 16- / 32-channel k-step for grids that cannot fill the chip) ccst_conv2d_igemm_f32 dispatches for
 * M = n*ho*wo output pixels, cout channels, cin (padded) input channels and taps = nky*nkx. */
int ccst_conv2d_igemm_tile(int M, int cout, int cin, int taps, int pool);

/* y = relu_mask ? (y + conv(x)) : 0 for a pointwise (1x1, stride 1, dense NHWC) problem with CCST_CONV_ACCUM: the gradient that
 * reaches a residual block's input is (identity share, already in y) + (first conv's backward-data), and that input is the previous
 * block's ReLU output -- masking the SUM here (relu_mask[M*cout/4] as written by ccst_bn_train_fwd_mask_f32: bit j of byte i <->
 * element 4 i + j) lets that block's BatchNorm backward take it as is: no mask reads there and no separate masked copy for its
 * skip connection (nets/resnet.py:160-165 inside loss.backward(), fed_run.py:79).  ccst_conv2d_pointwise_ok(d) = 1 where the form exists. */
int ccst_conv2d_pointwise_ok(const CcstConvDesc* d);
int ccst_conv2d_igemm_accum_masked_f32(const CcstConvDesc* d, const float* x, const float* w_packed, float* y,
                                       const uint8_t* relu_mask, const float* bn_x, const float* bn_mean,
                                       const float* bn_invstd, float* bn_partials, void* stream);
/* The same idea for a BatchNorm + ReLU WITHOUT a residual (bn2 -> conv3 of a bottleneck): y = (bn(bn_x) > 0) ? conv(x) : 0, i.e. the
 * backward-data result masked by the ReLU of the BatchNorm whose output gradient it is (mask recomputed from that BatchNorm's own
 * input), plus the partial sums for ccst_bn_train_bwd_partials_f32. */
int ccst_conv2d_igemm_bn_relu_bwd_f32(const CcstConvDesc* d, const float* x, const float* w_packed, float* y, const float* bn_x,
                                      const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                      float* bn_partials, void* stream);
/* bn_x .. bn_partials (all or none): the masked sum is the output gradient of a BatchNorm with input bn_x [M][cout] and saved
 * mean / invstd; the epilogue also leaves that BatchNorm's backward partial sums (sum g, sum g*xhat) per 32 rows in
 * bn_partials[2*ceil(M/64)][cout][2], for ccst_bn_train_bwd_partials_f32 -- one pass over (x, g) less. */

/* The decoder's last layer (net.py:34-35: 3x3 stride 1, 64 -> 3 channels; x NHWC [N,H,W,Cin], y NCHW [N,Cout,H,W]; HBM-bound at
 * 13 FLOP/B) as a 1x1 convolution to 9*Cout "tap planes" on the 16-bit MFMA (fp32 products as three half-piece products,
 * fp32 accumulate; 27 planes padded to 32 for Cout = 3) followed by the nine shifted fp32 adds (conv3x3_zform.hip, round 4): every
 * pixel's record is read once, whole.  Cin 32 or 64, Cout 1..3.  x_absmax / w_absmax: the |max| words of x and of the weight
 * (CCST_ABSMAX_WORDS each: the operands' power-of-two scales are derived from them on the device, any finite fp32 magnitude is safe).
 * w_packed: ccst_pack_conv_weight_zform_f32 of the [3][3][Cout][Cin] weight, ccst_conv3x3_zform_weight_floats(Cin) floats.
 * Replaces ReflectionPad2d + nn.Conv2d(64, 3, (3, 3)) of style_transfer/AdaIN/net.py:34-35.  (The VALU kernel ccst_conv3x3_smallco_f32
 * was retired in round 6 -- ABI version 2.) */
int64_t ccst_conv3x3_zform_weight_floats(int Cin);
int ccst_pack_conv_weight_zform_f32(const float* w_tap_co_ci, const uint32_t* w_absmax, float* packed, int Cin, int Cout, void* stream);
int ccst_conv3x3_zform_f32(const float* x, const uint32_t* x_absmax, const float* w_packed, const uint32_t* w_absmax,
                           const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int reflect, int relu, void* stream);

/* OIHW [cout][cin][kh][kw] -> packed [kh*kw][cin/4][cout_pad][4] (transpose=0), or the
 * backward-data operand [kh*kw][cout/4][cin_pad][4] (transpose=1: GEMM-K runs over cout).
 * The K-side extent (cin, or cout when transposed) must be a multiple of 4; it is padded
 * with zeros up to k_pad (a multiple of 16).  n_pad: multiple of 128. */
int ccst_pack_conv_weight_f32(const float* w_oihw, float* packed, int cout, int cin, int kh, int kw,
                              int transpose, int k_pad, int n_pad, void* stream);

/* ccst_pack_conv_weight_f32 for many tensors in one launch (the per-step refresh of a model's packed weights after
 * ccst_sgd_f32, fed_run.py:80).  jobs_device: device array [njobs][8] of int64 {src OIHW pointer, dst packed pointer,
 * cout, cin, kh*kw, transpose, k_pad, n_pad}. */
int ccst_pack_conv_weights_batch_f32(const int64_t* jobs_device, int njobs, void* stream);

/* NCHW [N,C<=4,H,W] -> zero/reflect padded NHWC4 [N, H+2*pad, Wp, 4] (Wp >= W+2*pad, extra
 * columns and channels C..3 zero).  Feeds the small-Cin stems (net.py:39-41; nets/resnet.py:136)
 * as "virtual pixel" convs. */
int ccst_nchw_to_nhwc4_pad_f32(const float* x, float* y, int N, int C, int H, int W, int pad, int Wp,
                               int reflect, void* stream);

/* Stand-alone layers of net.py:6-69 for un-fused use (arbitrary slices of the nn.Sequential), NHWC,
 * C % 4 == 0.  mode 0: nn.ReLU; 1: nn.ReflectionPad2d(pad) -> [H+2p,W+2p]; 2: nn.Upsample(x2 nearest)
 * -> [2H,2W]; 3: nn.MaxPool2d((2,2),(2,2),(0,0),ceil_mode=True) -> [ceil(H/2),ceil(W/2)]. */
int ccst_nhwc_layer_f32(int mode, const float* x, float* y, int N, int H, int W, int C, int pad, void* stream);
/* API-edge layout changes: x[N,C,HW] -> y[N,HW,Cp] (channels C..Cp-1 zero) and back (first C of Cs). */
int ccst_nchw_to_nhwc_f32(const float* x, float* y, int N, int C, int HW, int Cp, void* stream);
int ccst_nhwc_to_nchw_f32(const float* x, float* y, int N, int C, int HW, int Cs, void* stream);

/* Image output edge (CCST_OverallStyleTransfer.py:156-167, torchvision save_image semantics):
 * NCHW float -> NHWC uint8 with v*255+0.5 clamped to [0,255]. */
int ccst_quantize_u8_hwc_f32(const float* x_nchw, uint8_t* y_nhwc, int N, int C, int HW, void* stream);
/* --output_size (CCST_OverallStyleTransfer.py:154-157): transforms.Resize on the stylised float TENSOR = bilinear interpolation,
 * align_corners=False, no antialiasing, of `planes` = N*C planes [H,W] -> [oh,ow]. */
int ccst_resize_bilinear_nchw_f32(const float* x, float* y, int planes, int H, int W, int oh, int ow, void* stream);

/* Image input edge on the GPU.  Replaces, per image, the CPU transform chain of
 * data/data_helper.py:161-181 (train: RandomResizedCrop -> ToTensor -> Normalize -> RandomHorizontalFlip; val:
 * Resize -> ToTensor -> Normalize) and style_transfer/AdaIN/cjm_util/data_helper.py:46-49 (Resize -> ToTensor):
 *   out[n,c,oy,ox'] = ((PIL_bilinear_resize(crop(src_n))[oy,ox,c] / 255) - mean[c]) / std[c],  ox' = flip ? W-1-ox : ox
 * byte-identical to PIL's uint8 resize (libImaging/Resample.c fixed-point tables, built on the host by
 * ccst_image_plan) and bit-identical to the fp32 ToTensor/Normalize arithmetic.
 * The caller fills src_off, src_w, crop_i..crop_w and flip of every CcstImageXform; ccst_image_plan, a HOST function that needs no GPU, fills
 * kx/ky and the table offsets and writes the int32 tables; it returns the number of int32 used (call it with
 * tables == NULL to size the buffer), or a negative CCST_E* code.  xf and tables are then copied to the device
 * for ccst_crop_resize_norm_u8_f32: src = packed uint8 HWC (3-channel) decoded images, dst_nchw fp32 [n,3,H,W]
 * and/or dst_u8_nhwc uint8 [n,H,W,3] (the un-normalised resize, either may be NULL); mean3/std3 are HOST
 * pointers to three floats (NULL = 0 / 1, i.e. ToTensor only). */
typedef struct CcstImageXform {
    int64_t src_off;                            /* byte offset of the image's first pixel in src             */
    int32_t src_w;                              /* decoded width (row pitch = 3*src_w bytes)                 */
    int32_t crop_i, crop_j, crop_h, crop_w;     /* torchvision F.resized_crop rectangle: top, left, h, w     */
    int32_t flip;                               /* horizontal flip of the OUTPUT                             */
    int32_t kx, ky;                             /* (plan) taps per output column / row                       */
    int32_t bounds_x, coefs_x, bounds_y, coefs_y; /* (plan) int32 offsets into tables                        */
} CcstImageXform;
int64_t ccst_image_plan(int n, CcstImageXform* xf, int out_h, int out_w, int32_t* tables, int64_t cap);
int ccst_crop_resize_norm_u8_f32(const uint8_t* src, const CcstImageXform* xf, const int32_t* tables, float* dst_nchw,
                                 uint8_t* dst_u8_nhwc, int n, int out_h, int out_w, const float* mean3,
                                 const float* std3, void* stream);

/* Backward-weight: dW[tap][ci][co] = sum_m X[n, oy*ay+ky*by+cy, ox*ax+kx*bx+cx, ci] * dY[m, co]
 * (zero padding; x indexed with d's x strides, dY dense [M][cout]), split over `splits` pixel
 * ranges into ws[splits][ntap][cin][cout] partial slabs, then summed in fixed order (bitwise
 * reproducible, no float atomics) into OIHW dw[cout][cin][nky][nkx]; accumulate=1 adds to dw.
 * ws_bytes >= splits*ntap*cin*cout*4.  cin, cout multiples of 4.  Replaces Conv2d's weight
 * gradient inside loss.backward(), federated/fed_run.py:79. */
int ccst_conv2d_bwd_weight_f32(const CcstConvDesc* d, const float* x, const float* dy, float* dw_oihw,
                               int splits, int accumulate, void* ws, int64_t ws_bytes, void* stream);
/* Split count that fills the chip for this problem (M = n*ho*wo). */
int ccst_conv2d_bwd_weight_splits(int M, int cin, int cout, int ntap);
/* The same weight gradient with every fp32 product as three products of IEEE-half pieces on the 16-bit MFMA (22 significant bits,
 * fp32 accumulation; the LDS images stay [pixel][channel] and the pixel-major MFMA operands come from the transposing LDS read).
 * x_absmax / dy_absmax: the |max| words (CCST_ABSMAX_WORDS) of x and dy -- x's from the BatchNorm apply that produced it, dy's from
 * the BatchNorm backward that produced it (dx_absmax below), or ccst_absmax_f32; any upper bound is valid.  Both tensors are scaled
 * by powers of two derived from the words on the device and the result is scaled back: any finite fp32 magnitude is safe.  Same
 * slabs, reduce and determinism as ccst_conv2d_bwd_weight_f32; its own split count (the loop per pixel is ~3x shorter). */
int ccst_conv2d_bwd_weight_split_f32(const CcstConvDesc* d, const float* x, const uint32_t* x_absmax, const float* dy,
                                     const uint32_t* dy_absmax, float* dw_oihw, int splits, int accumulate, void* ws,
                                     int64_t ws_bytes, void* stream);
int ccst_conv2d_bwd_weight_split_splits(int M, int cin, int cout, int ntap);

/* ------------------------------------------------------------------------
 * AdaIN feature statistics / normalisation.  layout: 0 = NCHW planes, 1 = NHWC.
 * ------------------------------------------------------------------------ */
/* function.py:4-13 calc_mean_std: per-(n,c) mean and sqrt(var_unbiased + eps).  mean/std: [N*C]. */
int ccst_calc_mean_std_f32(const float* x, float* mean, float* std, int N, int C, int HW, int layout,
                           float eps, void* ws, int64_t ws_bytes, void* stream);
/* function.py:26-33 adaIN_StyleStat_ContentFeat fused with the alpha blend of
 * CCST_OverallStyleTransfer.py:45:  t = ((x-mu_c)/sigma_c)*sigma_s + mu_s ; y = t*alpha + x*(1-alpha).
 * style_mean/std: [C] if style_per_n==0 (broadcast over N) else [N*C] (function.py:16-24). */
int ccst_adain_f32(const float* x, const float* style_mean, const float* style_std, int style_per_n,
                   float alpha, float* y, int N, int C, int HW, int layout, float eps,
                   void* ws, int64_t ws_bytes, uint32_t* y_absmax /* NULL or zeroed |max| words of y (CCST_ABSMAX_WORDS) */, void* stream);
/* The same AdaIN (+ alpha blend) for an NHWC x [N][HW][C] whose producer already left the statistics: partials
 * [N * tiles_per_image][C][2] = per-(spatial tile, channel) (sum, sum of squares) of x, the tiles of image n contiguous -- what
 * ccst_conv3x3_wino4w_f32 writes into chan_sum_partials (tiles_per_image = ccst_wino4w_spatial_tiles(1, H, W)), or the centred
 * (sum, M2, count, 0) quadruples of the half-piece kernels.  Two launches: the records folded ONCE in fp64 (a thread per (image,
 * channel)) into mean_out / std_out, then x streamed once against them; no pass over x for the statistics.  C % 64 == 0.
 * mean_out / std_out: [N*C] floats each, REQUIRED (caller-owned): the content statistics, handed from the fold to the stream. */
int ccst_adain_tile_sums_f32(const float* x, const float* partials, int partial_floats /* 2 or 4, as ccst_chan_sums_finalize_f32 */,
                             int tiles_per_image, const float* style_mean,
                             const float* style_std, int style_per_n, float alpha, float* y, int N, int C, int HW, float eps,
                             float* mean_out, float* std_out, uint32_t* y_absmax /* NULL or zeroed |max| words of y */, void* stream);
/* CCST_OverallStyleTransfer.py:36-45, style_transfer's interpolation branch after the AdaIN of the K copies of one content image
 * against K styles: out[elems] = (sum_k weights[k] * base[k][elems], from zero in index order, products and sums rounded separately)
 * * alpha + content0[elems] * one_minus_alpha.  weights: K floats on the device.  Elementwise: any (common) layout. */
/* The AdaIN step WITHOUT a pass over the features (round 6): function.py:26-33 and the alpha blend of CCST_OverallStyleTransfer.py:45 as
 * the per-(image, channel) affine map y = a x + b (a = alpha sigma_s / sigma + 1 - alpha, b = alpha (mu_s - mu sigma_s / sigma), fp64,
 * rounded once) that ccst_conv3x3_f43_f32 applies through in_scale / in_shift.  partials: the centred records [N * tiles_per_image][C][4]
 * (sum, M2, count, max |x|) the F(4,3) kernel's epilogue left for x; mean_out / std_out / a_out / b_out: [N*C] floats each; y_absmax: zeroed
 * [N][CCST_ABSMAX_WORDS], receives |a| max |x| + |b| -- with x_nonneg (x >= 0: its producer applied ReLU) the exact max(|b|, |a max x + b|) --
 * maximised over the image's channels: a bound of max |y|, which is all a scale needs.
 * One small launch.  Result: within two fp32 roundings of the reference's four separately rounded operations (ccst_adain_tile_sums_f32
 * reproduces those bit for bit and stays the stand-alone entry). */
int ccst_adain_fold_affine_f32(const float* partials, int tiles_per_image, const float* style_mean, const float* style_std, int style_per_n,
                               float alpha, int x_nonneg, int N, int C, int HW, float eps, float* mean_out, float* std_out, float* a_out,
                               float* b_out, uint32_t* y_absmax, void* stream);
int ccst_interp_blend_f32(const float* base, const float* content0, const float* weights, int K, int64_t elems, float alpha,
                          float one_minus_alpha, float* out, void* stream);
/* mean_std_computation_effcientMem.py:103-115 calc_sum: per-channel sum and sum of squares over
 * N*H*W.  sum/sqsum: [C]. */
int ccst_chan_sums_f32(const float* x, float* sum, float* sqsum, int N, int C, int HW, int layout,
                       void* ws, int64_t ws_bytes, void* stream);
/* Workspace bytes the three calls above need. */
int64_t ccst_stats_workspace_bytes(int N, int C, int HW);

/* ------------------------------------------------------------------------
 * ResNet training ops (nets/resnet.py:132-191 + torchvision blocks; fed_run.py:49-80).  NHWC.
 * ------------------------------------------------------------------------ */
/* BatchNorm2d training forward: batch mean / biased var over the M = N*H*W rows, running-stat update
 * (momentum, unbiased var), y = (x-mean)*invstd*gamma+beta [+ residual] [ReLU].
 * save_mean/save_invstd: [C] for backward.  residual may be NULL.  stats_in (may be NULL): per-channel
 * (sum, sum^2) partials [stats_groups][C][2] from ccst_conv2d_igemm_stats_f32 -- then x is not re-read.
 * relu_mask (may be NULL): [M*C/4] bytes, bit j of byte i <-> element 4 i + j is positive before the ReLU -- the backward then reads
 * 1/16 of a tensor per pass instead of the whole saved output y. */
int ccst_bn_train_fwd_mask_f32(const float* x, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float momentum, float eps, const float* residual, int relu,
                               float* y, uint8_t* relu_mask, float* save_mean, float* save_invstd, int64_t M, int C,
                               const float* stats_in, int stats_groups, void* ws, int64_t ws_bytes,
                               uint32_t* y_absmax /* NULL, or zeroed |max| words (CCST_ABSMAX_WORDS) receiving max |y|: what the
                                                     half-piece pointwise conv that reads y scales it by */,
                               void* stream);
/* BatchNorm2d eval forward with running stats (fed_run.py:216). */
int ccst_bn_eval_fwd_f32(const float* x, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, const float* residual, int relu, float* y,
                         int64_t M, int C, uint32_t* y_absmax /* NULL or zeroed |max| words of y: the eval forward's pointwise convs then run on half pieces too */,
                        void* stream);
/* BatchNorm2d backward: dx, dgamma, dbeta (accumulate=1 adds into dgamma/dbeta).  With relu=1 the ReLU mask
 * comes from relu_mask (the forward's byte mask), else from the saved output y (y > 0), or -- when both are NULL, allowed only if no
 * residual was added in the forward -- is recomputed from x as (x-mean)*invstd*gamma+beta > 0 (one tensor read less per pass).
 * If d_residual != NULL it receives the masked incoming gradient (the skip connection's share). */
int ccst_bn_train_bwd_mask_f32(const float* dy, const float* x, const float* y, const uint8_t* relu_mask,
                               const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                               int relu, float* dx, float* d_residual, float* dgamma, float* dbeta, int accumulate,
                               int64_t M, int C, void* ws, int64_t ws_bytes,
                               uint32_t* dx_absmax /* NULL, or zeroed |max| words (CCST_ABSMAX_WORDS) receiving max |dx|: dx is the
                                                      output gradient of the conv in front of this BatchNorm, and the half-piece
                                                      weight-gradient kernel (ccst_conv2d_bwd_weight_split_f32) scales it by them */,
                               void* stream);
/* BatchNorm2d backward WITHOUT its reduction pass: the per-channel partial sums [groups][C][2] of (dy, dy*xhat) come from the kernel
 * that produced dy (ccst_conv2d_igemm_accum_masked_f32); dy carries no ReLU to undo and the skip connection's share is dy itself. */
int ccst_bn_train_bwd_partials_f32(const float* dy, const float* x, const float* gamma, const float* save_mean,
                                   const float* save_invstd, const float* partials, int groups, float* dx, float* dgamma,
                                   float* dbeta, int accumulate, int64_t M, int C, void* ws, int64_t ws_bytes,
                                   uint32_t* dx_absmax /* as ccst_bn_train_bwd_mask_f32 */, void* stream);
int64_t ccst_bn_workspace_bytes(int64_t M, int C);

/* BatchNorm2d (training) -> ReLU -> MaxPool2d(3, 2, 1), the ResNet stem (nets/resnet.py:138-140), without the full-resolution tensors in
 * between: the forward pools relu(bn(x)) straight from the conv output x [N,H,W,C] into y_pooled [N,Ho,Wo,C] + idx (as
 * ccst_maxpool3s2_fwd_f32); the backward takes the pooled gradient and gathers it through idx inside both BatchNorm-backward passes
 * (ReLU mask recomputed from x).  Same statistics / running-stat semantics as ccst_bn_train_fwd_mask_f32. */
int ccst_bn_relu_maxpool_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* running_mean,
                                       float* running_var, float momentum, float eps, float* y_pooled, uint32_t* idx,
                                       float* save_mean, float* save_invstd, int N, int H, int W, int C, int Ho, int Wo,
                                       const float* stats_in, int stats_groups, void* ws, int64_t ws_bytes,
                                       uint32_t* y_absmax /* NULL or zeroed |max| words of y_pooled */, void* stream);
int ccst_bn_relu_maxpool_train_bwd_f32(const float* dy_pooled, const uint32_t* idx, const float* x, const float* gamma,
                                       const float* beta, const float* save_mean, const float* save_invstd, float* dx,
                                       float* dgamma, float* dbeta, int accumulate, int N, int H, int W, int C, int Ho, int Wo,
                                       void* ws, int64_t ws_bytes, uint32_t* dx_absmax /* as ccst_bn_train_bwd_mask_f32 */, void* stream);

/* MaxPool2d(kernel 3, stride 2, padding 1) nets/resnet.py:140, NHWC, C % 4 == 0.  idx[N,Ho,Wo,C/4]
 * packs, per channel, the window position (0..8) of the first maximum (one byte each); backward
 * gathers through it (deterministic, no atomics). */
int ccst_maxpool3s2_fwd_f32(const float* x, float* y, uint32_t* idx, int N, int H, int W, int C, int Ho, int Wo,
                            void* stream);
int ccst_maxpool3s2_bwd_f32(const float* dy, const uint32_t* idx, float* dx, int N, int H, int W, int C,
                            int Ho, int Wo, void* stream);
/* AvgPool2d(7) on a 7x7 map + flatten (nets/resnet.py:145,189-190): [N,HW,C] -> [N,C]. */
int ccst_avgpool_fwd_f32(const float* x, float* y, int N, int HW, int C, void* stream);
int ccst_avgpool_bwd_f32(const float* dy, float* dx, int N, int HW, int C, void* stream);
/* Linear (nets/resnet.py:146): y[N,O] = x[N,K] w[O,K]^T + b. */
int ccst_linear_fwd_f32(const float* x, const float* w, const float* b, float* y, int N, int K, int O, void* stream);
int ccst_linear_bwd_f32(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                        int accumulate, int N, int K, int O, void* stream);
/* CrossEntropyLoss (mean) forward+backward (fed_run.py:554,65): loss[1], dlogits[N,O] (already
 * scaled by 1/N), correct[1] = #argmax==label (fed_run.py:67,71). */
int ccst_softmax_ce_f32(const float* logits, const int64_t* labels, float* loss, float* dlogits,
                        int32_t* correct, int N, int O, void* stream);
/* SGD p -= lr*g over a flat arena (fed_run.py:657,80), and the FedAvg pre-scale p *= s. */
int ccst_sgd_f32(float* p, const float* g, float lr, int64_t n, void* stream);
int ccst_scale_f32(float* p, float s, int64_t n, void* stream);
/* communication()'s fedavg branch (fed_run.py:400-414) over K <= 16 flat fp32 arenas of n floats on ONE GPU, in one pass: server[i] =
 * sum_k weights[k] * clients[k][i] (accumulated from zero in client order, the arithmetic of K ccst_sgd_f32 calls: bit-identical) and
 * every client overwritten with the result.  clients_host / weights_host: HOST arrays of K device pointers / K floats. */
int ccst_fedavg_f32(float* server, float* const* clients_host, const float* weights_host, int K, int64_t n, void* stream);
/* Glue the reference gets from ATen, as HIP launches so that no framework kernel runs inside a train step:
 * optimizer.zero_grad() (fed_run.py:58) as one fill of the flat gradient arena; BatchNorm2d's num_batches_tracked += 1
 * over the re-homed int64 counters; the chain-rule scale of CrossEntropyLoss's saved dlogits by the incoming gradient
 * (a device scalar); the stem conv's weight gradient from its virtual-pixel form [cout][kwp*4][kh] to OIHW (+=). */
int ccst_fill_f32(float* p, float value, int64_t n, void* stream);
int ccst_add_i64(int64_t* p, int64_t delta, int n, void* stream);
int ccst_mul_scalar_f32(float* y, const float* x, const float* s_dev, int64_t n, void* stream);
int ccst_stem_grad_unfold_f32(const float* gv, float* g_oihw, int cout, int kwp, int kh, int kw, int C, int accumulate,
                              void* stream);
/* elementwise helpers: y = relu(a + b) and its backward mask are folded into the BN calls above. */

#ifdef __cplusplus
}
#endif
#endif /* CCST_HIP_H */
