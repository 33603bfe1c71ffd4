"""Thin Python wrappers over the C ABI: torch tensors in, raw device pointers
out.  Activations are NHWC-contiguous fp32 tensors of shape [N,H,W,C];
`to_api` / `from_api` convert to/from the logical-NCHW (channels_last) tensors
the reference's Python surface exchanges -- zero-copy whenever possible.

PyTorch here is plumbing (allocator, streams); every op is a HIP kernel from
libccst_hip.so.  Nothing in this file falls back to torch compute on failure.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import CONV_POOL2, CONV_REFLECT, CONV_RELU, CONV_UPS2, CcstConvDesc, check, ptr, stream_ptr

NCHW, NHWC = 0, 1

# Bumped whenever a kernel rewrites parameters through raw pointers (SGD step, FedAvg), which torch's
# tensor version counters cannot see; packed-weight caches key on it.
WEIGHTS_EPOCH = 0


def bump_weights_epoch():
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1


# 3x3 reflect-pad convs go through the halo-in-LDS kernel (conv3x3_halo.hip); False keeps the gather kernel (a test compares the two).
USE_HALO = True
# The zero-padded 3x3 stride-1 convs of the ResNet trunk (forward with BN statistics, backward-data with flipped taps, y +=)
# CAN run on the halo kernel too (ccst_conv3x3_halo_train_f32) when its 8x16-pixel tiles cover the map well enough (56x56:
# 88 %, 28x28 / 14x14: 77 %; 7x7 would be 38 %).  Measured: per layer 97-105 TF vs 90-100 TF on the gather kernel, but
# the same step time (ResNet50 2837 vs 2839 img/s, ResNet18 within noise) -- short K loops and tile waste eat the
# advantage -- so the gather kernel stays the default (the fp32 train form remains the tests' reference of the half-piece one).
HALO_ZERO_PAD = False
HALO_MIN_COVER = 0.7


def halo_train_ok(H, W, cin, cout):
    if not (USE_HALO and HALO_ZERO_PAD) or cin % 16 != 0:
        return False
    return H * W >= HALO_MIN_COVER * (round_up(H, 8) * round_up(W, 16))


# ResNet trunk: 3x3 stride-1 convs (forward with BN statistics, backward-data) on the Winograd kernel when its 8x16-pixel tiles
# cover the map well enough (False: the gather kernel).
RESNET_WINO = True
RESNET_WINO_COVER = 0.7


def wino_train_ok(H, W, cin, cout):
    return RESNET_WINO and cin % 16 == 0 and cout % 16 == 0 and H * W >= RESNET_WINO_COVER * (round_up(H, 8) * round_up(W, 16))


def pack_wino(w_oihw, bwd=False, out=None):
    """Winograd-transformed copy of a 3x3 OIHW weight: (u, cout_pad, cout) of the forward conv, or with bwd=True of the
    backward-data conv dY -> dX (its `cout` is the forward Cin)."""
    cout, cin = w_oihw.shape[0], w_oihw.shape[1]
    lib = _lib.load()
    n_out, n_in = (cin, cout) if bwd else (cout, cin)
    pad = round_up(n_out, 32)
    nfl = int(lib.ccst_wino_weight_floats(n_in, pad))
    u = out if out is not None and out.numel() == nfl else torch.empty(nfl, device=w_oihw.device, dtype=torch.float32)
    w = w_oihw.contiguous()
    if bwd:
        check(lib.ccst_pack_conv_weight_wino_bwd_f32(ptr(w), ptr(u), cout, cin, pad, stream_ptr()), "pack_wino_bwd")
    else:
        check(lib.ccst_pack_conv_weight_wino_f32(ptr(w), ptr(u), cout, cin, pad, stream_ptr()), "pack_wino")
    return u, pad, n_out


def conv3x3_wino_train(x, packed, want_stats=False, accumulate_into=None, tag=""):
    """3x3 stride-1 zero-padded bias-free conv on the Winograd kernel (ResNet trunk).  packed = pack_wino(...)."""
    u, pad, cout = packed
    N, H, W, Cx = x.shape
    lib = _lib.load()
    if accumulate_into is not None:
        assert tuple(accumulate_into.shape) == (N, H, W, cout) and accumulate_into.is_contiguous() and not want_stats
        y = accumulate_into
    else:
        y = torch.empty((N, H, W, cout), device=x.device, dtype=torch.float32)
    stats = None
    if want_stats:
        stats = torch.empty((lib.ccst_conv3x3_wino_stats_groups(N, H, W), cout, 2), device=x.device, dtype=torch.float32)
    flags = _lib.CONV_ACCUM if accumulate_into is not None else 0
    args = (ptr(x), ptr(u), ptr(y), ptr(stats), N, H, W, Cx, cout, pad, flags, stream_ptr())
    if TIMING is None:
        check(lib.ccst_conv3x3_wino_train_f32(*args), "conv3x3_wino_train")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.ccst_conv3x3_wino_train_f32(*args), "conv3x3_wino_train")
        e1.record()
        TIMING.append((tag + "conv3x3_wino_kernel<train>", 2.0 * N * H * W * cout * Cx * 9, e0, e1,
                       "n%d %dx%d cin%d cout%d taps3x3 s1" % (N, H, W, Cx, cout)))
    return (y, stats) if want_stats else y


def conv3x3_halo_train(x, pc, want_stats=False, flip=False, accumulate_into=None):
    """3x3 stride-1 zero-padded bias-free conv on the halo kernel (ResNet trunk).  pc: PackedConv whose K side matches
    x's channels (the transposed pack + flip=True gives the backward-data).  Returns y or (y, stats)."""
    N, H, W, Cx = x.shape
    assert Cx == pc.k_pad and pc.kh == 3 and pc.kw == 3
    cout = pc.cin if pc.transpose else pc.cout
    lib = _lib.load()
    if accumulate_into is not None:
        assert tuple(accumulate_into.shape) == (N, H, W, cout) and accumulate_into.is_contiguous() and not want_stats
        y = accumulate_into
    else:
        y = torch.empty((N, H, W, cout), device=x.device, dtype=torch.float32)
    stats = None
    if want_stats:
        stats = torch.empty((lib.ccst_conv3x3_halo_stats_groups(N, H, W), cout, 2), device=x.device, dtype=torch.float32)
    flags = (_lib.CONV_FLIP if flip else 0) | (_lib.CONV_ACCUM if accumulate_into is not None else 0)
    args = (ptr(x), ptr(pc.w), ptr(y), ptr(stats), N, H, W, Cx, cout, pc.n_pad, flags, stream_ptr())
    if TIMING is None:
        check(lib.ccst_conv3x3_halo_train_f32(*args), "conv3x3_halo_train")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.ccst_conv3x3_halo_train_f32(*args), "conv3x3_halo_train")
        e1.record()
        TIMING.append((("bwd_data:" if flip else "") + "conv3x3_halo_kernel<%s,train>" % ("2,2,1" if lib.ccst_conv3x3_halo_narrow(N, H, W, cout) else "2,2,2"),
                       2.0 * N * H * W * cout * Cx * 9, e0, e1, "n%d %dx%d cin%d cout%d taps3x3 s1" % (N, H, W, Cx, cout)))
    return (y, stats) if want_stats else y

def pack_halo_split(w_oihw, w_absmax, bwd=False, out=None):
    """The half-piece weight image of a 3x3 OIHW weight for conv3x3_halo_train_split: (w_split, cout_pad, cout) of the forward conv, or
    with bwd=True of the backward-data conv dY -> dX (its `cout` is the forward Cin; run it with flip=True)."""
    cout, cin = w_oihw.shape[0], w_oihw.shape[1]
    n_out, n_in = (cin, cout) if bwd else (cout, cin)
    pad = round_up(n_out, 128)
    nfl = 9 * n_in * pad
    u = out if out is not None and out.numel() == nfl else torch.empty(nfl, device=w_oihw.device, dtype=torch.float32)
    check(_lib.load().ccst_pack_conv_weight_halo_split_f32(ptr(w_oihw.contiguous()), ptr(u), cout, cin, pad, ptr(w_absmax), int(bwd), stream_ptr()),
          "pack_conv_weight_halo_split")
    return u, pad, n_out


def halo_stats_groups(N, H, W):
    """Slabs the halo kernel's train forms leave per channel: one per (8x16-pixel tile, wave row)."""
    return int(_lib.load().ccst_conv3x3_halo_stats_groups(N, H, W))


def halo_train_split_ok(H, W, cin, cout):
    """3x3 stride-1 trunk layers the half-piece halo kernel takes: its k side a multiple of 16 (its 8x16-pixel tiles cover a 7x7 map by
    38 %, and it is still ahead of the fp32 gather kernel there)."""
    return cin % 16 == 0 and cout % 16 == 0


def conv3x3_halo_train_split(x, x_absmax, packed, w_absmax, want_stats=False, flip=False, accumulate_into=None, bn_relu=None):
    """3x3 stride-1 zero-padded bias-free conv on the half-piece halo kernel (ResNet trunk).  packed = pack_halo_split(...); the words of x
    and of the OIHW weight scale the operands.  flip=True with the bwd pack: backward-data.  Returns y or (y, stats).
    bn_relu = (bn input, mean, invstd, gamma, beta, partials out [halo_stats_groups(N, H, W), cout, 2]): y is the output gradient of that
    BatchNorm + ReLU -- stored masked, with the BatchNorm backward's partial sums."""
    u, pad, cout = packed
    N, H, W, Cx = x.shape
    lib = _lib.load()
    if accumulate_into is not None:
        assert tuple(accumulate_into.shape) == (N, H, W, cout) and accumulate_into.is_contiguous() and not want_stats
        y = accumulate_into
    else:
        y = torch.empty((N, H, W, cout), device=x.device, dtype=torch.float32)
    stats = None
    if want_stats:
        stats = torch.empty((lib.ccst_conv3x3_halo_stats_groups(N, H, W), cout, 2), device=x.device, dtype=torch.float32)
    flags = (_lib.CONV_FLIP if flip else 0) | (_lib.CONV_ACCUM if accumulate_into is not None else 0)
    bn = tuple(ptr(t) for t in bn_relu) if bn_relu is not None else (None,) * 6
    args = (ptr(x), ptr(x_absmax), ptr(u), ptr(w_absmax), ptr(y), ptr(stats), N, H, W, Cx, cout, pad, flags) + bn + (stream_ptr(),)
    if TIMING is None:
        check(lib.ccst_conv3x3_halo_train_split_f32(*args), "conv3x3_halo_train_split")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.ccst_conv3x3_halo_train_split_f32(*args), "conv3x3_halo_train_split")
        e1.record()
        TIMING.append((("bwd_data:" if flip else "") + "conv3x3_halo_kernel<%s,train,split>" % ("2,2,1" if lib.ccst_conv3x3_halo_narrow(N, H, W, cout) else "2,2,2"),
                       2.0 * N * H * W * cout * Cx * 9, e0, e1, "n%d %dx%d cin%d cout%d taps3x3 s1" % (N, H, W, Cx, cout)))
    return (y, stats) if want_stats else y


# bench.py sets TIMING = [] to collect (kernel name, algorithmic flops, start event, end event) per conv launch.
TIMING = None


def _conv_kernel_name(cout, pool, M, cin, taps=1):
    """Name of the kernel instance the C dispatcher picks (ccst_conv2d_igemm_tile)."""
    t = str(_lib.load().ccst_conv2d_igemm_tile(int(M), int(cout), int(cin), int(taps), int(bool(pool))))
    small = ""
    if len(t) == 4:          # 1xyz: the 64x64 tile (one MFMA tile row per wave)
        small, t = ",mt1" + ("" if t[3] == "1" else ",ck32"), t[1:3] + "1"
    return "conv_igemm_kernel<%s,%s,%s%s%s>" % (t[0], t[1], t[2], small, ",pool" if pool else "")


def _launch_conv(d, x, pc, out, flops, pool, what, stats=None, x_absmax=None, w_absmax=None, w_split=None):
    lib = _lib.load()
    half = x_absmax is not None and w_absmax is not None and w_split is not None and pc.bias is None and \
        bool(lib.ccst_conv2d_stream_ok(ctypes.byref(d)))

    def call():
        if half:        # (pointwise problems: half pieces on the 16-bit MFMA, x scaled by its words, the weight pre-split)
            check(lib.ccst_conv2d_pointwise_half_f32(ctypes.byref(d), ptr(x), ptr(x_absmax), ptr(w_split), ptr(w_absmax), ptr(out), ptr(stats),
                                                     None, None, None, None, None, None, None, stream_ptr()), what)
        elif stats is None:
            check(lib.ccst_conv2d_igemm_f32(ctypes.byref(d), ptr(x), ptr(pc.w), ptr(pc.bias), ptr(out), stream_ptr()), what)
        else:
            check(lib.ccst_conv2d_igemm_stats_f32(ctypes.byref(d), ptr(x), ptr(pc.w), ptr(pc.bias), ptr(out), ptr(stats),
                                                  stream_ptr()), what)
    if TIMING is None:
        call()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call()
    e1.record()
    TIMING.append((_conv_kernel_name(pc.cout, pool, d.n * d.ho * d.wo, d.cin, d.nky * d.nkx) + ("_h" if half else ""), flops, e0, e1,
                   "n%d %dx%d cin%d cout%d taps%dx%d flags%d" % (d.n, d.ho, d.wo, d.cin, d.cout, d.nky, d.nkx, d.flags)))


def _require_cuda(t, name="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("ccst_amd: %s must be a CUDA (ROCm) tensor -- the HIP path has no CPU fallback" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("ccst_amd: %s must be float32, got %s" % (name, t.dtype))


def round_up(a, b):
    return (a + b - 1) // b * b


# ---------------------------------------------------------------------------
# layout plumbing
# ---------------------------------------------------------------------------
def to_api(y_nhwc):
    """[N,H,W,C] NHWC buffer -> logical NCHW view (channels_last strides), no copy."""
    return y_nhwc.permute(0, 3, 1, 2)


def from_api(x, cpad=1):
    """Logical NCHW tensor -> NHWC-contiguous [N,H,W,Cp] (Cp = C rounded up to cpad, zero filled).
    Zero-copy if x is already channels_last with C % cpad == 0."""
    _require_cuda(x, "input")
    assert x.dim() == 4
    N, C, H, W = x.shape
    Cp = round_up(C, cpad)
    v = x.permute(0, 2, 3, 1)
    if Cp == C and v.is_contiguous():
        return v
    if x.requires_grad and torch.is_grad_enabled():
        raise RuntimeError("ccst_amd: a differentiable input must already be channels_last (NHWC in memory) with a "
                           "channel count the kernel accepts; the layout conversion kernels are not differentiable")
    x = as_nchw_contiguous(x)
    y = torch.empty((N, H, W, Cp), device=x.device, dtype=torch.float32)
    check(_lib.load().ccst_nchw_to_nhwc_f32(ptr(x), ptr(y), N, C, H * W, Cp, stream_ptr()), "nchw_to_nhwc")
    return y


# ---------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------
class _PackOrder(object):
    """Cross-stream ordering of lazily packed layouts: a layout is packed by a kernel on the stream that is current at its
    first use; the Python slot is set at once, so a use on ANOTHER stream (style._style_transfer_two_streams: the side half finds the
    slot filled by the main half) could launch its conv before that pack kernel has run.  `packed(slot)` records an event behind the
    pack launch; `use(slot)` makes any other stream wait for it (a completed event costs next to nothing)."""
    __slots__ = ("ev",)

    def __init__(self):
        self.ev = {}

    def packed(self, slot):
        e = torch.cuda.Event()
        e.record()
        self.ev[slot] = (e, torch.cuda.current_stream())

    def use(self, slot):
        # (a use on the packing stream is ordered by the stream itself; any other stream waits for the event EVERY time -- remembering
        #  "this stream has waited" by raw handle would skip the wait for a new stream that recycled a destroyed one's handle)
        ent = self.ev.get(slot)
        if ent is not None:
            cur = torch.cuda.current_stream()
            if cur != ent[1]:
                cur.wait_event(ent[0])


class PackedConv(object):
    """A conv weight in the layouts the kernels read (+ optional bias).  `w`: [kh*kw][K/4][n_pad][4] for the implicit-GEMM kernels.
    A weight packed with wino=... (the 3x3 layers of the AdaIN plan) keeps the OIHW source and builds each kernel's layout ON FIRST USE
    -- `uf43` (F(4,3) along x on half pieces), `wsplit` (direct kernel on half pieces), `wabsmax` (the weight's |max| words: the
    half-piece kernels derive their power-of-two weight scale from them on the device), `w` (the fp32-MFMA kernels' layout) -- so a
    plan holds ONE packed copy per layer, that of the kernel it runs.  can_split() says whether the half-piece layouts exist for
    this weight without building anything."""

    def __init__(self, w, bias, cin, cout, kh, kw, k_pad, n_pad, transpose, src=None, wino=False):
        self._w, self.bias, self.cin, self.cout, self.kh, self.kw = w, bias, cin, cout, kh, kw
        self.k_pad, self.n_pad, self.transpose = k_pad, n_pad, transpose
        self.src, self.wino = src, wino          # OIHW source (kept only for lazily packed weights) and the wino= argument
        self._wsplit = self._wabsmax = self._uf43 = None
        self._order = _PackOrder()

    # ---- which layouts exist (no packing) ----
    def lazy3x3(self):             # a 3x3 forward weight packed with wino=...: its kernel's layout is built on first use
        return self.src is not None and bool(self.wino) and self.kh == 3 and self.kw == 3 and not self.transpose

    def can_split(self):
        return self.lazy3x3() and HALO_SPLIT != "0" and self.cin % 16 == 0

    # ---- the layouts, built on first use ----
    def _lazy(self, slot, nfloats, fn, what, *tail):
        t = getattr(self, slot)
        if t is None:
            t = torch.empty(int(nfloats), device=self.src.device, dtype=torch.float32)
            check(fn(ptr(self.src), ptr(t), self.cout, self.cin, *tail, stream_ptr()), what)
            setattr(self, slot, t)
            self._order.packed(slot)
        else:
            self._order.use(slot)
        return t

    @property
    def w(self):
        if self._w is None:
            self._w = torch.empty(self.kh * self.kw * self.k_pad * self.n_pad, device=self.src.device, dtype=torch.float32)
            check(_lib.load().ccst_pack_conv_weight_f32(ptr(self.src), ptr(self._w), self.cout, self.cin, self.kh, self.kw, int(self.transpose),
                                                        self.k_pad, self.n_pad, stream_ptr()), "pack_conv_weight")
            self._order.packed("_w")
        elif self.src is not None:          # (eagerly packed weights -- src is None -- were complete before the object existed)
            self._order.use("_w")
        return self._w

    @property
    def wabsmax(self):
        if self._wabsmax is None and self.can_split():
            self._wabsmax = absmax(self.src)
            self._order.packed("_wabsmax")
        else:
            self._order.use("_wabsmax")
        return self._wabsmax

    @property
    def wsplit(self):
        return self._lazy("_wsplit", 9 * self.cin * self.n_pad, _lib.load().ccst_pack_conv_weight_halo_split_f32, "pack_conv_weight_halo_split",
                          self.n_pad, ptr(self.wabsmax), 0) if self.can_split() else None

    @property
    def uf43(self):
        return self._lazy("_uf43", 18 * self.cin * self.n_pad, _lib.load().ccst_pack_conv_weight_f43_f32, "pack_conv_weight_f43",
                          self.n_pad, ptr(self.wabsmax)) if self.can_split() else None


# The 3x3 stride-1 layers of the AdaIN encoder / decoder keep their OIHW weight and pack the layout of the kernel they run on first
# use (PackedConv): the half-piece kernels (CCST_HALO_SPLIT below), or with CCST_HALO_SPLIT=0 the direct fp32-MFMA halo kernel (the
# fp32 reference of the tests).  (Retired: F(2x2) / 32-channel F(4x4) in round 5; the 64-channel F(4x4) fp32 kernel and F(2,3) along x in
# round 6 -- no default path selected them; JOURNAL.md 3.06-3.07, 3.11 keep their measurements.)
USE_WINO = True          # pack_conv_weight(..., wino=USE_WINO): the AdaIN plan's 3x3 weights are packed lazily
# The direct 3x3 kernel with every fp32 product as three products of IEEE-half pieces on the 16-bit MFMA (conv3x3_halo.hip SPLIT form:
# x = hi + lo, 22 significant bits, fp32 accumulation; 5.3x the fp32 MFMA's rate at about its accuracy -- 1e-6 of max |y| per layer, 5x
# tighter than F(4x4) Winograd in fp32) runs every 3x3 layer of the AdaIN plan by default (CCST_HALO_SPLIT=2).  Measured per layer at B=6
# 512x512 (tools/halo_layers.py) against the 64-channel F(4x4) kernel: 1.23-1.49x on the six layers whose channel counts differ (the
# short-K, many-cout-group and partly-filled-round cases of F(4x4)), 0.96-1.11x on the ten where they are equal; and over the whole
# step, interleaved A/B on three boxes: all layers on SPLIT 1426-1442 images/s, all on F(4x4) 1382-1385, the per-layer mix
# (CCST_HALO_SPLIT=1: SPLIT where Cin != Cout) 1369-1428 -- the F(4x4) launches run 2-6 % slower between SPLIT launches on some boxes.
# Range: both operands are scaled by powers of two derived ON THE DEVICE from per-tensor |max| words (absmax_words / absmax below): the
# producing kernel's epilogue leaves max |y| (conv3x3_halo_split, conv3x3_stem3_nchw, the AdaIN kernels), the consumer reads it -- any
# finite fp32 magnitude is safe, nothing synchronises.  CCST_HALO_SPLIT=0: the direct fp32-MFMA kernel everywhere.
HALO_SPLIT = os.environ.get("CCST_HALO_SPLIT", "2")
ABSMAX_WORDS = 64        # CCST_ABSMAX_WORDS of include/ccst_hip.h


_ABSMAX_POOL = {}      # (device index, raw stream) -> [zeroed [1024, ABSMAX_WORDS] int32 tensor, rows handed out]


def absmax_words(device):
    """One zeroed |max| word set ([ABSMAX_WORDS] int32) for a kernel that max-accumulates the largest |value| it writes.  Rows of a
    [1024, ABSMAX_WORDS] tensor (256 KB) zeroed by ONE fill on the current stream, handed out once each (a plan of 17 layers costs a
    sixtieth of a fill launch per forward); a row is a view, so the block lives as long as any row of it does."""
    key = (torch.cuda.current_device() if device.index is None else device.index, _lib.raw_stream())
    ent = _ABSMAX_POOL.get(key)
    if ent is None or ent[1] >= ent[0].shape[0]:
        ent = [torch.zeros((1024, ABSMAX_WORDS), device=device, dtype=torch.int32), 0]
        _ABSMAX_POOL[key] = ent
    row = ent[0][ent[1]]
    ent[1] += 1
    return row


def sample_absmax_words(device, n):
    """n consecutive zeroed |max| word sets ([n, ABSMAX_WORDS] int32, one per IMAGE of a batch) from the same pool: what the AdaIN-path
    kernels take as x_absmax / y_absmax (include/ccst_hip.h: per image, so that a sample's scale -- and bits -- never depend on its
    batch-mates)."""
    n = int(n)
    key = (torch.cuda.current_device() if device.index is None else device.index, _lib.raw_stream())
    ent = _ABSMAX_POOL.get(key)
    if ent is None or ent[1] + n > ent[0].shape[0]:
        ent = [torch.zeros((max(1024, n), ABSMAX_WORDS), device=device, dtype=torch.int32), 0]
        _ABSMAX_POOL[key] = ent
    rows = ent[0][ent[1]:ent[1] + n]
    ent[1] += n
    return rows


def absmax_samples(t):
    """The per-image |max| words [N, ABSMAX_WORDS] of a contiguous fp32 CUDA batch t [N, ...] (one streaming pass,
    ccst_absmax_samples_f32) -- for AdaIN-path tensors whose producer left none."""
    _require_cuda(t, "tensor")
    t = t if t.is_contiguous() else t.contiguous()
    N = int(t.shape[0])
    out = sample_absmax_words(t.device, N)
    if t.numel() > 0:
        check(_lib.load().ccst_absmax_samples_f32(ptr(t), N, t.numel() // N, ptr(out), stream_ptr()), "absmax_samples")
    return out


def _sample_words(x, words):
    """The per-image words of batch x: `words` if they are [N, ABSMAX_WORDS] (a producer's), else computed by one pass."""
    if words is not None and words.numel() == x.shape[0] * ABSMAX_WORDS:
        return words
    if words is not None:
        raise ValueError("ccst_amd.ops: the AdaIN-path kernels take PER-IMAGE |max| words [N, %d] (got %d words for a batch of %d)"
                         % (ABSMAX_WORDS, words.numel(), x.shape[0]))
    return absmax_samples(x)


def reset_absmax_pool():
    """Forget the current blocks: the next absmax_words() zero-fills a fresh one.  fed._GraphedTrainStep calls it on both sides of a
    capture, so that a captured step takes its rows from a block whose zero fill is part of the graph (re-zeroed by every replay:
    words only grow, and a row zeroed once before the capture would freeze at the largest value any replay ever saw) and eager code
    never hands out rows of the graph's private block."""
    _ABSMAX_POOL.clear()


def absmax(t, out=None):
    """The |max| words of a contiguous fp32 CUDA tensor (one streaming pass, ccst_absmax_f32) -- for tensors whose producer left none."""
    _require_cuda(t, "tensor")
    t = t if t.is_contiguous() else t.contiguous()
    if out is None:
        out = absmax_words(t.device)
    if t.numel() > 0:
        check(_lib.load().ccst_absmax_f32(ptr(t), t.numel(), ptr(out), stream_ptr()), "absmax")
    return out


def tag_absmax(t, words):
    """Remember the |max| words a kernel left for tensor t (checked against t's version counter when they are used)."""
    t._ccst_absmax = (words, t._version)
    return t


def carry_absmax(src, dst):
    """dst is a view of src (to_api / from_api permutes): its |max| words are the same."""
    words = tagged_absmax(src)
    if words is not None:
        tag_absmax(dst, words)
    return dst


def tagged_absmax(t):
    tag = getattr(t, "_ccst_absmax", None)
    return tag[0] if tag is not None and tag[1] == t._version else None


def halo_split_wanted(pc):
    if not pc.can_split():
        return False
    return HALO_SPLIT == "2" or pc.cin != pc.cout


# Winograd F(4,3) along x on the half pieces (conv3x3_f43.hip: 1.5 instead of 3.0 executed MFMA FLOPs per algorithmic FLOP) for the layers
# of the plan whose grid of 8x32-pixel workgroups (x 128 channels, ONE per CU; Cout <= 64: x 64 channels, two per CU) is worth a launch;
# the others stay on the direct half-piece kernel.  CCST_CONV_F43=0: direct everywhere; 2: F(4,3) whatever the grid (tests: small images).
F43 = os.environ.get("CCST_CONV_F43", "1") != "0"
F43_FORCE = os.environ.get("CCST_CONV_F43", "1") == "2"
F43_MIN_TILES = 30      # per image
F43_MIN_COUT = 64       # (128 would leave the Cout = 64 layers on the direct half-piece kernel: 852 against 674 us for the three)
_N_CU = {}


def num_cus(device):
    idx = torch.cuda.current_device() if device.index is None else device.index
    if idx not in _N_CU:
        _N_CU[idx] = int(torch.cuda.get_device_properties(idx).multi_processor_count) or 256
    return _N_CU[idx]


def f43_wanted(pc, N, H, W, device):
    """Run this 3x3 layer (conv extent H x W) on the F(4,3) kernel?  The rule looks at ONE image's tiles, never at the batch size: a
    sample must not change kernels (and with them its rounding, at the 1e-5 level after sixteen layers) with the number of its
    batch-mates."""
    if not (F43 and halo_split_wanted(pc)) or pc.cout < F43_MIN_COUT or H * W * pc.cin >= 2 ** 30:
        return False
    if F43_FORCE:
        return True
    # (the bench's smallest layer -- 64 x 64, 512 -> 256: 32 tiles per image, 0.75 rounds of the chip at B = 6 -- is x1.3 the direct kernel)
    return int(_lib.load().ccst_conv3x3_f43_workgroups(1, H, W, pc.cout)) >= F43_MIN_TILES


def conv3x3_f43(x, pc, flags, sums=False, x_absmax=None, y_absmax=None, affine=None):
    """3x3 stride-1 pad-1 conv as Winograd F(4,3) along x on half pieces (conv3x3_f43.hip); same arguments and results as
    conv3x3_halo_split (sums: per-(tile, position group) centred records [ccst_conv3x3_f43_tiles, Cout, 4] = (sum, M2, count, max |y|)).
    affine = (a, b), [N, Cin] each: the conv of a * x + b (the fused AdaIN step, adain_fold_affine); x_absmax must then be the words of
    the MAPPED tensor."""
    N, Hs, Ws, Cx = x.shape
    if affine is not None:
        if x_absmax is None or (flags & (CONV_POOL2 | CONV_UPS2)) or not (flags & CONV_REFLECT):
            raise ValueError("ccst_amd.ops: a fused input affine needs the mapped tensor's words and a plain reflection-padded conv (no pool / upsample)")
        assert affine[0].numel() == N * Cx and affine[1].numel() == N * Cx and affine[0].is_contiguous() and affine[1].is_contiguous()
    x_absmax = _sample_words(x, x_absmax)
    assert y_absmax is None or y_absmax.numel() == N * ABSMAX_WORDS, "y_absmax: per-image words [N, ABSMAX_WORDS] (sample_absmax_words)"
    ups, pool = bool(flags & CONV_UPS2), bool(flags & CONV_POOL2)
    Hi, Wi = (2 * Hs, 2 * Ws) if ups else (Hs, Ws)
    oh, ow = ((Hi + 1) // 2, (Wi + 1) // 2) if pool else (Hi, Wi)
    out = torch.empty((N, oh, ow, pc.cout), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    part = None
    if sums:
        if pool:
            raise ValueError("ccst_amd.ops: the statistics epilogue is of the un-pooled output")
        part = torch.empty((int(lib.ccst_conv3x3_f43_tiles(N, Hi, Wi)), pc.cout, 4), device=x.device, dtype=torch.float32)
        part._ccst_f43 = True           # (records with the channel maxima: what adain_fold_affine needs, see f43_records)
        part._ccst_nonneg = bool(flags & CONV_RELU)
    args = (ptr(x), ptr(x_absmax), ptr(pc.uf43), ptr(pc.wabsmax), ptr(pc.bias), ptr(out), ptr(y_absmax), N, Hi, Wi, Cx,
            pc.cout, pc.n_pad, flags, ptr(part), ptr(None if affine is None else affine[0]), ptr(None if affine is None else affine[1]), stream_ptr())
    if TIMING is None:
        check(lib.ccst_conv3x3_f43_f32(*args), "conv3x3_f43")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.ccst_conv3x3_f43_f32(*args), "conv3x3_f43")
        e1.record()
        # (the kernel buckets of bench.py = rocprofv3's kernel names: the 64-channel tile is an instantiation of its own)
        TIMING.append(("conv3x3_f43_kernel<%s%s>" % ("pool" if pool else "nopool", ",half" if pc.cout <= 64 else ""), 2.0 * N * Hi * Wi * pc.cout * pc.cin * 9, e0, e1,
                       "n%d %dx%d cin%d cout%d taps3x3 flags%d" % (N, Hi, Wi, pc.cin, pc.cout, flags)))
    return (out, part) if sums else out


def conv3x3_halo_split(x, pc, flags, sums=False, x_absmax=None, y_absmax=None):
    """3x3 stride-1 pad-1 conv on the direct kernel's SPLIT form; x NHWC [N,Hs,Ws,Cin], pc packed with wino=4 (which also builds
    pc.wsplit).  sums=True (no pool): also the per-tile (sum, sum of squares) partials [tiles, Cout, 2] of the output.
    x_absmax: the per-image |max| words [N, ABSMAX_WORDS] of x left by its producer (None: one extra pass over x computes them);
    y_absmax: zeroed per-image words (sample_absmax_words) that receive max |out| for the next layer."""
    N, Hs, Ws, Cx = x.shape
    x_absmax = _sample_words(x, x_absmax)
    assert y_absmax is None or y_absmax.numel() == N * ABSMAX_WORDS, "y_absmax: per-image words [N, ABSMAX_WORDS] (sample_absmax_words)"
    ups, pool = bool(flags & CONV_UPS2), bool(flags & CONV_POOL2)
    Hi, Wi = (2 * Hs, 2 * Ws) if ups else (Hs, Ws)
    oh, ow = ((Hi + 1) // 2, (Wi + 1) // 2) if pool else (Hi, Wi)
    out = torch.empty((N, oh, ow, pc.cout), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    part = None
    if sums:
        if pool:
            raise ValueError("ccst_amd.ops: the statistics epilogue is of the un-pooled output")
        part = torch.empty((int(lib.ccst_conv3x3_halo_split_tiles(N, Hi, Wi)), pc.cout, 4), device=x.device, dtype=torch.float32)
    args = (ptr(x), ptr(x_absmax), ptr(pc.wsplit), ptr(pc.wabsmax), ptr(pc.bias), ptr(out), ptr(y_absmax), N, Hi, Wi, Cx, pc.cout, pc.n_pad, flags,
            ptr(part), stream_ptr())
    if TIMING is None:
        check(lib.ccst_conv3x3_halo_split_f32(*args), "conv3x3_halo_split")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.ccst_conv3x3_halo_split_f32(*args), "conv3x3_halo_split")
        e1.record()
        TIMING.append(("conv3x3_halo_split_kernel<%s>" % ("pool" if pool else "nopool"), 2.0 * N * Hi * Wi * pc.cout * pc.cin * 9, e0, e1,
                       "n%d %dx%d cin%d cout%d taps3x3 flags%d" % (N, Hi, Wi, pc.cin, pc.cout, flags)))
    return (out, part) if sums else out


def pack_conv_weight(w_oihw, bias=None, transpose=False, out=None, wino=False):
    """OIHW checkpoint tensor -> PackedConv.  transpose=True builds the backward-data operand.  wino (True or 4; 3x3 weights):
    nothing is packed here -- the PackedConv keeps the source and builds the layout of whichever kernel ends up running the layer
    (the fp32-MFMA one, the half-piece direct kernel, F(4,3) on half pieces) on first use."""
    _require_cuda(w_oihw, "weight")
    w = w_oihw.contiguous()
    cout, cin, kh, kw = w.shape
    kdim, ndim = (cout, cin) if transpose else (cin, cout)
    k_pad, n_pad = round_up(kdim, 16), round_up(ndim, 128)
    b = None
    if bias is not None:
        _require_cuda(bias, "bias")
        b = bias.detach().contiguous()
    if wino and kh == 3 and kw == 3 and not transpose and out is None:
        return PackedConv(None, b, cin, cout, kh, kw, k_pad, n_pad, transpose, src=w.detach(), wino=wino)
    if out is None:
        out = torch.empty(kh * kw * k_pad * n_pad, device=w.device, dtype=torch.float32)
    check(_lib.load().ccst_pack_conv_weight_f32(ptr(w), ptr(out), cout, cin, kh, kw, int(transpose), k_pad, n_pad,
                                                stream_ptr()), "pack_conv_weight")
    return PackedConv(out, b, cin, cout, kh, kw, k_pad, n_pad, transpose)


def pack_conv_weight_split(w_oihw, w_absmax, transpose=False, out=None):
    """An OIHW weight in the implicit-GEMM kernels' packed layout [tap][K/4][n_pad], PRE-SPLIT into half pieces scaled by the power of two
    of its |max| words (ccst_pack_conv_weight_split_f32): the `w_split` of conv2d_nhwc / nn_ops.conv_bwd_data (transpose=True)."""
    _require_cuda(w_oihw, "weight")
    w = w_oihw.contiguous()
    cout, cin, kh, kw = w.shape
    kdim, ndim = (cout, cin) if transpose else (cin, cout)
    k_pad, n_pad = round_up(kdim, 16), round_up(ndim, 128)
    if out is None or out.numel() != kh * kw * k_pad * n_pad:
        out = torch.empty(kh * kw * k_pad * n_pad, device=w.device, dtype=torch.float32)
    check(_lib.load().ccst_pack_conv_weight_split_f32(ptr(w), ptr(w_absmax), ptr(out), cout, cin, kh * kw, int(transpose), k_pad, n_pad,
                                                      stream_ptr()), "pack_conv_weight_split")
    return out


def chan_sums_finalize(partials):
    """[K, C, 2] per-tile (sum, sum of squares) pairs -> ([1,C,1,1] sum, [1,C,1,1] sqsum), folded in fp64 in a fixed order."""
    K, C, F = partials.shape
    s = torch.empty((1, C, 1, 1), device=partials.device, dtype=torch.float32)
    q = torch.empty((1, C, 1, 1), device=partials.device, dtype=torch.float32)
    if TIMING is None:
        check(_lib.load().ccst_chan_sums_finalize_f32(ptr(partials), F, K, C, ptr(s), ptr(q), stream_ptr()), "chan_sums_finalize")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().ccst_chan_sums_finalize_f32(ptr(partials), F, K, C, ptr(s), ptr(q), stream_ptr()), "chan_sums_finalize")
        e1.record()
        TIMING.append(("chan_sums_finalize", 0.0, e0, e1, "k%d c%d bytes%d" % (K, C, 8 * K * C)))
    return s, q


def conv_sums_ok(pc, stride, pad, pool, out_nchw):
    """Can this conv leave the per-channel records of its output in its epilogue (the half-piece kernels, un-pooled)?"""
    if stride != 1 or pad != 1 or pool or out_nchw or pc.kh != 3 or pc.kw != 3:
        return False
    return halo_split_wanted(pc)


def conv2d_nhwc(x, pc, stride=1, pad=0, reflect=False, relu=False, pool=False, ups=False, out_nchw=False, out=None,
                want_stats=False, chan_sums=False, x_absmax=None, y_absmax=None, w_absmax=None, w_split=None, affine=None):
    """Forward convolution of an NHWC tensor x [N,Hs,Ws,Cin_pad] with PackedConv pc.

    ups:  x is read through a nearest x2 upsample (logical input is [2Hs,2Ws]).
    pool: fused MaxPool2d(2,2,ceil_mode=True) epilogue.
    out_nchw: write a contiguous NCHW tensor (the image edge of the decoder).
    want_stats: also return the per-64-row (sum, sum^2) partials [groups, Cout, 2] of the output, produced in the
                conv epilogue for the following BatchNorm2d (returns (out, stats)).
    x_absmax / y_absmax: |max| words of x (from its producer) / zeroed words for max |out| -- used by the half-piece (SPLIT) kernel only;
                a caller that passes y_absmax must check halo_split_wanted(pc) (other kernels leave the words untouched).
    w_absmax + w_split (with x_absmax): the |max| words of the OIHW weight and its pre-split pack (pack_conv_weight_split) -- a
                pointwise problem then runs on half pieces (ccst_conv2d_pointwise_half_f32); without them the fp32 MFMA runs.
    """
    _require_cuda(x, "activation")
    assert x.is_contiguous() and x.dim() == 4 and not pc.transpose
    N, Hs, Ws, Cx = x.shape
    assert Cx == pc.k_pad, "activation has %d channels, packed weight expects %d" % (Cx, pc.k_pad)
    Hi, Wi = (2 * Hs, 2 * Ws) if ups else (Hs, Ws)
    ho = (Hi + 2 * pad - pc.kh) // stride + 1
    wo = (Wi + 2 * pad - pc.kw) // stride + 1
    assert ho > 0 and wo > 0
    d = CcstConvDesc()
    d.n, d.ho, d.wo, d.hi, d.wi = N, ho, wo, Hi, Wi
    d.cin, d.cout, d.cout_pad = pc.k_pad, pc.cout, pc.n_pad
    d.nky, d.nkx = pc.kh, pc.kw
    d.ay, d.by, d.cy = stride, 1, -pad
    d.ax, d.bx, d.cx = stride, 1, -pad
    d.tap_base, d.tap_sy, d.tap_sx = 0, pc.kw, 1
    d.xsN, d.xsH, d.xsW = Hs * Ws * Cx, Ws * Cx, Cx
    flags = (CONV_RELU if relu else 0) | (CONV_POOL2 if pool else 0) | (CONV_UPS2 if ups else 0) | \
        (CONV_REFLECT if reflect else 0)
    d.flags = flags
    oh, ow = ((ho + 1) // 2, (wo + 1) // 2) if pool else (ho, wo)
    if not reflect and pc.kh == 3 and pc.kw == 3 and stride == 1 and pad == 1 and not (relu or pool or ups or out_nchw) \
            and out is None and pc.bias is None and halo_train_ok(Hi, Wi, Cx, pc.cout):
        return conv3x3_halo_train(x, pc, want_stats=want_stats)
    if affine is not None:       # (the caller checked affine_ok: the F(4,3) kernel is the one that applies it)
        if not (halo_split_wanted(pc) and stride == 1 and pad == 1 and not out_nchw and out is None and not want_stats and not chan_sums
                and Cx == pc.cin and f43_wanted(pc, N, Hi, Wi, x.device)):
            raise ValueError("ccst_amd.ops: this conv cannot apply a fused input affine (affine_ok)")
        return conv3x3_f43(x, pc, flags, x_absmax=x_absmax, y_absmax=y_absmax, affine=affine)
    if chan_sums:       # (the caller checked conv_sums_ok)
        if not (halo_split_wanted(pc) and Cx == pc.cin and not pool):
            raise ValueError("ccst_amd.ops: this conv cannot leave channel records (conv_sums_ok)")
        if f43_wanted(pc, N, Hi, Wi, x.device):
            return conv3x3_f43(x, pc, flags, sums=True, x_absmax=x_absmax, y_absmax=y_absmax)
        return conv3x3_halo_split(x, pc, flags, sums=True, x_absmax=x_absmax, y_absmax=y_absmax)
    if halo_split_wanted(pc) and stride == 1 and pad == 1 and not out_nchw and out is None and not want_stats and Cx == pc.cin:
        if f43_wanted(pc, N, Hi, Wi, x.device):
            return conv3x3_f43(x, pc, flags, x_absmax=x_absmax, y_absmax=y_absmax)
        return conv3x3_halo_split(x, pc, flags, x_absmax=x_absmax, y_absmax=y_absmax)
    if USE_HALO and reflect and pc.kh == 3 and pc.kw == 3 and stride == 1 and pad == 1 and not out_nchw \
            and out is None and not want_stats:
        out = torch.empty((N, oh, ow, pc.cout), device=x.device, dtype=torch.float32)
        lib = _lib.load()
        args = (ptr(x), ptr(pc.w), ptr(pc.bias), ptr(out), N, Hi, Wi, Cx, pc.cout, pc.n_pad, flags, stream_ptr())
        if TIMING is None:
            check(lib.ccst_conv3x3_halo_f32(*args), "conv3x3_halo")
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.ccst_conv3x3_halo_f32(*args), "conv3x3_halo")
            e1.record()
            TIMING.append(("conv3x3_halo_kernel<%s%s>" % ("2,2,1" if lib.ccst_conv3x3_halo_narrow(N, Hi, Wi, pc.cout) else "2,2,2",
                                                           ",pool" if pool else ""),
                           2.0 * N * ho * wo * pc.cout * pc.cin * 9, e0, e1,
                           "n%d %dx%d cin%d cout%d taps3x3 flags%d" % (N, ho, wo, pc.cin, pc.cout, flags)))
        return out
    if out_nchw:
        assert not pool
        if out is None:
            out = torch.empty((N, pc.cout, oh, ow), device=x.device, dtype=torch.float32)
        d.y_off, d.ysN, d.ysH, d.ysW, d.ysC = 0, pc.cout * oh * ow, ow, 1, oh * ow
    else:
        if out is None:
            out = torch.empty((N, oh, ow, pc.cout), device=x.device, dtype=torch.float32)
        d.y_off, d.ysN, d.ysH, d.ysW, d.ysC = 0, oh * ow * pc.cout, ow * pc.cout, pc.cout, 1
    stats = None
    if want_stats:
        assert not (relu or pool or out_nchw), "statistics are of the raw dense NHWC conv output"
        groups = _lib.load().ccst_conv2d_igemm_stats_groups(N * ho * wo, pc.cout, pc.k_pad, pc.kh * pc.kw)
        stats = torch.empty((groups, pc.cout, 2), device=x.device, dtype=torch.float32)
    _launch_conv(d, x, pc, out, 2.0 * N * ho * wo * pc.cout * pc.cin * pc.kh * pc.kw, pool, "conv2d_igemm", stats,
                 x_absmax, w_absmax, w_split)
    return (out, stats) if want_stats else out


# The decoder's image edge (64 -> 3, NHWC in, NCHW out) as a 1x1 convolution to 9 * Cout tap planes on the 16-bit MFMA + nine shifted adds
# (conv3x3_zform.hip): Cin 32 / 64, Cout <= 3.
def zform_wanted(cin, cout):
    return cin in (32, 64) and 1 <= cout <= 3


class PackedZform:
    """The tap-plane weight of conv3x3_zform_nchw: |max| words and fragment-order half pieces of a [3][3][Cout][Cin] weight (built on first use)."""

    def __init__(self, w_tap_co_ci):
        self.w = w_tap_co_ci
        self._packed = None
        self._order = _PackOrder()

    def get(self):
        if self._packed is not None:
            self._order.use("packed")
        else:
            w = self.w
            cout, cin = int(w.shape[2]), int(w.shape[3])
            lib = _lib.load()
            words = absmax(w, out=torch.zeros(ABSMAX_WORDS, device=w.device, dtype=torch.int32))
            packed = torch.empty(int(lib.ccst_conv3x3_zform_weight_floats(cin)), device=w.device, dtype=torch.float32)
            check(lib.ccst_pack_conv_weight_zform_f32(ptr(w), ptr(words), ptr(packed), cin, cout, stream_ptr()), "pack_conv_weight_zform")
            self._packed = (packed, words)
            self._order.packed("packed")
        return self._packed


def conv3x3_zform_nchw(x, pz, bias, cout, reflect=True, relu=False, x_absmax=None):
    """3x3 stride-1 conv with <= 3 output channels, NHWC in, contiguous NCHW out (ccst_conv3x3_zform_f32); x_absmax: the |max| words of x
    (one extra pass if None)."""
    _require_cuda(x, "activation")
    assert x.is_contiguous() and x.dim() == 4
    N, H, W, Cin = x.shape
    if x_absmax is None:
        x_absmax = tagged_absmax(x)
    x_absmax = _sample_words(x, x_absmax)
    packed, words = pz.get()
    out = torch.empty((N, cout, H, W), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    args = (ptr(x), ptr(x_absmax), ptr(packed), ptr(words), ptr(bias), ptr(out), N, H, W, Cin, cout, int(reflect), int(relu), stream_ptr())
    if TIMING is None:
        check(lib.ccst_conv3x3_zform_f32(*args), "conv3x3_zform")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.ccst_conv3x3_zform_f32(*args), "conv3x3_zform")
        e1.record()
        TIMING.append(("conv3x3_zform_kernel<%d>" % cout, 2.0 * N * H * W * cout * Cin * 9, e0, e1,
                       "n%d %dx%d cin%d cout%d taps3x3" % (N, H, W, Cin, cout)))
    return out


def stem_virtual_weight(w_oihw):
    """[Cout,C<=4,kh,kw] -> [Cout, KWP*4, kh, 1]: the kx taps of a row become channels of a
    'virtual pixel' of the NHWC4 image (K per row = KWP*4, a multiple of 16)."""
    cout, c, kh, kw = w_oihw.shape
    assert c <= 4
    kwp = round_up(kw * 4, 16) // 4
    wv = torch.zeros((cout, kwp, 4, kh), device=w_oihw.device, dtype=torch.float32)
    wv[:, :kw, :c, :] = w_oihw.permute(0, 3, 1, 2)
    return wv.reshape(cout, kwp * 4, kh, 1), kwp


def conv2d_stem_nchw(x_nchw, pc_virtual, kwp, kw, stride=1, pad=0, reflect=False, relu=False):
    """Small-Cin (<=4) stem on a contiguous NCHW image: pad+transpose to NHWC4 once, then an
    implicit GEMM whose K runs over (ky, kx*4+ci).  Returns NHWC [N,ho,wo,Cout]."""
    _require_cuda(x_nchw, "image")
    x = as_nchw_contiguous(x_nchw)
    N, C, H, W = x.shape
    kh = pc_virtual.kh
    ho = (H + 2 * pad - kh) // stride + 1
    wo = (W + 2 * pad - kw) // stride + 1
    Hp = H + 2 * pad
    Wp = max(W + 2 * pad, (wo - 1) * stride + kwp)
    xp = torch.empty((N, Hp, Wp, 4), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    check(lib.ccst_nchw_to_nhwc4_pad_f32(ptr(x), ptr(xp), N, C, H, W, pad, Wp, int(reflect), stream_ptr()), "nchw_to_nhwc4_pad")
    d = CcstConvDesc()
    d.n, d.ho, d.wo, d.hi, d.wi = N, ho, wo, Hp, Wp
    d.cin, d.cout, d.cout_pad = pc_virtual.k_pad, pc_virtual.cout, pc_virtual.n_pad
    d.nky, d.nkx = kh, 1
    d.ay, d.by, d.cy = stride, 1, 0
    d.ax, d.bx, d.cx = stride, 0, 0
    d.tap_base, d.tap_sy, d.tap_sx = 0, 1, 0
    d.xsN, d.xsH, d.xsW = Hp * Wp * 4, Wp * 4, 4
    d.flags = CONV_RELU if relu else 0
    out = torch.empty((N, ho, wo, pc_virtual.cout), device=x.device, dtype=torch.float32)
    d.y_off, d.ysN, d.ysH, d.ysW, d.ysC = 0, ho * wo * pc_virtual.cout, wo * pc_virtual.cout, pc_virtual.cout, 1
    _launch_conv(d, xp, pc_virtual, out, 2.0 * N * ho * wo * pc_virtual.cout * C * kh * kw, False, "conv2d_igemm(stem)")
    return out


STEM3 = True      # the dedicated first-layer kernel (conv_stem3.hip); False: the generic stem path


def pack_stem3(w_oihw, bias=None):
    """[64,3,3,3] weight (+ bias [64]) -> the A operands of conv_stem3.hip."""
    w = w_oihw.contiguous()
    b = None if bias is None else bias.contiguous()
    wa = torch.empty(18 * 2 * 64, device=w.device, dtype=torch.float32)
    check(_lib.load().ccst_pack_stem3_weight_f32(ptr(w), ptr(b), ptr(wa), int(w.shape[0]), stream_ptr()), "pack_stem3")
    return wa


def conv3x3_stem3_nchw(x_nchw, wa, relu=True, y_absmax=None):
    """ReflectionPad2d(1) + Conv2d(3,64,3x3) (+ReLU) on a contiguous NCHW image -> NHWC [N,H,W,64].  y_absmax: zeroed per-image |max|
    words [N, ABSMAX_WORDS] (sample_absmax_words) that receive max |out| for a half-piece conv that follows."""
    _require_cuda(x_nchw, "image")
    x = as_nchw_contiguous(x_nchw)
    N, C, H, W = x.shape
    assert C == 3
    out = torch.empty((N, H, W, 64), device=x.device, dtype=torch.float32)
    assert y_absmax is None or y_absmax.numel() == N * ABSMAX_WORDS, "y_absmax: per-image words [N, ABSMAX_WORDS] (sample_absmax_words)"
    # the kernel addresses its output through one 32-bit buffer resource: slices of < 2^31 bytes along the batch
    per = max(1, (2 ** 31 - 1) // (H * W * 64 * 4))
    lib = _lib.load()
    for n0 in range(0, N, per):
        n = min(per, N - n0)
        args = (ptr(x[n0:]), ptr(wa), ptr(out[n0:]), n, H, W, int(relu), ptr(None if y_absmax is None else y_absmax.reshape(N, ABSMAX_WORDS)[n0:]), stream_ptr())
        if TIMING is None:
            check(lib.ccst_conv3x3_stem3_f32(*args), "conv3x3_stem3")
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.ccst_conv3x3_stem3_f32(*args), "conv3x3_stem3")
            e1.record()
            TIMING.append(("conv_stem3_kernel", 2.0 * n * H * W * 64 * 27, e0, e1, "n%d %dx%d cin3 cout64 taps3x3" % (n, H, W)))
    return out


# ---------------------------------------------------------------------------
# stand-alone NHWC layers (un-fused use)
# ---------------------------------------------------------------------------
def _nhwc_layer(mode, x, pad=0):
    _require_cuda(x, "activation")
    assert x.is_contiguous() and x.dim() == 4
    N, H, W, C = x.shape
    if mode == 0:
        shape = (N, H, W, C)
    elif mode == 1:
        shape = (N, H + 2 * pad, W + 2 * pad, C)
    elif mode == 2:
        shape = (N, 2 * H, 2 * W, C)
    else:
        shape = (N, (H + 1) // 2, (W + 1) // 2, C)
    y = torch.empty(shape, device=x.device, dtype=torch.float32)
    check(_lib.load().ccst_nhwc_layer_f32(mode, ptr(x), ptr(y), N, H, W, C, pad, stream_ptr()), "nhwc_layer")
    return y


def relu_nhwc(x):
    return _nhwc_layer(0, x)


def reflection_pad_nhwc(x, pad):
    return _nhwc_layer(1, x, pad)


def upsample2_nhwc(x):
    return _nhwc_layer(2, x)


def maxpool2_ceil_nhwc(x):
    return _nhwc_layer(3, x)


# ---------------------------------------------------------------------------
# AdaIN statistics
# ---------------------------------------------------------------------------
def _stats_ws(N, C, HW, device):
    nbytes = int(_lib.load().ccst_stats_workspace_bytes(N, C, HW))
    return torch.empty(nbytes, device=device, dtype=torch.uint8), nbytes


def nhwc_channel_stride(x):
    """If logical-NCHW x is a (possibly channel-sliced) view of an NHWC buffer, return its channel
    pitch Cs (>= C); else None."""
    N, C, H, W = x.shape
    sn, sc, sh, sw = x.stride()
    if sc != 1 and C > 1:
        return None
    Cs = sw if W > 1 else (sh if H > 1 else C)
    if Cs < C:
        return None
    if (W > 1 and sw != Cs) or (H > 1 and sh != W * Cs) or (N > 1 and sn != H * W * Cs):
        return None
    return Cs


def as_nchw_contiguous(x):
    """Logical-NCHW CUDA tensor -> NCHW-contiguous, using the HIP transpose when x is NHWC-strided."""
    _require_cuda(x, "input")
    if x.is_contiguous():
        return x
    Cs = nhwc_channel_stride(x)
    if Cs is None:
        raise RuntimeError("ccst_amd: tensor must be NCHW-contiguous or an NHWC (channels_last) view")
    N, C, H, W = x.shape
    out = torch.empty((N, C, H, W), device=x.device, dtype=torch.float32)
    check(_lib.load().ccst_nhwc_to_nchw_f32(ptr(x), ptr(out), N, C, H * W, Cs, stream_ptr()), "nhwc_to_nchw")
    return out


def _layout_of(feat):
    """feat: logical NCHW tensor.  Returns (buffer tensor, layout flag)."""
    _require_cuda(feat, "feature map")
    assert feat.dim() == 4
    if feat.is_contiguous():
        return feat, NCHW
    v = feat.permute(0, 2, 3, 1)
    if v.is_contiguous() and feat.shape[1] % 4 == 0:
        return v, NHWC
    return as_nchw_contiguous(feat), NCHW


def calc_mean_std(feat, eps=1e-5):
    """function.py:4-13 on a logical-NCHW CUDA tensor (either memory format)."""
    N, C, H, W = feat.shape
    buf, layout = _layout_of(feat)
    mean = torch.empty((N, C, 1, 1), device=feat.device, dtype=torch.float32)
    std = torch.empty((N, C, 1, 1), device=feat.device, dtype=torch.float32)
    ws, nb = _stats_ws(N, C, H * W, feat.device)
    check(_lib.load().ccst_calc_mean_std_f32(ptr(buf), ptr(mean), ptr(std), N, C, H * W, layout, eps, ptr(ws), nb,
                                             stream_ptr()), "calc_mean_std")
    return mean, std


def adain(feat, style_mean, style_std, alpha=1.0, eps=1e-5):
    """function.py:26-33 fused with the alpha blend (CCST_OverallStyleTransfer.py:45).
    style_mean/std: [1,C,1,1] (broadcast over N) or [N,C,1,1].  Output has feat's memory format."""
    N, C, H, W = feat.shape
    buf, layout = _layout_of(feat)
    sm = style_mean.to(device=feat.device, dtype=torch.float32).reshape(-1).contiguous()
    ss = style_std.to(device=feat.device, dtype=torch.float32).reshape(-1).contiguous()
    if sm.numel() == C and ss.numel() == C:
        per_n = 0
    elif sm.numel() == N * C and ss.numel() == N * C:
        per_n = 1
    else:
        raise RuntimeError("ccst_amd: style statistics must have C or N*C elements")
    out = torch.empty_like(buf)
    ws, nb = _stats_ws(N, C, H * W, feat.device)
    amax = sample_absmax_words(feat.device, N)      # the kernel leaves max |out| per image: the decoder's first half-piece conv scales by it
    args = (ptr(buf), ptr(sm), ptr(ss), per_n, float(alpha), ptr(out), N, C, H * W, layout, eps, ptr(ws), nb, ptr(amax), stream_ptr())
    if TIMING is None:
        check(_lib.load().ccst_adain_f32(*args), "adain")
    else:       # bench.py: the whole statistics + normalise step (3 launches), HBM-bound: algorithmic bytes = read x + write y
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().ccst_adain_f32(*args), "adain")
        e1.record()
        TIMING.append(("adain_step", 0.0, e0, e1, "n%d c%d hw%d bytes%d" % (N, C, H * W, 2 * 4 * N * C * H * W)))
    return tag_absmax(out if layout == NCHW else to_api(out), amax)


def adain_tile_sums_ok(feat, partials):
    """Can adain_from_tile_sums() take this feature map (NHWC in memory, C % 64 == 0, partials of whole images)?"""
    if partials is None or feat.dim() != 4 or not feat.is_cuda:
        return False
    N, C, H, W = feat.shape
    return C % 64 == 0 and H * W >= 2 and feat.permute(0, 2, 3, 1).is_contiguous() and partials.shape[0] % N == 0 and \
        partials.shape[1] == C and N <= 65535


def adain_from_tile_sums(feat, partials, style_mean, style_std, alpha=1.0, eps=1e-5):
    """function.py:26-33 + the alpha blend where the conv that produced feat left per-tile channel records (conv3x3_f43 /
    conv3x3_halo_split with sums=True: partials [N * tiles, C, 4]): one fold + one streaming launch, no statistics pass.  feat: logical NCHW view of an NHWC buffer."""
    N, C, H, W = feat.shape
    buf = feat.permute(0, 2, 3, 1)
    assert buf.is_contiguous() and partials.is_contiguous() and partials.shape[0] % N == 0 and partials.shape[1] == C
    sm = style_mean.to(device=feat.device, dtype=torch.float32).reshape(-1).contiguous()
    ss = style_std.to(device=feat.device, dtype=torch.float32).reshape(-1).contiguous()
    if sm.numel() == C and ss.numel() == C:
        per_n = 0
    elif sm.numel() == N * C and ss.numel() == N * C:
        per_n = 1
    else:
        raise RuntimeError("ccst_amd: style statistics must have C or N*C elements")
    out = torch.empty_like(buf)
    amax = sample_absmax_words(feat.device, N)      # the kernel leaves max |out| per image: the decoder's first half-piece conv scales by it
    stat = torch.empty((2, N * C), device=feat.device, dtype=torch.float32)      # the folded content statistics (fold kernel -> stream kernel)
    args = (ptr(buf), ptr(partials), int(partials.shape[2]), int(partials.shape[0] // N), ptr(sm), ptr(ss), per_n, float(alpha), ptr(out), N, C, H * W,
            eps, ptr(stat[0]), ptr(stat[1]), ptr(amax), stream_ptr())
    if TIMING is None:
        check(_lib.load().ccst_adain_tile_sums_f32(*args), "adain_tile_sums")
    else:       # bench.py: the AdaIN step of the path = this one launch; HBM-bound: algorithmic bytes = read x + write y
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().ccst_adain_tile_sums_f32(*args), "adain_tile_sums")
        e1.record()
        TIMING.append(("adain_step", 0.0, e0, e1, "n%d c%d hw%d bytes%d" % (N, C, H * W, 2 * 4 * N * C * H * W)))
    return tag_absmax(to_api(out), amax)


def f43_records(feat, partials):
    """Did the F(4,3) kernel write these records (its rows carry the slab's max |y| in the fourth float; the direct kernel's carry 0)?
    conv3x3_f43 marks the tensor it hands out."""
    return partials is not None and bool(getattr(partials, "_ccst_f43", False))


def adain_fold_affine(feat, partials, style_mean, style_std, alpha=1.0, eps=1e-5):
    """The AdaIN step as an affine map for the decoder's first conv (ccst_adain_fold_affine_f32): from the centred records the F(4,3)
    conv that produced feat left, ((a, b) [N, C] each, per-image words bounding |a feat + b|, (mean, std) [N, C, 1, 1]) -- one small
    launch, no pass over feat."""
    N, C, H, W = feat.shape
    assert partials.is_contiguous() and partials.shape[0] % N == 0 and partials.shape[1] == C and partials.shape[2] == 4
    sm = style_mean.to(device=feat.device, dtype=torch.float32).reshape(-1).contiguous()
    ss = style_std.to(device=feat.device, dtype=torch.float32).reshape(-1).contiguous()
    if sm.numel() == C and ss.numel() == C:
        per_n = 0
    elif sm.numel() == N * C and ss.numel() == N * C:
        per_n = 1
    else:
        raise RuntimeError("ccst_amd: style statistics must have C or N*C elements")
    buf = torch.empty((4, N, C), device=feat.device, dtype=torch.float32)       # mean | std | a | b
    words = sample_absmax_words(feat.device, N)
    args = (ptr(partials), int(partials.shape[0] // N), ptr(sm), ptr(ss), per_n, float(alpha), int(bool(getattr(partials, "_ccst_nonneg", False))),
            N, C, H * W, eps, ptr(buf[0]), ptr(buf[1]),
            ptr(buf[2]), ptr(buf[3]), ptr(words), stream_ptr())
    if TIMING is None:
        check(_lib.load().ccst_adain_fold_affine_f32(*args), "adain_fold_affine")
    else:       # bench.py: what is left of the AdaIN step -- it moves no feature bytes (bytes0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().ccst_adain_fold_affine_f32(*args), "adain_fold_affine")
        e1.record()
        TIMING.append(("adain_step", 0.0, e0, e1, "n%d c%d hw%d bytes0 fused" % (N, C, H * W)))
    return (buf[2], buf[3]), words, (buf[0].view(N, C, 1, 1), buf[1].view(N, C, 1, 1))


def interp_blend(base, content_f, weights, alpha=1.0):
    """CCST_OverallStyleTransfer.py:36-45: sum_k weights[k] * base[k] blended with content_f[0] -> [1,C,H,W] in base's memory format.
    base, content_f: [N,C,H,W] (same format), N >= len(weights)."""
    N, C, H, W = base.shape
    K = len(weights)
    if K > N or tuple(content_f.shape) != tuple(base.shape):
        raise RuntimeError("ccst_amd: %d interpolation weights for a batch of %d" % (K, N))
    bb, lb = _layout_of(base)
    cb, lc = _layout_of(content_f)
    if lb != lc:
        raise RuntimeError("ccst_amd: interp_blend operands differ in memory format")
    w = torch.tensor([float(x) for x in weights], dtype=torch.float32).to(base.device)
    out = torch.empty((1,) + tuple(bb.shape[1:]), device=base.device, dtype=torch.float32)
    check(_lib.load().ccst_interp_blend_f32(ptr(bb), ptr(cb), ptr(w), K, C * H * W, float(alpha), float(1 - alpha), ptr(out), stream_ptr()),
          "interp_blend")
    return out if lb == NCHW else to_api(out)


def chan_sums(feat):
    """calc_sum (mean_std_computation_effcientMem.py:103-115): ([1,C,1,1] sum, [1,C,1,1] sqsum, count)."""
    N, C, H, W = feat.shape
    buf, layout = _layout_of(feat)
    s = torch.empty((1, C, 1, 1), device=feat.device, dtype=torch.float32)
    q = torch.empty((1, C, 1, 1), device=feat.device, dtype=torch.float32)
    ws, nb = _stats_ws(N, C, H * W, feat.device)
    if TIMING is None:
        check(_lib.load().ccst_chan_sums_f32(ptr(buf), ptr(s), ptr(q), N, C, H * W, layout, ptr(ws), nb, stream_ptr()), "chan_sums")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().ccst_chan_sums_f32(ptr(buf), ptr(s), ptr(q), N, C, H * W, layout, ptr(ws), nb, stream_ptr()), "chan_sums")
        e1.record()
        TIMING.append(("chan_sums", 0.0, e0, e1, "n%d c%d hw%d bytes%d" % (N, C, H * W, 4 * N * C * H * W)))
    return s, q, N * H * W
