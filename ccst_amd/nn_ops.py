"""torch.autograd.Function wrappers of the ResNet training kernels (C ABI in include/ccst_hip.h).

Autograd is plumbing here: every forward/backward body is one or a few HIP launches.  Activations
are NHWC-contiguous [N,H,W,C] fp32 tensors.  Parameter gradients are accumulated by the kernels
directly into ``param.grad`` (allocated on first use, or a view of a flat arena set up by
``fed.FlatParams``) and the Functions return None for them, so autograd never runs its own
accumulation kernels.
"""
import ctypes

import torch

from . import _lib, ops
from ._lib import CcstConvDesc, check, ptr, stream_ptr

_WS = {}
import os as _os
BN_BYTE_MASK = True      # ReLU mask of the residual BNs as bytes (0: read the saved output)


def _workspace(nbytes, device):
    """Grow-only per-device scratch buffer (split-K slabs, BN partials).  Kernels on one stream run
    in order, so a single buffer is safe; it is allocated outside any graph capture at first use."""
    key = (device.index, _lib.raw_stream(device.index))
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), device=device, dtype=torch.uint8)
        _WS[key] = buf
    return buf


# Weight gradients are off the critical path of the backward pass (nothing reads dW before the optimiser step), and
# the ResNet GEMMs under-fill the chip individually, so they run on a second HIP stream, concurrently with the
# backward-data / BatchNorm chain.  The main stream re-joins at the end of the autograd pass (engine callback).
SIDE_STREAM = _os.environ.get("CCST_BWD_WEIGHT_STREAM", "1") != "0"
_SIDE = {}
_JOIN_PENDING = set()


def _side_stream(device):
    st = _SIDE.get(device.index)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SIDE[device.index] = st
    return st


# While a HIP graph is being CAPTURED, the weight-gradient launches are handed to the side stream in batches of GRAPH_FORK_BATCH: every
# fork of the side stream becomes a cross-queue dependency of the graph executor (~25 us each: a replay of the ResNet50 step with its 53
# forks ran 16.5 ms against 15.1 ms of eager device time, the ResNet18 step paid 20 of them in 4.2 ms).  A batch's launches wait for the
# main stream once; their operands (activations saved by the forward, gradients no later node modifies, |max| word rows) stay referenced
# by the pending closures until then.  The eager loop forks per conv as before (an event wait there costs nothing, and the earlier a
# weight gradient starts the more of it hides behind the backward-data chain).
GRAPH_FORK_BATCH = max(1, int(_os.environ.get("CCST_GRAPH_FORK_BATCH", "8")))
EAGER_FORK_BATCH = 1           # (measured 2 / 4 / 8 in the eager loop: ResNet50 14.49-14.67 ms against 14.57: inside the noise)
_DEFERRED = {}


def reset_deferred():
    """Forget weight-gradient launches a FAILED capture left pending (their tensors belong to an aborted step): called before a capture."""
    _DEFERRED.clear()


def _flush_deferred(device):
    pend = _DEFERRED.pop(device.index, None)
    if not pend:
        return
    main = torch.cuda.current_stream(device)
    side = _side_stream(device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _tensors, fn in pend:
            fn()
    for tensors, _fn in pend:
        for t in tensors:
            t.record_stream(side)


def _on_side_stream(device, tensors, fn):
    """Run fn() on the side stream after everything enqueued so far on the current stream; keep `tensors` alive for it."""
    batch = GRAPH_FORK_BATCH if torch.cuda.is_current_stream_capturing() else EAGER_FORK_BATCH
    if batch > 1:
        pend = _DEFERRED.setdefault(device.index, [])
        pend.append((tensors, fn))
        if len(pend) >= batch:
            _flush_deferred(device)
    else:
        main = torch.cuda.current_stream(device)
        side = _side_stream(device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fn()
        for t in tensors:
            t.record_stream(side)
    if device.index not in _JOIN_PENDING:
        _JOIN_PENDING.add(device.index)

        def join():
            _JOIN_PENDING.discard(device.index)
            _flush_deferred(device)
            torch.cuda.current_stream(device).wait_stream(_side_stream(device))
        torch.autograd.Variable._execution_engine.queue_callback(join)


_PREPACK_PENDING = set()
# The pointwise convs' TRAINING FORWARD on two IEEE-half pieces per operand (16-bit MFMA; conv_igemm.hip, CCST_CONV_BF=4): the
# BatchNorm applies leave max |y| in device words, the convs scale both operands by powers of two derived from them and from the
# weight's words (refreshed with the packed weights after each optimiser step) -- range-safe, no host synchronisation.  CCST_CONV_BF=0
# turns the words off as well (fp32 MFMA everywhere).
HALF_FWD = _os.environ.get("CCST_CONV_BF", "4") == "4"
# The weight gradients on two IEEE-half pieces per operand (conv_bwd_weight.hip, ccst_conv2d_bwd_weight_split_f32): x scaled by the |max|
# words its producing BatchNorm apply left, dy by the words the BatchNorm backward that produced it leaves (dx_absmax).  CCST_BWD_HALF=0:
# the fp32-MFMA kernel.  (Needs the forward's words, i.e. CCST_CONV_BF=4.)
HALF_BWD = HALF_FWD and _os.environ.get("CCST_BWD_HALF", "1") != "0"
# The 3x3 stride-1 trunk layers on the half-piece halo kernel (conv3x3_halo.hip, TRAIN + SPLIT form).  CCST_RESNET_HALF3X3:
#   1: backward-data (its operand is a gradient with words at hand; rounding there moves no ReLU mask);
#   2 (default): ... and the FORWARD of the layers on maps up to 28 x 28 (ResNet50 layers 2-4: ten of its thirteen stride-1 3x3 convs;
#      +3.1 % on the step).  The forward on half pieces is as accurate as the fp32 kernels (2-5e-7 of max |y| per layer, F(2x2) Winograd:
#      ~1e-6) but it is a DIFFERENT rounding, and the full-size gradient gate (tests/test_resnet_gpu.py: every gradient element within 7x
#      the reference's own fp32-vs-fp64 noise) sits behind ~50 ReLU masks: with the 56 x 56 layers on half pieces as well, ONE
#      borderline pre-activation of that fixture flips and one tensor needs 9.6x (9.0e-5 against 9.4e-6 of noise; three or four
#      products per fp32 product alike); with maps <= 28 the gate needs 5.5x, the same as with the forward on fp32;
#   3: the forward everywhere (+0.5 % more; that fixture's gate fails as described);   0: neither (fp32 F(2x2) Winograd).
HALF3X3 = int(_os.environ.get("CCST_RESNET_HALF3X3", "2")) if HALF_BWD else 0
HALF3X3_FWD_MAX_HW = 28


class GradWords(object):
    """Carries the |max| words of a conv-output gradient from the backward of the BatchNorm that produces it to the backward of the conv
    that consumes it.  One holder per (conv, BatchNorm) pair and forward pass: Conv2d.forward creates it, hangs it on its output and
    gives it to its ConvFn; the BatchNorm that reads that output gives it to its own Function, whose backward fills it (`put`); the
    conv's backward `take`s it -- only for the very tensor it was filled for (autograd may hand over a different one: a hook, an
    accumulation of several consumers' gradients)."""
    __slots__ = ("words", "ptr", "shape", "version")

    def __init__(self):
        self.words = self.ptr = self.shape = self.version = None

    def put(self, t, words):
        self.words, self.ptr, self.shape, self.version = words, t.data_ptr(), tuple(t.shape), t._version

    def take(self, t):
        # (same storage, same shape AND unmodified since: a hook that rescales the gradient in place bumps its version counter)
        words, ok = self.words, self.words is not None and self.ptr == t.data_ptr() and self.shape == tuple(t.shape) and self.version == t._version
        self.words = self.ptr = self.shape = self.version = None
        return words if ok else None


def _publish_grad_words(device, gw):
    return ops.absmax_words(device) if (HALF_BWD and gw is not None) else None


def _prepack_jobs(model, convs):
    """Device tables for the batched re-pack launches covering every packed copy the model's convs hold: the direct layouts
    (ccst_pack_conv_weights_batch_f32), the Winograd transforms (ccst_pack_conv_weights_wino_batch_f32), the weights' |max| words
    (ccst_absmax_batch_f32) and the pre-split half-piece packs scaled by them (ccst_pack_conv_weights_split_batch_f32).  Rebuilt only
    if a weight or a packed buffer moved (the key is the tuple of their addresses)."""
    slots, wslots, mslots, hslots, h3slots = [], [], [], [], []
    for m in convs:
        if m.in_channels <= 4:
            continue
        if m.__dict__.get("_ccst_wmax") is not None:          # the weight's |max| words (pointwise convs that ran a half-piece kernel)
            mslots.append(m)
        for name, transpose in (("_ccst_pk", 0), ("_ccst_pkt", 1)):
            slot = m.__dict__.get(name)
            if slot is not None:
                slots.append((m, name, slot[1], transpose))
        for name, bwd in (("_ccst_wu", 0), ("_ccst_wut", 1)):
            slot = m.__dict__.get(name)
            if slot is not None:
                wslots.append((m, name, slot[1], bwd))
        for name, transpose in (("_ccst_pkh", 0), ("_ccst_pkht", 1)):
            slot = m.__dict__.get(name)
            if slot is not None:
                hslots.append((m, name, slot[1], transpose))
        for name, transpose in (("_ccst_hh", 0), ("_ccst_hht", 1)):
            slot = m.__dict__.get(name)
            if slot is not None:
                h3slots.append((m, name, slot[1], transpose))
    sig = tuple((m.weight.data_ptr(), pc.w.data_ptr()) for m, _n, pc, _t in slots) + \
        tuple((m.weight.data_ptr(), pk[0].data_ptr()) for m, _n, pk, _b in wslots) + tuple(("wmax", m.weight.data_ptr()) for m in mslots) + \
        tuple((m.weight.data_ptr(), t.data_ptr()) for m, _n, t, _t in hslots) + tuple((m.weight.data_ptr(), pk[0].data_ptr()) for m, _n, pk, _t in h3slots)
    jobs = {"sig": sig, "table": None, "slots": slots, "wtable": None, "wslots": wslots, "mtable": None,
            "mwords": None, "mslots": mslots, "htable": None, "hslots": hslots, "h3table": None, "h3slots": h3slots}
    if not slots and not wslots:
        return jobs
    cached = model.__dict__.get("_ccst_prepack_jobs")
    if cached is not None and cached["sig"] == sig:
        return cached
    dev = (slots or wslots)[0][0].weight.device
    if slots:
        rows = [[m.weight.data_ptr(), pc.w.data_ptr(), m.out_channels, m.in_channels, pc.kh * pc.kw, t, pc.k_pad, pc.n_pad]
                for m, _n, pc, t in slots]
        jobs["table"] = torch.tensor(rows, dtype=torch.int64).to(dev)
    if wslots:
        rows = []
        for m, _n, (u, pad, n_out), bwd in wslots:
            n_in = m.out_channels if bwd else m.in_channels
            rows.append([m.weight.data_ptr(), u.data_ptr(), n_out, n_in, (n_in + 15) // 16 * 16, pad, bwd, 0])
        jobs["wtable"] = torch.tensor(rows, dtype=torch.int64).to(dev)
    if mslots:      # one batched |max| launch over the pointwise weights, into one [n, ABSMAX_WORDS] block (zeroed per step)
        jobs["mtable"] = torch.tensor([[m.weight.data_ptr(), m.weight.numel()] for m in mslots], dtype=torch.int64).to(dev)
        jobs["mwords"] = torch.zeros((len(mslots), ops.ABSMAX_WORDS), device=dev, dtype=torch.int32)
    if hslots:      # ... and the pre-split packs, scaled by those words (every conv that holds one also holds words: packed_h() asks for them)
        row_of = {id(m): i for i, m in enumerate(mslots)}
        rows = []
        for m, _n, t, transpose in hslots:
            kdim, ndim = (m.out_channels, m.in_channels) if transpose else (m.in_channels, m.out_channels)
            ntap = m.kernel_size[0] * m.kernel_size[1]
            rows.append([m.weight.data_ptr(), t.data_ptr(), m.out_channels, m.in_channels, jobs["mwords"][row_of[id(m)]].data_ptr(),
                         transpose + 2 * ntap, ops.round_up(kdim, 16), ops.round_up(ndim, 128)])
        jobs["htable"] = torch.tensor(rows, dtype=torch.int64).to(dev)
    if h3slots:     # ... and the 3x3 layers' half-piece weight images
        row_of = {id(m): i for i, m in enumerate(mslots)}
        rows = []
        for m, _n, (u, pad, n_out), transpose in h3slots:
            n_in = m.out_channels if transpose else m.in_channels
            rows.append([m.weight.data_ptr(), u.data_ptr(), n_out, n_in, pad, jobs["mwords"][row_of[id(m)]].data_ptr(), transpose, 0])
        jobs["h3table"] = torch.tensor(rows, dtype=torch.int64).to(dev)
    model.__dict__["_ccst_prepack_jobs"] = jobs
    return jobs


def prepack_on_side(model):
    """After an optimiser step: refresh every packed conv weight (into the existing buffers) with ONE batched launch per
    layout on the side stream, so it overlaps the next step's stem / first BN / max-pool instead of preceding them."""
    convs = model.__dict__.get("_ccst_convs")
    if convs is None:
        convs = model.__dict__["_ccst_convs"] = [m for m in model.modules() if hasattr(m, "prepack")]
    if not convs:
        return
    device = convs[0].weight.device
    j = _prepack_jobs(model, convs)
    if j["table"] is None and j["wtable"] is None:
        return

    def launch():
        lib = _lib.load()
        if j["table"] is not None:
            check(lib.ccst_pack_conv_weights_batch_f32(ptr(j["table"]), len(j["slots"]), stream_ptr()), "pack_weights_batch")
        if j["wtable"] is not None:
            check(lib.ccst_pack_conv_weights_wino_batch_f32(ptr(j["wtable"]), len(j["wslots"]), stream_ptr()), "pack_weights_wino_batch")
        if j["mtable"] is not None:
            check(lib.ccst_fill_f32(ptr(j["mwords"]), 0.0, j["mwords"].numel(), stream_ptr()), "zero weight |max| words")
            check(lib.ccst_absmax_batch_f32(ptr(j["mtable"]), len(j["mslots"]), ptr(j["mwords"]), stream_ptr()), "absmax_batch")
        if j["htable"] is not None:         # (after the words: the packs are scaled by them)
            check(lib.ccst_pack_conv_weights_split_batch_f32(ptr(j["htable"]), len(j["hslots"]), stream_ptr()), "pack_weights_split_batch")
        if j["h3table"] is not None:
            check(lib.ccst_pack_conv_weights_halo_split_batch_f32(ptr(j["h3table"]), len(j["h3slots"]), stream_ptr()), "pack_weights_halo_split_batch")
    if SIDE_STREAM:
        side = _side_stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            launch()
        _PREPACK_PENDING.add(device.index)
    else:
        launch()
    for m, name, pk, _t in j["slots"] + j["wslots"] + j["hslots"] + j["h3slots"]:       # the packed copies now match the weights of this epoch
        w = m.weight
        m.__dict__[name] = ((w._version, w.data_ptr(), ops.WEIGHTS_EPOCH), pk)
    for i, m in enumerate(j["mslots"]):
        w = m.weight
        m.__dict__["_ccst_wmax"] = ((w._version, w.data_ptr(), ops.WEIGHTS_EPOCH), j["mwords"][i])


def restamp_packs(model):
    """Stamp every packed copy the batched re-pack covers as current for the weights' present (version, address, epoch).  For a
    captured train step whose replay starts with that re-pack (fed._GraphedTrainStep): the step's own SGD bumped the epoch after
    the head re-pack stamped the slots, and the per-replay `prepack()` key compares must not re-pack eagerly what the replay
    re-packs itself."""
    convs = model.__dict__.get("_ccst_convs") or []
    if not convs:
        return
    j = _prepack_jobs(model, convs)
    for m, name, pk, _t in j["slots"] + j["wslots"] + j["hslots"] + j["h3slots"]:
        w = m.weight
        m.__dict__[name] = ((w._version, w.data_ptr(), ops.WEIGHTS_EPOCH), pk)
    if j["mwords"] is not None:
        for i, m in enumerate(j["mslots"]):
            w = m.weight
            m.__dict__["_ccst_wmax"] = ((w._version, w.data_ptr(), ops.WEIGHTS_EPOCH), j["mwords"][i])


def join_prepack(device):
    if device.index in _PREPACK_PENDING:
        _PREPACK_PENDING.discard(device.index)
        torch.cuda.current_stream(device).wait_stream(_side_stream(device))


def fill_(t, value=0.0):
    """t.fill_(value) as a HIP launch (contiguous fp32 tensors)."""
    check(_lib.load().ccst_fill_f32(ptr(t), float(value), t.numel(), stream_ptr()), "fill")
    return t


def _grad_slot(p):
    """Return (tensor to write, accumulate flag) for parameter p."""
    if p.grad is None:
        p.grad = fill_(torch.empty_like(p, memory_format=torch.contiguous_format))
    return p.grad


# ---------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------
def _fwd_desc(N, Hs, Ws, Cx, kh, kw, stride, pad, cin_k, cout, n_pad):
    ho = (Hs + 2 * pad - kh) // stride + 1
    wo = (Ws + 2 * pad - kw) // stride + 1
    d = CcstConvDesc()
    d.n, d.ho, d.wo, d.hi, d.wi = N, ho, wo, Hs, Ws
    d.cin, d.cout, d.cout_pad = cin_k, cout, n_pad
    d.nky, d.nkx = kh, kw
    d.ay, d.by, d.cy = stride, 1, -pad
    d.ax, d.bx, d.cx = stride, 1, -pad
    d.tap_base, d.tap_sy, d.tap_sx = 0, kw, 1
    d.xsN, d.xsH, d.xsW = Hs * Ws * Cx, Ws * Cx, Cx
    return d, ho, wo


def conv_bwd_data(dy, pc_t, x_shape, stride, pad, accumulate_into=None, relu_mask=None, bn_link=None, bn_relu=None, half=None):
    """dX of a zero-padded conv: an implicit GEMM over dY with the transposed packed weight.
    Stride 1: one launch (iy = oy + pad - ky).  Stride s: one launch per output parity class.
    accumulate_into (stride 1 only): a tensor of x's shape that already holds another gradient contribution (the
    identity branch of a residual block); the epilogue adds to it in place (CCST_CONV_ACCUM) and it is returned.
    half = (dy |max| words, weight |max| words, pre-split transposed pack): pointwise problems of the streaming kernel run on half
    pieces (ccst_conv2d_pointwise_half_f32); everything else, and half=None, on the fp32 MFMA."""
    N, H, W, Cin = x_shape
    _, Ho, Wo, Cout = dy.shape
    kh, kw = pc_t.kh, pc_t.kw
    lib = _lib.load()
    if stride == 1 and kh == 3 and kw == 3 and pad == 1 and Cout == pc_t.k_pad and ops.halo_train_ok(H, W, Cout, Cin):
        return ops.conv3x3_halo_train(dy, pc_t, flip=True, accumulate_into=accumulate_into)      # dX = conv(dY, flipped W^T)
    if stride == 1:
        if accumulate_into is not None:
            assert tuple(accumulate_into.shape) == tuple(x_shape) and accumulate_into.is_contiguous()
            dx = accumulate_into
        else:
            dx = torch.empty(x_shape, device=dy.device, dtype=torch.float32)
        classes = [(0, 0)]
    else:
        classes = [(py, px) for py in range(stride) for px in range(stride)]
        if accumulate_into is not None:     # y += over the parity classes that have taps; the others add nothing
            assert tuple(accumulate_into.shape) == tuple(x_shape) and accumulate_into.is_contiguous()
            dx = accumulate_into
        else:
            # every input pixel belongs to exactly one parity class; the buffer needs zeroing only if some class has no tap
            # (1x1 stride-2 downsample convs: three of four classes), not for 3x3 stride 2 where every class is written
            covered = all(len(range((py + pad) % stride, kh, stride)) > 0 and len(range((px + pad) % stride, kw, stride)) > 0
                          for py, px in classes)
            dx = torch.empty(x_shape, device=dy.device, dtype=torch.float32)
            if not covered:
                fill_(dx)
    for py, px in classes:
        ky0, kx0 = (py + pad) % stride, (px + pad) % stride
        nky, nkx = len(range(ky0, kh, stride)), len(range(kx0, kw, stride))
        Hc, Wc = len(range(py, H, stride)), len(range(px, W, stride))
        if nky == 0 or nkx == 0 or Hc == 0 or Wc == 0:
            continue
        d = CcstConvDesc()
        d.n, d.ho, d.wo, d.hi, d.wi = N, Hc, Wc, Ho, Wo
        d.cin, d.cout, d.cout_pad = pc_t.k_pad, Cin, pc_t.n_pad
        d.nky, d.nkx = nky, nkx
        d.ay, d.by, d.cy = 1, -1, (py + pad - ky0) // stride
        d.ax, d.bx, d.cx = 1, -1, (px + pad - kx0) // stride
        d.tap_base, d.tap_sy, d.tap_sx = ky0 * kw + kx0, stride * kw, stride
        d.xsN, d.xsH, d.xsW = Ho * Wo * Cout, Wo * Cout, Cout
        d.y_off, d.ysN, d.ysH, d.ysW, d.ysC = (py * W + px) * Cin, H * W * Cin, stride * W * Cin, stride * Cin, 1
        d.flags = _lib.CONV_ACCUM if accumulate_into is not None else 0
        use_half = half is not None and bool(lib.ccst_conv2d_stream_ok(ctypes.byref(d)))
        gather_half = half is not None and not use_half and relu_mask is None and bn_relu is None and d.cin % 32 == 0
        if gather_half:             # the gather GEMM on half pieces: parity classes of a strided 3x3 conv, strided 1x1 downsample branches
            dmax, wmax, wsp = half
            launch = lambda: check(lib.ccst_conv2d_igemm_half_f32(ctypes.byref(d), ptr(dy), ptr(dmax), ptr(wsp), ptr(wmax), ptr(dx), stream_ptr()),
                                   "conv bwd-data (gather, half pieces)")
        elif use_half:
            dmax, wmax, wsp = half
            bx, bm, bi, bg, bb, bp = (None,) * 6
            if relu_mask is not None:
                assert accumulate_into is not None and stride == 1
                if bn_link is not None:
                    bx, bm, bi, bp = bn_link
            elif bn_relu is not None:
                assert accumulate_into is None and stride == 1
                bx, bm, bi, bg, bb, bp = bn_relu
            launch = lambda: check(lib.ccst_conv2d_pointwise_half_f32(ctypes.byref(d), ptr(dy), ptr(dmax), ptr(wsp), ptr(wmax), ptr(dx), None,
                                                                      ptr(relu_mask), ptr(bx), ptr(bm), ptr(bi), ptr(bg), ptr(bb), ptr(bp), stream_ptr()),
                                   "conv bwd-data (half pieces)")
        elif relu_mask is not None:       # masked accumulate (see MaskLink); the caller checked masked_accum_ok()
            assert accumulate_into is not None and stride == 1
            bx, bm, bi, bp = bn_link if bn_link is not None else (None, None, None, None)     # (bn input, mean, invstd, partials out)
            launch = lambda: check(lib.ccst_conv2d_igemm_accum_masked_f32(ctypes.byref(d), ptr(dy), ptr(pc_t.w), ptr(dx), ptr(relu_mask),
                                                                          ptr(bx), ptr(bm), ptr(bi), ptr(bp), stream_ptr()),
                                   "conv bwd-data (masked)")
        elif bn_relu is not None:       # (bn input, mean, invstd, gamma, beta, partials out): masked by that BatchNorm's recomputed ReLU
            assert accumulate_into is None and stride == 1
            launch = lambda: check(lib.ccst_conv2d_igemm_bn_relu_bwd_f32(ctypes.byref(d), ptr(dy), ptr(pc_t.w), ptr(dx), *[ptr(t) for t in bn_relu],
                                                                         stream_ptr()), "conv bwd-data (bn+relu)")
        else:
            launch = lambda: check(lib.ccst_conv2d_igemm_f32(ctypes.byref(d), ptr(dy), ptr(pc_t.w), None, ptr(dx), stream_ptr()), "conv bwd-data")
        if ops.TIMING is None:
            launch()
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            ops.TIMING.append(("bwd_data:" + ops._conv_kernel_name(Cin, False, N * Hc * Wc, pc_t.k_pad, nky * nkx) + ("_h" if (use_half or gather_half) else ""), 2.0 * N * Hc * Wc * Cin * Cout * nky * nkx,
                               e0, e1, "n%d %dx%d cin%d cout%d taps%dx%d s%d" % (N, Hc, Wc, Cout, Cin, nky, nkx, stride)))
    return dx


def conv_bwd_weight(d, x, dy, weight_grad_oihw, accumulate=True, x_absmax=None, dy_absmax=None):
    """x_absmax + dy_absmax (the |max| words of both operands): half pieces on the 16-bit MFMA; else the fp32-MFMA kernel."""
    lib = _lib.load()
    M = d.n * d.ho * d.wo
    ntap = d.nky * d.nkx
    # (the half-piece kernel addresses x and dy with 32-bit byte offsets: operands of 2 GiB or more take the fp32-MFMA kernel)
    half = x_absmax is not None and dy_absmax is not None and x.numel() * 4 < 2 ** 31 and dy.numel() * 4 < 2 ** 31
    splits = (lib.ccst_conv2d_bwd_weight_split_splits if half else lib.ccst_conv2d_bwd_weight_splits)(M, d.cin, d.cout, ntap)
    need = splits * ntap * d.cin * d.cout * 4
    ws = _workspace(need, x.device)
    if ops.TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if half:
        check(lib.ccst_conv2d_bwd_weight_split_f32(ctypes.byref(d), ptr(x), ptr(x_absmax), ptr(dy), ptr(dy_absmax), ptr(weight_grad_oihw), splits,
                                                   int(accumulate), ptr(ws), ws.numel(), stream_ptr()), "conv bwd-weight (half pieces)")
    else:
        check(lib.ccst_conv2d_bwd_weight_f32(ctypes.byref(d), ptr(x), ptr(dy), ptr(weight_grad_oihw), splits, int(accumulate),
                                             ptr(ws), ws.numel(), stream_ptr()), "conv bwd-weight")
    if ops.TIMING is not None:
        e1.record()
        ops.TIMING.append(("bwd_weight" + ("_h" if half else ""), 2.0 * M * d.cin * d.cout * ntap, e0, e1,
                           "n%d %dx%d cin%d cout%d taps%dx%d splits%d" % (d.n, d.ho, d.wo, d.cin, d.cout, d.nky, d.nkx, splits)))


class MaskLink(object):
    """Ties the BatchNorm that closes a residual block (y = relu(bn(x) + identity), byte ReLU mask in `mask`) to the first conv of
    the NEXT block when that block's identity branch is y itself: the gradient of y is (next block's identity share) + (that
    conv's backward-data), summed in the conv's `y +=` epilogue -- which can apply y's ReLU mask to the sum while it is in
    registers.  It then sets `premasked`, and the BatchNorm's backward takes its incoming gradient as already masked: no mask
    reads in its two passes and its skip-connection share IS the incoming tensor (no masked copy: 1.06 GB of writes per ResNet50
    step).
    Contract (kept by nets/resnet.py's blocks, which create the links): every reader of y is such a conv or the identity add it
    completes -- a reader whose gradient reaches y unmasked (none exists in the reference's ResNets) would not be masked any more."""
    __slots__ = ("mask", "premasked", "bn_x", "bn_save", "partials", "gamma", "beta")

    def __init__(self, mask, bn_x=None, bn_save=None, gamma=None, beta=None):
        self.mask, self.premasked = mask, False
        # mask is None + (gamma, beta): the BatchNorm + ReLU WITHOUT a residual in front of a pointwise conv (bn2 -> conv3): that conv's
        # backward-data epilogue recomputes the mask from the BatchNorm's own input, stores the masked gradient and the partial sums
        self.gamma, self.beta = gamma, beta
        # the BatchNorm's input and saved (mean, invstd): with them that epilogue also leaves the BatchNorm backward's per-channel
        # partial sums (it has the masked gradient in registers and reads x at the same addresses), in `partials`
        self.bn_x, self.bn_save, self.partials = bn_x, bn_save, None


MASK_LINK = True
MASK_LINK_STATS = True      # BN-backward partial sums from that epilogue too


def lib_groups(M, cout, cin):
    """Partial-sum slabs the pointwise streaming kernel leaves for an M x cout result (see ccst_conv2d_igemm_stats_groups)."""
    return int(_lib.load().ccst_conv2d_igemm_stats_groups(int(M), int(cout), int(cin), 1))


def masked_accum_ok(dy, pc_t, x_shape, stride, pad):
    """True when conv_bwd_data(..., accumulate_into=..., relu_mask=...) exists for this problem (pointwise streaming kernel)."""
    if stride != 1 or pad != 0 or pc_t.kh != 1 or pc_t.kw != 1:
        return False
    N, H, W, Cin = x_shape
    Cout = dy.shape[3]
    d = CcstConvDesc()
    d.n, d.ho, d.wo, d.hi, d.wi = N, H, W, H, W
    d.cin, d.cout, d.cout_pad = pc_t.k_pad, Cin, pc_t.n_pad
    d.nky, d.nkx = 1, 1
    d.ay, d.by, d.cy = 1, -1, 0
    d.ax, d.bx, d.cx = 1, -1, 0
    d.xsN, d.xsH, d.xsW = H * W * Cout, W * Cout, Cout
    d.y_off, d.ysN, d.ysH, d.ysW, d.ysC = 0, H * W * Cin, W * Cin, Cin, 1
    d.flags = _lib.CONV_ACCUM
    return bool(_lib.load().ccst_conv2d_pointwise_ok(ctypes.byref(d)))


class GradSink(object):
    """Carries the identity-branch gradient of a residual block from the closing BatchNorm's backward to the block's
    first convolution, whose backward-data epilogue adds to it in place -- instead of autograd materialising both
    contributions and running an add kernel (16 per ResNet50 step, 3 passes over the block input each)."""
    __slots__ = ("grad", "pair")

    def __init__(self, pair=False):
        self.grad = None
        # pair=True: TWO convolutions read the same block input (the first conv and the downsample conv of a block that changes
        # shape, nets/resnet.py:156-165).  Whichever backward runs first leaves its dX here and reports no gradient; the second
        # adds to it in its epilogue and reports the sum -- instead of autograd adding the two tensors with an ATen kernel.
        self.pair = pair


class ConvFn(torch.autograd.Function):
    """Bias-free zero-padded Conv2d on NHWC (nets/resnet.py:160-161 + torchvision blocks)."""

    @staticmethod
    def forward(ctx, x, weight, mod, want_stats=False, sink=None, link=None, gw=None):
        ctx.save_for_backward(x, weight)
        ctx.gw = gw                # GradWords of this conv's output gradient (filled by the following BatchNorm's backward)
        ctx.mod = mod
        ctx.sink = sink
        ctx.link = link          # MaskLink of the ReLU that produced x (residual blocks), or None
        ctx.want_stats = bool(want_stats)
        ctx.xmax = ops.tagged_absmax(x) if HALF_BWD else None      # for the half-piece weight gradient
        ctx.set_materialize_grads(False)       # no zero tensor for the non-differentiable stats output
        ctx.wino = mod.wino_ok(x.shape[1], x.shape[2])
        ctx.half3 = HALF3X3 >= 1 and mod.halo_split_ok()         # 3x3 stride 1: backward-data (and with 2 the forward) on half pieces
        fwd_half3 = ctx.half3 and ctx.xmax is not None and (HALF3X3 >= 3 or (HALF3X3 == 2 and max(x.shape[1], x.shape[2]) <= HALF3X3_FWD_MAX_HW))
        pc = None if (ctx.wino or fwd_half3) else mod.packed()      # (the direct layout is packed only where it is used)
        if fwd_half3:
            out = ops.conv3x3_halo_train_split(x, ctx.xmax, mod.halo_h(), mod.wabsmax(), want_stats=bool(want_stats))
            if want_stats:
                ctx.mark_non_differentiable(out[1])
            return out
        if want_stats:      # BN batch statistics from the conv epilogue (non-differentiable side output)
            if ctx.wino:
                y, stats = ops.conv3x3_wino_train(x, mod.wino_fwd(), want_stats=True)
            else:
                # pointwise convs: half pieces on the 16-bit MFMA where the producer of x (a BatchNorm apply) left its |max| words and
                # the weight's are at hand (refreshed with the packed weights after every optimiser step) -- else the fp32 MFMA
                xmax = ops.tagged_absmax(x) if (HALF_FWD and mod.kernel_size == (1, 1)) else None
                y, stats = ops.conv2d_nhwc(x, pc, stride=mod.stride[0], pad=mod.padding[0], want_stats=True, x_absmax=xmax,
                                           w_absmax=mod.wabsmax() if xmax is not None else None,
                                           w_split=mod.packed_h() if xmax is not None else None)
            ctx.mark_non_differentiable(stats)
            return y, stats
        if ctx.wino:
            return ops.conv3x3_wino_train(x, mod.wino_fwd())
        xmax = ops.tagged_absmax(x) if (HALF_FWD and mod.kernel_size == (1, 1)) else None      # (the eval forward: same half-piece form, no statistics)
        return ops.conv2d_nhwc(x, pc, stride=mod.stride[0], pad=mod.padding[0], x_absmax=xmax,
                               w_absmax=mod.wabsmax() if xmax is not None else None, w_split=mod.packed_h() if xmax is not None else None)

    @staticmethod
    def backward(ctx, dy, *unused):
        if dy is None:
            return None, None, None, None, None, None, None
        x, weight = ctx.saved_tensors
        mod = ctx.mod
        dymax = ctx.gw.take(dy) if (HALF_BWD and ctx.gw is not None) else None
        dy = dy.contiguous()
        stride, pad = mod.stride[0], mod.padding[0]
        N, H, W, Cin = x.shape
        if weight.requires_grad:
            d, _, _ = _fwd_desc(N, H, W, Cin, weight.shape[2], weight.shape[3], stride, pad, Cin, weight.shape[0], 0)
            g = _grad_slot(weight)
            xmax = ctx.xmax if dymax is not None else None
            if SIDE_STREAM:
                # (the |max| word rows are views of pool blocks allocated on the main stream: they must outlive the side-stream kernel too)
                keep = (x, dy) + tuple(t for t in (xmax, dymax) if t is not None)
                _on_side_stream(x.device, keep, lambda: conv_bwd_weight(d, x, dy, g, x_absmax=xmax, dy_absmax=dymax))
            else:
                conv_bwd_weight(d, x, dy, g, x_absmax=xmax, dy_absmax=dymax)
        dx = None
        if ctx.needs_input_grad[0]:
            into, deposit = None, False
            if ctx.sink is not None:
                if ctx.sink.grad is not None:
                    into, ctx.sink.grad = ctx.sink.grad, None
                elif ctx.sink.pair:
                    deposit = True
            if ctx.half3 and dymax is not None:     # 3x3 stride 1: the half-piece halo kernel with the transposed image, taps flipped
                link, bn_relu = ctx.link, None
                # x is the output of a BatchNorm + ReLU that only this conv reads (bn1 -> conv2): mask and partial sums from the epilogue
                if link is not None and link.mask is None and link.gamma is not None and into is None and ctx.sink is None and MASK_LINK_STATS:
                    link.partials = torch.empty((ops.halo_stats_groups(N, H, W), Cin, 2), device=x.device, dtype=torch.float32)
                    bn_relu = (link.bn_x, link.bn_save[0], link.bn_save[1], link.gamma, link.beta, link.partials)
                dx = ops.conv3x3_halo_train_split(dy, dymax, mod.halo_ht(), mod.wabsmax(), flip=True, accumulate_into=into, bn_relu=bn_relu)
                if bn_relu is not None:
                    link.premasked = True
            elif ctx.wino:
                dx = ops.conv3x3_wino_train(dy, mod.wino_bwd(), accumulate_into=into, tag="bwd_data:")
            else:
                mask = None
                link = ctx.link
                # this launch completes the gradient of x (identity share + this conv): mask it by x's ReLU here (MaskLink)
                if link is not None and into is not None and not ctx.sink.pair and link.mask is not None and \
                        masked_accum_ok(dy, mod.packed_t(), tuple(x.shape), stride, pad):
                    mask = link.mask
                bn_link = bn_relu = None
                M = N * H * W
                if mask is not None and MASK_LINK_STATS and link.bn_x is not None:
                    link.partials = torch.empty((lib_groups(M, Cin, dy.shape[3]), Cin, 2), device=x.device, dtype=torch.float32)
                    bn_link = (link.bn_x, link.bn_save[0], link.bn_save[1], link.partials)
                # x is the output of a BatchNorm + ReLU that only this conv reads (bn2 -> conv3): mask and partial sums from here
                if link is not None and link.mask is None and link.gamma is not None and into is None and ctx.sink is None and \
                        MASK_LINK_STATS and masked_accum_ok(dy, mod.packed_t(), tuple(x.shape), stride, pad):
                    link.partials = torch.empty((lib_groups(M, Cin, dy.shape[3]), Cin, 2), device=x.device, dtype=torch.float32)
                    bn_relu = (link.bn_x, link.bn_save[0], link.bn_save[1], link.gamma, link.beta, link.partials)
                half = None
                if dymax is not None and mod.out_channels % 32 == 0:      # the gradient's words are at hand: half pieces (streaming or gather kernel)
                    half = (dymax, mod.wabsmax(), mod.packed_th())
                dx = conv_bwd_data(dy, mod.packed_t(), tuple(x.shape), stride, pad, accumulate_into=into, relu_mask=mask, bn_link=bn_link,
                                   bn_relu=bn_relu, half=half)
                if mask is not None or bn_relu is not None:
                    link.premasked = True
            if deposit:
                ctx.sink.grad, dx = dx, None
        return dx, None, None, None, None, None, None


class StemConvFn(torch.autograd.Function):
    """Conv2d with <= 4 input channels on the NCHW image (nets/resnet.py:136): no input gradient."""

    @staticmethod
    def forward(ctx, x_nchw, weight, mod, gw=None):
        ctx.gw = gw
        pc, kwp = mod.packed_stem()
        stride, pad, kw = mod.stride[0], mod.padding[0], weight.shape[3]
        x = ops.as_nchw_contiguous(x_nchw)
        N, C, H, W = x.shape
        kh = weight.shape[2]
        ho = (H + 2 * pad - kh) // stride + 1
        wo = (W + 2 * pad - kw) // stride + 1
        Hp, Wp = H + 2 * pad, max(W + 2 * pad, (wo - 1) * stride + kwp)
        xp = torch.empty((N, Hp, Wp, 4), device=x.device, dtype=torch.float32)
        lib = _lib.load()
        check(lib.ccst_nchw_to_nhwc4_pad_f32(ptr(x), ptr(xp), N, C, H, W, pad, Wp, 0, stream_ptr()), "nchw_to_nhwc4_pad")
        d = CcstConvDesc()
        d.n, d.ho, d.wo, d.hi, d.wi = N, ho, wo, Hp, Wp
        d.cin, d.cout, d.cout_pad = pc.k_pad, pc.cout, pc.n_pad
        d.nky, d.nkx = kh, 1
        d.ay, d.by, d.cy = stride, 1, 0
        d.ax, d.bx, d.cx = stride, 0, 0
        d.tap_base, d.tap_sy, d.tap_sx = 0, 1, 0
        d.xsN, d.xsH, d.xsW = Hp * Wp * 4, Wp * 4, 4
        d.flags = 0
        y = torch.empty((N, ho, wo, pc.cout), device=x.device, dtype=torch.float32)
        d.y_off, d.ysN, d.ysH, d.ysW, d.ysC = 0, ho * wo * pc.cout, wo * pc.cout, pc.cout, 1
        check(lib.ccst_conv2d_igemm_f32(ctypes.byref(d), ptr(xp), ptr(pc.w), None, ptr(y), stream_ptr()), "stem conv")
        ctx.save_for_backward(xp, weight)
        ctx.geom = (N, ho, wo, Hp, Wp, kh, kw, kwp, stride, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, weight = ctx.saved_tensors
        N, ho, wo, Hp, Wp, kh, kw, kwp, stride, C = ctx.geom
        dymax = ctx.gw.take(dy) if (HALF_BWD and ctx.gw is not None) else None
        if weight.requires_grad:
            cout = weight.shape[0]
            d = CcstConvDesc()
            d.n, d.ho, d.wo, d.hi, d.wi = N, ho, wo, Hp, Wp
            d.cin, d.cout = kwp * 4, cout
            d.nky, d.nkx = kh, 1
            d.ay, d.by, d.cy = stride, 1, 0
            d.ax, d.bx, d.cx = stride, 0, 0
            d.xsN, d.xsH, d.xsW = Hp * Wp * 4, Wp * 4, 4
            gv = torch.empty((cout, kwp * 4, kh, 1), device=dy.device, dtype=torch.float32)    # virtual-pixel gradient
            # (the padded image has no producer kernel that could leave its |max| words: one 53 MB pass, ~15 us, for a 3x shorter GEMM)
            conv_bwd_weight(d, xp, dy.contiguous(), gv, accumulate=False, x_absmax=ops.absmax(xp) if dymax is not None else None,
                            dy_absmax=dymax)
            # un-fold (kx, ci) <- virtual channel kx*4+ci : 9.4k floats of glue
            check(_lib.load().ccst_stem_grad_unfold_f32(ptr(gv), ptr(_grad_slot(weight)), cout, kwp, kh, kw, C, 1, stream_ptr()),
                  "stem_grad_unfold")
        return None, None, None, None


# ---------------------------------------------------------------------------
# batch norm (+ residual + ReLU)
# ---------------------------------------------------------------------------
class BNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, mod, relu, stats=None, sink=None, gw=None):
        lib = _lib.load()
        ctx.sink = sink
        ctx.gw = gw
        N, H, W, C = x.shape
        M = N * H * W
        y = torch.empty_like(x)
        if mod.training:
            ymax = ops.absmax_words(x.device) if HALF_FWD else None       # max |y| for a half-piece pointwise conv that reads y
            save = torch.empty((2, C), device=x.device, dtype=torch.float32)
            nb = int(lib.ccst_bn_workspace_bytes(M, C))
            ws = _workspace(nb, x.device)
            track = mod.track_running_stats and mod.running_mean is not None
            ctx.relu, ctx.has_res = bool(relu), residual is not None
            # ReLU mask for the backward.  With a residual it cannot be recomputed from x: the forward leaves a byte mask (4 elements
            # per byte), so both backward passes read 1/16 of a tensor instead of the whole saved output (the block outputs are the
            # 4P-channel tensors: 2 x 1.4 GB per ResNet50 step).  Without a residual it is recomputed from x in the kernels
            # (y == NULL), which measured 2373 vs 2360 img/s and keeps one tensor less alive per BN.
            mask = torch.empty((M * C // 4,), device=x.device, dtype=torch.uint8) if (ctx.relu and ctx.has_res and BN_BYTE_MASK) else None
            check(lib.ccst_bn_train_fwd_mask_f32(ptr(x), ptr(gamma), ptr(beta), ptr(mod.running_mean if track else None),
                                                 ptr(mod.running_var if track else None), float(mod.momentum), float(mod.eps),
                                                 ptr(residual), int(relu), ptr(y), ptr(mask), ptr(save[0]), ptr(save[1]), M, C, ptr(stats),
                                                 0 if stats is None else int(stats.shape[0]), ptr(ws), ws.numel(), ptr(ymax), stream_ptr()),
                  "bn_train_fwd")
            if ymax is not None:
                ops.tag_absmax(y, ymax)
            keep_y = ctx.relu and mask is None and ctx.has_res
            ctx.save_for_backward(x, y if keep_y else None, mask, gamma, beta, save)
            if mask is not None and MASK_LINK:
                ctx.link = mod._ccst_mask_link = MaskLink(mask, x, save)
            elif ctx.relu and not ctx.has_res and MASK_LINK:
                ctx.link = mod._ccst_mask_link = MaskLink(None, x, save, gamma, beta)
            else:
                ctx.link = mod._ccst_mask_link = None
        else:
            ymax = ops.absmax_words(x.device) if HALF_FWD else None       # (the eval forward's pointwise convs run on half pieces too)
            check(lib.ccst_bn_eval_fwd_f32(ptr(x), ptr(gamma), ptr(beta), ptr(mod.running_mean), ptr(mod.running_var),
                                           float(mod.eps), ptr(residual), int(relu), ptr(y), M, C, ptr(ymax), stream_ptr()), "bn_eval_fwd")
            if ymax is not None:
                ops.tag_absmax(y, ymax)
            ctx.save_for_backward()
            ctx.eval_mode = True
        return y

    @staticmethod
    def backward(ctx, dy):
        if getattr(ctx, "eval_mode", False):
            raise RuntimeError("ccst_amd: backward through eval-mode BatchNorm is not on the reference path")
        x, y, mask, gamma, beta, save = ctx.saved_tensors
        lib = _lib.load()
        N, H, W, C = x.shape
        M = N * H * W
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dxmax = _publish_grad_words(x.device, ctx.gw)       # max |dx|: the conv in front of this BatchNorm scales its backward's dy by it
        ws = _workspace(int(lib.ccst_bn_workspace_bytes(M, C)), x.device)
        link = getattr(ctx, "link", None)
        if link is not None and link.premasked:
            # the producer of dy already applied this BatchNorm's ReLU mask (MaskLink): plain backward, and the skip connection's
            # share is dy itself
            link.premasked = False
            dres = dy if ctx.has_res else None
            part, link.partials = link.partials, None
            if part is not None:        # ... and left this backward's partial sums: finalize + apply only
                check(lib.ccst_bn_train_bwd_partials_f32(ptr(dy), ptr(x), ptr(gamma), ptr(save[0]), ptr(save[1]), ptr(part), int(part.shape[0]),
                                                         ptr(dx), ptr(_grad_slot(gamma)), ptr(_grad_slot(beta)), 1, M, C, ptr(ws), ws.numel(),
                                                         ptr(dxmax), stream_ptr()), "bn_train_bwd")
            else:
                check(lib.ccst_bn_train_bwd_mask_f32(ptr(dy), ptr(x), None, None, ptr(gamma), ptr(beta), ptr(save[0]), ptr(save[1]),
                                                     0, ptr(dx), None, ptr(_grad_slot(gamma)), ptr(_grad_slot(beta)), 1, M, C,
                                                     ptr(ws), ws.numel(), ptr(dxmax), stream_ptr()), "bn_train_bwd")
        else:
            dres = torch.empty_like(x) if ctx.has_res else None
            check(lib.ccst_bn_train_bwd_mask_f32(ptr(dy), ptr(x), ptr(y), ptr(mask), ptr(gamma), ptr(beta), ptr(save[0]), ptr(save[1]),
                                                 int(ctx.relu), ptr(dx), ptr(dres), ptr(_grad_slot(gamma)), ptr(_grad_slot(beta)), 1, M, C,
                                                 ptr(ws), ws.numel(), ptr(dxmax), stream_ptr()), "bn_train_bwd")
        if dxmax is not None:
            ctx.gw.put(dx, dxmax)
        if ctx.sink is not None and dres is not None:
            ctx.sink.grad, dres = dres, None        # handed to the block's first conv (GradSink), not to autograd
        return dx, None, None, dres, None, None, None, None, None


class StemBnReluPoolFn(torch.autograd.Function):
    """BatchNorm2d (training) -> ReLU -> MaxPool2d(3, 2, 1) of the ResNet stem (nets/resnet.py:138-140) as one op: neither the
    normalised full-resolution map nor its gradient is ever stored (see ccst_bn_relu_maxpool_train_fwd_f32)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mod, stats=None, gw=None):
        lib = _lib.load()
        ctx.gw = gw
        N, H, W, C = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        M = N * H * W
        y = torch.empty((N, Ho, Wo, C), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, Ho, Wo, C // 4), device=x.device, dtype=torch.int32)
        save = torch.empty((2, C), device=x.device, dtype=torch.float32)
        ws = _workspace(int(lib.ccst_bn_workspace_bytes(M, C)), x.device)
        track = mod.track_running_stats and mod.running_mean is not None
        ymax = ops.absmax_words(x.device) if HALF_FWD else None
        check(lib.ccst_bn_relu_maxpool_train_fwd_f32(ptr(x), ptr(gamma), ptr(beta), ptr(mod.running_mean if track else None),
                                                     ptr(mod.running_var if track else None), float(mod.momentum), float(mod.eps), ptr(y), ptr(idx),
                                                     ptr(save[0]), ptr(save[1]), N, H, W, C, Ho, Wo, ptr(stats),
                                                     0 if stats is None else int(stats.shape[0]), ptr(ws), ws.numel(), ptr(ymax), stream_ptr()),
              "bn_relu_maxpool_fwd")
        if ymax is not None:
            ops.tag_absmax(y, ymax)
        ctx.save_for_backward(x, idx, gamma, beta, save)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, idx, gamma, beta, save = ctx.saved_tensors
        lib = _lib.load()
        N, H, W, C = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        dx = torch.empty_like(x)
        dxmax = _publish_grad_words(x.device, ctx.gw)
        ws = _workspace(int(lib.ccst_bn_workspace_bytes(N * H * W, C)), x.device)
        check(lib.ccst_bn_relu_maxpool_train_bwd_f32(ptr(dy.contiguous()), ptr(idx), ptr(x), ptr(gamma), ptr(beta), ptr(save[0]), ptr(save[1]),
                                                     ptr(dx), ptr(_grad_slot(gamma)), ptr(_grad_slot(beta)), 1, N, H, W, C, Ho, Wo, ptr(ws),
                                                     ws.numel(), ptr(dxmax), stream_ptr()), "bn_relu_maxpool_bwd")
        if dxmax is not None:
            ctx.gw.put(dx, dxmax)
        return dx, None, None, None, None, None


# ---------------------------------------------------------------------------
# pools, linear, loss
# ---------------------------------------------------------------------------
class MaxPool3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        N, H, W, C = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, Ho, Wo, C), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, Ho, Wo, C // 4), device=x.device, dtype=torch.int32)
        check(_lib.load().ccst_maxpool3s2_fwd_f32(ptr(x), ptr(y), ptr(idx), N, H, W, C, Ho, Wo, stream_ptr()), "maxpool_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (N, H, W, C, Ho, Wo)
        ops.carry_absmax(x, y)              # max |pooled| <= max |x|: the producer's words stay a valid bound
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, H, W, C, Ho, Wo = ctx.shape
        dx = torch.empty((N, H, W, C), device=dy.device, dtype=torch.float32)
        check(_lib.load().ccst_maxpool3s2_bwd_f32(ptr(dy.contiguous()), ptr(idx), ptr(dx), N, H, W, C, Ho, Wo, stream_ptr()),
              "maxpool_bwd")
        return dx


class AvgPoolFlattenFn(torch.autograd.Function):
    """AvgPool2d over the whole map + flatten: [N,H,W,C] -> [N,C]."""

    @staticmethod
    def forward(ctx, x):
        N, H, W, C = x.shape
        y = torch.empty((N, C), device=x.device, dtype=torch.float32)
        check(_lib.load().ccst_avgpool_fwd_f32(ptr(x), ptr(y), N, H * W, C, stream_ptr()), "avgpool_fwd")
        ctx.shape = (N, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C = ctx.shape
        dx = torch.empty((N, H, W, C), device=dy.device, dtype=torch.float32)
        check(_lib.load().ccst_avgpool_bwd_f32(ptr(dy.contiguous()), ptr(dx), N, H * W, C, stream_ptr()), "avgpool_bwd")
        return dx


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        N, K = x.shape
        O = weight.shape[0]
        y = torch.empty((N, O), device=x.device, dtype=torch.float32)
        check(_lib.load().ccst_linear_fwd_f32(ptr(x), ptr(weight), ptr(bias), ptr(y), N, K, O, stream_ptr()), "linear_fwd")
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        N, K = x.shape
        O = weight.shape[0]
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = _grad_slot(weight) if weight.requires_grad else None
        db = _grad_slot(bias) if (bias is not None and bias.requires_grad) else None
        check(_lib.load().ccst_linear_bwd_f32(ptr(x), ptr(weight), ptr(dy), ptr(dx), ptr(dw), ptr(db), 1, N, K, O, stream_ptr()),
              "linear_bwd")
        return dx, None, None


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean) with the argmax==label count of fed_run.py:67,71 as a by-product."""

    @staticmethod
    def forward(ctx, logits, labels, correct_out):
        logits = logits.contiguous()
        N, O = logits.shape
        loss = torch.empty((1,), device=logits.device, dtype=torch.float32)
        dlog = torch.empty_like(logits)
        check(_lib.load().ccst_softmax_ce_f32(ptr(logits), ptr(labels.contiguous()), ptr(loss), ptr(dlog), ptr(correct_out), N, O,
                                              stream_ptr()), "softmax_ce")
        ctx.save_for_backward(dlog)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        out = torch.empty_like(dlog)
        check(_lib.load().ccst_mul_scalar_f32(ptr(out), ptr(dlog), ptr(g.contiguous()), dlog.numel(), stream_ptr()), "ce_backward_scale")
        return out, None, None
