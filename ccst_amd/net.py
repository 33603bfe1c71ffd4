"""Drop-in for style_transfer/AdaIN/net.py: module-level ``vgg`` and ``decoder``
nn.Sequential objects with the reference's layer indices and state-dict keys
(net.py:6-36, :38-92), executed by hand-written HIP kernels.

* ``vgg`` / ``decoder`` are :class:`Sequential` instances.  Calling one (or any
  slice ``vgg[:31]``, or ``Sequential(*list(vgg.children())[:31])``) compiles a
  fused plan: ReflectionPad2d+Conv2d+ReLU(+MaxPool2d) and Upsample+Pad+Conv
  become single implicit-GEMM launches; the 1x1 colour conv (net.py:39) is
  folded into conv1_1's weights.
* The children are subclasses of the torch.nn layers whose ``forward`` also
  runs HIP kernels, so the reference idiom
  ``nn.Sequential(*list(vgg.children())[:31])`` (CCST_OverallStyleTransfer.py:124)
  still executes on this library -- un-fused and slower, same results.
* Tensors at the boundary are logical NCHW fp32 CUDA tensors; feature maps are
  returned as channels_last views of the internal NHWC buffers.

There is no CPU execution path: CPU tensors raise.
"""
import torch
import torch.nn as nn

from . import ops


# ---------------------------------------------------------------------------
# plan compiler
# ---------------------------------------------------------------------------
def _is_pad1(m):
    return isinstance(m, nn.ReflectionPad2d) and tuple(m.padding) == (1, 1, 1, 1)


def _is_conv(m, k=None):
    if not isinstance(m, nn.Conv2d):
        return False
    if m.groups != 1 or tuple(m.dilation) != (1, 1) or m.padding_mode != "zeros" or isinstance(m.padding, str):
        raise NotImplementedError("ccst_amd: unsupported Conv2d configuration %r" % (m,))
    return k is None or tuple(m.kernel_size) == (k, k)


def _is_valid3(m):
    return _is_conv(m, 3) and tuple(m.stride) == (1, 1) and tuple(m.padding) == (0, 0)


def _is_pool2(m):
    def pair(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    return isinstance(m, nn.MaxPool2d) and pair(m.kernel_size) == (2, 2) and pair(m.stride) == (2, 2) and \
        pair(m.padding) == (0, 0) and m.ceil_mode and pair(m.dilation) == (1, 1)


def _is_up2(m):
    return isinstance(m, nn.Upsample) and m.mode == "nearest" and m.scale_factor is not None and \
        float(m.scale_factor if not isinstance(m.scale_factor, (tuple, list)) else m.scale_factor[0]) == 2.0 and m.size is None


class _Step(object):
    """One launch (or launch pair) of the plan."""
    __slots__ = ("kind", "pc", "kw", "kwp", "stride", "pad", "reflect", "relu", "pool", "ups", "out_nchw",
                 "w_small", "b_small", "cout", "wa", "pz")

    def __init__(self, kind, **kw):
        self.kind = kind
        self.pc = self.kw = self.kwp = self.w_small = self.b_small = self.cout = self.wa = self.pz = None
        self.stride, self.pad = 1, 0
        self.reflect = self.relu = self.pool = self.ups = self.out_nchw = False
        for k, v in kw.items():
            setattr(self, k, v)


def _compile(mods):
    steps = []
    n = len(mods)
    i = 0
    pending_up = False

    def at(j):
        return mods[j] if j < n else None

    while i < n:
        m = mods[i]
        if _is_up2(m):
            nxt, nxt2 = at(i + 1), at(i + 2)
            if nxt is not None and _is_pad1(nxt) and nxt2 is not None and _is_valid3(nxt2) and nxt2.in_channels % 16 == 0:
                pending_up = True
            else:
                steps.append(_Step("up"))
            i += 1
            continue
        # folded colour conv: Conv2d(3,3,1x1) -> pad -> Conv2d(3,Cout,3x3)   (net.py:39-41)
        if _is_conv(m, 1) and m.in_channels <= 4 and m.out_channels <= 4 and tuple(m.stride) == (1, 1) and \
                tuple(m.padding) == (0, 0) and at(i + 1) is not None and _is_pad1(at(i + 1)) and at(i + 2) is not None and \
                _is_valid3(at(i + 2)) and at(i + 2).in_channels == m.out_channels:
            c0, c1 = m, at(i + 2)
            w0 = c0.weight.detach()[:, :, 0, 0]                       # [mid, in]
            w1 = c1.weight.detach()                                   # [out, mid, 3, 3]
            wf = torch.einsum("omyx,mi->oiyx", w1, w0).contiguous()   # W' = W o W0
            b0 = c0.bias.detach() if c0.bias is not None else torch.zeros(c0.out_channels, device=w1.device)
            bf = torch.einsum("omyx,m->o", w1, b0)
            if c1.bias is not None:
                bf = bf + c1.bias.detach()
            j = i + 3
            relu = isinstance(at(j), nn.ReLU)
            j += int(relu)
            if ops.STEM3 and tuple(wf.shape) == (64, 3, 3, 3):      # the encoder's first layer: its own kernel, NCHW image in
                steps.append(_Step("stem3", wa=ops.pack_stem3(wf, bf), relu=relu))
            else:
                wv, kwp = ops.stem_virtual_weight(wf)
                steps.append(_Step("stem", pc=ops.pack_conv_weight(wv, bf.contiguous()), kw=3, kwp=kwp, pad=1, reflect=True, relu=relu))
            i = j
            continue
        if _is_pad1(m) and at(i + 1) is not None and _is_valid3(at(i + 1)):
            conv = at(i + 1)
            j = i + 2
            relu = isinstance(at(j), nn.ReLU)
            j += int(relu)
            if conv.in_channels <= 4 and ops.STEM3 and tuple(conv.weight.shape) == (64, 3, 3, 3):
                steps.append(_Step("stem3", wa=ops.pack_stem3(conv.weight.detach(), None if conv.bias is None else conv.bias.detach()), relu=relu))
            elif conv.in_channels <= 4:
                wv, kwp = ops.stem_virtual_weight(conv.weight.detach())
                steps.append(_Step("stem", pc=ops.pack_conv_weight(wv, conv.bias), kw=3, kwp=kwp, pad=1, reflect=True, relu=relu))
            elif ops.zform_wanted(conv.in_channels, conv.out_channels) and not pending_up:
                # image edge of the decoder (net.py:35): tap planes on the 16-bit MFMA, NCHW out (conv3x3_zform.hip)
                w_small = conv.weight.detach().permute(2, 3, 0, 1).contiguous()
                steps.append(_Step("smallco", w_small=w_small, pz=ops.PackedZform(w_small),
                                   b_small=None if conv.bias is None else conv.bias.detach().contiguous(),
                                   cout=conv.out_channels, reflect=True, relu=relu))
            else:
                pool = at(j) is not None and _is_pool2(at(j)) and conv.out_channels % 16 == 0
                j += int(pool)
                steps.append(_Step("conv", pc=ops.pack_conv_weight(conv.weight.detach(), conv.bias, wino=ops.USE_WINO), pad=1, reflect=True,
                                   relu=relu, pool=pool, ups=pending_up))
                pending_up = False
            i = j
            continue
        if _is_conv(m):
            if m.kernel_size[0] != m.kernel_size[1] or m.stride[0] != m.stride[1] or m.padding[0] != m.padding[1]:
                raise NotImplementedError("ccst_amd: non-square Conv2d %r" % (m,))
            j = i + 1
            relu = isinstance(at(j), nn.ReLU)
            j += int(relu)
            if m.in_channels <= 4:
                wv, kwp = ops.stem_virtual_weight(m.weight.detach())
                steps.append(_Step("stem", pc=ops.pack_conv_weight(wv, m.bias), kw=m.kernel_size[1], kwp=kwp,
                                   stride=m.stride[0], pad=m.padding[0], relu=relu))
            else:
                steps.append(_Step("conv", pc=ops.pack_conv_weight(m.weight.detach(), m.bias), stride=m.stride[0],
                                   pad=m.padding[0], relu=relu))
            i = j
            continue
        if isinstance(m, nn.ReLU):
            steps.append(_Step("relu"))
        elif isinstance(m, nn.ReflectionPad2d):
            p = tuple(m.padding)
            if len(set(p)) != 1:
                raise NotImplementedError("ccst_amd: asymmetric ReflectionPad2d %r" % (p,))
            steps.append(_Step("pad", pad=p[0]))
        elif _is_pool2(m):
            steps.append(_Step("pool"))
        else:
            raise NotImplementedError("ccst_amd: layer %r has no HIP implementation in this path" % (m,))
        i += 1
    # image edges: a conv producing <= 4 channels at the end writes NCHW directly
    if steps and steps[-1].kind == "conv" and steps[-1].pc.cout <= 4:
        steps[-1].out_nchw = True
    for s in steps:
        if s.kind == "conv" and s.pc.cout % 16 != 0 and not s.out_nchw:
            raise NotImplementedError("ccst_amd: intermediate Conv2d with %d output channels (need a multiple of 16)" % s.pc.cout)
    return steps


# The kernels address a tensor with 32-bit element offsets (one VGPR per address, no 64-bit adds per load).  A batch whose widest
# activation (64 channels at full resolution) would not fit is run in slices of the batch dimension: every layer of this path and
# the AdaIN statistics are per sample, so the result is the same (to a few 1e-7 of the output range: the half-piece kernels' operand
# scale is a power of two derived from the whole tensor's largest |value|, test_sample_result_does_not_depend_on_its_batch).  (6 x 512 x 512: 100 M elements; the limit is reached at e.g.
# 8 x 2048 x 2048.)
MAX_ELEMS = 2 ** 31 - 1


def _peak_elems_per_sample(steps, C, H, W):
    """Largest NHWC buffer (elements per sample, channel padding included) a plan touches for a C x H x W input."""
    peak = max(C, 4) * H * W
    for s in steps:
        if s.kind == "stem3":
            C = 64
        elif s.kind in ("stem", "smallco"):
            C = s.pc.cout if s.kind == "stem" else s.cout
        elif s.kind == "conv":
            if s.ups:
                H, W = 2 * H, 2 * W
            peak = max(peak, s.pc.k_pad * H * W)
            H = (H + 2 * s.pad - s.pc.kh) // s.stride + 1
            W = (W + 2 * s.pad - s.pc.kw) // s.stride + 1
            if s.pool:                       # (the un-pooled map is never written)
                H, W = (H + 1) // 2, (W + 1) // 2
            C = s.pc.cout
        elif s.kind == "pad":
            H, W = H + 2 * s.pad, W + 2 * s.pad
        elif s.kind == "up":
            H, W = 2 * H, 2 * W
        elif s.kind == "pool":
            H, W = (H + 1) // 2, (W + 1) // 2
        peak = max(peak, ((C + 15) // 16 * 16) * H * W)
    return peak


def _run(steps, x, want_sums=False):
    """x: logical NCHW CUDA tensor.  Returns a logical NCHW tensor; with want_sums also the per-channel (sum, sum of squares, count)
    of the result (calc_sum, mean_std_computation_effcientMem.py:103-115) -- from the last conv's epilogue where that conv runs on
    the 64-channel F(4x4) kernel, else from the streaming-sums kernel."""
    if isinstance(x, torch.Tensor) and x.dim() == 4 and x.shape[0] > 0:
        per = _peak_elems_per_sample(steps, int(x.shape[1]), int(x.shape[2]), int(x.shape[3]))
        if per > MAX_ELEMS:
            raise ValueError("ccst_amd.net: a %dx%d image needs an activation of %d elements; the kernels address < 2^31 per tensor"
                             % (x.shape[2], x.shape[3], per))
        if per * int(x.shape[0]) > MAX_ELEMS:
            n = max(1, MAX_ELEMS // per)
            parts = [_run_batch(steps, x[i:i + n], want_sums) for i in range(0, int(x.shape[0]), n)]
            if not want_sums:
                return torch.cat(parts, dim=0)
            y = torch.cat([p[0] for p in parts], dim=0)
            return y, (sum(p[1][0] for p in parts), sum(p[1][1] for p in parts), sum(p[1][2] for p in parts))
    return _run_batch(steps, x, want_sums)


def _run_batch(steps, x, want_sums=False):
    if not want_sums:
        return _run_steps(steps, x, None)
    box = []
    y = _run_steps(steps, x, box)
    if box:
        s, q = ops.chan_sums_finalize(box[0])
        return y, (s, q, int(y.shape[0] * y.shape[2] * y.shape[3]))
    return y, ops.chan_sums(y)


def affine_ok(steps, x):
    """Can the plan `steps` take input x ([N,C,H,W], NHWC in memory) through a fused per-(image, channel) affine map -- i.e. is its
    first step a plain reflect-padded 3x3 conv that runs on the F(4,3) kernel for this extent, and does the batch go through in one
    piece?  (The decoder of net.py:6-36 at the metric's sizes: yes.)"""
    if not steps or steps[0].kind != "conv" or not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4):
        return False
    s = steps[0]
    N, C, H, W = (int(v) for v in x.shape)
    if not (s.stride == 1 and s.pad == 1 and s.reflect and not s.ups and not s.pool and not s.out_nchw and C == s.pc.cin and C % 16 == 0
            and ops.halo_split_wanted(s.pc) and ops.f43_wanted(s.pc, N, H, W, x.device) and x.permute(0, 2, 3, 1).is_contiguous()):
        return False
    return _peak_elems_per_sample(steps, C, H, W) * N <= MAX_ELEMS


def _run_steps(steps, x, sums_box, affine=None, affine_words=None):
    """sums_box: None, or a list that receives the last conv's per-tile channel-sum partials when that conv can produce them.
    affine = (a, b) [N, C] + affine_words (the per-image words of a x + b): the first step -- a conv, affine_ok() -- reads x through
    that map (the fused AdaIN step)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise RuntimeError("ccst_amd.net: input must be a CUDA (ROCm) tensor; the HIP path has no CPU fallback")
    if x.dim() != 4:
        raise RuntimeError("ccst_amd.net: expected a 4-D NCHW tensor")
    C = x.shape[1]
    cur = None          # NHWC buffer
    api = x             # logical NCHW tensor not yet converted
    # per-image |max| words of `cur` (ops.sample_absmax_words) where its producer left them: the half-piece (SPLIT) conv kernels scale their input by the
    # power of two derived from them, on the device.  ReLU / pad / upsample / pool keep them valid (an upper bound suffices); a tensor
    # that arrives without them costs the first half-piece conv one extra pass (ops.absmax_samples).
    amax = ops.tagged_absmax(x) if affine is None else affine_words
    if amax is not None and amax.numel() != x.shape[0] * ops.ABSMAX_WORDS:
        amax = None     # (words of another granularity -- a per-tensor set of the ResNet kernels: this plan's kernels take per-image words)

    def split_next(i):
        """Does the next conv after step i run on the SPLIT kernel (i.e. is max |out| of step i wanted)?"""
        for t in steps[i + 1:]:
            if t.kind == "conv":
                return t.stride == 1 and t.pad == 1 and not t.out_nchw and ops.halo_split_wanted(t.pc)
            if t.kind == "smallco":
                return t.pz is not None
            if t.kind not in ("relu", "pad", "up", "pool"):
                return False
        return False

    for si, s in enumerate(steps):
        if s.kind == "stem3":
            if cur is not None:
                api = ops.to_api(cur[..., :C]) if cur.shape[-1] != C else ops.to_api(cur)
            amax = ops.sample_absmax_words(x.device, api.shape[0]) if split_next(si) else None
            cur = ops.conv3x3_stem3_nchw(ops.as_nchw_contiguous(api), s.wa, relu=s.relu, y_absmax=amax)
            C, api = 64, None
            continue
        if s.kind == "stem":
            if cur is not None:
                api = ops.to_api(cur[..., :C]) if cur.shape[-1] != C else ops.to_api(cur)
            img = ops.as_nchw_contiguous(api)
            cur = ops.conv2d_stem_nchw(img, s.pc, s.kwp, s.kw, stride=s.stride, pad=s.pad, reflect=s.reflect, relu=s.relu)
            C = s.pc.cout
            api, amax = None, None
            continue
        if cur is None:
            cur = ops.from_api(api, cpad=16 if s.kind in ("conv", "smallco") else 4)
            api = None
        if s.kind == "smallco":
            if cur.shape[-1] % 16 != 0:
                cur = ops.from_api(ops.to_api(cur[..., :C]), cpad=16)
            if cur.shape[-1] != s.w_small.shape[-1]:
                raise RuntimeError("ccst_amd.net: the image-edge conv expects %d channels, got %d" % (s.w_small.shape[-1], cur.shape[-1]))
            out = ops.conv3x3_zform_nchw(cur, s.pz, s.b_small, s.cout, reflect=s.reflect, relu=s.relu, x_absmax=amax)
            cur, api, C, amax = None, out, s.cout, None
            continue
        if s.kind == "conv":
            if cur.shape[-1] != s.pc.k_pad:
                cur = ops.from_api(ops.to_api(cur[..., :C]), cpad=16)
            split = s.stride == 1 and s.pad == 1 and not s.out_nchw and ops.halo_split_wanted(s.pc) and cur.shape[-1] == s.pc.cin
            ymax = ops.sample_absmax_words(x.device, cur.shape[0]) if split and split_next(si) else None
            if sums_box is not None and s is steps[-1] and ops.conv_sums_ok(s.pc, s.stride, s.pad, s.pool, s.out_nchw):
                out, part = ops.conv2d_nhwc(cur, s.pc, stride=s.stride, pad=s.pad, reflect=s.reflect, relu=s.relu, pool=s.pool,
                                            ups=s.ups, chan_sums=True, x_absmax=amax, y_absmax=ymax)
                sums_box.append(part)
            else:
                out = ops.conv2d_nhwc(cur, s.pc, stride=s.stride, pad=s.pad, reflect=s.reflect, relu=s.relu, pool=s.pool,
                                      ups=s.ups, out_nchw=s.out_nchw, x_absmax=amax, y_absmax=ymax, affine=affine if si == 0 else None)
            amax = ymax
            C = s.pc.cout
            if s.out_nchw:
                return out
            cur = out
        elif s.kind == "relu":
            cur = ops.relu_nhwc(cur)
        elif s.kind == "pad":
            cur = ops.reflection_pad_nhwc(cur, s.pad)
        elif s.kind == "up":
            cur = ops.upsample2_nhwc(cur)
        elif s.kind == "pool":
            cur = ops.maxpool2_ceil_nhwc(cur)
    if cur is None:
        return api
    y = ops.to_api(cur)
    return y if cur.shape[-1] == C else y[:, :C]


def _signature(mods):
    sig = []
    for m in mods:
        for p in m.parameters(recurse=False):
            sig.append((id(p), p._version, p.data_ptr()))
    return (tuple(id(m) for m in mods), tuple(sig), ops.WEIGHTS_EPOCH)


class _PlanCache(object):
    def __init__(self):
        self.sig = None
        self.steps = None

    def get(self, mods):
        sig = _signature(mods)
        if sig != self.sig:
            for m in mods:
                for p in m.parameters(recurse=False):
                    if not p.is_cuda:
                        raise RuntimeError("ccst_amd.net: move the network to the GPU first (.to('cuda')); no CPU fallback")
            with torch.no_grad():
                self.steps = _compile(mods)
            self.sig = sig
        return self.steps


# ---------------------------------------------------------------------------
# modules
# ---------------------------------------------------------------------------
class Sequential(nn.Sequential):
    """nn.Sequential whose forward is the fused HIP plan.  Inference only (the CCST scripts run
    these networks frozen, under no_grad: CCST_OverallStyleTransfer.py:151)."""

    def forward(self, input):
        cache = self.__dict__.get("_ccst_plan")
        if cache is None:
            cache = _PlanCache()
            self.__dict__["_ccst_plan"] = cache
        return _run(cache.get(list(self.children())), input)

    def forward_with_chan_sums(self, input):
        """(self(input), (sum, sqsum, count)): the features AND their per-channel sums over (N, H, W) -- stage 1's
        ``feat = vgg(x); calc_sum(feat)`` (mean_std_computation_effcientMem.py:124-126) with the sums taken from the last conv's
        epilogue instead of a pass over the 268 MB tensor."""
        cache = self.__dict__.get("_ccst_plan")
        if cache is None:
            cache = _PlanCache()
            self.__dict__["_ccst_plan"] = cache
        return _run(cache.get(list(self.children())), input, want_sums=True)


def _forward_with_tile_sums(self, input):
    """(self(input), partials): the features and, where the last conv runs on the 64-channel F(4x4) kernel, the per-(spatial tile,
    channel) (sum, sum of squares) pairs its epilogue leaves ([N * tiles, C, 2], an image's tiles contiguous) -- what
    ops.adain_from_tile_sums needs instead of a statistics pass over the features; partials is None where that conv cannot
    produce them (other kernels, or a batch that has to be processed in slices)."""
    cache = self.__dict__.get("_ccst_plan")
    if cache is None:
        cache = _PlanCache()
        self.__dict__["_ccst_plan"] = cache
    steps = cache.get(list(self.children()))
    x = input
    if isinstance(x, torch.Tensor) and x.dim() == 4 and x.shape[0] > 0:
        per = _peak_elems_per_sample(steps, int(x.shape[1]), int(x.shape[2]), int(x.shape[3]))
        if per > MAX_ELEMS:
            return _run(steps, x), None                   # (raises: one image does not fit)
        if per * int(x.shape[0]) > MAX_ELEMS:             # slices of the batch: an image's tiles stay contiguous, so the partials concatenate
            n = max(1, MAX_ELEMS // per)
            ys, ps = [], []
            for i in range(0, int(x.shape[0]), n):
                bx = []
                ys.append(_run_steps(steps, x[i:i + n], bx))
                ps.append(bx[0] if bx else None)
            y = torch.cat([v.permute(0, 2, 3, 1) for v in ys], dim=0).permute(0, 3, 1, 2)        # (stays NHWC in memory)
            return y, (torch.cat(ps, dim=0) if all(p_ is not None for p_ in ps) else None)
    box = []
    y = _run_steps(steps, x, box)
    return y, (box[0] if box else None)


Sequential.forward_with_tile_sums = _forward_with_tile_sums


def _plan_of(self):
    cache = self.__dict__.get("_ccst_plan")
    if cache is None:
        cache = _PlanCache()
        self.__dict__["_ccst_plan"] = cache
    return cache.get(list(self.children()))


def _forward_affine(self, input, affine, words):
    """self(a * input + b) with the per-(image, channel) map applied inside the first conv (affine_ok(self, input) must hold): the
    fused AdaIN step of style.style_transfer.  affine = (a, b) [N, C]; words: per-image |max| words bounding |a x + b|."""
    steps = _plan_of(self)
    if not affine_ok(steps, input):
        raise ValueError("ccst_amd.net: this network / input cannot take a fused input affine (net.affine_ok)")
    return _run_steps(steps, input, None, affine=affine, affine_words=words)


Sequential.forward_affine = _forward_affine
Sequential.affine_ok = lambda self, x: affine_ok(_plan_of(self), x)


class _SingleMixin(object):
    def _ccst_forward(self, input):
        cache = self.__dict__.get("_ccst_plan")
        if cache is None:
            cache = _PlanCache()
            self.__dict__["_ccst_plan"] = cache
        return _run(cache.get([self]), input)


class Conv2d(_SingleMixin, nn.Conv2d):
    def forward(self, input):
        return self._ccst_forward(input)


class ReflectionPad2d(_SingleMixin, nn.ReflectionPad2d):
    def forward(self, input):
        return self._ccst_forward(input)


class ReLU(_SingleMixin, nn.ReLU):
    def forward(self, input):
        return self._ccst_forward(input)


class MaxPool2d(_SingleMixin, nn.MaxPool2d):
    def forward(self, input):
        return self._ccst_forward(input)


class Upsample(_SingleMixin, nn.Upsample):
    def forward(self, input):
        return self._ccst_forward(input)


def _pad():
    return ReflectionPad2d((1, 1, 1, 1))


def _block(cin, cout, relu=True):
    layers = [_pad(), Conv2d(cin, cout, (3, 3))]
    if relu:
        layers.append(ReLU())
    return layers


def _pool():
    return MaxPool2d((2, 2), (2, 2), (0, 0), ceil_mode=True)


def _build_decoder():      # net.py:6-36
    L = _block(512, 256) + [Upsample(scale_factor=2, mode='nearest')]
    L += _block(256, 256) + _block(256, 256) + _block(256, 256) + _block(256, 128) + [Upsample(scale_factor=2, mode='nearest')]
    L += _block(128, 128) + _block(128, 64) + [Upsample(scale_factor=2, mode='nearest')]
    L += _block(64, 64) + _block(64, 3, relu=False)
    return Sequential(*L)


def _build_vgg():          # net.py:38-92
    L = [Conv2d(3, 3, (1, 1))]
    L += _block(3, 64) + _block(64, 64) + [_pool()]
    L += _block(64, 128) + _block(128, 128) + [_pool()]
    L += _block(128, 256) + _block(256, 256) + _block(256, 256) + _block(256, 256) + [_pool()]
    L += _block(256, 512) + _block(512, 512) + _block(512, 512) + _block(512, 512) + [_pool()]
    L += _block(512, 512) + _block(512, 512) + _block(512, 512) + _block(512, 512)
    return Sequential(*L)


decoder = _build_decoder()
vgg = _build_vgg()


def fuse(seq):
    """Wrap any nn.Sequential of the supported layers into the fused HIP executor."""
    return seq if isinstance(seq, Sequential) else Sequential(*list(seq.children()))
