"""Drop-in for nets/resnet.py (ResNet :132-191, resnet18 :326-345, resnet50 :350-370) on HIP kernels.

Same attribute names (conv1, bn1, relu, maxpool, layer1..4, avgpool, class_classifier) and
state-dict keys as the reference + torchvision's BasicBlock/Bottleneck, so checkpoints interchange.
The blocks are restated here because the reference imports them from torchvision
(nets/resnet.py:3), which is not part of the reference tree: v1.5 layout, stride on the 3x3 conv.

Layers are subclasses of the torch.nn layers (isinstance checks and nn.init loops of the reference
keep working) whose forward runs HIP kernels; tensors between layers are logical NCHW views of NHWC
buffers.  BatchNorm is fused with the following ReLU and the residual add.  No CPU path.
"""
import os

import torch
from torch import nn

from .. import nn_ops, ops


def _to_nhwc(x):
    return ops.from_api(x, cpad=1)


class Conv2d(nn.Conv2d):
    """Bias-free zero-padded convolution (the only kind in the ResNet trunk)."""

    def _check(self):
        if self.bias is not None or self.groups != 1 or self.dilation != (1, 1) or self.padding_mode != "zeros" or \
                self.kernel_size[0] != self.kernel_size[1] or self.stride[0] != self.stride[1] or self.padding[0] != self.padding[1]:
            raise NotImplementedError("ccst_amd.nets: unsupported Conv2d %r" % (self,))

    def _cached(self, name, fn):
        w = self.weight
        key = (w._version, w.data_ptr(), ops.WEIGHTS_EPOCH)
        slot = self.__dict__.get("_ccst_" + name)
        if w.is_cuda and name != "pks":        # (the stem's virtual-pixel pack is never re-packed on the side stream)
            nn_ops.join_prepack(w.device)      # re-packs issued on the side stream by the optimiser step (reads AND
                                               # the rebuild below must not overlap them: same destination buffers)
        if slot is None or slot[0] != key:
            with torch.no_grad():
                slot = (key, fn(w.detach(), None if slot is None else slot[1]))
            self.__dict__["_ccst_" + name] = slot
        return slot[1]

    @staticmethod
    def _reuse(prev):
        return None if prev is None or not isinstance(prev, ops.PackedConv) else prev.w

    def packed(self):
        return self._cached("pk", lambda w, prev: ops.pack_conv_weight(w, out=self._reuse(prev)))

    def packed_t(self):
        return self._cached("pkt", lambda w, prev: ops.pack_conv_weight(w, transpose=True, out=self._reuse(prev)))

    def wabsmax(self):
        """|max| words of the weight (ops.absmax); after the first optimiser step the batched side-stream refresh keeps them current."""
        # (refreshed IN PLACE: a captured graph reads the words at the address it was captured with -- ADVICE r4)
        def build(w, prev):
            if prev is None:
                return ops.absmax(w)
            nn_ops.fill_(prev.view(torch.float32), 0.0)
            return ops.absmax(w, out=prev)
        return self._cached("wmax", build)

    def packed_h(self):
        """The pointwise kernel's pre-split half-piece pack of a 1x1 weight (scaled by the words of wabsmax())."""
        return self._cached("pkh", lambda w, prev: ops.pack_conv_weight_split(w, self.wabsmax(), out=prev))

    def packed_th(self):
        return self._cached("pkht", lambda w, prev: ops.pack_conv_weight_split(w, self.wabsmax(), transpose=True, out=prev))

    def halo_h(self):
        """Half-piece weight image of a 3x3 weight for the halo kernel's train form (scaled by the words of wabsmax())."""
        return self._cached("hh", lambda w, prev: ops.pack_halo_split(w, self.wabsmax(), out=None if prev is None else prev[0]))

    def halo_ht(self):
        return self._cached("hht", lambda w, prev: ops.pack_halo_split(w, self.wabsmax(), bwd=True, out=None if prev is None else prev[0]))

    def halo_split_ok(self):
        return self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1) and \
            ops.halo_train_split_ok(0, 0, self.in_channels, self.out_channels)

    def wino_ok(self, H, W):
        return self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1) and \
            ops.wino_train_ok(H, W, self.in_channels, self.out_channels)

    def wino_fwd(self):
        return self._cached("wu", lambda w, prev: ops.pack_wino(w, out=None if prev is None else prev[0]))

    def wino_bwd(self):
        return self._cached("wut", lambda w, prev: ops.pack_wino(w, bwd=True, out=None if prev is None else prev[0]))

    def prepack(self):
        """Refresh every packed copy this conv holds now (a key compare when they are current)."""
        if self.in_channels > 4:
            d = self.__dict__
            if "_ccst_pk" in d:
                self.packed()
            if "_ccst_pkt" in d:
                self.packed_t()
            if "_ccst_wu" in d:
                self.wino_fwd()
            if "_ccst_wut" in d:
                self.wino_bwd()
            if "_ccst_wmax" in d:         # the half-piece kernels scale the weight by these words: as current as the packs
                self.wabsmax()
            if "_ccst_pkh" in d:          # (after the words: the pre-split packs are scaled by them)
                self.packed_h()
            if "_ccst_pkht" in d:
                self.packed_th()
            if "_ccst_hh" in d:
                self.halo_h()
            if "_ccst_hht" in d:
                self.halo_ht()

    def packed_stem(self):
        def build(w, prev):
            wv, kwp = ops.stem_virtual_weight(w)
            return ops.pack_conv_weight(wv), kwp
        return self._cached("pks", build)

    def forward(self, x, want_stats=False, sink=None, sole_reader=False):
        """want_stats=True (training only): returns (y, stats) with the BatchNorm batch-statistic partials of y
        produced in the conv epilogue; pass them to the following BatchNorm2d(..., stats=stats).
        sink: nn_ops.GradSink whose content this conv's backward-data adds to (residual blocks)."""
        self._check()
        if not x.is_cuda:
            raise RuntimeError("ccst_amd.nets: CUDA (ROCm) tensors only; no CPU fallback")
        if self.in_channels <= 4:
            gw = nn_ops.GradWords() if (self.training and torch.is_grad_enabled()) else None
            y = ops.to_api(nn_ops.StemConvFn.apply(x, self.weight, self, gw))
            y._ccst_gw = gw
            return (y, None) if want_stats else y
        # x is the ReLU output of the previous block's closing BatchNorm and this conv completes x's gradient (residual sink):
        # hand its mask to the backward (nn_ops.MaskLink)
        link = getattr(x, "_ccst_mask_link", None)
        if link is not None and not _plain_reader(x):
            link = None             # someone watches x's gradient (hook / retain_grad): it must be complete and unmasked there
        if link is not None:
            # the residual form goes with a (non-pair) sink, the BatchNorm + ReLU form with a conv that is x's only reader (no sink)
            # (sole_reader: the caller -- a block's forward -- vouches that nothing else reads x, so x's gradient is this conv's alone)
            ok = (link.mask is not None and sink is not None and not sink.pair) or (link.mask is None and sink is None and sole_reader)
            link = link if ok else None
        if want_stats:
            # (the |max| words a BatchNorm apply left for x travel with the NHWC view: the half-piece pointwise forward scales by them)
            # (gw: the holder through which the BatchNorm that reads y hands the |max| words of y's gradient to this conv's backward)
            gw = nn_ops.GradWords() if torch.is_grad_enabled() else None
            y, stats = nn_ops.ConvFn.apply(ops.carry_absmax(x, _to_nhwc(x)), self.weight, self, True, sink, link, gw)
            out = ops.to_api(y)
            out._ccst_gw = gw
            return out, stats
        return ops.to_api(nn_ops.ConvFn.apply(ops.carry_absmax(x, _to_nhwc(x)), self.weight, self, False, sink, link))


class BatchNorm2d(nn.BatchNorm2d):
    """BatchNorm2d fused with an optional residual add and ReLU: y = relu(bn(x) + residual)."""

    def forward(self, x, residual=None, relu=False, stats=None, sink=None):
        if not (self.affine and x.is_cuda):
            raise NotImplementedError("ccst_amd.nets: affine CUDA BatchNorm2d only")
        if not self.training and not self.track_running_stats:
            raise NotImplementedError("ccst_amd.nets: eval-mode BatchNorm2d needs running statistics")
        res = _to_nhwc(residual) if residual is not None else None
        self._ccst_mask_link = None
        y = nn_ops.BNFn.apply(_to_nhwc(x), self.weight, self.bias, res, self, bool(relu), stats if self.training else None,
                              sink if self.training else None, getattr(x, "_ccst_gw", None) if self.training else None)
        out = ops.carry_absmax(y, ops.to_api(y))
        if self._ccst_mask_link is not None:        # travels with the block output to the next block's first conv
            out._ccst_mask_link, self._ccst_mask_link = self._ccst_mask_link, None
        return out


def conv_bn(conv, bn, x, residual=None, relu=False, sink_in=None, sink_out=None, sole_reader=False):
    """bn(conv(x)) [+ residual] [ReLU]; in training the batch statistics come out of the conv epilogue.
    sink_out / sink_in: the GradSink that this BatchNorm's backward fills with the residual gradient / that this
    conv's backward-data adds to (a residual block whose identity branch is its input, see _residual_sink)."""
    if bn.training:
        y, stats = conv(x, want_stats=True, sink=sink_in, sole_reader=sole_reader)
        return bn(y, residual=residual, relu=relu, stats=stats, sink=sink_out)
    return bn(conv(x), residual=residual, relu=relu)


USE_GRAD_SINK = True
FUSED_STEM = True      # bn1 + relu + maxpool of the stem as one op in training


def _plain_reader(x):
    """The gradient short-cuts between blocks (GradSink, MaskLink) hand x's gradient from one backward node straight to another:
    what autograd itself sees for x is then partial (one branch's share) or already masked.  That is only sound while nothing
    else looks at it, so a tensor hook or retain_grad() on x -- a feature tap registered on a block output, e.g. from a module
    forward hook -- switches the short-cuts off for the block that reads x."""
    return not (x.retains_grad or x._backward_hooks)


def _pair_sink(block, x):
    """The GradSink shared by the two convolutions that read x in a block with a downsample branch (see nn_ops.GradSink)."""
    if USE_GRAD_SINK and block.downsample is not None and block.training and torch.is_grad_enabled() and x.requires_grad \
            and _plain_reader(x):
        return nn_ops.GradSink(pair=True)
    return None


def _residual_sink(block, x, first_conv):
    """A GradSink when the block's identity branch is x itself, x needs a gradient and the first conv has stride 1
    (then d(identity) and the first conv's dX have the same shape and the conv can add to it in place)."""
    if USE_GRAD_SINK and block.downsample is None and block.training and torch.is_grad_enabled() and x.requires_grad \
            and first_conv.stride[0] == 1 and _plain_reader(x):
        return nn_ops.GradSink()
    return None


class ReLU(nn.ReLU):
    """Kept for attribute/state parity (``model.relu``); in the forward pass ReLU is fused into BatchNorm2d."""

    def forward(self, x):
        if x.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("ccst_amd.nets: stand-alone differentiable ReLU is not used by the ResNet path "
                                      "(it is fused: bn(x, relu=True))")
        return ops.to_api(ops.relu_nhwc(ops.from_api(x, cpad=4)))


class MaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        if (self.kernel_size, self.stride, self.padding, self.ceil_mode) != (3, 2, 1, False):
            raise NotImplementedError("ccst_amd.nets: only MaxPool2d(3, 2, 1) (nets/resnet.py:140)")
        y = nn_ops.MaxPool3s2Fn.apply(ops.carry_absmax(x, _to_nhwc(x)))
        return ops.carry_absmax(y, ops.to_api(y))


class AvgPool2d(nn.AvgPool2d):
    def forward(self, x):
        k = self.kernel_size if isinstance(self.kernel_size, int) else self.kernel_size[0]
        H, W = x.shape[2], x.shape[3]
        if H < k or W < k:
            raise RuntimeError("Given input size: (%dx%dx%d). Calculated output size: (%dx%dx%d). Output size is too small"
                               % (x.shape[1], H, W, x.shape[1], H - k + 1, W - k + 1))
        if (H, W) != (k, k):
            raise NotImplementedError("ccst_amd.nets: AvgPool2d(%d) is implemented for a %dx%d map (image_size 193..224)" % (k, k, k))
        y = nn_ops.AvgPoolFlattenFn.apply(_to_nhwc(x))
        return y.view(y.shape[0], y.shape[1], 1, 1)


class Linear(nn.Linear):
    def forward(self, x):
        return nn_ops.LinearFn.apply(x, self.weight, self.bias)


def conv3x3(in_planes, out_planes, stride=1):
    return Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = BatchNorm2d(planes)
        self.relu = ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        sink = _residual_sink(self, x, self.conv1)
        pair = _pair_sink(self, x)
        if self.downsample is not None:
            # evaluated FIRST so that its backward runs LAST (autograd runs later-created nodes first): the main branch's dense dX is
            # deposited in `pair`, and the strided downsample conv adds its (sparser) dX to it in place -- no zero-filled temporary
            identity = conv_bn(self.downsample[0], self.downsample[1], x, sink_in=pair)
        out = conv_bn(self.conv1, self.bn1, x, relu=True, sink_in=sink if sink is not None else pair)
        return conv_bn(self.conv2, self.bn2, out, residual=identity, relu=True, sink_out=sink, sole_reader=True)     # out = relu(bn1(..)): read here only


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv1x1(inplanes, planes)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = conv3x3(planes, planes, stride)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = conv1x1(planes, planes * self.expansion)
        self.bn3 = BatchNorm2d(planes * self.expansion)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        sink = _residual_sink(self, x, self.conv1)
        pair = _pair_sink(self, x)
        if self.downsample is not None:         # first, so that its backward runs last and adds to the main branch's dX (BasicBlock)
            identity = conv_bn(self.downsample[0], self.downsample[1], x, sink_in=pair)
        out = conv_bn(self.conv1, self.bn1, x, relu=True, sink_in=sink if sink is not None else pair)
        out = conv_bn(self.conv2, self.bn2, out, relu=True, sole_reader=True)       # (its input relu(bn1(..)) is read here only)
        return conv_bn(self.conv3, self.bn3, out, residual=identity, relu=True, sink_out=sink, sole_reader=True)     # out = relu(bn2(..)): read here only


class ResNet(nn.Module):
    """nets/resnet.py:132-191."""

    def __init__(self, block, layers, classes=100):
        self.inplanes = 64
        super(ResNet, self).__init__()
        self.conv1 = Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = ReLU(inplace=True)
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = AvgPool2d(7, stride=1)
        self.class_classifier = Linear(512 * block.expansion, classes)

        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def is_patch_based(self):
        return False

    def counter_arena(self):
        """Every BatchNorm2d.num_batches_tracked of the model as ONE int64 tensor (module order): the counters are re-homed into a small
        arena (as FlatParams does for the fp32 state), so that a train step bumps them with one launch and communication() copies
        them with one.  None if the model tracks none."""
        arena = self.__dict__.get("_ccst_nbt")
        cs = [m.num_batches_tracked for m in self.modules() if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None]
        if not cs:
            return None
        if arena is None or arena.device != cs[0].device or any(c.data_ptr() != arena.data_ptr() + 8 * i for i, c in enumerate(cs)):
            arena = torch.stack([c.detach().reshape(()) for c in cs]).to(torch.int64).contiguous()
            for i, c in enumerate(cs):
                c.data = arena[i]
            self.__dict__["_ccst_nbt"] = arena
        return arena

    def _bump_counters(self):
        arena = self.counter_arena()
        if arena is None:
            return
        from .._lib import check, load, ptr, stream_ptr
        check(load().ccst_add_i64(ptr(arena), 1, arena.numel(), stream_ptr()), "bump num_batches_tracked")

    def forward(self, x, **kwargs):
        if self.training:
            self._bump_counters()
        x = self._stem(x)
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        x = self.avgpool(x)
        x = x.view(x.size(0), -1)
        return self.class_classifier(x)

    def _stem(self, x):
        """conv1 -> bn1 -> relu -> maxpool (nets/resnet.py:136-140).  In training the last three are one op that never stores the
        normalised 111 x 111 map or its gradient (nn_ops.StemBnReluPoolFn)."""
        bn, mp = self.bn1, self.maxpool
        if FUSED_STEM and bn.training and torch.is_grad_enabled() and isinstance(bn, BatchNorm2d) and isinstance(mp, MaxPool2d) and bn.affine \
                and (mp.kernel_size, mp.stride, mp.padding, mp.ceil_mode) == (3, 2, 1, False):
            y, stats = self.conv1(x, want_stats=True)
            yp = nn_ops.StemBnReluPoolFn.apply(_to_nhwc(y), bn.weight, bn.bias, bn, stats, getattr(y, "_ccst_gw", None))
            return ops.carry_absmax(yp, ops.to_api(yp))
        return mp(conv_bn(self.conv1, bn, x, relu=True))

    def _apply(self, fn, *a, **k):
        self.__dict__.pop("_ccst_nbt", None)
        p0 = next(self.parameters(), None)
        before = None if p0 is None else (p0.device, p0.data_ptr())
        out = super()._apply(fn, *a, **k)
        p1 = next(self.parameters(), None)
        if before != (None if p1 is None else (p1.device, p1.data_ptr())):
            # captured train steps (fed.train) hold addresses of the tensors that just moved; a model.to(device) that moves nothing
            # (train() and test() call it every time) keeps them
            self.__dict__.pop("_ccst_graph_steps", None)
        return out


def _maybe_pretrained(model, name, pretrained):
    """nets/resnet.py:340-344,364-369 load ImageNet weights through model_zoo (network access; ``strict=False`` because the
    head is ``class_classifier``, not ``fc``).  Here the same state dict is read from $CCST_PRETRAINED_DIR/<name>.pth; asking for
    pretrained weights that are not there is an error, never a silent random initialisation."""
    if not pretrained:
        return model
    d = os.environ.get("CCST_PRETRAINED_DIR", "")
    path = os.path.join(d, name + ".pth")
    if not (d and os.path.exists(path)):
        raise FileNotFoundError("ccst_amd.nets: pretrained=True but %s does not exist (set CCST_PRETRAINED_DIR to a directory holding "
                                "torchvision's %s state dict as %s.pth; there is no network access to download it). Pass "
                                "pretrained=False for the reference's kaiming initialisation." % (path or "<CCST_PRETRAINED_DIR>/%s.pth" % name, name, name))
    model.load_state_dict(torch.load(path, map_location="cpu"), strict=False)
    print("Use pretrained %s" % name)
    return model


def _unsupported(args):
    dg = getattr(args, "dg_method", "") or ""
    if dg.lower() in ("jigsaw", "mixstyle"):
        raise NotImplementedError("ccst_amd.nets: --dg_method %s is outside the hot path (SURVEY.md section 2)" % dg)


def resnet18(args, pretrained=True, **kwargs):
    """Constructs a ResNet-18 model (nets/resnet.py:326-345)."""
    _unsupported(args)
    return _maybe_pretrained(ResNet(BasicBlock, [2, 2, 2, 2], **kwargs), "resnet18", pretrained)


def resnet50(args, pretrained=True, **kwargs):
    """Constructs a ResNet-50 model (nets/resnet.py:350-370)."""
    _unsupported(args)
    return _maybe_pretrained(ResNet(Bottleneck, [3, 4, 6, 3], **kwargs), "resnet50", pretrained)
