#!/usr/bin/env python3
"""Per-layer timing of the AdaIN path's sixteen 3x3 layers (B=6, 512x512) on the direct halo kernel in its SPLIT form (three products of half pieces per fp32 product on the 16-bit MFMA)
(conv3x3_halo.hip, ccst_conv3x3_halo_split_f32; HALO_FP32=1: the fp32-MFMA form) with the 64-channel F(4x4) kernel beside it: python tools/halo_layers.py [reps] -> per layer us, algorithmic
TFLOP/s of both and the largest difference from an fp64 convolution of the first image (relative to max |y|)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from ccst_amd import _lib, ops
from ccst_amd._lib import check, ptr, stream_ptr

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
B = 6
LAYERS = [(512, 512, 64, 64, True, False), (256, 256, 64, 128, False, False), (256, 256, 128, 128, True, False),
          (128, 128, 128, 256, False, False), (128, 128, 256, 256, False, False), (128, 128, 256, 256, False, False),
          (128, 128, 256, 256, True, False), (64, 64, 256, 512, False, False),
          (64, 64, 512, 256, False, False), (128, 128, 256, 256, False, True), (128, 128, 256, 256, False, False),
          (128, 128, 256, 256, False, False), (128, 128, 256, 128, False, False), (256, 256, 128, 128, False, True),
          (256, 256, 128, 64, False, False), (512, 512, 64, 64, False, True)]
if os.environ.get('WINO_LAYER'):
    LAYERS = [LAYERS[int(i)] for i in os.environ['WINO_LAYER'].split(',')]
g = torch.Generator().manual_seed(1)
lib = _lib.load()
toth = tot4 = 0.0
for (H, W, Cin, Cout, pool, ups) in LAYERS:
    Hs, Ws = (H // 2, W // 2) if ups else (H, W)
    x = torch.rand(B, Hs, Ws, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.05).to(dev)
    pc = ops.pack_conv_weight(w, b, wino=4)
    flags = 1 | 8 | (2 if pool else 0) | (4 if ups else 0)
    oh, ow = ((H + 1) // 2, (W + 1) // 2) if pool else (H, W)
    yh = torch.empty(B, oh, ow, Cout, device=dev)
    xmax, ymax = ops.absmax(x), ops.absmax_words(x.device)      # (ymax is max-accumulated over the repetitions: same value every time)

    def fh():
        if os.environ.get("HALO_FP32") == "1":
            check(lib.ccst_conv3x3_halo_f32(ptr(x), ptr(pc.w), ptr(pc.bias), ptr(yh), B, H, W, Cin, Cout, pc.n_pad, flags, stream_ptr()), "halo")
        else:
            check(lib.ccst_conv3x3_halo_split_f32(ptr(x), ptr(xmax), ptr(pc.wsplit), ptr(pc.wabsmax), ptr(pc.bias), ptr(yh), ptr(ymax), B, H, W, Cin, Cout, pc.n_pad, flags, None, stream_ptr()), "halo_split")

    def f4():
        return ops.conv3x3_wino4(x, pc, flags)
    res = []
    for fn in (fh, f4):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / reps)
    y4 = f4()
    # fp64 reference of image 0
    x0 = x[:1].permute(0, 3, 1, 2).double()
    if ups:
        x0 = F.interpolate(x0, scale_factor=2, mode="nearest")
    r = F.relu(F.conv2d(F.pad(x0, (1, 1, 1, 1), mode="reflect"), w.double(), b.double()))
    if pool:
        r = F.max_pool2d(r, 2, 2, 0, ceil_mode=True)
    r = r.permute(0, 2, 3, 1)
    eh = float((yh[:1].double() - r).abs().max() / r.abs().max())
    e4 = float((y4[:1].double() - r).abs().max() / r.abs().max())
    fl = 2.0 * B * H * W * Cin * Cout * 9
    toth += res[0]
    tot4 += res[1]
    print("%4dx%-4d %3d->%-3d %s%s  halo %7.1f us %6.1f TF (err %.1e)   F4w %7.1f us %6.1f TF (err %.1e)   x%.2f" % (
        H, W, Cin, Cout, "P" if pool else "-", "U" if ups else "-", res[0], fl / res[0] / 1e6, eh, res[1], fl / res[1] / 1e6, e4, res[1] / res[0]))
print("sum  halo %.1f us   F4w %.1f us" % (toth, tot4))
