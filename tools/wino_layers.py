#!/usr/bin/env python3
"""Per-layer timing of the AdaIN path's sixteen 3x3 layers (B=6, 512x512) on the F(2x2) and F(4x4) Winograd kernels:
python tools/wino_layers.py [reps]  -> one line per layer: shape, us and algorithmic TFLOP/s of each kernel, max |difference|."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccst_amd import _lib, ops
from ccst_amd._lib import check, ptr, stream_ptr

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
B = 6
# (H, W of the conv, Cin, Cout, pool, ups)  encoder conv1_2 .. conv4_1, decoder (net.py:6-36,38-69)
LAYERS = [(512, 512, 64, 64, True, False), (256, 256, 64, 128, False, False), (256, 256, 128, 128, True, False),
          (128, 128, 128, 256, False, False), (128, 128, 256, 256, False, False), (128, 128, 256, 256, False, False),
          (128, 128, 256, 256, True, False), (64, 64, 256, 512, False, False),
          (64, 64, 512, 256, False, False), (128, 128, 256, 256, False, True), (128, 128, 256, 256, False, False),
          (128, 128, 256, 256, False, False), (128, 128, 256, 128, False, False), (256, 256, 128, 128, False, True),
          (256, 256, 128, 64, False, False), (512, 512, 64, 64, False, True)]
g = torch.Generator().manual_seed(1)
if os.environ.get('WINO_LAYER'):
    LAYERS = [LAYERS[int(i)] for i in os.environ['WINO_LAYER'].split(',')]
tot2 = tot4 = 0.0
for (H, W, Cin, Cout, pool, ups) in LAYERS:
    Hs, Ws = (H // 2, W // 2) if ups else (H, W)
    x = torch.rand(B, Hs, Ws, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.05).to(dev)
    pc = ops.pack_conv_weight(w, b, wino=4)
    flags = 1 | 8 | (2 if pool else 0) | (4 if ups else 0)
    oh, ow = ((H + 1) // 2, (W + 1) // 2) if pool else (H, W)
    y2 = torch.empty(B, oh, ow, Cout, device=dev)
    lib = _lib.load()

    def f2():
        check(lib.ccst_conv3x3_wino_f32(ptr(x), ptr(pc.u), ptr(pc.bias), ptr(y2), B, H, W, Cin, Cout, pc.u_pad, flags, stream_ptr()), "w2")

    def f4():
        return ops.conv3x3_wino4(x, pc, flags)
    res = []
    for fn in (f2, f4):
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
    fl = 2.0 * B * H * W * Cin * Cout * 9
    tot2 += res[0]
    tot4 += res[1]
    print("%4dx%-4d %3d->%-3d %s%s  F2 %7.1f us %6.1f TF   F4 %7.1f us %6.1f TF   x%.2f   maxdiff %.2e (|y| %.2f)" % (
        H, W, Cin, Cout, "P" if pool else "-", "U" if ups else "-", res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6,
        res[0] / res[1], float((y4 - y2).abs().max()), float(y2.abs().max())), flush=True)
print("sum  F2 %.1f us   F4 %.1f us" % (tot2, tot4))
