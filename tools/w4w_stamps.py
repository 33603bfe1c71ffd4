#!/usr/bin/env python3
"""Phase timing of the wide F(4x4) kernel from in-kernel s_memtime stamps (diagnostic build -DABLW_STAMPS, CCST_HIP_LIB pointing at
it): per tile 0 start, 1 loop start, 2 loop end, 3 first epilogue pass done, 4 tile done, 5 after the closing barrier.
python tools/w4w_stamps.py <layer index of tools/wino_layers.py>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ccst_amd import ops

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
LAYERS = [(512, 512, 64, 64, True, False), (256, 256, 64, 128, False, False), (256, 256, 128, 128, True, False),
          (128, 128, 128, 256, False, False), (128, 128, 256, 256, False, False), (128, 128, 256, 256, False, False),
          (128, 128, 256, 256, True, False), (64, 64, 256, 512, False, False),
          (64, 64, 512, 256, False, False), (128, 128, 256, 256, False, True)]
H, W, Cin, Cout, pool, ups = LAYERS[L]
B = 6
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
Hs, Ws = (H // 2, W // 2) if ups else (H, W)
x = torch.rand(B, Hs, Ws, Cin, generator=g).to(dev)
w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
b = (torch.randn(Cout, generator=g) * 0.05).to(dev)
pc = ops.pack_conv_weight(w, b, wino=4)
flags = 1 | 8 | (2 if pool else 0) | (4 if ups else 0)
for _ in range(5):
    out = ops.conv3x3_wino4(x, pc, flags)
torch.cuda.synchronize()
ntiles = B * ((H + 15) // 16) * ((W + 31) // 32) * ((Cout + 63) // 64)
raw = out.view(-1)[:ntiles * 32].cpu().numpy().view(np.uint64).reshape(ntiles, 16).astype(np.int64)
d = np.diff(raw[:, :6], axis=1)
names = ["prologue", "loop", "epilogue pass 0", "epilogue pass 1", "closing barrier"]
print("layer %d: %dx%d %d->%d, %d tiles; s_memtime ticks (median / p10 / p90), whole tile median %d" % (L, H, W, Cin, Cout, ntiles, np.median(raw[:, 5] - raw[:, 0])))
for k, nm in enumerate(names):
    print("  %-18s %8d %8d %8d" % (nm, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
pro = [("set-up (tile decode, addresses)", 0, 8), ("issue 18 weight + 10 halo loads", 8, 9), ("wait + 10 ds_write", 9, 10), ("2 x 4 loads + barrier", 10, 11),
       ("first transform", 11, 1), ("pass 0: 36 LDS writes", 2, 12), ("pass 0: next tile set-up + 10 loads", 12, 13), ("pass 0: barrier", 13, 14),
       ("pass 0: 72 LDS reads + transform", 14, 15), ("pass 0: 16 stores (+ pass-1 writes)", 15, 3)]
for nm, a, b in pro:
    dd = raw[:, b] - raw[:, a]
    print("    %-34s %8d %8d %8d" % (nm, np.median(dd), np.percentile(dd, 10), np.percentile(dd, 90)))
t0 = raw[:, 0].min()
print("  launch span %d ticks; first tile starts spread %d; per-tile-slot start medians:" % (raw[:, 5].max() - t0, np.percentile(raw[:256, 0] - t0, 90)))
