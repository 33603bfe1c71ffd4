#!/usr/bin/env python3
"""Per-group timing of one pair-step of the wide F(4x4) kernel's loop (diagnostic build -DABLW_GSTAMPS): s_memtime before each of
the nine 4-MFMA groups of pair-step 1 of chunk 2, after the last group and after the column pass.  python tools/w4w_gstamps.py [layer]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ccst_amd import ops

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
LAYERS = [(512, 512, 64, 64, True, False), (256, 256, 64, 128, False, False), (256, 256, 128, 128, True, False),
          (128, 128, 128, 256, False, False), (128, 128, 256, 256, False, False)]
H, W, Cin, Cout, pool, ups = LAYERS[L]
B = 6
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.rand(B, H, W, Cin, generator=g).to(dev)
w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
b = (torch.randn(Cout, generator=g) * 0.05).to(dev)
pc = ops.pack_conv_weight(w, b, wino=4)
flags = 1 | 8 | (2 if pool else 0)
for _ in range(5):
    out = ops.conv3x3_wino4(x, pc, flags)
torch.cuda.synchronize()
ntiles = B * ((H + 15) // 16) * ((W + 31) // 32) * ((Cout + 63) // 64)
raw = out.view(-1)[:ntiles * 32].cpu().numpy().view(np.uint64).reshape(ntiles, 16).astype(np.int64)
d = np.diff(raw[:, :11], axis=1)
ok = (d > 0).all(axis=1) & (d < 100000).all(axis=1)
d = d[ok]
print("layer %d: %d of %d tiles with clean stamps; cycles per group (median / p10 / p90); a group = 4 MFMAs = 256 cycles of matrix pipe" % (L, len(d), ntiles))
for k in range(10):
    print("  %-28s %6d %6d %6d" % ("group %d" % k if k < 9 else "column pass (18 packed ops)", np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
print("  pair-step total %d" % np.median(d.sum(axis=1)))
