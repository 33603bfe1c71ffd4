#!/usr/bin/env python3
"""The implicit-GEMM kernel (conv_igemm.hip) alone on the ResNet50 B=64 222x222 pointwise shapes (forward form, with the BN
statistics epilogue as in training): python tools/igemm_time.py [reps]  -> per shape: tile code, us, TFLOP/s, GB/s of the
compulsory traffic.  IGEMM_SHAPES=i,j,.. picks rows; CCST_CONV_TILE overrides the tile."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccst_amd import _lib, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
N = 64
# (H = W, cin, cout, k, stride)
SHAPES = [(56, 64, 256, 1, 1), (56, 256, 64, 1, 1), (28, 128, 512, 1, 1), (28, 512, 128, 1, 1), (14, 256, 1024, 1, 1),
          (14, 1024, 256, 1, 1), (7, 512, 2048, 1, 1), (7, 2048, 512, 1, 1), (7, 512, 512, 3, 1), (56, 256, 128, 1, 1)]
if os.environ.get("IGEMM_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["IGEMM_SHAPES"].split(",")]
lib = _lib.load()
g = torch.Generator().manual_seed(5)
tot = 0.0
for (H, cin, cout, k, stride) in SHAPES:
    x = torch.randn(N, H, H, cin, generator=g).to(dev)
    w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(dev)
    pc = ops.pack_conv_weight(w, None)
    pad = k // 2

    def run():
        return ops.conv2d_nhwc(x, pc, stride=stride, pad=pad, want_stats=True)
    for _ in range(3):
        y, st = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y, st = run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    M = y.shape[0] * y.shape[1] * y.shape[2]
    fl = 2.0 * M * cin * cout * k * k
    byt = 4.0 * (x.numel() + y.numel())
    tile = lib.ccst_conv2d_igemm_tile(M, cout, cin, k * k, 0)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, stride=stride, padding=pad).permute(0, 2, 3, 1)
    err = float((y - ref).abs().max() / ref.abs().max())
    tot += us
    print("%3dx%-3d %4d->%-4d k%d  tile %4d  %7.1f us  %6.1f TF  %6.0f GB/s   rel.diff %.1e" % (H, H, cin, cout, k, tile, us, fl / us / 1e6, byt / us / 1e3, err),
          flush=True)
print("sum %.1f us" % tot)
