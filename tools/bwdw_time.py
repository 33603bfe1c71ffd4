#!/usr/bin/env python3
"""Weight-gradient kernel (conv_bwd_weight.hip) alone on the ResNet50 B=64 222x222 layer shapes:
python tools/bwdw_time.py [reps]   -> per shape: splits, us (kernel + slab reduce), TFLOP/s.
BWDW_SPLITS=<n> overrides the split count (workspace sized accordingly); BWDW_SHAPES=i,j,.. picks rows."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccst_amd import _lib, nn_ops
from ccst_amd._lib import check, ptr, stream_ptr

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
N = 64
# (H = W of the conv input, cin, cout, k, stride)
SHAPES = [(56, 64, 64, 3, 1), (28, 128, 128, 3, 1), (14, 256, 256, 3, 1), (7, 512, 512, 3, 1),
          (56, 64, 256, 1, 1), (56, 256, 64, 1, 1), (28, 128, 512, 1, 1), (28, 512, 128, 1, 1),
          (14, 256, 1024, 1, 1), (14, 1024, 256, 1, 1), (7, 512, 2048, 1, 1), (7, 2048, 512, 1, 1),
          (56, 256, 128, 1, 1), (56, 128, 128, 3, 2), (56, 256, 512, 1, 2)]
if os.environ.get("BWDW_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["BWDW_SHAPES"].split(",")]
lib = _lib.load()
g = torch.Generator().manual_seed(3)
tot = 0.0
for (H, cin, cout, k, stride) in SHAPES:
    pad = k // 2
    d, ho, wo = nn_ops._fwd_desc(N, H, H, cin, k, k, stride, pad, cin, cout, 0)
    x = torch.randn(N, H, H, cin, generator=g).to(dev)
    dy = torch.randn(N, ho, wo, cout, generator=g).to(dev)
    dw = torch.zeros(cout, cin, k, k, device=dev)
    M = N * ho * wo
    splits = lib.ccst_conv2d_bwd_weight_splits(M, cin, cout, k * k)
    if os.environ.get("BWDW_SPLITS"):
        splits = max(1, min(int(os.environ["BWDW_SPLITS"]), M // 128))
    ws = torch.empty(splits * k * k * cin * cout, device=dev)

    def run():
        check(lib.ccst_conv2d_bwd_weight_f32(ctypes.byref(d), ptr(x), ptr(dy), ptr(dw), splits, 0, ptr(ws), ws.numel() * 4,
                                             stream_ptr()), "bwd_weight")
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * M * cin * cout * k * k
    # reference: autograd of a torch conv (fp32) for a sanity difference
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2), (cout, cin, k, k), dy.permute(0, 3, 1, 2), stride=stride, padding=pad)
    err = float((dw - ref).abs().max() / ref.abs().max())
    tot += us
    print("%3dx%-3d %4d->%-4d k%d s%d  splits %3d  %7.1f us  %6.1f TF   rel.diff %.1e" % (H, H, cin, cout, k, stride, splits, us, fl / us / 1e6, err),
          flush=True)
print("sum %.1f us" % tot)
