#!/usr/bin/env python3
"""The streaming pointwise kernel's HALF-PIECE training forward (conv1x1_stream_kernel<.., HALFP>, BatchNorm-statistics epilogue) alone on
the ResNet50 B=64 222x222 pointwise shapes: python tools/pointwise_half_time.py [reps] -> per shape us, TFLOP/s, error vs fp64.
CCST_HIP_LIB selects a variant build (tools/variants/conv_igemm.hip.patch, IGA bits: timing experiments, wrong results)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccst_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
N = 64
SHAPES = [(56, 64, 256), (56, 256, 64), (28, 128, 512), (28, 512, 128), (14, 256, 1024), (14, 1024, 256), (7, 512, 2048), (7, 2048, 512), (56, 64, 64), (28, 256, 512)]
g = torch.Generator().manual_seed(5)
tot = 0.0
for (H, cin, cout) in SHAPES:
    x = torch.randn(N, H, H, cin, generator=g).to(dev)
    w = (torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).to(dev)
    pc = ops.pack_conv_weight(w, None)
    xmax, wmax = ops.absmax(x), ops.absmax(w)
    wsp = ops.pack_conv_weight_split(w, wmax)

    def run():
        return ops.conv2d_nhwc(x, pc, want_stats=True, x_absmax=xmax, w_absmax=wmax, w_split=wsp)
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
    M = N * H * H
    fl = 2.0 * M * cin * cout
    ref = (x[0].double().reshape(-1, cin) @ w.double().reshape(cout, cin).t())
    err = float((y[0].double().reshape(-1, cout) - ref).abs().max() / ref.abs().max())
    tot += us
    print("%3dx%-3d %4d->%-4d  %7.1f us  %6.1f TF  %6.0f GB/s   rel.err %.1e" % (H, H, cin, cout, us, fl / us / 1e6, 4.0 * (x.numel() + y.numel()) / us / 1e3, err), flush=True)
print("sum %.1f us" % tot)
