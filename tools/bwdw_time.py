#!/usr/bin/env python3
"""Weight-gradient kernels (conv_bwd_weight.hip) alone on the ResNet50 B=64 222x222 layer shapes:
python tools/bwdw_time.py [reps]   -> per shape and kernel (fp32 MFMA / half pieces on the 16-bit MFMA): splits, us (kernel + slab
reduce), TFLOP/s, and the error against an fp64 reference (max |dw - ref| / max |ref|).
BWDW_SPLITS=<n> overrides the split count (workspace sized accordingly); BWDW_SHAPES=i,j,.. picks rows; BWDW_SCALE=<f> multiplies dy
(gradient-sized operands: 1e-6); BWDW_SWEEP=a,b,.. times the half-piece kernel at those workgroup targets (splits = target / tiles)
instead of the library's choice."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccst_amd import _lib, nn_ops, ops
from ccst_amd._lib import check, ptr, stream_ptr

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
N = 64
# (H = W of the conv input, cin, cout, k, stride)
SHAPES = [(56, 64, 64, 3, 1), (28, 128, 128, 3, 1), (14, 256, 256, 3, 1), (7, 512, 512, 3, 1),
          (56, 64, 256, 1, 1), (56, 256, 64, 1, 1), (28, 128, 512, 1, 1), (28, 512, 128, 1, 1),
          (14, 256, 1024, 1, 1), (14, 1024, 256, 1, 1), (7, 512, 2048, 1, 1), (7, 2048, 512, 1, 1),
          (56, 256, 128, 1, 1), (56, 128, 128, 3, 2), (56, 256, 512, 1, 2), (56, 64, 64, 1, 1), (28, 512, 256, 1, 1),
          (14, 1024, 512, 1, 1), (28, 256, 256, 3, 2), (14, 512, 512, 3, 2), (28, 512, 1024, 1, 2), (14, 1024, 2048, 1, 2)]
if os.environ.get("BWDW_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["BWDW_SHAPES"].split(",")]
dscale = float(os.environ.get("BWDW_SCALE", "1"))
lib = _lib.load()
g = torch.Generator().manual_seed(3)


def ref64(x, dy, cin, cout, k, stride, pad):
    """dw[co][ci][ky][kx] in fp64: one GEMM per tap over the shifted, strided input."""
    Nn, H, W, _ = x.shape
    _, ho, wo, _ = dy.shape
    xp = torch.zeros(Nn, H + 2 * pad, W + 2 * pad, cin, device=x.device, dtype=torch.float64)
    xp[:, pad:pad + H, pad:pad + W] = x.double()
    d2 = dy.double().reshape(-1, cout)
    out = torch.empty(cout, cin, k, k, device=x.device, dtype=torch.float64)
    for ky in range(k):
        for kx in range(k):
            xs = xp[:, ky:ky + (ho - 1) * stride + 1:stride, kx:kx + (wo - 1) * stride + 1:stride].reshape(-1, cin)
            out[:, :, ky, kx] = d2.t() @ xs
    return out


tot = {"f32": 0.0, "half": 0.0}
for (H, cin, cout, k, stride) in SHAPES:
    pad = k // 2
    d, ho, wo = nn_ops._fwd_desc(N, H, H, cin, k, k, stride, pad, cin, cout, 0)
    x = torch.randn(N, H, H, cin, generator=g).to(dev)
    dy = (torch.randn(N, ho, wo, cout, generator=g) * dscale).to(dev)
    M = N * ho * wo
    ref = ref64(x, dy, cin, cout, k, stride, pad)
    xmax, dmax = ops.absmax(x), ops.absmax(dy)
    line = "%3dx%-3d %4d->%-4d k%d s%d " % (H, H, cin, cout, k, stride)
    kinds = [("f32", None), ("half", None)]
    if os.environ.get("BWDW_SWEEP"):
        ti, tj = (cin + 127) // 128 if cin >= 128 else 1, (cout + 127) // 128 if cout >= 128 else 1
        kinds = [("half", max(1, int(t) // (ti * tj * k * k))) for t in os.environ["BWDW_SWEEP"].split(",")]
    for kind, forced in kinds:
        splits = (lib.ccst_conv2d_bwd_weight_split_splits if kind == "half" else lib.ccst_conv2d_bwd_weight_splits)(M, cin, cout, k * k)
        if forced is not None:
            splits = max(1, min(forced, M // 64))
        if os.environ.get("BWDW_SPLITS"):
            splits = max(1, min(int(os.environ["BWDW_SPLITS"]), M // 128))
        ws = torch.empty(splits * k * k * cin * cout, device=dev)
        dw = torch.zeros(cout, cin, k, k, device=dev)

        def run():
            if kind == "half":
                check(lib.ccst_conv2d_bwd_weight_split_f32(ctypes.byref(d), ptr(x), ptr(xmax), ptr(dy), ptr(dmax), ptr(dw), splits, 0, ptr(ws),
                                                           ws.numel() * 4, stream_ptr()), "bwd_weight_split")
            else:
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
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
        tot[kind] += us
        line += " | %-4s splits %3d %7.1f us %6.1f TF err %.1e" % (kind, splits, us, fl / us / 1e6, err)
    print(line, flush=True)
print("sum f32 %.1f us   half %.1f us" % (tot["f32"], tot["half"]))
