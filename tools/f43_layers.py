"""Per-layer time of the AdaIN path's 3x3 layers at B=6, 512x512 on the F(4,3) kernel (conv3x3_f43.hip) beside the direct half-piece
kernel (conv3x3_halo.hip SPLIT): python tools/f43_layers.py [reps] -> per layer us, algorithmic TFLOP/s, error vs fp64.
F43_LAYER=i,j,...: only those rows."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccst_amd import ops  # noqa: E402

LAYERS = [  # H (= W, conv extent), Cin, Cout, pool, ups
    (256, 64, 128, False, False), (256, 128, 128, True, False), (128, 128, 256, False, False), (128, 256, 256, False, False),
    (128, 256, 256, True, False), (64, 256, 512, False, False), (64, 512, 256, False, False), (128, 256, 256, False, True),
    (128, 256, 128, False, False), (256, 128, 128, False, True),
    (512, 64, 64, True, False), (256, 128, 64, False, False), (512, 64, 64, False, True),        # the Cout = 64 layers (F(4,3)'s 64-channel tile)
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    B = int(os.environ.get("F43_B", "6"))
    g = torch.Generator().manual_seed(3)
    tot = [0.0, 0.0]
    sel = os.environ.get("F43_LAYER", os.environ.get("F23_LAYER"))
    for (H, Cin, Cout, pool, ups) in (LAYERS if sel is None else [LAYERS[int(i)] for i in sel.split(",")]):
        Hs = H // 2 if ups else H
        x = torch.rand(B, Hs, Hs, Cin, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
        b = (torch.randn(Cout, generator=g) * 0.05).to(dev)
        pc = ops.pack_conv_weight(w, b, wino=4)
        flags = 1 | 8 | (2 if pool else 0) | (4 if ups else 0)
        xmax = ops.absmax_samples(x)
        fns = (lambda: ops.conv3x3_halo_split(x, pc, flags, x_absmax=xmax), lambda: ops.conv3x3_f43(x, pc, flags, x_absmax=xmax))
        res = []
        for fn in fns:
            for _ in range(2):
                y = fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                y = fn()
            e1.record()
            torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) * 1e3 / reps, y))
        # fp64 reference of image 0
        xr = x[:1].permute(0, 3, 1, 2).double()
        if ups:
            xr = F.interpolate(xr, scale_factor=2, mode="nearest")
        ref = F.relu(F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), w.double(), b.double()))
        if pool:
            ref = F.max_pool2d(ref, 2, 2, 0, ceil_mode=True)
        ref = ref.permute(0, 2, 3, 1)
        errs = [float((r[1][:1].double() - ref).abs().max()) / float(ref.abs().max()) for r in res]
        fl = 2.0 * B * H * H * Cout * Cin * 9
        wgs = B * ((H + 7) // 8) * ((H + 31) // 32) * ((Cout + 127) // 128) if Cout >= 128 else B * ((H + 7) // 8) * ((H + 31) // 32)
        print("%4d^2 %3d->%3d %s%s  split %7.1f us %6.1f TF err %.2e | f43 %7.1f us %6.1f TF err %.2e | split/f43 x%.2f  (%d workgroups = %.2f rounds)" % (
            H, Cin, Cout, "pool " if pool else "     ", "ups" if ups else "   ", res[0][0], fl / res[0][0] / 1e6, errs[0], res[1][0], fl / res[1][0] / 1e6, errs[1],
            res[0][0] / res[1][0], wgs, wgs / 256.0))
        for k in range(2):
            tot[k] += res[k][0]
    print("sum of the layers: split %.1f us, f43 %.1f us" % tuple(tot))


if __name__ == "__main__":
    main()
