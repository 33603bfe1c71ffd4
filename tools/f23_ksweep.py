"""Per-tile fixed cost of the F(2,3) kernel: the same spatial extent and Cout with Cin = 32 ... 512 (2 ... 32 chunks of 16 channels):
time per launch is linear in the chunk count, the intercept is what a tile pays outside its k-loop (set-up, prologue, epilogue,
workgroup turn-around).   python tools/f23_ksweep.py [reps]   env: KS_H (128), KS_COUT (256), KS_B (6)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccst_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
H, Cout, B = int(os.environ.get("KS_H", 128)), int(os.environ.get("KS_COUT", 256)), int(os.environ.get("KS_B", 6))
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
tiles = B * ((H + 7) // 8) * ((H + 31) // 32) * ((Cout + 127) // 128)
pts = []
for Cin in (32, 64, 128, 256, 512):
    x = torch.rand(B, H, H, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.05).to(dev)
    pc = ops.pack_conv_weight(w, b, wino=4)
    xmax = ops.absmax(x)
    fn = lambda: ops.conv3x3_f23(x, pc, 1 | 8, x_absmax=xmax)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    per_tile = us / max(1.0, -(-tiles // 256))
    pts.append((Cin // 16, us, per_tile))
    print("%3d^2 %3d->%3d  %2d chunks  %7.1f us per launch  %6.2f us per tile-round (%d tiles = %.2f rounds)" % (H, Cin, Cout, Cin // 16, us, per_tile, tiles, tiles / 256.0))
(c0, _, t0), (c1, _, t1) = pts[-2], pts[-1]
slope = (t1 - t0) / (c1 - c0)
print("per chunk %.2f us; intercept (per tile, outside the k-loop) %.2f us" % (slope, t1 - slope * c1))
