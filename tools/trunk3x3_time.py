"""ResNet trunk 3x3 stride-1 layers (B images): the F(2x2) train kernel (conv3x3_wino.hip) against the 64-channel F(4x4)
kernel (conv3x3_wino4w.hip), both with the statistics epilogue; times and the difference of their outputs.
  python tools/trunk3x3_time.py [reps]    env: TRUNK_B (64)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccst_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(os.environ.get("TRUNK_B", 64))
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for H, C in ((56, 64), (28, 128), (14, 256), (7, 512)):
    x = torch.randn(B, H, H, C, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dev)
    pk2 = ops.pack_wino(w)
    pc4 = ops.pack_conv_weight(w, None, wino=4)
    y2, s2 = ops.conv3x3_wino_train(x, pk2, want_stats=True)
    y4, s4 = ops.conv3x3_wino4(x, pc4, 0, sums=True)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    e2, e4 = float((y2 - ref).abs().max()), float((y4 - ref).abs().max())
    ds = float((s2.double().sum(0) - s4.double().sum(0)).abs().max() / s2.double().sum(0).abs().max())
    t2 = timed(lambda: ops.conv3x3_wino_train(x, pk2, want_stats=True))
    t4 = timed(lambda: ops.conv3x3_wino4(x, pc4, 0, sums=True))
    gf = 2.0 * B * H * H * C * C * 9 / 1e9
    print("%3dx%-3d C=%3d  F(2x2) %7.1f us (%5.1f TF, err %.2e)   F(4x4)w %7.1f us (%5.1f TF, err %.2e)   stats rel diff %.1e"
          % (H, H, C, t2, gf / t2 * 1e-3 * 1e3 / 1e3 * 1e3, e2, t4, gf / t4 * 1e-3 * 1e3 / 1e3 * 1e3, e4, ds))
