"""Times the two image-edge kernels of the AdaIN path on their bench shapes (B images of SxS): the first layer
(conv_stem3.hip, NCHW image -> NHWC 64 channels) and the last (conv_small.hip, NHWC 64 channels -> NCHW image).
  python tools/edge_time.py [reps]      env: EDGE_B (6), EDGE_S (512), CCST_HIP_LIB for a variant library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccst_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, S = int(os.environ.get("EDGE_B", 6)), int(os.environ.get("EDGE_S", 512))
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
img = torch.rand(B, 3, S, S, generator=g).to(dev)
w1 = (torch.randn(64, 3, 3, 3, generator=g) * 0.2).to(dev)
b1 = torch.randn(64, generator=g).to(dev)
wa = ops.pack_stem3(w1, b1)
feat = torch.randn(B, S, S, 64, generator=g).to(dev)
w2 = (torch.randn(3, 64, 3, 3, generator=g) * 0.05).to(dev)
b2 = torch.randn(3, generator=g).to(dev)
w2t = w2.permute(2, 3, 0, 1).contiguous()       # [3][3][Cout][Cin], as net.py hands it over


pz = ops.PackedZform(w2t) if hasattr(ops, "PackedZform") else None
fwords = ops.absmax_samples(feat)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


out_bytes = B * S * S * 64 * 4
for name, fn, nbytes in (("stem3", lambda: ops.conv3x3_stem3_nchw(img, wa, relu=True), out_bytes + B * 3 * S * S * 4),
                         ("zform", lambda: ops.conv3x3_zform_nchw(feat, pz, b2, 3, reflect=True, relu=False, x_absmax=fwords), out_bytes + B * 3 * S * S * 4),
                         ("absmax", lambda: ops.absmax(feat), out_bytes),
                         ("fill", lambda: feat.fill_(1.0), out_bytes),
                         ("copy", lambda: feat.clone(), 2 * out_bytes)):
    try:
        med, best = timed(fn)
        print("%-8s median %7.1f us  best %7.1f us   %6.2f TB/s algorithmic" % (name, med, best, nbytes / med / 1e6))
    except Exception as e:  # a variant library may lack one of the paths
        print("%-8s failed: %s" % (name, e))
