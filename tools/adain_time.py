import os, sys, torch
sys.path.insert(0, os.getcwd())
from ccst_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.rand(6, 64, 64, 512, generator=g).to(dev)          # NHWC buffer
xa = x.permute(0, 3, 1, 2)                                   # logical NCHW, channels_last
sm, ss = torch.rand(512).to(dev), (torch.rand(512) + 0.5).to(dev)
for _ in range(5):
    y = ops.adain(xa, sm, ss)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    y = ops.adain(xa, sm, ss)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
print("adain %.1f us  %.2f TB/s" % (us, 100.66e6 / us / 1e6))
