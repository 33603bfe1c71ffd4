#!/usr/bin/env python3
"""The reference's OWN software stack on this GPU, as a yardstick: the two hot paths written the way the reference writes them --
stock torch.nn modules and autograd on PyTorch-ROCm, i.e. MIOpen / rocBLAS / ATen kernels -- timed on the bench shapes.

  python tools/torch_stack_baseline.py [adain|resnet|both] [steps]

  * AdaIN: encoder (net.py:38-69, first 31 modules) -> calc_mean_std / adaIN_StyleStat_ContentFeat (function.py:4-33) -> alpha blend
    -> decoder (net.py:6-36), B=6 512x512 fp32, no_grad, inputs resident (bench.py's `value` workload).
  * ResNet50 (torchvision v1.5 bottlenecks, nets/resnet.py:132-191), B=64 222x222, 7 classes, CrossEntropyLoss, SGD(lr=1e-3),
    zero_grad / forward / backward / step (fed_run.py:31-80; bench_resnet.py's workload).
Each in the reference's default settings (NCHW, cudnn.benchmark off) and in the fastest settings stock PyTorch offers
(channels_last and/or cudnn.benchmark = MIOpen's kernel search).  Nothing here is used by the product or the tests: it imports
neither ccst_amd nor oracle; the module lists are built from channel tables."""
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

what = sys.argv[1] if len(sys.argv) > 1 else "both"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")


def encoder():
    layers = [nn.Conv2d(3, 3, 1)]
    for c in ((3, 64), (64, 64), "P", (64, 128), (128, 128), "P", (128, 256), (256, 256), (256, 256), (256, 256), "P", (256, 512)):
        if c == "P":
            layers.append(nn.MaxPool2d(2, 2, 0, ceil_mode=True))
        else:
            layers += [nn.ReflectionPad2d(1), nn.Conv2d(c[0], c[1], 3), nn.ReLU()]
    return nn.Sequential(*layers)


def decoder():
    layers = []
    for c in ((512, 256), "U", (256, 256), (256, 256), (256, 256), (256, 128), "U", (128, 128), (128, 64), "U", (64, 64), (64, 3)):
        if c == "U":
            layers.append(nn.Upsample(scale_factor=2, mode="nearest"))
        else:
            layers += [nn.ReflectionPad2d(1), nn.Conv2d(c[0], c[1], 3)] + ([nn.ReLU()] if c[1] != 3 else [])
    return nn.Sequential(*layers)


def mean_std(feat, eps=1e-5):
    n, c = feat.shape[:2]
    var = feat.reshape(n, c, -1).var(dim=2) + eps
    return feat.reshape(n, c, -1).mean(dim=2).view(n, c, 1, 1), var.sqrt().view(n, c, 1, 1)


def style_transfer(enc, dec, content, stat, alpha=1.0):
    f = enc(content)
    m, s = mean_std(f)
    t = (f - m.expand(f.shape)) / s.expand(f.shape) * stat[1].expand(f.shape) + stat[0].expand(f.shape)
    return dec(t * alpha + f * (1 - alpha))


class Bottleneck(nn.Module):
    def __init__(self, cin, mid, stride):
        super().__init__()
        self.c1, self.b1 = nn.Conv2d(cin, mid, 1, bias=False), nn.BatchNorm2d(mid)
        self.c2, self.b2 = nn.Conv2d(mid, mid, 3, stride, 1, bias=False), nn.BatchNorm2d(mid)
        self.c3, self.b3 = nn.Conv2d(mid, 4 * mid, 1, bias=False), nn.BatchNorm2d(4 * mid)
        self.down = None
        if stride != 1 or cin != 4 * mid:
            self.down = nn.Sequential(nn.Conv2d(cin, 4 * mid, 1, stride, bias=False), nn.BatchNorm2d(4 * mid))

    def forward(self, x):
        y = F.relu(self.b1(self.c1(x)))
        y = F.relu(self.b2(self.c2(y)))
        y = self.b3(self.c3(y))
        return F.relu(y + (x if self.down is None else self.down(x)))


def resnet50(classes=7):
    layers = [nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1)]
    cin = 64
    for mid, n, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
        for i in range(n):
            layers.append(Bottleneck(cin, mid, stride if i == 0 else 1))
            cin = 4 * mid
    return nn.Sequential(*layers, nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(cin, classes))


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) * 1e3 / n


torch.manual_seed(0)
for bench, cl in ((False, False), (True, False), (True, True)):
    torch.backends.cudnn.benchmark = bench
    fmt = torch.channels_last if cl else torch.contiguous_format
    tag = "cudnn.benchmark=%s %s" % (bench, "channels_last" if cl else "NCHW")
    if what in ("adain", "both"):
        enc, dec = encoder().to(dev).to(memory_format=fmt).eval(), decoder().to(dev).to(memory_format=fmt).eval()
        x = torch.rand(6, 3, 512, 512, device=dev).to(memory_format=fmt)
        stat = (torch.rand(1, 512, 1, 1, device=dev), torch.rand(1, 512, 1, 1, device=dev) + 0.5)
        with torch.no_grad():
            ms, wall = timed(lambda: style_transfer(enc, dec, x, stat), steps)
        print("adain   B=6 512x512 fp32   %-40s %8.2f ms/batch  %8.1f images/s   (host wall %.2f ms)" % (tag, ms, 6e3 / ms, wall), flush=True)
        del enc, dec, x
    if what in ("resnet", "both"):
        model = resnet50().to(dev).to(memory_format=fmt).train()
        opt = torch.optim.SGD(model.parameters(), lr=1e-3)
        lossf = nn.CrossEntropyLoss()
        x = torch.randn(64, 3, 222, 222, device=dev).to(memory_format=fmt)
        y = torch.randint(0, 7, (64,), device=dev)

        def step():
            opt.zero_grad()
            loss = lossf(model(x), y)
            loss.backward()
            opt.step()
        ms, wall = timed(step, steps)
        print("resnet50 B=64 222x222 fp32 %-40s %8.2f ms/step   %8.1f images/s   (host wall %.2f ms)" % (tag, ms, 64e3 / ms, wall), flush=True)
        del model, opt, x, y
    torch.cuda.empty_cache()
