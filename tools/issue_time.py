"""Is the ResNet train step launch-bound?  Host time to ISSUE one step on an idle GPU (no synchronisation inside: fed.StepWindow does not
block within its first two steps) against the time until the GPU has finished it, and the steady-state step time of a run of steps.
python tools/issue_time.py [arch] [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import bench_resnet as B
dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model, opt, loss_fun, x, y = B.build(dev, arch=arch, batch=batch, seed=1)
for rep in range(6):
    step = B.make_step(model, opt, loss_fun, x, y)          # (a fresh StepWindow: nothing to wait for)
    if rep == 0:
        for _ in range(3):
            step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("one step on an idle GPU: host issue %.2f ms, done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
step = B.make_step(model, opt, loss_fun, x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print("20 steps back to back: %.2f ms/step" % ((time.perf_counter() - t0) * 50))
