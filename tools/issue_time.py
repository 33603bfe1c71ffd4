"""Is the ResNet train step launch-bound?  Host time to ISSUE ten steps (no sync) vs the time until the GPU
finishes them.  Measured on MI355X: 13 ms/step to issue, 24 ms/step to execute => GPU-bound, the host runs ahead
(which is also why a HIP-graph replay of the step is no faster than eager).   python tools/issue_time.py [arch] [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, types
import bench_resnet as B
dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model, opt, loss_fun, x, y = B.build(dev, arch=arch, batch=batch, seed=1)
step = B.make_step(model, opt, loss_fun, x, y)
for _ in range(3): step()
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(10): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("issue ms/step %.2f   total ms/step %.2f" % ((t1 - t0) * 100, (t2 - t0) * 100))
