"""Where the HOST time of a ResNet train step goes: cProfile over a few eager steps (python tools/issue_profile.py [arch] [batch])."""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_resnet as B
dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model, opt, loss_fun, x, y = B.build(dev, arch=arch, batch=batch, seed=1)
step = B.make_step(model, opt, loss_fun, x, y)
for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(8):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
