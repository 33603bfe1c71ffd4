"""Stress of the 'capture a train step, replay it, drop the graph, go on eagerly' flow (what bench_resnet's and fed.train's auto loop choice
do when the eager loop wins): python tools/repro_fork.py [rounds] [arch] [batch].  CCST_GRAPH_FORK_BATCH=1 / 8 to compare."""
import gc
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench_resnet
from ccst_amd import nn_ops, ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
arch = sys.argv[2] if len(sys.argv) > 2 else "resnet50"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
model, opt, loss_fun, x, y = bench_resnet.build(dev, arch=arch, batch=batch, classes=7)
step = bench_resnet.make_step(model, opt, loss_fun, x, y)
if os.environ.get("FREEZE") == "1":
    gc.collect()
    gc.freeze()
if os.environ.get("GC_STRESS") == "1":          # a full collection every few dozen allocations: does the cyclic GC ever take a live graph apart?
    gc.set_threshold(40, 2, 2)
fails = 0
for r in range(rounds):
    try:
        for _ in range(4):
            step()
        gstep = bench_resnet.make_step(model, opt, loss_fun, x, y, join_side=True)
        for _ in range(2):
            gstep()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        ops.reset_absmax_pool()
        nn_ops.reset_deferred()
        with torch.cuda.graph(g):
            gloss = gstep()
        ops.reset_absmax_pool()
        torch.cuda.synchronize()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        del g
        ops.bump_weights_epoch()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    except Exception:
        fails += 1
        print("round %d failed:" % r)
        traceback.print_exc(limit=3)
        torch.cuda.synchronize()
print("rounds %d, failures %d, K=%d" % (rounds, fails, nn_ops.GRAPH_FORK_BATCH))
