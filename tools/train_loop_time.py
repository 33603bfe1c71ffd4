"""fed.train() per-iteration time with and without args.hip_graph:  python tools/train_loop_time.py resnet18 32"""
import os, sys, time, types, torch
sys.path.insert(0, os.getcwd())
from ccst_amd import fed
from ccst_amd.nets import models
from oracle import resnet_ref as R
dev = torch.device("cuda:0")
arch, B = sys.argv[1], int(sys.argv[2])
xs = [tuple(t.to(dev) for t in R.synth_batch(B, 222, 7, seed=300 + i)) for i in range(4)]
loader = [xs[i % 4] for i in range(30)]
for mode in (False, True):
    args = types.SimpleNamespace(mode="fedavg", dg_method="no_DG", hip_graph=mode)
    model = models.get_network(arch)(args, pretrained=False, classes=7).to(dev)
    ce = fed.CrossEntropyLoss()
    fed.train(model, loader[:5], fed.SGD(model, lr=0.001), ce, 1, dev, args, 0, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = fed.train(model, loader, fed.SGD(model, lr=0.001), ce, 1, dev, args, 1, None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(arch, "B", B, "hip_graph", mode, "%.2f ms/iter  %.0f img/s" % (dt / 30 * 1e3, 30 * B / dt), r)
