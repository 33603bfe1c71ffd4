"""Second half of the BASELINE metric: ResNet50 train-step images/sec @222x222, B=64 (the body of
federated/fed_run.py:49-80: zero_grad -> forward -> CrossEntropy -> backward -> SGD step), synthetic
data resident in HBM.  Called by bench.py (``secondary``) and runnable on its own:

    python -m ccst_amd.bench_resnet [--steps K] [--warmup W] [--batch B] [--arch resnet50] [--graph]
"""
import argparse
import json
import os
import time
import types

import torch

GFLOP_PER_IMAGE = {"resnet50": 24.51, "resnet18": 10.87}      # SURVEY.md 8d (fwd + bwd-data + bwd-weight), 222x222
PEAK_F32_MFMA_TFLOPS = 157.3


def build(dev, arch="resnet50", classes=7, batch=64, size=222, lr=0.001, seed=1):
    from . import fed
    from .nets import models
    torch.manual_seed(seed)                                   # fed_run.py:495,510-514
    args = types.SimpleNamespace(dg_method="")
    model = models.get_network(arch)(args, pretrained=False, classes=classes).to(dev)
    model.train()
    opt = fed.SGD(model, lr=lr)
    loss_fun = fed.CrossEntropyLoss()
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(batch, 3, size, size, generator=g).to(dev)
    y = torch.randint(0, classes, (batch,), generator=g).to(dev)
    return model, opt, loss_fun, x, y


def make_step(model, opt, loss_fun, x, y):
    def step():
        opt.zero_grad()
        loss = loss_fun(model(x), y)
        loss.backward()
        opt.step()
        return loss
    return step


def layer_table(step):
    """Per-launch conv table of one train step (HIP events), to stderr."""
    import sys
    from . import ops
    ops.TIMING = []
    step()
    torch.cuda.synchronize()
    timing, ops.TIMING = ops.TIMING, None
    tot = {}
    for name, flops, e0, e1, info in timing:
        us = e0.elapsed_time(e1) * 1e3
        print("%-44s %-52s %8.1f us %6.1f TF" % (name, info, us, flops / us / 1e6), file=sys.stderr)
        k = name.split(":")[0] if ":" in name else ("fwd" if name.startswith("conv_igemm") else name)
        t = tot.setdefault(k, [0.0, 0.0])
        t[0] += us
        t[1] += flops
    for k, (us, fl) in tot.items():
        print("TOTAL %-12s %9.1f us  %7.1f GFLOP  %6.1f TF" % (k, us, fl / 1e9, fl / us / 1e6), file=sys.stderr)


def run(dev, world=1, steps=10, warmup=3, batch=64, arch="resnet50", graph=False, cpu_baseline=False, layers=False):
    model, opt, loss_fun, x, y = build(dev, arch=arch, batch=batch)
    step = make_step(model, opt, loss_fun, x, y)
    for _ in range(warmup):
        loss = step()
    torch.cuda.synchronize()
    if layers:
        layer_table(step)
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = step()
        torch.cuda.synchronize()
        run_step = g.replay
    else:
        run_step = step
    t0 = time.perf_counter()
    for _ in range(steps):
        run_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gflop = GFLOP_PER_IMAGE.get(arch, 0.0) * batch
    out = {"metric": "%s train images/sec @222x222 B=%d" % (arch, batch), "value": round(batch / dt, 2), "unit": "images/sec",
           "ms_per_step": round(dt * 1e3, 3), "steps": steps, "warmup": warmup, "dtype": "f32", "hip_graph": bool(graph),
           "tflops": round(gflop / dt / 1e3, 2), "frac_of_f32_mfma_peak": round(gflop / dt / 1e3 / PEAK_F32_MFMA_TFLOPS, 4),
           "final_loss": round(float(loss), 5)}
    if cpu_baseline:
        from oracle import resnet_ref as R
        torch.set_num_threads(os.cpu_count() or 1)
        ref = R.resnet50(7) if arch == "resnet50" else R.resnet18(7)
        nb = 8
        xc, yc = R.synth_batch(nb, 222, 7, seed=2)
        R.train_step(ref, xc[:2], yc[:2], 0.001)
        c0 = time.perf_counter()
        R.train_step(ref, xc, yc, 0.001)
        c1 = time.perf_counter()
        out["cpu_baseline"] = {"value": round(nb / (c1 - c0), 3), "unit": "images/sec", "cores": torch.get_num_threads(),
                               "kind": "port", "sample": "1 train step at B=%d (oracle/resnet_ref.py, torch CPU fp32)" % nb}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--cpu-baseline", action="store_true")
    ap.add_argument("--layers", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(torch.device("cuda:0"), steps=a.steps, warmup=a.warmup, batch=a.batch, arch=a.arch, graph=a.graph,
                         cpu_baseline=a.cpu_baseline, layers=a.layers)))
