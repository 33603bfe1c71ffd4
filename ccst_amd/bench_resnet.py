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


def host_cores():
    """CPU cores this process may actually use: min(affinity, cgroup quota) -- the GPU boxes expose 256
    logical CPUs but cap the container at a 16-core quota, and oversubscribed torch threads run 5x slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


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


def make_step(model, opt, loss_fun, x, y, join_side=False):
    def step():
        opt.zero_grad()
        loss = loss_fun(model(x), y)
        loss.backward()
        opt.step()
        if join_side:         # graph capture: every forked stream must re-join before the capture ends
            from . import nn_ops
            nn_ops.join_prepack(x.device)
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
        k = name.split(":")[0] if ":" in name else ("fwd" if name.startswith("conv") else name)
        t = tot.setdefault(k, [0.0, 0.0])
        t[0] += us
        t[1] += flops
    for k, (us, fl) in tot.items():
        print("TOTAL %-12s %9.1f us  %7.1f GFLOP  %6.1f TF" % (k, us, fl / 1e9, fl / us / 1e6), file=sys.stderr)


def run(dev, world=1, steps=10, warmup=3, batch=64, arch="resnet50", graph=False, cpu_baseline=False, layers=False):
    """One client per rank (weak scaling).  With world > 1 (torch.distributed already initialised by the
    caller) the timed region is K local train steps followed by ONE FedAvg all-reduce of the flat state
    (fed_run.py's round: local epoch(s) then communication()), barrier-bracketed, max over ranks."""
    import torch.distributed as dist
    from . import fed
    if graph and os.environ.get("CCST_GRAPH_SIDE", "1") == "0":       # single-stream capture
        from . import nn_ops
        nn_ops.SIDE_STREAM = False
    distributed = world > 1 and dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if distributed else 0
    model, opt, loss_fun, x, y = build(dev, arch=arch, batch=batch, seed=1 + rank)
    step = make_step(model, opt, loss_fun, x, y, join_side=graph)
    args = types.SimpleNamespace(mode="fedavg")
    for _ in range(warmup):
        loss = step()
    if distributed:
        fed.communication_distributed(args, model, 1.0 / world)
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
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_step()
    if distributed:
        fed.communication_distributed(args, model, 1.0 / world)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    total = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([total], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        total = float(tt.item())
    dt = total / steps
    gflop = GFLOP_PER_IMAGE.get(arch, 0.0) * batch
    out = {"metric": "%s train images/sec @222x222 B=%d" % (arch, batch), "value": round(world * batch / dt, 2), "unit": "images/sec",
           "n_gpus": world, "ms_per_step": round(dt * 1e3, 3), "steps": steps, "warmup": warmup, "dtype": "f32",
           "hip_graph": bool(graph), "scaling": "weak",
           "config": {"workload": "fed_run.py train() body, %s classes=7, SGD lr 0.001, one client per GPU%s"
                      % (arch, ", + 1 FedAvg all-reduce (RCCL) per %d steps" % steps if distributed else "")},
           "tflops_per_gpu": round(gflop / dt / 1e3, 2), "frac_of_f32_mfma_peak": round(gflop / dt / 1e3 / PEAK_F32_MFMA_TFLOPS, 4),
           "final_loss": round(float(loss.detach()), 5)}
    if cpu_baseline and rank == 0 and world == 1:
        from oracle import resnet_ref as R
        torch.set_num_threads(host_cores())
        ref = R.resnet50(7) if arch == "resnet50" else R.resnet18(7)
        nb, reps = batch, 4            # the metric's own batch; ~10 s of host work on 16 cores
        xc, yc = R.synth_batch(nb, 222, 7, seed=2)
        R.train_step(ref, xc[:2], yc[:2], 0.001)
        c0 = time.perf_counter()
        for _ in range(reps):
            R.train_step(ref, xc, yc, 0.001)
        c1 = time.perf_counter()
        out["cpu_baseline"] = {"value": round(reps * nb / (c1 - c0), 3), "unit": "images/sec", "cores": torch.get_num_threads(),
                               "kind": "port",
                               "sample": "%d train steps at B=%d (oracle/resnet_ref.py, torch CPU fp32)" % (reps, nb)}
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
