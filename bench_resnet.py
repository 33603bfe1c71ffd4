"""Second half of the BASELINE metric: ResNet50 train-step images/sec @222x222, B=64 (the body of
federated/fed_run.py:49-80: zero_grad -> forward -> CrossEntropy -> backward -> SGD step), synthetic
data resident in HBM.  Called by bench.py (``secondary``) and runnable on its own:

    python bench_resnet.py [--steps K] [--warmup W] [--batch B] [--arch resnet50] [--graph]

Lives next to bench.py, outside the ``ccst_amd`` package: its ``cpu_baseline`` leg imports the oracle, which nothing under
``ccst_amd/`` may do.
"""
import argparse
import gc
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE = {"resnet50": 24.51, "resnet18": 10.87}      # SURVEY.md 8d (fwd + bwd-data + bwd-weight), 222x222
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2516.6                               # the pipe the hot GEMMs run on (three half-piece products per fp32 product)
PEAK_HBM_GBPS = 8000.0                                      # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# ALGORITHMIC HBM bytes of a train step (SURVEY.md 8d "Roofline 2"): every conv output (fp32, 44.40 MB per image for ResNet50 at 222 x 222,
# 9.88 MB for ResNet18: the sum over the convs of Cout x Ho x Wo x 4 B) touched ~8.8 times per step by a fully fused schedule (written by the
# conv, read by the BatchNorm apply, written normalised, read by the next conv and by its weight gradient, and the backward's gradient
# passes) => 25 GB per B=64 ResNet50 step.  The step is HBM-bound: 3.1 ms at 8 TB/s against 1.9 ms of MFMA time on the 16-bit pipe.
EAGER_MARGIN = 1.03                                         # auto loop choice: eager must beat the replay by this factor
CONV_OUT_MB_PER_IMAGE = {"resnet50": 44.40, "resnet18": 9.88}
TOUCHES_PER_STEP = 8.8


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
    from ccst_amd import fed
    from ccst_amd.nets import models
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
    """join_side=True: the form that is captured into a HIP graph -- ROTATED as fed._GraphedTrainStep: the re-pack of the weights
    the previous step's SGD wrote opens the step (side stream, joined where the forward first reads a packed weight), so that the
    capture ends with every stream joined without serialising the re-pack behind the optimiser step."""
    from ccst_amd import fed, nn_ops

    window = fed.StepWindow() if (x.is_cuda and not join_side) else None      # as fed.train(): the host stays <= 2 steps ahead

    def step():
        if join_side:
            nn_ops.prepack_on_side(model)
        opt.zero_grad()
        loss = loss_fun(model(x), y)
        fed.backward(loss) if x.is_cuda else loss.backward()
        if join_side:
            opt.step(prepack=False)
            nn_ops.join_prepack(x.device)
        else:
            opt.step()
        if window is not None:
            window.tick()
        return loss
    return step


def layer_table(step):
    """Per-launch conv table of one train step (HIP events), to stderr."""
    from ccst_amd import ops
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


def build_stamp():
    """Digest of the kernel sources the loaded library was built from (ccst_amd/build.py writes it next to the .so)."""
    try:
        with open(os.path.join(ROOT, "ccst_amd", "csrc", ".build_stamp")) as f:
            return f.read().strip()[:16]
    except OSError:
        return None


def hbm_traffic_per_step(arch, batch):
    """(HBM bytes one train step moves, provenance) from the separate rocprofv3 --pmc passes over this same script (FETCH_SIZE x2
    per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE, summed over the kernels of the steady-state steps;
    tools/profile_resnet.sh writes profiles/traffic_resnet.json with the build stamp and date of the profiled build).  The number
    is a committed profile, not a measurement of this run: it is reported only when the profiled build IS the running build."""
    f = os.path.join(ROOT, "profiles", "traffic_resnet.json")
    if not os.path.exists(f):
        return None, None
    with open(f) as fh:
        tj = json.load(fh)
    key = "%s_b%d" % (arch, batch)
    src = {"file": "profiles/traffic_resnet.json", "build_stamp": tj.get("build_stamp"), "date": tj.get("date"),
           "profile": tj.get("profile"), "current_build": tj.get("build_stamp") is not None and tj.get("build_stamp") == build_stamp()}
    if key not in tj:
        return None, None
    if not src["current_build"]:
        src["note"] = "profiled build differs from the running one: bytes not reported"
        return None, src
    return tj.get(key), src


def run(dev, world=1, steps=10, warmup=3, batch=64, arch="resnet50", graph=False, cpu_baseline=False, layers=False,
        build_fn=None, scale_fn=None, sync=None, classes=7):
    """One client per rank (weak scaling).  With world > 1 (torch.distributed already initialised by the
    caller) the timed region is K local train steps followed by ONE FedAvg all-reduce of the flat state
    (fed_run.py's round: local epoch(s) then communication()), barrier-bracketed, max over ranks.
    build_fn / scale_fn / sync are injection points for the CPU (gloo) test of this protocol: a stand-in model + step,
    a host pre-scale instead of the HIP one, and a no-op instead of torch.cuda.synchronize."""
    import torch.distributed as dist
    from ccst_amd import fed
    sync = sync or torch.cuda.synchronize
    auto_graph = graph == "auto"
    graph = False if auto_graph else bool(graph)
    distributed = world > 1 and dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if distributed else 0
    n_ranks_seen = dist.get_world_size() if distributed else 1
    bkw = {"classes": classes} if build_fn is None else {}
    model, opt, loss_fun, x, y = (build_fn or build)(dev, arch=arch, batch=batch, seed=1 + rank, **bkw)
    step = make_step(model, opt, loss_fun, x, y, join_side=graph)
    args = types.SimpleNamespace(mode="fedavg")
    comm_kw = {"scale_fn": scale_fn} if scale_fn is not None else {}
    gc.collect()        # before the warm-up, see below
    gc.freeze()         # (what is alive now leaves the collector's generations: later passes scan new objects only)
    for _ in range(warmup):
        loss = step()
    if distributed:
        fed.communication_distributed(args, model, 1.0 / world, **comm_kw)
    sync()
    if layers:
        layer_table(step)
    graph_choice = None

    def capture():
        gstep = make_step(model, opt, loss_fun, x, y, join_side=True)
        for _ in range(2):
            gstep()
        sync()
        from ccst_amd import nn_ops, ops
        g = torch.cuda.CUDAGraph()
        ops.reset_absmax_pool()             # as fed._GraphedTrainStep: the step's |max| word rows come from a block zero-filled INSIDE the graph
        nn_ops.reset_deferred()
        with torch.cuda.graph(g):
            l = gstep()
        ops.reset_absmax_pool()
        sync()
        return g, l
    if auto_graph:
        # Eager or replayed?  Both loops give the same bits; which is faster depends on the host (the eager loop issues ~590 launches per
        # step) -- so MEASURE: `probe` steps of each, bracketed by synchronisation, and run the timed region on the faster one.
        probe = max(30, int(os.environ.get("CCST_BENCH_LOOP_PROBE", "30")))

        def timed(fn):
            sync()
            h0 = time.perf_counter()
            for _ in range(probe):
                fn()
            sync()
            return (time.perf_counter() - h0) * 1e3 / probe
        # host issue time of the eager loop alone (no window, no waiting): what a slower host would be bound by
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter()
        e0.record()
        for _ in range(6):
            step()
        e1.record()
        issue_ms = (time.perf_counter() - h0) * 1e3 / 6
        sync()
        eager_ms = timed(step)
        g, gloss = capture()
        for _ in range(3):
            g.replay()
        graph_ms = timed(g.replay)
        # (a replay's time does not depend on the host; a host-bound eager loop's does -- ResNet18 at B=32 issues in about its device time and
        #  read 4.27 ms in one probe and 4.98 in the timed region that followed: the eager loop has to win by a margin)
        graph = graph_ms < eager_ms * EAGER_MARGIN
        graph_choice = {"host_issue_ms_per_step": round(issue_ms, 3), "eager_ms_per_step": round(eager_ms, 3),
                        "graph_ms_per_step": round(graph_ms, 3), "probe_steps": probe,
                        "rule": "both loops timed over probe_steps; the eager loop runs the timed region if it is more than %d %% faster than the "
                                "replay (whose time does not depend on the host), else the replay" % round((EAGER_MARGIN - 1) * 100),
                        "hip_graph": bool(graph)}
        if not graph:
            from ccst_amd import ops
            ops.bump_weights_epoch()        # the replays moved the weights behind the host-side pack keys
            try:
                for _ in range(2):
                    step()
                sync()
                run_step = step
            except RuntimeError as e:       # (seen ONCE in ~30 driver-style runs of round 6: autograd refused the first eager backward after
                # the capture -- "backward through the graph a second time"; not reproduced in 100 rounds of tools/repro_fork.py.  The
                # replay gives the same bits: run the timed region on it and say so in the line.)
                graph_choice["eager_after_capture_error"] = str(e)[:200]
                graph_choice["hip_graph"] = graph = True
                sync()
        if graph:
            loss, run_step = gloss, g.replay
        else:
            del g
    elif graph:
        g, loss = capture()
        run_step = g.replay
    else:
        run_step = step
    # (a full collection of the interpreter's cyclic GC takes 60-80 ms in a process that has imported torch + the AdaIN bench, and its
    # allocation-count trigger landed deterministically inside the first timed step: that step's host issue 80 ms instead of 12, 2830
    # instead of 3160 images/s over 25 steps.  It runs BEFORE the warm-up: between warm-up and timed steps it would leave the GPU idle
    # long enough to drop its clocks.)
    if distributed:
        dist.barrier()
        sync()
    if os.environ.get("CCST_BENCH_MEM_TRACE") == "1":
        st = torch.cuda.memory_stats()
        print("mem before: reserved %.2f GB, device allocs %d, frees %d" % (st["reserved_bytes.all.current"] / 1e9, st["num_device_alloc"], st["num_device_free"]), file=sys.stderr)
    trace = os.environ.get("CCST_BENCH_STEP_TRACE") == "1"       # per-step host and device times to stderr (diagnosis only)
    evs, host_ts = [], []
    t0 = time.perf_counter()
    htrace = os.environ.get("CCST_BENCH_HOST_TRACE") == "1"
    hts = []
    dev_ev = None
    if sync is torch.cuda.synchronize:
        dev_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        dev_ev[0].record()
    for _ in range(steps):
        if htrace:
            hts.append(time.perf_counter())
        if trace:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
            host_ts.append(time.perf_counter())
        run_step()
    if trace:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
        host_ts.append(time.perf_counter())
        sync()
        print("step trace (device ms | host issue ms): " + " ".join("%.1f|%.1f" % (evs[i].elapsed_time(evs[i + 1]), (host_ts[i + 1] - host_ts[i]) * 1e3)
                                                                    for i in range(steps)), file=sys.stderr)
    if dev_ev is not None:
        dev_ev[1].record()
    allreduce_ms = None
    if distributed:
        sync()
        ta = time.perf_counter()
        fed.communication_distributed(args, model, 1.0 / world, **comm_kw)
        sync()
        allreduce_ms = (time.perf_counter() - ta) * 1e3
    sync()
    if distributed:
        dist.barrier()
        sync()
    total = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([total, allreduce_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        total, allreduce_ms = float(tt[0].item()), float(tt[1].item())
    if os.environ.get("CCST_BENCH_MEM_TRACE") == "1":
        st = torch.cuda.memory_stats()
        print("mem after: reserved %.2f GB, device allocs %d, frees %d, total %.1f ms" % (st["reserved_bytes.all.current"] / 1e9, st["num_device_alloc"], st["num_device_free"], total * 1e3), file=sys.stderr)
    if htrace:
        hts.append(t0 + total)
        print("host trace ms: " + " ".join("%.1f" % ((hts[i + 1] - hts[i]) * 1e3) for i in range(len(hts) - 1)), file=sys.stderr)
    dt = total / steps
    gflop = GFLOP_PER_IMAGE.get(arch, 0.0) * batch
    tf = gflop / dt / 1e3
    traffic, traffic_src = hbm_traffic_per_step(arch, batch)
    alg_bytes = CONV_OUT_MB_PER_IMAGE.get(arch, 0.0) * 1e6 * batch * TOUCHES_PER_STEP
    out = {"metric": "%s train images/sec @222x222 B=%d" % (arch, batch), "value": round(n_ranks_seen * batch / dt, 2), "unit": "images/sec",
           "n_gpus": world, "n_ranks_seen": n_ranks_seen, "ms_per_step": round(dt * 1e3, 3), "steps": steps, "warmup": warmup, "dtype": "f32",
           "hip_graph": bool(graph), "scaling": "weak",
           # HIP events around the same K steps on the compute stream (the host clock above is the contract's; the two agree unless
           # the host, not the GPU, is what the loop waits for)
           "device_ms_per_step": round(dev_ev[0].elapsed_time(dev_ev[1]) / steps, 3) if dev_ev is not None else None,
           "config": {"workload": "fed_run.py train() body, %s classes=%d, SGD lr 0.001, one client per GPU%s"
                      % (arch, classes, ", + 1 FedAvg all-reduce (RCCL) per %d steps" % steps if distributed else "")},
           "tflops_per_gpu": round(tf, 2),
           # The step is HBM-bound: its GEMMs run on the 16-bit MFMA pipe (3 products per fp32 product => 839 TFLOP/s, 1.9 ms
           # for ResNet50 at B=64) while a fully fused schedule moves ~25 GB (3.1 ms at 8 TB/s).  achieved = ALGORITHMIC bytes / step time;
           # traffic = the measured HBM bytes per step (committed --pmc passes of this build); mfma = the same step against the pipe it uses.
           "roofline": {"bound": "hbm", "achieved": round(alg_bytes / dt / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": round(alg_bytes / dt / 1e9 / PEAK_HBM_GBPS, 4), "algorithmic_bytes_per_step": int(alg_bytes),
                        "bound_images_per_s": round(batch / (alg_bytes / (PEAK_HBM_GBPS * 1e9)), 1),
                        "traffic": traffic, "traffic_source": traffic_src,
                        "traffic_over_algorithmic": round(traffic / alg_bytes, 3) if traffic else None,
                        "measured_GBps": round(traffic / dt / 1e9, 1) if traffic else None,
                        "measured_hbm_frac": round(traffic / dt / 1e9 / PEAK_HBM_GBPS, 4) if traffic else None,
                        "mfma": {"achieved": round(tf, 2), "peak": round(PEAK_F16_MFMA_TFLOPS / 3.0, 1), "unit": "TFLOP/s",
                                 "frac": round(tf / (PEAK_F16_MFMA_TFLOPS / 3.0), 4),
                                 "note": "fp32 products as three half-piece products on the 16-bit MFMA: bound = 2516.6 / 3",
                                 "frac_of_f32_mfma_peak": round(tf / PEAK_F32_MFMA_TFLOPS, 4)}},
           "final_loss": round(float(loss.detach()), 5)}
    if graph_choice is not None:
        out["graph_choice"] = graph_choice
    if distributed:
        nbytes = int(fed.FlatParams.of(model).n_total * 4)
        out["fedavg_allreduce_ms"] = round(allreduce_ms, 3)
        out["fedavg_bytes"] = nbytes
        # the collective alone, against the xGMI bounds (point-to-point links: a ring is bound by ONE 153 GB/s link per GPU; with a
        # direct link per peer pair every link carries S/N per phase): bus bandwidth = S x 2 (N-1)/N / time, the figure rccl-tests prints
        from bench import allreduce_bounds
        ar = allreduce_bounds(nbytes, world) or {}
        flat = fed.FlatParams.of(model).flat.clone()           # a scratch copy of the flat state: 5 timed all-reduces after one warm-up
        ts = []
        for i in range(6):
            sync()
            h0 = time.perf_counter()
            dist.all_reduce(flat)
            sync()
            ts.append((time.perf_counter() - h0) * 1e3)
        ts = sorted(ts[1:])
        tt = torch.tensor([ts[len(ts) // 2]], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        only = float(tt.item())
        ar.update({"bytes": nbytes, "round_ms_with_prescale_and_host": round(allreduce_ms, 3),
                   "collective_only_ms": round(only, 4) if only is not None else None,
                   "bus_GBps": round(nbytes * 2.0 * (world - 1) / world / (only * 1e-3) / 1e9, 1) if only else None,
                   "frac_of_one_link": round(nbytes * 2.0 * (world - 1) / world / (only * 1e-3) / 1e9 / 153.0, 3) if only else None})
        out["fedavg_allreduce"] = ar
    if cpu_baseline and rank == 0 and world == 1:
        from oracle import resnet_ref as R
        torch.set_num_threads(host_cores())
        ref = R.resnet50(classes) if arch == "resnet50" else R.resnet18(classes)
        nb, reps = batch, 4            # the metric's own batch; ~10 s of host work on 16 cores
        xc, yc = R.synth_batch(nb, 222, classes, seed=2)
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
    ap.add_argument("--graph", nargs="?", const="1", default="0", help="replay a HIP graph of the step; 'auto': time both loops, run the faster")
    ap.add_argument("--cpu-baseline", action="store_true")
    ap.add_argument("--layers", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(torch.device("cuda:0"), steps=a.steps, warmup=a.warmup, batch=a.batch, arch=a.arch, graph=("auto" if a.graph == "auto" else a.graph == "1"),
                         cpu_baseline=a.cpu_baseline, layers=a.layers)))
