"""Throughput lines for the BASELINE configs that are not the headline one (bench.py's ``secondary`` list):

  config 0/1  mean_std_computation_effcientMem.py --image_size 512 --batch 32: stage-1 style statistics, images/sec through
              vgg[:31] + calc_sum (mean_std_computation_effcientMem.py:117-132), with the HBM roofline of the statistics kernel
              (268.4 MB read per batch)
  config 3    CCST_SingleStyleTransfer.py 512x512 batch=32: one style image per batch, its mu / sigma from calc_sum
              (CCST_SingleStyleTransfer.py:178-212), then style_transfer over the content batch
  config 5    fed_run.py Camelyon17 ResNet18 B=32 classes=2 @222: train-step images/sec (bench_resnet.run, HIP graph chosen by
              measurement)

Lives next to bench.py, outside the package: the ``cpu_baseline`` legs import the oracle.
"""
import gc
import time

import torch

PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md, HBM3E


def _timed(fn, steps, warmup, dist_barrier):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    dist_barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dist_barrier()
    return time.perf_counter() - t0


def _max_over_ranks(elapsed, dev, world):
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())
    return elapsed


def stage1(dev, vgg31, A, world, rank, steps=6, warmup=2, batch=32, size=512, cpu=False, vgg_w=None):
    """One step = one batch of a domain through vgg[:31] and calc_sum, sums accumulated on the device (the stage-1 loop body)."""
    from ccst_amd import ops, style
    import torch.distributed as dist
    barrier = (lambda: (dist.barrier(), torch.cuda.synchronize())) if world > 1 else (lambda: None)
    x = A.synth_content(batch, size, size, seed=31 + rank).to(dev)
    acc = style.StyleStatAccumulator()

    def step():
        with torch.no_grad():
            acc.update_from_images(vgg31, x)      # the sums come out of conv4_1's epilogue (no pass over relu4_1)
    gc.collect()
    ops.TIMING = None
    elapsed = _timed(step, steps, warmup, barrier)
    ops.TIMING = ev = []
    step()
    with torch.no_grad():
        style.calc_sum(vgg31(x))                  # the stand-alone statistics kernel on the same tensor, for the HBM roofline beside it
    torch.cuda.synchronize()
    ops.TIMING = None
    us = [a.elapsed_time(b) * 1e3 for name, _f, a, b, _i in ev if name == "chan_sums"]
    fin = [a.elapsed_time(b) * 1e3 for name, _f, a, b, _i in ev if name == "chan_sums_finalize"]
    elapsed = _max_over_ranks(elapsed, dev, world)
    nbytes = 4 * batch * 512 * (size // 8) * (size // 8)          # the relu4_1 tensor read once (268.4 MB at B=32, 512^2)
    out = {"metric": "stage-1 style statistics images/sec @%dx%d B=%d (vgg[:31] + calc_sum)" % (size, size, batch),
           "value": round(world * batch * steps / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(elapsed / steps * 1e3, 3), "dtype": "f32", "scaling": "weak",
           "config": {"workload": "mean_std_computation_effcientMem PACS %dx%d batch=%d, one domain's batches per rank + one all-reduce of "
                                  "(sum, sqsum, n) at the end" % (size, size, batch)},
           "statistics": {"path": "fused: per-tile sums from conv4_1's epilogue (%s) + ccst_chan_sums_finalize_f32"
                                  % ("ccst_conv3x3_f43_f32, the default plan's kernel" if ops.HALO_SPLIT != "0" else "none: CCST_HALO_SPLIT=0 takes the streaming pass"),
                          "finalize_us": round(fin[0], 2) if fin else None, "tensor_bytes_not_read": nbytes},
           "roofline": {"bound": "hbm", "kernel": "ccst_chan_sums_f32 (the stand-alone pass over relu4_1, what the fused path removes)", "bytes": nbytes,
                        "avg_launch_us": round(us[0], 2) if us else None, "achieved": round(nbytes / us[0] / 1e3, 1) if us else None,
                        "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(nbytes / us[0] / 1e3 / PEAK_HBM_GBPS, 4) if us else None,
                        "traffic": None}}
    if cpu and rank == 0 and world == 1:
        n = 4
        xc = A.synth_content(n, size, size, seed=31)
        with torch.no_grad():
            A.calc_sum(A.encoder(xc[:1], vgg_w))
            c0 = time.perf_counter()
            A.calc_sum(A.encoder(xc, vgg_w))
            c1 = time.perf_counter()
        out["cpu_baseline"] = {"value": round(n / (c1 - c0), 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "%d images %dx%d through the oracle's encoder + calc_sum (oracle/adain_ref.py, torch CPU fp32)" % (n, size, size)}
    return out


def single_mode(dev, vgg31, dec, A, world, rank, steps=4, warmup=1, batch=32, size=512):
    """One step = one style image encoded -> calc_sum -> mu / sigma, then style_transfer over a content batch of 32
    (CCST_SingleStyleTransfer.py:178-212)."""
    from ccst_amd import style
    import torch.distributed as dist
    barrier = (lambda: (dist.barrier(), torch.cuda.synchronize())) if world > 1 else (lambda: None)
    content = A.synth_content(batch, size, size, seed=41 + rank).to(dev)
    style_img = A.synth_content(1, size, size, seed=43 + rank).to(dev)

    def step():
        with torch.no_grad():
            s, q, n = style.calc_sum(vgg31(style_img))
            stat = list(style.finalise_style_stats(s, q, n))
            return style.style_transfer(vgg31, dec, content, stat, 1.0)
    gc.collect()
    elapsed = _max_over_ranks(_timed(step, steps, warmup, barrier), dev, world)
    return {"metric": "AdaIN single-style stylised images/sec @%dx%d B=%d" % (size, size, batch),
            "value": round(world * batch * steps / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "dtype": "f32", "scaling": "weak",
            "config": {"workload": "CCST_SingleStyleTransfer PACS %dx%d batch=%d: one style image per batch (encode + calc_sum), then "
                                   "encoder -> AdaIN -> decoder over the content batch; source domains shard one per rank" % (size, size, batch)},
            "note": "same kernels as the headline line (its roofline applies); 33 encoder passes + 32 decoder passes per step"}


def eval_forward(dev, world, rank, steps=12, warmup=3, batch=64, size=222, arch="resnet50", classes=7):
    """a12: test() (fed_run.py:214-259) -- the eval-mode forward (BatchNorm on running statistics), cross-entropy and accuracy of one
    batch, data resident.  MFMA-bound like the train step's forward: 8.17 GFLOP per image."""
    import types
    import torch.distributed as dist
    from ccst_amd import fed
    from ccst_amd.nets import models
    barrier = (lambda: (dist.barrier(), torch.cuda.synchronize())) if world > 1 else (lambda: None)
    torch.manual_seed(1 + rank)
    model = models.get_network(arch)(types.SimpleNamespace(dg_method=""), pretrained=False, classes=classes).to(dev)
    model.eval()
    loss_fun = fed.CrossEntropyLoss()
    g = torch.Generator(device="cpu").manual_seed(7 + rank)
    x = torch.randn(batch, 3, size, size, generator=g).to(dev)
    y = torch.randint(0, classes, (batch,), generator=g).to(dev)

    def step():
        with torch.no_grad():
            return loss_fun(model(x), y)
    gc.collect()
    elapsed = _max_over_ranks(_timed(step, steps, warmup, barrier), dev, world)
    gflop = {"resnet50": 8.170, "resnet18": 3.623}[arch] * batch
    tf = gflop * steps / elapsed / 1e3
    return {"metric": "%s eval-forward images/sec @%dx%d B=%d (test() body)" % (arch, size, size, batch),
            "value": round(world * batch * steps / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "dtype": "f32", "scaling": "weak",
            "config": {"workload": "fed_run.py test() body: eval-mode forward + CrossEntropy + accuracy, %s classes=%d" % (arch, classes)},
            "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4),
                         "gflop_per_step": round(gflop, 1), "traffic": None,
                         "note": "8.17 GFLOP per image at 222x222 against the fp32-MFMA peak (SURVEY 8d's bound); the pointwise convs run on half "
                                 "pieces on the 16-bit MFMA (the eval BatchNorm applies leave the |max| words too), the 3x3 convs on fp32 F(2x2) "
                                 "Winograd; BatchNorm (eval) applies are HBM passes on top"}}


def communication_inprocess(dev, K=3, reps=10, arch="resnet50", classes=7):
    """a13: communication() (fed_run.py:385-455, fedavg branch :400-414) with the server and K clients on ONE GPU -- the in-process form
    of the reference (its Python loop over 320 keys x K clients on the CPU).  HBM-bound: algorithmic bytes = K arenas read + (K + 1)
    arenas written (every client is overwritten with the average)."""
    import types
    from ccst_amd import fed
    from ccst_amd.nets import models
    args = types.SimpleNamespace(mode="fedavg", dg_method="")
    torch.manual_seed(5)
    server = models.get_network(arch)(args, pretrained=False, classes=classes).to(dev)
    clients = []
    for k in range(K):
        m = models.get_network(arch)(args, pretrained=False, classes=classes).to(dev)
        clients.append(m)
    weights = [1.0 / K] * K
    fed.communication(args, server, clients, weights)          # builds the flat arenas
    torch.cuda.synchronize()
    n = int(fed.FlatParams.of(server).n_total)
    walls = []
    for _ in range(reps):
        h0 = time.perf_counter()
        fed.communication(args, server, clients, weights)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - h0) * 1e6)
    walls.sort()
    # the pass over the arenas alone (what communication() launches), HIP events around the launch
    import ctypes
    from ccst_amd import _lib
    from ccst_amd._lib import check, ptr, stream_ptr
    srv = fed.FlatParams.of(server)
    arenas = [fed.FlatParams.of(m) for m in clients]
    cl = (ctypes.c_void_p * K)(*[a.flat.data_ptr() for a in arenas])
    cw = (ctypes.c_float * K)(*weights)
    ts = []
    for _ in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().ccst_fedavg_f32(ptr(srv.flat), cl, cw, K, n, stream_ptr()), "fedavg")
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    dev_us, wall_us = ts[len(ts) // 2], walls[len(walls) // 2]
    nbytes = (2 * K + 1) * n * 4
    return {"metric": "communication() fedavg, K=%d %s clients on one GPU" % (K, arch), "value": round(wall_us, 1), "unit": "us per call (wall)",
            "higher_is_better": False, "n_gpus": 1, "kernel_us": round(dev_us, 1), "state_floats": n,
            "config": {"workload": "fed_run.py communication(), --mode fedavg, %d clients + server in one process" % K},
            "note": "the wall figure is host work: four arena validity checks (every tensor's address), the side-stream join, one int64 "
                    "counter-arena copy -- once per federated round, next to an epoch of training",
            "roofline": {"bound": "hbm", "bytes": nbytes, "achieved": round(nbytes / dev_us / 1e3, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": round(nbytes / dev_us / 1e3 / PEAK_HBM_GBPS, 4), "traffic": None,
                         "kernel": "fedavg_kernel (ccst_fedavg_f32): K reads + K + 1 writes per element in one pass (the reference: a Python "
                                   "loop over 320 keys x K clients on the CPU)"}}
