#!/usr/bin/env python3
"""Where does style.StylePipeline's time go?  (overlapped u8 1494 images/s against 1794 serial in the driver's run.)

Per batch: host time to stage / issue, and -- from timing events on the three streams -- when its H2D, compute and D2H ran on the
device.  Variants isolate the edges:  --no-h2d (compute reads a resident batch), --no-d2h (result stays on the device), --serial
(the bench's serial loop), --one-stream (everything on the compute stream).

    python tools/pipeline_diag.py [--batches 60] [--variant pipe|no_h2d|no_d2h|one_stream|serial|compute_only]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=60)
    ap.add_argument("--variant", default="all")
    a = ap.parse_args()
    from ccst_amd import data as cdata, net, style
    from oracle import adain_ref as A          # input synthesis only
    dev = torch.device("cuda:0")
    net.vgg.load_state_dict(A.he_weights(A.VGG_TABLE, seed=1234))
    net.decoder.load_state_dict(A.he_weights(A.DECODER_TABLE, seed=4321))
    vgg31, dec = net.vgg[:31].to(dev).eval(), net.decoder.to(dev).eval()
    B, S = 6, 512
    host_in = A.synth_content(B, S, S, seed=1).pin_memory()
    stat = [t.to(dev) for t in A.synth_style_stat(512, seed=7)]
    dev_in = host_in.to(dev)
    host_u8 = torch.empty((B, S, S, 3), dtype=torch.uint8).pin_memory()
    nb = a.batches

    def compute_only():
        with torch.no_grad():
            for _ in range(nb):
                cdata.quantize_u8(style.style_transfer(vgg31, dec, dev_in, stat, 1.0))
        torch.cuda.synchronize()

    def serial():
        for _ in range(nb):
            with torch.no_grad():
                o = style.style_transfer(vgg31, dec, host_in.to(dev, non_blocking=True), stat, 1.0)
            host_u8.copy_(cdata.quantize_u8(o), non_blocking=True)
            torch.cuda.synchronize()

    def pipe(**kw):
        p = style.StylePipeline(vgg31, dec, dev, **kw)
        for _ in p.run(((host_in, None) for _ in range(3)), stat, 1.0):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in p.run(((host_in, None) for _ in range(nb)), stat, 1.0):
            pass
        return time.perf_counter() - t0

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r if isinstance(r, float) else time.perf_counter() - t0
    variants = {"compute_only": lambda: timed(compute_only), "serial": lambda: timed(serial), "pipe": lambda: pipe()}
    for k in ("no_h2d", "no_d2h", "one_stream", "graph"):
        variants[k] = (lambda k_: (lambda: pipe(**{k_: True})))(k)
    names = list(variants) if a.variant == "all" else a.variant.split(",")
    for name in names:
        try:
            dt = variants[name]()
        except TypeError as e:          # a StylePipeline without that switch
            print("%-13s unsupported (%s)" % (name, e))
            continue
        print("%-13s %7.3f ms/batch  %8.1f images/s" % (name, dt / nb * 1e3, nb * B / dt), flush=True)
    # host / device timeline of the plain pipeline, batch by batch
    if "pipe" in names and hasattr(style.StylePipeline, "trace"):
        p = style.StylePipeline(vgg31, dec, dev)
        p.trace = []
        for _ in p.run(((host_in, None) for _ in range(12)), stat, 1.0):
            pass
        torch.cuda.synchronize()
        for row in p.trace_rows():
            print(row)


if __name__ == "__main__":
    main()
