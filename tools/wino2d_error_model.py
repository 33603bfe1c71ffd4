#!/usr/bin/env python3
"""Prices the NEXT reduced-MFMA form before it is built: Winograd F(2,3) along y x F(4,3) along x on half pieces
(1.0 executed MFMA FLOP per algorithmic FLOP) against today's F(4,3) along x only (1.5), in emulation on the CPU.

Rounding model of the kernels (conv3x3_f43.hip): fp32 transform chain on the scaled input (the power-of-two scale rides in the
coefficients), split into IEEE-half (hi, lo) pieces, three products a_lo b_hi + a_hi b_lo + a_hi b_hi per 16-channel chunk, each MFMA's
16-term dot product added to an fp32 accumulator (the dot itself in fp64 here: an under-estimate of the MFMA's internal rounding),
weights transformed in fp64, rounded to fp32, scaled and split once; output transform in fp32.

  * per layer: every 3x3 layer of the AdaIN path on its OWN oracle input (fp32 oracle activations), error against an fp64 convolution,
    relative to max |y| -- the quantity the GPU tests gate at 1e-5;
  * whole path: encoder -> AdaIN -> decoder with every 3x3 layer between the image edges emulated, against the fp64 path and the fp32
    oracle (oracle/adain_ref.py), relative and absolute -- the contract is 1e-3 absolute on the stylised image.

    python tools/wino2d_error_model.py [size=96] [batch=1]        (CPU, a few minutes)
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import adain_ref as A          # noqa: E402  (this tool IS test infrastructure: it prices a design against the oracle)

torch.set_num_threads(8)
X_TARGET, W_TARGET = 13, 9               # common.h: CCST_SPLIT_X_TARGET / CCST_SPLIT_W_TARGET
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
                   dtype=torch.float32)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                  dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float32)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def split(v):
    hi = v.to(torch.float16).to(torch.float32)
    lo = (v - hi).to(torch.float16).to(torch.float32)
    return hi.double(), lo.double()


def scale_exp(m, target):
    m = float(m)
    if m == 0.0:
        return 0
    import math
    return target - int(math.floor(math.log2(m)))


def chain(BT, d, dim):
    """fp32 fused-multiply-add chain of the rows of BT over the slices of d along `dim` (every step rounded to fp32)."""
    outs = []
    for q in range(BT.shape[0]):
        acc = None
        for j in range(BT.shape[1]):
            c = float(BT[q, j])
            if c != 0.0:
                t = d.select(dim, j) * c
                acc = t if acc is None else (acc + t)
        outs.append(acc)
    return torch.stack(outs, dim)


def conv_emulated(xp, w, b, form):
    """xp: [Cin, H + 2, W + 2] fp32 (already padded), w [Cout, Cin, 3, 3], b [Cout] -> [Cout, H, W] fp32.  form: 'x' (F(4,3) along x, three
    ky taps) or 'xy' (F(2,3) along y x F(4,3) along x)."""
    Cin, Hp, Wp = xp.shape
    H, W = Hp - 2, Wp - 2
    Cout = w.shape[0]
    nq = (W + 3) // 4
    nr = (H + 1) // 2
    x = torch.zeros(Cin, 2 * nr + 2, 4 * nq + 2)
    x[:, :Hp, :Wp] = xp
    head = 4 if form == "x" else 5          # a position is up to 10 x (x only) / 20 x (2-D) the largest pixel
    kx = scale_exp(x.abs().max(), X_TARGET - head)
    kw = scale_exp(w.abs().max(), W_TARGET - 1)
    xs = x * (2.0 ** kx)
    # quads: d[c, y, p, j] = xs[c, y, 4 p + j], j = 0..5
    d = torch.stack([xs[:, :, j:j + 4 * nq:4] for j in range(6)], dim=3)                     # [Cin, Hh, nq, 6]
    if form == "x":
        V = chain(BT4, d, 3)                                                                   # [Cin, Hh, nq, 6]
        U = torch.einsum('qk,oiyk->yqoi', G4, w.double()).float() * (2.0 ** kw)               # [3, 6, Cout, Cin]
        Vh, Vl = split(V)
        Uh, Ul = split(U)
        Hh = 2 * nr
        M = torch.zeros(6, Hh, nq, Cout)
        for c0 in range(0, Cin, 16):
            for ky in range(3):
                Ah = Vh[c0:c0 + 16, ky:ky + Hh].permute(3, 1, 2, 0).reshape(6, Hh * nq, 16)
                Al = Vl[c0:c0 + 16, ky:ky + Hh].permute(3, 1, 2, 0).reshape(6, Hh * nq, 16)
                Bh = Uh[ky, :, :, c0:c0 + 16].transpose(1, 2)                                  # [6, 16, Cout]
                Bl = Ul[ky, :, :, c0:c0 + 16].transpose(1, 2)
                for P in (torch.bmm(Al, Bh), torch.bmm(Ah, Bl), torch.bmm(Ah, Bh)):
                    M = M + P.float().reshape(6, Hh, nq, Cout)
        M = M * (2.0 ** -(kx + kw))
        Y = chain(AT4, M, 0)                                                                   # [4, Hh, nq, Cout]
        y = Y.permute(3, 1, 2, 0).reshape(Cout, Hh, 4 * nq)
    else:
        # y transform first (rows 2 i .. 2 i + 3), then x
        dy = torch.stack([d[:, r:r + 2 * nr:2] for r in range(4)], dim=1)                     # [Cin, 4, nr, nq, 6]
        Vy = chain(BT2, dy, 1)                                                                 # [Cin, 4, nr, nq, 6]
        V = chain(BT4, Vy, 4)                                                                  # [Cin, 4, nr, nq, 6]
        U = torch.einsum('rk,qm,oikm->rqoi', G2, G4, w.double()).float() * (2.0 ** kw)        # [4, 6, Cout, Cin]
        Vh, Vl = split(V)
        Uh, Ul = split(U)
        M = torch.zeros(24, nr * nq, Cout)
        for c0 in range(0, Cin, 16):
            Ah = Vh[c0:c0 + 16].permute(1, 4, 2, 3, 0).reshape(24, nr * nq, 16)
            Al = Vl[c0:c0 + 16].permute(1, 4, 2, 3, 0).reshape(24, nr * nq, 16)
            Bh = Uh[:, :, :, c0:c0 + 16].reshape(24, Cout, 16).transpose(1, 2)
            Bl = Ul[:, :, :, c0:c0 + 16].reshape(24, Cout, 16).transpose(1, 2)
            for P in (torch.bmm(Al, Bh), torch.bmm(Ah, Bl), torch.bmm(Ah, Bh)):
                M = M + P.float()
        M = (M * (2.0 ** -(kx + kw))).reshape(4, 6, nr, nq, Cout)
        Yy = chain(AT2, M, 0)                                                                  # [2, 6, nr, nq, Cout]
        Y = chain(AT4, Yy, 1)                                                                  # [2, 4, nr, nq, Cout]
        y = Y.permute(4, 2, 0, 3, 1).reshape(Cout, 2 * nr, 4 * nq)
    return y[:, :H, :W] + b.view(-1, 1, 1)


def run_path(table, x, weights, conv3, upto=None, record=None, dtype=torch.float32):
    for i, e in enumerate(table):
        if upto is not None and i >= upto:
            break
        if e[0] == "conv":
            w, b = weights["%d.weight" % e[1]].to(dtype), weights["%d.bias" % e[1]].to(dtype)
            if conv3 is not None and e[4] == 3 and e[2] % 16 == 0 and e[3] >= 64:
                if record is not None:
                    record.append((e, x, weights))
                x = torch.stack([conv3(x[n], w, b) for n in range(x.shape[0])])
            else:
                x = F.conv2d(x, w, b)
        elif e[0] == "pad":
            x = F.pad(x, (1, 1, 1, 1), mode="reflect")
        elif e[0] == "relu":
            x = F.relu(x)
        elif e[0] == "pool":
            x = F.max_pool2d(x, (2, 2), (2, 2), (0, 0), ceil_mode=True)
        elif e[0] == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
    return x


def whole(content, stat, vgg_w, dec_w, conv3, dtype=torch.float32, record=None):
    f = run_path(A.VGG_TABLE, content.to(dtype), vgg_w, conv3, upto=31, record=record, dtype=dtype)
    N, C = f.shape[:2]
    var = f.reshape(N, C, -1).var(dim=2) + 1e-5
    mean = f.reshape(N, C, -1).mean(dim=2).view(N, C, 1, 1)
    t = (f - mean) / var.sqrt().view(N, C, 1, 1) * stat[1].to(dtype) + stat[0].to(dtype)
    return run_path(A.DECODER_TABLE, t, dec_w, conv3, record=record, dtype=dtype)


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    vgg_w = A.he_weights(A.VGG_TABLE, seed=1234)
    dec_w = A.he_weights(A.DECODER_TABLE, seed=4321)
    content = A.synth_content(batch, size, size, seed=1)
    stat = A.synth_style_stat(512, seed=7)
    with torch.no_grad():
        # per layer, on the oracle's own fp32 activations
        rec = []
        whole(content, stat, vgg_w, dec_w, lambda x, w, b: F.conv2d(x[None], w, b)[0], record=rec)
        print("per layer (input = the fp32 oracle's activation; error / max |y| against an fp64 convolution)")
        print("%-22s %12s %12s %8s" % ("layer", "F(4,3) x", "F(2,3)yxF(4,3)x", "ratio"))
        worst = [0.0, 0.0]
        for e, x, ws in rec:
            w, b = ws["%d.weight" % e[1]], ws["%d.bias" % e[1]]
            ref = F.conv2d(x[0:1].double(), w.double(), b.double())[0]
            errs = []
            for form in ("x", "xy"):
                y = conv_emulated(x[0], w, b, form)
                errs.append(float((y.double() - ref).abs().max() / ref.abs().max()))
            worst = [max(worst[0], errs[0]), max(worst[1], errs[1])]
            print("%4d -> %4d @ %4d^2     %12.2e %12.2e %8.2f" % (e[2], e[3], x.shape[2] - 2, errs[0], errs[1], errs[1] / errs[0]))
        print("worst layer: F(4,3) x %.2e, 2-D %.2e   (GPU gate today: 1e-5; VERDICT's bound for the 2-D form: 2e-5)" % tuple(worst))
        # whole path
        ref64 = whole(content, stat, vgg_w, dec_w, None, dtype=torch.float64)
        o32 = A.style_transfer(vgg_w, dec_w, content, stat, 1.0)
        print("\nwhole path at %dx%d, B=%d (max |image| %.2f): max |difference|" % (size, size, batch, float(ref64.abs().max())))
        print("  fp32 oracle vs fp64 path                      %.2e" % float((o32.double() - ref64).abs().max()))
        for form, name in (("x", "F(4,3) along x (today)"), ("xy", "F(2,3) y x F(4,3) x")):
            out = whole(content, stat, vgg_w, dec_w, lambda x, w, b, f=form: conv_emulated(x, w, b, f))
            print("  %-26s vs fp64 path        %.2e    vs fp32 oracle %.2e" % (name, float((out.double() - ref64).abs().max()),
                                                                             float((out - o32).abs().max())))


if __name__ == "__main__":
    main()
