"""Random shapes through the F(4,3) kernel (conv3x3_f43.hip; both tiles, every flag combination): each launch three times (the results
must be the same bits: the kernel's hand-counted waits and two-barrier chunks leave no room for a benign race) and against the direct
half-piece kernel (1e-5 of max |y|).  python tools/f43_stress.py [cases] [seed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccst_amd import ops  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    worst = 0.0
    for k in range(cases):
        N, Cin, Cout = ri(1, 4), 16 * ri(1, 24), [48, 64, 64, 128, 160, 256, 384, 512][ri(0, 7)]
        pool, ups, reflect = bool(ri(0, 1)), bool(ri(0, 1)), bool(ri(0, 1))
        H, W = ri(2, 90), ri(2, 150)
        if ups:
            H, W = 2 * max(1, H // 2), 2 * max(1, W // 2)
        Hs, Ws = (H // 2, W // 2) if ups else (H, W)
        x = (torch.randn(N, Hs, Ws, Cin, generator=g) * 10.0 ** ri(-3, 3)).to(dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
        b = (torch.randn(Cout, generator=g) * 0.1).to(dev)
        pc = ops.pack_conv_weight(w, b, wino=4)
        flags = 1 | (2 if pool else 0) | (4 if ups else 0) | (8 if reflect else 0)
        y0 = ops.conv3x3_f43(x, pc, flags)
        for _ in range(2):
            assert torch.equal(y0, ops.conv3x3_f43(x, pc, flags)), ("not reproducible", N, H, W, Cin, Cout, flags)
        ref = ops.conv3x3_halo_split(x, pc, flags)
        err = float((y0 - ref).abs().max()) / max(1e-30, float(ref.abs().max()))
        worst = max(worst, err)
        assert err < 1e-5, (err, N, H, W, Cin, Cout, flags)
        if k % 10 == 9:
            print("%d cases, worst difference from the direct kernel %.2e of max |y|" % (k + 1, worst), flush=True)
    print("ok: %d cases, worst %.2e" % (cases, worst))


if __name__ == "__main__":
    main()
