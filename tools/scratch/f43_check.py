import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from ccst_amd import ops
dev = torch.device("cuda:0")
cases = [(2, 32, 64, 64, 128, False, False), (1, 17, 23, 32, 128, False, False), (2, 24, 40, 128, 256, True, False),
         (1, 22, 38, 64, 256, False, True), (1, 16, 32, 256, 512, False, False), (1, 9, 7, 16, 160, True, False),
         (3, 50, 84, 64, 128, False, False), (1, 64, 64, 512, 256, False, True), (2, 33, 17, 32, 200, True, False),
         (1, 8, 32, 16, 128, False, False), (1, 2, 2, 16, 128, True, False),
         (2, 40, 70, 64, 64, True, False), (1, 33, 47, 128, 64, False, False), (2, 48, 64, 64, 64, False, True), (1, 16, 32, 16, 48, False, False), (1, 5, 9, 32, 64, True, False)]
for reflect in (True, False):
    for case in cases:
        N, H, W, Cin, Cout, pool, ups = case
        g = torch.Generator().manual_seed(31)
        Hs, Ws = (H // 2, W // 2) if ups else (H, W)
        x = torch.randn(N, Hs, Ws, Cin, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
        b = (torch.randn(Cout, generator=g) * 0.1).to(dev)
        pc = ops.pack_conv_weight(w, b, wino=4)
        flags = 1 | (2 if pool else 0) | (4 if ups else 0) | (8 if reflect else 0)
        out = ops.conv3x3_f43(x, pc, flags)
        o2 = ops.conv3x3_f23(x, pc, flags) if Cout >= 128 else ops.conv3x3_halo_split(x, pc, flags)
        xr = x.permute(0, 3, 1, 2).double()
        if ups: xr = F.interpolate(xr, scale_factor=2, mode="nearest")
        xr = F.pad(xr, (1, 1, 1, 1), mode="reflect") if reflect else F.pad(xr, (1, 1, 1, 1))
        ref = F.relu(F.conv2d(xr, w.double(), b.double()))
        if pool: ref = F.max_pool2d(ref, 2, 2, 0, ceil_mode=True)
        ref = ref.permute(0, 2, 3, 1)
        den = max(1.0, float(ref.abs().max()))
        print(reflect, case, "f43 err %.2e  f23 err %.2e  same-twice %s" % (float((out.double() - ref).abs().max()) / den, float((o2.double() - ref).abs().max()) / den,
              torch.equal(out, ops.conv3x3_f43(x, pc, flags))), flush=True)
