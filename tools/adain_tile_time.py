"""Times the AdaIN step of the path (adain_tile_sums_nhwc_kernel: statistics folded from the conv epilogue's per-tile records, normalise +
blend streamed once) on the bench shape: python tools/adain_tile_time.py [reps]   env: TILE_FLOATS (4: centred records, 2: raw sums),
NO_WORDS=1 (no |max| words of the output)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccst_amd import ops, _lib
from ccst_amd.ops import ptr, stream_ptr, check

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
pf = int(os.environ.get("TILE_FLOATS", 4))
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
N, H, W, C, tpi = 6, 64, 64, 512, 32
x = torch.rand(N, H, W, C, generator=g).to(dev)
out = torch.empty_like(x)
cnt = float(H * W // tpi)
if pf == 4:
    part = torch.stack([torch.rand(N * tpi, C, generator=g) * cnt, torch.rand(N * tpi, C, generator=g) * cnt * 0.1,
                        torch.full((N * tpi, C), cnt), torch.zeros(N * tpi, C)], dim=2).contiguous().to(dev)
else:
    part = torch.stack([torch.rand(N * tpi, C, generator=g) * cnt, torch.rand(N * tpi, C, generator=g) * cnt], dim=2).contiguous().to(dev)
sm, ss = torch.rand(C).to(dev), (torch.rand(C) + 0.5).to(dev)
words = None if os.environ.get("NO_WORDS") == "1" else torch.zeros((N, 64), dtype=torch.int32, device=dev)      # per image (ABI version 2)
lib = _lib.load()
stat = torch.empty(2, N * C, device=dev)

def run():
    check(lib.ccst_adain_tile_sums_f32(ptr(x), ptr(part), pf, tpi, ptr(sm), ptr(ss), 0, 1.0, ptr(out), N, C, H * W, 1e-5, ptr(stat[0]), ptr(stat[1]),
                                       ptr(words), stream_ptr()), "adain_tile_sums")


for _ in range(5):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print("adain_tile_sums floats=%d words=%s  median %.1f us  best %.1f us  %.2f TB/s" % (pf, words is not None, ts[len(ts) // 2], ts[0], 2 * x.numel() * 4 / ts[len(ts) // 2] / 1e6))
