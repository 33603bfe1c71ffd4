"""bench.py's secondary flow in a loop: AdaIN passes + the CPU oracle leg first (as bench.py does), then bench_resnet.run(graph='auto') with
the eager loop FORCED to follow the capture, N times in one process.  Hunting the one 'backward through the graph a second time' of round 6."""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench_resnet
from ccst_amd import net, style
from oracle import adain_ref as A

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
vgg_w, dec_w = A.he_weights(A.VGG_TABLE, seed=1234), A.he_weights(A.DECODER_TABLE, seed=4321)
net.vgg.load_state_dict(vgg_w)
net.decoder.load_state_dict(dec_w)
vgg31, dec = net.vgg[:31].to(dev).eval(), net.decoder.to(dev).eval()
content = A.synth_content(6, 512, 512, seed=1).to(dev)
stat = [t.to(dev) for t in A.synth_style_stat(512, seed=7)]
with torch.no_grad():
    for _ in range(20):
        out = style.style_transfer(vgg31, dec, content, stat, 1.0)
torch.cuda.synchronize()
torch.set_num_threads(bench_resnet.host_cores())
with torch.no_grad():
    A.style_transfer(vgg_w, dec_w, A.synth_content(2, 256, 256, seed=1), A.synth_style_stat(512, seed=7), 1.0)
bench_resnet.EAGER_MARGIN = 0.0          # the eager loop always "wins": capture, replays, then eager again
fails = 0
for i in range(n):
    try:
        r = bench_resnet.run(dev, steps=10, warmup=5, graph="auto")
        err = r["graph_choice"].get("eager_after_capture_error")
        print(i, r["value"], r["graph_choice"]["hip_graph"], err, flush=True)
        fails += int(err is not None)
    except Exception:
        fails += 1
        traceback.print_exc(limit=4)
print("runs %d, failures %d" % (n, fails))
