#!/usr/bin/env python3
"""Per-tensor fp32 rounding noise of the full-size ResNet50 step (B=64, 222x222, the weights / batch of
tests/test_resnet_gpu.py::test_resnet50_full_size_step_vs_oracle): the CPU oracle's train step in fp32 and in fp64, and for each
of the 161 gradient tensors max|g32 - g64|, max|g64|, ||g32 - g64||_2, ||g64||_2  ->  tests/golden/resnet50_fullsize_grad_noise.npz
(a few KB).  The fp64 step takes ~10 minutes on 8 cores, which is why it is a fixture and not part of the GPU test.
    python tools/make_resnet_grad_noise.py"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import resnet_ref as R  # noqa: E402

torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
classes, nb, lr = 7, 64, 0.001
oracle = R.resnet50(classes)
sd = R.seeded_state_dict(oracle, 77, residual_gamma=0.25, fc_gain=8.0)
oracle.load_state_dict(sd)
x, y = R.synth_batch(nb, 222, classes, seed=78)
o64 = copy.deepcopy(oracle).double()
R.train_step(oracle, x, y, lr)
R.train_step(o64, x.double(), y, lr)
out = {}
g64 = dict((k, p.grad) for k, p in o64.named_parameters())
for k, p in oracle.named_parameters():
    d = p.grad.double() - g64[k]
    out["noise_max/" + k] = np.float64(d.abs().max())
    out["gmax/" + k] = np.float64(g64[k].abs().max())
    out["noise_l2/" + k] = np.float64(d.norm())
    out["g_l2/" + k] = np.float64(g64[k].norm())
np.savez(os.path.join(ROOT, "tests", "golden", "resnet50_fullsize_grad_noise.npz"), **out)
print("wrote %d tensors; worst noise_max/gmax = %.3g" % (len(out) // 4, max(float(out["noise_max/" + k[5:]]) / float(out[k]) for k in out if k.startswith("gmax/"))))
