"""Script-level functions of the three AdaIN CLIs, on HIP kernels.

style_transfer   : CCST_OverallStyleTransfer.py:32-46 (same copy at CCST_SingleStyleTransfer.py:39-53)
calc_sum         : mean_std_computation_effcientMem.py:103-115 / CCST_SingleStyleTransfer.py:55-67
StyleStatAccumulator : the stage-1 loop and finalisation, mean_std_computation_effcientMem.py:117-137
"""
import os

import numpy as np
import torch

from . import ops

# CCST_ADAIN_STREAMS=2: run the two halves of a content batch on two HIP streams, so one half's launch ramps, tails and
# HBM-bound edges (stem, last decoder layer, AdaIN) overlap the other half's MFMA work: 548 -> 561 images/s at B=6 512x512.
# Off by default: with kernels of two streams sharing the chip a per-launch duration no longer says anything about the
# kernel (bench.py's roofline line is measured on the plain single-stream schedule).
HALF_BATCH_STREAMS = os.environ.get("CCST_ADAIN_STREAMS", "1") == "2"
_SIDE = {}


def _style_transfer_two_streams(vgg, decoder, content, style_stat, alpha):
    dev = content.device
    main = torch.cuda.current_stream(dev)
    side = _SIDE.get(dev.index)
    if side is None:
        side = _SIDE[dev.index] = torch.cuda.Stream(device=dev)
    h = content.shape[0] // 2
    parts, outs = (content[:h], content[h:]), [None, None]
    side.wait_stream(main)
    for i, st in enumerate((main, side)):
        with torch.cuda.stream(st):
            f = ops.adain(vgg(parts[i]), style_stat[0], style_stat[1], alpha=alpha)
            outs[i] = decoder(f)
    main.wait_stream(side)
    outs[1].record_stream(main)
    return torch.cat(outs, 0)



def style_transfer(vgg, decoder, content, style_stat, alpha=1.0, interpolation_weights=None):
    assert (0.0 <= alpha <= 1.0)
    if interpolation_weights:
        # unreachable from the reference CLIs (do_interpolation is never set, CCST_OverallStyleTransfer.py:109)
        raise NotImplementedError("ccst_amd: style interpolation is outside the hot path")
    if HALF_BATCH_STREAMS and content.shape[0] >= 2 and content.is_cuda:
        return _style_transfer_two_streams(vgg, decoder, content, style_stat, alpha)
    content_f = vgg(content)
    style_mean, style_std = style_stat
    feat = ops.adain(content_f, style_mean, style_std, alpha=alpha)   # AdaIN + alpha blend in one pass
    return decoder(feat)


def calc_sum(feat):
    feat = feat.detach()
    size = feat.shape
    assert (len(size) == 4)
    return ops.chan_sums(feat)


def finalise_style_stats(feat_sum, feat_square_sum, count):
    """mean_std_computation_effcientMem.py:135-137 (CCST_SingleStyleTransfer.py:201-203): BIASED
    variance E[x^2]-mu^2 in fp32, sigma = sqrt(var + 1e-5).  [1,C,1,1] tensors; 2*C flops of glue."""
    feat_mean = feat_sum / float(count)
    feat_var = feat_square_sum / float(count) - feat_mean ** 2
    feat_std = torch.sqrt(feat_var + 1e-5)
    return feat_mean, feat_std


class StyleStatAccumulator(object):
    """Running per-channel sums over batches (additive, so also shardable across ranks)."""

    def __init__(self):
        self.sum, self.sqsum, self.count, self.images = 0, 0, 0, 0

    def update(self, feat):
        s, q, n = calc_sum(feat)
        self.sum = self.sum + s
        self.sqsum = self.sqsum + q
        self.count += n
        self.images += feat.shape[0]

    def all_reduce(self):
        """Intra-domain sharding: one all_reduce(SUM) of 2*C floats + the count."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            buf = torch.cat([self.sum.reshape(-1), self.sqsum.reshape(-1),
                             torch.tensor([float(self.count), float(self.images)], device=self.sum.device)])
            buf = buf.double()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            C = self.sum.numel()
            self.sum = buf[:C].float().reshape(self.sum.shape)
            self.sqsum = buf[C:2 * C].float().reshape(self.sqsum.shape)
            self.count, self.images = int(round(buf[2 * C].item())), int(round(buf[2 * C + 1].item()))

    def finalise(self):
        return finalise_style_stats(self.sum, self.sqsum, self.count)


def domain_style_stat(vgg, loader, device, world=1, rank=0, progress=None):
    """Stage 1 as a function (mean_std_computation_effcientMem.py:117-137): stream one domain's loader through
    vgg[:31], accumulate the per-channel sums, all-reduce the additive triple, finalise.  Under torchrun `loader` is
    already this rank's shard of the domain's list (data.get_train_dataloader(..., rank, world)): the shards partition
    the list whatever each rank's RNG does, so every image is counted exactly once.  Returns ([mean, std] as stage 2 consumes them, accumulator).  Used by the stage-1 CLI and by stage
    2's --fuse_stats, which skips the .npy round trip and keeps the file only as a cache."""
    acc = StyleStatAccumulator()
    with torch.no_grad():
        for it, (batch, _) in enumerate(loader):
            acc.update(vgg(batch.to(device)))
            if progress is not None:
                progress(it, len(loader))
    if acc.images == 0:         # a rank without batches still takes part in the all-reduce
        acc.sum = torch.zeros((1, 512, 1, 1), device=device)
        acc.sqsum = torch.zeros((1, 512, 1, 1), device=device)
    acc.all_reduce()
    mean, std = acc.finalise()
    return [mean, std], acc


def save_style_stat(path, mean, std):
    """The stage-1 -> stage-2 hand-off: np.save of [mean, std] => float32 [2,1,C,1,1]
    (mean_std_computation_effcientMem.py:146, read at CCST_OverallStyleTransfer.py:140-144)."""
    np.save(path, np.stack([mean.detach().cpu().numpy(), std.detach().cpu().numpy()]).astype(np.float32))


def load_style_stat(path, device):
    stat = np.load(path)
    return [torch.from_numpy(np.ascontiguousarray(s)).float().to(device) for s in stat]
