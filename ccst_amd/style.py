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
# partly filled rounds and HBM-bound edges (first layer, last decoder layer, AdaIN) overlap the other half's MFMA work: 1382 -> 1416
# images/s at B=6 512x512 (`bench.py --two-stream` reports it as `two_stream_schedule`).  Off by default: with kernels of two streams sharing the
# chip a per-launch duration no longer says anything about the kernel (bench.py's `value` and roofline are measured on the plain
# single-stream schedule).
HALF_BATCH_STREAMS = os.environ.get("CCST_ADAIN_STREAMS", "1") == "2"
_SIDE = {}


# The AdaIN statistics of the content features come out of the encoder's last conv (per-tile channel sums in its epilogue,
# net.Sequential.forward_with_tile_sums) where that conv can produce them: the AdaIN step is then ONE streaming launch over the
# features (ops.adain_from_tile_sums) instead of load-everything / two-pass / store.  TILE_SUM_ADAIN = False: always ops.adain (the stand-alone two-pass kernel).
TILE_SUM_ADAIN = True


# The AdaIN step WITHOUT its pass over the features (round 6): where conv4_1 left centred records with the channel maxima (the F(4,3)
# kernel) and the decoder's first conv runs on that kernel too, the step is one small launch that turns the statistics into the affine map
# y = a x + b per (image, channel) (ops.adain_fold_affine), applied by the decoder's first conv on its loads (net.Sequential.forward_affine):
# the normalised tensor is never written.  FUSE_ADAIN = False: the streaming form (one read + one write of the features).
FUSE_ADAIN = os.environ.get("CCST_ADAIN_FUSE", "1") != "0"


def _encode_adain_decode(vgg, decoder, content, style_stat, alpha):
    """decoder(alpha-blend(adaIN_StyleStat_ContentFeat(vgg(content), style_stat))) (CCST_OverallStyleTransfer.py:35,43,45-46)."""
    style_mean, style_std = style_stat
    if TILE_SUM_ADAIN and hasattr(vgg, "forward_with_tile_sums"):
        content_f, part = vgg.forward_with_tile_sums(content)
        if ops.adain_tile_sums_ok(content_f, part):
            if FUSE_ADAIN and part.shape[2] == 4 and ops.f43_records(content_f, part) and hasattr(decoder, "affine_ok") and decoder.affine_ok(content_f):
                affine, words, _stats = ops.adain_fold_affine(content_f, part, style_mean, style_std, alpha=alpha)
                return decoder.forward_affine(content_f, affine, words)
            return decoder(ops.adain_from_tile_sums(content_f, part, style_mean, style_std, alpha=alpha))
    else:
        content_f = vgg(content)
    return decoder(ops.adain(content_f, style_mean, style_std, alpha=alpha))   # AdaIN + alpha blend in one pass


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
            outs[i] = _encode_adain_decode(vgg, decoder, parts[i], style_stat, alpha)
    main.wait_stream(side)
    outs[1].record_stream(main)
    return torch.cat(outs, 0)



def style_transfer(vgg, decoder, content, style_stat, alpha=1.0, interpolation_weights=None):
    assert (0.0 <= alpha <= 1.0)
    if interpolation_weights:
        # CCST_OverallStyleTransfer.py:36-42: the batch holds one content image per style; their stylised features are mixed by the
        # weights, then blended with the first image's features (unreachable from the reference CLIs: do_interpolation is never set, :109)
        content_f = vgg(content)
        style_mean, style_std = style_stat
        base_feat = ops.adain(content_f, style_mean, style_std, alpha=1.0)
        return decoder(ops.interp_blend(base_feat, content_f, interpolation_weights, alpha))
    if HALF_BATCH_STREAMS and content.shape[0] >= 2 and content.is_cuda:
        return _style_transfer_two_streams(vgg, decoder, content, style_stat, alpha)
    return _encode_adain_decode(vgg, decoder, content, style_stat, alpha)


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
        self._add(calc_sum(feat), feat.shape[0])

    def update_from_images(self, vgg, images):
        """Stage 1's loop body on a batch of images: encode and accumulate, the sums coming out of the encoder's last conv
        (net.Sequential.forward_with_chan_sums) when the encoder is the fused plan; returns the features."""
        if hasattr(vgg, "forward_with_chan_sums"):
            feat, triple = vgg.forward_with_chan_sums(images)
        else:
            feat = vgg(images)
            triple = calc_sum(feat)
        self._add(triple, feat.shape[0])
        return feat

    def _add(self, triple, images):
        s, q, n = triple
        self.sum = self.sum + s
        self.sqsum = self.sqsum + q
        self.count += n
        self.images += images

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
            acc.update_from_images(vgg, batch.to(device))
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


class StylePipeline(object):
    """The stage-2 batch loop with its edges overlapped (CCST_OverallStyleTransfer.py:149-167 runs load -> transfer -> .cpu() ->
    save strictly in turn): the H2D copy of batch k+1 and the quantise + D2H of batch k-1 run on their own HIP streams under the
    compute of batch k, from / into pinned host buffers; finished batches are handed to `sink` (e.g. the image writers of
    data.ImageWriterPool) in order.  Results are the bytes of the serial path (same kernels, same order per batch).

        pipe = StylePipeline(vgg, decoder, device, output_size=-1)
        for u8, meta in pipe.run(((batch, fpaths) for ...), style_stat, alpha): ...      # u8: [N,H,W,3] uint8 numpy, valid until the next item
    """

    def __init__(self, vgg, decoder, device, output_size=-1, depth=3, u8=True, no_h2d=False, no_d2h=False, one_stream=False):
        # depth 3: before batch k+1 can be staged its slot's previous result (batch k-2) is handed out -- finished long ago -- so the
        # host queues batch k's kernels while the GPU still runs batch k-1 (with 2 slots it would first wait for batch k-1 itself)
        self.vgg, self.decoder, self.device = vgg, decoder, torch.device(device)
        self.output_size, self.depth, self.u8 = output_size, max(2, int(depth)), u8
        # (diagnosis switches, tools/pipeline_diag.py: no_h2d / no_d2h skip that edge's copy, one_stream puts both on the compute stream)
        self.no_h2d, self.no_d2h = bool(no_h2d), bool(no_d2h)
        self.h2d = torch.cuda.current_stream(self.device) if one_stream else torch.cuda.Stream(device=self.device)
        self.d2h = torch.cuda.current_stream(self.device) if one_stream else torch.cuda.Stream(device=self.device)
        self.trace = None               # a list: per batch (host seconds, timing events) for tools/pipeline_diag.py
        self.slots = [dict(pin_in=None, dev_in=None, pin_out=None, ev_in=torch.cuda.Event(), ev_c=torch.cuda.Event(),
                           ev_out=torch.cuda.Event(), meta=None, busy=False) for _ in range(self.depth)]

    def _stage_in(self, slot, batch):
        if slot["pin_in"] is None or slot["pin_in"].shape != batch.shape:
            slot["pin_in"] = torch.empty(batch.shape, dtype=batch.dtype).pin_memory()
            slot["dev_in"] = torch.empty(batch.shape, dtype=batch.dtype, device=self.device)
            # the caching allocator may hand out a block that kernels already queued on the compute stream still read (it is free in
            # that stream's order only): the copy stream must not write it before they are done
            self.h2d.wait_stream(torch.cuda.current_stream(self.device))
        src = batch
        if not batch.is_pinned():
            slot["pin_in"].copy_(batch)
            src = slot["pin_in"]
        with torch.cuda.stream(self.h2d):
            self.h2d.wait_event(slot["ev_c"])      # the slot's previous batch has been consumed by its kernels
            if self.trace is not None:
                slot["t_h2d0"] = torch.cuda.Event(enable_timing=True)
                slot["t_h2d0"].record(self.h2d)
            if not (self.no_h2d and slot.get("filled")):
                slot["dev_in"].copy_(src, non_blocking=True)
                slot["filled"] = True
            if self.trace is not None:
                slot["t_h2d1"] = torch.cuda.Event(enable_timing=True)
                slot["t_h2d1"].record(self.h2d)
            slot["ev_in"].record(self.h2d)
        # dev_in was allocated in the compute stream's order but is written on the copy stream: tell the caching allocator, so that a
        # buffer dropped on a shape change (the short last batch of a list) is not handed out again before this copy has run
        slot["dev_in"].record_stream(self.h2d)
        slot["keep"] = src                     # the pinned source stays alive until the copy has run

    def _compute(self, slot, style_stat, alpha):
        from . import data as cdata
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(slot["ev_in"])
        if self.trace is not None:
            import time
            h0 = time.perf_counter()
            tc0, tc1, td0, td1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            tc0.record(cur)
        with torch.no_grad():
            out = style_transfer(self.vgg, self.decoder, slot["dev_in"], style_stat, alpha)
            if self.output_size and self.output_size > 0:
                out = cdata.resize_tensor(out, self.output_size)
            res = cdata.quantize_u8(out) if self.u8 else out
        slot["ev_c"].record(cur)
        if self.trace is not None:
            tc1.record(cur)
        if slot["pin_out"] is None or slot["pin_out"].shape != res.shape or slot["pin_out"].dtype != res.dtype:
            slot["pin_out"] = torch.empty(res.shape, dtype=res.dtype).pin_memory()
        with torch.cuda.stream(self.d2h):
            self.d2h.wait_event(slot["ev_c"])
            if self.trace is not None:
                td0.record(self.d2h)
            if not self.no_d2h:
                slot["pin_out"].copy_(res, non_blocking=True)
            if self.trace is not None:
                td1.record(self.d2h)
            slot["ev_out"].record(self.d2h)
        res.record_stream(self.d2h)            # the allocator must not hand `res` out again before the copy has read it
        if self.trace is not None:
            self.trace.append({"issue_ms": (time.perf_counter() - h0) * 1e3, "h0": h0, "h2d": (slot.get("t_h2d0"), slot.get("t_h2d1")),
                               "c": (tc0, tc1), "d2h": (td0, td1)})

    def trace_rows(self):
        """Text rows of the recorded trace: per batch, host issue ms and, relative to the first batch's compute start, when its H2D,
        compute and D2H started / ended on the device (ms)."""
        t = self.trace or []
        if not t:
            return []
        z = t[0]["c"][0]
        rows = ["batch  host_issue  host_gap |  h2d start-end  |  compute start-end (dur)  |  d2h start-end"]
        for i, r in enumerate(t):
            def rel(e):
                return z.elapsed_time(e) if e is not None else float("nan")
            gap = (r["h0"] - t[i - 1]["h0"]) * 1e3 if i else 0.0
            rows.append("%5d  %9.2f  %8.2f | %7.2f-%7.2f | %7.2f-%7.2f (%5.2f) | %7.2f-%7.2f" % (
                i, r["issue_ms"], gap, rel(r["h2d"][0]), rel(r["h2d"][1]), rel(r["c"][0]), rel(r["c"][1]), r["c"][0].elapsed_time(r["c"][1]),
                rel(r["d2h"][0]), rel(r["d2h"][1])))
        return rows

    def run(self, batches, style_stat, alpha=1.0):
        """batches: iterable of (CPU tensor [N,3,H,W], meta).  Yields (numpy view of the pinned result, meta) in order; a view is
        valid until the generator is advanced again (its pinned buffer is reused)."""
        it = iter(batches)
        pending = []                               # slots with a batch in flight, oldest first
        i = 0
        nxt = next(it, None)
        if nxt is not None:
            self._stage_in(self.slots[0], nxt[0])
            self.slots[0]["meta"] = nxt[1]
        while nxt is not None:
            slot = self.slots[i % self.depth]
            nxt = next(it, None)
            if nxt is not None:                    # batch k+1 goes up while batch k computes
                s2 = self.slots[(i + 1) % self.depth]
                if s2["busy"]:                     # its previous result has not been handed out yet
                    done = pending.pop(0)
                    done["ev_out"].synchronize()
                    done["busy"] = False
                    yield done["pin_out"].numpy(), done["meta_out"]
                self._stage_in(s2, nxt[0])
                s2["meta"] = nxt[1]
            self._compute(slot, style_stat, alpha)
            slot["busy"], slot["meta_out"] = True, slot["meta"]
            pending.append(slot)
            i += 1
        for done in pending:
            done["ev_out"].synchronize()
            done["busy"] = False
            yield done["pin_out"].numpy(), done["meta_out"]
