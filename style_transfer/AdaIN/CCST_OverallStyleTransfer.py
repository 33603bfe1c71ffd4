#!/usr/bin/env python3
"""Drop-in for the reference's style_transfer/AdaIN/CCST_OverallStyleTransfer.py (stage 2, overall
mode): for every other domain's {style}_mean_std.npy, encoder -> AdaIN(stat) -> alpha blend -> decoder
on every content batch, save the images under all_style_transferred_Overall (:138-175).  Same flags
(:49-93).  Under torchrun the content LIST is sharded over ranks by entry (disjoint and complete whatever each
rank's shuffle would have drawn; images are independent: no collective).

--fuse_stats (addition): stage 1 and stage 2 in one process -- a style domain whose
{style}_mean_std.npy is absent (or every domain with --refresh_stats) has its statistics computed here by
streaming that domain through the encoder (style.domain_style_stat, the stage-1 code), used straight from
device memory, and written back as the [2,1,512,1,1] float32 .npy cache the other tools read."""
import copy
import os
from datetime import datetime

from _common import ALL_CLIENTS, base_parser, device_or_die, load_networks, settle_gc

import torch

from ccst_amd import data, style

parser = base_parser(image_size_default=512)
parser.add_argument('--output_size', type=int, default=-1, help='transform images into final size')
parser.add_argument('--no_save', action='store_true', help='skip PIL encoding (throughput runs)')
parser.add_argument('--serial', action='store_true', help="the reference's serial batch loop instead of the overlapped pipeline (same images)")
parser.add_argument('--fuse_stats', action='store_true', help='compute missing style statistics in-process (stage 1 + 2 fused)')
parser.add_argument('--refresh_stats', action='store_true', help='with --fuse_stats: recompute even if the .npy cache exists')
args = parser.parse_args()

# the image writers first: worker processes are forked before this process creates its HIP context (data.ImageWriterPool)
writers = None if (args.serial or args.no_save) else data.ImageWriterPool()
all_clients = ALL_CLIENTS[args.dataset.lower()]
style_domains = sorted(set(all_clients) - set([args.target]))    # :107 (sorted: the reference's set order is hash-dependent)
device = device_or_die()
os.makedirs(args.output, exist_ok=True)
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))

if world > 1 and args.fuse_stats:
    import torch.distributed as dist
    dist.init_process_group(backend="nccl", device_id=device)       # only the fused statistics need a collective

vgg, decoder = load_networks(args, device)
settle_gc()
pipeline = None if args.serial else style.StylePipeline(vgg, decoder, device, output_size=args.output_size)
data_loader = data.get_train_dataloader(args, args.txt_root, rank, world)      # this rank's shard of the content list


def style_stat_of(style_name):
    path = f"style_stats/{args.dataset}/{style_name}_mean_std.npy"
    if not args.fuse_stats or (os.path.exists(path) and not args.refresh_stats):
        return style.load_style_stat(path, device)                    # :140-144
    sargs = copy.copy(args)
    sargs.target = style_name
    stat, acc = style.domain_style_stat(vgg, data.get_train_dataloader(sargs, args.txt_root, rank, world), device, world, rank)
    print(f"    computed style statistics of {style_name} from {acc.images} images")
    if rank == 0:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        style.save_style_stat(path, stat[0], stat[1])
    return stat


for style_name in style_domains:
    print(f"Content: {args.target} | Style: {style_name}")
    style_stat = style_stat_of(style_name)
    start_time = datetime.now()
    img_count = 0
    if args.serial:          # the reference's loop shape: load -> transfer -> .cpu() -> save, strictly in turn
        for it, (batch, fpaths) in enumerate(data_loader):
            img_count += len(batch)
            with torch.no_grad():
                output = style.style_transfer(vgg, decoder, batch.to(device), style_stat, args.alpha)
            print(f"    Target: {args.target}, Style: {style_name}, Iteration: {it}/{len(data_loader)}")
            if not args.no_save:
                names = [data.stylised_name(f, args.target, style_name, 'all_style_transferred_Overall') for f in fpaths]
                data.save_images(output, names, args.output_size)
    else:                    # same images, edges overlapped: H2D / quantise + D2H on their own streams, encoding in worker processes
        for it, (u8, fpaths) in enumerate(pipeline.run(data_loader, style_stat, args.alpha)):
            img_count += len(u8)
            print(f"    Target: {args.target}, Style: {style_name}, Iteration: {it}/{len(data_loader)}")
            if writers is not None:
                writers.submit(u8, [data.stylised_name(f, args.target, style_name, 'all_style_transferred_Overall') for f in fpaths])
        if writers is not None:
            writers.drain()
    torch.cuda.synchronize()
    end_time = datetime.now()
    if rank == 0:
        with open(f"{args.dataset}_{args.target}_overall_stylize_time.txt", 'w') as f:      # :171-175
            f.write(f"Target {args.target} with style {style_name}: Finished in {(end_time - start_time).seconds} seconds\n")
            f.write(f"Images number: {img_count}\n")
            f.write(f"Image resolution: {args.image_size}\n")
            f.write(f"Batch_size: {args.batch}\n")
if writers is not None:
    writers.close()
print(f"Target {args.target}: Finished in {(end_time - start_time).seconds} seconds")
if world > 1 and args.fuse_stats:
    dist.barrier()
    dist.destroy_process_group()
