#!/usr/bin/env python3
"""Drop-in for the reference's style_transfer/AdaIN/CCST_OverallStyleTransfer.py (stage 2, overall
mode): for every other domain's {style}_mean_std.npy, encoder -> AdaIN(stat) -> alpha blend -> decoder
on every content batch, save the images under all_style_transferred_Overall (:138-175).  Same flags
(:49-93).  Under torchrun, content batches are sharded over ranks (images are independent: no
collective)."""
import os
from datetime import datetime

from _common import ALL_CLIENTS, base_parser, device_or_die, load_networks

import torch

from ccst_amd import data, style

parser = base_parser(image_size_default=512)
parser.add_argument('--output_size', type=int, default=-1, help='transform images into final size')
parser.add_argument('--no_save', action='store_true', help='skip PIL encoding (throughput runs)')
args = parser.parse_args()

all_clients = ALL_CLIENTS[args.dataset.lower()]
style_domains = sorted(set(all_clients) - set([args.target]))    # :107 (sorted: the reference's set order is hash-dependent)
device = device_or_die()
os.makedirs(args.output, exist_ok=True)
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))

vgg, decoder = load_networks(args, device)
data_loader = data.get_train_dataloader(args, args.txt_root)

for style_name in style_domains:
    print(f"Content: {args.target} | Style: {style_name}")
    style_stat = style.load_style_stat(f"style_stats/{args.dataset}/{style_name}_mean_std.npy", device)   # :140-144
    start_time = datetime.now()
    img_count = 0
    for it, (batch, fpaths) in enumerate(data_loader):
        if it % world != rank:
            continue
        img_count += len(batch)
        with torch.no_grad():
            output = style.style_transfer(vgg, decoder, batch.to(device), style_stat, args.alpha)
        print(f"    Target: {args.target}, Style: {style_name}, Iteration: {it}/{len(data_loader)}")
        if not args.no_save:
            names = [data.stylised_name(f, args.target, style_name, 'all_style_transferred_Overall') for f in fpaths]
            data.save_images(output, names, args.output_size)
    torch.cuda.synchronize()
    end_time = datetime.now()
    if rank == 0:
        with open(f"{args.dataset}_{args.target}_overall_stylize_time.txt", 'w') as f:      # :171-175
            f.write(f"Target {args.target} with style {style_name}: Finished in {(end_time - start_time).seconds} seconds\n")
            f.write(f"Images number: {img_count}\n")
            f.write(f"Image resolution: {args.image_size}\n")
            f.write(f"Batch_size: {args.batch}\n")
print(f"Target {args.target}: Finished in {(end_time - start_time).seconds} seconds")
