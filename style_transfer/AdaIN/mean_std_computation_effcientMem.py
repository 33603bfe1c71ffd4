#!/usr/bin/env python3
"""Drop-in for the reference's style_transfer/AdaIN/mean_std_computation_effcientMem.py (stage 1,
"overall style"): stream one domain through vgg[:31], accumulate per-channel sum / sum of squares,
finalise mu, sigma = sqrt(E[x^2]-mu^2+1e-5) (:117-137), write style_stats/{dataset}/{target}_mean_std.npy
([2,1,512,1,1] float32 -- the np.save the reference has commented out at :146 but stage 2 needs) and the
timing file (:150-155).  Same flags (:29-66).  Under torchrun the domain is sharded over ranks and the
additive (sum, sqsum, count) triple is all-reduced once at the end."""
import os
from datetime import datetime

from _common import base_parser, device_or_die, load_networks, settle_gc

import torch

from ccst_amd import data, style

parser = base_parser(image_size_default=222)      # :48 default 222
parser.set_defaults(output_name=None)              # :49-50: this script declares --output_name without a default
args = parser.parse_args()
device = device_or_die()
os.makedirs(args.output, exist_ok=True)            # :72-73

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
if world > 1:
    import torch.distributed as dist
    dist.init_process_group(backend="nccl", device_id=device)

vgg, decoder = load_networks(args, device)
settle_gc()
data_loader = data.get_train_dataloader(args, args.txt_root, rank, world)      # this rank's shard of the list
start_time = datetime.now()
(feat_mean, feat_std), acc = style.domain_style_stat(vgg, data_loader, device, world, rank,
                                                     progress=lambda it, n: print(f"{it}/{n}"))
torch.cuda.synchronize()
end_time = datetime.now()
print(feat_mean.shape, feat_std.shape)

if rank == 0:
    os.makedirs(f'style_stats/{args.dataset}/', exist_ok=True)
    style.save_style_stat(f'style_stats/{args.dataset}/{args.target}_mean_std.npy', feat_mean, feat_std)
    print(f"Target {args.target}: Finished in {(end_time - start_time).seconds} seconds")
    with open(f"style_stats/{args.dataset}/{args.target}_style_comp_time.txt", 'w') as f:
        f.write(f"Target {args.target}: Finished in {(end_time - start_time).seconds} seconds\n")
        f.write(f"Images number: {acc.images}\n")
        f.write(f"Image resolution: {args.image_size}\n")
        f.write(f"Batch_size: {args.batch}\n")
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
