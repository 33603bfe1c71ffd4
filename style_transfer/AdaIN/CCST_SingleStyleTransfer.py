#!/usr/bin/env python3
"""Drop-in for the reference's style_transfer/AdaIN/CCST_SingleStyleTransfer.py (stage 2, single
mode): per content batch pick ONE random style image of the style domain (random.choice, seed 1,
:21-26,195), encode it, mu/sigma with the sum / sum-of-squares formula (:199-203), then the same
transfer (:207-212); outputs under all_style_transferred_Single.  Same flags (:70-117).  (The
reference stops in pdb after the first style domain, :232; this runs all of them.)"""
import os
import random
from datetime import datetime

from _common import ALL_CLIENTS, base_parser, device_or_die, load_networks, settle_gc

import numpy as np
import torch

from ccst_amd import data, style

seed = 1
random.seed(a=seed)
np.random.seed(seed)
torch.manual_seed(seed)
torch.cuda.manual_seed_all(seed)

parser = base_parser(image_size_default=512)
parser.add_argument('--style_size', type=int, default=512, help='New (minimum) size for the style image')
parser.add_argument('--output_size', type=int, default=-1, help='transform images into final size')
parser.add_argument('--no_save', action='store_true')
args = parser.parse_args()

all_clients = ALL_CLIENTS[args.dataset.lower()]
style_domains = sorted(set(all_clients) - set([args.target]))
device = device_or_die()
os.makedirs(args.output, exist_ok=True)
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))

vgg, decoder = load_networks(args, device)
settle_gc()
data_loader = data.get_train_dataloader(args, args.txt_root, rank, world)      # this rank's shard of the content list


def load_style_image(path):
    if args.synthetic:
        g = torch.Generator().manual_seed(abs(hash(path)) % (2 ** 31))
        return torch.rand(3, args.style_size, args.style_size, generator=g)
    u8 = data.decode_rgb_u8(str(path))
    h, w = int(u8.shape[0]), int(u8.shape[1])
    s = args.style_size
    oh, ow = h, w
    if s:                                   # transforms.Resize(size): shorter side -> size (:29-30)
        oh, ow = (max(1, int(s * h / w)), s) if w <= h else (s, max(1, int(s * w / h)))
    img = data.gpu_transform([u8], [(0, 0, h, w, 0)], (oh, ow), device)[0]      # PIL-exact resize + ToTensor on the GPU
    if args.crop:                           # transforms.CenterCrop(size) (:31-32)
        t, l = int(round((oh - s) / 2.0)), int(round((ow - s) / 2.0))
        img = img[:, t:t + s, l:l + s]
    return img


for style_name in style_domains:
    print(f"Content: {args.target} | Style: {style_name}")
    if args.synthetic:
        style_img_list = ['synthetic/%s/%d.jpg' % (style_name, i) for i in range(64)]
    else:
        sub = 'camelyon17_discardBlackWhite' if args.dataset == 'camelyon17' else args.dataset        # :165-168
        with open(os.path.join(args.txt_root, sub, f"{style_name}_train.txt"), 'r') as f:
            style_img_list = [mm.split(' ')[0] for mm in f.readlines()]
    start_time = datetime.now()
    img_count = 0
    for it, (batch, fpaths) in enumerate(data_loader):
        style_img_path = random.choice(style_img_list)
        img_count += len(batch)
        print(f"    Style: {style_name}, Iteration: {it}/{len(data_loader)}")
        with torch.no_grad():
            # (features and their per-channel sums in one go: the sums come out of the encoder's last conv)
            style_feat, (feat_sum, feat_square_sum, count) = vgg.forward_with_chan_sums(load_style_image(style_img_path).to(device).unsqueeze(0))
            style_stat = list(style.finalise_style_stats(feat_sum, feat_square_sum, count))
            output = style.style_transfer(vgg, decoder, batch.to(device), style_stat, args.alpha)
        if not args.no_save:
            names = [data.stylised_name(f, args.target, style_name, 'all_style_transferred_Single') for f in fpaths]
            data.save_images(output, names, args.output_size)
    torch.cuda.synchronize()
    end_time = datetime.now()
    if rank == 0:
        with open(f"{args.dataset}_{args.target}_single_stylize_time.txt", 'w') as f:
            f.write(f"Target {args.target} with style {style_name}: Finished in {(end_time - start_time).seconds} seconds\n")
            f.write(f"Images number: {img_count}\n")
            f.write(f"Image resolution: {args.image_size}\n")
            f.write(f"Batch_size: {args.batch}\n")
