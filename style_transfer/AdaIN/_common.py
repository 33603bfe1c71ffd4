"""Shared by the three AdaIN CLI drop-ins: repo path, flags, model loading."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

ALL_CLIENTS = {      # CCST_OverallStyleTransfer.py:97-105
    'pacs': ["art_painting", "cartoon", "photo", "sketch"],
    'officehome': ['art', 'clipart', 'product', 'real_world'],
    'digitsfive': ['MNIST', 'MNIST_M', 'SVHN', 'SynthDigits', 'USPS'],
    'camelyon17': ['hospital1', 'hospital2', 'hospital3', 'hospital4', 'hospital5'],
}


def base_parser(image_size_default):
    """Flag names/defaults of CCST_OverallStyleTransfer.py:49-93 (the other two scripts share them)."""
    p = argparse.ArgumentParser()
    p.add_argument('--dataset', type=str, default='pacs', help='dataset')
    p.add_argument('--target', type=str, help='content domain')
    p.add_argument('--style_stat', type=str, help='File path to the style statistics')
    p.add_argument('--vgg', type=str, default='models/vgg_normalised.pth')
    p.add_argument('--decoder', type=str, default='models/decoder.pth')
    p.add_argument('--crop', action='store_true', help='do center crop to create squared image')
    p.add_argument('--save_ext', default='.jpg', help='The extension name of the output image')
    p.add_argument('--output', type=str, default='output', help='Directory to save the output image(s)')
    p.add_argument('--output_name', type=str, default='out_image.png', help='output image path')
    p.add_argument('--preserve_color', action='store_true', help='If specified, preserve color of the content image')
    p.add_argument('--alpha', type=float, default=1.0,
                   help='The weight that controls the degree of stylization. Should be between 0 and 1')
    p.add_argument('--style_interpolation_weights', type=str, default='',
                   help='The weight for blending the style of multiple style images')
    p.add_argument('--batch', type=int, default=32, help='batch size')
    p.add_argument('--image_size', type=int, default=image_size_default, help='image size')
    # additions (not in the reference): run without datasets / checkpoints, shard over ranks
    p.add_argument('--synthetic', type=int, default=0, help='use N seeded synthetic content images instead of the list files')
    p.add_argument('--random_weights', action='store_true', help='seeded He-normal weights when the .pth files are absent')
    p.add_argument('--txt_root', type=str, default='cjm_util/txt_lists')
    return p


def load_networks(args, device):
    """CCST_OverallStyleTransfer.py:116-127: eval, load_state_dict, truncate vgg at relu4_1, to(device)."""
    from ccst_amd import net
    decoder, vgg = net.decoder, net.vgg
    decoder.eval()
    vgg.eval()
    if args.random_weights or not (os.path.exists(args.vgg) and os.path.exists(args.decoder)):
        if not args.random_weights:
            raise FileNotFoundError("%s / %s not found (README.md:21); pass --random_weights for seeded weights" % (args.vgg, args.decoder))
        g = torch.Generator().manual_seed(1234)
        for seq in (vgg, decoder):
            for m in seq.modules():
                if isinstance(m, torch.nn.Conv2d):
                    fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                    with torch.no_grad():
                        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                        m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
    else:
        decoder.load_state_dict(torch.load(args.decoder))
        vgg.load_state_dict(torch.load(args.vgg))
    vgg = net.Sequential(*list(vgg.children())[:31])
    vgg.to(device)
    decoder.to(device)
    return vgg, decoder


def settle_gc():
    """Call once after set-up, before a batch loop: collect, then move everything alive into the permanent generation.  The loops issue
    a few dozen launches per 3 ms batch from ONE host thread; a generation-2 pass of the cyclic GC over the ~10^5 objects torch and the
    plan hold takes 60-80 ms and stalls that thread (measured: -25 % on the overlapped stage-2 loop when one lands in a 0.25 s run)."""
    import gc
    gc.collect()
    gc.freeze()


def device_or_die():
    if not torch.cuda.is_available():
        raise SystemExit("ccst_amd: an MI355X (ROCm) device is required; this implementation has no CPU path")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    return torch.device("cuda", local)
