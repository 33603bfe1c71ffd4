"""Oracle (test infrastructure only): the reference's image input/output edges on CPU.

The reference builds its loaders from torchvision transforms over PIL images
(data/data_helper.py:161-181, style_transfer/AdaIN/cjm_util/data_helper.py:46-49) and writes results with
torchvision.utils.save_image (CCST_OverallStyleTransfer.py:154-167).  torchvision is a third-party dependency
that is NOT under /root/reference (requirements.txt:7 ``torchvision>=0.8.1``) and is not installed in this image;
Pillow (its backend for these transforms) IS installed, here and on the GPU box.  So:

  * ``resized_crop`` / ``to_tensor`` / ``normalize`` / ``hflip`` restate torchvision.transforms.functional of the
    0.8-0.15 generation on top of **PIL itself** -- PIL does the resize, which is the only non-trivial arithmetic;
  * ``random_resized_crop_params`` / ``random_flip`` restate RandomResizedCrop.get_params / RandomHorizontalFlip
    (torch global RNG, same draw order).  No reference test pins that RNG stream: **parity unpinned** for the
    stream itself, pinned for everything downstream of the drawn rectangle;
  * ``pil_resize_restated`` is a numpy restatement of Pillow's libImaging/Resample.c (8 bpc, BILINEAR): the published
    algorithm the HIP kernel implements.  It is pinned against PIL in tests/test_data_cpu.py (byte-equal);
  * ``save_image_bytes`` restates save_image's quantisation ``mul(255).add_(0.5).clamp_(0,255).permute(1,2,0).to(uint8)``.
"""
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2


# ---------------------------------------------------------------- PIL-backed reference chain (the checker)
def resized_crop(img, i, j, h, w, size):
    """torchvision.transforms.functional.resized_crop on a PIL image: crop((j, i, j+w, i+h)).resize(size[::-1], BILINEAR)."""
    from PIL import Image
    return img.crop((j, i, j + w, i + h)).resize((size[1], size[0]), Image.BILINEAR)


def to_tensor(img):
    """transforms.ToTensor on an RGB PIL image: uint8 HWC -> float32 CHW, .div(255)."""
    a = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())
    return a.permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def normalize(t, mean, std):
    """transforms.Normalize: tensor.sub_(mean[:, None, None]).div_(std[:, None, None]) with float32 mean / std."""
    m = torch.as_tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.as_tensor(std, dtype=torch.float32).view(-1, 1, 1)
    return t.clone().sub_(m).div_(s)


def hflip(t):
    return t.flip(-1)


def random_resized_crop_params(height, width, scale, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """RandomResizedCrop.get_params (torchvision 0.8+): up to 10 draws from the torch global RNG, then the centre-crop fallback."""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,)).item()
            j = torch.randint(0, width - w + 1, size=(1,)).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def random_flip(p):
    """RandomHorizontalFlip(p).forward: flips iff torch.rand(1) < p."""
    return bool(torch.rand(1) < p)


def train_transform(img, size, scale, mean, std, flip_p, params=None, flip=None):
    """data/data_helper.py:173-178: RandomResizedCrop((S,S), scale) -> ToTensor -> Normalize [-> RandomHorizontalFlip(p) if p > 0]."""
    W, H = img.size
    i, j, h, w = params if params is not None else random_resized_crop_params(H, W, scale)
    t = normalize(to_tensor(resized_crop(img, i, j, h, w, (size, size))), mean, std)
    if flip_p > 0.0:
        if random_flip(flip_p) if flip is None else flip:
            t = hflip(t)
    return t


def val_transform(img, size, mean=None, std=None):
    """data/data_helper.py:183-186 (Resize -> ToTensor -> Normalize); cjm_util/data_helper.py:46-49 without the Normalize."""
    from PIL import Image
    t = to_tensor(img.resize((size, size), Image.BILINEAR))
    return normalize(t, mean, std) if mean is not None else t


# ---------------------------------------------------------------- Pillow's resample, restated (what the kernel implements)
def _axis_tables(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc of libImaging/Resample.c for the bilinear (triangle, support 1) filter
    over the whole axis (box = [0, in_size))."""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coefs = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w, ww = [], 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            v = 1.0 - a if a < 1.0 else 0.0
            w.append(v)
            ww += v
        for x, v in enumerate(w):
            if ww != 0.0:
                v = v / ww
            coefs[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, coefs


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_resize_restated(img, out_h, out_w):
    """uint8 HWC array -> uint8 [out_h, out_w, C]: horizontal pass (skipped when the width is unchanged), uint8 intermediate,
    vertical pass (skipped when the height is unchanged) -- ImagingResample's two-pass schedule."""
    h, w, _ = img.shape
    out = img
    if out_w != w:
        _, b, kk = _axis_tables(w, out_w)
        t = np.zeros((h, out_w, img.shape[2]), np.uint8)
        for xx in range(out_w):
            xmin, xmax = b[xx]
            acc = np.full((h, img.shape[2]), 1 << (PRECISION_BITS - 1), np.int64)
            for x in range(xmax):
                acc += out[:, xmin + x, :].astype(np.int64) * int(kk[xx, x])
            t[:, xx, :] = _clip8(acc)
        out = t
    if out_h != h:
        _, b, kk = _axis_tables(h, out_h)
        t = np.zeros((out_h, out.shape[1], img.shape[2]), np.uint8)
        for yy in range(out_h):
            ymin, ymax = b[yy]
            acc = np.full((out.shape[1], img.shape[2]), 1 << (PRECISION_BITS - 1), np.int64)
            for y in range(ymax):
                acc += out[ymin + y].astype(np.int64) * int(kk[yy, y])
            t[yy] = _clip8(acc)
        out = t
    return out


# ---------------------------------------------------------------- output edge
def save_image_bytes(t):
    """torchvision.utils.save_image's array for ONE image tensor [C,H,W] (make_grid of a single image is the image):
    ``grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to('cpu', torch.uint8).numpy()``
    (CCST_OverallStyleTransfer.py:167 calls it per image)."""
    return t.detach().cpu().to(torch.float32).mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
