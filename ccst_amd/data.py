"""Minimal torchvision-free data plane so the CLIs run: txt-list parsing
(cjm_util/ImageLoader.py:31-42), PIL decode + Resize((S,S)) + ToTensor (cjm_util/data_helper.py:38-49),
the training transforms of data/data_helper.py:161-181, torchvision.utils.save_image semantics with the
uint8 quantisation done on the GPU, and synthetic stand-ins when the datasets are absent.
CPU image decode/IO is not a kernel target; it exists to make the path usable end to end."""
import os
import random

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import _lib
from ._lib import check, ptr, stream_ptr

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)     # data/data_helper.py:170-171


def _dataset_info(txt_labels):
    """cjm_util/ImageLoader.py:31-42: 'path label' per line."""
    with open(txt_labels, 'r') as f:
        rows = f.readlines()
    names, labels = [], []
    for row in rows:
        row = row.split(' ')
        names.append(row[0])
        labels.append(int(row[1]))
    return names, labels


def _load_rgb(path, size):
    from PIL import Image
    img = Image.open(path).convert('RGB')
    if size:
        img = img.resize((size, size), Image.BILINEAR)        # transforms.Resize((S,S)) on a PIL image
    return img


def _to_tensor(img):
    a = np.asarray(img, dtype=np.uint8)
    return torch.from_numpy(a).permute(2, 0, 1).float().div_(255.0)   # transforms.ToTensor


class ImageTestDataset(Dataset):
    """cjm_util/ImageLoader.py:74-85: returns (tensor, framename)."""

    def __init__(self, names, labels, image_size):
        self.names, self.labels, self.image_size, self.data_path = names, labels, image_size, ""

    def __len__(self):
        return len(self.names)

    def __getitem__(self, index):
        framename = self.data_path + '/' + self.names[index]
        return _to_tensor(_load_rgb(framename, self.image_size)), framename


class SyntheticImages(Dataset):
    """Stand-in when the image files are absent: seeded uniform[0,1) images (ToTensor range) with the
    list's own file names, so output naming and the rest of the CLI behave as with real data."""

    def __init__(self, names, labels, image_size, seed=1, normalized=False, with_path=True):
        self.names, self.labels, self.image_size = names, labels, image_size
        self.seed, self.normalized, self.with_path = seed, normalized, with_path

    def __len__(self):
        return len(self.names)

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        S = self.image_size
        x = torch.randn(3, S, S, generator=g) if self.normalized else torch.rand(3, S, S, generator=g)
        return (x, self.names[index]) if self.with_path else (x, int(self.labels[index]))


def get_train_dataloader(args, txt_root='cjm_util/txt_lists'):
    """cjm_util/data_helper.py:38-44 (AdaIN scripts): shuffle=True, num_workers=0, (tensor, path) items."""
    lst = os.path.join(txt_root, args.dataset.lower(), '%s_train.txt' % args.target)
    synthetic = int(getattr(args, 'synthetic', 0) or 0)
    if synthetic:
        out = getattr(args, 'output', 'output')
        names = [os.path.join(out, 'synthetic/kfold/%s/class0/img_%05d.jpg' % (args.target, i)) for i in range(synthetic)]
        import zlib
        ds = SyntheticImages(names, [0] * synthetic, args.image_size, seed=1 + zlib.crc32(str(args.target).encode()) % 1000)
    else:
        names, labels = _dataset_info(lst)
        ds = ImageTestDataset(names, labels, args.image_size)
    return DataLoader(ds, batch_size=args.batch, shuffle=True)


def quantize_u8(images):
    """[N,C,H,W] float CUDA -> [N,H,W,C] uint8 (x*255+0.5 clamped), one HIP launch."""
    x = images.contiguous()
    N, C, H, W = x.shape
    y = torch.empty((N, H, W, C), device=x.device, dtype=torch.uint8)
    check(_lib.load().ccst_quantize_u8_hwc_f32(ptr(x), ptr(y), N, C, H * W, stream_ptr()), "quantize_u8")
    return y


def save_images(output, paths, output_size=-1):
    """torchvision.utils.save_image per image (CCST_OverallStyleTransfer.py:158-167): quantise on the
    GPU, copy uint8 to the host, encode with PIL.  `paths` are the final file names."""
    from PIL import Image
    u8 = quantize_u8(output).cpu().numpy()
    for arr, name in zip(u8, paths):
        d = os.path.dirname(name)
        if d and not os.path.exists(d):
            os.makedirs(d)
        img = Image.fromarray(arr)
        if output_size and output_size > 0:
            img = img.resize((output_size, output_size), Image.BILINEAR)   # transforms.Resize(output_size) (:154-155)
        img.save(name)


def stylised_name(fpath, target, style, tree):
    """Output path rule of CCST_OverallStyleTransfer.py:160-163 / CCST_SingleStyleTransfer.py:217-219."""
    file_name, ext = os.path.splitext(os.path.basename(fpath))
    out_name = fpath.replace('kfold', tree)
    out_name = out_name.replace('%s' % target, '%s/%s' % (target, style))
    out_name = out_name.replace('%s' % ext, '_%s%s' % (style, ext))
    return out_name


# ---------------------------------------------------------------------------
# federated training loaders (data/data_helper.py:46-181, plain no_DG path)
# ---------------------------------------------------------------------------
class _TrainImages(Dataset):
    def __init__(self, names, labels, args, train):
        self.names, self.labels, self.args, self.train = names, labels, args, train

    def __len__(self):
        return len(self.names)

    def __getitem__(self, index):
        from PIL import Image
        a = self.args
        img = Image.open('/' + self.names[index]).convert('RGB')
        S = a.image_size
        if self.train:      # RandomResizedCrop((S,S), (min_scale, max_scale)) -> ToTensor -> Normalize -> RandomHorizontalFlip
            W, H = img.size
            area = W * H
            for _ in range(10):
                t = area * random.uniform(a.min_scale, a.max_scale)
                logr = random.uniform(np.log(3. / 4.), np.log(4. / 3.))
                r = float(np.exp(logr))
                w, h = int(round(np.sqrt(t * r))), int(round(np.sqrt(t / r)))
                if 0 < w <= W and 0 < h <= H:
                    i, j = random.randint(0, H - h), random.randint(0, W - w)
                    img = img.crop((j, i, j + w, i + h))
                    break
            img = img.resize((S, S), Image.BILINEAR)
        else:               # Resize((S,S)) -> ToTensor -> Normalize
            img = img.resize((S, S), Image.BILINEAR)
        x = _to_tensor(img)
        x = (x - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)
        if self.train and random.random() < a.random_horiz_flip:
            x = x.flip(-1)
        return x, int(self.labels[index])


def get_fed_dataloaders(args, txt_root='data/txt_lists'):
    """(train_loaders, val_loaders, target_test_loader) for the source clients.  With --synthetic N every
    client gets N seeded N(0,1) images (post-Normalize statistics) instead of files."""
    synthetic = int(getattr(args, 'synthetic', 0) or 0)
    train_loaders, val_loaders = [], []
    for di, dname in enumerate(args.source):
        if synthetic:
            names = ['%s_%d' % (dname, i) for i in range(synthetic)]
            labels = [i % args.n_classes for i in range(synthetic)]
            tr = SyntheticImages(names, labels, args.image_size, seed=args.seed + di, normalized=True, with_path=False)
            va = SyntheticImages(names[:max(1, synthetic // 10)], labels, args.image_size, seed=100 + di, normalized=True, with_path=False)
        else:
            p = os.path.join(txt_root, '%s_%s/%s' % (args.dataset.lower(), args.fusion_mode, args.target), '%s_train.txt' % dname)
            names, labels = _dataset_info(p)
            nval = max(1, int(len(names) * args.val_size))
            tr = _TrainImages(names, labels, args, True)
            va = _TrainImages(names[:nval], labels[:nval], args, False)
        train_loaders.append(DataLoader(tr, batch_size=args.batch, shuffle=True))
        val_loaders.append(DataLoader(va, batch_size=args.batch, shuffle=False))
    if synthetic:
        names = ['target_%d' % i for i in range(synthetic)]
        te = SyntheticImages(names, [i % args.n_classes for i in range(synthetic)], args.image_size, seed=999, normalized=True,
                             with_path=False)
    else:
        names, labels = _dataset_info(os.path.join(txt_root, args.dataset, '%s_test.txt' % args.target))
        te = _TrainImages(names, labels, args, False)
    return train_loaders, val_loaders, DataLoader(te, batch_size=args.batch, shuffle=True)
