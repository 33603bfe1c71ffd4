"""torchvision-free data plane of the two CLIs, with the per-image transform chain on the GPU.

Host side (what the reference's data/ and cjm_util/ packages do around the hot path):
  * txt-list parsing ``_dataset_info`` (data/ImageLoader.py:31-42), the disjoint random train/val split
    ``get_random_subset`` (data/ImageLoader.py:13-28; Python ``random.sample``, seeded by fed_run.py:510),
    ``creat_train_loader_list`` -- the CCST list expansion / renaming rules (data/data_helper.py:125-145) -- the
    ``--mode deepall`` concatenation (data/data_helper.py:66-121) and the ``--limit_source`` / ``--limit_target`` subsets
    (data/data_helper.py:33-43,96-98,156-158);
  * file decode (PIL, CPU -- JPEG/PNG entropy decoding is not a GPU job) and the draw of the crop rectangle / flip
    (torchvision's RandomResizedCrop.get_params / RandomHorizontalFlip: torch global RNG, same draw order);
  * output naming (CCST_OverallStyleTransfer.py:160-163) and image writing.
Device side (HIP, ``csrc/image_ops.hip``): crop -> PIL-exact bilinear resize -> ToTensor -> Normalize -> flip in one
launch per batch from the decoded uint8 pixels (``gpu_transform`` / ``DeviceImageLoader``), byte-identical to the PIL
chain, and the save_image quantisation (``quantize_u8``).  There is no CPU fallback for the device side."""
import ctypes
import math
import os
import random

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import _lib
from ._lib import CcstImageXform, check, ptr, stream_ptr

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)     # data/data_helper.py:20-30 (pacs, officehome, camelyon17)


# ---------------------------------------------------------------------------
# list handling
# ---------------------------------------------------------------------------
def _dataset_info(txt_labels):
    """data/ImageLoader.py:31-42 (= cjm_util/ImageLoader.py:31-42): 'path label' per line."""
    with open(txt_labels, 'r') as f:
        rows = f.readlines()
    names, labels = [], []
    for row in rows:
        row = row.split(' ')
        names.append(row[0])
        labels.append(int(row[1]))
    return names, labels


def get_random_subset(names, labels, percent):
    """data/ImageLoader.py:13-28: ``int(len * percent)`` indices drawn with ``random.sample`` become the validation set (in
    draw order); the training set is the complement in list order.  Disjoint by construction."""
    samples = len(names)
    amount = int(samples * percent)
    random_index = random.sample(range(samples), amount)
    picked = set(random_index)
    name_val = [names[k] for k in random_index]
    name_train = [v for k, v in enumerate(names) if k not in picked]
    labels_val = [labels[k] for k in random_index]
    labels_train = [v for k, v in enumerate(labels) if k not in picked]
    return name_train, name_val, labels_train, labels_val


def get_split_dataset_info(txt_list, val_percentage=None):
    """data/ImageLoader.py:45-47."""
    names, labels = _dataset_info(txt_list)
    return get_random_subset(names, labels, val_percentage)


def creat_train_loader_list(name_train, labels_train, mode, source, target):
    """data/data_helper.py:125-145 (name kept, typo included): substitute / expand the training list for a fusion mode.
      * a fusion mode without '-K' reads the stylised tree ``kfold_overall-multi/<target>`` instead of ``kfold``;
      * 'single-K' modes read ``single-multi`` instead of ``overall-multi``;
      * 'multi' modes append, for every source domain, a copy of every entry that does not already carry that domain's
        name, renamed ``x.jpg -> x_<domain>.jpg`` (the stylised variants), labels repeated.
    For the shipped ``adain-*-K*`` modes the lists under ``txt_lists/<dataset>_<mode>/<target>/`` are already expanded
    (data/data_list_generator.py) and only the second rule can fire."""
    if mode != 'no_fusion' and '-K' not in mode:
        name_train = [name.replace('kfold', 'kfold_overall-multi' + '/' + target) for name in name_train]
    if "single-K" in mode:
        name_train = [name.replace('overall-multi', 'single-multi') for name in name_train]
    if 'multi' in mode:
        temp_name_list, temp_label_list = [], []
        for domain in source:
            temp_name_list += [name.replace('.', '_' + domain + '.') for name in name_train if domain not in name]
            temp_label_list += labels_train          # the reference repeats the WHOLE label list per domain (:141)
        name_train += temp_name_list
        labels_train += temp_label_list
    return name_train, labels_train


def stylised_name(fpath, target, style, tree):
    """Output path rule of CCST_OverallStyleTransfer.py:160-163 / CCST_SingleStyleTransfer.py:217-219."""
    file_name, ext = os.path.splitext(os.path.basename(fpath))
    out_name = fpath.replace('kfold', tree)
    out_name = out_name.replace('%s' % target, '%s/%s' % (target, style))
    out_name = out_name.replace('%s' % ext, '_%s%s' % (style, ext))
    return out_name


# ---------------------------------------------------------------------------
# decode + transform parameters (host)
# ---------------------------------------------------------------------------
def decode_rgb_u8(path):
    """Image.open(path).convert('RGB') as a contiguous uint8 [H,W,3] tensor (the only per-pixel CPU work left)."""
    from PIL import Image
    with Image.open(path) as im:
        a = np.array(im.convert('RGB'), dtype=np.uint8)      # a writable, contiguous copy
    return torch.from_numpy(a)


def random_resized_crop_params(height, width, scale, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """transforms.RandomResizedCrop.get_params (torchvision >= 0.8, the reference's pin): torch global RNG."""
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


class RawBatch(object):
    """What the raw datasets collate to: decoded uint8 images of any sizes + per-image (i, j, h, w, flip) + labels/paths."""

    def __init__(self, images, params, tags):
        self.images, self.params, self.tags = images, params, tags

    def __len__(self):
        return len(self.images)


def collate_raw(items):
    images = [it[0] for it in items]
    params = torch.tensor([it[1] for it in items], dtype=torch.int32).view(len(items), 5)
    tags = [it[2] for it in items]
    if tags and isinstance(tags[0], int):
        tags = torch.tensor(tags, dtype=torch.int64)
    return RawBatch(images, params, tags)


class ImageDataset(Dataset):
    """data/ImageLoader.py:50-69 with the train transform list of data/data_helper.py:173-178: decodes the file, draws the
    RandomResizedCrop rectangle and the flip; the pixels are transformed on the GPU (DeviceImageLoader)."""

    def __init__(self, names, labels, scale=(0.8, 1.0), flip_p=0.0):
        self.data_path = ""
        self.names, self.labels, self.scale, self.flip_p = names, labels, scale, flip_p

    def __len__(self):
        return len(self.names)

    def __getitem__(self, index):
        img = decode_rgb_u8(self.data_path + '/' + self.names[index])
        i, j, h, w = random_resized_crop_params(img.shape[0], img.shape[1], self.scale)
        flip = int(bool(torch.rand(1) < self.flip_p)) if self.flip_p > 0.0 else 0     # RandomHorizontalFlip is only appended if p > 0
        return img, (i, j, h, w, flip), int(self.labels[index])


class ImageTestDataset(Dataset):
    """data/ImageLoader.py:73-85 (returns the label) and cjm_util/ImageLoader.py:74-85 (returns the file name): whole image,
    Resize((S,S)) on the GPU."""

    def __init__(self, names, labels, with_path=False):
        self.data_path = ""
        self.names, self.labels, self.with_path = names, labels, with_path

    def __len__(self):
        return len(self.names)

    def __getitem__(self, index):
        framename = self.data_path + '/' + self.names[index]
        img = decode_rgb_u8(framename)
        return img, (0, 0, img.shape[0], img.shape[1], 0), (framename if self.with_path else int(self.labels[index]))


class Subset(Dataset):
    """data/data_helper.py:33-43: the first `limit` entries of a torch.randperm."""

    def __init__(self, dataset, limit):
        self.dataset = dataset
        self.indices = torch.randperm(len(dataset))[:limit]

    def __getitem__(self, idx):
        return self.dataset[int(self.indices[idx])]

    def __len__(self):
        return len(self.indices)


class SyntheticImages(Dataset):
    """Stand-in when the image files are absent: seeded uniform[0,1) images (ToTensor range) with the
    list's own file names, so output naming and the rest of the CLI behave as with real data."""

    def __init__(self, names, labels, image_size, seed=1, normalized=False, with_path=True, ids=None):
        self.names, self.labels, self.image_size = names, labels, image_size
        self.seed, self.normalized, self.with_path = seed, normalized, with_path
        self.ids = ids          # global image numbers when this is one rank's shard of a list

    def __len__(self):
        return len(self.names)

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + (self.ids[index] if self.ids is not None else index))
        S = self.image_size
        x = torch.randn(3, S, S, generator=g) if self.normalized else torch.rand(3, S, S, generator=g)
        return (x, self.names[index]) if self.with_path else (x, int(self.labels[index]))


# ---------------------------------------------------------------------------
# device side
# ---------------------------------------------------------------------------
def plan_transform(sizes, params, out_h, out_w):
    """Host half of the GPU transform: (CcstImageXform array, int32 tables) for images of `sizes` [(H,W)] and
    `params` [(i,j,h,w,flip)] -- ccst_image_plan, no GPU needed."""
    lib = _lib.load()
    n = len(sizes)
    xf = (CcstImageXform * n)()
    off = 0
    for k, ((H, W), (i, j, h, w, flip)) in enumerate(zip(sizes, params)):
        if not (0 <= i and 0 <= j and 0 < h and 0 < w and i + h <= H and j + w <= W):
            raise ValueError("crop (%d,%d,%d,%d) outside a %dx%d image" % (i, j, h, w, H, W))
        t = xf[k]
        t.src_off, t.src_w, t.crop_i, t.crop_j, t.crop_h, t.crop_w, t.flip = off, W, i, j, h, w, int(bool(flip))
        off += H * W * 3
    need = lib.ccst_image_plan(n, ctypes.byref(xf), out_h, out_w, None, 0)
    if need < 0:
        check(int(need), "image_plan")
    tables = torch.empty(int(need), dtype=torch.int32)
    used = lib.ccst_image_plan(n, ctypes.byref(xf), out_h, out_w, ctypes.c_void_p(tables.data_ptr()), int(need))
    if used != need:
        check(int(used) if used < 0 else -1, "image_plan")
    return xf, tables


def gpu_transform(images, params, size, device, mean=None, std=None, want_u8=False):
    """[uint8 HWC tensors], [(i,j,h,w,flip)] -> float32 [N,3,S,S] on `device` (and, with want_u8, the un-normalised uint8
    [N,S,S,3] resize): one H2D of the packed decoded pixels, one of the tables, one launch."""
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise RuntimeError("ccst_amd.data: the image transform runs on the GPU; there is no CPU fallback")
    out_h, out_w = (size, size) if isinstance(size, int) else size
    n = len(images)
    if isinstance(params, torch.Tensor):
        params = params.tolist()
    for im in images:
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
            raise ValueError("gpu_transform: images must be uint8 [H,W,3]")
    xf, tables = plan_transform([(int(im.shape[0]), int(im.shape[1])) for im in images], params, out_h, out_w)
    src = torch.cat([im.reshape(-1) for im in images]).to(dev)
    xf_dev = torch.frombuffer(xf, dtype=torch.uint8).to(dev)
    tab_dev = tables.to(dev)
    out = torch.empty((n, 3, out_h, out_w), device=dev, dtype=torch.float32)
    u8 = torch.empty((n, out_h, out_w, 3), device=dev, dtype=torch.uint8) if want_u8 else None
    m = (ctypes.c_float * 3)(*(mean if mean is not None else (0.0, 0.0, 0.0)))
    s = (ctypes.c_float * 3)(*(std if std is not None else (1.0, 1.0, 1.0)))
    with torch.cuda.device(dev):
        check(_lib.load().ccst_crop_resize_norm_u8_f32(ptr(src), ptr(xf_dev), ptr(tab_dev), ptr(out), ptr(u8), n, out_h, out_w,
                                                       m, s, stream_ptr()), "crop_resize_norm")
    return (out, u8) if want_u8 else out


class DeviceImageLoader(object):
    """Wraps a DataLoader of RawBatch into the (tensor, labels-or-paths) batches train()/test()/style_transfer() take,
    with the tensor already transformed and resident on the GPU."""

    def __init__(self, loader, size, device, mean=None, std=None):
        self.loader, self.size, self.device, self.mean, self.std = loader, size, device, mean, std
        self.dataset = loader.dataset
        self.batch_size = loader.batch_size

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        dev = self.device if self.device is not None else torch.device('cuda', torch.cuda.current_device())
        for raw in self.loader:
            if isinstance(raw, RawBatch):
                yield gpu_transform(raw.images, raw.params, self.size, dev, self.mean, self.std), raw.tags
            else:                       # synthetic stand-ins are already tensors
                yield raw


def _device(args):
    d = getattr(args, 'device', None)
    return torch.device(d) if d is not None else None        # None: the current CUDA device when a batch is first transformed


def get_train_dataloader(args, txt_root='cjm_util/txt_lists', rank=0, world=1):
    """cjm_util/data_helper.py:38-49 (AdaIN scripts): Resize((S,S)) -> ToTensor, (tensor, path) items, shuffle=True.
    Under torchrun the LIST is sharded (entries rank, rank+world, ...; disjoint and complete whatever each rank's RNG
    does) and the shard is read in order."""
    lst = os.path.join(txt_root, args.dataset.lower(), '%s_train.txt' % args.target)
    synthetic = int(getattr(args, 'synthetic', 0) or 0)
    if synthetic:
        out = getattr(args, 'output', 'output')
        names = [os.path.join(out, 'synthetic/kfold/%s/class0/img_%05d.jpg' % (args.target, i)) for i in range(synthetic)]
        labels = [0] * synthetic
    else:
        names, labels = _dataset_info(lst)
    ids = list(range(len(names)))
    if world > 1:
        names, labels, ids = names[rank::world], labels[rank::world], ids[rank::world]
    if synthetic:
        import zlib
        ds = SyntheticImages(names, labels, args.image_size, seed=1 + zlib.crc32(str(args.target).encode()) % 1000, ids=ids)
        return DataLoader(ds, batch_size=args.batch, shuffle=(world == 1))
    ds = ImageTestDataset(names, labels, with_path=True)
    loader = DataLoader(ds, batch_size=args.batch, shuffle=(world == 1), collate_fn=collate_raw,
                        num_workers=int(getattr(args, 'workers', 0) or 0))
    return DeviceImageLoader(loader, args.image_size, _device(args))


def quantize_u8(images):
    """[N,C,H,W] float CUDA -> [N,H,W,C] uint8, torchvision.utils.save_image's ``mul(255).add_(0.5).clamp_(0,255).to(uint8)``
    byte for byte, one HIP launch."""
    x = images.contiguous()
    N, C, H, W = x.shape
    y = torch.empty((N, H, W, C), device=x.device, dtype=torch.uint8)
    check(_lib.load().ccst_quantize_u8_hwc_f32(ptr(x), ptr(y), N, C, H * W, stream_ptr()), "quantize_u8")
    return y


def resize_tensor(images, size):
    """transforms.Resize(size) on a float tensor batch, as CCST_OverallStyleTransfer.py:154-157 applies it to the stylised output:
    the smaller edge becomes `size` (aspect kept; torchvision's int-size rule), bilinear, align_corners=False, no antialiasing."""
    x = images.contiguous()
    N, C, H, W = x.shape
    if W <= H:
        ow, oh = size, int(size * H / W)
    else:
        oh, ow = size, int(size * W / H)
    if (oh, ow) == (H, W):
        return x
    y = torch.empty((N, C, oh, ow), device=x.device, dtype=torch.float32)
    check(_lib.load().ccst_resize_bilinear_nchw_f32(ptr(x), ptr(y), N * C, H, W, oh, ow, stream_ptr()), "resize_bilinear")
    return y


def save_images(output, paths, output_size=-1):
    """torchvision.utils.save_image per image (CCST_OverallStyleTransfer.py:154-167): the optional tensor resize, the uint8
    quantisation on the GPU, copy to the host, encode with PIL.  `paths` are the final file names."""
    from PIL import Image
    if output_size and output_size > 0:
        output = resize_tensor(output, output_size)
    u8 = quantize_u8(output).cpu().numpy()
    for arr, name in zip(u8, paths):
        d = os.path.dirname(name)
        if d and not os.path.exists(d):
            os.makedirs(d)
        Image.fromarray(arr).save(name)


def _encode_one(arr, name):
    from PIL import Image
    d = os.path.dirname(name)
    if d:
        os.makedirs(d, exist_ok=True)
    Image.fromarray(arr).save(name)
    return name


class ImageWriterPool(object):
    """PNG / JPEG encoding of finished batches off the main thread: at > 1000 images/s of stylised output one Python thread
    encoding ~50 images/s is what the stage-2 loop waits for (the reference encodes inline, CCST_OverallStyleTransfer.py:158-167).
    Worker PROCESSES when the pool is created before this process has touched the GPU (they are forked at once, idle, and never
    see a HIP context -- the CLIs create the pool first thing); otherwise worker threads (zlib releases the GIL while it
    compresses).  submit() copies the uint8 HWC arrays out of the (reused) pinned buffer; close() waits for every file."""

    def __init__(self, workers=None):
        import concurrent.futures as cf
        n = workers or max(1, min(16, (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4) - 1))
        self.futures = []
        self.pids = None
        if torch.cuda.is_initialized():
            self.kind, self.pool = "threads", cf.ThreadPoolExecutor(max_workers=n)
        else:
            # multiprocessing.Pool forks ALL n workers in its constructor (documented behaviour, not an executor's lazy start-up): they
            # exist before this process touches the GPU and never see a HIP context.  A worker that dies would be replaced by a fork
            # of a process that by then holds a HIP context and live streams: that is treated as fatal instead (check_workers).
            import multiprocessing as mp
            self.kind, self.pool = "processes", mp.get_context("fork").Pool(n)
            self.pids = sorted(p.pid for p in self.pool._pool)

    def check_workers(self):
        if self.pids is not None and sorted(p.pid for p in self.pool._pool) != self.pids:
            raise RuntimeError("ccst_amd.data.ImageWriterPool: an encoder process died and was re-forked after GPU initialisation; "
                               "output files may be missing -- rerun (or create the pool with threads)")

    def submit(self, u8_batch, paths):
        # (the array is a view of a pinned buffer the pipeline reuses, and both pool kinds serialise / run a task later, on another
        #  thread: the copy here is what makes that safe; pickling it for a worker process is the second, unavoidable one)
        for arr, name in zip(u8_batch, paths):
            a = np.ascontiguousarray(arr).copy()
            self.futures.append(self.pool.apply_async(_encode_one, (a, name)) if self.kind == "processes" else self.pool.submit(_encode_one, a, name))
        if len(self.futures) > 4096:
            self.drain()

    def drain(self):
        import multiprocessing as mp
        for f in self.futures:
            if self.kind != "processes":
                f.result()
                continue
            # multiprocessing.Pool never completes the AsyncResult of a task whose worker died (OOM kill, signal): poll, and look at the
            # worker set between polls, so that a dead encoder raises here instead of blocking the run for ever
            while True:
                try:
                    f.get(timeout=0.5)
                    break
                except mp.TimeoutError:
                    self.check_workers()
        self.futures = []
        self.check_workers()

    def close(self):
        self.drain()
        if self.kind == "processes":
            self.pool.close()
            self.pool.join()
        else:
            self.pool.shutdown()


# ---------------------------------------------------------------------------
# federated training loaders (data/data_helper.py:46-123,148-159)
# ---------------------------------------------------------------------------
def _train_list_dir(args, txt_root):
    return os.path.join(txt_root, '%s_%s/%s' % (args.dataset.lower(), args.fusion_mode, args.target))


def fed_lists(args, txt_root='data/txt_lists'):
    """The list logic of data_helper.get_train_dataloader without any image access: per client
    (name_train, labels_train, name_val, labels_val), in the reference's order of RNG draws.  ``--mode deepall``
    yields ONE client: every source's training entries concatenated, validated on the LAST source's split
    (data/data_helper.py:66-70,82-84,104-113: name_val / labels_val are simply the loop's last values)."""
    out, all_names, all_labels = [], [], []
    name_val, labels_val = [], []
    for dname in args.source:
        print("Prepare %s ..." % dname)
        name_train, name_val, labels_train, labels_val = get_split_dataset_info(
            os.path.join(_train_list_dir(args, txt_root), '%s_train.txt' % dname), args.val_size)
        name_train, labels_train = creat_train_loader_list(name_train, labels_train, args.fusion_mode, args.source, args.target)
        if args.mode == 'deepall':
            all_names += name_train
            all_labels += labels_train
        else:
            out.append((name_train, labels_train, name_val, labels_val))
    if args.mode == 'deepall':
        out.append((all_names, all_labels, name_val, labels_val))
    return out


def get_fed_dataloaders(args, txt_root='data/txt_lists'):
    """(train_loaders, val_loaders, target_test_loader): data_helper.get_train_dataloader + get_test_dataloader
    (data/data_helper.py:46-123,148-159).  With --synthetic N every client gets N seeded N(0,1) images
    (post-Normalize statistics) instead of files."""
    synthetic = int(getattr(args, 'synthetic', 0) or 0)
    train_loaders, val_loaders = [], []
    limit = getattr(args, 'limit_source', None)
    workers = int(getattr(args, 'workers', 0) or 0)
    if synthetic:
        sources = args.source if args.mode != 'deepall' else ['+'.join(args.source)]
        for di, dname in enumerate(sources):
            names = ['%s_%d' % (dname, i) for i in range(synthetic)]
            labels = [i % args.n_classes for i in range(synthetic)]
            nval = max(1, synthetic // 10)
            tr = SyntheticImages(names, labels, args.image_size, seed=args.seed + di, normalized=True, with_path=False)
            va = SyntheticImages(['val_' + n for n in names[:nval]], labels[:nval], args.image_size, seed=100 + di, normalized=True,
                                 with_path=False)
            if limit:
                tr, va = Subset(tr, limit), Subset(va, limit)
            train_loaders.append(DataLoader(tr, batch_size=args.batch, shuffle=True))
            val_loaders.append(DataLoader(va, batch_size=args.batch, shuffle=False))
        names = ['target_%d' % i for i in range(synthetic)]
        te = SyntheticImages(names, [i % args.n_classes for i in range(synthetic)], args.image_size, seed=999, normalized=True,
                             with_path=False)
        if getattr(args, 'limit_target', None) and len(te) > args.limit_target:
            te = Subset(te, args.limit_target)
        return train_loaders, val_loaders, DataLoader(te, batch_size=args.batch, shuffle=True)
    dev = _device(args)
    for name_train, labels_train, name_val, labels_val in fed_lists(args, txt_root):
        tr = ImageDataset(name_train, labels_train, scale=(args.min_scale, args.max_scale), flip_p=args.random_horiz_flip)
        va = ImageTestDataset(name_val, labels_val)
        if limit:
            tr, va = Subset(tr, limit), Subset(va, limit)
        train_loaders.append(DeviceImageLoader(
            DataLoader(tr, batch_size=args.batch, shuffle=True, collate_fn=collate_raw, num_workers=workers),
            args.image_size, dev, MEAN, STD))
        val_loaders.append(DeviceImageLoader(
            DataLoader(va, batch_size=args.batch, shuffle=False, collate_fn=collate_raw, num_workers=workers),
            args.image_size, dev, MEAN, STD))
    names, labels = _dataset_info(os.path.join(txt_root, args.dataset, '%s_test.txt' % args.target))
    te = ImageTestDataset(names, labels)
    if getattr(args, 'limit_target', None) and len(te) > args.limit_target:
        te = Subset(te, args.limit_target)
        print("Using %d subset of val dataset" % args.limit_target)
    test_loader = DeviceImageLoader(                                        # shuffle=True as in the reference (:158)
        DataLoader(te, batch_size=args.batch, shuffle=True, collate_fn=collate_raw, num_workers=workers),
        args.image_size, dev, MEAN, STD)
    return train_loaders, val_loaders, test_loader
