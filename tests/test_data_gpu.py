"""GPU: the image input / output edges through the C ABI against the PIL / torchvision-semantics chain of
oracle/image_ref.py.  Byte work is held to byte equality, the ToTensor / Normalize floats to bit equality."""
import types

import numpy as np
import pytest
import torch

from oracle import image_ref as I

pytestmark = pytest.mark.gpu
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _img(rs, h, w, smooth=False):
    a = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    if smooth:          # photo-like gradients as well as noise
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([(yy * 255 // max(1, h - 1)), (xx * 255 // max(1, w - 1)), ((yy + xx) % 256)], -1).astype(np.uint8)
    return a


def test_crop_resize_normalize_flip_is_the_pil_chain(dev):
    """A ragged batch (different source sizes, crops, up- and down-scales to 7.6x, flips): uint8 resize byte-equal to
    img.crop().resize(BILINEAR); float output bit-equal to ToTensor -> Normalize -> hflip."""
    from PIL import Image
    from ccst_amd import data
    rs = np.random.RandomState(3)
    S = 222
    cases = [((227, 227), (11, 7, 205, 212), 0), ((227, 227), (0, 0, 227, 227), 1), ((96, 96), (3, 5, 90, 88), 0),
             ((500, 375), (20, 30, 400, 300), 1), ((1700, 1300), (0, 0, 1700, 1300), 0), ((222, 222), (0, 0, 222, 222), 0),
             ((40, 333), (2, 100, 30, 222), 1), ((227, 227), (5, 5, 222, 200), 0)]
    imgs = [_img(rs, h, w, smooth=(k % 3 == 2)) for k, ((h, w), _, _) in enumerate(cases)]
    params = [(i, j, h, w, f) for _, (i, j, h, w), f in cases]
    out, u8 = data.gpu_transform([torch.from_numpy(a) for a in imgs], params, S, dev, MEAN, STD, want_u8=True)
    out, u8 = out.cpu(), u8.cpu().numpy()
    for k, (a, (i, j, h, w, f)) in enumerate(zip(imgs, params)):
        pil = I.resized_crop(Image.fromarray(a), i, j, h, w, (S, S))
        ref_u8 = np.asarray(pil)
        if f:
            ref_u8 = ref_u8[:, ::-1]
        assert np.array_equal(u8[k], ref_u8), k
        ref = I.normalize(I.to_tensor(pil), MEAN, STD)
        if f:
            ref = I.hflip(ref)
        assert torch.equal(out[k], ref), (k, float((out[k] - ref).abs().max()))


def test_resize_to_tensor_only_and_non_square(dev):
    """The AdaIN loaders' chain (cjm_util/data_helper.py:46-49: Resize((S,S)) -> ToTensor, no Normalize) and a non-square target."""
    from PIL import Image
    from ccst_amd import data
    rs = np.random.RandomState(4)
    a, b = _img(rs, 227, 227), _img(rs, 300, 200)
    out = data.gpu_transform([torch.from_numpy(a), torch.from_numpy(b)], [(0, 0, 227, 227, 0), (0, 0, 300, 200, 0)], 512, dev).cpu()
    for k, im in enumerate((a, b)):
        assert torch.equal(out[k], I.val_transform(Image.fromarray(im), 512))
    out2 = data.gpu_transform([torch.from_numpy(b)], [(0, 0, 300, 200, 0)], (64, 100), dev).cpu()
    assert torch.equal(out2[0], I.to_tensor(Image.fromarray(b).resize((100, 64), Image.BILINEAR)))
    with pytest.raises(RuntimeError):
        data.gpu_transform([torch.from_numpy(a)], [(0, 0, 227, 227, 0)], 64, "cpu")           # no CPU fallback


def test_loaders_on_real_files(dev, tmp_path):
    """PNG / JPEG files written in-test -> the fed train / val loaders and the AdaIN loader -> the oracle's PIL chain with the
    same RNG draws (per item: crop rectangle, then flip), labels and paths intact."""
    from PIL import Image
    from ccst_amd import data
    rs = np.random.RandomState(5)
    names, labels = [], []
    for k, (h, w) in enumerate([(227, 227), (227, 227), (180, 240), (96, 96), (300, 260)]):
        p = tmp_path / ("img_%d.%s" % (k, "jpg" if k % 2 else "png"))
        Image.fromarray(_img(rs, h, w, smooth=True)).save(str(p))
        names.append(str(p)[1:])                    # the datasets prepend '/' (data/ImageLoader.py:57)
        labels.append(k % 3)
    S = 222
    ds = data.ImageDataset(names, labels, scale=(0.8, 1.0), flip_p=0.5)
    torch.manual_seed(7)
    items = [ds[k] for k in range(len(ds))]
    batch = data.collate_raw(items)
    got = data.gpu_transform(batch.images, batch.params, S, dev, MEAN, STD).cpu()
    torch.manual_seed(7)
    for k in range(len(ds)):
        ref = I.train_transform(Image.open('/' + names[k]).convert('RGB'), S, (0.8, 1.0), MEAN, STD, 0.5)
        assert torch.equal(got[k], ref), k
    assert batch.tags.tolist() == labels
    # val / test loader (Resize -> ToTensor -> Normalize), through DeviceImageLoader + DataLoader
    from torch.utils.data import DataLoader
    ld = data.DeviceImageLoader(DataLoader(data.ImageTestDataset(names, labels), batch_size=2, shuffle=False, collate_fn=data.collate_raw),
                                S, dev, MEAN, STD)
    assert len(ld) == 3
    seen = 0
    for x, y in ld:
        assert x.is_cuda and x.shape[1:] == (3, S, S) and y.dtype == torch.int64
        for r in range(x.shape[0]):
            ref = I.val_transform(Image.open('/' + names[seen]).convert('RGB'), S, MEAN, STD)
            assert torch.equal(x[r].cpu(), ref) and int(y[r]) == labels[seen]
            seen += 1
    assert seen == 5
    # AdaIN loader: list file -> (tensor in [0,1], path)
    lst = tmp_path / "pacs"
    lst.mkdir()
    (lst / "photo_train.txt").write_text("".join("%s %d\n" % (n, l) for n, l in zip(names, labels)))
    args = types.SimpleNamespace(dataset="pacs", target="photo", batch=5, image_size=64, synthetic=0)
    x, paths = next(iter(data.get_train_dataloader(args, str(tmp_path))))
    assert sorted(paths) == sorted('/' + n for n in names) and float(x.min()) >= 0.0 and float(x.max()) <= 1.0
    for r, p in enumerate(paths):
        assert torch.equal(x[r].cpu(), I.val_transform(Image.open(p).convert('RGB'), 64))


def test_quantize_u8_is_save_image_bytes(dev):
    """f1: quantize_u8 == torchvision.utils.save_image's ``mul(255).add_(0.5).clamp_(0,255).permute(1,2,0).to(uint8)``
    (CCST_OverallStyleTransfer.py:167), byte for byte: values exactly on k/255 and (k+0.5)/255 boundaries and one float either
    side of them, negatives, values above 1, and a realistic decoder output range."""
    from ccst_amd import data
    k = torch.arange(0, 256, dtype=torch.float32)
    edges = torch.cat([k / 255, (k + 0.5) / 255, (k - 0.5) / 255])
    near = torch.cat([torch.nextafter(edges, torch.tensor(2.0)), torch.nextafter(edges, torch.tensor(-2.0))])
    special = torch.tensor([-3.0, -1e-8, 0.0, 1.0, 1.0000001, 2.5, 254.49999 / 255, 254.5 / 255, 0.999999, 1e-3])
    g = torch.Generator().manual_seed(1)
    rand = torch.rand(3 * 40 * 50 - edges.numel() - near.numel() - special.numel(), generator=g) * 1.4 - 0.2
    x = torch.cat([edges, near, special, rand]).view(1, 3, 40, 50)
    x = torch.cat([x, torch.randn(1, 3, 40, 50, generator=g) * 0.5 + 0.5])
    got = data.quantize_u8(x.to(dev)).cpu().numpy()
    for n in range(2):
        assert np.array_equal(got[n], I.save_image_bytes(x[n])), n


def test_save_images_writes_the_quantised_bytes(dev, tmp_path):
    """data.save_images: the PNG on disk decodes to exactly save_image's uint8 array; directories are created (:164-166)."""
    from PIL import Image
    from ccst_amd import data
    g = torch.Generator().manual_seed(2)
    out = torch.rand(2, 3, 33, 47, generator=g) * 1.2 - 0.1
    paths = [str(tmp_path / "a" / "b" / ("x_%d.png" % i)) for i in range(2)]
    data.save_images(out.to(dev), paths)
    for i, p in enumerate(paths):
        assert np.array_equal(np.asarray(Image.open(p)), I.save_image_bytes(out[i]))
