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


def _write_pacs_tree(root, size=40):
    """A miniature PACS: root/PACS/kfold/<domain>/<class>/pic_<k>.png + list files in both CLIs' layouts; returns the list roots."""
    import os
    from PIL import Image
    rs = np.random.RandomState(9)
    domains, classes = ["art_painting", "cartoon", "photo", "sketch"], ["dog", "house"]
    rows = {d: [] for d in domains}
    for d in domains:
        for ci, c in enumerate(classes):
            os.makedirs(os.path.join(root, "PACS", "kfold", d, c), exist_ok=True)
            for k in range(3):
                p = os.path.join(root, "PACS", "kfold", d, c, "pic_%03d.png" % k)
                Image.fromarray(_img(rs, size + 3 * k, size + ci, smooth=(k == 1))).save(p)
                rows[d].append((p, ci))       # absolute paths, as in the reference's lists (the loaders prepend one more '/', ImageLoader.py:57)
    for sub in ("adain_lists/pacs", "fed_lists/pacs", "fed_lists/pacs_no_fusion/photo"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    for d in domains:
        text = "".join("%s %d\n" % r for r in rows[d])
        for name in ("%s_train.txt" % d, "%s_test.txt" % d):
            open(os.path.join(root, "adain_lists/pacs", name), "w").write(text)
            open(os.path.join(root, "fed_lists/pacs", name), "w").write(text)
        if d != "photo":
            open(os.path.join(root, "fed_lists/pacs_no_fusion/photo", "%s_train.txt" % d), "w").write(text)
    return rows


def test_adain_clis_on_real_files_vs_oracle(dev, tmp_path):
    """Stage 1 + stage 2 (Overall and Single) on image FILES through the drop-in CLIs: list parsing, decode, GPU resize, encoder /
    AdaIN / decoder, GPU quantise, PNG write, output naming.  A stylised image is compared, byte for byte up to +-1 level, with the
    oracle run on the same file (PIL Resize -> ToTensor -> the oracle's style_transfer -> save_image's bytes)."""
    import os
    import subprocess
    import sys
    from PIL import Image
    from oracle import adain_ref as A
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "style_transfer", "AdaIN")
    rows = _write_pacs_tree(str(tmp_path))
    env = dict(os.environ, PYTHONPATH=root)
    common = ["--dataset", "pacs", "--random_weights", "--batch", "4", "--image_size", "64", "--txt_root", str(tmp_path / "adain_lists"),
              "--output", str(tmp_path / "out")]
    for dom in ("cartoon", "photo", "sketch"):
        subprocess.check_call([sys.executable, os.path.join(d, "mean_std_computation_effcientMem.py"), "--target", dom] + common,
                              cwd=str(tmp_path), env=env, stdout=subprocess.DEVNULL)
    subprocess.check_call([sys.executable, os.path.join(d, "CCST_OverallStyleTransfer.py"), "--target", "art_painting"] + common,
                          cwd=str(tmp_path), env=env, stdout=subprocess.DEVNULL)
    subprocess.check_call([sys.executable, os.path.join(d, "CCST_SingleStyleTransfer.py"), "--target", "art_painting", "--style_size", "48"] + common,
                          cwd=str(tmp_path), env=env, stdout=subprocess.DEVNULL)
    src = rows["art_painting"][1][0]
    out_overall = ("/" + src).replace("kfold", "all_style_transferred_Overall").replace("art_painting", "art_painting/cartoon").replace(".png", "_cartoon.png")
    out_single = ("/" + src).replace("kfold", "all_style_transferred_Single").replace("art_painting", "art_painting/sketch").replace(".png", "_sketch.png")
    assert os.path.exists(out_overall) and os.path.exists(out_single), (out_overall, os.listdir(str(tmp_path)))
    n_out = sum(len(fs) for _, _, fs in os.walk(str(tmp_path / "PACS" / "all_style_transferred_Overall")))
    assert n_out == 3 * 6                                           # 3 style domains x 6 content images
    # the oracle on the same file: the CLI's --random_weights are seeded He-normal in module order (_common.load_networks)
    g = torch.Generator().manual_seed(1234)
    vgg_w, dec_w = {}, {}
    for table, wd in ((A.VGG_TABLE, vgg_w), (A.DECODER_TABLE, dec_w)):
        for idx, cin, cout, k in A.conv_keys(table):
            wd["%d.weight" % idx] = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
            wd["%d.bias" % idx] = torch.randn((cout,), generator=g) * 0.05
    stat = np.load(str(tmp_path / "style_stats" / "pacs" / "cartoon_mean_std.npy"))
    content = I.val_transform(Image.open(src).convert("RGB"), 64).unsqueeze(0)
    with torch.no_grad():
        ref = A.style_transfer(vgg_w, dec_w, content, [torch.from_numpy(stat[0]), torch.from_numpy(stat[1])], 1.0)
    want = I.save_image_bytes(ref[0]).astype(np.int32)
    got = np.asarray(Image.open(out_overall).convert("RGB")).astype(np.int32)
    assert got.shape == want.shape == (64, 64, 3)
    assert np.abs(got - want).max() <= 1 and (got != want).mean() < 0.02      # fp32 differences of <= 1e-3 can flip a rounding at a level boundary
    # ... and the style statistics file against the oracle's stage-1 loop over the cartoon files (list order, batch 4)
    batches, cur = [], []
    for name, _ in rows["cartoon"]:
        cur.append(I.val_transform(Image.open(name).convert("RGB"), 64))
        if len(cur) == 4:
            batches.append(torch.stack(cur))
            cur = []
    if cur:
        batches.append(torch.stack(cur))
    with torch.no_grad():
        mean, std = A.overall_style_stats(batches, vgg_w)
    assert np.abs(stat[0] - mean.numpy()).max() < 1e-3 and np.abs(stat[1] - std.numpy()).max() < 1e-3


def test_fed_run_cli_on_real_files(dev, tmp_path):
    """federated/fed_run.py on image files: the reference's list layout (txt_lists/<dataset>_<fusion>/<target>/<source>_train.txt),
    the random disjoint validation split, GPU-side train / val transforms inside train() / test(), two rounds of FedAvg."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _write_pacs_tree(str(tmp_path), size=230)
    env = dict(os.environ, PYTHONPATH=root)
    cmd = [sys.executable, os.path.join(root, "federated", "fed_run.py"), "--mode", "fedavg", "--fusion_mode", "no_fusion",
           "--source", "art_painting", "cartoon", "sketch", "--target", "photo", "--n_classes", "2", "--network", "resnet18",
           "--lr", "0.001", "--image_size", "222", "--batch", "4", "--val_size", "0.34", "--random_horiz_flip", "0.5",
           "--txt_root", str(tmp_path / "fed_lists"), "--save_path", str(tmp_path / "ckpt"), "--iters", "2"]
    out = subprocess.check_output(cmd, cwd=str(tmp_path), env=env, text=True)
    assert out.count("| Train Loss:") == 6 and out.count("| Global Val Class Acc:") == 6 and "| Global Test Class Acc:" in out
    losses = [float(l.split(":")[1]) for l in out.splitlines() if "| Train Loss:" in l]
    assert all(np.isfinite(losses)) and all(0.0 < x < 5.0 for x in losses)


def test_output_size_resizes_the_tensor_like_the_reference(dev, tmp_path):
    """--output_size (CCST_OverallStyleTransfer.py:154-157): transforms.Resize on the float tensor BEFORE save_image = F.interpolate
    (bilinear, align_corners=False, no antialias); up- and down-scaling, non-square."""
    import torch.nn.functional as F
    from PIL import Image
    from ccst_amd import data
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 40, 56, generator=g) * 1.2 - 0.1
    for size in (24, 40, 77):
        got = data.resize_tensor(x.to(dev), size).cpu()
        oh, ow = (size, int(size * 56 / 40))
        ref = F.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=False)
        assert got.shape == ref.shape and float((got - ref).abs().max()) < 2e-6
    paths = [str(tmp_path / ("o_%d.png" % i)) for i in range(2)]
    data.save_images(x.to(dev), paths, output_size=24)
    ref = F.interpolate(x, size=(24, 33), mode="bilinear", align_corners=False)
    for i, p in enumerate(paths):
        got = np.asarray(Image.open(p)).astype(np.int32)
        want = I.save_image_bytes(ref[i]).astype(np.int32)
        assert got.shape == want.shape and np.abs(got - want).max() <= 1 and (got != want).mean() < 0.01


def test_config5_camelyon17_k4_end_to_end(dev, tmp_path):
    """BASELINE config 5 end to end on generated files (README.md:109; SURVEY App. C-10 for the 222 size): a miniature Camelyon17
    (five hospitals, two classes, 96x96 patches), hospital1's training patches stylised with the other hospitals' overall style by the
    drop-in stage-2 CLI (--fuse_stats computes the statistics in-process, the overlapped pipeline + writer processes save the
    files), the expanded adain-overall-K4 lists (originals + stylised variants, data/data_list_generator.py), then
    fed_run.py --dataset camelyon17 --network resnet18 --n_classes 2 --fusion_mode adain-overall-K4 with four clients."""
    import os
    import subprocess
    import sys
    from PIL import Image
    from ccst_amd import data
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rs = np.random.RandomState(17)
    hospitals, classes = ["hospital%d" % k for k in range(1, 6)], ["normal", "tumor"]
    rows = {h: [] for h in hospitals}
    for h in hospitals:
        for ci, c in enumerate(classes):
            os.makedirs(str(tmp_path / "Camelyon17" / "kfold" / h / c), exist_ok=True)
            for k in range(3):
                p = str(tmp_path / "Camelyon17" / "kfold" / h / c / ("patch_%03d.png" % k))
                Image.fromarray(_img(rs, 96, 96, smooth=(k != 1))).save(p)
                rows[h].append((p, ci))
    os.makedirs(str(tmp_path / "adain_lists" / "camelyon17"))
    os.makedirs(str(tmp_path / "fed_lists" / "camelyon17"))
    os.makedirs(str(tmp_path / "fed_lists" / "camelyon17_adain-overall-K4" / "hospital5"))
    for h in hospitals:
        text = "".join("%s %d\n" % r for r in rows[h])
        (tmp_path / "adain_lists" / "camelyon17" / ("%s_train.txt" % h)).write_text(text)
        (tmp_path / "fed_lists" / "camelyon17" / ("%s_test.txt" % h)).write_text(text)
    env = dict(os.environ, PYTHONPATH=root)
    # stage 2 for hospital1 (statistics of the four style hospitals computed in the same process)
    out = subprocess.check_output([sys.executable, os.path.join(root, "style_transfer", "AdaIN", "CCST_OverallStyleTransfer.py"),
                                   "--dataset", "camelyon17", "--target", "hospital1", "--random_weights", "--batch", "4", "--image_size", "96",
                                   "--fuse_stats", "--txt_root", str(tmp_path / "adain_lists"), "--output", str(tmp_path / "out")],
                                  cwd=str(tmp_path), env=env, text=True)
    assert out.count("computed style statistics of") == 4
    sources = hospitals[:4]
    for h in sources:
        names, labels = [], []
        for p, ci in rows[h]:
            names.append(p)
            labels.append(ci)
            for s in sources:
                if s == h:
                    continue
                q = data.stylised_name(p, h, s, "all_style_transferred_Overall")
                if h == "hospital1":
                    assert os.path.exists(q), q                       # written by the stage-2 CLI above
                    assert Image.open(q).size == (96, 96)
                else:                                                  # (the other clients: stand-in files under the same naming rule)
                    os.makedirs(os.path.dirname(q), exist_ok=True)
                    Image.fromarray(_img(rs, 96, 96, smooth=True)).save(q)
                names.append(q)
                labels.append(ci)
        (tmp_path / "fed_lists" / "camelyon17_adain-overall-K4" / "hospital5" / ("%s_train.txt" % h)).write_text(
            "".join("%s %d\n" % r for r in zip(names, labels)))
    cmd = [sys.executable, os.path.join(root, "federated", "fed_run.py"), "--dataset", "camelyon17", "--mode", "fedavg",
           "--fusion_mode", "adain-overall-K4", "--source"] + sources + ["--target", "hospital5", "--n_classes", "2", "--network", "resnet18",
           "--lr", "0.001", "--image_size", "222", "--batch", "8", "--val_size", "0.25", "--txt_root", str(tmp_path / "fed_lists"),
           "--save_path", str(tmp_path / "ckpt"), "--iters", "2"]
    out = subprocess.check_output(cmd, cwd=str(tmp_path), env=env, text=True)
    assert out.count("| Train Loss:") == 8 and out.count("| Global Val Class Acc:") == 8 and out.count("| Global Test Class Acc:") == 2
    losses = [float(l.split(":")[1]) for l in out.splitlines() if "| Train Loss:" in l]
    assert all(np.isfinite(losses)) and all(0.0 < x < 5.0 for x in losses)
    d = tmp_path / "ckpt" / "camelyon17" / "fedavg_adain-overall-K4_no_DG_resnet18_locIter1" / "Target_hospital5_seed_1"
    ck = torch.load(str(d / "fedavg_latest"), map_location="cpu")
    assert set(ck.keys()) == {"server_model", "a_iter"} and tuple(ck["server_model"]["class_classifier.weight"].shape) == (2, 512)
