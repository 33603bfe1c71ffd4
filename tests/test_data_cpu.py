"""CPU: the host half of the data plane against the reference's own list logic (tests/golden/data_lists.npz, made by
tools/make_golden.py running data/data_helper.py's get_train_dataloader / get_test_dataloader / creat_train_loader_list and
data/ImageLoader.py's get_random_subset), and the PIL-resample tables of the C-ABI planner against PIL."""
import json
import os
import random
import types

import numpy as np
import pytest
import torch

from oracle import image_ref as I


@pytest.fixture(scope="module")
def lists(golden):
    return json.loads(bytes(golden("data_lists")["json"]).decode())


def test_creat_train_loader_list(lists):
    from ccst_amd import data
    for c in lists["direct"]:
        names, labels = data.creat_train_loader_list(list(c["names_in"]), list(c["labels_in"]), c["mode"],
                                                     ["art_painting", "cartoon", "sketch"], "photo")
        assert names == c["names"] and labels == c["labels"], c["mode"]
    multi = [c for c in lists["direct"] if c["mode"] == "adain-overall-multi"][0]
    assert len(multi["names"]) > len(multi["names_in"]) and "kfold_overall-multi/photo" in multi["names"][0]


def test_random_val_split_is_the_references_and_disjoint(lists):
    from ccst_amd import data
    random.seed(a=1)                                                    # fed_run.py:510
    got = data.get_random_subset(["n%d" % i for i in range(57)], [i % 7 for i in range(57)], 0.1)
    assert [list(x) for x in got] == lists["split57"]
    name_train, name_val = got[0], got[1]
    assert len(name_val) == 5 and not set(name_train) & set(name_val) and len(name_train) + len(name_val) == 57


def _describe(loader):
    from ccst_amd import data
    inner = loader.loader if isinstance(loader, data.DeviceImageLoader) else loader
    ds, idx = inner.dataset, None
    if isinstance(ds, data.Subset):
        idx, ds = [int(i) for i in ds.indices], ds.dataset
    return {"names": list(ds.names), "labels": [int(x) for x in ds.labels], "indices": idx,
            "kind": "train" if isinstance(ds, data.ImageDataset) else "test", "batch_size": inner.batch_size,
            "shuffle": isinstance(inner.sampler, torch.utils.data.RandomSampler)}


def test_fed_loaders_match_reference_lists(lists, tmp_path):
    """fedavg and deepall, every fusion-mode family, --limit_source / --limit_target: same training / validation / test
    entries, in the same order, same Subset permutations, same batch size and shuffle flags as the reference's loaders."""
    from ccst_amd import data
    assert len(lists["cases"]) >= 6
    for c in lists["cases"]:
        root = tmp_path / c["root"]
        for rel, text in c["lists"].items():
            f = root / rel
            f.parent.mkdir(parents=True, exist_ok=True)
            f.write_text(text)
        args = types.SimpleNamespace(source=list(c["source"]), target=c["target"], dataset="pacs", fusion_mode=c["fusion_mode"],
                                     mode=c["mode"], val_size=0.1, dg_method="no_DG", limit_source=c["limit_source"],
                                     limit_target=c["limit_target"], batch=4, image_size=222, min_scale=0.8, max_scale=1.0,
                                     random_horiz_flip=0.0, n_classes=3, seed=1)
        random.seed(a=1)
        torch.manual_seed(1)
        train, val, test = data.get_fed_dataloaders(args, str(root / "txt_lists"))
        assert [_describe(x) for x in train] == c["train"], (c["fusion_mode"], c["mode"])
        assert [_describe(x) for x in val] == c["val"], (c["fusion_mode"], c["mode"])
        assert _describe(test) == c["test"]
        if c["mode"] == "deepall":
            assert len(train) == 1 and len(train[0].dataset if c["limit_source"] is None else train[0].dataset.dataset) > 40
        for tr, va in zip(c["train"], c["val"]):            # the property the old prefix split violated
            assert not set(tr["names"]) & set(va["names"])


def test_pil_resample_restatement_is_pil():
    """oracle.image_ref.pil_resize_restated (the algorithm the kernel implements) == Image.resize(BILINEAR), byte for byte."""
    from PIL import Image
    rs = np.random.RandomState(0)
    for h, w, oh, ow in [(227, 227, 222, 222), (96, 96, 222, 222), (61, 83, 40, 222), (300, 451, 64, 64), (50, 84, 50, 40), (205, 190, 222, 190)]:
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(I.pil_resize_restated(img, oh, ow), ref), (h, w, oh, ow)


def test_image_plan_tables_match_restatement():
    """ccst_image_plan (host function of the C ABI, no GPU): tap windows and 22-bit coefficients of both axes equal the
    restated libImaging tables; sizing call and too-small buffer behave as documented."""
    import ctypes
    from ccst_amd import _lib, data
    sizes = [(227, 227), (96, 120), (500, 333), (222, 222)]
    params = [(10, 20, 200, 190, 1), (0, 0, 96, 120, 0), (3, 5, 480, 300, 0), (0, 0, 222, 222, 0)]
    xf, tables = data.plan_transform(sizes, params, 222, 64)
    t = tables.numpy()
    off = 0
    for k, ((H, W), (i, j, h, w, flip)) in enumerate(zip(sizes, params)):
        x = xf[k]
        assert (x.src_off, x.src_w, x.crop_i, x.crop_j, x.crop_h, x.crop_w, x.flip) == (off, W, i, j, h, w, flip)
        off += H * W * 3
        kx, bx, cx = I._axis_tables(w, 64)
        ky, by, cy = I._axis_tables(h, 222)
        assert (x.kx, x.ky) == (kx, ky)
        assert np.array_equal(t[x.bounds_x:x.bounds_x + 2 * 64].reshape(64, 2), bx)
        assert np.array_equal(t[x.coefs_x:x.coefs_x + 64 * kx].reshape(64, kx), cx)
        assert np.array_equal(t[x.bounds_y:x.bounds_y + 2 * 222].reshape(222, 2), by)
        assert np.array_equal(t[x.coefs_y:x.coefs_y + 222 * ky].reshape(222, ky), cy)
    assert int(xf[3].kx) == 9 and int(xf[3].ky) == 3                      # 222 -> 64 is a 3.47x reduction; 222 -> 222 is the identity
    lib = _lib.load()
    small = torch.empty(10, dtype=torch.int32)
    assert lib.ccst_image_plan(len(sizes), ctypes.byref(xf), 222, 64, ctypes.c_void_p(small.data_ptr()), 10) == -2
    with pytest.raises(ValueError):
        data.plan_transform([(10, 10)], [(0, 0, 11, 10, 0)], 8, 8)


def test_crop_params_restate_torchvision_draw_order():
    """random_resized_crop_params: product copy == oracle copy (same torch RNG stream), rectangle inside the image,
    area within scale, and the 10-failures centre-crop fallback."""
    from ccst_amd import data
    for seed in range(20):
        torch.manual_seed(seed)
        a = data.random_resized_crop_params(227, 227, (0.8, 1.0))
        torch.manual_seed(seed)
        b = I.random_resized_crop_params(227, 227, (0.8, 1.0))
        assert a == b
        i, j, h, w = a
        assert 0 <= i and i + h <= 227 and 0 <= j and j + w <= 227 and 0.78 <= h * w / 227.0 ** 2 <= 1.01
    torch.manual_seed(0)
    i, j, h, w = data.random_resized_crop_params(10, 100, (1.0, 1.0))       # 10:1 image: every draw fails -> fallback, ratio clamp 4/3
    assert (h, w) == (10, 13) and i == 0 and j == (100 - 13) // 2


def test_adain_loader_shards_are_a_partition(tmp_path):
    """Under torchrun the AdaIN content list is sharded by entry: ranks' shards are disjoint, complete and
    independent of each rank's RNG state."""
    from ccst_amd import data
    lst = tmp_path / "pacs"
    lst.mkdir()
    rows = ["/x/PACS/kfold/photo/dog/p%03d.jpg 0" % i for i in range(23)]
    (lst / "photo_train.txt").write_text("\n".join(rows) + "\n")
    args = types.SimpleNamespace(dataset="pacs", target="photo", batch=4, image_size=64, synthetic=0)
    seen = []
    for rank in range(3):
        torch.manual_seed(100 + rank)                         # deliberately different per rank
        ld = data.get_train_dataloader(args, str(tmp_path), rank=rank, world=3)
        assert not isinstance(ld.loader.sampler, torch.utils.data.RandomSampler)
        seen.append(list(ld.dataset.names))
    flat = [n for s in seen for n in s]
    assert len(flat) == len(set(flat)) == 23 and set(flat) == set(r.split(" ")[0] for r in rows)
    one = data.get_train_dataloader(args, str(tmp_path))
    assert isinstance(one.loader.sampler, torch.utils.data.RandomSampler) and len(one.dataset) == 23     # the reference's shuffle=True


@pytest.mark.timeout(60)
def test_image_writer_pool_raises_when_an_encoder_dies(tmp_path):
    """A worker process killed under a task never completes that task's AsyncResult: drain() must notice the changed worker set and
    raise, not block for ever; an intact pool writes every file."""
    import os
    import signal
    import time
    import numpy as np
    from ccst_amd import data
    pool = data.ImageWriterPool(workers=2)
    assert pool.kind == "processes"               # (no GPU was touched in this process)
    img = np.zeros((2, 8, 8, 3), dtype=np.uint8)
    pool.submit(img, [str(tmp_path / "a.png"), str(tmp_path / "b.png")])
    pool.drain()
    assert (tmp_path / "a.png").exists() and (tmp_path / "b.png").exists()
    pool.futures += [pool.pool.apply_async(time.sleep, (30,)) for _ in range(2)]      # both workers busy ...
    time.sleep(0.5)
    for pid in pool.pids:
        os.kill(pid, signal.SIGKILL)                                                   # ... and killed under their tasks
    t0 = time.time()
    with pytest.raises(RuntimeError, match="encoder process died"):
        pool.drain()
    assert time.time() - t0 < 20
    pool.pool.terminate()
