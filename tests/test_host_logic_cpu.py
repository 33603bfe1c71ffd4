"""CPU: host-side logic of the drop-in surface (no kernels run)."""
import types

import numpy as np
import pytest
import torch

from oracle import adain_ref as A
from oracle import resnet_ref as R


def test_net_state_dict_keys_match_reference_layout():
    from ccst_amd import net
    assert len(net.vgg) == 53 and len(net.decoder) == 29
    vk = ["%d.%s" % (i, s) for i, _, _, _ in A.conv_keys(A.VGG_TABLE) for s in ("weight", "bias")]
    dk = ["%d.%s" % (i, s) for i, _, _, _ in A.conv_keys(A.DECODER_TABLE) for s in ("weight", "bias")]
    assert list(net.vgg.state_dict().keys()) == vk and list(net.decoder.state_dict().keys()) == dk
    net.vgg.load_state_dict(A.he_weights(A.VGG_TABLE, 1))
    sliced = net.vgg[:31]
    assert isinstance(sliced, net.Sequential) and len(sliced) == 31
    assert isinstance(torch.nn.Sequential(*list(net.vgg.children())[:31])[2], torch.nn.Conv2d)


def test_no_cpu_fallback():
    from ccst_amd import function, net
    with pytest.raises(RuntimeError):
        net.vgg[:4](torch.zeros(1, 3, 8, 8))
    with pytest.raises(RuntimeError):
        function.calc_mean_std(torch.zeros(1, 4, 4, 4))
    with pytest.raises(AssertionError):
        function.calc_mean_std(torch.zeros(4, 4, 4))


def test_registries():
    from ccst_amd.nets import model_factory, models
    args = types.SimpleNamespace(dg_method="no_DG")
    assert set(models.nets_map) == {'resnet18', 'resnet18IN', 'resnet50', 'DigitModel', 'densenet'}
    assert set(model_factory.nets_map) == {'caffenet', 'alexnet', 'resnet18', 'resnet50', 'lenet'}
    with pytest.raises(ValueError, match="Name of network unknown foo"):
        models.get_network("foo")
    with pytest.raises(ValueError):
        model_factory.get_network("foo")
    m = models.get_network("resnet18")(args, pretrained=False, classes=2)
    assert list(m.state_dict().keys()) == list(R.resnet18(2).state_dict().keys())
    for name in ("conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4", "avgpool", "class_classifier"):
        assert hasattr(m, name)
    m2 = model_factory.get_network("resnet50")(pretrained=False, classes=7)
    assert m2.class_classifier.out_features == 7
    with pytest.raises(NotImplementedError):
        models.get_network("densenet")(args)
    with pytest.raises(NotImplementedError):
        models.get_network("resnet50")(types.SimpleNamespace(dg_method="Jigsaw"), classes=7)


def test_style_stat_finalise_and_file_format(tmp_path):
    from ccst_amd import style
    rs = np.random.RandomState(0)
    s = torch.from_numpy(rs.uniform(100, 200, (1, 512, 1, 1)).astype(np.float32))
    q = s * s / 1000 + torch.from_numpy(rs.uniform(50, 60, (1, 512, 1, 1)).astype(np.float32))
    m, sd = style.finalise_style_stats(s, q, 1000)
    rm, rsd = A.finalise_stats(s, q, 1000)
    assert torch.equal(m, rm) and torch.equal(sd, rsd)
    p = str(tmp_path / "x_mean_std.npy")
    style.save_style_stat(p, m, sd)
    a = np.load(p)
    assert a.shape == (2, 1, 512, 1, 1) and a.dtype == np.float32        # CCST_OverallStyleTransfer.py:140-144
    lm, ls = style.load_style_stat(p, "cpu")
    assert torch.equal(lm, m) and torch.equal(ls, sd)


def test_output_naming_rule():
    from ccst_amd import data
    f = "/disk1/cjm/research/CCST/data/PACS/kfold/art_painting/dog/pic_001.jpg"
    assert data.stylised_name(f, "art_painting", "cartoon", "all_style_transferred_Overall") == \
        "/disk1/cjm/research/CCST/data/PACS/all_style_transferred_Overall/art_painting/cartoon/dog/pic_001_cartoon.jpg"


def test_flat_params_arena_views():
    from ccst_amd import fed
    from ccst_amd.nets import resnet
    m = resnet.ResNet(resnet.BasicBlock, [1, 1, 1, 1], classes=3)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    a = fed.FlatParams(m)
    assert a.n_param >= sum(p.numel() for p in m.parameters()) and a.n_total > a.n_param
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    lo, hi = a.flat.data_ptr(), a.flat.data_ptr() + 4 * a.n_total
    for p in m.parameters():
        assert lo <= p.data_ptr() < hi and p.data_ptr() % 16 == 0
        assert a.grad.data_ptr() <= p.grad.data_ptr() < a.grad.data_ptr() + 4 * a.n_param
    nb = [b for n, b in m.named_buffers() if n.endswith("num_batches_tracked")]
    assert all(not (lo <= b.data_ptr() < hi) for b in nb)
    a.flat[:a.n_param].fill_(2.0)
    assert float(m.conv1.weight[0, 0, 0, 0]) == 2.0
    assert fed.FlatParams.of(m) is a


def test_bench_conv_kernel_name_mirror():
    from ccst_amd import ops
    assert ops._conv_kernel_name(256, False, 98304, 256, 9) == "conv_igemm_kernel<2,2,2>"            # large map, long K
    assert ops._conv_kernel_name(256, False, 200704, 64, 1) == "conv_igemm_kernel<2,2,1>"            # large map, short K
    assert ops._conv_kernel_name(64, True, 1572864, 64, 9) == "conv_igemm_kernel<2,2,1,pool>"
    assert ops._conv_kernel_name(64, False, 200704, 256, 1) == "conv_igemm_kernel<2,2,1,mt1,ck32>"   # Cout = 64
    assert ops._conv_kernel_name(1024, False, 12544, 256, 1) == "conv_igemm_kernel<2,2,1,mt1,ck32>"  # maps up to 28x28 at B=64
    assert ops._conv_kernel_name(256, False, 12544, 48, 9) == "conv_igemm_kernel<2,2,1,mt1>"         # no 32-channel k-step
    assert ops._conv_kernel_name(64, False, 1572864, 16, 3) == "conv_igemm_kernel<2,2,1>"            # AdaIN stem


def test_bench_traffic_keys_find_the_committed_profile():
    """bench.py copies the dominant kernel's HBM bytes from profiles/traffic.json: the bucket name of ops.TIMING must map onto the
    rocprofv3 names of the committed profile (a silent miss would print `traffic: null` in the driver's bench line)."""
    import json
    import os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "traffic.json")) as fh:
        tj = json.load(fh)
    assert tj["_source"]["build_stamp"] and tj["_source"]["profile"].startswith("profiles/")
    # the buckets of the round-5 path: F(4,3) along x, <POOL, ZP, HALF, NT> in the kernel trace
    keys = bench.traffic_keys("conv3x3_f43_kernel<nopool>", tj)
    assert keys and all(k.startswith("void conv3x3_f43_kernel<false, ") and k.split(",")[2].strip() == "false" for k in keys)
    pooled = bench.traffic_keys("conv3x3_f43_kernel<pool>", tj)
    assert pooled and all(k.startswith("void conv3x3_f43_kernel<true, ") for k in pooled) and not set(pooled) & set(keys)
    half = bench.traffic_keys("conv3x3_f43_kernel<nopool,half>", tj)       # (both store policies of the 64-channel tile in one bucket)
    assert half and all(k.split(",")[2].strip() == "true" for k in half) and not set(half) & set(keys)
    for k in keys + pooled + half:
        assert tj[k]["total_bytes_per_launch"] > 0 and tj[k]["launches"] > 0
    # ... and of the kernels behind the switches (names only: the committed profile does not run them)
    assert bench.traffic_keys("conv3x3_halo_split_kernel<nopool>", {"void conv3x3_halo_kernel<4, 1, 2, false, false, true>": 1}) == \
        ["void conv3x3_halo_kernel<4, 1, 2, false, false, true>"]
    assert bench.traffic_keys("conv_igemm_kernel<2,2,1>", tj) == ["void conv_igemm_kernel<2, 2, 1, false, 2, 16>"]


def test_flat_params_follow_the_model():
    """an arena whose tensors the model no longer holds (deepcopy, .to() round trip, assignment) is rebuilt, never
    silently updated in place of the model; kernels that need the GPU refuse a CPU arena."""
    import copy
    import pytest
    from ccst_amd import fed
    from ccst_amd.nets import resnet
    m = resnet.ResNet(resnet.BasicBlock, [1, 1, 1, 1], classes=3)
    a = fed.FlatParams.of(m)
    m2 = copy.deepcopy(m)                               # fed_run.py:577 builds the clients like this
    assert m2.__dict__.get("_ccst_arena") is None
    a2 = fed.FlatParams.of(m2)
    assert a2 is not a and a2.valid(full=True) and a.valid(full=True)
    a2.flat.fill_(3.0)
    assert float(m2.conv1.weight[0, 0, 0, 0]) == 3.0 and float(m.conv1.weight[0, 0, 0, 0]) != 3.0
    # what model.to('cpu') -> .to(device) does to every parameter: a fresh tensor
    opt = fed.SGD(m, lr=0.1)
    for p in m.parameters():
        p.data = p.data.clone()
    for b in m.buffers():
        b.data = b.data.clone()
    assert not a.valid()
    a3 = opt.arena
    assert a3 is not a and a3.valid(full=True) and fed.FlatParams.of(m) is a3 and opt.param_groups[0]["params"] is a3.params
    a3.flat[:a3.n_param].fill_(5.0)
    assert float(m.class_classifier.bias[0]) == 5.0
    with pytest.raises(RuntimeError, match="GPU"):
        opt.step()
    import types
    with pytest.raises(RuntimeError, match="GPU"):
        fed.communication(types.SimpleNamespace(mode="fedavg"), m, [m2], [1.0])


def test_pretrained_weights_must_exist(tmp_path, monkeypatch):
    import types
    import pytest
    from ccst_amd.nets import models
    args = types.SimpleNamespace(dg_method="no_DG")
    monkeypatch.delenv("CCST_PRETRAINED_DIR", raising=False)
    with pytest.raises(FileNotFoundError):
        models.get_network("resnet18")(args, classes=7)                  # the reference's default is pretrained=True
    monkeypatch.setenv("CCST_PRETRAINED_DIR", str(tmp_path))
    with pytest.raises(FileNotFoundError):
        models.get_network("resnet18")(args, pretrained=True, classes=7)
    ref = models.get_network("resnet18")(args, pretrained=False, classes=1000)
    sd = {k.replace("class_classifier", "fc"): v + 1.0 if v.dtype == torch.float32 else v for k, v in ref.state_dict().items()}
    torch.save(sd, str(tmp_path / "resnet18.pth"))
    m = models.get_network("resnet18")(args, pretrained=True, classes=7)     # strict=False: the 1000-way 'fc' head is ignored
    assert torch.equal(m.conv1.weight, ref.conv1.weight + 1.0) and m.class_classifier.weight.shape == (7, 512)


def test_resume_restores_every_clients_own_model_under_fedbn():
    """fed_run.py:626-640: --resume under --mode fedbn loads 'model_{k}' into client k (the server average used to
    overwrite the local BN state); other modes load the server model everywhere."""
    import importlib.util
    import os
    from ccst_amd.nets import resnet
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ccst_fed_run", os.path.join(root, "federated", "fed_run.py"))
    fr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fr)

    def mk(seed):
        torch.manual_seed(seed)
        m = resnet.ResNet(resnet.BasicBlock, [1, 1, 1, 1], classes=3)
        with torch.no_grad():
            m.bn1.running_mean.normal_()
        return m
    ck = {"server_model": mk(1).state_dict(), "a_iter": 4, "model_0": mk(2).state_dict(), "model_1": mk(3).state_dict()}
    server, models = mk(9), {0: mk(10), 1: mk(11)}
    assert fr.restore(ck, server, models, fedbn=True) == 5
    assert torch.equal(server.bn1.running_mean, ck["server_model"]["bn1.running_mean"])
    for ci in (0, 1):
        assert torch.equal(models[ci].bn1.running_mean, ck["model_%d" % ci]["bn1.running_mean"])
        assert torch.equal(models[ci].conv1.weight, ck["model_%d" % ci]["conv1.weight"])
    only1 = {1: mk(12)}                                     # torchrun: rank 1 holds only client 1
    fr.restore(ck, server, only1, fedbn=True)
    assert torch.equal(only1[1].bn1.running_mean, ck["model_1"]["bn1.running_mean"])
    fr.restore(ck, server, models, fedbn=False)
    assert torch.equal(models[1].bn1.running_mean, ck["server_model"]["bn1.running_mean"])


def test_conv_kernel_selection_does_not_depend_on_the_batch_size():
    """ops.f43_wanted decides from the tiles of ONE image (ccst_conv3x3_f43_workgroups with N = 1): a sample keeps its kernels -- and its
    rounding -- whatever the number of its batch-mates (a rule on the whole launch's workgroups once ran the Cout = 64 layers of a
    96 x 160 image on F(4,3) inside a batch of three and on the direct kernel alone: 2e-5 of the output range apart)."""
    import torch
    from ccst_amd import _lib, ops

    class PC(object):          # the fields the rule reads
        def __init__(self, cin, cout):
            self.cin, self.cout = cin, cout

        def can_split(self):
            return True

        def lazy3x3(self):
            return True

    dev = torch.device("cpu")
    lib = _lib.load()
    assert int(lib.ccst_conv3x3_f43_workgroups(1, 512, 512, 64)) == 64 * 16           # 8 x 32-pixel x 64-channel tiles
    assert int(lib.ccst_conv3x3_f43_workgroups(6, 64, 64, 256)) == 6 * 8 * 2 * 2      # ... x 128 channels
    for (h, w, cin, cout) in ((512, 512, 64, 64), (64, 64, 512, 256), (96, 160, 64, 64), (48, 80, 64, 128), (24, 40, 128, 256), (8, 8, 512, 512)):
        picks = {ops.f43_wanted(PC(cin, cout), n, h, w, dev) for n in (1, 2, 3, 6, 32)}
        assert len(picks) == 1, (h, w, cin, cout, picks)
    assert ops.f43_wanted(PC(64, 64), 1, 512, 512, dev) and not ops.f43_wanted(PC(512, 512), 32, 8, 8, dev)


def test_timing_experiment_patches_apply_to_the_shipped_sources(tmp_path):
    """The timing-experiment branches live outside the product sources as tools/variants/<file>.patch (tools/build_variant.sh applies them to a
    scratch copy): every patch must still apply to the file it is named after."""
    import glob
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    patches = sorted(glob.glob(os.path.join(root, "tools", "variants", "*.patch")))
    assert patches
    for pf in patches:
        scratch = tmp_path / os.path.basename(pf)
        shutil.copytree(os.path.join(root, "ccst_amd", "csrc"), scratch / "ccst_amd" / "csrc", ignore=shutil.ignore_patterns("*.o", "*.so"))
        with open(pf) as fh:
            r = subprocess.run(["patch", "-s", "-p1", "--dry-run", "-d", str(scratch)], stdin=fh, capture_output=True, text=True)
        assert r.returncode == 0, (os.path.basename(pf), r.stdout[-400:])


def test_cli_flags_match_the_reference_scripts():
    """every LIVE flag of the reference's four scripts exists here with the same option strings, default and action
    (tests/golden/cli_flags.json: names and literal defaults AST-read from the reference by tools/make_golden.py).  Flags whose
    feature is outside the hot path parse and refuse when used (federated/fed_run.py main()).  Documented deviations only."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "cli_flags.json")) as f:
        golden = json.load(f)
    here = {"mean_std_computation_effcientMem.py": "style_transfer/AdaIN/mean_std_computation_effcientMem.py",
            "CCST_OverallStyleTransfer.py": "style_transfer/AdaIN/CCST_OverallStyleTransfer.py",
            "CCST_SingleStyleTransfer.py": "style_transfer/AdaIN/CCST_SingleStyleTransfer.py", "fed_run.py": "federated/fed_run.py"}
    # the reference's --network default 'resnet' is not a key of its own nets_map (nets/models.py:114-123): the drop-in defaults to resnet50
    # ... and the two stage-2 scripts declare --dataset without a default and then call args.dataset.lower() (CCST_OverallStyleTransfer.py:51,97):
    # here it defaults to 'pacs' (stage 1's default) instead of failing with AttributeError
    deviations = {("fed_run.py", "--network", "default"), ("CCST_OverallStyleTransfer.py", "--dataset", "default"),
                  ("CCST_SingleStyleTransfer.py", "--dataset", "default")}

    class _Stop(Exception):
        pass

    def flags_of(path):
        """Run the script up to its parse_args() call and read the parser it built (the AdaIN scripts share _common.base_parser)."""
        import argparse
        import runpy
        import sys
        box = {}

        def grab(self, *a, **k):
            box["parser"] = self
            raise _Stop()
        real, argv, syspath = argparse.ArgumentParser.parse_args, sys.argv, list(sys.path)
        argparse.ArgumentParser.parse_args = grab
        sys.argv = [path]
        sys.path.insert(0, os.path.dirname(path))
        try:
            runpy.run_path(path, run_name="__main__")
        except _Stop:
            pass
        finally:
            argparse.ArgumentParser.parse_args, sys.argv, sys.path[:] = real, argv, syspath
        out = {}
        for a in box["parser"]._actions:
            if a.option_strings and a.option_strings[0] != "-h":
                action = {"_StoreTrueAction": "store_true", "_StoreFalseAction": "store_false"}.get(type(a).__name__)
                kw = {"default": None if action == "store_true" else a.default}
                if action:
                    kw["action"] = action
                out[a.option_strings[0]] = (list(a.option_strings), kw)
        return out
    for script, rel in here.items():
        mine = flags_of(os.path.join(root, rel))
        for ref in golden[script]:
            name = ref["names"][0]
            assert name in mine, "%s: flag %s of the reference is missing" % (script, name)
            names, kw = mine[name]
            assert names == ref["names"], (script, name, names)
            for field in ("default", "action"):
                if (script, name, field) in deviations:
                    continue
                assert kw.get(field) == ref.get(field), (script, name, field, kw.get(field), ref.get(field))
