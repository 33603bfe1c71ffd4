#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF on seeded inputs.

Runs only where /root/reference exists (the build container).  It
  * imports the reference's style_transfer/AdaIN/{net,function}.py as they lie,
  * AST-extracts (and exec's, unmodified) the script-level functions
    ``style_transfer`` (CCST_OverallStyleTransfer.py:32), ``calc_sum``
    (mean_std_computation_effcientMem.py:103 and CCST_SingleStyleTransfer.py:55)
    and ``train`` / ``test`` / ``communication`` (federated/fed_run.py:31,214,385; ``train``/``test`` with a stub logger)
    because the scripts themselves execute argparse/model loading on import and
    need torchvision,
  * imports nets/resnet.py behind a stub ``torchvision.models.resnet`` that
    supplies the oracle's restated BasicBlock/Bottleneck (torchvision is not
    installed and not part of /root/reference),
and stores inputs' seeds + the reference's outputs.  Only data is written:
no reference source text is copied anywhere.

Usage:  python tools/make_golden.py            (rewrites tests/golden/)
"""
import ast
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import numpy as np
import torch
from torch import nn

from oracle import adain_ref as A
from oracle import resnet_ref as R

torch.manual_seed(0)
torch.set_num_threads(8)
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def extract_functions(path, names, namespace):
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), namespace)
    return namespace


ONLY = set(sys.argv[1:])      # e.g. `python tools/make_golden.py communication_fedbn` rewrites just that fixture


def wanted(name):
    return not ONLY or name in ONLY


def save(name, **arrs):
    if not wanted(name):
        print("skipped", name)
        return
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


# --------------------------------------------------------------------------
# AdaIN path
# --------------------------------------------------------------------------
sys.path.insert(0, os.path.join(REF, "style_transfer", "AdaIN"))
import function as ref_function      # noqa: E402  (reference module)
import net as ref_net                # noqa: E402  (reference module)

VGG_W = A.he_weights(A.VGG_TABLE, seed=1234)
DEC_W = A.he_weights(A.DECODER_TABLE, seed=4321)
ref_vgg = ref_net.vgg
ref_dec = ref_net.decoder
ref_vgg.eval()
ref_dec.eval()
ref_vgg.load_state_dict(VGG_W)       # all 17 convs of the 53-layer net
ref_dec.load_state_dict(DEC_W)
ref_vgg31 = nn.Sequential(*list(ref_vgg.children())[:31])   # CCST_OverallStyleTransfer.py:124

ns = {"torch": torch, "device": torch.device("cpu"),
      "adaIN_StyleStat_ContentFeat": ref_function.adaIN_StyleStat_ContentFeat}
extract_functions(os.path.join(REF, "style_transfer/AdaIN/CCST_OverallStyleTransfer.py"),
                  {"style_transfer"}, ns)
ref_style_transfer = ns["style_transfer"]
ns1 = {"torch": torch}
extract_functions(os.path.join(REF, "style_transfer/AdaIN/mean_std_computation_effcientMem.py"),
                  {"calc_sum"}, ns1)
ns2 = {"torch": torch}
extract_functions(os.path.join(REF, "style_transfer/AdaIN/CCST_SingleStyleTransfer.py"),
                  {"calc_sum"}, ns2)

with torch.no_grad():
    # 1. calc_mean_std on [2,8,5,7]
    rs = np.random.RandomState(11)
    feat = torch.from_numpy(rs.normal(0.3, 1.2, (2, 8, 5, 7)).astype(np.float32))
    m, s = ref_function.calc_mean_std(feat)
    save("calc_mean_std", seed=11, mean=m, std=s)

    # 2. adaIN_StyleStat_ContentFeat / adaptive_instance_normalization on [2,512,8,8]
    rs = np.random.RandomState(12)
    cf = torch.from_numpy(np.abs(rs.normal(0.0, 0.6, (2, 512, 8, 8))).astype(np.float32))
    sf = torch.from_numpy(np.abs(rs.normal(0.2, 0.9, (2, 512, 6, 10))).astype(np.float32))
    stat = A.synth_style_stat(512, seed=7)
    save("adain_feat", seed=12,
         out_stat=ref_function.adaIN_StyleStat_ContentFeat(cf, stat),
         out_feat=ref_function.adaptive_instance_normalization(cf, sf))

    # 3. calc_sum over 3 batches + finalise (mean_std...py:117-137)
    tot_s, tot_q, tot_n = 0, 0, 0
    per = []
    for b in range(3):
        data = A.synth_content(2, 32, 48, seed=100 + b)
        f = ref_vgg31(data)
        s1, q1, n1 = ns1["calc_sum"](f)
        s2, q2, n2 = ns2["calc_sum"](f)
        assert torch.equal(s1, s2) and torch.equal(q1, q2) and n1 == n2
        per.append((s1, q1, n1))
        tot_s = tot_s + s1
        tot_q = tot_q + q1
        tot_n += n1
    mean = tot_s / float(tot_n)
    var = tot_q / float(tot_n) - mean ** 2
    std = torch.sqrt(var + 1e-5)
    save("overall_stats", seeds=[100, 101, 102], sum0=per[0][0], sq0=per[0][1], n0=per[0][2],
         tot_sum=tot_s, tot_sq=tot_q, tot_n=tot_n, mean=mean, std=std)

    # 4. encoder relu4_1 and style_transfer on [2,3,64,64]
    content = A.synth_content(2, 64, 64, seed=1)
    stat = A.synth_style_stat(512, seed=7)
    enc = ref_vgg31(content)
    out = ref_style_transfer(ref_vgg31, ref_dec, content, stat, 1.0)
    out_a = ref_style_transfer(ref_vgg31, ref_dec, content, stat, 0.5)
    save("style_transfer_64", seed=1, relu4_1=enc, out=out, out_alpha05=out_a)

    # 4b. the interpolation branch (CCST_OverallStyleTransfer.py:36-42): one content image three times, three styles, weights
    one = A.synth_content(1, 64, 64, seed=5)
    content3 = one.repeat(3, 1, 1, 1)
    stats3 = [A.synth_style_stat(512, seed=s) for s in (7, 8, 9)]
    stat3 = [torch.cat([s[0] for s in stats3]), torch.cat([s[1] for s in stats3])]
    wts = [0.5, 0.3, 0.2]
    out_i = ref_style_transfer(ref_vgg31, ref_dec, content3, stat3, 1.0, wts)
    out_ia = ref_style_transfer(ref_vgg31, ref_dec, content3, stat3, 0.6, wts)
    assert tuple(out_i.shape) == (1, 3, 64, 64)
    save("style_transfer_interp", seed=5, style_seeds=[7, 8, 9], weights=np.array(wts), out=out_i, out_alpha06=out_ia)

    # 5. odd size [1,3,222,222] -> [1,3,224,224]; non-square [1,3,50,84]
    content = A.synth_content(1, 222, 222, seed=2)
    out = ref_style_transfer(ref_vgg31, ref_dec, content, stat, 1.0)
    assert tuple(out.shape) == (1, 3, 224, 224)
    content2 = A.synth_content(1, 50, 84, seed=3)
    out2 = ref_style_transfer(ref_vgg31, ref_dec, content2, stat, 1.0)
    save("style_transfer_odd", seed=2, out_sub4=out[:, :, ::4, ::4].contiguous(),
         out_shape=list(out.shape), chan_sum=out.sum(dim=(0, 2, 3)), chan_abs=out.abs().sum(dim=(0, 2, 3)),
         seed2=3, out2=out2, out2_shape=list(out2.shape))

# --------------------------------------------------------------------------
# ResNet / FedAvg path
# --------------------------------------------------------------------------
tv = types.ModuleType("torchvision")
tvm = types.ModuleType("torchvision.models")
tvr = types.ModuleType("torchvision.models.resnet")
tvr.BasicBlock, tvr.Bottleneck, tvr.model_urls = R.BasicBlock, R.Bottleneck, {}
tv.models, tvm.resnet = tvm, tvr
sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.resnet": tvr})
sys.path.insert(0, REF)
from nets import resnet as ref_resnet   # noqa: E402  (reference module: ResNet class, _make_layer, forward)


def _run_case(model, x, y, lr):
    model.eval()
    with torch.no_grad():
        logit_eval = model(x)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=lr)
    opt.zero_grad()
    logit_train = model(x)
    loss = nn.CrossEntropyLoss()(logit_train, y)
    loss.backward()
    g = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    opt.step()
    model.eval()
    with torch.no_grad():
        logit_after = model(x)
    return logit_eval, logit_train, loss, g, logit_after, model.state_dict()


def resnet_case(name, block, layers, classes, nb, seed, lr=0.001, residual_gamma=0.25, fc_gain=8.0):
    if not wanted(name):
        return
    model = ref_resnet.ResNet(block, layers, classes=classes)
    ours = R.ResNet(block, layers, classes=classes)
    # conditioned like a trained net (small closing-BN gammas), see oracle.resnet_ref.seeded_state_dict
    sd = R.seeded_state_dict(ours, seed, residual_gamma=residual_gamma, fc_gain=fc_gain)
    model.load_state_dict(sd)
    x, y = R.synth_batch(nb, 222, classes, seed=seed + 1)
    model.eval()
    with torch.no_grad():
        logit_eval = model(x)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=lr)         # fed_run.py:657 form, README.md:99 lr
    opt.zero_grad()
    logit_train = model(x)
    loss = nn.CrossEntropyLoss()(logit_train, y)               # fed_run.py:554
    loss.backward()
    g = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    opt.step()
    model.eval()
    with torch.no_grad():
        logit_after = model(x)
    sd_after = model.state_dict()
    probe = {}
    for k in ["conv1.weight", "bn1.weight", "bn1.bias", "class_classifier.weight", "class_classifier.bias",
              "layer1.0.conv1.weight", "layer2.0.downsample.0.weight", "layer4.%d.conv2.weight" % (layers[3] - 1),
              "layer3.0.bn2.weight"]:
        probe["grad_sum/" + k] = g[k].sum()
        probe["grad_abs/" + k] = g[k].abs().sum()
        probe["grad_head/" + k] = g[k].flatten()[:16]
    for k in ["bn1.running_mean", "bn1.running_var", "layer4.0.bn1.running_mean", "layer2.0.downsample.1.running_var",
              "bn1.num_batches_tracked"]:
        probe["state/" + k] = sd_after[k]
    # conditioning of the problem: the SAME reference model in float64.  |fp32 - fp64| per probe is the
    # noise floor any fp32 implementation sits on (tiny batch + train-mode BN + ReLU/max-pool masks).
    m64 = ref_resnet.ResNet(block, layers, classes=classes)
    m64.load_state_dict(sd)
    m64 = m64.double()
    le64, lt64, loss64, g64, la64, _ = _run_case(m64, x.double(), y, lr)
    probe["noise/logit_eval"] = (logit_eval.double() - le64).abs().max()
    probe["noise/logit_train"] = (logit_train.double() - lt64).abs().max()
    probe["noise/logit_after"] = (logit_after.double() - la64).abs().max()
    probe["noise/loss"] = (loss.double() - loss64).abs()
    for k in list(probe):
        if k.startswith("grad_head/"):
            n_ = k[10:]
            probe["noise/grad_head/" + n_] = (g[n_].flatten()[:16].double() - g64[n_].flatten()[:16]).abs().max()
            probe["noise/grad_abs/" + n_] = (g[n_].abs().sum().double() - g64[n_].abs().sum()).abs()
            # the 16-element head is a small sample of the tensor's error distribution: also keep the reference's own
            # worst fp32 error over the WHOLE gradient tensor (same conditioning, far steadier estimate)
            probe["noise_full/grad/" + n_] = (g[n_].double() - g64[n_]).abs().max()
    save(name, seed=seed, classes=classes, nb=nb, lr=lr, residual_gamma=residual_gamma, fc_gain=fc_gain, logit_eval=logit_eval, logit_train=logit_train,
         loss=loss, logit_after=logit_after, **probe)


resnet_case("resnet18_step", R.BasicBlock, [2, 2, 2, 2], 2, 8, seed=50)
resnet_case("resnet50_step", R.Bottleneck, [3, 4, 6, 3], 7, 16, seed=60)

# communication() on 3 perturbed clients (fed_run.py:385-455, fedavg branch)
nsf = {"torch": torch, "nn": nn}
extract_functions(os.path.join(REF, "federated/fed_run.py"), {"communication"}, nsf)
args = types.SimpleNamespace(mode="fedavg")
server = ref_resnet.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3)
server.load_state_dict(R.seeded_state_dict(R.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3), 70))
import copy  # noqa: E402
clients = [copy.deepcopy(server) for _ in range(3)]
for ci, c in enumerate(clients):
    rs = np.random.RandomState(71 + ci)
    with torch.no_grad():
        for k, v in c.state_dict().items():
            if "num_batches_tracked" in k:
                v.fill_(5 + ci)
            else:
                v += torch.from_numpy(rs.normal(0, 0.02, tuple(v.shape)).astype(np.float32))
weights = [0.5, 0.3, 0.2]
server, clients = nsf["communication"](args, server, clients, weights)
keys = list(server.state_dict().keys())
ksum = np.array([float(server.state_dict()[k].double().sum()) for k in keys])
kabs = np.array([float(server.state_dict()[k].double().abs().sum()) for k in keys])
nbt_server = [int(server.state_dict()[k]) for k in keys if "num_batches_tracked" in k]
nbt_clients = [[int(c.state_dict()[k]) for k in keys if "num_batches_tracked" in k] for c in clients]
same = all(torch.equal(server.state_dict()[k], c.state_dict()[k]) for c in clients for k in keys
           if "num_batches_tracked" not in k)
save("communication", seed=70, weights=weights, keys=np.array(keys), key_sum=ksum, key_abs=kabs,
     conv1_head=server.state_dict()["conv1.weight"].flatten()[:32],
     fc_bias=server.state_dict()["class_classifier.bias"],
     nbt_server=nbt_server, nbt_clients=nbt_clients, clients_equal_server=same)

# communication() --mode fedbn (fed_run.py:388-399): same 3 perturbed clients, BN keys stay local
args = types.SimpleNamespace(mode="fedbn")
server = ref_resnet.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3)
server.load_state_dict(R.seeded_state_dict(R.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3), 70))
clients = [copy.deepcopy(server) for _ in range(3)]
for ci, c in enumerate(clients):
    rs = np.random.RandomState(71 + ci)
    with torch.no_grad():
        for k, v in c.state_dict().items():
            if "num_batches_tracked" in k:
                v.fill_(5 + ci)
            else:
                v += torch.from_numpy(rs.normal(0, 0.02, tuple(v.shape)).astype(np.float32))
server, clients = nsf["communication"](args, server, clients, weights)
keys = list(server.state_dict().keys())
fkeys = [k for k in keys if "num_batches_tracked" not in k]
save("communication_fedbn", seed=70, weights=weights, keys=np.array(fkeys),
     server_sum=np.array([float(server.state_dict()[k].double().sum()) for k in fkeys]),
     server_abs=np.array([float(server.state_dict()[k].double().abs().sum()) for k in fkeys]),
     client_sum=np.array([[float(c.state_dict()[k].double().sum()) for k in fkeys] for c in clients]),
     client_abs=np.array([[float(c.state_dict()[k].double().abs().sum()) for k in fkeys] for c in clients]),
     shared=np.array([all(torch.equal(server.state_dict()[k], c.state_dict()[k]) for c in clients) for k in fkeys]),
     bn1_weight_client1=clients[1].state_dict()["bn1.weight"].flatten()[:16],
     ds_bn_weight_client1=clients[1].state_dict()["layer2.0.downsample.1.weight"].flatten()[:16],
     nbt_server=[int(server.state_dict()[k]) for k in keys if "num_batches_tracked" in k],
     nbt_clients=[[int(c.state_dict()[k]) for k in keys if "num_batches_tracked" in k] for c in clients])
# train() / test() (fed_run.py:31-88, :214-259): the reference's own loops, AST-extracted, on a 3-batch seeded loader with a
# ragged last batch (4, 4, 3 images).  The reference's ResNet class carries the oracle's restated blocks (see above).
import math  # noqa: E402
from collections import OrderedDict  # noqa: E402


class _StubLogger(object):
    """utils/Logger.py needs tensorflow; train() only calls .log(it, n, losses, samples_right, total_samples)."""

    def __init__(self):
        self.rows = []

    def log(self, it, iters, losses, samples_right, total_samples):
        self.rows.append((it, iters, float(losses["train_loss"]), int(samples_right["class_acc"]), int(total_samples)))


nsl = {"torch": torch, "nn": nn, "copy": copy, "math": math, "OrderedDict": OrderedDict}
extract_functions(os.path.join(REF, "federated/fed_run.py"), {"train", "test"}, nsl)
largs = types.SimpleNamespace(dg_method="no_DG", network="resnet18", n_classes=3, IN_test=False, jig_weight=0.0)
lmodel = ref_resnet.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3)
lmodel.load_state_dict(R.seeded_state_dict(R.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3), 91))
train_loader = [R.synth_batch(n, 222, 3, seed=100 + i) for i, n in enumerate((4, 4, 3))]
test_loader = [R.synth_batch(n, 222, 3, seed=200 + i) for i, n in enumerate((4, 2))]
logger = _StubLogger()
opt = torch.optim.SGD(params=lmodel.parameters(), lr=0.01)                      # fed_run.py:657
loss_fun = nn.CrossEntropyLoss()                                                # fed_run.py:554
tr1 = nsl["train"](lmodel, train_loader, opt, loss_fun, 3, torch.device("cpu"), largs, 0, logger)
te1 = nsl["test"](lmodel, test_loader, loss_fun, torch.device("cpu"), largs)
tr2 = nsl["train"](lmodel, train_loader, opt, loss_fun, 3, torch.device("cpu"), largs, 1, logger)
te2 = nsl["test"](lmodel, train_loader, loss_fun, torch.device("cpu"), largs)
lsd = lmodel.state_dict()
save("fed_loop", seed=91, train_seeds=[100, 101, 102], train_sizes=[4, 4, 3], test_seeds=[200, 201], test_sizes=[4, 2], lr=0.01,
     train1=np.array(tr1, dtype=np.float64), test1=np.array(te1, dtype=np.float64),
     train2=np.array(tr2, dtype=np.float64), test2=np.array(te2, dtype=np.float64),
     log_loss=np.array([r[2] for r in logger.rows], dtype=np.float64), log_right=np.array([r[3] for r in logger.rows]),
     log_total=np.array([r[4] for r in logger.rows]), log_it=np.array([r[0] for r in logger.rows]),
     log_iters=np.array([r[1] for r in logger.rows]),
     conv1_head=lsd["conv1.weight"].flatten()[:64], fc_weight=lsd["class_classifier.weight"], fc_bias=lsd["class_classifier.bias"],
     bn1_running_mean=lsd["bn1.running_mean"], bn1_running_var=lsd["bn1.running_var"],
     l4_bn2_weight=lsd["layer4.0.bn2.weight"], nbt=int(lsd["bn1.num_batches_tracked"]),
     key_sum=np.array([float(v.double().sum()) for v in lsd.values()]),
     key_abs=np.array([float(v.double().abs().sum()) for v in lsd.values()]))
# --------------------------------------------------------------------------
# Data plane lists (data/data_helper.py:46-159, data/ImageLoader.py:13-47): the reference's own get_train_dataloader /
# get_test_dataloader / creat_train_loader_list / get_random_subset, AST-extracted, with the image datasets replaced by
# name-capturing stubs (the list logic never touches a pixel) and the stray pdb.set_trace() at :76 neutralised.
# --------------------------------------------------------------------------
if wanted("data_lists"):
    import json
    import random
    import tempfile
    from os.path import dirname, join

    def extract_defs(path, names, namespace):
        with open(path) as f:
            tree = ast.parse(f.read(), filename=path)
        for node in tree.body:
            if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
                exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), namespace)
        return namespace

    class _CapDataset(torch.utils.data.Dataset):
        def __init__(self, names, labels, img_transformer=None):
            self.names, self.labels = names, labels

        def __len__(self):
            return len(self.names)

        def __getitem__(self, i):
            return self.names[i], int(self.labels[i])

    class _Train(_CapDataset):
        kind = "train"

    class _Test(_CapDataset):
        kind = "test"

    pdb_stub = types.ModuleType("pdb")
    pdb_stub.set_trace = lambda *a, **k: None
    real_pdb = sys.modules.get("pdb")
    sys.modules["pdb"] = pdb_stub
    tmp = tempfile.mkdtemp()
    nsd = {"sample": random.sample}
    extract_defs(os.path.join(REF, "data/ImageLoader.py"), {"get_random_subset", "_dataset_info", "get_split_dataset_info"}, nsd)
    nsd.update({"torch": torch, "join": join, "dirname": dirname, "__file__": os.path.join(tmp, "data_helper.py"),
                "get_train_transformers": lambda a: None, "get_val_transformer": lambda a: None,
                "ImageDataset": _Train, "ImageTestDataset": _Test})
    extract_defs(os.path.join(REF, "data/data_helper.py"),
                 {"get_train_dataloader", "get_test_dataloader", "creat_train_loader_list", "Subset"}, nsd)

    domains = ["art_painting", "cartoon", "photo", "sketch"]
    classes = ["dog", "elephant", "giraffe"]
    rsl = np.random.RandomState(5)

    def make_lists(root, fusion_mode, target, sources):
        """Lists shaped like the shipped ones (data/txt_lists/pacs/*.txt; expanded '-K' lists as data_list_generator.py writes them)."""
        files = {}
        for d in domains:
            rows = []
            for ci, c in enumerate(classes):
                for k in range(int(rsl.randint(5, 9))):
                    rows.append(("/disk1/x/CCST/data/PACS/kfold/%s/%s/pic_%03d.jpg" % (d, c, k), ci))
            files[d] = rows
        base = os.path.join(root, "txt_lists", "pacs")
        os.makedirs(base, exist_ok=True)
        for d in domains:
            for split in ("train", "test"):
                with open(os.path.join(base, "%s_%s.txt" % (d, split)), "w") as f:
                    for n_, l_ in files[d]:
                        f.write("%s %d\n" % (n_, l_))
        sub = os.path.join(root, "txt_lists", "pacs_%s" % fusion_mode, target)
        os.makedirs(sub, exist_ok=True)
        for d in sources:
            with open(os.path.join(sub, "%s_train.txt" % d), "w") as f:
                for n_, l_ in files[d]:
                    if "-K" in fusion_mode:
                        style = fusion_mode.split("-")[1]
                        outp = n_.replace("kfold/", "kfold_adain-%s-multi/%s/" % (style, target))
                        K = int(fusion_mode[-1])
                        for t_ in list(rsl.choice(sources, K, replace=False)):
                            f.write("%s %d\n" % (outp if t_ == d else outp.replace(".", "_" + t_ + "."), l_))
                    else:
                        f.write("%s %d\n" % (n_, l_))
        return files

    cases = []
    for fusion_mode, mode, limit_source, limit_target in [("no_fusion", "fedavg", None, None), ("adain-overall-K3", "fedavg", None, None),
                                                           ("adain-single-K2", "fedavg", None, 7), ("adain-overall-K1", "deepall", None, None),
                                                           ("adain-overall-K2", "fedavg", 11, None), ("no_fusion", "deepall", 13, 5)]:
        target = "photo"
        sources = [d for d in domains if d != target]
        root = os.path.join(tmp, "%s_%s_%s" % (fusion_mode, mode, limit_source))
        os.makedirs(root)
        make_lists(root, fusion_mode, target, sources)
        nsd["__file__"] = os.path.join(root, "data_helper.py")
        a = types.SimpleNamespace(source=list(sources), target=target, dataset="pacs", fusion_mode=fusion_mode, mode=mode, val_size=0.1,
                                  dg_method="no_DG", limit_source=limit_source, limit_target=limit_target, batch=4, image_size=222,
                                  min_scale=0.8, max_scale=1.0, random_horiz_flip=0.0)
        random.seed(a=1)
        torch.manual_seed(1)
        loaders, val_loaders = nsd["get_train_dataloader"](a)
        test_loader = nsd["get_test_dataloader"](a)

        def describe(loader):
            ds = loader.dataset
            idx = None
            if hasattr(ds, "indices"):
                idx, ds = [int(i) for i in ds.indices], ds.dataset
            return {"names": list(ds.names), "labels": [int(l_) for l_ in ds.labels], "indices": idx, "kind": ds.kind,
                    "batch_size": loader.batch_size, "shuffle": isinstance(loader.sampler, torch.utils.data.RandomSampler)}
        cases.append({"fusion_mode": fusion_mode, "mode": mode, "limit_source": limit_source, "limit_target": limit_target,
                      "target": target, "source": sources, "root": os.path.relpath(root, tmp),
                      "lists": {os.path.relpath(os.path.join(dp, fn), root): open(os.path.join(dp, fn)).read()
                                for dp, _, fns in os.walk(root) for fn in fns},
                      "train": [describe(l_) for l_ in loaders], "val": [describe(l_) for l_ in val_loaders], "test": describe(test_loader)})
    # creat_train_loader_list on the modes the CLI cannot reach but the function handles ('multi' expansion, :128-143)
    direct = []
    for fusion_mode in ["no_fusion", "adain-overall-K3", "adain-single-K1", "adain-overall-multi", "adain-single-multi", "multi"]:
        names_in = ["/d/PACS/kfold/%s/%s/p%d.jpg" % (d, c, k) for d in ("cartoon", "sketch") for c in classes for k in range(2)]
        labels_in = [classes.index(n_.split("/")[-2]) for n_ in names_in]
        n_out, l_out = nsd["creat_train_loader_list"](list(names_in), list(labels_in), fusion_mode, ["art_painting", "cartoon", "sketch"], "photo")
        direct.append({"mode": fusion_mode, "names_in": names_in, "labels_in": labels_in, "names": n_out, "labels": [int(x_) for x_ in l_out]})
    random.seed(a=1)
    split = nsd["get_random_subset"](["n%d" % i for i in range(57)], [i % 7 for i in range(57)], 0.1)
    blob = json.dumps({"cases": cases, "direct": direct, "split57": [list(x_) for x_ in split]})
    save("data_lists", json=np.frombuffer(blob.encode(), dtype=np.uint8))
    if real_pdb is not None:
        sys.modules["pdb"] = real_pdb
    else:
        del sys.modules["pdb"]
# ---------------------------------------------------------------------------------------------------------------------------------
# cli_flags: the LIVE add_argument calls of the four scripts the drop-in CLIs mirror: option strings, default, action.
# Commented-out flags are not in the AST.  Data only (names and literal defaults).
if wanted("cli_flags"):
    import json

    def flags_of(path):
        with open(path) as f:
            tree = ast.parse(f.read(), filename=path)
        out = []
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
                names = [a.value for a in node.args if isinstance(a, ast.Constant) and isinstance(a.value, str)]
                kw = {}
                for k in node.keywords:
                    if k.arg in ("default", "action", "nargs"):
                        try:
                            kw[k.arg] = ast.literal_eval(k.value)
                        except ValueError:
                            kw[k.arg] = None
                    elif k.arg == "type" and isinstance(k.value, ast.Name):
                        kw["type"] = k.value.id
                out.append({"names": names, **kw})
        return out
    scripts = {"mean_std_computation_effcientMem.py": "style_transfer/AdaIN/mean_std_computation_effcientMem.py",
               "CCST_OverallStyleTransfer.py": "style_transfer/AdaIN/CCST_OverallStyleTransfer.py",
               "CCST_SingleStyleTransfer.py": "style_transfer/AdaIN/CCST_SingleStyleTransfer.py",
               "fed_run.py": "federated/fed_run.py"}
    blob = json.dumps({k: flags_of(os.path.join(REF, v)) for k, v in scripts.items()}, indent=1, sort_keys=True)
    with open(os.path.join(OUT, "cli_flags.json"), "w") as f:
        f.write(blob + "\n")
    print("wrote cli_flags.json")
print("done")
