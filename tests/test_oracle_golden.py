"""CPU: the oracle (oracle/) reproduces the reference's own outputs
(tests/golden/, made by tools/make_golden.py from /root/reference) bit-for-bit."""
import copy

import numpy as np
import torch
from torch import nn

from oracle import adain_ref as A
from oracle import fed_ref as Fd
from oracle import resnet_ref as R

torch.set_num_threads(8)
VGG_W = A.he_weights(A.VGG_TABLE, seed=1234)
DEC_W = A.he_weights(A.DECODER_TABLE, seed=4321)


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_calc_mean_std(golden):
    g = golden("calc_mean_std")
    rs = np.random.RandomState(int(g["seed"]))
    feat = torch.from_numpy(rs.normal(0.3, 1.2, (2, 8, 5, 7)).astype(np.float32))
    m, s = A.calc_mean_std(feat)
    assert torch.equal(m, t(g["mean"])) and torch.equal(s, t(g["std"]))


def test_adain_feat(golden):
    g = golden("adain_feat")
    rs = np.random.RandomState(int(g["seed"]))
    cf = torch.from_numpy(np.abs(rs.normal(0.0, 0.6, (2, 512, 8, 8))).astype(np.float32))
    sf = torch.from_numpy(np.abs(rs.normal(0.2, 0.9, (2, 512, 6, 10))).astype(np.float32))
    stat = A.synth_style_stat(512, seed=7)
    assert torch.equal(A.adain_style_stat(cf, stat), t(g["out_stat"]))
    assert torch.equal(A.adain(cf, sf), t(g["out_feat"]))


def test_overall_stats(golden):
    g = golden("overall_stats")
    batches = [A.synth_content(2, 32, 48, seed=int(s)) for s in g["seeds"]]
    s0, q0, n0 = A.calc_sum(A.encoder(batches[0], VGG_W))
    assert torch.equal(s0, t(g["sum0"])) and torch.equal(q0, t(g["sq0"])) and n0 == int(g["n0"])
    mean, std = A.overall_style_stats(batches, VGG_W)
    assert torch.equal(mean, t(g["mean"])) and torch.equal(std, t(g["std"]))


def test_style_transfer_64(golden):
    g = golden("style_transfer_64")
    content = A.synth_content(2, 64, 64, seed=int(g["seed"]))
    stat = A.synth_style_stat(512, seed=7)
    assert torch.equal(A.encoder(content, VGG_W), t(g["relu4_1"]))
    assert torch.equal(A.style_transfer(VGG_W, DEC_W, content, stat, 1.0), t(g["out"]))
    assert torch.equal(A.style_transfer(VGG_W, DEC_W, content, stat, 0.5), t(g["out_alpha05"]))


def test_style_transfer_interpolation(golden):
    """CCST_OverallStyleTransfer.py:36-42 against the reference's own outputs."""
    g = golden("style_transfer_interp")
    content3 = A.synth_content(1, 64, 64, seed=int(g["seed"])).repeat(3, 1, 1, 1)
    stats3 = [A.synth_style_stat(512, seed=int(s)) for s in g["style_seeds"]]
    stat3 = [torch.cat([s[0] for s in stats3]), torch.cat([s[1] for s in stats3])]
    wts = [float(w) for w in g["weights"]]
    assert torch.equal(A.style_transfer(VGG_W, DEC_W, content3, stat3, 1.0, wts), t(g["out"]))
    assert torch.equal(A.style_transfer(VGG_W, DEC_W, content3, stat3, 0.6, wts), t(g["out_alpha06"]))


def test_style_transfer_odd(golden):
    g = golden("style_transfer_odd")
    stat = A.synth_style_stat(512, seed=7)
    out = A.style_transfer(VGG_W, DEC_W, A.synth_content(1, 222, 222, seed=int(g["seed"])), stat, 1.0)
    assert list(out.shape) == list(g["out_shape"]) == [1, 3, 224, 224]
    assert torch.equal(out[:, :, ::4, ::4], t(g["out_sub4"]))
    assert torch.equal(out.sum(dim=(0, 2, 3)), t(g["chan_sum"]))
    out2 = A.style_transfer(VGG_W, DEC_W, A.synth_content(1, 50, 84, seed=int(g["seed2"])), stat, 1.0)
    assert torch.equal(out2, t(g["out2"]))


def _resnet_case(g, model):
    seed, classes, nb = int(g["seed"]), int(g["classes"]), int(g["nb"])
    model.load_state_dict(R.seeded_state_dict(model, seed, float(g["residual_gamma"]), float(g["fc_gain"])))
    x, y = R.synth_batch(nb, 222, classes, seed=seed + 1)
    model.eval()
    with torch.no_grad():
        assert torch.equal(model(x), t(g["logit_eval"]))
    loss, logit = R.train_step(model, x, y, float(g["lr"]))
    assert torch.equal(logit, t(g["logit_train"])) and torch.equal(loss, t(g["loss"]))
    model.eval()
    with torch.no_grad():
        assert torch.equal(model(x), t(g["logit_after"]))
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("state/"):
            assert torch.equal(sd[k[6:]], t(g[k])), k
    named = dict(model.named_parameters())
    for k in g.files:
        if k.startswith("grad_head/"):
            assert torch.equal(named[k[10:]].grad.flatten()[:16], t(g[k])), k


def test_resnet18_step(golden):
    _resnet_case(golden("resnet18_step"), R.resnet18(2))


def test_resnet50_step(golden):
    _resnet_case(golden("resnet50_step"), R.resnet50(7))


def test_communication(golden):
    g = golden("communication")
    server = R.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3)
    server.load_state_dict(R.seeded_state_dict(server, int(g["seed"])))
    clients = [copy.deepcopy(server) for _ in range(3)]
    for ci, c in enumerate(clients):
        rs = np.random.RandomState(71 + ci)
        with torch.no_grad():
            for k, v in c.state_dict().items():
                if "num_batches_tracked" in k:
                    v.fill_(5 + ci)
                else:
                    v += torch.from_numpy(rs.normal(0, 0.02, tuple(v.shape)).astype(np.float32))
    server, clients = Fd.communication_fedavg(server, clients, [float(w) for w in g["weights"]])
    keys = list(server.state_dict().keys())
    assert keys == [str(k) for k in g["keys"]]
    ksum = np.array([float(server.state_dict()[k].double().sum()) for k in keys])
    assert np.array_equal(ksum, g["key_sum"])
    assert torch.equal(server.state_dict()["conv1.weight"].flatten()[:32], t(g["conv1_head"]))
    nbt = [k for k in keys if "num_batches_tracked" in k]
    assert [int(server.state_dict()[k]) for k in nbt] == list(g["nbt_server"]) == [5] * len(nbt)
    for ci, c in enumerate(clients):
        assert [int(c.state_dict()[k]) for k in nbt] == list(g["nbt_clients"][ci])


def test_communication_fedbn(golden):
    """--mode fedbn (fed_run.py:388-399): server averages everything, clients keep keys containing 'bn'."""
    g = golden("communication_fedbn")
    server = R.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3)
    server.load_state_dict(R.seeded_state_dict(server, int(g["seed"])))
    clients = [copy.deepcopy(server) for _ in range(3)]
    for ci, c in enumerate(clients):
        rs = np.random.RandomState(71 + ci)
        with torch.no_grad():
            for k, v in c.state_dict().items():
                if "num_batches_tracked" in k:
                    v.fill_(5 + ci)
                else:
                    v += torch.from_numpy(rs.normal(0, 0.02, tuple(v.shape)).astype(np.float32))
    server, clients = Fd.communication_fedbn(server, clients, [float(w) for w in g["weights"]])
    fkeys = [str(k) for k in g["keys"]]
    assert fkeys == [k for k in server.state_dict().keys() if "num_batches_tracked" not in k]
    assert np.array_equal(np.array([float(server.state_dict()[k].double().sum()) for k in fkeys]), g["server_sum"])
    for ci, c in enumerate(clients):
        assert np.array_equal(np.array([float(c.state_dict()[k].double().sum()) for k in fkeys]), g["client_sum"][ci])
    shared = [all(torch.equal(server.state_dict()[k], c.state_dict()[k]) for c in clients) for k in fkeys]
    assert shared == [bool(b) for b in g["shared"]] == ['bn' not in k for k in fkeys]
    assert shared[fkeys.index("layer2.0.downsample.1.weight")] and not shared[fkeys.index("bn1.weight")]
    assert torch.equal(clients[1].state_dict()["bn1.weight"].flatten()[:16], t(g["bn1_weight_client1"]))


def test_fed_train_test_loops(golden):
    """oracle.fed_ref.train_epoch / test_epoch vs the reference's own train() / test() (fed_run.py:31-88, :214-259,
    AST-extracted and run by tools/make_golden.py with a stub logger): two epochs over a ragged 3-batch loader with
    one optimiser (fed_run.py:657), a test pass on held-out batches and one on the training batches -- bit for bit."""
    g = golden("fed_loop")
    model = R.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3)
    model.load_state_dict(R.seeded_state_dict(model, int(g["seed"])))
    train_loader = [R.synth_batch(int(n), 222, 3, seed=int(s)) for s, n in zip(g["train_seeds"], g["train_sizes"])]
    test_loader = [R.synth_batch(int(n), 222, 3, seed=int(s)) for s, n in zip(g["test_seeds"], g["test_sizes"])]
    ce, lr = nn.CrossEntropyLoss(), float(g["lr"])
    log = []
    tr1 = Fd.train_epoch(model, train_loader, lr, ce, log=log)
    te1 = Fd.test_epoch(model, test_loader, ce)
    tr2 = Fd.train_epoch(model, train_loader, lr, ce, log=log)
    te2 = Fd.test_epoch(model, train_loader, ce)
    for got, key in ((tr1, "train1"), (te1, "test1"), (tr2, "train2"), (te2, "test2")):
        assert np.array_equal(np.array(got, dtype=np.float64), g[key]), (key, got, g[key])
    assert np.array_equal(np.array([r[0] for r in log], dtype=np.float64), g["log_loss"])
    assert [r[1] for r in log] == list(g["log_right"]) and [r[2] for r in log] == list(g["log_total"])
    sd = model.state_dict()
    assert torch.equal(sd["conv1.weight"].flatten()[:64], t(g["conv1_head"]))
    assert torch.equal(sd["class_classifier.weight"], t(g["fc_weight"])) and torch.equal(sd["class_classifier.bias"], t(g["fc_bias"]))
    assert torch.equal(sd["bn1.running_mean"], t(g["bn1_running_mean"])) and torch.equal(sd["bn1.running_var"], t(g["bn1_running_var"]))
    assert int(sd["bn1.num_batches_tracked"]) == int(g["nbt"]) == 6
    assert np.array_equal(np.array([float(v.double().sum()) for v in sd.values()]), g["key_sum"])
