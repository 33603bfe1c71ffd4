"""Oracle (test infrastructure only): AdaIN style-transfer path on CPU, fp32.

Every function cites the reference lines it restates (paths relative to
/root/reference).  Weights are passed as a plain ``{key: tensor}`` dict with
the reference's state-dict keys ("<sequential index>.weight" / ".bias").
"""
import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------
# Layer tables.  style_transfer/AdaIN/net.py:38-92 (vgg) and :6-36 (decoder).
# ("conv", idx, cin, cout, k) | ("pad",) | ("relu",) | ("pool",) | ("up",)
# ---------------------------------------------------------------------------
def _vgg_table():
    t = [("conv", 0, 3, 3, 1)]
    idx = 1
    cfg = [(3, 64), (64, 64), "P", (64, 128), (128, 128), "P", (128, 256),
           (256, 256), (256, 256), (256, 256), "P", (256, 512), (512, 512),
           (512, 512), (512, 512), "P", (512, 512), (512, 512), (512, 512),
           (512, 512)]
    for c in cfg:
        if c == "P":
            t.append(("pool",))
            idx += 1
        else:
            t.append(("pad",))
            t.append(("conv", idx + 1, c[0], c[1], 3))
            t.append(("relu",))
            idx += 3
    return t


def _decoder_table():
    t = []
    idx = 0
    cfg = [(512, 256), "U", (256, 256), (256, 256), (256, 256), (256, 128),
           "U", (128, 128), (128, 64), "U", (64, 64), (64, 3)]
    for i, c in enumerate(cfg):
        if c == "U":
            t.append(("up",))
            idx += 1
        else:
            t.append(("pad",))
            t.append(("conv", idx + 1, c[0], c[1], 3))
            idx += 2
            if i != len(cfg) - 1:          # net.py:35 -- last conv has no ReLU
                t.append(("relu",))
                idx += 1
    return t


VGG_TABLE = _vgg_table()          # 53 entries, net.py:38-92
DECODER_TABLE = _decoder_table()  # 29 entries, net.py:6-36
assert len(VGG_TABLE) == 53 and len(DECODER_TABLE) == 29


def conv_keys(table):
    return [(e[1], e[2], e[3], e[4]) for e in table if e[0] == "conv"]


def he_weights(table, seed, bias_std=0.05):
    """Seeded He-normal weights (the pretrained .pth files are
    absent, default init collapses activations).  numpy RandomState is a
    frozen stream, so fixtures only need to store outputs."""
    rs = np.random.RandomState(seed)
    w = {}
    for idx, cin, cout, k in conv_keys(table):
        fan_in = cin * k * k
        if k == 1:   # the 3->3 colour pre-transform: near-identity, like the real one
            wt = np.eye(3, dtype=np.float32).reshape(3, 3, 1, 1) + \
                rs.normal(0, 0.1, (cout, cin, 1, 1)).astype(np.float32)
        else:
            wt = rs.normal(0, np.sqrt(2.0 / fan_in), (cout, cin, k, k)).astype(np.float32)
        w["%d.weight" % idx] = torch.from_numpy(wt)
        w["%d.bias" % idx] = torch.from_numpy(rs.normal(0, bias_std, (cout,)).astype(np.float32))
    return w


def run_table(table, x, weights, upto=None):
    """Sequential forward.  Semantics per SURVEY.md Appendix D:
    ReflectionPad2d((1,1,1,1)) net.py:7; Conv2d stride 1 no padding with bias
    net.py:8; MaxPool2d((2,2),(2,2),(0,0),ceil_mode=True) net.py:46;
    Upsample(scale_factor=2, mode='nearest') net.py:10."""
    for i, e in enumerate(table):
        if upto is not None and i >= upto:
            break
        if e[0] == "conv":
            x = F.conv2d(x, weights["%d.weight" % e[1]], weights["%d.bias" % e[1]])
        elif e[0] == "pad":
            x = F.pad(x, (1, 1, 1, 1), mode="reflect")
        elif e[0] == "relu":
            x = F.relu(x)
        elif e[0] == "pool":
            x = F.max_pool2d(x, (2, 2), (2, 2), (0, 0), ceil_mode=True)
        elif e[0] == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
    return x


def encoder(x, vgg_w):
    """vgg[:31] = up to relu4_1.  CCST_OverallStyleTransfer.py:124."""
    return run_table(VGG_TABLE, x, vgg_w, upto=31)


def decoder(x, dec_w):
    """net.decoder, net.py:6-36."""
    return run_table(DECODER_TABLE, x, dec_w)


def calc_mean_std(feat, eps=1e-5):
    """function.py:4-13 -- per-(n,c) mean and sqrt(UNBIASED var + eps)."""
    assert feat.dim() == 4
    N, C = feat.shape[:2]
    var = feat.reshape(N, C, -1).var(dim=2) + eps
    std = var.sqrt().view(N, C, 1, 1)
    mean = feat.reshape(N, C, -1).mean(dim=2).view(N, C, 1, 1)
    return mean, std


def adain_style_stat(content_feat, style_stat):
    """function.py:26-33 (adaIN_StyleStat_ContentFeat)."""
    size = content_feat.size()
    style_mean, style_std = style_stat
    c_mean, c_std = calc_mean_std(content_feat)
    normalized = (content_feat - c_mean.expand(size)) / c_std.expand(size)
    return normalized * style_std.expand(size) + style_mean.expand(size)


def adain(content_feat, style_feat):
    """function.py:16-24 (adaptive_instance_normalization)."""
    assert content_feat.size()[:2] == style_feat.size()[:2]
    size = content_feat.size()
    s_mean, s_std = calc_mean_std(style_feat)
    c_mean, c_std = calc_mean_std(content_feat)
    normalized = (content_feat - c_mean.expand(size)) / c_std.expand(size)
    return normalized * s_std.expand(size) + s_mean.expand(size)


def style_transfer(vgg_w, dec_w, content, style_stat, alpha=1.0, interpolation_weights=None):
    """CCST_OverallStyleTransfer.py:32-46.  The interpolation branch (:36-42;
    unreachable from the CLIs: do_interpolation is never set, :109) mixes the
    stylised features of the batch by the weights, from a zero tensor in index
    order, and blends with the first image's features."""
    assert 0.0 <= alpha <= 1.0
    content_f = encoder(content, vgg_w)
    if interpolation_weights:
        _, C, H, W = content_f.shape
        feat = torch.zeros(1, C, H, W, dtype=torch.float32)
        base_feat = adain_style_stat(content_f, style_stat)
        for i, w in enumerate(interpolation_weights):
            feat = feat + w * base_feat[i:i + 1]
        content_f = content_f[0:1]
    else:
        feat = adain_style_stat(content_f, style_stat)
    feat = feat * alpha + content_f * (1 - alpha)
    return decoder(feat, dec_w)


def calc_sum(feat):
    """mean_std_computation_effcientMem.py:103-115 (copy at
    CCST_SingleStyleTransfer.py:55-67): per-channel sum, sum of squares over
    N*H*W, and the count."""
    feat = feat.detach()
    N, C, H, W = feat.shape
    count = N * H * W
    f = feat.swapaxes(1, 0)
    s = f.reshape(C, -1).sum(axis=1).reshape(1, C, 1, 1)
    sq = (f ** 2).reshape(C, -1).sum(axis=1).reshape(1, C, 1, 1)
    return s, sq, count


def finalise_stats(all_sum, all_sqsum, all_count):
    """mean_std_computation_effcientMem.py:135-137 (and
    CCST_SingleStyleTransfer.py:201-203): BIASED variance E[x^2]-mu^2, +1e-5."""
    mean = all_sum / float(all_count)
    var = all_sqsum / float(all_count) - mean ** 2
    std = torch.sqrt(var + 1e-5)
    return mean, std


def overall_style_stats(batches, vgg_w):
    """The stage-1 loop, mean_std_computation_effcientMem.py:117-137."""
    tot_s, tot_q, tot_n = 0, 0, 0
    for data in batches:
        s, q, n = calc_sum(encoder(data, vgg_w))
        tot_s = tot_s + s
        tot_q = tot_q + q
        tot_n += n
    return finalise_stats(tot_s, tot_q, tot_n)


# ---------------------------------------------------------------------------
# Deterministic synthetic inputs shared by fixtures, tests and bench
# (SURVEY.md 8d "Synthetic inputs 1").
# ---------------------------------------------------------------------------
def synth_content(n, h, w, seed=1):
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.uniform(0.0, 1.0, (n, 3, h, w)).astype(np.float32))


def synth_style_stat(c=512, seed=7):
    rs = np.random.RandomState(seed)
    mean = rs.normal(0.5, 0.3, (1, c, 1, 1)).astype(np.float32)
    std = rs.uniform(0.5, 1.5, (1, c, 1, 1)).astype(np.float32)
    return [torch.from_numpy(mean), torch.from_numpy(std)]
