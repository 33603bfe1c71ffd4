"""GPU parity of the ResNet training path (through the C ABI) against torch-CPU op references, the CPU
oracle and the reference-generated golden vectors.  Tolerance: 1e-3 (fp32), per BASELINE.json."""
import copy
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu
ARGS = types.SimpleNamespace(dg_method="", mode="fedavg")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32))


def cl(t, dev):
    """channels_last device copy (differentiable inputs must already be NHWC in memory)."""
    return t.to(dev).contiguous(memory_format=torch.channels_last)


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("cfg", [
    # N, Cin, H, W, Cout, k, stride, pad
    (2, 64, 14, 14, 64, 3, 1, 1),
    (2, 64, 15, 13, 128, 3, 2, 1),
    (3, 128, 9, 9, 256, 1, 1, 0),
    (2, 256, 14, 14, 512, 1, 2, 0),
    (2, 128, 7, 7, 128, 3, 1, 1),
    (1, 64, 28, 28, 256, 1, 1, 0),
])
def test_conv_fwd_bwd(dev, cfg):
    from ccst_amd.nets import resnet
    N, Cin, H, W, Cout, k, stride, pad = cfg
    x = rnd((N, Cin, H, W), 1)
    w = rnd((Cout, Cin, k, k), 2, (2.0 / (Cin * k * k)) ** 0.5)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=stride, padding=pad)
    g = rnd(tuple(yr.shape), 3)
    yr.backward(g)
    conv = resnet.Conv2d(Cin, Cout, k, stride=stride, padding=pad, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w)
    xd = cl(x, dev).requires_grad_(True)
    y = conv(xd)
    y.backward(g.to(dev))
    assert relerr(y, yr) < 1e-4
    assert relerr(xd.grad, xr.grad) < 1e-4, "bwd-data"
    assert relerr(conv.weight.grad, wr.grad) < 1e-4, "bwd-weight"
    # gradients accumulate (second backward adds)
    y2 = conv(xd)
    y2.backward(g.to(dev))
    assert relerr(conv.weight.grad, 2 * wr.grad) < 1e-4


@pytest.mark.parametrize("cfg", [
    # N, Cin, H, W, Cout, stride: the pointwise streaming kernel's corner cases
    (5, 32, 9, 7, 192, 1),        # partial last row tile (M = 315), three column tiles (not a power of two)
    (3, 64, 13, 13, 128, 2),      # strided (downsample branch), M = 147
    (8, 96, 48, 48, 256, 1),      # more tiles (1152) than workgroups (1024): several tiles per workgroup, unequal counts
    (2, 2048, 7, 7, 64, 1),       # long K (64 k-steps), one column tile
])
def test_pointwise_conv_statistics(dev, cfg):
    """Forward 1x1 conv with the BatchNorm-statistics epilogue (ccst_conv2d_igemm_stats_f32 on conv1x1_stream_kernel): output against
    torch, and the partial slabs -- as many as ccst_conv2d_igemm_stats_groups promises -- must add up to the output's per-channel
    sum and sum of squares."""
    from ccst_amd import _lib, ops
    N, Cin, H, W, Cout, stride = cfg
    g = torch.Generator().manual_seed(29)
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * (1.0 / Cin) ** 0.5
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, stride=stride).permute(0, 2, 3, 1)
    pc = ops.pack_conv_weight(w.to(dev))
    y, st = ops.conv2d_nhwc(x.to(dev), pc, stride=stride, pad=0, want_stats=True)
    M = ref.shape[0] * ref.shape[1] * ref.shape[2]
    assert st.shape[0] == _lib.load().ccst_conv2d_igemm_stats_groups(M, Cout, Cin, 1) and st.shape[1:] == (Cout, 2)
    assert float((y.cpu() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max())) * (Cin / 64) ** 0.5 * 4
    tot = st.double().sum(0).cpu()
    r2 = ref.reshape(-1, Cout).double()
    assert float((tot[:, 0] - r2.sum(0)).abs().max()) < 1e-4 * max(1.0, float(r2.sum(0).abs().max()))
    assert float((tot[:, 1] - (r2 * r2).sum(0)).abs().max()) < 1e-4 * float((r2 * r2).sum(0).abs().max())


@pytest.mark.parametrize("xscale,wscale", [(1.0, 1.0), (1e-6, 1.0), (3e4, 1.0), (1e5, 1000.0), (1e20, 1e-10), (1.0, 1e4)])
def test_pointwise_conv_half_pieces_any_magnitude(dev, xscale, wscale):
    """the pointwise training forward multiplies IEEE-half pieces; with round 3's fixed 2^8 weight scale and unscaled
    activations |w| >= 256 or |x| >= 65504 became inf / NaN.  ccst_conv2d_pointwise_half_f32 scales both operands by powers of two
    derived on the device from their |max| words: the result must be at the fp32 level at any magnitude (incl. weights of 1e4 and
    activations of 1e5), and the BatchNorm statistics of its epilogue must be those of its output."""
    from ccst_amd import ops
    g = torch.Generator().manual_seed(31)
    N, H, W, Cin, Cout = 4, 28, 28, 256, 128
    x = (torch.randn(N, H, W, Cin, generator=g) * xscale).to(dev)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * (1.0 / Cin) ** 0.5 * wscale).to(dev)
    pc = ops.pack_conv_weight(w)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu()).permute(0, 2, 3, 1)
    wmax = ops.absmax(w)
    y, st = ops.conv2d_nhwc(x, pc, want_stats=True, x_absmax=ops.absmax(x), w_absmax=wmax, w_split=ops.pack_conv_weight_split(w, wmax))
    y32, _ = ops.conv2d_nhwc(x, pc, want_stats=True)                       # (no words: the fp32 MFMA)
    assert bool(torch.isfinite(y).all())
    scale = float(ref.abs().max())
    e16, e32 = float((y.double().cpu() - ref).abs().max()) / scale, float((y32.double().cpu() - ref).abs().max()) / scale
    print("x scale %g, w scale %g: half pieces %.2e, fp32 MFMA %.2e of max |y|" % (xscale, wscale, e16, e32))
    assert e16 < 4e-6, e16
    tot = st.double().sum(0).cpu()
    r2 = y.double().cpu().reshape(-1, Cout)
    assert float((tot[:, 0] - r2.sum(0)).abs().max()) < 1e-4 * max(1e-30, float(r2.sum(0).abs().max()))
    assert float((tot[:, 1] - (r2 * r2).sum(0)).abs().max()) < 1e-4 * float((r2 * r2).sum(0).abs().max())


def test_resnet_step_runs_the_half_piece_forward_with_words(dev):
    """In a train step every pointwise conv behind a BatchNorm gets the |max| words of its input (left by that BatchNorm's apply
    kernel) and of its weight, and after an optimiser step the weight words are the batched side-stream refresh's."""
    import types
    from ccst_amd import fed, nn_ops, ops
    from ccst_amd.nets import models
    assert nn_ops.HALF_FWD
    torch.manual_seed(3)
    model = models.get_network("resnet50")(types.SimpleNamespace(dg_method=""), pretrained=False, classes=7).to(dev)
    model.train()
    opt, loss_fun = fed.SGD(model, lr=0.01), fed.CrossEntropyLoss()
    x, y = torch.randn(2, 3, 222, 222, device=dev), torch.randint(0, 7, (2,), device=dev)
    seen = []
    real = ops.conv2d_nhwc

    def spy(xx, pc, *a, **k):
        if k.get("want_stats") and pc.kh == 1:
            seen.append((k.get("x_absmax") is not None, k.get("w_absmax") is not None))
        return real(xx, pc, *a, **k)
    ops.conv2d_nhwc = spy
    try:
        for _ in range(2):
            opt.zero_grad()
            loss = loss_fun(model(x), y)
            fed.backward(loss)
            opt.step()
    finally:
        ops.conv2d_nhwc = real
    torch.cuda.synchronize()
    assert seen and all(a and b for a, b in seen), seen[:8]
    conv = model.layer1[0].conv1
    nn_ops.join_prepack(conv.weight.device)
    torch.cuda.synchronize()
    words = conv.__dict__["_ccst_wmax"][1]
    got = float(torch.tensor(int(words.max()), dtype=torch.int32).view(torch.float32))
    assert got == float(conv.weight.detach().abs().max()), (got, float(conv.weight.detach().abs().max()))
    assert bool(torch.isfinite(loss))


@pytest.mark.parametrize("shape", [(3, 28, 28, 128, 128), (2, 56, 56, 64, 64), (2, 14, 14, 256, 192), (1, 13, 19, 32, 48)])
def test_halo_train_form_vs_gather(dev, shape):
    """ccst_conv3x3_halo_train_f32 (ResNet-trunk form of the halo kernel): forward + BN statistics, backward-data by
    flipped taps, and y += conv, against the gather kernel's results for the same packed weights."""
    from ccst_amd import nn_ops, ops
    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
    dy = torch.randn(N, H, W, Cout, generator=g).to(dev)
    base = torch.randn(N, H, W, Cin, generator=g).to(dev)
    pc, pct = ops.pack_conv_weight(w), ops.pack_conv_weight(w, transpose=True)
    assert not ops.HALO_ZERO_PAD
    y_ref, st_ref = ops.conv2d_nhwc(x, pc, stride=1, pad=1, want_stats=True)
    dx_ref = nn_ops.conv_bwd_data(dy, pct, (N, H, W, Cin), 1, 1)
    y, st = ops.conv3x3_halo_train(x, pc, want_stats=True)
    assert float((y - y_ref).abs().max()) < 1e-4 * max(1.0, float(y_ref.abs().max()))
    tot, tot_ref = st.double().sum(0), st_ref.double().sum(0)          # the partials are grouped differently; totals agree
    assert float((tot - tot_ref).abs().max()) < 1e-4 * float(tot_ref.abs().max())
    dx = ops.conv3x3_halo_train(dy, pct, flip=True)
    assert float((dx - dx_ref).abs().max()) < 1e-4 * max(1.0, float(dx_ref.abs().max()))
    acc = base.clone()
    out = ops.conv3x3_halo_train(dy, pct, flip=True, accumulate_into=acc)
    assert out.data_ptr() == acc.data_ptr() and float((acc - (base + dx_ref)).abs().max()) < 1e-4 * max(1.0, float(dx_ref.abs().max()))
    acc2 = base.clone()
    nn_ops.conv_bwd_data(dy, pct, (N, H, W, Cin), 1, 1, accumulate_into=acc2)          # the gather kernel's CCST_CONV_ACCUM
    assert float((acc2 - (base + dx_ref)).abs().max()) < 1e-5 * max(1.0, float(dx_ref.abs().max()))


@pytest.mark.parametrize("shape", [(3, 28, 28, 128, 128), (2, 56, 56, 64, 64), (2, 14, 14, 256, 192), (1, 13, 19, 32, 48), (5, 7, 7, 512, 512)])
@pytest.mark.parametrize("gscale", [1.0, 1e-7])
def test_halo_train_form_on_half_pieces_vs_fp64(dev, shape, gscale):
    """ccst_conv3x3_halo_train_split_f32 (the trunk's 3x3 stride-1 layers on the 16-bit MFMA): forward + BatchNorm statistics,
    backward-data by flipped taps with the transposed half-piece image, and y += conv, against fp64 convolutions; the gradient
    operand at 1 and at 1e-7 of the activations' magnitude."""
    from ccst_amd import ops
    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
    dy = (torch.randn(N, H, W, Cout, generator=g) * gscale).to(dev)
    base = (torch.randn(N, H, W, Cin, generator=g) * gscale).to(dev)
    wmax = ops.absmax(w)
    ph, pht = ops.pack_halo_split(w, wmax), ops.pack_halo_split(w, wmax, bwd=True)
    y_ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    dx_ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w.double(), dy.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
    y, st = ops.conv3x3_halo_train_split(x, ops.absmax(x), ph, wmax, want_stats=True)
    assert float((y.double() - y_ref).abs().max()) < 4e-6 * float(y_ref.abs().max())
    tot = st.double().sum(0)
    ref_tot = torch.stack([y_ref.sum(dim=(0, 1, 2)), (y_ref * y_ref).sum(dim=(0, 1, 2))], dim=1)
    assert float((tot - ref_tot).abs().max()) < 1e-4 * float(ref_tot.abs().max())
    dmax = ops.absmax(dy)
    dx = ops.conv3x3_halo_train_split(dy, dmax, pht, wmax, flip=True)
    assert float((dx.double() - dx_ref).abs().max()) < 4e-6 * float(dx_ref.abs().max())
    acc = base.clone()
    out = ops.conv3x3_halo_train_split(dy, dmax, pht, wmax, flip=True, accumulate_into=acc)
    assert out.data_ptr() == acc.data_ptr()
    assert float((acc.double() - (base.double() + dx_ref)).abs().max()) < 4e-6 * float(dx_ref.abs().max()) + 1e-6 * float(base.abs().max())


@pytest.mark.parametrize("shape", [(3, 28, 28, 128, 128), (2, 56, 56, 64, 64), (1, 13, 19, 32, 48), (5, 7, 7, 512, 512)])
def test_halo_backward_data_with_the_batchnorm_relu_link(dev, shape):
    """The bn1 -> conv2 link of the half-piece halo kernel: its backward-data epilogue recomputes the ReLU mask of the BatchNorm whose
    output the conv read (bn(bn_x) > 0), stores the masked gradient and leaves that BatchNorm's backward partial sums -- against the
    un-linked launch masked and summed on the host side of the test; then ccst_bn_train_bwd_partials_f32 on those sums against the
    ordinary BatchNorm backward."""
    from ccst_amd import _lib, ops
    from ccst_amd._lib import check, ptr, stream_ptr
    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(29)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
    dy = torch.randn(N, H, W, Cout, generator=g).to(dev)
    bx = torch.randn(N, H, W, Cin, generator=g).to(dev)
    mean, invstd = (torch.randn(Cin, generator=g) * 0.1).to(dev), (torch.rand(Cin, generator=g) + 0.5).to(dev)
    gam, bet = (torch.rand(Cin, generator=g) + 0.5).to(dev), (torch.randn(Cin, generator=g) * 0.3).to(dev)
    wmax, dmax = ops.absmax(w), ops.absmax(dy)
    pht = ops.pack_halo_split(w, wmax, bwd=True)
    plain = ops.conv3x3_halo_train_split(dy, dmax, pht, wmax, flip=True)
    part = torch.zeros((ops.halo_stats_groups(N, H, W), Cin, 2), device=dev)
    got = ops.conv3x3_halo_train_split(dy, dmax, pht, wmax, flip=True, bn_relu=(bx, mean, invstd, gam, bet, part))
    xh = (bx - mean) * invstd
    want = torch.where(xh * gam + bet > 0, plain, torch.zeros((), device=dev))
    assert torch.equal(got, want)
    tot = part.double().sum(0)
    ref = torch.stack([want.double().sum(dim=(0, 1, 2)), (want.double() * xh.double()).sum(dim=(0, 1, 2))], dim=1)
    assert float((tot - ref).abs().max()) < 1e-4 * max(float(ref.abs().max()), 1e-30)
    lib = _lib.load()
    M = N * H * W
    ws = torch.empty(int(lib.ccst_bn_workspace_bytes(M, Cin)) // 4, device=dev)
    dx_a, dg_a, db_a = torch.empty_like(bx), torch.zeros(Cin, device=dev), torch.zeros(Cin, device=dev)
    check(lib.ccst_bn_train_bwd_partials_f32(ptr(got), ptr(bx), ptr(gam), ptr(mean), ptr(invstd), ptr(part), int(part.shape[0]), ptr(dx_a),
                                             ptr(dg_a), ptr(db_a), 0, M, Cin, ptr(ws), ws.numel() * 4, None, stream_ptr()), "bn bwd partials")
    dx_b, dg_b, db_b = torch.empty_like(bx), torch.zeros(Cin, device=dev), torch.zeros(Cin, device=dev)
    check(lib.ccst_bn_train_bwd_mask_f32(ptr(plain), ptr(bx), None, None, ptr(gam), ptr(bet), ptr(mean), ptr(invstd), 1, ptr(dx_b), None,
                                         ptr(dg_b), ptr(db_b), 0, M, Cin, ptr(ws), ws.numel() * 4, None, stream_ptr()), "bn bwd")
    for a, b in ((dx_a, dx_b), (dg_a, dg_b), (db_a, db_b)):
        assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))


def test_pointwise_backward_data_masked_accumulate(dev):
    """ccst_conv2d_igemm_accum_masked_f32: y = mask ? y + dX : 0 (the residual-block input gradient, masked by the previous block's
    ReLU in the conv's own epilogue) against torch; the byte mask has the layout ccst_bn_train_fwd_mask_f32 writes."""
    from ccst_amd import nn_ops, ops
    N, H, W, Cin, Cout = 3, 14, 10, 256, 64            # forward conv Cin -> Cout (1x1); backward-data yields Cin channels
    g = torch.Generator().manual_seed(23)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1
    dy = torch.randn(N, H, W, Cout, generator=g)
    base = torch.randn(N, H, W, Cin, generator=g)
    keep = torch.rand(N, H, W, Cin, generator=g) > 0.4
    bits = keep.reshape(-1, 4).to(torch.uint8)
    mask = (bits[:, 0] | (bits[:, 1] << 1) | (bits[:, 2] << 2) | (bits[:, 3] << 3)).contiguous()
    pct = ops.pack_conv_weight(w.to(dev), transpose=True)
    assert nn_ops.masked_accum_ok(dy.to(dev), pct, (N, H, W, Cin), 1, 0)
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w, dy.permute(0, 3, 1, 2)).permute(0, 2, 3, 1) + base
    ref = torch.where(keep, ref, torch.zeros(()))
    acc = base.to(dev).contiguous()
    out = nn_ops.conv_bwd_data(dy.to(dev), pct, (N, H, W, Cin), 1, 0, accumulate_into=acc, relu_mask=mask.to(dev))       # (without the BatchNorm link)
    assert out.data_ptr() == acc.data_ptr()
    assert float((out.cpu() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))
    assert bool((out.cpu()[~keep] == 0).all())
    # with the BatchNorm link the same launch also leaves that BatchNorm's backward partial sums: finalize + apply from them must
    # give the gradients of the plain backward on the masked sum
    import ctypes
    from ccst_amd import _lib
    from ccst_amd._lib import check, ptr, stream_ptr
    lib = _lib.load()
    M, C = N * H * W, Cin
    bx = torch.randn(N, H, W, Cin, generator=g).to(dev)
    gam = (torch.rand(C, generator=g) + 0.5).to(dev)
    bet = torch.zeros(C, device=dev)
    mean, invstd = bx.reshape(-1, C).mean(0).contiguous(), (1.0 / torch.sqrt(bx.reshape(-1, C).var(0, unbiased=False) + 1e-5)).contiguous()
    acc2 = base.to(dev).contiguous()
    part = torch.empty((nn_ops.lib_groups(M, C, Cout), C, 2), device=dev)
    out2 = nn_ops.conv_bwd_data(dy.to(dev), pct, (N, H, W, Cin), 1, 0, accumulate_into=acc2, relu_mask=mask.to(dev), bn_link=(bx, mean, invstd, part))
    assert torch.equal(out2, out)
    ws = torch.empty(int(lib.ccst_bn_workspace_bytes(M, C)) // 4, device=dev)
    dx_a, dg_a, db_a = torch.empty_like(bx), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    check(lib.ccst_bn_train_bwd_partials_f32(ptr(out2), ptr(bx), ptr(gam), ptr(mean), ptr(invstd), ptr(part), int(part.shape[0]), ptr(dx_a),
                                             ptr(dg_a), ptr(db_a), 0, M, C, ptr(ws), ws.numel() * 4, None, stream_ptr()), "bn bwd partials")
    dx_b, dg_b, db_b = torch.empty_like(bx), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    check(lib.ccst_bn_train_bwd_mask_f32(ptr(out2), ptr(bx), None, None, ptr(gam), ptr(bet), ptr(mean), ptr(invstd), 0, ptr(dx_b), None,
                                         ptr(dg_b), ptr(db_b), 0, M, C, ptr(ws), ws.numel() * 4, None, stream_ptr()), "bn bwd")
    for a, b in ((dx_a, dx_b), (dg_a, dg_b), (db_a, db_b)):
        assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))
    # the BatchNorm + ReLU form (bn2 -> conv3): y = (bn(bx) > 0) ? dX : 0 with the mask recomputed from the BatchNorm's input, and the
    # same partial sums
    z = ((bx - mean) * invstd) * gam + 0.3
    bet.fill_(0.3)
    dxr = torch.nn.grad.conv2d_input((N, Cin, H, W), w, dy.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).to(dev)
    want = torch.where(z > 0, dxr, torch.zeros((), device=dev))
    part2 = torch.empty_like(part)
    got = nn_ops.conv_bwd_data(dy.to(dev), pct, (N, H, W, Cin), 1, 0, bn_relu=(bx, mean, invstd, gam, bet, part2))
    assert float((got - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))
    dx_c, dg_c, db_c = torch.empty_like(bx), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    check(lib.ccst_bn_train_bwd_partials_f32(ptr(got), ptr(bx), ptr(gam), ptr(mean), ptr(invstd), ptr(part2), int(part2.shape[0]), ptr(dx_c),
                                             ptr(dg_c), ptr(db_c), 0, M, C, ptr(ws), ws.numel() * 4, None, stream_ptr()), "bn bwd partials")
    dx_d, dg_d, db_d = torch.empty_like(bx), torch.zeros(C, device=dev), torch.zeros(C, device=dev)       # plain: ReLU mask recomputed by the BN kernels
    check(lib.ccst_bn_train_bwd_mask_f32(ptr(dxr.contiguous()), ptr(bx), None, None, ptr(gam), ptr(bet), ptr(mean), ptr(invstd), 1, ptr(dx_d), None,
                                         ptr(dg_d), ptr(db_d), 0, M, C, ptr(ws), ws.numel() * 4, None, stream_ptr()), "bn bwd")
    for a, b in ((dx_c, dx_d), (dg_c, dg_d), (db_c, db_d)):
        assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))
    # shapes the streaming kernel does not take are refused by the predicate (3x3, stride 2)
    assert not nn_ops.masked_accum_ok(dy.to(dev), ops.pack_conv_weight(torch.randn(Cout, Cin, 3, 3).to(dev), transpose=True), (N, H, W, Cin), 1, 1)


def test_stem_conv_fwd_bwd(dev):
    from ccst_amd.nets import resnet
    x = rnd((2, 3, 38, 38), 4)
    w = rnd((64, 3, 7, 7), 5, 0.1)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(x, wr, stride=2, padding=3)
    g = rnd(tuple(yr.shape), 6)
    yr.backward(g)
    conv = resnet.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w)
    y = conv(x.to(dev))
    y.backward(g.to(dev))
    assert relerr(y, yr) < 1e-4 and relerr(conv.weight.grad, wr.grad) < 1e-4


@pytest.mark.parametrize("relu,res,bytemask", [(False, False, True), (True, False, True), (True, True, True), (True, True, False)])
def test_batchnorm_fwd_bwd(dev, relu, res, bytemask, monkeypatch):
    from ccst_amd import nn_ops
    from ccst_amd.nets import resnet
    monkeypatch.setattr(nn_ops, "BN_BYTE_MASK", bytemask)       # ReLU mask of a residual BN: the forward's byte mask / the saved output
    N, C, H, W = 4, 64, 9, 11
    x = rnd((N, C, H, W), 7, 2.0) + 0.5
    r = rnd((N, C, H, W), 8)
    gam, bet = rnd((C,), 9, 0.2) + 1.0, rnd((C,), 10, 0.1)
    ref = nn.BatchNorm2d(C)
    with torch.no_grad():
        ref.weight.copy_(gam)
        ref.bias.copy_(bet)
    xr, rr = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    yr = ref(xr)
    if res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    g = rnd(tuple(yr.shape), 11)
    yr.backward(g)
    bn = resnet.BatchNorm2d(C).to(dev)
    with torch.no_grad():
        bn.weight.copy_(gam)
        bn.bias.copy_(bet)
    xd, rd = cl(x, dev).requires_grad_(True), cl(r, dev).requires_grad_(True)
    y = bn(xd, residual=rd if res else None, relu=relu)
    y.backward(g.to(dev))
    assert relerr(y, yr) < 1e-5
    assert relerr(xd.grad, xr.grad) < 1e-4
    if res:
        assert relerr(rd.grad, rr.grad) < 1e-5
    assert relerr(bn.weight.grad, ref.weight.grad) < 1e-4 and relerr(bn.bias.grad, ref.bias.grad) < 1e-4
    assert relerr(bn.running_mean, ref.running_mean) < 1e-5 and relerr(bn.running_var, ref.running_var) < 1e-5
    bn.eval()
    ref.eval()
    with torch.no_grad():
        assert relerr(bn(x.to(dev), relu=relu), F.relu(ref(x)) if relu else ref(x)) < 1e-5


def test_fused_stem_bn_relu_maxpool(dev, monkeypatch):
    """The stem's bn1 -> relu -> maxpool as one op (no full-resolution normalised map, pooled gradient gathered inside both
    BatchNorm-backward passes) against the three separate ops: same output, same gradients, same running statistics."""
    from ccst_amd.nets import resnet
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(resnet, "FUSED_STEM", fused)
        torch.manual_seed(3)
        m = resnet.resnet18(types.SimpleNamespace(), pretrained=False, classes=7).to(dev).train()
        x = rnd((3, 3, 45, 39), 21).to(dev)
        y = m._stem(x)
        g = rnd(tuple(y.shape), 22).to(dev)
        y.backward(g)
        res[fused] = (y.detach().clone(), m.conv1.weight.grad.clone(), m.bn1.weight.grad.clone(), m.bn1.bias.grad.clone(),
                      m.bn1.running_mean.clone(), m.bn1.running_var.clone())
    for a, b in zip(res[True], res[False]):
        assert relerr(a, b) < 1e-6


def test_pools_linear_ce(dev):
    from ccst_amd import fed
    from ccst_amd.nets import resnet
    x = rnd((2, 64, 13, 15), 12)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    g = rnd(tuple(yr.shape), 13)
    yr.backward(g)
    xd = cl(x, dev).requires_grad_(True)
    y = resnet.MaxPool2d(3, 2, 1)(xd)
    y.backward(g.to(dev))
    assert relerr(y, yr) == 0.0 and relerr(xd.grad, xr.grad) < 1e-6
    # avgpool(7) + flatten + linear + CE
    f = rnd((5, 128, 7, 7), 14)
    lin = nn.Linear(128, 7)
    lab = torch.tensor([0, 3, 6, 2, 2])
    fr = f.clone().requires_grad_(True)
    lr_ = F.cross_entropy(lin(F.avg_pool2d(fr, 7).view(5, -1)), lab)
    lr_.backward()
    ours = resnet.Linear(128, 7).to(dev)
    ours.load_state_dict(lin.state_dict())
    fd = cl(f, dev).requires_grad_(True)
    ce = fed.CrossEntropyLoss()
    logit = ours(resnet.AvgPool2d(7, stride=1)(fd).view(5, -1))
    loss = ce(logit, lab.to(dev))
    loss.backward()
    assert abs(float(loss) - float(lr_)) < 1e-5
    assert relerr(fd.grad, fr.grad) < 1e-5 and relerr(ours.weight.grad, lin.weight.grad) < 1e-5
    assert relerr(ours.bias.grad, lin.bias.grad) < 1e-5
    assert int(ce.correct) == int((logit.argmax(1).cpu() == lab).sum())
    with pytest.raises(RuntimeError):
        resnet.AvgPool2d(7, stride=1)(torch.zeros(1, 8, 3, 3, device=dev))      # fed: image_size < 193


# Multiplier of the reference's OWN fp32-vs-fp64 difference that the gradient gates allow on top of 1e-3 of the tensor's largest
# gradient.  Two fp32 evaluations differ by the sum of their rounding errors (>= 2 x), and single elements sit behind ~50 ReLU /
# max-pool masks that a different rounding sequence flips.  Round 3 used 8 for both criteria; the default path's measured need is
# printed by test_resnet50_full_size_step_vs_oracle (round 4, VERDICT r3 #8: element-wise 5.49 on layer3.4.conv2.weight, 2-norm 1.12 on
# layer3.4.bn1.weight) and each gate is the smallest integer above its need plus a margin of one.
NOISE_MULT = 7.0          # element-wise criteria (and the fixtures' 16-element gradient probes)
NOISE_MULT_L2 = 3.0       # 2-norm of a whole gradient tensor


def _model_case(dev, g, arch):
    """One train step vs the reference's own outputs (tests/golden/resnet*_step.npz).  Logits (eval, train, post-step) and
    the loss are held to BASELINE.json's 1e-3 flat: the fixture's weights are conditioned like a trained net (small closing-BN
    gammas, oracle.resnet_ref.seeded_state_dict), so the reference's own fp32-vs-fp64 difference on them is <= 1e-5 and is
    asserted to be.  Gradient PROBES (single elements of conv1 / early-layer gradients behind ~50 ReLU / max-pool masks) keep a
    term scaled by the reference's own |fp32 - fp64| on the probed elements."""
    from ccst_amd import fed
    from ccst_amd.nets import models
    from oracle import resnet_ref as R
    seed, classes, nb, lr = int(g["seed"]), int(g["classes"]), int(g["nb"]), float(g["lr"])

    def tol(key):
        assert float(g["noise/" + key]) < 1e-4, ("fixture is ill-conditioned", key, float(g["noise/" + key]))
        return 1e-3

    model = models.get_network(arch)(ARGS, pretrained=False, classes=classes)
    oracle = R.resnet18(classes) if arch == "resnet18" else R.resnet50(classes)
    model.load_state_dict(R.seeded_state_dict(oracle, seed, float(g["residual_gamma"]), float(g["fc_gain"])))
    model.to(dev)
    x, y = R.synth_batch(nb, 222, classes, seed=seed + 1)
    x, y = x.to(dev), y.to(dev)
    model.eval()
    with torch.no_grad():
        d = float((model(x).cpu() - torch.from_numpy(g["logit_eval"])).abs().max())
    assert d < tol("logit_eval"), "eval logits %g" % d
    model.train()
    opt = fed.SGD(model, lr=lr)
    ce = fed.CrossEntropyLoss()
    opt.zero_grad()
    logit = model(x)
    loss = ce(logit, y)
    loss.backward()
    d = float((logit.detach().cpu() - torch.from_numpy(g["logit_train"])).abs().max())
    assert d < tol("logit_train"), "train logits %g" % d
    assert abs(float(loss.detach()) - float(g["loss"])) < tol("loss")
    named = dict(model.named_parameters())
    for k in g.files:
        if k.startswith("grad_abs/"):
            name = k[9:]
            got = float(named[name].grad.abs().sum())
            assert abs(got - float(g[k])) < 1e-3 * float(g[k]) + NOISE_MULT * float(g["noise/" + k]) + 1e-6, (name, got, float(g[k]))
        if k.startswith("grad_head/"):
            name = k[10:]
            ref = torch.from_numpy(g[k])
            got = named[name].grad.flatten()[:ref.numel()].cpu()
            scale = float(g["grad_abs/" + name]) / named[name].numel()
            d = float((got - ref).abs().max())
            # conditioning: 8x the reference's own fp32 error on these 16 elements, or 2x its worst fp32 error over the
            # whole tensor (the 16-element sample under-estimates it up to 36x: layer2.0.downsample.0.weight of resnet18
            # has 2.7e-5 on the head but 9.9e-4 over the tensor)
            noise = max(NOISE_MULT * float(g["noise/" + k]), 2.0 * float(g["noise_full/grad/" + name]))
            assert d < 1e-3 * max(scale, float(ref.abs().max())) + noise + 1e-6, (name, d)
    opt.step()
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("state/"):
            ref = torch.from_numpy(np.asarray(g[k]))
            got = sd[k[6:]].cpu()
            if ref.dtype == torch.int64:
                assert int(got) == int(ref)
            else:
                assert float((got - ref).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max())), k
    model.eval()
    with torch.no_grad():
        d = float((model(x).cpu() - torch.from_numpy(g["logit_after"])).abs().max())
    assert d < tol("logit_after"), "post-step logits %g (tol %g)" % (d, tol("logit_after"))


def test_resnet18_step_golden(dev, golden):
    _model_case(dev, golden("resnet18_step"), "resnet18")


def test_resnet50_step_golden(dev, golden):
    _model_case(dev, golden("resnet50_step"), "resnet50")


def test_resnet50_full_size_step_vs_oracle(dev):
    """BASELINE config 4 at its own size -- ResNet50, B=64, 222x222, SGD lr 0.001 (SURVEY 8d "Synthetic inputs 2") -- one
    whole train step against ONE step of the CPU oracle (pinned bit-for-bit to the reference's ResNet / train(), see
    tests/test_oracle_golden.py): train-mode logits, loss, accuracy count, ALL 161 gradient tensors, every BN's running
    statistics, every updated weight and the post-step eval-mode logits, all at 1e-3."""
    from ccst_amd import fed
    from ccst_amd.nets import models
    from oracle import resnet_ref as R
    torch.set_num_threads(__import__("conftest").host_cores())
    classes, nb, lr = 7, 64, 0.001
    oracle = R.resnet50(classes)
    sd = R.seeded_state_dict(oracle, 77, residual_gamma=0.25, fc_gain=8.0)
    oracle.load_state_dict(sd)
    x, y = R.synth_batch(nb, 222, classes, seed=78)
    loss_ref, logit_ref = R.train_step(oracle, x, y, lr)
    grads_ref = {k: p.grad.clone() for k, p in oracle.named_parameters()}          # (train_step leaves .grad in place)
    # what fp32 rounding alone costs on each gradient tensor: the oracle's own fp32 step against its fp64 step on these weights
    # and this batch (tools/make_resnet_grad_noise.py; ~10 minutes of CPU, hence a fixture)
    noise = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "resnet50_fullsize_grad_noise.npz"))
    oracle.eval()
    with torch.no_grad():
        after_ref = oracle(x[:16])
    model = models.get_network("resnet50")(ARGS, pretrained=False, classes=classes)
    model.load_state_dict(sd)
    model.to(dev).train()
    opt, ce = fed.SGD(model, lr=lr), fed.CrossEntropyLoss()
    opt.zero_grad()
    logit = model(x.to(dev))
    loss = ce(logit, y.to(dev))
    loss.backward()
    # every one of the 161 gradient tensors, element by element and in the 2-norm, against the oracle's: within 1e-3 of the
    # tensor's largest gradient (norm) plus NOISE_MULT (7; 2-norm: 3) x what the reference's OWN fp32 run loses against fp64 on that tensor -- two fp32
    # evaluations differ by the sum of their rounding errors, and single elements sit behind ~50 ReLU / max-pool masks.  (The
    # step's lr = 1e-3 would hide a gradient error of 1.0 in the weight comparison further down.)
    n_checked, worst = 0, (0.0, "")
    need_max, need_l2 = (0.0, ""), (0.0, "")       # the noise multiplier each criterion actually needs (printed; the gate below is NOISE_MULT)
    for k, prm in model.named_parameters():
        g32 = grads_ref[k]
        dg = prm.grad.detach().cpu() - g32
        d_max, d_l2 = float(dg.abs().max()), float(dg.double().norm())
        gmax, gl2 = float(noise["gmax/" + k]), float(noise["g_l2/" + k])
        assert abs(float(g32.abs().max()) - gmax) <= 1e-3 * gmax + float(noise["noise_max/" + k]), k    # the fixture is of THIS step
        need_max = max(need_max, ((d_max - 1e-3 * gmax) / max(float(noise["noise_max/" + k]), 1e-30), k))
        need_l2 = max(need_l2, ((d_l2 - 1e-3 * gl2) / max(float(noise["noise_l2/" + k]), 1e-30), k))
        assert d_max <= 1e-3 * gmax + NOISE_MULT * float(noise["noise_max/" + k]), (k, d_max, float(noise["noise_max/" + k]), gmax)
        assert d_l2 <= 1e-3 * gl2 + NOISE_MULT_L2 * float(noise["noise_l2/" + k]), (k, d_l2, float(noise["noise_l2/" + k]), gl2)
        worst = max(worst, (d_l2 / gl2, k))
        n_checked += 1
    print("worst relative 2-norm gradient difference: %.3g (%s)" % worst)
    print("noise multiplier needed beyond 1e-3 of the tensor's largest gradient: element-wise %.2f (%s), 2-norm %.2f (%s); gate %.1f" % (
        need_max + need_l2 + (NOISE_MULT,)) + " / %.1f" % NOISE_MULT_L2)
    assert n_checked == len(grads_ref) == 161
    opt.step()
    assert float(logit_ref.abs().max()) > 0.5                                      # the 1e-3 below is not vacuous
    assert float((logit.detach().cpu() - logit_ref).abs().max()) < 1e-3
    assert abs(float(loss.detach()) - float(loss_ref)) < 1e-3
    assert int(ce.correct) == int((logit_ref.argmax(1) == y).sum())
    osd, msd = oracle.state_dict(), model.state_dict()
    for k, v in osd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert float((msd[k].cpu() - v).abs().max()) < 1e-3 * max(1.0, float(v.abs().max())), k
    worst = max(float((msd[k].cpu() - v).abs().max()) for k, v in osd.items() if v.dtype == torch.float32)
    assert worst < 1e-3, worst                                                     # every updated weight
    model.eval()
    with torch.no_grad():
        after = model(x[:16].to(dev))
    assert float((after.cpu() - after_ref).abs().max()) < 1e-3


def test_communication_golden(dev, golden):
    from ccst_amd import fed
    from ccst_amd.nets import resnet
    from oracle import resnet_ref as R
    g = golden("communication")
    server = resnet.ResNet(resnet.BasicBlock, [1, 1, 1, 1], classes=3)
    server.load_state_dict(R.seeded_state_dict(R.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3), int(g["seed"])))
    clients = [copy.deepcopy(server) for _ in range(3)]
    for ci, c in enumerate(clients):
        rs = np.random.RandomState(71 + ci)
        with torch.no_grad():
            for k, v in c.state_dict().items():
                if "num_batches_tracked" in k:
                    v.fill_(5 + ci)
                else:
                    v += torch.from_numpy(rs.normal(0, 0.02, tuple(v.shape)).astype(np.float32))
    server.to(dev)
    clients = [c.to(dev) for c in clients]
    server, clients = fed.communication(ARGS, server, clients, [float(w) for w in g["weights"]])
    keys = list(server.state_dict().keys())
    assert keys == [str(k) for k in g["keys"]]
    ksum = np.array([float(server.state_dict()[k].double().sum()) for k in keys])
    kabs = np.array([float(server.state_dict()[k].double().abs().sum()) for k in keys])
    assert np.allclose(kabs, g["key_abs"], rtol=1e-5) and np.allclose(ksum, g["key_sum"], rtol=1e-4, atol=1e-3)
    assert float((server.state_dict()["conv1.weight"].flatten()[:32].cpu() - torch.from_numpy(g["conv1_head"])).abs().max()) < 1e-6
    nbt = [k for k in keys if "num_batches_tracked" in k]
    assert [int(server.state_dict()[k]) for k in nbt] == list(g["nbt_server"])
    for ci, c in enumerate(clients):
        assert [int(c.state_dict()[k]) for k in nbt] == list(g["nbt_clients"][ci])
        for k in keys:
            if "num_batches_tracked" not in k:
                assert torch.equal(c.state_dict()[k], server.state_dict()[k])


def test_communication_fedbn_golden(dev, golden):
    """--mode fedbn through the flat arenas: server = full average, clients keep keys containing 'bn'."""
    import types
    from ccst_amd import fed
    from ccst_amd.nets import resnet
    from oracle import resnet_ref as R
    g = golden("communication_fedbn")
    server = resnet.ResNet(resnet.BasicBlock, [1, 1, 1, 1], classes=3)
    server.load_state_dict(R.seeded_state_dict(R.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3), int(g["seed"])))
    clients = [copy.deepcopy(server) for _ in range(3)]
    for ci, c in enumerate(clients):
        rs = np.random.RandomState(71 + ci)
        with torch.no_grad():
            for k, v in c.state_dict().items():
                if "num_batches_tracked" in k:
                    v.fill_(5 + ci)
                else:
                    v += torch.from_numpy(rs.normal(0, 0.02, tuple(v.shape)).astype(np.float32))
    server.to(dev)
    clients = [c.to(dev) for c in clients]
    server, clients = fed.communication(types.SimpleNamespace(mode="fedbn"), server, clients, [float(w) for w in g["weights"]])
    fkeys = [str(k) for k in g["keys"]]
    ssum = np.array([float(server.state_dict()[k].double().sum()) for k in fkeys])
    sabs = np.array([float(server.state_dict()[k].double().abs().sum()) for k in fkeys])
    assert np.allclose(sabs, g["server_abs"], rtol=1e-5) and np.allclose(ssum, g["server_sum"], rtol=1e-4, atol=1e-3)
    for ci, c in enumerate(clients):
        cabs = np.array([float(c.state_dict()[k].double().abs().sum()) for k in fkeys])
        assert np.allclose(cabs, g["client_abs"][ci], rtol=1e-5)
        for k, sh in zip(fkeys, g["shared"]):
            assert torch.equal(c.state_dict()[k], server.state_dict()[k]) == bool(sh), k
    assert float((clients[1].state_dict()["bn1.weight"].flatten()[:16].cpu() - torch.from_numpy(g["bn1_weight_client1"])).abs().max()) == 0.0
    nbt = [k for k in server.state_dict().keys() if "num_batches_tracked" in k]
    assert [int(server.state_dict()[k]) for k in nbt] == list(g["nbt_server"])


def test_train_and_test_loops(dev, golden):
    """train()/test() vs THE REFERENCE'S OWN train()/test() (fed_run.py:31-88,214-259, run by tools/make_golden.py ->
    tests/golden/fed_loop.npz): two epochs over a ragged 3-batch loader with one optimiser, a held-out test pass and
    a test pass over the training batches; per-iteration logger rows, epoch returns and the post-epoch state."""
    from ccst_amd import fed
    from ccst_amd.nets import resnet
    from oracle import resnet_ref as R
    g = golden("fed_loop")
    classes = 3
    ours = resnet.ResNet(resnet.BasicBlock, [1, 1, 1, 1], classes=classes)
    ours.load_state_dict(R.seeded_state_dict(R.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=classes), int(g["seed"])))
    train_loader = [R.synth_batch(int(n), 222, classes, seed=int(s)) for s, n in zip(g["train_seeds"], g["train_sizes"])]
    test_loader = [R.synth_batch(int(n), 222, classes, seed=int(s)) for s, n in zip(g["test_seeds"], g["test_sizes"])]

    class Log(object):
        rows = []

        def log(self, it, iters, losses, samples_right, total_samples):
            self.rows.append((it, iters, float(losses["train_loss"]), int(samples_right["class_acc"]), int(total_samples)))

    logger = Log()
    ours.to(dev)
    opt = fed.SGD(ours, lr=float(g["lr"]))                       # one optimiser for both epochs, as fed_run.py:657-664
    ce = fed.CrossEntropyLoss()
    tr1 = fed.train(ours, train_loader, opt, ce, 3, dev, ARGS, 0, logger)
    te1 = fed.test(ours, test_loader, ce, dev, ARGS)
    tr2 = fed.train(ours, train_loader, opt, ce, 3, dev, ARGS, 1, logger)
    te2 = fed.test(ours, train_loader, ce, dev, ARGS)
    for got, key in ((tr1, "train1"), (te1, "test1"), (tr2, "train2"), (te2, "test2")):
        assert abs(got[0] - float(g[key][0])) < 1e-3, (key, got, g[key])
        assert got[1] == float(g[key][1]), (key, got, g[key])                 # accuracies are exact ratios
    assert [r[0] for r in logger.rows] == list(g["log_it"]) and [r[1] for r in logger.rows] == list(g["log_iters"])
    assert [r[3] for r in logger.rows] == list(g["log_right"]) and [r[4] for r in logger.rows] == list(g["log_total"])
    assert np.abs(np.array([r[2] for r in logger.rows]) - g["log_loss"]).max() < 1e-3
    sd = ours.state_dict()
    for key, name in (("conv1_head", "conv1.weight"), ("fc_weight", "class_classifier.weight"), ("fc_bias", "class_classifier.bias"),
                      ("bn1_running_mean", "bn1.running_mean"), ("bn1_running_var", "bn1.running_var"),
                      ("l4_bn2_weight", "layer4.0.bn2.weight")):
        ref = torch.from_numpy(g[key]).flatten()
        got = sd[name].flatten()[:ref.numel()].cpu()
        assert float((got - ref).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max())), key
    assert int(sd["bn1.num_batches_tracked"]) == int(g["nbt"])
    kabs = np.array([float(v.double().abs().sum()) for v in sd.values()])
    assert np.allclose(kabs, g["key_abs"], rtol=1e-3)


def test_offload_models_round_trip_keeps_training(dev):
    """args.offload_models=True reproduces fed_run.py:85,258 (model.to('cpu') after every train()/test()).  Every parameter tensor is
    replaced on each round trip; the optimiser created BEFORE the moves (fed_run.py:657) must keep updating the model the
    modules hold (it used to update a stale arena).  Result: bit-identical to the resident run."""
    from ccst_amd import fed
    from ccst_amd.nets import resnet
    from oracle import resnet_ref as R
    loader = [R.synth_batch(4, 222, 3, seed=500 + i) for i in range(2)]
    out = {}
    for offload in (False, True):
        args = types.SimpleNamespace(dg_method="", mode="fedavg", offload_models=offload)
        m = resnet.ResNet(resnet.BasicBlock, [1, 1, 1, 1], classes=3)
        m.load_state_dict(R.seeded_state_dict(R.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3), 92))
        opt = fed.SGD(m, lr=0.01)                       # model still on the CPU here, as in fed_run.py:657
        ce = fed.CrossEntropyLoss()
        r = [fed.train(m, loader, opt, ce, 1, dev, args, 0, None), fed.test(m, loader, ce, dev, args),
             fed.train(m, loader, opt, ce, 1, dev, args, 1, None)]
        assert next(m.parameters()).device.type == ("cpu" if offload else "cuda")
        out[offload] = (r, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    assert out[False][0] == out[True][0]
    assert out[True][0][2][0] != out[True][0][0][0]                    # the second epoch saw updated weights
    for k, v in out[False][1].items():
        assert torch.equal(v, out[True][1][k]), k


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_gradient_shortcuts_on_vs_off(dev, arch):
    """The gradient short-cuts between blocks -- GradSink (the first conv's backward-data adds to the identity branch's gradient in
    place), MaskLink (the closing BatchNorm's ReLU mask applied by the next block's conv epilogue) and its partial-sum variant --
    against the same step with all of them off (autograd's own add, masks in the BatchNorm backward): every parameter gradient
    within 2e-5 of the tensor's largest entry (the two orders of summation differ by fp32 rounding only).  And the reader
    contract: a tensor hook on a block output switches the short-cuts off for the block that reads it, so the hook
    sees the complete gradient."""
    from ccst_amd import fed, nn_ops
    from ccst_amd.nets import models, resnet
    from oracle import resnet_ref as R
    classes, nb = 7, 4
    sd = R.seeded_state_dict(R.resnet18(classes) if arch == "resnet18" else R.resnet50(classes), 55, residual_gamma=0.25, fc_gain=8.0)
    x, y = R.synth_batch(nb, 222, classes, seed=56)
    x, y = x.to(dev), y.to(dev)
    saved = (nn_ops.MASK_LINK, nn_ops.MASK_LINK_STATS, resnet.USE_GRAD_SINK)

    def run(on, tap=False):
        nn_ops.MASK_LINK, nn_ops.MASK_LINK_STATS, resnet.USE_GRAD_SINK = on, on, on
        model = models.get_network(arch)(ARGS, pretrained=False, classes=classes)
        model.load_state_dict(sd)
        model.to(dev).train()
        seen = {}
        if tap:     # a feature tap on layer1's output, registered the way DG methods do it: from a module forward hook
            def tap_hook(_m, _i, o):
                o.register_hook(lambda g: seen.__setitem__("g", g.detach().clone()))       # (returns None: the output stays the output)
            model.layer1.register_forward_hook(tap_hook)
        ce = fed.CrossEntropyLoss()
        model.zero_grad()
        loss = ce(model(x), y)
        loss.backward()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters()}, seen.get("g")
    try:
        g_on, _ = run(True)
        g_off, _ = run(False)
        g_tap, seen_on = run(True, tap=True)
        _, seen_off = run(False, tap=True)
    finally:
        nn_ops.MASK_LINK, nn_ops.MASK_LINK_STATS, resnet.USE_GRAD_SINK = saved
    for k in g_off:
        tol = 2e-5 * float(g_off[k].abs().max()) + 1e-9
        assert float((g_on[k] - g_off[k]).abs().max()) <= tol, k
        assert float((g_tap[k] - g_off[k]).abs().max()) <= tol, k
    assert seen_on is not None and seen_off is not None
    assert float((seen_on - seen_off).abs().max()) <= 2e-5 * float(seen_off.abs().max())      # complete and unmasked, as autograd defines it


def test_offload_models_with_hip_graph_recaptures(dev):
    """with --hip_graph the captured train steps hold raw addresses of the parameter arena.  args.offload_models moves
    the model to the CPU and back between train() calls, so the second call must NOT replay the first call's graphs (they would
    read and write freed memory and leave the live tensors untrained): the arena is re-resolved before the cache is read, the
    cache dropped with it, and the result stays bit-identical to the resident eager run."""
    from ccst_amd import fed
    from ccst_amd.nets import resnet
    from oracle import resnet_ref as R
    loader = [R.synth_batch(4, 222, 3, seed=600 + i) for i in range(5)]       # 2 eager iterations, then capture + replays
    out = {}
    for offload, graph in ((False, False), (True, True), (False, True)):
        args = types.SimpleNamespace(dg_method="", mode="fedavg", offload_models=offload, hip_graph=graph)
        m = resnet.ResNet(resnet.BasicBlock, [1, 1, 1, 1], classes=3)
        m.load_state_dict(R.seeded_state_dict(R.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3), 93))
        opt = fed.SGD(m, lr=0.01)
        ce = fed.CrossEntropyLoss()
        r = [fed.train(m, loader, opt, ce, 1, dev, args, 0, None), fed.train(m, loader, opt, ce, 1, dev, args, 1, None)]
        if graph and not offload:
            assert m.__dict__.get("_ccst_graph_steps"), "the resident model keeps its captured steps across train() calls"
        out[(offload, graph)] = (r, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    ref = out[(False, False)]
    assert ref[0][1][0] != ref[0][0][0]                                        # the second epoch saw updated weights
    for key in ((True, True), (False, True)):
        assert out[key][0] == ref[0], key
        for k, v in ref[1].items():
            assert torch.equal(v, out[key][1][k]), (key, k)


def test_test_fedbn_merge_and_eval(dev):
    """fed.test_fedbn vs the reference's merge (fed_run.py:350-362, restated in oracle.fed_ref.test_fedbn_merge) + eval loop."""
    from ccst_amd import fed
    from ccst_amd.nets import resnet
    from oracle import fed_ref, resnet_ref as R

    def make(cls, blk):
        server = cls(blk, [1, 1, 1, 1], classes=3)
        server.load_state_dict(R.seeded_state_dict(R.ResNet(R.BasicBlock, [1, 1, 1, 1], classes=3), 70))
        clients = [copy.deepcopy(server) for _ in range(3)]
        for ci, c in enumerate(clients):
            rs = np.random.RandomState(81 + ci)
            with torch.no_grad():
                for k, v in c.state_dict().items():
                    if "num_batches_tracked" in k:
                        v.fill_(5 + ci)
                    elif "running_var" in k:
                        v += torch.from_numpy(rs.uniform(0, 0.2, tuple(v.shape)).astype(np.float32))
                    else:
                        v += torch.from_numpy(rs.normal(0, 0.02, tuple(v.shape)).astype(np.float32))
        return server, clients
    loader = [R.synth_batch(4, 222, 3, seed=600 + i) for i in range(2)]
    rs_, rc = make(R.ResNet, R.BasicBlock)
    fed_ref.test_fedbn_merge(rs_, rc)
    ref = fed_ref.test_epoch(rs_, loader, nn.CrossEntropyLoss())
    server, clients = make(resnet.ResNet, resnet.BasicBlock)
    got = fed.test_fedbn(server, clients, loader, fed.CrossEntropyLoss(), dev, ARGS)
    for k, v in rs_.state_dict().items():
        d = (server.state_dict()[k].cpu().double() - v.double()).abs().max()
        assert float(d) <= 1e-6 * max(1.0, float(v.double().abs().max())), k
    assert abs(got[0] - ref[0]) < 1e-3 and got[1] == ref[1]


def test_train_hip_graph_matches_eager(dev, monkeypatch):
    """fed.train(..., args.hip_graph=True): iterations 3.. are HIP-graph replays; same losses, accuracies and -- bit for
    bit -- the same weights / BN statistics as the eager loop, across two calls with a FedAvg-style weight rewrite between."""
    import types
    from ccst_amd import fed, ops
    from ccst_amd.nets import models
    from oracle import resnet_ref as R
    xs = [R.synth_batch(4, 222, 7, seed=300 + i) for i in range(5)] + [R.synth_batch(3, 222, 7, seed=400)]   # ragged last batch
    loader = [(x, y) for x, y in xs]
    out = {}
    monkeypatch.setattr(fed, "AUTO_PROBE", 1)      # "auto": 2 warm + 1 timed eager iterations, capture, 1 + 1 timed replays (spans both calls)
    for mode in (False, True, "auto"):
        args = types.SimpleNamespace(mode="fedavg", dg_method="no_DG", hip_graph=mode)
        model = models.get_network("resnet18")(args, pretrained=False, classes=7)
        model.load_state_dict(R.seeded_state_dict(R.resnet18(7), 11))
        model.to(dev)
        ce = fed.CrossEntropyLoss()
        r1 = fed.train(model, loader, fed.SGD(model, lr=0.01), ce, 1, dev, args, 0, None)
        with torch.no_grad():                                  # weights rewritten outside the graph, as communication() does
            fed.FlatParams.of(model).flat.mul_(0.999)
            ops.bump_weights_epoch()
        r2 = fed.train(model, loader, fed.SGD(model, lr=0.01), ce, 1, dev, args, 1, None)
        r3 = fed.test(model, loader, ce, dev, args)
        out[mode] = (r1, r2, r3, {k: v.detach().clone() for k, v in model.state_dict().items()})
        if mode is True:
            assert len([k for k in model.__dict__["_ccst_graph_steps"] if k not in ("_sums", "_auto")]) == 2        # full batch + the ragged last batch
        if mode == "auto":          # both loops were timed for the full-batch shape, and a decision exists
            st = [v for v in model.__dict__["_ccst_graph_steps"]["_auto"].values() if v["graph_ms"] is not None]
            assert len(st) == 1 and st[0]["eager_ms"] > 0 and st[0]["graph_ms"] > 0, model.__dict__["_ccst_graph_steps"]["_auto"]
    for mode in (True, "auto"):
        for a, b in zip(out[False][:3], out[mode][:3]):
            assert abs(a[0] - b[0]) < 1e-6 * abs(a[0]) and a[1] == b[1], (mode, a, b)      # the running loss is summed per graph, then added
        for k, v in out[False][3].items():
            assert torch.equal(v, out[mode][3][k]), (mode, k)


def test_fed_run_cli_checkpoint_resume_test(dev, tmp_path):
    """federated/fed_run.py drop-in end to end on synthetic clients: console lines, checkpoint dict
    {'server_model','a_iter'} (best + _latest), --resume and --test (fed_run.py:582-640,733-766)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    base = [sys.executable, os.path.join(root, "federated", "fed_run.py"), "--mode", "fedavg", "--fusion_mode", "adain-overall-K3",
            "--source", "art_painting", "cartoon", "sketch", "--target", "photo", "--n_classes", "7", "--network", "resnet18",
            "--lr", "0.001", "--image_size", "222", "--batch", "4", "--synthetic", "8", "--save_path", str(tmp_path / "ckpt")]
    out = subprocess.check_output(base + ["--iters", "2"], cwd=str(tmp_path), env=env, text=True)
    assert "=============Global iter is 1 ===============" in out and "| Train Loss:" in out and "| Global Test Class Acc:" in out
    d = tmp_path / "ckpt" / "pacs" / "fedavg_adain-overall-K3_no_DG_resnet18_locIter1" / "Target_photo_seed_1"
    latest = torch.load(str(d / "fedavg_latest"), map_location="cpu")
    assert set(latest.keys()) == {"server_model", "a_iter"} and int(latest["a_iter"]) == 1
    from oracle import resnet_ref as R
    assert list(latest["server_model"].keys()) == list(R.resnet18(7).state_dict().keys())
    if not (d / "fedavg").exists():       # the best checkpoint is only written when val acc improves (strict >, fed_run.py:749)
        import shutil
        shutil.copy(str(d / "fedavg_latest"), str(d / "fedavg"))
    out2 = subprocess.check_output(base + ["--iters", "3", "--resume"], cwd=str(tmp_path), env=env, text=True)
    assert "Resume training from epoch 2" in out2 and "Global iter is 2" in out2 and "Global iter is 1 " not in out2
    out3 = subprocess.check_output(base + ["--test"], cwd=str(tmp_path), env=env, text=True)
    assert "| Test  Acc:" in out3


def test_fed_run_cli_fedbn(dev, tmp_path):
    """--mode fedbn end to end: clients validated with their local model, checkpoint carries 'model_{k}' dicts whose
    'bn' entries differ from the server's while the shared entries are identical (fed_run.py:388-399,735-759)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    cmd = [sys.executable, os.path.join(root, "federated", "fed_run.py"), "--mode", "fedbn", "--fusion_mode", "adain-overall-K3",
           "--source", "art_painting", "cartoon", "sketch", "--target", "photo", "--n_classes", "7", "--network", "resnet18",
           "--lr", "0.001", "--image_size", "222", "--batch", "4", "--synthetic", "8", "--save_path", str(tmp_path / "ckpt"),
           "--iters", "2"]
    out = subprocess.check_output(cmd, cwd=str(tmp_path), env=env, text=True)
    assert "| Global Val Class Acc:" in out and "| Global Test Class Acc:" in out
    d = tmp_path / "ckpt" / "pacs" / "fedbn_adain-overall-K3_no_DG_resnet18_locIter1" / "Target_photo_seed_1"
    ck = torch.load(str(d / "fedbn_latest"), map_location="cpu")
    assert set(ck.keys()) == {"server_model", "a_iter", "model_0", "model_1", "model_2"}
    srv, m1 = ck["server_model"], ck["model_1"]
    assert torch.equal(srv["conv1.weight"], m1["conv1.weight"]) and torch.equal(srv["layer2.0.downsample.1.weight"], m1["layer2.0.downsample.1.weight"])
    assert not torch.equal(srv["bn1.running_mean"], m1["bn1.running_mean"])
    # --resume: every client continues from ITS model_k; --test: test_fedbn (fed_run.py:585-590)
    if not (d / "fedbn").exists():
        import shutil
        shutil.copy(str(d / "fedbn_latest"), str(d / "fedbn"))
    out2 = subprocess.check_output(cmd[:-1] + ["3", "--resume"], cwd=str(tmp_path), env=env, text=True)
    assert "Resume training from epoch 2" in out2
    out3 = subprocess.check_output(cmd + ["--test"], cwd=str(tmp_path), env=env, text=True)
    assert "| Test  Acc:" in out3 and out3.strip().splitlines()[-1].startswith(" sketch")


def test_fed_run_cli_deepall(dev, tmp_path):
    """--mode deepall (data/data_helper.py:66-121, fed_run.py:570-575): ONE client trained on every source's data."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    cmd = [sys.executable, os.path.join(root, "federated", "fed_run.py"), "--mode", "deepall", "--fusion_mode", "no_fusion",
           "--source", "art_painting", "cartoon", "sketch", "--target", "photo", "--n_classes", "7", "--network", "resnet18",
           "--lr", "0.001", "--image_size", "222", "--batch", "4", "--synthetic", "8", "--save_path", str(tmp_path / "ckpt"),
           "--iters", "2"]
    out = subprocess.check_output(cmd, cwd=str(tmp_path), env=env, text=True)
    assert out.count("| Train Loss:") == 2 and out.count("| Global Val Class Acc:") == 2          # one client per round


def test_train_step_bitwise_reproducible(dev):
    """No float atomics anywhere on the path (split partials + fixed-order reduces): the same step from the same
    state gives bit-identical logits, loss, gradients and updated weights."""
    from ccst_amd import fed
    from ccst_amd.nets import resnet
    from oracle import resnet_ref as R

    def run():
        m = resnet.ResNet(resnet.Bottleneck, [1, 1, 1, 1], classes=5)
        m.load_state_dict(R.seeded_state_dict(R.ResNet(R.Bottleneck, [1, 1, 1, 1], classes=5), 123))
        m.to(dev).train()
        opt = fed.SGD(m, lr=0.01)
        ce = fed.CrossEntropyLoss()
        x, y = R.synth_batch(6, 222, 5, seed=124)
        outs = []
        for _ in range(2):
            opt.zero_grad()
            logit = m(x.to(dev))
            loss = ce(logit, y.to(dev))
            loss.backward()
            outs += [logit.detach().clone(), loss.detach().clone(), fed.FlatParams.of(m).grad.clone()]
            opt.step()
        outs.append(fed.FlatParams.of(m).flat.clone())
        torch.cuda.synchronize()
        return outs

    a, b = run(), run()
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize("shape", [(3, 28, 28, 128, 128), (2, 56, 56, 64, 64), (2, 14, 14, 256, 192), (1, 13, 19, 32, 48)])
def test_wino_train_form_vs_gather(dev, shape):
    """ccst_conv3x3_wino_train_f32: forward + BN statistics, backward-data through the transposed / flipped weight transform,
    and y += conv, against the gather kernel."""
    from ccst_amd import nn_ops, ops
    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(6)
    x = torch.randn(N, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
    dy = torch.randn(N, H, W, Cout, generator=g).to(dev)
    base = torch.randn(N, H, W, Cin, generator=g).to(dev)
    pc, pct = ops.pack_conv_weight(w), ops.pack_conv_weight(w, transpose=True)
    y_ref, st_ref = ops.conv2d_nhwc(x, pc, stride=1, pad=1, want_stats=True)
    dx_ref = nn_ops.conv_bwd_data(dy, pct, (N, H, W, Cin), 1, 1)
    y, st = ops.conv3x3_wino_train(x, ops.pack_wino(w), want_stats=True)
    assert float((y - y_ref).abs().max()) < 3e-5 * max(1.0, float(y_ref.abs().max()))
    tot, tot_ref = st.double().sum(0), st_ref.double().sum(0)
    assert float((tot - tot_ref).abs().max()) < 1e-4 * float(tot_ref.abs().max())
    dx = ops.conv3x3_wino_train(dy, ops.pack_wino(w, bwd=True))
    assert float((dx - dx_ref).abs().max()) < 3e-5 * max(1.0, float(dx_ref.abs().max()))
    acc = base.clone()
    ops.conv3x3_wino_train(dy, ops.pack_wino(w, bwd=True), accumulate_into=acc)
    assert float((acc - (base + dx_ref)).abs().max()) < 3e-5 * max(1.0, float(dx_ref.abs().max()))


@pytest.mark.parametrize("cfg", [
    # N, H, W, Cin, Cout, k, stride          (tiles: 128x128, 64x128, 128x64, 64x64; ragged M, Cin / Cout that do not fill a tile)
    (3, 14, 14, 256, 128, 3, 1), (2, 28, 28, 64, 256, 1, 1), (2, 28, 28, 256, 64, 1, 1), (2, 30, 26, 64, 64, 3, 1),
    (2, 29, 27, 128, 128, 3, 2), (5, 7, 7, 512, 192, 1, 1), (1, 9, 11, 96, 160, 3, 1), (2, 28, 28, 128, 256, 1, 2),
])
@pytest.mark.parametrize("gscale", [1.0, 1e-8, 1e6])
def test_conv_bwd_weight_half_pieces_vs_fp64(dev, cfg, gscale):
    """ccst_conv2d_bwd_weight_split_f32 (three half-piece products per fp32 product on the 16-bit MFMA, operands scaled by their |max|
    words) against an fp64 weight gradient: every tile shape, the gathered (3x3, strided) and the pointwise loader, zero padding
    through the buffer bounds, pixel counts and channel counts that leave tiles partly empty, gradients from 1e-8 to 1e6; and the
    fp32-MFMA kernel on the same operands for scale.  Bitwise reproducible."""
    import ctypes
    from ccst_amd import _lib, nn_ops, ops
    from ccst_amd._lib import check, ptr, stream_ptr
    N, H, W, Cin, Cout, k, stride = cfg
    pad = k // 2
    g = torch.Generator().manual_seed(41)
    x = torch.randn(N, H, W, Cin, generator=g).to(dev)
    d, ho, wo = nn_ops._fwd_desc(N, H, W, Cin, k, k, stride, pad, Cin, Cout, 0)
    dy = (torch.randn(N, ho, wo, Cout, generator=g) * gscale).to(dev)
    xp = torch.zeros(N, H + 2 * pad, W + 2 * pad, Cin, device=dev, dtype=torch.float64)
    xp[:, pad:pad + H, pad:pad + W] = x.double()
    d2 = dy.double().reshape(-1, Cout)
    ref = torch.empty(Cout, Cin, k, k, device=dev, dtype=torch.float64)
    for ky in range(k):
        for kx in range(k):
            xs = xp[:, ky:ky + (ho - 1) * stride + 1:stride, kx:kx + (wo - 1) * stride + 1:stride].reshape(-1, Cin)
            ref[:, :, ky, kx] = d2.t() @ xs
    lib = _lib.load()
    M = N * ho * wo
    out = []
    for rep in range(2):
        dw = torch.full((Cout, Cin, k, k), 7.0, device=dev)
        splits = lib.ccst_conv2d_bwd_weight_split_splits(M, Cin, Cout, k * k)
        ws = torch.empty(splits * k * k * Cin * Cout, device=dev)
        check(lib.ccst_conv2d_bwd_weight_split_f32(ctypes.byref(d), ptr(x), ptr(ops.absmax(x)), ptr(dy), ptr(ops.absmax(dy)), ptr(dw), splits, rep,
                                                   ptr(ws), ws.numel() * 4, stream_ptr()), "bwd_weight_split")
        out.append(dw)
    scale = float(ref.abs().max())
    err = float((out[0].double() - ref).abs().max()) / scale
    assert err < 2e-6, err
    assert float((out[1].double() - (ref + 7.0)).abs().max()) < 2e-6 * scale + 1e-5         # accumulate = 1 adds to what was there
    dw32 = torch.empty((Cout, Cin, k, k), device=dev)
    nn_ops.conv_bwd_weight(d, x, dy, dw32, accumulate=False)
    err32 = float((dw32.double() - ref).abs().max()) / scale
    assert err < 4 * err32 + 1e-7, (err, err32)


@pytest.mark.parametrize("half", [True, False])
def test_pointwise_conv_forward_vs_fp64(dev, half):
    """The streaming pointwise kernel's training forward -- every fp32 product as three products of IEEE-half pieces on the 16-bit MFMA
    (the default where the |max| words are at hand: activations split in the loader, weights pre-split by pack_conv_weight_split) or
    on the fp32 MFMA -- against an fp64 convolution, dense and strided (the downsample branches), with the statistics epilogue."""
    from ccst_amd import ops
    g = torch.Generator().manual_seed(5)
    worst = 0.0
    for (N, H, cin, cout, stride) in ((8, 56, 64, 128, 2), (8, 56, 64, 256, 1), (8, 14, 256, 512, 2), (4, 28, 128, 512, 1), (3, 7, 2048, 512, 1)):
        x = torch.randn(N, H, H, cin, generator=g).to(dev)
        w = (torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).to(dev)
        kw = {}
        if half:
            wmax = ops.absmax(w)
            kw = dict(x_absmax=ops.absmax(x), w_absmax=wmax, w_split=ops.pack_conv_weight_split(w, wmax))
        y, st = ops.conv2d_nhwc(x, ops.pack_conv_weight(w, None), stride=stride, pad=0, want_stats=True, **kw)
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), stride=stride).permute(0, 2, 3, 1)
        worst = max(worst, float((y - ref).abs().max() / ref.abs().max()))
        s_ref = torch.stack([ref.sum(dim=(0, 1, 2)), (ref * ref).sum(dim=(0, 1, 2))], dim=1)
        worst = max(worst, float((st.double().sum(0) - s_ref).abs().max() / s_ref.abs().max()))
    assert worst < 4e-6, worst


@pytest.mark.parametrize("gscale", [1.0, 1e-7, 3e5])
def test_pointwise_backward_data_half_pieces_match_fp32_forms(dev, gscale):
    """Backward-data of a pointwise conv on half pieces (ccst_conv2d_pointwise_half_f32: the gradient scaled by its |max| words, the
    transposed weight pre-split) in all four forms -- plain, y +=, masked accumulate with the BatchNorm link, BatchNorm + ReLU link --
    against the fp32-MFMA forms on the same operands, at gradient magnitudes from 1e-7 to 3e5."""
    from ccst_amd import nn_ops, ops
    g = torch.Generator().manual_seed(17)
    N, H, W, Cin, Cout = 4, 28, 28, 128, 256                # forward conv Cin -> Cout; dX [N,H,W,Cin]
    M = N * H * W
    dy = (torch.randn(N, H, W, Cout, generator=g) * gscale).to(dev)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).to(dev)
    pct = ops.pack_conv_weight(w, transpose=True)
    wmax = ops.absmax(w)
    half = (ops.absmax(dy), wmax, ops.pack_conv_weight_split(w, wmax, transpose=True))
    ref = (dy.double().reshape(M, Cout) @ w.double().reshape(Cout, Cin)).reshape(N, H, W, Cin)
    scale = float(ref.abs().max())
    tol = 4e-6 * scale
    # plain
    a = nn_ops.conv_bwd_data(dy, pct, (N, H, W, Cin), 1, 0, half=half)
    b = nn_ops.conv_bwd_data(dy, pct, (N, H, W, Cin), 1, 0)
    assert float((a.double() - ref).abs().max()) < tol and float((b.double() - ref).abs().max()) < tol
    # y +=
    base = (torch.randn(N, H, W, Cin, generator=g) * gscale).to(dev)
    a = nn_ops.conv_bwd_data(dy, pct, (N, H, W, Cin), 1, 0, accumulate_into=base.clone(), half=half)
    assert float((a.double() - (ref + base.double())).abs().max()) < tol
    # masked accumulate + BatchNorm link
    mask = (torch.rand(M * Cin // 4, generator=g) * 16).to(torch.uint8).to(dev)
    bx = torch.randn(N, H, W, Cin, generator=g).to(dev)
    mean, invstd = torch.randn(Cin, generator=g).to(dev) * 0.1, (torch.rand(Cin, generator=g) + 0.5).to(dev)
    groups = nn_ops.lib_groups(M, Cin, Cout)
    pa, pb = torch.empty((groups, Cin, 2), device=dev), torch.empty((groups, Cin, 2), device=dev)
    a = nn_ops.conv_bwd_data(dy, pct, (N, H, W, Cin), 1, 0, accumulate_into=base.clone(), relu_mask=mask, bn_link=(bx, mean, invstd, pa), half=half)
    b = nn_ops.conv_bwd_data(dy, pct, (N, H, W, Cin), 1, 0, accumulate_into=base.clone(), relu_mask=mask, bn_link=(bx, mean, invstd, pb))
    assert float((a - b).abs().max()) < 2 * tol
    assert float((pa.double().sum(0) - pb.double().sum(0)).abs().max()) < 1e-4 * max(float(pb.double().sum(0).abs().max()), 1e-30)
    # BatchNorm + ReLU link
    gam, bet = (torch.rand(Cin, generator=g) + 0.5).to(dev), (torch.randn(Cin, generator=g) * 0.3).to(dev)
    a = nn_ops.conv_bwd_data(dy, pct, (N, H, W, Cin), 1, 0, bn_relu=(bx, mean, invstd, gam, bet, pa), half=half)
    b = nn_ops.conv_bwd_data(dy, pct, (N, H, W, Cin), 1, 0, bn_relu=(bx, mean, invstd, gam, bet, pb))
    assert float((a - b).abs().max()) < 2 * tol
    assert float((pa.double().sum(0) - pb.double().sum(0)).abs().max()) < 1e-4 * max(float(pb.double().sum(0).abs().max()), 1e-30)


@pytest.mark.parametrize("cfg", [
    # N, H, W, Cin, Cout, k, stride, pad            (forward conv; dX [N,H,W,Cin])
    (3, 28, 28, 128, 128, 3, 2, 1), (2, 15, 13, 64, 128, 3, 2, 1), (4, 28, 28, 256, 512, 1, 2, 0), (2, 14, 14, 96, 160, 3, 1, 1), (2, 9, 9, 64, 32, 3, 1, 1),
])
@pytest.mark.parametrize("gscale", [1.0, 1e-7])
def test_gather_backward_data_on_half_pieces_vs_fp64(dev, cfg, gscale):
    """ccst_conv2d_igemm_half_f32: backward-data through the gather GEMM on half pieces -- the parity classes of a stride-2 3x3 conv, a
    strided 1x1 downsample branch, stride-1 3x3 shapes no other half-piece kernel takes -- plain and accumulating, against fp64."""
    from ccst_amd import nn_ops, ops
    N, H, W, Cin, Cout, k, stride, pad = cfg
    g = torch.Generator().manual_seed(23)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(dev)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = (torch.randn(N, Ho, Wo, Cout, generator=g) * gscale).to(dev)
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w.double(), dy.permute(0, 3, 1, 2).double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    pct = ops.pack_conv_weight(w, transpose=True)
    wmax = ops.absmax(w)
    half = (ops.absmax(dy), wmax, ops.pack_conv_weight_split(w, wmax, transpose=True))
    tol = 4e-6 * float(ref.abs().max())
    a = nn_ops.conv_bwd_data(dy, pct, (N, H, W, Cin), stride, pad, half=half)
    b = nn_ops.conv_bwd_data(dy, pct, (N, H, W, Cin), stride, pad)
    assert float((a.double() - ref).abs().max()) < tol and float((b.double() - ref).abs().max()) < tol
    base = (torch.randn(N, H, W, Cin, generator=g) * gscale).to(dev)
    acc = nn_ops.conv_bwd_data(dy, pct, (N, H, W, Cin), stride, pad, accumulate_into=base.clone(), half=half)
    assert float((acc.double() - (ref + base.double())).abs().max()) < tol + 1e-6 * float(base.abs().max())


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_reference_initialisation_within_twice_the_references_own_rounding(dev, arch):
    """every other fixture conditions the weights like a trained net (closing-BN gammas x 0.25, classifier x 8).  This
    one uses the REFERENCE'S OWN initialisation (nets/resnet.py:149-154: kaiming-normal fan_out convolutions, every BatchNorm gamma = 1,
    beta = 0), on which the reference's own fp32 and fp64 runs differ by ~1e-3 in the logits -- so the gate is relative to that:
    train-mode logits and the loss within TWICE |oracle fp32 - oracle fp64| of the oracle's fp64 result (an fp32 implementation
    cannot be asked for less than the reference's own fp32 run delivers; twice = a different, not a worse, rounding sequence)."""
    import copy
    from ccst_amd import fed
    from ccst_amd.nets import models
    from oracle import resnet_ref as R
    torch.set_num_threads(__import__("conftest").host_cores())
    classes, nb = 7, 8
    torch.manual_seed(5)
    oracle = R.resnet18(classes) if arch == "resnet18" else R.resnet50(classes)      # the reference's initialisation
    sd = {k: v.clone() for k, v in oracle.state_dict().items()}
    gam = [v for k, v in sd.items() if k.endswith("bn2.weight") or k.endswith("bn3.weight")]
    assert gam and all(bool((g_ == 1).all()) for g_ in gam)                            # gamma = 1 everywhere, as the reference initialises
    x, y = R.synth_batch(nb, 222, classes, seed=9)
    o64 = copy.deepcopy(oracle).double().train()
    oracle.train()
    with torch.no_grad():
        l32, l64 = oracle(x), o64(x.double())
    loss32, loss64 = float(F.cross_entropy(l32, y)), float(F.cross_entropy(l64, y))
    n_logit, n_loss = float((l32.double() - l64).abs().max()), abs(loss32 - loss64)
    model = models.get_network(arch)(ARGS, pretrained=False, classes=classes)
    model.load_state_dict(sd)
    model.to(dev).train()
    ce = fed.CrossEntropyLoss()
    with torch.no_grad():
        logit = model(x.to(dev))
        loss = ce(logit, y.to(dev))
    d_logit, d_loss = float((logit.double().cpu() - l64).abs().max()), abs(float(loss) - loss64)
    print("%s, reference initialisation: max |logit| %.3g; reference fp32 vs fp64: logits %.3g, loss %.3g; HIP vs fp64: logits %.3g (%.2fx), loss %.3g (%.2fx)" % (
        arch, float(l64.abs().max()), n_logit, n_loss, d_logit, d_logit / max(n_logit, 1e-30), d_loss, d_loss / max(n_loss, 1e-30)))
    assert d_logit <= 2.0 * n_logit + 1e-6 * float(l64.abs().max()), (d_logit, n_logit)
    assert d_loss <= 2.0 * n_loss + 2e-6 * max(1.0, abs(loss64)), (d_loss, n_loss)
