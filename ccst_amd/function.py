"""Drop-in for style_transfer/AdaIN/function.py (calc_mean_std :4-13,
adaptive_instance_normalization :16-24, adaIN_StyleStat_ContentFeat :26-33) on
HIP kernels.  Same names, argument meaning and AssertionErrors; tensors are
logical-NCHW fp32 CUDA tensors (NCHW-contiguous or channels_last)."""
from . import ops


def calc_mean_std(feat, eps=1e-5):
    # eps is a small value added to the variance to avoid divide-by-zero.
    size = feat.size()
    assert (len(size) == 4)
    return ops.calc_mean_std(feat, eps)


def adaptive_instance_normalization(content_feat, style_feat):
    assert (content_feat.size()[:2] == style_feat.size()[:2])
    style_mean, style_std = calc_mean_std(style_feat)
    return ops.adain(content_feat, style_mean, style_std)


def adaIN_StyleStat_ContentFeat(content_feat, style_stat):
    style_mean, style_std = style_stat
    return ops.adain(content_feat, style_mean, style_std)
