"""Drop-in for nets/models.py:114-133: name -> constructor registry, get_network(name)(args, **kw)."""
from . import resnet


def _out_of_scope(name):
    def ctor(*a, **k):
        raise NotImplementedError("ccst_amd.nets: '%s' is outside the hot path this library implements "
                                  "(SURVEY.md section 2: only resnet18 / resnet50 are on it)" % name)
    return ctor


nets_map = {
    'resnet18': resnet.resnet18,
    'resnet18IN': _out_of_scope('resnet18IN'),
    'resnet50': resnet.resnet50,
    'DigitModel': _out_of_scope('DigitModel'),
    'densenet': _out_of_scope('densenet'),
}


def get_network(name):
    if name not in nets_map:
        raise ValueError('Name of network unknown %s' % name)

    def get_network_fn(args, **kwargs):
        return nets_map[name](args, **kwargs)

    return get_network_fn
