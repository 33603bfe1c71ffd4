"""Drop-in for nets/model_factory.py:7-23 (legacy JiGen registry): get_network(name)(**kwargs).
The reference module cannot be imported (it targets a non-existent ``models`` package, :1-5); the
API is kept: same keys, same ValueError, constructors called with keyword arguments only."""
import types

from . import resnet
from .models import _out_of_scope

_ARGS = types.SimpleNamespace(dg_method="")


def _bind(fn):
    def ctor(**kwargs):
        return fn(_ARGS, **kwargs)
    return ctor


nets_map = {
    'caffenet': _out_of_scope('caffenet'),
    'alexnet': _out_of_scope('alexnet'),
    'resnet18': _bind(resnet.resnet18),
    'resnet50': _bind(resnet.resnet50),
    'lenet': _out_of_scope('lenet'),
}


def get_network(name):
    if name not in nets_map:
        raise ValueError('Name of network unknown %s' % name)

    def get_network_fn(**kwargs):
        return nets_map[name](**kwargs)

    return get_network_fn
