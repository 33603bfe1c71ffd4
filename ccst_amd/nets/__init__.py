"""Drop-in for the reference's ``nets`` package (nets/models.py registry, nets/resnet.py models)."""
from . import resnet  # noqa: F401
from .models import get_network, nets_map  # noqa: F401
