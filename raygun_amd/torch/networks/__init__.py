"""Network architectures (mirror of raygun.torch.networks)."""
from .NLayerDiscriminator import NLayerDiscriminator
from .UNet import UNet
from .ResidualUNet import ResidualUNet
from .ResNet import ResNet
from .utils import *  # noqa: F401,F403
