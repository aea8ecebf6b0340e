from .BaseSystem import BaseSystem
from .CycleGAN import CycleGAN
from .MTLSD import MTLSD
