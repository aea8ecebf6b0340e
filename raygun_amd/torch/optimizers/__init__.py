from .BaseDummyOptimizer import BaseDummyOptimizer
from .utils import get_base_optimizer
from .FusedAdam import FusedAdam
