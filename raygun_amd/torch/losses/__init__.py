from .GANLoss import GANLoss
from .BaseCompetentLoss import BaseCompetentLoss
from .LinkCycleLoss import LinkCycleLoss
from .SplitCycleLoss import SplitCycleLoss
from .WeightedMSELoss_MTLSD import WeightedMSELoss_MTLSD
