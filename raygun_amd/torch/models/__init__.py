from .BaseModel import BaseModel
from .FreezableModel import FreezableModel
from .CycleModel import CycleModel
from .MTLSDModel import MTLSDModel
