from .driver import Trainer, train_step
