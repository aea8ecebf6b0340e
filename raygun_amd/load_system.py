"""Config -> System instance — mirror of raygun.load_system (reference: src/raygun/load_system.py:9-21): the system class is
looked up as <package>.<config["framework"]>.systems.<config["system"]>.<config["system"]> and built from the config path."""
from importlib import import_module

from .read_config import read_config


def load_system(config_path, checkpoint=None):
    config = read_config(config_path)
    System = getattr(import_module(".".join([__package__, config["framework"], "systems", config["system"]])), config["system"])
    system = System(config_path)
    # (the reference leaves `system.load_saved_model(checkpoint=checkpoint)` to the caller, load_system.py:19)
    return system
