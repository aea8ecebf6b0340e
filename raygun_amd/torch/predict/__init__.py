from .worker import worker  # noqa: F401
