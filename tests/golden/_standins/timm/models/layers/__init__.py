from . import mlp  # noqa: F401
