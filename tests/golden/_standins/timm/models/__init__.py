from . import layers, vision_transformer  # noqa: F401
from oracle.cara_oracle import create_vit as create_model  # noqa: F401
