"""cara_amd -- MI355X-native CaRA fine-tuning hot path (hand-written HIP for gfx950 behind a C ABI).

Public surface mirrors the reference's ``src/cara`` package: ``cara(config)`` installs the
Canonical-Polyadic adapters on a timm-named ViT container; the arithmetic lives in
``libcara_hip.so`` (see include/cara_hip.h).  There is no CPU or eager fallback.
"""
from .cara import cara, set_cara  # noqa: F401
from .vit import VisionTransformer, create_model  # noqa: F401
from ._lib import CaraError  # noqa: F401
from . import optim  # noqa: F401
