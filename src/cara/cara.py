"""Import-path compatibility: the reference's users write ``from src.cara.cara import cara``
(/root/reference/image_classification/vit_cp.py:15, tests/test_cara.py:10).  The implementation
is cara_amd (HIP on MI355X)."""
from cara_amd.cara import cara, cp_attn, cp_mlp, set_cara  # noqa: F401
