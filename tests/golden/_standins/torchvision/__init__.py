"""Import-only stand-in: image_classification/vtab.py:6 does `from torchvision import transforms` at module level (the
data loader is not on the recorded path)."""
from . import transforms  # noqa: F401
