"""Stand-in for timm==0.4.12 exposing the names /root/reference/src/cara/cara.py type-checks
against (cara.py:110,147,157).  The classes are the oracle's restatement of timm's published
source.  Used ONLY by tests/golden/make_golden.py."""
from . import models  # noqa: F401
