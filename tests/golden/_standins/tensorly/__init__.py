"""Stand-in for tensorly==0.8.1 exposing only what /root/reference/src/cara/cara.py uses
(``set_backend`` at cara.py:10, ``cp_to_tensor`` at cara.py:27,52,76,88).  Build-own code:
the arithmetic is the oracle's restatement of tensorly's published definition.  Used ONLY by
tests/golden/make_golden.py in the build container; never shipped in the product path."""
from oracle.cara_oracle import cp_to_tensor  # noqa: F401


def set_backend(name):
    assert name == "pytorch"
