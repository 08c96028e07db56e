from oracle.cara_oracle import Mlp  # noqa: F401
