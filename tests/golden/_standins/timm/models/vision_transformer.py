from oracle.cara_oracle import Attention, Block, VisionTransformer  # noqa: F401
