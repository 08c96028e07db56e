class Accuracy:   # import-only stand-in (see avalanche/__init__.py)
    def __init__(self, *a, **k):
        raise RuntimeError("stand-in: not a metric implementation")
