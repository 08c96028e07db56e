"""Import-only stand-in for timm.scheduler (dim_experiment.py:14); the recipe's scheduler is restated in cara_amd/recipe.py."""


class CosineLRScheduler:
    def __init__(self, *a, **k):
        raise RuntimeError("stand-in: not a scheduler implementation")
