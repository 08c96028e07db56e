"""Import-only stand-in for wandb (dim_experiment.py:13 imports it at module level; logging is off on the recorded path)."""
