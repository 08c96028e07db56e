"""Import-only stand-in: /root/reference/image_classification/dim_experiment.py:7 imports avalanche's Accuracy metric at
module level; the adapter code recorded by tests/golden/make_golden.py never touches it.  Used ONLY there."""
