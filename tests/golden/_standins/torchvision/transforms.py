# import-only stand-in (see torchvision/__init__.py)
