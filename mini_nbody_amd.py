"""Alias so that `import mini_nbody_amd` works: the package directory is `mini-nbody_amd/` (hyphen)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("mini-nbody_amd")
