"""Alias for the package's old directory name: importlib.import_module("mini-nbody_amd") gives `mini_nbody_amd`."""
import sys

import mini_nbody_amd

sys.modules[__name__] = mini_nbody_amd
