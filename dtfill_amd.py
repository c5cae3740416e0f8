"""Importable alias of the `distancetransform-depthcompletion_amd` package (its directory name has
a '-' and so cannot appear in an `import` statement)."""
import importlib
import sys

_pkg = importlib.import_module("distancetransform-depthcompletion_amd")
sys.modules[__name__] = _pkg
