"""Importable alias for the hyphenated package directory ``gp-derivatives-variational-inference_amd``."""
import importlib
import sys

_PKG = "gp-derivatives-variational-inference_amd"
_pkg = importlib.import_module(_PKG)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_PKG + "."):
        sys.modules[__name__ + _name[len(_PKG):]] = _mod
sys.modules[__name__] = _pkg
