"""Generate tests/golden/select_cols.npz by calling the REFERENCE's ``select_cols_of_y``
(directionalvi/directional_vi.py:68-90) in this container.  TEST INFRASTRUCTURE.

The function itself is pure Python / torch / numpy / ``random``; only the module around it needs gpytorch (absent) at
import time -- for base classes, decorators and names that are never executed here.  A throw-away import hook answers
every ``gpytorch.*`` / ``wandb`` import with a module whose attributes are inert placeholders (usable as a base class,
a decorator or a decorator factory).  Nothing else in the repo uses it; the reference source is imported from where it
lies and never copied.  The GPU box only sees the committed .npz.

Usage:  python oracle/make_host_fixtures.py
"""
import importlib.abc
import importlib.machinery
import importlib.util
import os
import random
import sys
import types

import numpy as np
import torch

REF_DIR = "/root/reference/directionalvi"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "select_cols.npz")


class _InertModule(types.ModuleType):
    """answers ``import gpytorch...``: Capitalised attributes are empty classes (usable as bases), everything else is
    another inert module that can also be called as a decorator or decorator factory (``@cached(name=...)``)"""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name.lstrip("_")[:1].isupper():
            return type(name, (torch.nn.Module,), {})
        return _InertModule(self.__name__ + "." + name)

    def __call__(self, *a, **k):
        return a[0] if len(a) == 1 and callable(a[0]) and not k else self


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in ("gpytorch", "wandb"):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _InertModule(spec.name)

    def exec_module(self, module):
        pass


def load_reference_harness():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REF_DIR)
    cwd = os.getcwd()
    os.chdir(REF_DIR)                       # the file appends the relative path "utils" to sys.path
    try:
        spec = importlib.util.spec_from_file_location("_ref_directional_vi", os.path.join(REF_DIR, "directional_vi.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        os.chdir(cwd)
    return mod


# (dim, minibatch_dim, batch rows, python `random` seed, calls in a row)
CASES = [(2, 2, 5, 0, 3), (5, 2, 4, 1, 4), (20, 5, 3, 7, 4), (10, 10, 2, 3, 2), (50, 5, 2, 11, 3), (4, 1, 6, 5, 5)]


def main():
    ref = load_reference_harness()
    out = {}
    g = torch.Generator().manual_seed(99)
    for ci, (dim, p, B, seed, calls) in enumerate(CASES):
        random.seed(seed)
        out["case%d_meta" % ci] = np.array([dim, p, B, seed, calls])
        for k in range(calls):
            y = torch.rand(B, dim + 1, generator=g)
            y_sel, D = ref.select_cols_of_y(y, p, dim)
            out["case%d_call%d_y" % (ci, k)] = y.numpy()
            out["case%d_call%d_ysel" % (ci, k)] = y_sel.numpy()
            out["case%d_call%d_D" % (ci, k)] = D.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
