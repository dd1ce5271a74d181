"""Importable alias of the ``multi-pass-gan_amd`` package (its directory name is
not a Python identifier): ``import mpgan_amd`` and ``from mpgan_amd.x import y``
resolve to the very same module objects as ``multi-pass-gan_amd[.x]``."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_REAL = "multi-pass-gan_amd"
_ALIAS = __name__

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, module):
        self._module = module

    def create_module(self, spec):
        return self._module

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_ALIAS + "."):
            return None
        real = importlib.import_module(_REAL + fullname[len(_ALIAS):])
        return importlib.util.spec_from_loader(fullname, _AliasLoader(real))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
sys.modules[_ALIAS] = importlib.import_module(_REAL)
