"""Importable alias of the ``x-ggm_amd/`` package directory (a hyphen cannot appear in
an ``import`` statement).  All code lives in ``x-ggm_amd/``; this file only redirects
the package search path there and runs that directory's ``__init__.py``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "x-ggm_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
