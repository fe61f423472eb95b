"""Importable alias of the package directory `bournemouth-forced-aligner_amd/` (a hyphen cannot be
imported).  All code lives in that directory; this file only points the package path at it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "bournemouth-forced-aligner_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _real
