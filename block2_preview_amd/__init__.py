"""Import shim: the package directory is named ``block2-preview_amd`` (not a valid Python
identifier), so this stub extends its search path to that directory and runs its __init__."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "block2-preview_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _real
