"""Import alias: Python cannot import a directory named ``handwritten-ocr_amd`` (hyphen), so this package
points its search path at that directory and runs its ``__init__``.  All code lives in ``handwritten-ocr_amd/``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "handwritten-ocr_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py"), "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
