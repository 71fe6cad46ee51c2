"""knp-emi-cgx_amd: MI355X-native assemble-and-solve path for KNP-EMI.

The directory name is not a valid Python identifier; import it with
``importlib.import_module("knp-emi-cgx_amd")`` or put this directory on ``sys.path`` and use
``cgx_hip`` (native names) / ``CGx`` (the reference's module paths) directly.
"""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
if _here not in _sys.path:
    _sys.path.insert(0, _here)

from cgx_hip import *  # noqa: E402,F401,F403
