"""MI355X-native hot path of Bissmella/Small-object-detection-transformers.

The directory name contains hyphens, so import it with
``importlib.import_module("small-object-detection-transformers_amd")`` (``sodt_amd``
is registered as an alias in ``sys.modules`` on first import).
"""
import sys as _sys

_sys.modules.setdefault("sodt_amd", _sys.modules[__name__])

from . import _lib  # noqa: E402,F401  (loads libsodt_hip.so; raises if it is missing)

__all__ = ["_lib"]
