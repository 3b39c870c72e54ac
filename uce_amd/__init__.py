"""Importable alias of the `unified-concept-editing_amd/` package directory.

The contract names the package directory `unified-concept-editing_amd` (a hyphen cannot appear in
a Python import), so this thin package extends its `__path__` over that directory: every module
there is importable as `uce_amd.<module>`.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "unified-concept-editing_amd")
__path__.append(_PKG_DIR)
PKG_DIR = _PKG_DIR
REPO_ROOT = _os.path.dirname(_PKG_DIR)
__version__ = "0.1.0"
