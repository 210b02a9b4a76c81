"""Drop-in mirror of the reference's ``i2c`` package for the cubature hot path.

Works both as a sub-package (``input-inference-for-control_amd.i2c``) and as the top-level
package ``i2c`` when ``input-inference-for-control_amd/`` itself is put on ``sys.path`` (which is
how a script written for the reference picks up the MI355X build without edits).
"""
import importlib
import os
import sys

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_core():
    name = os.path.basename(_PKG_DIR)
    try:
        return importlib.import_module(name)
    except ImportError:
        sys.path.insert(0, os.path.dirname(_PKG_DIR))
        return importlib.import_module(name)


core = _load_core()  # the engine package: BatchedI2c, load_library, ...
