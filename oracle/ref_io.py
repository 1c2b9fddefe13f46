"""Loads the REAL reference `rec.io` (ArithmeticCoder, write/read_compressed_code) for fixture generation and
cross-checks.  TEST INFRASTRUCTURE ONLY, and only usable in the build container: it needs /root/reference (python
sources rec/io/utils.py, rec/io/data_structures.py) and compiles the reference's Cython coder on demand with
oracle/build_ref.sh into a TEMPORARY directory outside the repository, removed when the process ends -- the compiled
reference never sits in the tree and never travels to the GPU box.
Nothing here may be imported by -m gpu tests, smoke() or bench.py.
"""
import atexit
import importlib.util
import os
import shutil
import subprocess
import sys
import tempfile
import types

REF = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
_build_dir = None


def available():
    """True where the reference's sources and the tools to compile its coder exist (the build container)."""
    return (os.path.isfile(os.path.join(REF, "rec", "io", "entropy_coding.pyx")) and
            importlib.util.find_spec("Cython") is not None and shutil.which("gcc") is not None)


def _build():
    global _build_dir
    if _build_dir is None:
        d = tempfile.mkdtemp(prefix="irec_ref_")
        atexit.register(shutil.rmtree, d, ignore_errors=True)
        subprocess.check_call(["bash", os.path.join(_HERE, "build_ref.sh"), d], stdout=subprocess.DEVNULL)
        _build_dir = d
    return _build_dir


def load():
    """Returns the reference's rec.io.utils module (with .ArithmeticCoder bound from the compiled extension)."""
    if "rec.io.utils" in sys.modules:
        return sys.modules["rec.io.utils"]
    if not available():
        raise RuntimeError("reference rec.io not available (needs /root/reference, Cython and gcc: the build container)")
    # synthetic packages: rec.io resolves python files in the reference tree and the extension in the temporary build
    # directory; the reference's own rec/io/__init__.py (`from .utils import *`) is bypassed, rec/__init__.py is empty anyway
    rec = types.ModuleType("rec"); rec.__path__ = [os.path.join(REF, "rec")]
    io = types.ModuleType("rec.io"); io.__path__ = [os.path.join(REF, "rec", "io"), _build()]
    sys.modules.setdefault("rec", rec)
    sys.modules["rec.io"] = io
    import importlib
    return importlib.import_module("rec.io.utils")
