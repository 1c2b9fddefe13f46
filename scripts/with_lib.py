"""Runs a script or the test suite against a diagnostic build of the library (csrc/variants/<name>.so): the A/B hook.

    python scripts/with_lib.py LIB script.py [args ...]
    python scripts/with_lib.py LIB -m pytest [args ...]

The product loader (irec/_lib.py) reads no environment variable; this wrapper makes the one explicit call
irec._lib.load(LIB) before the target runs, so every later irec._lib.load() of the process returns that build."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    lib, target, rest = sys.argv[1], sys.argv[2], sys.argv[3:]
    import irec
    if lib not in ("main", "product"):
        irec._lib.load(lib if os.path.sep in lib else os.path.join(ROOT, "relative-entropy-coding_amd", "csrc", "variants", lib + ".so"))
    if target == "-m":
        sys.argv = rest
        runpy.run_module(rest[0], run_name="__main__", alter_sys=True)
    else:
        sys.argv = [target] + rest
        runpy.run_path(target, run_name="__main__")


if __name__ == "__main__":
    main()
