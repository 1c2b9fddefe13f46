#!/bin/bash
# Builds the REAL reference's only native component -- the Cython arithmetic coder rec/io/entropy_coding.pyx -- from the
# source where it lies under /root/reference, the way the reference's own setup.py does when Cython is installed
# (setup.py:7-21: Extension("rec.io.entropy_coding", ["rec/io/entropy_coding.pyx"])).
#
#   usage: build_ref.sh OUT_DIR        (OUT_DIR must lie OUTSIDE the repository: oracle/ref_io.py passes a fresh
#                                       temporary directory and removes it when the process ends)
#
# The compiled reference never stays in the tree and never travels to the GPU box (round 5; until then it was kept in
# oracle/_ref/): it exists for the lifetime of the process that asked for it -- tests/golden/make_golden_rec.py, which
# produces the committed .rec fixtures, and tests/test_rec_io.py::test_against_live_reference_when_available.
# Needed directive: cpow=True (Cython >= 3 otherwise rejects `cdef long whole = 2**precision`, entropy_coding.pyx:60).
# The shipped pre-generated entropy_coding.c (Cython 0.29.16) targets the numpy 1.x C API and does not compile against
# numpy 2.2 -- it is not used and nothing is patched.
set -e
REF=/root/reference
[ -n "$1" ] || { echo "usage: build_ref.sh OUT_DIR (outside the repository)"; exit 2; }
OUT="$(mkdir -p "$1" && cd "$1" && pwd)"
REPO="$(cd "$(dirname "$0")/.." && pwd)"
case "$OUT/" in "$REPO"/*) echo "build_ref.sh: $OUT lies inside the repository; the compiled reference must not"; exit 2;; esac
[ -f "$REF/rec/io/entropy_coding.pyx" ] || { echo "build_ref.sh: $REF not present, nothing to do"; exit 0; }
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy; print(numpy.get_include())")
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
SO="$OUT/entropy_coding$EXT"
(cd "$REF" && python3 -m cython -3 -X cpow=True rec/io/entropy_coding.pyx -o "$OUT/entropy_coding.c")
gcc -O2 -fPIC -shared -w -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION -I"$PYINC" -I"$NPINC" \
    "$OUT/entropy_coding.c" -o "$SO"
rm -f "$OUT/entropy_coding.c"
echo "built $SO"
