#!/bin/bash
# Builds the REAL reference's only native component -- the Cython arithmetic coder rec/io/entropy_coding.pyx -- from the
# source where it lies under /root/reference, the way the reference's own setup.py does when Cython is installed
# (setup.py:7-21: Extension("rec.io.entropy_coding", ["rec/io/entropy_coding.pyx"])).  The generated C and the .so land
# in oracle/_ref/ only (git-ignored).  Needed directive: cpow=True (Cython >= 3 otherwise rejects
# `cdef long whole = 2**precision`, entropy_coding.pyx:60).  The shipped pre-generated entropy_coding.c (Cython 0.29.16)
# targets the numpy 1.x C API and does not compile against numpy 2.2 -- it is not used and nothing is patched.
# Skipped when /root/reference is absent (GPU box).  oracle/ref_io.py loads the result; tests/golden/make_golden_rec.py
# uses it to produce the committed .rec fixtures.
set -e
REF=/root/reference
OUT="$(cd "$(dirname "$0")" && pwd)/_ref"
[ -f "$REF/rec/io/entropy_coding.pyx" ] || { echo "build_ref.sh: $REF not present, nothing to do"; exit 0; }
mkdir -p "$OUT"
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy; print(numpy.get_include())")
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
SO="$OUT/entropy_coding$EXT"
if [ ! -f "$SO" ] || [ "$REF/rec/io/entropy_coding.pyx" -nt "$SO" ]; then
  (cd "$REF" && python3 -m cython -3 -X cpow=True rec/io/entropy_coding.pyx -o "$OUT/entropy_coding.c")
  gcc -O2 -fPIC -shared -w -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION -I"$PYINC" -I"$NPINC" \
      "$OUT/entropy_coding.c" -o "$SO"
  rm -f "$OUT/entropy_coding.c"
fi
echo "built $SO"
