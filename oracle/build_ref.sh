#!/usr/bin/env bash
# Build the REFERENCE's own hot-path code (HM TLibCommon, from /root/reference, where it lies)
# plus our tap harness (oracle/ref_tap.cpp) into oracle/_ref/libhmref.so.
#
#  * Test infrastructure only: it validates oracle/hmx_oracle.c and generates tests/golden/.
#  * Nothing from /root/reference is copied into the repository; objects and the .so go to
#    oracle/_ref/ (git-ignored; it does travel to the GPU box, where only the committed golden
#    vectors and the CPU restatement are needed).
#  * The reference's own build system is not run: every TLibCommon/*.cpp is compiled directly
#    with g++.  g++ 11 rejects one pre-standard construct in TComTrQuant.cpp (a for-init
#    variable reused after its loop, lines 1862/2118 and 2164/2173, inside the RDOQ function
#    that is outside this round's scope).  That file is therefore compiled from a stream:
#    `sed` hoists the two declarations and pipes the text to g++ on stdin; no patched copy is
#    stored anywhere.  No headers, libraries or generated code are substituted.
set -euo pipefail
REF=${REF_ROOT:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
SRC=$REF/source/Lib
if [ ! -d "$SRC/TLibCommon" ]; then
  echo "build_ref: $SRC not present (GPU box?) - skipping" >&2
  exit 0
fi
mkdir -p "$OUT/obj"
CXX=${CXX:-g++}
FLAGS="-O2 -w -fPIC -DMSYS_LINUX -I$SRC -I$SRC/TLibCommon"
pids=()
for f in "$SRC"/TLibCommon/*.cpp; do
  b=$(basename "$f" .cpp)
  o=$OUT/obj/$b.o
  if [ "$o" -nt "$f" ]; then continue; fi
  if [ "$b" = TComTrQuant ]; then
    ( sed -e 's/for (Int iCGScanPos = uiCGNum-1;/Int iCGScanPos; for (iCGScanPos = uiCGNum-1;/' \
          -e 's/for ( Int scanPos = 0; scanPos < iBestLastIdxP1; scanPos++ )/Int scanPos; for ( scanPos = 0; scanPos < iBestLastIdxP1; scanPos++ )/' \
          "$f" | $CXX $FLAGS -x c++ -c - -o "$o" ) &
  else
    $CXX $FLAGS -c "$f" -o "$o" &
  fi
  pids+=($!)
  if [ ${#pids[@]} -ge 8 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
gcc -O2 -w -fPIC -c "$SRC/libmd5/libmd5.c" -o "$OUT/obj/libmd5.o"
$CXX $FLAGS -c "$SRC/TLibVideoIO/TVideoIOYuv.cpp" -o "$OUT/obj/TVideoIOYuv.o"
$CXX $FLAGS -I"$HERE" -c "$HERE/ref_tap.cpp" -o "$OUT/obj/ref_tap.o"
$CXX -shared -o "$OUT/libhmref.so" "$OUT"/obj/*.o
echo "build_ref: wrote $OUT/libhmref.so"
