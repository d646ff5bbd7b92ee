#!/usr/bin/env bash
# Build the REFERENCE's encoder application and a decision tap on its decoder library, from the sources where they
# lie under /root/reference, into oracle/_ref/:
#   oracle/_ref/TAppEncoder        the reference encoder, unmodified (makes real bitstreams from synthetic YUV)
#   oracle/_ref/hm_decision_tap    oracle/ref_decision_tap.cpp linked against the reference's decoder library
#
#  * Test infrastructure only (tests/golden/make_stream_golden.py uses both to make the "real stream" fixtures).
#  * Nothing from /root/reference is copied into the repository; objects and binaries go to oracle/_ref/ (git-ignored).
#  * The reference's own build system is not run: its .cpp files are compiled directly with g++ (oracle/build_ref.sh
#    has already compiled TLibCommon, TLibVideoIO and libmd5 into oracle/_ref/obj).  g++ 11 rejects one pre-standard
#    construct reached from TAppEncTop.cpp (AnnexBwrite.h:80 binds a temporary std::string to a non-const reference);
#    that translation unit is preprocessed, the one declaration is given its `const` by sed in the stream, and the
#    stream is compiled -- no patched copy is stored, no header or library is substituted.
set -euo pipefail
REF=${REF_ROOT:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
SRC=$REF/source
if [ ! -d "$SRC/Lib/TLibDecoder" ]; then
  echo "build_ref_apps: $SRC not present (GPU box?) - skipping" >&2
  exit 0
fi
bash "$HERE/build_ref.sh"
mkdir -p "$OUT/obj_apps"
CXX=${CXX:-g++}
FLAGS="-O2 -w -DMSYS_LINUX -I$SRC/Lib -I$SRC/Lib/TLibCommon"
pids=()
compile() { # $1 source, $2 object
  if [ "$2" -nt "$1" ]; then return; fi
  if [ "$(basename "$1")" = TAppEncTop.cpp ]; then
    ( $CXX $FLAGS -I"$SRC/App/TAppEncoder" -E "$1" | sed -e 's/string &P = nalu.m_nalUnitData.str();/const string \&P = nalu.m_nalUnitData.str();/' \
        | $CXX $FLAGS -x c++-cpp-output -c - -o "$2" ) &
  else
    $CXX $FLAGS -I"$(dirname "$1")" -c "$1" -o "$2" &
  fi
  pids+=($!)
  if [ ${#pids[@]} -ge 8 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
}
for d in TLibEncoder TLibDecoder TAppCommon; do
  for f in "$SRC"/Lib/$d/*.cpp; do compile "$f" "$OUT/obj_apps/${d}_$(basename "$f" .cpp).o"; done
done
for f in "$SRC"/App/TAppEncoder/*.cpp; do compile "$f" "$OUT/obj_apps/enc_$(basename "$f" .cpp).o"; done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
COMMON=$(ls "$OUT"/obj/*.o | grep -v ref_tap.o)
$CXX -o "$OUT/TAppEncoder" "$OUT"/obj_apps/enc_*.o "$OUT"/obj_apps/TLibEncoder_*.o "$OUT"/obj_apps/TAppCommon_*.o $COMMON
$CXX $FLAGS -c "$HERE/ref_decision_tap.cpp" -o "$OUT/obj_apps/ref_decision_tap.o"
$CXX -o "$OUT/hm_decision_tap" "$OUT/obj_apps/ref_decision_tap.o" "$OUT"/obj_apps/TLibDecoder_*.o $COMMON
echo "build_ref_apps: wrote $OUT/TAppEncoder and $OUT/hm_decision_tap"
