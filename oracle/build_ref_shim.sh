#!/usr/bin/env bash
# Build oracle/_ref/TAppDecoder_hmx: the REFERENCE's decoder application with the bodies of its hot-path members replaced
# by the libhmx calls of INTEGRATION.md section 3 -- the drop-in, shown rather than asserted.
#   TComTrQuant::xIT, xDeQuant, xITransformSkip; TComPrediction::predIntraLumaAng, predIntraChromaAng;
#   TComInterpolationFilter::filterHorLuma/VerLuma/HorChroma/VerChroma; TComYuv::addAvg
# Everything else (parsing, CABAC, the CU walk, reference sample preparation, deblocking, SAO, MD5) is the reference's own
# code, compiled from the sources where they lie under /root/reference.  TEST INFRASTRUCTURE:
#  * nothing of the reference is copied into the repository; the four edited translation units exist only in a pipe
#    (oracle/ref_shim_edit.py -> g++ -x c++ -); objects and the binary go to oracle/_ref/ (git-ignored, travels to the GPU box);
#  * the reference's build system is not run; no header, library or generated file is substituted.
# tests/test_ref_shim.py decodes committed bitstreams (made by the reference's encoder, tests/golden/make_bitstreams.py)
# through this binary on the GPU and requires the reference's own picture-digest check to say (OK) for every picture.
set -euo pipefail
REF=${REF_ROOT:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
OUT=$HERE/_ref
SRC=$REF/source
if [ ! -d "$SRC/Lib/TLibDecoder" ]; then
  echo "build_ref_shim: $SRC not present (GPU box?) - skipping" >&2
  exit 0
fi
[ -f "$OUT/obj_apps/TLibDecoder_TDecCu.o" ] || bash "$HERE/build_ref_apps.sh"
mkdir -p "$OUT/obj_shim"
CXX=${CXX:-g++}
FLAGS="-O2 -w -DMSYS_LINUX -I$SRC/Lib -I$SRC/Lib/TLibCommon"
for u in TComTrQuant TComPrediction TComInterpolationFilter TComYuv; do
  python3 "$HERE/ref_shim_edit.py" $u | $CXX $FLAGS -I"$ROOT/include" -include "$HERE/ref_shim.h" -x c++ -c - -o "$OUT/obj_shim/$u.o" &
done
for f in "$SRC"/App/TAppDecoder/*.cpp; do
  $CXX $FLAGS -I"$SRC/App/TAppDecoder" -c "$f" -o "$OUT/obj_shim/dec_$(basename "$f" .cpp).o" &
done
wait
COMMON=$(ls "$OUT"/obj/*.o | grep -v -e ref_tap.o -e /TComTrQuant.o -e /TComPrediction.o -e /TComInterpolationFilter.o -e /TComYuv.o)
$CXX -o "$OUT/TAppDecoder_hmx" "$OUT"/obj_shim/*.o "$OUT"/obj_apps/TLibDecoder_*.o "$OUT"/obj_apps/TAppCommon_*.o $COMMON \
  -L"$ROOT/thevc_amd" -lhmx -Wl,-rpath,'$ORIGIN/../../thevc_amd'
# the unmodified decoder next to it (the CPU test decodes the same fixtures with it: the fixtures and the digest parsing are sound)
$CXX -o "$OUT/TAppDecoder" "$OUT"/obj_shim/dec_*.o "$OUT"/obj_apps/TLibDecoder_*.o "$OUT"/obj_apps/TAppCommon_*.o $(ls "$OUT"/obj/*.o | grep -v ref_tap.o)
echo "build_ref_shim: wrote $OUT/TAppDecoder_hmx and $OUT/TAppDecoder"
