#!/bin/bash
# Build a compile-time variant of libamsm.so for A/B runs:  tools/build_variant.sh NAME "-DFOO=1 ..." [unit ...]
# Recompiles the listed units (default: kern_pallas.hip) with the extra flags and links them with the default objects
# into build/variants/libamsm_NAME.so; select it with AMSM_LIB_PATH.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; shift 2 || true
UNITS=${@:-kern_pallas.hip}
OUT=$ROOT/build/variants; mkdir -p $OUT/obj_$NAME
OBJS=""
for u in api.hip kern_pallas.hip kern_bls12_381.hip kern_fr.hip; do
  o=$ROOT/build/obj/${u%.hip}.o
  for v in $UNITS; do
    if [ "$v" = "$u" ]; then
      o=$OUT/obj_$NAME/${u%.hip}.o
      hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-pass-failed -Rpass-analysis=kernel-resource-usage $FLAGS -c $ROOT/accumulation_amd/csrc/$u -o $o > $o.log 2>&1
    fi
  done
  OBJS="$OBJS $o"
done
hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libamsm_$NAME.so $OBJS
echo $OUT/libamsm_$NAME.so
