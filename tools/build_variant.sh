#!/bin/bash
# tools/build_variant.sh NAME SOURCE.hip FLAGS...: variants/libcoevo_NAME.so = the current objects with SOURCE rebuilt under FLAGS
# (A/B experiments: COEVO_ALLOW_VARIANT=1 COEVO_LIB=variants/libcoevo_NAME.so python tools/... - coevonet_amd.lib.load() refuses a
# library whose coevo_build_flags() is not empty unless COEVO_ALLOW_VARIANT=1; the flags given here are what it reports)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p variants
base="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $base "$@" "-DCOEVO_TU_FLAGS=\"$*\"" -c coevonet_amd/csrc/$src -o variants/${src%.hip}_$name.o
objs=""
for o in coevonet_amd/csrc/_obj/*.o; do
  if [ "$(basename $o)" == "${src%.hip}.o" ]; then objs="$objs variants/${src%.hip}_$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o variants/libcoevo_$name.so
echo variants/libcoevo_$name.so
