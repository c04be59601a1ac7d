#!/bin/bash
# Build a VARIANT of libevt_hip.so (e.g. -DEVT_PROF phase timing, -DEVT_ABLATE=n) into scripts/probes/bin/libevt_<name>.so
# (git-ignored; travels to the GPU box with the snapshot).  Use it through EVT_LIB=<path>.
#   bash scripts/build_variant.sh prof -DEVT_PROF
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=/tmp/evt_variant_$NAME
mkdir -p $OBJ $ROOT/scripts/probes/bin
pids=()
for f in $ROOT/eventful-transformer_amd/csrc/*.hip; do
  o=$OBJ/$(basename ${f%.hip}).o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fvisibility=hidden "$@" -c $f -o $o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/scripts/probes/bin/libevt_$NAME.so $OBJ/*.o
echo $ROOT/scripts/probes/bin/libevt_$NAME.so
