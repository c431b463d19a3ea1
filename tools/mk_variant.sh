#!/bin/bash
# Build a VARIANT of libemap_hip.so into tools/ab/<name>.so: one translation unit recompiled with extra -D flags, the other objects
# taken from the in-tree build (csrc/_obj, built first if stale).  For same-box A/B runs with EMAP_HIP_LIB (tools/ab.sh):
#   tools/mk_variant.sh c8 emap_binned.hip -DSPLIT_CAP=8192u
#   tools/mk_variant.sh tileorder emap_kernels.hip -DRAY_TILE_ORDER
# Several variants can be built in parallel (each ~40-60 s): run the script in the background and `wait`.
set -e
name=$1; unit=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/elevation_mapping_cupy_amd/csrc
python $C/build.py > /dev/null
mkdir -p $R/tools/ab /tmp/emap_variants
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
obj=/tmp/emap_variants/${name}_${unit%.*}.o
/opt/rocm/bin/hipcc $FL "$@" -c $C/$unit -o $obj
objs=""
for o in $C/_obj/*.o; do [ "$(basename $o)" = "${unit%.*}.o" ] && objs="$objs $obj" || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fno-gpu-rdc -shared -fPIC $objs -o $R/tools/ab/$name.so
echo $R/tools/ab/$name.so
