#!/bin/bash
# Builds an experiment variant of libdabx.so next to the product library, for same-box A/B runs (tools/ab.sh):
#   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DDABX_VIT_GRID_CAP=2048]   ->  tools/_build/ab/libdabx_<name>.so
NAME=$1; shift
mkdir -p tools/_build/ab /tmp/dabx_variant_$NAME
for f in dabstar_amd/csrc/*.hip dabstar_amd/csrc/*.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -w "$@" -x hip -c $f -o /tmp/dabx_variant_$NAME/$(basename $f).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/dabx_variant_$NAME/*.o -lhsa-runtime64 -o tools/_build/ab/libdabx_$NAME.so && echo built tools/_build/ab/libdabx_$NAME.so
rm -rf /tmp/dabx_variant_$NAME
