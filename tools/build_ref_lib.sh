#!/bin/bash
# Builds libdabx of a committed revision into tools/_build/ab/libdabx_<name>.so (git worktree in /tmp), for same-box A/B runs against
# the product library with tools/ab.sh / tools/ab_single.sh:   tools/build_ref_lib.sh <git ref> <name>
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=$1; NAME=$2
mkdir -p "$ROOT/tools/_build/ab"
W=$(mktemp -d /tmp/dabx_ref.XXXXXX)
git -C "$ROOT" worktree add -f "$W" "$REF" -q
EXTRA=""; grep -q hsa_ "$W"/dabstar_amd/csrc/*.cpp 2>/dev/null && EXTRA="-lhsa-runtime64"
( cd "$W" && mkdir -p o && for f in dabstar_amd/csrc/*.cpp dabstar_amd/csrc/*.hip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -w -x hip -c $f -o o/$(basename $f).o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC o/*.o $EXTRA -o "$ROOT/tools/_build/ab/libdabx_$NAME.so" )
git -C "$ROOT" worktree remove --force "$W"
ls -la "$ROOT/tools/_build/ab/libdabx_$NAME.so"
