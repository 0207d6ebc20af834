#!/bin/bash
# Builds libdabx of the committed HEAD into tools/_build/ab/libdabx_head.so (git worktree in /tmp), for same-box A/B runs against the
# working tree's build with tools/ab.sh.  tools/_build/ is not tracked (*.so is git-ignored) but travels to the GPU box.
set -e
mkdir -p "$(cd "$(dirname "$0")/.." && pwd)/tools/_build/ab"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/dabx_head.XXXXXX)
git -C "$ROOT" worktree add -f "$W" HEAD -q
( cd "$W" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-fast-math -ffp-contract=off -Wno-unused-function \
    -Wno-unused-result -x hip dabstar_amd/csrc/*.cpp dabstar_amd/csrc/*.hip -o "$ROOT/tools/_build/ab/libdabx_head.so" )
git -C "$ROOT" worktree remove --force "$W"
ls -la "$ROOT/tools/_build/ab/libdabx_head.so"
