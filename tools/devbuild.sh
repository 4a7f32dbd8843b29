#!/bin/bash
# Experiment build of the library WITHOUT touching the working tree:
#   tools/devbuild.sh <name> <git-ref | WORK> [-DTSDF_DEV_ONLY32 ...]   ->   build/libtsdf_hip_<name>.so
# Takes csrc/ and include/ from the given commit (or from the working tree: WORK), applies
# tools/patches/r05_removed_knobs.diff there (the parked experiment knobs: TSDF_DEV_ONLY32 / _ONLY64 builds take 15-25 s
# instead of 3 min) and compiles with the product's flags plus the given ones.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; REF=$2; shift 2
PKG=handposeestimation-with-3d-cnns_amd
TMP=$(mktemp -d /tmp/tsdf_devbuild.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
if [ "$REF" = WORK ]; then
  mkdir -p "$TMP/$PKG" && cp -r "$ROOT/$PKG/csrc" "$TMP/$PKG/" && cp -r "$ROOT/include" "$TMP/"
else
  git -C "$ROOT" archive "$REF" "$PKG/csrc" include | tar -x -C "$TMP"
fi
(cd "$TMP" && patch -s -p1 < "$ROOT/tools/patches/r05_removed_knobs.diff")
mkdir -p "$ROOT/build"
(cd "$TMP/$PKG/csrc" && ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
   -Wall -Wextra -Wno-unused-parameter "$@" -shared -o "$ROOT/build/libtsdf_hip_$NAME.so" tsdf_hip.hip)
echo "build/libtsdf_hip_$NAME.so"
