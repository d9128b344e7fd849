#!/bin/bash
# Builds smolscale_amd/libsmolscale_amd_tuning.so: the library with -DSMOL_TUNING (the SMOL_* ablation /
# geometry knobs of smol_jobdev.h are read from the environment). Developer tool: select it with
# SMOLSCALE_AMD_LIB=smolscale_amd/libsmolscale_amd_tuning.so. The shipped library never reads those knobs.
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"
W="$(mktemp -d /tmp/smol_tuning.XXXXXX)"
mkdir -p "$W/include" "$W/smolscale_amd/csrc"
cp "$HERE"/include/*.h "$W/include/"
cp "$HERE"/smolscale_amd/csrc/*.h "$HERE"/smolscale_amd/csrc/*.hip "$HERE"/smolscale_amd/csrc/*.c "$HERE"/smolscale_amd/csrc/Makefile "$W/smolscale_amd/csrc/"
# usage: build_tuning.sh ["extra compiler flags" [output .so name]]
FLAGS="-DSMOL_TUNING ${1:-}"
OUTNAME="${2:-libsmolscale_amd_tuning.so}"
make -s -j8 -C "$W/smolscale_amd/csrc" EXTRA="$FLAGS"
cp "$W/smolscale_amd/libsmolscale_amd.so" "$HERE/smolscale_amd/$OUTNAME"
rm -rf "$W"
echo "built smolscale_amd/$OUTNAME ($FLAGS)"
