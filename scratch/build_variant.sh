#!/bin/bash
# build_variant.sh <k_warp source> <out.so> [extra hipcc flags]: libvideo-stab with another warp kernel (A/B measurements)
set -e
SRC=$1; OUT=$2; shift 2
CS=/root/repo/video-stab_amd/csrc
[ -d "$CS" ] || CS=$(dirname $(readlink -f $0))/../video-stab_amd/csrc
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc -x hip -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -w -I$CS "$@" -c $SRC -o $TMP/k_warp.o
OBJS=$(ls $CS/_build/*.o | grep -v k_warp.hip.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT $TMP/k_warp.o $OBJS
rm -rf $TMP
