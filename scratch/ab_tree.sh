#!/bin/bash
# ab_tree.sh <name> <file under video-stab_amd/csrc> <sed expression>: scratch/labs/libvs_<name>.so = the work tree's library with one
# sed edit applied (same-box A/B against the work tree: VS_LIB=scratch/labs/libvs_<name>.so ...).  Built in a temporary copy.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
mkdir -p $T/video-stab_amd
cp -r $ROOT/include $T/include
cp -r $ROOT/video-stab_amd/csrc $T/video-stab_amd/ && rm -rf $T/video-stab_amd/csrc/_build $T/video-stab_amd/csrc/*.so
sed -i "$3" $T/video-stab_amd/csrc/$2
diff -q $ROOT/video-stab_amd/csrc/$2 $T/video-stab_amd/csrc/$2 > /dev/null && { echo "the edit changed nothing"; exit 1; }
make -s -j8 -C $T/video-stab_amd/csrc 2>&1 | grep -v warning || true
mkdir -p $ROOT/scratch/labs
cp $T/video-stab_amd/csrc/libvideo-stab.so $ROOT/scratch/labs/libvs_$1.so
rm -rf $T
echo built scratch/labs/libvs_$1.so
