#!/bin/bash
# ab_rev.sh <name> <git-rev>: scratch/labs/libvs_<name>.so = the whole library as of <git-rev> (same-box A/B against the work tree:
# VS_LIB=scratch/labs/libvs_<name>.so python bench.py ...).  Built in a temporary copy; nothing of the work tree is touched.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
git -C $ROOT archive $2 video-stab_amd/csrc include | tar -x -C $T
make -s -j8 -C $T/video-stab_amd/csrc
mkdir -p $ROOT/scratch/labs
cp $T/video-stab_amd/csrc/libvideo-stab.so $ROOT/scratch/labs/libvs_$1.so
rm -rf $T
echo built scratch/labs/libvs_$1.so from $2
