#!/bin/bash
# ab_lib.sh build <name> [git-rev]: scratch/labs/libvs_<name>.so from k_warp.hip of the work tree (or of a commit) + the other objects
# ab_lib.sh run <outdir> <workload args...> -- <name> <name> ...: bench.py with each library in turn, three rounds, on one box
#   (name "cur" = the library of the work tree).  Kernel times differ by ~5 % from box to box: only same-box pairs compare.
set -e
ROOT=$(pwd); C=$ROOT/video-stab_amd/csrc
if [ "$1" = build ]; then
  mkdir -p scratch/labs; make -s -C $C
  SRC=$C/k_warp.hip
  if [ -n "$3" ]; then git show $3:video-stab_amd/csrc/k_warp.hip > $C/_k_warp_rev.hip; SRC=$C/_k_warp_rev.hip; fi
  /opt/rocm/bin/hipcc -x hip -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -w -mllvm -amdgpu-kernarg-preload-count=16 \
      $EXTRA -c $SRC -o scratch/labs/k_warp_$2.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/labs/libvs_$2.so scratch/labs/k_warp_$2.o $(ls $C/_build/*.o | grep -v k_warp.hip.o)
  rm -f scratch/labs/k_warp_$2.o $C/_k_warp_rev.hip
  exit 0
fi
OUT=$2; shift 2; mkdir -p $OUT
ARGS=(); while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
export VS_LAB=1
for rep in 1 2 3; do for n in "$@"; do
  if [ $n = cur ]; then unset VS_LIB_PATH; else export VS_LIB_PATH=$ROOT/scratch/labs/libvs_$n.so; fi
  timeout -k 10 300 python3 bench.py "${ARGS[@]}" --no-cpu-baseline --regions 3 > $OUT/${n}_$rep.json 2>> $OUT/err.log
  python3 - <<PY
import json
b=json.load(open("$OUT/${n}_$rep.json")); b=b.get("nv12_stabilize", b)
print("%-10s rep $rep: %.0f f/s warp %.1f us frac %.4f" % ("$n", b["value"], b["roofline"]["avg_launch_us"], b["roofline"]["frac"]))
PY
done; done
