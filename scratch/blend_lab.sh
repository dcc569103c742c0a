#!/bin/bash
# blend_lab.sh build | run <outdir>: timing variants of the plane kernels' blend (k_warp.hip, VS_WARP_BLEND_LAB: 1 no weight reads,
# 2 no tap reads, 3 neither) as separate libraries under scratch/labs/, and bench.py --workload configs2 on each (results are wrong
# pictures by construction: timing only).
set -e
ROOT=$(pwd); C=$ROOT/video-stab_amd/csrc
if [ "$1" = build ]; then
  mkdir -p scratch/labs
  make -s -C $C
  for n in 1 2 3; do
    /opt/rocm/bin/hipcc -x hip -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 \
        -DVS_WARP_BLEND_LAB=$n -c $C/k_warp.hip -o scratch/labs/k_warp_lab$n.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/labs/libvs_lab$n.so scratch/labs/k_warp_lab$n.o $(ls $C/_build/*.o | grep -v k_warp.hip.o)
    rm scratch/labs/k_warp_lab$n.o
  done
  exit 0
fi
OUT=$2; mkdir -p $OUT
export VS_LAB=1
for n in 0 1 2 3; do
  if [ $n = 0 ]; then unset VS_LIB_PATH; else export VS_LIB_PATH=$ROOT/scratch/labs/libvs_lab$n.so; fi
  timeout -k 10 300 python3 bench.py --workload configs2 --no-cpu-baseline --regions 3 > $OUT/lab$n.json 2>> $OUT/err.log
  python3 - <<PY
import json
b=json.load(open("$OUT/lab$n.json")); b=b.get("nv12_stabilize", b)
print("blend lab $n: %.0f f/s warp %.1f us" % (b["value"], b["roofline"]["avg_launch_us"]))
PY
done
