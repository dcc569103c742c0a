#!/bin/bash
# profile_r03.sh <tag> [stats|pmc|all]: the rocprofv3 passes behind profiles/r03_* (run on the GPU box through gpurun).  Every pass is its
# own run of the program (python3 directly behind `--`), PMC passes with --kernel-trace only, one pass per TCC counter and one per
# SQ counter set, as MI355X_MICROARCH.md prescribes.  Workloads: configs1 = bench.py's headline stream (1080p BGR8), configs2 =
# bench.py --workload configs2 (3840x2160 NV12).  Both read > 256 MB of distinct input (128 / 64 distinct frames in a cycle).
set -e
TAG=$1; WHAT=${2:-all}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export VS_BENCH_PREROLL_BATCHES=40 VS_BENCH_4K_WARM=20
A1="--no-extras --no-cpu-baseline --regions 3"
A2="--workload configs2 --regions 3"
cd /tmp && export TMPDIR=/tmp
run() {  # run <name> <rocprof args...> -- <bench args...>
    local name=$1; shift
    local pa=()
    while [ "$1" != "--" ]; do pa+=("$1"); shift; done
    shift
    rm -rf "$OUT/$name"
    rocprofv3 "${pa[@]}" -d "$OUT/$name" --output-format csv -- python3 "$ROOT/bench.py" "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
    echo "pass $name done" >> "$OUT/progress.txt"
}
if [ "$WHAT" = stats ] || [ "$WHAT" = all ]; then
    run c1_stats --kernel-trace --stats -- $A1
    run c2_stats --kernel-trace --stats -- $A2
fi
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
    for wl in c1 c2; do
        if [ $wl = c1 ]; then A=$A1; else A=$A2; fi
        run ${wl}_fetch --kernel-trace --pmc FETCH_SIZE -- $A
        run ${wl}_write --kernel-trace --pmc WRITE_SIZE -- $A
        run ${wl}_sqa --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- $A
        run ${wl}_sqb --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- $A
    done
fi
cd "$ROOT"
python3 profiles/summarize_r03.py "$TAG" "$OUT" > "$OUT/summary.txt" 2>&1 || true
# the raw counter / trace csv files are large: the summaries and per-kernel stats are what is kept
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
