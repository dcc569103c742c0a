"""Turns the rocprofv3 output of a profiled bench.py run (see README.md) into the files kept here.

usage: python profiles/summarize.py <tag> <stats_dir> <fetch_dir> <write_dir> <build_tag>
  stats_dir : rocprofv3 --kernel-trace --stats            -- python3 bench.py ...
  fetch_dir : rocprofv3 --kernel-trace --pmc FETCH_SIZE   -- python3 bench.py ...   (own pass: TCC slots)
  write_dir : rocprofv3 --kernel-trace --pmc WRITE_SIZE   -- python3 bench.py ...
build_tag : vs_build_tag() of the library the passes ran on (bench.py prints it as config.build): bench.py reports
            the traffic only when it runs on the same build.
Writes profiles/<tag>_kernel_stats.csv and profiles/warp_traffic.json.
HBM bytes per launch of the warp kernel (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE and WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so the read side is doubled.
"""
import csv, glob, json, os, shutil, statistics, sys

tag, stats_dir, fetch_dir, write_dir, build_tag = sys.argv[1:6]
KERNEL = "warp_tab_kernel"      # the 3-channel warp of the batched pipeline (its coordinate tables come from warp_tables_kernel)
here = os.path.dirname(os.path.abspath(__file__))
src = glob.glob(os.path.join(stats_dir, "*", "*_kernel_stats.csv"))[0]
shutil.copy(src, os.path.join(here, tag + "_kernel_stats.csv"))


def counter(d, name):
    vals = []
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and KERNEL in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return vals


fetch, write = counter(fetch_dir, "FETCH_SIZE"), counter(write_dir, "WRITE_SIZE")
# full launches only (the first launches of a run carry fewer frames): those that write the most
full = max(write)
is_full = [w >= 0.99 * full for w in write]
fetch_full = [f for f, ok in zip(fetch, is_full) if ok] if len(fetch) == len(write) else fetch[-8:]
fetch_kib, write_kib = statistics.mean(fetch_full), statistics.mean([w for w, ok in zip(write, is_full) if ok])
out = {
    "kernel": KERNEL,
    "build_tag": build_tag,
    "fetch_size_kib_per_launch": round(fetch_kib, 1),
    "write_size_kib_per_launch": round(write_kib, 1),
    "gfx950_fetch_correction": 2.0,
    "hbm_bytes_per_launch": int(round(2.0 * fetch_kib * 1024 + write_kib * 1024)),
    "frames_per_launch": int(round(write_kib * 1024 / (1920 * 1080 * 3))),   # bench.py default geometry
    "launches_sampled": len(fetch_full),
    "source": tag,
}
json.dump(out, open(os.path.join(here, "warp_traffic.json"), "w"), indent=1)
print(out)
