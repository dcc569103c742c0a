"""Turns the rocprofv3 passes of scratch/profile_r03.sh (rounds 3 and 4: `profile_r03.sh r04_<x>`) into the files kept under profiles/.

usage: python profiles/summarize_r03.py <tag> <dir with c1_stats, c1_fetch, c1_write, c1_sqa, c1_sqb, c2_...>

Writes
  profiles/<tag>_configs1_kernel_stats.csv / <tag>_configs2_kernel_stats.csv   rocprofv3 --kernel-trace --stats
  profiles/<tag>_configs1_pmc.txt / <tag>_configs2_pmc.txt                      per kernel: launches, mean duration in the pass,
                                                                                 mean of every counter per launch
  profiles/<tag>_timeline_one_batch.txt                                         one steady-state batch (kernel trace of c1_stats)
  profiles/warp_traffic.json                                                    HBM bytes per warp launch (FETCH_SIZE x 2 + WRITE_SIZE)
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of 16-byte-per-lane streaming reads at
64 B (MI355X_MICROARCH.md, "HBM"), so the read side is doubled - checked in the same pass on copy_rate_kernel (a plain
16-byte-per-lane copy of a known size, vs_dev_copy_rate): its doubled FETCH_SIZE must equal its WRITE_SIZE.
"""
import collections, csv, glob, json, os, re, shutil, statistics, subprocess, sys

tag, base = sys.argv[1], sys.argv[2]
repo_profiles = os.path.dirname(os.path.abspath(__file__))
# on the GPU box only gpurun_out/ travels back: the files are written to <dir>/profiles and copied into profiles/ afterwards
here = os.path.join(base, "profiles")
os.makedirs(here, exist_ok=True)
NAMES = {"c1": "configs1", "c2": "configs2"}


def short(name):
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def read_pass(d):
    """-> {kernel: {counter: [per-dispatch values]}, 'dur': {kernel: [ns]}} in dispatch order"""
    kt = {}
    for f in glob.glob(os.path.join(d, "*", "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            kt[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    vals = collections.defaultdict(lambda: collections.defaultdict(dict))
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            did = int(r["Dispatch_Id"])
            vals[k][r["Counter_Name"]][did] = vals[k][r["Counter_Name"]].get(did, 0.0) + float(r["Counter_Value"])
            if str(did) in kt:
                vals[k]["dur_ns"][did] = kt[str(did)]
    return vals


def build_tag(jsonfile):
    try:
        j = json.load(open(jsonfile))
        return j.get("config", {}).get("build") or j.get("build")
    except Exception:
        return None


traffic_path = os.path.join(here, "warp_traffic.json")
try:
    traffic = json.load(open(os.path.join(repo_profiles, "warp_traffic.json")))
    if "hbm_bytes_per_launch" in traffic:      # round 2's flat layout
        traffic = {}
except Exception:
    traffic = {}
btag = None
try:
    sys.path.insert(0, os.path.join(os.path.dirname(repo_profiles), "video-stab_amd"))
    from vsamd import capi
    btag = capi.load().lib.vs_build_tag().decode()
except Exception as e:   # noqa: BLE001
    print("no build tag:", e)

for wl, wname in NAMES.items():
    st = glob.glob(os.path.join(base, wl + "_stats", "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(here, "%s_%s_kernel_stats.csv" % (tag, wname)))
    lines = []
    passes = {p: read_pass(os.path.join(base, "%s_%s" % (wl, p))) for p in ("fetch", "write", "sqa", "sqb")}
    kernels = sorted(set().union(*[set(v) for v in passes.values()]))
    for k in kernels:
        line = "%-34s" % k
        for p in ("fetch", "write", "sqa", "sqb"):
            v = passes[p].get(k)
            if not v:
                continue
            n = len(v.get("dur_ns", {})) or max(len(x) for x in v.values())
            line += " | %s n=%d dur_us=%.1f" % (p, n, statistics.mean(v["dur_ns"].values()) / 1e3 if v.get("dur_ns") else 0)
            for c in sorted(v):
                if c != "dur_ns":
                    line += " %s=%.0f" % (c.replace("SQ_", "").replace("GRBM_", ""), statistics.mean(v[c].values()))
        lines.append(line)
    if lines:
        open(os.path.join(here, "%s_%s_pmc.txt" % (tag, wname)), "w").write(
            "# per kernel and pass: launches, mean duration IN THAT PASS (PMC passes serialise kernels), mean counter value per launch\n"
            "# FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE to be doubled on gfx950 for 16-byte-per-lane reads); SQ_* summed over the 8 XCDs\n"
            + "\n".join(lines) + "\n")
    # warp traffic: full launches only (those that write the most)
    fpass, wpass = passes["fetch"], passes["write"]
    # (round 4: luma and chroma tiles of the NV12 surfaces in one launch, warp_nv12_kernel; before: one launch per plane)
    nv12_k = [k for k in fpass if k.startswith("warp_nv12_kernel<0")]       # (the BORDER_CONSTANT instance: the stabilizer's warp)
    warp_k = ["warp_tab_kernel"] if wl == "c1" else (nv12_k if nv12_k else ["warp_plane_kernel<1>", "warp_plane_kernel<2>"])
    if all(k in fpass and k in wpass for k in warp_k):
        rd = wr = 0.0
        nl = 0
        for k in warp_k:
            w = list(wpass[k]["WRITE_SIZE"].values())
            f = list(fpass[k]["FETCH_SIZE"].values())
            full_w = max(w)
            wf = [x for x in w if x >= 0.99 * full_w]
            # the fetch pass is another run of the same program: its full launches are the ones that fetch the most
            full_f = statistics.median(sorted(f)[-max(1, len(wf) // 2):])
            ff = [x for x in f if x >= 0.9 * full_f]
            rd += 2.0 * statistics.mean(ff) * 1024
            wr += statistics.mean(wf) * 1024
            nl = len(wf)
        fb = 1920 * 1080 * 3 if wl == "c1" else 3840 * 2160 * 3 // 2
        frames = int(round(wr / fb))
        cal = None
        if "copy_rate_kernel" in fpass and "copy_rate_kernel" in wpass:
            cal = round(2.0 * statistics.mean(fpass["copy_rate_kernel"]["FETCH_SIZE"].values()) /
                        statistics.mean(wpass["copy_rate_kernel"]["WRITE_SIZE"].values()), 4)
        traffic[wname] = {
            "kernel": " + ".join(warp_k), "build_tag": btag, "source": tag,
            "read_bytes_per_launch": int(round(rd)), "write_bytes_per_launch": int(round(wr)),
            "hbm_bytes_per_launch": int(round(rd + wr)), "frames_per_launch": frames,
            "algorithmic_bytes_per_launch": 2 * fb * frames, "ratio_to_algorithmic": round((rd + wr) / (2.0 * fb * max(frames, 1)), 4),
            "read_ratio": round(rd / (fb * max(frames, 1)), 4), "launches_sampled": nl, "gfx950_fetch_correction": 2.0,
            "fetch_calibration_copy_kernel": cal,
            "distinct_input_MB": 796.3,
        }
        print(wname, traffic[wname])
json.dump(traffic, open(traffic_path, "w"), indent=1)
# one steady-state batch
try:
    tl = subprocess.run([sys.executable, os.path.join(os.path.dirname(repo_profiles), "scratch", "timeline.py"), os.path.join(base, "c1_stats")],
                        capture_output=True, text=True, timeout=120).stdout
    if tl.strip():
        open(os.path.join(here, "%s_timeline_one_batch.txt" % tag), "w").write(tl)
except Exception as e:   # noqa: BLE001
    print("no timeline:", e)
