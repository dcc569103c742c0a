#!/usr/bin/env python3
"""Headline benchmark: stabilized frames/s on synthetic 1080p clips (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (the driver launches N>1 through torch.distributed.run;
RANK / LOCAL_RANK / WORLD_SIZE come from the environment).  Each rank owns
`--streams` independent video streams (default 1 = BASELINE configs[1]: one
1920x1080 BGR8 stream, 200 corners, 3-level LK 21x21, RANSAC partial affine,
warpAffine); streams never exchange data, so scaling is weak and there is no
data-path collective - torch.distributed only provides the barrier and the
max-over-ranks of the timed region.

A STEP = one pass of the hot path over one batch of synthetic input: `--batch`
(64) consecutive stabilize() calls (vs_stab_push_dev) per stream on frames that
are already resident in HBM - the unit the batch mode issues its launches in, so
that any K is a whole number of steady-state batches.  `value` is frames/s.

The input of a stream is a closed-loop clip of `--clip-frames` (128) DISTINCT
frames rendered on the device (796 MB at 1080p: three times the 256 MB Infinity
Cache) and played in a cycle, so the warp reads every source frame from HBM;
the results go to a ring of 3 x batch output frames (1.2 GB).  The timed region
of exactly K steps is repeated `--regions` (9) times, each bracketed by barrier +
device sync; `value` is the MEDIAN region (all of them are in `regions`).

Prints ONE JSON line on rank 0; at N = 1 it also carries the PCIe-inclusive rate
of the host-pointer entry point, BASELINE configs[2] (3840x2160) and the CPU
baseline.  `--workload configs2` makes the 4K NV12 stream the measured workload
(profiling runs of its kernels).
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "video-stab_amd"))

from vsamd import capi, dist as vsdist, synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured copy rate
# algorithmic bytes of the WHOLE step per 1080p BGR8 frame (downscale+gray, pyramid, GFTT every 2nd frame,
# LK gathers, warp)
WHOLE_STEP_BYTES_1080P = 22.7e6


def make_params(vs, config=None, **over):
    # BASELINE.json configs[1]: 200 corners, 3-level LK (maxLevel 2) with a 21x21 window;
    # everything else is the reference's live default (Stabilizer.h:76-175, Stabilizer.cpp:611-649).
    kw = dict(max_corners=200, lk_win_size=21, lk_max_level=2, lk_max_iters=20, lk_epsilon=0.03, smoothing_radius=30)
    kw.update(over)
    p = vs.params(**kw)
    if config:
        # the "stabilizer" section of a config.yaml of the reference's apps on top of that (vs_config_read_stab;
        # keys the file does not name keep the values above)
        p, present = capi.Config(vs, path=config).stab_params(base=p)
        if not present:
            raise SystemExit("bench.py: %s has no 'stabilizer' section" % config)
    return p


def clip_order(n_frames, n_steps):
    """Ping-pong through a (short, host-side) clip so consecutive frames always differ by one camera step."""
    fwd = list(range(n_frames)) + list(range(n_frames - 2, 0, -1))
    return [fwd[i % len(fwd)] for i in range(n_steps)]


def cpu_baseline(width, height, frames, order_fn, keep=0):
    """The oracle (CPU restatement of src/Stabilizer.cpp, kind "port") on the same workload,
    single thread, bounded sample.  Also returns the push order and the first `keep` frames the oracle
    produced, for check_outputs()."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    o = oracle_lib.load()
    p = o.params(max_corners=200, lk_win_size=21, lk_max_level=2, lk_max_iters=20, lk_epsilon=0.03,
                 smoothing_radius=30)
    o.lib.vso_set_threads(1)
    s = o.stabilizer(p)
    warm, timed = 34, 90
    order = order_fn(len(frames), warm + timed)
    kept = []
    for i in order[:warm]:
        r = s.push(frames[i])
        if r is not None and len(kept) < keep:
            kept.append(r)
    t0 = time.perf_counter()
    n_out = 0
    for i in order[warm:]:
        r = s.push(frames[i])
        if r is not None:
            n_out += 1
            if len(kept) < keep:
                kept.append(r)
    dt = time.perf_counter() - t0
    s.close()
    # all-core variant of the same port (resize / corner response / LK points / warp rows spread over threads)
    ncores = os.cpu_count() or 1
    nthr = min(ncores, 64)
    o.lib.vso_set_threads(nthr)
    s = o.stabilizer(p)
    for i in order[:warm]:
        s.push(frames[i])
    t1 = time.perf_counter()
    for i in order[warm:warm + 60]:
        s.push(frames[i])
    dt_mt = time.perf_counter() - t1
    s.close()
    o.lib.vso_set_threads(1)
    return order, kept, {
        "value": round(timed / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": "%d steady-state stabilize() calls on %d frames of the same %dx%d clip, oracle/ single thread" % (
            timed, len(frames), width, height),
        "all_cores": {"value": round(60 / dt_mt, 3), "cores": nthr, "host_cores": ncores,
                      "note": "same port, threaded stages spread over %d threads" % nthr},
    }


def check_outputs(vs, device, params, clip, w, h, batch, order, ref):
    """The HIP path with the bench's own settings (batch mode, zero-copy, the resident clip of stream 0) over the pushes the
    cpu_baseline leg has just put through the oracle: its first len(ref) outputs must equal the oracle's, byte for byte.  The
    oracle is the checker here, nothing else; a mismatch ends the run."""
    fb = w * h * 3
    s = vs.stabilizer(params, device=device)
    s.set_batch(batch)
    s.set_zero_copy(True)
    d_out = capi.DevBuf(vs, fb * (len(order) + 1))
    k = 0
    for i in order:
        k += s.push_dev(clip.ptr + i * fb, w, h, w * 3, capi.FMT_BGR8, d_out.ptr + k * fb, w * 3)
    s.sync()
    n = min(k, len(ref))
    for j in range(n):
        if not np.array_equal(d_out.download((h, w, 3), np.uint8, j * fb), ref[j]):
            raise SystemExit("bench.py: output %d of the HIP path differs from the oracle's (batch %d, zero-copy)" % (j, batch))
    s.close()
    d_out.free()
    return n


class StreamSet:
    """S streams in batch mode on one GPU, each with a resident clip (one DevBuf of `clip_frames` packed frames, played in a
    cycle) and a ring of output frames.  S > 1: one vs_batch group (one launch per stage over the frames of all streams; a step
    is `batch // S` frames per stream, so a step carries as many frames as the one-stream step) unless group=False (S
    independent vs_stab instances, each with steps of `batch` frames: round 2's --streams)."""

    def __init__(self, vs, device, params, clips, clip_frames, w, h, fmt, batch, warp_batch, zero_copy, group=True):
        self.vs, self.w, self.h, self.fmt = vs, w, h, fmt
        self.fb = w * h * 3 if fmt == capi.FMT_BGR8 else w * h * 3 // 2
        self.stride = w * 3 if fmt == capi.FMT_BGR8 else w
        self.clip_frames = clip_frames
        self.d_in = clips
        S = len(clips)
        self.grouped = bool(group) and S > 1 and batch > 1
        self.BT = max(1, min(64, batch)) if not self.grouped else max(1, min(64, batch) // S)     # pushes per stream and step
        self.WB = max(1, min(32, warp_batch if batch == 1 else self.BT * (S if self.grouped else 1)))
        frames_per_step = self.BT * (S if self.grouped else 1)                # frames between one pair of warp-stage events
        self.WB_frames = frames_per_step if batch > 1 else self.WB
        self.NOUT = max(2 * self.WB, 3 * max(self.BT, 64 // S if self.grouped else self.BT))    # a result stays untouched until its step and the next have been issued
        self.d_out = [[capi.DevBuf(vs, self.fb) for _ in range(self.NOUT)] for _ in clips]
        if self.grouped:
            self.group = vs.batch(params, S, self.BT, device=device)
            self.group.set_zero_copy(bool(zero_copy))
            self.stabs = [self.group.stream(j) for j in range(S)]
            self.timed = [self.stabs[0]]                                      # a group books its stage times on its first member
            # the pointer arrays of every clip position and output slot, marshalled once
            self.fr_arrays = [self.group.pointer_array([b.ptr + fi * self.fb for b in self.d_in]) for fi in range(clip_frames)]
            self.ou_arrays = [self.group.pointer_array([o[oi].ptr for o in self.d_out]) for oi in range(self.NOUT)]
        else:
            self.group = None
            self.stabs = [vs.stabilizer(params, device=device) for _ in clips]
            for s in self.stabs:
                s.set_batch(self.BT)
                s.set_zero_copy(bool(zero_copy))
                s.set_warp_batch(self.WB)
            self.timed = self.stabs
        self.i = 0

    def push(self, n):
        """n consecutive pushes per stream."""
        for _ in range(n):
            fi = self.i % self.clip_frames
            oi = self.i % self.NOUT
            if self.group is not None:
                self.group.push_dev_arrays(self.fr_arrays[fi], self.w, self.h, self.stride, self.fmt, self.ou_arrays[oi], self.stride)
            else:
                for j, s in enumerate(self.stabs):
                    s.push_dev(self.d_in[j].ptr + fi * self.fb, self.w, self.h, self.stride, self.fmt, self.d_out[j][oi].ptr, self.stride)
            self.i += 1

    def sync(self):
        if self.group is not None:
            self.group.sync()
        else:
            for s in self.stabs:
                s.sync()

    def frames_out(self):
        return sum(s.counters().frames_out for s in self.stabs)

    def set_profiling(self, mode):
        for s in self.timed:
            s.set_profiling(mode)
            s.stage_times()          # drop anything recorded so far

    def stage_totals(self):
        n_st = capi.STAGE_COUNT
        ms, n = [0.0] * n_st, [0] * n_st
        for s in self.timed:
            a, b = s.stage_times()
            for k in range(n_st):
                ms[k] += a[k]
                n[k] += b[k]
        return ms, n

    def host_frames(self, j, count):
        """The first `count` frames of stream j's clip as numpy arrays (the CPU baseline and the host entry points)."""
        shape = (self.h, self.w, 3) if self.fmt == capi.FMT_BGR8 else (self.h * 3 // 2, self.w)
        return [self.d_in[j].download(shape, np.uint8, i * self.fb) for i in range(min(count, self.clip_frames))]

    def close(self, free_clips=True):
        if self.group is not None:
            self.group.close()
        else:
            for s in self.stabs:
                s.close()
        if free_clips:
            for b in self.d_in:
                b.free()
        for bs in self.d_out:
            for b in bs:
                b.free()


def timed_regions(ss, comm, steps, regions):
    """`regions` timed regions of exactly `steps` steps each, every one bracketed by barrier + device sync on both sides;
    elapsed = max over ranks.  Returns the list of elapsed seconds."""
    def sync_all():
        ss.sync()
        comm.device_sync()
    out = []
    for _ in range(regions):
        sync_all()
        comm.barrier()
        sync_all()
        t0 = time.perf_counter()
        ss.push(steps * ss.BT)
        sync_all()
        comm.barrier()
        sync_all()
        out.append(comm.max_over_ranks(time.perf_counter() - t0))
    return out


def warp_roofline(ss, alg_bytes_per_frame, frames_out_timed, kernel, stage_ms, stage_n):
    """Warp kernel(s) of the timed regions: HIP-event time of their launches on the stream they run on."""
    # (a batch of more than 32 frames is warped by ceil(batch / 32) launches back to back inside one pair of events)
    launches = max(stage_n[capi.STAGE_WARP], 1) * max(1, (ss.WB_frames + 31) // 32)
    frames_per_launch = frames_out_timed / launches
    avg_ms = stage_ms[capi.STAGE_WARP] / launches
    byts = alg_bytes_per_frame * frames_per_launch
    achieved = byts / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "bytes_per_launch": byts,
            "frames_per_launch": round(frames_per_launch, 3), "avg_launch_us": round(avg_ms * 1e3, 3), "launches": launches}


def tables_pass(ss, roof, steps):
    """The coordinate-table kernel (warp_tables_kernel) is queued right behind the batch tail, outside the warp stage's
    event pair; its time is measured here, in a region of its own (profiling mode 3: one more event pair per batch on the
    critical stream, so not inside the headline regions) and folded into `stage_frac` = algorithmic bytes over
    warp + tables time."""
    ss.set_profiling(3)
    ss.push(steps * ss.BT)
    ss.sync()
    ms, n = ss.stage_totals()
    ss.set_profiling(1)
    if n[capi.STAGE_WARP] == 0:
        roof["tables_note"] = "no warp launches in the tables pass"
        return
    if n[capi.STAGE_WARP_TABLES] == 0:
        # round 4: the coordinate tables of a frame's warp are built by the workgroup that computes the frame's inverse map
        # (release kernel of the batch tail, k_ransac.hip): no launch of their own, nothing to add to the warp kernel's time
        roof["tables_us_per_launch"] = 0.0
        roof["stage_frac"] = roof["frac"]
        roof["stage_note"] = ("stage_frac = frac: the coordinate tables are built inside the batch tail's release kernel "
                              "(no launch of their own on the warp's stream); the warp stage is the warp kernel(s)")
        return
    # a batch's tables are built by one event-bracketed group of launches (one per 32 frames and plane), like its warps
    per_launch = max(1, (ss.WB_frames + 31) // 32)
    tab_us = ms[capi.STAGE_WARP_TABLES] / n[capi.STAGE_WARP_TABLES] * 1e3 / per_launch
    warp_us = ms[capi.STAGE_WARP] / n[capi.STAGE_WARP] * 1e3 / per_launch
    roof["tables_us_per_launch"] = round(tab_us, 3)
    roof["warp_us_in_tables_pass"] = round(warp_us, 3)
    stage_gbps = roof["bytes_per_launch"] / ((roof["avg_launch_us"] + tab_us) * 1e-6) / 1e9
    roof["stage_frac"] = round(stage_gbps / HBM_PEAK_GBPS, 4)
    roof["stage_note"] = ("frac = the warp kernel(s) alone (avg_launch_us); stage_frac = warp kernel(s) + the coordinate-table "
                          "kernel (warp_tables_kernel: tables_us_per_launch, measured in a separate region)")


def traffic_from_counters(vs, roof, key):
    """HBM bytes per launch from the PMC passes kept in profiles/warp_traffic.json - only when that entry was measured on
    THIS build of the kernels (vs_build_tag); otherwise the field stays null and says why."""
    pmc = os.path.join(ROOT, "profiles", "warp_traffic.json")
    try:
        t = json.load(open(pmc)).get(key)
    except Exception:
        t = None
    if not t:
        roof["traffic_note"] = "no counter entry '%s' in profiles/warp_traffic.json" % key
        return
    tag = vs.lib.vs_build_tag().decode()
    if t.get("build_tag") != tag:
        roof["traffic_note"] = "stale: profiles/warp_traffic.json[%s] was measured on build %s, this is %s" % (key, t.get("build_tag"), tag)
        return
    scale = roof["frames_per_launch"] / t["frames_per_launch"]
    roof["traffic"] = round(t["hbm_bytes_per_launch"] * scale)
    roof["traffic_read"] = round(t["read_bytes_per_launch"] * scale)
    roof["traffic_write"] = round(t["write_bytes_per_launch"] * scale)
    roof["traffic_note"] = "FETCH_SIZE x2 + WRITE_SIZE of %s (%d-frame launches, %s MB of distinct input), scaled to this launch" % (
        t.get("source"), t["frames_per_launch"], t.get("distinct_input_MB", "?"))


def copy_yardstick(vs, nbytes):
    """What a plain device copy of the same number of bytes reaches on this device (vs_dev_copy_rate: 16 bytes per lane,
    streaming stores, HIP events around 20 back-to-back launches over buffers that together exceed the Infinity Cache)."""
    import ctypes as C
    rate = C.c_double(0.0)
    vs.check(vs.lib.vs_dev_copy_rate(C.c_size_t(int(nbytes)), 20, C.byref(rate)))
    return round(rate.value, 1)


def host_api_rate(vs, device, params, frames, n_timed=240):
    """vs_stab_push: host frame in, host frame out (what vs::Stabilizer::stabilize(cv::Mat) calls) - PCIe both ways.
    Frames in pageable memory (numpy arrays as they come: what a cv::Mat holds) or page-locked (vs_host_alloc); the
    synchronous call or the host pipeline (vs_stab_set_host_pipeline; Parameters::hostPipeline for the C++ class: a call
    returns the previous call's frame, its download runs beside this call's upload)."""
    order = clip_order(len(frames), 64 + n_timed)

    def run(pinned, pipeline, fresh_out=False):
        s = vs.stabilizer(params, device=device)
        if pipeline:
            s.set_host_pipeline(True)
        src = frames
        out = None if fresh_out else np.empty_like(frames[0])
        bufs = []
        if pinned:
            bufs = [capi.HostBuf(vs, f.shape) for f in frames] + [capi.HostBuf(vs, frames[0].shape)]
            for b, f in zip(bufs, frames):
                b.array[...] = f
            src = [b.array for b in bufs[:-1]]
            out = bufs[-1].array
        for i in order[:64]:
            s.push(src[i], out=out)
        t0 = time.perf_counter()
        n_out = 0
        for i in order[64:]:
            if s.push(src[i], out=out) is not None:
                n_out += 1
        dt = time.perf_counter() - t0
        s.close()
        for b in bufs:
            b.free()
        return {"value": round(n_out / dt, 1), "ms_per_frame": round(dt / max(n_out, 1) * 1e3, 4)}
    piped = run(False, True)
    return {"value": piped["value"], "unit": "frames/s", "ms_per_frame": piped["ms_per_frame"],
            "what": "vs_stab_push with the host pipeline, %d calls: PAGEABLE host frame in, stabilized frame into the caller's "
                    "(pageable) buffer per call - H2D + D2H inside the call, the download of the previous result issued by the "
                    "instance's helper thread beside this call's upload" % n_timed,
            "pageable_host_pipeline_fresh_output": dict(run(False, True, True), what="the same, a newly allocated output array per call (first-touch page faults inside the call)"),
            "pageable_synchronous": dict(run(False, False), what="the synchronous call (a call returns the frame it made due): upload, device work and download in a row"),
            "pageable_synchronous_fresh_output": dict(run(False, False, True), what="the same, a newly allocated output array per call (round 1's figure: 1 985)"),
            "page_locked_synchronous": dict(run(True, False), what="synchronous call, frames in page-locked memory (vs_host_alloc)"),
            "page_locked_host_pipeline": dict(run(True, True), what="host pipeline, frames in page-locked memory")}


def cpp_class_rate(device):
    """vs::Stabilizer::stabilize(cv::Mat) measured through the C++ class itself (tests/cpp/wrapper_smoke.cpp built against
    the test cv::Mat, `--time`): default construction (synchronous) and the pipelined call."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "wrapper_time")
    if not os.path.exists(exe):
        return {"note": "tests/cpp/_build/wrapper_time not built (make -C tests/cpp)"}
    try:
        r = subprocess.run([exe, str(device)], capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and line:
            out = json.loads(line[-1])
            r2 = subprocess.run([exe, str(device), "checker"], capture_output=True, text=True, timeout=300)
            line2 = [ln for ln in r2.stdout.splitlines() if ln.startswith("{")]
            if r2.returncode == 0 and line2:
                out["busy_picture"] = json.loads(line2[-1])
            return out
        return {"note": "wrapper_time failed (%d): %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
    except Exception as e:       # noqa: BLE001 - a missing extra must not lose the headline line
        return {"note": "wrapper_time: %r" % (e,)}


def run_stream_workload(comm, ss, steps, warmup, regions, preroll_batches, profile_stages, alg_bytes_per_frame, kernel):
    """Setup (not a step): past the 29-frame warm-up of smoothingRadius 30 (every later push produces a frame) and
    `preroll_batches` batches of the same work so that the device's clocks and the allocator's first touches are not part
    of the W warm-up steps; then W warm-up steps and the timed regions."""
    BT = ss.BT
    ss.push(64 + preroll_batches * BT)
    ss.sync()
    comm.device_sync()
    ss.push(warmup * BT)
    ss.set_profiling(2 if profile_stages else 1)
    frames_before = ss.frames_out()
    elapsed = timed_regions(ss, comm, steps, regions)
    frames_out_timed = ss.frames_out() - frames_before
    assert frames_out_timed == steps * BT * len(ss.stabs) * regions, "timed steps did not all produce frames"
    stage_ms, stage_n = ss.stage_totals()
    roof = warp_roofline(ss, alg_bytes_per_frame, frames_out_timed, kernel, stage_ms, stage_n)
    return elapsed, roof, stage_ms, stage_n


def config2(vs, comm, device, args, full=True):
    """BASELINE configs[2]: 3840x2160, 400 corners, RollCorrection + AutoZoomCrop enabled."""
    W, H = 3840, 2160
    out = {"workload": "configs[2]: 1 stream 3840x2160, 400 corners, 3-level LK 21x21; frames resident in HBM"}
    p = make_params(vs, max_corners=400)
    # (a) the stream as a decoder hands it over: NV12 surfaces, batch mode.  64 distinct surfaces (796 MB) in a cycle.
    NF = int(os.environ.get("VS_BENCH_4K_CLIP", "64"))
    clip = synth.make_clip_dev(vs, synth.SEED_CONFIG3, W, H, NF, nv12=True)
    BT = int(os.environ.get("VS_BENCH_4K_BATCH", "64"))
    ss = StreamSet(vs, device, p, [clip], NF, W, H, capi.FMT_NV12, BT, BT, True)
    steps = int(os.environ.get("VS_BENCH_4K_TIMED", "20"))
    regions = int(os.environ.get("VS_BENCH_4K_REGIONS", str(args.regions)))
    fb = W * H * 3 // 2
    elapsed, roof, _, _ = run_stream_workload(comm, ss, steps, 4, regions, int(os.environ.get("VS_BENCH_4K_WARM", "40")),
                                              False, 2.0 * fb, "warp_nv12_kernel (luma and interleaved chroma tiles of 32 surfaces in one launch)")
    med = statistics.median(elapsed)
    tables_pass(ss, roof, steps)
    traffic_from_counters(vs, roof, "configs2")
    roof["distinct_input_MB"] = round(NF * fb / 1e6, 1)
    out["nv12_stabilize"] = {"value": round(steps * BT / med, 1), "unit": "frames/s", "batch": BT, "steps": steps,
                             "regions": [round(steps * BT / e, 1) for e in elapsed], "roofline": roof}
    ss.close()
    if not full:
        return out
    # (b) configs[2] as ONE chain on the decoder surfaces: roll correction -> stabilize -> auto zoom/crop (the reference's order
    # of operators, examples/vs.cpp:553-562), each stage through its asynchronous device entry point, chunks of 128 surfaces:
    # while the roll stage works on chunk c, the stabilizer takes chunk c-1 and the zoom stage the stabilized surfaces of chunk
    # c-2; one host wait per stage and chunk.
    out["chain_nv12"] = config2_chain(vs, device, p, W, H)
    return out


def config2_chain(vs, device, p, W, H, chunks_timed=10, chunks_warm=5):
    CH = int(os.environ.get("VS_BENCH_CHAIN_CHUNK", "128"))
    NF = int(os.environ.get("VS_BENCH_4K_CLIP", "64"))
    clip = synth.make_clip_dev(vs, synth.SEED_CONFIG3, W, H, NF, nv12=True)
    sb = W * H * 3 // 2
    R_RING, S_RING, Z_RING = 4, 3, 2        # chunks a roll result / a stabilized surface / a zoom result stays untouched
    d_roll = [capi.DevBuf(vs, sb * CH) for _ in range(R_RING)]
    d_stab = [capi.DevBuf(vs, sb * CH) for _ in range(S_RING)]
    d_zoom = [capi.DevBuf(vs, sb * CH) for _ in range(Z_RING)]      # (room for the fall-back: the unchanged 4K surface)
    rc, az = vs.roll_correction(), vs.auto_zoom_crop()
    st = vs.stabilizer(p, device=device)
    st.set_batch(min(CH, 64))
    st.set_zero_copy(True)
    produced = {}                      # chunk -> stabilized surfaces it yielded
    tickets = []

    import threading
    busy = {"roll": 0.0, "stab": 0.0, "zoom": 0.0}       # host time of a stage's thread (its calls and its wait), timed chunks only
    timing = [False]

    def timed(name, f):
        def g(c):
            t = time.perf_counter()
            f(c)
            if timing[0]:
                busy[name] += time.perf_counter() - t
        return g

    def roll_stage(c):
        rc.correct_nv12_dev_n([clip.ptr + ((c * CH + i) % NF) * sb for i in range(CH)], W, H, W,
                              [d_roll[c % R_RING].ptr + i * sb for i in range(CH)], W)
        rc.sync()

    def stab_stage(c):                                                         # (the rotations of chunk c are complete)
        produced[c] = st.push_dev_n([d_roll[c % R_RING].ptr + i * sb for i in range(CH)], W, H, W, capi.FMT_NV12,
                                    [d_stab[c % S_RING].ptr + j * sb for j in range(CH)], W)
        st.sync()

    def zoom_stage(c):
        k = produced[c]                                                        # what the stabilizer let go during chunk c (complete)
        if k:
            tickets.extend(az.apply_nv12_dev_n([d_stab[c % S_RING].ptr + j * sb for j in range(k)], W, H, W,
                                               [d_zoom[c % Z_RING].ptr + j * sb for j in range(k)], W, W * H))
        az.sync()

    # One host thread per stage (a stage's calls queue a dozen launches per surface: three stages from one thread cost the sum of
    # their host times), free-running: a stage takes chunk c as soon as the stage before it has finished c and the ring slot it
    # writes is no longer read - roll(c) after stab(c - 3) (the stabilizer reads a surface until radius + 16 pushes later), stab(c)
    # after zoom(c - 3).  Timed: from the moment the zoom stage has finished chunk `chunks_warm - 1` to the moment it has finished
    # chunk `chunks_warm + chunks_timed - 1`, while the earlier stages keep working on two more chunks behind (the timed chunks run
    # in a full pipeline from end to end).
    total = chunks_warm + chunks_timed + 2
    done = {"roll": -1, "stab": -1, "zoom": -1}
    stamp = {}
    marks = {}
    cond = threading.Condition()
    failed = []
    import resource

    def wait_for(stage, c):
        with cond:
            cond.wait_for(lambda: done[stage] >= c or failed)
        return not failed

    def stage_loop(name, f, before, after, lag):
        try:
            g = timed(name, f)
            for c in range(total):
                if before and not wait_for(before, c):
                    return
                if after and not wait_for(after, c - lag):
                    return
                g(c)
                with cond:
                    done[name] = c
                    if name == "zoom" and c in (chunks_warm - 1, chunks_warm + chunks_timed - 1):
                        stamp[c] = time.perf_counter()
                        marks[c] = (len(tickets), resource.getrusage(resource.RUSAGE_SELF))
                        timing[0] = c == chunks_warm - 1
                    cond.notify_all()
        except BaseException as e:          # (a failed stage must not leave the others waiting)
            with cond:
                failed.append(e)
                cond.notify_all()

    ths = [threading.Thread(target=stage_loop, args=a) for a in (("roll", roll_stage, None, "stab", 3), ("stab", stab_stage, "roll", "zoom", 3),
                                                                 ("zoom", zoom_stage, "stab", None, 0))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if failed:
        raise failed[0]
    c0, c1 = chunks_warm - 1, chunks_warm + chunks_timed - 1
    dt = stamp[c1] - stamp[c0]
    n0, u0 = marks[c0]
    n1, u1 = marks[c1]
    n = n1 - n0
    ow, oh, info = az.result(tickets[-1])
    wt = az.worker_times()
    res = {"value": round(n / dt, 1), "unit": "frames/s", "ms_per_frame": round(dt / max(n, 1) * 1e3, 4), "frames": n,
           "last_result": [ow, oh], "last_crop": [int(v) for v in info[2:6]], "chunk": CH,
           "stage_thread_ms_per_chunk": {k: round(v / chunks_timed * 1e3, 3) for k, v in busy.items()},
           "zoom_worker_us_per_frame": {k: round(v / max(wt[0], 1) * 1e6, 1) for k, v in zip(("idle", "masks", "contour", "publish"), wt[1:5])},
           "zoom_batch_us": {k: round(v / max(wt[5], 1) * 1e6, 1) for k, v in zip(("launch_to_masks", "masks_to_crop", "caller_waits_for_slot"), wt[6:9])},
           "host_cores_busy": round((u1.ru_utime - u0.ru_utime + u1.ru_stime - u0.ru_stime) / dt, 2),
           "what": "vs_roll_correct_nv12_dev -> vs_stab_push_dev (batch 64, zero-copy) -> vs_azc_apply_nv12_dev on 3840x2160 NV12 surfaces "
                   "resident in HBM, chunks of %d, one free-running host thread per stage (a stage takes a chunk when the stage before it has finished it), one host wait per stage and chunk; every "
                   "surface goes through all three stages (640x360 NV12 out)" % CH}
    st.close()
    rc.close()
    az.close()
    for b in [clip] + d_roll + d_stab + d_zoom:
        b.free()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps per region; a step = one batch of --batch frames per stream")
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps in front of the first region")
    ap.add_argument("--regions", type=int, default=9, help="the timed region of --steps steps is repeated this often; value = the median")
    ap.add_argument("--streams", type=int, default=1, help="independent streams per GPU (batch mode)")
    ap.add_argument("--group", type=int, default=1,
                    help="--streams > 1: 1 = the streams of a GPU as ONE vs_batch group (one launch per stage over the frames of all "
                         "streams, steps of batch // streams frames per stream); 0 = independent vs_stab instances")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--clip-frames", type=int, default=128,
                    help="distinct frames per stream, played in a cycle (a multiple of 4; 128 x 6.2 MB = 796 MB at 1080p)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive and configs[2] measurements")
    ap.add_argument("--profile-stages", action="store_true", help="time every stage (adds event records)")
    ap.add_argument("--workload", default="configs1", choices=["configs1", "configs2"],
                    help="configs2: the 3840x2160 NV12 stream of BASELINE configs[2] alone (profiling runs of its kernels)")
    ap.add_argument("--zero-copy", type=int, default=1,
                    help="frames are read where they lie in HBM instead of being copied into the instance's queue")
    ap.add_argument("--batch", type=int, default=64,
                    help="frames per step (at most 64): the analysis stages of this many consecutive pushes run as one launch each "
                         "(1 = per-frame pipeline); the warps of a step go out 32 frames per launch")
    ap.add_argument("--warp-batch", type=int, default=8,
                    help="per-frame pipeline only (--batch 1): results of this many consecutive pushes are warped by one launch")
    ap.add_argument("--config", default=None,
                    help="a config.yaml of the reference's apps: its 'stabilizer' section replaces the configs[1] "
                         "parameters (not the headline workload any more: no cpu_baseline, the workload string says so)")
    ap.add_argument("--fanout", action="store_true",
                    help="N > 1 only: rank 0 owns ingest - it renders the clips of ALL streams and scatters them to the "
                         "owning ranks (RCCL send/recv over xGMI) before the timed region; timed separately, reported as `fanout`")
    args = ap.parse_args()

    comm = vsdist.Comm()           # nccl (= RCCL) when WORLD_SIZE > 1 (or VS_DIST_FORCE=1), nothing otherwise
    rank, local_rank, world = comm.rank, comm.local_rank, comm.world
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    vs = capi.load(os.environ.get("VS_LIB"))    # VS_LIB: another build of the library (A/B measurements)
    if vs.lib.vs_device_count() <= 0:
        raise SystemExit("bench.py: no GPU visible - libvideo-stab has no CPU fallback")
    if os.environ.get("VS_BENCH_DEVICE"):          # rehearsal: several ranks on one GPU (with VS_DIST_BACKEND=gloo)
        local_rank = int(os.environ["VS_BENCH_DEVICE"])
    vs.check(vs.lib.vs_dev_set_device(local_rank))

    if args.workload == "configs2":
        out = config2(vs, comm, local_rank, args, full=os.environ.get("VS_BENCH_CHAIN") == "1")   # (VS_BENCH_CHAIN=1: and the chain)
        if rank == 0:
            print(json.dumps(out), flush=True)
        comm.close()
        return

    W, H = args.width, args.height
    fb = W * H * 3
    S = args.streams
    NF = args.clip_frames
    # synthetic clips: global stream g (owned by rank g % world) uses seed base + g
    my_streams = vsdist.streams_of_rank(rank, n_gpus, S * n_gpus)
    assert len(my_streams) == S
    fanout = None
    if args.fanout and world > 1:
        # ingest on rank 0: it renders every stream's clip on its own GPU; the frame payloads travel to the
        # owning GPUs with one scatter, timed on its own
        payloads = None
        if rank == 0:
            payloads = []
            for r in range(world):
                parts = []
                for g in vsdist.streams_of_rank(r, n_gpus, S * n_gpus):
                    c = synth.make_clip_dev(vs, synth.SEED_CONFIG2 + g, W, H, NF)
                    parts.append(c.download((NF * fb,), np.uint8))
                    c.free()
                payloads.append(np.concatenate(parts))
        nbytes = fb * NF * S
        recv, secs = comm.fan_out(payloads, nbytes)
        clips = []
        for j in range(S):
            c = capi.DevBuf(vs, NF * fb)
            if comm.device == "cuda":       # device to device: out of the tensor RCCL filled, into the stream's clip
                vs.check(vs.lib.vs_dev_memcpy_d2d(c.ptr, recv.data_ptr() + j * NF * fb, NF * fb))
            else:
                c.upload(recv.numpy()[j * NF * fb:(j + 1) * NF * fb])
            clips.append(c)
        sent = nbytes * (world - 1)
        fanout = {"bytes": sent, "ms": round(secs * 1e3, 3), "GBps": round(sent / secs / 1e9, 2) if secs > 0 else None,
                  "what": "scatter of %d frames per stream from rank 0 to %d ranks (torch.distributed.scatter, %s)" % (
                      NF, world - 1, "RCCL" if comm.device == "cuda" else "gloo, host memory")}
    else:
        # (VS_BENCH_SAME_CLIP=1: every stream plays the clip of stream 0 - scheduling comparisons between stream counts without
        # the differences in content: the candidate count of the corner detector varies by a factor of two between seeds)
        same = os.environ.get("VS_BENCH_SAME_CLIP") == "1"
        clips = [synth.make_clip_dev(vs, synth.SEED_CONFIG2 + (0 if same else g), W, H, NF) for g in my_streams]

    params = make_params(vs, args.config)
    ss = StreamSet(vs, local_rank, params, clips, NF, W, H, capi.FMT_BGR8, args.batch, args.warp_batch, args.zero_copy, group=args.group)
    BT = ss.BT
    elapsed, roof, stage_ms, stage_n = run_stream_workload(
        comm, ss, args.steps, args.warmup, args.regions, int(os.environ.get("VS_BENCH_PREROLL_BATCHES", "200")),
        args.profile_stages, 2.0 * fb, "warp_tab_kernel" if BT >= 4 else "warp_affine_kernel<3>")
    med = statistics.median(elapsed)
    # throughput counters of every rank (the only inter-GPU traffic of the path)
    per_rank = comm.gather_counters([rank, args.steps * BT * S, ss.frames_out()])

    out = None
    if rank == 0:
        tables_pass(ss, roof, args.steps)
        traffic_from_counters(vs, roof, "configs1")
        roof["copy_yardstick_GBps"] = copy_yardstick(vs, roof["bytes_per_launch"] / 2)
        roof["distinct_input_MB"] = round(NF * fb / 1e6, 1)
        total_frames = args.steps * BT * S * n_gpus
        whole_bytes = WHOLE_STEP_BYTES_1080P * (W * H) / (1920.0 * 1080.0)
        whole = whole_bytes * total_frames / med / 1e9
        out = {
            "metric": "stabilized frames/sec @1080p (whole job; warp-stage HBM GB/s in roofline)",
            "value": round(total_frames / med, 2),
            "unit": "frames/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(med / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": ("configs[1]: %d stream(s)/GPU %dx%d BGR8, 200 corners, 3-level LK 21x21, "
                                    "RANSAC partial affine, warpAffine; frames resident in HBM" % (S, W, H))
                       if not args.config else
                       ("custom: %d stream(s)/GPU %dx%d BGR8, stabilizer parameters from %s; frames resident in HBM"
                        % (S, W, H, os.path.basename(args.config))),
                       "streams_per_gpu": S, "frames_per_step": BT, "step": "one batch of %d stabilize() calls per stream" % BT,
                       "streams_grouped": ss.grouped,
                       "clip": "%d distinct frames per stream rendered on the device, played in a cycle" % NF,
                       "warp_batch": ss.WB, "zero_copy": bool(args.zero_copy),
                       "timed_frames_per_region_per_rank": [int(r[1]) for r in per_rank], "timed_regions": args.regions, "build": vs.lib.vs_build_tag().decode()},
            "regions": {"count": args.regions, "value": "median", "frames_per_s": [round(total_frames / e, 1) for e in elapsed]},
            "roofline": roof,
            "whole_step": {"bytes_per_frame": round(whole_bytes), "achieved": round(whole, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": round(whole / HBM_PEAK_GBPS, 4),
                           "note": "SURVEY 8d: algorithmic bytes of the whole step (gray, pyramid, GFTT every 2nd frame, LK gathers, "
                                   "warp) x frames / elapsed"},
        }
        if fanout is not None:
            out["fanout"] = fanout
        if args.profile_stages:
            names = ["copy_in", "gray", "pyramid", "lk", "ransac", "traj", "gftt", "warp", "warp_tables"]
            out["stage_us_per_launch"] = {names[k]: round(stage_ms[k] / max(stage_n[k], 1) * 1e3, 2) for k in range(capi.STAGE_COUNT)}
            out["stage_launches"] = {names[k]: stage_n[k] for k in range(capi.STAGE_COUNT)}
    baseline = rank == 0 and n_gpus == 1 and not args.no_cpu_baseline and not args.config
    # (the CPU leg replays the FIRST pushes of the timed clip in the timed order: as many distinct frames as it pushes)
    host_frames = ss.host_frames(0, 124 if baseline else 12) if rank == 0 and n_gpus == 1 and not args.config else []
    ss.close(free_clips=False)
    if rank == 0:
        if baseline:
            N_CHECK = 48
            order, ref, out["cpu_baseline"] = cpu_baseline(W, H, host_frames, lambda nf, n: [i % nf for i in range(n)], keep=N_CHECK)
            out["outputs_checked"] = check_outputs(vs, local_rank, params, clips[0], W, H, max(1, min(64, args.batch)), order, ref)
            out["outputs_note"] = ("the first %d stabilized frames of the timed clip (stream 0, the timed push order and settings: batch %d, "
                                   "zero-copy) equal the frames the oracle produced for the same pushes in the cpu_baseline leg, byte for byte"
                                   % (out["outputs_checked"], max(1, min(64, args.batch))))
            del ref
        if n_gpus == 1 and not args.no_extras and not args.config:
            out["with_pcie"] = host_api_rate(vs, local_rank, params, host_frames[:12])
            out["with_pcie"]["cpp_class"] = cpp_class_rate(local_rank)
    for c in clips:
        c.free()
    if rank == 0:
        if n_gpus == 1 and not args.no_extras and not args.config:
            out["configs"] = {"configs[2]": config2(vs, comm, local_rank, args)}
        print(json.dumps(out), flush=True)
    comm.close()


if __name__ == "__main__":
    main()
