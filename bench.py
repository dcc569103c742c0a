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
Prints ONE JSON line on rank 0; at N = 1 it also carries the PCIe-inclusive rate
of the host-pointer entry point, BASELINE configs[2] (3840x2160) and the CPU
baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "video-stab_amd"))

from vsamd import capi, dist as vsdist, synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured copy rate


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
    """Ping-pong through the clip so consecutive frames always differ by one camera step."""
    fwd = list(range(n_frames)) + list(range(n_frames - 2, 0, -1))
    return [fwd[i % len(fwd)] for i in range(n_steps)]


def cpu_baseline(width, height, frames, order_fn):
    """The oracle (CPU restatement of src/Stabilizer.cpp, kind "port") on the same workload,
    single thread, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    o = oracle_lib.load()
    p = o.params(max_corners=200, lk_win_size=21, lk_max_level=2, lk_max_iters=20, lk_epsilon=0.03,
                 smoothing_radius=30)
    o.lib.vso_set_threads(1)
    s = o.stabilizer(p)
    warm, timed = 34, 90
    order = order_fn(len(frames), warm + timed)
    for i in order[:warm]:
        s.push(frames[i])
    t0 = time.perf_counter()
    n_out = 0
    for i in order[warm:]:
        if s.push(frames[i]) is not None:
            n_out += 1
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
    return {
        "value": round(timed / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": "%d steady-state stabilize() calls on the same %dx%d clip, oracle/ single thread" % (timed, width, height),
        "all_cores": {"value": round(60 / dt_mt, 3), "cores": nthr, "host_cores": ncores,
                      "note": "same port, threaded stages spread over %d threads" % nthr},
    }


class StreamSet:
    """S stabilizer instances in batch mode on one GPU, each with a resident clip and a ring of output frames."""

    def __init__(self, vs, device, params, frames_per_stream, w, h, fmt, batch, warp_batch, zero_copy):
        self.vs, self.w, self.h, self.fmt = vs, w, h, fmt
        self.fb = frames_per_stream[0][0].nbytes
        self.stride = w * 3 if fmt == capi.FMT_BGR8 else w
        self.clip_frames = len(frames_per_stream[0])
        self.d_in = []
        for frames in frames_per_stream:
            buf = capi.DevBuf(vs, self.fb * len(frames))
            for i, f in enumerate(frames):
                buf.upload(f, i * self.fb)
            self.d_in.append(buf)
        self.BT = max(1, min(64, batch))
        self.WB = max(1, min(32, warp_batch if self.BT == 1 else self.BT))
        self.WB_frames = self.BT if self.BT > 1 else self.WB       # frames between one pair of warp-stage events
        self.NOUT = max(2 * self.WB, 3 * self.BT)    # a result stays untouched until its batch and the next one have been issued
        self.d_out = [[capi.DevBuf(vs, self.fb) for _ in range(self.NOUT)] for _ in frames_per_stream]
        self.stabs = [vs.stabilizer(params, device=device) for _ in frames_per_stream]
        for s in self.stabs:
            s.set_batch(self.BT)
            s.set_zero_copy(bool(zero_copy))
            s.set_warp_batch(self.WB)
        self.i = 0
        self.order = clip_order(self.clip_frames, 1 << 17)

    def push(self, n):
        """n consecutive pushes per stream."""
        for _ in range(n):
            fi = self.order[self.i % len(self.order)]
            for j, s in enumerate(self.stabs):
                s.push_dev(self.d_in[j].ptr + fi * self.fb, self.w, self.h, self.stride, self.fmt,
                           self.d_out[j][self.i % self.NOUT].ptr, self.stride)
            self.i += 1

    def sync(self):
        for s in self.stabs:
            s.sync()

    def frames_out(self):
        return sum(s.counters().frames_out for s in self.stabs)

    def close(self):
        for s in self.stabs:
            s.close()
        for b in self.d_in:
            b.free()
        for bs in self.d_out:
            for b in bs:
                b.free()


def warp_roofline(ss, alg_bytes_per_frame, frames_out_timed, kernel):
    """Warp stage of the timed region: HIP-event time of its launches on the stream they run on."""
    stage_ms = [0.0] * 8
    stage_n = [0] * 8
    for s in ss.stabs:
        ms, n = s.stage_times()
        for k in range(8):
            stage_ms[k] += ms[k]
            stage_n[k] += n[k]
    # (a batch of more than 32 frames is warped by ceil(batch / 32) launches back to back inside one pair of events)
    launches = max(stage_n[7], 1) * max(1, (ss.WB_frames + 31) // 32)
    frames_per_launch = frames_out_timed / launches
    avg_ms = stage_ms[7] / launches
    byts = alg_bytes_per_frame * frames_per_launch
    achieved = byts / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    roof = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "bytes_per_launch": byts,
            "frames_per_launch": round(frames_per_launch, 3), "avg_launch_us": round(avg_ms * 1e3, 3), "launches": launches}
    return roof, stage_ms, stage_n


def traffic_from_counters(vs, roof, frame_bytes):
    """HBM bytes per launch from the PMC passes kept in profiles/warp_traffic.json - only when that file was measured on
    THIS build of the kernels (vs_build_tag); otherwise the field stays null and says why."""
    pmc = os.path.join(ROOT, "profiles", "warp_traffic.json")
    try:
        t = json.load(open(pmc))
    except Exception:
        roof["traffic_note"] = "no counter file"
        return
    tag = vs.lib.vs_build_tag().decode()
    if t.get("build_tag") != tag:
        roof["traffic_note"] = "stale: profiles/warp_traffic.json was measured on build %s, this is %s" % (t.get("build_tag"), tag)
        return
    per_frame = t["hbm_bytes_per_launch"] / t["frames_per_launch"] * (frame_bytes / (1920 * 1080 * 3.0))
    roof["traffic"] = round(per_frame * roof["frames_per_launch"])
    roof["traffic_note"] = "FETCH_SIZE x2 + WRITE_SIZE of %s (%d-frame launches), scaled to this launch" % (t.get("source"), t["frames_per_launch"])


def host_api_rate(vs, device, params, frames, n_timed=240):
    """vs_stab_push: host frame in, host frame out (what vs::Stabilizer::stabilize(cv::Mat) calls) - PCIe both ways.
    Frames in pageable memory (numpy arrays as they come: what a cv::Mat holds) or page-locked (vs_host_alloc); the
    synchronous call or the host pipeline (vs_stab_set_host_pipeline, VS_STAB_HOST_PIPELINE=1 for the C++ class: a call
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


def config2(vs, device, args):
    """BASELINE configs[2]: 3840x2160, 400 corners, RollCorrection + AutoZoomCrop enabled."""
    W, H = 3840, 2160
    out = {"workload": "configs[2]: 1 stream 3840x2160, 400 corners, 3-level LK 21x21; frames resident in HBM"}
    bgr = synth.make_clip(synth.SEED_CONFIG3, W, H, 6)
    p = make_params(vs, max_corners=400)
    # (a) the stream as a decoder hands it over: NV12 surfaces, batch mode
    nv = [synth.bgr_to_nv12(f) for f in bgr]
    # batches of 64 like configs[1], and the same kind of run-in (40 batches: the device's clocks are still rising during the
    # first ones - 16-frame batches behind 2 warm-up batches, the first form of this measurement, read 45 k frames/s where
    # batches of 32 behind 40 read 61 k and batches of 64 67 k; scratch/config2_sweep.py).  VS_BENCH_4K_* override the three numbers.
    BT = int(os.environ.get("VS_BENCH_4K_BATCH", "64"))
    ss = StreamSet(vs, device, p, [nv], W, H, capi.FMT_NV12, BT, BT, True)
    ss.push(64)
    ss.sync()
    ss.push(int(os.environ.get("VS_BENCH_4K_WARM", "40")) * BT)
    for s in ss.stabs:
        s.set_profiling(1)
        s.stage_times()
    ss.sync()
    f0 = ss.frames_out()
    nb = int(os.environ.get("VS_BENCH_4K_TIMED", "40"))
    t0 = time.perf_counter()
    ss.push(nb * BT)
    ss.sync()
    dt = time.perf_counter() - t0
    fo = ss.frames_out() - f0
    roof, _, _ = warp_roofline(ss, 2.0 * nv[0].nbytes, fo, "warp_plane_kernel<1> + <2> (Y and interleaved UV plane)")
    out["nv12_stabilize"] = {"value": round(fo / dt, 1), "unit": "frames/s", "batch": BT, "timed_frames": fo, "roofline": roof}
    ss.close()
    # (b) the reference's order of operators on a 4K BGR frame: roll correction -> stabilize -> auto zoom/crop (each
    # call finished before the next: the reference's loop is synchronous, examples/roll-correction-file.cpp:58-70)
    fb = bgr[0].nbytes
    d_f = capi.DevBuf(vs, fb * len(bgr))
    for i, f in enumerate(bgr):
        d_f.upload(f, i * fb)
    d_r, d_s, d_z = capi.DevBuf(vs, fb), capi.DevBuf(vs, fb), capi.DevBuf(vs, fb)   # (the zoom stage returns the frame as it is when it finds no crop)
    rc, az = vs.roll_correction(), vs.auto_zoom_crop()
    st = vs.stabilizer(p, device=device)
    order = clip_order(len(bgr), 40 + 100)

    def one(i):
        rc.correct_dev(d_f.ptr + order[i] * fb, W, H, W * 3, d_r.ptr, W * 3)
        rc.sync()
        k = st.push_dev(d_r.ptr, W, H, W * 3, capi.FMT_BGR8, d_s.ptr, W * 3)
        st.sync()
        if k:
            az.apply_dev(d_s.ptr, W, H, W * 3, 3, d_z.ptr, W * 3)
            az.sync()
        return k
    for i in range(40):
        one(i)
    t0 = time.perf_counter()
    n = sum(one(i) for i in range(40, 140))
    dt = time.perf_counter() - t0
    out["roll_stabilize_zoomcrop_bgr"] = {"value": round(n / dt, 1), "unit": "frames/s", "ms_per_frame": round(dt / max(n, 1) * 1e3, 3),
                                          "what": "autoCorrectRoll -> stabilize -> autoZoomCrop per 4K BGR frame, one frame at a time"}
    st.close()
    for b in (d_f, d_r, d_s, d_z):
        b.free()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed steps; a step = one batch of --batch frames per stream")
    ap.add_argument("--warmup", type=int, default=8, help="untimed steps in front of them")
    ap.add_argument("--streams", type=int, default=1, help="independent streams per GPU (batch mode)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--clip-frames", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive and configs[2] measurements")
    ap.add_argument("--profile-stages", action="store_true", help="time every stage (adds event records)")
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
                    help="N > 1 only: rank 0 owns ingest - it generates the clips of ALL streams and scatters them to the "
                         "owning ranks (RCCL send/recv over xGMI) before the timed region; timed separately, reported as `fanout`")
    args = ap.parse_args()

    comm = vsdist.Comm()           # nccl (= RCCL) when WORLD_SIZE > 1, nothing otherwise
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

    W, H = args.width, args.height
    fb = W * H * 3
    S = args.streams
    # synthetic clips: global stream g (owned by rank g % world) uses seed base + g
    my_streams = vsdist.streams_of_rank(rank, n_gpus, S * n_gpus)
    assert len(my_streams) == S
    fanout = None
    clips = [synth.make_clip(synth.SEED_CONFIG2 + g, W, H, args.clip_frames) for g in my_streams]
    if args.fanout and world > 1:
        # ingest on rank 0: the frame payloads travel to the owning GPUs with one scatter, timed on its own
        payloads = None
        if rank == 0:
            payloads = [np.concatenate([np.stack(synth.make_clip(synth.SEED_CONFIG2 + g, W, H, args.clip_frames)).reshape(-1)
                                        for g in vsdist.streams_of_rank(r, n_gpus, S * n_gpus)]) for r in range(world)]
        nbytes = fb * args.clip_frames * S
        recv, secs = comm.fan_out(payloads, nbytes)
        got = (recv.cpu().numpy() if comm.device == "cuda" else recv.numpy()).reshape(S, args.clip_frames, H, W, 3)
        clips = [list(got[j]) for j in range(S)]
        sent = nbytes * (world - 1)
        fanout = {"bytes": sent, "ms": round(secs * 1e3, 3), "GBps": round(sent / secs / 1e9, 2) if secs > 0 else None,
                  "what": "scatter of %d frames per stream from rank 0 to %d ranks (torch.distributed.scatter, %s)" % (
                      args.clip_frames, world - 1, "RCCL" if comm.device == "cuda" else "gloo, host memory")}

    params = make_params(vs, args.config)
    ss = StreamSet(vs, local_rank, params, clips, W, H, capi.FMT_BGR8, args.batch, args.warp_batch, args.zero_copy)
    BT = ss.BT
    # setup, not a step: past the 29-frame warm-up of smoothingRadius 30 (every later push produces a frame), and ~0.1 s of the
    # same work so that the device's clocks and the allocator's first touches are not part of the W warm-up steps' job
    preroll = 64 + int(os.environ.get("VS_BENCH_PREROLL_BATCHES", "200")) * BT

    def sync_all():
        ss.sync()
        comm.device_sync()

    ss.push(preroll)
    sync_all()
    ss.push(args.warmup * BT)
    for s in ss.stabs:
        s.set_profiling(2 if args.profile_stages else 1)
        s.stage_times()          # drop anything recorded so far
    sync_all()
    comm.barrier()
    sync_all()
    frames_before = ss.frames_out()
    t0 = time.perf_counter()
    ss.push(args.steps * BT)
    sync_all()
    comm.barrier()
    sync_all()
    elapsed = time.perf_counter() - t0
    elapsed = comm.max_over_ranks(elapsed)

    frames_out_timed = ss.frames_out() - frames_before
    assert frames_out_timed == args.steps * BT * S, "timed steps did not all produce frames"
    roof, stage_ms, stage_n = warp_roofline(ss, 2.0 * fb, frames_out_timed,
                                            "warp_tab_kernel (+ warp_tables_kernel)" if BT >= 4 else "warp_affine_kernel<3>")
    # throughput counters of every rank (the only inter-GPU traffic of the path)
    per_rank = comm.gather_counters([rank, args.steps * BT * S, ss.frames_out()])

    if rank == 0:
        traffic_from_counters(vs, roof, fb)
        total_frames = args.steps * BT * S * n_gpus
        out = {
            "metric": "stabilized frames/sec @1080p (whole job; warp-stage HBM GB/s in roofline)",
            "value": round(total_frames / elapsed, 2),
            "unit": "frames/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
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
                       "warp_batch": ss.WB, "zero_copy": bool(args.zero_copy),
                       "timed_frames_per_rank": [int(r[1]) for r in per_rank], "build": vs.lib.vs_build_tag().decode()},
            "roofline": roof,
        }
        if fanout is not None:
            out["fanout"] = fanout
        if args.profile_stages:
            names = ["copy_in", "gray", "pyramid", "lk", "ransac", "traj", "gftt", "warp"]
            out["stage_us_per_launch"] = {names[k]: round(stage_ms[k] / max(stage_n[k], 1) * 1e3, 2) for k in range(8)}
            out["stage_launches"] = {names[k]: stage_n[k] for k in range(8)}
    ss.close()
    if rank == 0:
        if n_gpus == 1 and not args.no_extras and not args.config:
            out["with_pcie"] = host_api_rate(vs, local_rank, params, clips[0])
            out["configs"] = {"configs[2]": config2(vs, local_rank, args)}
        if n_gpus == 1 and not args.no_cpu_baseline and not args.config:
            out["cpu_baseline"] = cpu_baseline(W, H, clips[0], clip_order)
        print(json.dumps(out), flush=True)
    comm.close()


if __name__ == "__main__":
    main()
