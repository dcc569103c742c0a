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

A step = one stabilize() (vs_stab_push_dev) per stream on a frame that is
already resident in HBM.  Prints ONE JSON line on rank 0.
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


def make_params(vs, config=None):
    # BASELINE.json configs[1]: 200 corners, 3-level LK (maxLevel 2) with a 21x21 window;
    # everything else is the reference's live default (Stabilizer.h:76-175, Stabilizer.cpp:611-649).
    p = vs.params(max_corners=200, lk_win_size=21, lk_max_level=2, lk_max_iters=20, lk_epsilon=0.03,
                  smoothing_radius=30)
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
    # all-core variant of the same port (point-parallel LK, row-parallel warp)
    ncores = os.cpu_count() or 1
    o.lib.vso_set_threads(ncores)
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
        "all_cores": {"value": round(60 / dt_mt, 3), "cores": ncores,
                      "note": "same port, LK points and warp rows spread over threads"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=640)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--streams", type=int, default=1, help="independent streams per GPU (batch mode)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--clip-frames", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-stages", action="store_true", help="time every stage (adds event records)")
    ap.add_argument("--zero-copy", type=int, default=1,
                    help="frames are read where they lie in HBM instead of being copied into the instance's queue")
    ap.add_argument("--batch", type=int, default=32,
                    help="batch mode: analysis stages of this many consecutive pushes run as one launch each (1 = per-frame pipeline)")
    ap.add_argument("--warp-batch", type=int, default=8,
                    help="deferred output: results of this many consecutive pushes are warped by one launch (1 = one launch per push)")
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
    clips, d_in = [], []
    my_streams = vsdist.streams_of_rank(rank, n_gpus, S * n_gpus)
    assert len(my_streams) == S
    fanout = None
    if args.fanout and world > 1:
        # ingest on rank 0: the frame payloads travel to the owning GPUs with one scatter; the received
        # device memory is what the stabilizers read (zero-copy), so the timed region is unchanged
        import numpy as np
        payloads = None
        if rank == 0:
            payloads = [np.concatenate([np.stack(synth.make_clip(synth.SEED_CONFIG2 + g, W, H, args.clip_frames)).reshape(-1)
                                        for g in vsdist.streams_of_rank(r, n_gpus, S * n_gpus)]) for r in range(world)]
        nbytes = fb * args.clip_frames * S
        recv, secs = comm.fan_out(payloads, nbytes)

        class _View:                      # same interface as DevBuf for the step loop; `recv` keeps the memory alive
            def __init__(self, ptr):
                self.ptr = ptr
        if comm.device == "cuda":
            base = recv.data_ptr()
        else:                             # gloo rehearsal: the payload arrived in host memory
            staged = capi.DevBuf.from_array(vs, recv.numpy())
            base = staged.ptr
        for j in range(S):
            d_in.append(_View(base + j * fb * args.clip_frames))
        if rank == 0:
            clips.append(list(payloads[0][:fb * args.clip_frames].reshape(args.clip_frames, H, W, 3)))
        sent = nbytes * (world - 1)
        fanout = {"bytes": sent, "ms": round(secs * 1e3, 3), "GBps": round(sent / secs / 1e9, 2) if secs > 0 else None,
                  "what": "scatter of %d frames per stream from rank 0 to %d ranks (torch.distributed.scatter, %s)" % (
                      args.clip_frames, world - 1, "RCCL" if comm.device == "cuda" else "gloo, host memory")}
    else:
        for g in my_streams:
            seed = synth.SEED_CONFIG2 + g
            frames = synth.make_clip(seed, W, H, args.clip_frames)
            clips.append(frames)
            buf = capi.DevBuf(vs, fb * len(frames))
            for i, f in enumerate(frames):
                buf.upload(f, i * fb)
            d_in.append(buf)
    BT = max(1, min(32, args.batch))
    WB = max(1, min(32, args.warp_batch if BT == 1 else BT))
    NOUT = max(2 * WB, 3 * BT)        # a result stays untouched until its batch and the next one have been issued
    d_out = [[capi.DevBuf(vs, fb) for _ in range(NOUT)] for _ in range(S)]
    stabs = [vs.stabilizer(make_params(vs, args.config), device=local_rank) for _ in range(S)]
    for s in stabs:
        s.set_batch(BT)
        s.set_zero_copy(bool(args.zero_copy))
        s.set_warp_batch(WB)

    preroll = 64   # past the 29-frame warm-up of smoothingRadius 30: every timed step produces a frame
    total = preroll + args.warmup + args.steps
    order = clip_order(args.clip_frames, total)

    def step(i):
        for j in range(S):
            stabs[j].push_dev(d_in[j].ptr + order[i] * fb, W, H, W * 3, capi.FMT_BGR8, d_out[j][i % NOUT].ptr, W * 3)

    def sync_all():
        for s in stabs:
            s.sync()
        comm.device_sync()

    barrier = comm.barrier

    for i in range(preroll):
        step(i)
    sync_all()
    for i in range(preroll, preroll + args.warmup):
        step(i)
    for s in stabs:
        s.set_profiling(2 if args.profile_stages else 1)
        s.stage_times()          # drop anything recorded so far
    sync_all()
    barrier()
    sync_all()
    frames_before = sum(s.counters().frames_out for s in stabs)
    t0 = time.perf_counter()
    for i in range(preroll + args.warmup, total):
        step(i)
    sync_all()
    barrier()
    sync_all()
    elapsed = time.perf_counter() - t0

    elapsed = comm.max_over_ranks(elapsed)

    # per-stage device time of the timed region (HIP events on the instance streams)
    stage_ms = [0.0] * 8
    stage_n = [0] * 8
    for s in stabs:
        ms, n = s.stage_times()
        for k in range(8):
            stage_ms[k] += ms[k]
            stage_n[k] += n[k]
    frames_out = sum(s.counters().frames_out for s in stabs)
    frames_out_timed = frames_out - frames_before
    assert frames_out_timed == args.steps * S, "timed steps did not all produce frames"
    # throughput counters of every rank (the only inter-GPU traffic of the path)
    per_rank = comm.gather_counters([rank, args.steps * S, frames_out])

    if rank == 0:
        frames_per_launch = frames_out_timed / max(stage_n[7], 1)   # WB when every launch is full
        warp_bytes = 2.0 * fb * frames_per_launch                # algorithmic bytes per launch (2 x frame bytes per frame)
        warp_avg_ms = stage_ms[7] / max(stage_n[7], 1)
        achieved = warp_bytes / (warp_avg_ms * 1e-3) / 1e9 if warp_avg_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "warp_traffic.json")
        if os.path.exists(pmc):
            try:
                t = json.load(open(pmc))
                # measured on full launches of t["frames_per_launch"] frames of 1920x1080 BGR8; scale to this run's launch
                traffic = round(t["hbm_bytes_per_launch"] / t["frames_per_launch"] * frames_per_launch * (fb / (1920 * 1080 * 3.0)))
            except Exception:
                traffic = None
        total_frames = args.steps * S * n_gpus
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
                       "streams_per_gpu": S, "batch": BT, "warp_batch": WB, "zero_copy": bool(args.zero_copy),
                       "timed_frames_per_rank": [int(r[1]) for r in per_rank]},
            "roofline": {"bound": "hbm", "kernel": "warp_affine_kernel<3>", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "bytes_per_launch": warp_bytes,
                         "frames_per_launch": round(frames_per_launch, 3),
                         "avg_launch_us": round(warp_avg_ms * 1e3, 3), "launches": stage_n[7]},
        }
        if fanout is not None:
            out["fanout"] = fanout
        if args.profile_stages:
            names = ["copy_in", "gray", "pyramid", "lk", "ransac", "traj", "gftt", "warp"]
            out["stage_us_per_launch"] = {names[k]: round(stage_ms[k] / max(stage_n[k], 1) * 1e3, 2) for k in range(8)}
            out["stage_launches"] = {names[k]: stage_n[k] for k in range(8)}
        if n_gpus == 1 and not args.no_cpu_baseline and not args.config:
            out["cpu_baseline"] = cpu_baseline(W, H, clips[0], clip_order)
        print(json.dumps(out), flush=True)

    for s in stabs:
        s.close()
    comm.close()


if __name__ == "__main__":
    main()
