"""End-to-end: vs_stab_push() (HIP pipeline) vs the oracle's restated
vs::Stabilizer on the same synthetic clips, frame by frame.

Per frame we compare: the analysis gray image, keypoints handed to LK, LK
positions and status, the inlier mask and the hypothesis kept by RANSAC (all
bit-exact), the refined model (bit-exact, double), the measured transform
(dx, dy, da), the smoothed transform, the warp matrix and the stabilized frame -
ALL bit for bit: the device evaluates cosf / sinf / atan2f with glibc's own
algorithms (vs_libm.h; tests/test_libm.py holds them against the host libm on
every float), so nothing on the path differs from the oracle in a last place.
"""
import numpy as np
import pytest

from vsamd import capi, synth

pytestmark = pytest.mark.gpu

def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def run_both(gpu, oracle, clip, fmt=0, **params):
    ps = gpu.params(**params)
    po = oracle.params(**params)
    sg = gpu.stabilizer(ps)
    so = oracle.stabilizer(po)
    n_out = 0
    worst = 0.0
    for k, f in enumerate(clip):
        og = sg.push(f, fmt)
        oo = so.push(f, fmt)
        dg, do = sg.debug(), so.debug()
        ag, ao = sg.debug_arrays(), so.debug_arrays()
        assert (og is None) == (oo is None), k
        assert np.array_equal(ag["gray"], ao["gray"]), k
        assert (dg.n_prev, dg.n_valid) == (do.n_prev, do.n_valid), k
        assert np.array_equal(ag["prev"], ao["prev"]), k
        assert np.array_equal(ag["status"], ao["status"]), k
        assert np.array_equal(ag["curr"].view(np.uint32), ao["curr"].view(np.uint32)), k
        if k > 0:
            assert (dg.ransac_best_iter, dg.ransac_iters_run, dg.n_inliers) == \
                   (do.ransac_best_iter, do.ransac_iters_run, do.n_inliers), k
            assert np.array_equal(ag["inliers"], ao["inliers"]), k
            mg, mo = np.array(dg.model), np.array(do.model)
            assert np.array_equal(np.isnan(mg), np.isnan(mo)), k
            if not np.isnan(mo).any():
                assert np.array_equal(mg.view(np.uint64), mo.view(np.uint64)), k
            tg, to = np.array(dg.transform), np.array(do.transform)
            assert np.array_equal(bits(tg), bits(to)), k
        assert dg.detected == do.detected and dg.n_detected == do.n_detected, k
        assert np.array_equal(ag["detected"], ao["detected"]), k
        if oo is not None:
            n_out += 1
            assert dg.out_index == do.out_index, k
            assert dg.box_radius == do.box_radius and dg.intent == do.intent, k
            assert np.array_equal(bits(dg.smoothed), bits(do.smoothed)), k
            assert np.array_equal(bits(dg.warp_matrix), bits(do.warp_matrix)), k
            assert np.array_equal(og, oo), k
    while True:
        og = sg.flush(clip[0], fmt)
        oo = so.flush(clip[0], fmt)
        assert (og is None) == (oo is None)
        if oo is None:
            break
        n_out += 1
        assert np.array_equal(og, oo)
    sg.close()
    so.close()
    return n_out, worst


def test_pipeline_box_default(gpu, oracle):
    clip = synth.make_clip(synth.SEED_CONFIG1, 320, 240, 30)
    n_out, worst = run_both(gpu, oracle, clip, smoothing_radius=8)
    assert n_out == 30


def test_pipeline_intent_segments(gpu, oracle):
    """fast pan / rotation jitter / direction changes: exercises all four intent gains."""
    segs = [(0, 512, 0, 384, 200), (18, 2200, 0, 60, 20), (40, 200, 0, 40, 900), (56, 1000, 900, 700, 100)]
    clip = synth.make_clip(synth.SEED_CONFIG1 + 1, 320, 240, 72, segments=segs)
    n_out, worst = run_both(gpu, oracle, clip, smoothing_radius=6)
    assert n_out == 72


@pytest.mark.parametrize("method", [capi.SMOOTH_GAUSSIAN, capi.SMOOTH_KALMAN])
def test_pipeline_other_smoothers(gpu, oracle, method):
    clip = synth.make_clip(synth.SEED_CONFIG1 + 2, 256, 192, 24)
    n_out, _ = run_both(gpu, oracle, clip, smoothing_radius=5, smoothing_method=method)
    assert n_out == 24


def test_pipeline_lk21_and_horizon_lock(gpu, oracle):
    clip = synth.make_clip(synth.SEED_CONFIG1 + 3, 320, 240, 16)
    run_both(gpu, oracle, clip, smoothing_radius=5, lk_win_size=21, horizon_lock=1, max_corners=120)


def test_pipeline_border_modes(gpu, oracle):
    clip = synth.make_clip(synth.SEED_CONFIG1 + 4, 256, 192, 12)
    for border in (capi.BORDER_REFLECT, capi.BORDER_REPLICATE, capi.BORDER_WRAP, capi.BORDER_REFLECT_101, capi.BORDER_BLACK):
        run_both(gpu, oracle, clip, smoothing_radius=5, border_size=16, border_type=border)


@pytest.mark.parametrize("alpha,duration", [(0.1, 30), (0.35, 4), (0.9, 0)])
def test_pipeline_border_fade(gpu, oracle, alpha, duration):
    """borderType "fade" (Stabilizer.cpp:914-978, 1069-1106): the padded frame is blended with a history that every output
    updates; the fade-in of the history weight over fadeDuration frames, and the steady state behind it."""
    clip = synth.make_clip(synth.SEED_CONFIG1 + 5, 256, 192, 16)
    n_out, _ = run_both(gpu, oracle, clip, smoothing_radius=5, border_size=12, border_type=capi.BORDER_FADE, fade_alpha=alpha,
                        fade_duration=duration)
    assert n_out == 16


def test_fade_history_survives_clean_and_batch_mode_is_declined(gpu, oracle):
    """borderHistory_ is not reset by Stabilizer::clean() (Stabilizer.cpp:221-256): a second pass over the clip starts from
    the history of the first; and set_batch() leaves a fade stream with the per-frame pipeline (each output depends on the
    history the output before it updated)."""
    clip = synth.make_clip(synth.SEED_CONFIG1 + 6, 192, 144, 10)
    kw = dict(smoothing_radius=5, border_size=8, border_type=capi.BORDER_FADE, fade_alpha=0.5, fade_duration=3)
    sg, so = gpu.stabilizer(gpu.params(**kw)), oracle.stabilizer(oracle.params(**kw))
    sg.set_batch(8)
    for rep in range(2):
        for f in clip:
            a, b = sg.push(f), so.push(f)
            assert (a is None) == (b is None)
            if a is not None:
                assert np.array_equal(a, b), rep
        sg.clean(); so.clean()
    sg.close(); so.close()


def _canvas_run(gpu, oracle, clip, passes=1, batch=0, **kw):
    """Both stabilizers over the clip with the virtual canvas on: the canvas state (size, scale, regions, fills, window)
    must agree output by output; returns the outputs and the number of fills."""
    kw = dict(dict(smoothing_radius=5, enable_virtual_canvas=1), **kw)
    sg, so = gpu.stabilizer(gpu.params(**kw)), oracle.stabilizer(oracle.params(**kw))
    if batch:
        sg.set_batch(batch)
    outs, fills = [], 0
    for rep in range(passes):
        for k, f in enumerate(clip):
            a, b = sg.push(f), so.push(f)
            assert (a is None) == (b is None), k
            if a is None:
                continue
            assert a.shape == b.shape == f.shape
            ig, io = sg.canvas_info(), so.canvas_info()
            assert ig.tolist() == io.tolist(), (rep, k, ig, io)
            fills += int(io[4])
            outs.append((a, b))
        while True:
            a, b = sg.flush(clip[0]), so.flush(clip[0])
            assert (a is None) == (b is None)
            if a is None:
                break
            outs.append((a, b))
        sg.clean(); so.clean()
    sg.close(); so.close()
    return outs, fills


def test_virtual_canvas_default_scale_is_a_shifted_copy(gpu, oracle):
    """enableVirtualCanvas with the defaults (Stabilizer.cpp:1130-1134, 2066-2149): scale >= 1.5, the one empty region is
    the whole canvas, no frame covers half of it, nothing is filled: the output is the unwarped frame moved by the
    integer part of the correction, black where the canvas shows.  Bit-exact; a border pad does not change its size."""
    clip = synth.make_clip(synth.SEED_CONFIG1 + 11, 256, 192, 14)
    outs, fills = _canvas_run(gpu, oracle, clip, border_size=8, border_type=capi.BORDER_REFLECT)
    assert fills == 0 and len(outs) == 14
    for a, b in outs:
        assert np.array_equal(a, b)
    assert any((a == 0).all(axis=2).any() for a, _ in outs[1:-1])      # some canvas shows


@pytest.mark.parametrize("scale", [1.2, 1.0, 0.8])
def test_virtual_canvas_temporal_fill(gpu, oracle, scale):
    """Scales at which the reference fills: 1.2 (the ring region = whole canvas, filled from the previous frame stretched
    to the canvas), 1.0 (no ring: the regions are the dark parts of the picture, from the gray <= 1 mask and its external
    contours), 0.8 (the window does not fit: the frame comes back as it is).  Two passes: the temporal buffer and the
    canvas outlive clean()."""
    clip = synth.make_clip(synth.SEED_CONFIG1 + 12, 224, 160, 12)
    clip = [f.copy() for f in clip]
    for k, f in enumerate(clip):
        f[40:70, 60 + k:110 + k] = 0                # a dark object that moves
        f[100:160, 0:25] = 0                        # one that touches the picture's border
        f[10:16, 10:20] = 1                         # small: area <= 100, ignored
    outs, fills = _canvas_run(gpu, oracle, clip, passes=2, adaptive_canvas_size=0, canvas_scale_factor=scale, temporal_buffer_size=4,
                              edge_blend_radius=6, batch=8)
    assert fills > 0           # (at 0.8 too: the reference blends into a canvas it then does not use)
    for a, b in outs:
        assert np.array_equal(a, b)


@pytest.mark.parametrize("extra", [dict(temporal_buffer_size=0), dict(temporal_buffer_size=1), dict(temporal_buffer_size=2),
                                   dict(border_size=10, border_type=capi.BORDER_FADE, fade_alpha=0.4, fade_duration=3),
                                   dict(canvas_blend_weight=0.0), dict(edge_blend_radius=0), dict(canvas_scale_factor=1.3, edge_blend_radius=400),
                                   dict(crop_n_zoom=1, border_size=12)])
def test_virtual_canvas_corner_settings(gpu, oracle, extra):
    """Temporal buffers too short to fill from (0, 1) and just long enough (2); a fade border under the canvas (the canvas
    replaces the warped frame, so the fade cannot be seen); a zero blend weight (nothing is ever the best fill); edge radii
    of zero and larger than any region; crop-and-zoom, behind whose returns the reference never reaches the canvas."""
    clip = synth.make_clip(synth.SEED_CONFIG1 + 14, 192, 144, 10)
    kw = dict(dict(adaptive_canvas_size=0, canvas_scale_factor=1.2, temporal_buffer_size=3), **extra)
    outs, _ = _canvas_run(gpu, oracle, clip, **kw)
    assert len(outs) == 10
    for a, b in outs:
        assert np.array_equal(a, b)


def test_virtual_canvas_adaptive_scale_follows_the_motion(gpu, oracle):
    """calculateOptimalCanvasSize (:2281-2314): the scale is chosen once, from the largest of the last 30 transforms."""
    clip = synth.make_clip(synth.SEED_CONFIG1 + 13, 256, 192, 10, pan_q8=16 * 256)      # 60 analysis pixels per frame
    kw = dict(smoothing_radius=5, enable_virtual_canvas=1, canvas_scale_factor=1.1, min_canvas_scale=1.05, max_canvas_scale=3.0)
    so = oracle.stabilizer(oracle.params(**kw))
    for f in clip:
        so.push(f)
    scale = so.canvas_info()[2:3].view(np.float32)[0]
    so.close()
    assert 1.15 < scale < 3.0, scale          # the clip does move the scale off its base value
    outs, _ = _canvas_run(gpu, oracle, clip, **{k: v for k, v in kw.items() if k not in ("smoothing_radius", "enable_virtual_canvas")})
    for a, b in outs:
        assert np.array_equal(a, b)


def test_pipeline_crop_n_zoom(gpu, oracle):
    clip = synth.make_clip(synth.SEED_CONFIG1 + 5, 256, 192, 12)
    run_both(gpu, oracle, clip, smoothing_radius=5, border_size=12, crop_n_zoom=1)


def test_pipeline_drone_mode(gpu, oracle):
    clip = synth.make_clip(synth.SEED_CONFIG1 + 6, 320, 240, 30, pan_q8=60, jitter_q8=200, rot_1e5=50)
    run_both(gpu, oracle, clip, smoothing_radius=6, drone_high_freq_mode=1, horizon_lock=1)


def test_pipeline_adaptive_smoothing(gpu, oracle):
    clip = synth.make_clip(synth.SEED_CONFIG1 + 7, 256, 192, 40)
    run_both(gpu, oracle, clip, smoothing_radius=10, adaptive_smoothing=1, min_smoothing_radius=5, max_smoothing_radius=12)


def test_pipeline_nv12(gpu, oracle):
    clip = [synth.bgr_to_nv12(f) for f in synth.make_clip(synth.SEED_CONFIG3, 320, 240, 14)]
    n_out, _ = run_both(gpu, oracle, clip, fmt=capi.FMT_NV12, smoothing_radius=5, max_corners=400)
    assert n_out == 14


def test_pipeline_flat_frames_no_features(gpu, oracle):
    """No corners anywhere: zero transforms, frames pass through unchanged (Stabilizer.cpp:675-677)."""
    clip = [np.full((120, 160, 3), 128, np.uint8) for _ in range(9)]
    n_out, _ = run_both(gpu, oracle, clip, smoothing_radius=5)
    assert n_out == 9


def test_clean_restarts_the_stream(gpu, oracle):
    clip = synth.make_clip(synth.SEED_CONFIG1 + 8, 256, 192, 8)
    s = gpu.stabilizer(gpu.params(smoothing_radius=5))
    a = [s.push(f) for f in clip]
    s.clean()
    b = [s.push(f) for f in clip]
    for x, y in zip(a, b):
        assert (x is None) == (y is None)
    # detection cadence is per instance and survives clean() (reference: function-static counter),
    # so outputs may differ after clean(); the latency contract must not.
    assert [x is None for x in b] == [True] * 4 + [False] * 4
    s.close()


def test_device_resident_push_matches_host_push(gpu):
    clip = synth.make_clip(synth.SEED_CONFIG1 + 9, 320, 240, 12)
    p = gpu.params(smoothing_radius=5)
    s1 = gpu.stabilizer(p)
    s2 = gpu.stabilizer(p)
    d_in = capi.DevBuf(gpu, clip[0].nbytes)
    d_out = capi.DevBuf(gpu, clip[0].nbytes)
    for f in clip:
        ref = s1.push(f)
        d_in.upload(f)
        got = s2.push_dev(d_in.ptr, 320, 240, 320 * 3, capi.FMT_BGR8, d_out.ptr, 320 * 3)
        s2.sync()
        assert bool(got) == (ref is not None)
        if got:
            assert np.array_equal(d_out.download(f.shape, np.uint8), ref)
    c = s2.counters()
    assert c.frames_in == 12 and c.frames_out == 8 and c.detections >= 6
    s1.close()
    s2.close()


@pytest.mark.parametrize("batch", [2, 5, 16])
def test_deferred_warp_batches_match_immediate_output(gpu, batch):
    """vs_stab_set_warp_batch: the warps of `batch` consecutive results go out as one launch, each
    into its own buffer; after sync they equal what one launch per push produces, flush included."""
    n = 40
    clip = synth.make_clip(synth.SEED_CONFIG1 + 11, 320, 240, n)
    p = gpu.params(smoothing_radius=7)
    s1, s2 = gpu.stabilizer(p), gpu.stabilizer(p)
    s2.set_warp_batch(batch)
    fb = clip[0].nbytes
    d_in = capi.DevBuf(gpu, fb * n)
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
    d_ref, d_got = capi.DevBuf(gpu, fb * (n + 8)), capi.DevBuf(gpu, fb * (n + 8))
    k1 = k2 = 0
    for i in range(n):
        k1 += s1.push_dev(d_in.ptr + i * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_ref.ptr + k1 * fb, 320 * 3)
        k2 += s2.push_dev(d_in.ptr + i * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_got.ptr + k2 * fb, 320 * 3)
    while s1.flush_dev(d_ref.ptr + k1 * fb, 320 * 3):
        k1 += 1
    while s2.flush_dev(d_got.ptr + k2 * fb, 320 * 3):
        k2 += 1
    s1.sync(); s2.sync()
    assert k1 == k2 == n
    ref = d_ref.download((k1, 240, 320, 3), np.uint8)
    got = d_got.download((k2, 240, 320, 3), np.uint8)
    assert np.array_equal(ref, got)
    assert s2.counters().frames_out == n
    # a host push in between drains what is pending and still delivers synchronously
    s3 = gpu.stabilizer(p)
    s3.set_warp_batch(batch)
    outs = []
    for i in range(12):
        if i % 3 == 2:
            r = s3.push(clip[i])
            if r is not None:
                outs.append(r)
        else:
            if s3.push_dev(d_in.ptr + i * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_got.ptr + i * fb, 320 * 3):
                s3.sync()
                outs.append(d_got.download((240, 320, 3), np.uint8, offset=i * fb))
    assert len(outs) == 6 and all(np.array_equal(outs[j], ref[j]) for j in range(6))
    for s in (s1, s2, s3):
        s.close()


@pytest.mark.parametrize("batch,radius,n", [(2, 7, 30), (8, 7, 45), (16, 12, 61), (5, 30, 80), (32, 9, 100), (27, 5, 70), (64, 9, 200), (50, 20, 170)])
def test_batch_mode_matches_per_frame_pipeline(gpu, batch, radius, n):
    """vs_stab_set_batch: GFTT / LK / RANSAC scoring of `batch` frames per launch; outputs (flush included),
    the last frame's debug record and the counters equal the per-frame pipeline's."""
    clip = synth.make_clip(synth.SEED_CONFIG1 + 21, 320, 240, 24)
    order = [i % 24 if (i // 24) % 2 == 0 else 23 - i % 24 for i in range(n)]
    p = gpu.params(smoothing_radius=radius, lk_win_size=21)
    s1, s2 = gpu.stabilizer(p), gpu.stabilizer(p)
    s2.set_batch(batch)
    fb = clip[0].nbytes
    d_in = capi.DevBuf(gpu, fb * 24)
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
    d_ref, d_got = capi.DevBuf(gpu, fb * (n + 4)), capi.DevBuf(gpu, fb * (n + 4))
    k1 = k2 = 0
    for i in range(n):
        k1 += s1.push_dev(d_in.ptr + order[i] * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_ref.ptr + k1 * fb, 320 * 3)
        k2 += s2.push_dev(d_in.ptr + order[i] * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_got.ptr + k2 * fb, 320 * 3)
        assert k1 == k2
    s1.sync(); s2.sync()
    da, db = s1.debug(), s2.debug()
    for name, _ in da._fields_:
        va, vb = getattr(da, name), getattr(db, name)
        assert (list(va) == list(vb)) if hasattr(va, "__len__") else (va == vb), name
    while s1.flush_dev(d_ref.ptr + k1 * fb, 320 * 3):
        k1 += 1
    while s2.flush_dev(d_got.ptr + k2 * fb, 320 * 3):
        k2 += 1
    s1.sync(); s2.sync()
    assert k1 == k2 == n
    ref = d_ref.download((k1, 240, 320, 3), np.uint8)
    got = d_got.download((k2, 240, 320, 3), np.uint8)
    assert np.array_equal(ref, got)
    c1, c2 = s1.counters(), s2.counters()
    assert (c1.frames_in, c1.frames_out, c1.detections) == (c2.frames_in, c2.frames_out, c2.detections)
    # the host entry point still delivers synchronously in batch mode
    s3 = gpu.stabilizer(p)
    s3.set_batch(batch)
    outs = [o for o in (s3.push(clip[order[i]]) for i in range(radius + 3)) if o is not None]
    assert len(outs) == 4 and all(np.array_equal(outs[j], ref[j]) for j in range(4))
    with pytest.raises(capi.VsError):
        s3.set_batch(2)                      # only before the first frame
    for s in (s1, s2, s3):
        s.close()


@pytest.mark.parametrize("size,levels,win", [((200, 150), 4, 9), ((322, 242), 3, 15), ((134, 98), 5, 5)])
def test_batch_mode_pyramid_levels_of_odd_sizes(gpu, size, levels, win):
    """Batch mode builds the pyramid levels of all frames of a batch per launch: one launch per level that produces the
    derivatives of level l and the image of level l+1 from one staged tile (k_pyr.hip pyr_level_kernel; the per-frame pipeline
    runs the two stencils apart).  Drone mode analyses at the frame's own size,
    so the levels here have odd widths and heights (200x150 -> 100x75 -> 50x38 -> 25x19 -> ...), widths that are not
    multiples of four and tiles that hang over every border.  Same tracks, same frames as the per-frame pipeline."""
    w, h = size
    clip = synth.make_clip(synth.SEED_CONFIG1 + 31, w, h, 12)
    p = gpu.params(smoothing_radius=5, drone_high_freq_mode=1, hf_analysis_max_width=1024, lk_max_level=levels, lk_win_size=win, max_corners=120)
    s1, s2 = gpu.stabilizer(p), gpu.stabilizer(p)
    s2.set_batch(4)
    fb = clip[0].nbytes
    n = 22
    d_in = capi.DevBuf(gpu, fb * 12)
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
    d_ref, d_got = capi.DevBuf(gpu, fb * (n + 4)), capi.DevBuf(gpu, fb * (n + 4))
    k1 = k2 = 0
    for i in range(n):
        j = i % 12 if (i // 12) % 2 == 0 else 11 - i % 12
        k1 += s1.push_dev(d_in.ptr + j * fb, w, h, w * 3, capi.FMT_BGR8, d_ref.ptr + k1 * fb, w * 3)
        k2 += s2.push_dev(d_in.ptr + j * fb, w, h, w * 3, capi.FMT_BGR8, d_got.ptr + k2 * fb, w * 3)
        if i % 4 == 3:                       # the batch has just run: its last frame's tracks against the per-frame pipeline's
            s1.sync(); s2.sync()
            a, b = s1.debug_arrays(), s2.debug_arrays()
            for key in ("prev", "curr", "status", "inliers", "gray"):
                assert np.array_equal(a[key].view(np.uint8), b[key].view(np.uint8)), (i, key)
    while s1.flush_dev(d_ref.ptr + k1 * fb, w * 3):
        k1 += 1
    while s2.flush_dev(d_got.ptr + k2 * fb, w * 3):
        k2 += 1
    s1.sync(); s2.sync()
    assert k1 == k2 == n
    assert np.array_equal(d_ref.download((k1, h, w, 3), np.uint8), d_got.download((k2, h, w, 3), np.uint8))
    s1.close(); s2.close()


@pytest.mark.parametrize("extra", [dict(smoothing_method=capi.SMOOTH_GAUSSIAN, gaussian_sigma=3.0),
                                   dict(smoothing_method=capi.SMOOTH_KALMAN),
                                   dict(drone_high_freq_mode=1, horizon_lock=1),
                                   dict(max_corners=60, lk_win_size=15, lk_max_level=3)])
def test_batch_mode_with_other_smoothers_and_drone_state(gpu, extra):
    """The ordered tail of a batch keeps the whole trajectory state (Kalman, drone filters) in LDS: same
    frames as the per-frame pipeline for every smoothing method, with intent segments in the clip."""
    segs = [(0, 512, 0, 384, 200), (14, 2200, 0, 60, 20), (30, 200, 0, 40, 900), (44, 1000, 900, 700, 100)]
    n = 56
    clip = synth.make_clip(synth.SEED_CONFIG1 + 41, 320, 240, n, segments=segs)
    p = gpu.params(smoothing_radius=6, **extra)
    s1, s2 = gpu.stabilizer(p), gpu.stabilizer(p)
    s2.set_batch(7)
    s2.set_zero_copy(True)
    fb = clip[0].nbytes
    d_in = capi.DevBuf(gpu, fb * n)
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
    d_ref, d_got = capi.DevBuf(gpu, fb * n), capi.DevBuf(gpu, fb * n)
    k1 = k2 = 0
    for i in range(n):
        k1 += s1.push_dev(d_in.ptr + i * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_ref.ptr + k1 * fb, 320 * 3)
        k2 += s2.push_dev(d_in.ptr + i * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_got.ptr + k2 * fb, 320 * 3)
    while s1.flush_dev(d_ref.ptr + k1 * fb, 320 * 3):
        k1 += 1
    while s2.flush_dev(d_got.ptr + k2 * fb, 320 * 3):
        k2 += 1
    s1.sync(); s2.sync()
    assert k1 == k2 == n
    assert np.array_equal(d_ref.download((n, 240, 320, 3), np.uint8), d_got.download((n, 240, 320, 3), np.uint8))
    s1.close(); s2.close()


def test_pipeline_full_hd_config2_against_oracle(gpu, oracle):
    """BASELINE configs[1] at its full size: 1920x1080 BGR8, 200 corners, LK 21x21 / 3 levels, frame by frame
    against the oracle (every intermediate, as in run_both), then the same clip through the batch pipeline."""
    clip = synth.make_clip(synth.SEED_CONFIG2, 1920, 1080, 14)
    kw = dict(smoothing_radius=5, max_corners=200, lk_win_size=21, lk_max_level=2)
    oracle.lib.vso_set_threads(8)
    try:
        n_out, worst = run_both(gpu, oracle, clip, **kw)
    finally:
        oracle.lib.vso_set_threads(1)
    assert n_out == 14
    p = gpu.params(**kw)
    s1, s2 = gpu.stabilizer(p), gpu.stabilizer(p)
    s2.set_batch(16)
    s2.set_zero_copy(True)
    fb = clip[0].nbytes
    d_in = capi.DevBuf(gpu, fb * 14)
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
    d_got = capi.DevBuf(gpu, fb * 14)
    k = 0
    ref = []
    for i, f in enumerate(clip):
        r = s1.push(f)
        if r is not None:
            ref.append(r)
        k += s2.push_dev(d_in.ptr + i * fb, 1920, 1080, 1920 * 3, capi.FMT_BGR8, d_got.ptr + k * fb, 1920 * 3)
    s2.sync()
    assert k == len(ref) == 10
    got = d_got.download((k, 1080, 1920, 3), np.uint8)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    s1.close(); s2.close()


def test_pipeline_4k_nv12_config3_against_oracle(gpu, oracle):
    """BASELINE configs[2] at its full size: 3840x2160 NV12, 400 corners (per-frame pipeline), followed by
    roll correction + auto zoom/crop on a 4K BGR frame."""
    import roll_scene
    clip = [synth.bgr_to_nv12(f) for f in synth.make_clip(synth.SEED_CONFIG3, 3840, 2160, 8)]
    oracle.lib.vso_set_threads(8)
    try:
        # (every frame array_equal: at 3840 columns a last-ulp difference of cosf / sinf would move the 1/1024-px coordinate of a
        # few columns across a rounding boundary - which is why both sides evaluate one definition of them, vs_libm.h)
        n_out, worst = run_both(gpu, oracle, clip, fmt=capi.FMT_NV12, smoothing_radius=5, max_corners=400)
        assert n_out == 8
        f = roll_scene.horizon_frame(3840, 2160, 45, seed=7)
        ro, rg = oracle.roll_correction(), gpu.roll_correction()
        for _ in range(2):
            a, b = ro.correct(f), rg.correct(f)
            assert np.array_equal(a, b) and ro.state() == rg.state()
        z, info = oracle.auto_zoom_crop(a)
        az = gpu.auto_zoom_crop()
        assert np.array_equal(az.apply(b), z) and az.info().tolist() == info.tolist()
    finally:
        oracle.lib.vso_set_threads(1)


@pytest.mark.parametrize("size,batch", [((320, 240), 8), ((640, 360), 16), ((320, 240), 64)])
def test_batch_mode_nv12(gpu, size, batch):
    """NV12 surfaces in batch mode (Y plane analysed and warped, interleaved chroma warped with the halved
    translation): same surfaces as the per-frame pipeline, flush included."""
    w, h = size
    n = max(44, 2 * batch + 12)           # two full batches (of 64: four warp launches of 32) and a drain
    clip = [synth.bgr_to_nv12(f) for f in synth.make_clip(synth.SEED_CONFIG3 + 5, w, h, n)]
    p = gpu.params(smoothing_radius=6, max_corners=400)
    s1, s2 = gpu.stabilizer(p), gpu.stabilizer(p)
    s2.set_batch(batch)
    s2.set_zero_copy(True)
    fb = clip[0].nbytes
    d_in = capi.DevBuf(gpu, fb * n)
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
    d_ref, d_got = capi.DevBuf(gpu, fb * n), capi.DevBuf(gpu, fb * n)
    k1 = k2 = 0
    for i in range(n):
        k1 += s1.push_dev(d_in.ptr + i * fb, w, h, w, capi.FMT_NV12, d_ref.ptr + k1 * fb, w)
        k2 += s2.push_dev(d_in.ptr + i * fb, w, h, w, capi.FMT_NV12, d_got.ptr + k2 * fb, w)
    while s1.flush_dev(d_ref.ptr + k1 * fb, w):
        k1 += 1
    while s2.flush_dev(d_got.ptr + k2 * fb, w):
        k2 += 1
    s1.sync(); s2.sync()
    assert k1 == k2 == n
    assert np.array_equal(d_ref.download((n, h * 3 // 2, w), np.uint8), d_got.download((n, h * 3 // 2, w), np.uint8))
    s1.close(); s2.close()


def test_batch_mode_nv12_at_4k_with_the_bench_batch(gpu):
    """BASELINE configs[2] as bench.py runs it: 3840x2160 NV12, 400 corners, batches of 64 zero-copy surfaces (Y and UV plane
    through warp_plane_kernel, 32 surfaces per launch).  Same surfaces as the per-frame pipeline over 140 pushes (two full
    batches and a drain), flush included."""
    w, h, n = 3840, 2160, 140
    base = [synth.bgr_to_nv12(f) for f in synth.make_clip(synth.SEED_CONFIG3 + 11, w, h, 8)]
    order = [(i % 14) if (i % 14) < 8 else 14 - (i % 14) for i in range(n)]       # ping-pong: neighbours differ by one step
    p = gpu.params(smoothing_radius=6, max_corners=400)
    s1, s2 = gpu.stabilizer(p), gpu.stabilizer(p)
    s2.set_batch(64)
    s2.set_zero_copy(True)
    fb = base[0].nbytes
    d_in = capi.DevBuf(gpu, fb * len(base))
    for i, f in enumerate(base):
        d_in.upload(f, i * fb)
    d_ref, d_got = capi.DevBuf(gpu, fb * n), capi.DevBuf(gpu, fb * n)
    k1 = k2 = 0
    for i in range(n):
        src = d_in.ptr + order[i] * fb
        k1 += s1.push_dev(src, w, h, w, capi.FMT_NV12, d_ref.ptr + k1 * fb, w)
        k2 += s2.push_dev(src, w, h, w, capi.FMT_NV12, d_got.ptr + k2 * fb, w)
    while s1.flush_dev(d_ref.ptr + k1 * fb, w):
        k1 += 1
    while s2.flush_dev(d_got.ptr + k2 * fb, w):
        k2 += 1
    s1.sync(); s2.sync()
    assert k1 == k2 == n
    for i in range(0, n, 10):                       # ten surfaces at a time: 124 MB per side on the host
        m = min(10, n - i)
        a = d_ref.download((m, h * 3 // 2, w), np.uint8, offset=i * fb)
        b = d_got.download((m, h * 3 // 2, w), np.uint8, offset=i * fb)
        assert np.array_equal(a, b), i
    s1.close(); s2.close()
    for d in (d_in, d_ref, d_got):
        d.free()


@pytest.mark.parametrize("mode", ["per_frame_copy", "per_frame_zero_copy", "batch_zero_copy"])
def test_nv12_decoder_surfaces(gpu, mode):
    """Decoder hand-off: NV12 surfaces with a padded pitch and the UV plane at
    pitch * aligned_height behind the Y pointer (rocDecode / VA-API layout), read in place and written
    into surfaces of another such layout: same planes as with packed frames, flush included."""
    w, h, n = 322, 198, 40
    clip = [synth.bgr_to_nv12(f) for f in synth.make_clip(synth.SEED_CONFIG3 + 9, w, h, n)]
    in_pitch, in_rows = 512, 208                      # 256-byte pitch, height aligned to 16
    out_pitch, out_rows = 384, 224
    in_uv, out_uv = in_pitch * in_rows, out_pitch * out_rows
    in_bytes, out_bytes = in_pitch * (in_rows + in_rows // 2), out_pitch * (out_rows + out_rows // 2)
    p = gpu.params(smoothing_radius=6, max_corners=300)
    s1, s2 = gpu.stabilizer(p), gpu.stabilizer(p)
    if mode == "batch_zero_copy":
        s2.set_batch(8)
    if mode != "per_frame_copy":
        s2.set_zero_copy(True)
    s2.set_nv12_layout(in_uv, out_uv)
    fb = clip[0].nbytes
    d_in, d_ref = capi.DevBuf(gpu, fb * n), capi.DevBuf(gpu, fb * n)
    d_surf, d_got = capi.DevBuf(gpu, in_bytes * n), capi.DevBuf(gpu, out_bytes * n)
    rng = np.random.default_rng(3)
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
        surf = rng.integers(0, 256, (in_rows + in_rows // 2, in_pitch), dtype=np.uint8)   # padding holds garbage
        surf[:h, :w] = f[:h]
        surf[in_rows:in_rows + h // 2, :w] = f[h:]
        d_surf.upload(surf, i * in_bytes)
    d_got.zero()
    k1 = k2 = 0
    for i in range(n):
        k1 += s1.push_dev(d_in.ptr + i * fb, w, h, w, capi.FMT_NV12, d_ref.ptr + k1 * fb, w)
        k2 += s2.push_dev(d_surf.ptr + i * in_bytes, w, h, in_pitch, capi.FMT_NV12, d_got.ptr + k2 * out_bytes, out_pitch)
    while s1.flush_dev(d_ref.ptr + k1 * fb, w):
        k1 += 1
    while s2.flush_dev(d_got.ptr + k2 * out_bytes, out_pitch):
        k2 += 1
    s1.sync(); s2.sync()
    assert k1 == k2 == n
    ref = d_ref.download((n, h * 3 // 2, w), np.uint8)
    got = d_got.download((n, out_rows + out_rows // 2, out_pitch), np.uint8)
    assert np.array_equal(got[:, :h, :w], ref[:, :h])
    assert np.array_equal(got[:, out_rows:out_rows + h // 2, :w], ref[:, h:])
    assert not got[:, :h, w:].any() and not got[:, h:out_rows].any() and not got[:, out_rows + h // 2:].any()   # padding untouched
    s1.close(); s2.close()


def test_batch_mode_output_pitch_may_change(gpu):
    """A different output pitch closes the batch being collected (one pitch per batched warp launch)."""
    n = 30
    clip = synth.make_clip(synth.SEED_CONFIG1 + 51, 320, 240, n)
    p = gpu.params(smoothing_radius=5)
    s1, s2 = gpu.stabilizer(p), gpu.stabilizer(p)
    s2.set_batch(8)
    fb = clip[0].nbytes
    pitch2 = 336 * 3
    d_in = capi.DevBuf(gpu, fb * n)
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
    d_ref, d_got = capi.DevBuf(gpu, fb * n), capi.DevBuf(gpu, pitch2 * 240 * n)
    d_got.zero()
    k1 = k2 = 0
    pitches = []
    for i in range(n):
        k1 += s1.push_dev(d_in.ptr + i * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_ref.ptr + k1 * fb, 320 * 3)
        pitch = pitch2 if (i // 5) % 2 else 320 * 3          # changes every 5 pushes, i.e. inside batches
        got = s2.push_dev(d_in.ptr + i * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_got.ptr + k2 * pitch2 * 240, pitch)
        if got:
            pitches.append(pitch)
        k2 += got
    s1.sync(); s2.sync()
    assert k1 == k2 == n - 4
    ref = d_ref.download((k1, 240, 320, 3), np.uint8)
    raw = d_got.download((n, pitch2 * 240), np.uint8)
    for j, pitch in enumerate(pitches):
        got = raw[j, :pitch * 240].reshape(240, pitch)[:, :960].reshape(240, 320, 3)
        assert np.array_equal(got, ref[j]), j
    s1.close(); s2.close()


@pytest.mark.parametrize("batch", [1, 8])
def test_zero_copy_input_matches_queued_copy(gpu, batch):
    """vs_stab_set_zero_copy: frames are read where the caller holds them (here: a resident clip that stays
    untouched); same results as with the copy into the instance's queue, per-frame and batch mode alike."""
    n = 40
    clip = synth.make_clip(synth.SEED_CONFIG1 + 31, 320, 240, n)
    p = gpu.params(smoothing_radius=6)
    s1, s2 = gpu.stabilizer(p), gpu.stabilizer(p)
    s2.set_batch(batch)
    s2.set_zero_copy(True)
    fb = clip[0].nbytes
    d_in = capi.DevBuf(gpu, fb * n)
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
    d_ref, d_got = capi.DevBuf(gpu, fb * n), capi.DevBuf(gpu, fb * n)
    k1 = k2 = 0
    for i in range(n):
        k1 += s1.push_dev(d_in.ptr + i * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_ref.ptr + k1 * fb, 320 * 3)
        k2 += s2.push_dev(d_in.ptr + i * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_got.ptr + k2 * fb, 320 * 3)
    while s1.flush_dev(d_ref.ptr + k1 * fb, 320 * 3):
        k1 += 1
    while s2.flush_dev(d_got.ptr + k2 * fb, 320 * 3):
        k2 += 1
    s1.sync(); s2.sync()
    assert k1 == k2 == n
    assert np.array_equal(d_ref.download((n, 240, 320, 3), np.uint8), d_got.download((n, 240, 320, 3), np.uint8))
    s3 = gpu.stabilizer(p)                 # padded rows are read in place too, but all queued frames share one pitch
    s3.set_zero_copy(True)
    s3.push_dev(d_in.ptr, 316, 240, 320 * 3, capi.FMT_BGR8, d_got.ptr, 316 * 3)
    with pytest.raises(capi.VsError):
        s3.push_dev(d_in.ptr + fb, 316, 240, 316 * 3, capi.FMT_BGR8, d_got.ptr, 316 * 3)
    s3.sync(); s3.close()
    s1.close(); s2.close()


def test_nv12_layout_and_zero_copy_switch_need_an_empty_queue(gpu):
    w, h = 64, 48
    f = synth.bgr_to_nv12(synth.make_clip(synth.SEED_CONFIG3, w, h, 1)[0])
    d_in, d_out = capi.DevBuf.from_array(gpu, f), capi.DevBuf(gpu, f.nbytes)
    s = gpu.stabilizer(gpu.params(smoothing_radius=5))
    s.set_nv12_layout(0, 0)
    s.push_dev(d_in.ptr, w, h, w, capi.FMT_NV12, d_out.ptr, w)
    with pytest.raises(capi.VsError):
        s.set_nv12_layout(w * 64, 0)          # a frame is queued
    with pytest.raises(capi.VsError):
        s.set_zero_copy(True)
    while s.flush_dev(d_out.ptr, w):
        pass
    s.set_nv12_layout(w * 64, 0)              # queue drained: allowed again
    s.sync(); s.close()


def test_errors_are_loud(gpu):
    with pytest.raises(capi.VsError):
        gpu.stabilizer(gpu.params(enable_virtual_canvas=1, canvas_blend_weight=2.0))
    s = gpu.stabilizer(gpu.params(enable_virtual_canvas=1, smoothing_radius=5))
    with pytest.raises(capi.VsError):
        s.push(np.zeros((120, 160), np.uint8), capi.FMT_GRAY8)      # the reference's canvas needs three channels
    s.close()
    with pytest.raises(capi.VsError):
        gpu.stabilizer(gpu.params(max_corners=0))
    s = gpu.stabilizer(gpu.params(smoothing_radius=5))
    s.push(np.zeros((120, 160, 3), np.uint8))
    with pytest.raises(capi.VsError):
        s.push(np.zeros((100, 160, 3), np.uint8))     # geometry change without clean()
    s.close()


def _unsynced_outputs(gpu, clip, order, batch, **params):
    """All frames pushed without a host synchronisation in between (device entry point); the stabilized frames."""
    w, h = clip[0].shape[1], clip[0].shape[0]
    fb = clip[0].nbytes
    s = gpu.stabilizer(gpu.params(**params))
    if batch > 1:
        s.set_batch(batch)
    d_in = capi.DevBuf(gpu, fb * len(clip))
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
    d_out = capi.DevBuf(gpu, fb * (len(order) + 2))
    k = 0
    for i in order:
        k += s.push_dev(d_in.ptr + i * fb, w, h, w * 3, capi.FMT_BGR8, d_out.ptr + k * fb, w * 3)
    while s.flush_dev(d_out.ptr + k * fb, w * 3):
        k += 1
    s.sync()
    out = d_out.download((k, h, w, 3), np.uint8)
    s.close()
    return out


@pytest.mark.parametrize("batch", [1, 8])
def test_keypoint_buffers_are_not_recycled_under_the_ransac_kernels(gpu, oracle, monkeypatch, batch):
    """Regression test for a write-after-read hazard: the event that lets a re-detection overwrite a keypoint buffer used to
    be recorded behind the tracker, but the RANSAC scoring / selection kernels read the same buffer after it.  With a
    spin kernel between tracker and RANSAC (VS_STAB_DEBUG_DELAY_US) and no host synchronisation between pushes the window
    is wide open: the result must still be the oracle's."""
    clip = synth.make_clip(synth.SEED_CONFIG1 + 31, 320, 240, 20)
    order = list(range(20))
    ref = []
    so = oracle.stabilizer(oracle.params(smoothing_radius=5))
    for i in order:
        r = so.push(clip[i])
        if r is not None:
            ref.append(r)
    while True:
        r = so.flush(clip[0])
        if r is None:
            break
        ref.append(r)
    so.close()
    monkeypatch.setenv("VS_LAB", "1")
    monkeypatch.setenv("VS_STAB_DEBUG_DELAY_US", "400")
    got = _unsynced_outputs(gpu, clip, order, batch, smoothing_radius=5)
    assert len(got) == len(ref) == 20
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("extra", [dict(), dict(smoothing_method=capi.SMOOTH_GAUSSIAN, gaussian_sigma=2.0),
                                   dict(smoothing_method=capi.SMOOTH_KALMAN), dict(drone_high_freq_mode=1)])
@pytest.mark.parametrize("batch", [1, 32, 64])
def test_long_stream_wraps_the_trajectory_rings(gpu, oracle, extra, batch):
    """340 frames: the 256-entry device rings of transforms / path (traj_state.h) wrap, the incremental Kalman walk and the
    Gaussian reflect window run past the wrap; per-frame pipeline and batch mode (flush included) against the oracle, whose
    history grows without bound like the reference's."""
    clip = synth.make_clip(synth.SEED_CONFIG1 + 33, 192, 144, 20)
    order = [i % 20 if (i // 20) % 2 == 0 else 19 - i % 20 for i in range(340)]
    params = dict(smoothing_radius=9, **extra)
    so = oracle.stabilizer(oracle.params(**params))
    ref = []
    for i in order:
        r = so.push(clip[i])
        if r is not None:
            ref.append(r)
    while True:
        r = so.flush(clip[0])
        if r is None:
            break
        ref.append(r)
    so.close()
    got = _unsynced_outputs(gpu, clip, order, batch, **params)
    assert len(got) == len(ref) == 340
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)


def test_host_pipeline_returns_the_same_frames_one_call_later(gpu):
    """vs_stab_set_host_pipeline: a call hands out the frame the call before it computed (its download overlaps the next
    upload); the sequence of frames, flush included, is the synchronous entry point's.  Page-locked and pageable buffers."""
    clip = synth.make_clip(synth.SEED_CONFIG1 + 35, 320, 240, 26)
    p = gpu.params(smoothing_radius=6)
    s1, s2 = gpu.stabilizer(p), gpu.stabilizer(p)
    s2.set_host_pipeline(True)
    pin_in = capi.HostBuf(gpu, clip[0].shape)
    pin_out = capi.HostBuf(gpu, clip[0].shape)
    ref, got = [], []
    for k, f in enumerate(clip):
        r = s1.push(f)
        if r is not None:
            ref.append(r)
        if k % 2:                       # page-locked frames on odd calls, pageable ones on even calls
            pin_in.array[...] = f
            g = s2.push(pin_in.array, out=pin_out.array)
        else:
            g = s2.push(f)
        assert (g is not None) == (len(ref) > 1 or (len(ref) == 1 and r is None)), k
        if g is not None:
            got.append(g.copy())
    assert len(got) == len(ref) - 1
    while True:
        r = s1.flush(clip[0])
        if r is None:
            break
        ref.append(r)
    while True:
        g = s2.flush(clip[0])
        if g is None:
            break
        got.append(g)
    assert len(got) == len(ref) == 26
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    with pytest.raises(capi.VsError):
        s1.push(clip[0]); s1.set_host_pipeline(True)      # only while nothing is queued
    for s in (s1, s2):
        s.close()
    pin_in.free(); pin_out.free()


def test_host_pipeline_helper_threads_of_several_instances(gpu):
    """Pageable output buffers: the download of the previous result is issued by the instance's helper thread, beside the
    caller's upload.  Four pipelined instances fed in turn (four helper threads alive at once), made and destroyed twice:
    every stream gets the frames of its own synchronous run."""
    clips = [synth.make_clip(synth.SEED_CONFIG1 + 50 + g, 640, 360, 18) for g in range(4)]
    p = gpu.params(smoothing_radius=5)
    refs = []
    for clip in clips:
        s = gpu.stabilizer(p)
        refs.append([r for r in (s.push(f) for f in clip) if r is not None])
        s.close()
    for _ in range(2):
        stabs = [gpu.stabilizer(p) for _ in clips]
        for s in stabs:
            s.set_host_pipeline(True)
        outs = [np.empty_like(clips[0][0]) for _ in clips]          # pageable, reused: a late download would show
        got = [[] for _ in clips]
        for k in range(18):
            for g, s in enumerate(stabs):
                r = s.push(clips[g][k], out=outs[g])
                if r is not None:
                    got[g].append(r.copy())
        for g, s in enumerate(stabs):
            r = s.flush(clips[g][0])                                 # the frame the last push computed
            assert r is not None
            got[g].append(r)
            s.close()
        for g in range(4):
            assert len(got[g]) == len(refs[g]) == 14
            for a, b in zip(got[g], refs[g]):
                assert np.array_equal(a, b)


@pytest.mark.parametrize("extra", [dict(border_size=16, border_type=capi.BORDER_REFLECT), dict(border_size=9, border_type=capi.BORDER_REPLICATE),
                                   dict(border_size=12, border_type=capi.BORDER_WRAP), dict(border_size=16, border_type=capi.BORDER_REFLECT_101),
                                   dict(border_size=8, border_type=capi.BORDER_BLACK), dict(border_size=20, crop_n_zoom=1),
                                   dict(border_size=16, border_type=capi.BORDER_REFLECT, batch=40), dict(border_size=20, crop_n_zoom=1, batch=64)])
def test_batch_mode_with_border_pad_and_crop_n_zoom(gpu, extra):
    """copyMakeBorder in front of the warp (Stabilizer.cpp:981-990) and crop-and-zoom behind it (:1108-1124) in batch mode:
    the frames of the per-frame pipeline, flush included."""
    clip = synth.make_clip(synth.SEED_CONFIG1 + 37, 320, 240, 24)
    extra = dict(extra)
    batch = extra.pop("batch", 8)          # more than 32: the batch's warps are two launches, each with its share of the tables
    n = 45 if batch == 8 else 2 * batch + 20
    order = [i % 24 if (i // 24) % 2 == 0 else 23 - i % 24 for i in range(n)]
    p = gpu.params(smoothing_radius=7, **extra)
    s1, s2 = gpu.stabilizer(p), gpu.stabilizer(p)
    s2.set_batch(batch)
    oh, ow, _ = s1.out_shape(320, 240, capi.FMT_BGR8)
    fb, ob = clip[0].nbytes, oh * ow * 3
    d_in = capi.DevBuf(gpu, fb * 24)
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
    d_ref, d_got = capi.DevBuf(gpu, ob * (n + 2)), capi.DevBuf(gpu, ob * (n + 2))
    k1 = k2 = 0
    for i in range(n):
        k1 += s1.push_dev(d_in.ptr + order[i] * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_ref.ptr + k1 * ob, ow * 3)
        k2 += s2.push_dev(d_in.ptr + order[i] * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_got.ptr + k2 * ob, ow * 3)
        assert k1 == k2
    s1.sync(); s2.sync()
    assert k1 == n - 6
    ref = d_ref.download((k1, oh, ow, 3), np.uint8)
    got = d_got.download((k2, oh, ow, 3), np.uint8)
    assert np.array_equal(ref, got)
    for s in (s1, s2):
        s.close()


@pytest.mark.parametrize("extra", [dict(border_size=32, border_type=capi.BORDER_REFLECT), dict(border_size=32, crop_n_zoom=1)])
def test_batch_mode_border_and_crop_full_hd_against_oracle(gpu, oracle, extra):
    """configs[1] size (1920x1080, 200 corners, 21x21 LK) with a border pad / crop-and-zoom, batch mode against the oracle."""
    clip = synth.make_clip(synth.SEED_CONFIG2 + 3, 1920, 1080, 14)
    params = dict(smoothing_radius=5, max_corners=200, lk_win_size=21, **extra)
    so = oracle.stabilizer(oracle.params(**params))
    ref = [r for r in (so.push(f) for f in clip) if r is not None]
    so.close()
    s = gpu.stabilizer(gpu.params(**params))
    s.set_batch(8)
    oh, ow, _ = s.out_shape(1920, 1080, capi.FMT_BGR8)
    fb, ob = clip[0].nbytes, oh * ow * 3
    d_in, d_out = capi.DevBuf(gpu, fb * len(clip)), capi.DevBuf(gpu, ob * len(clip))
    k = 0
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
        k += s.push_dev(d_in.ptr + i * fb, 1920, 1080, 1920 * 3, capi.FMT_BGR8, d_out.ptr + k * ob, ow * 3)
    s.sync()
    assert k == len(ref) == 10
    got = d_out.download((k, oh, ow, 3), np.uint8)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    s.close()


def test_config0_640x480_300_frames_radius_25(gpu, oracle):
    """BASELINE configs[0]: 300 frames of 640x480, smoothingRadius 25 (the analysis image is an UPSCALE to 960x540): inputs
    0-23 return nothing, input 24 returns frame 0, 276 frames from stabilize() and 24 from flush() - on the
    HIP path in both execution models, every frame against the oracle."""
    clip = synth.make_clip(synth.SEED_CONFIG1, 640, 480, 60)
    order = [i % 60 if (i // 60) % 2 == 0 else 59 - i % 60 for i in range(300)]
    so = oracle.stabilizer(oracle.params(smoothing_radius=25))
    ref, first_out = [], None
    for k, i in enumerate(order):
        r = so.push(clip[i])
        if r is not None:
            if first_out is None:
                first_out = k
            ref.append(r)
    n_push = len(ref)
    while True:
        r = so.flush(clip[0])
        if r is None:
            break
        ref.append(r)
    so.close()
    assert first_out == 24 and n_push == 276 and len(ref) == 300
    for batch in (1, 32):
        got = _unsynced_outputs(gpu, clip, order, batch, smoothing_radius=25)
        assert len(got) == 300
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)


def test_eight_instances_on_one_gpu_keep_their_streams_apart(gpu):
    """BASELINE configs[4], one GPU's share: 8 streams in batch mode on the device's shared HIP streams, pushes interleaved
    frame by frame; every stream's frames equal those of a run on its own."""
    n_streams, n = 8, 70
    clips = [synth.make_clip(synth.SEED_CONFIG1 + 40 + g, 320, 240, 16) for g in range(n_streams)]
    order = [i % 16 if (i // 16) % 2 == 0 else 15 - i % 16 for i in range(n)]
    alone = [_unsynced_outputs(gpu, clips[g], order, 16, smoothing_radius=6 + g % 3) for g in range(n_streams)]
    fb = clips[0][0].nbytes
    stabs, d_in, d_out, k = [], [], [], [0] * n_streams
    for g in range(n_streams):
        s = gpu.stabilizer(gpu.params(smoothing_radius=6 + g % 3))
        s.set_batch(16)
        s.set_zero_copy(True)
        stabs.append(s)
        b = capi.DevBuf(gpu, fb * 16)
        for i, f in enumerate(clips[g]):
            b.upload(f, i * fb)
        d_in.append(b)
        d_out.append(capi.DevBuf(gpu, fb * (n + 2)))
    for i in order:
        for g in range(n_streams):
            k[g] += stabs[g].push_dev(d_in[g].ptr + i * fb, 320, 240, 320 * 3, capi.FMT_BGR8, d_out[g].ptr + k[g] * fb, 320 * 3)
    for g in range(n_streams):
        while stabs[g].flush_dev(d_out[g].ptr + k[g] * fb, 320 * 3):
            k[g] += 1
    for g in range(n_streams):
        stabs[g].sync()
        got = d_out[g].download((k[g], 240, 320, 3), np.uint8)
        assert k[g] == n and np.array_equal(got, alone[g]), g
        stabs[g].close()


@pytest.mark.parametrize("batch", [1, 8])
@pytest.mark.parametrize("extra", [dict(border_size=8, border_type=capi.BORDER_REFLECT), dict(border_size=8, crop_n_zoom=1)])
def test_nv12_with_a_border_is_refused_in_both_execution_models(gpu, batch, extra):
    """Border padding and crop-and-zoom exist for BGR8 frames only (the reference pads / crops cv::Mat frames of three
    channels); an NV12 stream with border_size > 0 is turned down by the first push - per frame and in batch mode alike, so
    the two execution models cannot diverge on it."""
    nv = synth.bgr_to_nv12(synth.make_clip(synth.SEED_CONFIG1 + 40, 64, 48, 1)[0])
    s = gpu.stabilizer(gpu.params(smoothing_radius=5, **extra))
    s.set_batch(batch)
    d_in, d_out = capi.DevBuf.from_array(gpu, nv), capi.DevBuf(gpu, 4 * nv.nbytes)
    with pytest.raises(capi.VsError, match="BGR8"):
        s.push_dev(d_in.ptr, 64, 48, 64, capi.FMT_NV12, d_out.ptr, 64 + 16)
    s.close()


@pytest.mark.parametrize("fmt", [capi.FMT_BGR8, capi.FMT_NV12])
@pytest.mark.parametrize("n_streams,step,extra", [(8, 8, dict()), (3, 16, dict(smoothing_method=capi.SMOOTH_KALMAN)),
                                                   (5, 4, dict(smoothing_method=capi.SMOOTH_GAUSSIAN, drone_high_freq_mode=1)), (2, 64, dict())])
def test_streams_of_a_vs_batch_equal_their_solo_runs(gpu, fmt, n_streams, step, extra):
    """BASELINE configs[4], one GPU's share, through vs_batch_*: the frames of all streams in one launch per stage (one
    workgroup per stream in the ordered tail, one warp launch per 32 due frames).  Every stream's frames, flush included, equal
    those of a vs_stab instance run on its own - which the other tests hold against the oracle.  70 pushes per stream: steps of
    8 x 8, 3 x 16, 5 x 4 and 2 x 64 frames, the last step partial; one stream sits out every seventh call."""
    n = 70
    clips = [synth.make_clip(synth.SEED_CONFIG1 + 60 + g, 320, 240, 16) for g in range(n_streams)]
    if fmt == capi.FMT_NV12:
        clips = [[synth.bgr_to_nv12(f) for f in c] for c in clips]
    order = [i % 16 if (i // 16) % 2 == 0 else 15 - i % 16 for i in range(n)]
    params = dict(smoothing_radius=7, **extra)
    fb = clips[0][0].nbytes
    stride = 320 * 3 if fmt == capi.FMT_BGR8 else 320

    def solo(g):
        s = gpu.stabilizer(gpu.params(**params))
        s.set_batch(16)
        s.set_zero_copy(True)
        d_in, d_out = capi.DevBuf(gpu, fb * 16), capi.DevBuf(gpu, fb * (n + 2))
        for i, f in enumerate(clips[g]):
            d_in.upload(f, i * fb)
        k = 0
        for i in order:
            k += s.push_dev(d_in.ptr + i * fb, 320, 240, stride, fmt, d_out.ptr + k * fb, stride)
        while s.flush_dev(d_out.ptr + k * fb, stride):
            k += 1
        s.sync()
        out = d_out.download((k,) + clips[g][0].shape, np.uint8)
        s.close()
        return out
    alone = [solo(g) for g in range(n_streams)]
    b = gpu.batch(gpu.params(**params), n_streams, step)
    b.set_zero_copy(True)
    d_in, d_out, k, pos = [], [], [0] * n_streams, [0] * n_streams
    for g in range(n_streams):
        buf = capi.DevBuf(gpu, fb * 16)
        for i, f in enumerate(clips[g]):
            buf.upload(f, i * fb)
        d_in.append(buf)
        d_out.append(capi.DevBuf(gpu, fb * (n + 2)))
    call = 0
    while min(pos) < n:
        # stream (call mod n_streams) sits out every seventh call: the members' queues run out of step
        skip = call % n_streams if call % 7 == 6 else -1
        fr = [d_in[g].ptr + order[pos[g]] * fb if (g != skip and pos[g] < n) else None for g in range(n_streams)]
        ou = [d_out[g].ptr + k[g] * fb for g in range(n_streams)]
        prod = b.push_dev(fr, 320, 240, stride, fmt, ou, stride)
        for g in range(n_streams):
            if fr[g] is not None:
                pos[g] += 1
                k[g] += prod[g]
        call += 1
    while True:
        prod = b.flush_dev([d_out[g].ptr + k[g] * fb for g in range(n_streams)], stride)
        for g in range(n_streams):
            k[g] += prod[g]
        if not any(prod):
            break
    b.sync()
    for g in range(n_streams):
        got = d_out[g].download((k[g],) + clips[g][0].shape, np.uint8)
        assert k[g] == n and np.array_equal(got, alone[g]), g
    # the per-stream getters work on the members
    assert b.stream(0).counters().frames_out == n and b.stream(n_streams - 1).debug().out_index == n - 1
    b.close()


def test_vs_batch_refuses_per_stream_modes(gpu):
    """Modes whose outputs depend on each other or on a host decision per output stay with vs_stab_* (per-frame pipeline)."""
    for kw in (dict(adaptive_smoothing=1), dict(border_size=8, border_type=capi.BORDER_FADE), dict(enable_virtual_canvas=1)):
        with pytest.raises(capi.VsError):
            gpu.batch(gpu.params(**kw), 2, 8)


def _group_vs_solo(gpu, per_stream_params, step, n=50, size=(320, 240), solo_batch=16):
    """Every stream of a vs_batch (one parameter block per stream) against a standalone instance with the same block."""
    w, h = size
    S = len(per_stream_params)
    clips = [synth.make_clip(synth.SEED_CONFIG1 + 80 + g, w, h, 16) for g in range(S)]
    order = [i % 16 if (i // 16) % 2 == 0 else 15 - i % 16 for i in range(n)]
    fb = clips[0][0].nbytes
    probe = gpu.stabilizer(per_stream_params[0])
    oh, ow, _ = probe.out_shape(w, h, capi.FMT_BGR8)
    probe.close()
    ob = ow * oh * 3

    def load(g):
        buf = capi.DevBuf(gpu, fb * 16)
        for i, f in enumerate(clips[g]):
            buf.upload(f, i * fb)
        return buf
    alone = []
    for g in range(S):
        s = gpu.stabilizer(per_stream_params[g])
        s.set_batch(solo_batch)
        s.set_zero_copy(True)
        d_in, d_out = load(g), capi.DevBuf(gpu, ob * (n + 2))
        k = 0
        for i in order:
            k += s.push_dev(d_in.ptr + i * fb, w, h, w * 3, capi.FMT_BGR8, d_out.ptr + k * ob, ow * 3)
        # (the last queued frame has no transform and comes back at its own size: not part of this comparison)
        s.sync()
        alone.append(d_out.download((k, oh, ow, 3), np.uint8))
        s.close()
    b = gpu.batch(list(per_stream_params), S, step)
    b.set_zero_copy(True)
    d_in = [load(g) for g in range(S)]
    d_out = [capi.DevBuf(gpu, ob * (n + 2)) for _ in range(S)]
    k = [0] * S
    for i in order:
        prod = b.push_dev([d_in[g].ptr + i * fb for g in range(S)], w, h, w * 3, capi.FMT_BGR8, [d_out[g].ptr + k[g] * ob for g in range(S)], ow * 3)
        for g in range(S):
            k[g] += prod[g]
    b.sync()
    for g in range(S):
        got = d_out[g].download((k[g], oh, ow, 3), np.uint8)
        assert k[g] == len(alone[g]) > 0 and np.array_equal(got, alone[g]), g
    b.close()


@pytest.mark.parametrize("extra", [dict(border_size=16, border_type=capi.BORDER_REFLECT), dict(border_size=9, border_type=capi.BORDER_REPLICATE),
                                   dict(border_size=12, crop_n_zoom=1)])
def test_vs_batch_with_border_pad_and_crop_n_zoom_members(gpu, extra):
    """Border pad and crop-and-zoom are launch shapes, not per-output decisions: a group batches them like a standalone
    instance does (it IS the same schedule: an instance is a group of one)."""
    p = gpu.params(smoothing_radius=7, **extra)
    _group_vs_solo(gpu, [p, p, p], 8, n=44)


def test_vs_batch_members_with_their_own_parameters(gpu):
    """vs_batch_create_params: the streams of a group may differ in what does not shape a launch - smoothing radius and
    method (the Kalman stream's releases stay inside its tail workgroup, the others' run apart), horizon lock, corner count."""
    ps = [gpu.params(smoothing_radius=7), gpu.params(smoothing_radius=12, smoothing_method=capi.SMOOTH_KALMAN),
          gpu.params(smoothing_radius=5, smoothing_method=capi.SMOOTH_GAUSSIAN, gaussian_sigma=2.5, horizon_lock=1),
          gpu.params(smoothing_radius=9, max_corners=90)]
    _group_vs_solo(gpu, ps, 8, n=60)
    # ... but not in what does: another tracking window is refused when the step runs
    b = gpu.batch([gpu.params(smoothing_radius=7), gpu.params(smoothing_radius=7, lk_win_size=21)], 2, 4)
    clip = synth.make_clip(synth.SEED_CONFIG1, 320, 240, 6)
    fb = clip[0].nbytes
    d_in, d_out = capi.DevBuf(gpu, fb * 6), capi.DevBuf(gpu, fb * 12)
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
    with pytest.raises(capi.VsError):
        for i in range(6):
            b.push_dev([d_in.ptr + i * fb] * 2, 320, 240, 960, capi.FMT_BGR8, [d_out.ptr + i * fb, d_out.ptr + (6 + i) * fb], 960)
    b.close()


def test_members_of_a_vs_batch_are_driven_through_the_group_only(gpu):
    """vs_batch_stream hands out the members for their getters; pushing, flushing, cleaning or re-configuring one behind the
    group's back would leave the group with dangling entries, so those calls are refused (and destroy is ignored)."""
    import ctypes as C
    b = gpu.batch(gpu.params(smoothing_radius=5), 2, 4)
    m = b.stream(0)
    clip = synth.make_clip(synth.SEED_CONFIG1, 320, 240, 2)
    d = capi.DevBuf(gpu, clip[0].nbytes * 2)
    prod = C.c_int(0)
    L = gpu.lib
    assert L.vs_stab_push_dev(m.h, C.c_void_p(d.ptr), 320, 240, 960, capi.FMT_BGR8, C.c_void_p(d.ptr + clip[0].nbytes), 960, C.byref(prod)) == 1   # VS_ERR_INVALID_ARG
    assert L.vs_stab_clean(m.h) == 1 and L.vs_stab_set_batch(m.h, 8) == 1 and L.vs_stab_set_zero_copy(m.h, 1) == 1
    assert L.vs_stab_set_warp_batch(m.h, 4) == 1 and L.vs_stab_set_nv12_layout(m.h, 0, 0) == 1
    L.vs_stab_destroy(m.h)                     # ignored: the group owns the member
    assert m.counters().frames_in == 0         # ... which is still alive
    b.set_zero_copy(True)
    b.push_dev([d.ptr, d.ptr], 320, 240, 960, capi.FMT_BGR8, [d.ptr + clip[0].nbytes] * 2, 960)
    b.sync()
    assert m.counters().frames_in == 1
    b.close()


def _oracle_outputs(oracle, frames, order, **params):
    """The oracle's stabilized frames for the pushes `order` of `frames`, flush included (8 threads)."""
    oracle.lib.vso_set_threads(8)
    try:
        so = oracle.stabilizer(oracle.params(**params))
        out = []
        for i in order:
            r = so.push(frames[i])
            if r is not None:
                out.append(r)
        while True:
            r = so.flush(frames[0])
            if r is None:
                break
            out.append(r)
        so.close()
        return out
    finally:
        oracle.lib.vso_set_threads(1)


def test_bench_settings_full_hd_batches_of_64_against_oracle(gpu, oracle):
    """What bench.py times, proven: BASELINE configs[1] with the bench's own settings - 1920x1080 BGR8, 200 corners, 21x21 LK,
    smoothing radius 30, batches of 64, zero-copy input, a closed-loop clip rendered on the device (synth.make_clip_dev, here
    downloaded for the oracle) - over two FULL batches (each batch's warps are two 32-frame warp_tab_kernel launches with their
    halves of the batch's table set), a partial third one and the drain.  Every output frame array_equal to the oracle's."""
    W, H, NF = 1920, 1080, 40
    fb = W * H * 3
    clip = synth.make_clip_dev(gpu, synth.SEED_CONFIG2, W, H, NF)
    frames = [clip.download((H, W, 3), np.uint8, i * fb) for i in range(NF)]
    n = 1 + 64 + 64 + 37                      # first frame, two full batches, a partial one
    order = [i % NF for i in range(n)]
    params = dict(max_corners=200, lk_win_size=21, lk_max_level=2, lk_max_iters=20, lk_epsilon=0.03, smoothing_radius=30)
    ref = _oracle_outputs(oracle, frames, order, **params)
    s = gpu.stabilizer(gpu.params(**params))
    s.set_batch(64)
    s.set_zero_copy(True)
    d_out = capi.DevBuf(gpu, fb * (n + 1))
    k = 0
    for i in order:
        k += s.push_dev(clip.ptr + i * fb, W, H, W * 3, capi.FMT_BGR8, d_out.ptr + k * fb, W * 3)
    while s.flush_dev(d_out.ptr + k * fb, W * 3):
        k += 1
    s.sync()
    assert k == len(ref) == n
    for j in range(k):
        assert np.array_equal(d_out.download((H, W, 3), np.uint8, j * fb), ref[j]), j
    s.close()
    clip.free()
    d_out.free()


def test_vs_batch_full_hd_8_streams_of_8_frames_against_oracle(gpu, oracle):
    """BASELINE configs[4], one GPU's share at its real size: a vs_batch of 8 streams of 1920x1080, 8 frames per stream and step
    (what `bench.py --streams 8` times), every stream's frames against the ORACLE directly (the small-size group tests compare
    with standalone instances).  Each stream plays its own device-rendered clip; 44 pushes per stream: the 29-frame warm-up of
    radius 30 and two steps of 64 due frames (two 32-frame warp launches each), then the drain."""
    W, H, NF, S, n = 1920, 1080, 12, 8, 46
    fb = W * H * 3
    params = dict(max_corners=200, lk_win_size=21, lk_max_level=2, lk_max_iters=20, lk_epsilon=0.03, smoothing_radius=30)
    order = [i % NF if (i // NF) % 2 == 0 else NF - 1 - i % NF for i in range(n)]
    clips = [synth.make_clip_dev(gpu, synth.SEED_CONFIG2 + g, W, H, NF) for g in range(S)]
    b = gpu.batch(gpu.params(**params), S, 8)
    b.set_zero_copy(True)
    d_out = [capi.DevBuf(gpu, fb * (n + 1)) for _ in range(S)]
    k = [0] * S
    for i in order:
        prod = b.push_dev([c.ptr + i * fb for c in clips], W, H, W * 3, capi.FMT_BGR8, [d_out[g].ptr + k[g] * fb for g in range(S)], W * 3)
        for g in range(S):
            k[g] += prod[g]
    while True:
        prod = b.flush_dev([d_out[g].ptr + k[g] * fb for g in range(S)], W * 3)
        for g in range(S):
            k[g] += prod[g]
        if not any(prod):
            break
    b.sync()
    for g in range(S):
        frames = [clips[g].download((H, W, 3), np.uint8, i * fb) for i in range(NF)]
        ref = _oracle_outputs(oracle, frames, order, **params)
        assert k[g] == len(ref) == n, g
        for j in range(n):
            assert np.array_equal(d_out[g].download((H, W, 3), np.uint8, j * fb), ref[j]), (g, j)
    b.close()
    for buf in clips + d_out:
        buf.free()


def test_config3_chain_4k_nv12_roll_stabilize_zoomcrop_against_oracle(gpu, oracle):
    """BASELINE configs[2] as ONE chain on decoder surfaces: 3840x2160 NV12, RollCorrection -> stabilize (400 corners, batch
    mode) -> AutoZoomCrop, the reference's order of operators (examples/vs.cpp:553-562), every stage through its device entry
    point without a wait per frame: the roll stage closes a frame four calls later, the stabilizer runs batches of 8, the zoom
    stage's contour logic runs on its worker threads.  One host wait per stage and chunk.  Every 640x360 result (and the
    stages' states) against the oracle's chain on the same surfaces."""
    import roll_scene
    W, H, N = 3840, 2160, 14
    base = [synth.bgr_to_nv12(f) for f in synth.make_clip(synth.SEED_CONFIG3, W, H, 7)]
    tilt = synth.bgr_to_nv12(roll_scene.horizon_frame(W, H, 45, seed=7))
    surfs = []
    for i in range(N):
        s = base[i % 7].copy()
        s[H // 3:H // 3 + 400] = tilt[H // 3:H // 3 + 400]          # a band with a tilted horizon: lines for the roll stage
        surfs.append(s)
    sb = W * H * 3 // 2
    params = dict(smoothing_radius=5, max_corners=400)
    # ---- oracle chain
    oracle.lib.vso_set_threads(8)
    try:
        ro, so = oracle.roll_correction(), oracle.stabilizer(oracle.params(**params))
        ref = []
        for s in surfs:
            r = ro.correct_nv12(s, W, H)
            o = so.push(r, capi.FMT_NV12)
            if o is not None:
                ref.append(oracle.auto_zoom_crop_nv12(o, W, H))
        while True:
            o = so.flush(surfs[0], capi.FMT_NV12)
            if o is None:
                break
            ref.append(oracle.auto_zoom_crop_nv12(o, W, H))
        so.close()
    finally:
        oracle.lib.vso_set_threads(1)
    # ---- device chain
    rg, az = gpu.roll_correction(), gpu.auto_zoom_crop()
    sg = gpu.stabilizer(gpu.params(**params))
    sg.set_batch(8)
    sg.set_zero_copy(True)
    d_in = capi.DevBuf(gpu, sb * N)
    for i, s in enumerate(surfs):
        d_in.upload(s, i * sb)
    d_roll, d_stab = capi.DevBuf(gpu, sb * N), capi.DevBuf(gpu, sb * (N + 1))
    zp, zh = W, H                                   # result surfaces that fit the fall-back (unchanged frame) as well
    zb = zp * zh * 3 // 2
    d_zoom = capi.DevBuf(gpu, zb * (N + 1))
    for i in range(N):
        rg.correct_nv12_dev(d_in.ptr + i * sb, W, H, W, d_roll.ptr + i * sb, W)
    rg.sync()
    k = 0
    for i in range(N):
        k += sg.push_dev(d_roll.ptr + i * sb, W, H, W, capi.FMT_NV12, d_stab.ptr + k * sb, W)
    while sg.flush_dev(d_stab.ptr + k * sb, W):
        k += 1
    sg.sync()
    assert k == len(ref) == N
    tickets = [az.apply_nv12_dev(d_stab.ptr + j * sb, W, H, W, d_zoom.ptr + j * zb, zp, zp * zh) for j in range(k)]
    az.sync()
    assert ro.state() == rg.state()
    n_crop = 0
    for j in range(k):
        want, info = ref[j]
        ow, oh, ginfo = az.result(tickets[j])
        assert ginfo.tolist() == info.tolist(), j
        assert (ow, oh) == ((640, 360) if info[7] else (W, H)), j
        got = d_zoom.download((zh * 3 // 2, zp), np.uint8, j * zb)
        assert np.array_equal(got[:oh, :ow], want[:oh]) and np.array_equal(got[zh:zh + oh // 2, :ow], want[oh:]), j
        n_crop += int(info[7])
    assert n_crop >= N - 1
    for o in (rg, az, sg):
        o.close()
    for b in (d_in, d_roll, d_stab, d_zoom):
        b.free()


def test_push_dev_n_equals_the_pushes(gpu):
    """vs_stab_push_dev_n: n consecutive pushes with one trip through the binding; the j-th result that becomes due lands in
    d_outs[j]."""
    clip = synth.make_clip(synth.SEED_CONFIG1 + 5, 320, 240, 12)
    fb = clip[0].nbytes
    n = 40
    order = [i % 12 if (i // 12) % 2 == 0 else 11 - i % 12 for i in range(n)]
    d_in = capi.DevBuf(gpu, fb * 12)
    for i, f in enumerate(clip):
        d_in.upload(f, i * fb)
    res = []
    for form in (0, 1):
        s = gpu.stabilizer(gpu.params(smoothing_radius=6))
        s.set_batch(8)
        s.set_zero_copy(True)
        d_out = capi.DevBuf(gpu, fb * (n + 1))
        if form == 0:
            k = 0
            for i in order:
                k += s.push_dev(d_in.ptr + i * fb, 320, 240, 960, capi.FMT_BGR8, d_out.ptr + k * fb, 960)
        else:
            k = s.push_dev_n([d_in.ptr + i * fb for i in order[:17]], 320, 240, 960, capi.FMT_BGR8, [d_out.ptr + j * fb for j in range(17)], 960)
            k += s.push_dev_n([d_in.ptr + i * fb for i in order[17:]], 320, 240, 960, capi.FMT_BGR8, [d_out.ptr + (k + j) * fb for j in range(n - 17)], 960)
        s.sync()
        res.append((k, d_out.download((k, 240, 320, 3), np.uint8)))
        s.close()
    assert res[0][0] == res[1][0] == n - 5 and np.array_equal(res[0][1], res[1][1])
