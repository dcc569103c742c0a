"""Roll correction: oracle known-answer tests (CPU) and HIP-vs-oracle
parity (GPU) for Canny, HoughLines, warpAffine(BORDER_REPLICATE) and the whole
RollCorrection::autoCorrectRoll step (RollCorrection.cpp:16-155).

Bar: bit-exact edges, identical line lists (order included), identical corrected frames;
the smoothed angle is compared bit-exactly too (same double recursion on both sides).
"""
import math

import numpy as np
import pytest

import roll_scene

THETA = np.float32(math.pi / 180.0)


# ---------------------------------------------------------------- oracle KATs (CPU)
def test_sobel_on_ramp(oracle):
    g = np.tile(np.arange(0, 64, 2, dtype=np.uint8), (16, 1))     # d/dx = 2 per pixel
    dx, dy = oracle.sobel16(g)
    assert (dx[:, 1:-1] == 16).all()          # (1+2+1) * (2*2)
    assert (dx[:, 0] == 8).all() and (dx[:, -1] == 8).all()       # BORDER_REPLICATE halves it
    assert (dy == 0).all()


def test_canny_step_edge_single_line(oracle):
    g = np.zeros((32, 48), np.uint8)
    g[16:] = 200
    e = oracle.canny(g, 50, 150)
    rows = np.nonzero(e.any(axis=1))[0]
    assert len(rows) == 1 and rows[0] in (15, 16)      # one-pixel-thin edge
    assert (e[rows[0]] == 255).all()
    assert set(np.unique(e)) <= {0, 255}


def test_canny_hysteresis_keeps_weak_only_when_connected(oracle):
    g = np.zeros((40, 80), np.uint8)
    g[20:, :40] = 200          # strong step (|dy| = 800 > 150)
    g[20:, 40:] = 25           # weak step (|dy| = 100: between the thresholds), connected to the strong one
    h = np.zeros((40, 80), np.uint8)
    h[20:, :] = 25             # the same weak step, nothing strong to attach to
    e1, e2 = oracle.canny(g, 50, 150), oracle.canny(h, 50, 150)
    assert e1[:, 45:75].any() and not e2.any()


def test_hough_horizontal_and_vertical_lines(oracle):
    e = np.zeros((100, 160), np.uint8)
    e[37, :] = 255             # horizontal: theta = 90 deg, rho = 37
    lines = oracle.hough_lines(e, 1.0, THETA, 100)
    assert len(lines) >= 1
    assert lines[0, 0] == 37.0 and lines[0, 1] == np.float32(90) * THETA
    e = np.zeros((160, 100), np.uint8)
    e[:, 22] = 255             # vertical: theta = 0, rho = 22
    lines = oracle.hough_lines(e, 1.0, THETA, 100)
    assert lines[0, 0] == 22.0 and lines[0, 1] == 0.0


def test_hough_orders_by_votes(oracle):
    e = np.zeros((120, 200), np.uint8)
    e[30, :] = 255
    e[80, :150] = 255
    lines = oracle.hough_lines(e, 1.0, THETA, 100)
    rhos = [l[0] for l in lines if l[1] == np.float32(90) * THETA]
    assert rhos[:2] == [30.0, 80.0]


def test_warp_replicate_border_translation(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    out = oracle.warp_affine_d(img, [1, 0, 4, 0, 1, -3], border=3)
    assert (out[:17, 4:] == img[3:, :26]).all()
    assert (out[:17, :4] == img[3:, :1]).all()          # left columns replicate column 0
    assert (out[17:, 4:] == img[19:, :26]).all()        # bottom rows replicate the last row


def test_roll_converges_to_horizon_tilt(oracle):
    # horizon dropping to the right by atan(0.05) = 2.86 deg; the loop walks at most 0.5 deg/frame
    # with alpha 0.1 towards the detected angle and must not overshoot it.
    f = roll_scene.horizon_frame(640, 360, 51, texture=False)
    r = oracle.roll_correction()
    prev = 0.0
    for i in range(12):
        r.correct(f)
        s, d, n, u = r.state()
        assert n > 0 and u > 0
        assert abs(d - math.degrees(math.atan(51 / 1024))) < 1.0
        assert abs(s - prev) <= 0.5 + 1e-12 and abs(s) <= abs(d) + 1e-9 and s * d >= 0
        prev = s
    assert abs(prev) > 1.0


def test_roll_decays_without_lines(oracle):
    f = roll_scene.horizon_frame(320, 240, 40, texture=False)
    r = oracle.roll_correction(oracle.roll_params(hough_threshold=40))
    for _ in range(5):
        r.correct(f)
    s0 = r.state()[0]
    assert s0 != 0.0
    flat = np.full((240, 320, 3), 128, np.uint8)
    out = r.correct(flat)
    s1, _, n, _ = r.state()
    assert n == 0 and s1 == s0 * 0.995
    assert (out == 128).all()


def test_roll_params_default_match(oracle, vs):
    a, b = oracle.roll_params(), vs.roll_params()
    assert bytes(a) == bytes(b)
    assert a.scale_factor == 0.25 and a.hough_threshold == 100 and a.angle_decay == 0.995


# ---------------------------------------------------------------- HIP parity (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("shape,seed", [((135, 240), 1), ((270, 480), 2), ((101, 67), 3), ((8, 8), 4),
                                        ((200, 1100), 5), ((130, 1024), 6), ((63, 64), 7), ((62, 65), 8)])
def test_canny_bit_exact(gpu, oracle, shape, seed):
    g = roll_scene.noisy_gray(shape[1], shape[0], seed)
    for lo, hi in ((50, 150), (10, 30), (150, 50)):
        ref = oracle.canny(g, lo, hi)
        got = gpu.canny(g, lo, hi)
        assert np.array_equal(ref, got)
    assert shape[0] < 32 or oracle.canny(g, 10, 30).any()


@pytest.mark.gpu
def test_canny_hysteresis_follows_a_long_serpentine(gpu, oracle):
    """One strong spot at the head of a weak serpentine bar: the growth has to cross the 62-row bands of the
    hysteresis kernel a few dozen times (several groups of passes) and to run along rows and columns."""
    h, w = 400, 300
    g = np.full((h, w), 100, np.uint8)
    xs = list(range(10, w - 20, 24))
    for i, x in enumerate(xs):
        g[20:h - 20, x:x + 6] = 112
        if i + 1 < len(xs):
            y = h - 26 if i % 2 == 0 else 20
            g[y:y + 6, x:x + 30] = 112
    g[20:26, 10:16] = 230
    ref = oracle.canny(g, 20, 100)
    assert ref[h // 2, xs[-1] - 1] or ref[h // 2, xs[-1]] or ref[h // 2, xs[-1] + 1]      # the far end is reached
    assert np.array_equal(ref, gpu.canny(g, 20, 100))
    weak_only = g.copy()
    weak_only[20:26, 10:16] = 112
    ref0 = oracle.canny(weak_only, 20, 100)
    assert not ref0.any()
    assert np.array_equal(ref0, gpu.canny(weak_only, 20, 100))


@pytest.mark.gpu
def test_canny_on_rendered_frames(gpu, oracle):
    from vsamd import synth
    for f in synth.make_clip(synth.SEED_CONFIG1, 320, 240, 3):
        g = oracle.analysis_gray(f, 480, 360)
        assert np.array_equal(oracle.canny(g, 50, 150), gpu.canny(g, 50, 150))


@pytest.mark.gpu
@pytest.mark.parametrize("thr", [100, 40, 15])
def test_hough_lines_exact(gpu, oracle, thr):
    f = roll_scene.horizon_frame(480, 270, 37, seed=5)
    g = oracle.bgr2gray(f)
    e = oracle.canny(g, 50, 150)
    ref = oracle.hough_lines(e, 1.0, THETA, thr)
    got = gpu.hough_lines(e, 1.0, THETA, thr)
    assert len(ref) > 0 and len(ref) < 8192
    assert ref.shape == got.shape and np.array_equal(ref, got)


@pytest.mark.gpu
def test_hough_other_resolutions(gpu, oracle):
    e = oracle.canny(roll_scene.noisy_gray(200, 150, 9), 20, 60)
    for rho, theta, thr in ((2.0, THETA, 30), (1.0, np.float32(math.pi / 90), 25), (0.5, np.float32(math.pi / 360), 20)):
        ref = oracle.hough_lines(e, rho, theta, thr)
        got = gpu.hough_lines(e, rho, theta, thr)
        assert ref.shape == got.shape and np.array_equal(ref, got)


@pytest.mark.gpu
def test_hough_empty_edges(gpu, oracle):
    e = np.zeros((64, 64), np.uint8)
    assert len(gpu.hough_lines(e, 1.0, THETA, 10)) == 0


@pytest.mark.gpu
def test_hough_resolution_out_of_range_is_an_error_not_a_crash(gpu):
    """hough_rho / hough_theta come from config.yaml: values that would overflow the accumulator geometry are refused."""
    from vsamd import capi
    e = np.zeros((64, 64), np.uint8)
    for rho, theta in ((1.0, 1e-12), (1e-9, THETA), (1.0, -1.0), (0.0, THETA), (1.0, float("nan"))):
        with pytest.raises(capi.VsError):
            gpu.hough_lines(e, rho, theta, 10)
    assert len(gpu.hough_lines(e, 1.0, THETA, 10)) == 0          # the library is still usable


@pytest.mark.gpu
@pytest.mark.parametrize("deg", [0.0, 0.7, -2.3, 9.5, 45.0])
def test_warp_replicate_bit_exact(gpu, oracle, deg):
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (180, 320, 3), dtype=np.uint8)
    a = math.radians(deg)
    al, be = math.cos(a), math.sin(a)
    cx, cy = 160.0, 90.0
    M = [al, be, (1 - al) * cx - be * cy, -be, al, be * cx + (1 - al) * cy]
    for border in (0, 3):
        assert np.array_equal(oracle.warp_affine_d(img, M, border), gpu.warp_affine_ex(img, M, border))


@pytest.mark.gpu
@pytest.mark.parametrize("size,slope", [((640, 360), 51), ((1280, 720), -33), ((322, 242), 20)])
def test_roll_correct_matches_oracle(gpu, oracle, size, slope):
    w, h = size
    thr = 100 if w >= 640 else 40
    ro = oracle.roll_correction(oracle.roll_params(hough_threshold=thr))
    rg = gpu.roll_correction(gpu.roll_params(hough_threshold=thr))
    moved = False
    for i in range(8):
        f = roll_scene.horizon_frame(w, h, slope + i, seed=i, offset=i - 3)
        a, b = ro.correct(f), rg.correct(f)
        assert ro.state() == rg.state()
        assert np.array_equal(a, b)
        moved |= ro.state()[0] != 0.0
    assert moved
    flat = np.full((h, w, 3), 77, np.uint8)
    assert np.array_equal(ro.correct(flat), rg.correct(flat))      # decay branch
    assert ro.state() == rg.state()


@pytest.mark.gpu
def test_roll_correct_when_the_edge_growth_needs_many_passes(gpu, oracle):
    """The roll stage builds its line search on the edge map of the first four hysteresis passes and redoes it
    when the growth had not ended by then: a weak serpentine with one strong spot needs dozens of passes."""
    h, w = 400, 300
    g = np.full((h, w), 100, np.uint8)
    xs = list(range(10, w - 20, 24))
    for i, x in enumerate(xs):
        g[20:h - 20, x:x + 6] = 112
        if i + 1 < len(xs):
            y = h - 26 if i % 2 == 0 else 20
            g[y:y + 6, x:x + 30] = 112
    g[20:26, 10:16] = 230
    f = np.repeat(g[:, :, None], 3, axis=2)
    kw = dict(scale_factor=1.0, canny_threshold_low=20, canny_threshold_high=100, hough_threshold=60,
              angle_filter_min=-100.0, angle_filter_max=100.0)
    ro, rg = oracle.roll_correction(oracle.roll_params(**kw)), gpu.roll_correction(gpu.roll_params(**kw))
    for _ in range(2):
        a, b = ro.correct(f), rg.correct(f)
        assert ro.state() == rg.state()
        assert np.array_equal(a, b)
    assert ro.state()[2] > 4                                   # the long bars were found as lines: the full edge map was used
    weak = f.copy()
    weak[20:26, 10:16] = 112
    ro2, rg2 = oracle.roll_correction(oracle.roll_params(**kw)), gpu.roll_correction(gpu.roll_params(**kw))
    assert np.array_equal(ro2.correct(weak), rg2.correct(weak)) and ro2.state() == rg2.state() and ro2.state()[2] == 0


@pytest.mark.gpu
def test_roll_device_entry_point(gpu, oracle):
    from vsamd.capi import DevBuf
    f = roll_scene.horizon_frame(640, 360, 45, seed=2)
    ro = oracle.roll_correction()
    rg = gpu.roll_correction()
    d_in, d_out = DevBuf.from_array(gpu, f), DevBuf(gpu, f.nbytes)
    for _ in range(3):
        ref = ro.correct(f)
        rg.correct_dev(d_in.ptr, 640, 360, 640 * 3, d_out.ptr, 640 * 3)
        rg.sync()
        assert np.array_equal(ref, d_out.download(f.shape, np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("size,slope,padded", [((640, 360), 51, False), ((1280, 720), -33, True), ((322, 242), 20, True)])
def test_roll_correct_nv12_async_matches_oracle(gpu, oracle, size, slope, padded):
    """vs_roll_correct_nv12_dev: NV12 surfaces in HBM, no wait for the device per frame - the line searches of four frames are
    in flight on worker streams, the smoothed angle advances in call order when a frame's 24-byte result has arrived, and a
    frame's rotation (both planes) is queued four calls later or by the sync.  Eleven frames (a flat one for the decay
    branch in the middle), decoder-style surfaces (padded pitch, chroma plane behind padding rows) in two of three cases:
    planes and state equal the oracle's, padding untouched."""
    from vsamd.capi import DevBuf
    from vsamd import synth
    w, h = size
    thr = 100 if w >= 640 else 40
    ro = oracle.roll_correction(oracle.roll_params(hough_threshold=thr))
    rg = gpu.roll_correction(gpu.roll_params(hough_threshold=thr))
    frames = [roll_scene.horizon_frame(w, h, slope + i, seed=i, offset=i - 3) for i in range(10)]
    frames.insert(5, np.full((h, w, 3), 77, np.uint8))
    surfs = [synth.bgr_to_nv12(f) for f in frames]
    pitch = (w + 63) // 64 * 64 + 64 if padded else w
    hal = h + 6 if padded else h                     # rows the luma plane is allotted
    sb = pitch * (hal + h // 2)
    host = np.full((len(surfs), hal + h // 2, pitch), 0xA5, np.uint8)
    for i, s in enumerate(surfs):
        host[i, :h, :w] = s[:h]
        host[i, hal:hal + h // 2, :w] = s[h:]
    d_in = DevBuf.from_array(gpu, host)
    d_out = DevBuf(gpu, host.nbytes)
    gpu.check(gpu.lib.vs_dev_memset(d_out.ptr, 0x5A, host.nbytes))
    for i in range(len(surfs)):
        rg.correct_nv12_dev(d_in.ptr + i * sb, w, h, pitch, d_out.ptr + i * sb, pitch, uv_offset=pitch * hal, out_uv_offset=pitch * hal)
    rg.sync()
    got = d_out.download(host.shape, np.uint8)
    for i, s in enumerate(surfs):
        ref = ro.correct_nv12(s, w, h)
        assert np.array_equal(got[i, :h, :w], ref[:h]), i
        assert np.array_equal(got[i, hal:hal + h // 2, :w], ref[h:]), i
    assert ro.state() == rg.state() and ro.state()[0] != 0.0
    if padded:
        assert (got[:, :h, w:] == 0x5A).all() and (got[:, h:hal] == 0x5A).all() and (got[:, hal:, w:] == 0x5A).all()


@pytest.mark.gpu
def test_roll_correct_nv12_async_when_the_edge_growth_needs_many_passes(gpu, oracle):
    """The asynchronous path checks the hysteresis flag when it closes a frame and finishes the growth then (the rare frame)."""
    from vsamd.capi import DevBuf
    h, w = 400, 300
    g = np.full((h, w), 100, np.uint8)
    xs = list(range(10, w - 20, 24))
    for i, x in enumerate(xs):
        g[20:h - 20, x:x + 6] = 112
        if i + 1 < len(xs):
            y = h - 26 if i % 2 == 0 else 20
            g[y:y + 6, x:x + 30] = 112
    g[20:26, 10:16] = 230
    surf = np.concatenate([g, np.full((h // 2, w), 128, np.uint8)])
    kw = dict(scale_factor=1.0, canny_threshold_low=20, canny_threshold_high=100, hough_threshold=60,
              angle_filter_min=-100.0, angle_filter_max=100.0)
    ro, rg = oracle.roll_correction(oracle.roll_params(**kw)), gpu.roll_correction(gpu.roll_params(**kw))
    d_in, d_out = DevBuf.from_array(gpu, surf), DevBuf(gpu, surf.nbytes * 6)
    for i in range(6):
        rg.correct_nv12_dev(d_in.ptr, w, h, w, d_out.ptr + i * surf.nbytes, w)
    rg.sync()
    got = d_out.download((6,) + surf.shape, np.uint8)
    for i in range(6):
        assert np.array_equal(got[i], ro.correct_nv12(surf, w, h)), i
    assert ro.state() == rg.state() and ro.state()[2] > 4


@pytest.mark.gpu
def test_roll_correct_nv12_array_form_equals_the_calls(gpu, oracle):
    """vs_roll_correct_nv12_dev_n: n surfaces with one trip through the binding = n calls."""
    from vsamd.capi import DevBuf
    from vsamd import synth
    w, h, n = 640, 360, 19
    surfs = np.stack([synth.bgr_to_nv12(roll_scene.horizon_frame(w, h, 40 + i, seed=i, offset=i % 5 - 2)) for i in range(n)])
    sb = surfs[0].nbytes
    d_in = DevBuf.from_array(gpu, surfs)
    outs = []
    for form in (0, 1):
        rg = gpu.roll_correction()
        d_out = DevBuf(gpu, surfs.nbytes)
        if form == 0:
            for i in range(n):
                rg.correct_nv12_dev(d_in.ptr + i * sb, w, h, w, d_out.ptr + i * sb, w)
        else:
            rg.correct_nv12_dev_n([d_in.ptr + i * sb for i in range(n)], w, h, w, [d_out.ptr + i * sb for i in range(n)], w)
        rg.sync()
        outs.append((d_out.download(surfs.shape, np.uint8), rg.state()))
        rg.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    ro = oracle.roll_correction()
    for i in range(n):
        assert np.array_equal(outs[1][0][i], ro.correct_nv12(surfs[i], w, h)), i


@pytest.mark.gpu
def test_roll_correct_nv12_async_full_size_analysis(gpu, oracle):
    """scale_factor 1 at 1920 x 1080: the batched line search on an analysis image of 30 words x 18 hysteresis bands per frame
    (two workgroups side by side per band), nine frames = a batch of eight and one of one."""
    from vsamd.capi import DevBuf
    from vsamd import synth
    w, h, n = 1920, 1080, 9
    kw = dict(scale_factor=1.0, hough_threshold=300)
    ro, rg = oracle.roll_correction(oracle.roll_params(**kw)), gpu.roll_correction(gpu.roll_params(**kw))
    surfs = np.stack([synth.bgr_to_nv12(roll_scene.horizon_frame(w, h, 30 + 2 * i, seed=i, offset=i - 4)) for i in range(n)])
    d_in, d_out = DevBuf.from_array(gpu, surfs), DevBuf(gpu, surfs.nbytes)
    sb = surfs[0].nbytes
    rg.correct_nv12_dev_n([d_in.ptr + i * sb for i in range(n)], w, h, w, [d_out.ptr + i * sb for i in range(n)], w)
    rg.sync()
    got = d_out.download(surfs.shape, np.uint8)
    for i in range(n):
        assert np.array_equal(got[i], ro.correct_nv12(surfs[i], w, h)), i
    assert ro.state() == rg.state() and ro.state()[0] != 0.0


@pytest.mark.gpu
def test_roll_and_zoom_async_objects_survive_a_change_of_geometry(gpu, oracle):
    """One vs_roll and one vs_azc object fed 640 x 360 surfaces, then 800 x 450, then 640 x 360 again without a sync in between:
    a batch closes when the geometry changes, work areas and mask buffers are rebuilt per size, state carries over; against the
    oracle objects fed the same sequence."""
    from vsamd.capi import DevBuf
    from vsamd import synth
    seq = [(640, 360)] * 5 + [(800, 450)] * 11 + [(640, 360)] * 3
    surfs = [synth.bgr_to_nv12(roll_scene.horizon_frame(w, h, 30 + 3 * i, seed=i, offset=i % 7 - 3)) for i, (w, h) in enumerate(seq)]
    kw = dict(hough_threshold=60)
    ro, rg = oracle.roll_correction(oracle.roll_params(**kw)), gpu.roll_correction(gpu.roll_params(**kw))
    az = gpu.auto_zoom_crop()
    d_in = [DevBuf.from_array(gpu, s) for s in surfs]
    d_rot = [DevBuf(gpu, s.nbytes) for s in surfs]
    d_zoom = [DevBuf(gpu, max(s.nbytes, 640 * 360 * 3 // 2)) for s in surfs]
    for (w, h), a, b in zip(seq, d_in, d_rot):
        rg.correct_nv12_dev(a.ptr, w, h, w, b.ptr, w)
    rg.sync()
    tickets = [az.apply_nv12_dev(b.ptr, w, h, w, c.ptr, max(w, 640), max(w, 640) * max(h, 360)) for (w, h), b, c in zip(seq, d_rot, d_zoom)]
    az.sync()
    for i, ((w, h), s) in enumerate(zip(seq, surfs)):
        rot = ro.correct_nv12(s, w, h)
        assert np.array_equal(d_rot[i].download(s.shape, np.uint8), rot), i
        ref, info = oracle.auto_zoom_crop_nv12(rot, w, h)
        ow, oh, ginfo = az.result(tickets[i])
        assert ginfo.tolist() == info.tolist() and (ow, oh) == ((640, 360) if info[7] else (w, h)), i
        op = max(w, 640)
        got = d_zoom[i].download((max(h, 360) * 3 // 2, op), np.uint8)
        assert np.array_equal(got[:oh, :ow], ref[:oh]) and np.array_equal(got[max(h, 360):max(h, 360) + oh // 2, :ow], ref[oh:]), i
    assert ro.state() == rg.state()
    rg.close()
    az.close()
