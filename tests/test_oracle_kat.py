"""Known-answer tests that pin the CPU oracle (oracle/).

The reference ships no tests or golden vectors, so the
oracle is pinned analytically: closed-form answers each primitive must give,
plus the one numeric vector the reference's dependency fixes (the cv::RNG
stream for seed (uint64)-1).
"""
import numpy as np
import pytest

from vsamd import synth


def test_rng_stream_kat(oracle):
    # cv::RNG MWC: state = (uint32)state*4164903690 + (state>>32); independent re-derivation
    s = (1 << 64) - 1
    exp = []
    for _ in range(8):
        s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & ((1 << 64) - 1)
        exp.append(s & 0xFFFFFFFF)
    got = oracle.rng_stream((1 << 64) - 1, 8)
    assert list(got) == exp
    # SURVEY.md 8a R1 quotes the first outputs and their residues mod 200
    assert [hex(x) for x in got[:4]] == ["0x7c09cf5", "0xbac3439c", "0x99ae7b8c", "0x37275f45"]
    assert [int(x) % 200 for x in got[:4]] == [5, 4, 140, 173]


def test_rng_stream_golden_file(oracle):
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cv_rng_mwc_stream.json")))
    got = oracle.rng_stream(int(g["seed"], 16), len(g["next_u32"]))
    assert [int(x) for x in got] == g["next_u32"]
    assert [int(x) % 200 for x in got] == g["uniform_0_200"]


def test_bgr2gray_known_colors(oracle):
    img = np.zeros((1, 5, 3), np.uint8)
    img[0, 0] = (255, 255, 255)
    img[0, 1] = (255, 0, 0)     # pure blue
    img[0, 2] = (0, 255, 0)     # pure green
    img[0, 3] = (0, 0, 255)     # pure red
    g = oracle.bgr2gray(img)[0]
    assert list(g) == [255, 29, 150, 76, 0]


def test_resize_half_is_box_average(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    out = oracle.resize(img, 48, 32)
    s = img.astype(np.int32)
    exp = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(out, exp.astype(np.uint8))


def test_resize_constant_and_identity(oracle):
    img = np.full((30, 40), 77, np.uint8)
    assert np.all(oracle.resize(img, 60, 45) == 77)          # upscale keeps a constant
    assert np.all(oracle.resize(img, 10, 7) == 77)           # 4x-ish downscale too
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (33, 47), dtype=np.uint8)
    assert np.array_equal(oracle.resize(img, 47, 33), img)    # same size = identity


def test_resize_upscale_2x_midpoints(oracle):
    # 2x upscale of a horizontal ramp: fx = (dx+0.5)/2-0.5 -> weights (0.75,0.25)/(0.25,0.75)
    row = np.arange(0, 160, 16, dtype=np.uint8)
    img = np.tile(row, (4, 1))
    out = oracle.resize(img, 20, 8)
    # interior samples: exact multiples of 4 -> exact lerp
    assert out[3, 1] == 4 and out[3, 2] == 12 and out[3, 3] == 20
    assert out[3, 0] == 0 and out[3, 19] == 144               # clamped ends


def test_pyr_down_constant_and_impulse(oracle):
    img = np.full((40, 50), 200, np.uint8)
    assert np.all(oracle.pyr_down(img) == 200)
    img = np.zeros((41, 51), np.uint8)
    img[20, 24] = 255
    out = oracle.pyr_down(img)
    assert out.shape == (21, 26)
    k = np.array([1, 4, 6, 4, 1])
    # output (10,12) is centred on the impulse: 6*6*255/256 rounded
    assert out[10, 12] == (36 * 255 + 128) >> 8
    assert out[10, 11] == (6 * 1 * 255 + 128) >> 8            # source col 22 is 2 left of the impulse
    assert out[9, 12] == (1 * 6 * 255 + 128) >> 8
    assert k.sum() == 16


def test_scharr_ramp(oracle):
    x = np.arange(40, dtype=np.uint8)
    img = np.tile(x * 3, (20, 1)).astype(np.uint8)
    d = oracle.scharr(img)
    assert np.all(d[5:15, 5:35, 0] == 16 * 2 * 3)              # (3+10+3) * (I[x+1]-I[x-1])
    assert np.all(d[5:15, 5:35, 1] == 0)
    assert np.all(d[:, 0, 0] == 0)                             # REFLECT_101: I[-1] == I[1]


def test_warp_identity_and_integer_shift(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    assert np.array_equal(oracle.warp_affine(img, [1, 0, 0, 0, 1, 0]), img)
    out = oracle.warp_affine(img, [1, 0, 7, 0, 1, 4])
    assert np.array_equal(out[4:, 7:], img[:-4, :-7])
    assert out[:4].max() == 0 and out[:, :7].max() == 0        # BORDER_CONSTANT black
    out = oracle.warp_affine(img, [1, 0, -3, 0, 1, -2])
    assert np.array_equal(out[:-2, :-3], img[2:, 3:])


def test_warp_half_pixel_is_average(oracle):
    img = np.zeros((8, 8), np.uint8)
    img[:, 4] = 200
    out = oracle.warp_affine(img, [1, 0, 0.5, 0, 1, 0])
    # dst(x) samples src(x-0.5): columns 4 and 5 each see half of the bright column
    assert out[3, 4] == 100 and out[3, 5] == 100 and out[3, 3] == 0


def test_warp_mt_equals_st(oracle):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    M = [0.9995, -0.03, 2.25, 0.03, 0.9995, -1.75]
    assert np.array_equal(oracle.warp_affine(img, M, threads=1), oracle.warp_affine(img, M, threads=4))


def test_copy_make_border_modes(oracle):
    row = np.arange(1, 6, dtype=np.uint8)[None, :]
    img = np.repeat(row, 3, 0)
    assert list(oracle.copy_make_border(img, 2, 1)[2]) == [2, 1, 1, 2, 3, 4, 5, 5, 4]     # reflect
    assert list(oracle.copy_make_border(img, 2, 2)[2]) == [3, 2, 1, 2, 3, 4, 5, 4, 3]     # reflect_101
    assert list(oracle.copy_make_border(img, 2, 3)[2]) == [1, 1, 1, 2, 3, 4, 5, 5, 5]     # replicate
    assert list(oracle.copy_make_border(img, 2, 4)[2]) == [4, 5, 1, 2, 3, 4, 5, 1, 2]     # wrap
    assert list(oracle.copy_make_border(img, 2, 0)[2]) == [0, 0, 1, 2, 3, 4, 5, 0, 0]     # black


def test_two_point_similarity_closed_form(oracle):
    a, b, tx, ty = 0.98, 0.05, 3.0, -2.0
    src = np.array([[10, 20], [200, 150]], np.float32)
    dst = np.stack([a * src[:, 0] - b * src[:, 1] + tx, b * src[:, 0] + a * src[:, 1] + ty], 1)
    ok, model, inl, info = oracle.estimate_affine_partial2d(src, dst)
    assert ok and list(inl) == [1, 1]
    assert np.allclose(model, [a, -b, tx, b, a, ty], atol=1e-5)


def test_ransac_recovers_similarity_with_outliers(oracle):
    rng = np.random.default_rng(5)
    n = 200
    src = np.stack([rng.integers(5, 950, n), rng.integers(5, 530, n)], 1).astype(np.float32)
    ang = 0.01
    a, b, tx, ty = 1.002 * np.cos(ang), 1.002 * np.sin(ang), -3.25, 1.5
    dst = np.stack([a * src[:, 0] - b * src[:, 1] + tx, b * src[:, 0] + a * src[:, 1] + ty], 1)
    dst += rng.normal(0, 0.05, dst.shape)
    out_idx = rng.choice(n, 60, replace=False)
    dst[out_idx] += rng.uniform(20, 80, (60, 2)) * rng.choice([-1, 1], (60, 2))
    dst = dst.astype(np.float32)
    ok, model, inl, info = oracle.estimate_affine_partial2d(src, dst)
    assert ok
    exp_inl = np.ones(n, np.uint8)
    exp_inl[out_idx] = 0
    assert np.array_equal(inl, exp_inl)
    # 0.05 px noise over 140 inliers: rotation/scale to 1e-4, translation to a few 1e-2 px
    assert np.allclose(model, [a, -b, tx, b, a, ty], atol=3e-2)
    assert abs(model[0] - a) < 1e-4 and abs(model[3] - b) < 1e-4
    assert info[3] == 140 and 0 <= info[1] < info[2] + 1


def test_ransac_degenerate_inputs(oracle):
    pts = np.array([[1, 1]], np.float32)
    ok, model, inl, info = oracle.estimate_affine_partial2d(pts, pts)
    assert not ok and np.all(np.isnan(model))


def test_kalman_constant_and_box_ramp(oracle):
    p = np.full(50, 3.5, np.float32)
    assert np.allclose(oracle.kalman_filter(p), 3.5)
    ramp = np.arange(60, dtype=np.float32) * 0.5
    out = oracle.box_filter(ramp, 5)
    assert np.allclose(out[5:-5], ramp[5:-5], atol=1e-5)       # symmetric window of a ramp
    assert np.array_equal(oracle.box_filter(ramp[:4], 5), ramp[:4])   # n <= r: unchanged
    # radius actually used is clamp(r,2,8)
    assert np.array_equal(oracle.box_filter(ramp, 30), oracle.box_filter(ramp, 8))


def test_kalman_ramp_tracks_with_lag(oracle):
    ramp = np.arange(200, dtype=np.float32)
    out = oracle.kalman_filter(ramp)
    assert out[0] == 0
    assert abs(out[-1] - ramp[-1]) < 0.5                      # constant-velocity model locks on


def test_gaussian_filter_constant_and_symmetry(oracle):
    p = np.full(40, -2.0, np.float32)
    assert np.allclose(oracle.gaussian_filter(p, 2.0), -2.0, atol=1e-5)
    ramp = np.arange(40, dtype=np.float32)
    out = oracle.gaussian_filter(ramp, 2.0)
    assert np.allclose(out[8:-8], ramp[8:-8], atol=1e-4)


def test_adaptive_radius_and_intent(oracle):
    z = np.zeros(30, np.float32)
    assert oracle.adaptive_radius(z, z, z, 30) == 5            # no variance -> min 5
    assert oracle.adaptive_radius(z[:5], z[:5], z[:5], 30) == 30   # < 10 samples -> smoothingRadius
    big = np.linspace(0, 400, 30).astype(np.float32)
    assert oracle.adaptive_radius(big, z, z, 30) == 25
    # steady fast pan: consistent direction and magnitude > 5 -> DELIBERATE_PAN (1)
    t = np.tile(np.array([8.0, 0.1, 0.0], np.float32), (30, 1))
    assert oracle.motion_intent(t, 20) == 1
    # slow pan stays NORMAL (0)
    t = np.tile(np.array([3.0, 0.0, 0.0], np.float32), (30, 1))
    assert oracle.motion_intent(t, 20) == 0
    # fewer than 15 transforms -> NORMAL
    assert oracle.motion_intent(t[:10], 5) == 0


def test_gftt_finds_rectangle_corners(oracle):
    img = np.full((120, 160), 40, np.uint8)
    img[30:80, 50:120] = 220
    pts, nc = oracle.gftt(img, 10, 0.01, 10.0, 3)
    assert len(pts) == 4
    got = sorted((int(x), int(y)) for x, y in pts)
    for (x, y), (ex, ey) in zip(got, sorted([(50, 30), (119, 30), (50, 79), (119, 79)])):
        assert abs(x - ex) <= 1 and abs(y - ey) <= 1


def test_gftt_min_distance_and_order(oracle):
    clip = synth.make_clip(synth.SEED_CONFIG1, 320, 240, 1)
    g = oracle.bgr2gray(clip[0])
    pts, nc = oracle.gftt(g, 50, 0.02, 15.0, 3)
    assert 4 < len(pts) <= 50 and nc >= len(pts)
    eig = oracle.min_eigen(g, 3)
    vals = [eig[int(y), int(x)] for x, y in pts]
    assert all(vals[i] >= vals[i + 1] for i in range(len(vals) - 1))    # strongest first
    d = pts[:, None, :] - pts[None, :, :]
    d2 = (d ** 2).sum(-1) + np.eye(len(pts)) * 1e9
    assert d2.min() >= 15.0 ** 2 - 1e-3


def test_lk_recovers_subpixel_shift(oracle):
    world = synth.make_world(0x1234, 320, 240)
    f0 = synth.render_frame(world, 320, 240, (256 * 256, 256 * 256, 0))
    f1 = synth.render_frame(world, 320, 240, (256 * 256 + 448, 256 * 256 - 320, 0))   # +1.75, -1.25 px
    g0, g1 = oracle.bgr2gray(f0), oracle.bgr2gray(f1)
    pts, _ = oracle.gftt(g0, 60, 0.02, 12.0, 3)
    nxt, st, err = oracle.pyr_lk(g0, g1, pts)
    assert st.sum() >= 0.8 * len(pts)
    flow = (nxt - pts)[st > 0]
    med = np.median(flow, 0)
    assert abs(med[0] + 1.75) < 0.05 and abs(med[1] - 1.25) < 0.05


def test_stabilizer_latency_contract(oracle):
    """E0: radius r -> clamp(r,5,35)-1 empties, then 1:1; flush drains the rest."""
    clip = synth.make_clip(synth.SEED_CONFIG1, 160, 120, 20)
    s = oracle.stabilizer(oracle.params(smoothing_radius=7))
    produced = [s.push(f) is not None for f in clip]
    assert produced == [False] * 6 + [True] * 14
    n = 0
    while s.flush(clip[0]) is not None:
        n += 1
    assert n == 6
    d = s.debug()
    assert d.out_index == 19


def test_stabilizer_static_clip_is_identity(oracle):
    """A static scene: from the second transform on the measured motion is exactly
    zero, so those outputs equal the input frame.  Output 0 is warped by
    transforms_[0], which is measured between the 480x270 first-frame image
    (upscaled) and the 960x540 one with keypoints in 480x270 coordinates
    (reference quirks Q2 + Q7) and is therefore not the identity."""
    world = synth.make_world(7, 160, 120)
    f = synth.render_frame(world, 160, 120, (256 * 256, 256 * 256, 0))
    s = oracle.stabilizer(oracle.params(smoothing_radius=5))
    outs = []
    for _ in range(12):
        o = s.push(f)
        if o is not None:
            outs.append((s.debug().out_index, o))
    assert [i for i, _ in outs] == list(range(8))
    for i, o in outs[1:]:
        assert np.array_equal(o, f), i


def test_debug_arrays_never_write_past_the_counts_they_report(oracle):
    """Round-1 crash record gpu2.log: the oracle copied a stale list of detected points (200 of them, kept from the
    frame before) into the 2-float array the test had sized by n_detected = 0 - a heap overflow in the test process that
    surfaced as SIGSEGV inside the NEXT call, vs_stab_push.  The copies are bounded by the reported counts now; this
    walks a clip that alternates detecting and non-detecting frames with canaries behind every buffer."""
    import ctypes as C
    from vsamd import synth
    clip = synth.make_clip(synth.SEED_CONFIG1 + 21, 320, 240, 8)
    so = oracle.stabilizer(oracle.params(smoothing_radius=5))
    seen_quiet_frame = False
    for f in clip:
        so.push(f)
        d = so.debug()
        sizes = dict(prev=d.n_prev * 2, cur=d.n_prev * 2, det=d.n_detected * 2)
        bufs = {k: np.full(n + 4096, np.float32(-7.0), np.float32) for k, n in sizes.items()}
        st = np.full(d.n_prev + 4096, 0xAB, np.uint8)
        inl = np.full(d.n_valid + 4096, 0xAB, np.uint8)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        bp = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
        oracle.lib.vso_stab_get_debug_arrays(so.h, fp(bufs["prev"]), fp(bufs["cur"]), bp(st), bp(inl), fp(bufs["det"]), None, None, None)
        for k, n in sizes.items():
            assert (bufs[k][n:] == np.float32(-7.0)).all(), k
        assert (st[d.n_prev:] == 0xAB).all() and (inl[d.n_valid:] == 0xAB).all()
        seen_quiet_frame |= d.n_detected == 0 and d.n_prev > 0
    assert seen_quiet_frame
    so.close()


def test_threaded_stages_do_not_depend_on_the_thread_count(oracle):
    """vso_set_threads spreads rows / points over threads (resize, min-eigenvalue map, LK, warp): same bits as one thread."""
    from vsamd import synth
    f0, f1 = synth.make_clip(synth.SEED_CONFIG1 + 22, 331, 247, 2)

    def everything():
        small = oracle.resize(f0, 223, 131)
        g0, g1 = oracle.bgr2gray(f0), oracle.bgr2gray(f1)
        pts, nc = oracle.gftt(g0, 120, 0.02, 9.0, 3)
        nxt, st, err = oracle.pyr_lk(g0, g1, pts, 15, 2, 20, 0.03)
        so = oracle.stabilizer(oracle.params(smoothing_radius=5))
        outs = [so.push(f) for f in (f0, f1, f0, f1, f0, f1)]
        so.close()
        return [small, pts, np.int64(nc), nxt, st, err] + [o for o in outs if o is not None]

    one = everything()
    try:
        oracle.lib.vso_set_threads(5)
        many = everything()
    finally:
        oracle.lib.vso_set_threads(1)
    assert len(one) == len(many) > 6
    for a, b in zip(one, many):
        assert np.array_equal(a, b)


def test_oracle_snapshot(oracle):
    """The oracle is the parity target of every GPU test: its outputs on fixed inputs are pinned by digest
    (tests/golden/oracle_snapshot.json, made by tests/golden/make_oracle_snapshot.py), so that an edit of oracle/ that
    changes results cannot go unnoticed.  A deliberate correction regenerates the file in the same commit."""
    import importlib.util
    import json
    import os
    here = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_oracle_snapshot", os.path.join(here, "make_oracle_snapshot.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(here, "oracle_snapshot.json")))
    got = mod.snapshot(oracle)
    assert sorted(got) == sorted(want)
    changed = [k for k in want if got[k] != want[k]]
    assert not changed, "oracle outputs changed for: %s" % ", ".join(changed)
