"""Pins the CPU oracle to a real OpenCV, primitive by primitive (docs/opencv_semantics.md lists what each test decides).

This image has no OpenCV, so the whole module SKIPS here and on the GPU box: the oracle stays "parity unpinned".  On any
machine with `cv2` (4.x) importable, `python -m pytest tests/test_opencv_pin.py -q` is the one command that pins or
refutes every restated primitive; it needs no GPU and nothing of the reference.  The C++ twin for a machine with the
OpenCV development package but no Python binding is oracle/opencv_pin/ (`make -C oracle opencv-pin`).

Exactness: the 8-bit primitives are compared byte for byte.  OpenCV builds differ in SIMD back ends (and IPP), which
for the float stages (min-eigenvalue map, LK) may move the last bits: those compare with the tolerance written at the
assertion, and the test prints how many values were not bit-identical so that a maintainer can see which it is.
"""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

from vsamd import synth      # noqa: E402

import roll_scene             # noqa: E402


def frames(n=3, w=320, h=240, seed=3):
    return synth.make_clip(synth.SEED_CONFIG1 + seed, w, h, n)


def gray_of(f):
    return cv2.cvtColor(f, cv2.COLOR_BGR2GRAY)


# ---------------------------------------------------------------------------------------- analysis front end
@pytest.mark.parametrize("dst", [(960, 540), (480, 270), (160, 120), (640, 480), (333, 211)])
def test_resize_linear(oracle, dst):
    f = frames(1)[0]
    assert np.array_equal(oracle.resize(f, *dst), cv2.resize(f, dst, interpolation=cv2.INTER_LINEAR))
    g = gray_of(f)
    assert np.array_equal(oracle.resize(g, *dst), cv2.resize(g, dst, interpolation=cv2.INTER_LINEAR))


def test_bgr2gray(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    assert np.array_equal(oracle.bgr2gray(img), gray_of(img))


def test_pyr_down_and_scharr_as_the_lk_pyramid_uses_them(oracle):
    g = gray_of(frames(1, 321, 243)[0])
    assert np.array_equal(oracle.pyr_down(g), cv2.pyrDown(g))
    d = oracle.scharr(g)
    assert np.array_equal(d[..., 0], cv2.Scharr(g, cv2.CV_16S, 1, 0))
    assert np.array_equal(d[..., 1], cv2.Scharr(g, cv2.CV_16S, 0, 1))


# ---------------------------------------------------------------------------------------- detection / tracking
def test_min_eigen_and_good_features(oracle):
    g = gray_of(frames(1, 480, 270)[0])
    a, b = oracle.min_eigen(g, 3), cv2.cornerMinEigenVal(g, 3, ksize=3)
    print("min-eigen values not bit-identical:", int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))))
    assert np.allclose(a, b, rtol=1e-5, atol=1e-9)
    for max_corners, q, dist, bs in [(200, 0.01, 30.0, 3), (200, 0.02, 15.0, 3), (50, 0.05, 8.0, 5)]:
        pts, _ = oracle.gftt(g, max_corners, q, dist, bs)
        ref = cv2.goodFeaturesToTrack(g, max_corners, q, dist, blockSize=bs)
        ref = np.zeros((0, 2), np.float32) if ref is None else ref.reshape(-1, 2)
        assert np.array_equal(pts, ref)


def test_pyr_lk(oracle):
    f0, f1 = frames(2, 480, 270)
    g0, g1 = gray_of(f0), gray_of(f1)
    pts, _ = oracle.gftt(g0, 200, 0.01, 30.0, 3)
    nxt, st, err = oracle.pyr_lk(g0, g1, pts, win=15, max_level=2, iters=20, eps=0.03)
    crit = (cv2.TERM_CRITERIA_COUNT + cv2.TERM_CRITERIA_EPS, 20, 0.03)
    rn, rs, re = cv2.calcOpticalFlowPyrLK(g0, g1, pts.reshape(-1, 1, 2), None, winSize=(15, 15), maxLevel=2, criteria=crit)
    assert np.array_equal(st, rs.reshape(-1))
    ok = st != 0
    d = np.abs(nxt[ok] - rn.reshape(-1, 2)[ok])
    print("LK points not bit-identical:", int(np.count_nonzero(d)), "max", float(d.max(initial=0)))
    assert d.max(initial=0) <= 1e-2          # float accumulation order of the 2x2 system differs per OpenCV build (vso_lk.cpp)


def test_estimate_affine_partial_2d(oracle):
    rng = np.random.default_rng(5)
    for n, out_frac in [(4, 0.0), (30, 0.2), (150, 0.4), (200, 0.0)]:
        a = rng.uniform(0, 900, (n, 2)).astype(np.float32)
        ang, s = 0.01, 1.002
        R = np.array([[np.cos(ang) * s, -np.sin(ang) * s], [np.sin(ang) * s, np.cos(ang) * s]])
        b = (a @ R.T + [3.5, -2.25] + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
        k = int(n * out_frac)
        b[:k] += rng.uniform(-80, 80, (k, 2)).astype(np.float32)
        ok, model, inl, _ = oracle.estimate_affine_partial2d(a, b, 5.0, 500)
        M, mask = cv2.estimateAffinePartial2D(a, b, method=cv2.RANSAC, ransacReprojThreshold=5.0, maxIters=500)
        assert bool(ok) == (M is not None)
        if M is not None:
            assert np.array_equal(inl, mask.reshape(-1))
            assert np.array_equal(model.reshape(2, 3), M)       # the RNG stream, the sample order and LM refinement agree


# ---------------------------------------------------------------------------------------- output stage
@pytest.mark.parametrize("mode,name", [(cv2.BORDER_CONSTANT, 0), (cv2.BORDER_REFLECT, 1), (cv2.BORDER_REFLECT_101, 2),
                                       (cv2.BORDER_REPLICATE, 3), (cv2.BORDER_WRAP, 4)])
def test_copy_make_border(oracle, mode, name):
    f = frames(1, 97, 61)[0]
    assert np.array_equal(oracle.copy_make_border(f, 20, name), cv2.copyMakeBorder(f, 20, 20, 20, 20, mode, value=(0, 0, 0)))


def test_warp_affine_float_matrix_constant_border(oracle):
    f = frames(1, 640, 360)[0]
    rng = np.random.default_rng(2)
    for _ in range(6):
        da = float(rng.normal(0, 0.01))
        M = np.array([[np.cos(da), -np.sin(da), rng.normal(0, 8)], [np.sin(da), np.cos(da), rng.normal(0, 8)]], np.float32)
        ref = cv2.warpAffine(f, M, (f.shape[1], f.shape[0]), flags=cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT)
        assert np.array_equal(oracle.warp_affine(f, M), ref)
        g = gray_of(f)
        assert np.array_equal(oracle.warp_affine(g, M), cv2.warpAffine(g, M, (g.shape[1], g.shape[0]), flags=cv2.INTER_LINEAR))


@pytest.mark.parametrize("mode,name", [(cv2.BORDER_REPLICATE, 3), (cv2.BORDER_REFLECT, 1)])
def test_warp_affine_other_borders(oracle, mode, name):
    """roll correction rotates with REPLICATE (RollCorrection.cpp:84-85); the canvas compensates with REFLECT (:2439)."""
    f = frames(1, 320, 200)[0]
    M = cv2.getRotationMatrix2D((160.0, 100.0), 3.7, 1.0)
    M[:, 2] += (12.5, -7.25)
    ref = cv2.warpAffine(f, M, (320, 200), flags=cv2.INTER_LINEAR, borderMode=mode)
    assert np.array_equal(oracle.warp_affine_d(f, M.reshape(6), border=name), ref)


def test_add_weighted_and_fade_history_update(oracle):
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    for alpha in (0.1, 0.35, np.float32(0.1) * np.float32(7 / 30)):
        alpha = float(np.float32(alpha))
        beta = float(np.float32(1.0) - np.float32(alpha))
        assert np.array_equal(oracle.add_weighted(a, alpha, b, beta), cv2.addWeighted(a, alpha, b, beta, 0.0))


# ---------------------------------------------------------------------------------------- contours (auto zoom/crop, canvas)
def test_find_contours_external_simple_order_and_points(oracle):
    import test_azc
    for m in test_azc.random_masks():
        ref, _ = cv2.findContours(m.copy(), cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)
        got = oracle.find_contours(m)
        assert len(got) == len(ref)
        for g, r in zip(got, ref):                    # same contours in the same order, same starting points
            assert np.array_equal(g, r.reshape(-1, 2))
        for i, r in enumerate(ref):
            want = np.zeros_like(m)
            cv2.drawContours(want, ref, i, 255, -1)
            assert np.array_equal(oracle.fill_contour(r.reshape(-1, 2), m.shape[1], m.shape[0]) != 0, want != 0)


def test_content_mask_threshold_and_close(oracle):
    f = frames(1, 200, 120)[0].copy()
    f[30:50, 40:90] = 0
    f[0:8, :] = 1
    g = gray_of(f)
    _, t = cv2.threshold(g, 1, 255, cv2.THRESH_BINARY)
    want = cv2.morphologyEx(t, cv2.MORPH_CLOSE, cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (5, 5)))
    assert np.array_equal(oracle.content_mask(f), want)


# ---------------------------------------------------------------------------------------- roll correction
def test_canny_and_hough_lines(oracle):
    g = gray_of(roll_scene.horizon_frame(480, 270, 45, seed=1))
    for lo, hi in [(50, 150), (30, 90)]:
        e = oracle.canny(g, lo, hi)
        assert np.array_equal(e, cv2.Canny(g, lo, hi))
        for thr in (60, 100):
            ref = cv2.HoughLines(e, 1.0, np.pi / 180.0, thr)
            ref = np.zeros((0, 2), np.float32) if ref is None else ref.reshape(-1, 2)
            assert np.array_equal(oracle.hough_lines(e, 1.0, np.pi / 180.0, thr), ref)     # same lines, same (vote) order


# ---------------------------------------------------------------------------------------- enhancer
def test_gaussian_blur_clahe_and_nl_means(oracle):
    f = frames(1, 160, 120)[0]
    for sigma in (1.0, 2.5):
        assert np.array_equal(oracle.gaussian_blur(f, sigma), cv2.GaussianBlur(f, (0, 0), sigma))
    lab = cv2.cvtColor(f, cv2.COLOR_BGR2Lab)
    for clip, tiles in [(2.0, 8), (4.0, 4)]:
        want = cv2.createCLAHE(clipLimit=clip, tileGridSize=(tiles, tiles)).apply(lab[..., 0])
        assert np.array_equal(oracle.clahe(np.ascontiguousarray(lab[..., 0]), clip, tiles), want)
    small = f[:48, :64]
    assert np.array_equal(oracle.denoise_colored(small, 5.0, 5.0), cv2.fastNlMeansDenoisingColored(small, None, 5.0, 5.0, 7, 21))
