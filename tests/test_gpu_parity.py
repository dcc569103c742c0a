"""HIP path vs CPU oracle, through the C ABI (libvideo-stab.so), on a real MI355X.

Bar: bit-exact for EVERY compared output.  Integer/byte/index outputs (gray,
pyramid, derivatives, warped pixels under an identical matrix, feature lists and
their order, LK status, inlier masks, chosen hypothesis) and float outputs alike
(eigenvalue map, LK positions, refined model: the same IEEE operation sequence on
both sides; atan2f / sinf / cosf: one frozen restatement of glibc's algorithms on
both sides, vs_libm.h, held against the host libm by tests/test_libm.py).  No
tolerance is used anywhere in this file.
"""
import numpy as np
import pytest

from vsamd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def clip_small():
    return synth.make_clip(synth.SEED_CONFIG1, 320, 240, 6)


@pytest.fixture(scope="module")
def grays(oracle, clip_small):
    return [oracle.analysis_gray(f, 480, 360) for f in clip_small]


# ---- W1 warpAffine -----------------------------------------------------------
MATS = [
    [1, 0, 0, 0, 1, 0],
    [1, 0, 5, 0, 1, -3],
    [0.99995, -0.01, 3.25, 0.01, 0.99995, -7.5],
    [0.9986, 0.0523, -14.2, -0.0523, 0.9986, 9.9],      # 3 degrees
    [0.7071, -0.7071, 80.0, 0.7071, 0.7071, -40.0],       # 45 degrees: LDS bbox overflow -> direct path
    [1.5, 0.0, -20.0, 0.0, 1.5, 10.0],                    # zoom in
    [0.5, 0.0, 30.0, 0.0, 0.5, 20.0],                     # zoom out (wide source footprint)
    [1, 0, 1000.0, 0, 1, 0],                              # everything out of frame
]


@pytest.mark.parametrize("M", MATS)
def test_warp_bgr_bit_exact(gpu, oracle, clip_small, M):
    img = clip_small[0]
    assert np.array_equal(gpu.warp_affine(img, M), oracle.warp_affine(img, M))


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (17, 129), (240, 321), (33, 130)])
def test_warp_ragged_sizes(gpu, oracle, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    h, w = shape
    M = [0.9998, 0.02, 1.3, -0.02, 0.9998, -0.6]
    for cn in (1, 3):
        img = rng.integers(0, 256, (h, w) if cn == 1 else (h, w, 3), dtype=np.uint8)
        assert np.array_equal(gpu.warp_affine(img, M), oracle.warp_affine(img, M)), (shape, cn)


def test_warp_batch_matches_single(gpu, oracle, clip_small):
    imgs = np.stack(clip_small[:5])
    Ms = np.array([[np.cos(a), -np.sin(a), dx, np.sin(a), np.cos(a), dy]
                   for a, dx, dy in [(0.0, 0, 0), (0.002, 1.5, -2), (-0.004, -3, 4), (0.01, 7, 7), (0.0, -9.25, 0.5)]],
                  np.float32)
    out = gpu.warp_affine(imgs, Ms)
    for i in range(5):
        assert np.array_equal(out[i], oracle.warp_affine(imgs[i], Ms[i])), i


def test_warp_multi_frame_launch_every_matrix_class(gpu, oracle, clip_small):
    """Launches of four and more frames take the persistent kernel (coordinate tables, double-buffered tiles, prefetch):
    every matrix class of the single-frame test in ONE launch, so interior tiles, tiles that leave the image on every
    side, boxes that do not fit the staging area and frames that are entirely out of view alternate inside the
    workgroups' runs of tiles."""
    imgs = np.stack([clip_small[i % len(clip_small)] for i in range(len(MATS))])
    out = gpu.warp_affine(imgs, np.array(MATS, np.float32))
    for i, M in enumerate(MATS):
        assert np.array_equal(out[i], oracle.warp_affine(imgs[i], M)), i


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (17, 129), (240, 321), (33, 130), (270, 482), (16, 128), (48, 515)])
def test_warp_multi_frame_launch_ragged_sizes(gpu, oracle, shape):
    """Widths that are no multiple of 4 put groups of staged pixels across the right image edge; heights that are no
    multiple of 16 leave partial tiles."""
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    h, w = shape
    Ms = np.array([[0.9998, 0.02, 1.3, -0.02, 0.9998, -0.6], [1, 0, 0, 0, 1, 0], [1, 0, -2.5, 0, 1, 3.75],
                   [0.9999, -0.012, -4.0, 0.012, 0.9999, 2.0], [1.0, 0.0, 6.0, 0.0, 1.0, -5.0]], np.float32)
    imgs = rng.integers(0, 256, (len(Ms), h, w, 3), dtype=np.uint8)
    out = gpu.warp_affine(imgs, Ms)
    for i in range(len(Ms)):
        assert np.array_equal(out[i], oracle.warp_affine(imgs[i], Ms[i])), (shape, i)


@pytest.mark.parametrize("cn", [1, 2])
def test_warp_plane_kernel_every_matrix_class(gpu, oracle, clip_small, cn):
    """One- and two-channel planes (NV12: Y, interleaved UV) in launches of four and more frames take the plane kernel
    (128 x 64 / 128 x 32 tiles, source box staged as bytes): every matrix class in ONE launch - interior tiles, tiles that
    leave the image on every side, boxes that do not fit the staging area (direct path), frames entirely out of view."""
    imgs = np.stack([clip_small[i % len(clip_small)] for i in range(len(MATS))])
    planes = np.ascontiguousarray(imgs[..., 1] if cn == 1 else imgs[..., :2])
    out = gpu.warp_affine(planes, np.array(MATS, np.float32))
    for i, M in enumerate(MATS):
        assert np.array_equal(out[i], oracle.warp_affine(planes[i], M)), i


@pytest.mark.parametrize("cn", [1, 2])
@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (17, 129), (240, 322), (33, 130), (270, 482), (64, 128), (65, 132), (96, 516), (200, 1024)])
def test_warp_plane_kernel_ragged_sizes(gpu, oracle, shape, cn):
    """Partial tiles in both directions, widths that leave the 4-pixel stores and the 16-byte staging chunks hanging over the
    right edge, planes smaller than a tile."""
    rng = np.random.default_rng(shape[0] * 1000 + shape[1] + cn)
    h, w = shape
    Ms = np.array([[0.9998, 0.02, 1.3, -0.02, 0.9998, -0.6], [1, 0, 0, 0, 1, 0], [1, 0, -2.5, 0, 1, 3.75],
                   [0.9999, -0.012, -4.0, 0.012, 0.9999, 2.0], [1.0, 0.0, 6.0, 0.0, 1.0, -5.0], [0.998, 0.06, 0.5, -0.06, 0.998, 0.25]], np.float32)
    planes = rng.integers(0, 256, (len(Ms), h, w) if cn == 1 else (len(Ms), h, w, 2), dtype=np.uint8)
    out = gpu.warp_affine(planes, Ms, batch=True)
    for i in range(len(Ms)):
        assert np.array_equal(out[i], oracle.warp_affine(planes[i], Ms[i])), (shape, i)


def test_warp_plane_kernel_4k_luma(gpu, oracle):
    """Four 3840x2160 luma planes with the small rotations of a stabilizer, checked in full."""
    rng = np.random.default_rng(11)
    world = synth.make_world(synth.SEED_CONFIG2, 1920, 1080)
    base = synth.render_frame(world, 1920, 1080, (300 * 256, 280 * 256, 90))[..., 1]
    big = np.ascontiguousarray(np.kron(base, np.ones((2, 2), np.uint8)))
    planes = np.stack([np.roll(big, 3 * b, axis=1) for b in range(4)])
    Ms = np.array([[np.cos(a), -np.sin(a), dx, np.sin(a), np.cos(a), dy]
                   for a, dx, dy in zip(rng.normal(0, 0.004, 4), rng.normal(0, 8, 4), rng.normal(0, 8, 4))], np.float32)
    out = gpu.warp_affine(planes, Ms)
    for i in range(4):
        assert np.array_equal(out[i], oracle.warp_affine(planes[i], Ms[i], threads=8)), i


def test_warp_multi_frame_launch_full_hd(gpu, oracle):
    """Six 1920x1080 frames (6120 tiles: runs of several tiles per workgroup) with the small rotations of a
    stabilizer, checked in full."""
    world = synth.make_world(synth.SEED_CONFIG2, 1920, 1080)
    rng = np.random.default_rng(5)
    imgs = np.stack([synth.render_frame(world, 1920, 1080, ((300 + 3 * b) * 256, (280 + 2 * b) * 256, 90 + 5 * b)) for b in range(6)])
    Ms = np.array([[np.cos(a), -np.sin(a), dx, np.sin(a), np.cos(a), dy]
                   for a, dx, dy in zip(rng.normal(0, 0.004, 6), rng.normal(0, 6, 6), rng.normal(0, 6, 6))], np.float32)
    out = gpu.warp_affine(imgs, Ms)
    for i in range(6):
        assert np.array_equal(out[i], oracle.warp_affine(imgs[i], Ms[i], threads=8)), i


def test_warp_full_hd_properties(gpu, oracle):
    """BASELINE config 2 size: identity = exact copy; integer shift = exact shift;
    a checksum of the rotated frame equals the oracle's."""
    world = synth.make_world(synth.SEED_CONFIG2, 1920, 1080)
    img = synth.render_frame(world, 1920, 1080, (300 * 256, 280 * 256, 90))
    assert np.array_equal(gpu.warp_affine(img, [1, 0, 0, 0, 1, 0]), img)
    out = gpu.warp_affine(img, [1, 0, 16, 0, 1, 9])
    assert np.array_equal(out[9:, 16:], img[:-9, :-16]) and out[:9].max() == 0 and out[:, :16].max() == 0
    M = [0.999998, -0.002, 2.75, 0.002, 0.999998, -1.25]
    assert np.array_equal(gpu.warp_affine(img, M), oracle.warp_affine(img, M, threads=8))


def test_warp_table_scratch_grows_between_launches_on_one_stream(gpu, oracle):
    """The standalone operator keeps ONE grow-only block of coordinate tables per stream (k_warp.hip op_tabs): a 32-frame launch,
    then a launch with larger frames (the block is replaced behind a stream sync), then the small geometry again - every frame
    equal to the oracle's, i.e. no launch ever read tables of another geometry or of a freed block.  (Round 2 saw zeroed table
    records in in-flight builds that took this scratch from the stream-ordered pool; the cause could not be settled from what
    was kept - DESIGN section 8 - and this is the regression test for the path as it is built now.)"""
    rng = np.random.default_rng(940)

    def mats(n):
        a = rng.normal(0, 0.004, n)
        return np.array([[np.cos(t), -np.sin(t), dx, np.sin(t), np.cos(t), dy] for t, dx, dy in zip(a, rng.normal(0, 4, n), rng.normal(0, 4, n))], np.float32)
    for (h, w, n) in ((120, 200, 32), (270, 482, 32), (120, 200, 7), (300, 644, 33)):
        imgs = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        Ms = mats(n)
        out = gpu.warp_affine(imgs, Ms)
        for i in range(n):
            assert np.array_equal(out[i], oracle.warp_affine(imgs[i], Ms[i])), (h, w, i)


def test_warp_nv12(gpu, oracle, clip_small):
    nv = synth.bgr_to_nv12(clip_small[0])
    M = [0.99998, -0.006, 2.5, 0.006, 0.99998, -3.0]
    assert np.array_equal(gpu.warp_affine_nv12(nv, 320, 240, M), oracle.warp_affine_nv12(nv, 320, 240, M))


@pytest.mark.parametrize("shape", [(2, 2), (6, 4), (18, 130), (240, 322), (34, 130), (270, 482), (64, 128), (66, 132), (96, 516), (130, 1024), (128, 256)])
def test_warp_nv12_surfaces_in_one_launch(gpu, oracle, shape):
    """Four and more NV12 surfaces per call: luma and chroma tiles of all of them in ONE grid (warp_nv12_kernel; the frames'
    table blocks hold the luma table followed by the chroma table).  Sizes around the tile edges of both planes (128 x 64 luma,
    128 x 32 chroma pixels), planes smaller than a tile, and every matrix class: interior tiles, tiles leaving the surface on
    every side, boxes that do not fit the staging area (direct path), a surface entirely out of view."""
    h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    surf = rng.integers(0, 256, (len(MATS), h * 3 // 2, w), dtype=np.uint8)
    Ms = np.array(MATS, np.float32)
    out = gpu.warp_affine_nv12(surf, w, h, Ms)
    for i in range(len(MATS)):
        assert np.array_equal(out[i], oracle.warp_affine_nv12(surf[i], w, h, Ms[i])), (shape, i)


def test_warp_nv12_33_surfaces_span_two_launches(gpu, oracle):
    """33 surfaces: a launch of 32 and one of a single surface (plane by plane, no tables)."""
    w, h = 322, 242 - 2
    rng = np.random.default_rng(77)
    surf = rng.integers(0, 256, (33, h * 3 // 2, w), dtype=np.uint8)
    ang = rng.normal(0, 0.004, 33)
    Ms = np.array([[np.cos(a), -np.sin(a), dx, np.sin(a), np.cos(a), dy] for a, dx, dy in zip(ang, rng.normal(0, 5, 33), rng.normal(0, 5, 33))], np.float32)
    out = gpu.warp_affine_nv12(surf, w, h, Ms)
    for i in range(33):
        assert np.array_equal(out[i], oracle.warp_affine_nv12(surf[i], w, h, Ms[i])), i


# ---- G1/G2 resize + gray --------------------------------------------------------
@pytest.mark.parametrize("src,dst", [((240, 320), (480, 360)),      # upscale (config 1 regime)
                                     ((240, 320), (160, 120)),      # exact 2x
                                     ((240, 320), (80, 60)),        # 4x (first-frame regime)
                                     ((241, 323), (100, 77))])      # odd sizes
def test_resize_gray_bgr(gpu, oracle, src, dst):
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (src[0], src[1], 3), dtype=np.uint8)
    dw, dh = dst
    assert np.array_equal(gpu.resize_gray(img, dw, dh), oracle.analysis_gray(img, dw, dh))


def test_resize_gray_single_channel(gpu, oracle):
    rng = np.random.default_rng(12)
    g = rng.integers(0, 256, (270, 480), dtype=np.uint8)
    assert np.array_equal(gpu.resize_gray(g, 960, 540), oracle.resize(g, 960, 540))   # G2: 480x270 -> 960x540
    assert np.array_equal(gpu.resize_gray(g, 240, 135), oracle.resize(g, 240, 135))


def test_resize_gray_full_hd(gpu, oracle):
    world = synth.make_world(synth.SEED_CONFIG2, 1920, 1080)
    img = synth.render_frame(world, 1920, 1080, (256 * 256, 256 * 256, 0))
    assert np.array_equal(gpu.resize_gray(img, 960, 540), oracle.analysis_gray(img, 960, 540))
    assert np.array_equal(gpu.resize_gray(img, 480, 270), oracle.analysis_gray(img, 480, 270))


# ---- pyramid + derivatives ------------------------------------------------------
@pytest.mark.parametrize("shape", [(360, 480), (135, 241), (7, 9), (2, 2)])
def test_pyr_down_and_scharr(gpu, oracle, shape):
    rng = np.random.default_rng(shape[0])
    g = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(gpu.pyr_down(g), oracle.pyr_down(g))
    assert np.array_equal(gpu.scharr(g), oracle.scharr(g))


@pytest.mark.parametrize("shape", [(540, 960), (270, 480), (135, 240), (19, 25), (75, 100), (1, 7), (3, 1), (97, 131), (16, 64), (17, 65), (33, 129)])
def test_pyramid_level_kernel(gpu, oracle, shape):
    """One launch per pyramid level (batch mode): derivatives and the next level from one staged tile - tiles inside the
    image (dword loads), on every border (REFLECT_101 while staging), images smaller than a tile, odd sizes."""
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    g = rng.integers(0, 256, shape, dtype=np.uint8)
    der, nxt = gpu.pyr_level(g)
    assert np.array_equal(der, oracle.scharr(g))
    assert np.array_equal(nxt, oracle.pyr_down(g))
    der, nxt = gpu.pyr_level(g, down=False)
    assert nxt is None and np.array_equal(der, oracle.scharr(g))


# ---- F1 goodFeaturesToTrack -----------------------------------------------------
@pytest.mark.parametrize("args", [(200, 0.02, 15.0, 3), (200, 0.01, 30.0, 3), (50, 0.05, 8.0, 5), (400, 0.01, 5.0, 3),
                                  (30, 0.01, 0.0, 3)])
def test_gftt_exact_list_and_order(gpu, oracle, grays, args):
    g = grays[0]
    pts_o, _ = oracle.gftt(g, *args)
    pts_g, eig_g = gpu.gftt(g, *args, want_eig=True)
    assert np.array_equal(eig_g.view(np.uint32), oracle.min_eigen(g, args[3]).view(np.uint32))   # float map bit-exact
    assert pts_g.shape == pts_o.shape and np.array_equal(pts_g, pts_o)


def test_gftt_flat_image_returns_nothing(gpu, oracle):
    g = np.full((120, 160), 90, np.uint8)
    assert len(gpu.gftt(g, 100, 0.01, 10.0, 3)) == 0
    assert len(oracle.gftt(g, 100, 0.01, 10.0, 3)[0]) == 0


def test_gftt_many_candidates_chunked_sort(gpu, oracle):
    """> 8192 local maxima: exercises the chunked selection path of the sort kernel."""
    rng = np.random.default_rng(99)
    g = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    pts_o, nc = oracle.gftt(g, 3000, 0.0001, 3.0, 3)
    assert nc > 8192
    pts_g = gpu.gftt(g, 3000, 0.0001, 3.0, 3)
    assert np.array_equal(pts_g, pts_o)


# ---- L1 pyramidal LK ---------------------------------------------------------------
@pytest.mark.parametrize("win,levels,iters,eps", [(15, 2, 20, 0.03), (21, 2, 20, 0.03), (21, 3, 30, 0.01)])
def test_pyr_lk_exact(gpu, oracle, grays, win, levels, iters, eps):
    g0, g1 = grays[0], grays[1]
    pts, _ = oracle.gftt(g0, 200, 0.02, 15.0, 3)
    # add points that leave the image / sit on flat areas to exercise status=0 paths
    extra = np.array([[0.5, 0.5], [479.0, 359.0], [-30.0, 10.0], [240.3, 180.7], [600.0, 20.0]], np.float32)
    pts = np.vstack([pts, extra])
    no, so, eo = oracle.pyr_lk(g0, g1, pts, win, levels, iters, eps)
    ng, sg, eg = gpu.pyr_lk(g0, g1, pts, win, levels, iters, eps)
    assert np.array_equal(sg, so)
    assert np.array_equal(ng.view(np.uint32), no.view(np.uint32))     # positions bit-exact
    assert np.array_equal(eg.view(np.uint32), eo.view(np.uint32))


@pytest.mark.parametrize("shift", [(23, 0), (-17, 29), (40, -35), (3, 60)])
def test_pyr_lk_large_motion_restages_the_search_region(gpu, oracle, grays, shift):
    """Motion of more than LK_MARGIN (6) pixels at a pyramid level: the tracker's search region is staged again
    around the point (the oracle samples the image directly); points near the border take the reflected padding."""
    g0 = grays[0]
    dx, dy = shift
    g1 = np.roll(np.roll(g0, dy, axis=0), dx, axis=1)               # wraps at the borders: some tracks fail, same on both sides
    pts, _ = oracle.gftt(g0, 150, 0.02, 12.0, 3)
    h, w = g0.shape
    extra = np.array([[2.0, 3.0], [w - 3.0, h - 2.0], [w - 1.0, 5.5], [7.25, h - 1.0]], np.float32)
    pts = np.vstack([pts, extra])
    for win, levels in ((21, 2), (15, 3), (31, 1)):
        no, so, eo = oracle.pyr_lk(g0, g1, pts, win, levels, 20, 0.03)
        ng, sg, eg = gpu.pyr_lk(g0, g1, pts, win, levels, 20, 0.03)
        assert np.array_equal(sg, so)
        assert np.array_equal(ng.view(np.uint32), no.view(np.uint32))
        assert np.array_equal(eg.view(np.uint32), eo.view(np.uint32))
    moved = np.linalg.norm(no - pts, axis=1)[so.astype(bool)]
    assert moved.size and np.median(moved) > 6                       # the case is what it claims to be


def test_pyr_lk_points_whose_fourth_bilinear_weight_is_minus_one(gpu, oracle, grays):
    """The window weights are round(.. * 2^14) for three corners and the remainder for the fourth, which is -1 for about
    one sub-pixel offset in 80 000 (three round-ups): the tracker's 24-bit multiplies must be the signed ones.  (Round-1
    crash record gpu6.log, 142 of 143 tracked: an unsigned 24-bit multiply took the -1 for 2^24 - 1 in one iteration of
    one point.)  Every point here starts on such an offset, so its template patch is built with w11 = -1."""
    g0, g1 = grays[0], grays[1]
    f = np.float32
    rng = np.random.default_rng(7)
    pts = []
    while len(pts) < 64:
        x = (rng.integers(40, g0.shape[1] - 40, 200000).astype(f) + (rng.random(200000) * 0.02).astype(f)).astype(f)
        y = (rng.integers(40, g0.shape[0] - 40, 200000).astype(f) + (rng.random(200000) * 0.02).astype(f)).astype(f)
        a, b = x - np.floor(x), y - np.floor(y)          # halfWin is an integer: the window's offset is the point's
        w00 = np.rint((f(1) - a) * (f(1) - b) * f(16384)); w01 = np.rint(a * (f(1) - b) * f(16384)); w10 = np.rint((f(1) - a) * b * f(16384))
        for i in np.nonzero(16384 - w00 - w01 - w10 < 0)[0]:
            pts.append((x[i], y[i]))
    pts = np.array(pts[:64], f)
    no, so, eo = oracle.pyr_lk(g0, g1, pts, 15, 2, 20, 0.03)
    ng, sg, eg = gpu.pyr_lk(g0, g1, pts, 15, 2, 20, 0.03)
    assert np.array_equal(sg, so) and so.sum() > 8
    assert np.array_equal(ng.view(np.uint32), no.view(np.uint32))
    assert np.array_equal(eg.view(np.uint32), eo.view(np.uint32))


# ---- R1 RANSAC ------------------------------------------------------------------------
def _correspondences(seed, n, n_out, noise=0.2):
    rng = np.random.default_rng(seed)
    src = np.stack([rng.integers(5, 950, n), rng.integers(5, 530, n)], 1).astype(np.float32)
    ang = rng.uniform(-0.02, 0.02)
    a, b = np.cos(ang), np.sin(ang)
    tx, ty = rng.uniform(-8, 8, 2)
    dst = np.stack([a * src[:, 0] - b * src[:, 1] + tx, b * src[:, 0] + a * src[:, 1] + ty], 1)
    dst += rng.normal(0, noise, dst.shape)
    idx = rng.choice(n, n_out, replace=False)
    dst[idx] += rng.uniform(10, 60, (n_out, 2)) * rng.choice([-1, 1], (n_out, 2))
    return src, dst.astype(np.float32)


@pytest.mark.parametrize("n,n_out", [(200, 0), (200, 60), (200, 150), (57, 20), (4, 0), (3, 0), (2, 0), (400, 390)])
def test_ransac_exact(gpu, oracle, n, n_out):
    src, dst = _correspondences(n * 7 + n_out, n, n_out)
    oko, mo, io, fo = oracle.estimate_affine_partial2d(src, dst)
    okg, mg, ig, fg = gpu.estimate_affine_partial2d(src, dst)
    assert okg == oko
    assert np.array_equal(fg, fo)                     # ok, kept hypothesis, iterations run, inlier count
    assert np.array_equal(ig, io)                     # inlier mask
    if oko:
        assert np.array_equal(mg.view(np.uint64), mo.view(np.uint64))   # refined model, double, bit-exact


def test_ransac_too_few_points(gpu, oracle):
    src = np.array([[3, 4]], np.float32)
    okg, mg, ig, fg = gpu.estimate_affine_partial2d(src, src)
    assert okg == 0 and np.all(np.isnan(mg)) and fg[1] == -1
