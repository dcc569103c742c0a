"""AutoZoomCrop (SURVEY.md 8a row AZ, /root/reference/src/AutoZoomCrop.cpp:10-283).

CPU part (no GPU): known-answer tests of the oracle's findContours / drawContours restatement,
and the product's host logic (vs_azc_crop_from_mask: an independent implementation with
row-span fill and prefix counts) against the oracle on masks of every kind.
GPU part: the fused gray/threshold/close kernel and the whole autoZoomCrop step, bit-exact.
"""
import math

import numpy as np
import pytest
from scipy import ndimage

import roll_scene

EIGHT = np.ones((3, 3), bool)
FOUR = ndimage.generate_binary_structure(2, 1)


def rotated_frame(oracle, w, h, deg, seed=0, cn=3):
    """A textured frame rotated about its centre with black corners (what roll correction with a
    constant border hands to autoZoomCrop)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(40, 256, (h, w, 3), dtype=np.uint8)
    img[h // 3:h // 3 + 9, w // 4:w // 4 + 14] = 0          # a black object inside the content
    if cn == 1:
        img = np.ascontiguousarray(img[..., 0])
    a = math.radians(deg)
    al, be = math.cos(a), math.sin(a)
    cx, cy = w / 2.0, h / 2.0
    M = [al, be, (1 - al) * cx - be * cy, -be, al, be * cx + (1 - al) * cy]
    return oracle.warp_affine_d(img, M, border=0)


def random_masks():
    rng = np.random.default_rng(1234)
    out = []
    for k in range(6):                                        # noise of several densities: many contours, holes
        out.append((rng.random((37 + 5 * k, 53 + 3 * k)) < (0.15 + 0.14 * k)).astype(np.uint8) * 255)
    for seed in range(4):                                     # blobs
        out.append((roll_scene.noisy_gray(90, 70, seed) > 60).astype(np.uint8) * 255)
    m = np.zeros((40, 60), np.uint8); m[5:30, 10:50] = 255; m[12:20, 20:35] = 0; m[14:17, 25:28] = 255
    out.append(m)                                             # ring with an island in the hole
    m = np.zeros((30, 30), np.uint8); m[:, :] = 255
    out.append(m)                                             # everything set (touches all borders)
    m = np.zeros((30, 30), np.uint8); m[10, 3:20] = 255; m[3:25, 12] = 255
    out.append(m)                                             # one-pixel-wide cross
    m = np.zeros((20, 20), np.uint8); m[4, 4] = 255
    out.append(m)                                             # isolated pixel
    m = np.zeros((24, 24), np.uint8)
    for i in range(4, 20):
        m[i, i] = 255; m[i, 23 - i] = 255
    out.append(m)                                             # diagonal X (pure 8-connectivity)
    m = np.zeros((12, 40), np.uint8); m[2:10, 2] = 255; m[2, 2:30] = 255; m[9, 2:30] = 255; m[2:10, 29] = 255; m[5:7, 10:20] = 255
    out.append(m)                                             # thin-walled box with a blob inside (RETR_EXTERNAL quirk)
    return out


# ------------------------------------------------------------------ oracle KATs
def test_contour_of_rectangle_and_order(oracle):
    m = np.zeros((20, 30), np.uint8)
    m[4:12, 5:21] = 255
    cs = oracle.find_contours(m)
    assert len(cs) == 1
    # OpenCV's well known answer for a filled rectangle: TL, BL, BR, TR
    assert cs[0].tolist() == [[5, 4], [5, 11], [20, 11], [20, 4]]


def test_contour_special_shapes(oracle):
    m = np.zeros((9, 9), np.uint8); m[4, 4] = 255
    assert [c.tolist() for c in oracle.find_contours(m)] == [[[4, 4]]]
    m = np.zeros((9, 9), np.uint8); m[3, 2:7] = 255
    assert [c.tolist() for c in oracle.find_contours(m)] == [[[2, 3], [6, 3]]]
    m = np.zeros((9, 9), np.uint8); m[2:7, 5] = 255
    assert [c.tolist() for c in oracle.find_contours(m)] == [[[5, 2], [5, 6]]]
    m = np.zeros((9, 9), np.uint8)      # plus sign with one-pixel arms
    m[4, 2:7] = 255; m[2:7, 4] = 255
    c = oracle.find_contours(m)[0]
    # followed by hand: down the top arm, diagonally past the centre to each tip and back (8-connectivity
    # cuts the inner corners, so the centre pixel itself is never visited)
    assert c.tolist() == [[4, 2], [4, 3], [3, 4], [2, 4], [3, 4], [4, 5], [4, 6], [4, 5], [5, 4], [6, 4], [5, 4], [4, 3]]
    m = np.zeros((9, 9), np.uint8); m[0:9, 0:9] = 255    # touching every image border
    assert oracle.find_contours(m)[0].tolist() == [[0, 0], [0, 8], [8, 8], [8, 0]]


def test_contours_external_only_and_last_found_first(oracle):
    m = np.zeros((40, 60), np.uint8)
    m[5:30, 10:50] = 255
    m[12:20, 20:35] = 0                 # hole
    m[14:17, 25:28] = 255               # island inside the hole: not external
    m[33:36, 3:8] = 255                 # second external component, later in raster order
    cs = oracle.find_contours(m)
    assert len(cs) == 2
    # every finished contour goes to the head of OpenCV's list (cvInsertNodeIntoTree): the vector runs from the
    # bottom of the image to the top
    assert cs[0][0].tolist() == [3, 33] and cs[1][0].tolist() == [10, 5]


def test_external_boxes_match_the_oracle_contours_in_order(vs, oracle):
    """The product's region list for the virtual canvas (vs_op_external_boxes, host code) = cv::boundingRect of the
    oracle's contours, same order."""
    total = 0
    for m in random_masks():
        want = []
        for c in oracle.find_contours(m):
            x0, y0, x1, y1 = c[:, 0].min(), c[:, 1].min(), c[:, 0].max(), c[:, 1].max()
            want.append([x0, y0, x1 - x0 + 1, y1 - y0 + 1])
        got = vs.external_boxes(m)
        assert got.tolist() == want
        total += len(want)
    assert total > 100
    ring = np.full((50, 70), 255, np.uint8); ring[10:40, 12:60] = 0; ring[20:25, 30:40] = 255   # canvas ring + a dark blob in the frame
    assert vs.external_boxes(ring).tolist() == [[0, 0, 70, 50]]


def test_largest_contour_tie_goes_to_the_first_of_opencvs_vector(vs, oracle):
    """AutoZoomCrop.cpp:155-164 keeps the first contour of maximal size in the vector cv::findContours returned, i.e.
    the one found LAST in raster order."""
    m = np.zeros((40, 60), np.uint8)
    m[4:14, 5:25] = 255                 # two rectangles: 4 contour points each
    m[22:36, 30:55] = 255
    info = vs.azc_crop_from_mask(m)
    want = oracle.azc_crop_rect(m)
    assert info.tolist() == want.tolist()
    assert info[3] >= 22                # the crop sits in the lower rectangle


def _true_outer_fill(mask, start_xy):
    """Component of start_xy (8-connected) plus everything it encloses (4-connected outside)."""
    lab, _ = ndimage.label(mask != 0, structure=EIGHT)
    comp = lab == lab[start_xy[1], start_xy[0]]
    pad = np.pad(~comp, 1, constant_values=True)
    out_lab, _ = ndimage.label(pad, structure=FOUR)
    outside = out_lab == out_lab[0, 0]
    return (~outside[1:-1, 1:-1]), comp


def test_filled_contour_is_component_plus_holes(oracle):
    checked = 0
    for m in random_masks():
        h, w = m.shape
        lab, _ = ndimage.label(m != 0, structure=EIGHT)
        for c in oracle.find_contours(m):
            x, y = c[0]
            comp = lab == lab[y, x]
            ys, xs = np.nonzero(comp)
            first = (xs[ys == ys.min()].min(), ys.min())
            if (x, y) != first:
                continue                  # not a true outer border (see the RETR_EXTERNAL quirk test)
            expect, _ = _true_outer_fill(m, (x, y))
            got = oracle.fill_contour(c, w, h) != 0
            assert np.array_equal(got, expect)
            checked += 1
    assert checked > 50


def test_external_with_one_pixel_walls(oracle):
    # A blob inside a box whose walls are one pixel thick: the left wall keeps the left-edge mark
    # (its east neighbour is not examined on the way down), so the blob is recognised as nested.
    m = random_masks()[-1]
    cs = oracle.find_contours(m)
    assert len(cs) == 1 and cs[0].tolist() == [[2, 2], [2, 9], [29, 9], [29, 2]]


def test_every_contour_is_a_true_outer_border(oracle):
    # RETR_EXTERNAL: exactly one contour per 8-connected component that is not enclosed by another one
    for m in random_masks():
        lab, n = ndimage.label(m != 0, structure=EIGHT)
        cs = oracle.find_contours(m)
        pad = np.pad(m == 0, 1, constant_values=True)
        bl, _ = ndimage.label(pad, structure=FOUR)
        outside = (bl == bl[0, 0])[1:-1, 1:-1]
        # components touching the outside background (4-adjacent) or the image border
        near = ndimage.binary_dilation(np.pad(outside, 1, constant_values=True), structure=FOUR)[1:-1, 1:-1]
        expect = {int(l) for l in np.unique(lab[near & (m != 0)])}
        got = [int(lab[c[0][1], c[0][0]]) for c in cs]
        assert len(got) == len(set(got)) and set(got) == expect


def test_crop_rect_of_rotated_content(oracle):
    f = rotated_frame(oracle, 480, 270, 4.0)
    out, info = oracle.auto_zoom_crop(f)
    assert out.shape == (360, 640, 3) and info[7] == 1 and info[0] >= 1
    x, y, w, h = info[2:6]
    cm = oracle.content_mask(f)
    # the four edges of the rectangle found by the shrink loop are inside the content; the aspect
    # fix may then widen it, so test the height and the centre columns
    assert cm[y:y + h, x + w // 2].all()
    assert abs(w - int(h * 480 / 270)) <= 1 or x == 0 or x + w == 480
    assert h < 270 and h > 200
    # unrotated: the whole frame (minus the one-pixel quirk of Rect(min, max-min))
    f0 = rotated_frame(oracle, 480, 270, 0.0)
    _, info0 = oracle.auto_zoom_crop(f0)
    assert info0[2:6].tolist() == [0, 0, 478, 269]


def test_black_frame_falls_back(oracle):
    f = np.zeros((90, 160, 3), np.uint8)
    out, info = oracle.auto_zoom_crop(f)
    assert out.shape == f.shape and info[0] == 0 and info[7] == 0


def test_content_mask_closes_small_holes(oracle):
    g = np.full((40, 60), 200, np.uint8)
    g[10:12, 10:13] = 0                 # 2x3 hole: closed by the 5x5 ellipse
    g[25:35, 30:45] = 1                 # large dark area at the threshold value: stays open
    g[0, 0] = 0                         # corner pixel: outside pixels are ignored, so it closes
    m = oracle.content_mask(g)
    assert m[10:12, 10:13].all() and m[0, 0] == 255
    assert not m[28:32, 34:41].any()
    assert set(np.unique(m)) == {0, 255}


# ------------------------------------------------------------------ product host logic vs oracle (CPU)
def _oracle_filled(oracle, m):
    cs = oracle.find_contours(m)
    if not cs:
        return np.zeros_like(m)
    best = max(range(len(cs)), key=lambda i: (len(cs[i]), -i))
    return oracle.fill_contour(cs[best], m.shape[1], m.shape[0])


def test_host_crop_logic_matches_oracle(oracle, vs):
    masks = random_masks()
    rng = np.random.default_rng(7)
    for deg in (0.0, 1.5, -3.0, 8.0, 30.0):
        masks.append(oracle.content_mask(rotated_frame(oracle, 320, 180, deg, seed=int(deg * 10) % 7)))
    masks.append(np.zeros((16, 16), np.uint8))
    for _ in range(40):
        h, w = rng.integers(3, 60, 2)
        masks.append((rng.random((h, w)) < rng.random()).astype(np.uint8) * 255)
    for w in (63, 64, 65, 128, 129, 200):                    # rows of several 64-pixel words, noise and blobs
        masks.append((rng.random((40, w)) < 0.55).astype(np.uint8) * 255)
        blobs = np.kron((rng.random((10, (w + 7) // 8)) < 0.6).astype(np.uint8), np.ones((6, 8), np.uint8))[:, :w] * 255
        blobs[rng.random(blobs.shape) < 0.03] ^= 255
        masks.append(blobs)
    masks.append(np.full((5, 192), 255, np.uint8))
    for w, bw in ((300, 40), (517, 70), (700, 130)):        # long straight edges: the word-at-a-time stretch of the follower
        blobs = np.kron((rng.random((6, (w + bw - 1) // bw)) < 0.6).astype(np.uint8), np.ones((12, bw), np.uint8))[:, :w] * 255
        masks.append(blobs.copy())
        blobs[rng.random(blobs.shape) < 0.004] ^= 255
        masks.append(blobs)
        masks.append(np.pad(np.full((20, w - 2), 255, np.uint8), 1))
    masks.append(np.full((40, 400), 255, np.uint8))
    for m in masks:
        ref = oracle.azc_crop_rect(m)
        got, filled = vs.azc_crop_from_mask(m, want_filled=True)
        assert ref.tolist() == got.tolist()
        assert np.array_equal(filled, _oracle_filled(oracle, m))


def test_host_crop_logic_full_hd(oracle, vs):
    m = oracle.content_mask(rotated_frame(oracle, 1920, 1080, 2.0, cn=1))
    ref = oracle.azc_crop_rect(m)
    got = vs.azc_crop_from_mask(m)
    assert ref.tolist() == got.tolist() and ref[7] == 1 and ref[6] > 10


# ------------------------------------------------------------------ HIP parity (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("shape,cn", [((270, 480), 3), ((101, 67), 3), ((64, 64), 1), ((5, 7), 3), ((1080, 1920), 3),
                                      ((33, 70), 1), ((40, 1030), 3), ((9, 1025), 1), ((120, 129), 3), ((57, 64), 3)])
def test_content_mask_bit_exact(gpu, oracle, shape, cn):
    rng = np.random.default_rng(shape[0] * 7 + cn)
    h, w = shape
    # mostly dark values around the threshold so that the mask is busy, plus bright patches
    img = rng.integers(0, 4, (h, w, 3), dtype=np.uint8)
    img[rng.random((h, w)) < 0.6] = 0
    img[h // 4:h // 2, w // 4:w // 2] = 180
    if cn == 1:
        img = np.ascontiguousarray(img[..., 1])
    ref = oracle.content_mask(img)
    assert h < 32 or 0 < int((ref != 0).sum()) < ref.size
    assert np.array_equal(ref, gpu.content_mask(img))


@pytest.mark.gpu
@pytest.mark.parametrize("size,deg,cn", [((480, 270), 4.0, 3), ((1280, 720), -2.5, 3), ((1920, 1080), 1.0, 3),
                                         ((640, 360), 10.0, 1), ((333, 201), 3.0, 3)])
def test_auto_zoom_crop_matches_oracle(gpu, oracle, size, deg, cn):
    w, h = size
    f = rotated_frame(oracle, w, h, deg, seed=w, cn=cn)
    ref, info = oracle.auto_zoom_crop(f)
    az = gpu.auto_zoom_crop()
    got = az.apply(f)
    assert info.tolist() == az.info().tolist() and info[7] == 1
    assert ref.shape == got.shape and np.array_equal(ref, got)


@pytest.mark.gpu
def test_auto_zoom_crop_fallback_and_device_entry(gpu, oracle):
    from vsamd.capi import DevBuf
    az = gpu.auto_zoom_crop()
    black = np.zeros((90, 700, 3), np.uint8)
    assert np.array_equal(az.apply(black), black) and az.info()[7] == 0
    f = rotated_frame(oracle, 800, 450, -5.0, seed=3)
    ref, info = oracle.auto_zoom_crop(f)
    d_in, d_out = DevBuf.from_array(gpu, f), DevBuf(gpu, 800 * 3 * 450)
    ow, oh = az.apply_dev(d_in.ptr, 800, 450, 800 * 3, 3, d_out.ptr, 800 * 3)
    az.sync()
    assert (ow, oh) == (640, 360)
    got = d_out.download((450, 800, 3), np.uint8).reshape(-1)[:360 * 800 * 3].reshape(360, 800, 3)[:, :640]
    assert np.array_equal(ref, got)


@pytest.mark.gpu
def test_roll_then_zoom_chain(gpu, oracle):
    """examples/roll-correction-file.cpp:60-66: autoCorrectRoll followed by autoZoomCrop."""
    ro, rg, az = oracle.roll_correction(), gpu.roll_correction(), gpu.auto_zoom_crop()
    for i in range(4):
        f = roll_scene.horizon_frame(640, 360, 60, seed=i)
        a, b = ro.correct(f), rg.correct(f)
        assert np.array_equal(a, b)
        ref, _ = oracle.auto_zoom_crop(a)
        assert np.array_equal(ref, az.apply(b))


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(800, 450), (808, 454)], ids=["rows_of_16_pixel_groups", "ragged_rows"])
def test_auto_zoom_crop_nv12_async_matches_oracle(gpu, oracle, size):
    """vs_azc_apply_nv12_dev: the mask kernels of a frame are queued by the call, the contour logic runs on worker threads (four
    frames at a time), which queue the crop-and-scale of both planes.  Twelve surfaces - rotated content in black corners at
    several angles and sizes of the black object, an all-black one (no contour: the surface comes back unchanged), an all-content
    one - pushed without waiting; tickets, rectangles and planes against the oracle."""
    from vsamd.capi import DevBuf
    from vsamd import synth
    w, h = size              # (a width of 16-pixel groups takes the mask's wide-load kernel, the other one the general kernel)
    frames = [rotated_frame(oracle, w, h, deg, seed=i) for i, deg in enumerate([4.0, -2.5, 1.0, 7.5, -6.0, 0.5, 3.0, -1.0, 2.0, -4.5])]
    frames.insert(3, np.zeros((h, w, 3), np.uint8))
    frames.insert(8, np.full((h, w, 3), 200, np.uint8))
    surfs = [synth.bgr_to_nv12(f) for f in frames]
    for s, f in zip(surfs, frames):
        s[:h][(f == 0).all(axis=2)] = 0              # (BT.601 black is 16; the warp's black border in an NV12 stream is 0)
    sb = w * h * 3 // 2
    op, oh_max = w, h
    ob = op * oh_max * 3 // 2
    d_in = DevBuf.from_array(gpu, np.stack(surfs))
    d_out = DevBuf(gpu, ob * len(surfs))
    az = gpu.auto_zoom_crop()
    # (the first five one by one, the rest through the array form: one trip through the binding)
    tickets = [az.apply_nv12_dev(d_in.ptr + i * sb, w, h, w, d_out.ptr + i * ob, op, op * oh_max) for i in range(5)]
    tickets += az.apply_nv12_dev_n([d_in.ptr + i * sb for i in range(5, len(surfs))], w, h, w, [d_out.ptr + i * ob for i in range(5, len(surfs))], op, op * oh_max)
    assert tickets == list(range(len(surfs)))
    az.sync()
    got = d_out.download((len(surfs), oh_max * 3 // 2, op), np.uint8)
    cropped = 0
    for i, s in enumerate(surfs):
        ref, info = oracle.auto_zoom_crop_nv12(s, w, h)
        ow, oh, ginfo = az.result(tickets[i])
        assert ginfo.tolist() == info.tolist(), i
        assert (ow, oh) == ((640, 360) if info[7] else (w, h)), i
        assert np.array_equal(got[i, :oh, :ow], ref[:oh]), i
        assert np.array_equal(got[i, oh_max:oh_max + oh // 2, :ow], ref[oh:]), i
        cropped += int(info[7])
    assert 9 <= cropped <= 11
    wt = az.worker_times()               # diagnostics: every frame went through a worker, in two batches (8 + 4)
    assert wt[0] == len(surfs) and wt[5] == 2 and wt[3] > 0 and all(v >= 0 for v in wt)
    az.close()
