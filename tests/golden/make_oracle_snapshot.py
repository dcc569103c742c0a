"""Generates oracle_snapshot.json: SHA-256 digests of what the CPU oracle (oracle/, built by `make -C oracle`) returns for
fixed, integer-generated inputs.  NOT reference vectors - the reference cannot run here - but a snapshot of the parity
TARGET: the HIP path is tested bit for bit against the oracle, so an unintended edit of the oracle would silently move
what "parity" means.  tests/test_oracle_kat.py::test_oracle_snapshot fails when any digest changes; after a deliberate
correction of the oracle (say, from an OpenCV pin run, docs/opencv_semantics.md) rerun this script and commit the result
together with the correction.

    python tests/golden/make_oracle_snapshot.py          # from the repository root
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "video-stab_amd"))
import numpy as np          # noqa: E402
import oracle_lib           # noqa: E402
from vsamd import synth     # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    return h.hexdigest()


def snapshot(o):
    clip = synth.make_clip(synth.SEED_CONFIG1 + 41, 320, 240, 14)
    f0, f1 = clip[0], clip[1]
    g0, g1 = o.bgr2gray(f0), o.bgr2gray(f1)
    out = {}
    out["resize_linear"] = digest(o.resize(f0, 213, 131), o.resize(f0, 160, 120), o.resize(g0, 480, 270))
    out["bgr2gray"] = digest(g0)
    out["pyr_down_scharr"] = digest(o.pyr_down(g0), o.scharr(g0))
    out["min_eigen"] = digest(o.min_eigen(g0, 3))
    pts, nc = o.gftt(g0, 200, 0.02, 15.0, 3)
    out["gftt"] = digest(pts, np.int64(nc))
    nxt, st, err = o.pyr_lk(g0, g1, pts, 15, 2, 20, 0.03)
    out["pyr_lk"] = digest(nxt, st, err)
    ok, model, inl, info = o.estimate_affine_partial2d(pts[st != 0], nxt[st != 0], 5.0, 500)
    out["estimate_affine_partial2d"] = digest(np.int64(ok), model, inl, info)
    M = [0.99995, -0.01, 3.25, 0.01, 0.99995, -7.5]
    out["warp_affine"] = digest(o.warp_affine(f0, M), o.warp_affine(g0, M), o.warp_affine_d(f0, M, border=3), o.warp_affine_d(f0, M, border=1))
    out["copy_make_border"] = digest(*[o.copy_make_border(f0, 9, b) for b in range(5)])
    e = o.canny(g0, 50, 150)
    out["canny_hough"] = digest(e, o.hough_lines(e, 1.0, np.float32(np.pi / 180), 40))
    mask = (g0 > 110).astype(np.uint8) * 255
    cs = o.find_contours(mask)
    out["find_contours"] = digest(np.int64(len(cs)), *cs[:200])
    out["content_mask_crop"] = digest(o.content_mask(f0), o.azc_crop_rect(o.content_mask(f0)))
    small = np.ascontiguousarray(f0[:48, :64])
    out["enhancer_primitives"] = digest(o.gaussian_blur(f0, 1.0), o.clahe(np.ascontiguousarray(g0), 2.0, 8), o.add_weighted(f0, 1.5, f1, -0.5),
                                        o.denoise_colored(small, 5.0, 5.0))
    path = np.cumsum(np.sin(np.arange(80, dtype=np.float32) * np.float32(0.37)) * np.float32(3.0)).astype(np.float32)
    out["trajectory_filters"] = digest(o.box_filter(path, 30), o.box_filter(path, 30, True), o.gaussian_filter(path, 2.0), o.kalman_filter(path))

    def run(**kw):
        so = o.stabilizer(o.params(**kw))
        frames = [so.push(f) for f in clip]
        while True:
            f = so.flush(clip[0])
            if f is None:
                break
            frames.append(f)
        so.close()
        return digest(*[f for f in frames if f is not None])
    out["stabilizer_box"] = run(smoothing_radius=5)
    out["stabilizer_gaussian_border"] = run(smoothing_radius=6, smoothing_method=1, gaussian_sigma=2.0, border_size=8, border_type=1)
    out["stabilizer_kalman_crop"] = run(smoothing_radius=5, smoothing_method=2, border_size=10, crop_n_zoom=1)
    out["stabilizer_drone"] = run(smoothing_radius=5, drone_high_freq_mode=1)
    out["stabilizer_fade"] = run(smoothing_radius=5, border_size=8, border_type=5, fade_alpha=0.3, fade_duration=4)
    out["stabilizer_canvas"] = run(smoothing_radius=5, enable_virtual_canvas=1, adaptive_canvas_size=0, canvas_scale_factor=1.2, temporal_buffer_size=4)
    return out


if __name__ == "__main__":
    snap = snapshot(oracle_lib.load())
    json.dump(snap, open(os.path.join(HERE, "oracle_snapshot.json"), "w"), indent=1, sort_keys=True)
    print(len(snap), "digests written")
