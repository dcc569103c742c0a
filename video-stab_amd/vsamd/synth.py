"""Deterministic synthetic clips (SURVEY.md 8d "Synthetic clip generator").

Integer-only arithmetic (numpy int64 + a Python-int xorshift64*), so the same
seed gives the same bytes on every machine: a world texture (square-wave
gratings + filled rectangles + hash noise) sampled through a per-frame
translation / small rotation with 8-bit fixed-point bilinear weights.
"""
import numpy as np

_MASK = (1 << 64) - 1

SEED_CONFIG1 = 0x5EED0001   # 640x480x300, smoothingRadius 25 (BASELINE.json configs[0])
SEED_CONFIG2 = 0x5EED0002   # 1920x1080 BGR8 (configs[1])
SEED_CONFIG3 = 0x5EED0003   # 3840x2160 NV12 (configs[2])


class XorShift64Star:
    def __init__(self, seed):
        self.s = (seed & _MASK) or 0x9E3779B97F4A7C15

    def next(self):
        x = self.s
        x ^= x >> 12
        x ^= (x << 25) & _MASK
        x ^= x >> 27
        self.s = x
        return (x * 0x2545F4914F6CDD1D) & _MASK

    def randint(self, lo, hi):
        return lo + (self.next() >> 11) % (hi - lo + 1)


def make_world(seed, width, height):
    """World texture (height+512, width+512, 3) uint8, BGR."""
    rng = XorShift64Star(seed)
    wt, ht = width + 512, height + 512
    yy, xx = np.mgrid[0:ht, 0:wt].astype(np.int64)
    img = np.full((ht, wt, 3), 96, dtype=np.int64)
    for _ in range(24):
        ux, uy = rng.randint(-8, 8), rng.randint(-8, 8)
        if ux == 0 and uy == 0:
            ux = 1
        period = rng.randint(96, 768)
        amp = [rng.randint(-22, 22) for _ in range(3)]
        wave = ((ux * xx + uy * yy) // period) & 1
        for c in range(3):
            img[:, :, c] += amp[c] * wave
    noise = (((xx * 73856093) ^ (yy * 19349663)) >> 7) & 7
    img += (noise - 3)[:, :, None]
    for _ in range(400):
        x0, y0 = rng.randint(0, wt - 9), rng.randint(0, ht - 9)
        rw, rh = rng.randint(8, 72), rng.randint(8, 72)
        col = [rng.randint(0, 255) for _ in range(3)]
        img[y0:y0 + rh, x0:x0 + rw, :] = col
    return np.clip(img, 0, 255).astype(np.uint8)


def motion_script(seed, n_frames, pan_q8=512, jitter_q8=384, rot_1e5=200, segments=None):
    """Per-frame (ox_q8, oy_q8, sin_q16) camera pose.

    pan_q8: pan in 1/256 px per frame (default 2 px/frame); jitter_q8: jitter
    std in 1/256 px (default 1.5 px); rot_1e5: rotation std in 1e-5 rad
    (default 0.002 rad).  `segments` = list of (first_frame, pan_x_q8,
    pan_y_q8, jitter_q8, rot_1e5) overrides, for clips that exercise all four
    motion-intent gains (SURVEY.md 8 "Clip design note").
    """
    rng = XorShift64Star(seed ^ 0xA5A5A5A5)
    poses = []
    px, py = 256 * 256, 256 * 256
    seg = (0, pan_q8, 0, jitter_q8, rot_1e5)
    segs = sorted(segments or [])
    for k in range(n_frames):
        while segs and segs[0][0] <= k:
            seg = segs.pop(0)
        _, panx, pany, jit, rot = seg
        if k > 0:
            px += panx
            py += pany

        def gauss(std):
            # Irwin-Hall(4) scaled: integer-only approximately normal sample
            s = sum(rng.randint(-1000, 1000) for _ in range(4))
            return (s * std) // 1155   # std of the sum is 1155
        jx, jy, ja = gauss(jit), gauss(jit), gauss(rot)
        sin_q16 = (ja * 65536) // 100000
        poses.append((px + jx, py + jy, sin_q16))
    return poses


def render_frame(world, width, height, pose):
    """Sample the world at `pose` -> (height, width, 3) uint8 BGR."""
    ox, oy, s16 = pose
    c16 = 65536 - ((s16 * s16) >> 17)
    ht, wt = world.shape[:2]
    cx, cy = width // 2, height // 2
    y, x = np.mgrid[0:height, 0:width].astype(np.int64)
    rx, ry = x - cx, y - cy
    sx = ox + cx * 256 + ((c16 * rx - s16 * ry) >> 8)
    sy = oy + cy * 256 + ((s16 * rx + c16 * ry) >> 8)
    ix, iy = sx >> 8, sy >> 8
    fx, fy = (sx & 255)[:, :, None], (sy & 255)[:, :, None]
    x0, x1 = ix % wt, (ix + 1) % wt
    y0, y1 = iy % ht, (iy + 1) % ht
    w = world.astype(np.int64)
    acc = ((256 - fx) * (256 - fy)) * w[y0, x0]
    acc += (fx * (256 - fy)) * w[y0, x1]
    acc += ((256 - fx) * fy) * w[y1, x0]
    acc += (fx * fy) * w[y1, x1]
    return ((acc + 32768) >> 16).astype(np.uint8)


def make_clip(seed, width, height, n_frames, **kw):
    world = make_world(seed, width, height)
    poses = motion_script(seed, n_frames, **kw)
    return [render_frame(world, width, height, p) for p in poses]


def bgr_to_nv12(frame):
    """BT.601 limited-range integer BGR -> NV12 (h*3/2, w) uint8; w,h even."""
    b = frame[:, :, 0].astype(np.int64)
    g = frame[:, :, 1].astype(np.int64)
    r = frame[:, :, 2].astype(np.int64)
    yp = ((66 * r + 129 * g + 25 * b + 128) >> 8) + 16
    u = ((-38 * r - 74 * g + 112 * b + 128) >> 8) + 128
    v = ((112 * r - 94 * g - 18 * b + 128) >> 8) + 128
    h, w = yp.shape

    def sub(p):
        return (p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2] + 2) >> 2
    uv = np.empty((h // 2, w), dtype=np.int64)
    uv[:, 0::2] = sub(u)
    uv[:, 1::2] = sub(v)
    return np.clip(np.vstack([yp, uv]), 0, 255).astype(np.uint8)


# ---- long clips rendered on the device (bench.py: more distinct input than the 256 MB Infinity Cache holds) -----------
def loop_script(seed, n_frames, pan_q8=512, jitter_q8=384, rot_1e5=200):
    """Per-frame camera pose of a CLOSED pan path: n/4 frames right, down, left, up at pan_q8 per frame (the same jitter
    and rotation model as motion_script), so that the clip can be played in a cycle without a cut - frame 0 follows frame
    n-1 by one camera step - and no frame is read again before all the others have been."""
    assert n_frames % 4 == 0 and n_frames >= 8
    rng = XorShift64Star(seed ^ 0xA5A5A5A5)
    q = n_frames // 4
    px, py = 256 * 256, 256 * 256
    poses = []
    for k in range(n_frames):
        if k > 0:
            side = (k - 1) // q
            px += (pan_q8, 0, -pan_q8, 0)[side]
            py += (0, pan_q8, 0, -pan_q8)[side]

        def gauss(std):
            s = sum(rng.randint(-1000, 1000) for _ in range(4))
            return (s * std) // 1155
        jx, jy, ja = gauss(jitter_q8), gauss(jitter_q8), gauss(rot_1e5)
        poses.append((px + jx, py + jy, (ja * 65536) // 100000))
    return poses


def _pose_forward_matrix(pose, width, height, scale=1.0):
    """Forward 2x3 matrix (cv::warpAffine's M) whose inverse samples the world at `pose`: frame(x, y) =
    world(R (x - cx, y - cy) + (cx + ox, cy + oy)).  scale 0.5: the same pose in the coordinates of a half-size plane."""
    ox, oy, s16 = pose
    s = s16 / 65536.0
    c = (1.0 - s * s) ** 0.5
    cx, cy = width * 0.5 * scale, height * 0.5 * scale
    tx, ty = cx + ox / 256.0 * scale, cy + oy / 256.0 * scale
    inv = np.array([[c, -s, tx - c * cx + s * cy], [s, c, ty - s * cx - c * cy], [0, 0, 1]], np.float64)
    return np.ascontiguousarray(np.linalg.inv(inv)[:2].reshape(6))


def make_clip_dev(vs, seed, width, height, n_frames, nv12=False, **kw):
    """A closed-loop clip of n_frames frames rendered ON THE DEVICE from the world texture (one upload) with the library's
    own warp operator (vs_op_warp_affine_ex); returns one DevBuf holding the packed frames (BGR8, or NV12: Y plane then
    the interleaved UV plane).  Synthetic input only - nothing is compared against these frames' provenance."""
    import ctypes as C
    from . import capi
    world = make_world(seed, width, height)
    hw, ww = world.shape[:2]
    poses = loop_script(seed, n_frames, **kw)
    f64p = C.POINTER(C.c_double)
    if not nv12:
        fb = width * height * 3
        d_world = capi.DevBuf.from_array(vs, world)
        clip = capi.DevBuf(vs, fb * n_frames)
        for i, p in enumerate(poses):
            M = _pose_forward_matrix(p, width, height)
            vs.check(vs.lib.vs_op_warp_affine_ex(d_world.ptr, ww * 3, ww, hw, clip.ptr + i * fb, width * 3, width, height, 3,
                                                 M.ctypes.data_as(f64p), capi.BORDER_BLACK, None))
        vs.sync()
        d_world.free()
        return clip
    wnv = bgr_to_nv12(world)
    d_y = capi.DevBuf.from_array(vs, wnv[:hw])
    d_uv = capi.DevBuf.from_array(vs, wnv[hw:])
    fb = width * height * 3 // 2
    clip = capi.DevBuf(vs, fb * n_frames)
    for i, p in enumerate(poses):
        M = _pose_forward_matrix(p, width, height)
        Mh = _pose_forward_matrix(p, width, height, 0.5)
        vs.check(vs.lib.vs_op_warp_affine_ex(d_y.ptr, ww, ww, hw, clip.ptr + i * fb, width, width, height, 1,
                                             M.ctypes.data_as(f64p), capi.BORDER_BLACK, None))
        vs.check(vs.lib.vs_op_warp_affine_ex(d_uv.ptr, ww, ww // 2, hw // 2, clip.ptr + i * fb + width * height, width,
                                             width // 2, height // 2, 2, Mh.ctypes.data_as(f64p), capi.BORDER_BLACK, None))
    vs.sync()
    d_y.free()
    d_uv.free()
    return clip
