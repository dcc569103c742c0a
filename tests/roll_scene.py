"""Synthetic tilted-horizon scenes for the roll-correction tests (integer arithmetic only)."""
import numpy as np


def horizon_frame(w, h, slope_q10, seed=1, offset=0, texture=True):
    """BGR frame: bright sky above, dark ground below a line of slope slope_q10/1024
    through the centre (+offset rows), plus deterministic block texture so Canny sees
    more than one edge."""
    rng = np.random.RandomState(seed)
    x = np.arange(w, dtype=np.int64)[None, :]
    y = np.arange(h, dtype=np.int64)[:, None]
    ground = (y - h // 2 - offset) * 1024 > (x - w // 2) * slope_q10
    f = np.empty((h, w, 3), np.uint8)
    f[..., 0] = np.where(ground, 40, 230)
    f[..., 1] = np.where(ground, 90, 200)
    f[..., 2] = np.where(ground, 60, 170)
    if texture:
        for _ in range(24):
            bw, bh = rng.randint(8, w // 6), rng.randint(8, h // 6)
            bx, by = rng.randint(0, w - bw), rng.randint(0, h - bh)
            col = rng.randint(0, 256, 3)
            f[by:by + bh, bx:bx + bw] = (f[by:by + bh, bx:bx + bw].astype(np.int32) * 3 + col) // 4
    return f


def noisy_gray(w, h, seed):
    """Smooth blobs + noise: many weak and strong Canny candidates, long hysteresis chains."""
    rng = np.random.RandomState(seed)
    g = np.zeros((h, w), np.int32)
    for _ in range(40):
        cx, cy, r = rng.randint(0, w), rng.randint(0, h), rng.randint(6, max(8, min(w, h) // 3))
        yy, xx = np.ogrid[:h, :w]
        d2 = (xx - cx) ** 2 + (yy - cy) ** 2
        g += np.where(d2 < r * r, rng.randint(10, 80), 0)
    g += rng.randint(0, 12, (h, w))
    return np.clip(g, 0, 255).astype(np.uint8)
