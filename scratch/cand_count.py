import sys
sys.path.insert(0, 'video-stab_amd')
from vsamd import capi, synth
vs = capi.load()
W, H = 1920, 1080
fb = W * H * 3
frames = synth.make_clip(synth.SEED_CONFIG2, W, H, 6)
d_in = capi.DevBuf(vs, fb * 6); d_out = capi.DevBuf(vs, fb)
for i, f in enumerate(frames): d_in.upload(f, i * fb)
s = vs.stabilizer(vs.params(smoothing_radius=30, max_corners=200, lk_win_size=21, lk_max_level=2))
for i in range(6):
    s.push_dev(d_in.ptr + i * fb, W, H, W * 3, 0, d_out.ptr, W * 3)
    c = s.counters()
    print(i, c.last_candidates, c.last_features, c.last_tracked, c.last_inliers, c.gftt_overflow)
