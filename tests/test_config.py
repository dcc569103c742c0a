"""Config layer: the YAML of the reference's example mains and the key -> parameter mapping
(examples/config.yaml; examples/vs.cpp:50-168; examples/vsg.cpp:1007-1112) through the C ABI vs_config_*.
Host only: runs without a GPU.  The conversions checked here are those of cv::FileNode `>>` as restated in
csrc/config.cpp (OpenCV is absent from this image, so these are known-answer tests of the restated rules)."""
import os
import time

import pytest

from vsamd import capi

# Written for this test with the key names of examples/config.yaml; values differ from every default so that each
# mapping shows.
CONFIG = """%YAML:1.0
video_source: "rtsp://192.168.144.119:554"   # trailing comment

mode:
  width:  1280
  height:  720
  optimize_fps:  1
  use_cuda:  1
  tracker_enabled: 0
  enhancer_enabled:              1
  roll_correction_enabled:       0
  stabilizer_enabled:            1

enhancer:
  brightness:  1.5
  contrast:  1.1

  enable_white_balance:  1
  wb_strength:  0.1
  enable_vibrance:  0
  vibrance_strength:  0.02
  enable_unsharp:  1
  sharpness:  2.0
  blur_sigma: 1.25
  enable_denoise:  0
  denoise_strength:  7
  gamma:  1.2
  enable_clahe:  true
  clahe_clip_limit:  3.5
  clahe_tile_grid_size:  4
  use_cuda: true               # booleans are ints to the reader

roll_correction:
  scale_factor:  0.5
  canny_threshold_low:  40
  canny_threshold_high:  120
  canny_aperture:  3
  hough_rho:  2.0
  hough_theta: 0.0174533
  hough_threshold:  80
  angle_smoothing_alpha:  0.2
  angle_decay:  0.98
  angle_filter_min:  -70.0
  angle_filter_max:  70.0

stabilizer:
  smoothing_radius: 15            # comment
  border_type: "reflect_101"
  fadeDuration: 20
  fadeAlpha: 0.9
  border_size: 30
  crop_n_zoom: false
  logging: true
  use_cuda: true
  max_corners: 300
  quality_level: 0.02
  min_distance: 10.0
  block_size: 5
  smoothing_method: "gaussian"
  gaussian_sigma: 15.0
  adaptive_smoothing: true
  min_smoothing_radius: 10
  max_smoothing_radius: 35
  horizon_lock: true
  enable_virtual_canvas: false
  drone_high_freq_mode: true
  hf_shake_px: 0.8
  hf_analysis_max_width: 640
  hf_rot_lp_alpha: 0.1
  enable_conditional_clahe: false
  hf_dead_zone_threshold: 3.0
  hf_freeze_duration: 30
  hf_motion_accumulator_decay: 0.85
  roi: [192, 108, 1536, 864]
  'quoted key': 'it''s'

deepstream_tracker:
  model_engine: "/opt/engines/best.engine"
"""


def test_scalars_follow_the_filestorage_rules(vs):
    c = capi.Config(vs, text=CONFIG)
    assert c.kind("video_source") == "string" and c.get_string("video_source") == "rtsp://192.168.144.119:554"
    assert c.kind("mode") == "map" and c.size("mode") == 8
    assert c.kind("mode.width") == "int" and c.get_int("mode.width") == 1280
    assert c.kind("enhancer.use_cuda") == "int" and c.get_int("enhancer.use_cuda") == 1       # true -> 1
    assert c.kind("stabilizer.crop_n_zoom") == "int" and c.get_bool("stabilizer.crop_n_zoom") is False
    assert c.kind("enhancer.brightness") == "real" and c.get_double("enhancer.brightness") == 1.5
    assert c.kind("roll_correction.canny_threshold_low") == "int" and c.get_double("roll_correction.canny_threshold_low") == 40.0
    assert c.get_float("roll_correction.hough_theta") == pytest.approx(0.0174533, rel=1e-7)
    assert c.kind("stabilizer.roi") == "seq" and c.get_seq("stabilizer.roi") == [192.0, 108.0, 1536.0, 864.0]
    assert c.get_string("stabilizer.quoted key") == "it's"
    assert c.get_string("deepstream_tracker.model_engine") == "/opt/engines/best.engine"


def test_conversions_of_absent_and_mismatched_nodes(vs):
    """`node["k"] >> v`: absent -> 0 / 0.0 / "", wrong kind -> INT_MAX / DBL_MAX / "", real -> int rounds half to even."""
    c = capi.Config(vs, text="a: 2.5\nb: 3.5\nc: -0.5\ns: hello\nn: 7\nm:\n  x: 1\nip: 192.168.1.1\nneg: -12\nhexa: 0x10\nsci: 1e3\ne:\n")
    assert c.get_int("a") == 2 and c.get_int("b") == 4 and c.get_int("c") == 0
    assert c.get_int("missing") == 0 and c.get_double("missing") == 0.0 and c.get_string("missing") == ""
    assert c.kind("missing") == "none" and c.kind("m.y") == "none" and c.kind("s.t") == "none" and c.kind("e") == "none"
    assert c.get_int("s") == 2**31 - 1 and c.get_double("s") == 1.7976931348623157e308
    assert c.get_float("s") == pytest.approx(3.4028234663852886e38)
    assert c.get_bool("s") is True                    # what `>> bool` makes of a string
    assert c.get_string("n") == "" and c.get_string("m") == ""
    assert c.get_int("m") == 2**31 - 1
    assert c.kind("ip") == "string" and c.get_string("ip") == "192.168.1.1"
    assert c.get_int("neg") == -12 and c.get_int("hexa") == 16
    assert c.kind("sci") == "real" and c.get_double("sci") == 1000.0


def test_sections_map_onto_the_parameter_structs(vs):
    c = capi.Config(vs, text=CONFIG)
    p, present = c.stab_params()
    assert present
    assert (p.smoothing_radius, p.border_type, p.border_size, p.crop_n_zoom, p.logging) == (15, 2, 30, 0, 1)
    assert (p.max_corners, p.quality_level, p.min_distance, p.block_size) == (300, 0.02, 10.0, 5)
    assert (p.smoothing_method, p.gaussian_sigma) == (1, 15.0)
    assert (p.adaptive_smoothing, p.min_smoothing_radius, p.max_smoothing_radius, p.horizon_lock) == (1, 10, 35, 1)
    assert (p.fade_duration, p.enable_virtual_canvas, p.drone_high_freq_mode) == (20, 0, 1)
    assert p.fade_alpha == pytest.approx(0.9) and p.hf_shake_px == pytest.approx(0.8)
    assert (p.hf_analysis_max_width, p.enable_conditional_clahe, p.hf_freeze_duration) == (640, 0, 30)
    assert p.hf_rot_lp_alpha == pytest.approx(0.1) and p.hf_dead_zone_threshold == 3.0
    assert p.hf_motion_accumulator_decay == pytest.approx(0.85)
    assert p.lk_win_size == vs.params().lk_win_size                      # not a config key: untouched

    r, present = c.roll_params()
    assert present
    assert (r.scale_factor, r.canny_threshold_low, r.canny_threshold_high, r.canny_aperture) == (0.5, 40.0, 120.0, 3)
    assert (r.hough_rho, r.hough_threshold) == (2.0, 80)
    assert r.hough_theta == pytest.approx(0.0174533, rel=1e-6)
    assert (r.angle_smoothing_alpha, r.angle_decay, r.angle_filter_min, r.angle_filter_max) == (0.2, 0.98, -70.0, 70.0)
    assert r.max_angle_change_deg == vs.roll_params().max_angle_change_deg

    e, present = c.enh_params()
    assert present
    assert (e.brightness, e.enable_white_balance, e.enable_vibrance, e.enable_unsharp) == (1.5, 1, 0, 1)
    assert e.contrast == pytest.approx(1.1) and e.wb_strength == pytest.approx(0.1)
    assert (e.sharpness, e.blur_sigma, e.enable_denoise, e.denoise_strength) == (2.0, 1.25, 0, 7.0)
    assert (e.enable_clahe, e.clahe_clip_limit, e.clahe_tile_grid_size, e.use_cuda) == (1, 3.5, 4, 1)
    assert e.gamma == pytest.approx(1.2)


def test_absent_section_and_absent_keys(vs):
    c = capi.Config(vs, text="stabilizer:\n  smoothing_radius: 9\n  border_type: \"wrap\"\n  smoothing_method: \"gausian\"\n")
    base = vs.params()
    p, present = c.stab_params()
    assert present and p.smoothing_radius == 9 and p.border_type == 4
    assert p.smoothing_method == 0                                # an unknown method is the box filter
    assert (p.max_corners, p.block_size, p.quality_level) == (base.max_corners, base.block_size, base.quality_level)
    # to the letter of `node["k"] >> field`: what the file does not name becomes zero
    z, _ = c.stab_params(zero_missing=True)
    assert z.smoothing_radius == 9 and (z.max_corners, z.block_size, z.quality_level, z.fade_alpha) == (0, 0, 0.0, 0.0)
    assert z.lk_win_size == base.lk_win_size
    r, present = c.roll_params()
    assert not present and r.hough_threshold == vs.roll_params().hough_threshold
    e, present = c.enh_params(zero_missing=True)
    assert not present and e.gamma == 1.0
    # crop-and-zoom forces the black border (Stabilizer.cpp:67-71)
    c2 = capi.Config(vs, text="stabilizer:\n  border_type: reflect\n  crop_n_zoom: 1\n")
    assert c2.stab_params()[0].border_type == 0


@pytest.mark.parametrize("text,needle", [
    ("a: 1\n\tb: 2\n", "line 2"),
    ("a: 1\n  b: 2\n", "line 2"),
    ("a: \"open\n", "line 1"),
    ("just a scalar\n", "line 1"),
    ("a: 1\na: 2\n", "duplicate"),
    ("a: {x: 1}\n", "flow maps"),
    ("a: [1, [2]]\n", "nested"),
])
def test_malformed_text_is_rejected_with_the_line(vs, text, needle):
    with pytest.raises(capi.VsError) as e:
        capi.Config(vs, text=text)
    assert needle in str(e.value)


def test_open_and_hot_reload_by_mtime(vs, tmp_path):
    path = tmp_path / "config.yaml"
    with pytest.raises(capi.VsError):
        capi.Config(vs, path=str(path))
    with pytest.raises(capi.VsError):
        capi.Config.mtime(vs, str(path))
    path.write_text(CONFIG)
    c = capi.Config(vs, path=str(path))
    assert c.stab_params()[0].smoothing_radius == 15
    seen = capi.Config.mtime(vs, str(path))
    assert seen == int(os.stat(path).st_mtime)
    # the loop of the mains: reload when st_mtime differs from the one remembered (vs.cpp:381-394)
    path.write_text(CONFIG.replace("smoothing_radius: 15", "smoothing_radius: 40"))
    os.utime(path, (time.time() + 5, seen + 5))
    now = capi.Config.mtime(vs, str(path))
    assert now != seen
    assert capi.Config(vs, path=str(path)).stab_params()[0].smoothing_radius == 40


def test_empty_and_comment_only_documents(vs):
    for text in ("", "%YAML:1.0\n---\n# nothing\n", "\n\n"):
        c = capi.Config(vs, text=text)
        assert c.kind("stabilizer") == "none" and c.stab_params()[1] is False
    c = capi.Config(vs, text="seq:\n  - 1\n  - 2.5\n  - x\nempty: []\nat_key_level:\n- 3\n- 4\n")
    assert c.get_seq("seq")[:2] == [1.0, 2.5] and c.size("seq") == 3 and c.size("empty") == 0 and c.kind("empty") == "seq"
    assert c.get_seq("at_key_level") == [3.0, 4.0]


def test_cpp_face_compiles_and_reads_like_the_mains(vs, tmp_path):
    """include/video/Config.h (vs::ConfigFile, vs::read, vs::loadConfig, vs::ConfigWatcher) against the same text."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "video-stab_amd", "csrc")
    exe = os.path.join(root, "tests", "cpp", "_build", "config_smoke")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(root, "tests", "mock_opencv"),
                           "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "config_smoke.cpp"),
                           "-L" + csrc, "-lvideo-stab", "-Wl,-rpath," + csrc, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
                           "-o", exe])
    path = tmp_path / "config.yaml"
    path.write_text(CONFIG)
    r = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "config ok" in r.stdout, r.stdout + r.stderr


def test_example_config_and_bench_option(vs):
    """The shipped example parses and gives the library defaults; bench.py takes a config file for its parameters."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = capi.Config(vs, path=os.path.join(root, "video-stab_amd", "config.example.yaml"))
    p, present = c.stab_params()
    d = vs.params()
    assert present and (p.smoothing_radius, p.max_corners, p.min_distance, p.block_size) == \
        (d.smoothing_radius, d.max_corners, d.min_distance, d.block_size)
    assert c.roll_params()[0].hough_threshold == vs.roll_params().hough_threshold and c.get_int("mode.stabilizer_enabled") == 1
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    q = bench.make_params(vs, os.path.join(root, "video-stab_amd", "config.example.yaml"))
    assert (q.max_corners, q.lk_win_size, q.smoothing_radius) == (200, 21, 30)
