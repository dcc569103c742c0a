"""The source-compatible C++ class (include/video/Stabilizer.h + video-stab_amd/host/Stabilizer.cpp):
compiles against an OpenCV-shaped cv::Mat and drives the GPU library like the reference's
examples/file-capture.cpp does."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "video-stab_amd", "csrc")
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "wrapper_smoke")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "tests", "mock_opencv"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "wrapper_smoke.cpp"), os.path.join(ROOT, "video-stab_amd", "host", "Stabilizer.cpp"),
           "-L" + CSRC, "-lvideo-stab", "-Wl,-rpath," + CSRC, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)


def test_wrapper_compiles_and_links(vs):
    build()
    assert os.path.exists(EXE)


def test_wrapper_fails_loudly_without_gpu(vs):
    if vs.lib.vs_device_count() > 0:
        pytest.skip("a GPU is present")
    build()
    r = subprocess.run([EXE, "3"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_wrapper_runs_file_capture_loop(gpu):
    build()
    r = subprocess.run([EXE, "40"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    n, outs, flushed, _ = r.stdout.strip().splitlines()[-1].split()
    assert (int(n), int(outs), int(flushed)) == (40, 21, 19)     # radius 20 -> 19 warm-up empties


def _run(mode, n=40, **env):
    r = subprocess.run([EXE, str(n), mode], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    frames = [tuple(int(v) for v in ln.split()[1:]) for ln in lines if ln.startswith("F ")]
    return frames, [int(v) for v in lines[-1].split()[:3]]


@pytest.mark.gpu
@pytest.mark.parametrize("mode,size", [("reflect", (240, 320)), ("fade", (264, 344)), ("canvas", (240, 320)), ("canvas12", (240, 320))])
def test_wrapper_modes_and_the_pipelined_host_call(gpu, mode, size):
    """vs::Stabilizer through its Parameters for the border / canvas modes of round 2 (frame sizes as the reference returns
    them: the fade border pads, the canvas window has the unpadded size), and VS_STAB_HOST_PIPELINE=1: the same frames in
    the same order, one stabilize() call later (the held frame comes out of flush() first)."""
    build()
    frames, (n, outs, flushed) = _run(mode)
    assert (n, outs, flushed) == (40, 21, 19) and len(frames) == 40
    # every frame but the last one (no transform: returned as it came, unpadded) has the mode's size
    assert all(f[:2] == size for f in frames[:-1]) and frames[-1][:2] == (240, 320)
    assert len(set(f[2] for f in frames)) > 30                      # a moving picture, not one frame forty times
    piped, (n2, outs2, flushed2) = _run(mode, VS_STAB_HOST_PIPELINE="1")
    assert piped == frames and (n2, outs2, flushed2) == (40, 20, 20)


@pytest.mark.gpu
def test_wrapper_output_ring_and_the_parameters_of_this_implementation(gpu):
    """Parameters::pinHostFrames (default on: results come from a ring of page-locked Mats that is reused once the caller has let
    go of them) and Parameters::hostPipeline: the same frames as with plain allocations; results the caller keeps alive are never
    written again (every one of forty is hashed a second time at the end); the pipelined call by parameter equals the one by
    environment variable."""
    build()
    frames, counts = _run("reflect")
    assert _run("unpinned") == (frames, counts)
    assert _run("keep") == (frames, counts)
    piped, (n, outs, flushed) = _run("piped")
    assert piped == frames and (n, outs, flushed) == (40, 20, 20)


def test_timing_program_of_the_class_builds(vs):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "_build", "wrapper_time"))


ROLL_EXE = os.path.join(ROOT, "tests", "cpp", "_build", "roll_smoke")


def build_roll():
    os.makedirs(os.path.dirname(ROLL_EXE), exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "tests", "mock_opencv"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "roll_smoke.cpp"), os.path.join(ROOT, "video-stab_amd", "host", "RollCorrection.cpp"),
           "-L" + CSRC, "-lvideo-stab", "-Wl,-rpath," + CSRC, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", ROLL_EXE]
    subprocess.check_call(cmd)


def test_roll_wrapper_compiles_and_fails_loudly_without_gpu(vs):
    build_roll()
    if vs.lib.vs_device_count() > 0:
        return
    r = subprocess.run([ROLL_EXE, "2"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_roll_wrapper_runs_example_loop(gpu):
    """examples/roll-correction-file.cpp:52-70 through the C++ classes."""
    build_roll()
    r = subprocess.run([ROLL_EXE, "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    n, ow, oh, _ = r.stdout.split()
    assert (int(n), int(ow), int(oh)) == (6, 640, 360)


ENH_EXE = os.path.join(ROOT, "tests", "cpp", "_build", "enhance_smoke")


def build_enh():
    os.makedirs(os.path.dirname(ENH_EXE), exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "tests", "mock_opencv"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "enhance_smoke.cpp"), os.path.join(ROOT, "video-stab_amd", "host", "Enhancer.cpp"),
           "-L" + CSRC, "-lvideo-stab", "-Wl,-rpath," + CSRC, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", ENH_EXE]
    subprocess.check_call(cmd)


def test_enhancer_wrapper_compiles_and_fails_loudly_without_gpu(vs):
    build_enh()
    if vs.lib.vs_device_count() > 0:
        return
    r = subprocess.run([ENH_EXE], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_enhancer_wrapper_matches_oracle(gpu, oracle):
    """examples/vs.cpp:547-550 with examples/config.yaml:23-47 through the C++ class."""
    import numpy as np
    build_enh()
    w, h = 333, 201
    r = subprocess.run([ENH_EXE, str(w), str(h)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    yy, xx = np.mgrid[0:h, 0:w]
    frame = np.stack([(xx * 3 + yy * 5) & 255, (xx ^ yy) & 255, (xx * yy) & 255], 2).astype(np.uint8)
    p = oracle.enh_params(brightness=1.5, contrast=1.1, enable_unsharp=1, sharpness=2.0, blur_sigma=1.0, gamma=1.2, use_cuda=1)
    want = int(oracle.enhance(frame, p).astype(np.uint64).sum())
    assert r.stdout.split() == [str(w), str(h), str(want)]
