"""The pin harness must stay buildable although this image cannot build it.

oracle/opencv_pin/cv_pin.cpp and tests/test_opencv_pin.py are the only route from "parity unpinned" to a pinned oracle: they
compare every restated primitive (and, with REF=..., the reference's own Stabilizer.cpp end to end) with a real OpenCV 4.11.
No OpenCV exists here or on the GPU box, so neither has ever run.  These tests keep them from rotting: the C++ harness is
parsed (g++ -fsyntax-only) against a DECLARATIONS-ONLY header of the OpenCV calls it makes
(tests/mock_opencv/pin_decls/opencv2/opencv.hpp - signatures, no bodies; it cannot build or emulate anything) together with
this repository's own vso.h and include/video/Stabilizer.h, and the Python twin is byte-compiled.

This pins nothing and earns no parity credit."""
import os
import py_compile
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "opencv_pin", "cv_pin.cpp")
DECLS = os.path.join(ROOT, "tests", "mock_opencv", "pin_decls")


@pytest.mark.parametrize("defines", [["-DVS_WITH_OPENCV_ORACLE"], ["-DVS_WITH_OPENCV_ORACLE", "-DVS_PIN_REFERENCE"]],
                         ids=["primitives", "primitives+reference"])
def test_cpp_harness_parses_against_the_opencv_signatures(defines):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra"] + defines + [
        "-I", DECLS, "-I", os.path.join(ROOT, "oracle"), "-I", os.path.join(ROOT, "include"), HARNESS]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "error" not in r.stderr


def test_cpp_harness_is_an_empty_program_without_its_switch():
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", HARNESS], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_every_cv_call_of_the_harness_is_declared():
    """The parse above proves it for the calls the compiler sees; this keeps the declarations file honest the other way round -
    it declares nothing the harness does not name (no emulation creeping in)."""
    import re
    src = open(HARNESS).read()
    decl = open(os.path.join(DECLS, "opencv2", "opencv.hpp")).read()
    used = set(re.findall(r"cv::([a-zA-Z_]\w*)\s*\(", src))
    declared = set(re.findall(r"^(?:[\w:<>,&\s\*]+?)\b([a-zA-Z_]\w*)\s*\((?:InputArray|InputOutputArray|int nthreads)", decl, re.M))
    functions = {n for n in declared if n[0].islower() and n != "noArray"}
    assert functions, "no function declarations recognised"
    assert functions <= used, sorted(functions - used)
    assert "{" not in "".join(ln for ln in decl.splitlines() if re.match(r"^(void|double|int|Mat) \w+\(", ln)), "a declared function has a body"


def test_python_harness_compiles():
    py_compile.compile(os.path.join(ROOT, "tests", "test_opencv_pin.py"), doraise=True)
