"""cosf / sinf / atan2f of the product (video-stab_amd/csrc/vs_libm.h: glibc's algorithms restated for host and device) against
the host's own libm - the functions the reference's std::cos(float) / std::sin(float) / std::atan2(float, float) call
(/root/reference/src/Stabilizer.cpp:662, 902-908, 1689).

CPU: the header's host build, every float through cosf, sinf and atanf, 2^31 argument pairs through atan2f (tests/cpp/libm_check.cpp).
GPU: the device build's checksum over the same arguments against the checksum of the host libm's values (oracle/vso_libm.cpp).
Together: the matrix entries and the rotation component the device computes are the reference's own, bit for bit - which is why
the pipeline tests compare whole frames with np.array_equal."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "cpp", "_build")


def test_host_build_equals_the_host_libm_on_every_float():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "libm_check")
    src = os.path.join(ROOT, "tests", "cpp", "libm_check.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-o", exe, src])
    r = subprocess.run([exe, "all"], capture_output=True, text=True, timeout=1500)
    lines = [ln.split() for ln in r.stdout.splitlines()]
    assert r.returncode == 0, r.stdout + r.stderr
    assert [(a, c) for a, _, c in lines] == [("cosf", "0"), ("sinf", "0"), ("atanf", "0"), ("atan2f", "0")], r.stdout
    assert [b for _, b, _ in lines] == ["4294967296"] * 3 + ["2147483648"]


@pytest.mark.gpu
@pytest.mark.parametrize("fn,name", [(0, "cosf"), (1, "sinf"), (2, "atanf")])
def test_device_build_equals_the_host_libm_on_every_float(gpu, oracle, fn, name):
    threads = min(os.cpu_count() or 1, 32)
    got = C.c_uint64(0)
    gpu.check(gpu.lib.vs_op_libm_checksum(fn, 0, 1 << 32, C.byref(got)))
    assert got.value == oracle.lib.vso_libm_checksum(fn, 0, 1 << 32, threads), name
    # and a range of its own where a single wrong value cannot hide behind a colliding sum: |x| <= 0.25 rad, positive floats
    gpu.check(gpu.lib.vs_op_libm_checksum(fn, 0, 0x3E800000, C.byref(got)))
    assert got.value == oracle.lib.vso_libm_checksum(fn, 0, 0x3E800000, threads), name


@pytest.mark.gpu
def test_device_atan2f_equals_the_host_libm(gpu, oracle):
    threads = min(os.cpu_count() or 1, 32)
    got = C.c_uint64(0)
    for start, count in ((0, 1 << 30), (1 << 40, 1 << 28)):
        gpu.check(gpu.lib.vs_op_libm_checksum(3, start, count, C.byref(got)))
        assert got.value == oracle.lib.vso_libm_checksum(3, start, count, threads), (start, count)
