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


def _host_libm_outside_the_verified_set():
    """vs_libm.h restates the float code glibc ships from 2.28 up to (at least) 2.35 as x86-64 hosts with FMA and aarch64 hosts
    execute it.  Other hosts may legitimately answer differently in a last place (glibc >= 2.41: correctly rounded atan2f;
    x86 without FMA: the unfused sincosf variant): there this module's host checks are skipped with the reason, and the
    pipeline parity tests are unaffected because the oracle evaluates the same frozen definition (oracle/vso_internal.h)."""
    try:
        libc = C.CDLL(None)
        libc.gnu_get_libc_version.restype = C.c_char_p
        ver = tuple(int(x) for x in libc.gnu_get_libc_version().decode().split(".")[:2])
    except Exception:       # noqa: BLE001 - not glibc
        return "the host libc is not glibc"
    if not ((2, 28) <= ver <= (2, 40)):
        return "glibc %d.%d is outside the range vs_libm.h was verified against (2.28 .. 2.40; verified on 2.35)" % ver
    import platform
    if platform.machine() in ("x86_64", "AMD64"):
        try:
            flags = open("/proc/cpuinfo").read()
        except OSError:
            flags = ""
        if " fma" not in flags:
            return "x86-64 host without FMA: glibc selects the unfused sincosf variant"
    return None


def test_host_build_equals_the_host_libm_on_every_float():
    why = _host_libm_outside_the_verified_set()
    if why:
        pytest.skip(why)
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
    why = _host_libm_outside_the_verified_set()
    if why:
        pytest.skip(why)
    threads = min(os.cpu_count() or 1, 32)
    got = C.c_uint64(0)
    gpu.check(gpu.lib.vs_op_libm_checksum(fn, 0, 1 << 32, C.byref(got)))
    assert got.value == oracle.lib.vso_libm_checksum(fn, 0, 1 << 32, threads), name
    # and a range of its own where a single wrong value cannot hide behind a colliding sum: |x| <= 0.25 rad, positive floats
    gpu.check(gpu.lib.vs_op_libm_checksum(fn, 0, 0x3E800000, C.byref(got)))
    assert got.value == oracle.lib.vso_libm_checksum(fn, 0, 0x3E800000, threads), name


@pytest.mark.gpu
def test_device_atan2f_equals_the_host_libm(gpu, oracle):
    why = _host_libm_outside_the_verified_set()
    if why:
        pytest.skip(why)
    threads = min(os.cpu_count() or 1, 32)
    got = C.c_uint64(0)
    for start, count in ((0, 1 << 30), (1 << 40, 1 << 28)):
        gpu.check(gpu.lib.vs_op_libm_checksum(3, start, count, C.byref(got)))
        assert got.value == oracle.lib.vso_libm_checksum(3, start, count, threads), (start, count)
