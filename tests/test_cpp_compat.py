"""The reference's own smoke test (60bit_ntt_test.cu, check = 1) rebuilt on the C++ compat headers."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "ntt_test_60bit.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "ntt_test_60bit")


def build(native, oracle):
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-x", "hip", "--offload-arch=gfx950", SRC, "-x", "none",
                               "-L", os.path.join(ROOT, "ntt-cuda_amd"), "-lmi355ntt", "-L", os.path.join(ROOT, "oracle"), "-loracle",
                               "-Wl,-rpath," + os.path.join(ROOT, "ntt-cuda_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-o", EXE])
    return EXE


def test_compat_program_builds(native, oracle):
    """CPU: the compat headers compile against the C ABI and link (no GPU needed to build)."""
    assert os.path.exists(build(native, oracle))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2048, 4096, 32768])
def test_compat_program_matches_schoolbook(native, oracle, gpu, n):
    """60bit_ntt_test.cu:72-98 through the compiled drop-in program; above n = 8192 the program takes its expected product from
    the oracle's transforms instead of the O(n^2) refPolyMul128"""
    exe = build(native, oracle)
    r = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "errors = 0" in r.stdout


DEMO_SRC = os.path.join(ROOT, "tests", "cpp", "bfv_demo.cpp")
DEMO_EXE = os.path.join(ROOT, "tests", "cpp", "bfv_demo")


def build_demo(native):
    hdr = os.path.join(ROOT, "ntt-cuda_amd", "compat", "bfv_launch.hpp")
    if not os.path.exists(DEMO_EXE) or os.path.getmtime(DEMO_EXE) < max(os.path.getmtime(DEMO_SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-x", "hip", "--offload-arch=gfx950", DEMO_SRC, "-x", "none",
                               "-L", os.path.join(ROOT, "ntt-cuda_amd"), "-lmi355ntt", "-Wl,-rpath," + os.path.join(ROOT, "ntt-cuda_amd"),
                               "-o", DEMO_EXE])
    return DEMO_EXE


def test_bfv_demo_builds(native):
    """CPU: demo.cu rebuilt on compat/bfv_launch.hpp compiles and links against the C ABI alone (no oracle)."""
    assert os.path.exists(build_demo(native))


@pytest.mark.gpu
@pytest.mark.parametrize("primes", [16, 5, 3])
def test_bfv_demo_runs_like_the_reference_demo(native, gpu, primes):
    """demo.cu:275-320: keygen_rns -> encryption_rns -> decryption_rns from the keystream, decrypted == message"""
    r = subprocess.run([build_demo(native), str(primes)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Decryption is correct" in r.stdout


@pytest.mark.gpu
def test_bfv_demo_batched_drivers(native, gpu):
    """the same program with 24 ciphertexts per call through compat/bfv_launch.hpp's encryption_rns_batch / decryption_rns_batch"""
    r = subprocess.run([build_demo(native), "5", "24"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Decryption is correct" in r.stdout and "Batched decryption is correct" in r.stdout


SH_SRC = os.path.join(ROOT, "tests", "cpp", "shards_test.cpp")
SH_EXE = os.path.join(ROOT, "tests", "cpp", "shards_test")


def build_shards(native):
    hdr = os.path.join(ROOT, "include", "mi355ntt.h")
    if not os.path.exists(SH_EXE) or os.path.getmtime(SH_EXE) < max(os.path.getmtime(SH_SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-x", "hip", "--offload-arch=gfx950", SH_SRC, "-x", "none",
                               "-L", os.path.join(ROOT, "ntt-cuda_amd"), "-lmi355ntt", "-Wl,-rpath," + os.path.join(ROOT, "ntt-cuda_amd"),
                               "-o", SH_EXE])
    return SH_EXE


def test_shards_program_builds(native):
    """CPU: a C++ program against the multi-device driver and the element-wise wrappers compiles and links against the C ABI alone."""
    assert os.path.exists(build_shards(native))


@pytest.mark.gpu
@pytest.mark.parametrize("world,num", [(4, 600), (8, 37), (1, 64)])
def test_shards_program_runs(native, gpu, world, num):
    """mi355ntt_shards_scatter_transform_gather / _transform with `world` logical shards on the one GPU against the whole-batch
    call, and poly_add_device ... poly_mul_int_t against the reference's arithmetic, from compiled C++."""
    r = subprocess.run([build_shards(native), str(world), str(num)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "errors = 0" in r.stdout
