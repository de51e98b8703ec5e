"""The header's thread-safety contract (include/mi355ntt.h:18-19) from compiled C++: eight std::threads on shared contexts
(tests/cpp/threads_test.cpp) -- persistent and small-batch kernels at n = 2^15, the class-0 context of the reference's own
decryption_test.cu moduli (one launch, two passes per workgroup), n = 2^16 pair launches (pair_acquire) and the raw API's LRU under eviction;
every result against the CPU oracle (linked into the test program: test infrastructure)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "threads_test.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "threads_test")


def build(native):
    deps = [SRC, os.path.join(ROOT, "include", "mi355ntt.h"), os.path.join(ROOT, "oracle", "ntt_oracle.h")]
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(p) for p in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-pthread", "-x", "hip", "--offload-arch=gfx950", SRC, "-x", "none",
                               "-L", os.path.join(ROOT, "ntt-cuda_amd"), "-lmi355ntt", "-L", os.path.join(ROOT, "oracle"), "-loracle",
                               "-Wl,-rpath," + os.path.join(ROOT, "ntt-cuda_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-o", EXE])
    return EXE


def test_threads_program_builds(native):
    """CPU: the program compiles and links against the C ABI (and the oracle, its checker)."""
    assert os.path.exists(build(native))


@pytest.mark.gpu
def test_eight_host_threads_share_the_contexts(native, gpu):
    r = subprocess.run([build(native), "8", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "errors = 0" in r.stdout and "FAILED" not in r.stdout, r.stdout
