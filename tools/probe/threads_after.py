"""probe: run a set of GPU tests in this process, then tests/cpp/threads_test as a child several times"""
import os, subprocess, sys
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(ROOT)
sel = sys.argv[1:] or ["tests/test_ntt30.py"]
def child(tag, n=4):
    for i in range(n):
        r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "threads_test"), "8", "4"], capture_output=True, text=True, timeout=300)
        print(tag, i, "rc", r.returncode, "| stdout lines", len(r.stdout.splitlines()), "| last:", (r.stdout.strip().splitlines() or [""])[-1], flush=True)
        if r.returncode != 0:
            print(r.stdout[-800:]); print(r.stderr[-4000:], flush=True)
child("before")
rc = pytest.main(sel + ["-m", "gpu", "-x", "-q"])
print("pytest rc", rc, flush=True)
child("after", 6)
