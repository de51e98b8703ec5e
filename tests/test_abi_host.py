"""CPU: the C-ABI library loads, exports every symbol include/mi355ntt.h declares, its host-only
parameter helpers agree with the oracle, and argument errors are reported (no GPU compute here)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import params as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi355ntt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355ntt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(native):
    syms = declared_symbols()
    assert len(syms) >= 30
    raw = ctypes.CDLL(native.LIB_PATH)
    missing = [s for s in syms if not hasattr(raw, s)]
    assert not missing, missing
    # and the Python binding covers the same set
    assert sorted(native._SIGNATURES) == syms


def test_library_is_the_in_tree_hip_build(native):
    assert os.path.dirname(native.LIB_PATH) == os.path.join(ROOT, "ntt-cuda_amd")
    assert b"gfx950" in native.lib().mi355ntt_version()


def test_host_helpers_match_oracle(native, oracle):
    L = oracle.lib()
    moduli = [v[0] for v in P.REF_PARAMS.values()] + P.Q60 + P.Q55 + [v[0] for v in P.EDGE_PRIMES.values()] + [P.GAMMA61]
    rng = np.random.default_rng(3)
    for q in moduli:
        k = native.bit_length(q)
        assert k == L.orc_bit_length(q) == int(q).bit_length()       # demo.cu:69 formula agrees with the exact length
        assert native.barrett_mu(q, k) == L.orc_mu(q, k)
        for a in [2, 3, q - 1] + [int(x) % q for x in rng.integers(1, 1 << 62, 5, dtype=np.uint64)]:
            e = int(rng.integers(0, 1 << 62))
            assert native.modpow128(a, e, q) == pow(a, e, q) == L.orc_modpow(a, e, q)
            assert native.lib().mi355ntt_mulmod(a, e, q) == (a * e) % q
    for bits in (1, 5, 12, 15, 16):
        for a in (0, 1, 5, (1 << bits) - 1):
            assert native.bitReverse(a, bits) == L.orc_bitrev(a, bits)


@pytest.mark.parametrize("n", sorted(P.REF_PARAMS))
def test_get_params_and_tables(native, oracle, n):
    q, psi, psiinv, ninv, qbit = P.REF_PARAMS[n]
    assert native.getParams(n) == (q, psi, psiinv, ninv, qbit)
    assert native.modinv128(psi, q) == psiinv and native.modinv128(n, q) == ninv
    tp, ti = native.fillTablePsi128(psi, q, psiinv, n)
    prm = oracle.Params(n, [q], [psi])
    assert np.array_equal(tp, prm.psi_tabs[0]) and np.array_equal(ti, prm.psiinv_tabs[0])


def test_tables_60bit_digest(native):
    (n, q, psi), (d_tab, _, _) = [(k, v) for k, v in P.GOLDEN_DIGESTS.items() if k[1] == P.Q60[0]][0]
    tp, _ = native.fillTablePsi128(psi, q, native.modinv128(psi, q), n)
    assert P.digest(tp) == d_tab


def test_argument_errors_are_reported(native):
    L = native.lib()
    h = ctypes.c_void_p()
    q = (ctypes.c_ulonglong * 1)(P.Q60[0])
    psi = (ctypes.c_ulonglong * 1)(P.PSI60[0])
    # unsupported ring degree (the reference silently launches nothing, ntt_60bit.cuh:344-347)
    assert L.mi355ntt_ctx_create(ctypes.byref(h), 1000, 1, q, psi, 0) == native.EUNSUPPORTED
    assert L.mi355ntt_ctx_create(ctypes.byref(h), 1024, 1, q, psi, 0) == native.EUNSUPPORTED
    assert L.mi355ntt_ctx_create(ctypes.byref(h), 32768, 17, q, psi, 0) == native.EUNSUPPORTED
    assert L.mi355ntt_ctx_create(ctypes.byref(h), 32768, 0, q, psi, 0) == native.EUNSUPPORTED
    assert L.mi355ntt_ctx_create(None, 32768, 1, q, psi, 0) == native.EINVAL
    # psi that is not a primitive 2n-th root, even modulus, modulus too wide
    bad = (ctypes.c_ulonglong * 1)(P.PSI60[0] + 1)
    assert L.mi355ntt_ctx_create(ctypes.byref(h), 32768, 1, q, bad, 0) == native.EPARAM
    even = (ctypes.c_ulonglong * 1)(P.Q60[0] + 1)
    assert L.mi355ntt_ctx_create(ctypes.byref(h), 32768, 1, even, psi, 0) == native.EUNSUPPORTED
    wide = (ctypes.c_ulonglong * 1)((1 << 63) + 1)
    assert L.mi355ntt_ctx_create(ctypes.byref(h), 32768, 1, wide, psi, 0) == native.EUNSUPPORTED
    assert h.value is None
    # null handles / pointers
    assert L.mi355ntt_forward_batch(None, None, 1, 1, None) == native.EINVAL
    assert L.mi355ntt_forward_raw(None, 4096, None, P.Q60[0], 1, 60, None) == native.EINVAL
    assert L.mi355ntt_ctx_destroy(None) == native.OK
    assert native.lib().mi355ntt_strerror(native.EPARAM).startswith(b"inconsistent")


def test_no_gpu_means_loud_failure_not_fallback(native):
    """The product path has no CPU fallback: without a device the context constructor raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.NTTError) as e:
        native.NTTContext(32768, P.Q60[:1], P.PSI60[:1])
    assert e.value.code == native.EHIP


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under ntt-cuda_amd/ or include/ may reference it."""
    bad = []
    for base in ("ntt-cuda_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dirpath:
                continue
            for f in files:
                if f.endswith((".py", ".cpp", ".hpp", ".hip", ".cuh", ".h", "Makefile")):
                    t = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"oracle_py|liboracle|ntt_oracle|orc_", t):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_kernel_sources_carry_no_experiment_switches():
    """the lab left the product.  The kernel sources hold no conditional compilation on an MI355NTT_* macro (rounds
    1-5 carried ~55 of them: ablations that produced wrong results, in-kernel stamps, alternative code paths); the tuning VALUES live in
    ONE struct, csrc/tune.hpp, which is also the only place a measurement build can hook into (MI355NTT_TUNE_HEADER)."""
    csrc = os.path.join(ROOT, "ntt-cuda_amd", "csrc")
    for f in ("ntt_core.cuh", "kernels_fast_impl.cuh", "kernels_lit.cuh", "kernels_lat.cuh", "modarith.cuh"):
        t = open(os.path.join(csrc, f)).read()
        cond = [l for l in t.splitlines() if re.match(r"\s*#\s*(if|ifdef|ifndef|elif)\b", l)]
        assert not cond, (f, cond[:5])
        assert not re.search(r"MI355NTT_(ABLATE|STAMP|TOUCH|LAB|POLY_SLOT|ONLY_HL4N)", t), f
    tune = open(os.path.join(csrc, "tune.hpp")).read()
    assert tune.count("#ifdef") == 1 and "MI355NTT_TUNE_HEADER" in tune and "struct Tune" in tune


def test_stray_defines_do_not_change_a_kernel(tmp_path):
    """A -DMI355NTT_... left in CXXFLAGS can no longer ship a different kernel: the same translation unit compiled with and without a
    handful of the former switches yields the same gfx950 assembly."""
    src = tmp_path / "probe.hip"
    src.write_text('#include "kernels_fast_impl.cuh"\nnamespace mi355ntt {\n'
                   'template __global__ void k_forward<11, 4, true>(u64*, const TwPair*, const PrimeDev*, unsigned, unsigned, unsigned);\n}\n')
    inc = ["-I", os.path.join(ROOT, "ntt-cuda_amd", "csrc"), "-I", os.path.join(ROOT, "include")]
    base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only"] + inc
    outs = []
    for k, flags in enumerate(([], ["-DMI355NTT_ABLATE_EXCHANGE", "-DMI355NTT_STAMPS=1", "-DMI355NTT_PRIO_R1=2", "-DMI355NTT_MAD_CHAIN=0", "-DMI355NTT_LAB"])):
        out = tmp_path / ("k%d.s" % k)
        r = subprocess.run(base + flags + [str(src), "-o", str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in open(out).read().splitlines() if not l.lstrip().startswith((";", ".ident", ".file")) and "__hip_cuid" not in l])
    assert outs[0] == outs[1]


@pytest.fixture(scope="module")
def kernel_compiles(tmp_path_factory):
    """ONE compilation per translation unit of the throughput kernels (n = 2^11 .. 2^16), all six at once (about two minutes on 8
    cores): compiler remarks on stdout of tools/kernel_resources.py for the scratch check of every size, and the gfx950 assembly of
    the n = 2^15 unit for the VALU-ceiling drift check."""
    tool = os.path.join(ROOT, "tools", "kernel_resources.py")
    asm = str(tmp_path_factory.mktemp("n15") / "n15.s")
    procs = {}
    for tag in ("11", "12", "13", "14", "15", "15e", "16"):
        src = os.path.join(ROOT, "ntt-cuda_amd", "csrc", "kernels_fast_n%s.hip" % tag)
        # (n = 2^16, beyond the reference's dispatch: its forward kernels are checked; k_inverse15_split<4, false, true> -- general 60-bit
        # primes with the pointwise factor -- keeps 12 bytes, listed by the tool, a known leftover)
        cmd = [sys.executable, tool, src, "", "--require-no-scratch", "k_forward15" if tag == "16" else "k_"] + (["--asm-out", asm] if tag == "15" else [])
        procs[tag] = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    out = {}
    for tag, p in procs.items():
        text, _ = p.communicate(timeout=1800)
        out[tag] = (p.returncode, text)
    return out, asm


@pytest.fixture(scope="module")
def n15_compile(kernel_compiles):
    return kernel_compiles


def test_throughput_kernels_use_no_scratch(kernel_compiles):
    """EVERY instantiation of the throughput kernels -- persistent, fused-product and small-batch kernels of n = 2^11 .. 2^15 in every
    headroom class (6, 5, 4, 3, 2) x near / general prime, and the n = 2^16 split / pair kernels -- fits its VGPR budget without scratch
    memory (compiler remarks; tools/kernel_resources.py).  Round 2 shipped the general-prime inverse and fused kernels of n = 2^15 with
    28-104 bytes of scratch per lane, round 3 k_inverse<13|14, 4, false> with 12."""
    out, _ = kernel_compiles
    want = {"11": 80, "12": 80, "13": 80, "14": 80, "15": 80}      # 8 kernels x (classes 6, 5, 4, 3, 2 near-2^k + 6, 4, 3, 2 general) + the 3 + 5 of class 0 (kernels_lit.cuh: single-pass + small-batch)
    for tag, (rc, text) in out.items():
        rows = [l for l in text.splitlines() if "VGPRs" in l]
        assert rc == 0, (tag, text[-3000:])
        if tag in want:
            assert len(rows) == want[tag], (tag, len(rows))
        elif tag == "15e":       # the fused products with an epilogue: the classes that would spill with it are not instantiated (kernels_epi.cuh)
            assert len(rows) >= 6 and all("scratch    0 B" in l for l in rows if "_epi" in l), text[-2000:]
        else:
            assert len(rows) >= 18, (tag, len(rows))
            spilling = [l.split()[0] + l.split()[1] + l.split()[2] for l in rows if "scratch    0 B" not in l]
            assert all(x.startswith("k_inverse15_split<4,false,true>") for x in spilling), spilling


def test_valu_ceiling_profile_matches_shipped_kernels(n15_compile):
    """profiles/valu_ceiling_r05.json (the VALU ceiling bench.py prints beside the HBM roofline) is recomputed from the shipped
    sources: instruction counts of the three polynomial loops exactly, issue cycles to 1e-6.  A kernel change without
    `python3 tools/valu_ceiling.py` in the same commit fails here (round 3 shipped a stale k_polymul15 entry)."""
    _, asm = n15_compile
    assert os.path.exists(asm) and os.path.getsize(asm) > 1 << 20
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_ceiling.py"), "--asm", asm, "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]


def test_traffic_profile_belongs_to_the_shipped_kernels(native):
    """bench.py prints roofline.traffic from profiles/traffic_rNN.json -- a committed figure, not one collected in the run.  The file
    records the instruction streams of the kernels it was taken on (tools/codeobj_digest.py over the library that ran); a change of
    k_forward15 / k_inverse15 / k_polymul15 or their class-0 counterparts without new FETCH_SIZE / WRITE_SIZE passes
    (tools/profile_traffic.sh) fails here, as the VALU-ceiling profile does above."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import codeobj_digest
    import bench
    path = bench.newest_profile("traffic_r%02d.json")
    prof = json.load(open(path))
    assert prof.get("commit") and isinstance(prof.get("kernel_digest"), dict) and len(prof["kernel_digest"]) >= 5, path
    now = codeobj_digest.named_digests(os.path.join(ROOT, "ntt-cuda_amd", "build", "kernels_fast_n15.hip.o"))
    stale = [k for k, h in prof["kernel_digest"].items() if now.get(k) != h]
    assert not stale, ("the shipped kernels differ from the ones %s was taken on: re-run tools/profile_traffic.sh" % os.path.relpath(path, ROOT), stale)
    for k in ("k_forward15", "k_inverse15", "k_forward15_lit", "k_inverse15_lit"):
        assert 1.0 <= prof[k]["ratio"] < 1.06, (k, prof[k]["ratio"])      # measured traffic within 6 % of the algorithmic bytes


def test_host_objects_under_asan_ubsan(native, tmp_path):
    """The host side of the C ABI (capi.cpp, hostparams.cpp, bfv_host.cpp, shard.cpp) compiled by g++ with AddressSanitizer and
    UndefinedBehaviorSanitizer and linked in front of libmi355ntt.so (which supplies the kernel launchers): tests/cpp/host_sanitize.cpp
    walks the host-only helpers, every argument check and the creation-failure unwinding (no GPU here: mi355ntt_ctx_create fails
    with MI355NTT_EHIP after building its host state).  No sanitizer report, exit status 0.  (Sanitizers on the CPU build only: GPU
    sanitizers are not available on this pool.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-build sanitizer test (a sanitized process next to the GPU runtime is not what it is for)")
    csrc = os.path.join(ROOT, "ntt-cuda_amd", "csrc")
    flags = ["-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
             "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include")]
    objs = []
    procs = []
    for f in ("capi", "hostparams", "bfv_host", "shard"):
        o = str(tmp_path / (f + ".o"))
        objs.append(o)
        procs.append(subprocess.Popen(["g++"] + flags + ["-c", os.path.join(csrc, f + ".cpp"), "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]
    exe = str(tmp_path / "host_sanitize")
    libdir = os.path.join(ROOT, "ntt-cuda_amd")
    r = subprocess.run(["g++"] + flags + [os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp")] + objs +
                       ["-L", libdir, "-lmi355ntt", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "all checks passed" in r.stdout, (r.stdout + r.stderr)[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-4000:]
