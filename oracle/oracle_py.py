"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(), bench.py's
cpu_baseline leg.  The product path (ntt-cuda_amd/) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64 = ctypes.c_ulonglong
u64p = ctypes.POINTER(u64)
u32p = ctypes.POINTER(ctypes.c_uint)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        L = _LIB
        L.orc_bit_length.restype = ctypes.c_uint
        L.orc_bit_length.argtypes = [u64]
        for name in ("orc_mu",):
            getattr(L, name).restype = u64
        L.orc_mu.argtypes = [u64, ctypes.c_uint]
        L.orc_mulmod.restype = u64
        L.orc_mulmod.argtypes = [u64, u64, u64]
        L.orc_modpow.restype = u64
        L.orc_modpow.argtypes = [u64, u64, u64]
        L.orc_modinv.restype = u64
        L.orc_modinv.argtypes = [u64, u64]
        L.orc_bitrev.restype = u64
        L.orc_bitrev.argtypes = [u64, ctypes.c_int]
        L.orc_barrett.restype = u64
        L.orc_barrett.argtypes = [u64, u64, u64, u64, ctypes.c_uint]
        L.orc_fill_table.restype = None
        L.orc_fill_table.argtypes = [u64, u64, u64p, ctypes.c_uint]
        L.orc_get_params.restype = ctypes.c_int
        L.orc_get_params.argtypes = [u64, u64p, u64p, u64p, u64p, u32p]
        L.orc_ct_stage.restype = None
        L.orc_ct_stage.argtypes = [u64p, ctypes.c_uint, ctypes.c_uint, u64, u64, ctypes.c_uint, u64p]
        L.orc_gs_stage.restype = None
        L.orc_gs_stage.argtypes = [u64p, ctypes.c_uint, ctypes.c_uint, u64, u64, ctypes.c_uint, u64p]
        L.orc_forward.restype = None
        L.orc_forward.argtypes = [u64p, ctypes.c_uint, u64, u64, ctypes.c_uint, u64p]
        L.orc_inverse.restype = None
        L.orc_inverse.argtypes = [u64p, ctypes.c_uint, u64, u64, ctypes.c_uint, u64p]
        for name in ("orc_forward_batch", "orc_inverse_batch"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [u64p, ctypes.c_uint, u64p, ctypes.c_uint, ctypes.c_uint, u64p, u64p, u32p, ctypes.c_int]
        L.orc_pointwise.restype = None
        L.orc_pointwise.argtypes = [u64p, u64p, ctypes.c_uint, u64, u64, ctypes.c_uint]
        L.orc_pointwise_batch.restype = None
        L.orc_pointwise_batch.argtypes = [u64p, u64p, u64p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, u64p, u64p, u32p]
        L.orc_pointwise_scalar.restype = None
        L.orc_pointwise_scalar.argtypes = [u64p, u64, ctypes.c_uint, u64, u64, ctypes.c_uint]
        L.orc_ref_polymul.restype = None
        L.orc_ref_polymul.argtypes = [u64p, u64p, u64p, u64, ctypes.c_uint]
        L.orc_bfv_decrypt.restype = ctypes.c_int
        L.orc_bfv_decrypt.argtypes = [u64p, u64p, u64p, u64p, ctypes.c_uint, ctypes.c_uint, u64, u64, u64p, u64p]
        L.orc_bfv_keygen_core.restype = ctypes.c_int
        L.orc_bfv_keygen_core.argtypes = [u64p, u64p, u64p, u64p, u64p, ctypes.c_uint, ctypes.c_uint]
        L.orc_bfv_encrypt_core.restype = ctypes.c_int
        L.orc_bfv_encrypt_core.argtypes = [u64p, u64p, u64p, u64p, u64p, u64p, ctypes.c_uint, ctypes.c_uint, u64]
        u8p = ctypes.POINTER(ctypes.c_ubyte)
        L.orc_salsa20_keystream.restype = None
        L.orc_salsa20_keystream.argtypes = [u8p, ctypes.c_ulong, u8p, u64]
        for nm in ("orc_sample_ternary_xq", "orc_sample_uniform_xq", "orc_sample_gaussian_xq"):
            getattr(L, nm).restype = None
            getattr(L, nm).argtypes = [u8p, u64p, ctypes.c_uint, ctypes.c_uint, u64p]
        for nm in ("orc30_forward", "orc30_inverse"):
            getattr(L, nm).restype = None
            getattr(L, nm).argtypes = [u32p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, u32p]
        L.orc30_pointwise.restype = None
        L.orc30_pointwise.argtypes = [u32p, u32p, ctypes.c_ulong, ctypes.c_uint, ctypes.c_uint, ctypes.c_int]
        L.orc_bfv_constants.restype = None
        L.orc_bfv_constants.argtypes = [u64p, u64p, ctypes.c_uint, u64, u64, u64p, u64p, u64p, u64p, u64p, u64p]
        L.orc_splitmix_fill.restype = None
        L.orc_splitmix_fill.argtypes = [u64p, ctypes.c_ulong, u64, u64]
    return _LIB


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _p32(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


class Params:
    """Per-prime parameters derived the reference's way (bit length, mu, psi^-1, both tables)."""

    def __init__(self, n, qs, psis, tables=True):
        L = lib()
        self.n = int(n)
        self.q = np.array(qs, dtype=np.uint64)
        self.psi = np.array(psis, dtype=np.uint64)
        P = len(self.q)
        self.k = np.array([L.orc_bit_length(int(q)) for q in self.q], dtype=np.uint32)
        self.mu = np.array([L.orc_mu(int(q), int(k)) for q, k in zip(self.q, self.k)], dtype=np.uint64)
        self.psiinv = np.array([L.orc_modinv(int(p), int(q)) for p, q in zip(self.psi, self.q)], dtype=np.uint64)
        if tables:
            self.psi_tabs = np.empty((P, self.n), dtype=np.uint64)
            self.psiinv_tabs = np.empty((P, self.n), dtype=np.uint64)
            for i in range(P):
                L.orc_fill_table(int(self.psi[i]), int(self.q[i]), _p(self.psi_tabs[i]), self.n)
                L.orc_fill_table(int(self.psiinv[i]), int(self.q[i]), _p(self.psiinv_tabs[i]), self.n)


def splitmix(count, seed, q):
    a = np.empty(int(count), dtype=np.uint64)
    lib().orc_splitmix_fill(_p(a), int(count), int(seed), int(q))
    return a


def synth_batch(n, num, qs, seed_base=1):
    """polynomial y = splitmix64(seed = seed_base + y) mod q[y % P]."""
    out = np.empty((num, n), dtype=np.uint64)
    P = len(qs)
    for y in range(num):
        lib().orc_splitmix_fill(_p(out[y]), n, seed_base + y, int(qs[y % P]))
    return out


def forward(a, prm, idx=0):
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().orc_forward(_p(a), prm.n, int(prm.q[idx]), int(prm.mu[idx]), int(prm.k[idx]), _p(prm.psi_tabs[idx]))
    return a


def inverse(a, prm, idx=0):
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().orc_inverse(_p(a), prm.n, int(prm.q[idx]), int(prm.mu[idx]), int(prm.k[idx]), _p(prm.psiinv_tabs[idx]))
    return a


def forward_batch(a, prm, division=None, threads=1):
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    num = a.size // prm.n
    division = division or len(prm.q)
    lib().orc_forward_batch(_p(a), prm.n, _p(prm.psi_tabs), num, division, _p(prm.q), _p(prm.mu), _p32(prm.k), threads)
    return a


def inverse_batch(a, prm, division=None, threads=1):
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    num = a.size // prm.n
    division = division or len(prm.q)
    lib().orc_inverse_batch(_p(a), prm.n, _p(prm.psiinv_tabs), num, division, _p(prm.q), _p(prm.mu), _p32(prm.k), threads)
    return a


def pointwise_batch(a, b, prm, division=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    c = np.empty_like(a)
    num = a.size // prm.n
    division = division or len(prm.q)
    lib().orc_pointwise_batch(_p(c), _p(a), _p(b), prm.n, num, division, _p(prm.q), _p(prm.mu), _p32(prm.k))
    return c


def pointwise_scalar(a, b, prm, idx=0):
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().orc_pointwise_scalar(_p(a), int(b), a.size, int(prm.q[idx]), int(prm.mu[idx]), int(prm.k[idx]))
    return a


# ---- the stand-alone element-wise kernels of poly_arithmetic.cuh (numpy restatements; uint64 arithmetic wraps as the reference's) ----
def poly_add(a, b, q):
    """poly_add, poly_arithmetic.cuh:144-154: ra = a[i] + b[i]; if (ra > q) ra -= q  (note `>`: a sum equal to q stays q)"""
    r = a.astype(np.uint64) + np.asarray(b, dtype=np.uint64)
    return np.where(r > np.uint64(q), r - np.uint64(q), r)


def poly_add_integer(a, b, q):
    """poly_add_integer, poly_arithmetic.cuh:156-166: the same with a scalar b"""
    return poly_add(a, np.uint64(b), q)


def poly_sub(a, b, q):
    """poly_sub, poly_arithmetic.cuh:168-179: ra = a[i]; if (ra < b[i]) ra += q; a[i] = ra  -- b is never subtracted"""
    a = a.astype(np.uint64)
    return np.where(a < b.astype(np.uint64), a + np.uint64(q), a)


def poly_negate(a, q):
    """poly_negate, poly_arithmetic.cuh:334-338: a[i] = q - a[i]; a[i] *= (a[i] != q)"""
    r = np.uint64(q) - a.astype(np.uint64)
    return np.where(r == np.uint64(q), np.uint64(0), r)


def poly_mul_int_t(a, b, t):
    """mod_t, poly_arithmetic.cuh:128-142: the low 64 bits of a[i] * b, masked with t - 1 held in a 32-bit `unsigned`"""
    with np.errstate(over="ignore"):
        lo = a.astype(np.uint64) * np.uint64(b)
    return lo & np.uint64((int(t) - 1) & 0xffffffff)


def ref_polymul(a, b, q):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    d = np.empty_like(a)
    lib().orc_ref_polymul(_p(a), _p(b), _p(d), int(q), a.size)
    return d


def bfv_decrypt(c, sk, qs, psis, n, t, gamma, want_stages=False):
    c = np.ascontiguousarray(c, dtype=np.uint64).copy()
    sk = np.ascontiguousarray(sk, dtype=np.uint64)
    qs = np.array(qs, dtype=np.uint64)
    psis = np.array(psis, dtype=np.uint64)
    r = len(qs) - 1
    out = np.empty(n, dtype=np.uint64)
    stages = np.empty((3, r * n), dtype=np.uint64) if want_stages else None
    rc = lib().orc_bfv_decrypt(_p(c), _p(sk), _p(qs), _p(psis), len(qs), n, int(t), int(gamma), _p(out),
                               _p(stages) if want_stages else None)
    assert rc == 0
    return (out, stages) if want_stages else out


def bfv_keygen_core(sk, pk, e, qs, psis, n):
    """keygen_rns after its samplers: returns (secret key in the NTT domain [R][n], public key [2][R][n])."""
    sk = np.ascontiguousarray(sk, dtype=np.uint64).copy()
    pk = np.ascontiguousarray(pk, dtype=np.uint64).copy()
    e = np.ascontiguousarray(e, dtype=np.uint64)
    qs = np.array(qs, dtype=np.uint64)
    psis = np.array(psis, dtype=np.uint64)
    assert lib().orc_bfv_keygen_core(_p(sk), _p(pk), _p(e), _p(qs), _p(psis), len(qs), n) == 0
    return sk, pk


def bfv_encrypt_core(c, pk, e, m, qs, psis, n, t):
    """encryption_rns after its samplers: c holds the ternary sample u in both halves; returns the ciphertext."""
    c = np.ascontiguousarray(c, dtype=np.uint64).copy()
    pk = np.ascontiguousarray(pk, dtype=np.uint64)
    e = np.ascontiguousarray(e, dtype=np.uint64)
    m = np.ascontiguousarray(m, dtype=np.uint64)
    qs = np.array(qs, dtype=np.uint64)
    psis = np.array(psis, dtype=np.uint64)
    assert lib().orc_bfv_encrypt_core(_p(c), _p(pk), _p(e), _p(m), _p(qs), _p(psis), len(qs), n, int(t)) == 0
    return c


def bfv_sample(qs, n, seed):
    """Test inputs shaped like the reference's samplers' outputs (NOT its Salsa20 stream): a ternary polynomial
    {0, 1, q-1} repeated per prime, small centred errors as residues, and uniform residues."""
    rng = np.random.default_rng(seed)
    R = len(qs)
    tern = rng.integers(-1, 2, size=n)
    def residues(x):
        return np.stack([np.where(x < 0, np.uint64(q) - (-x).astype(np.uint64), x.astype(np.uint64)).astype(np.uint64) for q in qs])
    def err(count):
        return [np.rint(rng.normal(0, 3.2, size=n)).astype(np.int64) for _ in range(count)]
    uniform = np.stack([rng.integers(0, q, size=n, dtype=np.uint64) for q in qs])
    return dict(ternary=residues(tern), err=lambda: residues(err(1)[0]), uniform=uniform, rng=rng)


def _p8(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))


def salsa20_keystream(nbytes, key32, nonce=0):
    """generate_random(_default) over a zeroed buffer (distributions.cuh:192-276): floor(nbytes/64) blocks"""
    out = np.zeros(nbytes // 64 * 64, dtype=np.uint8)
    key = np.frombuffer(bytes(key32), dtype=np.uint8).copy()
    lib().orc_salsa20_keystream(_p8(out), nbytes // 64, _p8(key), int(nonce))
    return out


def sample_xq(kind, rnd, n, qs):
    """ternary_dist_xq / uniform_dist_xq / gaussian_dist_xq on the byte array `rnd` (already offset as the caller wants)"""
    qs = np.array(qs, dtype=np.uint64)
    out = np.empty((len(qs), n), dtype=np.uint64)
    rnd = np.ascontiguousarray(rnd, dtype=np.uint8)
    getattr(lib(), "orc_sample_%s_xq" % kind)(_p8(rnd), _p(out), n, len(qs), _p(qs))
    return out


class Params30:
    """30-bit path (old/ntt_30bit.cuh): bit length, mu = floor(2^(2k)/q) (old/30bit_ntt_test.cu:46-48), 32-bit tables"""

    def __init__(self, n, q, psi):
        L = lib()
        self.n, self.q, self.psi = int(n), int(q), int(psi)
        self.k = int(L.orc_bit_length(self.q))
        self.mu = (1 << (2 * self.k)) // self.q
        self.psiinv = int(L.orc_modinv(self.psi, self.q))
        t64 = np.empty(self.n, dtype=np.uint64)
        L.orc_fill_table(self.psi, self.q, _p(t64), self.n)
        self.psi_tab = t64.astype(np.uint32)
        L.orc_fill_table(self.psiinv, self.q, _p(t64), self.n)
        self.psiinv_tab = t64.astype(np.uint32)


def forward30(a, prm):
    a = np.ascontiguousarray(a, dtype=np.uint32).copy()
    for row in a.reshape(-1, prm.n):
        lib().orc30_forward(_p32(row), prm.n, prm.q, prm.mu, prm.k, _p32(prm.psi_tab))
    return a


def inverse30(a, prm):
    a = np.ascontiguousarray(a, dtype=np.uint32).copy()
    for row in a.reshape(-1, prm.n):
        lib().orc30_inverse(_p32(row), prm.n, prm.q, prm.mu, prm.k, _p32(prm.psiinv_tab))
    return a


def pointwise30(a, b, prm):
    a = np.ascontiguousarray(a, dtype=np.uint32).copy()
    b = np.ascontiguousarray(b, dtype=np.uint32)
    lib().orc30_pointwise(_p32(a.reshape(-1)), _p32(b.reshape(-1)), a.size, prm.q, prm.mu, prm.k)
    return a


def bfv_constants(qs, psis, t, gamma):
    qs = np.array(qs, dtype=np.uint64)
    psis = np.array(psis, dtype=np.uint64)
    R = len(qs)
    r = R - 1
    psiinv = np.empty(R, dtype=np.uint64)
    ipq = np.empty(r, dtype=np.uint64)
    neg = np.empty(2, dtype=np.uint64)
    ptg = np.empty(r, dtype=np.uint64)
    iql = np.empty(r, dtype=np.uint64)
    qdt = np.empty(R, dtype=np.uint64)
    lib().orc_bfv_constants(_p(qs), _p(psis), R, int(t), int(gamma), _p(psiinv), _p(ipq), _p(neg), _p(ptg), _p(iql), _p(qdt))
    return dict(psiinv=psiinv, inv_punctured_q=ipq, neg_inv_q_mod_t_gamma=neg, prod_t_gamma_mod_q=ptg,
                inv_q_last_mod_q=iql, qi_div_t=qdt)
