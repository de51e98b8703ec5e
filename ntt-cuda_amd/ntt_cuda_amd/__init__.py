"""ntt_cuda_amd -- Python plumbing over the MI355X-native NTT engine's C ABI (libmi355ntt.so).

The product is the HIP library; this module only loads it through ctypes, mirrors the reference's host
call surface (BFV_Scheme/ntt_60bit.cuh:267-386,608-697 and poly_arithmetic.cuh:9-126,277-310 of
ozgunozerk/NTT-Cuda) with the same names and argument meaning, and uses torch for device memory and
streams.  There is NO CPU fallback: if the library is missing, import of the native handle fails loudly.

Polynomials are torch.int64 CUDA tensors holding the bit patterns of unsigned 64-bit residues.
"""
import ctypes
import os
import subprocess

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG_DIR)                  # ntt-cuda_amd/
LIB_PATH = os.path.join(_ROOT, "libmi355ntt.so")

u64 = ctypes.c_ulonglong
u64p = ctypes.POINTER(u64)
u32p = ctypes.POINTER(ctypes.c_uint)
vp = ctypes.c_void_p

OK = 0
EINVAL, EUNSUPPORTED, EHIP, ENOMEM, EPARAM = -1, -2, -3, -4, -5

# name -> (restype, argtypes); mirrors include/mi355ntt.h one to one
_SIGNATURES = {
    "mi355ntt_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "mi355ntt_last_hip_error": (ctypes.c_int, []),
    "mi355ntt_pair_fault_count": (ctypes.c_ulonglong, [ctypes.c_int]),
    "mi355ntt_version": (ctypes.c_char_p, []),
    "mi355ntt_bit_length": (ctypes.c_uint, [u64]),
    "mi355ntt_barrett_mu": (u64, [u64, ctypes.c_uint]),
    "mi355ntt_mulmod": (u64, [u64, u64, u64]),
    "mi355ntt_modpow": (u64, [u64, u64, u64]),
    "mi355ntt_modinv": (u64, [u64, u64]),
    "mi355ntt_bit_reverse": (u64, [u64, ctypes.c_int]),
    "mi355ntt_fill_tables": (ctypes.c_int, [u64, u64, u64, ctypes.c_uint, u64p, u64p]),
    "mi355ntt_get_params": (ctypes.c_int, [ctypes.c_uint, u64p, u64p, u64p, u64p, u32p]),
    "mi355ntt_barrett_is_exact": (ctypes.c_int, [u64]),
    "mi355ntt_ctx_create": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.c_uint, ctypes.c_uint, u64p, u64p, ctypes.c_int]),
    "mi355ntt_ctx_create_ex": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.c_uint, ctypes.c_uint, u64p, u64p, ctypes.c_int, ctypes.c_uint]),
    "mi355ntt_ctx_uses_literal_kernels": (ctypes.c_int, [vp]),
    "mi355ntt_ctx_kernel_class": (ctypes.c_int, [vp]),
    "mi355ntt_bfv_create": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.c_uint, ctypes.c_uint, u64p, u64p, u64, u64, ctypes.c_int, ctypes.c_uint]),
    "mi355ntt_bfv_destroy": (ctypes.c_int, [vp]),
    "mi355ntt_bfv_ntt": (vp, [vp]),
    "mi355ntt_bfv_constants": (ctypes.c_int, [vp, u64p, u64p, u64p, u64p, u64p, u64p, u64p]),
    "mi355ntt_bfv_keygen": (ctypes.c_int, [vp, vp, vp, vp, vp]),
    "mi355ntt_bfv_encrypt": (ctypes.c_int, [vp, vp, vp, vp, vp, vp]),
    "mi355ntt_bfv_decrypt": (ctypes.c_int, [vp, vp, vp, vp]),
    "mi355ntt_forward30_raw": (ctypes.c_int, [vp, ctypes.c_uint, vp, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, vp]),
    "mi355ntt_inverse30_raw": (ctypes.c_int, [vp, ctypes.c_uint, vp, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, vp]),
    "mi355ntt_forward30_batch_raw": (ctypes.c_int, [vp, ctypes.c_uint, vp, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, vp]),
    "mi355ntt_inverse30_batch_raw": (ctypes.c_int, [vp, ctypes.c_uint, vp, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, vp]),
    "mi355ntt_barrett30_raw": (ctypes.c_int, [vp, vp, ctypes.c_size_t, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, vp]),
    "mi355ntt_salsa20_keystream": (ctypes.c_int, [vp, ctypes.c_size_t, ctypes.c_char_p, u64, vp]),
    "mi355ntt_bfv_keygen_random_bytes": (ctypes.c_size_t, [vp]),
    "mi355ntt_bfv_encrypt_random_bytes": (ctypes.c_size_t, [vp]),
    "mi355ntt_bfv_sample_keygen": (ctypes.c_int, [vp, vp, vp, vp, vp, vp]),
    "mi355ntt_bfv_sample_encrypt": (ctypes.c_int, [vp, vp, vp, vp, vp]),
    "mi355ntt_bfv_keygen_rns": (ctypes.c_int, [vp, vp, vp, vp, vp, u64, vp]),
    "mi355ntt_bfv_encryption_rns": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, u64, vp]),
    "mi355ntt_ctx_destroy": (ctypes.c_int, [vp]),
    "mi355ntt_ctx_n": (ctypes.c_uint, [vp]),
    "mi355ntt_ctx_num_primes": (ctypes.c_uint, [vp]),
    "mi355ntt_ctx_device": (ctypes.c_int, [vp]),
    "mi355ntt_ctx_prime": (ctypes.c_int, [vp, ctypes.c_uint, u64p, u64p, u32p, u64p, u64p]),
    "mi355ntt_ctx_psi_tables": (vp, [vp]),
    "mi355ntt_ctx_psiinv_tables": (vp, [vp]),
    "mi355ntt_forward": (ctypes.c_int, [vp, vp, ctypes.c_uint, vp]),
    "mi355ntt_inverse": (ctypes.c_int, [vp, vp, ctypes.c_uint, vp]),
    "mi355ntt_forward_double": (ctypes.c_int, [vp, vp, vp, ctypes.c_uint, vp, vp]),
    "mi355ntt_forward_batch": (ctypes.c_int, [vp, vp, ctypes.c_uint, ctypes.c_uint, vp]),
    "mi355ntt_inverse_batch": (ctypes.c_int, [vp, vp, ctypes.c_uint, ctypes.c_uint, vp]),
    "mi355ntt_pointwise_mul": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_uint, ctypes.c_uint, vp]),
    "mi355ntt_pointwise_mul_scalar": (ctypes.c_int, [vp, vp, u64, ctypes.c_uint, vp]),
    "mi355ntt_polymul_batch": (ctypes.c_int, [vp, vp, vp, ctypes.c_uint, ctypes.c_uint, vp]),
    "mi355ntt_polymul_batch_shared": (ctypes.c_int, [vp, vp, vp, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, vp]),
    "mi355ntt_bfv_encrypt_batch": (ctypes.c_int, [vp, vp, vp, vp, vp, ctypes.c_uint, vp]),
    "mi355ntt_bfv_decrypt_batch": (ctypes.c_int, [vp, vp, vp, ctypes.c_uint, vp]),
    "mi355ntt_shard_range": (ctypes.c_int, [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, u32p, u32p]),
    "mi355ntt_shards_create": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_uint, ctypes.c_uint]),
    "mi355ntt_shards_destroy": (ctypes.c_int, [vp]),
    "mi355ntt_shards_world": (ctypes.c_uint, [vp]),
    "mi355ntt_shards_transform": (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_uint, ctypes.c_uint, vp]),
    "mi355ntt_shards_scatter_transform_gather": (ctypes.c_int, [vp, ctypes.c_int, vp, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, vp]),
    "mi355ntt_synth_splitmix": (ctypes.c_int, [vp, vp, ctypes.c_uint, ctypes.c_uint, u64, vp]),
    "mi355ntt_ctx_clock_probe": (ctypes.c_int, [vp, vp]),
    "mi355ntt_ctx_occupy": (ctypes.c_int, [vp, ctypes.c_uint, ctypes.c_uint, vp]),
    "mi355ntt_ctx_probed_clock_mhz": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_double)]),
    "mi355ntt_raw_cache_clear": (ctypes.c_int, []),
    "mi355ntt_raw_uses_fast_kernels": (ctypes.c_int, [ctypes.c_uint, vp, ctypes.c_int, ctypes.c_uint, u64p, u64p, u32p]),
    "mi355ntt_raw_trust_tables": (ctypes.c_int, [ctypes.c_uint, vp, ctypes.c_int, ctypes.c_uint, u64p, u64p, u32p]),
    "mi355ntt_forward_raw": (ctypes.c_int, [vp, ctypes.c_uint, vp, u64, u64, ctypes.c_int, vp]),
    "mi355ntt_inverse_raw": (ctypes.c_int, [vp, ctypes.c_uint, vp, u64, u64, ctypes.c_int, vp]),
    "mi355ntt_forward_batch_raw": (ctypes.c_int, [vp, ctypes.c_uint, vp, ctypes.c_uint, ctypes.c_uint, u64p, u64p, u32p, vp]),
    "mi355ntt_inverse_batch_raw": (ctypes.c_int, [vp, ctypes.c_uint, vp, ctypes.c_uint, ctypes.c_uint, u64p, u64p, u32p, vp]),
    "mi355ntt_barrett_raw": (ctypes.c_int, [vp, vp, vp, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, u64p, u64p, u32p, vp]),
    "mi355ntt_poly_add_raw": (ctypes.c_int, [vp, vp, ctypes.c_uint, vp, u64]),
    "mi355ntt_poly_sub_raw": (ctypes.c_int, [vp, vp, ctypes.c_uint, vp, u64]),
    "mi355ntt_poly_negate_raw": (ctypes.c_int, [vp, ctypes.c_uint, vp, u64]),
    "mi355ntt_poly_add_integer_raw": (ctypes.c_int, [vp, u64, ctypes.c_uint, vp, u64]),
    "mi355ntt_poly_mul_int_t_raw": (ctypes.c_int, [vp, u64, ctypes.c_uint, vp, u64]),
    "mi355ntt_barrett_int_raw": (ctypes.c_int, [vp, u64, ctypes.c_uint, u64, u64, ctypes.c_int, vp]),
}

_lib = None


class NTTError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().mi355ntt_strerror(code).decode()
        if code == EHIP:
            msg += " (hipError_t %d)" % lib().mi355ntt_last_hip_error()
        super().__init__("%s: %s [%d]" % (where, msg, code))


def build(force=False):
    """Compile libmi355ntt.so for gfx950 in tree (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _ROOT, "-s", "-j4"])
    return LIB_PATH


def lib():
    """The native library handle.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libmi355ntt.so not found at %s -- run __graft_entry__.build() / make -C ntt-cuda_amd" % LIB_PATH)
        # torch bundles its own HIP runtime (SONAME libamdhip64.so.7).  Load it first so that this library
        # binds to the same runtime instance: two HIP runtimes in one process do not share devices/streams.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            f = getattr(L, name)       # AttributeError if the ABI drifted from the header
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


CTX_EXACT_ON_INEXACT_PRIMES = 1   # MI355NTT_CTX_EXACT_ON_INEXACT_PRIMES


def barrett_is_exact(q):
    """1 when the reference's single-subtraction Barrett is exact for every pair of canonical operands mod q."""
    return bool(lib().mi355ntt_barrett_is_exact(int(q)))


def _check(code, where):
    if code != OK:
        raise NTTError(code, where)


def _np_u64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def _ptr(t):
    """Device pointer of a torch tensor (or raw int address)."""
    if isinstance(t, int):
        return vp(t)
    assert t.is_cuda and t.is_contiguous() and t.element_size() == 8, "need a contiguous 64-bit CUDA tensor"
    return vp(t.data_ptr())


def _ptr_n(t, words, device=None):
    """Device pointer of a tensor that must hold at least `words` 64-bit words (and live on `device`, if given): the C ABI
    takes plain pointers, so a short view or a tensor of another GPU would be silent out-of-bounds traffic."""
    if isinstance(t, int):
        return vp(t)
    assert t.is_cuda and t.is_contiguous() and t.element_size() == 8, "need a contiguous 64-bit CUDA tensor"
    if t.numel() < words:
        raise ValueError("tensor holds %d words, the call addresses %d" % (t.numel(), words))
    if device is not None and t.device.index != device:
        raise ValueError("tensor is on cuda:%s, the context on cuda:%d" % (t.device.index, device))
    return vp(t.data_ptr())


def _byte_ptr(t):
    assert t.is_cuda and t.is_contiguous() and t.element_size() == 1, "need a contiguous byte CUDA tensor"
    return vp(t.data_ptr())


def _ptr32(t):
    assert t.is_cuda and t.is_contiguous() and t.element_size() == 4, "need a contiguous 32-bit CUDA tensor"
    return vp(t.data_ptr())


# the reference's 30-bit path (old/ntt_30bit.cuh): int32 CUDA tensors holding the 32-bit words
def forward30(a, n, q, mu, bit_length, psi_table, num=1, stream=None):
    _check(lib().mi355ntt_forward30_batch_raw(_ptr32(a), int(n), _ptr32(psi_table), int(num), int(q), int(mu), int(bit_length),
                                              _stream(stream)), "mi355ntt_forward30_batch_raw")


def inverse30(a, n, q, mu, bit_length, psiinv_table, num=1, stream=None):
    _check(lib().mi355ntt_inverse30_batch_raw(_ptr32(a), int(n), _ptr32(psiinv_table), int(num), int(q), int(mu), int(bit_length),
                                              _stream(stream)), "mi355ntt_inverse30_batch_raw")


def barrett30(a, b, q, mu, bit_length, stream=None):
    _check(lib().mi355ntt_barrett30_raw(_ptr32(a), _ptr32(b), a.numel(), int(q), int(mu), int(bit_length), _stream(stream)),
           "mi355ntt_barrett30_raw")


def salsa20_keystream(out, key32, nonce=0, stream=None):
    """generate_random / generate_random_default (distributions.cuh:192-276): fills the byte tensor `out` (whole 64-byte
    blocks) with the Salsa20/20 keystream of (key32, 64-bit nonce), block counter from 0."""
    assert len(key32) == 32
    _check(lib().mi355ntt_salsa20_keystream(_byte_ptr(out), out.numel(), bytes(key32), int(nonce), _stream(stream)), "mi355ntt_salsa20_keystream")


def _stream(stream=None):
    if stream is None:
        import torch
        stream = torch.cuda.current_stream()
    if isinstance(stream, int):
        return vp(stream)
    return vp(stream.cuda_stream)


# --------------------------------------------------------------------------- host-only helpers
def bit_length(q):
    return int(lib().mi355ntt_bit_length(int(q)))


def barrett_mu(q, bits):
    return int(lib().mi355ntt_barrett_mu(int(q), int(bits)))


def modpow128(a, b, mod):
    """helper.h:8-28"""
    return int(lib().mi355ntt_modpow(int(a), int(b), int(mod)))


def modinv128(a, q):
    """helper.h:52-56"""
    return int(lib().mi355ntt_modinv(int(a), int(q)))


def bitReverse(a, bit_length_):
    """helper.h:58-70"""
    return int(lib().mi355ntt_bit_reverse(int(a), int(bit_length_)))


def fillTablePsi128(psi, q, psiinv, n):
    """parameter.h:5-12; returns (psiTable, psiinvTable) as numpy uint64 arrays."""
    tp = np.empty(n, dtype=np.uint64)
    ti = np.empty(n, dtype=np.uint64)
    _check(lib().mi355ntt_fill_tables(int(psi), int(psiinv), int(q), int(n), tp.ctypes.data_as(u64p), ti.ctypes.data_as(u64p)),
           "fillTablePsi128")
    return tp, ti


def getParams(n):
    """parameter.h:31-79; returns (q, psi, psiinv, ninv, q_bit)."""
    q, psi, psiinv, ninv = u64(), u64(), u64(), u64()
    bits = ctypes.c_uint()
    _check(lib().mi355ntt_get_params(int(n), ctypes.byref(q), ctypes.byref(psi), ctypes.byref(psiinv), ctypes.byref(ninv),
                                     ctypes.byref(bits)), "getParams")
    return q.value, psi.value, psiinv.value, ninv.value, bits.value


# --------------------------------------------------------------------------- tensors
def to_device(a, device="cuda:0"):
    """numpy uint64 array -> torch.int64 CUDA tensor with the same bits."""
    import torch
    return torch.from_numpy(_np_u64(a).view(np.int64)).to(device)


def to_host(t):
    """torch.int64 CUDA tensor -> numpy uint64 array."""
    return t.detach().cpu().numpy().view(np.uint64)


# --------------------------------------------------------------------------- context API
class NTTContext:
    """Immutable per-(n, primes) state: replaces the bootstrap at demo.cu:62-196 and the __constant__
    q_cons/q_bit_cons/mu_cons symbols (ntt_60bit.cuh:8-10)."""

    def __init__(self, n, q, psi, device=0, exact_on_inexact_primes=False):
        """exact_on_inexact_primes: run the exact kernels even when some modulus makes the reference's single-subtraction
        Barrett inexact (barrett_is_exact(q) == 0); by default such a context reproduces the reference's words with the
        literal kernels (include/mi355ntt.h, "Arithmetic contract")."""
        self._h = vp()
        qs, ps = _np_u64(np.atleast_1d(q)), _np_u64(np.atleast_1d(psi))
        assert qs.size == ps.size
        _check(lib().mi355ntt_ctx_create_ex(ctypes.byref(self._h), int(n), int(qs.size), qs.ctypes.data_as(u64p),
                                            ps.ctypes.data_as(u64p), int(device),
                                            CTX_EXACT_ON_INEXACT_PRIMES if exact_on_inexact_primes else 0), "mi355ntt_ctx_create_ex")
        self.n = int(n)
        self.num_primes = int(qs.size)
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().mi355ntt_ctx_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def uses_literal_kernels(self):
        return bool(lib().mi355ntt_ctx_uses_literal_kernels(self._h))

    @property
    def kernel_class(self):
        """(headroom class 2..6, near-2^k) the throughput kernels were selected by (mi355ntt_ctx_kernel_class)"""
        v = lib().mi355ntt_ctx_kernel_class(self._h)
        return v & 15, bool(v & 16)

    @property
    def literal_routing(self):
        """0: throughput kernels for every prime; 1: literal kernels for every prime; 2: per prime -- the polynomials of the
        Barrett-inexact primes run the literal kernels, all others the throughput kernels (mi355ntt_ctx_uses_literal_kernels)"""
        return int(lib().mi355ntt_ctx_uses_literal_kernels(self._h))

    def prime(self, i):
        q, mu, psi, psiinv = u64(), u64(), u64(), u64()
        bits = ctypes.c_uint()
        _check(lib().mi355ntt_ctx_prime(self._h, int(i), ctypes.byref(q), ctypes.byref(mu), ctypes.byref(bits), ctypes.byref(psi),
                                        ctypes.byref(psiinv)), "mi355ntt_ctx_prime")
        return dict(q=q.value, mu=mu.value, bit_length=bits.value, psi=psi.value, psiinv=psiinv.value)

    @property
    def psi_tables_ptr(self):
        return int(lib().mi355ntt_ctx_psi_tables(self._h) or 0)

    @property
    def psiinv_tables_ptr(self):
        return int(lib().mi355ntt_ctx_psiinv_tables(self._h) or 0)

    # forwardNTT / inverseNTT (ntt_60bit.cuh:314,350)
    def _p(self, t, polys=1):
        return _ptr_n(t, int(polys) * self.n, self.device)

    def forward(self, a, prime_idx=0, stream=None):
        _check(lib().mi355ntt_forward(self._h, self._p(a), int(prime_idx), _stream(stream)), "mi355ntt_forward")

    def inverse(self, a, prime_idx=0, stream=None):
        _check(lib().mi355ntt_inverse(self._h, self._p(a), int(prime_idx), _stream(stream)), "mi355ntt_inverse")

    def forward_double(self, a, b, prime_idx=0, stream1=None, stream2=None):
        _check(lib().mi355ntt_forward_double(self._h, self._p(a), self._p(b), int(prime_idx), _stream(stream1), _stream(stream2)),
               "mi355ntt_forward_double")

    # forwardNTT_batch / inverseNTT_batch (ntt_60bit.cuh:608,652)
    def forward_batch(self, a, num, division=None, stream=None):
        _check(lib().mi355ntt_forward_batch(self._h, self._p(a, num), int(num), int(division or self.num_primes), _stream(stream)),
               "mi355ntt_forward_batch")

    def inverse_batch(self, a, num, division=None, stream=None):
        _check(lib().mi355ntt_inverse_batch(self._h, self._p(a, num), int(num), int(division or self.num_primes), _stream(stream)),
               "mi355ntt_inverse_batch")

    # barrett / barrett_batch / barrett_batch_3param (poly_arithmetic.cuh:9-98)
    def pointwise_mul(self, c, a, b, num, division=None, stream=None):
        _check(lib().mi355ntt_pointwise_mul(self._h, self._p(c, num), self._p(a, num), self._p(b, num), int(num),
                                            int(division or self.num_primes), _stream(stream)), "mi355ntt_pointwise_mul")

    # barrett_int (poly_arithmetic.cuh:100)
    def pointwise_mul_scalar(self, a, b, prime_idx=0, stream=None):
        _check(lib().mi355ntt_pointwise_mul_scalar(self._h, self._p(a), int(b), int(prime_idx), _stream(stream)),
               "mi355ntt_pointwise_mul_scalar")

    # forwardNTT_batch -> barrett_batch -> inverseNTT_batch (bfv_encryption.cuh:268-271), fused
    def polymul_batch(self, a, bhat, num, division=None, stream=None):
        _check(lib().mi355ntt_polymul_batch(self._h, self._p(a, num), self._p(bhat, num), int(num), int(division or self.num_primes),
                                            _stream(stream)), "mi355ntt_polymul_batch")


    def synth_splitmix(self, a, num, seed_base=1, division=None, stream=None):
        """a[y] = splitmix64(seed_base + y) mod q[y % division]: the benchmark's synthetic inputs"""
        division = self.num_primes if division is None else division
        _check(lib().mi355ntt_synth_splitmix(self._h, self._p(a, num), int(num), int(division), int(seed_base), _stream(stream)),
               "mi355ntt_synth_splitmix")
        return a

    def clock_probe(self, stream=None):
        """enqueue a clock probe on the stream: shader cycles counted over 20 us of the 100 MHz clock"""
        _check(lib().mi355ntt_ctx_clock_probe(self._h, _stream(stream)), "mi355ntt_ctx_clock_probe")

    def occupy(self, workgroups, microseconds, stream=None):
        """foreign load: `workgroups` whole CUs held for `microseconds` on `stream` (mi355ntt_ctx_occupy)"""
        _check(lib().mi355ntt_ctx_occupy(self._h, int(workgroups), int(microseconds), _stream(stream)), "mi355ntt_ctx_occupy")

    def probed_clock_mhz(self):
        """shader clock measured by the last probe: what the launches enqueued in front of it left the chip at (synchronises); 0.0 = none"""
        v = ctypes.c_double(0.0)
        _check(lib().mi355ntt_ctx_probed_clock_mhz(self._h, ctypes.byref(v)), "mi355ntt_ctx_probed_clock_mhz")
        return float(v.value)

    def polymul_batch_shared(self, a, bhat, num, division=None, group=0, stream=None):
        """polymul_batch with shared second operands: polynomial y multiplies with bhat[(y // group) * division + y % division]
        (group = 0: the whole batch is one group)"""
        d, g = int(division or self.num_primes), int(group)
        groups = -(-int(num) // g) if g else 1
        _check(lib().mi355ntt_polymul_batch_shared(self._h, self._p(a, num), self._p(bhat, d * groups), int(num), d, g, _stream(stream)),
               "mi355ntt_polymul_batch_shared")


OP_FORWARD, OP_INVERSE, OP_FORWARD_INVERSE, OP_POLYMUL = 0, 1, 2, 3     # MI355NTT_OP_*


def shard_range(num, division, rank, world):
    """mi355ntt_shard_range: (first polynomial, count) of shard `rank` of `world` (host only; the C-ABI twin of shard.shard_range)"""
    first, count = ctypes.c_uint(), ctypes.c_uint()
    _check(lib().mi355ntt_shard_range(int(num), int(division), int(rank), int(world), ctypes.byref(first), ctypes.byref(count)),
           "mi355ntt_shard_range")
    return first.value, count.value


class ShardSet:
    """mi355ntt_shards: several devices (or several logical shards of one device) driven from ONE process -- device-resident shards
    transformed concurrently, or a root-resident batch scattered by peer copies, transformed and gathered (include/mi355ntt.h)."""

    def __init__(self, contexts, max_polys_per_piece=0):
        self.contexts = list(contexts)          # (kept alive: the object holds their handles)
        arr = (vp * len(self.contexts))(*[c._h for c in self.contexts])
        self._h = vp()
        _check(lib().mi355ntt_shards_create(ctypes.byref(self._h), arr, len(self.contexts), int(max_polys_per_piece)), "mi355ntt_shards_create")
        self.n = self.contexts[0].n
        self.num_primes = self.contexts[0].num_primes

    @property
    def world(self):
        return int(lib().mi355ntt_shards_world(self._h))

    def transform(self, op, shards, num, division=None, bhat=None, stream=None):
        """shards[r]: tensor on contexts[r]'s device holding the polynomials shard_range(num, division, r, world) names"""
        division = int(division or self.num_primes)
        ptrs, bptrs = [], []
        for r, t in enumerate(shards):
            _, c = shard_range(num, division, r, len(shards))
            ptrs.append(_ptr_n(t, c * self.n, self.contexts[r].device) if c else None)
            if bhat is not None:
                bptrs.append(_ptr_n(bhat[r], c * self.n, self.contexts[r].device) if c else None)
        arr = (vp * len(ptrs))(*ptrs)
        barr = (vp * len(bptrs))(*bptrs) if bhat is not None else None
        _check(lib().mi355ntt_shards_transform(self._h, int(op), arr, barr, int(num), division, _stream(stream)), "mi355ntt_shards_transform")

    def scatter_transform_gather(self, op, full, num, division=None, chunks=4, stream=None):
        """full: [num, n] tensor on contexts[0]'s device, transformed in place through every lane"""
        division = int(division or self.num_primes)
        _check(lib().mi355ntt_shards_scatter_transform_gather(self._h, int(op), _ptr_n(full, int(num) * self.n, self.contexts[0].device), int(num),
                                                              division, int(chunks), _stream(stream)), "mi355ntt_shards_scatter_transform_gather")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().mi355ntt_shards_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------- reference-named raw API
def forwardNTT(device_a, n, stream, q, mu, bit_length_, psi_powers):
    """ntt_60bit.cuh:314 -- same argument order; psi_powers is a device tensor in the reference table format."""
    _check(lib().mi355ntt_forward_raw(_ptr(device_a), int(n), _stream(stream), int(q), int(mu), int(bit_length_),
                                      _ptr(psi_powers)), "forwardNTT")


def inverseNTT(device_a, n, stream, q, mu, bit_length_, psiinv_powers):
    """ntt_60bit.cuh:350"""
    _check(lib().mi355ntt_inverse_raw(_ptr(device_a), int(n), _stream(stream), int(q), int(mu), int(bit_length_),
                                      _ptr(psiinv_powers)), "inverseNTT")


def forwardNTTdouble(device_a, device_b, n, stream1, stream2, q, mu, bit_length_, psi_powers):
    """ntt_60bit.cuh:267"""
    forwardNTT(device_a, n, stream1, q, mu, bit_length_, psi_powers)
    forwardNTT(device_b, n, stream2, q, mu, bit_length_, psi_powers)


class Moduli:
    """Stand-in for the __constant__ q_cons / mu_cons / q_bit_cons the batch kernels read (ntt_60bit.cuh:8-10)."""

    def __init__(self, q, mu=None, bits=None):
        self.q = _np_u64(np.atleast_1d(q))
        self.bits = np.ascontiguousarray(bits if bits is not None else [bit_length(x) for x in self.q], dtype=np.uint32)
        self.mu = _np_u64(mu if mu is not None else [barrett_mu(x, k) for x, k in zip(self.q, self.bits)])

    def args(self):
        return self.q.ctypes.data_as(u64p), self.mu.ctypes.data_as(u64p), self.bits.ctypes.data_as(u32p)


# the element-wise host wrappers of poly_arithmetic.cuh:312-352 (same names and argument order)
def poly_add_device(device_a, device_b, n, stream, q):
    _check(lib().mi355ntt_poly_add_raw(_ptr_n(device_a, n), _ptr_n(device_b, n), int(n), _stream(stream), int(q)), "poly_add_device")


def poly_mul_int_t(device_a, b, n, stream, t):
    _check(lib().mi355ntt_poly_mul_int_t_raw(_ptr_n(device_a, n), int(b), int(n), _stream(stream), int(t)), "poly_mul_int_t")


def poly_sub_device(device_a, device_b, n, stream, q):
    """the reference's poly_sub (poly_arithmetic.cuh:168-179) adds q where a[i] < b[i] and never subtracts b: mirrored literally"""
    _check(lib().mi355ntt_poly_sub_raw(_ptr_n(device_a, n), _ptr_n(device_b, n), int(n), _stream(stream), int(q)), "poly_sub_device")


def poly_negate_device(device_a, n, stream, q):
    _check(lib().mi355ntt_poly_negate_raw(_ptr_n(device_a, n), int(n), _stream(stream), int(q)), "poly_negate_device")


def poly_add_integer_device(device_a, b, n, stream, q):
    _check(lib().mi355ntt_poly_add_integer_raw(_ptr_n(device_a, n), int(b), int(n), _stream(stream), int(q)), "poly_add_integer_device")


def raw_cache_clear():
    """Forget every context the raw entry points derived from caller tables (include/mi355ntt.h, "Routing")."""
    _check(lib().mi355ntt_raw_cache_clear(), "mi355ntt_raw_cache_clear")


def raw_uses_fast_kernels(n, table, moduli, inverse=False):
    """True when forwardNTT_batch / inverseNTT_batch with this table and these moduli run the throughput kernels."""
    return bool(lib().mi355ntt_raw_uses_fast_kernels(int(n), _ptr(table), 1 if inverse else 0, int(moduli.q.size), *moduli.args()))


def raw_trust_tables(n, table, moduli, inverse=False):
    """The caller's promise that `table` keeps its contents: raw calls on it skip the per-call table comparison."""
    return bool(lib().mi355ntt_raw_trust_tables(int(n), _ptr(table), 1 if inverse else 0, int(moduli.q.size), *moduli.args()))


def forwardNTT_batch(device_a, n, psi_powers, num, division, moduli, stream=None):
    """ntt_60bit.cuh:608 (+ the moduli the reference reads from __constant__ memory)"""
    _check(lib().mi355ntt_forward_batch_raw(_ptr_n(device_a, int(num) * int(n)), int(n), _ptr_n(psi_powers, int(division) * int(n)), int(num),
                                            int(division), *moduli.args(), _stream(stream)), "forwardNTT_batch")


def inverseNTT_batch(device_a, n, psiinv_powers, num, division, moduli, stream=None):
    """ntt_60bit.cuh:652"""
    _check(lib().mi355ntt_inverse_batch_raw(_ptr_n(device_a, int(num) * int(n)), int(n), _ptr_n(psiinv_powers, int(division) * int(n)), int(num),
                                            int(division), *moduli.args(), _stream(stream)), "inverseNTT_batch")


def barrett(a, b, q, mu, qbit, stream=None):
    """poly_arithmetic.cuh:9 -- a[i] = a[i]*b[i] mod q over a.numel() coefficients"""
    m = Moduli([q], [mu], [qbit])
    _check(lib().mi355ntt_barrett_raw(_ptr(a), _ptr(a), _ptr(b), int(a.numel()), 1, 1, *m.args(), _stream(stream)), "barrett")


def barrett_batch(a, b, n, division, moduli, num=None, stream=None):
    """poly_arithmetic.cuh:36"""
    num = int(num if num is not None else a.numel() // n)
    _check(lib().mi355ntt_barrett_raw(_ptr(a), _ptr(a), _ptr(b), int(n), num, int(division), *moduli.args(), _stream(stream)),
           "barrett_batch")


def barrett_batch_3param(c, a, b, n, division, moduli, num=None, stream=None):
    """poly_arithmetic.cuh:68"""
    num = int(num if num is not None else a.numel() // n)
    _check(lib().mi355ntt_barrett_raw(_ptr(c), _ptr(a), _ptr(b), int(n), num, int(division), *moduli.args(), _stream(stream)),
           "barrett_batch_3param")


def barrett_int(a, b, q, mu, qbit, stream=None):
    """poly_arithmetic.cuh:100"""
    _check(lib().mi355ntt_barrett_int_raw(_ptr(a), int(b), int(a.numel()), int(q), int(mu), int(qbit), _stream(stream)),
           "barrett_int")


def half_poly_mul_device(device_a, device_b, n, stream, q, mu, bit_length_, psi_powers, psiinv_powers):
    """poly_arithmetic.cuh:303-310: a = INTT(NTT(a) (.) b), b already in the NTT domain"""
    forwardNTT(device_a, n, stream, q, mu, bit_length_, psi_powers)
    barrett(device_a, device_b, q, mu, bit_length_, stream)
    inverseNTT(device_a, n, stream, q, mu, bit_length_, psiinv_powers)


def full_poly_mul_device(device_a, device_b, n, stream1, stream2, q, mu, bit_length_, psi_powers):
    """poly_arithmetic.cuh:296-301: NTT both operands and multiply pointwise (result stays in the NTT domain)"""
    import torch
    forwardNTTdouble(device_a, device_b, n, stream1, stream2, q, mu, bit_length_, psi_powers)
    s2 = stream2 if stream2 is not None else torch.cuda.current_stream()
    if stream1 is not None and stream1 is not s2:
        s2.wait_stream(stream1)   # the reference relies on legacy default-stream ordering here
    barrett(device_a, device_b, q, mu, bit_length_, s2)


def full_poly_mul(host_a, host_b, n, q, mu, bit_length_, psi_powers, psiinv_powers, device="cuda:0"):
    """poly_arithmetic.cuh:277-294: H2D, NTT x2, pointwise, INTT, D2H; returns a new host array"""
    import torch
    da, db = to_device(host_a, device), to_device(host_b, device)
    s = torch.cuda.current_stream()
    full_poly_mul_device(da, db, n, s, s, q, mu, bit_length_, psi_powers)
    inverseNTT(da, n, s, q, mu, bit_length_, psiinv_powers)
    return to_host(da)
