"""The NTT sections of the reference's BFV drivers as launch sequences on an NTTContext (Python mirror of
compat/bfv_launch.hpp; same buffer layouts and num/division arguments as bfv_keygen.cuh:129-145,
bfv_encryption.cuh:268-271, bfv_decryption.cuh:98-101 of ozgunozerk/NTT-Cuda)."""


def keygen_ntt_a(ctx, secret_key, public_key, n, q_amount, stream=None):
    """bfv_keygen.cuh:129-133: sk -> NTT(sk) in place; pk0 = INTT(pk1 (.) NTT(sk)).  public_key is [2][r][n], pk1 second."""
    ctx.forward_batch(secret_key, q_amount, q_amount, stream)
    pk = public_key.reshape(-1)
    ctx.pointwise_mul(pk[: q_amount * n], pk[q_amount * n: 2 * q_amount * n], secret_key, q_amount, q_amount, stream)
    ctx.inverse_batch(pk[: q_amount * n], q_amount, q_amount, stream)


def keygen_ntt_b(ctx, public_key, q_amount, stream=None):
    """bfv_keygen.cuh:145"""
    ctx.forward_batch(public_key, q_amount, q_amount, stream)


def encryption_ntt(ctx, c, public_key, q_amount, stream=None):
    """bfv_encryption.cuh:268-271 as one fused launch"""
    ctx.polymul_batch(c, public_key, 2 * q_amount, q_amount, stream)


def decryption_ntt(ctx, c, secret_key, n, q_amount, stream=None):
    """bfv_decryption.cuh:98-101: c1 = c[(r+1) n :], num = r, division = r + 1"""
    c1 = c.reshape(-1)[(q_amount + 1) * n:]
    ctx.polymul_batch(c1, secret_key, q_amount, q_amount + 1, stream)


class BFVContext:
    """The BFV launch layer on the GPU (C ABI section "BFV" of include/mi355ntt.h): parameter bootstrap of
    demo.cu:62-272 and keygen_rns / encryption_rns / decryption_rns after their samplers.  `q`, `psi` list all primes,
    the special one (dropped by encryption) last."""

    def __init__(self, n, q, psi, t, gamma, device=0, exact_on_inexact_primes=False):
        import ctypes
        import numpy as np
        from . import lib, _check, _np_u64, u64p, vp, CTX_EXACT_ON_INEXACT_PRIMES
        self._h = vp()
        qs, ps = _np_u64(np.atleast_1d(q)), _np_u64(np.atleast_1d(psi))
        assert qs.size == ps.size
        _check(lib().mi355ntt_bfv_create(ctypes.byref(self._h), int(n), int(qs.size), qs.ctypes.data_as(u64p),
                                         ps.ctypes.data_as(u64p), int(t), int(gamma), int(device),
                                         CTX_EXACT_ON_INEXACT_PRIMES if exact_on_inexact_primes else 0), "mi355ntt_bfv_create")
        self.n, self.num_primes, self.t, self.gamma = int(n), int(qs.size), int(t), int(gamma)
        self.device = int(device)

    def _p(self, t, polys):
        """pointer of a buffer that must hold `polys` polynomials on this object's device (the C ABI takes raw pointers)"""
        from . import _ptr_n
        return _ptr_n(t, int(polys) * self.n, self.device)

    def close(self):
        from . import lib, vp
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().mi355ntt_bfv_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def uses_literal_kernels(self):
        from . import lib
        return bool(lib().mi355ntt_ctx_uses_literal_kernels(lib().mi355ntt_bfv_ntt(self._h)))

    def constants(self):
        import numpy as np
        from . import lib, _check, u64p
        r = self.num_primes - 1
        out = dict(inv_punctured_q=np.zeros(r, np.uint64), neg_inv_q_mod_t_gamma=np.zeros(2, np.uint64),
                   prod_t_gamma_mod_q=np.zeros(r, np.uint64), inv_q_last_mod_q=np.zeros(r, np.uint64),
                   qi_div_t=np.zeros(r + 1, np.uint64), base_change_matrix=np.zeros(2 * r, np.uint64), mu_gamma=np.zeros(1, np.uint64))
        _check(lib().mi355ntt_bfv_constants(self._h, *[v.ctypes.data_as(u64p) for v in out.values()]), "mi355ntt_bfv_constants")
        out["mu_gamma"] = int(out["mu_gamma"][0])
        return out

    def keygen(self, secret_key, public_key, e, stream=None):
        from . import lib, _check, _ptr, _stream
        R = self.num_primes
        _check(lib().mi355ntt_bfv_keygen(self._h, self._p(secret_key, R), self._p(public_key, 2 * R), self._p(e, R), _stream(stream)),
               "mi355ntt_bfv_keygen")

    def encrypt(self, c, public_key, e, m, stream=None):
        from . import lib, _check, _ptr, _stream
        R = self.num_primes
        _check(lib().mi355ntt_bfv_encrypt(self._h, self._p(c, 2 * R), self._p(public_key, 2 * R), self._p(e, 2 * R), self._p(m, 1),
                                          _stream(stream)), "mi355ntt_bfv_encrypt")

    def decrypt(self, c, secret_key, stream=None):
        """In place on c; returns the view of c holding the plaintext (c + n (num_primes - 2))."""
        from . import lib, _check, _ptr, _stream
        R = self.num_primes
        _check(lib().mi355ntt_bfv_decrypt(self._h, self._p(c, 2 * R), self._p(secret_key, R - 1), _stream(stream)), "mi355ntt_bfv_decrypt")
        off = self.n * (self.num_primes - 2)
        return c.reshape(-1)[off: off + self.n]

    # ---- batches of ciphertexts, layout [2][count][num_primes][n]
    def encrypt_batch(self, c, public_key, e, m, count, stream=None):
        from . import lib, _check, _stream
        R = self.num_primes
        _check(lib().mi355ntt_bfv_encrypt_batch(self._h, self._p(c, 2 * R * count), self._p(public_key, 2 * R), self._p(e, 2 * R * count),
                                                self._p(m, count), int(count), _stream(stream)), "mi355ntt_bfv_encrypt_batch")

    def decrypt_batch(self, c, secret_key, count, stream=None):
        """In place; the plaintext of ciphertext z is c.reshape(-1)[(z R + R - 2) n : ... + n]."""
        from . import lib, _check, _stream
        R = self.num_primes
        _check(lib().mi355ntt_bfv_decrypt_batch(self._h, self._p(c, 2 * R * count), self._p(secret_key, R), int(count), _stream(stream)),
               "mi355ntt_bfv_decrypt_batch")

    # ---- samplers and the complete drivers
    @property
    def keygen_random_bytes(self):
        from . import lib
        return int(lib().mi355ntt_bfv_keygen_random_bytes(self._h))

    @property
    def encrypt_random_bytes(self):
        from . import lib
        return int(lib().mi355ntt_bfv_encrypt_random_bytes(self._h))

    def sample_keygen(self, rnd, secret_key, public_key, temp, stream=None):
        from . import lib, _check, _ptr, _byte_ptr, _stream
        _check(lib().mi355ntt_bfv_sample_keygen(self._h, _byte_ptr(rnd), _ptr(secret_key), _ptr(public_key), _ptr(temp), _stream(stream)),
               "mi355ntt_bfv_sample_keygen")

    def sample_encrypt(self, rnd, c, e, stream=None):
        from . import lib, _check, _ptr, _byte_ptr, _stream
        _check(lib().mi355ntt_bfv_sample_encrypt(self._h, _byte_ptr(rnd), _ptr(c), _ptr(e), _stream(stream)), "mi355ntt_bfv_sample_encrypt")

    def keygen_rns(self, rnd, secret_key, public_key, temp, nonce=0, stream=None):
        """keygen_rns complete (bfv_keygen.cuh:95-151): keystream (reference default key) -> samplers -> key generation"""
        from . import lib, _check, _ptr, _byte_ptr, _stream
        _check(lib().mi355ntt_bfv_keygen_rns(self._h, _byte_ptr(rnd), _ptr(secret_key), _ptr(public_key), _ptr(temp), int(nonce),
                                             _stream(stream)), "mi355ntt_bfv_keygen_rns")

    def encryption_rns(self, c, public_key, rnd, e, m, nonce=0, stream=None):
        """encryption_rns complete (bfv_encryption.cuh:223-290)"""
        from . import lib, _check, _ptr, _byte_ptr, _stream
        _check(lib().mi355ntt_bfv_encryption_rns(self._h, _ptr(c), _ptr(public_key), _byte_ptr(rnd), _ptr(e), _ptr(m), int(nonce),
                                                 _stream(stream)), "mi355ntt_bfv_encryption_rns")
