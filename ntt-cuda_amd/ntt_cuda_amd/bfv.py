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
