"""GPU: the NTT sections of the reference's BFV drivers (host launch layer) on the throughput
kernels, against the oracle and the reference's known-answer test."""
import os

import numpy as np
import pytest

import params as P

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat1_decryption_n4096.npz")


def test_decryption_section_replays_reference_kat(native, oracle, gpu):
    """decryption_test.cu: the embedded ciphertext, pushed through decryption_ntt (bfv_decryption.cuh:98-101 on the
    fused kernel) and then the rest of decryption_rns (oracle), yields m[i] = i % 10."""
    import torch
    from ntt_cuda_amd import bfv
    z = np.load(GOLD)
    n, r = int(z["n"]), 2
    q, psi = [int(x) for x in z["q"]], [int(x) for x in z["psi"]]
    ctx = native.NTTContext(n, q, psi)                       # r + 1 primes, as decryption_test.cu builds them
    c = native.to_device(z["c_host"])
    sk = native.to_device(z["sk_host"])
    bfv.decryption_ntt(ctx, c, sk, n, r)
    torch.cuda.synchronize()
    got = native.to_host(c)
    assert P.digest(got[(r + 1) * n:(r + 1) * n + r * n]) == P.KAT1_STAGE_DIGESTS[2]
    assert np.array_equal(got[: (r + 1) * n], z["c_host"][: (r + 1) * n])          # c0 untouched
    # finish decryption on the host (everything after the NTT section) and compare with the reference plaintext
    _, stages = oracle.bfv_decrypt(z["c_host"], z["sk_host"], q, psi, n, int(z["t"]), int(z["gamma"]), want_stages=True)
    assert np.array_equal(got[(r + 1) * n:(r + 1) * n + r * n], stages[2])
    ctx.close()


@pytest.mark.parametrize("n,qs,psis", [(4096, [68719403009, 68719230977, 137438822401], [24250113, 29008497, 8625844]),
                                        (32768, P.Q55, P.PSI55)])
def test_keygen_and_encryption_sections_match_oracle(native, oracle, gpu, n, qs, psis):
    from ntt_cuda_amd import bfv
    import torch
    r = len(qs)
    prm = oracle.Params(n, qs, psis)
    ctx = native.NTTContext(n, qs, psis)
    # --- keygen (bfv_keygen.cuh:129-145): sk ternary-like small values, pk1 uniform
    sk = np.stack([np.where(oracle.splitmix(n, 50, 3) == 2, q - 1, oracle.splitmix(n, 50, 3)).astype(np.uint64) for q in qs])
    pk = np.zeros((2, r, n), dtype=np.uint64)
    pk[1] = oracle.synth_batch(n, r, qs, 60)
    d_sk, d_pk = native.to_device(sk), native.to_device(pk)
    bfv.keygen_ntt_a(ctx, d_sk, d_pk, n, r)
    torch.cuda.synchronize()
    SK = oracle.forward_batch(sk, prm).reshape(r, n)
    pk0 = oracle.inverse_batch(oracle.pointwise_batch(pk[1], SK, prm), prm).reshape(r, n)
    assert np.array_equal(native.to_host(d_sk).reshape(r, n), SK)
    got_pk = native.to_host(d_pk).reshape(2, r, n)
    assert np.array_equal(got_pk[0], pk0) and np.array_equal(got_pk[1], pk[1])
    bfv.keygen_ntt_b(ctx, d_pk, r)
    assert np.array_equal(native.to_host(d_pk).reshape(2, r, n)[0], oracle.forward_batch(pk0, prm).reshape(r, n))
    # --- encryption (bfv_encryption.cuh:268-271): c = [u | u], public key in the NTT domain
    u = np.stack([oracle.splitmix(n, 70, 3).astype(np.uint64) % q for q in qs])
    c = np.concatenate([u, u]).reshape(2 * r, n)
    pkhat = oracle.synth_batch(n, 2 * r, qs, 80)
    d_c, d_pkhat = native.to_device(c), native.to_device(pkhat)
    bfv.encryption_ntt(ctx, d_c, d_pkhat, r)
    want = oracle.inverse_batch(oracle.pointwise_batch(oracle.forward_batch(c, prm), pkhat, prm), prm).reshape(2 * r, n)
    assert np.array_equal(native.to_host(d_c).reshape(2 * r, n), want)
    ctx.close()
