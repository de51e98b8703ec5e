"""Samplers of the BFV drivers: Salsa20/20 keystream and the conversion kernels."""
import struct

import numpy as np
import pytest

import params as P

ECRYPT_KEY = bytes([0x80] + [0] * 31)        # ECRYPT Salsa20/20, 256-bit key, set 1 vector 0, IV = 0: stream[0..63]
ECRYPT_STREAM0 = bytes.fromhex("E3BE8FDD8BECA2E3EA8EF9475B29A6E7003951E1097A5C38D23B7A5FAD9F6844"
                               "B22C97559E2723C7CBBD3FE4FC8D9A0744652A83E72A9C461876AF4D7EF1A117")
DEFAULT_KEY = bytes([1] * 32)                # generate_random_default, distributions.cuh:236
OTHER_KEY = bytes([77] * 32)                 # generate_random, distributions.cuh:206


def salsa20_block_py(key, nonce, ctr):
    """independent Python statement of the Salsa20/20 block function (Bernstein's specification)"""
    rotl = lambda x, c: ((x << c) & 0xffffffff) | (x >> (32 - c))
    le = lambda b, i: struct.unpack_from("<I", b, i)[0]
    sig = b"expand 32-byte k"
    j = [le(sig, 0), le(key, 0), le(key, 4), le(key, 8), le(key, 12), le(sig, 4), nonce & 0xffffffff, nonce >> 32, ctr & 0xffffffff,
         ctr >> 32, le(sig, 8), le(key, 16), le(key, 20), le(key, 24), le(key, 28), le(sig, 12)]
    x = j[:]

    def qr(a, b, c, d):
        x[b] ^= rotl((x[a] + x[d]) & 0xffffffff, 7)
        x[c] ^= rotl((x[b] + x[a]) & 0xffffffff, 9)
        x[d] ^= rotl((x[c] + x[b]) & 0xffffffff, 13)
        x[a] ^= rotl((x[d] + x[c]) & 0xffffffff, 18)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(5, 9, 13, 1); qr(10, 14, 2, 6); qr(15, 3, 7, 11)
        qr(0, 1, 2, 3); qr(5, 6, 7, 4); qr(10, 11, 8, 9); qr(15, 12, 13, 14)
    return b"".join(struct.pack("<I", (x[i] + j[i]) & 0xffffffff) for i in range(16))


def test_salsa20_oracle_matches_known_answer_and_specification(oracle):
    assert salsa20_block_py(ECRYPT_KEY, 0, 0) == ECRYPT_STREAM0
    assert bytes(oracle.salsa20_keystream(64, ECRYPT_KEY, 0)) == ECRYPT_STREAM0
    for key, nonce in ((DEFAULT_KEY, 0), (OTHER_KEY, 0), (DEFAULT_KEY, 0x0123456789abcdef)):
        ks = bytes(oracle.salsa20_keystream(64 * 40, key, nonce))
        for blk in (0, 1, 7, 39):
            assert ks[64 * blk: 64 * blk + 64] == salsa20_block_py(key, nonce, blk)
    assert len(oracle.salsa20_keystream(100, DEFAULT_KEY)) == 64          # NBLKS = n / 64: the tail is not generated


def test_integer_samplers_oracle(oracle):
    n, qs = 4096, P.Q55[:3]
    rng = np.random.default_rng(1)
    by = np.concatenate([np.arange(256, dtype=np.uint8), rng.integers(0, 256, size=n - 256, dtype=np.uint8)])
    tern = oracle.sample_xq("ternary", by, n, qs)
    for i, q in enumerate(qs):
        b = by.astype(np.int64) // 85 - 1                     # int(float(byte) / 85.0f) - 1; byte 255 gives 2
        assert np.array_equal(tern[i], np.where(b < 0, q - 1, b).astype(np.uint64))
    assert tern[0][255] == 2 and tern[0][0] == qs[0] - 1 and tern[0][85] == 0 and tern[0][170] == 1
    words = rng.integers(0, 1 << 64, size=len(qs) * n, dtype=np.uint64)
    words[:4] = [0, (1 << 64) - 1, 1 << 63, 12345]
    uni = oracle.sample_xq("uniform", words.view(np.uint8), n, qs)
    for i, q in enumerate(qs):
        want = [int(float(int(w)) / float(2 ** 64) * float(q - 1)) for w in words[i * n: i * n + 64]]
        assert [int(x) for x in uni[i][:64]] == want
        assert uni[i].max() < q


def test_gaussian_sampler_oracle_statistics(oracle):
    n, qs = 1 << 16, [P.Q55[0], P.Q60[0]]
    rng = np.random.default_rng(2)
    w = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    g = oracle.sample_xq("gaussian", w.view(np.uint8), n, qs)
    signed = np.where(g[0] > qs[0] // 2, g[0].astype(np.int64) - qs[0], g[0].astype(np.int64))
    assert np.array_equal(np.where(g[1] > qs[1] // 2, g[1].astype(np.int64) - qs[1], g[1].astype(np.int64)), signed)   # same words for every prime
    assert abs(signed.mean()) < 0.05 and 2.7 < signed.std() < 3.0 and np.abs(signed).max() <= 19
    # truncation toward zero: P(0) = P(|x| < 1) for x ~ N(0, 3.2^2) = 0.2453
    assert abs((signed == 0).mean() - 0.2453) < 0.01


@pytest.mark.gpu
def test_keystream_on_gpu(native, oracle, gpu):
    import torch
    for key, nonce, nbytes in ((ECRYPT_KEY, 0, 64), (DEFAULT_KEY, 0, 1 << 20), (OTHER_KEY, 5, 64 * 1000 + 17), (DEFAULT_KEY, 1 << 40, 4096)):
        out = torch.full((nbytes,), 0xAA, dtype=torch.uint8, device=gpu)
        native.salsa20_keystream(out, key, nonce)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        want = oracle.salsa20_keystream(nbytes, key, nonce)
        assert np.array_equal(got[: want.size], want)
        assert (got[want.size:] == 0xAA).all()                 # only whole blocks are written
    assert bytes(got[:0]) == b""


@pytest.mark.gpu
def test_samplers_and_complete_drivers_on_gpu(native, oracle, gpu):
    """keygen_rns -> encryption_rns -> decryption_rns from the keystream, as demo.cu:275-311 runs them; the integer
    samplers word for word against the oracle, the Gaussian one by mismatch rate."""
    import torch
    from ntt_cuda_amd import bfv
    n = 32768
    qs, psis = P.Q60 + [P.Q60_SPECIAL], P.PSI60 + [P.PSI60_SPECIAL]
    R, t = len(qs), 1024
    ctx = bfv.BFVContext(n, qs, psis, t, P.GAMMA61)
    assert ctx.keygen_random_bytes == 9 * R * n + 4 * n and ctx.encrypt_random_bytes == 9 * n
    z64 = lambda *shape: torch.zeros(*shape, dtype=torch.int64, device=gpu)
    rnd = torch.zeros(ctx.keygen_random_bytes, dtype=torch.uint8, device=gpu)
    sk, pk, tmp = z64(R, n), z64(2, R, n), z64(R, n)
    native.salsa20_keystream(rnd, DEFAULT_KEY, 0)
    ctx.sample_keygen(rnd, sk, pk, tmp)
    torch.cuda.synchronize()
    ks = oracle.salsa20_keystream(ctx.keygen_random_bytes, DEFAULT_KEY, 0)
    assert np.array_equal(rnd.cpu().numpy()[: ks.size], ks)
    assert np.array_equal(native.to_host(sk), oracle.sample_xq("ternary", ks[:n], n, qs))
    assert np.array_equal(native.to_host(pk)[1], oracle.sample_xq("uniform", ks[n: n + 8 * R * n], n, qs))
    want_e = oracle.sample_xq("gaussian", ks[n + 8 * R * n: n + 8 * R * n + 4 * n], n, qs)
    got_e = native.to_host(tmp)
    assert (got_e != want_e).mean() < 2e-3                     # normcdfinvf is not specified to the ulp
    se = np.where(got_e[0] > qs[0] // 2, got_e[0].astype(np.int64) - qs[0], got_e[0].astype(np.int64))
    assert abs(se.mean()) < 0.08 and 2.7 < se.std() < 3.0 and np.abs(se).max() <= 19

    # complete drivers; nonce 0 for both = the reference's behaviour (u equals the secret key's ternary sample), then fresh nonces
    m = torch.randint(0, t, (n,), dtype=torch.int64, device=gpu)
    for nonce_k, nonce_e in ((0, 0), (1, 2)):
        sk, pk, tmp = z64(R, n), z64(2, R, n), z64(R, n)
        ctx.keygen_rns(rnd, sk, pk, tmp, nonce=nonce_k)
        c, e = z64(2, R, n), z64(2, R, n)
        rnd_e = torch.zeros(ctx.encrypt_random_bytes, dtype=torch.uint8, device=gpu)
        ctx.encryption_rns(c, pk, rnd_e, e, m, nonce=nonce_e)
        got = ctx.decrypt(c, sk)
        torch.cuda.synchronize()
        assert torch.equal(got, m)
    ctx.close()
