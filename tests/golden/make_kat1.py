#!/usr/bin/env python3
"""Extract the reference's only known-answer vectors for the NTT hot path into a data fixture.

Source (read in THIS container only; /root/reference does not exist on the GPU box):
  /root/reference/BFV_Scheme/decryption_test.cu:348  unsigned long long c_host[24576]  (ciphertext c0|pad|c1|pad)
  /root/reference/BFV_Scheme/decryption_test.cu:355  unsigned long long sk_host[8192]  (secret key, NTT domain)
  parameters :26,27,47,48,91 (n, t, q_array, psi_roots, gamma); expected plaintext m[i] = i % 10 (:230-232)

Output: tests/golden/kat1_decryption_n4096.npz  (numbers only: the two arrays + the scalar parameters)

Usage: python tests/golden/make_kat1.py
"""
import os
import re
import hashlib
import numpy as np

SRC = "/root/reference/BFV_Scheme/decryption_test.cu"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat1_decryption_n4096.npz")


def grab(text, name):
    m = re.search(r"^\s*unsigned long long %s\[\]\s*=\s*\{\s*(\d[^}]*)\}" % name, text, re.M)
    return np.array([int(x) for x in m.group(1).replace("\n", " ").split(",")], dtype=np.uint64)


def main():
    text = open(SRC).read()
    c = grab(text, "c_host")
    sk = grab(text, "sk_host")
    assert c.size == 24576 and sk.size == 8192, (c.size, sk.size)
    np.savez_compressed(
        OUT,
        c_host=c,
        sk_host=sk,
        n=np.uint64(4096),
        t=np.uint64(1024),
        gamma=np.uint64(2305843009213683713),
        q=np.array([68719403009, 68719230977, 137438822401], dtype=np.uint64),
        psi=np.array([24250113, 29008497, 8625844], dtype=np.uint64),
    )
    print("c_host  sha256", hashlib.sha256(c.astype("<u8").tobytes()).hexdigest())
    print("sk_host sha256", hashlib.sha256(sk.astype("<u8").tobytes()).hexdigest())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
