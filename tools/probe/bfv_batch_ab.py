#!/usr/bin/env python3
"""probe (round 5): bench.py's BFV legs (single ciphertext and 64 per call) on the shipped library or another build (MI355NTT_LIB)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd"))
import torch
import ntt_cuda_amd as ntt
if os.environ.get("MI355NTT_LIB"):
    ntt.LIB_PATH = os.environ["MI355NTT_LIB"]
import bench
dev = torch.device("cuda", 0)
r = bench.bfv_round_trip(torch, ntt, 32768, dev, False, bench.Q60 + [bench.Q60_SPECIAL], bench.PSI60 + [bench.PSI60_SPECIAL], "4 + 1")
b = r["batch64"]
print("# lib = %s : encrypt %.1f us  decrypt %.1f us per 64 ciphertexts;  one ciphertext: keygen %.1f encrypt %.1f decrypt %.1f us" % (
    os.path.basename(os.environ.get("MI355NTT_LIB") or "shipped"), b["encrypt_us_per_call"], b["decrypt_us_per_call"], r["keygen_us"], r["encrypt_us"], r["decrypt_us"]))
