#!/usr/bin/env python3
"""Per-kernel table for the BFV profile (tools/prof_driver_bfv.py under rocprofv3 --kernel-trace --stats): mean us per launch,
algorithmic bytes per launch (the words each kernel must read and write once), TB/s and the fraction of 6.29 TB/s (what the
n = 2^15 access pattern alone reaches on this chip, profiles/r01_memory_system_experiments.txt).
    python3 tools/prof_summary_bfv.py OUT R [batch=64]      (OUT/trace/**/_kernel_stats.csv of ONE prime set; R primes incl. the special one)"""
import csv
import glob
import os
import sys

out, R = sys.argv[1], int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n, r = 32768, R - 1
W = 8
# algorithmic bytes per launch: (batched launch, single-ciphertext launch); None = not HBM-streaming (listed with time only)
alg = {
    "k_encrypt_tail": lambda c: c * n * W * (2 * 3 * R + 1),            # per column and half: c, e read, c written (R words each); m once
    "k_decrypt_scale": lambda c: c * n * W * 3 * r,                      # c0, c1 read, c1 written on r polynomials
    "k_decrypt_round": lambda c: c * n * W * (r + 3),                    # r words read, 3 written per column
    "k_add_negate": lambda c: c * n * W * 3 * R,
    "k_keygen_pk0": lambda c: c * n * W * 4 * R,
    "k_sample_keygen": lambda c: c * n * (1 + 4 + 8 * R + 3 * W * R),    # byte, word, R uniform words read; 3 R words written
    "k_sample_encrypt": lambda c: c * n * (1 + 4 + 4 + 4 * W * R),
    "k_salsa20_keystream": None,
    "k_polymul15": None, "k_forward15": None, "k_inverse15": None,
}
rows = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        rows.append((row["Name"], int(row["Calls"]), float(row["AverageNs"]), float(row["TotalDurationNs"]), float(row["MinNs"]), float(row["MaxNs"])))
# per-launch durations from the kernel trace, to separate the batched launches (64 ciphertexts) from the single-ciphertext ones
per = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        nm = row["Kernel_Name"]
        gz = int(row.get("Grid_Size_Z", row.get("Grid_Size_z", "1")) or 1)
        wz = int(row.get("Workgroup_Size_Z", row.get("Workgroup_Size_z", "1")) or 1)
        key = (nm, gz // max(wz, 1), int(row.get("Grid_Size_X", row.get("Grid_Size", "0")) or 0))
        per.setdefault(key, []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
print("%-64s %8s %6s %10s %12s %8s %8s" % ("kernel (grid z = ciphertexts; grid x in threads)", "calls", "z", "mean us", "alg bytes", "TB/s", "of 6.29"))
for (nm, z, gx), ts in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    short = nm.replace("(anonymous namespace)::", "").replace("mi355ntt::", "").split("(")[0]
    base = short.split("<")[0].replace("void ", "")
    mean = sum(ts) / len(ts)
    a = alg.get(base)
    if a is not None:
        by = a(z)
        print("%-64s %8d %6d %10.2f %12d %8.2f %8.2f" % (short[:64], len(ts), z, mean * 1e-3, by, by / mean * 1e-3, by / mean * 1e-3 / 6.29))
    else:
        print("%-64s %8d %6d %10.2f %12s %8s %8s   (grid x %d)" % (short[:64], len(ts), z, mean * 1e-3, "-", "-", "-", gx))
