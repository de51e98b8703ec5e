#!/usr/bin/env python3
"""Fit the package-power model of the n = 2^15 workload to a tools/power_model run (round 5).

    P(u, f, B) = S + A u (f / f0)^alpha + M B^beta              [W]
u = VALU utilisation of the butterfly stream (1 = every issue slot), f = shader clock, f0 = 2.385 GHz, B = HBM traffic in TB/s
(1 read : 1 write).  S, A from the points without memory traffic (the clock stays at f0 there), M, beta from the points without
butterflies, alpha from the points the firmware throttled (PVIOL > 50 %: the package sits on its power limit, so P = the measured
power and f is what the SMU chose).  Then the NTT kernels' own term: E_L joules per fwd+inv pair for everything the synthetic load
does not have (three trips of every polynomial through LDS, the twiddle stream through L1 / L2, scalar work) from a measured
operating point (pairs/s, clock, power), and the throughput the cap admits at a given VALU utilisation:
    cap = S + A u (f/f0)^alpha + M (bytes_per_pair T)^beta + E_L T,   u = T c / (CUs f)        (c = VALU issue cycles per pair and CU)
    python3 tools/power_fit.py gpurun_out/r05b.txt [pairs_per_s clock_ghz watts]  -> profiles/r05_power_model_fit.json"""
import json
import math
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F0 = 2.385
pts = []
for l in open(sys.argv[1]):
    m = re.match(r"^(grid|memonly)\s+busy\s+(\d+) vsleep\s+\d+ msleep\s+(\d+) \| VALU util ([0-9.]+)\s+([0-9.]+) TB/s \|\s+([0-9.]+) W .*in-kernel ([0-9.]+) GHz.*PVIOL\s+([0-9.]+) %", l)
    if m:
        pts.append(dict(kind=m.group(1), duty=int(m.group(2)), msleep=int(m.group(3)), u=float(m.group(4)), bw=float(m.group(5)), w=float(m.group(6)),
                        f=float(m.group(7)), pviol=float(m.group(8))))
assert pts, "no power_model lines found"
valu = [p for p in pts if p["kind"] == "grid" and p["bw"] == 0 and p["pviol"] < 5]
mem = [p for p in pts if p["kind"] == "memonly"]
# S, A: least squares of w = S + A u over the unthrottled VALU-only points
n = len(valu)
su, sw = sum(p["u"] for p in valu), sum(p["w"] for p in valu)
suu, suw = sum(p["u"] ** 2 for p in valu), sum(p["u"] * p["w"] for p in valu)
A = (n * suw - su * sw) / (n * suu - su * su)
S = (sw - A * su) / n
# M, beta: w - S0 = M B^beta over the memory-only points (S0 = the memory-only point without traffic)
S0 = min(p["w"] for p in mem if p["bw"] == 0)
mm = [p for p in mem if p["bw"] > 0]
lx, ly = [math.log(p["bw"]) for p in mm], [math.log(p["w"] - S0) for p in mm]
k = len(mm)
beta = (k * sum(x * y for x, y in zip(lx, ly)) - sum(lx) * sum(ly)) / (k * sum(x * x for x in lx) - sum(lx) ** 2)
M = math.exp((sum(ly) - beta * sum(lx)) / k)
# alpha: throttled points
thr = [p for p in pts if p["kind"] == "grid" and p["pviol"] > 50 and p["u"] > 0.3]
al = []
for p in thr:
    dyn = p["w"] - S - M * p["bw"] ** beta
    x = dyn / (A * p["u"])
    if 0 < x < 1 and p["f"] < F0 * 0.985:
        al.append(math.log(x) / math.log(p["f"] / F0))
al.sort()
alpha = al[len(al) // 2]
res = [abs(S + A * p["u"] * (p["f"] / F0) ** alpha + (M * p["bw"] ** beta if p["bw"] > 0 else 0) - p["w"]) for p in pts if p["kind"] == "grid"]
fit = {"model": "P = S + A u (f/f0)^alpha + M B^beta  [W; u = VALU utilisation of the butterfly stream, f in GHz, B in TB/s of HBM traffic]",
       "S_w": S, "A_w": A, "f0_ghz": F0, "alpha": alpha, "M_w": M, "beta": beta, "alpha_samples": len(al), "alpha_range": [al[0], al[-1]],
       "mean_abs_residual_w": sum(res) / len(res), "max_abs_residual_w": max(res), "points": len(pts),
       "source": os.path.relpath(sys.argv[1], ROOT) if os.path.isabs(sys.argv[1]) else sys.argv[1]}
# the kernels' own operating point -> E_L and the throughput the cap admits at a given utilisation
valu_json = json.load(open(os.path.join(ROOT, "profiles", "valu_ceiling_r05.json")))
cyc_pair = valu_json["k_forward15"]["cycles_per_polynomial_per_cu"] + valu_json["k_inverse15"]["cycles_per_polynomial_per_cu"]
CUS, BYTES_PAIR_TB = 256, 2 * 524288 * 1.03 / 1e12           # measured traffic 1.03 x algorithmic (profiles/traffic_r04.json)


def solve(T_meas, f_meas, w_meas, cap=None):
    u = T_meas * cyc_pair / (CUS * f_meas * 1e9)
    E_L = (w_meas - S - A * u * (f_meas / F0) ** alpha - M * (BYTES_PAIR_TB * T_meas) ** beta) / T_meas
    cap = cap or w_meas
    out = {"measured": {"pairs_per_s": T_meas, "clock_ghz": f_meas, "package_w": w_meas, "valu_utilisation": u}, "E_L_joule_per_pair": E_L, "cap_w": cap,
           "cycles_per_pair_and_cu": cyc_pair, "throughput_the_cap_admits": {}}
    for ut in (0.70, 0.75, 0.80, 0.85, 0.90, 1.00):
        lo, hi = 1e6, 8e6
        for _ in range(60):
            T = 0.5 * (lo + hi)
            f = min(T * cyc_pair / (CUS * ut * 1e9), 2.4)
            uu = T * cyc_pair / (CUS * f * 1e9)
            P = S + A * uu * (f / F0) ** alpha + M * (BYTES_PAIR_TB * T) ** beta + E_L * T
            if P > cap:
                hi = T
            else:
                lo = T
        out["throughput_the_cap_admits"]["%.2f" % ut] = {"pairs_per_s": lo, "clock_ghz": min(lo * cyc_pair / (CUS * ut * 1e9), 2.4)}
    return out


if len(sys.argv) > 4:
    fit["n32768_60bit"] = solve(float(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5]) if len(sys.argv) > 5 else None)
path = os.path.join(ROOT, "profiles", "r05_power_model_fit.json")
json.dump(fit, open(path, "w"), indent=1)
print(json.dumps(fit, indent=1))
