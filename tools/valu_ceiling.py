#!/usr/bin/env python3
"""Secondary (VALU) ceiling of the shipped n = 2^15 kernels -> profiles/valu_ceiling_r06.json (read by bench.py).

Compiles ntt-cuda_amd/csrc/kernels_fast_n15.hip to gfx950 assembly, sums the measured steady-state issue cost
(tools/ubench_issue.hip, profiles/r02_ubench_issue_costs.txt) over the instructions of each kernel's polynomial loop
and converts to transforms/s: one polynomial per CU at a time, 4 waves per SIMD, 256 CUs at a NOMINAL 2.35 GHz.  bench.py
re-prices `cycles_per_polynomial_per_cu` at the shader clock its own timed launches ran at (sampled inside the kernels:
mi355ntt_ctx_last_kernel_clock_mhz) and at the device's CU count.  Run here (no GPU needed)."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_cost  # noqa: E402  (prints nothing on import: guarded below)

CLOCK_HZ, CUS = 2.35e9, 256


def loop_cost(lines, name_part):
    kern, cur = {}, None
    for l in lines:
        m = re.match(r'^(_Z\w+):', l)
        if m:
            cur = m.group(1)
            kern[cur] = []
            continue
        if cur is not None:
            kern[cur].append(l)
            if 's_endpgm' in l:
                cur = None
    for name, body in kern.items():
        if name_part not in name:
            continue
        labels, best = {}, None
        for i, l in enumerate(body):
            m = re.match(r'^(\.LBB\d+_\d+):', l)
            if m:
                labels[m.group(1)] = i
            m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels:
                seg = [x.split()[0] for x in body[labels[m.group(1)]:i] if x.strip() and not x.strip().startswith((';', '.'))]
                if best is None or len(seg) > len(best):
                    best = seg
        tot, nv = 0.0, 0
        for op in best:
            if op.startswith('v_'):
                nv += 1
                k = isa_cost.cost(op)
                tot += 4.1 if k is None else k
        return tot, nv, len(best)
    raise SystemExit("kernel %s not found" % name_part)


KERNELS = (("k_forward15", "k_forward15ILi4ELb1ELi0"), ("k_inverse15", "k_inverse15ILi4ELb1"), ("k_polymul15", "k_polymul15ILi4ELb1"),
           # kernel class 0 (round 6): the longest loop of the two-pass kernels is the literal pass
           ("k_forward15_lit", "k_forward15_litILi15"), ("k_inverse15_lit", "k_inverse15_litILi15"))
PROFILE = os.path.join(ROOT, "profiles", "valu_ceiling_r06.json")


def ceiling_from_asm(lines):
    res = {}
    for k, sym in KERNELS:
        cyc, nv, ni = loop_cost(lines, sym)
        per_poly = 4 * cyc                                    # 4 waves share a SIMD
        res[k] = {"valu_issue_cycles_per_wave": cyc, "valu_instructions_per_wave": nv, "loop_instructions": ni,
                  "cycles_per_polynomial_per_cu": per_poly, "transforms_per_s": CUS * CLOCK_HZ / per_poly}
    res["pairs_per_s"] = 1.0 / (1.0 / res["k_forward15"]["transforms_per_s"] + 1.0 / res["k_inverse15"]["transforms_per_s"])
    res["literal_pairs_per_s"] = 1.0 / (1.0 / res["k_forward15_lit"]["transforms_per_s"] + 1.0 / res["k_inverse15_lit"]["transforms_per_s"])
    res["source"] = ("tools/valu_ceiling.py: measured steady-state issue cycles per instruction (profiles/r02_ubench_issue_costs.txt) summed "
                     "over the polynomial loop of the shipped <HL 4, near-2^k> kernels, 4 waves per SIMD; transforms_per_s here at a nominal %d CUs x %.2f GHz "
                     "(bench.py re-prices cycles_per_polynomial_per_cu at the clock sampled inside its timed launches); "
                     "the bare butterfly stream measures 11.5 M transforms/s sustained (profiles/r04_power_cap_and_overlap.txt).  "
                     "tests/test_abi_host.py recomputes these figures from the shipped sources and fails on drift: regenerate with "
                     "`python3 tools/valu_ceiling.py` in the same commit as any kernel change" % (CUS, CLOCK_HZ / 1e9))
    return res


def drift(res, ref):
    """differences between a fresh computation and the committed profile (empty = in step)"""
    bad = []
    for k, _ in KERNELS:
        for f in ("valu_instructions_per_wave", "loop_instructions"):
            if res[k][f] != ref.get(k, {}).get(f):
                bad.append("%s.%s: shipped sources %s, profile %s" % (k, f, res[k][f], ref.get(k, {}).get(f)))
        a, b = res[k]["cycles_per_polynomial_per_cu"], ref.get(k, {}).get("cycles_per_polynomial_per_cu", 0.0)
        if abs(a - b) > 1e-6 * max(a, b, 1.0):
            bad.append("%s.cycles_per_polynomial_per_cu: shipped sources %.1f, profile %.1f" % (k, a, b))
    return bad


def main():
    # usage: valu_ceiling.py [--asm FILE] [--check]   (--asm: an existing `hipcc -S --cuda-device-only` listing of kernels_fast_n15.hip;
    # --check: compare with the committed profile instead of rewriting it, exit status 1 on drift)
    args = sys.argv[1:]
    asm = args[args.index("--asm") + 1] if "--asm" in args else None
    if asm:
        lines = open(asm).read().split('\n')
    else:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "n15.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "ntt-cuda_amd", "csrc"),
                                   "-I", os.path.join(ROOT, "include"), "--cuda-device-only", "-S",
                                   os.path.join(ROOT, "ntt-cuda_amd", "csrc", "kernels_fast_n15.hip"), "-o", out], stderr=subprocess.DEVNULL)
            lines = open(out).read().split('\n')
    res = ceiling_from_asm(lines)
    if "--check" in args:
        bad = drift(res, json.load(open(PROFILE)))
        for b in bad:
            print(b)
        print("valu ceiling profile %s" % ("DRIFTED: regenerate profiles/valu_ceiling_r06.json" if bad else "in step with the shipped sources"))
        return 1 if bad else 0
    json.dump(res, open(PROFILE, "w"), indent=1)
    print(json.dumps(res, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
