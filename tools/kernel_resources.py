#!/usr/bin/env python3
"""Register / scratch usage of every kernel in one translation unit of the library (compiler remarks).

usage: python tools/kernel_resources.py ntt-cuda_amd/csrc/kernels_fast_n15.hip [name-filter] [-- extra hipcc flags]
Prints one line per kernel: demangled name, VGPRs, AGPRs, scratch bytes per lane, occupancy (waves per SIMD).
Exit status 1 if any kernel whose name matches --require-no-scratch uses scratch.
--asm-out PATH: the same compilation also leaves the gfx950 assembly at PATH (tools/valu_ceiling.py --asm reads it: one compile
of the translation unit serves the scratch check and the VALU-ceiling drift check of the CPU test suite).
"""
import re
import subprocess
import sys

def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    need = None
    if "--require-no-scratch" in args:
        k = args.index("--require-no-scratch")
        need = args[k + 1]
        del args[k:k + 2]
    asm_out = None
    if "--asm-out" in args:
        k = args.index("--asm-out")
        asm_out = args[k + 1]
        del args[k:k + 2]
    src = args[0]
    flt = args[1] if len(args) > 1 else ""
    out = ["-S", "--cuda-device-only", "-o", asm_out] if asm_out else ["-c", "-o", "/dev/null"]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", src] + out + ["-Rpass-analysis=kernel-resource-usage"] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], {}
    for line in err.splitlines():
        m = re.search(r"remark: (?:.*: )?\s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            if cur:
                rows.append(cur)
            cur = {"name": val}
        else:
            cur[key.split(" ")[0]] = val
    if cur:
        rows.append(cur)
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    bad = 0
    for r, nm in zip(rows, names):
        nm = re.sub(r"\(.*", "", nm).replace("void ", "").replace("mi355ntt::", "")
        if flt and flt not in nm:
            continue
        print("%-44s VGPRs %4s  AGPRs %3s  SGPRs %4s  scratch %4s B/lane  LDS %6s  occupancy %s" % (
            nm, r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("SGPRs", "?"), r.get("ScratchSize", "?"), r.get("LDS", "?"), r.get("Occupancy", "?")))
        if need and need in nm and r.get("ScratchSize", "0") != "0":
            bad += 1
    if bad:
        print("%d kernel(s) matching '%s' use scratch" % (bad, need))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
