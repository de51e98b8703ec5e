#!/usr/bin/env python3
"""Per-kernel digest of the gfx950 code in a hipcc object file or shared library ("the code object of every
shipped instantiation is byte-identical before and after").

usage: python tools/codeobj_digest.py OBJECT [OBJECT ...]      -> one line per kernel: sha256 of its instruction stream, size, name
       python tools/codeobj_digest.py --diff OLD NEW           -> kernels whose code differs / appeared / disappeared (exit 1 if any differ)
       python tools/codeobj_digest.py --loose --diff OLD NEW   -> the same comparison modulo scalar-register allocation (see loosen)

The instruction stream is the disassembly (llvm-objdump -d of the unbundled gfx950 code object) with addresses and encodings
stripped: symbol-relative branch targets stay, so two kernels with the same digest execute the same instructions.
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path, tmp):
    """the gfx950 ELF(s) inside a host object / shared library (clang offload bundle), or the file itself if it is one"""
    out = os.path.join(tmp, os.path.basename(path) + ".co")
    for kind in ("o", "a"):
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=" + kind, "--input=" + path, "--output=" + out,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(out) and os.path.getsize(out) > 0:
            return [out]
    # shared library: the fat binary sits in .hip_fatbin
    fb = os.path.join(tmp, os.path.basename(path) + ".fatbin")
    r = subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fb], capture_output=True, text=True)
    if r.returncode == 0 and os.path.exists(fb) and os.path.getsize(fb) > 0:
        data = open(fb, "rb").read()
        outs, pos, k = [], 0, 0
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        while True:
            i = data.find(magic, pos)
            if i < 0:
                break
            j = data.find(magic, i + 1)
            chunk = data[i:j if j > 0 else len(data)]
            cf = os.path.join(tmp, "%s.%d.bundle" % (os.path.basename(path), k))
            open(cf, "wb").write(chunk)
            co = cf + ".co"
            r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + cf, "--output=" + co,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
            if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
                outs.append(co)
            pos, k = i + 1, k + 1
        return outs
    return [path]


LOOSE = False


def loosen(line):
    """--loose: scalar register numbers, scalar-only bookkeeping and branch distances dropped -- two kernels with the same loose digest run
    the same vector / memory / LDS instruction stream on the same vector registers and differ at most in SGPR allocation"""
    op = line.split()[0] if line.split() else ""
    if op.startswith(("s_mov", "s_movk", "s_nop", "s_add", "s_addc", "s_sub", "s_lshl", "s_lshr", "s_not", "s_mul", "s_and", "s_or", "s_cselect", "s_cmp", "s_bitcmp", "s_load", "s_cbranch", "s_branch")):
        return op.split("_e")[0] if op.startswith(("s_cbranch", "s_branch", "s_load")) else None
    return re.sub(r"\bs\[\d+:\d+\]|\bs\d+\b|\bvcc\b|\bttmp\d+", "S", line)


def digests(path):
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(path, tmp):
            dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", co], capture_output=True, text=True).stdout
            name, body = None, []
            def flush():
                if name and body and not name.startswith("__hip_cuid"):
                    h = hashlib.sha256("\n".join(body).encode()).hexdigest()
                    res[name] = (h, len(body))
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]* ?<(.+)>:$", line.strip())
                if m:
                    flush()
                    name, body = m.group(1), []
                elif name is not None and line.strip():
                    l = re.sub(r"\s*//.*$", "", line).strip()
                    if LOOSE:
                        l = loosen(l)
                        if l is None:
                            continue
                    body.append(l)
            flush()
    return res


def named_digests(path):
    """{demangled kernel name: sha256 of its instruction stream} of one object / library"""
    d = digests(path)
    names = subprocess.run(["c++filt"], input="\n".join(d), capture_output=True, text=True).stdout.splitlines()
    return {re.sub(r"\(.*", "", nm).replace("void ", "").replace("mi355ntt::", ""): h for (raw, (h, n)), nm in zip(d.items(), names)}


def main():
    global LOOSE
    args = sys.argv[1:]
    if args and args[0] == "--loose":
        LOOSE, args = True, args[1:]
    if args and args[0] == "--diff":
        old, new = {}, {}
        for p in args[1].split(","):
            old.update(digests(p))
        for p in args[2].split(","):
            new.update(digests(p))
        def by_name(d):
            names = subprocess.run(["c++filt"], input="\n".join(d), capture_output=True, text=True).stdout.splitlines()
            out = {}
            for (raw, v), nm in zip(d.items(), names):
                nm = re.sub(r"\(.*", "", nm).replace("void ", "").replace("mi355ntt::", "")
                # (round 6 appended a defaulted template argument CHECKED = false to k_forward15 / k_inverse15: the same kernels)
                nm = re.sub(r"^(k_forward15<\d, (?:true|false), \d), false>$", r"\1>", nm)
                nm = re.sub(r"^(k_inverse15<\d, (?:true|false)), false>$", r"\1>", nm)
                out[nm] = v
            return out
        old, new = by_name(old), by_name(new)
        bad = 0
        for raw in sorted(set(old) | set(new)):
            nm = raw
            if raw not in old:
                print("NEW      %6d  %s" % (new[raw][1], nm))
            elif raw not in new:
                print("GONE     %6d  %s" % (old[raw][1], nm))
            elif old[raw][0] != new[raw][0]:
                print("DIFFERS  %6d -> %6d  %s" % (old[raw][1], new[raw][1], nm))
                bad += 1
        print("%d kernels compared, %d differ" % (len(set(old) & set(new)), bad))
        sys.exit(1 if bad else 0)
    for p in args:
        d = digests(p)
        names = subprocess.run(["c++filt"], input="\n".join(d), capture_output=True, text=True).stdout.splitlines()
        for (raw, (h, n)), nm in zip(d.items(), names):
            print("%s %6d  %s" % (h[:16], n, re.sub(r"\(.*", "", nm).replace("void ", "").replace("mi355ntt::", "")))


if __name__ == "__main__":
    main()
