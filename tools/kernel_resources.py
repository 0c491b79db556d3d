#!/usr/bin/env python3
"""Registers, scratch, LDS and occupancy of every kernel of the library, from hipcc's -Rpass-analysis=kernel-resource-usage
(compile only; runs without a GPU).  A kernel with a non-zero ScratchSize or a dynamic stack makes the runtime (re)allocate
the queue's scratch memory the first time a dispatch needs more wave slots than any dispatch before it.

    python tools/kernel_resources.py [file.hip ...]   ->  one line per kernel"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "model_matching_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
         "-Rpass-analysis=kernel-resource-usage"]
PATS = (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
        ("dyn_stack", r"Dynamic Stack: (\w+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"))


def kernels_of(path):
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", path, "-o", "/dev/null"], capture_output=True, text=True, cwd=CSRC, stdin=subprocess.DEVNULL)
    out, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"Name: (\S+) \[", line)
        if m:
            cur = {"name": m.group(1)}
            out.append(cur)
            continue
        for key, pat in PATS:
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = m.group(1)
    if not out:
        return out
    names = subprocess.run(["c++filt"] + [k["name"] for k in out], capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.splitlines()
    for k, n in zip(out, names):
        k["demangled"] = n
    return out


def main():
    files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    for f in files:
        for k in kernels_of(os.path.join(CSRC, f)):
            own = "stocs::" in k["demangled"].split("(")[0] and not k["demangled"].startswith("void rocprim")
            if own or k.get("scratch", "0") != "0" or k.get("dyn_stack") == "True":
                print("%-14s %-90s %s" % (f, k["demangled"].split("(")[0][:90], " ".join("%s=%s" % (x, k.get(x)) for x, _ in PATS)))


if __name__ == "__main__":
    main()
