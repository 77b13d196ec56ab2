#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of libsgx_hip.so, from the compiler's own resource remarks.

``make -C stylegan/pytorch_amd/csrc`` compiles with ``-Rpass-analysis=kernel-resource-usage`` and keeps the remarks in
``csrc/build/<file>.res``; this script turns them into one table (no GPU needed):

    python tools/kernel_resources.py > profiles/rNN_kernel_resources.tsv

Occupancy is waves per SIMD as the compiler derives it from the register budget (the LDS a launch asks for dynamically
can lower it further: the convolution kernels size their stages at launch time)."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "stylegan", "pytorch_amd", "csrc", "build")
FIELDS = [("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("TotalSGPRs", "sgpr"), ("ScratchSize [bytes/lane]", "scratch"),
          ("Occupancy [waves/SIMD]", "occupancy"), ("SGPRs Spill", "sgpr_spill"), ("VGPRs Spill", "vgpr_spill"),
          ("LDS Size [bytes/block]", "static_lds")]


def demangle(names):
    import shutil
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not filt:
        return names
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return out if len(out) == len(names) else names


def parse(path):
    """-> list of dicts, one per kernel, in file order"""
    kernels, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis=kernel-resource-usage\]", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = {"file": os.path.basename(path)[:-4] + ".hip", "mangled": body.split(":", 1)[1].strip()}
            kernels.append(cur)
            continue
        for label, key in FIELDS:
            if cur is not None and body.startswith(label + ":"):
                val = body[len(label) + 1:].strip()
                cur[key] = int(val) if val.lstrip("-").isdigit() else val
    return kernels


def collect():
    kernels = []
    for path in sorted(glob.glob(os.path.join(BUILD, "*.res"))):
        kernels += parse(path)
    for k, name in zip(kernels, demangle([k["mangled"] for k in kernels])):
        k["name"] = re.sub(r"^void ", "", name)
    return kernels


def main():
    kernels = collect()
    if not kernels:
        sys.exit("no csrc/build/*.res: run make -C stylegan/pytorch_amd/csrc first")
    cols = ["file", "name"] + [k for _, k in FIELDS]
    print("\t".join(cols))
    for k in kernels:
        print("\t".join(str(k.get(c, "")) for c in cols))


if __name__ == "__main__":
    main()
