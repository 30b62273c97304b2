#!/usr/bin/env python3
"""Register / scratch budget of every kernel in libferromic_hip.so, read from the gfx950 code objects' metadata notes.

    tools/kernel_resources.py [--spills-only] [library.so]

Prints one line per kernel: vgpr (arch + acc), agpr, sgpr, scratch bytes (.private_segment_fixed_size), spilled VGPRs / SGPRs.
Exit code 1 when a kernel uses scratch memory or spills VGPRs (tests/test_abi_library.py runs this on every CPU pass: a spilling
instantiation is a performance bug that no parity test would notice)."""

from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(library: str):
    out = []
    with tempfile.TemporaryDirectory(prefix="fmh_co_") as tmp:
        local = os.path.join(tmp, "lib.so")
        os.symlink(os.path.abspath(library), local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" not in name:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, name)], check=True, capture_output=True, text=True).stdout
            for block in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                block = ".agpr_count:" + block
                rec = {}
                for key in ("agpr_count", "name", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "group_segment_fixed_size"):
                    m = re.search(r"\." + key + r":\s+(\S+)", block)
                    if m:
                        rec[key] = m.group(1) if key == "name" else int(m.group(1))
                if "name" in rec:
                    out.append(rec)
    return out


def demangle(names):
    try:
        res = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True, check=True)
        return res.stdout.splitlines()
    except (OSError, subprocess.CalledProcessError):
        return list(names)


def main() -> int:
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    spills_only = "--spills-only" in sys.argv
    library = args[0] if args else os.path.join(ROOT, "ferromic_amd", "lib", "libferromic_hip.so")
    ks = kernels_of(library)
    pretty = demangle([k["name"] for k in ks])
    bad = 0
    for k, name in zip(ks, pretty):
        # SGPRs spilled into VGPR lanes (sgpr_spill_count) cost a few v_writelane / v_readlane and no memory: not counted
        spilled = k.get("private_segment_fixed_size", 0) or k.get("vgpr_spill_count", 0)
        bad += 1 if spilled else 0
        if spills_only and not spilled:
            continue
        print(f"vgpr {k.get('vgpr_count', 0):3d} agpr {k.get('agpr_count', 0):3d} sgpr {k.get('sgpr_count', 0):3d} scratch {k.get('private_segment_fixed_size', 0):5d} B "
              f"spill v{k.get('vgpr_spill_count', 0)} s{k.get('sgpr_spill_count', 0)} lds {k.get('group_segment_fixed_size', 0):6d}  {name}")
    print(f"{len(ks)} kernels, {bad} with scratch memory or spilled VGPRs")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
