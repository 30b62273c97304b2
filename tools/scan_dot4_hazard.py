#!/usr/bin/env python3
"""Static guard for a gfx950 hazard the fuzz tests tripped over: the result of a v_dot4 read by a different VALU op
(in this code base: the DPP ops of the 16-lane row reductions) needs 3 wait states, and hipcc (ROCm 7.2) left only
2 when the last dot4 of a loop and the reduction sat in different basic blocks - the reduction then missed that dot4.

Compiles the device code of every ferromic_amd/csrc/*.hip to assembly (hipcc, no GPU needed) and walks every v_dot4:
following fall-through paths and branch targets (s_nop N counts N + 1 wait states, every other instruction 1), a read of the result by
anything but the accumulator operand of another v_dot4 within fewer than 3 wait states is reported.
Exit code 1 when anything is reported.  Usage: tools/scan_dot4_hazard.py [existing.s]
"""

from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEEDED = 3


def device_asm(path: str, source: str) -> None:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-fno-fast-math",
           "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", path, source]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def all_units_asm(tmp: str) -> list:
    """One .s per translation unit of the library, compiled a few at a time."""
    import glob
    from concurrent.futures import ThreadPoolExecutor

    sources = sorted(glob.glob(os.path.join(ROOT, "ferromic_amd", "csrc", "*.hip")))
    outs = [os.path.join(tmp, os.path.basename(src)[:-4] + ".s") for src in sources]
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as pool:
        list(pool.map(lambda so: device_asm(so[1], so[0]), zip(sources, outs)))
    return outs


def regs(tok: str) -> set:
    tok = tok.strip().split(" ")[0]
    m = re.match(r"^v(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def scan(path: str):
    func, ins, labels = None, [], {}
    for no, line in enumerate(open(path), 1):
        s = line.strip()
        if not s or s.startswith(";"):
            continue
        m = re.match(r"^(_Z\w+):", s)
        if m:
            func = m.group(1)
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = len(ins)  # the next instruction
            continue
        if s.startswith(".") or s.endswith(":"):
            continue  # directives
        ins.append((no, func, s.split(";")[0].strip()))

    def walk(start, fn, dest, waits, findings, seen):
        """Follows fall-through AND branch targets from instruction index `start` until NEEDED wait states have passed."""
        j = start
        while j < len(ins) and waits < NEEDED and ins[j][1] == fn:
            if (j, waits) in seen:
                return
            seen.add((j, waits))
            t = ins[j][2]
            parts = t.split(None, 1)
            op = parts[0]
            if op == "s_nop":
                waits += int(parts[1]) + 1
                j += 1
                continue
            if op == "s_endpgm":
                return
            toks = parts[1].split(",") if len(parts) > 1 else []
            has_dest = not (op.startswith(("global_store", "buffer_store", "flat_store", "ds_write", "v_cmp", "s_")))
            srcs = set()
            for idx, tk in enumerate(toks):
                if idx == 0 and has_dest:
                    continue
                srcs |= regs(tk)
            if srcs & dest:
                accumulate = op.startswith("v_dot4") and len(toks) == 4 and regs(toks[3]) & dest and not ((regs(toks[1]) | regs(toks[2])) & dest)
                if not accumulate:
                    findings.append((fn, ins[j][0], waits, t))
                return
            if has_dest and toks and regs(toks[0]) & dest:
                return  # overwritten
            if op.startswith(("s_cbranch", "s_branch")) and toks and toks[0].strip() in labels:
                walk(labels[toks[0].strip()], fn, dest, waits + 1, findings, seen)
                if op == "s_branch":
                    return
            waits += 1
            j += 1

    findings, dots = [], 0
    for k, (no, fn, text) in enumerate(ins):
        if not text.startswith("v_dot4"):
            continue
        dots += 1
        dest = regs(text.split(None, 1)[1].split(",")[0])
        walk(k + 1, fn, dest, 0, findings, set())
    return dots, sorted(set(findings))


def main() -> int:
    if len(sys.argv) > 1:
        paths, tmp = sys.argv[1:], None
    else:
        tmp = tempfile.mkdtemp(prefix="fmh_asm_")
        paths = all_units_asm(tmp)
    dots, findings = 0, []
    for path in paths:
        d, f = scan(path)
        dots += d
        findings += f
    for fn, no, waits, text in findings:
        print(f"{fn}: line {no}: dot4 result read after {waits} wait state(s) by `{text}`")
    print(f"{dots} v_dot4 instructions scanned, {len(findings)} hazard(s)")
    if tmp:
        import shutil

        shutil.rmtree(tmp, ignore_errors=True)
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
