#!/usr/bin/env python3
"""Scan the device assembly of a kernel file for one hazard the compiler does not see: an inline-asm vector-memory instruction
(the hidden LDS-DMA pieces of attention.hip: `global_load_lds_dword[x4] v, s[a:b]`) whose scalar base s[a:b] was written by a VALU
instruction (v_readlane / v_readfirstlane -- hipcc parks scalars in VGPR lanes under register pressure) fewer than five wait
states earlier.  gfx950 needs five; the hazard recogniser does not look inside inline asm (docs/findings/r05.md, "The backward
without the second recompute").  Usage:  python tools/isa_sgpr_hazard_scan.py [file.hip] [-DCARA_F16_OPERANDS]"""
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = next((a for a in sys.argv[1:] if a.endswith(".hip")), os.path.join(root, "cara_amd", "csrc", "attention.hip"))
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-Wno-inline-asm", "-fno-slp-vectorize",
                    *defs, "--cuda-device-only", "-S", src, "-o", out], check=True, cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
hits = total = 0
for i, line in enumerate(lines):
    m = re.search(r"global_(load_lds_dword(x4)?|load_dword\w*|store_\w+)\s+\S+,(?:\s+\S+,)?\s+s\[(\d+):(\d+)\]", line)
    if not m or ";;#ASMSTART" not in "\n".join(lines[max(0, i - 6):i]):
        continue
    total += 1
    a, b = int(m.group(3)), int(m.group(4))
    seen, j = 0, i - 1
    while j > 0 and seen < 8:
        t = lines[j].strip()
        if t and not t.startswith((";", ".")):
            seen += 1
            w = re.match(r"v_read(first)?lane_b32 s(\d+),", t)
            if w and int(w.group(2)) in (a, b):
                ws = 0
                for k in range(j + 1, i):
                    tt = lines[k].strip()
                    if tt and not tt.startswith((";", ".")):
                        n = re.match(r"s_nop (\d+)", tt)
                        ws += (int(n.group(1)) + 1) if n else 1
                if ws < 5:
                    hits += 1
                    print(f"line {i}: {ws} wait state(s) between `{t}` and `{line.strip()}`")
                break
        j -= 1
print(f"{os.path.basename(src)} {' '.join(defs)}: {total} inline-asm vector-memory instructions with a scalar base, {hits} closer than five wait states to a VALU write of it")
sys.exit(1 if hits else 0)
