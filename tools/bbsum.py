#!/usr/bin/env python3
"""Basic-block summary of one kernel's gfx950 ISA: tools/bbsum.py <file.hip> <kernel-name-regex> [extra hipcc flags]
(counts of matrix, LDS, global, scratch, barrier instructions per block; blocks under 8 instructions are skipped)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "rq-vae-recommender_amd", "csrc")
src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
out = f"/tmp/bbsum_{os.getpid()}.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-fno-slp-vectorize", "-munsafe-fp-atomics",
                f"-I{ROOT}/include", f"-I{C}", "--cuda-device-only", "-S", os.path.join(C, src), "-o", out] + extra,
               check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
os.remove(out)
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel", s, re.S | re.M):
    if not re.search(pat, m.group(1)):
        continue
    print("==", m.group(1))
    blocks, cur = [], ["entry", []]
    for l in m.group(2).split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur); cur = [l.strip()[:60], []]
        else:
            cur[1].append(l)
    blocks.append(cur)
    tot = 0
    for name, ls in blocks:
        c = lambda p: sum(1 for l in ls if re.search(p, l))
        n = c(r"^\s+[a-z]")
        tot += n
        if n < 8: continue
        pats = [("mfma", "v_mfma"), ("valu", r"^\s+v_(?!mfma)"), ("salu", r"^\s+s_(?!waitcnt|barrier|nop)"), ("wait", "s_waitcnt"),
                ("scr", "scratch_"), ("dsr", "ds_read"), ("dsw", "ds_write"), ("dsa", r"ds_(min|add|max)"),
                ("gld", "global_load|buffer_load"), ("gst", "global_store|buffer_store"), ("bar", "s_barrier")]
        print(f"{name:60s} n={n:4d} " + " ".join(f"{k}={c(v):3d}" for k, v in pats))
    print("total instructions", tot)
