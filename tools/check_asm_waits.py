#!/usr/bin/env python3
"""Build-time check of the hand-placed vector-memory waits (csrc/wgrad_jobs.hip and csrc/wgrad_split.hip).

Those kernels issue their pipelined global loads from inline asm and wait for them with hand-counted `s_waitcnt vmcnt(N)`: the
compiler does not know that the destination registers of an asm load are still pending, so nothing but the register allocation of the
build at hand keeps it from reading (or overwriting) one of them between the load and its wait -- a copy on a loop back edge would
read data that has not arrived and the gradients would be silently wrong.  This tool re-derives that property from the ISA of the
build: it compiles the given .hip files to gfx950 assembly, walks every kernel in program order with a model of the vmcnt counter
(every vector-memory instruction -- load or store, the compiler's or an asm block's -- enters a FIFO with its destination registers;
`s_waitcnt vmcnt(N)` retires all but the N youngest; gfx9: loads and stores share the counter and retire in order) and reports every
instruction that mentions a register whose load is still in flight.  A backward branch re-walks its loop body once with the state at
the latch (what the second iteration sees).  Compiler-managed loads satisfy the same rule by construction, so they are checked too.

usage: python tools/check_asm_waits.py [file.hip ...]      (default: the two kernels with hand-placed waits); exit code 1 on a finding
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rq-vae-recommender_amd", "csrc")
DEFAULT = ["wgrad_jobs.hip", "wgrad_split.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-fast-math", "-fno-slp-vectorize", "-munsafe-fp-atomics", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
         "--cuda-device-only", "-S"]          # == csrc/Makefile's FLAGS

_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_VMEM = re.compile(r"^(global|flat|buffer|scratch)_(load|store|atomic)\w*")


def regs_of(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def compile_to_asm(src):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        pr = subprocess.run([hipcc, *FLAGS, "-o", out, src], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if pr.returncode != 0:
            raise RuntimeError(pr.stderr[-2000:])
        return open(out).read()


def kernels(asm):
    """{name: [(line number, instruction text)]} of every .amdhsa kernel body (labels kept as 'LABEL:')."""
    out, name, body = {}, None, []
    for no, raw in enumerate(asm.splitlines(), 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        m = re.match(r"^(_Z\w+):", line)
        if m and not line.startswith(".L"):
            name, body = m.group(1), []
            out[name] = body
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end") or line.startswith(".section") or line.startswith(".end_amdhsa_kernel"):
            name = None
            continue
        if line.startswith(".") and not line.startswith(".LBB"):
            continue
        body.append((no, line))
    return {k: v for k, v in out.items() if any(t.startswith("s_endpgm") for _, t in v)}


def vmcnt_of(text):
    m = re.search(r"vmcnt\((\d+)\)", text)
    if m:
        return int(m.group(1))
    m = re.match(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", text)
    if m:                                     # raw immediate (gfx9): vmcnt = imm[3:0] | imm[15:14] << 4
        imm = int(m.group(1), 0)
        return (imm & 0xF) | ((imm >> 14) & 0x3) << 4
    return None                               # only other counters named: vmcnt untouched


def walk(body, findings, name):
    """Forward dataflow over the kernel's control-flow graph.  State: {register: (c, line of the load)} for every register with a load possibly in flight,
    c = the number of vector-memory operations issued after that load on the path with the FEWEST of them (`s_waitcnt vmcnt(N)` retires
    a load once c >= N: at most N operations outstanding means all but the N youngest have completed; they complete in order).  Joins
    take the union of the registers and the minimum of c.  An instruction that mentions a register of the state is a finding."""
    # basic blocks
    leaders = {0}
    labels = {t[:-1]: i for i, (_, t) in enumerate(body) if t.endswith(":")}
    br = re.compile(r"^(s_cbranch_\w+|s_branch)\s+(\.LBB\w+)")
    for i, (_, t) in enumerate(body):
        m = br.match(t)
        if m:
            if m.group(2) in labels:
                leaders.add(labels[m.group(2)])
            leaders.add(i + 1)
        elif t.startswith("s_endpgm") or t.startswith("s_setpc"):
            leaders.add(i + 1)
    starts = sorted(x for x in leaders if x < len(body))
    block_of = {}
    blocks = []
    for bi, st in enumerate(starts):
        en = starts[bi + 1] if bi + 1 < len(starts) else len(body)
        blocks.append((st, en))
        block_of[st] = bi
    succ = []
    for st, en in blocks:
        _, last = body[en - 1]
        m = br.match(last)
        out = []
        if m:
            if m.group(2) in labels:
                out.append(block_of[labels[m.group(2)]])
            if m.group(1) != "s_branch" and en < len(body):
                out.append(block_of[en])
        elif not (last.startswith("s_endpgm") or last.startswith("s_setpc")) and en < len(body):
            out.append(block_of[en])
        succ.append(out)

    def transfer(bi, state, report):
        state = dict(state)
        st, en = blocks[bi]
        for i in range(st, en):
            no, t = body[i]
            if t.endswith(":"):
                continue
            op = t.split()[0]
            if op == "s_waitcnt":
                n = vmcnt_of(t)
                if n is not None:
                    state = {r: v for r, v in state.items() if v[0] < n}
                continue
            if report:
                # (a load whose DESTINATION is still pending from an older load is fine: loads complete in order, the younger one
                # lands last -- the compiler itself re-issues conditional fetches that way; its address operands are reads)
                vm = _VMEM.match(op)
                read = regs_of(t[len(op):].split(",", 1)[1] if "," in t else "") if (vm and vm.group(2) == "load") else regs_of(t)
                hit = read & state.keys()
                if hit:
                    r = sorted(hit)[0]
                    findings.append(f"{name}: line {no}: `{t}` mentions v{r} while the load of line {state[r][1]} into it may be in flight "
                                    f"(as few as {state[r][0]} younger vector-memory operations and no s_waitcnt vmcnt(<= {state[r][0]}) on some path)")
            m = _VMEM.match(op)
            if m:
                state = {r: (min(v[0] + 1, 64), v[1]) for r, v in state.items()}
                if m.group(2) == "load" and " lds" not in t:
                    for r in regs_of(t[len(op):].split(",")[0]):
                        state[r] = (0, no)
        return state

    # Divergent if / else as the compiler lowers it: `s_and_saveexec; s_xor s, exec, s; s_cbranch_execz ELSE; THEN; ELSE: s_or_saveexec;
    # s_xor exec, exec, s; s_cbranch_execz END; ELSE body; END:`.  The path that takes BOTH skips needs an empty exec mask on entry -- a wave
    # with no active lane, whose registers nobody reads.  Without excluding it, a wave-uniform `if (role) loop A else loop B` whose loops
    # end with their own waits looks as if the loads issued before it could reach the code behind it unawaited.  So the dataflow runs on
    # (block, f) nodes, f = "arrived here by skipping a THEN side": in such a node an ELSE-side skip is not followed.
    def skip_kind(bi):
        st, en = blocks[bi]
        _, last = body[en - 1]
        if not last.startswith("s_cbranch_execz"):
            return None
        head = [t for _, t in body[max(st, en - 8):en - 1]]
        if any(t.startswith("s_or_saveexec") or re.match(r"s_xor_b64\s+exec,\s*exec", t) for t in head):
            return "else"
        if any(t.startswith("s_and_saveexec") for t in head):
            return "then"
        return None

    def header_only(bi):      # an ELSE header: nothing but exec bookkeeping / moves before its skip
        st, en = blocks[bi]
        return en - st <= 8 and not any(_VMEM.match(t.split()[0]) or t.startswith("s_waitcnt") for _, t in body[st:en] if not t.endswith(":"))

    n = len(blocks)
    inn = {(0, 0): {}}
    work = [(0, 0)]
    while work:
        bi, f = work.pop()
        out = transfer(bi, inn[(bi, f)], False)
        kind = skip_kind(bi)
        for k, sj in enumerate(succ[bi]):
            taken = k == 0 and kind is not None           # succ lists the branch target first
            if taken and kind == "else" and f and header_only(bi):
                continue                                   # both sides skipped: empty exec
            nf = 1 if (taken and kind == "then") else 0
            key = (sj, nf)
            if key not in inn:
                merged = dict(out)
            else:
                merged = dict(inn[key])
                for r, v in out.items():
                    if r not in merged or v[0] < merged[r][0]:
                        merged[r] = v
            if merged != inn.get(key):
                inn[key] = merged
                work.append(key)
    seen = set()
    for (bi, f), state in sorted(inn.items()):
        before = len(findings)
        transfer(bi, state, True)
        for x in findings[before:]:
            if x in seen:
                findings.remove(x)
            seen.add(x)


def check(asm):
    findings = []
    ks = kernels(asm)
    for name, body in ks.items():
        walk(body, findings, name)
    return ks, findings


def main():
    files = sys.argv[1:] or [os.path.join(CSRC, f) for f in DEFAULT]
    bad = 0
    for f in files:
        ks, findings = check(compile_to_asm(f))
        n_asm = sum(1 for body in ks.values() for _, t in body if t.startswith("global_load") or t.startswith("s_waitcnt"))
        print(f"{os.path.basename(f)}: {len(ks)} kernels, {n_asm} loads / waits walked, {len(findings)} findings")
        for x in findings[:20]:
            print("  " + x)
        bad += len(findings)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
