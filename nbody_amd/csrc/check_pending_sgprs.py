#!/usr/bin/env python3
"""Build-time check of the device assembly of tree_walk.hip.

The walk issues its node loads through inline assembly so that a load and the wait for it can be a whole visit
apart.  The compiler does not know that the destination registers of such a load are in flight in between: a
register copy or a reuse scheduled into that window would read or clobber them silently.  This script walks the
control-flow graph of every kernel in the assembly and fails if any instruction between a scalar load and the next
`s_waitcnt lgkmcnt(0)` on some path touches the load's destination registers.  It also checks that M0, which the
walk's inline assembly sets without the compiler's knowledge, is used by nothing else (check_m0).

usage: check_pending_sgprs.py file.s
"""
import re
import sys

REG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")
TERMINATORS = ("s_branch", "s_cbranch", "s_endpgm", "s_setpc")


def sgprs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def kernels(asm):
    """Yields (name, [instruction or label lines]) per function."""
    for m in re.finditer(r"^(\w+):\s*(?:;.*)?\n", asm, re.M):
        name = m.group(1)
        end = asm.find(".end_amdhsa_kernel", m.end())
        nxt = asm.find("\t.section", m.end())
        stop = min(x for x in (end, nxt, len(asm)) if x >= 0)
        if name.startswith(".") or "s_endpgm" not in asm[m.end():stop]:
            continue
        lines = []
        for raw in asm[m.end():stop].split("\n"):
            line = raw.split(";")[0].strip()
            if not line:
                continue
            if line.startswith(".LBB") and line.endswith(":"):
                lines.append(line)
            elif not line.startswith("."):
                lines.append(line)
        yield name, lines


def check(name, lines):
    # basic blocks
    blocks, cur, label_of = [], [], {}
    for line in lines:
        if line.endswith(":"):
            if cur:
                blocks.append(cur)
            cur = [line]
            continue
        cur.append(line)
        if line.startswith(TERMINATORS):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    for i, b in enumerate(blocks):
        if b[0].endswith(":"):
            label_of[b[0][:-1]] = i
    succ = []
    for i, b in enumerate(blocks):
        last = b[-1]
        s = []
        if last.startswith("s_branch"):
            s.append(label_of[last.split()[1]])
        elif last.startswith("s_cbranch"):
            s.append(label_of[last.split()[1]])
            if i + 1 < len(blocks):
                s.append(i + 1)
        elif last.startswith(("s_endpgm", "s_setpc")):
            pass
        elif i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    pending_in = [set() for _ in blocks]
    problems = set()
    work = list(range(len(blocks)))
    while work:
        i = work.pop()
        pend = set(pending_in[i])
        for line in blocks[i]:
            if line.endswith(":"):
                continue
            if line.startswith("s_waitcnt"):
                if "lgkmcnt(0)" in line:
                    pend = set()
                continue
            if line.startswith(("s_branch", "s_cbranch", "s_nop", "s_endpgm")):
                continue
            ops = line.split(None, 1)
            regs = sgprs(ops[1]) if len(ops) > 1 else set()
            if regs & pend:
                problems.add(line)
            if line.startswith(("s_load_", "s_buffer_load_")):
                pend |= sgprs(ops[1].split(",")[0])
        for j in succ[i]:
            if not pend <= pending_in[j]:
                pending_in[j] |= pend
                work.append(j)
    return sorted(problems)


def check_m0(lines):
    """The walk pushes onto its stack of cell ends with `s_mov_b32 m0, <lane>; s_nop 0; v_writelane_b32 v, s, m0` from
    inline assembly and does not tell the compiler (M0 is reserved: it cannot be named as clobbered).  That is sound only
    while nothing else in the kernel uses M0: every other mention of it is reported."""
    bad = []
    insts = [l for l in lines if not l.endswith(":")]
    for k, line in enumerate(insts):
        if not re.search(r"\bm0\b", line):
            continue
        if line.startswith("s_mov_b32 m0,"):
            follow = insts[k + 1:k + 3]
            if len(follow) == 2 and follow[0].startswith("s_nop") and follow[1].startswith("v_writelane_b32") and follow[1].endswith(", m0"):
                continue
        elif line.startswith("v_writelane_b32") and line.endswith(", m0") and k >= 2 and insts[k - 2].startswith("s_mov_b32 m0,"):
            continue
        bad.append(line)
    return bad


def main():
    asm = open(sys.argv[1]).read()
    bad = 0
    n = 0
    for name, lines in kernels(asm):
        n += 1
        for line in check(name, lines):
            print(f"{name}: touches scalar registers of a load in flight: {line}", file=sys.stderr)
            bad += 1
        for line in check_m0(lines):
            print(f"{name}: uses M0 outside the walk's lane write: {line}", file=sys.stderr)
            bad += 1
    if n == 0:
        print("no kernels found in", sys.argv[1], file=sys.stderr)
        return 1
    print(f"{sys.argv[1]}: {n} kernels checked, {bad} problems")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
