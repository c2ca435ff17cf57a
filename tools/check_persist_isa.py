#!/usr/bin/env python3
"""Check the code object of the persistent pile kernel (pile_runs_kernel<.., kPersist = true>).

The kernel requests the next read's events and offsets by inline assembly that the compiler's wait-count pass does
not know (pile_runs_kernel.hip, "kPersist"), into registers the compiler does not have (v64 .. v71 of an
instantiation compiled for 64).  This script reads the assembly (hipcc -save-temps) and fails unless, in every
persistent instantiation,
  * nothing but the kernel's own statements names v64 .. v71: global_load_dword into them, v_mov_b32 / v_readlane_b32
    out of them;
  * every read of one of them has an `s_waitcnt vmcnt(0)` between itself and the last request in front of it
    (in the order of the code; the statements are volatile, the order of the code is the order of the source);
  * the kernel takes 72 registers (seven wavefronts per SIMD).
Scratch accesses (spills) are listed with their line: one on the straight path of an item would be a wait for the
item's row stores.
usage: check_persist_isa.py file.s
"""
import re
import sys


def kernels(text):
    cur, name = None, None
    for line in text.splitlines():
        m = re.match(r"^(_ZN8rala_hip16pile_runs_kernelI\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                yield name, cur
                cur = None


def high_regs(line):
    out = set()
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return {r for r in out if r >= 64}


def check(name, lines):
    code = [(i, l.strip()) for i, l in enumerate(lines) if l.startswith("\t") and not l.strip().startswith((";", "."))]
    problems, pending, n_req, n_read = [], False, 0, 0
    for i, l in code:
        if l.startswith("s_waitcnt vmcnt(0)"):
            pending = False
        hi = high_regs(l)
        if not hi:
            continue
        if re.match(r"global_load_dword v(6[4-9]|7[01]), ", l) and not high_regs(l.split(",", 1)[1]):
            pending = True
            n_req += 1
        elif re.match(r"(v_mov_b32 v\d+, v(6[4-9]|70)|v_readlane_b32 s\d+, v71, \d)$", l) and max(high_regs(l.split(",")[0]) | {0}) < 64:
            n_read += 1
            if pending:
                problems.append("%s: line %d reads a requested register without a wait behind the request: %s" % (name, i, l))
        else:
            problems.append("%s: line %d names a register of the requests: %s" % (name, i, l))
    if n_req != 15 or n_read != 17:
        problems.append("%s: %d requests and %d reads found (7 + 8 requests - the first item's, then one site in the loop - and 7 + 2 x 5 reads expected)" % (name, n_req, n_read))
    # a wait for every outstanding access that the compiler put in (not the kernel's own three and the ones behind a returning atomic)
    own = 0
    for k, (i, l) in enumerate(code):
        if l.startswith("s_waitcnt vmcnt(0)"):
            before = [x for _, x in code[max(0, k - 4):k]]
            if not any(b.startswith(("global_atomic", "global_load_dword v7", "global_load_dword v", "scratch_load")) for b in before):
                own += 1
    if own > 3:
        problems.append("%s: %d waits for all outstanding accesses that do not follow a load or a returning atomic (3 are the kernel's own)" % (name, own))
    scratch = [(i, l) for i, l in code if l.startswith("scratch_")]
    print("%s\n  %d request instructions, %d reads, scratch accesses at lines %s" % (name, n_req, n_read, [i for i, _ in scratch]))
    return problems


def main():
    text = open(sys.argv[1]).read()
    bad, seen = [], 0
    for name, lines in kernels(text):
        # pile_runs_kernel<kCap, kDiag, kSens, kOne, kBases, kWaves, kPersist = true, ..>
        if not re.search(r"pile_runs_kernelILj\d+ELb[01]ELi\d+ELb[01]ELj\d+ELj\d+ELb1E", name):
            continue
        seen += 1
        bad += check(name, lines)
        m = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % re.escape(name), text, re.S)
        nf = re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(1)) if m else None
        if not nf or int(nf.group(1)) != 72:
            bad.append("%s: .amdhsa_next_free_vgpr is %s, not 72" % (name, nf.group(1) if nf else "missing"))
    if not seen:
        bad.append("no persistent instantiation in %s" % sys.argv[1])
    for b in bad:
        print("PROBLEM:", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
