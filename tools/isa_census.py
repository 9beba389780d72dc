#!/usr/bin/env python3
"""Static instruction census of one kernel in a --save-temps .s file: per basic block, the scalar-ALU, vector-ALU and memory
instruction counts (where do a kernel's SQ_INSTS_SALU come from?).  usage: isa_census.py file.s kernel-name-substring [-v]"""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    verbose = "-v" in sys.argv[3:]
    s = open(path).read()
    m = None
    for m in re.finditer(r"^(_Z\w*):\s*;", s, re.M):
        if pat in m.group(1):
            break
    else:
        raise SystemExit("no kernel matching %r" % pat)
    name = m.group(1)
    body = s[m.end():s.index(".Lfunc_end", m.end())]
    blocks, cur = [], ("entry", [])
    for l in body.split("\n"):
        l = l.strip()
        if not l or l.startswith(";"):
            continue
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur)
            cur = (l.split(":")[0], [])
        elif not l.startswith("."):
            cur[1].append(l.split(";")[0].strip())
    blocks.append(cur)
    tot = [0, 0, 0]
    kinds = {}
    for n, ins in blocks:
        ns = sum(1 for i in ins if i.startswith("s_") and not i.startswith(("s_waitcnt", "s_nop", "s_load", "s_buffer_load")))
        nv = sum(1 for i in ins if i.startswith("v_"))
        nm = sum(1 for i in ins if i.startswith(("global_", "buffer_", "flat_", "ds_", "s_load", "s_buffer_load", "scratch_")))
        tot[0] += ns
        tot[1] += nv
        tot[2] += nm
        for i in ins:
            if i.startswith("s_"):
                op = i.split()[0]
                kinds[op] = kinds.get(op, 0) + 1
        print("%-12s total %4d  salu %4d  valu %4d  mem %3d" % (n, len(ins), ns, nv, nm))
        if verbose:
            for i in ins:
                print("      " + i)
    print(name[:60], "static totals: salu %d valu %d mem %d" % tuple(tot))
    print("scalar opcodes:", sorted(kinds.items(), key=lambda kv: -kv[1])[:25])
    k = s.index(".amdhsa_kernel " + name)
    print(re.findall(r"\.amdhsa_next_free_[vs]gpr \d+", s[k:k + 6000]))


if __name__ == "__main__":
    main()
