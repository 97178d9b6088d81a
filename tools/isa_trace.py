#!/usr/bin/env python3
"""Compact trace of a kernel's instruction stream from a hipcc -S listing: runs of instruction classes
(V = VALU, DR/DW = LDS read/write, GL/GS = buffer/global load/store, W(..) = s_waitcnt, B = s_barrier,
s = other scalar), one line per basic block.  Used to see where the compiler put the LDS traffic and
the waits relative to the butterflies.
usage: isa_trace.py file.s kernel-name-substring"""
import re, sys

def classify(ins, ops):
    if ins.startswith("v_"): return "V"
    if ins.startswith("ds_read") or ins.startswith("ds_load"): return "DR"
    if ins.startswith("ds_write") or ins.startswith("ds_store"): return "DW"
    if ins.startswith("buffer_load") or ins.startswith("global_load"): return "GL"
    if ins.startswith("buffer_store") or ins.startswith("global_store"): return "GS"
    if ins == "s_waitcnt": return "W(" + ops.replace(" ", "") + ")"
    if ins == "s_barrier": return "B"
    if ins.startswith("s_cbranch") or ins == "s_branch": return "BR"
    if ins == "s_nop": return "n"
    return "s"

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
    out, cur, last, cnt = [], [], None, 0
    total = {}
    def flush():
        nonlocal last, cnt
        if last is not None:
            cur.append(last if cnt == 1 else "%s*%d" % (last, cnt))
        last, cnt = None, 0
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"): break
        if re.match(r"^\.LBB\S+:", s):
            flush()
            if cur: out.append(" ".join(cur))
            cur = [s]
            continue
        if not s or s.startswith(";") or s.startswith("."): continue
        m = re.match(r"^(\S+)\s*(.*?)(;.*)?$", s)
        c = classify(m.group(1), m.group(2))
        total[c.split("(")[0]] = total.get(c.split("(")[0], 0) + 1
        if c == last: cnt += 1
        else:
            flush(); last, cnt = c, 1
    flush()
    if cur: out.append(" ".join(cur))
    for o in out: print(o); print()
    print("totals:", total)

main()
