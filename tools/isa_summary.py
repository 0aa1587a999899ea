#!/usr/bin/env python3
"""Static instruction census of a gfx950 ISA listing (hipcc -S): per kernel, instructions by class, in total and per
segment between s_barrier instructions -- straight-line text order, NOT an execution count (branches not followed)."""
import re, sys, collections

def klass(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        if op.startswith(("s_load", "s_buffer_load", "s_store")):
            return "smem"
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_endpgm", "s_branch", "s_cbranch", "s_setprio", "s_code_end")):
            return "sctl"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"

txt = open(sys.argv[1]).read()
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*s_endpgm", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    tail = txt[m.end():m.end() + 6000]
    meta = re.search(r"; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", tail, re.S)
    sg = re.search(r"; TotalNumSgprs: (\d+)", tail)
    seg, segs = collections.Counter(), []
    for line in body.splitlines():
        line = line.split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        op = line.split()[0]
        seg[klass(op)] += 1
        if op == "s_barrier":
            segs.append(seg)
            seg = collections.Counter()
    segs.append(seg)
    total = sum(segs, collections.Counter())
    short = re.sub(r"^_ZN2sl2rl\d+", "", name)[:70]
    print(short)
    if meta:
        print("  vgpr %s  sgpr %s  scratch %s  occupancy %s" % (meta.group(1), sg.group(1) if sg else "?", meta.group(2), meta.group(3)))
    keys = ("valu", "salu", "smem", "lds", "vmem", "sctl")
    print("  %-10s" % "segment" + "".join("%7s" % k for k in keys))
    for i, s in enumerate(segs):
        print("  %-10d" % i + "".join("%7d" % s[k] for k in keys))
    print("  %-10s" % "total" + "".join("%7d" % total[k] for k in keys))
