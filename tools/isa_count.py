#!/usr/bin/env python3
"""Count instructions per kernel in a hipcc -S device listing (development aid).
usage: isa_count.py file.s substring [substring...]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pats = sys.argv[2:]
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.Lfunc_end", s, flags=re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pats and not any(p in name for p in pats):
        continue
    ops = collections.Counter()
    for line in body.split("\n"):
        mm = re.match(r"\s+([a-z][a-z0-9_]+)\s", line)
        if mm and not line.strip().startswith((".", ";")):
            ops[mm.group(1)] += 1
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    print("%s\n  VALU %d  SALU %d  LDS %d  VMEM %d  total %d" % (
        name, valu, sum(v for k, v in ops.items() if k.startswith("s_")),
        sum(v for k, v in ops.items() if k.startswith("ds_")),
        sum(v for k, v in ops.items() if k.startswith(("global_", "buffer_", "scratch_", "flat_"))), sum(ops.values())))
    print("  " + ", ".join("%s:%d" % kv for kv in ops.most_common(28)))
