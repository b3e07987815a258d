"""Probe: instruction mix per loop of a kernel in a `hipcc -S` listing (uses the compiler's "Loop Header" block comments).
usage: python tools/probe/isa_mix.py file.s mangled_kernel_name_substring"""
import re
import sys
from collections import Counter, defaultdict

s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r"^(\S*" + re.escape(name) + r"\S*):[^\n]*\n(.*?)^\.Lfunc_end", s, re.S | re.M)
mix = defaultdict(Counter)
ops = defaultdict(Counter)
cur = None
for raw in m.group(2).split("\n"):
    l = raw.strip()
    if not l or l.startswith(";"):
        continue
    lab = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
    if lab:
        c = lab.group(2) or ""
        h = re.search(r"Header=(BB\d+_\d+)", c)
        if "Loop Header" in c:
            cur = lab.group(1)[2:]
        elif h:
            cur = h.group(1)
        else:
            cur = None
        continue
    if cur is None:
        continue
    op = l.split()[0]
    k = ("mfma" if op.startswith("v_mfma") else "trans" if re.match(r"v_(exp|log|rcp|rsq|sqrt)", op) else
         "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
         "vmem" if op.startswith(("global_", "buffer_", "scratch_")) else "wait" if op.startswith("s_waitcnt") else
         "salu" if op.startswith("s_") else "other")
    mix[cur][k] += 1
    if k == "valu":
        ops[cur][op] += 1
print(m.group(1))
for h, c in mix.items():
    print(" loop", h, dict(c))
    print("   ", ops[h].most_common(14))
