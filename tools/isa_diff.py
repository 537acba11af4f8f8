#!/usr/bin/env python3
"""Which kernels' gfx950 ISA changed between two `make -C raytracertest_amd/csrc asm` outputs (labels and comments
normalised).  Usage: isa_diff.py old.s new.s"""
import hashlib, re, sys


def funcs(path):
    t = open(path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", t, re.S | re.M):
        body = re.sub(r"\.LBB\d+_\d+", ".LBB", m.group(2))
        body = re.sub(r";.*", "", body)
        out[m.group(1)] = (hashlib.md5(body.encode()).hexdigest(), body.count("\n"))
    return out


a, b = funcs(sys.argv[1]), funcs(sys.argv[2])
same = 0
for k in sorted(a):
    if k not in b:
        print("GONE", k[:110])
    elif a[k][0] == b[k][0]:
        same += 1
    else:
        print("DIFF", k[:110], a[k][1], "->", b[k][1], "lines")
print("same %d, new %d" % (same, len(set(b) - set(a))))
