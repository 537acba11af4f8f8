#!/usr/bin/env python3
"""Static instruction counts of one kernel in the ISA listing `make -C raytracertest_amd/csrc asm` leaves behind.
Usage: tools/isa_count.py [listing.s] [mangled-name-substring]"""
import collections
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "raytracertest_amd/csrc/_build/rt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
want = sys.argv[2] if len(sys.argv) > 2 else "trace_kernelILb1ELi2ELb1ELb0ELb1ELb1ELb0ELb0E"
inside = False
count = collections.Counter()
for line in open(path):
    if not inside:
        if re.match(r"^_Z\w*%s\w*:" % re.escape(want), line):
            inside = True
        continue
    if line.startswith(".Lfunc_end"):                # (blocks may be placed behind the first s_endpgm)
        break
    m = re.match(r"^\s+([a-z_0-9]+)\s", line)
    if not m:
        continue
    op = m.group(1)
    key = ("ds_bpermute" if op.startswith("ds_bpermute") else "v_readlane/readfirstlane" if op.startswith("v_read") else
           "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
           "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
    count[key] += 1
print(want, dict(count), "total", sum(count.values()))
