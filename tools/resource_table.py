#!/usr/bin/env python3
"""VGPRs / scratch / occupancy of every trace_kernel instantiation, from `make -C raytracertest_amd/csrc asm`
(-Rpass-analysis=kernel-resource-usage).  Template arguments: FMA, K, FILTER, STATS, BIN, ONEPASS."""
import os, re, subprocess, sys
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                          "raytracertest_amd", "csrc", "_build", "resource_usage.txt")
t = open(path).read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split()[0]
    if "trace_kernel" not in name:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    d = re.sub(r".*trace_kernel<([^>]*)>.*", r"\1", d)
    print("%-45s vgpr %3s scratch %3s waves/SIMD %s" % (d, g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
