#!/usr/bin/env python3
"""Summarise `-Rpass-analysis=kernel-resource-usage` remarks (stderr of hipcc) per kernel."""
import re
import sys

txt = open(sys.argv[1]).read()
seen = set()
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    if name in seen:
        continue
    seen.add(name)

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"

    print("%-64s VGPR %3s AGPR %3s SGPR %3s scratch %3s occ %s vspill %s sspill %s" % (
        name[:64], g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"),
        g(r"Occupancy \[waves/SIMD\]"), g("VGPRs Spill"), g("SGPRs Spill")))
