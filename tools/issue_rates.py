#!/usr/bin/env python3
"""Measured issue cost per instruction class (run on a GPU box from the repo root): shader cycles a wave spends per
instruction while four waves share each SIMD -> divide by 4 for the SIMD's cost per instruction where the class is
pipe-bound.  Writes gpurun_out/issue_rates.json; profiles/rNN_issue_rates.json is what bench.py prices VALU slots with."""
import importlib
import json
import sys

sys.path.insert(0, ".")
pkg = importlib.import_module("ray_tracing-rendering_amd")
with pkg.Context(0) as ctx:
    r = ctx.issue_rates()
for k, v in r.items():
    print("%-22s %7.2f cycles per wave-instruction with 4 waves per SIMD  (%.2f per SIMD)" % (k, v, v / 4))
json.dump({"cycles_per_wave_instruction_at_4_waves_per_simd": r}, open("gpurun_out/issue_rates.json", "w"), indent=1)
