#!/usr/bin/env python3
"""Throughput of the seeded random scenes (tests/_randscene.py) by object count: how the linear scan over instances
(one per distinct transform chain) scales.  tools/time_random.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401
import _golden as G, _randscene as R
A = G.A
ctx = G.rtr.Context(0)
W, H, SPP = 640, 360, int(os.environ.get("SPP", "16"))
counts = [int(x) for x in os.environ.get("OBJECTS", "8,24,60,120,200,400,800").split(",")]
for n in counts:
    sc = R.random_scene(19, n_objects=n)
    info = G.rtr.native.validate_scene(sc)
    ctx.upload(sc)
    for integ in (1, 4):
        p = A.make_params(W, H, SPP, integrator=integ, seed=3, spp_chunks=0)
        ctx.render(p)
        best = 1e9
        for _ in range(2):
            ctx.render(p)
            best = min(best, ctx.stats()["device_ms"])
        st = ctx.stats()
        seg = (st["closest_segments"] + st["shadow_segments"]) / st["samples"]
        print("objects %4d  instances %4d refs %5d  i%d: %8.1f Msamples/s (%.2f ms)  %.2f segments/sample  %8.1f Msegments/s" %
              (n, info["fast_instances"], info["fast_refs"], integ, W * H * SPP / best * 1e-3, best, seg, W * H * SPP * seg / best * 1e-3), flush=True)
