#!/usr/bin/env python3
"""Device time of a few non-headline fixture scenes (run on a GPU box from the repo root):
scene 1 (462-sphere box tree, RR), 35 (image-textured PBR), 30 / 40 (hollow glass: guarded references),
24 (HDR environment map), 23 (Cook-Torrance, half size).  Prints Msamples/s from the
library's own HIP-event timing."""
import sys

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import _golden as G  # fixtures only (scenes); nothing of the oracle is used here

A, rtr = G.A, G.rtr
ctx = rtr.Context(0)
for sid, integ, W, H, spp in ((1, 1, 800, 450, 100), (35, 4, 800, 450, 100), (30, 4, 800, 450, 400),
                              (40, 4, 800, 450, 400), (24, 4, 800, 450, 100), (23, 4, 960, 540, 256)):
    ctx.upload(G.scene(sid))
    p = A.make_params(W, H, spp, integrator=integ, seed=1)
    ctx.render(p)  # warm-up
    ctx.render(p)
    st = ctx.stats()
    seg = (st["closest_segments"] + st["shadow_segments"]) / st["samples"]
    rate = W * H * spp / (st["device_ms"] * 1e-3) * 1e-6
    print("scene %d i%d: %.1f Msamples/s (device %.1f ms)  %.2f segments/sample  %.0f Msegments/s" %
          (sid, integ, rate, st["device_ms"], seg, rate * seg))
