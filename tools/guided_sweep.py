#!/usr/bin/env python3
"""One rank's eighth of C2 on one GPU under other guided-chunk splits (RTR_GUIDED=share,divisor,eighths)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import bench
pkg = importlib.import_module("ray_tracing-rendering_amd")
A = pkg._abi
sc = bench.load_scene(pkg, 21)
W = H = 800
with pkg.Context(0) as ctx:
    ctx.upload(sc)
    fb = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    for n in (8, 4):
        for g in ("0.90,3,6", "0.85,3,6", "0.80,3,6", "0.95,3,6", "0.90,2,6", "0.90,4,6", "0.90,6,6", "0.90,3,7", "0.85,4,7",
                  "0.80,4,6", "0.85,4,6", "0.75,4,6", "0.85,6,6", "0.80,6,5"):
            os.environ["RTR_GUIDED"] = g
            p = A.make_params(W, H, 400, integrator=4, seed=1, tile_first=0, tile_stride=n, spp_chunks=0)
            best = 1e9
            for _ in range(4):
                ctx.render_into(p, fb.data_ptr(), W, blocking=True)
                best = min(best, ctx.stats()["device_ms"])
            print("N=%d  RTR_GUIDED=%-10s chunks %2d  %.3f ms" % (n, g, ctx.stats()["spp_chunks"], best), flush=True)
