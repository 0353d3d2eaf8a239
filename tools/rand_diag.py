import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch  # noqa
import numpy as np, _golden as G, _randscene as R, test_random_scenes as T
A = G.A; rtr = G.rtr
ctx = rtr.Context(0)
def bits(a): return np.ascontiguousarray(a).view(np.uint64)
for seed, kw in T.CASES:
    sc = R.random_scene(seed, **kw)
    try:
        ctx.upload(sc)
    except Exception as e:
        print(seed, 'upload', e); continue
    rays = R.random_rays(seed, 6000)
    ora = G.oracle_records(sc, "rto_hits", rays)
    out = {}
    for name, flag in (("compiled", False), ("exact", True)):
        ctx.reference_order(flag)
        try:
            out[name] = ctx.test_records("hits", rays)
        except Exception as e:
            print(seed, name, 'ERR', e)
        ctx.reference_order(False)
    def diff(a, b):
        hd = int((a["hit"] != b["hit"]).sum())
        h = (a["hit"] == 1) & (b["hit"] == 1)
        td = int((bits(a["t"][h]) != bits(b["t"][h])).sum())
        md = int((a["material"][h] != b["material"][h]).sum())
        rd = int((a["rng_out"] != b["rng_out"]).sum())
        return "hit %d t %d mat %d rng %d" % (hd, td, md, rd)
    msg = "seed %d %s | " % (seed, kw)
    if "compiled" in out: msg += "ora~compiled: %s | " % diff(ora, out["compiled"])
    if "exact" in out: msg += "ora~exact: %s | " % diff(ora, out["exact"])
    if len(out) == 2: msg += "compiled~exact: %s" % diff(out["compiled"], out["exact"])
    print(msg, flush=True)
    info = rtr.native.validate_scene(sc)
    for integ in (0, 1, 3, 4):
        p = A.make_params(48, 32, 4, integrator=integ, seed=100 + seed)
        want, wst = G.oracle_render(sc, p)
        res = []
        for pipe, flags in [(1, 0), (1, A.FLAG_REFERENCE_ORDER)] + ([(2, 0)] if (info["fast_ok"] or info["program_steps"] > 0) else []):
            try:
                got = ctx.render(A.make_params(48, 32, 4, integrator=integ, seed=100 + seed, pipeline=pipe, flags=flags))
                st = ctx.stats()
                res.append("p%d f%d: err %.2e seg %d/%d vs %d/%d" % (pipe, flags, G.rel_l2(got, want), st["closest_segments"], st["shadow_segments"], wst["closest_segments"], wst["shadow_segments"]))
            except Exception as e:
                res.append("p%d f%d: %s" % (pipe, flags, str(e)[:60]))
        print("   i%d  " % integ + " | ".join(res), flush=True)
