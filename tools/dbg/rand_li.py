import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import torch  # noqa
import numpy as np, _golden as G, _randscene as R, test_random_scenes as T
A = G.A; rtr = G.rtr
ctx = rtr.Context(0)
for seed, integ in [(17, 3), (11, 3), (16, 3), (14, 3), (18, 0), (15, 1)]:
    kw = dict(T.CASES)[seed]
    sc = R.random_scene(seed, **kw)
    ctx.upload(sc)
    W, H, spp = 48, 32, 4
    p = A.make_params(W, H, spp, integrator=integ, seed=100 + seed)
    recs = np.zeros(W * H * spp, dtype=A.LI_DTYPE)
    ii, jj, ss = np.meshgrid(np.arange(W), np.arange(H), np.arange(spp), indexing="ij")
    recs["i"], recs["j"], recs["s"] = ii.ravel(), jj.ravel(), ss.ravel()
    ora = G.oracle_records(sc, "rto_li", recs, p)
    dev = ctx.test_records("li", recs, p)
    bad = np.flatnonzero((ora["L"] != dev["L"]).any(axis=1) | (ora["rng_exit"] != dev["rng_exit"]) | (ora["n_closest"] != dev["n_closest"]))
    print("seed", seed, "integ", integ, "differing samples", len(bad), "of", len(recs))
    for k in bad[:6]:
        print("   ", recs["i"][k], recs["j"][k], recs["s"][k], "ora L", ora["L"][k], "seg", ora["n_closest"][k], ora["n_shadow"][k], "rng", ora["rng_exit"][k],
              "| dev L", dev["L"][k], "seg", dev["n_closest"][k], dev["n_shadow"][k], "rng", dev["rng_exit"][k])
