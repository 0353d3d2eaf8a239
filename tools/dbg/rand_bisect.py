import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import torch  # noqa
import numpy as np, _golden as G, _randscene as R, test_random_scenes as T
A = G.A; rtr = G.rtr
ctx = rtr.Context(0)
NAMES = {0:'bvh',1:'list',2:'translate',3:'rotate',4:'flip',5:'medium',6:'sphere',7:'moving',8:'xy',9:'xz',10:'yz'}
MAT = {0:'lambert',1:'metal',2:'diel',3:'light',4:'pbr',5:'iso'}
def describe(sc, k):
    n = sc.nodes[k]; t = int(n['type'])
    if t in (2,3,4): return NAMES[t] + '(' + describe(sc, int(n['a'])) + ')'
    if t == 1: return 'list[' + ','.join(describe(sc, int(c)) for c in sc.list_children[n['a']:n['a']+n['b']][:2]) + ('..%d' % n['b']) + ']'
    if t == 5: return 'medium(' + describe(sc, int(n['a'])) + ')'
    m = sc.materials[n['a']]
    tex = sc.textures[m['tex'][0]]['type'] if m['type'] in (0,3,5) else -1
    return NAMES[t] + ':' + MAT[int(m['type'])] + ('/checker' if tex == 1 else '')
def count_bad(sc, integ, seed):
    ctx.upload(sc)
    W, H, spp = 48, 32, 4
    p = A.make_params(W, H, spp, integrator=integ, seed=100 + seed)
    recs = np.zeros(W * H * spp, dtype=A.LI_DTYPE)
    ii, jj, ss = np.meshgrid(np.arange(W), np.arange(H), np.arange(spp), indexing="ij")
    recs["i"], recs["j"], recs["s"] = ii.ravel(), jj.ravel(), ss.ravel()
    ora = G.oracle_records(sc, "rto_li", recs, p)
    dev = ctx.test_records("li", recs, p)
    rel = np.abs(ora["L"] - dev["L"]).max(axis=1) / np.maximum(np.abs(ora["L"]).max(axis=1), 1e-30)
    return int((rel > 1e-9).sum())
for seed, integ in [(17, 3), (16, 3)]:
    kw = dict(T.CASES)[seed]
    sc = R.random_scene(seed, **kw)
    root = sc.nodes[sc.root]
    kids = sc.list_children[root['a']:root['a'] + root['b']].copy()
    print("seed", seed, "all:", count_bad(sc, integ, seed))
    for drop in range(len(kids)):
        keep = np.delete(kids, drop)
        lc = np.concatenate([sc.list_children, keep]).astype(np.int32)
        nodes = sc.nodes.copy()
        nodes['a'][sc.root] = len(sc.list_children); nodes['b'][sc.root] = len(keep)
        s2 = rtr.Scene(sc.root, nodes, lc, sc.materials, sc.textures, sc.perlin, sc.images, sc.image_bytes, sc.lights, sc.camera, sc.background)
        print("   drop", drop, describe(sc, int(kids[drop])), "->", count_bad(s2, integ, seed), flush=True)
