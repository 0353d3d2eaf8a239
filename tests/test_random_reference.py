"""Random scenes answered by the UNMODIFIED reference (oracle/gen_random_golden.py: the harness builds the
reference's own classes from the scene file): hit records and small images of ten seeded scenes that none of its
demo builders produces.  CPU: the oracle reproduces them bit for bit (its pin beyond the forty demo scenes).
GPU: the HIP path against the same vectors, through the C ABI."""
import gzip
import os

import numpy as np
import pytest

import _golden as G

A = G.A
rtr = G.rtr
# "<seed>": a flat hittable_list as generated; "<seed>b": the same objects under the reference's own bvh_node
# 32 / 33: a constant_medium UNDER translate / rotate_y (a step with a transform chain, FStep::xf_first)
SEEDS = ["11", "13", "14", "15", "16", "17", "18", "19", "27", "28", "32", "33",
         "13b", "15b", "16b", "18b", "19b", "23b", "26b", "28b", "32b", "33b"]
W, H, SPP = 48, 32, 4


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def _scene(seed):
    with gzip.open(os.path.join(G.GOLD, "random_%s.rtrs.gz" % seed), "rb") as f:
        return rtr.Scene.from_bytes(f.read())


def _writes_uv(sc, gold):
    """moving_sphere::hit and constant_medium::hit write no (u,v) (moving_sphere.h:36-62, constant_medium.h:95-101):
    the reference's record then holds whatever an earlier object left in the hittable_list's temp_rec -- no
    material of these scenes reads it (checker and noise textures take p), so it is not part of the comparison."""
    moving = np.isin(gold["material"], sc.nodes["a"][sc.nodes["type"] == A.NODE_MOVING_SPHERE])
    fog = np.isin(gold["material"], np.flatnonzero(sc.materials["type"] == A.MAT_ISOTROPIC))
    return ~moving & ~fog & ~np.isnan(gold["u"])


def _image(seed, integ):
    return np.fromfile(os.path.join(G.GOLD, "random_%s_i%d.f64" % (seed, integ)), dtype="<f8").reshape(H, W, 3)


@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_equals_the_reference_on_random_scenes(seed):
    sc = _scene(seed)
    gold = G.records("random_%s_hits.bin" % seed, A.HIT_DTYPE)
    out = G.oracle_records(sc, "rto_hits", gold)
    h = gold["hit"] == 1
    assert 150 < h.sum() and np.array_equal(out["hit"], gold["hit"]) and np.array_equal(out["rng_out"], gold["rng_out"])
    for f in ("front_face", "material"):
        assert np.array_equal(out[f][h], gold[f][h]), f
    for f in ("t", "p", "n"):
        assert np.array_equal(_bits(out[f][h]), _bits(gold[f][h])), f
    uv = h & _writes_uv(sc, gold)
    for f in ("u", "v"):
        assert np.array_equal(_bits(out[f][uv]), _bits(gold[f][uv])), f
    for integ in (1, 4):
        info = G.MANIFEST["files"]["random_%s_i%d.f64" % (seed, integ)]["info"]
        img, st = G.oracle_render(sc, A.make_params(W, H, SPP, integrator=integ, seed=100 + int(seed.rstrip("b"))))
        assert np.array_equal(_bits(img), _bits(_image(seed, integ))), integ
        # (the harness counts a cast as a shadow ray by its finite t_max: rays towards a directional or
        # environment light have none, so the split differs there; the sum does not)
        assert st["closest_segments"] + st["shadow_segments"] == info["closest_segments"] + info["shadow_segments"]


def _xties():
    with gzip.open(os.path.join(G.GOLD, "xties.rtrs.gz"), "rb") as f:
        sc = rtr.Scene.from_bytes(f.read())
    return sc, G.records("xties_hits.bin", A.HIT_DTYPE)


def test_oracle_resolves_ties_across_transform_chains_like_the_reference():
    """Faces of translated / rotated boxes in the planes of rects the list visits before AND after them: the
    material of the reference's record says which object its walk kept (the later one)."""
    sc, gold = _xties()
    out = G.oracle_records(sc, "rto_hits", gold)
    h = gold["hit"] == 1
    assert h.all() and np.array_equal(out["hit"], gold["hit"])
    assert np.array_equal(out["material"], gold["material"]) and np.array_equal(out["front_face"], gold["front_face"])
    for f in ("t", "p", "n"):
        assert np.array_equal(_bits(out[f]), _bits(gold[f])), f
    assert len(np.unique(gold["material"])) >= 5  # walls before, walls after and boxes all win somewhere


@pytest.fixture(scope="module")
def ctx():
    c = rtr.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_device_equals_the_reference_on_random_scenes(ctx, seed):
    sc = _scene(seed)
    ctx.upload(sc)
    info = rtr.native.validate_scene(sc)
    gold = G.records("random_%s_hits.bin" % seed, A.HIT_DTYPE)
    h = gold["hit"] == 1
    fog = h & np.isin(gold["material"], np.flatnonzero(sc.materials["type"] == A.MAT_ISOTROPIC))
    surf = h & ~fog
    for ref_order in (False, True):
        ctx.reference_order(ref_order)
        out = ctx.test_records("hits", gold)
        ctx.reference_order(False)
        assert np.array_equal(out["hit"], gold["hit"]) and np.array_equal(out["rng_out"], gold["rng_out"]), ref_order
        for f in ("front_face", "material"):
            assert np.array_equal(out[f][h], gold[f][h]), (f, ref_order)
        for f in ("t", "p", "n"):
            assert np.array_equal(_bits(out[f][surf]), _bits(gold[f][surf])), (f, ref_order)
            assert np.allclose(out[f][fog], gold[f][fog], rtol=1e-13, atol=1e-13), (f, ref_order)  # OCML log
        uv = surf & _writes_uv(sc, gold)
        for f in ("u", "v"):
            assert np.allclose(out[f][uv], gold[f][uv], rtol=0, atol=1e-12), (f, ref_order)  # OCML acos / atan2
    worst = 0.0
    pipes = [A.PIPELINE_MEGAKERNEL] + ([A.PIPELINE_WAVEFRONT] if info["fast_ok"] or info["program_steps"] > 0 else [])
    for integ in (1, 4):
        want = _image(seed, integ)
        for pipe in pipes:
            got = ctx.render(A.make_params(W, H, SPP, integrator=integ, seed=100 + int(seed.rstrip("b")), pipeline=pipe))
            err = G.rel_l2(got, want)
            worst = max(worst, err)
            assert err <= 1e-12, (seed, integ, pipe, err)
    G.residue("random%s.vs_reference.worst_rel_l2" % seed, worst, 1e-12)


@pytest.mark.gpu
def test_device_resolves_ties_across_transform_chains_like_the_reference(ctx):
    """The compiled traversal scans instances (one per transform chain) in the order their FIRST primitive is visited;
    where that contradicts the visiting order of two coplanar faces (floor_b sits in the first instance but is
    visited after the box whose bottom lies in its plane) upload flags the pair and their visiting positions decide."""
    sc, gold = _xties()
    ctx.upload(sc)
    assert rtr.native.validate_scene(sc)["fast_ok"]
    for ref_order in (False, True):
        ctx.reference_order(ref_order)
        out = ctx.test_records("hits", gold)
        ctx.reference_order(False)
        assert np.array_equal(out["hit"], gold["hit"]), ref_order
        wrong = int((out["material"] != gold["material"]).sum())
        assert wrong == 0, (ref_order, wrong)
        assert np.array_equal(out["front_face"], gold["front_face"]), ref_order
        for f in ("t", "p", "n"):
            assert np.array_equal(_bits(out[f]), _bits(gold[f])), (f, ref_order)
    want, _ = G.oracle_render(sc, A.make_params(W, H, SPP, integrator=1, seed=9))
    for pipe in (A.PIPELINE_MEGAKERNEL, A.PIPELINE_WAVEFRONT):
        got = ctx.render(A.make_params(W, H, SPP, integrator=1, seed=9, pipeline=pipe))
        assert np.array_equal(_bits(got), _bits(want)), pipe
