"""Random scenes (tests/_randscene.py) through the C ABI against the pinned CPU oracle: hit records of every
traversal and small renders of every integrator on both pipelines.  The golden scenes are the reference's
forty demo scenes; these graphs add what none of them contains -- coplanar overlapping rects (exact ties in t),
media in the middle of the visiting order next to transformed boxes, dozens of small instances, nested lists
under transforms -- with the oracle (bit-exact against the reference on every golden vector) as the checker."""
import numpy as np
import pytest

import _golden as G
import _randscene as R

A = G.A
rtr = G.rtr

CASES = [(11, {}), (12, {}), (13, dict(n_objects=90)), (14, dict(media=True)), (15, dict(media=True, n_objects=60)),
         (16, dict(hollow=True)), (17, dict(n_objects=8, ties=True)), (18, dict(media=True, hollow=True)),
         (19, dict(n_objects=200)), (20, dict(n_objects=40)), (21, dict(media=True, n_objects=12)),
         (22, dict(n_objects=3)), (23, dict(media=True, n_objects=100)), (24, dict(n_objects=60)),
         (25, dict(media=True)), (26, dict(n_objects=30, hollow=True)),
         (27, dict(delta_lights=True)), (28, dict(delta_lights=True, media=True)), (29, dict(delta_lights=True, n_objects=70)),
         (30, dict(lens=0.12)), (31, dict(lens=0.05, media=True, n_objects=40)),
         (32, dict(moved_media=True)), (33, dict(moved_media=True, media=True, n_objects=50))]


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


@pytest.mark.parametrize("seed,kw", CASES)
def test_random_scenes_validate_and_run_on_the_oracle(seed, kw):
    """CPU: the generator's scenes are well-formed and the oracle traces them (no GPU)."""
    sc = R.random_scene(seed, **kw)
    info = rtr.native.validate_scene(sc)
    assert info["has_media"] == bool(kw.get("media") or kw.get("moved_media")) and info["inverted_boxes"] == (1 if kw.get("hollow") else 0)
    if not kw.get("media") and not kw.get("hollow") and not kw.get("moved_media"):
        assert info["fast_ok"] and info["fast_refs"] >= 3
        # dozens of transformed instances sit in a box tree of their own (FSub::top_root): seeds 13 / 19 / 24 / 29
        # are the cases that put the per-lane instance walk in front of the oracle
        assert info["top_trees"] == (1 if kw.get("n_objects", 24) >= 200 else 0)  # seed 19: 80 instances
    if kw.get("media") and not kw.get("hollow") and not kw.get("moved_media"):
        assert info["program_steps"] >= 4  # media under lists only: the step program exists
    if kw.get("hollow"):
        # transformed boxes next to the hollow sphere: no single scan in visiting order -> a guarded step (FStep kind 3)
        assert not info["fast_ok"] and info["program_steps"] >= 2
    if kw.get("moved_media"):
        assert info["program_steps"] >= 4  # a medium under translate / rotate_y: a step with a transform chain
    rays = R.random_rays(seed, 500)
    out = G.oracle_records(sc, "rto_hits", rays)
    assert 50 < int(out["hit"].sum()) <= 500
    img, st = G.oracle_render(sc, A.make_params(24, 16, 2, integrator=4, seed=seed))
    assert np.isfinite(img).all() and img.mean() > 0 and st["samples"] == 24 * 16 * 2


@pytest.fixture(scope="module")
def ctx():
    c = rtr.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kw", CASES)
def test_random_scene_hit_records_equal_the_oracle(ctx, seed, kw):
    sc = R.random_scene(seed, **kw)
    ctx.upload(sc)
    rays = R.random_rays(seed, 6000)
    ora = G.oracle_records(sc, "rto_hits", rays)
    h = ora["hit"] == 1
    # moving_sphere::hit writes no (u,v) (moving_sphere.h:36-62): nothing to compare there
    moving = np.isin(ora["material"], sc.nodes["a"][sc.nodes["type"] == A.NODE_MOVING_SPHERE])
    for exact_order in (False, True):
        ctx.reference_order(exact_order)
        dev = ctx.test_records("hits", rays)
        ctx.reference_order(False)
        tag = "random%02d.%s" % (seed, "reference_order" if exact_order else "compiled")
        assert np.array_equal(dev["hit"], ora["hit"]), tag
        for f in ("front_face", "material", "rng_out"):
            assert np.array_equal(dev[f][h] if f != "rng_out" else dev[f], ora[f][h] if f != "rng_out" else ora[f]), (tag, f)
        # a constant_medium's t is t1 + hit_distance / |d| with hit_distance = -1/density * log(random)
        # (constant_medium.h:88-96): OCML's log against glibc's, last bits; surfaces are +-*/sqrt only
        fog = h & np.isin(ora["material"], np.flatnonzero(sc.materials["type"] == A.MAT_ISOTROPIC))
        surf = h & ~fog
        for f in ("t", "p", "n"):
            bad = int((_bits(dev[f][surf]) != _bits(ora[f][surf])).reshape(int(surf.sum()), -1).any(axis=1).sum())
            assert bad == 0, (tag, f, bad)
            assert np.allclose(dev[f][fog], ora[f][fog], rtol=1e-13, atol=1e-13), (tag, f, "medium")
        uv = surf & ~moving
        # sphere (u,v) go through acos / atan2 (OCML on the device, glibc in the oracle): last bits may differ
        for f in ("u", "v"):
            assert np.allclose(dev[f][uv], ora[f][uv], rtol=0, atol=1e-12), (tag, f)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kw", CASES)
def test_random_scene_renders_equal_the_oracle(ctx, seed, kw):
    sc = R.random_scene(seed, **kw)
    ctx.upload(sc)
    info = rtr.native.validate_scene(sc)
    wavefront = info["fast_ok"] or info["program_steps"] > 0
    worst = 0.0
    for integ in (0, 1, 2, 3, 4):
        p = A.make_params(48, 32, 4, integrator=integ, seed=100 + seed)
        want, wst = G.oracle_render(sc, p)
        runs = [(A.PIPELINE_MEGAKERNEL, 0), (A.PIPELINE_MEGAKERNEL, A.FLAG_REFERENCE_ORDER)]
        if wavefront:
            runs.append((A.PIPELINE_WAVEFRONT, 0))
        for pipe, flags in runs:
            got = ctx.render(A.make_params(48, 32, 4, integrator=integ, seed=100 + seed, pipeline=pipe, flags=flags))
            st = ctx.stats()
            tag = "random%02d.i%d.pipe%d.flags%d" % (seed, integ, pipe, flags)
            assert st["closest_segments"] == wst["closest_segments"] and st["shadow_segments"] == wst["shadow_segments"], tag
            err = G.rel_l2(got, want)
            worst = max(worst, err)
            assert err <= 1e-12, (tag, err)  # same paths (segment counts equal); libm last bits only
    G.residue("random%02d.renders.worst_rel_l2" % seed, worst, 1e-12)


TOP_CASES = [c for c in CASES if not (c[1].get("media") or c[1].get("hollow") or c[1].get("moved_media"))] + \
            [(41, dict(big_group=True)), (42, dict(big_group=True, n_objects=70))]


@pytest.fixture
def forced_top_tree(monkeypatch):
    """RTR_TOP_MIN=2 (read at every upload, rt_compile.h): a sub-scene with two transformed instances already gets its
    top tree, so every media-free random scene goes through the per-lane instance walk (RT_TRAV_TOP)."""
    monkeypatch.setenv("RTR_TOP_MIN", "2")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kw", TOP_CASES)
def test_top_tree_walk_equals_the_oracle(ctx, forced_top_tree, seed, kw):
    """The many-instance path (FSub::top_root, trace_top: one box tree over the transformed instances and the leaves
    of the untransformed instance's tree, walked per lane, instance records through vector loads, packed runs with
    per-lane shared reciprocals) against the oracle's in-order scan: hit records bit for bit -- including the exact ties
    between coplanar rects, which this walk meets in tree order -- and small renders of integrators 1 and 4."""
    sc = R.random_scene(seed, **kw)
    assert rtr.native.validate_scene(sc)["top_trees"] == (1 if rtr.native.validate_scene(sc)["fast_instances"] >= 3 else 0)
    ctx.upload(sc)
    rays = R.random_rays(seed, 6000)
    ora = G.oracle_records(sc, "rto_hits", rays)
    dev = ctx.test_records("hits", rays)
    h = ora["hit"] == 1
    assert np.array_equal(dev["hit"], ora["hit"])
    for f in ("front_face", "material"):
        assert np.array_equal(dev[f][h], ora[f][h]), f
    for f in ("t", "p", "n"):
        assert np.array_equal(_bits(dev[f][h]), _bits(ora[f][h])), f
    for integ in (1, 4):
        p = A.make_params(48, 32, 4, integrator=integ, seed=200 + seed, pipeline=A.PIPELINE_MEGAKERNEL)
        want, wst = G.oracle_render(sc, p)
        got = ctx.render(p)
        st = ctx.stats()
        assert st["closest_segments"] == wst["closest_segments"] and st["shadow_segments"] == wst["shadow_segments"]
        assert G.rel_l2(got, want) <= 1e-12


@pytest.mark.gpu
def test_top_tree_resolves_ties_across_transform_chains_like_the_reference(ctx, forced_top_tree):
    """tests/_randscene.cross_instance_tie_scene: faces of translated boxes in the planes of rects the list visits before
    and after them.  With a top tree the instances are met in tree order, so every such pair carries its visiting
    position (RT_TIE_FLAG) and the later visit wins, as in the reference's in-order scan."""
    sc = R.cross_instance_tie_scene()
    assert rtr.native.validate_scene(sc)["top_trees"] == 1
    ctx.upload(sc)
    rays = R.cross_instance_tie_rays()
    ora = G.oracle_records(sc, "rto_hits", rays)
    dev = ctx.test_records("hits", rays)
    h = ora["hit"] == 1
    assert np.array_equal(dev["hit"], ora["hit"])
    assert np.array_equal(dev["material"][h], ora["material"][h])
    assert np.array_equal(_bits(dev["t"][h]), _bits(ora["t"][h]))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kw", [(16, dict(hollow=True)), (32, dict(moved_media=True))])
def test_traversal_machine_refuses_what_it_cannot_run(ctx, seed, kw):
    """Step programs with a guarded step (hollow sphere under bvh_nodes next to transformed boxes) or a medium under
    translate / rotate_y run on the megakernel and the lockstep wavefront stages (the render test above); the
    persistent-threads machine (RTR_FLAG_WF_PERSISTENT) has no such step and says so instead of rendering something else."""
    ctx.upload(R.random_scene(seed, **kw))
    p = A.make_params(32, 16, 2, integrator=4, seed=1, pipeline=A.PIPELINE_WAVEFRONT, flags=A.FLAG_WF_PERSISTENT)
    with pytest.raises(rtr.RtrError) as e:
        ctx.render(p)
    assert e.value.code == A.RTR_ERR_UNSUPPORTED
    ctx.render(A.make_params(32, 16, 2, integrator=4, seed=1, pipeline=A.PIPELINE_WAVEFRONT))
