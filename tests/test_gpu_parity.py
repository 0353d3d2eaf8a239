"""GPU parity: the HIP path (through the C ABI of include/rtr_hip.h) against the golden vectors
of the unmodified reference and against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star: per-pixel L2 <= 1e-3 vs the reference on identical seeds):
  * scenes whose path uses only + - * / sqrt (the Cornell boxes, scene 07 / 21): BIT-EXACT,
    the device keeps the reference's operation order in IEEE double, no FMA contraction;
  * scenes that call libm (scene 23: sin/cos/pow; scene 9/22: log, sin, floor): OCML and glibc
    may differ in the last ulp, which can flip a rare branch; the bar is rel-L2 <= 1e-3 on linear
    radiance (measured: far below), and a per-record mismatch budget on the unit vectors.
"""
import numpy as np
import pytest

import _golden as G

A = G.A
pytestmark = pytest.mark.gpu

REL_L2_BAR = 1e-3  # BASELINE.json north_star tolerance


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


@pytest.fixture(scope="module")
def ctx(rtr):
    c = rtr.Context(0)
    yield c
    c.close()


def _upload(ctx, sid):
    sc = G.scene(sid)
    ctx.upload(sc)
    return sc


def _wavefront_runs(sc):
    """The wavefront pipeline runs the compiled traversals; graphs that need the reference-order walk
    (hollow spheres, a medium under a transform) are the megakernel's."""
    info = G.rtr.native.validate_scene(sc)
    return info["fast_ok"] or info["program_steps"] > 0


def _render_or_unsupported(ctx, sc, params):
    """Render; a wavefront request for a graph that pipeline does not run must fail with RTR_ERR_UNSUPPORTED."""
    if params.pipeline == A.PIPELINE_WAVEFRONT and (not _wavefront_runs(sc) or params.flags & A.FLAG_REFERENCE_ORDER):
        with pytest.raises(G.rtr.RtrError) as e:
            ctx.render(params)
        assert e.value.code == A.RTR_ERR_UNSUPPORTED
        return None
    return ctx.render(params)


def _rows(mask, how):
    """reduce a per-component boolean array to one value per record (empty arrays included)"""
    mask = np.asarray(mask)
    if mask.ndim == 1:
        return mask
    flat = mask.reshape(mask.shape[0], int(np.prod(mask.shape[1:])))
    return flat.any(axis=1) if how == "any" else flat.all(axis=1)


def _close(a, b, rtol):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) <= rtol * np.maximum(np.abs(b), 1e-300) + 1e-300


@pytest.mark.parametrize("sid", [21, 23, 9, 1, 8, 35])
def test_unit_closest_hit(ctx, sid):
    """Closest hit of whole scenes.  t / p / n / front_face / material involve only + - * / sqrt on the device
    and in the reference, so they are BIT-EXACT on every scene -- except behind a constant_medium, whose free
    path goes through log() (scenes 8, 9: the medium's own records and nothing else).  (u,v) of spheres come
    from acos / atan2 (OCML here, glibc there): exact counts of records outside 1e-12 are recorded."""
    sc = _upload(ctx, sid)
    gold = G.records("hits_scene%02d.bin" % sid, A.HIT_DTYPE)
    out = ctx.test_records("hits", gold)
    h = gold["hit"] == 1
    iso = np.flatnonzero(sc.materials["type"] == A.MAT_ISOTROPIC)
    # a ray decided by a medium's random free path: log() of the draw
    via_medium = np.isin(gold["material"], iso) | np.isin(out["material"], iso) | (out["rng_out"] != out["rng_in"])
    exact = ~via_medium
    assert np.array_equal(out["hit"][exact], gold["hit"][exact])
    assert np.array_equal(out["rng_out"][exact], gold["rng_out"][exact])
    he = h & exact
    for f in ("front_face", "material"):
        assert np.array_equal(out[f][he], gold[f][he]), f
    tie_budget = 0
    for f in ("t", "p", "n"):
        bad = _rows(_bits(out[f][he]) != _bits(gold[f][he]), "any")
        G.residue("hits%02d.%s.not_bit_exact" % (sid, f), int(bad.sum()), 0)
    # rays that went through a medium: same decision and, where it matters, within the log() ulp
    vm = via_medium
    G.residue("hits%02d.medium.hit_flag_differs" % sid, int((out["hit"][vm] != gold["hit"][vm]).sum()), 2)
    G.residue("hits%02d.medium.rng_differs" % sid, int((out["rng_out"][vm] != gold["rng_out"][vm]).sum()), 2)
    bm = vm & h & (out["hit"] == 1) & (out["material"] == gold["material"])
    for f in ("t", "p"):
        ok = _rows(_close(out[f][bm], gold[f][bm], 1e-12), "all")
        G.residue("hits%02d.medium.%s.outside_1e-12" % (sid, f), int((~ok).sum()), 2)
    # (u,v) only where the final primitive writes them: moving_sphere and constant_medium leave
    # whatever an earlier, farther hit of the same walk stored (hittable.h:10-17 is never reset),
    # which depends on visiting order inside compiled subtrees
    no_uv_mats = set(iso.tolist())
    no_uv_mats |= set(sc.nodes["a"][sc.nodes["type"] == A.NODE_MOVING_SPHERE].tolist())
    both = h & (out["hit"] == 1) & (out["material"] == gold["material"])
    uv = both & ~np.isnan(gold["u"]) & ~np.isin(gold["material"], sorted(no_uv_mats))
    # (a ray through the coplanar side faces of two adjacent ground boxes of scene 9 ties exactly in t; the winner
    # -- in the reference: 1-ulp noise of its BVH box tests -- only changes the face-relative (u,v))
    for f in ("u", "v"):
        bits_differ = int((_bits(out[f][uv]) != _bits(gold[f][uv])).sum())
        outside = int((~_close(out[f][uv], gold[f][uv], 1e-12)).sum())
        G.residue("hits%02d.%s.not_bit_exact" % (sid, f), bits_differ, int(0.05 * uv.sum()) + 1 if sid != 21 else 0)
        G.residue("hits%02d.%s.outside_1e-12" % (sid, f), outside, int(0.015 * uv.sum()) + 1 if sid != 21 else 0)
    del tie_budget


@pytest.mark.parametrize("sid", [23, 9, 35, 1011])
def test_unit_materials(ctx, sid):
    sc = _upload(ctx, sid)
    gold = G.records("materials_scene%02d.bin" % sid, A.MAT_DTYPE)
    out = ctx.test_records("materials", gold)
    types = sc.materials["type"][gold["material"]]
    assert np.array_equal(out["sample_ok"], gold["sample_ok"])
    assert np.array_equal(out["rng_out"], gold["rng_out"])
    assert np.array_equal(out["is_transmission"], gold["is_transmission"])
    writes = np.isin(types, [A.MAT_LAMBERTIAN, A.MAT_METAL, A.MAT_DIELECTRIC])
    ok = writes | ((types == A.MAT_PBR) & (gold["sample_ok"] == 1))
    for f in ("s_wi", "s_f", "s_pdf"):
        assert np.all(_close(out[f][ok], gold[f][ok], 1e-11)), f
    for f in ("eval", "pdf", "emitted"):
        assert np.all(_close(out[f], gold[f], 1e-11)), f
    # transcendental-free materials are bit-exact
    exact = np.isin(types, [A.MAT_METAL]) | ((types == A.MAT_LAMBERTIAN) & (sid == 23))
    for f in ("s_wi", "s_f", "s_pdf", "eval", "pdf"):
        assert np.array_equal(_bits(out[f][exact]), _bits(gold[f][exact])), f


@pytest.mark.parametrize("sid", [21, 23])
def test_unit_lights(ctx, sid):
    _upload(ctx, sid)
    gold = G.records("lights_scene%02d.bin" % sid, A.LIGHTREC_DTYPE)
    out = ctx.test_records("lights", gold)
    for f in ("Li", "wi", "pdf", "dist", "pdf_dir"):  # + - * / sqrt only
        assert np.array_equal(_bits(out[f]), _bits(gold[f])), f


@pytest.mark.parametrize("sid,integ", [(7, 1), (7, 4), (21, 4), (23, 4), (9, 1), (22, 4), (1, 1), (8, 1), (35, 4)])
def test_li_records(ctx, sid, integ):
    """Per camera sample: radiance, RNG state at exit (pins the draw count), segment counts."""
    name = "li_scene%02d_i%d.bin" % (sid, integ)
    info = G.MANIFEST["files"][name]
    _upload(ctx, sid)
    gold = G.records(name, A.LI_DTYPE)
    p = A.make_params(info["info"]["width"], info["info"]["height"], info["spp"], integrator=integ,
                      seed=info["seed"])
    out = ctx.test_records("li", gold, params=p)
    if sid in (7, 21):
        assert np.array_equal(out["rng_exit"], gold["rng_exit"])
        assert np.array_equal(out["n_closest"], gold["n_closest"])
        assert np.array_equal(out["n_shadow"], gold["n_shadow"])
        assert np.array_equal(_bits(out["L"]), _bits(gold["L"]))
        return
    # every sample whose path met no libm call is bit-exact; the others are counted
    same_path = (out["rng_exit"] == gold["rng_exit"]) & (out["n_closest"] == gold["n_closest"]) & \
                (out["n_shadow"] == gold["n_shadow"])
    tag = "li%02d_i%d" % (sid, integ)
    G.residue(tag + ".other_path", int((~same_path).sum()), int(0.005 * len(gold)))
    bit_exact = np.all(_bits(out["L"]) == _bits(gold["L"]), axis=1)
    G.residue(tag + ".L_not_bit_exact", int((~bit_exact).sum()), len(gold))
    ok = np.all(_close(out["L"], gold["L"], 1e-9), axis=1)
    G.residue(tag + ".L_outside_1e-9_same_path", int((~ok & same_path).sum()), int(0.001 * len(gold)) + 1)
    rel = np.abs(out["L"][same_path] - gold["L"][same_path]) / np.maximum(np.abs(gold["L"][same_path]), 1e-300)
    G.residue(tag + ".max_rel_err_same_path", float(rel.max()) if rel.size else 0.0, 1e-9)
    G.residue(tag + ".rel_l2", G.rel_l2(out["L"], gold["L"]), 5e-2)


def test_per_ray_entry_of_the_main_abi(ctx):
    """rtr_li_samples (include/rtr_hip.h): Integrator::Li per camera sample, the per-ray entry of the boundary
    (renderer/integrator.h:12-19) -- against the reference's per-sample records and against a render."""
    _upload(ctx, 21)
    name = "li_scene21_i4.bin"
    info = G.MANIFEST["files"][name]
    gold = G.records(name, A.LI_DTYPE)
    p = A.make_params(info["info"]["width"], info["info"]["height"], info["spp"], integrator=4, seed=info["seed"])
    ijs = np.stack([gold["i"], gold["j"], gold["s"]], axis=1)
    L = ctx.li_samples(p, ijs)
    assert np.array_equal(_bits(L), _bits(gold["L"]))
    # mean of a pixel's samples == the pixel of a render (one running sum, renderer.h:72-79)
    img = ctx.render(A.make_params(p.image_width, p.image_height, 4, integrator=4, seed=info["seed"], spp_chunks=1))
    i, j = 17, 23
    acc = np.zeros(3)
    for row in ctx.li_samples(p, [[i, j, s] for s in range(4)]):
        acc = acc + row
    assert np.array_equal(_bits((1.0 / 4) * acc), _bits(img[j, i]))
    with pytest.raises(G.rtr.RtrError):
        ctx.li_samples(p, [[p.image_width, 0, 0]])


def _camera_rays(sc, p, ijs):
    """camera::get_ray (renderer/camera.h:32-40) of camera samples on the host, in the reference's operation order
    (numpy float64 = IEEE binary64): origins, directions, times and the generator state after the ray was made."""
    cam = sc.camera[0]
    org, llc = np.array(cam["origin"], dtype=np.float64), np.array(cam["lower_left_corner"], dtype=np.float64)
    hor, ver = np.array(cam["horizontal"], dtype=np.float64), np.array(cam["vertical"], dtype=np.float64)
    cu, cv = np.array(cam["u"], dtype=np.float64), np.array(cam["v"], dtype=np.float64)
    lens, t0, t1 = float(cam["lens_radius"]), float(cam["time0"]), float(cam["time1"])
    lib = G.rtr.native.lib()
    M = 0xFFFFFFFF

    def nxt(state):
        state ^= (state << 13) & M
        state ^= state >> 17
        state ^= (state << 5) & M
        return state, np.float64(state) * np.float64(2.3283064365386963e-10)

    o, d, tm, st = [], [], [], []
    W, H = p.image_width, p.image_height
    for i, j, s in ijs:
        state = lib.rtr_sample_seed(p.seed, W, int(i), int(j), int(s))
        state, r = nxt(state)
        u = (np.float64(i) + r) / np.float64(W - 1)
        state, r = nxt(state)
        v = (np.float64(j) + r) / np.float64(H - 1)
        while True:  # random_in_unit_disk (vec3.h:250-257): y takes the first draw
            state, r = nxt(state)
            y = np.float64(-1.0) + np.float64(2.0) * r
            state, r = nxt(state)
            x = np.float64(-1.0) + np.float64(2.0) * r
            if x * x + y * y + np.float64(0.0) < 1:
                break
        rd = np.array([lens * x, lens * y, lens * np.float64(0.0)])
        offset = rd[0] * cu + rd[1] * cv
        direction = llc + u * hor + v * ver - org - offset
        state, r = nxt(state)
        o.append(org + offset), d.append(direction), tm.append(np.float64(t0) + (np.float64(t1) - np.float64(t0)) * r)
        st.append(state)
    return np.array(o), np.array(d), np.array(tm), np.array(st, dtype=np.uint32)


@pytest.mark.parametrize("name,sid,integ", [("li_scene21_i4.bin", 21, 4), ("li_scene07_i1.bin", 7, 1), ("li_scene09_i1.bin", 9, 1)])
def test_arbitrary_ray_entry_of_the_main_abi(ctx, name, sid, integ):
    """rtr_li_rays (include/rtr_hip.h): Integrator::Li takes ANY ray (renderer/integrator.h:12-19).  Fed the camera rays of
    the reference's per-sample records -- made on the host, generator state after camera::get_ray included -- it must
    return the records' radiance bit for bit (scene 9: media draw inside the casts, so the state matters all the way)."""
    sc = _upload(ctx, sid)
    info = G.MANIFEST["files"][name]
    gold = G.records(name, A.LI_DTYPE)[:512]
    p = A.make_params(info["info"]["width"], info["info"]["height"], info["spp"], integrator=integ, seed=info["seed"])
    ijs = np.stack([gold["i"], gold["j"], gold["s"]], axis=1)
    o, d, tm, st = _camera_rays(sc, p, ijs)
    L = ctx.li_rays(p, o, d, tm, st)
    same = ctx.li_samples(p, ijs)
    assert np.array_equal(_bits(L), _bits(same)), "host-made camera rays differ from the device's"
    if sid != 9:
        assert np.array_equal(_bits(L), _bits(gold["L"]))
    else:  # OCML log / sin vs glibc: the recorded residue of test_li_records applies
        assert np.allclose(L, gold["L"], rtol=1e-9, atol=1e-12)
    with pytest.raises(G.rtr.RtrError):
        ctx.li_rays(p, o[:1], d[:1], tm[:1], np.zeros(1, dtype=np.uint32))


IMG_CASES = ["img_scene07_i1_64_spp16.f64", "img_scene07_i4_64_spp16.f64", "img_scene21_i4_64_spp16.f64",
             "img_scene23_i4_64_spp16.f64", "img_scene09_i1_64_spp16.f64", "img_scene22_i4_64_spp16.f64",
             "img_scene21_i4_128_spp32.f64", "img_scene01_i1_64_spp16.f64", "img_scene08_i1_64_spp16.f64",
             "img_scene35_i4_64_spp16.f64"]


@pytest.mark.parametrize("name", IMG_CASES)
@pytest.mark.parametrize("pipeline", [A.PIPELINE_MEGAKERNEL, A.PIPELINE_WAVEFRONT])
def test_images_vs_reference(ctx, name, pipeline):
    """Whole render through rtr_render_host vs the reference's image (golden) and the oracle."""
    img, info = G.image(name)
    sc = _upload(ctx, info["scene"])
    p = A.make_params(info["width"], info["height"], info["spp"], integrator=info["integrator"], seed=info["seed"],
                      pipeline=pipeline, spp_chunks=1)
    out = ctx.render(p)
    st = ctx.stats()
    assert st["samples"] == info["width"] * info["height"] * info["spp"]
    err = G.rel_l2(out, img)
    if info["scene"] in (7, 21):
        assert np.array_equal(_bits(out), _bits(img)), "rel L2 %.3e" % err
        assert st["closest_segments"] == info["info"]["closest_segments"]
        assert st["shadow_segments"] == info["info"]["shadow_segments"]
    else:
        assert err <= REL_L2_BAR, err
        assert abs(st["closest_segments"] - info["info"]["closest_segments"]) <= 1e-3 * st["closest_segments"]
    # reference output quantity: sqrt-gamma, clamped (renderer.h:126-140)
    g_out = np.clip(np.sqrt(out), 0, 1)
    g_ref = np.clip(np.sqrt(img), 0, 1)
    assert np.sqrt(np.mean((g_out - g_ref) ** 2)) <= REL_L2_BAR
    ora, _ = G.oracle_render(sc, p, threads=0)
    assert G.rel_l2(out, ora) <= (0.0 if info["scene"] in (7, 21) else REL_L2_BAR)


@pytest.mark.parametrize("pipeline", [A.PIPELINE_MEGAKERNEL, A.PIPELINE_WAVEFRONT])
def test_headline_scene_mid_size_vs_oracle(ctx, pipeline):
    """scene21 / MIS at 200x200 spp 8 (ragged: 200 is not a multiple of 16), seeded, vs the oracle."""
    sc = _upload(ctx, 21)
    p = A.make_params(200, 200, 8, seed=99, pipeline=pipeline, spp_chunks=1)
    out = ctx.render(p)
    ora, ost = G.oracle_render(sc, p, threads=0)
    assert np.array_equal(_bits(out), _bits(ora))
    st = ctx.stats()
    assert st["closest_segments"] == ost["closest_segments"] and st["shadow_segments"] == ost["shadow_segments"]


@pytest.mark.parametrize("pipeline", [A.PIPELINE_MEGAKERNEL, A.PIPELINE_WAVEFRONT])
def test_properties_at_full_size(ctx, pipeline):
    """BASELINE config C2 size (800x800) at reduced spp: size-independent properties."""
    _upload(ctx, 21)
    W = H = 800
    base = A.make_params(W, H, 4, seed=5, pipeline=pipeline, spp_chunks=1)
    full = ctx.render(base)
    assert np.isfinite(full).all() and full.min() >= 0
    # determinism
    assert np.array_equal(full, ctx.render(base))
    # tile sharding: ranks own disjoint tiles, union is bit-identical to 1 rank
    acc = np.full_like(full, np.nan)
    import ctypes as C
    for r in range(4):
        p = A.make_params(W, H, 4, seed=5, pipeline=pipeline, spp_chunks=1, tile_first=r, tile_stride=4)
        part = np.full_like(full, np.nan)
        ctx._chk(ctx._L.rtr_render_host(ctx._h, C.byref(p), part.ctypes.data, W))
        own = ~np.isnan(part[..., 0])
        assert not np.any(own & ~np.isnan(acc[..., 0]))
        acc[own] = part[own]
    assert np.array_equal(acc, full)
    # a sub-region equals the same pixels of the full image
    sub = ctx.render(A.make_params(W, H, 4, seed=5, pipeline=pipeline, spp_chunks=1, region=(100, 333, 421, 590)))
    assert np.array_equal(sub, full[333:590, 100:421])
    # chunked summation only reorders the per-pixel sum
    ch = ctx.render(A.make_params(W, H, 4, seed=5, pipeline=pipeline, spp_chunks=4))
    assert G.rel_l2(ch, full) <= 1e-14
    # another seed is another (statistically equal) estimate
    other = ctx.render(A.make_params(W, H, 4, seed=6, pipeline=pipeline, spp_chunks=1))
    assert not np.array_equal(other, full)
    assert abs(other.mean() - full.mean()) <= 0.05 * full.mean()


@pytest.mark.parametrize("sid", [7, 21, 23, 1, 4, 35])
def test_compiled_scene_equals_reference_order(ctx, sid):
    """Scenes without media: the order-free compiled-scene traversal (default) and the
    reference-order traversal (RTR_FLAG_REFERENCE_ORDER) must agree bit for bit."""
    sc = _upload(ctx, sid)
    info = rtr_info(sc)
    assert info["fast_ok"] and not info["has_media"]
    gold = G.records("hits_scene%02d.bin" % (21 if sid == 7 else sid), A.HIT_DTYPE)
    if sid == 1:
        assert info["fast_stack_words"] > 1  # 462 spheres: the instance has a box tree
    ctx.reference_order(False)
    fast = ctx.test_records("hits", gold)
    ctx.reference_order(True)
    exact = ctx.test_records("hits", gold)
    ctx.reference_order(False)
    assert np.array_equal(fast["hit"], exact["hit"])
    h = exact["hit"] == 1
    for f in ("front_face", "material"):
        assert np.array_equal(fast[f][h], exact[f][h]), f
    for f in ("t", "p", "n"):
        assert np.array_equal(_bits(fast[f][h]), _bits(exact[f][h])), f
    # moving_sphere::hit writes no (u,v): the reference-order walk then still holds those of an
    # earlier, farther hit, the compiled path the caller's initial value
    sets_uv = h & ~np.isin(exact["material"], sc.nodes["a"][sc.nodes["type"] == A.NODE_MOVING_SPHERE])
    for f in ("u", "v"):
        assert np.array_equal(_bits(fast[f][sets_uv]), _bits(exact[f][sets_uv])), f
    integ = 1 if sid in (7, 1, 4) else 4  # 4 and 35: image textures, (u,v) rebuilt after the order-free cast
    b = ctx.render(A.make_params(96, 64, 6, integrator=integ, seed=21, pipeline=A.PIPELINE_MEGAKERNEL,
                                 flags=A.FLAG_REFERENCE_ORDER))
    sb = ctx.stats()
    for pipe in (A.PIPELINE_MEGAKERNEL, A.PIPELINE_WAVEFRONT):  # compiled traversal, both pipelines
        a = ctx.render(A.make_params(96, 64, 6, integrator=integ, seed=21, pipeline=pipe))
        sa = ctx.stats()
        assert np.array_equal(_bits(a), _bits(b)), pipe
        assert sa["closest_segments"] == sb["closest_segments"] and sa["shadow_segments"] == sb["shadow_segments"]
    with pytest.raises(G.rtr.RtrError) as e:  # the wavefront stages run compiled traversals only
        ctx.render(A.make_params(32, 32, 1, integrator=integ, pipeline=A.PIPELINE_WAVEFRONT, flags=A.FLAG_REFERENCE_ORDER))
    assert e.value.code == A.RTR_ERR_UNSUPPORTED


def rtr_info(sc):
    return G.rtr.native.validate_scene(sc)


@pytest.mark.parametrize("sid,integ", [(8, 1), (9, 1), (22, 4), (22, 3)])
def test_media_program_equals_reference_order(ctx, sid, integ):
    """Scenes with media run the step program (media in the reference's order, the media-free
    runs between them compiled); it must reproduce the reference-order walk bit for bit, RNG
    draws of constant_medium::hit included."""
    sc = _upload(ctx, sid)
    info = rtr_info(sc)
    assert info["has_media"] and not info["fast_ok"] and info["program_steps"] >= 3
    gold = G.records("hits_scene%02d.bin" % (9 if sid == 22 else sid), A.HIT_DTYPE)
    if sid != 22:
        ctx.reference_order(False)
        prog = ctx.test_records("hits", gold)
        ctx.reference_order(True)
        walk = ctx.test_records("hits", gold)
        ctx.reference_order(False)
        assert np.array_equal(prog["hit"], walk["hit"])
        assert np.array_equal(prog["rng_out"], walk["rng_out"])
        h = walk["hit"] == 1
        for f in ("front_face", "material"):
            assert np.array_equal(prog[f][h], walk[f][h]), f
        for f in ("t", "p", "n"):
            assert np.array_equal(_bits(prog[f][h]), _bits(walk[f][h])), f
    b = ctx.render(A.make_params(96, 64, 6, integrator=integ, seed=21, pipeline=A.PIPELINE_MEGAKERNEL,
                                 flags=A.FLAG_REFERENCE_ORDER))
    sb = ctx.stats()
    for pipe in (A.PIPELINE_MEGAKERNEL, A.PIPELINE_WAVEFRONT):  # the step program: lockstep (megakernel) and as the
        a = ctx.render(A.make_params(96, 64, 6, integrator=integ, seed=21, pipeline=pipe))  # resumable machine (wavefront)
        sa = ctx.stats()
        assert np.array_equal(_bits(a), _bits(b)), pipe
        assert sa["closest_segments"] == sb["closest_segments"], pipe
        # a blocked shadow ray of the machine stops at its first hit once no medium is left to draw: same count
        assert sa["shadow_segments"] == sb["shadow_segments"], pipe


N1_CASES = [(7, 0, "img_scene07_i0_48_spp8.f64"), (23, 2, "img_scene23_i2_64_spp16.f64"),
            (21, 3, "img_scene21_i3_64_spp16.f64"), (23, 3, "img_scene23_i3_64_spp16.f64")]


@pytest.mark.parametrize("sid,integ,img_name", N1_CASES)
def test_next_integrators(ctx, sid, integ, img_name):
    """PathIntegrator (0), PBRPathIntegrator (2), DirectLightIntegrator (3) on the
    megakernel, vs the reference's per-sample records and images.  Integrator 0 sums the
    reference's nested `emitted + attenuation * Li(...)` front to back, so it is compared with a
    1e-12 relative tolerance; scene21/i3 has no libm on its path and is bit-exact."""
    name = "li_scene%02d_i%d.bin" % (sid, integ)
    info = G.MANIFEST["files"][name]
    sc = _upload(ctx, sid)
    gold = G.records(name, A.LI_DTYPE)
    for flags in (0, A.FLAG_REFERENCE_ORDER):
        p = A.make_params(info["info"]["width"], info["info"]["height"], info["spp"], integrator=integ,
                          seed=info["seed"], flags=flags)
        out = ctx.test_records("li", gold, params=p)
        same = (out["rng_exit"] == gold["rng_exit"]) & (out["n_closest"] == gold["n_closest"]) & \
               (out["n_shadow"] == gold["n_shadow"])
        if sid in (7, 21):
            assert same.all()
        else:  # OCML vs glibc ulps: the measured number of samples that took another path
            G.residue("next_integrators.scene%02d.i%d.flags%d.paths_differ" % (sid, integ, flags), int((~same).sum()),
                      int(0.005 * same.size))
        if (sid, integ) == (21, 3):
            assert np.array_equal(_bits(out["L"]), _bits(gold["L"]))
        else:
            assert np.all(_close(out["L"][same], gold["L"][same], 1e-9).all(axis=1))
    img, iinfo = G.image(img_name)
    p = A.make_params(iinfo["width"], iinfo["height"], iinfo["spp"], integrator=integ, seed=iinfo["seed"])
    out = ctx.render(p)
    assert G.rel_l2(out, img) <= (1e-12 if sid in (7, 21) else REL_L2_BAR)
    ora, _ = G.oracle_render(sc, p, threads=0)
    assert G.rel_l2(out, ora) <= (1e-12 if sid in (7, 21) else REL_L2_BAR)
    # the wavefront stages run all five integrators: same bits as the megakernel
    p.pipeline = A.PIPELINE_WAVEFRONT
    wf = ctx.render(p)
    assert ctx.stats()["pipeline"] == A.PIPELINE_WAVEFRONT
    assert np.array_equal(_bits(wf), _bits(out))


@pytest.mark.parametrize("sid", [15, 17, 18, 19, 24, 26])
def test_delta_and_environment_lights(ctx, sid):
    """PointLight (scene 15), DirectionalLight (17), SpotLight (18), EnvironmentLight
    without a map (19), with an equirectangular HDR map (24) and with an angular-probe map (26):
    light records, per-sample records and images vs the reference, both pipelines."""
    sc = _upload(ctx, sid)
    gold = G.records("lights_scene%02d.bin" % sid, A.LIGHTREC_DTYPE)
    out = ctx.test_records("lights", gold)
    if sid in (24, 26):  # sin / cos / acos / atan2 come from OCML here, glibc there
        assert np.array_equal(out["pdf"] == 0, gold["pdf"] == 0)
        for f in ("Li", "pdf", "pdf_dir"):  # an ulp in (u,v) can select the neighbouring texel: rare
            same_nan = np.isnan(out[f]) & np.isnan(gold[f])  # pdf() of the zero direction of a failed sample
            ok = _close(out[f], gold[f], 1e-9) | same_nan
            G.residue("lights_scene%02d.%s.records_outside_1e-9" % (sid, f), int((~ok.reshape(len(ok), -1).all(axis=1)).sum()),
                      int(0.01 * len(ok)))
        assert np.abs(out["wi"] - gold["wi"]).max() <= 1e-12
        assert np.all(np.isinf(out["dist"]))
    else:
        for f in ("Li", "wi", "pdf", "dist", "pdf_dir"):
            assert np.array_equal(_bits(out[f]), _bits(gold[f])), f
    assert np.array_equal(out["is_delta"], gold["is_delta"])
    cases = [(4, "img_scene%02d_i4_64_spp16.f64" % sid)] + \
            ([(3, "img_scene%02d_i3_64_spp16.f64" % sid)] if sid in (18, 19, 24, 26) else [])  # 19: uniform EnvironmentLight
    for integ, img_name in cases:
        name = "li_scene%02d_i%d.bin" % (sid, integ)
        info = G.MANIFEST["files"][name]
        grec = G.records(name, A.LI_DTYPE)
        p = A.make_params(info["info"]["width"], info["info"]["height"], info["spp"], integrator=integ,
                          seed=info["seed"])
        o = ctx.test_records("li", grec, params=p)
        # (the harness files a directional light's shadow rays, t_max = inf, under "closest")
        same = (o["rng_exit"] == grec["rng_exit"]) & \
               (o["n_closest"] + o["n_shadow"] == grec["n_closest"] + grec["n_shadow"])
        G.residue("lights_scene%02d.i%d.paths_differ" % (sid, integ), int((~same).sum()), int(0.005 * same.size))
        assert np.all(_close(o["L"][same], grec["L"][same], 1e-9).all(axis=1))
        img, iinfo = G.image(img_name)
        for pipe in (A.PIPELINE_MEGAKERNEL, A.PIPELINE_WAVEFRONT):
            q = A.make_params(iinfo["width"], iinfo["height"], iinfo["spp"], integrator=integ, seed=iinfo["seed"],
                              pipeline=pipe)
            assert G.rel_l2(ctx.render(q), img) <= REL_L2_BAR


ALL_OTHER_SCENES = [2, 5, 6, 10, 11, 12, 13, 14, 16, 20, 25, 27, 28, 30, 31, 32, 33, 34, 36, 37, 38, 39, 40, 41, 42]


@pytest.mark.parametrize("sid", ALL_OTHER_SCENES)
def test_every_other_reference_scene(ctx, sid):
    """Breadth: each remaining scene id of the reference's select_scene, rendered small with the MIS
    integrator by the reference (golden), the oracle and the device (both pipelines).  At 32 x H x 4
    samples one path that takes another branch on an OCML-vs-glibc ulp would dominate a whole-image
    norm, so the bar is per pixel: >= 99 % of the pixels within 1e-9, the image within 5e-2."""
    img, info = G.image("img_scene%02d_i4_32_spp4.f64" % sid)
    sc = _upload(ctx, sid)
    for pipe in (A.PIPELINE_MEGAKERNEL, A.PIPELINE_WAVEFRONT):
        p = A.make_params(info["width"], info["height"], info["spp"], integrator=4, seed=info["seed"], pipeline=pipe)
        out = _render_or_unsupported(ctx, sc, p)
        if out is None:
            continue
        st = ctx.stats()
        assert st["samples"] == info["width"] * info["height"] * info["spp"]
        close = np.all(_close(out, img, 1e-9) | (np.abs(out - img) <= 1e-12), axis=-1)
        G.residue("scene%02d.pipe%d.pixels_outside_1e-9" % (sid, pipe), int((~close).sum()), int(0.01 * close.size))
        G.residue("scene%02d.pipe%d.rel_l2" % (sid, pipe), G.rel_l2(out, img), 5e-2)


@pytest.mark.parametrize("sid", list(range(1001, 1011)))
def test_primitive_hit_vectors(ctx, sid):
    """SURVEY 8c item 2 on the device: one object of each geometry class as the whole world (sphere,
    moving_sphere, the three rects, box, translate(rotate_y(box)), flip_face, constant_medium x 2), 256 rays
    each from the reference; whatever traversal upload picks, and the reference-order walk."""
    sc = _upload(ctx, sid)
    gold = G.records("hits_scene%d.bin" % sid, A.HIT_DTYPE)
    exact = sid not in (1001, 1009)  # spheres: (u,v) through acos / atan2, the medium through log()
    for ref_order in (False, True):
        ctx.reference_order(ref_order)
        out = ctx.test_records("hits", gold)
        ctx.reference_order(False)
        assert np.array_equal(out["hit"], gold["hit"]) and np.array_equal(out["rng_out"], gold["rng_out"])
        h = gold["hit"] == 1
        for f in ("front_face", "material"):
            assert np.array_equal(out[f][h], gold[f][h]), f
        for f in ("t", "p", "n"):
            if exact or sid == 1001:
                assert np.array_equal(_bits(out[f][h]), _bits(gold[f][h])), f
            else:
                assert np.all(_close(out[f][h], gold[f][h], 1e-12)), f
        uv = h & ~np.isnan(gold["u"])
        for f in ("u", "v"):
            if sid == 1001:
                assert np.all(np.abs(out[f][uv] - gold[f][uv]) <= 1e-14), f
            else:
                assert np.array_equal(_bits(out[f][uv]), _bits(gold[f][uv])), f


@pytest.mark.parametrize("sid", [1012, 1013])
def test_exact_ties_in_t_resolve_like_the_reference(ctx, sid):
    """Coplanar overlapping rects, a box face in a rect's plane, the same sphere twice (harness scenes 1012: a
    hittable_list, 1013: the same objects under a bvh_node): the material of the reference's hit record says which
    object its walk kept.  The compiled traversal (tie-capable references carry their visiting position) and the
    reference-order walk must keep the same one; a small render per pipeline on top."""
    sc = _upload(ctx, sid)
    gold = G.records("hits_scene%d.bin" % sid, A.HIT_DTYPE)
    h = gold["hit"] == 1
    assert G.rtr.native.validate_scene(sc)["fast_ok"]
    for ref_order in (False, True):
        ctx.reference_order(ref_order)
        out = ctx.test_records("hits", gold)
        ctx.reference_order(False)
        assert np.array_equal(out["hit"], gold["hit"]), ref_order
        for f in ("front_face", "material"):
            assert np.array_equal(out[f][h], gold[f][h]), (f, ref_order)
        for f in ("t", "p", "n"):
            assert np.array_equal(_bits(out[f][h]), _bits(gold[f][h])), (f, ref_order)
    want, wst = G.oracle_render(sc, A.make_params(64, 64, 4, integrator=1, seed=5))
    for pipe, flags in ((A.PIPELINE_MEGAKERNEL, 0), (A.PIPELINE_MEGAKERNEL, A.FLAG_REFERENCE_ORDER), (A.PIPELINE_WAVEFRONT, 0)):
        got = ctx.render(A.make_params(64, 64, 4, integrator=1, seed=5, pipeline=pipe, flags=flags))
        assert np.array_equal(_bits(got), _bits(want)), (pipe, flags)
        assert ctx.stats()["closest_segments"] == wst["closest_segments"]


EVERY_SCENE = sorted(set(ALL_OTHER_SCENES) | {1, 4, 7, 8, 9, 15, 17, 18, 19, 21, 22, 23, 24, 26, 35})


@pytest.mark.parametrize("sid", EVERY_SCENE)
def test_default_traversal_equals_reference_order_on_every_scene(ctx, sid):
    """Whatever upload picks for a scene (flat / compiled scene / step program / walk with compiled
    subtrees) must give the bits of the reference-order walk: 3072 random rays through the scene
    (hit flag, t, p, n, material, front_face, RNG state) and a small render per pipeline."""
    sc = _upload(ctx, sid)
    cam = np.array(sc.camera["origin"], dtype=np.float64).reshape(3)
    corner = np.array(sc.camera["lower_left_corner"], dtype=np.float64).reshape(3)
    hor = np.array(sc.camera["horizontal"], dtype=np.float64).reshape(3)
    ver = np.array(sc.camera["vertical"], dtype=np.float64).reshape(3)
    rng = np.random.default_rng(1000 + sid)
    n = 3072

    def both(rays):
        ctx.reference_order(False)
        a = ctx.test_records("hits", rays)
        ctx.reference_order(True)
        b = ctx.test_records("hits", rays)
        ctx.reference_order(False)
        assert np.array_equal(a["hit"], b["hit"]) and np.array_equal(a["rng_out"], b["rng_out"])
        h = b["hit"] == 1
        for f in ("front_face", "material"):
            assert np.array_equal(a[f][h], b[f][h]), f
        for f in ("t", "p", "n"):
            assert np.array_equal(_bits(a[f][h]), _bits(b[f][h])), f
        return b

    # rays a render can produce: camera rays, then rays leaving the surfaces those hit (the artificial
    # alternative, origins inside solids or below floors, meets exact ties the reference itself
    # resolves by 1-ulp noise of unpadded box tests -- see RT_TIE_FLAG in rt_device.h)
    rays = np.zeros(n, dtype=A.HIT_DTYPE)
    uv = rng.random((n, 2))
    rays["o"] = cam
    rays["d"] = corner + uv[:, :1] * hor + uv[:, 1:] * ver - cam
    rays["time"] = rng.random(n)
    rays["t_min"], rays["t_max"], rays["rng_in"] = 0.001, np.inf, 12345
    first = both(rays)
    assert 0.05 < first["hit"].mean(), "the camera rays miss the scene"
    h = first["hit"] == 1
    second = rays[h].copy()
    d = rng.normal(0.0, 1.0, (len(second), 3))
    d = np.where((np.sum(d * first["n"][h], axis=1) < 0)[:, None], -d, d)
    second["o"], second["d"] = first["p"][h], d
    both(second)
    integ = 4 if len(sc.lights) else 1
    y = ctx.render(A.make_params(48, 32, 2, integrator=integ, seed=77, pipeline=A.PIPELINE_MEGAKERNEL,
                                 flags=A.FLAG_REFERENCE_ORDER))
    for pipe in (A.PIPELINE_MEGAKERNEL, A.PIPELINE_WAVEFRONT):
        x = _render_or_unsupported(ctx, sc, A.make_params(48, 32, 2, integrator=integ, seed=77, pipeline=pipe))
        assert x is None or np.array_equal(_bits(x), _bits(y)), pipe


@pytest.mark.parametrize("sid", EVERY_SCENE)
def test_other_integrators_on_every_scene(ctx, sid):
    """Breadth for integrators 0-3 (plain path, roulette path, BSDF-only, NEE): every scene, small
    render, device vs the oracle (which the golden vectors pin to the reference for all five
    integrators).  Same per-pixel bar as test_every_other_reference_scene."""
    sc = _upload(ctx, sid)
    W, H = 32, 20
    for integ in (0, 1, 2, 3):
        p = A.make_params(W, H, 3, integrator=integ, seed=5, max_depth=12 if integ == 0 else 50)
        ora, ost = G.oracle_render(sc, p, threads=0)
        for pipe in (A.PIPELINE_MEGAKERNEL, A.PIPELINE_WAVEFRONT):
            q = A.make_params(W, H, 3, integrator=integ, seed=5, max_depth=12 if integ == 0 else 50, pipeline=pipe)
            out = _render_or_unsupported(ctx, sc, q)
            if out is None:
                continue
            close = np.all(_close(out, ora, 1e-9) | (np.abs(out - ora) <= 1e-12), axis=-1)
            G.residue("other_integrators.scene%02d.i%d.pipe%d.pixels_outside_1e-9" % (sid, integ, pipe), int((~close).sum()),
                      int(0.015 * close.size))
            assert G.rel_l2(out, ora) <= 5e-2 or np.abs(out - ora).max() <= 1e-12, (integ, pipe)


def test_image_texture(ctx):
    """image_texture with real texels (scene 4 + synthetic picture): (u,v) from
    acos/atan2 on the sphere, nearest-texel fetch.  Runs the compiled traversal, which rebuilds
    (u,v) for the winning primitive (test_compiled_scene_equals_reference_order covers both)."""
    sc = _upload(ctx, 4)
    assert G.rtr.native.validate_scene(sc)["needs_uv"]
    gold = G.records("hits_scene04.bin", A.HIT_DTYPE)
    out = ctx.test_records("hits", gold)
    assert np.array_equal(out["hit"], gold["hit"])
    h = gold["hit"] == 1
    for f in ("t", "p", "n"):
        assert np.array_equal(_bits(out[f][h]), _bits(gold[f][h])), f
    for f in ("u", "v"):  # (atan2 + pi) / 2pi cancels near u = 0: absolute, not relative, agreement
        assert np.all(np.abs(out[f][h] - gold[f][h]) <= 1e-14), f
    name = "li_scene04_i1.bin"
    info = G.MANIFEST["files"][name]
    grec = G.records(name, A.LI_DTYPE)
    p = A.make_params(info["info"]["width"], info["info"]["height"], info["spp"], integrator=1, seed=info["seed"])
    o = ctx.test_records("li", grec, params=p)
    same = o["rng_exit"] == grec["rng_exit"]
    G.residue("image_texture.scene04.i1.paths_differ", int((~same).sum()), int(0.005 * same.size))
    # an ulp of difference in (u,v) can select a neighbouring texel: the measured number of samples that differ
    ok = np.all(_close(o["L"][same], grec["L"][same], 1e-9), axis=1)
    G.residue("image_texture.scene04.i1.samples_outside_1e-9", int((~ok).sum()), int(0.01 * ok.size))
    img, iinfo = G.image("img_scene04_i1_64_spp16.f64")
    for pipe in (A.PIPELINE_MEGAKERNEL, A.PIPELINE_WAVEFRONT):
        q = A.make_params(iinfo["width"], iinfo["height"], iinfo["spp"], integrator=1, seed=iinfo["seed"],
                          pipeline=pipe)
        assert G.rel_l2(ctx.render(q), img) <= REL_L2_BAR


@pytest.mark.parametrize("pipeline", [A.PIPELINE_MEGAKERNEL, A.PIPELINE_WAVEFRONT])
def test_cancel_from_another_thread(ctx, rtr, pipeline):
    """Renderer::cancel (renderer.h:113-115): thread-safe, the running render returns
    RTR_ERR_CANCELLED, the context stays usable."""
    import threading
    import time
    _upload(ctx, 21)
    W = H = 2048
    p = A.make_params(W, H, 256, seed=1, pipeline=pipeline, spp_chunks=1)  # ~0.4 s+ of work
    res = {}
    buf = np.full((H, W, 3), -7.0)  # sentinel: what the caller's buffer held before the call

    def run():
        try:
            ctx.render(p, out=buf)
            res["rc"] = 0
        except rtr.RtrError as e:
            res["rc"] = e.code

    th = threading.Thread(target=run)
    th.start()
    time.sleep(0.1)
    ctx.cancel()
    th.join(60)
    assert not th.is_alive() and res["rc"] == A.RTR_ERR_CANCELLED
    assert ctx.stats()["cancelled"]
    # like the reference's workers (renderer.h:52-59): a tile is either finished or untouched
    full = ctx.render(p)
    assert not ctx.stats()["cancelled"]
    tiles_b = buf.reshape(H // 16, 16, W // 16, 16, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 16 * 16 * 3)
    tiles_f = full.reshape(H // 16, 16, W // 16, 16, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 16 * 16 * 3)
    untouched = np.all(tiles_b == -7.0, axis=1)
    finished = np.all(tiles_b == tiles_f, axis=1)
    assert np.all(untouched | finished), "%d tiles hold partial sums" % (~(untouched | finished)).sum()
    assert untouched.any(), "the cancel came too late to test anything"
    if pipeline == A.PIPELINE_WAVEFRONT:
        assert untouched.all()
    again = ctx.render(A.make_params(64, 64, 2, seed=1, pipeline=pipeline))
    assert np.isfinite(again).all() and again.mean() > 0


def test_cancel_covers_queued_renders_only(ctx, rtr):
    """A cancel covers every render issued so far -- also one still waiting in the stream behind another --
    and none issued afterwards."""
    import torch
    _upload(ctx, 21)
    W = H = 1024
    p = A.make_params(W, H, 256, seed=1, spp_chunks=0)
    fb = torch.full((H, W, 3), -7.0, dtype=torch.float64, device="cuda")
    ctx.render_into(p, fb.data_ptr(), W, blocking=False)
    ctx.render_into(p, fb.data_ptr(), W, blocking=False)  # queued behind the first
    ctx.cancel()
    st = ctx.stats()  # of the second render: it must not have run to the end
    assert st["cancelled"] and st["samples"] < W * H * 256
    ctx.render_into(p, fb.data_ptr(), W, blocking=True)  # issued after the cancel: unaffected
    st = ctx.stats()
    assert not st["cancelled"] and st["samples"] == W * H * 256
    assert float(fb.min()) >= 0.0
    ctx.cancel()  # nothing running: must not leak into the next render
    ctx.render_into(p, fb.data_ptr(), W, blocking=True)
    assert not ctx.stats()["cancelled"]


@pytest.mark.parametrize("workload", ["cornell_mis", "cornell_literal", "final_rr", "final_mis", "mis_spheres", "c5_shard"])
def test_full_spp_crop_of_every_baseline_config(ctx, rtr, workload):
    """What is timed and shipped (spp_chunks = 0: the library's own partial sums) against the oracle at the
    configuration's REAL spp.  (1) The megakernel renders the configuration at FULL size exactly as bench.py times it
    (c5_shard: the tiles one rank of eight owns) and the 64x64 centre crop of THAT framebuffer -- the summation that is
    timed -- is compared with the oracle; (2) the crop rendered on its own (another tile count, so other partial sums),
    both pipelines.  Scenes 07 / 21 (no libm on the path): bit-exact with one running sum, <= 1e-13 with partial sums;
    the others within the 1e-3 tolerance of BASELINE.json (measured: <= 1e-12).  Same check bench.py prints as `parity`."""
    import bench
    import torch
    wl = dict(bench.WORKLOADS[workload])
    sc = bench.load_scene(rtr, wl["scene"])
    ctx.upload(sc)
    W, H, stride = wl["W"], wl["H"], wl.get("stride", 1)
    full = A.make_params(W, H, wl["spp"], integrator=wl["integ"], seed=1, spp_chunks=0, tile_first=0, tile_stride=stride)
    fb = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    ctx.render_into(full, fb.data_ptr(), W, blocking=True)
    timed_chunks = ctx.stats()["spp_chunks"]
    image = fb.cpu().numpy()
    del fb
    mask = G.rtr.renderer.ownership_mask(W, H, 0, stride)
    for pipe in (A.PIPELINE_MEGAKERNEL, A.PIPELINE_WAVEFRONT):
        timed = pipe == A.PIPELINE_MEGAKERNEL
        res = bench.crop_parity(rtr, ctx, sc, wl, timed_chunks, pipe, image if timed else None, mask if timed else None)
        print(workload, pipe, res)
        assert res["ok"], res
        assert res["rel_l2_chunked_vs_chunks1"] <= 1e-13
        G.residue("crop.%s.pipe%d.rel_l2_vs_oracle" % (workload, pipe), res["rel_l2_vs_oracle"], 1e-3)
        if timed:
            assert res["timed_framebuffer_pixels_compared"] > 0
            G.residue("crop.%s.timed_framebuffer.rel_l2_vs_oracle" % workload, res["rel_l2_timed_framebuffer_vs_oracle"], 1e-3)


def test_sharded_render_equals_unsharded(ctx, rtr):
    """Tile sharding on the device: two contexts on GPU 0 render the tiles index % 2 == 0 / 1 of scene 21; their union
    against the unsharded image.  With one running sum per pixel (spp_chunks = 1) every pixel is the same bits for any
    sharding; with the library's own partial sums (spp_chunks = 0) the number of sums depends on how many tiles a call
    owns, so the union agrees within 1e-13 -- rounding of another summation order, documented in include/rtr_hip.h."""
    sc = _upload(ctx, 21)
    W, H, spp = 160, 128, 64
    with rtr.Context(0) as other:
        other.upload(sc)
        for chunks in (1, 0):
            whole = ctx.render(A.make_params(W, H, spp, integrator=4, seed=9, spp_chunks=chunks))
            union = np.zeros_like(whole)
            for rank, c in enumerate((ctx, other)):
                c.render(A.make_params(W, H, spp, integrator=4, seed=9, spp_chunks=chunks, tile_first=rank, tile_stride=2), out=union)
            if chunks == 1:
                assert np.array_equal(_bits(union), _bits(whole))
            else:
                assert G.rel_l2(union, whole) <= 1e-13


@pytest.mark.parametrize("sid,integ", [(21, 4), (9, 1), (22, 4), (22, 3), (1, 1), (23, 4), (8, 1)])
def test_wavefront_persistent_threads_equal_lockstep(ctx, sid, integ):
    """The traversal machine (rt_machine.h: per-lane program position, lane refill from the wave's blocks) is a
    re-scheduling of run_program() / trace_fast(): same image and same cast counts as the lockstep stages and
    as the megakernel, on flat, box-tree and step-program scenes."""
    _upload(ctx, sid)
    ref = ctx.render(A.make_params(96, 64, 6, integrator=integ, seed=21, pipeline=A.PIPELINE_MEGAKERNEL))
    sr = ctx.stats()
    for flags in (0, A.FLAG_WF_PERSISTENT):
        out = ctx.render(A.make_params(96, 64, 6, integrator=integ, seed=21, pipeline=A.PIPELINE_WAVEFRONT, flags=flags))
        st = ctx.stats()
        assert np.array_equal(_bits(out), _bits(ref)), flags
        assert (st["closest_segments"], st["shadow_segments"]) == (sr["closest_segments"], sr["shadow_segments"])


@pytest.mark.parametrize("sid,expect", [(23, True), (7, False), (21, False), (9, False)])
def test_sorted_shading_equals_unsorted(ctx, sid, expect):
    """RTR_FLAG_SORTED_SHADING (rt_kernels.h: the lanes of a workgroup exchange their hits through LDS once per bounce
    so that a wave shades one material type -- the reference's virtual scatter(), material.h:31-60) is a re-scheduling:
    same image bit for bit and same cast counts as the default kernel, with one running sum and with the library's
    partial sums, whole image and tile-sharded.  The variant exists for (MIS, flat scene, quad lights only) = scene 23;
    elsewhere the flag is ignored and stats()["flags_in_effect"] says so."""
    _upload(ctx, sid)
    for chunks, stride in ((1, 1), (0, 1), (3, 2)):
        kw = dict(integrator=4, seed=5, spp_chunks=chunks, tile_first=stride - 1, tile_stride=stride,
                  pipeline=A.PIPELINE_MEGAKERNEL)
        ref = ctx.render(A.make_params(112, 80, 24, **kw))
        sr = ctx.stats()
        assert sr["flags_in_effect"] == 0
        out = ctx.render(A.make_params(112, 80, 24, flags=A.FLAG_SORTED_SHADING, **kw))
        st = ctx.stats()
        assert (st["flags_in_effect"] == A.FLAG_SORTED_SHADING) == expect
        assert np.array_equal(_bits(out), _bits(ref)), (chunks, stride)
        assert (st["samples"], st["closest_segments"], st["shadow_segments"]) == \
            (sr["samples"], sr["closest_segments"], sr["shadow_segments"])


def test_headline_image_at_full_size_and_spp_is_bit_exact(ctx, rtr):
    """BASELINE C2 itself, not a crop: all 640 000 pixels of the Cornell box 800x800 at spp 400 (MIS) from the
    megakernel with one running sum per pixel == the CPU oracle, bit for bit (the oracle is bit-exact against
    the reference's own renders of this scene); the shipped summation (spp_chunks = 0) within 1e-13.
    About 25 s of oracle time on the GPU box's host cores."""
    import os
    import bench
    wl = bench.WORKLOADS["cornell_mis"]
    sc = bench.load_scene(rtr, wl["scene"])
    ctx.upload(sc)
    p1 = A.make_params(wl["W"], wl["H"], wl["spp"], integrator=wl["integ"], seed=1, spp_chunks=1)
    ref, st = G.oracle_render(sc, p1, threads=os.cpu_count() or 1)
    out = ctx.render(p1)
    gs = ctx.stats()
    assert (gs["samples"], gs["closest_segments"], gs["shadow_segments"]) == \
        (st["samples"], st["closest_segments"], st["shadow_segments"])
    assert np.array_equal(_bits(out), _bits(ref))
    p0 = A.make_params(wl["W"], wl["H"], wl["spp"], integrator=wl["integ"], seed=1, spp_chunks=0)
    G.residue("c2_full_image.chunked.rel_l2", G.rel_l2(ctx.render(p0), ref), 1e-13)


def test_large_image_shape_of_config_c5(ctx):
    """BASELINE C5 is 4096x4096 (65 536 tiles); at reduced spp the whole image path still holds:
    every pixel written, statistics complete, both pipelines agree bit for bit."""
    _upload(ctx, 21)
    W = H = 4096
    a = ctx.render(A.make_params(W, H, 2, seed=4, pipeline=A.PIPELINE_MEGAKERNEL, spp_chunks=1))
    assert ctx.stats()["samples"] == W * H * 2
    b = ctx.render(A.make_params(W, H, 2, seed=4, pipeline=A.PIPELINE_WAVEFRONT, spp_chunks=1))
    assert ctx.stats()["samples"] == W * H * 2
    assert np.array_equal(a, b) and np.isfinite(a).all()
    assert (a.sum(axis=2) > 0).mean() > 0.5


def test_cpp_host_renderer_cli(ctx, rtr, tmp_path):
    """host/rtr_cli.cpp: the C++ mirror of main.cpp + Renderer::render driving the same C ABI."""
    import os
    import subprocess
    cli = os.path.join(G.ROOT, "ray_tracing-rendering_amd", "rtr_cli")
    if not os.path.exists(cli):
        pytest.skip("rtr_cli not built")
    out = str(tmp_path / "cli.ppm")
    r = subprocess.run([cli, "21", "4", "--width", "64", "--spp", "4", "--seed", "3", "--out", out],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    raw = open(out, "rb").read()
    head = b"P6\n64 64\n255\n"
    assert raw.startswith(head)
    got = np.frombuffer(raw[len(head):], dtype=np.uint8).reshape(64, 64, 3)
    _upload(ctx, 21)
    rb = rtr.RenderBuffer(64, 64)
    rb.store_linear(ctx.render(A.make_params(64, 64, 4, seed=3, spp_chunks=0)))
    assert np.array_equal(got, rb.to_rgb8())
    # progressive bands (what keeps the reference's polling UI fed): same pixels for any band count
    for bands in (1, 3):
        out_b = str(tmp_path / ("cli_b%d.ppm" % bands))
        r = subprocess.run([cli, "21", "4", "--width", "64", "--spp", "4", "--seed", "3", "--bands", str(bands), "--out", out_b],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr
        assert open(out_b, "rb").read() == raw
    # N contexts, one host thread each (here: two contexts on GPU 0), tiles dealt round-robin: same bytes; the
    # flattened scene is uploaded once however often render() is called
    out_n = str(tmp_path / "cli_two.ppm")
    r = subprocess.run([cli, "21", "4", "--width", "64", "--spp", "4", "--seed", "3", "--devices", "0,0", "--repeat", "2",
                        "--out", out_n], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert open(out_n, "rb").read() == raw
    assert b"contexts: 2  scene uploads: 1" in r.stdout, r.stdout
    # end to end against the reference's own writer: the CLI's scene21 image at the golden configuration
    # (bit-exact on this scene) must carry the pixels of the PNG the reference wrote from its own render
    out_g = str(tmp_path / "cli_golden.ppm")
    r = subprocess.run([cli, "21", "4", "--width", "64", "--spp", "16", "--seed", "1", "--out", out_g],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    want = np.fromfile(os.path.join(G.GOLD, "png_scene21_i4_64_spp16.rgb8"), dtype=np.uint8)
    assert np.array_equal(np.frombuffer(open(out_g, "rb").read()[len(head):], dtype=np.uint8), want)
    # ... and as the PNG main.cpp:138-151 saves (RenderBuffer::save_to_png of the mirror), from two contexts
    from test_output_stage import _decode_png
    out_p = str(tmp_path / "cli_golden.png")
    r = subprocess.run([cli, "21", "4", "--width", "64", "--spp", "16", "--seed", "1", "--devices", "0,0", "--out", out_p],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(_decode_png(out_p).reshape(-1), want)


def test_large_flat_list(ctx, rtr):
    """50 000 spheres as direct children of one hittable_list: the compiled traversal renders it through a box
    tree; the reference-order walk steps through the list with a two-word continuation (not one LDS stack word
    per child) and gives the same bits by brute force; rays are checked against the oracle as well."""
    base = G.scene(23)
    n = 50_000
    rng = np.random.default_rng(3)
    nodes = np.zeros(n + 1, dtype=A.NODE_DTYPE)
    nodes["type"][0], nodes["a"][0], nodes["b"][0] = A.NODE_LIST, 0, n
    nodes["type"][1:] = A.NODE_SPHERE
    nodes["a"][1:] = rng.integers(0, len(base.materials), n)
    nodes["f"][1:, 0:3] = rng.uniform(-4.0, 4.0, (n, 3))
    nodes["f"][1:, 3] = 0.05
    sc = rtr.Scene(0, nodes, np.arange(1, n + 1, dtype=np.int32), base.materials, base.textures, base.perlin,
                   base.images, base.image_bytes, base.lights, base.camera, base.background)
    ctx.upload(sc)
    out = ctx.render(A.make_params(96, 54, 4, integrator=4, seed=2))
    assert np.isfinite(out).all() and out.mean() > 0
    small = ctx.render(A.make_params(32, 18, 2, integrator=4, seed=2))
    brute = ctx.render(A.make_params(32, 18, 2, integrator=4, seed=2, flags=A.FLAG_REFERENCE_ORDER))
    assert np.array_equal(_bits(small), _bits(brute))
    rays = np.zeros(256, dtype=A.HIT_DTYPE)
    rays["o"] = rng.uniform(-6.0, 6.0, (256, 3))
    rays["d"] = rng.normal(0.0, 1.0, (256, 3))
    rays["t_min"], rays["t_max"], rays["rng_in"] = 0.001, np.inf, 9
    ora = G.oracle_records(sc, "rto_hits", rays)
    h = ora["hit"] == 1
    for ref_order in (False, True):
        ctx.reference_order(ref_order)
        dev = ctx.test_records("hits", rays)
        ctx.reference_order(False)
        assert dev["hit"].sum() > 20 and np.array_equal(dev["hit"], ora["hit"]), ref_order
        assert np.array_equal(dev["material"][h], ora["material"][h]), ref_order
        assert np.array_equal(_bits(dev["t"][h]), _bits(ora["t"][h])), ref_order


def test_one_sincos_equals_sin_and_cos_on_every_sampler_angle(ctx):
    """random_cosine_direction (vec3.h:261-269) and PBRMaterial::sample (material.h:268-275) call cos(phi) and
    sin(phi); the device takes both from one sincos().  Exhaustive over the 2^32 states of the generator."""
    assert ctx.sincos_mismatches() == 0


def test_error_behaviour(ctx, rtr):
    sc = _upload(ctx, 21)
    with pytest.raises(rtr.RtrError) as e:
        ctx.render(A.make_params(64, 64, 1, integrator=5))
    assert e.value.code == A.RTR_ERR_UNSUPPORTED
    with pytest.raises(rtr.RtrError) as e:
        ctx.render(A.make_params(64, 64, 1, region=(0, 0, 65, 64)))
    assert e.value.code == A.RTR_ERR_INVALID
    for w, h in [(2 ** 31 - 1, 2 ** 31 - 1), (2 ** 20, 2 ** 20), (1, 64), (64, 0)]:  # tile count would overflow / empty image
        with pytest.raises(rtr.RtrError) as e:
            ctx.plan_chunks(A.make_params(w, h, 1))
        assert e.value.code == A.RTR_ERR_INVALID, (w, h)
    for kw in [dict(spp_chunks=2), dict(tile_first=3, tile_stride=2), dict(flags=64), dict(pipeline=7), dict(max_depth=0)]:
        with pytest.raises(rtr.RtrError) as e:
            ctx.plan_chunks(A.make_params(64, 64, 1, **kw))
        assert e.value.code == A.RTR_ERR_INVALID, kw
    bad = rtr.Scene.from_bytes(sc.to_bytes())
    bad.nodes["a"][bad.root] = 10 ** 6
    with pytest.raises(rtr.RtrError) as e:
        ctx.upload(bad)
    assert e.value.code == A.RTR_ERR_INVALID
    fresh = rtr.Context(0)
    with pytest.raises(rtr.RtrError) as e:
        fresh.render(A.make_params(64, 64, 1))
    assert e.value.code == A.RTR_ERR_NO_SCENE
    fresh.close()


def test_shared_reciprocal_division_equals_plain_division(ctx):
    """rt_device.h: div_shared -- the primitive tests divide by a ray-direction component (aarect.h:80,99,118) or by
    |d|^2 (sphere.h:43-47) through ONE refined reciprocal per ray and frame; the last three instructions of the
    compiler's own division sequence then give the quotient.  2^32 operand pairs of the range it is used in: 0 differ."""
    assert ctx.shared_division_mismatches() == 0
