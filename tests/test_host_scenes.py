"""The host C++ layer (host/rtr_scene_api.h): scenes described through the mirrored API flatten
to the same bytes as the reference's own object graph (tests/golden/scene*.rtrs were walked
from it), and the reference's UNMODIFIED scene builders compile and run against it."""
import os
import subprocess
import tempfile

import pytest

import _golden as G

rtr = G.rtr
DROPIN = os.path.join(G.ROOT, "oracle", "_ref", "dropin_scenes")


@pytest.mark.parametrize("sid", [7, 21, 23, 9, 22])
def test_host_builder_matches_reference_graph(sid):
    sc, d = rtr.hostscene.build_scene(sid, with_defaults=True)
    gold = G.scene(sid)
    assert sc.to_bytes() == gold.to_bytes()
    info = G.MANIFEST["files"]["scene%02d.rtrs%s" % (sid, ".gz" if sid in (9, 22) else "")]
    assert sc.sha256() == info["raw_sha256"]
    assert (d["width"], d["height"], d["spp"]) == (info["info"]["default_width"], info["info"]["default_height"],
                                                   info["info"]["default_spp"])


def test_scene_seed_changes_random_scenes_only():
    a = rtr.hostscene.build_scene(9, scene_seed=1)
    b = rtr.hostscene.build_scene(9, scene_seed=2)
    assert a.to_bytes() != b.to_bytes() and len(a.nodes) == len(b.nodes)
    c1 = rtr.hostscene.build_scene(23, scene_seed=1)
    c2 = rtr.hostscene.build_scene(23, scene_seed=1)
    assert c1.to_bytes() == c2.to_bytes()


def test_unknown_scene_id_fails_loudly():
    with pytest.raises(rtr.native.RtrError):
        rtr.hostscene.build_scene(3)


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/dropin_scenes is built only where the reference is")
@pytest.mark.parametrize("sid", [7, 21, 23, 9, 22])
def test_reference_scene_builders_drop_in(sid):
    """scene/scenes.cpp of the reference, compiled against host/compat (oracle/Makefile)."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "s.rtrs")
        r = subprocess.run([DROPIN, str(sid), "12345", out], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        assert r.returncode == 0, r.stdout
        assert open(out, "rb").read() == G.scene(sid).to_bytes()


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/dropin_scenes is built only where the reference is")
@pytest.mark.parametrize("sid", [15, 17, 18, 19])
def test_delta_light_scenes_drop_in(sid):
    """Point / directional / spot light scenes of the reference, built against host/compat."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "s.rtrs")
        r = subprocess.run([DROPIN, str(sid), "12345", out], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        assert r.returncode == 0, r.stdout
        assert open(out, "rb").read() == G.scene(sid).to_bytes()


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/dropin_scenes is built only where the reference is")
@pytest.mark.parametrize("sid", [24, 26])
def test_environment_map_scenes_drop_in(sid):
    """EnvironmentLight of the host layer (own RGBE reader, Distribution2D tables) on the synthetic
    maps the goldens were made with: the flattened scene equals the one walked out of the
    reference's objects (stb_image texels, its Distribution2D), byte for byte."""
    gen = _gen_golden_module()
    fname, w, h, sun = gen.HDR_ASSETS[sid]
    with tempfile.TemporaryDirectory() as td:
        gen.write_hdr(os.path.join(td, fname), w, h, sun)
        out = os.path.join(td, "s.rtrs")
        r = subprocess.run([DROPIN, str(sid), "12345", out], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=td)
        assert r.returncode == 0, r.stdout
        assert open(out, "rb").read() == G.scene(sid).to_bytes()


def _gen_golden_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(G.ROOT, "oracle", "gen_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/dropin_scenes is built only where the reference is")
def test_png_textured_pbr_scene_drops_in():
    """Scene 35 (PBRMaterial with image albedo / roughness / metallic / normal maps): the host layer's
    own PNG decoder (zlib inflate + the five row filters; grey, grey+alpha, RGB, RGBA and palette
    images of 4, 8 and 16 bits) yields the texels stb_image gave the reference -- the flattened
    scene is byte-identical to the golden walked out of the reference's objects."""
    gen = _gen_golden_module()
    with tempfile.TemporaryDirectory() as td:
        gen.write_pbr_textures(td)
        out = os.path.join(td, "s.rtrs")
        r = subprocess.run([DROPIN, "35", "12345", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=td)
        assert r.returncode == 0, r.stdout
        assert b"Could not load" not in r.stderr
        sc = rtr.Scene.load(out)
        assert len(sc.images) == 10 and sc.to_bytes() == G.scene(35).to_bytes()


def _rle_scanline(row):
    """Radiance adaptive run-length scanline: 2 2 hi lo, then each of the 4 components coded as runs
    (count > 128: repeat) and literals (count <= 128)."""
    w = len(row) // 4
    out = bytearray((2, 2, w >> 8, w & 255))
    for comp in range(4):
        vals = row[comp::4]
        i = 0
        while i < w:
            run = 1
            while i + run < w and run < 127 and vals[i + run] == vals[i]:
                run += 1
            if run >= 3:
                out += bytes((128 + run, vals[i]))
                i += run
            else:
                lit = bytearray()
                while i < w and len(lit) < 128:
                    if i + 2 < w and vals[i] == vals[i + 1] == vals[i + 2]:
                        break
                    lit.append(vals[i])
                    i += 1
                out += bytes((len(lit),)) + bytes(lit)
    return bytes(out)


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/dropin_scenes is built only where the reference is")
def test_environment_map_run_length_scanlines():
    """The same picture stored with run-length scanlines decodes to the same scene as the flat file."""
    w, h = 32, 16
    px = bytearray()
    for j in range(h):
        for i in range(w):
            px += bytes((40 + (i // 5) * 9, 30 + (j * 11) % 200, 20 + (i * 3 + j * 13) % 220, 128 + (j % 3)))
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w)
    outs = []
    for body in (bytes(px), b"".join(_rle_scanline(px[j * w * 4:(j + 1) * w * 4]) for j in range(h))):
        with tempfile.TemporaryDirectory() as td:
            with open(os.path.join(td, "brown_photostudio_02_4k.hdr"), "wb") as f:
                f.write(head + body)
            out = os.path.join(td, "s.rtrs")
            r = subprocess.run([DROPIN, "24", "12345", out], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=td)
            assert r.returncode == 0, r.stdout
            outs.append(open(out, "rb").read())
    sc = rtr.Scene.from_bytes(outs[0])
    assert sc.lights["type"][0] == G.A.LIGHT_ENV_MAP and (sc.lights["f"][0][0], sc.lights["f"][0][1]) == (w, h)
    assert outs[0] == outs[1]


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/dropin_scenes is built only where the reference is")
def test_every_reference_scene_flattens():
    """All scene ids of the reference's select_scene (scenes.cpp:1523-2096) build against the host
    layer and flatten to something the device accepts (environment maps are absent -> uniform sky)."""
    ids = [1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28,
           30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42]
    with tempfile.TemporaryDirectory() as td:
        for sid in ids:
            out = os.path.join(td, "s%d.rtrs" % sid)
            r = subprocess.run([DROPIN, str(sid), "12345", out], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            assert r.returncode == 0, (sid, r.stdout)
            info = rtr.native.validate_scene(rtr.Scene.load(out))
            assert info["stack_words"] >= 1
