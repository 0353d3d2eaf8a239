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
