#!/usr/bin/env python3
"""Vectors of the UNMODIFIED reference on seeded random scenes (tests/_randscene.py).

The reference has no scene-file loader; oracle/ref_harness.cpp builds its own classes (sphere, xy_rect, box sides,
translate, rotate_y, flip_face, constant_medium, bvh_node, the materials, textures and lights) from a flattened
scene file (commands `hits-rtrs`, `render-rtrs`) and lets them answer.  Written under tests/golden/:
  random_<seed>.rtrs.gz        the scene as generated (the fixture holds the scene itself, not the generator)
  random_<seed>_hits.bin       512 rays: hit flag, t, p, n, u, v, front_face, material, RNG state after the cast
  random_<seed>_i1.f64 / _i4.f64   48x32 spp 4 linear images of RRPathInterator / MISPathIntegrator
and their entries in manifest.json (argv, sha256).  Run from the repository root:  python oracle/gen_random_golden.py
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import _randscene as R  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
HARNESS = os.path.join(HERE, "_ref", "ref_harness")
CASES = [(11, {}), (13, dict(n_objects=90)), (14, dict(media=True)), (15, dict(media=True, n_objects=60)),
         (16, dict(hollow=True)), (17, dict(n_objects=8, ties=True)), (18, dict(media=True, hollow=True)),
         (19, dict(n_objects=200)), (27, dict(delta_lights=True)), (28, dict(delta_lights=True, media=True)),
         (32, dict(moved_media=True)), (33, dict(moved_media=True, media=True, n_objects=50))]
W, H, SPP, N_RAYS = 48, 32, 4, 512
# the same generator output under the reference's own bvh_node (harness command wrap-bvh: its constructor draws the
# split axes): fixtures random_<seed>b.*
BVH_CASES = [(13, dict(n_objects=90)), (15, dict(media=True, n_objects=60)), (16, dict(hollow=True)),
             (18, dict(media=True, hollow=True)), (19, dict(n_objects=200)), (23, dict(media=True, n_objects=100)),
             (26, dict(n_objects=30, hollow=True)), (28, dict(delta_lights=True, media=True)),
             (32, dict(moved_media=True)), (33, dict(moved_media=True, media=True, n_objects=50))]


def sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def main():
    subprocess.run(["make", "-C", HERE, "_ref/ref_harness"], check=True, stdout=subprocess.DEVNULL)
    with open(os.path.join(GOLD, "manifest.json")) as f:
        manifest = json.load(f)

    def note(name, argv, info, **extra):
        p = os.path.join(GOLD, name)
        manifest["files"][name] = {"argv": argv, "info": info, "sha256": sha(p), "bytes": os.path.getsize(p), **extra}

    with tempfile.TemporaryDirectory() as td:
        for seed, kw, wrap in [(s_, k_, False) for s_, k_ in CASES] + [(s_, k_, True) for s_, k_ in BVH_CASES]:
            sc = R.random_scene(seed, **kw)
            raw = os.path.join(td, "scene.rtrs")
            sc.save(raw)
            tag = "%02d" % seed
            if wrap:
                flat, raw, tag = raw, os.path.join(td, "scene_bvh.rtrs"), "%02db" % seed
                subprocess.run([HARNESS, "wrap-bvh", flat, str(777 + seed), raw], check=True, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL)
            name = "random_%s.rtrs.gz" % tag
            with open(raw, "rb") as f, open(os.path.join(GOLD, name), "wb") as fo, \
                    gzip.GzipFile(filename="", fileobj=fo, mode="wb", mtime=0) as g:
                g.write(f.read())
            note(name, ["tests/_randscene.py: random_scene(%d, **%r)" % (seed, kw)] +
                 (["ref_harness", "wrap-bvh", "<that scene>", str(777 + seed), "random_%s.rtrs" % tag] if wrap else []), {},
                 raw_sha256=sha(raw))
            rays = os.path.join(td, "rays.bin")
            R.random_rays(seed, N_RAYS).tofile(rays)
            name = "random_%s_hits.bin" % tag
            out = subprocess.run([HARNESS, "hits-rtrs", raw, rays, os.path.join(GOLD, name)], check=True,
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
            note(name, ["ref_harness", "hits-rtrs", "random_%s.rtrs" % tag, "random_rays(%d, %d)" % (seed, N_RAYS), name],
                 json.loads(out.strip().splitlines()[-1]), scene="random_%s" % tag)
            for integ in (1, 4):
                name = "random_%s_i%d.f64" % (tag, integ)
                out = subprocess.run([HARNESS, "render-rtrs", raw, str(integ), str(W), str(H), str(SPP), str(100 + seed),
                                      os.path.join(GOLD, name)], check=True, stdout=subprocess.PIPE,
                                     stderr=subprocess.DEVNULL).stdout.decode()
                note(name, ["ref_harness", "render-rtrs", "random_%s.rtrs" % tag, str(integ), str(W), str(H), str(SPP),
                            str(100 + seed), name], json.loads(out.strip().splitlines()[-1]), scene="random_%s" % tag,
                     integrator=integ, width=W, height=H, spp=SPP, seed=100 + seed)
            print("random_%s done" % tag, flush=True)
        # exact ties in t ACROSS transform chains (box faces in the planes of rects visited before and after them)
        sc = R.cross_instance_tie_scene()
        raw = os.path.join(td, "xties.rtrs")
        sc.save(raw)
        with open(raw, "rb") as f, open(os.path.join(GOLD, "xties.rtrs.gz"), "wb") as fo, \
                gzip.GzipFile(filename="", fileobj=fo, mode="wb", mtime=0) as g:
            g.write(f.read())
        note("xties.rtrs.gz", ["tests/_randscene.py: cross_instance_tie_scene()"], {}, raw_sha256=sha(raw))
        rays = os.path.join(td, "xrays.bin")
        R.cross_instance_tie_rays().tofile(rays)
        out = subprocess.run([HARNESS, "hits-rtrs", raw, rays, os.path.join(GOLD, "xties_hits.bin")], check=True,
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        note("xties_hits.bin", ["ref_harness", "hits-rtrs", "xties.rtrs", "cross_instance_tie_rays()", "xties_hits.bin"],
             json.loads(out.strip().splitlines()[-1]), scene="xties")
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
