#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: regenerate tests/golden/ from the UNMODIFIED reference.

Runs oracle/_ref/ref_harness (built by oracle/Makefile from /root/reference, which exists
only in the build container) and writes small data fixtures: flattened scenes, RNG
known-answer vectors, per-primitive/material/light vectors, per-sample Li records and
small linear-radiance images.  Every fixture is data (inputs + the reference's outputs);
no reference source text is stored.  The manifest records the exact command of each file.

    python oracle/gen_golden.py            # needs /root/reference
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
HARNESS = os.path.join(HERE, "_ref", "ref_harness")
SCENE_SEED = 12345  # xorshift state before select_scene(): BVH axes, perlin tables, random geometry


def write_hdr(path, w, h, sun):
    px = bytearray()
    for j in range(h):
        for i in range(w):
            e = 128 + (i + 2 * j) % 3
            if (i, j) in sun:
                e = 135
            px += bytes((40 + (i * 7 + j * 3) % 200, 30 + (i * 5 + j * 11) % 200, 20 + (i * 3 + j * 13) % 220, e))
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w) + bytes(px))


def write_ppm(path, w, h, k):
    """synthetic picture number k as binary PPM (stb_image sniffs the format from the content, so the
    file may carry the .png / .jpg name a scene asks for)"""
    px = bytearray()
    for j in range(h):
        for i in range(w):
            px += bytes(((i * (5 + k) + j * 3 + 17 * k) % 256, (i * i + (7 + k) * j + 40) % 256,
                         (128 + ((i ^ (3 * j)) + 29 * k) % 128) % 256))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h) + bytes(px))


def write_png(path, w, h, k, ctype, depth):
    """synthetic picture number k as a PNG of the given colour type / bit depth, every row with another
    filter type (0..4), so that a decoder has to implement all of them"""
    import struct
    import zlib
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    maxv = (1 << depth) - 1

    def sample(i, j, ch):
        v = (i * (5 + k + 2 * ch) + j * (3 + ch) + 17 * k + 40 * ch + ((i ^ (3 * j)) & 15)) % 256
        return (v * maxv) // 255 if depth < 8 else (v * 257 if depth == 16 else v)

    rows = []
    for j in range(h):
        if depth >= 8:
            row = bytearray()
            for i in range(w):
                for ch in range(channels):
                    v = sample(i, j, ch)
                    row += struct.pack(">H", v) if depth == 16 else bytes((v,))
        else:  # packed samples, most significant bits first
            bits = 0
            nb = 0
            row = bytearray()
            for i in range(w):
                bits = (bits << depth) | sample(i, j, 0)
                nb += depth
                if nb == 8:
                    row.append(bits)
                    bits = nb = 0
            if nb:
                row.append(bits << (8 - nb))
        rows.append(bytes(row))
    bpp = max(1, channels * depth // 8)
    raw = bytearray()
    prev = bytes(len(rows[0]))
    for j, row in enumerate(rows):
        f = (j + k) % 5
        out = bytearray()
        for i, x in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 0:
                pred = 0
            elif f == 1:
                pred = a
            elif f == 2:
                pred = b
            elif f == 3:
                pred = (a + b) >> 1
            else:
                p_ = a + b - c
                pa, pb, pc = abs(p_ - a), abs(p_ - b), abs(p_ - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out.append((x - pred) & 255)
        raw += bytes((f,)) + bytes(out)
        prev = row

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if ctype == 3:
        png += chunk(b"PLTE", bytes((((e * 7 + 3 * k) % 256) if c == 0 else ((e * 13 + 50) % 256) if c == 1 else
                                     (255 - e) % 256) for e in range(1 << depth) for c in range(3)))
    comp = zlib.compress(bytes(raw), 6)
    half = len(comp) // 2
    png += chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")  # split stream
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(png)


# scene 35 (pbr_texture_demo, scenes.cpp:1244-1300): albedo / roughness / metallic / normal maps as PNG files of
# every colour type the decoders must handle: (path, colour type, bit depth)
PBR_TEXTURE_ASSETS = [("tex/oak/oak_veneer_01_diff_1k.png", 2, 8), ("tex/oak/oak_veneer_01_rough_1k.png", 0, 8),
                      ("tex/oak/oak_veneer_01_nor_dx_1k.png", 6, 8), ("tex/brick/red_brick_diff_1k.png", 3, 8),
                      ("tex/brick/red_brick_rough_1k.png", 0, 4), ("tex/brick/red_brick_nor_dx_1k.png", 2, 16),
                      ("tex/rust/rusty_metal_04_diff_1k.png", 2, 8), ("tex/rust/rusty_metal_04_rough_1k.png", 4, 8),
                      ("tex/rust/rusty_metal_04_metal_1k.png", 3, 4), ("tex/rust/rusty_metal_04_nor_dx_1k.png", 2, 8)]


def write_pbr_textures(td):
    for k, (rel, ctype, depth) in enumerate(PBR_TEXTURE_ASSETS):
        write_png(os.path.join(td, rel), 24 + 2 * (k % 3), 20 + k, k, ctype, depth)


# scene ids without a dedicated fixture set above
ALL_OTHER_SCENES = [2, 5, 6, 10, 11, 12, 13, 14, 16, 20, 25, 27, 28, 30, 31, 32, 33, 34, 36, 37, 38, 39, 40, 41, 42]

PRIMITIVE_SCENES = list(range(1001, 1011))

HDR_ASSETS = {24: ("brown_photostudio_02_4k.hdr", 32, 16, {(20, 4), (21, 4), (20, 5)}),
              26: ("rnl_probe.hdr", 16, 16, {(11, 5), (4, 9)})}


def run(*args, cwd=None):
    cmd = [HARNESS] + [str(a) for a in args]
    out = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=cwd).stdout.decode()
    return cmd, json.loads(out.strip().splitlines()[-1]) if out.strip().startswith("{") else {}


def sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def png_fixtures(manifest, note):
    """the reference's own output stage (Renderer::write_color_to_buffer + RenderBuffer::save_to_png,
    the PNG read back with its stb_image) on linear images: two golden renders and one synthetic image of edge
    values (0, denormal-small, values either side of every k/255 step after the sqrt, 1, > 1, huge)."""
    import numpy as np
    edge = np.zeros((8, 32, 3), dtype="<f8")
    vals = [0.0, 1e-300, 1e-9, 0.25, 0.999999, 1.0, 1.0000001, 4.0, 1e300]
    for k in range(0, 256, 3):  # g = k / 255 in gamma space -> linear g*g, nudged either side of the step
        g = k / 255.0
        vals += [np.nextafter(g * g, 0.0), g * g, np.nextafter(g * g, 2.0)]
    flat = edge.reshape(-1)
    flat[:len(vals)] = vals[:flat.size]
    edge.tofile(os.path.join(GOLD, "png_edge_in.f64"))
    manifest["files"]["png_edge_in.f64"] = {"argv": ["gen_golden.py: synthetic linear image 32x8"], "info": {},
                                            "sha256": sha(os.path.join(GOLD, "png_edge_in.f64")),
                                            "bytes": edge.nbytes, "width": 32, "height": 8}
    for src, w, h in (("img_scene21_i4_64_spp16.f64", 64, 64), ("img_scene23_i4_64_spp16.f64", 64, 36),
                      ("png_edge_in.f64", 32, 8)):
        if src.startswith("img_"):
            w, h = manifest["files"][src]["width"], manifest["files"][src]["height"]
        name = "png_" + src.replace("img_", "").replace("png_", "").replace(".f64", "") + ".rgb8"
        cmd, info = run("png", os.path.join(GOLD, src), w, h, os.path.join(GOLD, name))
        cmd[2] = src
        note(name, cmd, info, source=src, width=w, height=h)


TIE_SCENES = [1012, 1013]  # exact ties in t (harness scenes; a hittable_list and the same objects under a bvh_node)


def tie_fixtures(note):
    for sid in TIE_SCENES:
        name = "scene%d.rtrs" % sid
        cmd, info = run("dump-scene", sid, SCENE_SEED, os.path.join(GOLD, name))
        note(name, cmd, info, raw_sha256=sha(os.path.join(GOLD, name)))
        name = "hits_scene%d.bin" % sid
        cmd, info = run("hits", sid, SCENE_SEED, 2048, 900 + sid, os.path.join(GOLD, name))
        note(name, cmd, info, scene=sid)


def main():
    if "--add-ties" in sys.argv:  # only the tie fixtures, into the existing manifest
        subprocess.run(["make", "-C", HERE, "_ref/ref_harness"], check=True)
        with open(os.path.join(GOLD, "manifest.json")) as f:
            manifest = json.load(f)

        def note(name, cmd, info, **extra):
            p = os.path.join(GOLD, name)
            manifest["files"][name] = {"argv": [os.path.basename(cmd[0])] + cmd[1:-1] + [name], "info": info,
                                       "sha256": sha(p), "bytes": os.path.getsize(p), **extra}
        tie_fixtures(note)
        with open(os.path.join(GOLD, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return 0
    if "--add-png" in sys.argv:  # only the N3 fixtures, into the existing manifest
        subprocess.run(["make", "-C", HERE, "_ref/ref_harness"], check=True)
        with open(os.path.join(GOLD, "manifest.json")) as f:
            manifest = json.load(f)

        def note(name, cmd, info, **extra):
            p = os.path.join(GOLD, name)
            manifest["files"][name] = {"argv": [os.path.basename(cmd[0])] + cmd[1:-1] + [name], "info": info,
                                       "sha256": sha(p), "bytes": os.path.getsize(p), **extra}
        png_fixtures(manifest, note)
        with open(os.path.join(GOLD, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return 0
    if not os.path.exists(HARNESS):
        subprocess.run(["make", "-C", HERE, "_ref/ref_harness"], check=True)
    os.makedirs(GOLD, exist_ok=True)
    manifest = {"scene_seed": SCENE_SEED, "generator": "oracle/gen_golden.py", "files": {}}

    def note(name, cmd, info, **extra):
        p = os.path.join(GOLD, name)
        manifest["files"][name] = {"argv": [os.path.basename(cmd[0])] + cmd[1:-1] + [name], "info": info,
                                   "sha256": sha(p), "bytes": os.path.getsize(p), **extra}

    # RNG known answers
    cmd, info = run("rng", os.path.join(GOLD, "rng.bin"))
    note("rng.bin", cmd, info)

    # flattened scenes (walked from the reference's own object graph)
    for sid in (7, 21, 23, 9, 22, 15, 17, 18, 19, 1, 8):
        name = "scene%02d.rtrs" % sid
        path = os.path.join(GOLD, name)
        cmd, info = run("dump-scene", sid, SCENE_SEED, path)
        raw_sha = sha(path)
        if os.path.getsize(path) > 60_000:  # scene 9 / 22 / 1: keep the repo small
            with open(path, "rb") as f:
                data = f.read()
            with open(path + ".gz", "wb") as f:
                f.write(gzip.compress(data, 9, mtime=0))
            os.remove(path)
            note(name + ".gz", cmd, info, raw_sha256=raw_sha)
        else:
            note(name, cmd, info, raw_sha256=raw_sha)

    # closest-hit vectors on whole scenes
    for sid, n in ((21, 1024), (23, 1024), (9, 1536), (1, 768), (8, 768)):
        name = "hits_scene%02d.bin" % sid
        cmd, info = run("hits", sid, SCENE_SEED, n, 777 + sid, os.path.join(GOLD, name))
        note(name, cmd, info, scene=sid)

    # material / light vectors
    for sid, n in ((23, 96), (9, 48)):
        name = "materials_scene%02d.bin" % sid
        cmd, info = run("materials", sid, SCENE_SEED, n, 4242 + sid, os.path.join(GOLD, name))
        note(name, cmd, info, scene=sid)
    for sid, n in ((21, 256), (23, 256), (15, 128), (17, 128), (18, 128), (19, 128)):
        name = "lights_scene%02d.bin" % sid
        cmd, info = run("lights", sid, SCENE_SEED, n, 99 + sid, os.path.join(GOLD, name))
        note(name, cmd, info, scene=sid)

    # per-sample Li records and small images: (scene, integrator, W, spp, seed)
    cases = [(7, 1, 64, 16, 1), (7, 4, 64, 16, 1), (21, 4, 64, 16, 1), (23, 4, 64, 16, 1), (9, 1, 64, 16, 1),
             (22, 4, 64, 16, 1),
             # integrators 0 (plain path), 2 (BSDF-only), 3 (NEE without MIS)
             (7, 0, 48, 8, 1), (23, 2, 64, 16, 1), (21, 3, 64, 16, 1), (23, 3, 64, 16, 1),
             # delta lights (point 15, directional 17, spot 18)
             (15, 4, 64, 16, 1), (17, 4, 64, 16, 1), (18, 4, 64, 16, 1), (18, 3, 64, 16, 1),
             # the map-less EnvironmentLight (uniform white sky): scene 19
             (19, 4, 64, 16, 1), (19, 3, 64, 16, 1),
             # scene 1 = RTIOW random spheres (checker, metal, glass, moving spheres, thin lens; 485-leaf BVH)
             # scene 8 = Cornell smoke (media whose boundaries sit under translate/rotate_y)
             (1, 1, 64, 16, 1), (8, 1, 64, 16, 1)]
    for sid, integ, W, spp, seed in cases:
        name = "li_scene%02d_i%d.bin" % (sid, integ)
        n_li = 2048 if (sid, integ) in ((7, 1), (7, 4), (21, 4), (23, 4), (9, 1), (22, 4)) else 768
        cmd, info = run("li", sid, integ, W, spp, seed, SCENE_SEED, n_li, os.path.join(GOLD, name))
        note(name, cmd, info, scene=sid, integrator=integ, width=W, spp=spp, seed=seed)
        name = "img_scene%02d_i%d_%d_spp%d.f64" % (sid, integ, W, spp)
        cmd, info = run("render", sid, integ, W, spp, seed, SCENE_SEED, os.path.join(GOLD, name), 8)
        note(name, cmd, info, scene=sid, integrator=integ, width=W, height=info["height"], spp=spp, seed=seed)
    # a REAL image texture.  None of the reference's assets ship, so a
    # synthetic 96x48 picture is written as binary PPM under the name the scene asks for
    # ("earthmap.jpg"; stb_image sniffs the format from the content) and the harness runs in that
    # directory: scene 4 (earth(): one textured sphere) then loads it.
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        w, h = 96, 48
        px = bytearray()
        for j in range(h):
            for i in range(w):
                px += bytes(((i * 5 + j * 3) % 256, (i * i + 7 * j) % 256, (i ^ (3 * j)) % 256))
        with open(os.path.join(td, "earthmap.jpg"), "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (w, h) + bytes(px))
        name = "scene04.rtrs"
        cmd, info = run("dump-scene", 4, SCENE_SEED, os.path.join(GOLD, name), cwd=td)
        note(name, cmd, info, raw_sha256=sha(os.path.join(GOLD, name)), asset="synthetic 96x48 PPM as earthmap.jpg")
        name = "hits_scene04.bin"
        cmd, info = run("hits", 4, SCENE_SEED, 512, 781, os.path.join(GOLD, name), cwd=td)
        note(name, cmd, info, scene=4)
        name = "li_scene04_i1.bin"
        cmd, info = run("li", 4, 1, 64, 16, 1, SCENE_SEED, 768, os.path.join(GOLD, name), cwd=td)
        note(name, cmd, info, scene=4, integrator=1, width=64, spp=16, seed=1)
        name = "img_scene04_i1_64_spp16.f64"
        cmd, info = run("render", 4, 1, 64, 16, 1, SCENE_SEED, os.path.join(GOLD, name), 8, cwd=td)
        note(name, cmd, info, scene=4, integrator=1, width=64, height=info["height"], spp=16, seed=1)

    # EnvironmentLight WITH its HDR map.  The reference's .hdr assets do not ship either,
    # so two small synthetic Radiance RGBE pictures (flat scanlines, which stb_image's stbi_loadf
    # accepts) are written under the names scenes 24 / 26 ask for: a 32x16 equirectangular map with
    # a bright "sun" and a 16x16 square map, which the reference treats as an angular light probe.
    with tempfile.TemporaryDirectory() as td:
        for fname, w, h, sun in HDR_ASSETS.values():
            write_hdr(os.path.join(td, fname), w, h, sun)
        for sid, what in ((24, "synthetic 32x16 RGBE as brown_photostudio_02_4k.hdr (equirectangular)"),
                          (26, "synthetic 16x16 RGBE as rnl_probe.hdr (angular probe)")):
            name = "scene%02d.rtrs" % sid
            cmd, info = run("dump-scene", sid, SCENE_SEED, os.path.join(GOLD, name), cwd=td)
            note(name, cmd, info, raw_sha256=sha(os.path.join(GOLD, name)), asset=what)
            name = "lights_scene%02d.bin" % sid
            cmd, info = run("lights", sid, SCENE_SEED, 256, 99 + sid, os.path.join(GOLD, name), cwd=td)
            note(name, cmd, info, scene=sid)
            for integ in (4, 3):
                name = "li_scene%02d_i%d.bin" % (sid, integ)
                cmd, info = run("li", sid, integ, 64, 16, 1, SCENE_SEED, 768, os.path.join(GOLD, name), cwd=td)
                note(name, cmd, info, scene=sid, integrator=integ, width=64, spp=16, seed=1)
                name = "img_scene%02d_i%d_64_spp16.f64" % (sid, integ)
                cmd, info = run("render", sid, integ, 64, 16, 1, SCENE_SEED, os.path.join(GOLD, name), 8, cwd=td)
                note(name, cmd, info, scene=sid, integrator=integ, width=64, height=info["height"], spp=16, seed=1)

    # PBRMaterial with albedo / roughness / metallic / NORMAL maps from image files
    # (scene 35); ten small synthetic pictures stand in for the absent tex/*.png assets.
    with tempfile.TemporaryDirectory() as td:
        write_pbr_textures(td)
        what = "synthetic PNGs (grey, grey+alpha, RGB, RGBA, palette; 4/8/16 bit) as tex/{oak,brick,rust}/*.png"
        name = "scene35.rtrs"
        cmd, info = run("dump-scene", 35, SCENE_SEED, os.path.join(GOLD, name), cwd=td)
        note(name, cmd, info, raw_sha256=sha(os.path.join(GOLD, name)), asset=what)
        name = "hits_scene35.bin"
        cmd, info = run("hits", 35, SCENE_SEED, 768, 812, os.path.join(GOLD, name), cwd=td)
        note(name, cmd, info, scene=35)
        name = "materials_scene35.bin"
        cmd, info = run("materials", 35, SCENE_SEED, 96, 4277, os.path.join(GOLD, name), cwd=td)
        note(name, cmd, info, scene=35)
        name = "li_scene35_i4.bin"
        cmd, info = run("li", 35, 4, 64, 16, 1, SCENE_SEED, 768, os.path.join(GOLD, name), cwd=td)
        note(name, cmd, info, scene=35, integrator=4, width=64, spp=16, seed=1)
        name = "img_scene35_i4_64_spp16.f64"
        cmd, info = run("render", 35, 4, 64, 16, 1, SCENE_SEED, os.path.join(GOLD, name), 8, cwd=td)
        note(name, cmd, info, scene=35, integrator=4, width=64, height=info["height"], spp=16, seed=1)

    # breadth: every other scene id of the reference's select_scene (scenes.cpp:1523-2096), flattened and
    # rendered small by the reference itself (no assets present: missing-file fallbacks), so that
    # oracle and device are compared with the reference on each material / light / geometry mix
    for sid in ALL_OTHER_SCENES:
        name = "scene%02d.rtrs" % sid
        path = os.path.join(GOLD, name)
        cmd, info = run("dump-scene", sid, SCENE_SEED, path)
        raw_sha = sha(path)
        if os.path.getsize(path) > 60_000:
            with open(path, "rb") as f:
                data = f.read()
            with open(path + ".gz", "wb") as f:
                f.write(gzip.compress(data, 9, mtime=0))
            os.remove(path)
            note(name + ".gz", cmd, info, raw_sha256=raw_sha)
        else:
            note(name, cmd, info, raw_sha256=raw_sha)
        name = "img_scene%02d_i4_32_spp4.f64" % sid
        cmd, info = run("render", sid, 4, 32, 4, 1, SCENE_SEED, os.path.join(GOLD, name), 8)
        note(name, cmd, info, scene=sid, integrator=4, width=32, height=info["height"], spp=4, seed=1)

    # hit() vectors per geometry class -- scene ids 1001.. of the harness hold ONE object each
    # (sphere, moving_sphere, xy/xz/yz_rect, box, translate(rotate_y(box)), flip_face, constant_medium over a
    # sphere and over a transformed box), built with the reference's own constructors
    for sid in PRIMITIVE_SCENES:
        name = "scene%d.rtrs" % sid
        cmd, info = run("dump-scene", sid, SCENE_SEED, os.path.join(GOLD, name))
        note(name, cmd, info, raw_sha256=sha(os.path.join(GOLD, name)))
        name = "hits_scene%d.bin" % sid
        cmd, info = run("hits", sid, SCENE_SEED, 256, 900 + sid, os.path.join(GOLD, name))
        note(name, cmd, info, scene=sid)

    # material::sample / eval / pdf over a roughness x metallic grid of PBRMaterials (+ metal,
    # dielectric, lambertian, diffuse_light): harness scene 1011
    name = "scene1011.rtrs"
    cmd, info = run("dump-scene", 1011, SCENE_SEED, os.path.join(GOLD, name))
    note(name, cmd, info, raw_sha256=sha(os.path.join(GOLD, name)))
    name = "materials_scene1011.bin"
    cmd, info = run("materials", 1011, SCENE_SEED, 32, 4300, os.path.join(GOLD, name))
    note(name, cmd, info, scene=1011)

    # one mid-size image of the headline config's scene
    name = "img_scene21_i4_128_spp32.f64"
    cmd, info = run("render", 21, 4, 128, 32, 7, SCENE_SEED, os.path.join(GOLD, name), 8)
    note(name, cmd, info, scene=21, integrator=4, width=128, height=info["height"], spp=32, seed=7)

    png_fixtures(manifest, note)
    tie_fixtures(note)

    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    total = sum(v["bytes"] for v in manifest["files"].values())
    print("wrote %d fixtures, %.2f MB" % (len(manifest["files"]), total / 1e6))


if __name__ == "__main__":
    sys.exit(main())
