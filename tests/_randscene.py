"""Seeded random scenes in the flattened form of include/rtr_hip.h, built from the same object kinds the
reference's scene builders use (scene/scenes.cpp): spheres, moving spheres, the three rects, `box`
(geometry/box.h:18-31: six rects in a hittable_list), translate(rotate_y(box)), flip_face, nested
hittable_lists, constant_medium, QuadLights over emissive rects.  Test infrastructure: the golden scenes pin
the oracle against the reference; these scenes let the pinned oracle check the device on graphs no demo
scene contains (coplanar overlapping rects, media next to transforms, many small instances, ...)."""
import numpy as np

import _golden as G

A = G.A
rtr = G.rtr


class Builder:
    def __init__(self, rng):
        self.rng = rng
        self.nodes, self.kids, self.mats, self.texs, self.lights = [], [], [], [], []
        self.uses_noise = False

    # ---- records ---------------------------------------------------------------------------------
    def node(self, type_, a=0, b=0, f=()):
        n = np.zeros(1, dtype=A.NODE_DTYPE)
        n["type"], n["a"], n["b"] = type_, a, b
        n["f"][0, :len(f)] = f
        self.nodes.append(n)
        return len(self.nodes) - 1

    def solid(self, rgb):
        t = np.zeros(1, dtype=A.TEXTURE_DTYPE)
        t["type"] = A.TEX_SOLID
        t["f"][0, :3] = rgb
        self.texs.append(t)
        return len(self.texs) - 1

    def checker(self, even, odd):
        t = np.zeros(1, dtype=A.TEXTURE_DTYPE)
        t["type"], t["a"], t["b"] = A.TEX_CHECKER, even, odd
        self.texs.append(t)
        return len(self.texs) - 1

    def noise(self, scale):
        t = np.zeros(1, dtype=A.TEXTURE_DTYPE)
        t["type"], t["a"] = A.TEX_NOISE, 0  # perlin table 0 (the scene carries scene 9's)
        t["f"][0, 0] = scale
        self.texs.append(t)
        self.uses_noise = True
        return len(self.texs) - 1

    def material(self, type_, tex=(), f=()):
        m = np.zeros(1, dtype=A.MATERIAL_DTYPE)
        m["type"] = type_
        m["tex"][0, :] = -1
        m["tex"][0, :len(tex)] = tex
        if type_ not in (A.MAT_PBR,):
            m["tex"][0, len(tex):] = 0
        m["f"][0, :len(f)] = f
        self.mats.append(m)
        return len(self.mats) - 1

    # ---- materials the reference's scenes use -------------------------------------------------------
    def random_material(self, allow_glass=True):
        r = self.rng
        k = int(r.integers(0, 6 if allow_glass else 5))
        if k == 0 and r.random() < 0.3:  # marble-like noise_texture (texture.h:78-92)
            return self.material(A.MAT_LAMBERTIAN, [self.noise(float(r.uniform(0.5, 6.0)))])
        if k == 0:
            return self.material(A.MAT_LAMBERTIAN, [self.solid(r.uniform(0.1, 0.9, 3))])
        if k == 1:
            return self.material(A.MAT_LAMBERTIAN, [self.checker(self.solid(r.uniform(0.1, 0.4, 3)),
                                                                 self.solid(r.uniform(0.6, 0.9, 3)))])
        if k == 2:
            return self.material(A.MAT_METAL, f=list(r.uniform(0.5, 0.95, 3)) + [float(r.choice([0.0, 0.1, 0.5]))])
        if k in (3, 4):
            rough, metal = float(r.choice([0.01, 0.05, 0.2, 0.4, 1.0])), float(r.choice([0.0, 0.5, 1.0]))
            return self.material(A.MAT_PBR, [self.solid(r.uniform(0.2, 0.9, 3)), self.solid([rough] * 3),
                                             self.solid([metal] * 3), -1])
        return self.material(A.MAT_DIELECTRIC, f=[1.5])

    # ---- geometry ----------------------------------------------------------------------------------------
    def sphere(self, c, radius, mat):
        return self.node(A.NODE_SPHERE, mat, f=list(c) + [radius])

    def moving_sphere(self, c0, c1, radius, mat):
        return self.node(A.NODE_MOVING_SPHERE, mat, f=list(c0) + list(c1) + [0.0, 1.0, radius])

    def rect(self, axis, a0, a1, b0, b1, k, mat):
        return self.node({"xy": A.NODE_XY_RECT, "xz": A.NODE_XZ_RECT, "yz": A.NODE_YZ_RECT}[axis], mat, f=[a0, a1, b0, b1, k])

    def hlist(self, children):
        first = len(self.kids)
        self.kids += list(children)
        return self.node(A.NODE_LIST, first, len(children))

    def box(self, p0, p1, mat):
        """geometry/box.h:18-31, the order of its six sides"""
        s = [self.rect("xy", p0[0], p1[0], p0[1], p1[1], p1[2], mat), self.rect("xy", p0[0], p1[0], p0[1], p1[1], p0[2], mat),
             self.rect("xz", p0[0], p1[0], p0[2], p1[2], p1[1], mat), self.rect("xz", p0[0], p1[0], p0[2], p1[2], p0[1], mat),
             self.rect("yz", p0[1], p1[1], p0[2], p1[2], p1[0], mat), self.rect("yz", p0[1], p1[1], p0[2], p1[2], p0[0], mat)]
        return self.hlist(s)

    def translate(self, child, off):
        return self.node(A.NODE_TRANSLATE, child, f=list(off))

    def rotate_y(self, child, degrees):
        rad = degrees * np.pi / 180.0  # hittable.h:98-100 (degrees_to_radians, then sin / cos)
        return self.node(A.NODE_ROTATE_Y, child, f=[np.sin(rad), np.cos(rad)])

    def flip_face(self, child):
        return self.node(A.NODE_FLIP_FACE, child)

    def medium(self, boundary, density, rgb):
        phase = self.material(A.MAT_ISOTROPIC, [self.solid(rgb)])
        return self.node(A.NODE_MEDIUM, boundary, phase, f=[-1.0 / density])

    def quad_light(self, q, u, v, intensity):
        n = np.cross(u, v)
        area = float(np.linalg.norm(n))
        l = np.zeros(1, dtype=A.LIGHT_DTYPE)
        l["type"] = A.LIGHT_QUAD
        l["f"][0, :] = list(q) + list(u) + list(v) + list(intensity) + list(n / area) + [area]
        self.lights.append(l)


    def simple_light(self, type_, f):
        l = np.zeros(1, dtype=A.LIGHT_DTYPE)
        l["type"] = type_
        l["f"][0, :len(f)] = f
        self.lights.append(l)


def _cat(parts, dtype):
    return np.concatenate(parts) if parts else np.zeros(0, dtype=dtype)


def random_scene(seed, n_objects=24, media=False, hollow=False, ties=True, delta_lights=False, lens=0.0, moved_media=False,
                 big_group=False):
    """One scene in front of scene 23's camera (origin (0,3,8), looking at the origin)."""
    rng = np.random.default_rng(seed)
    b = Builder(rng)
    base = G.scene(23)
    top = []
    ground = b.material(A.MAT_LAMBERTIAN, [b.checker(b.solid([0.2, 0.3, 0.1]), b.solid([0.9, 0.9, 0.9]))])
    # (not at y = 0: there sin(10 * p.y) of the checker would be the sign of the hit point's rounding noise, and a
    # last-bit difference between two libms upstream of the hit would flip whole checker cells)
    top.append(b.rect("xz", -12.0, 12.0, -12.0, 12.0, -0.013, ground))
    # an area light: emissive rect facing down behind a flip_face (scenes.cpp:779-809) + its QuadLight
    emit = b.material(A.MAT_DIFFUSE_LIGHT, [b.solid([7.0, 7.0, 7.0])])
    lx0, lx1, lz0, lz1, ly = -2.0, 2.0, -3.0, 0.0, 6.0
    top.append(b.flip_face(b.rect("xz", lx0, lx1, lz0, lz1, ly, emit)))
    b.quad_light([lx0, ly, lz0], [lx1 - lx0, 0.0, 0.0], [0.0, 0.0, lz1 - lz0], [7.0, 7.0, 7.0])

    if delta_lights:  # lighting/point_light.h, spot_light.h, directional_light.h, environmental_light.h without a map
        b.simple_light(A.LIGHT_POINT, [3.0, 4.5, 1.0, 9.0, 8.0, 6.0])
        d = np.array([-0.3, -1.0, -0.2])
        b.simple_light(A.LIGHT_SPOT, [-3.0, 5.0, 0.0] + list(d / np.linalg.norm(d)) + [20.0, 20.0, 25.0, np.cos(np.radians(35.0))])
        d = np.array([0.4, -1.0, 0.3])
        b.simple_light(A.LIGHT_DIRECTIONAL, list(d / np.linalg.norm(d)) + [0.6, 0.5, 0.4])
        b.simple_light(A.LIGHT_ENV_UNIFORM, [])

    def pos(lo=(-4.5, 0.3, -6.0), hi=(4.5, 3.5, 2.0)):
        return rng.uniform(lo, hi)

    for _ in range(n_objects):
        kind = int(rng.integers(0, 8))
        mat = b.random_material()
        if kind == 0:
            top.append(b.sphere(pos(), float(rng.uniform(0.15, 0.7)), mat))
        elif kind == 1:
            c = pos()
            top.append(b.moving_sphere(c, c + rng.uniform(-0.3, 0.3, 3), float(rng.uniform(0.15, 0.5)), mat))
        elif kind == 2:
            axis = ["xy", "xz", "yz"][int(rng.integers(0, 3))]
            a0, b0 = rng.uniform(-4.0, 3.0, 2)
            k = float(rng.uniform(-5.0, 3.0)) if axis != "xz" else float(rng.uniform(0.2, 3.0))
            r1 = b.rect(axis, a0, a0 + rng.uniform(0.3, 2.0), b0, b0 + rng.uniform(0.3, 2.0), k, mat)
            top.append(r1)
            if ties and rng.random() < 0.5:
                # a second rect in the SAME plane that overlaps the first one, other material: exact ties in t,
                # the reference keeps the one its hittable_list visits later (aarect.h accepts t == t_max)
                n1 = b.nodes[r1]["f"][0]
                top.append(b.rect(axis, n1[0] + 0.1, n1[1] + 0.4, n1[2] - 0.2, n1[3] - 0.05, k, b.random_material(False)))
        elif kind == 3:
            p0 = pos()
            top.append(b.box(p0, p0 + rng.uniform(0.2, 1.2, 3), mat))
        elif kind in (4, 5):
            size = rng.uniform(0.3, 1.3, 3)
            inner = b.box([0.0, 0.0, 0.0], size, mat)
            top.append(b.translate(b.rotate_y(inner, float(rng.uniform(-60.0, 60.0))), pos()))
        elif kind == 6:
            # a small group of its own (nested hittable_list), partly under a translate
            grp = [b.sphere(rng.uniform(-0.5, 0.5, 3), float(rng.uniform(0.1, 0.3)), b.random_material()) for _ in range(3)]
            top.append(b.translate(b.hlist(grp), pos()))
        else:
            top.append(b.flip_face(b.sphere(pos(), float(rng.uniform(0.2, 0.5)), mat)))
    if big_group:
        # dozens of primitives under ONE transform chain (scenes.cpp:278-287: translate(rotate_y(bvh_node(boxes2)))): an
        # instance with a box tree of its own, which a top tree over the instances walks nested on the same stack
        grp = [b.sphere(rng.uniform(-1.2, 1.2, 3), float(rng.uniform(0.08, 0.25)), b.random_material()) for _ in range(40)]
        grp += [b.box(p0, p0 + rng.uniform(0.15, 0.5, 3), b.random_material()) for p0 in rng.uniform(-1.2, 1.0, (6, 3))]
        top.append(b.translate(b.rotate_y(b.hlist(grp), float(rng.uniform(-40.0, 40.0))), pos((-3.0, 1.2, -4.0), (3.0, 2.5, 0.0))))
    if hollow:  # hollow glass (scenes.cpp:903): negative radius inside a glass sphere
        c = pos()
        glass = b.material(A.MAT_DIELECTRIC, f=[1.5])
        top.append(b.sphere(c, 0.6, glass))
        top.append(b.sphere(c, -0.5, glass))
    if media:
        c = pos()
        top.append(b.medium(b.sphere(c, 0.9, b.material(A.MAT_DIELECTRIC, f=[1.5])), 0.8, [0.2, 0.4, 0.9]))
        p0 = pos()
        top.append(b.medium(b.box(p0, p0 + rng.uniform(0.6, 1.5, 3), ground), 1.5, [0.9, 0.9, 0.9]))
        top.append(b.medium(b.sphere([0.0, 0.0, 0.0], 40.0, ground), 0.01, [1.0, 1.0, 1.0]))  # mist around everything
        order = rng.permutation(len(top))  # media anywhere in the visiting order, not only at its end
        top = [top[i] for i in order]
    if moved_media:  # a medium UNDER a transform (scenes.cpp:214-217: cornell_smoke's boxes): no step program for it
        smoke = b.medium(b.box([0.0, 0.0, 0.0], [1.2, 1.6, 1.1], ground), 1.2, [0.1, 0.1, 0.1])
        top.insert(len(top) // 2, b.translate(b.rotate_y(smoke, 25.0), pos()))
        top.insert(len(top) // 3, b.translate(b.medium(b.sphere([0.0, 0.0, 0.0], 0.8, ground), 2.0, [0.9, 0.8, 0.7]), pos()))
    root = b.hlist(top)
    camera = base.camera.copy()
    camera["lens_radius"] = lens  # camera.h:33-39: the defocus disk is drawn for every ray, used when the lens is open
    sc = rtr.Scene(root, _cat(b.nodes, A.NODE_DTYPE), np.asarray(b.kids, dtype=np.int32), _cat(b.mats, A.MATERIAL_DTYPE),
                   _cat(b.texs, A.TEXTURE_DTYPE), G.scene(9).perlin[:1] if b.uses_noise else base.perlin[:0],
                   base.images[:0], base.image_bytes[:0],
                   _cat(b.lights, A.LIGHT_DTYPE), camera, np.array([0.55, 0.65, 0.8]))
    return sc


def random_rays(seed, n):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=A.HIT_DTYPE)
    rays["o"] = rng.uniform((-6.0, 0.2, -7.0), (6.0, 6.0, 8.0), (n, 3))
    target = rng.uniform((-4.5, 0.0, -6.0), (4.5, 3.5, 2.0), (n, 3))
    rays["d"] = target - rays["o"]
    rays["time"] = rng.uniform(0.0, 1.0, n)
    rays["t_min"], rays["t_max"] = 0.001, np.inf
    rays["rng_in"] = rng.integers(1, 2 ** 32 - 1, n, dtype=np.uint64).astype(np.uint32)
    return rays


def cross_instance_tie_scene():
    """Exact ties in t between objects under DIFFERENT transform chains: faces of translated (and rotated) boxes in
    the planes of rects that the list visits before and after them.  The reference keeps whatever it visits later."""
    b = Builder(np.random.default_rng(0))
    m = [b.material(A.MAT_LAMBERTIAN, [b.solid(c)]) for c in ([.9, .1, .1], [.1, .9, .1], [.1, .1, .9], [.9, .9, .1],
                                                             [.1, .9, .9], [.9, .1, .9], [.5, .5, .5])]
    floor_a = b.rect("xz", -5, 5, -5, 5, 0.0, m[0])
    back = b.rect("xy", -5, 5, -1, 4, 1.5, m[5])
    t1 = b.translate(b.rotate_y(b.box([0, 0, 0], [1, 1.5, 1], m[1]), 20.0), [0.5, 0.0, -1.0])  # bottom in y = 0
    floor_b = b.rect("xz", -1, 3, -3, 1, 0.0, m[2])  # same transform chain as floor_a, visited AFTER t1
    t2 = b.translate(b.box([0, 0, 0], [1, 1, 1], m[3]), [-2.0, 0.0, 0.5])  # faces in x = -2, z = 1.5, y = 0
    side = b.rect("yz", -1, 4, -5, 5, -2.0, m[4])  # visited after t2
    t3 = b.translate(b.translate(b.box([0, 0, 0], [0.5, 0.5, 0.5], m[6]), [1.0, 0.0, 0.0]), [1.5, 0.0, 1.0])  # z = 1.5 again
    root = b.hlist([floor_a, back, t1, floor_b, t2, side, t3])
    base = G.scene(23)
    return rtr.Scene(root, _cat(b.nodes, A.NODE_DTYPE), np.asarray(b.kids, dtype=np.int32), _cat(b.mats, A.MATERIAL_DTYPE),
                     _cat(b.texs, A.TEXTURE_DTYPE), base.perlin[:0], base.images[:0], base.image_bytes[:0], base.lights[:0],
                     base.camera.copy(), np.array([0.5, 0.6, 0.8]))


def cross_instance_tie_rays(n=1536):
    rng = np.random.default_rng(5)
    r = np.zeros(n, dtype=A.HIT_DTYPE)
    k = n // 3
    o, t = np.empty((n, 3)), np.empty((n, 3))
    o[:k] = rng.uniform((-3, -3, -3), (3.5, -0.5, 3), (k, 3))  # from below the floors, up into the boxes
    t[:k] = rng.uniform((-2.2, 0.2, -1.2), (3.2, 1.2, 1.8), (k, 3))
    o[k:2 * k] = rng.uniform((-6, 0.1, -1), (-2.5, 2, 3), (k, 3))  # from x < -2 into the translated box
    t[k:2 * k] = rng.uniform((-1.9, 0.1, 0.6), (-1.1, 0.9, 1.4), (k, 3))
    o[2 * k:] = rng.uniform((-3, 0.1, 2), (4, 2, 5), (n - 2 * k, 3))  # from z > 1.5
    t[2 * k:] = rng.uniform((-1.9, 0.05, 0.6), (3.0, 0.9, 1.45), (n - 2 * k, 3))
    r["o"], r["d"] = o, t - o
    r["time"], r["t_min"], r["t_max"], r["rng_in"] = 0.5, 0.001, np.inf, 7
    return r
