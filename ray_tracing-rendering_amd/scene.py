"""Flattened scene container: numpy arrays laid out exactly as ``rtr_scene_desc``
(include/rtr_hip.h), plus the ``.rtrs`` file form (include/rtr_scene_io.h)."""
import ctypes as C
import hashlib
import struct

import numpy as np

from . import _abi as A

_MAGIC = b"RTRS0001"


class Scene:
    """An immutable flattened scene: what ``rtr_upload_scene`` consumes.

    It replaces the ``world`` / ``cam`` / ``background`` / ``lights`` arguments of the
    reference's ``Renderer::render`` (renderer/renderer.h:30-32)."""

    def __init__(self, root, nodes, list_children, materials, textures, perlin, images, image_bytes, lights,
                 camera, background):
        self.root = int(root)
        self.nodes = np.ascontiguousarray(nodes, dtype=A.NODE_DTYPE)
        self.list_children = np.ascontiguousarray(list_children, dtype="<i4")
        self.materials = np.ascontiguousarray(materials, dtype=A.MATERIAL_DTYPE)
        self.textures = np.ascontiguousarray(textures, dtype=A.TEXTURE_DTYPE)
        self.perlin = np.ascontiguousarray(perlin, dtype=A.PERLIN_DTYPE)
        self.images = np.ascontiguousarray(images, dtype=A.IMAGE_DTYPE)
        self.image_bytes = np.ascontiguousarray(image_bytes, dtype=np.uint8)
        self.lights = np.ascontiguousarray(lights, dtype=A.LIGHT_DTYPE)
        self.camera = np.ascontiguousarray(camera, dtype=A.CAMERA_DTYPE).reshape(1)
        self.background = np.ascontiguousarray(background, dtype="<f8").reshape(3)

    # ---- (de)serialisation -------------------------------------------------------------
    @classmethod
    def from_bytes(cls, buf):
        buf = bytes(buf)
        if buf[:8] != _MAGIC:
            raise ValueError("not an RTRS0001 scene")
        if len(buf) < 48:
            raise ValueError("truncated scene file")
        h = struct.unpack_from("<8i", buf, 8)
        (nb,) = struct.unpack_from("<Q", buf, 40)
        off = 48
        if min(h[1:]) < 0:
            raise ValueError("negative count in scene header")

        def take(dtype, n):
            nonlocal off
            dt = np.dtype(dtype)
            end = off + dt.itemsize * n
            if end > len(buf):
                raise ValueError("truncated scene file")
            a = np.frombuffer(buf, dtype=dt, count=n, offset=off).copy()
            off = end
            return a

        camera = take(A.CAMERA_DTYPE, 1)
        background = take("<f8", 3)
        nodes = take(A.NODE_DTYPE, h[1])
        kids = take("<i4", h[2])
        mats = take(A.MATERIAL_DTYPE, h[3])
        texs = take(A.TEXTURE_DTYPE, h[4])
        perlin = take(A.PERLIN_DTYPE, h[5])
        images = take(A.IMAGE_DTYPE, h[6])
        image_bytes = take(np.uint8, nb)
        lights = take(A.LIGHT_DTYPE, h[7])
        if off != len(buf):
            raise ValueError("trailing bytes in scene file")
        return cls(h[0], nodes, kids, mats, texs, perlin, images, image_bytes, lights, camera, background)

    @classmethod
    def load(cls, path):
        with open(path, "rb") as f:
            return cls.from_bytes(f.read())

    def to_bytes(self):
        head = _MAGIC + struct.pack("<8i", self.root, len(self.nodes), len(self.list_children),
                                    len(self.materials), len(self.textures), len(self.perlin), len(self.images),
                                    len(self.lights)) + struct.pack("<Q", len(self.image_bytes))
        parts = [head, self.camera.tobytes(), self.background.tobytes(), self.nodes.tobytes(),
                 self.list_children.tobytes(), self.materials.tobytes(), self.textures.tobytes(),
                 self.perlin.tobytes(), self.images.tobytes(), self.image_bytes.tobytes(), self.lights.tobytes()]
        return b"".join(parts)

    def save(self, path):
        with open(path, "wb") as f:
            f.write(self.to_bytes())

    def sha256(self):
        return hashlib.sha256(self.to_bytes()).hexdigest()

    # ---- C view ---------------------------------------------------------------------------
    def desc(self):
        """``rtr_scene_desc`` whose pointers alias this object's arrays (keep ``self`` alive)."""
        d = A.SceneDescC()
        d.abi_version = A.RTR_ABI_VERSION
        d.root = self.root
        d.n_nodes, d.n_list_children = len(self.nodes), len(self.list_children)
        d.n_materials, d.n_textures = len(self.materials), len(self.textures)
        d.n_perlin, d.n_images, d.n_lights = len(self.perlin), len(self.images), len(self.lights)
        d.n_image_bytes = len(self.image_bytes)
        d.nodes = self.nodes.ctypes.data
        d.list_children = self.list_children.ctypes.data
        d.materials = self.materials.ctypes.data
        d.textures = self.textures.ctypes.data
        d.perlin = self.perlin.ctypes.data
        d.images = self.images.ctypes.data
        d.image_bytes = self.image_bytes.ctypes.data
        d.lights = self.lights.ctypes.data
        C.memmove(C.byref(d.camera), self.camera.ctypes.data, 192)
        for c in range(3):
            d.background[c] = float(self.background[c])
        return d

    # ---- facts used by the host logic -------------------------------------------------------
    def has_media(self):
        """RNG is consumed inside traversal when a constant_medium exists."""
        return bool(np.any(self.nodes["type"] == A.NODE_MEDIUM))

    def census(self):
        t = self.nodes["type"]
        return {"nodes": len(t), "bvh": int(np.sum(t == A.NODE_BVH)), "lists": int(np.sum(t == A.NODE_LIST)),
                "spheres": int(np.sum((t == A.NODE_SPHERE) | (t == A.NODE_MOVING_SPHERE))),
                "rects": int(np.sum(t >= A.NODE_XY_RECT)), "media": int(np.sum(t == A.NODE_MEDIUM)),
                "materials": len(self.materials), "lights": len(self.lights)}
