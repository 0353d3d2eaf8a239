"""Host-side mirror of the reference's ``Renderer`` / ``RenderBuffer``
(renderer/renderer.h:17-142, renderer/render_buffer.h:11-84) on top of the HIP library, and the
multi-GPU tile sharding (one process per GPU, no data-path collective)."""
import numpy as np

from . import _abi as A

TILE = 16  # renderer.h:40


class RenderBuffer:
    """``RenderBuffer`` (render_buffer.h:11-33): height x width gamma-space colours, row 0 =
    bottom row.  ``linear`` additionally keeps the linear mean radiance the device produced."""

    def __init__(self, width, height):
        self.width, self.height = int(width), int(height)
        self.pixels = np.zeros((self.height, self.width, 3), dtype=np.float64)
        self.linear = np.zeros_like(self.pixels)

    def get_data(self):
        return self.pixels

    def store_linear(self, linear, region=None):
        """write_color_to_buffer (renderer.h:126-140): sqrt gamma, clamp to [0, 1]."""
        x0, y0, x1, y1 = region if region is not None else (0, 0, self.width, self.height)
        self.linear[y0:y1, x0:x1] = linear
        self.pixels[y0:y1, x0:x1] = np.clip(np.sqrt(linear), 0.0, 1.0)

    def to_rgb8(self):
        """The bytes save_to_png writes (render_buffer.h:35-55): Y flipped, uchar(c * 255) truncation."""
        return (self.pixels[::-1] * 255.0).astype(np.uint8)

    def save_to_png(self, filename):
        """8-bit RGB PNG with the reference's pixel bytes (render_buffer.h:35-55).  The reference
        encodes with stb_image_write; the zlib stream differs, the decoded pixels do not."""
        import struct
        import zlib
        rgb = self.to_rgb8()
        raw = b"".join(b"\x00" + rgb[j].tobytes() for j in range(self.height))

        def chunk(tag, data):
            return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

        png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", self.width, self.height, 8, 2, 0, 0, 0)) +
               chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
        with open(filename, "wb") as f:
            f.write(png)
        return True


def output_filename(scene_id, integrator_id, timestamp=None):
    """``output/sceneNN_integratorK_<unixtime>.png`` (main.cpp:134-142)."""
    import time
    t = int(time.time()) if timestamp is None else int(timestamp)
    return "output/scene%02d_integrator%d_%d.png" % (scene_id, integrator_id, t)


def tiles_of_rank(width, height, rank, world):
    """Tile indices (reference dispatch order, renderer.h:61-62) owned by ``rank`` of ``world``:
    index % world == rank.  The union over ranks is every tile exactly once."""
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    return list(range(rank, tx * ty, world))


def tile_rect(width, height, tile_index):
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    y = (ty - 1) - tile_index // tx
    x = tile_index % tx
    return x * TILE, y * TILE, min(x * TILE + TILE, width), min(y * TILE + TILE, height)


def ownership_mask(width, height, rank, world):
    m = np.zeros((height, width), dtype=bool)
    for t in tiles_of_rank(width, height, rank, world):
        x0, y0, x1, y1 = tile_rect(width, height, t)
        m[y0:y1, x0:x1] = True
    return m


class Renderer:
    """``Renderer`` (renderer.h:17-120) driving one GPU through the C ABI."""

    def __init__(self, device=0, context=None):
        from .native import Context
        self._ctx = context if context is not None else Context(device)
        self._spp = 10          # Settings::samples_per_pixel default (renderer.h:20)
        self._max_depth = 50
        self._integrator = A.INTEGRATOR_MIS
        self.pipeline = A.PIPELINE_AUTO
        self.seed = 1
        self._rendering = False

    def set_integrator(self, integrator_id):
        """Integrator ids of the reference CLI (main.cpp:52): 1 = RR path, 4 = MIS path."""
        self._integrator = int(integrator_id)

    def set_samples(self, samples):
        self._spp = int(samples)

    def set_max_depth(self, depth):
        self._max_depth = int(depth)

    def cancel(self):
        self._ctx.cancel()

    def is_rendering(self):
        return self._rendering

    def render(self, scene, target_buffer, rank=0, world=1):
        """``Renderer::render(world, cam, background, target_buffer, lights)`` with the scene
        already flattened; fills the tiles ``rank`` owns."""
        self._rendering = True
        try:
            if self._ctx.scene is not scene:
                self._ctx.upload(scene)
            p = A.make_params(target_buffer.width, target_buffer.height, self._spp, integrator=self._integrator,
                              seed=self.seed, max_depth=self._max_depth, pipeline=self.pipeline, tile_first=rank,
                              tile_stride=world, spp_chunks=0)
            linear = self._ctx.render(p)
            if world > 1:
                own = ownership_mask(target_buffer.width, target_buffer.height, rank, world)
                linear = np.where(own[..., None], linear, target_buffer.linear)
            target_buffer.store_linear(linear)
        finally:
            self._rendering = False
        return target_buffer


def render_sharded(render_fn, width, height, rank, world, group=None, dst=0):
    """Tile-sharded render over the ranks of a ``torch.distributed`` job.

    ``render_fn(tile_first, tile_stride) -> (H, W, 3) float64`` renders this rank's tiles (other
    pixels are ignored).  The framebuffer is gathered on the host: every rank sends its
    buffer to ``dst`` (any backend that moves CPU tensors, e.g. gloo), which keeps, per pixel, the
    value of the owning rank.  No collective takes part in the render itself.  Returns the full
    image on ``dst`` and None elsewhere."""
    part = np.ascontiguousarray(render_fn(rank, world), dtype=np.float64)
    if world == 1:
        return part
    import torch
    import torch.distributed as dist
    mine = torch.from_numpy(part)
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, parts, dst=dst, group=group)
    if rank != dst:
        return None
    out = np.zeros_like(part)
    covered = np.zeros((height, width), dtype=np.int32)
    for r in range(world):
        own = ownership_mask(width, height, r, world)
        out[own] = parts[r].numpy()[own]
        covered += own
    if not np.all(covered == 1):
        raise RuntimeError("tile sharding does not partition the image")
    return out
