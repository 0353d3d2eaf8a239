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


def _tile_view(t, width, height):
    """(H, W, 3) tensor -> view [tile row (j // 16), tile column, 16, 16, 3]; H and W must be multiples of 16."""
    return t.view(height // TILE, TILE, width // TILE, TILE, 3).permute(0, 2, 1, 3, 4)


def _tile_coords(tiles, width, height):
    import torch
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    idx = torch.as_tensor(list(tiles), dtype=torch.long)
    return (ty - 1) - idx // tx, idx % tx


def pack_tiles(fb, tiles, width, height):
    """The 16x16 tiles ``tiles`` (reference dispatch numbering) of framebuffer ``fb`` (torch tensor
    (H, W, 3), row 0 = bottom row, host or device) as one dense (n, 16, 16, 3) tensor on the same device:
    what a rank sends to the gathering rank -- its own tiles and nothing else (50 MB per GPU
    for C5 instead of the 403 MB frame).  Pixels past the image edge are zero."""
    import torch
    hp, wp = (height + TILE - 1) // TILE * TILE, (width + TILE - 1) // TILE * TILE
    if (hp, wp) != (height, width):
        padded = torch.zeros((hp, wp, 3), dtype=fb.dtype, device=fb.device)
        padded[:height, :width] = fb
        fb = padded
    ty, tx = _tile_coords(tiles, width, height)
    return _tile_view(fb.contiguous(), wp, hp)[ty.to(fb.device), tx.to(fb.device)].contiguous()


def unpack_tiles(parts, tile_lists, width, height, out=None):
    """Inverse of pack_tiles on the host: scatter each rank's dense tile block into the (H, W, 3) image.
    Returns (image as a numpy array, per-tile coverage count as a (tiles_y, tiles_x) array)."""
    import torch
    hp, wp = (height + TILE - 1) // TILE * TILE, (width + TILE - 1) // TILE * TILE
    full = torch.zeros((hp, wp, 3), dtype=torch.float64)
    if out is not None:
        full[:height, :width] = torch.from_numpy(np.ascontiguousarray(out))
    covered = torch.zeros((hp // TILE, wp // TILE), dtype=torch.int32)
    view = _tile_view(full, wp, hp)
    for part, tiles in zip(parts, tile_lists):
        ty, tx = _tile_coords(tiles, width, height)
        view[ty, tx] = part[:len(ty)].to(torch.float64)
        covered.index_put_((ty, tx), torch.ones(len(ty), dtype=torch.int32), accumulate=True)
    return full[:height, :width].numpy(), covered.numpy()


def gather_tiles(fb, width, height, rank, world, group=None, dst=0, tile_first=None, tile_stride=None):
    """Host-side framebuffer gather of a tile-sharded render: every rank packs the tiles it owns
    (``index % tile_stride == tile_first``, default rank / world), copies them to the host once and sends
    them to ``dst`` over a CPU-capable process group (gloo); no collective touches the render itself.
    Returns on ``dst`` (image, coverage) as unpack_tiles does, elsewhere (None, None).  ``bytes_sent`` of
    the last call is kept on the function for reporting."""
    import torch
    import torch.distributed as dist
    stride = world if tile_stride is None else tile_stride
    firsts = list(range(world)) if tile_first is None else [tile_first - rank + r for r in range(world)]
    lists = [tiles_of_rank(width, height, firsts[r], stride) for r in range(world)]
    mine = pack_tiles(fb, lists[rank], width, height).cpu()
    gather_tiles.bytes_sent = int(mine.numel() * mine.element_size())
    if world == 1:
        return unpack_tiles([mine], lists, width, height)
    n_max = max(len(t) for t in lists)
    if len(mine) < n_max:  # ragged shares (tile count not a multiple of world): pad to the largest
        pad = torch.zeros((n_max, TILE, TILE, 3), dtype=mine.dtype)
        pad[:len(mine)] = mine
        mine = pad
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, parts, dst=dst, group=group)
    if rank != dst:
        return None, None
    return unpack_tiles(parts, lists, width, height)


def render_sharded(render_fn, width, height, rank, world, group=None, dst=0):
    """Tile-sharded render over the ranks of a ``torch.distributed`` job.

    ``render_fn(tile_first, tile_stride) -> (H, W, 3) float64`` renders this rank's tiles (other
    pixels are ignored).  The framebuffer is gathered on the host: every rank sends the tiles it owns
    (and only those) to ``dst`` over any backend that moves CPU tensors (gloo).  No collective takes part
    in the render itself.  Returns the full image on ``dst`` and None elsewhere."""
    part = np.ascontiguousarray(render_fn(rank, world), dtype=np.float64)
    if world == 1:
        return part
    import torch
    image, covered = gather_tiles(torch.from_numpy(part), width, height, rank, world, group=group, dst=dst)
    if rank != dst:
        return None
    if not np.all(covered == 1):
        raise RuntimeError("tile sharding does not partition the image")
    return image
