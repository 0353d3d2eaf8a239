"""Multi-GPU path on CPU: tile ownership (renderer.h:40-62 numbering) and the host-side
framebuffer gather over a world_size-2 gloo job.  The per-rank renderer is injected; here it is
the CPU oracle (test infrastructure), on the GPU box bench.py uses the HIP context."""
import os
import socket
import sys

import numpy as np
import pytest

import _golden as G

A = G.A
rtr = G.rtr


@pytest.mark.parametrize("w,h,world", [(800, 800, 8), (40, 27, 3), (1920, 1080, 4), (17, 16, 2)])
def test_tiles_partition(w, h, world):
    seen = []
    for r in range(world):
        seen += rtr.renderer.tiles_of_rank(w, h, r, world)
    tx, ty = (w + 15) // 16, (h + 15) // 16
    assert sorted(seen) == list(range(tx * ty))
    cover = sum(rtr.renderer.ownership_mask(w, h, r, world).astype(int) for r in range(world))
    assert np.all(cover == 1)


def test_tile_order_is_top_row_first():
    # renderer.h:61-62: tile 0 is the top-left tile (largest y)
    assert rtr.renderer.tile_rect(64, 48, 0) == (0, 32, 16, 48)
    assert rtr.renderer.tile_rect(64, 48, 5) == (16, 16, 32, 32)
    assert rtr.renderer.tile_rect(40, 27, 2) == (32, 16, 40, 27)


def test_render_buffer_store_matches_reference_output_stage():
    rb = rtr.RenderBuffer(4, 2)
    lin = np.array([[[0.25, 4.0, 0.0]] * 4] * 2)
    rb.store_linear(lin)
    assert np.allclose(rb.get_data()[0, 0], [0.5, 1.0, 0.0])  # sqrt gamma, clamp (renderer.h:126-140)
    assert rb.to_rgb8()[0, 0].tolist() == [127, 255, 0]        # uchar(c * 255) truncation (render_buffer.h:44-49)


def test_png_writer_and_filename(tmp_path):
    """PNG with the reference's pixel bytes and its output file naming."""
    import struct
    import zlib
    rb = rtr.RenderBuffer(5, 3)
    rng = np.random.default_rng(0)
    rb.store_linear(rng.random((3, 5, 3)) * 1.5)
    path = str(tmp_path / "x.png")
    assert rb.save_to_png(path)
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    w, h, depth, ctype = struct.unpack(">IIBB", data[16:26])
    assert (w, h, depth, ctype) == (5, 3, 8, 2)
    idat_len = struct.unpack(">I", data[33:37])[0]
    raw = zlib.decompress(data[41:41 + idat_len])
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(3, 1 + 5 * 3)[:, 1:].reshape(3, 5, 3)
    assert np.array_equal(rows, rb.to_rgb8())
    assert rtr.renderer.output_filename(7, 1, 1765799945) == "output/scene07_integrator1_1765799945.png"


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = G.scene(21)
        W, H, spp = 40, 40, 2

        def render_fn(first, stride):  # oracle stands in for the GPU here
            out = np.full((H, W, 3), -1.0)
            p = A.make_params(W, H, spp, seed=11, tile_first=first, tile_stride=stride)
            d = sc.desc()
            assert G.oracle().rto_render(G.C.byref(d), G.C.byref(p), out.ctypes.data, W, 1, None) == 0
            return out

        img = rtr.render_sharded(render_fn, W, H, rank, world)
        if rank == 0:
            full, _ = G.oracle_render(sc, A.make_params(W, H, spp, seed=11), threads=1)
            q.put(bool(np.array_equal(img, full)))
        else:
            assert img is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_gather_is_bit_identical_to_one_rank(world):
    """world 2, and the BASELINE's 8: a 40x40 image has 9 tiles, so with 8 ranks the shares are ragged (rank 0 owns two
    tiles, the others one) -- the N = 8 gather and coverage bookkeeping run here even where no 8-GPU node is at hand."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_cpp_renderer_tile_bookkeeping_matches_python():
    """host/rtr_renderer.h (the N-context Renderer): which context owns a pixel == index % n of the reference's
    tile numbering (renderer.h:61-62), the same partition render_sharded / bench.py use."""
    import ctypes as C
    import importlib
    rtr = importlib.import_module("ray_tracing-rendering_amd")
    lib = rtr.hostscene.lib()
    lib.rtr_host_tile_owner.argtypes = [C.c_int] * 5
    for (W, H) in ((64, 64), (70, 50), (800, 800), (33, 17)):
        for n in (1, 2, 3, 8):
            owner = np.array([[lib.rtr_host_tile_owner(W, H, i, j, n) for i in range(0, W, 5)] for j in range(0, H, 3)])
            want = np.zeros_like(owner)
            for r in range(n):
                m = rtr.renderer.ownership_mask(W, H, r, n)[::3, ::5]
                want[m] = r
            assert np.array_equal(owner, want), (W, H, n)
