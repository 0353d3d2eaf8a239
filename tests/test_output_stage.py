"""the output stage (gamma-2 store, clamp, uchar(c * 255) truncation, Y flip, PNG) pinned against
the reference's OWN writer: tests/golden/png_*.rgb8 are the pixels of PNG files that the reference's
Renderer::write_color_to_buffer (renderer.h:126-140) + RenderBuffer::save_to_png (render_buffer.h:35-55) wrote
from the linear images beside them, read back with the reference's stb_image (oracle/ref_harness.cpp `png`)."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest

import _golden as G

rtr = G.rtr
CASES = [("img_scene21_i4_64_spp16.f64", "png_scene21_i4_64_spp16.rgb8"),
         ("img_scene23_i4_64_spp16.f64", "png_scene23_i4_64_spp16.rgb8"),
         ("png_edge_in.f64", "png_edge_in.rgb8")]


def _load(src, fixture):
    info = G.MANIFEST["files"][fixture]
    w, h = info["width"], info["height"]
    lin = np.fromfile(os.path.join(G.GOLD, src), dtype="<f8").reshape(h, w, 3)
    want = np.fromfile(os.path.join(G.GOLD, fixture), dtype=np.uint8).reshape(h, w, 3)
    return lin, want, w, h


def _decode_png(path):
    """Minimal reader for the 8-bit RGB, non-interlaced PNGs save_to_png writes (filter type 0-4)."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(tag + body) & 0xFFFFFFFF == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert (depth, ctype, interlace) == (8, 2, 0)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = zlib.decompress(idat)
    out = np.zeros((h, w * 3), dtype=np.uint8)
    stride = w * 3 + 1
    for j in range(h):
        assert raw[j * stride] == 0, "save_to_png writes unfiltered scanlines"
        out[j] = np.frombuffer(raw[j * stride + 1:(j + 1) * stride], dtype=np.uint8)
    return out.reshape(h, w, 3)


@pytest.mark.parametrize("src,fixture", CASES)
def test_python_render_buffer_matches_reference_writer(src, fixture, tmp_path):
    lin, want, w, h = _load(src, fixture)
    buf = rtr.RenderBuffer(w, h)
    buf.store_linear(lin)
    assert np.array_equal(buf.to_rgb8(), want)
    path = str(tmp_path / "out.png")
    assert buf.save_to_png(path)
    assert np.array_equal(_decode_png(path), want)


@pytest.mark.parametrize("src,fixture", CASES)
def test_cpp_render_buffer_matches_reference_writer(src, fixture):
    """host/rtr_renderer.h: RenderBuffer::store_linear_rows + to_rgb8 (what save_to_ppm writes)."""
    lin, want, w, h = _load(src, fixture)
    lib = rtr.hostscene.lib()
    lib.rtr_host_output_stage.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lin = np.ascontiguousarray(lin)
    out = np.zeros((h, w, 3), dtype=np.uint8)
    assert lib.rtr_host_output_stage(lin.ctypes.data, w, h, out.ctypes.data) == 0
    assert np.array_equal(out, want)


@pytest.mark.parametrize("src,fixture", CASES)
@pytest.mark.parametrize("through_tiles", [0, 1])
def test_cpp_render_buffer_png_decodes_to_the_reference_pixels(src, fixture, through_tiles, tmp_path):
    """host/rtr_renderer.h: RenderBuffer::save_to_png (render_buffer.h:35-55), fed by rows and by packed 16x16 tiles
    (what a Renderer worker stores): the file decodes to the pixels of the reference's own PNG.  save_to_jpg exists for
    the reference's callers (main.cpp:138-151 compiles) and reports failure: the JPEG encoder is outside this path."""
    lin, want, w, h = _load(src, fixture)
    lib = rtr.hostscene.lib()
    lib.rtr_host_save_png.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
    lin = np.ascontiguousarray(lin)
    path = str(tmp_path / "cpp.png")
    assert lib.rtr_host_save_png(lin.ctypes.data, w, h, path.encode(), through_tiles) == 1
    assert np.array_equal(_decode_png(path), want)
    lib.rtr_host_save_jpg.argtypes = [C.c_int, C.c_int, C.c_char_p]
    assert lib.rtr_host_save_jpg(w, h, str(tmp_path / "cpp.jpg").encode()) == 0


def test_edge_fixture_covers_every_byte_value_step():
    """the synthetic image walks the sqrt-gamma steps: the fixture must contain clamped, zero and mid values"""
    _, want, _, _ = _load("png_edge_in.f64", "png_edge_in.rgb8")
    vals = set(np.unique(want).tolist())
    assert {0, 255}.issubset(vals) and len(vals) > 80
