"""MI355X-native path tracer behind the reference's ``Renderer::render`` boundary.

The directory name carries a hyphen (it mirrors the upstream repository name), so import it
with ``importlib.import_module("ray_tracing-rendering_amd")``.

Layout:
  csrc/      hand-written HIP kernels (gfx950: register-resident megakernel, wavefront stages) + the C ABI of
             include/rtr_hip.h -> librtr_hip.so
  host/      C++ host layer mirroring the reference's scene-description API + flattening
  scene.py   flattened scene container / .rtrs files
  native.py  ctypes binding of librtr_hip.so (fails loudly when the library is missing)
  renderer.py  Renderer / RenderBuffer mirror of renderer/renderer.h, tile sharding across ranks
"""
from . import _abi  # noqa: F401
from ._abi import (INTEGRATOR_MIS, INTEGRATOR_RR, PIPELINE_AUTO, PIPELINE_MEGAKERNEL,  # noqa: F401
                   PIPELINE_WAVEFRONT, make_params)
from .scene import Scene  # noqa: F401


def __getattr__(name):
    # device-facing pieces are imported lazily so CPU-only tooling can use Scene / _abi
    if name in ("native", "renderer", "hostscene", "build"):
        import importlib
        return importlib.import_module(__name__ + "." + name)
    if name in ("Context", "RtrError", "library_path"):
        from . import native
        return getattr(native, name)
    if name in ("Renderer", "RenderBuffer", "render_sharded", "tiles_of_rank", "gather_tiles", "pack_tiles", "unpack_tiles"):
        from . import renderer
        return getattr(renderer, name)
    raise AttributeError(name)
