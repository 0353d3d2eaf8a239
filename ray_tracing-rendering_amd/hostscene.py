"""Python face of the C++ host layer (``host/rtr_scene_api.h`` + ``host/rtr_host.cpp`` ->
``librtr_host.so``): builds the BASELINE scenes through the mirrored scene-description API and
returns them flattened."""
import ctypes as C
import os

from . import _abi as A
from .scene import Scene

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
SCENE_SEED = 12345  # xorshift32 state before scene construction
SCENES = (7, 9, 21, 22, 23)


class _Info(C.Structure):
    _fields_ = [("default_width", C.c_int32), ("default_height", C.c_int32), ("default_spp", C.c_int32),
                ("reserved", C.c_int32)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "librtr_host.so")
        if not os.path.exists(path):
            raise OSError("%s is missing: run __graft_entry__.build()" % path)
        L = C.CDLL(path)
        L.rtr_host_build_scene.argtypes = [C.c_int, C.c_uint32, C.POINTER(C.POINTER(C.c_uint8)),
                                           C.POINTER(C.c_size_t), C.POINTER(_Info), C.c_char_p, C.c_size_t]
        L.rtr_host_free.argtypes = [C.POINTER(C.c_uint8)]
        L.rtr_host_free.restype = None
        _LIB = L
    return _LIB


def build_scene(scene_id, scene_seed=SCENE_SEED, with_defaults=False):
    """Flattened scene ``scene_id`` (reference numbering: 7, 9, 21, 22, 23)."""
    L = lib()
    buf = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    info = _Info()
    err = C.create_string_buffer(256)
    rc = L.rtr_host_build_scene(int(scene_id), int(scene_seed), C.byref(buf), C.byref(n), C.byref(info), err, len(err))
    if rc != 0:
        from .native import RtrError
        raise RtrError(rc, err.value.decode())
    try:
        sc = Scene.from_bytes(C.string_at(buf, n.value))
    finally:
        L.rtr_host_free(buf)
    if with_defaults:
        return sc, {"width": info.default_width, "height": info.default_height, "spp": info.default_spp}
    return sc
