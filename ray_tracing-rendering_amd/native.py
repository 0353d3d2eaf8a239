"""ctypes binding of ``librtr_hip.so`` (the C ABI of include/rtr_hip.h).

There is no CPU fallback: if the library is missing or no GPU is visible, every device entry
point raises.  The oracle under ``oracle/`` is test infrastructure and is never loaded here."""
import ctypes as C
import os

import numpy as np

from . import _abi as A
from .scene import Scene

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_STATUS = {A.RTR_ERR_INVALID: "RTR_ERR_INVALID", A.RTR_ERR_UNSUPPORTED: "RTR_ERR_UNSUPPORTED",
           A.RTR_ERR_DEVICE: "RTR_ERR_DEVICE", A.RTR_ERR_NO_SCENE: "RTR_ERR_NO_SCENE",
           A.RTR_ERR_CANCELLED: "RTR_ERR_CANCELLED", A.RTR_ERR_NOMEM: "RTR_ERR_NOMEM"}

# every symbol include/rtr_hip.h declares (librtr_hip.so) ...
EXPORTS = ("rtr_abi_version", "rtr_device_count", "rtr_create", "rtr_destroy", "rtr_set_stream",
           "rtr_upload_scene", "rtr_render_device", "rtr_render_host", "rtr_render_tiles_host", "rtr_plan_chunks", "rtr_li_samples", "rtr_li_rays",
           "rtr_synchronize", "rtr_cancel", "rtr_get_stats", "rtr_last_error", "rtr_sample_seed", "rtr_validate_scene")
# ... and include/rtr_hip_test.h (librtr_hip_test.so: device unit kernels of the parity tests, not part of the product)
TEST_EXPORTS = ("rtr_test_hits", "rtr_test_materials", "rtr_test_lights", "rtr_test_li", "rtr_test_reference_order",
                "rtr_test_stream8", "rtr_test_sincos_exhaustive", "rtr_test_shared_division", "rtr_test_issue_rates")
_TEST_LIB = None


class SceneInfoC(C.Structure):
    _fields_ = [("stack_words", C.c_int32), ("has_media", C.c_int32), ("needs_uv", C.c_int32),
                ("graph_depth", C.c_int32), ("fast_ok", C.c_int32), ("fast_instances", C.c_int32),
                ("fast_refs", C.c_int32), ("fast_stack_words", C.c_int32), ("compiled_subtrees", C.c_int32),
                ("program_steps", C.c_int32), ("inverted_boxes", C.c_int32), ("top_trees", C.c_int32)]


class RtrError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (_STATUS.get(code, "rtr_status"), code, message))
        self.code = code
        self.message = message


def library_path():
    # RTR_HIP_LIBRARY lets a tuning run point at an alternative build of the same ABI
    return os.environ.get("RTR_HIP_LIBRARY") or os.path.join(_HERE, "librtr_hip.so")


def lib():
    """Load librtr_hip.so; raises if it has not been built (``__graft_entry__.build()``)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RtrError(A.RTR_ERR_DEVICE, "HIP library %s is missing: run __graft_entry__.build(); there is no "
                       "CPU fallback" % path)
    L = C.CDLL(path)
    P = C.POINTER
    vp = C.c_void_p
    L.rtr_abi_version.restype = C.c_uint32
    L.rtr_device_count.restype = C.c_int
    L.rtr_create.argtypes = [C.c_int, P(vp)]
    L.rtr_destroy.argtypes = [vp]
    L.rtr_destroy.restype = None
    L.rtr_set_stream.argtypes = [vp, vp]
    L.rtr_upload_scene.argtypes = [vp, P(A.SceneDescC)]
    L.rtr_render_device.argtypes = [vp, P(A.RenderParamsC), vp, C.c_int64, C.c_int]
    L.rtr_render_host.argtypes = [vp, P(A.RenderParamsC), vp, C.c_int64]
    L.rtr_plan_chunks.argtypes = [vp, P(A.RenderParamsC)]
    L.rtr_li_samples.argtypes = [vp, P(A.RenderParamsC), vp, vp, C.c_int64]
    L.rtr_synchronize.argtypes = [vp]
    L.rtr_cancel.argtypes = [vp]
    L.rtr_get_stats.argtypes = [vp, P(A.RenderStatsC)]
    L.rtr_last_error.argtypes = [vp]
    L.rtr_last_error.restype = C.c_char_p
    L.rtr_sample_seed.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.rtr_sample_seed.restype = C.c_uint32
    L.rtr_validate_scene.argtypes = [P(A.SceneDescC), P(SceneInfoC), C.c_char_p, C.c_size_t]
    L.rtr_li_rays.argtypes = [vp, P(A.RenderParamsC), vp, vp, C.c_int64]
    if L.rtr_abi_version() != A.RTR_ABI_VERSION:
        raise RtrError(A.RTR_ERR_INVALID, "librtr_hip.so ABI version mismatch")
    _LIB = L
    return L


def test_lib():
    """librtr_hip_test.so (include/rtr_hip_test.h): the device unit kernels the parity tests and tools/ drive.  Loaded on
    first use; never needed to render."""
    global _TEST_LIB
    if _TEST_LIB is not None:
        return _TEST_LIB
    lib()  # the product library first: the test library links against it
    path = os.environ.get("RTR_HIP_TEST_LIBRARY") or os.path.join(os.path.dirname(library_path()), "librtr_hip_test.so")
    if not os.path.exists(path):
        path = os.path.join(_HERE, "librtr_hip_test.so")
    if not os.path.exists(path):
        raise RtrError(A.RTR_ERR_DEVICE, "test library %s is missing: run __graft_entry__.build()" % path)
    T = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    for name in ("rtr_test_hits", "rtr_test_materials", "rtr_test_lights"):
        getattr(T, name).argtypes = [vp, vp, C.c_int64]
    T.rtr_test_li.argtypes = [vp, C.POINTER(A.RenderParamsC), vp, C.c_int64]
    T.rtr_test_reference_order.argtypes = [vp, C.c_int]
    T.rtr_test_stream8.argtypes = [vp, C.c_int64, C.c_int]
    T.rtr_test_sincos_exhaustive.argtypes = [vp, C.POINTER(C.c_uint64)]
    T.rtr_test_issue_rates.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
    T.rtr_test_shared_division.argtypes = [vp, C.POINTER(C.c_uint64)]
    _TEST_LIB = T
    return T


def validate_scene(scene):
    """Host-only scene check (no GPU needed).  Returns the scene facts; raises RtrError."""
    L = lib()
    d = scene.desc()
    info = SceneInfoC()
    msg = C.create_string_buffer(256)
    rc = L.rtr_validate_scene(C.byref(d), C.byref(info), msg, len(msg))
    if rc != 0:
        raise RtrError(rc, msg.value.decode())
    return {"stack_words": info.stack_words, "has_media": bool(info.has_media), "needs_uv": bool(info.needs_uv),
            "graph_depth": info.graph_depth, "fast_ok": bool(info.fast_ok), "fast_instances": info.fast_instances,
            "fast_refs": info.fast_refs, "fast_stack_words": info.fast_stack_words,
            "compiled_subtrees": info.compiled_subtrees, "program_steps": info.program_steps,
            "inverted_boxes": info.inverted_boxes, "top_trees": info.top_trees}


class Context:
    """One GPU.  Mirrors the life cycle of the reference's ``Renderer`` object
    (renderer/renderer.h:22-28): create, give it a scene, render, cancel."""

    def __init__(self, device=0):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.rtr_create(int(device), C.byref(h))
        if rc != 0:
            raise RtrError(rc, self._L.rtr_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self.scene = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.rtr_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RtrError(rc, self._L.rtr_last_error(self._h).decode())

    def set_stream(self, stream_handle):
        """Run the kernels on an existing hipStream_t (e.g. ``torch.cuda.current_stream().cuda_stream``)."""
        self._chk(self._L.rtr_set_stream(self._h, C.c_void_p(stream_handle or None)))

    def upload(self, scene):
        if not isinstance(scene, Scene):
            raise TypeError("expected a flattened Scene")
        d = scene.desc()
        self._chk(self._L.rtr_upload_scene(self._h, C.byref(d)))
        self.scene = scene

    def render_into(self, params, device_ptr, row_stride, blocking=False):
        """Linear mean radiance of params' region into a device buffer of doubles."""
        self._chk(self._L.rtr_render_device(self._h, C.byref(params), C.c_void_p(device_ptr), int(row_stride),
                                            1 if blocking else 0))

    def render(self, params, out=None):
        """Blocking render to a host array (H, W, 3) of the region (includes the D2H copy).  Pixels of
        tiles the call does not own or, after a cancel, did not finish keep the values of ``out``."""
        h, w = params.y1 - params.y0, params.x1 - params.x0
        if out is None:
            out = np.zeros((h, w, 3), dtype=np.float64)
        if out.shape != (h, w, 3) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float64 array of shape (%d, %d, 3)" % (h, w))
        self._chk(self._L.rtr_render_host(self._h, C.byref(params), out.ctypes.data, w))
        return out

    def plan_chunks(self, params):
        """Partial sums per pixel the library would use for ``params`` (its own choice when spp_chunks = 0)."""
        n = self._L.rtr_plan_chunks(self._h, C.byref(params))
        if n < 0:
            self._chk(n)
        return n

    def li_samples(self, params, ijs):
        """``Integrator::Li`` of the camera samples ``ijs`` ((n, 3) int32: pixel i, pixel j, sample index):
        radiance (n, 3), not divided by spp."""
        ijs = np.ascontiguousarray(ijs, dtype=np.int32).reshape(-1, 3)
        out = np.zeros((len(ijs), 3), dtype=np.float64)
        self._chk(self._L.rtr_li_samples(self._h, C.byref(params), ijs.ctypes.data, out.ctypes.data, len(ijs)))
        return out

    def li_rays(self, params, origins, directions, times, rng_states):
        """``Integrator::Li`` of arbitrary rays (n, 3) / (n, 3) / (n,) with the xorshift32 state (n,) the reference's
        generator would hold on entry: radiance (n, 3)."""
        n = len(origins)
        rays = np.zeros(n, dtype=A.LI_RAY_DTYPE)
        rays["origin"], rays["direction"], rays["time"], rays["rng_state"] = origins, directions, times, rng_states
        out = np.zeros((n, 3), dtype=np.float64)
        self._chk(self._L.rtr_li_rays(self._h, C.byref(params), rays.ctypes.data, out.ctypes.data, n))
        return out

    def synchronize(self):
        self._chk(self._L.rtr_synchronize(self._h))

    def cancel(self):
        self._chk(self._L.rtr_cancel(self._h))

    def stats(self):
        s = A.RenderStatsC()
        self._chk(self._L.rtr_get_stats(self._h, C.byref(s)))
        return {"samples": s.samples, "closest_segments": s.closest_segments, "shadow_segments": s.shadow_segments,
                "device_ms": s.device_ms, "kernel_launches": s.kernel_launches, "pipeline": s.pipeline,
                "spp_chunks": s.spp_chunks, "cancelled": bool(s.cancelled), "flags_in_effect": s.flags_in_effect}

    def reference_order(self, on):
        """Force the reference-order traversal for rtr_test_hits (renders use params.flags)."""
        self._chk(test_lib().rtr_test_reference_order(self._h, 1 if on else 0))

    def stream8(self, n_doubles, repeat=1):
        """Counter calibration: stream n_doubles doubles in and out, 8 bytes per lane (include/rtr_hip_test.h)."""
        self._chk(test_lib().rtr_test_stream8(self._h, int(n_doubles), int(repeat)))

    def sincos_mismatches(self):
        """All 2^32 sampler angles: how many give sincos(phi) != (sin(phi), cos(phi)) in some bit (include/rtr_hip_test.h)."""
        n = C.c_uint64(0)
        self._chk(test_lib().rtr_test_sincos_exhaustive(self._h, C.byref(n)))
        return int(n.value)

    ISSUE_CLASSES = ("v_fma_f64", "v_add_f64", "v_mul_f64", "v_rcp_f64", "v_rsq_f64", "v_cmp_lt_f64", "v_cndmask_b32",
                     "v_mov_b32", "v_fma_f32", "s_and_b64", "v_div_scale_f64", "v_div_fixup_f64", "v_cmp_f64+s_and_b64",
                     "v_cndmask_b32 (scalar-pair mask)", "v_min_f32", "v_cmp_f32+v_cndmask_b32", "v_cndmask_b32 (4 destinations)",
                     "v_cmp_f64+v_cndmask_b32")

    def issue_rates(self):
        """Shader cycles per wave-instruction and class with four waves on every SIMD (include/rtr_hip_test.h)."""
        out = (C.c_double * len(self.ISSUE_CLASSES))()
        self._chk(test_lib().rtr_test_issue_rates(self._h, out, len(self.ISSUE_CLASSES)))
        return dict(zip(self.ISSUE_CLASSES, [float(x) for x in out]))

    def shared_division_mismatches(self):
        """2^32 operand pairs: how many quotients of the shared-reciprocal division differ from n / d (include/rtr_hip_test.h)."""
        n = C.c_uint64(0)
        self._chk(test_lib().rtr_test_shared_division(self._h, C.byref(n)))
        return int(n.value)

    # device unit kernels over golden-vector records (include/rtr_hip_test.h)
    def test_records(self, kind, recs, params=None):
        out = np.ascontiguousarray(recs.copy())
        fn = getattr(test_lib(), "rtr_test_" + kind)
        if kind == "li":
            self._chk(fn(self._h, C.byref(params), out.ctypes.data, len(out)))
        else:
            self._chk(fn(self._h, out.ctypes.data, len(out)))
        return out
