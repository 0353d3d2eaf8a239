"""ctypes / numpy mirrors of the C ABI in ``include/rtr_hip.h`` and of the golden-vector
records in ``include/rtr_testrec.h``.  Pure layout: no device code, no oracle."""
import ctypes as C

import numpy as np

RTR_ABI_VERSION = 3

# status codes (rtr_status)
RTR_OK = 0
RTR_ERR_INVALID = -1
RTR_ERR_UNSUPPORTED = -2
RTR_ERR_DEVICE = -3
RTR_ERR_NO_SCENE = -4
RTR_ERR_CANCELLED = -5
RTR_ERR_NOMEM = -6

# node / material / texture / light tags
(NODE_BVH, NODE_LIST, NODE_TRANSLATE, NODE_ROTATE_Y, NODE_FLIP_FACE, NODE_MEDIUM, NODE_SPHERE,
 NODE_MOVING_SPHERE, NODE_XY_RECT, NODE_XZ_RECT, NODE_YZ_RECT) = range(11)
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_PBR, MAT_ISOTROPIC = range(6)
TEX_SOLID, TEX_CHECKER, TEX_NOISE, TEX_IMAGE = range(4)
LIGHT_QUAD, LIGHT_POINT, LIGHT_SPOT, LIGHT_DIRECTIONAL, LIGHT_ENV_UNIFORM, LIGHT_ENV_MAP = 0, 1, 2, 3, 4, 5

INTEGRATOR_PATH, INTEGRATOR_RR, INTEGRATOR_PBR, INTEGRATOR_NEE, INTEGRATOR_MIS = 0, 1, 2, 3, 4
PIPELINE_AUTO, PIPELINE_MEGAKERNEL, PIPELINE_WAVEFRONT = 0, 1, 2
FLAG_REFERENCE_ORDER = 1
FLAG_WF_PERSISTENT = 2
FLAG_SORTED_SHADING = 4

NODE_DTYPE = np.dtype([("type", "<i4"), ("a", "<i4"), ("b", "<i4"), ("reserved", "<i4"), ("f", "<f8", (10,))])
MATERIAL_DTYPE = np.dtype([("type", "<i4"), ("tex", "<i4", (4,)), ("reserved", "<i4", (3,)), ("f", "<f8", (4,))])
TEXTURE_DTYPE = np.dtype([("type", "<i4"), ("a", "<i4"), ("b", "<i4"), ("reserved", "<i4"), ("f", "<f8", (4,))])
PERLIN_DTYPE = np.dtype([("ranvec", "<f8", (256, 3)), ("perm_x", "<i4", (256,)), ("perm_y", "<i4", (256,)),
                         ("perm_z", "<i4", (256,))])
IMAGE_DTYPE = np.dtype([("width", "<i4"), ("height", "<i4"), ("offset", "<u8")])
LIGHT_DTYPE = np.dtype([("type", "<i4"), ("reserved", "<i4"), ("f", "<f8", (16,))])
CAMERA_DTYPE = np.dtype([("origin", "<f8", (3,)), ("lower_left_corner", "<f8", (3,)), ("horizontal", "<f8", (3,)),
                         ("vertical", "<f8", (3,)), ("u", "<f8", (3,)), ("v", "<f8", (3,)), ("w", "<f8", (3,)),
                         ("lens_radius", "<f8"), ("time0", "<f8"), ("time1", "<f8")])
assert NODE_DTYPE.itemsize == 96 and MATERIAL_DTYPE.itemsize == 64 and TEXTURE_DTYPE.itemsize == 48
assert PERLIN_DTYPE.itemsize == 9216 and IMAGE_DTYPE.itemsize == 16 and LIGHT_DTYPE.itemsize == 136
assert CAMERA_DTYPE.itemsize == 192
LI_RAY_DTYPE = np.dtype([("origin", "<f8", (3,)), ("direction", "<f8", (3,)), ("time", "<f8"), ("rng_state", "<u4"),
                         ("pad", "<u4")])  # rtr_li_ray
assert LI_RAY_DTYPE.itemsize == 64

# golden-vector records (rtr_testrec.h, packed)
LI_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("s", "<i4"), ("rng_exit", "<u4"), ("L", "<f8", (3,)),
                     ("n_closest", "<i4"), ("n_shadow", "<i4")])
HIT_DTYPE = np.dtype([("o", "<f8", (3,)), ("d", "<f8", (3,)), ("time", "<f8"), ("t_min", "<f8"), ("t_max", "<f8"),
                      ("rng_in", "<u4"), ("rng_out", "<u4"), ("hit", "<i4"), ("front_face", "<i4"),
                      ("material", "<i4"), ("pad", "<i4"), ("t", "<f8"), ("p", "<f8", (3,)), ("n", "<f8", (3,)),
                      ("u", "<f8"), ("v", "<f8")])
MAT_DTYPE = np.dtype([("material", "<i4"), ("front_face", "<i4"), ("rng_in", "<u4"), ("rng_out", "<u4"),
                      ("p", "<f8", (3,)), ("n", "<f8", (3,)), ("u", "<f8"), ("v", "<f8"), ("wo", "<f8", (3,)),
                      ("wi_in", "<f8", (3,)), ("sample_ok", "<i4"), ("is_specular", "<i4"),
                      ("is_transmission", "<i4"), ("pad", "<i4"), ("s_wi", "<f8", (3,)), ("s_f", "<f8", (3,)),
                      ("s_pdf", "<f8"), ("eval", "<f8", (3,)), ("pdf", "<f8"), ("emitted", "<f8", (3,))])
LIGHTREC_DTYPE = np.dtype([("light", "<i4"), ("pad", "<i4"), ("p", "<f8", (3,)), ("u", "<f8", (2,)),
                           ("dir", "<f8", (3,)), ("Li", "<f8", (3,)), ("wi", "<f8", (3,)), ("pdf", "<f8"),
                           ("dist", "<f8"), ("is_delta", "<i4"), ("pad2", "<i4"), ("pdf_dir", "<f8")])
assert LI_DTYPE.itemsize == 48 and HIT_DTYPE.itemsize == 168 and MAT_DTYPE.itemsize == 256
assert LIGHTREC_DTYPE.itemsize == 152
RNG_BLOCK_DOUBLES = 1 + 16 + 8 + 3 + 2 + 3 + 3 + 3 + 3 + 1


class CameraC(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("lower_left_corner", C.c_double * 3), ("horizontal", C.c_double * 3),
                ("vertical", C.c_double * 3), ("u", C.c_double * 3), ("v", C.c_double * 3), ("w", C.c_double * 3),
                ("lens_radius", C.c_double), ("time0", C.c_double), ("time1", C.c_double)]


class SceneDescC(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("root", C.c_int32), ("n_nodes", C.c_int32),
                ("n_list_children", C.c_int32), ("n_materials", C.c_int32), ("n_textures", C.c_int32),
                ("n_perlin", C.c_int32), ("n_images", C.c_int32), ("n_lights", C.c_int32), ("reserved", C.c_int32),
                ("n_image_bytes", C.c_uint64), ("nodes", C.c_void_p), ("list_children", C.c_void_p),
                ("materials", C.c_void_p), ("textures", C.c_void_p), ("perlin", C.c_void_p),
                ("images", C.c_void_p), ("image_bytes", C.c_void_p), ("lights", C.c_void_p),
                ("camera", CameraC), ("background", C.c_double * 3)]


class RenderParamsC(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32),
                ("x1", C.c_int32), ("y1", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("rr_start_depth", C.c_int32), ("integrator", C.c_int32), ("seed", C.c_uint32),
                ("pipeline", C.c_int32), ("tile_first", C.c_int32), ("tile_stride", C.c_int32),
                ("spp_chunks", C.c_int32), ("flags", C.c_int32)]


class RenderStatsC(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("closest_segments", C.c_uint64), ("shadow_segments", C.c_uint64),
                ("device_ms", C.c_double), ("kernel_launches", C.c_int32), ("pipeline", C.c_int32),
                ("spp_chunks", C.c_int32), ("cancelled", C.c_int32), ("flags_in_effect", C.c_int32),
                ("reserved", C.c_int32)]


def make_params(width, height, spp, *, integrator=INTEGRATOR_MIS, seed=1, max_depth=50, rr_start_depth=3,
                region=None, pipeline=PIPELINE_AUTO, tile_first=0, tile_stride=1, spp_chunks=1, flags=0):
    """Build an ``rtr_render_params``.  Defaults follow the reference driver (main.cpp:102,
    mis_path_integrator.h:237)."""
    x0, y0, x1, y1 = region if region is not None else (0, 0, width, height)
    p = RenderParamsC()
    p.image_width, p.image_height = int(width), int(height)
    p.x0, p.y0, p.x1, p.y1 = int(x0), int(y0), int(x1), int(y1)
    p.spp, p.max_depth, p.rr_start_depth = int(spp), int(max_depth), int(rr_start_depth)
    p.integrator, p.seed, p.pipeline = int(integrator), int(seed) & 0xFFFFFFFF, int(pipeline)
    p.tile_first, p.tile_stride = int(tile_first), int(tile_stride)
    p.spp_chunks = int(spp_chunks)
    p.flags = int(flags)
    return p
