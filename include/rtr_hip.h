/*
 * rtr_hip.h -- C ABI of the MI355X (gfx950) path tracer.
 *
 * This is the drop-in boundary for ONE hot path of JiGuang283/Ray_Tracing-Rendering:
 * the tile-threaded integrator loop behind
 *     void Renderer::render(shared_ptr<hittable> world, shared_ptr<camera> cam,
 *                           const color& background, RenderBuffer& target,
 *                           const std::vector<shared_ptr<Light>>& lights)
 * (reference src/renderer/renderer.h:30-102).  The reference has no C plugin ABI;
 * the entry points below are what an FFI for that call would bind.  Plain
 * pointers and sizes only: no C++ types, no torch types.
 *
 * Data model: the caller flattens the (immutable) scene graph into the POD
 * arrays of `rtr_scene_desc`, uploads it once, then asks for linear mean
 * radiance of a pixel region.  The host applies the reference's gamma-2 /
 * clamp store (renderer.h:126-140) itself.
 *
 * All scene quantities are IEEE double, like the reference (core/vec3.h:88).
 * All functions return RTR_OK (0) or a negative rtr_status; the text of the
 * last failure of a context is available from rtr_last_error().  No exception
 * crosses this boundary.
 */
#ifndef RTR_HIP_H
#define RTR_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTR_ABI_VERSION 3 /* 2: rtr_render_stats grew spp_chunks / cancelled; 3: ... flags_in_effect */

/* ------------------------------------------------------------------------- */
/* status codes                                                              */
typedef enum rtr_status {
    RTR_OK = 0,
    RTR_ERR_INVALID = -1,     /* bad argument / malformed scene            */
    RTR_ERR_UNSUPPORTED = -2, /* scene uses a node/material the device lacks */
    RTR_ERR_DEVICE = -3,      /* HIP runtime error                           */
    RTR_ERR_NO_SCENE = -4,    /* render before upload                        */
    RTR_ERR_CANCELLED = -5,   /* rtr_cancel() observed (partial image)       */
    RTR_ERR_NOMEM = -6
} rtr_status;

/* ------------------------------------------------------------------------- */
/* flattened scene graph                                                     */

/* hittable nodes: one record per object of the reference's hittable graph
 * (geometry/hittable.h:25-32).  Indices refer to rtr_scene_desc.nodes.
 * The graph may be a DAG (bvh_node with left == right, bvh.h:68-69; a sphere
 * that is both a world object and a medium boundary, scenes.cpp:255-258). */
typedef enum rtr_node_type {
    RTR_NODE_BVH = 0,           /* bvh_node        geometry/bvh.h:12-31        a=left b=right f[0..5]=box min,max */
    RTR_NODE_LIST = 1,          /* hittable_list   geometry/hittable_list.h    a=first (into list_children) b=count */
    RTR_NODE_TRANSLATE = 2,     /* translate       geometry/hittable.h:34-73   a=child f[0..2]=offset */
    RTR_NODE_ROTATE_Y = 3,      /* rotate_y        geometry/hittable.h:75-156  a=child f[0]=sin f[1]=cos */
    RTR_NODE_FLIP_FACE = 4,     /* flip_face       geometry/hittable.h:158-179 a=child */
    RTR_NODE_MEDIUM = 5,        /* constant_medium geometry/constant_medium.h  a=boundary b=phase material f[0]=neg_inv_density */
    RTR_NODE_SPHERE = 6,        /* sphere          geometry/sphere.h           a=material f[0..2]=center f[3]=radius */
    RTR_NODE_MOVING_SPHERE = 7, /* moving_sphere   geometry/moving_sphere.h    a=material f[0..2]=c0 f[3..5]=c1 f[6]=t0 f[7]=t1 f[8]=radius */
    RTR_NODE_XY_RECT = 8,       /* xy_rect         geometry/aarect.h:10-32     a=material f[0]=x0 f[1]=x1 f[2]=y0 f[3]=y1 f[4]=k */
    RTR_NODE_XZ_RECT = 9,       /* xz_rect         geometry/aarect.h:34-54     a=material f[0]=x0 f[1]=x1 f[2]=z0 f[3]=z1 f[4]=k */
    RTR_NODE_YZ_RECT = 10,      /* yz_rect         geometry/aarect.h:56-76     a=material f[0]=y0 f[1]=y1 f[2]=z0 f[3]=z1 f[4]=k */
    RTR_NODE_TYPE_COUNT = 11
} rtr_node_type;

typedef struct rtr_node {
    int32_t type; /* rtr_node_type */
    int32_t a;
    int32_t b;
    int32_t reserved;
    double f[10];
} rtr_node; /* 96 bytes */

/* materials (materials/material.h:72-439, geometry/constant_medium.h:12-29) */
typedef enum rtr_material_type {
    RTR_MAT_LAMBERTIAN = 0,    /* tex[0]=albedo */
    RTR_MAT_METAL = 1,         /* f[0..2]=albedo f[3]=fuzz (already clamped <=1) */
    RTR_MAT_DIELECTRIC = 2,    /* f[0]=ir */
    RTR_MAT_DIFFUSE_LIGHT = 3, /* tex[0]=emit */
    RTR_MAT_PBR = 4,           /* tex[0]=albedo tex[1]=roughness tex[2]=metallic tex[3]=normal_map or -1 */
    RTR_MAT_ISOTROPIC = 5,     /* tex[0]=albedo */
    RTR_MAT_TYPE_COUNT = 6
} rtr_material_type;

typedef struct rtr_material {
    int32_t type;
    int32_t tex[4];
    int32_t reserved[3];
    double f[4];
} rtr_material; /* 64 bytes */

/* textures (materials/texture.h:11-162) */
typedef enum rtr_texture_type {
    RTR_TEX_SOLID = 0,   /* f[0..2]=color */
    RTR_TEX_CHECKER = 1, /* a=even texture b=odd texture */
    RTR_TEX_NOISE = 2,   /* a=perlin table index f[0]=scale */
    RTR_TEX_IMAGE = 3,   /* a=image index, or -1 = file missing (cyan fallback, texture.h:116-118) */
    RTR_TEX_TYPE_COUNT = 4
} rtr_texture_type;

typedef struct rtr_texture {
    int32_t type;
    int32_t a;
    int32_t b;
    int32_t reserved;
    double f[4];
} rtr_texture; /* 48 bytes */

/* perlin noise tables (materials/perlin.h:10-19,57-60) */
typedef struct rtr_perlin {
    double ranvec[256][3];
    int32_t perm_x[256];
    int32_t perm_y[256];
    int32_t perm_z[256];
} rtr_perlin; /* 9216 bytes */

/* 8-bit RGB images for image_texture (materials/texture.h:96-107); image_bytes also holds the
 * float texels and sampling tables of RTR_LIGHT_ENV_MAP lights */
typedef struct rtr_image {
    int32_t width;
    int32_t height;
    uint64_t offset; /* byte offset of texel (0,0) in rtr_scene_desc.image_bytes, rows of 3*width bytes */
} rtr_image;

/* lights (lighting/light.h:15-47) */
typedef enum rtr_light_type {
    RTR_LIGHT_QUAD = 0,        /* QuadLight lighting/quad_light.h:9-92: f[0..2]=Q f[3..5]=u f[6..8]=v f[9..11]=intensity f[12..14]=normal f[15]=area */
    RTR_LIGHT_POINT = 1,       /* PointLight lighting/point_light.h:6-37: f[0..2]=position f[3..5]=intensity */
    RTR_LIGHT_SPOT = 2,        /* SpotLight lighting/spot_light.h:6-41: f[0..2]=position f[3..5]=unit direction f[6..8]=intensity f[9]=cos_cutoff */
    RTR_LIGHT_DIRECTIONAL = 3, /* DirectionalLight lighting/directional_light.h:7-31: f[0..2]=unit direction f[3..5]=radiance */
    RTR_LIGHT_ENV_UNIFORM = 4, /* EnvironmentLight whose map file is missing (lighting/environmental_light.h:126-131,
                                  187-192,251-252,293-294): uniform white sky, sampled with random_unit_vector() */
    RTR_LIGHT_ENV_MAP = 5,     /* EnvironmentLight with its HDR map (lighting/environmental_light.h:120-374):
                                  f[0]=width f[1]=height f[2]=is_light_probe (square map = angular probe, :137-140)
                                  f[3]=byte offset in rtr_scene_desc.image_bytes of the float32 RGB texels as stbi_loadf
                                       returns them (row 0 first, 3*width floats per row; multiple of 4)
                                  f[4]=byte offset (multiple of 8) of the float64 sampling tables of Distribution2D
                                       (:60-117): per map row v {func[width], cdf[width+1], func_int}, then the
                                       marginal {func[height], cdf[height+1], func_int} */
    RTR_LIGHT_TYPE_COUNT = 6
} rtr_light_type;

typedef struct rtr_light {
    int32_t type;
    int32_t reserved;
    double f[16];
} rtr_light; /* 136 bytes */

/* thin-lens camera, the private state of renderer/camera.h:43-50 */
typedef struct rtr_camera {
    double origin[3];
    double lower_left_corner[3];
    double horizontal[3];
    double vertical[3];
    double u[3];
    double v[3];
    double w[3];
    double lens_radius;
    double time0;
    double time1;
} rtr_camera; /* 192 bytes */

typedef struct rtr_scene_desc {
    uint32_t abi_version; /* RTR_ABI_VERSION */
    int32_t root;         /* node index of `world` */
    int32_t n_nodes;
    int32_t n_list_children;
    int32_t n_materials;
    int32_t n_textures;
    int32_t n_perlin;
    int32_t n_images;
    int32_t n_lights;
    int32_t reserved;
    uint64_t n_image_bytes;
    const rtr_node* nodes;
    const int32_t* list_children; /* node indices, hittable_list order */
    const rtr_material* materials;
    const rtr_texture* textures;
    const rtr_perlin* perlin;
    const rtr_image* images;
    const uint8_t* image_bytes;
    const rtr_light* lights;
    rtr_camera camera;
    double background[3];
} rtr_scene_desc;

/* ------------------------------------------------------------------------- */
/* render request                                                            */

/* integrator ids follow the reference CLI (main.cpp:52,78-100) */
#define RTR_INTEGRATOR_PATH 0 /* PathIntegrator        renderer/path_integrator.h:22-44 (summed front to back on the device) */
#define RTR_INTEGRATOR_RR 1   /* RRPathInterator       renderer/rr_path_integrator.h:21-59  */
#define RTR_INTEGRATOR_PBR 2  /* PBRPathIntegrator     renderer/pbr_path_integrator.h:21-73 */
#define RTR_INTEGRATOR_NEE 3  /* DirectLightIntegrator renderer/direct_light_integrator.h:25-142 */
#define RTR_INTEGRATOR_MIS 4  /* MISPathIntegrator     renderer/mis_path_integrator.h:25-150 */

/* device pipeline selection */
#define RTR_PIPELINE_AUTO 0       /* the megakernel (the faster one on every measured scene: DESIGN.md 4.4) */
#define RTR_PIPELINE_MEGAKERNEL 1 /* one lane per pixel, in-register bounce loop     */
#define RTR_PIPELINE_WAVEFRONT 2  /* SoA path pool in HBM, extend/shade/connect stages over live-block lists; all five
                                     integrators; compiled traversals only (RTR_ERR_UNSUPPORTED for graphs that need the
                                     reference-order walk -- e.g. a list holding a medium under a transform -- or with
                                     RTR_FLAG_REFERENCE_ORDER; hollow spheres and a medium straight under transforms are
                                     compiled) */

typedef struct rtr_render_params {
    int32_t image_width;  /* W of the full image (pixel (i,j), j=0 is the bottom row, renderer.h:69-74) */
    int32_t image_height; /* H */
    int32_t x0, y0;       /* region [x0,x1) x [y0,y1) to render                        */
    int32_t x1, y1;
    int32_t spp;          /* Renderer::set_samples (renderer.h:104)                    */
    int32_t max_depth;    /* Integrator::set_max_depth, reference uses 50 (main.cpp:102) */
    int32_t rr_start_depth; /* 3 (mis_path_integrator.h:237)                           */
    int32_t integrator;   /* RTR_INTEGRATOR_*                                          */
    uint32_t seed;        /* render seed; per-sample xorshift32 state = rtr_sample_seed() */
    int32_t pipeline;     /* RTR_PIPELINE_*                                            */
    /* tile sharding (renderer.h:40-62): the image is cut in 16x16 tiles, numbered in
     * the reference's dispatch order; this call renders the tiles with
     * index % tile_stride == tile_first that intersect the region.  tile_stride <= 1
     * renders every tile. */
    int32_t tile_first;
    int32_t tile_stride;
    /* The spp samples of a pixel may be summed as `spp_chunks` consecutive partial sums that
     * are then added in order (more parallelism on small images).  1 = one running sum in
     * sample order, exactly like renderer.h:72-79; 0 = let the library choose.  The library's choice depends on
     * the GPU, on the scene's kernel variant and on how many tiles the call owns, so with 0 the same image rendered
     * as ONE call, or tile-sharded over N contexts, or on another device agrees within 1e-13 relative (another
     * summation order) but not bit for bit; with 1, or any explicit count, every pixel is the same bits under every
     * sharding (tests: test_sharded_render_equals_unsharded). */
    int32_t spp_chunks;
    int32_t flags; /* RTR_FLAG_* */
} rtr_render_params;

/* Visit the hittable graph in the reference's own order (bvh_node left-then-right, lists in
 * order) instead of the compiled scene.  Always on for scenes with constant_medium, whose RNG
 * draws depend on that order; elsewhere both give the same image and this flag is a
 * cross-check. */
#define RTR_FLAG_REFERENCE_ORDER 1
/* Wavefront pipeline only: run the two ray-casting stages as PERSISTENT THREADS on the resumable traversal machine
 * (every lane takes the next ray of its wave's blocks as soon as its own is finished) instead of lockstep waves.
 * Same image bit for bit; measured slower on MI355X for every BASELINE scene, kept selectable. */
#define RTR_FLAG_WF_PERSISTENT 2
/* Megakernel only: once per bounce the 256 lanes of a workgroup exchange their hits through LDS so that every wave
 * shades hits of ONE material type (the reference's virtual scatter() call, material.h:31-60, is what diverges).
 * Exists for the MIS integrator on flat scenes lit by quad lights only (scene 23's kernel); ignored elsewhere.
 * Same image bit for bit; measured slower on MI355X (4.06 against 6.62 Gsamples/s on scene 23), kept
 * selectable. */
#define RTR_FLAG_SORTED_SHADING 4

typedef struct rtr_render_stats {
    uint64_t samples;          /* camera samples finished                               */
    uint64_t closest_segments; /* closest-hit traversals (hot loop 3)       */
    uint64_t shadow_segments;  /* shadow-ray traversals                                 */
    double device_ms;          /* HIP-event time of the render kernels of the last call */
    int32_t kernel_launches;
    int32_t pipeline;          /* pipeline that actually ran                            */
    int32_t spp_chunks;        /* partial sums per pixel that were used (params.spp_chunks, or the library's choice for 0) */
    int32_t cancelled;         /* 1: rtr_cancel() stopped this render before its last sample */
    int32_t flags_in_effect;   /* the RTR_FLAG_* bits of params.flags that selected another kernel than 0 would have */
    int32_t reserved;
} rtr_render_stats;

typedef struct rtr_context rtr_context;

/* ------------------------------------------------------------------------- */
/* entry points                                                              */

/* ABI version of the loaded library. */
uint32_t rtr_abi_version(void);

/* Number of HIP devices visible, or a negative status. */
int rtr_device_count(void);

/* Create / destroy a context bound to one GPU.  Replaces the construction of
 * `Renderer` (renderer.h:22-24).  One context is driven by one host thread. */
int rtr_create(int device_ordinal, rtr_context** out_ctx);
void rtr_destroy(rtr_context* ctx);

/* Launch the render kernels on an existing HIP stream (hipStream_t as void*;
 * NULL = the context's own stream).  Lets a host framework time the kernels
 * with its own events. */
int rtr_set_stream(rtr_context* ctx, void* hip_stream);

/* Validate + upload an immutable flattened scene.  Replaces the `world`, `cam`,
 * `background`, `lights` arguments of Renderer::render (renderer.h:30-32).
 * Every record of every array is checked, reachable from `root` or not: indices and
 * types in range, no cycle, image texels and environment tables inside image_bytes,
 * perlin permutation entries in 0..255, and finite parameters for the primitives and
 * transforms the graph reaches (RTR_ERR_INVALID / RTR_ERR_UNSUPPORTED with a message
 * in rtr_last_error; nothing is uploaded then).  These checks are what keeps the host
 * and the kernels inside the arrays they were given: a scene file is untrusted input. */
int rtr_upload_scene(rtr_context* ctx, const rtr_scene_desc* scene);

/* Render the region into a DEVICE buffer of doubles, 3 per pixel:
 * d_rgb[((j - y0) * row_stride + (i - x0)) * 3 + c] = linear mean radiance
 * (sum over samples * (1/spp)); pixels of tiles this call does not own are
 * left untouched.  Asynchronous on the context stream unless `blocking`; further
 * non-blocking calls queue behind it (the host waits only if the new call needs
 * another tile list or larger buffers than the one still running). */
int rtr_render_device(rtr_context* ctx, const rtr_render_params* params,
                      double* d_rgb, int64_t row_stride, int blocking);

/* Same, into a HOST buffer (blocking; includes the D2H copy). */
int rtr_render_host(rtr_context* ctx, const rtr_render_params* params,
                    double* h_rgb, int64_t row_stride);

/* Like rtr_render_host, but only what the call OWNS crosses PCIe: the tiles with index % tile_stride == tile_first that
 * intersect the region, packed.  On return *n_tiles tiles were rendered; tile k is the reference's tile (*tile_ids)[k]
 * (dispatch numbering, renderer.h:61-62) and occupies (*tiles)[k * 768 ...]: 16 rows of 16 pixels of 3 doubles, lowest
 * row first, linear mean radiance (pixels of a border tile outside the image or region are undefined);
 * (*tile_done)[k] = 1 when the tile was finished, 0 when a cancel came first (the reference's workers leave such a
 * tile untouched, renderer.h:52-59).  The three arrays live in pinned host memory owned by the context and stay valid
 * until its next render call.  Blocking.  This is what one worker of an N-GPU renderer calls: nothing is uploaded,
 * one D2H copy of n_tiles x 6 KiB. */
int rtr_render_tiles_host(rtr_context* ctx, const rtr_render_params* params, const double** tiles, const int32_t** tile_ids,
                          const uint8_t** tile_done, int64_t* n_tiles);

/* The number of partial sums per pixel the library would use for `params` (params->spp_chunks, or its own
 * choice for 0: depends on the scene's kernel variant, the pipeline and the number of owned tiles).  Returns
 * the count (>= 1) or a negative status.  Lets a caller render a crop with the summation of a full-size render. */
int rtr_plan_chunks(rtr_context* ctx, const rtr_render_params* params);

/* Per-ray entry: `Integrator::Li(r, world, background, lights)` (renderer/integrator.h:12-19) for `n` camera
 * samples of the uploaded scene, one GPU lane each.  Sample k is (pixel i = ijs[3k], j = ijs[3k+1], sample index
 * s = ijs[3k+2]) of the image params describe: its ray is camera::get_ray under rtr_sample_seed(seed, W, i, j, s)
 * exactly as in a render, and L[3k..3k+2] receives its radiance (not divided by spp).  params->integrator,
 * max_depth, rr_start_depth and flags apply; region, tiles, chunks and pipeline do not.  Host arrays; blocking. */
int rtr_li_samples(rtr_context* ctx, const rtr_render_params* params, const int32_t* ijs, double* L, int64_t n);

/* The same for ARBITRARY rays (Integrator::Li takes any `ray`: renderer/integrator.h:12-19): ray k starts at
 * origin with direction (not normalised by the library, like the reference's), time, and the xorshift32 state
 * (core/rtweekend.h:24-34; must not be 0) the reference's thread-local generator would hold when Li is entered --
 * for a camera ray that is the state after camera::get_ray.  L[3k..3k+2] receives the radiance. */
typedef struct rtr_li_ray {
    double origin[3], direction[3], time;
    uint32_t rng_state, pad;
} rtr_li_ray;
int rtr_li_rays(rtr_context* ctx, const rtr_render_params* params, const rtr_li_ray* rays, double* L, int64_t n);

/* Wait for everything queued on the context stream. */
int rtr_synchronize(rtr_context* ctx);

/* Thread-safe cooperative cancel (Renderer::cancel, renderer.h:113-115).  Covers every render
 * issued on the context so far, running or still queued behind another one; a render issued
 * afterwards is not affected.  A covered render stops at its next poll (every 8th sample of a
 * pixel / every wavefront batch) and reports RTR_ERR_CANCELLED: from the blocking call itself,
 * otherwise as rtr_render_stats.cancelled.  Output buffer after a cancel: like the reference's
 * workers (renderer.h:52-59), tiles whose samples all finished hold their final values and every
 * other tile keeps what the caller's buffer held before the call (megakernel: per 16x16 tile;
 * wavefront: no tile is written).  A cancel that arrives after the last sample has no effect. */
int rtr_cancel(rtr_context* ctx);

/* Statistics of the LAST render call (blocks until it has finished).  The
 * statistics of earlier queued calls that were never asked for are dropped. */
int rtr_get_stats(rtr_context* ctx, rtr_render_stats* out);

/* Text of the last error on this context ("" if none).  ctx may be NULL for
 * errors of rtr_create. */
const char* rtr_last_error(const rtr_context* ctx);

/* Per-sample RNG seed shared by the oracle and the device: the
 * xorshift32 state (core/rtweekend.h:24-34) used for sample `s` of pixel
 * (i, j) under render seed `seed`; never 0. */
uint32_t rtr_sample_seed(uint32_t seed, int32_t image_width, int32_t i, int32_t j, int32_t s);

/* Facts rtr_upload_scene() derives from a scene, and the host-only check it runs before touching the GPU. */
typedef struct rtr_scene_info {
    int32_t stack_words; /* LDS traversal-stack words per lane the reference-order traversal needs */
    int32_t has_media;   /* constant_medium present: RNG is consumed inside traversal */
    int32_t needs_uv;    /* some texture reads (u,v) */
    int32_t graph_depth; /* longest root-to-leaf chain of hittables */
    int32_t fast_ok;     /* a compiled scene exists (no media): the order-free traversal is the default */
    int32_t fast_instances, fast_refs, fast_stack_words;
    int32_t compiled_subtrees; /* media scenes: media-free subtrees compiled inside the reference-order walk */
    int32_t program_steps;     /* media scenes: steps of the ray-cast program (0: media not directly under the root list) */
    int32_t inverted_boxes;    /* spheres with a negative radius (hollow glass): sphere::bounding_box (sphere.h:62-66)
                                  then has min > max, the bvh_node boxes built from it do not enclose the sphere,
                                  and which rays still reach it depends on the reference's visiting order */
    int32_t top_trees;         /* compiled sub-scenes whose many transformed instances sit in a box tree of their own
                                  (walked per lane by the megakernel) instead of being scanned one after the other */
} rtr_scene_info;

/* Host-only: the checks rtr_upload_scene() runs before touching the GPU.  Returns RTR_OK,
 * RTR_ERR_INVALID or RTR_ERR_UNSUPPORTED; `msg` (may be NULL) receives the reason. */
int rtr_validate_scene(const rtr_scene_desc* scene, rtr_scene_info* info, char* msg, size_t msg_cap);


#ifdef __cplusplus
}
#endif
#endif /* RTR_HIP_H */
