/*
 * rtr_capi.hip -- implementation of the C ABI (include/rtr_hip.h) on top of the HIP
 * kernels of rt_kernels.h.  Host side only: scene validation and upload, launch geometry,
 * workspace, cancel, statistics.  Built by hipcc for gfx950 into librtr_hip.so.
 */
#define RTR_TU_CAPI
#include "rt_compile.h"
#include "rt_kernels.h"
#include "rt_launch.h"
#include "rt_machine.h"
#include "rt_debug.h"

#include <atomic>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

} // namespace

struct rtr_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr; /* cancel flag writes */
    std::string err;
    /* scene */
    bool has_scene = false;
    rtr_scene_info info{};
    DScene ds{};
    DevBuf b_nodes, b_kids, b_mats, b_tex, b_perlin, b_images, b_imgbytes, b_lights;
    DevBuf b_finst, b_fxf, b_fref, b_fexit, b_fbvh, b_dscene, b_fprim, b_fsub, b_fstep, b_fvisit, b_fscan, b_fleaf, b_fmat, b_fguard;
    int fast_stack_words = 1;
    int walk_extra_words = 0; /* stack of a compiled subtree's box tree on top of the walk's own */
    bool lean_materials = false; /* only lambertian / diffuse_light with solid_color textures, only QuadLights */
    bool quad_lights_only = false;
    bool flat_scene = false; /* compiled scene without box trees and without tie-capable references */
    /* a moving_sphere (its hit() writes no u,v: the record keeps those of an earlier, farther hit of the
     * reference's walk) carries a material that reads (u,v): only the reference-order walk reproduces that */
    bool uv_order_dependent = false;
    int n_material_types = 0;
    /* per-render workspace */
    DevBuf b_tiles, b_partial, b_done, b_stats, b_cancel, b_test, b_stage;
    std::vector<int> last_tiles; /* what b_tiles holds */
    WavefrontPool pool;
    void* h_stage = nullptr; /* pinned: rtr_render_tiles_host */
    size_t h_stage_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool stats_pending = false;
    bool in_flight = false; /* stream-ordered work of a render call is queued whose statistics are not pending (an error return) */
    rtr_render_stats stats{};
    /* Cancel.  Renders are numbered; rtr_cancel() covers every render issued so far: it stores the newest
     * id in `cancelled_upto` and in the device word the kernels poll.  A render issued afterwards carries a
     * larger id, so nothing has to be reset between renders and a cancel that arrives while a render waits
     * in the stream behind another one is not lost. */
    std::atomic<uint32_t> render_seq{0};
    std::atomic<uint32_t> cancelled_upto{0};
    uint32_t pending_id = 0; /* id of the render whose statistics are pending */
    std::mutex cancel_mu;
    int n_materials = 0;
    int n_cus = 256; /* hipDeviceProp.multiProcessorCount */
    bool flat_guarded = false; /* a flat scene but for guarded references: RT_TRAV_FAST everywhere, RT_TRAV_FLAT_GUARD in the megakernel */
    bool machine_ok = false; /* the compiled scene fits the position word of the traversal machine (rt_machine.h) */
    bool guarded_program = false; /* the step program holds guarded primitives (FStep kind 3) or media under wrappers: not a program of the machine */
};

namespace {

int fail(rtr_context* c, int code, const std::string& msg) {
    if (c)
        c->err = msg;
    else
        g_create_error = msg;
    return code;
}

#define HIPCHK(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(ctx, RTR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

int ensure(rtr_context* c, DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return RTR_OK;
    if (b.p) {
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) return fail(c, RTR_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    b.cap = bytes;
    return RTR_OK;
}

int upload(rtr_context* c, DevBuf& b, const void* src, size_t bytes) {
    int rc = ensure(c, b, bytes);
    if (rc) return rc;
    if (bytes) HIPCHK(c, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return RTR_OK;
}

/* ---- host-only scene validation + traversal stack analysis -------------------------------- */
struct Validator {
    const rtr_scene_desc* s;
    std::string msg;
    std::vector<int> state;  /* 0 new, 1 on DFS stack, 2 done */
    std::vector<int> need;   /* stack words used below a node (see traverse()) */
    std::vector<int> depth;
    std::vector<char> media; /* subtree contains a constant_medium */
    int code = RTR_OK;

    bool bad(int c, const std::string& m) {
        if (code == RTR_OK) {
            code = c;
            msg = m;
        }
        return false;
    }
    bool node_ix(int i) const { return i >= 0 && i < s->n_nodes; }

    bool texture_ok(int ix, int guard) {
        if (ix < 0 || ix >= s->n_textures) return bad(RTR_ERR_INVALID, "texture index out of range");
        if (guard > 7) return bad(RTR_ERR_UNSUPPORTED, "checker textures nested deeper than 8");
        const rtr_texture& t = s->textures[ix];
        switch (t.type) {
        case RTR_TEX_SOLID: return true;
        case RTR_TEX_CHECKER: return texture_ok(t.a, guard + 1) && texture_ok(t.b, guard + 1);
        case RTR_TEX_NOISE:
            if (t.a < 0 || t.a >= s->n_perlin) return bad(RTR_ERR_INVALID, "perlin table index out of range");
            return true;
        case RTR_TEX_IMAGE:
            if (t.a >= s->n_images) return bad(RTR_ERR_INVALID, "image index out of range");
            if (t.a >= 0) {
                const rtr_image& im = s->images[t.a];
                /* width * height * 3 of two positive int32 fits a uint64; the offset is compared first so
                 * the sum cannot wrap */
                if (im.width <= 0 || im.height <= 0 || im.offset > s->n_image_bytes ||
                    (uint64_t)im.width * (uint64_t)im.height * 3 > s->n_image_bytes - im.offset)
                    return bad(RTR_ERR_INVALID, "image texels out of range");
            }
            return true;
        default: return bad(RTR_ERR_UNSUPPORTED, "unknown texture type");
        }
    }

    bool material_ok(int ix) {
        if (ix < 0 || ix >= s->n_materials) return bad(RTR_ERR_INVALID, "material index out of range");
        const rtr_material& m = s->materials[ix];
        switch (m.type) {
        case RTR_MAT_LAMBERTIAN:
        case RTR_MAT_DIFFUSE_LIGHT:
        case RTR_MAT_ISOTROPIC: return texture_ok(m.tex[0], 0);
        case RTR_MAT_METAL:
        case RTR_MAT_DIELECTRIC: return true;
        case RTR_MAT_PBR:
            if (!texture_ok(m.tex[0], 0) || !texture_ok(m.tex[1], 0) || !texture_ok(m.tex[2], 0)) return false;
            return m.tex[3] < 0 || texture_ok(m.tex[3], 0);
        default: return bad(RTR_ERR_UNSUPPORTED, "unknown material type");
        }
    }

    /* children of a node by position (no per-visit copies: a hittable_list can hold a million of them) */
    int child_count(const rtr_node& n) const {
        switch (n.type) {
        case RTR_NODE_BVH: return 2;
        case RTR_NODE_LIST: return n.b;
        case RTR_NODE_TRANSLATE:
        case RTR_NODE_ROTATE_Y:
        case RTR_NODE_FLIP_FACE:
        case RTR_NODE_MEDIUM: return 1;
        default: return 0;
        }
    }
    int child_at(const rtr_node& n, int k) const {
        if (n.type == RTR_NODE_BVH) return k == 0 ? n.a : n.b;
        if (n.type == RTR_NODE_LIST) return s->list_children[n.a + k];
        return n.a;
    }

    /* iterative post-order DFS over the hittable DAG */
    bool walk(int root) {
        struct Frame {
            int node, next;
        };
        std::vector<Frame> stk;
        stk.push_back({root, 0});
        state[root] = 1;
        while (!stk.empty()) {
            Frame& f = stk.back();
            const rtr_node& n = s->nodes[f.node];
            if (f.next == 0) { /* first visit: check the record itself */
                int n_geom = 0; /* leading f[] entries the scene compiler builds boxes from: must be finite */
                switch (n.type) {
                case RTR_NODE_BVH: /* its box is only ever compared with a ray: any value is safe */
                case RTR_NODE_FLIP_FACE: break;
                case RTR_NODE_TRANSLATE: n_geom = 3; break;
                case RTR_NODE_ROTATE_Y: n_geom = 2; break;
                case RTR_NODE_LIST:
                    if (n.a < 0 || n.b < 0 || (int64_t)n.a + n.b > s->n_list_children)
                        return bad(RTR_ERR_INVALID, "hittable_list children out of range");
                    break;
                case RTR_NODE_MEDIUM:
                    if (!material_ok(n.b)) return false;
                    break;
                case RTR_NODE_SPHERE: n_geom = 4; break;
                case RTR_NODE_MOVING_SPHERE: n_geom = 9; break;
                case RTR_NODE_XY_RECT:
                case RTR_NODE_XZ_RECT:
                case RTR_NODE_YZ_RECT: n_geom = 5; break;
                default: return bad(RTR_ERR_UNSUPPORTED, "unknown hittable node type");
                }
                if (n.type >= RTR_NODE_SPHERE && !material_ok(n.a)) return false;
                for (int k = 0; k < n_geom; ++k)
                    if (!std::isfinite(n.f[k])) return bad(RTR_ERR_INVALID, "non-finite primitive or transform parameter");
            }
            const int m = child_count(n);
            if (f.next < m) {
                int k = child_at(n, f.next++);
                if (!node_ix(k)) return bad(RTR_ERR_INVALID, "hittable child index out of range");
                if (state[k] == 1) return bad(RTR_ERR_INVALID, "hittable graph has a cycle");
                if (state[k] == 0) {
                    state[k] = 1;
                    stk.push_back({k, 0});
                }
                continue;
            }
            /* children done: stack words, depth, media */
            const int me = f.node;
            int u = 0, d = 0;
            char md = 0;
            for (int k = 0; k < m; ++k) {
                const int ck = child_at(n, k);
                d = std::max(d, depth[ck]);
                md |= media[ck];
            }
            switch (n.type) {
            case RTR_NODE_BVH: u = std::max(2, std::max(1 + need[n.a], need[n.b])); break;
            case RTR_NODE_LIST:
                if (m > RT_LIST_BULK) { /* continuation (2 words) + the child being walked */
                    u = 3;
                    for (int k = 0; k < m; ++k) u = std::max(u, 2 + need[child_at(n, k)]);
                } else {
                    u = m;
                    for (int k = 0; k < m; ++k) u = std::max(u, (m - 1 - k) + need[child_at(n, k)]);
                }
                break;
            case RTR_NODE_TRANSLATE: u = RT_FRAME_TRANSLATE + std::max(1, need[n.a]); break;
            case RTR_NODE_ROTATE_Y: u = RT_FRAME_ROTATE + std::max(1, need[n.a]); break;
            case RTR_NODE_FLIP_FACE: u = RT_FRAME_FLIP + std::max(1, need[n.a]); break;
            case RTR_NODE_MEDIUM:
                if (md) return bad(RTR_ERR_UNSUPPORTED, "constant_medium nested inside a medium boundary");
                u = std::max(1, need[n.a]);
                md = 1;
                break;
            default: break;
            }
            need[me] = u;
            depth[me] = d + 1;
            media[me] = md;
            state[me] = 2;
            stk.pop_back();
        }
        return true;
    }

    int run(rtr_scene_info* info) {
        if (!s) return (bad(RTR_ERR_INVALID, "null scene"), code);
        if (s->abi_version != RTR_ABI_VERSION) return (bad(RTR_ERR_INVALID, "ABI version mismatch"), code);
        if (s->n_nodes >= RT_LIST_MARK) return (bad(RTR_ERR_INVALID, "more than 2^30 nodes"), code);
    if (s->n_nodes <= 0 || s->n_list_children < 0 || s->n_materials < 0 || s->n_textures < 0 ||
            s->n_perlin < 0 || s->n_images < 0 || s->n_lights < 0)
            return (bad(RTR_ERR_INVALID, "negative or empty counts"), code);
        if (!node_ix(s->root)) return (bad(RTR_ERR_INVALID, "root out of range"), code);
        if (!s->nodes || (s->n_list_children && !s->list_children) || (s->n_materials && !s->materials) ||
            (s->n_textures && !s->textures) || (s->n_perlin && !s->perlin) || (s->n_images && !s->images) ||
            (s->n_image_bytes && !s->image_bytes) || (s->n_lights && !s->lights))
            return (bad(RTR_ERR_INVALID, "null array with non-zero count"), code);
        for (int k = 0; k < s->n_lights; ++k) {
            const rtr_light& l = s->lights[k];
            if (l.type < 0 || l.type >= RTR_LIGHT_TYPE_COUNT)
                return (bad(RTR_ERR_UNSUPPORTED, "unknown light type (QuadLight, PointLight, SpotLight, DirectionalLight, EnvironmentLight are on the device)"), code);
            if (l.type == RTR_LIGHT_ENV_MAP) { /* texels and sampling tables live in image_bytes */
                const double w = l.f[0], h = l.f[1], to = l.f[3], tb = l.f[4];
                if (!(w >= 1 && h >= 1 && w <= 65536 && h <= 65536) || w != (double)(int)w || h != (double)(int)h)
                    return (bad(RTR_ERR_INVALID, "environment map size out of range"), code);
                const double nb = (double)s->n_image_bytes;
                const double texel_bytes = w * h * 3 * 4, table_bytes = (h * (2 * w + 2) + (2 * h + 2)) * 8;
                if (!(to >= 0 && tb >= 0) || to != (double)(uint64_t)to || tb != (double)(uint64_t)tb ||
                    (uint64_t)to % 4 || (uint64_t)tb % 8 || to + texel_bytes > nb || tb + table_bytes > nb)
                    return (bad(RTR_ERR_INVALID, "environment map texels / tables out of range or misaligned"), code);
            }
        }
        /* every record is checked, also those the graph under `root` does not reach: upload and the
         * scene-wide facts (material classes, texture classes) loop over whole arrays */
        for (int k = 0; k < s->n_textures; ++k)
            if (!texture_ok(k, 0)) return code;
        for (int k = 0; k < s->n_materials; ++k)
            if (!material_ok(k)) return code;
        /* materials/perlin.h:35 indexes ranvec[perm_x ^ perm_y ^ perm_z]: the device does the same, unchecked */
        for (int k = 0; k < s->n_perlin; ++k)
            for (int i = 0; i < 256; ++i)
                if ((unsigned)s->perlin[k].perm_x[i] > 255u || (unsigned)s->perlin[k].perm_y[i] > 255u ||
                    (unsigned)s->perlin[k].perm_z[i] > 255u)
                    return (bad(RTR_ERR_INVALID, "perlin permutation entry out of range"), code);
        for (int k = 0; k < s->n_nodes; ++k) {
            const rtr_node& n = s->nodes[k];
            if (n.type < 0 || n.type >= RTR_NODE_TYPE_COUNT)
                return (bad(RTR_ERR_UNSUPPORTED, "unknown hittable node type"), code);
            if (n.type >= RTR_NODE_SPHERE && (n.a < 0 || n.a >= s->n_materials))
                return (bad(RTR_ERR_INVALID, "material index out of range"), code);
            if (n.type == RTR_NODE_MEDIUM && (n.b < 0 || n.b >= s->n_materials))
                return (bad(RTR_ERR_INVALID, "material index out of range"), code);
        }
        state.assign(s->n_nodes, 0);
        need.assign(s->n_nodes, 0);
        depth.assign(s->n_nodes, 0);
        media.assign(s->n_nodes, 0);
        if (!walk(s->root)) return code;
        if (info) {
            info->stack_words = std::max(1, need[s->root]);
            info->has_media = media[s->root];
            int inverted = 0;
            for (int k = 0; k < s->n_nodes; ++k)
                if (state[k] == 2 && ((s->nodes[k].type == RTR_NODE_SPHERE && s->nodes[k].f[3] < 0) ||
                                      (s->nodes[k].type == RTR_NODE_MOVING_SPHERE && s->nodes[k].f[8] < 0)))
                    ++inverted;
            info->inverted_boxes = inverted;
            info->graph_depth = depth[s->root];
            int uv = 0;
            for (int k = 0; k < s->n_textures; ++k)
                if (s->textures[k].type == RTR_TEX_IMAGE && s->textures[k].a >= 0) uv = 1;
            info->needs_uv = uv;
        }
        return RTR_OK;
    }
};

int params_check(rtr_context* c, const rtr_render_params* p) {
    if (!p) return fail(c, RTR_ERR_INVALID, "null params");
    if (p->image_width < 2 || p->image_height < 2) return fail(c, RTR_ERR_INVALID, "image smaller than 2x2");
    /* tile indices are int32 and the tile list is materialised: 2^26 tiles = 131 072 x 131 072 pixels, more
     * than a framebuffer in 288 GB of HBM holds */
    if (((int64_t)p->image_width + 15) / 16 * (((int64_t)p->image_height + 15) / 16) > ((int64_t)1 << 26))
        return fail(c, RTR_ERR_INVALID, "image larger than 2^26 tiles");
    if (p->x0 < 0 || p->y0 < 0 || p->x1 > p->image_width || p->y1 > p->image_height || p->x0 >= p->x1 ||
        p->y0 >= p->y1)
        return fail(c, RTR_ERR_INVALID, "region outside the image or empty");
    if (p->spp < 1 || p->max_depth < 1 || p->rr_start_depth < 0)
        return fail(c, RTR_ERR_INVALID, "spp/max_depth/rr_start_depth out of range");
    if (p->integrator < RTR_INTEGRATOR_PATH || p->integrator > RTR_INTEGRATOR_MIS)
        return fail(c, RTR_ERR_UNSUPPORTED, "integrator id not supported (0 path, 1 RR, 2 PBR, 3 NEE, 4 MIS)");
    if (p->tile_stride > 1 && (p->tile_first < 0 || p->tile_first >= p->tile_stride))
        return fail(c, RTR_ERR_INVALID, "tile_first must be in [0, tile_stride)");
    if (p->spp_chunks < 0 || p->spp_chunks > p->spp) return fail(c, RTR_ERR_INVALID, "spp_chunks must be in [0, spp]");
    if (p->flags & ~(RTR_FLAG_REFERENCE_ORDER | RTR_FLAG_WF_PERSISTENT | RTR_FLAG_SORTED_SHADING)) return fail(c, RTR_ERR_INVALID, "unknown flag bits");
    if (p->pipeline < RTR_PIPELINE_AUTO || p->pipeline > RTR_PIPELINE_WAVEFRONT)
        return fail(c, RTR_ERR_INVALID, "unknown pipeline");
    return RTR_OK;
}

/* tiles this call owns, in the reference's dispatch order (renderer.h:40-62) */
std::vector<int> owned_tiles(const rtr_render_params& p, int& tiles_x, int& tiles_y) {
    tiles_x = (p.image_width + 15) / 16;
    tiles_y = (p.image_height + 15) / 16;
    const int stride = p.tile_stride > 1 ? p.tile_stride : 1;
    const int first = p.tile_stride > 1 ? p.tile_first : 0;
    std::vector<int> out;
    for (int t = first; t < tiles_x * tiles_y; t += stride) {
        int ty = (tiles_y - 1) - t / tiles_x, tx = t % tiles_x;
        int xs = tx * 16, ys = ty * 16;
        if (xs + 16 <= p.x0 || xs >= p.x1 || ys + 16 <= p.y0 || ys >= p.y1) continue;
        out.push_back(t);
    }
    return out;
}

/* which traversal a call uses: the compiled scene unless it does not exist or the caller asks
 * for the reference's visiting order */
int pick_trav(const rtr_context* c, int flags) {
    if (c->info.has_media || (c->info.inverted_boxes && !c->info.fast_ok)) /* (hollow spheres as guarded references: compiled) */
        return c->info.program_steps > 0 && !(flags & RTR_FLAG_REFERENCE_ORDER) ? RT_TRAV_PROGRAM
                                                                                                     : RT_TRAV_MEDIA;
    if (!c->info.fast_ok || c->uv_order_dependent || (flags & RTR_FLAG_REFERENCE_ORDER))
        return RT_TRAV_EXACT;
    return c->flat_scene ? RT_TRAV_FLAT : RT_TRAV_FAST;
}
size_t stack_bytes(const rtr_context* c, int trav) {
    const int words = trav == RT_TRAV_FAST || trav == RT_TRAV_FLAT || trav == RT_TRAV_PROGRAM || trav == RT_TRAV_TOP || trav == RT_TRAV_FLAT_GUARD ? c->fast_stack_words
                                                                      : c->info.stack_words + c->walk_extra_words;
    return (size_t)words * RTR_BLOCK * sizeof(int);
}

template <typename K>
int set_lds(rtr_context* c, K kernel, size_t bytes) {
    /* the per-lane traversal stack lives in LDS: a graph that needs more than the CU has (e.g. the
     * reference-order walk of a hittable_list with thousands of direct children) cannot run that way */
    if (bytes > 160 * 1024)
        return fail(c, RTR_ERR_UNSUPPORTED, "this traversal of the scene needs a deeper stack than 160 KiB of LDS holds");
    if (bytes > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return RTR_OK;
}

/* `dry`: only what can fail without touching the stream (the LDS size check / attribute, the occupancy query) */
int launch_mega(rtr_context* c, const RenderK& P, int integrator, int trav_in, bool dry, int* blocks_per_cu, int flags,
                int* flags_in_effect = nullptr) {
    MegaLaunch L{};
    /* the flat variants exist for integrators 1 and 4; the others take the general compiled-scene kernel */
    L.trav = trav_in == RT_TRAV_FLAT && integrator != RTR_INTEGRATOR_MIS && integrator != RTR_INTEGRATOR_RR ? RT_TRAV_FAST : trav_in;
    if (L.trav == RT_TRAV_FAST && c->ds.top_root0 >= 0) L.trav = RT_TRAV_TOP; /* many instances: the per-lane walk (FSub) */
    if (L.trav == RT_TRAV_FAST && c->flat_guarded && (integrator == RTR_INTEGRATOR_MIS || integrator == RTR_INTEGRATOR_RR))
        L.trav = RT_TRAV_FLAT_GUARD;
    L.integrator = integrator;
    L.stack_words = (int)(stack_bytes(c, L.trav) / (RTR_BLOCK * sizeof(int)));
    L.dsc = static_cast<const DScene*>(c->b_dscene.p);
    L.lean = c->lean_materials && L.trav != RT_TRAV_MEDIA && L.trav != RT_TRAV_PROGRAM;
    L.quadlit = c->quad_lights_only && !c->info.needs_uv;
    /* (the sorted variant packs the material index into 16 bits) */
    L.sorted = (flags & RTR_FLAG_SORTED_SHADING) && c->n_materials <= 65535 && mega_sortable(integrator, L.trav, L.lean ? RT_MS_LEAN : (L.quadlit ? RT_MS_QUADLIT : RT_MS_FULL));
    L.lds = stack_bytes(c, L.trav) + (size_t)(L.sorted ? SK_WORDS : park_words(integrator, L.trav)) * RTR_BLOCK * sizeof(double);
    L.program_ext = c->guarded_program;
    L.stream = c->stream;
    L.P = P;
    L.dry = dry;
    L.blocks_per_cu = blocks_per_cu;
    if (flags_in_effect && L.sorted) *flags_in_effect |= RTR_FLAG_SORTED_SHADING;
    switch (integrator) {
    case RTR_INTEGRATOR_MIS: return rtr_mega_launch_mis(L, c->err);
    case RTR_INTEGRATOR_RR:
    case RTR_INTEGRATOR_PATH: return rtr_mega_launch_rr_path(L, c->err);
    default: return rtr_mega_launch_pbr_nee(L, c->err);
    }
}

/* auto chunking.  A workgroup renders one tile for one chunk of the samples.  More chunks = more, shorter
 * workgroups: the resident slots drain more evenly at the end of the launch, but every workgroup pays its
 * start-up once.  Model fitted to sweeps on scenes 21 / 23 / 9 (4-8 chunks beat 1-2 by 3-10 %, 32 lose
 * 10 %): efficiency = R / (R + 0.75) * s / (s + 2) with R = rounds over the resident slots and s = samples
 * per pixel and chunk; the best power of two is taken.  It picks 8 for C2 on one GPU and 16 for the 313
 * tiles one of 8 ranks owns. */
int auto_chunks(int pipeline, double resident_slots, int n_tiles, int spp) {
    int chunks = 1;
    if (pipeline == RTR_PIPELINE_WAVEFRONT) { /* one pool slot per pixel and chunk: keep the pool small */
        while ((long long)n_tiles * chunks < 8192 && chunks * 2 * 8 <= spp && chunks < 64) chunks *= 2;
        return chunks; /* about 2 M slots (0.5 GB of path state) where the image allows it */
    }
    double best = 0;
    for (int cand = 1; cand <= 64 && cand <= spp; cand *= 2) {
        const double rounds = (double)n_tiles * cand / resident_slots, s_per = (double)spp / cand;
        const double eff = rounds / (rounds + 0.75) * s_per / (s_per + 2.0);
        if (eff > best) best = eff, chunks = cand;
    }
    return chunks;
}


/* spp_chunks = 0: the library's choice for this scene, pipeline and number of owned tiles */
int choose_chunks(rtr_context* c, RenderK P, int integrator, int pipeline, int trav, int spp, int flags, int* chunks, int* guided) {
    /* workgroups of this kernel variant the chip holds at once (registers / LDS decide: 2-5 per CU) */
    int per_cu = 4;
    P.chunks = 1;
    if (pipeline == RTR_PIPELINE_MEGAKERNEL)
        if (int rc = launch_mega(c, P, integrator, trav, true, &per_cu, flags)) return rc;
    const double resident = (double)c->n_cus * (per_cu > 0 ? per_cu : 1);
    *chunks = auto_chunks(pipeline, resident, P.n_tiles, spp);
    /* Guided chunks.  With equal chunks a launch ends with part of the chip waiting for the last full-size
     * workgroups -- a tenth of the time of a rank's eighth of C2 (313 tiles: 4.9 rounds over the resident slots).
     * Instead three quarters of the chunks carry 90 % of a pixel's samples and run first; the rest is cut into thirds
     * of that size and drains the launch.  Measured on one GPU, scene 21 800x800 spp 400, share of 1 / 2 / 4 / 8
     * ranks: 75.2 / 38.6 / 20.3 / 10.9 ms with equal chunks, 74.5 / 37.9 / 19.8 / 10.2 ms guided (100 / 98 / 94 / 91 %
     * of the ideal share).  Not worth it once the launch has tens of rounds anyway. */
    guided[0] = guided[1] = guided[2] = 0;
    const double rounds = (double)P.n_tiles * *chunks / resident;
    if (pipeline == RTR_PIPELINE_MEGAKERNEL && *chunks >= 4 && rounds < 30.0 && spp >= 8 * *chunks) {
        /* (RTR_GUIDED = "big-share,small-divisor,big-chunk-eighths" overrides the split for tools/shard_sweep.py) */
        double share = 0.90;
        int div = 3, eighths = 6;
        if (const char* g = getenv("RTR_GUIDED")) std::sscanf(g, "%lf,%d,%d", &share, &div, &eighths);
        const int n_big = std::max(1, *chunks * eighths / 8);
        const int big = (int)((share * spp + n_big - 1) / n_big);
        const int rem = spp - n_big * big;
        const int small = std::max(1, big / std::max(1, div));
        if (rem > 0) {
            const int n_small = (rem + small - 1) / small;
            guided[0] = n_big, guided[1] = big, guided[2] = small;
            *chunks = n_big + n_small;
        }
    }
    return RTR_OK;
}

int finish_stats(rtr_context* c) {
    if (!c->stats_pending) {
        if (c->in_flight) { /* a call that failed after its first stream-ordered step: wait for what it queued */
            c->in_flight = false;
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        return RTR_OK;
    }
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    unsigned long long h[RT_STATS_WORDS] = {0};
    HIPCHK(c, hipMemcpy(h, c->b_stats.p, sizeof h, hipMemcpyDeviceToHost));
    if (getenv("RTR_REGION_PROFILE")) { /* a -DRTR_REGION_PROFILE build filled these (rt_device.h: RT_REGION) */
        static const char* names[RG_N] = {"other / loop", "closest: instance setup", "closest: rect runs", "closest: sphere runs",
                                          "closest: generic scan", "closest: tree inner nodes", "closest: tree leaves",
                                          "closest: hit record (fast_finish)", "shadow: instance setup", "shadow: rect runs",
                                          "shadow: sphere runs", "shadow: generic scan", "shadow: tree inner nodes",
                                          "shadow: tree leaves", "media steps", "mat_prepare", "shade_a (emission, light sample)",
                                          "shade_b (BSDF sample, roulette)", "miss", "end of sample + regeneration",
                                          "shade_rr / shade_path", "park path state", "sorted shading: barrier waits",
                                          "sorted shading: tickets + exchange"};
        double total = 0;
        for (int k = 0; k < RG_N; ++k) total += (double)h[RT_PROF_BASE + k];
        std::fprintf(stderr, "[region profile] %.4g wave cycles in all, %llu samples\n", total, h[0]);
        for (int k = 0; k < RG_N; ++k)
            if (h[RT_PROF_BASE + RT_PROF_REGIONS + k])
                std::fprintf(stderr, "[region profile] %-36s %6.2f %%  %12llu visits  %8.1f cycles/visit\n", names[k],
                             100.0 * (double)h[RT_PROF_BASE + k] / total, h[RT_PROF_BASE + RT_PROF_REGIONS + k],
                             (double)h[RT_PROF_BASE + k] / (double)h[RT_PROF_BASE + RT_PROF_REGIONS + k]);
    }
#ifdef RTR_PHASE_CLOCKS
    std::fprintf(stderr, "[phase clocks] closest %.3e  shade %.3e  shadow %.3e  other %.3e (wave cycles)\n", (double)h[3],
                 (double)h[4], (double)h[5], (double)h[6]);
#endif
    c->stats.samples = h[0];
    c->stats.closest_segments = h[1];
    c->stats.shadow_segments = h[2];
    c->stats.device_ms = ms;
    if (h[7]) c->stats.cancelled = 1; /* workgroups that saw the cancel before their last sample */
    c->stats_pending = false;
    c->in_flight = false;
    return RTR_OK;
}

} // namespace

void rtr_launch_resolve(const ResolveK& R, hipStream_t stream) {
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)R.r.n_tiles), dim3(RTR_BLOCK), 0, stream, R);
}

extern "C" {

uint32_t rtr_abi_version(void) { return RTR_ABI_VERSION; }

int rtr_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return RTR_ERR_DEVICE;
    return n;
}

uint32_t rtr_sample_seed(uint32_t seed, int32_t image_width, int32_t i, int32_t j, int32_t s) {
    return rtr_sample_seed_inline(seed, image_width, i, j, s);
}

int rtr_validate_scene(const rtr_scene_desc* scene, rtr_scene_info* info, char* msg, size_t msg_cap) {
    Validator v;
    v.s = scene;
    int rc = v.run(info);
    if (msg && msg_cap) {
        std::snprintf(msg, msg_cap, "%s", v.msg.c_str());
    }
    if (rc == RTR_OK && info) {
        CompiledScene cs = compile_scene(scene, info->has_media != 0 || info->inverted_boxes != 0, info->has_media == 0 && info->inverted_boxes != 0);
        info->fast_ok = cs.ok;
        info->fast_instances = (int32_t)cs.inst.size();
        info->fast_refs = (int32_t)cs.ref.size();
        info->fast_stack_words = cs.stack_words;
        info->compiled_subtrees = cs.n_compiled_subtrees;
        info->program_steps = (int32_t)cs.steps.size();
        info->top_trees = 0;
        for (const FSub& sub : cs.subs) info->top_trees += sub.top_root >= 0;
    }
    return rc;
}

int rtr_create(int device_ordinal, rtr_context** out_ctx) {
    if (!out_ctx) return fail(nullptr, RTR_ERR_INVALID, "null out_ctx");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, RTR_ERR_DEVICE,
                    std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count is 0"));
    if (device_ordinal < 0 || device_ordinal >= n) return fail(nullptr, RTR_ERR_INVALID, "device ordinal out of range");
    rtr_context* c = new rtr_context();
    c->device = device_ordinal;
#define CREATE_CHK(expr)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            g_create_error = std::string(#expr) + ": " + hipGetErrorString(e_);              \
            delete c;                                                                        \
            return RTR_ERR_DEVICE;                                                           \
        }                                                                                    \
    } while (0)
    CREATE_CHK(hipSetDevice(device_ordinal));
    {
        hipDeviceProp_t prop;
        CREATE_CHK(hipGetDeviceProperties(&prop, device_ordinal));
        c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    CREATE_CHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    CREATE_CHK(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    CREATE_CHK(hipEventCreate(&c->ev0));
    CREATE_CHK(hipEventCreate(&c->ev1));
#undef CREATE_CHK
    c->stream = c->own_stream;
    int rc = ensure(c, c->b_stats, RT_STATS_WORDS * sizeof(unsigned long long));
    if (!rc) rc = ensure(c, c->b_cancel, sizeof(uint32_t));
    if (!rc && hipMemset(c->b_cancel.p, 0, sizeof(uint32_t)) != hipSuccess) rc = RTR_ERR_DEVICE;
    if (rc) {
        g_create_error = c->err;
        rtr_destroy(c);
        return rc;
    }
    *out_ctx = c;
    return RTR_OK;
}

void rtr_destroy(rtr_context* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->b_nodes, &c->b_kids,  &c->b_mats,    &c->b_tex,   &c->b_perlin, &c->b_images, &c->b_imgbytes,
                      &c->b_lights, &c->b_tiles, &c->b_partial, &c->b_done, &c->b_stats, &c->b_cancel, &c->b_test, &c->b_stage,
                      &c->b_finst, &c->b_fxf, &c->b_fref, &c->b_fexit, &c->b_fbvh, &c->b_dscene, &c->b_fprim, &c->b_fsub, &c->b_fstep, &c->b_fvisit, &c->b_fscan, &c->b_fleaf, &c->b_fmat, &c->b_fguard};
    for (DevBuf* b : bufs)
        if (b->p) hipFree(b->p);
    c->pool.release();
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    if (c->side_stream) hipStreamDestroy(c->side_stream);
    delete c;
}

int rtr_set_stream(rtr_context* c, void* hip_stream) {
    if (!c) return RTR_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    return RTR_OK;
}

int rtr_upload_scene(rtr_context* c, const rtr_scene_desc* s) {
    if (!c) return RTR_ERR_INVALID;
    Validator v;
    v.s = s;
    rtr_scene_info info{};
    int rc = v.run(&info);
    if (rc) return fail(c, rc, "scene rejected: " + v.msg);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->has_scene = false;
    CompiledScene cs = compile_scene(s, info.has_media != 0 || info.inverted_boxes != 0, info.has_media == 0 && info.inverted_boxes != 0);
    if ((rc = upload(c, c->b_nodes, cs.dev_nodes.data(), sizeof(rtr_node) * cs.dev_nodes.size()))) return rc;
    if ((rc = upload(c, c->b_kids, s->list_children, sizeof(int32_t) * s->n_list_children))) return rc;
    if ((rc = upload(c, c->b_mats, s->materials, sizeof(rtr_material) * s->n_materials))) return rc;
    if ((rc = upload(c, c->b_tex, s->textures, sizeof(rtr_texture) * s->n_textures))) return rc;
    if ((rc = upload(c, c->b_perlin, s->perlin, sizeof(rtr_perlin) * s->n_perlin))) return rc;
    if ((rc = upload(c, c->b_images, s->images, sizeof(rtr_image) * s->n_images))) return rc;
    if ((rc = upload(c, c->b_imgbytes, s->image_bytes, s->n_image_bytes))) return rc;
    if ((rc = upload(c, c->b_lights, s->lights, sizeof(rtr_light) * s->n_lights))) return rc;
    info.fast_ok = cs.ok;
    info.fast_instances = (int32_t)cs.inst.size();
    info.fast_refs = (int32_t)cs.ref.size();
    info.fast_stack_words = cs.stack_words;
    info.compiled_subtrees = cs.n_compiled_subtrees;
    if ((rc = upload(c, c->b_fxf, cs.xf.data(), sizeof(FXf) * cs.xf.size()))) return rc;
    if ((rc = upload(c, c->b_fref, cs.ref.data(), sizeof(FRef) * cs.ref.size()))) return rc;
    if ((rc = upload(c, c->b_fexit, cs.exits.data(), sizeof(int32_t) * cs.exits.size()))) return rc;
    if ((rc = upload(c, c->b_fbvh, cs.bvh.data(), sizeof(FBvh) * cs.bvh.size()))) return rc;
    if ((rc = upload(c, c->b_fsub, cs.subs.data(), sizeof(FSub) * cs.subs.size()))) return rc;
    info.program_steps = (int32_t)cs.steps.size();
    info.top_trees = 0;
    for (const FSub& sub : cs.subs) info.top_trees += sub.top_root >= 0;
    /* the traversal machine of the wavefront stages always runs a step program: a scene without media is
     * the one-step program "sub-scene 0" */
    std::vector<FStep> dev_steps = cs.steps;
    if (cs.ok && dev_steps.empty()) {
        FStep whole{};
        whole.kind = 0, whole.sub = 0;
        dev_steps.push_back(whole);
    }
    if ((rc = upload(c, c->b_fstep, dev_steps.data(), sizeof(FStep) * dev_steps.size()))) return rc;
    if ((rc = upload(c, c->b_fguard, cs.guards.data(), sizeof(FGuard) * cs.guards.size()))) return rc;
    /* ... flattened into instance visits in execution order */
    std::vector<FVisit> visits;
    for (size_t k = 0; k < dev_steps.size(); ++k) {
        const FStep& st = dev_steps[k];
        const FSub& sub = cs.subs[st.sub];
        const int first = (int)visits.size();
        for (int q = 0; q < sub.n_inst; ++q) {
            const FInst& I = cs.inst[sub.inst_first + q];
            FVisit v{};
            v.flags = (q == 0 ? FV_FIRST : 0) | (q == sub.n_inst - 1 ? FV_LAST : 0) | (st.kind != 0 ? FV_MEDIUM : 0) |
                      (sub.n_inst > RT_FAST_NO_BOX_MAX ? FV_BOXES : 0) | ((int)k >= cs.step_tail ? FV_TAIL : 0);
            v.step = (int32_t)k, v.inst = sub.inst_first + q, v.step_first = first;
            v.xf_first = I.xf_first, v.n_xf = I.n_xf, v.ref_first = I.ref_first, v.n_ref = I.n_ref;
            v.bvh_root = I.bvh_root, v.bound = I.bound;
            v.neg_inv_density = st.neg_inv_density;
            visits.push_back(v);
        }
    }
    if ((rc = upload(c, c->b_fvisit, visits.data(), sizeof(FVisit) * visits.size()))) return rc;
    c->machine_ok = !visits.empty();
    c->guarded_program = false;
    for (const FStep& st : dev_steps) c->guarded_program |= st.kind == 3 || st.n_xf > 0 || st.n_exit > 0;
    bool any_tie = false;
    {
        std::vector<rtr_node> prims(cs.ref.size());
        for (size_t k = 0; k < cs.ref.size(); ++k) {
            prims[k] = s->nodes[cs.ref[k].node]; /* original records */
            prims[k].reserved = cs.ref[k].pad;      /* visiting order of the reference's walk */
            /* the wrappers above the reference as a code in f[9] (see RT_EXIT_LONG) */
            unsigned long long code = 0;
            bool fits = prims[k].type != RTR_NODE_MOVING_SPHERE && cs.ref[k].n_exit <= 31;
            for (int e = 0; e < cs.ref[k].n_exit && fits; ++e) {
                const int wt = s->nodes[cs.exits[cs.ref[k].exit_first + e]].type;
                code |= (unsigned long long)(wt == RTR_NODE_FLIP_FACE ? 2 : 1) << (2 * e);
            }
            if (!fits) code = RT_EXIT_LONG;
            if (prims[k].type != RTR_NODE_MOVING_SPHERE) std::memcpy(&prims[k].f[9], &code, 8);
            const auto guard = cs.guard_of_ref.find((int)k);
            if (guard != cs.guard_of_ref.end()) { /* RT_GUARD_FLAG: first guard and count in a sphere's free words */
                const long long first = guard->second.first, count = guard->second.second;
                std::memcpy(&prims[k].f[4], &first, 8), std::memcpy(&prims[k].f[5], &count, 8);
                prims[k].reserved |= RT_GUARD_FLAG;
            }
        }
        /* references that can tie exactly in t with another one of their instance (see RT_TIE_FLAG) */
        for (const FInst& I : cs.inst) {
            std::map<std::vector<uint64_t>, std::vector<int>> groups; /* same plane / same sphere */
            auto bits = [](double v) {
                uint64_t u;
                std::memcpy(&u, &v, 8);
                return u;
            };
            for (int r = I.ref_first; r < I.ref_first + I.n_ref; ++r) {
                const rtr_node& n = prims[r];
                if (n.type >= RTR_NODE_XY_RECT)
                    groups[{(uint64_t)n.type, bits(n.f[4])}].push_back(r);
                else if (n.type == RTR_NODE_SPHERE)
                    groups[{(uint64_t)n.type, bits(n.f[0]), bits(n.f[1]), bits(n.f[2]), bits(std::fabs(n.f[3]))}].push_back(r);
            }
            for (const auto& g : groups) {
                const std::vector<int>& v = g.second;
                if (v.size() > 512) { /* a huge coplanar set (tiled floor): flag all rather than test every pair */
                    for (int r : v) prims[r].reserved |= RT_TIE_FLAG;
                    any_tie = true;
                    continue;
                }
                for (size_t x = 0; x < v.size(); ++x)
                    for (size_t y = x + 1; y < v.size(); ++y) {
                        rtr_node &p = prims[v[x]], &q = prims[v[y]];
                        const bool overlap = p.type == RTR_NODE_SPHERE ||
                                             (std::max(p.f[0], q.f[0]) <= std::min(p.f[1], q.f[1]) &&
                                              std::max(p.f[2], q.f[2]) <= std::min(p.f[3], q.f[3]));
                        if (overlap) p.reserved |= RT_TIE_FLAG, q.reserved |= RT_TIE_FLAG, any_tie = true;
                    }
            }
        }
        /* Ties ACROSS instances of a sub-scene.  Instances are scanned in the order their first primitive is
         * visited and every test accepts t == t_max, so of two instances the later one wins a tie -- which is the
         * reference's choice (it keeps what it visits later) unless the EARLIER instance holds the later-visited
         * primitive (all primitives under the same transform chain share an instance: [wall, box, floor] puts the
         * floor into the first instance, in front of the box whose bottom face lies in its plane).  Exactly those
         * pairs -- rects whose planes coincide in world space, visiting order against instance order -- get the
         * tie flag; their visiting positions then decide.  A y-plane keeps its orientation under every chain
         * (translate, rotate_y), x- and z-planes under translations only; rotated side faces of different
         * chains are not looked at. */
        for (const FSub& sub : cs.subs) {
            struct PlaneRef {
                double k;
                int axis, inst, ref, visit;
            };
            std::vector<PlaneRef> planes;
            for (int ii = sub.inst_first; ii < sub.inst_first + sub.n_inst; ++ii) {
                const FInst& I = cs.inst[ii];
                double off[3] = {0, 0, 0};
                bool rotated = false;
                for (int k = 0; k < I.n_xf; ++k) {
                    const FXf& x = cs.xf[I.xf_first + k];
                    if (x.type == RTR_NODE_TRANSLATE)
                        off[0] += x.f[0], off[1] += x.f[1], off[2] += x.f[2];
                    else
                        rotated = true;
                }
                for (int r = I.ref_first; r < I.ref_first + I.n_ref; ++r) {
                    const rtr_node& n = prims[r];
                    if (n.type < RTR_NODE_XY_RECT) continue;
                    const int axis = n.type == RTR_NODE_XY_RECT ? 2 : (n.type == RTR_NODE_XZ_RECT ? 1 : 0);
                    if (rotated && axis != 1) continue;
                    planes.push_back({n.f[4] + off[axis], axis, ii, r, n.reserved & ~RT_TIE_FLAG});
                }
            }
            std::sort(planes.begin(), planes.end(), [](const PlaneRef& a, const PlaneRef& b) {
                return a.axis != b.axis ? a.axis < b.axis : a.k < b.k;
            });
            for (size_t lo = 0; lo < planes.size();) { /* clusters of (nearly) the same world plane */
                size_t hi = lo + 1;
                while (hi < planes.size() && planes[hi].axis == planes[lo].axis &&
                       planes[hi].k - planes[hi - 1].k <= 1e-9 * std::max(1.0, std::fabs(planes[hi].k)))
                    ++hi;
                if (hi - lo > 1) {
                    std::vector<PlaneRef> cl(planes.begin() + lo, planes.begin() + hi);
                    std::sort(cl.begin(), cl.end(), [](const PlaneRef& a, const PlaneRef& b) { return a.inst < b.inst; });
                    /* flag P (earlier instance) and Q (later instance) whenever visit(P) > visit(Q) */
                    std::vector<int> max_before(cl.size()), min_after(cl.size());
                    int mx = -1;
                    for (size_t i = 0, j = 0; i < cl.size(); i = j) { /* per instance block */
                        for (j = i; j < cl.size() && cl[j].inst == cl[i].inst; ++j) max_before[j] = mx;
                        for (size_t q = i; q < j; ++q) mx = std::max(mx, cl[q].visit);
                    }
                    int mn = INT32_MAX;
                    for (size_t j = cl.size(), i; j > 0; j = i) {
                        for (i = j; i > 0 && cl[i - 1].inst == cl[j - 1].inst; --i) min_after[i - 1] = mn;
                        for (size_t q = i; q < j; ++q) mn = std::min(mn, cl[q].visit);
                    }
                    /* (a sub-scene with a top tree meets its instances in any order: every such pair then) */
                    const bool any_order = sub.top_root >= 0;
                    for (size_t q = 0; q < cl.size(); ++q)
                        if (max_before[q] > cl[q].visit || min_after[q] < cl[q].visit ||
                            (any_order && (max_before[q] >= 0 || min_after[q] < INT32_MAX)))
                            prims[cl[q].ref].reserved |= RT_TIE_FLAG, any_tie = true;
                }
                lo = hi;
            }
        }
        if ((rc = upload(c, c->b_fprim, prims.data(), sizeof(rtr_node) * prims.size()))) return rc;
        {
            const std::vector<FLeaf> leaves = rtc::build_leaf_records(cs, prims);
            if ((rc = upload(c, c->b_fleaf, leaves.data(), sizeof(FLeaf) * leaves.size()))) return rc;
        }
        rtc::build_scan_runs(cs, prims);
        if ((rc = upload(c, c->b_fscan, cs.scan.data(), sizeof(double) * cs.scan.size()))) return rc;
        if ((rc = upload(c, c->b_finst, cs.inst.data(), sizeof(FInst) * cs.inst.size()))) return rc;
    }
    c->fast_stack_words = cs.stack_words;
    /* (guarded references -- hollow spheres -- are tested by the generic loop of the kernels that know about ties: the
     * flat kernels carry neither) */
    c->flat_scene = cs.ok && cs.bvh.empty() && !any_tie && cs.guard_of_ref.empty();
    c->flat_guarded = cs.ok && cs.bvh.empty() && !any_tie && !cs.guard_of_ref.empty(); /* the megakernel's RT_TRAV_FLAT_GUARD */
    c->walk_extra_words = cs.n_compiled_subtrees ? cs.stack_words : 0;
    DScene& d = c->ds;
    d.finst = static_cast<const FInst*>(c->b_finst.p);
    d.fxf = static_cast<const FXf*>(c->b_fxf.p);
    d.fref = static_cast<const FRef*>(c->b_fref.p);
    d.fprim = static_cast<const rtr_node*>(c->b_fprim.p);
    d.fscan = static_cast<const double*>(c->b_fscan.p);
    d.fleaf = static_cast<const FLeaf*>(c->b_fleaf.p);
    d.fexit = static_cast<const int32_t*>(c->b_fexit.p);
    d.fbvh = static_cast<const FBvh*>(c->b_fbvh.p);
    d.fsub = static_cast<const FSub*>(c->b_fsub.p);
    d.n_finst = cs.ok ? cs.subs[0].n_inst : 0;
    d.top_root0 = cs.ok ? cs.subs[0].top_root : -1;
    d.world_inst0 = cs.ok ? cs.subs[0].world_inst : -1;
    d.world_linear0 = cs.ok ? cs.subs[0].world_linear : 0;
    d.top_bound0 = cs.ok ? cs.subs[0].top_bound : 0.0f;
    d.fstep = static_cast<const FStep*>(c->b_fstep.p);
    d.n_fstep = (int32_t)dev_steps.size();
    d.fstep_tail = cs.step_tail;
    d.fguard = static_cast<const FGuard*>(c->b_fguard.p);
    d.fvisit = static_cast<const FVisit*>(c->b_fvisit.p);
    d.n_fvisit = (int32_t)visits.size();
    d.nodes = static_cast<const rtr_node*>(c->b_nodes.p);
    d.list_children = static_cast<const int32_t*>(c->b_kids.p);
    d.materials = static_cast<const rtr_material*>(c->b_mats.p);
    d.textures = static_cast<const rtr_texture*>(c->b_tex.p);
    d.perlin = static_cast<const rtr_perlin*>(c->b_perlin.p);
    d.images = static_cast<const rtr_image*>(c->b_images.p);
    d.image_bytes = static_cast<const uint8_t*>(c->b_imgbytes.p);
    d.lights = static_cast<const rtr_light*>(c->b_lights.p);
    d.camera = s->camera;
    for (int k = 0; k < 3; ++k) d.background[k] = s->background[k];
    d.root = s->root;
    d.n_nodes = s->n_nodes;
    d.n_lights = s->n_lights;
    d.needs_uv = info.needs_uv;
    /* div_shared's range argument: numerators are differences of scene coordinates and ray origins */
    d.shared_div = 1;
    for (int k = 0; k < s->n_nodes && d.shared_div; ++k) {
        const rtr_node& n = s->nodes[k];
        const int nf = n.type == RTR_NODE_TRANSLATE ? 3 : n.type == RTR_NODE_SPHERE ? 4 : n.type == RTR_NODE_MOVING_SPHERE ? 9
                       : n.type >= RTR_NODE_XY_RECT ? 5 : 0;
        for (int q = 0; q < nf; ++q)
            if (!(std::fabs(n.f[q]) <= 0x1p60)) d.shared_div = 0;
        /* moving_sphere::center(time) scales (c1 - c0) by (time - t0) / (t1 - t0) */
        if (n.type == RTR_NODE_MOVING_SPHERE && !(std::fabs(n.f[7] - n.f[6]) >= 0x1p-20)) d.shared_div = 0;
    }
    if (!(std::fabs(s->camera.time0) <= 0x1p60 && std::fabs(s->camera.time1) <= 0x1p60)) d.shared_div = 0;
    for (const FInst& I : cs.inst)
        if (I.n_xf > 30) d.shared_div = 0;
    if (getenv("RTR_NO_SHARED_DIV")) d.shared_div = 0; /* experiments: the plain divisions */
    c->info = info;
    c->n_materials = s->n_materials;
    c->lean_materials = true;
    unsigned type_mask = 0;
    for (int k = 0; k < s->n_materials; ++k) type_mask |= 1u << s->materials[k].type;
    c->n_material_types = __builtin_popcount(type_mask);
    c->quad_lights_only = true;
    for (int k = 0; k < s->n_lights; ++k)
        if (s->lights[k].type != RTR_LIGHT_QUAD) c->quad_lights_only = false;
    if (!c->quad_lights_only) c->lean_materials = false; /* the lean kernels know QuadLights only */
    for (int k = 0; k < s->n_materials; ++k) {
        const rtr_material& m = s->materials[k];
        if (m.type != RTR_MAT_LAMBERTIAN && m.type != RTR_MAT_DIFFUSE_LIGHT) c->lean_materials = false;
        else if (s->textures[m.tex[0]].type != RTR_TEX_SOLID) c->lean_materials = false;
    }
    c->uv_order_dependent = false;
    if (info.needs_uv) {
        std::function<bool(int, int)> tex_reads_uv = [&](int t, int guard) {
            if (t < 0 || guard > 8) return false;
            const rtr_texture& x = s->textures[t];
            if (x.type == RTR_TEX_IMAGE) return x.a >= 0;
            if (x.type == RTR_TEX_CHECKER) return tex_reads_uv(x.a, guard + 1) || tex_reads_uv(x.b, guard + 1);
            return false;
        };
        for (int k = 0; k < s->n_nodes; ++k) {
            if (s->nodes[k].type != RTR_NODE_MOVING_SPHERE) continue;
            const rtr_material& m = s->materials[s->nodes[k].a];
            const int n_tex = m.type == RTR_MAT_PBR ? 4 : (m.type == RTR_MAT_METAL || m.type == RTR_MAT_DIELECTRIC ? 0 : 1);
            for (int q = 0; q < n_tex; ++q)
                if (tex_reads_uv(m.tex[q], 0)) c->uv_order_dependent = true;
        }
    }
    { /* FMat: materials with their solid textures' values inline */
        std::vector<FMat> fm((size_t)s->n_materials);
        auto solid = [&](int t) { return t >= 0 && s->textures[t].type == RTR_TEX_SOLID; };
        auto clampd = [](double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }; /* rtweekend.h:40-46 */
        for (int k = 0; k < s->n_materials; ++k) {
            const rtr_material& m = s->materials[k];
            FMat f{};
            f.type = m.type;
            for (int q = 0; q < 4; ++q) f.tex[q] = m.tex[q], f.f[q] = m.f[q];
            switch (m.type) {
            case RTR_MAT_LAMBERTIAN:
            case RTR_MAT_DIFFUSE_LIGHT:
            case RTR_MAT_ISOTROPIC: f.solid = solid(m.tex[0]); break;
            case RTR_MAT_PBR: f.solid = solid(m.tex[0]) && solid(m.tex[1]) && solid(m.tex[2]) && m.tex[3] < 0; break;
            default: f.solid = 1; /* metal, dielectric: no texture */
            }
            if (f.solid && m.type != RTR_MAT_METAL && m.type != RTR_MAT_DIELECTRIC) {
                for (int q = 0; q < 3; ++q) f.albedo[q] = s->textures[m.tex[0]].f[q];
                if (m.type == RTR_MAT_PBR) {
                    f.rough = clampd(s->textures[m.tex[1]].f[0], 0.01, 1.0);
                    f.metal = s->textures[m.tex[2]].f[0];
                }
            }
            fm[(size_t)k] = f;
        }
        if ((rc = upload(c, c->b_fmat, fm.data(), sizeof(FMat) * fm.size()))) return rc;
        d.fmat = static_cast<const FMat*>(c->b_fmat.p);
    }
    if ((rc = upload(c, c->b_dscene, &c->ds, sizeof(DScene)))) return rc;
    c->has_scene = true;
    return RTR_OK;
}

/* the render call proper; tile_done != nullptr: packed output (see ResolveK) */
static int render_core(rtr_context* c, const rtr_render_params* p, double* d_rgb, int64_t row_stride, unsigned char* tile_done,
                       int blocking);

int rtr_render_device(rtr_context* c, const rtr_render_params* p, double* d_rgb, int64_t row_stride, int blocking) {
    if (!c) return RTR_ERR_INVALID;
    if (!c->has_scene) return fail(c, RTR_ERR_NO_SCENE, "rtr_render before rtr_upload_scene");
    if (int prc = params_check(c, p)) return prc;
    if (!d_rgb || row_stride < (int64_t)(p->x1 - p->x0)) return fail(c, RTR_ERR_INVALID, "bad output buffer / stride");
    return render_core(c, p, d_rgb, row_stride, nullptr, blocking);
}

static int render_core(rtr_context* c, const rtr_render_params* p, double* d_rgb, int64_t row_stride, unsigned char* tile_done,
                       int blocking) {
    HIPCHK(c, hipSetDevice(c->device));
    (void)hipGetLastError(); /* a launch error of an earlier call (ours or the host framework's) is not this call's */
    int rc = RTR_OK;

    RenderK P{};
    P.W = p->image_width, P.H = p->image_height;
    P.x0 = p->x0, P.y0 = p->y0, P.x1 = p->x1, P.y1 = p->y1;
    P.spp = p->spp, P.max_depth = p->max_depth, P.rr_start = p->rr_start_depth;
    P.seed = p->seed;
    P.integrator = p->integrator;
    std::vector<int> tiles = owned_tiles(*p, P.tiles_x, P.tiles_y);
    P.n_tiles = (int)tiles.size();
    if (P.n_tiles == 0) { /* nothing to do: this call's statistics are all zero (an earlier render's are dropped) */
        if ((rc = finish_stats(c))) return rc;
        c->stats = rtr_render_stats{};
        return RTR_OK;
    }

    int pipeline = p->pipeline;
    if (pipeline == RTR_PIPELINE_AUTO) pipeline = RTR_PIPELINE_MEGAKERNEL;
    const int trav = pick_trav(c, p->flags);
    if (pipeline == RTR_PIPELINE_WAVEFRONT && (!c->machine_ok || (trav != RT_TRAV_FLAT && trav != RT_TRAV_FAST && trav != RT_TRAV_PROGRAM)))
        return fail(c, RTR_ERR_UNSUPPORTED, "the wavefront pipeline runs the compiled traversals only: this graph (or "
                                            "RTR_FLAG_REFERENCE_ORDER) needs the reference-order walk of the megakernel");
    if (pipeline == RTR_PIPELINE_WAVEFRONT && (p->flags & RTR_FLAG_WF_PERSISTENT) && c->guarded_program)
        return fail(c, RTR_ERR_UNSUPPORTED, "RTR_FLAG_WF_PERSISTENT: the traversal machine does not run step programs with "
                                            "guarded primitives (hollow spheres under bvh_nodes) or media under transforms; "
                                            "the lockstep stages do");
    int chunks = p->spp_chunks, guided[3] = {0, 0, 0};
    if (chunks == 0 && (rc = choose_chunks(c, P, p->integrator, pipeline, trav, p->spp, p->flags, &chunks, guided))) return rc;
    P.n_big = guided[0], P.big_spp = guided[1], P.small_spp = guided[2];
    P.chunks = chunks;

    /* A render that was queued without blocking may still be running.  Everything below is ordered behind
     * it on the stream, so the host only has to wait where it would touch memory that render still reads:
     * another tile list, larger workspace buffers, the wavefront pool.  Back-to-back renders of the same
     * shape (bench.py's steps) then queue up without a bubble between them; the statistics of a render nobody
     * asked for are dropped. */
    const size_t partial_bytes = (size_t)P.n_tiles * chunks * 3 * RTR_BLOCK * sizeof(double);
    const size_t done_bytes = (size_t)P.n_tiles * chunks * sizeof(int);
    const bool same_tiles = tiles == c->last_tiles;
    if ((c->stats_pending || c->in_flight) && (!same_tiles || partial_bytes > c->b_partial.cap || done_bytes > c->b_done.cap ||
                                               pipeline == RTR_PIPELINE_WAVEFRONT)) {
        if ((rc = finish_stats(c))) return rc;
    }
    /* everything that can fail comes before the first stream-ordered side effect, so an error return leaves
     * a render that is still in flight (and its pending statistics) alone */
    if (!same_tiles) {
        c->last_tiles.clear();
        if ((rc = upload(c, c->b_tiles, tiles.data(), tiles.size() * sizeof(int)))) return rc;
        c->last_tiles = tiles;
    }
    if ((rc = ensure(c, c->b_partial, partial_bytes))) return rc;
    if ((rc = ensure(c, c->b_done, done_bytes))) return rc;
    P.tile_ids = static_cast<const int*>(c->b_tiles.p);
    P.stats = static_cast<unsigned long long*>(c->b_stats.p);
    P.cancel = static_cast<const uint32_t*>(c->b_cancel.p);
    P.partial = static_cast<double*>(c->b_partial.p);
    P.done = static_cast<int*>(c->b_done.p);
    if (pipeline == RTR_PIPELINE_MEGAKERNEL && (rc = launch_mega(c, P, p->integrator, trav, true, nullptr, p->flags))) return rc;

    const uint32_t id = c->render_seq.fetch_add(1) + 1; /* rtr_cancel() from now on covers this render */
    P.render_id = id;
    c->stats_pending = false; /* an unfinished earlier render's statistics are dropped here */
    c->in_flight = true;      /* ... but what it queued, and what this call queues from here on, is still tracked: an error
                                 return below leaves it set, and the next call (or rtr_get_stats) waits before it touches
                                 the tile list, the workspace or the wavefront pool */
    c->stats = rtr_render_stats{};
    c->stats.spp_chunks = chunks;
    c->pending_id = id;
    HIPCHK(c, hipMemsetAsync(c->b_stats.p, 0, RT_STATS_WORDS * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (pipeline == RTR_PIPELINE_WAVEFRONT) {
        int launches = 0;
        WavefrontPlan plan{};
        plan.has_lights = c->ds.n_lights > 0;
        plan.media = trav == RT_TRAV_PROGRAM;
        plan.lean = c->lean_materials && !plan.media;
        plan.quadlit = c->quad_lights_only && !c->info.needs_uv;
        plan.sort = !plan.lean && c->n_material_types > 1;
        plan.n_cus = c->n_cus;
        plan.lds = stack_bytes(c, trav);
        plan.trav = trav;
        plan.machine = (p->flags & RTR_FLAG_WF_PERSISTENT) != 0;
        if (plan.machine) c->stats.flags_in_effect |= RTR_FLAG_WF_PERSISTENT;
        rc = wavefront_render(c->pool, static_cast<const DScene*>(c->b_dscene.p), plan, P, p->integrator, d_rgb, row_stride,
                              tile_done, c->stream, &c->cancelled_upto, &launches, c->err);
        if (rc && rc != RTR_ERR_CANCELLED) return rc;
        if (rc == RTR_ERR_CANCELLED) c->stats.cancelled = 1;
        c->stats.kernel_launches = launches;
    } else {
        if ((rc = launch_mega(c, P, p->integrator, trav, false, nullptr, p->flags, &c->stats.flags_in_effect))) return rc;
        ResolveK R{P, d_rgb, (long long)row_stride, tile_done};
        rtr_launch_resolve(R, c->stream);
        HIPCHK(c, hipGetLastError());
        c->stats.kernel_launches = 2;
    }
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->stats.pipeline = pipeline;
    c->stats_pending = true;
    if (blocking || c->stats.cancelled) { /* the host-driven wavefront loop has already waited */
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if ((rc = finish_stats(c))) return rc;
        if (c->stats.cancelled) return fail(c, RTR_ERR_CANCELLED, "render cancelled");
    }
    return RTR_OK;
}

int rtr_render_host(rtr_context* c, const rtr_render_params* p, double* h_rgb, int64_t row_stride) {
    if (!c) return RTR_ERR_INVALID;
    if (int prc = params_check(c, p)) return prc;
    if (!h_rgb || row_stride < (int64_t)(p->x1 - p->x0)) return fail(c, RTR_ERR_INVALID, "bad output buffer / stride");
    HIPCHK(c, hipSetDevice(c->device));
    const int w = p->x1 - p->x0, h = p->y1 - p->y0;
    const size_t bytes = (size_t)w * h * 3 * sizeof(double);
    /* staging framebuffer kept across calls (a progressive host render calls once per band); a render
     * still queued on the stream may be writing it */
    if (bytes > c->b_stage.cap) HIPCHK(c, hipStreamSynchronize(c->stream));
    int rc = ensure(c, c->b_stage, bytes);
    if (rc) return rc;
    void* d = c->b_stage.p;
    /* pixels of tiles this call does not own (or does not finish: cancel) keep the caller's values */
    hipError_t ce = hipMemcpy2DAsync(d, (size_t)w * 3 * sizeof(double), h_rgb, (size_t)row_stride * 3 * sizeof(double),
                                     (size_t)w * 3 * sizeof(double), h, hipMemcpyHostToDevice, c->stream);
    rc = ce == hipSuccess ? rtr_render_device(c, p, static_cast<double*>(d), w, 1)
                          : fail(c, RTR_ERR_DEVICE, hipGetErrorString(ce));
    if (rc == RTR_OK || rc == RTR_ERR_CANCELLED) {
        hipError_t e2 = hipMemcpy2D(h_rgb, (size_t)row_stride * 3 * sizeof(double), d, (size_t)w * 3 * sizeof(double),
                                    (size_t)w * 3 * sizeof(double), h, hipMemcpyDeviceToHost);
        if (e2 != hipSuccess) rc = fail(c, RTR_ERR_DEVICE, hipGetErrorString(e2));
    }
    return rc;
}

int rtr_render_tiles_host(rtr_context* c, const rtr_render_params* p, const double** tiles, const int32_t** tile_ids,
                          const uint8_t** tile_done, int64_t* n_tiles) {
    if (!c) return RTR_ERR_INVALID;
    if (!c->has_scene) return fail(c, RTR_ERR_NO_SCENE, "rtr_render before rtr_upload_scene");
    if (int prc = params_check(c, p)) return prc;
    if (!tiles || !tile_ids || !tile_done || !n_tiles) return fail(c, RTR_ERR_INVALID, "null output pointer");
    HIPCHK(c, hipSetDevice(c->device));
    int tx, ty;
    const std::vector<int> owned = owned_tiles(*p, tx, ty);
    const size_t n = owned.size();
    /* [pixels: n x 768 doubles][done: n bytes, padded][ids: n int32] -- on the device and, pinned, on the host */
    const size_t px_bytes = n * 768 * sizeof(double), done_bytes = (n + 15) & ~(size_t)15, id_bytes = n * sizeof(int32_t);
    const size_t total = px_bytes + done_bytes + id_bytes + 16;
    if (total > c->b_stage.cap || total > c->h_stage_cap) HIPCHK(c, hipStreamSynchronize(c->stream));
    int rc = ensure(c, c->b_stage, total);
    if (rc) return rc;
    if (total > c->h_stage_cap) {
        if (c->h_stage) HIPCHK(c, hipHostFree(c->h_stage));
        c->h_stage = nullptr, c->h_stage_cap = 0;
        if (hipHostMalloc(&c->h_stage, total, hipHostMallocDefault) != hipSuccess)
            return fail(c, RTR_ERR_NOMEM, "hipHostMalloc of the tile staging buffer");
        c->h_stage_cap = total;
    }
    char* d = static_cast<char*>(c->b_stage.p);
    char* h = static_cast<char*>(c->h_stage);
    *n_tiles = (int64_t)n;
    *tiles = reinterpret_cast<const double*>(h);
    *tile_done = reinterpret_cast<const uint8_t*>(h + px_bytes);
    *tile_ids = reinterpret_cast<const int32_t*>(h + px_bytes + done_bytes);
    if (n == 0) return RTR_OK;
    std::memcpy(h + px_bytes + done_bytes, owned.data(), id_bytes);
    HIPCHK(c, hipMemsetAsync(d + px_bytes, 0, done_bytes, c->stream));
    rc = render_core(c, p, reinterpret_cast<double*>(d), -1, reinterpret_cast<unsigned char*>(d + px_bytes), 0);
    if (rc && rc != RTR_ERR_CANCELLED) return rc;
    /* one D2H of the owned tiles and their flags into pinned memory */
    HIPCHK(c, hipMemcpyAsync(h, d, px_bytes + done_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (int frc = finish_stats(c)) return frc;
    if (c->stats.cancelled) return fail(c, RTR_ERR_CANCELLED, "render cancelled");
    return RTR_OK;
}

int rtr_plan_chunks(rtr_context* c, const rtr_render_params* p) {
    if (!c) return RTR_ERR_INVALID;
    if (!c->has_scene) return fail(c, RTR_ERR_NO_SCENE, "rtr_plan_chunks before rtr_upload_scene");
    if (int prc = params_check(c, p)) return prc;
    if (p->spp_chunks > 0) return p->spp_chunks;
    HIPCHK(c, hipSetDevice(c->device));
    RenderK P{};
    rtr_render_params q = *p;
    P.n_tiles = (int)owned_tiles(q, P.tiles_x, P.tiles_y).size();
    if (P.n_tiles == 0) return 1;
    const int pipeline = p->pipeline == RTR_PIPELINE_AUTO ? RTR_PIPELINE_MEGAKERNEL : p->pipeline;
    int chunks = 1, guided[3];
    if (int rc = choose_chunks(c, P, p->integrator, pipeline, pick_trav(c, p->flags), p->spp, p->flags, &chunks, guided)) return rc;
    return chunks;
}

int rtr_synchronize(rtr_context* c) {
    if (!c) return RTR_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RTR_OK;
}

int rtr_cancel(rtr_context* c) {
    if (!c) return RTR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(c->cancel_mu);
    /* static: the copy may still read the word after this function has returned an error */
    static thread_local uint32_t upto;
    upto = c->render_seq.load();
    if (upto == 0 || upto == c->cancelled_upto.load()) return RTR_OK; /* nothing issued since the last cancel */
    c->cancelled_upto.store(upto);
    hipSetDevice(c->device);
    hipError_t e = hipMemcpyAsync(c->b_cancel.p, &upto, sizeof(uint32_t), hipMemcpyHostToDevice, c->side_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->side_stream);
    return e == hipSuccess ? RTR_OK : RTR_ERR_DEVICE;
}

int rtr_get_stats(rtr_context* c, rtr_render_stats* out) {
    if (!c || !out) return RTR_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = finish_stats(c);
    if (rc) return rc;
    *out = c->stats;
    return RTR_OK;
}

const char* rtr_last_error(const rtr_context* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

/* ---- per-ray entry: Integrator::Li of camera samples or caller-given rays ------------------------------------ */
static int li_run(rtr_context* c, const rtr_render_params* p, const int32_t* ijs, const rtr_li_ray* rays, LiOut* host_out,
                  int64_t n) {
    if (!c) return RTR_ERR_INVALID;
    if (!c->has_scene) return fail(c, RTR_ERR_NO_SCENE, "rtr_li_* before rtr_upload_scene");
    if (int prc = params_check(c, p)) return prc;
    if (n == 0) return RTR_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t in_bytes = rays ? (size_t)n * sizeof(rtr_li_ray) : (size_t)n * 3 * sizeof(int32_t);
    const size_t in_pad = (in_bytes + 15) & ~(size_t)15;
    int rc = ensure(c, c->b_test, in_pad + (size_t)n * sizeof(LiOut));
    if (rc) return rc;
    char* base = static_cast<char*>(c->b_test.p);
    HIPCHK(c, hipMemcpy(base, rays ? (const void*)rays : (const void*)ijs, in_bytes, hipMemcpyHostToDevice));
    RenderK P{};
    P.W = p->image_width, P.H = p->image_height;
    P.spp = p->spp, P.max_depth = p->max_depth, P.rr_start = p->rr_start_depth;
    P.seed = p->seed;
    const int32_t* d_ijs = rays ? nullptr : reinterpret_cast<const int32_t*>(base);
    const rtr_li_ray* d_rays = rays ? reinterpret_cast<const rtr_li_ray*>(base) : nullptr;
    LiOut* d_out = reinterpret_cast<LiOut*>(base + in_pad);
    int trav = pick_trav(c, p->flags);
    if (trav == RT_TRAV_FLAT) trav = RT_TRAV_FAST; /* same results; one per-ray kernel for both */
    const size_t lds = stack_bytes(c, trav);
    const dim3 grid((unsigned)((n + RTR_BLOCK - 1) / RTR_BLOCK));
#define RTR_LAUNCH(I, T)                                                                                            \
    do {                                                                                                            \
        if ((rc = set_lds(c, k_li<I, T>, lds))) return rc;                                                          \
        hipLaunchKernelGGL((k_li<I, T>), grid, dim3(RTR_BLOCK), lds, c->stream, c->ds, P, d_ijs, d_rays, d_out, (long long)n); \
    } while (0)
#define RTR_LAUNCH_T(I)                                        \
    do {                                                       \
        if (trav == RT_TRAV_FAST)                              \
            RTR_LAUNCH(I, RT_TRAV_FAST);                       \
        else if (trav == RT_TRAV_PROGRAM)                      \
            RTR_LAUNCH(I, RT_TRAV_PROGRAM_EXT);                \
        else if (trav == RT_TRAV_MEDIA)                        \
            RTR_LAUNCH(I, RT_TRAV_MEDIA);                      \
        else                                                   \
            RTR_LAUNCH(I, RT_TRAV_EXACT);                      \
    } while (0)
#define RTR_LAUNCH_N1(I)                                   \
    do {                                                   \
        if (trav == RT_TRAV_FAST)                          \
            RTR_LAUNCH(I, RT_TRAV_FAST);                   \
        else if (trav == RT_TRAV_PROGRAM)                  \
            RTR_LAUNCH(I, RT_TRAV_PROGRAM_EXT);            \
        else                                               \
            RTR_LAUNCH(I, RT_TRAV_MEDIA);                  \
    } while (0)
    switch (p->integrator) {
    case RTR_INTEGRATOR_MIS: RTR_LAUNCH_T(RTR_INTEGRATOR_MIS); break;
    case RTR_INTEGRATOR_RR: RTR_LAUNCH_T(RTR_INTEGRATOR_RR); break;
    case RTR_INTEGRATOR_PATH: RTR_LAUNCH_N1(RTR_INTEGRATOR_PATH); break;
    case RTR_INTEGRATOR_PBR: RTR_LAUNCH_N1(RTR_INTEGRATOR_PBR); break;
    default: RTR_LAUNCH_N1(RTR_INTEGRATOR_NEE); break;
    }
#undef RTR_LAUNCH_N1
#undef RTR_LAUNCH_T
#undef RTR_LAUNCH
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(host_out, d_out, (size_t)n * sizeof(LiOut), hipMemcpyDeviceToHost));
    return RTR_OK;
}

int rtr_li_samples(rtr_context* c, const rtr_render_params* p, const int32_t* ijs, double* L, int64_t n) {
    if (!c) return RTR_ERR_INVALID;
    if (n < 0 || (n > 0 && (!ijs || !L))) return fail(c, RTR_ERR_INVALID, "bad sample / radiance arrays");
    if (int prc = params_check(c, p)) return prc;
    for (int64_t k = 0; k < n; ++k)
        if (ijs[3 * k] < 0 || ijs[3 * k] >= p->image_width || ijs[3 * k + 1] < 0 || ijs[3 * k + 1] >= p->image_height ||
            ijs[3 * k + 2] < 0)
            return fail(c, RTR_ERR_INVALID, "sample outside the image");
    std::vector<LiOut> out((size_t)n);
    int rc = li_run(c, p, ijs, nullptr, out.data(), n);
    if (rc) return rc;
    for (int64_t k = 0; k < n; ++k)
        for (int q = 0; q < 3; ++q) L[3 * k + q] = out[(size_t)k].L[q];
    return RTR_OK;
}

int rtr_li_rays(rtr_context* c, const rtr_render_params* p, const rtr_li_ray* rays, double* L, int64_t n) {
    if (!c) return RTR_ERR_INVALID;
    if (n < 0 || (n > 0 && (!rays || !L))) return fail(c, RTR_ERR_INVALID, "bad ray / radiance arrays");
    for (int64_t k = 0; k < n; ++k)
        if (rays[k].rng_state == 0) return fail(c, RTR_ERR_INVALID, "a xorshift32 state must not be 0 (rtweekend.h:24-34)");
    std::vector<LiOut> out((size_t)n);
    int rc = li_run(c, p, nullptr, rays, out.data(), n);
    if (rc) return rc;
    for (int64_t k = 0; k < n; ++k)
        for (int q = 0; q < 3; ++q) L[3 * k + q] = out[(size_t)k].L[q];
    return RTR_OK;
}

/* ---- the seam librtr_hip_test.so reaches the context through (csrc/rt_debug.h; not part of include/) ------------- */
int rtr_debug_view_get(rtr_context* c, int flags, rtr_debug_view* v, size_t size) {
    if (!c || !v || size != sizeof(rtr_debug_view)) return RTR_ERR_INVALID;
    if (!c->has_scene) return fail(c, RTR_ERR_NO_SCENE, "rtr_test_* before rtr_upload_scene");
    v->ds = c->ds;
    v->stream = c->stream;
    v->device = c->device;
    v->n_cus = c->n_cus;
    v->n_materials = c->n_materials;
    int trav = pick_trav(c, flags);
    if (trav == RT_TRAV_FLAT) trav = RT_TRAV_FAST;
    if (trav == RT_TRAV_FAST && c->ds.top_root0 >= 0) trav = RT_TRAV_TOP; /* the unit kernels walk what the megakernel walks */
    v->trav = trav;
    v->stack_bytes = stack_bytes(c, trav);
    return RTR_OK;
}
int rtr_debug_li(rtr_context* c, const rtr_render_params* p, const int32_t* ijs, rtr_debug_li_out* out, int64_t n) {
    static_assert(sizeof(rtr_debug_li_out) == sizeof(LiOut), "one layout");
    if (n < 0 || (n > 0 && (!ijs || !out))) return RTR_ERR_INVALID;
    return li_run(c, p, ijs, nullptr, reinterpret_cast<LiOut*>(out), n);
}
void rtr_debug_set_error(rtr_context* c, const char* msg) {
    if (c) c->err = msg ? msg : "";
}

} /* extern "C" */
