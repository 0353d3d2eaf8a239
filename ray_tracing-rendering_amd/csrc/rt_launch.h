/*
 * rt_launch.h -- host-side seams between the translation units of librtr_hip.so.  The kernels are
 * compiled in parallel: rtr_capi.hip (C ABI, scene upload, unit-test kernels, k_resolve), rtr_mega.hip
 * three times (one integrator group each: RTR_MEGA_GROUP 0 = MIS, 1 = RR + path, 2 = PBR + NEE) and
 * rtr_wavefront.hip (stage kernels + their host driver).
 */
#pragma once

#include "rt_render.h"

#include <atomic>
#include <string>

/* what rtr_render_device decided about one megakernel launch */
struct MegaLaunch {
    int integrator, trav;
    bool lean, quadlit; /* material / light set of the scene (rtr_upload_scene) */
    bool sorted;        /* RTR_FLAG_SORTED_SHADING and a sorted instantiation exists for this launch (rt_kernels.h) */
    bool program_ext;   /* RT_TRAV_PROGRAM: the program holds guarded steps or media under wrappers (RT_TRAV_PROGRAM_EXT kernels) */
    size_t lds;         /* traversal stack + parked path state, bytes per workgroup */
    int stack_words;
    hipStream_t stream;
    const DScene* dsc;
    RenderK P;
    bool dry;           /* only what can fail without touching the stream: LDS attribute, occupancy query */
    int* blocks_per_cu; /* dry: resident workgroups per CU of the variant that would run */
};
/* return an rtr_status; `err` receives the text of a failure */
int rtr_mega_launch_mis(const MegaLaunch& L, std::string& err);
int rtr_mega_launch_rr_path(const MegaLaunch& L, std::string& err);
int rtr_mega_launch_pbr_nee(const MegaLaunch& L, std::string& err);

void rtr_launch_resolve(const ResolveK& R, hipStream_t stream);

struct WavefrontPool {
    void* slab = nullptr;
    size_t slab_bytes = 0;
    uint32_t* h_live = nullptr; /* pinned + mapped: live blocks after the newest compaction */
    hipEvent_t ev[2] = {nullptr, nullptr};
    void release();
};
struct WavefrontPlan {
    bool has_lights, lean, quadlit, sort, media;
    bool machine; /* casting stages as persistent threads on the traversal machine instead of lockstep waves */
    int trav;     /* RT_TRAV_FLAT / RT_TRAV_FAST / RT_TRAV_PROGRAM */
    int n_cus;
    size_t lds; /* traversal stack of the extend / connect stages */
};
int wavefront_render(WavefrontPool& pool, const DScene* sc, const WavefrontPlan& plan, const RenderK& P, int integrator,
                     double* d_rgb, int64_t row_stride, unsigned char* tile_done, hipStream_t stream,
                     std::atomic<uint32_t>* cancelled_upto,
                     int* launches, std::string& err);
