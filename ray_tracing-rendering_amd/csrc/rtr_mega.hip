/*
 * rtr_mega.hip -- megakernel instantiations of one integrator group (see rt_launch.h); compiled three
 * times with -DRTR_MEGA_GROUP=0/1/2 so the variants build in parallel.
 */
#include "rt_kernels.h"
#include "rt_launch.h"

#ifndef RTR_MEGA_GROUP
#error "compile with -DRTR_MEGA_GROUP=0|1|2"
#endif

namespace {

int mega_fail(std::string& err, int code, const std::string& m) {
    err = m;
    return code;
}

template <typename K>
int launch_one(K kernel, const MegaLaunch& L, std::string& err) {
    /* the per-lane traversal stack lives in LDS: a graph that needs more than the CU has (e.g. the
     * reference-order walk of a hittable_list with thousands of direct children) cannot run that way */
    hipFuncAttributes fa{};
    hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel));
    /* the kernel's own static LDS (a few words of the workgroup vote) counts against the same 160 KiB */
    if (e == hipSuccess && L.lds + fa.sharedSizeBytes > 160 * 1024)
        return mega_fail(err, RTR_ERR_UNSUPPORTED, "this traversal of the scene needs a deeper stack than 160 KiB of LDS holds");
    if (e == hipSuccess && L.lds > 64 * 1024)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds);
    if (e == hipSuccess && L.dry && L.blocks_per_cu)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(L.blocks_per_cu, kernel, RTR_BLOCK, L.lds);
    if (e == hipSuccess && !L.dry) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)(L.P.n_tiles * L.P.chunks)), dim3(RTR_BLOCK), L.lds, L.stream, L.dsc, L.P,
                           L.stack_words);
        e = hipGetLastError();
    }
    if (e != hipSuccess) return mega_fail(err, RTR_ERR_DEVICE, std::string("megakernel launch: ") + hipGetErrorString(e));
    return RTR_OK;
}

#define RTR_LAUNCH(I, T, M) return launch_one(k_mega<I, T, M>, L, err)

/* integrators 1 and 4: every traversal, material-set variants.  FULLQ = the variant for "every material,
 * QuadLights only" (the RR integrator has no light code) */
template <int I, int FULLQ>
int launch_t(const MegaLaunch& L, std::string& err) {
    const int trav = L.trav;
    if (trav == RT_TRAV_FLAT) {
        if (L.lean) RTR_LAUNCH(I, RT_TRAV_FLAT, RT_MS_LEAN);
        if (L.quadlit) {
            if (mega_sortable(I, RT_TRAV_FLAT, FULLQ) && L.sorted)
                return launch_one(k_mega<I, RT_TRAV_FLAT, FULLQ, mega_sortable(I, RT_TRAV_FLAT, FULLQ)>, L, err);
            RTR_LAUNCH(I, RT_TRAV_FLAT, FULLQ);
        }
        RTR_LAUNCH(I, RT_TRAV_FLAT, RT_MS_FULL);
    }
    if (trav == RT_TRAV_FLAT_GUARD) {
        if (L.quadlit) RTR_LAUNCH(I, RT_TRAV_FLAT_GUARD, FULLQ);
        RTR_LAUNCH(I, RT_TRAV_FLAT_GUARD, RT_MS_FULL);
    }
    if (trav == RT_TRAV_FAST) {
        if (L.lean) RTR_LAUNCH(I, RT_TRAV_FAST, RT_MS_LEAN);
        if (L.quadlit) RTR_LAUNCH(I, RT_TRAV_FAST, FULLQ);
        RTR_LAUNCH(I, RT_TRAV_FAST, RT_MS_FULL);
    }
    if (trav == RT_TRAV_TOP) {
        if (L.lean) RTR_LAUNCH(I, RT_TRAV_TOP, RT_MS_LEAN);
        if (L.quadlit) RTR_LAUNCH(I, RT_TRAV_TOP, FULLQ);
        RTR_LAUNCH(I, RT_TRAV_TOP, RT_MS_FULL);
    }
    if (trav == RT_TRAV_PROGRAM && L.program_ext) {
        if (L.quadlit) RTR_LAUNCH(I, RT_TRAV_PROGRAM_EXT, FULLQ);
        RTR_LAUNCH(I, RT_TRAV_PROGRAM_EXT, RT_MS_FULL);
    }
    if (trav == RT_TRAV_PROGRAM) {
        if (L.quadlit) RTR_LAUNCH(I, RT_TRAV_PROGRAM, FULLQ);
        RTR_LAUNCH(I, RT_TRAV_PROGRAM, RT_MS_FULL);
    }
    if (trav == RT_TRAV_MEDIA) RTR_LAUNCH(I, RT_TRAV_MEDIA, RT_MS_FULL);
    if (L.lean) RTR_LAUNCH(I, RT_TRAV_EXACT, RT_MS_LEAN);
    RTR_LAUNCH(I, RT_TRAV_EXACT, RT_MS_FULL);
}
/* integrators 0 / 2 / 3: generic material set; the media kernel also serves the
 * reference-order traversal of scenes without media */
template <int I>
int launch_n1(const MegaLaunch& L, std::string& err) {
    if (L.trav == RT_TRAV_FAST) RTR_LAUNCH(I, RT_TRAV_FAST, RT_MS_FULL);
    if (L.trav == RT_TRAV_TOP) RTR_LAUNCH(I, RT_TRAV_TOP, RT_MS_FULL);
    if (L.trav == RT_TRAV_PROGRAM) RTR_LAUNCH(I, RT_TRAV_PROGRAM_EXT, RT_MS_FULL); /* (one program kernel here: the general one) */
    RTR_LAUNCH(I, RT_TRAV_MEDIA, RT_MS_FULL);
}

} // namespace

#if RTR_MEGA_GROUP == 0
int rtr_mega_launch_mis(const MegaLaunch& L, std::string& err) { return launch_t<RTR_INTEGRATOR_MIS, RT_MS_QUADLIT>(L, err); }
#elif RTR_MEGA_GROUP == 1
int rtr_mega_launch_rr_path(const MegaLaunch& L, std::string& err) {
    if (L.integrator == RTR_INTEGRATOR_RR) return launch_t<RTR_INTEGRATOR_RR, RT_MS_FULL>(L, err);
    return launch_n1<RTR_INTEGRATOR_PATH>(L, err);
}
#else
int rtr_mega_launch_pbr_nee(const MegaLaunch& L, std::string& err) {
    if (L.integrator == RTR_INTEGRATOR_PBR) return launch_n1<RTR_INTEGRATOR_PBR>(L, err);
    return launch_n1<RTR_INTEGRATOR_NEE>(L, err);
}
#endif
