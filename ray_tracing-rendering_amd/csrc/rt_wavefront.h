/*
 * rt_wavefront.h -- wavefront pipeline (SoA path pool in HBM; extend / shade / connect
 * stage kernels).  Placeholder until the stages land: the entry point reports
 * RTR_ERR_UNSUPPORTED so RTR_PIPELINE_WAVEFRONT fails loudly instead of silently
 * running something else.
 */
#pragma once

#include "rt_kernels.h"
#include "rtr_hip_test.h"

#include <atomic>
#include <string>

struct WavefrontPool {
    void release() {}
};

inline int wavefront_render(WavefrontPool&, const DScene&, const rtr_scene_info&, const RenderK&, int, double*,
                            int64_t, hipStream_t, std::atomic<int>*, int*, std::string& err) {
    err = "wavefront pipeline not built into this library";
    return RTR_ERR_UNSUPPORTED;
}
