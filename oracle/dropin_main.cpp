/*
 * dropin_main.cpp -- TEST INFRASTRUCTURE.  Proves the drop-in claim of the host layer: the
 * reference's scene builders (scene/scenes.cpp, UNMODIFIED, compiled from where it lies under
 * /root/reference) are built against ray_tracing-rendering_amd/host/compat/ -- the project's
 * own headers that carry the reference's header names -- and the scenes they produce are
 * flattened with rtr::flatten().  Output goes to oracle/_ref/ only.
 *
 *   dropin_scenes <scene_id> <scene_seed> <out.rtrs>     exit 0 ok, 3 = scene uses an
 *                                                        object without device counterpart
 */
#include "scenes.h"

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
    if (argc != 4) return 2;
    const int id = std::atoi(argv[1]);
    rtr::rng_state() = (uint32_t)std::strtoul(argv[2], nullptr, 0);
    SceneConfig c = select_scene(id); /* the reference's select_scene */
    camera cam(c.lookfrom, c.lookat, c.vup, c.vfov, c.aspect_ratio, c.aperture, c.focus_dist, 0.0, 1.0);
    rtr_scene_storage st;
    std::string why;
    if (!rtr::flatten(*c.world, c.lights, cam, c.background, st, why)) {
        std::printf("{\"scene\": %d, \"flattened\": false, \"why\": \"%s\"}\n", id, why.c_str());
        return 3;
    }
    if (!st.save(argv[3])) return 4;
    std::printf("{\"scene\": %d, \"flattened\": true, \"nodes\": %zu, \"materials\": %zu, \"lights\": %zu}\n", id,
                st.nodes.size(), st.materials.size(), st.lights.size());
    return 0;
}
