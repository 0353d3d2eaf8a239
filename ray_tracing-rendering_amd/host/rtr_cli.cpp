/*
 * rtr_cli.cpp -- headless driver in the shape of the reference's main.cpp:49-153 without the
 * SDL window: scene id and integrator id as the first two arguments (main.cpp:54-59), plus the
 * overrides the BASELINE configurations need (--width --spp --seed --out).  It is the
 * reference-side usage of the host layer: select_scene -> camera -> Renderer::render -> file.
 *
 *   rtr_cli <scene 7|9|21|22|23> <integrator 0..4> [--width W] [--spp N] [--seed S] [--bands N] [--out img.ppm|img.png]
 *           [--devices 0,1,...|all] [--repeat N]   one context + host thread per listed GPU (an ordinal may repeat)
 */
#include "rtr_renderer.h"

#include <cstring>

int main(int argc, char** argv) {
    int scene_id = 21, integrator_id = 4, width = 0, spp = 0, bands = 0, repeat = 1;
    std::vector<int> devices{0};
    unsigned seed = 1;
    std::string out;
    int pos = 0;
    for (int k = 1; k < argc; ++k) {
        if (!std::strcmp(argv[k], "--width") && k + 1 < argc) width = std::atoi(argv[++k]);
        else if (!std::strcmp(argv[k], "--spp") && k + 1 < argc) spp = std::atoi(argv[++k]);
        else if (!std::strcmp(argv[k], "--seed") && k + 1 < argc) seed = (unsigned)std::strtoul(argv[++k], nullptr, 0);
        else if (!std::strcmp(argv[k], "--out") && k + 1 < argc) out = argv[++k];
        else if (!std::strcmp(argv[k], "--bands") && k + 1 < argc) bands = std::atoi(argv[++k]);
        else if (!std::strcmp(argv[k], "--repeat") && k + 1 < argc) repeat = std::atoi(argv[++k]);
        else if (!std::strcmp(argv[k], "--devices") && k + 1 < argc) {
            const std::string v = argv[++k];
            devices.clear();
            if (v == "all") devices = Renderer::all_devices();
            else
                for (size_t a = 0; a < v.size();) {
                    size_t b = v.find(',', a);
                    if (b == std::string::npos) b = v.size();
                    devices.push_back(std::atoi(v.substr(a, b - a).c_str()));
                    a = b + 1;
                }
        }
        else if (pos == 0) scene_id = std::atoi(argv[k]), ++pos;
        else if (pos == 1) integrator_id = std::atoi(argv[k]), ++pos;
    }
    rtr::rng_state() = 12345u; /* scene-construction seed */
    SceneConfig config;
    try {
        config = select_scene(scene_id);
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return 2;
    }
    if (width > 0) config.image_width = width;
    if (spp > 0) config.samples_per_pixel = spp;
    auto cam = make_shared<camera>(config.lookfrom, config.lookat, config.vup, config.vfov, config.aspect_ratio,
                                   config.aperture, config.focus_dist, 0.0, 1.0); /* main.cpp:63-66 */
    const int W = config.image_width, H = static_cast<int>(W / config.aspect_ratio);
    RenderBuffer buffer(W, H);
    Renderer renderer(devices);
    renderer.set_samples(config.samples_per_pixel);
    switch (integrator_id) { /* main.cpp:80-100 */
    case 0: renderer.set_integrator(make_shared<PathIntegrator>()); break;
    case 1: renderer.set_integrator(make_shared<RRPathInterator>()); break;
    case 2: renderer.set_integrator(make_shared<PBRPathIntegrator>()); break;
    case 3: renderer.set_integrator(make_shared<DirectLightIntegrator>()); break;
    default: renderer.set_integrator(make_shared<MISPathIntegrator>()); break;
    }
    renderer.set_max_depth(50); /* main.cpp:102 */
    renderer.set_seed(seed);
    renderer.set_progress_bands(bands);
    for (int r = 0; r < repeat; ++r) { /* a second call finds the flattened scene on the GPUs */
        renderer.render(config.world, cam, config.background, buffer, config.lights);
        if (renderer.last_status() != RTR_OK) return 1;
    }
    std::cout << "contexts: " << renderer.device_contexts() << "  scene uploads: " << renderer.scene_uploads() << "\n";
    std::cout << "Msamples/s: " << (double)W * H * config.samples_per_pixel / renderer.last_seconds() * 1e-6
              << " (includes flatten, upload and D2H)\n";
    const bool png = out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0; /* main.cpp:138-151 writes a PNG */
    if (!out.empty() && !(png ? buffer.save_to_png(out) : buffer.save_to_ppm(out))) {
        std::cerr << "Failed to save image to " << out << "\n";
        return 1;
    }
    return 0;
}
