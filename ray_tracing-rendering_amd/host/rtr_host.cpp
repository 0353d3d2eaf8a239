/*
 * rtr_host.cpp -- librtr_host.so: the BASELINE scenes described through the host
 * scene-description API (rtr_scene_api.h), flattened for the device.
 *
 * These are this project's own descriptions of the five scenes BASELINE.json names (the
 * reference builds them in scene/scenes.cpp: cornell_box :159-187, cornell_box_nee :779-809,
 * final_scene :221-290, final_scene_nee :811-882, mis_comparison_scene :580-626, with the
 * camera / light settings of select_scene :1572-1604,1729-1781).  Scene content is data; the
 * ORDER in which objects are created is kept, because object creation consumes the scene RNG
 * (box heights, perlin tables, sphere cloud, one BVH axis draw per node) and the order of the
 * object list decides BVH topology: with the same scene seed the result flattens to the same
 * bytes as the reference's own graph (tests/test_host_scenes.py).
 */
#include "rtr_renderer.h"

#include <cstring>

namespace {

struct Built {
    SceneConfig cfg;
};

using Mat = shared_ptr<material>;
using Obj = shared_ptr<hittable>;

Mat diffuse(double r, double g, double b) { return make_shared<lambertian>(color(r, g, b)); }
Mat emitter(double v) { return make_shared<diffuse_light>(color(v, v, v)); }

Obj tilted_box(double sx, double sy, double sz, double degrees, const vec3& at, const Mat& m) {
    Obj b = make_shared<box>(point3(0, 0, 0), point3(sx, sy, sz), m);
    b = make_shared<rotate_y>(b, degrees);
    return make_shared<translate>(b, at);
}

/* Cornell box: 555-unit room, green +x wall, red -x wall, 130x105 ceiling lamp just below the
 * ceiling, two rotated boxes.  `one_sided_lamp` wraps the lamp in flip_face so it emits
 * downwards under the front-face-only emitted(rec, wo). */
Obj cornell_room(bool one_sided_lamp) {
    const Mat red = diffuse(.65, .05, .05), white = diffuse(.73, .73, .73), green = diffuse(.12, .45, .15);
    const Mat lamp = emitter(15);
    hittable_list room;
    room.add(make_shared<yz_rect>(0, 555, 0, 555, 555, green));
    room.add(make_shared<yz_rect>(0, 555, 0, 555, 0, red));
    Obj lamp_rect = make_shared<xz_rect>(213, 343, 227, 332, 554, lamp);
    room.add(one_sided_lamp ? Obj(make_shared<flip_face>(lamp_rect)) : lamp_rect);
    room.add(make_shared<xz_rect>(0, 555, 0, 555, 0, white));
    room.add(make_shared<xz_rect>(0, 555, 0, 555, 555, white));
    room.add(make_shared<xy_rect>(0, 555, 0, 555, 555, white));
    room.add(tilted_box(165, 330, 165, 15, vec3(265, 0, 295), white));
    room.add(tilted_box(165, 165, 165, -18, vec3(130, 0, 65), white));
    return make_shared<bvh_node>(room, 0, 1);
}

void cornell_camera(SceneConfig& c) {
    c.aspect_ratio = 1.0;
    c.image_width = 600;
    c.samples_per_pixel = 400;
    c.background = color(0, 0, 0);
    c.lookfrom = point3(278, 278, -800);
    c.lookat = point3(278, 278, 0);
    c.vfov = 40.0;
    c.aperture = 0.0;
}

/* "The Next Week" final scene: 20x20 field of random-height boxes, lamp, moving sphere,
 * glass, fuzzy metal, a glass ball filled with blue fog, a scene-wide thin mist, a globe whose
 * image is absent here, a marble (perlin) ball, and a rotated cloud of 1000 small spheres. */
Obj next_week_final(bool one_sided_lamp) {
    hittable_list field;
    const Mat ground = diffuse(0.48, 0.83, 0.53);
    for (int i = 0; i < 20; i++)
        for (int j = 0; j < 20; j++) {
            const double w = 100.0, x0 = -1000.0 + i * w, z0 = -1000.0 + j * w;
            const double top = random_double(1, 101);
            field.add(make_shared<box>(point3(x0, 0.0, z0), point3(x0 + w, top, z0 + w), ground));
        }
    hittable_list all;
    all.add(make_shared<bvh_node>(field, 0, 1));

    Obj lamp_rect = make_shared<xz_rect>(123, 423, 147, 412, 554, emitter(7));
    all.add(one_sided_lamp ? Obj(make_shared<flip_face>(lamp_rect)) : lamp_rect);

    const point3 c1(400, 400, 200);
    all.add(make_shared<moving_sphere>(c1, c1 + vec3(30, 0, 0), 0, 1, 50, diffuse(0.7, 0.3, 0.1)));
    all.add(make_shared<sphere>(point3(260, 150, 45), 50, make_shared<dielectric>(1.5)));
    all.add(make_shared<sphere>(point3(0, 150, 145), 50, make_shared<metal>(color(0.8, 0.8, 0.9), 1.0)));

    Obj shell = make_shared<sphere>(point3(360, 150, 145), 70, make_shared<dielectric>(1.5));
    all.add(shell);
    all.add(make_shared<constant_medium>(shell, 0.2, color(0.2, 0.4, 0.9)));
    shell = make_shared<sphere>(point3(0, 0, 0), 5000, make_shared<dielectric>(1.5));
    all.add(make_shared<constant_medium>(shell, .0001, color(1, 1, 1)));

    all.add(make_shared<sphere>(point3(400, 200, 400), 100,
                                make_shared<lambertian>(make_shared<image_texture>("earthmap.jpg"))));
    all.add(make_shared<sphere>(point3(220, 280, 300), 80, make_shared<lambertian>(make_shared<noise_texture>(0.1))));

    hittable_list cloud;
    const Mat white = diffuse(.73, .73, .73);
    for (int j = 0; j < 1000; j++) cloud.add(make_shared<sphere>(point3::random(0, 165), 10, white));
    all.add(make_shared<translate>(make_shared<rotate_y>(make_shared<bvh_node>(cloud, 0.0, 1.0), 15),
                                   vec3(-100, 270, 395)));
    return make_shared<bvh_node>(all, 0, 1);
}

/* MIS comparison: ground, near-mirror gold (roughness 0.001, clamped to 0.01 on use), rough
 * silver, glass; a large dim lamp above and a small bright one at the right, both flipped. */
Obj mis_three_spheres() {
    hittable_list w;
    w.add(make_shared<sphere>(point3(0, -1000, 0), 1000, diffuse(0.5, 0.5, 0.5)));
    auto pbr = [](double r, double g, double b, double rough) {
        return make_shared<PBRMaterial>(make_shared<solid_color>(r, g, b), make_shared<solid_color>(rough, rough, rough),
                                        make_shared<solid_color>(1.0, 1.0, 1.0));
    };
    w.add(make_shared<sphere>(point3(-2.5, 1, 0), 1.0, pbr(0.9, 0.6, 0.2, 0.001)));
    w.add(make_shared<sphere>(point3(0, 1, 0), 1.0, pbr(0.8, 0.8, 0.8, 0.4)));
    w.add(make_shared<sphere>(point3(2.5, 1, 0), 1.0, make_shared<dielectric>(1.5)));
    w.add(make_shared<flip_face>(make_shared<xz_rect>(-10, 10, -10, 10, 10, emitter(5))));
    w.add(make_shared<flip_face>(make_shared<yz_rect>(3.75, 4.25, 1.75, 2.25, 6, emitter(50))));
    return make_shared<bvh_node>(w, 0, 1);
}

bool build(int scene_id, SceneConfig& c, std::string& err) {
    switch (scene_id) {
    case 7:
    case 21:
        c.world = cornell_room(scene_id == 21);
        cornell_camera(c);
        if (scene_id == 21)
            c.lights.push_back(
                make_shared<QuadLight>(point3(213, 554, 227), vec3(130, 0, 0), vec3(0, 0, 105), color(15, 15, 15)));
        return true;
    case 9:
    case 22:
        c.world = next_week_final(scene_id == 22);
        c.aspect_ratio = 1.0;
        c.image_width = 800;
        c.samples_per_pixel = 500;
        c.background = color(0, 0, 0);
        c.lookfrom = point3(478, 278, -600);
        c.lookat = point3(278, 278, 0);
        c.vfov = 40.0;
        if (scene_id == 22)
            c.lights.push_back(
                make_shared<QuadLight>(point3(123, 554, 147), vec3(300, 0, 0), vec3(0, 0, 265), color(7, 7, 7)));
        return true;
    case 23:
        c.world = mis_three_spheres();
        c.aspect_ratio = 16.0 / 9.0;
        c.image_width = 800;
        c.samples_per_pixel = 64;
        c.background = color(0.0, 0.0, 0.0);
        c.lookfrom = point3(0, 3, 8);
        c.lookat = point3(0, 1, 0);
        c.vfov = 35.0;
        c.lights.push_back(make_shared<QuadLight>(point3(-10, 10, -10), vec3(20, 0, 0), vec3(0, 0, 20), color(5, 5, 5)));
        c.lights.push_back(make_shared<QuadLight>(point3(6, 4, 2), vec3(0, 0.5, 0), vec3(0, 0, 0.5), color(50, 50, 50)));
        return true;
    default: err = "scene id not described in librtr_host (7, 9, 21, 22, 23)"; return false;
    }
}

} // namespace

/* the product's select_scene: the BASELINE scenes only (the reference's has 40 ids) */
SceneConfig select_scene(int scene_id) {
    SceneConfig c;
    std::string err;
    if (!build(scene_id, c, err)) throw std::invalid_argument(err);
    return c;
}

extern "C" {

struct rtr_host_scene_info {
    int32_t default_width, default_height, default_spp, reserved;
};

/* Builds scene `scene_id` with the scene RNG seeded to `scene_seed`, the reference driver's
 * camera (shutter 0..1, main.cpp:45-46,63-66), flattens it and returns the .rtrs bytes
 * (malloc'ed; free with rtr_host_free).  Returns 0, or a negative rtr_status with `err`. */
int rtr_host_build_scene(int scene_id, uint32_t scene_seed, uint8_t** bytes, size_t* n_bytes,
                         rtr_host_scene_info* info, char* err, size_t err_cap) {
    auto fail = [&](int code, const std::string& m) {
        if (err && err_cap) std::snprintf(err, err_cap, "%s", m.c_str());
        return code;
    };
    if (!bytes || !n_bytes) return fail(RTR_ERR_INVALID, "null output");
    try {
        rtr::rng_state() = scene_seed ? scene_seed : 1u;
        SceneConfig c;
        std::string why;
        if (!build(scene_id, c, why)) return fail(RTR_ERR_UNSUPPORTED, why);
        camera cam(c.lookfrom, c.lookat, c.vup, c.vfov, c.aspect_ratio, c.aperture, c.focus_dist, 0.0, 1.0);
        rtr_scene_storage st;
        if (!rtr::flatten(*c.world, c.lights, cam, c.background, st, why)) return fail(RTR_ERR_UNSUPPORTED, why);
        std::vector<uint8_t> b = st.serialize();
        *bytes = static_cast<uint8_t*>(std::malloc(b.size()));
        if (!*bytes) return fail(RTR_ERR_NOMEM, "malloc");
        std::memcpy(*bytes, b.data(), b.size());
        *n_bytes = b.size();
        if (info) {
            info->default_width = c.image_width;
            info->default_height = static_cast<int>(c.image_width / c.aspect_ratio); /* main.cpp:69 */
            info->default_spp = c.samples_per_pixel;
            info->reserved = 0;
        }
        return RTR_OK;
    } catch (const std::exception& e) {
        return fail(RTR_ERR_INVALID, e.what());
    }
}

void rtr_host_free(uint8_t* p) { std::free(p); }

/* The output stage of the C++ host layer on a linear image (H rows of W pixels, row 0 = bottom row):
 * RenderBuffer::store_linear_rows + to_rgb8 (host/rtr_renderer.h), i.e. what save_to_ppm writes.  `rgb8` receives
 * W*H*3 bytes, top row first.  Lets the CPU tests pin the stage against the reference's own writer. */
int rtr_host_output_stage(const double* linear, int width, int height, uint8_t* rgb8) {
    if (!linear || !rgb8 || width <= 0 || height <= 0) return RTR_ERR_INVALID;
    RenderBuffer buf(width, height);
    buf.store_linear_rows(linear, 0, height, width);
    const std::vector<unsigned char> out = buf.to_rgb8();
    std::memcpy(rgb8, out.data(), out.size());
    return RTR_OK;
}

/* RenderBuffer::save_to_png / save_to_jpg of the host layer (render_buffer.h:35-80) on a linear image; returns 1 / 0
 * like the methods.  For the packed-tile path the image goes in through store_linear_tile, 16x16 tile by tile, the way
 * a Renderer worker stores what rtr_render_tiles_host returns. */
int rtr_host_save_png(const double* linear, int width, int height, const char* path, int through_tiles) {
    if (!linear || !path || width <= 0 || height <= 0) return 0;
    RenderBuffer buf(width, height);
    if (!through_tiles) {
        buf.store_linear_rows(linear, 0, height, width);
    } else {
        double tile[768];
        for (int ty = 0; ty < height; ty += 16)
            for (int tx = 0; tx < width; tx += 16) {
                for (int r = 0; r < 16; ++r)
                    for (int q = 0; q < 16; ++q)
                        for (int c = 0; c < 3; ++c)
                            tile[(r * 16 + q) * 3 + c] = (ty + r < height && tx + q < width)
                                                             ? linear[((size_t)(ty + r) * width + tx + q) * 3 + c]
                                                             : -1.0; /* outside the image: must never be stored */
                buf.store_linear_tile(tile, tx, ty, 0, height);
            }
    }
    return buf.save_to_png(path) ? 1 : 0;
}
int rtr_host_save_jpg(int width, int height, const char* path) {
    RenderBuffer buf(width, height);
    return buf.save_to_jpg(path) ? 1 : 0;
}

/* tile bookkeeping of the multi-context Renderer (host/rtr_renderer.h), for the CPU tests */
int rtr_host_tile_owner(int width, int height, int i, int j, int n_workers) { return rtr::tile_owner(width, height, i, j, n_workers); }

} /* extern "C" */
