/*
 * rtr_scene_io.h -- header-only C++ helper: an owning container for a flattened
 * scene (rtr_scene_desc of rtr_hip.h) and its ".rtrs" file form.
 *
 * File layout (little endian, no padding between sections):
 *   char     magic[8] = "RTRS0001"
 *   int32    root, n_nodes, n_list_children, n_materials, n_textures,
 *            n_perlin, n_images, n_lights
 *   uint64   n_image_bytes
 *   rtr_camera camera (192 B); double background[3]
 *   rtr_node[n_nodes]; int32[n_list_children]; rtr_material[]; rtr_texture[];
 *   rtr_perlin[]; rtr_image[]; uint8[n_image_bytes]; rtr_light[]
 */
#ifndef RTR_SCENE_IO_H
#define RTR_SCENE_IO_H

#include "rtr_hip.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct rtr_scene_storage {
    int32_t root = -1;
    std::vector<rtr_node> nodes;
    std::vector<int32_t> list_children;
    std::vector<rtr_material> materials;
    std::vector<rtr_texture> textures;
    std::vector<rtr_perlin> perlin;
    std::vector<rtr_image> images;
    std::vector<uint8_t> image_bytes;
    std::vector<rtr_light> lights;
    rtr_camera camera{};
    double background[3] = {0, 0, 0};

    /* view valid while *this is alive and unmodified */
    rtr_scene_desc desc() const {
        rtr_scene_desc d;
        std::memset(&d, 0, sizeof d);
        d.abi_version = RTR_ABI_VERSION;
        d.root = root;
        d.n_nodes = (int32_t)nodes.size();
        d.n_list_children = (int32_t)list_children.size();
        d.n_materials = (int32_t)materials.size();
        d.n_textures = (int32_t)textures.size();
        d.n_perlin = (int32_t)perlin.size();
        d.n_images = (int32_t)images.size();
        d.n_lights = (int32_t)lights.size();
        d.n_image_bytes = image_bytes.size();
        d.nodes = nodes.data();
        d.list_children = list_children.data();
        d.materials = materials.data();
        d.textures = textures.data();
        d.perlin = perlin.data();
        d.images = images.data();
        d.image_bytes = image_bytes.data();
        d.lights = lights.data();
        d.camera = camera;
        for (int c = 0; c < 3; ++c) d.background[c] = background[c];
        return d;
    }

    void assign(const rtr_scene_desc& d) {
        root = d.root;
        nodes.assign(d.nodes, d.nodes + d.n_nodes);
        list_children.assign(d.list_children, d.list_children + d.n_list_children);
        materials.assign(d.materials, d.materials + d.n_materials);
        textures.assign(d.textures, d.textures + d.n_textures);
        perlin.assign(d.perlin, d.perlin + d.n_perlin);
        images.assign(d.images, d.images + d.n_images);
        image_bytes.assign(d.image_bytes, d.image_bytes + d.n_image_bytes);
        lights.assign(d.lights, d.lights + d.n_lights);
        camera = d.camera;
        for (int c = 0; c < 3; ++c) background[c] = d.background[c];
    }

    std::vector<uint8_t> serialize() const {
        std::vector<uint8_t> out;
        auto put = [&out](const void* p, size_t n) {
            const uint8_t* b = static_cast<const uint8_t*>(p);
            out.insert(out.end(), b, b + n);
        };
        put("RTRS0001", 8);
        int32_t h[8] = {root,
                        (int32_t)nodes.size(),
                        (int32_t)list_children.size(),
                        (int32_t)materials.size(),
                        (int32_t)textures.size(),
                        (int32_t)perlin.size(),
                        (int32_t)images.size(),
                        (int32_t)lights.size()};
        put(h, sizeof h);
        uint64_t nb = image_bytes.size();
        put(&nb, 8);
        put(&camera, sizeof camera);
        put(background, sizeof background);
        put(nodes.data(), nodes.size() * sizeof(rtr_node));
        put(list_children.data(), list_children.size() * sizeof(int32_t));
        put(materials.data(), materials.size() * sizeof(rtr_material));
        put(textures.data(), textures.size() * sizeof(rtr_texture));
        put(perlin.data(), perlin.size() * sizeof(rtr_perlin));
        put(images.data(), images.size() * sizeof(rtr_image));
        put(image_bytes.data(), image_bytes.size());
        put(lights.data(), lights.size() * sizeof(rtr_light));
        return out;
    }

    bool deserialize(const uint8_t* p, size_t n, std::string* err = nullptr) {
        size_t off = 0;
        auto fail = [&](const char* m) {
            if (err) *err = m;
            return false;
        };
        auto get = [&](void* dst, size_t k) {
            if (off + k > n) return false;
            std::memcpy(dst, p + off, k);
            off += k;
            return true;
        };
        char magic[8];
        if (!get(magic, 8) || std::memcmp(magic, "RTRS0001", 8) != 0) return fail("bad magic");
        int32_t h[8];
        uint64_t nb;
        if (!get(h, sizeof h) || !get(&nb, 8)) return fail("truncated header");
        for (int k = 1; k < 8; ++k)
            if (h[k] < 0) return fail("negative count");
        if (!get(&camera, sizeof camera) || !get(background, sizeof background)) return fail("truncated camera");
        root = h[0];
        nodes.resize(h[1]);
        list_children.resize(h[2]);
        materials.resize(h[3]);
        textures.resize(h[4]);
        perlin.resize(h[5]);
        images.resize(h[6]);
        lights.resize(h[7]);
        image_bytes.resize(nb);
        if (!get(nodes.data(), nodes.size() * sizeof(rtr_node)) ||
            !get(list_children.data(), list_children.size() * sizeof(int32_t)) ||
            !get(materials.data(), materials.size() * sizeof(rtr_material)) ||
            !get(textures.data(), textures.size() * sizeof(rtr_texture)) ||
            !get(perlin.data(), perlin.size() * sizeof(rtr_perlin)) ||
            !get(images.data(), images.size() * sizeof(rtr_image)) || !get(image_bytes.data(), image_bytes.size()) ||
            !get(lights.data(), lights.size() * sizeof(rtr_light)))
            return fail("truncated arrays");
        if (off != n) return fail("trailing bytes");
        return true;
    }

    bool save(const std::string& path) const {
        std::vector<uint8_t> b = serialize();
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) return false;
        size_t w = std::fwrite(b.data(), 1, b.size(), f);
        std::fclose(f);
        return w == b.size();
    }

    bool load(const std::string& path, std::string* err = nullptr) {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) {
            if (err) *err = "cannot open " + path;
            return false;
        }
        std::vector<uint8_t> b;
        uint8_t buf[65536];
        size_t r;
        while ((r = std::fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + r);
        std::fclose(f);
        return deserialize(b.data(), b.size(), err);
    }
};

#endif /* RTR_SCENE_IO_H */
