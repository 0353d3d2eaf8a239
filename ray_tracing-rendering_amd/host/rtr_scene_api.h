/*
 * rtr_scene_api.h -- host-side mirror of the reference's scene-description API.
 *
 * The reference describes a scene by building a graph of C++ objects (hittable, material,
 * texture, Light, camera) and hands it to Renderer::render (renderer/renderer.h:30-32).
 * This header declares classes with the SAME names, constructors and public fields, so the
 * reference's scene builders (scene/scenes.cpp) compile against it unchanged, but the objects
 * here are descriptors: the per-ray work (hit / sample / eval / pdf) runs in the HIP kernels
 * after `rtr::flatten()` lowered the graph to the POD arrays of include/rtr_hip.h.  What does
 * run on the host, with the reference's exact arithmetic, is everything the reference does at
 * scene-build time: bounding boxes, the random-axis median-split BVH (geometry/bvh.h:52-94),
 * rotate_y's box (geometry/hittable.h:96-125), perlin tables (materials/perlin.h:10-19),
 * QuadLight's normal/area (lighting/quad_light.h:11-17) and the camera frame
 * (renderer/camera.h:9-30) -- so a scene built here flattens to the same bytes as the
 * reference's own object graph (tests/test_host_scenes.py).
 *
 * Calling hit()/sample()/... on the host throws: there is no CPU fallback.  A user subclass
 * of hittable/material/texture/Light that the device does not know is rejected by flatten().
 *
 * The directory compat/ holds one-line headers named like the reference's (sphere.h,
 * material.h, ...) that forward here.
 */
#ifndef RTR_SCENE_API_H
#define RTR_SCENE_API_H

#include "rtr_hip.h"
#include "rtr_scene_io.h"

#if defined(__has_include)
#if __has_include(<zlib.h>) && !defined(RTR_NO_ZLIB)
#include <zlib.h> /* PNG textures; link with -lz (define RTR_NO_ZLIB to build without) */
#define RTR_HAVE_ZLIB 1
#endif
#endif
#ifndef RTR_HAVE_ZLIB
#define RTR_HAVE_ZLIB 0
#endif

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

using std::make_shared;
using std::make_unique;
using std::shared_ptr;
using std::sqrt;
using std::unique_ptr;

/* ---- core/rtweekend.h ------------------------------------------------------------------- */
constexpr double infinity = std::numeric_limits<double>::infinity();
constexpr double pi = 3.1415926535897932385;

inline constexpr double degrees_to_radians(double degrees) { return degrees * pi / 180.0; }

namespace rtr {
/* xorshift32 state of the calling thread (core/rtweekend.h:26-27 seeds it from the thread id);
 * settable so scene construction is reproducible */
inline uint32_t& rng_state() {
    static thread_local uint32_t s = static_cast<uint32_t>(std::hash<std::thread::id>{}(std::this_thread::get_id()));
    return s;
}
[[noreturn]] inline void device_only(const char* what) {
    throw std::logic_error(std::string(what) + " runs in the HIP kernels; flatten the scene and render through "
                                               "rtr_hip.h (no CPU fallback)");
}
} // namespace rtr

inline double random_double() { /* core/rtweekend.h:24-34 */
    uint32_t& s = rtr::rng_state();
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s * 2.3283064365386963e-10;
}
inline double random_double(double min, double max) noexcept { return min + (max - min) * random_double(); }
inline double clamp(double x, double min, double max) noexcept { return x < min ? min : (x > max ? max : x); }
inline int random_int(int min, int max) { return static_cast<int>(random_double(min, max + 1)); }

/* ---- core/vec3.h --------------------------------------------------------------------------- */
class vec3 {
  public:
    vec3() : e{0, 0, 0} {}
    vec3(double e0, double e1, double e2) : e{e0, e1, e2} {}
    double x() const noexcept { return e[0]; }
    double y() const noexcept { return e[1]; }
    double z() const noexcept { return e[2]; }
    vec3 operator-() const { return vec3(-e[0], -e[1], -e[2]); }
    double operator[](int i) const { return e[i]; }
    double& operator[](int i) { return e[i]; }
    vec3& operator+=(const vec3& v) {
        e[0] += v.e[0], e[1] += v.e[1], e[2] += v.e[2];
        return *this;
    }
    vec3& operator*=(const double t) {
        e[0] *= t, e[1] *= t, e[2] *= t;
        return *this;
    }
    vec3& operator*=(const vec3& v) {
        e[0] *= v.e[0], e[1] *= v.e[1], e[2] *= v.e[2];
        return *this;
    }
    vec3& operator/=(const double t) { return *this *= 1 / t; }
    double length() const { return sqrt(length_squared()); }
    double length_squared() const noexcept { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
    /* the reference writes vec3(r(), r(), r()); g++ evaluates right to left, stated
     * explicitly here so the draw order does not depend on the compiler */
    static vec3 random() {
        double c = random_double(), b = random_double(), a = random_double();
        return vec3(a, b, c);
    }
    static vec3 random(double min, double max) {
        double c = random_double(min, max), b = random_double(min, max), a = random_double(min, max);
        return vec3(a, b, c);
    }
    bool near_zero() const noexcept {
        const double s = 1e-8;
        return (fabs(e[0]) < s) && (fabs(e[1]) < s) && (fabs(e[2]) < s);
    }
    double e[3];
};
using point3 = vec3;
using color = vec3;

class vec2 {
  public:
    vec2() : e{0, 0} {}
    vec2(double e0, double e1) : e{e0, e1} {}
    double x() const { return e[0]; }
    double y() const { return e[1]; }
    double operator[](int i) const { return e[i]; }
    double& operator[](int i) { return e[i]; }
    double length_squared() const { return e[0] * e[0] + e[1] * e[1]; }
    double length() const { return sqrt(length_squared()); }
    double e[2];
};

inline std::ostream& operator<<(std::ostream& out, const vec3& v) { return out << v.e[0] << ' ' << v.e[1] << ' ' << v.e[2]; }
inline vec3 operator+(const vec3& u, const vec3& v) { return vec3(u.e[0] + v.e[0], u.e[1] + v.e[1], u.e[2] + v.e[2]); }
inline vec3 operator-(const vec3& u, const vec3& v) { return vec3(u.e[0] - v.e[0], u.e[1] - v.e[1], u.e[2] - v.e[2]); }
inline vec3 operator*(const vec3& u, const vec3& v) { return vec3(u.e[0] * v.e[0], u.e[1] * v.e[1], u.e[2] * v.e[2]); }
inline vec3 operator*(double t, const vec3& v) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator*(const vec3& v, double t) { return t * v; }
inline vec3 operator/(vec3 v, double t) { return (1 / t) * v; }
inline double dot(const vec3& u, const vec3& v) { return u.e[0] * v.e[0] + u.e[1] * v.e[1] + u.e[2] * v.e[2]; }
inline vec3 cross(const vec3& u, const vec3& v) {
    return vec3(u.e[1] * v.e[2] - u.e[2] * v.e[1], u.e[2] * v.e[0] - u.e[0] * v.e[2], u.e[0] * v.e[1] - u.e[1] * v.e[0]);
}
inline vec3 unit_vector(const vec3& v) { return v / v.length(); }

/* ---- core/ray.h ------------------------------------------------------------------------------- */
class ray {
  public:
    ray() = default;
    ray(const point3& origin, const vec3& direction, double time = 0.0) noexcept : orig(origin), dir(direction), tm(time) {}
    point3 origin() const noexcept { return orig; }
    vec3 direction() const noexcept { return dir; }
    double time() const noexcept { return tm; }
    point3 at(double t) const noexcept { return orig + t * dir; }

  private:
    point3 orig;
    vec3 dir;
    double tm = 0.0;
};

/* ---- geometry/aabb.h ----------------------------------------------------------------------------- */
class aabb {
  public:
    aabb() : minimum(point3(0, 0, 0)), maximum(point3(0, 0, 0)) {}
    aabb(const point3& a, const point3& b) : minimum(a), maximum(b) {}
    point3 min() const { return minimum; }
    point3 max() const { return maximum; }
    point3 minimum;
    point3 maximum;
};
inline aabb surrounding_box(aabb box0, aabb box1) { /* aabb.h:50-59 */
    point3 small(fmin(box0.min().x(), box1.min().x()), fmin(box0.min().y(), box1.min().y()),
                 fmin(box0.min().z(), box1.min().z()));
    point3 big(fmax(box0.max().x(), box1.max().x()), fmax(box0.max().y(), box1.max().y()),
               fmax(box0.max().z(), box1.max().z()));
    return aabb(small, big);
}

namespace rtr {
struct Flattener;
}

/* ---- materials/texture.h, perlin.h ------------------------------------------------------------------ */
class texture {
  public:
    virtual ~texture() = default;
    virtual color value(double, double, const point3&) const { rtr::device_only("texture::value"); }
    virtual int rtr_flatten(rtr::Flattener&) const;
};

class solid_color : public texture {
  public:
    solid_color() {}
    solid_color(color c) : color_value(c) {}
    solid_color(double red, double green, double blue) : solid_color(color(red, green, blue)) {}
    int rtr_flatten(rtr::Flattener&) const override;
    color color_value;
};

class checker_texture : public texture {
  public:
    checker_texture() {}
    checker_texture(shared_ptr<texture> _even, shared_ptr<texture> _odd) : odd(_odd), even(_even) {}
    checker_texture(color c1, color c2) : odd(make_shared<solid_color>(c2)), even(make_shared<solid_color>(c1)) {}
    int rtr_flatten(rtr::Flattener&) const override;
    shared_ptr<texture> odd;
    shared_ptr<texture> even;
};

class perlin { /* materials/perlin.h:10-19,62-79: 256 unit vectors, then three shuffles */
  public:
    perlin() {
        ranvec.resize(point_count);
        for (int i = 0; i < point_count; ++i) ranvec[i] = unit_vector(vec3::random(-1, 1));
        perm_x = generate_perm();
        perm_y = generate_perm();
        perm_z = generate_perm();
    }
    static constexpr int point_count = 256;
    std::vector<vec3> ranvec;
    std::vector<int> perm_x, perm_y, perm_z;

  private:
    static std::vector<int> generate_perm() {
        std::vector<int> p(point_count);
        for (int i = 0; i < point_count; i++) p[i] = i;
        for (int i = point_count - 1; i > 0; i--) std::swap(p[i], p[random_int(0, i)]);
        return p;
    }
};

class noise_texture : public texture {
  public:
    noise_texture() {}
    noise_texture(double sc) : scale(sc) {}
    int rtr_flatten(rtr::Flattener&) const override;
    perlin noise;
    double scale;
};

/* image_texture (materials/texture.h:82-146).  No image decoder ships with this layer: a file
 * that cannot be read as binary PPM (P6, maxval 255) is "missing", which reproduces the
 * reference's behaviour here, where none of its assets exist: cyan (0,1,1). */
class image_texture : public texture {
  public:
    image_texture(const image_texture&) = delete;
    image_texture& operator=(const image_texture&) = delete;
    image_texture() {}
    image_texture(const char* filename);
    int rtr_flatten(rtr::Flattener&) const override;
    std::vector<unsigned char> data; /* empty = missing */
    int width = 0, height = 0;
};

/* ---- geometry/hittable.h ---------------------------------------------------------------------------- */
class material;

struct hit_record { /* hittable.h:10-23 */
    point3 p;
    vec3 normal;
    material* mat_ptr;
    double t;
    double u;
    double v;
    bool front_face;
    inline void set_face_normal(const ray& r, const vec3& outward_normal) {
        front_face = dot(r.direction(), outward_normal) < 0;
        normal = front_face ? outward_normal : -outward_normal;
    }
};

class hittable {
  public:
    virtual ~hittable() = default;
    virtual bool hit(const ray&, double, double, hit_record&) const { rtr::device_only("hittable::hit"); }
    virtual bool bounding_box(double time0, double time1, aabb& output_box) const = 0;
    virtual int rtr_flatten(rtr::Flattener&) const; /* default: unsupported subclass */
};

/* ---- materials/material.h ------------------------------------------------------------------------------ */
struct BSDFSample { /* material.h:13-20 */
    vec3 wi;
    color f;
    double pdf;
    bool is_specular;
    bool is_transmission = false;
};

class material {
  public:
    virtual ~material() = default;
    virtual color emitted(double, double, const point3&) const { rtr::device_only("material::emitted"); }
    virtual color emitted(const hit_record&, const vec3&) const { rtr::device_only("material::emitted"); }
    virtual bool is_specular() const { return false; }
    virtual bool sample(const hit_record&, const vec3&, BSDFSample&) const { rtr::device_only("material::sample"); }
    virtual color eval(const hit_record&, const vec3&, const vec3&) const { rtr::device_only("material::eval"); }
    virtual double pdf(const hit_record&, const vec3&, const vec3&) const { rtr::device_only("material::pdf"); }
    virtual bool scatter(const ray&, const hit_record&, color&, ray&, double&) const { rtr::device_only("material::scatter"); }
    virtual bool scatter(const ray&, const hit_record&, color&, ray&) const { rtr::device_only("material::scatter"); }
    virtual int rtr_flatten(rtr::Flattener&) const;
};

class lambertian : public material {
  public:
    lambertian(const color& a) : albedo(make_shared<solid_color>(a)) {}
    lambertian(shared_ptr<texture> a) : albedo(a) {}
    int rtr_flatten(rtr::Flattener&) const override;
    shared_ptr<texture> albedo;
};
class metal : public material {
  public:
    metal(const color& a, double f) : albedo(a), fuzz(f < 1 ? f : 1) {}
    int rtr_flatten(rtr::Flattener&) const override;
    color albedo;
    double fuzz;
};
class dielectric : public material {
  public:
    dielectric(double index_of_refraction) : ir(index_of_refraction) {}
    int rtr_flatten(rtr::Flattener&) const override;
    double ir;
};
class diffuse_light : public material {
  public:
    diffuse_light(shared_ptr<texture> a) : emit(a) {}
    diffuse_light(color c) : emit(make_shared<solid_color>(c)) {}
    int rtr_flatten(rtr::Flattener&) const override;
    shared_ptr<texture> emit;
};
class PBRMaterial : public material {
  public:
    PBRMaterial(shared_ptr<texture> a, shared_ptr<texture> r, shared_ptr<texture> m, shared_ptr<texture> n = nullptr)
        : albedo(a), roughness(r), metallic(m), normal_map(n) {}
    int rtr_flatten(rtr::Flattener&) const override;
    shared_ptr<texture> albedo, roughness, metallic, normal_map;
};
class isotropic : public material { /* geometry/constant_medium.h:12-29 */
  public:
    isotropic(color c) : albedo(make_shared<solid_color>(c)) {}
    isotropic(shared_ptr<texture> a) : albedo(a) {}
    int rtr_flatten(rtr::Flattener&) const override;
    shared_ptr<texture> albedo;
};

/* ---- instance wrappers, geometry/hittable.h:34-179 -------------------------------------------------------- */
class translate : public hittable {
  public:
    translate(shared_ptr<hittable> p, const vec3& displacement) : ptr(p), offset(displacement) {}
    bool bounding_box(double time0, double time1, aabb& output_box) const override {
        if (!ptr->bounding_box(time0, time1, output_box)) return false;
        output_box = aabb(output_box.min() + offset, output_box.max() + offset);
        return true;
    }
    int rtr_flatten(rtr::Flattener&) const override;
    shared_ptr<hittable> ptr;
    vec3 offset;
};

class rotate_y : public hittable {
  public:
    rotate_y(shared_ptr<hittable> p, double angle) : ptr(p) { /* hittable.h:96-125 */
        const double radians = degrees_to_radians(angle);
        sin_theta = sin(radians);
        cos_theta = cos(radians);
        hasbox = ptr->bounding_box(0, 1, bbox);
        point3 lo(infinity, infinity, infinity), hi(-infinity, -infinity, -infinity);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    const double x = i * bbox.max().x() + (1 - i) * bbox.min().x();
                    const double y = j * bbox.max().y() + (1 - j) * bbox.min().y();
                    const double z = k * bbox.max().z() + (1 - k) * bbox.min().z();
                    const vec3 corner(cos_theta * x + sin_theta * z, y, -sin_theta * x + cos_theta * z);
                    for (int c = 0; c < 3; c++) {
                        lo[c] = fmin(lo[c], corner[c]);
                        hi[c] = fmax(hi[c], corner[c]);
                    }
                }
        bbox = aabb(lo, hi);
    }
    bool bounding_box(double, double, aabb& output_box) const override {
        output_box = bbox;
        return hasbox;
    }
    int rtr_flatten(rtr::Flattener&) const override;
    shared_ptr<hittable> ptr;
    double sin_theta, cos_theta;
    bool hasbox;
    aabb bbox;
};

class flip_face : public hittable {
  public:
    flip_face(shared_ptr<hittable> p) : ptr(p) {}
    bool bounding_box(double time0, double time1, aabb& output_box) const override {
        return ptr->bounding_box(time0, time1, output_box);
    }
    int rtr_flatten(rtr::Flattener&) const override;
    shared_ptr<hittable> ptr;
};

/* ---- geometry/hittable_list.h ---------------------------------------------------------------------------------- */
class hittable_list : public hittable {
  public:
    hittable_list() {}
    hittable_list(shared_ptr<hittable> object) { add(object); }
    void clear() { objects.clear(); }
    void add(shared_ptr<hittable> object) { objects.push_back(object); }
    bool bounding_box(double time0, double time1, aabb& output_box) const override { /* :49-66 */
        if (objects.empty()) return false;
        aabb temp_box;
        bool first_box = true;
        for (const auto& object : objects) {
            if (!object->bounding_box(time0, time1, temp_box)) return false;
            output_box = first_box ? temp_box : surrounding_box(output_box, temp_box);
            first_box = false;
        }
        return true;
    }
    int rtr_flatten(rtr::Flattener&) const override;
    std::vector<shared_ptr<hittable>> objects;
};

/* ---- primitives ---------------------------------------------------------------------------------------------------- */
class sphere : public hittable { /* geometry/sphere.h */
  public:
    sphere(point3 cen, double r, shared_ptr<material> m) : center(cen), radius(r), mat_ptr(std::move(m)) {}
    bool bounding_box(double, double, aabb& output_box) const override {
        output_box = aabb(center - vec3(radius, radius, radius), center + vec3(radius, radius, radius));
        return true;
    }
    int rtr_flatten(rtr::Flattener&) const override;
    point3 center;
    double radius;
    shared_ptr<material> mat_ptr;
};

class moving_sphere : public hittable { /* geometry/moving_sphere.h */
  public:
    moving_sphere() {}
    moving_sphere(point3 cen0, point3 cen1, double _time0, double _time1, double r, shared_ptr<material> m)
        : center0(cen0), center1(cen1), time0(_time0), time1(_time1), radius(r), mat_ptr(m) {}
    point3 center(double time) const { return center0 + ((time - time0) / (time1 - time0)) * (center1 - center0); }
    bool bounding_box(double _time0, double _time1, aabb& output_box) const override {
        const vec3 r3(radius, radius, radius);
        aabb box0(center(_time0) - r3, center(_time0) + r3);
        aabb box1(center(_time1) - r3, center(_time1) + r3);
        output_box = surrounding_box(box0, box1);
        return true;
    }
    int rtr_flatten(rtr::Flattener&) const override;
    point3 center0, center1;
    double time0, time1;
    double radius;
    shared_ptr<material> mat_ptr;
};

/* geometry/aarect.h: boxes are padded by 1e-4 along the constant axis */
class xy_rect : public hittable {
  public:
    xy_rect() {}
    xy_rect(double _x0, double _x1, double _y0, double _y1, double _k, shared_ptr<material> mat)
        : mp(mat), x0(_x0), x1(_x1), y0(_y0), y1(_y1), k(_k) {}
    bool bounding_box(double, double, aabb& output_box) const override {
        output_box = aabb(point3(x0, y0, k - 0.0001), point3(x1, y1, k + 0.0001));
        return true;
    }
    int rtr_flatten(rtr::Flattener&) const override;
    shared_ptr<material> mp;
    double x0, x1, y0, y1, k;
};
class xz_rect : public hittable {
  public:
    xz_rect() {}
    xz_rect(double _x0, double _x1, double _z0, double _z1, double _k, shared_ptr<material> mat)
        : mp(mat), x0(_x0), x1(_x1), z0(_z0), z1(_z1), k(_k) {}
    bool bounding_box(double, double, aabb& output_box) const override {
        output_box = aabb(vec3(x0, k - 0.0001, z0), vec3(x1, k + 0.0001, z1));
        return true;
    }
    int rtr_flatten(rtr::Flattener&) const override;
    shared_ptr<material> mp;
    double x0, x1, z0, z1, k;
};
class yz_rect : public hittable {
  public:
    yz_rect() {}
    yz_rect(double _y0, double _y1, double _z0, double _z1, double _k, shared_ptr<material> mat)
        : mp(mat), y0(_y0), y1(_y1), z0(_z0), z1(_z1), k(_k) {}
    bool bounding_box(double, double, aabb& output_box) const override {
        output_box = aabb(vec3(k - 0.0001, y0, z0), vec3(k + 0.0001, y1, z1));
        return true;
    }
    int rtr_flatten(rtr::Flattener&) const override;
    shared_ptr<material> mp;
    double y0, y1, z0, z1, k;
};

class box : public hittable { /* geometry/box.h: six rects in a hittable_list, +z -z +y -y +x -x */
  public:
    box() {}
    box(const point3& p0, const point3& p1, shared_ptr<material> ptr) : box_min(p0), box_max(p1) {
        sides.add(make_shared<xy_rect>(p0.x(), p1.x(), p0.y(), p1.y(), p1.z(), ptr));
        sides.add(make_shared<xy_rect>(p0.x(), p1.x(), p0.y(), p1.y(), p0.z(), ptr));
        sides.add(make_shared<xz_rect>(p0.x(), p1.x(), p0.z(), p1.z(), p1.y(), ptr));
        sides.add(make_shared<xz_rect>(p0.x(), p1.x(), p0.z(), p1.z(), p0.y(), ptr));
        sides.add(make_shared<yz_rect>(p0.y(), p1.y(), p0.z(), p1.z(), p1.x(), ptr));
        sides.add(make_shared<yz_rect>(p0.y(), p1.y(), p0.z(), p1.z(), p0.x(), ptr));
    }
    bool bounding_box(double, double, aabb& output_box) const override {
        output_box = aabb(box_min, box_max);
        return true;
    }
    int rtr_flatten(rtr::Flattener&) const override;
    point3 box_min, box_max;
    hittable_list sides;
};

class constant_medium : public hittable { /* geometry/constant_medium.h:31-53 */
  public:
    constant_medium(shared_ptr<hittable> b, double d, shared_ptr<texture> a)
        : boundary(b), phase_function(make_shared<isotropic>(a)), neg_inv_density(-1 / d) {}
    constant_medium(shared_ptr<hittable> b, double d, color c)
        : boundary(b), phase_function(make_shared<isotropic>(c)), neg_inv_density(-1 / d) {}
    bool bounding_box(double time0, double time1, aabb& output_box) const override {
        return boundary->bounding_box(time0, time1, output_box);
    }
    int rtr_flatten(rtr::Flattener&) const override;
    shared_ptr<hittable> boundary;
    shared_ptr<material> phase_function;
    double neg_inv_density;
};

/* ---- geometry/bvh.h --------------------------------------------------------------------------------------------------- */
class bvh_node : public hittable {
  public:
    bvh_node(const hittable_list& list, double time0, double time1)
        : bvh_node(list.objects, 0, list.objects.size(), time0, time1) {}
    /* random split axis per node, sort the span by box minimum on that axis, split at the
     * median; spans of one put the object on both sides (bvh.h:52-94).  Each node works on its
     * own copy of the object vector, like the reference, so sibling sorts do not interact. */
    bvh_node(const std::vector<shared_ptr<hittable>>& src_objects, size_t start, size_t end, double time0,
             double time1) {
        std::vector<shared_ptr<hittable>> objs = src_objects;
        const int axis = random_int(0, 2);
        auto less_on_axis = [axis](const shared_ptr<hittable>& a, const shared_ptr<hittable>& b) {
            aabb ba, bb;
            if (!a->bounding_box(0, 0, ba) || !b->bounding_box(0, 0, bb))
                std::cerr << "No bounding box in bvh_node constructor.\n";
            return ba.min()[axis] < bb.min()[axis];
        };
        const size_t span = end - start;
        if (span == 1) {
            left = right = objs[start];
        } else if (span == 2) {
            const bool in_order = less_on_axis(objs[start], objs[start + 1]);
            left = objs[in_order ? start : start + 1];
            right = objs[in_order ? start + 1 : start];
        } else {
            std::sort(objs.begin() + start, objs.begin() + end, less_on_axis);
            const size_t mid = start + span / 2;
            left = make_shared<bvh_node>(objs, start, mid, time0, time1);
            right = make_shared<bvh_node>(objs, mid, end, time0, time1);
        }
        aabb bl, br;
        if (!left->bounding_box(time0, time1, bl) || !right->bounding_box(time0, time1, br))
            std::cerr << "No bounding box in bvh_node constructor.\n";
        box = surrounding_box(bl, br);
    }
    bool bounding_box(double, double, aabb& output_box) const override {
        output_box = box;
        return true;
    }
    int rtr_flatten(rtr::Flattener&) const override;
    shared_ptr<hittable> left;
    shared_ptr<hittable> right;
    aabb box;
};

/* ---- lighting ----------------------------------------------------------------------------------------------------------- */
struct LightSample { /* lighting/light.h:7-13 */
    color Li;
    vec3 wi;
    double pdf;
    double dist;
    bool is_delta;
};

class Light {
  public:
    virtual ~Light() = default;
    virtual LightSample sample(const point3&, const vec2&) const { rtr::device_only("Light::sample"); }
    virtual double pdf(const point3&, const vec3&) const { rtr::device_only("Light::pdf"); }
    virtual bool is_delta() const { return false; }
    virtual bool is_infinite() const { return false; }
    virtual color Le(const ray&) const { return color(0, 0, 0); }
    virtual color power() const { return color(0, 0, 0); }
    virtual bool rtr_flatten(rtr_light& out, std::string& why) const {
        (void)out;
        why = "this Light subclass has no device implementation";
        return false;
    }
    /* lights with bulk data (environment maps) append it to the scene's byte blob */
    virtual bool rtr_flatten(rtr_light& out, std::vector<uint8_t>& blob, std::string& why) const {
        (void)blob;
        return rtr_flatten(out, why);
    }
};

class QuadLight : public Light { /* lighting/quad_light.h:9-17 */
  public:
    QuadLight(const point3& _Q, const vec3& _u, const vec3& _v, const color& _c) : Q(_Q), u(_u), v(_v), intensity(_c) {
        const vec3 n = cross(u, v);
        area = n.length();
        normal = unit_vector(n);
    }
    bool rtr_flatten(rtr_light& out, std::string&) const override {
        out = rtr_light{};
        out.type = RTR_LIGHT_QUAD;
        const vec3* parts[5] = {&Q, &u, &v, &intensity, &normal};
        for (int k = 0; k < 5; ++k)
            for (int c = 0; c < 3; ++c) out.f[3 * k + c] = (*parts[k])[c];
        out.f[15] = area;
        return true;
    }
    point3 Q;
    vec3 u, v;
    color intensity;
    vec3 normal;
    double area;
};

/* delta lights (lighting/point_light.h, spot_light.h, directional_light.h) */
class PointLight : public Light {
  public:
    PointLight(const point3& pos, const color& intensity) : m_position(pos), m_intensity(intensity) {}
    bool is_delta() const override { return true; }
    bool rtr_flatten(rtr_light& out, std::string&) const override {
        out = rtr_light{};
        out.type = RTR_LIGHT_POINT;
        for (int c = 0; c < 3; ++c) out.f[c] = m_position[c], out.f[3 + c] = m_intensity[c];
        return true;
    }
    point3 m_position;
    color m_intensity;
};
class SpotLight : public Light {
  public:
    SpotLight(point3 pos, vec3 dir, double cutoff, color intensity_)
        : position(pos), direction(unit_vector(dir)), intensity(intensity_) {
        const double r = cutoff * (pi / 180.0); /* spot_light.h:10-11 */
        cos_cutoff = cos(r);
    }
    bool is_delta() const override { return true; }
    bool rtr_flatten(rtr_light& out, std::string&) const override {
        out = rtr_light{};
        out.type = RTR_LIGHT_SPOT;
        for (int c = 0; c < 3; ++c) out.f[c] = position[c], out.f[3 + c] = direction[c], out.f[6 + c] = intensity[c];
        out.f[9] = cos_cutoff;
        return true;
    }
    point3 position;
    vec3 direction;
    color intensity;
    double cos_cutoff;
};
class DirectionalLight : public Light {
  public:
    DirectionalLight(const vec3& dir, const color& c) : direction(unit_vector(dir)), L(c) {}
    bool is_delta() const override { return true; }
    bool rtr_flatten(rtr_light& out, std::string&) const override {
        out = rtr_light{};
        out.type = RTR_LIGHT_DIRECTIONAL;
        for (int c = 0; c < 3; ++c) out.f[c] = direction[c], out.f[3 + c] = L[c];
        return true;
    }
    vec3 direction;
    color L;
};
/* EnvironmentLight (lighting/environmental_light.h:120-374).  The map is read by this layer's own
 * Radiance RGBE reader (flat and run-length scanlines; texel = mantissa * 2^(e-136) in float, the
 * conversion stbi_loadf applies), the luminance * sin(theta) table and its Distribution2D are built
 * like :147-179,:14-26,:64-80, and everything is handed to the device through the scene's byte blob
 * (rtr_hip.h, RTR_LIGHT_ENV_MAP).  A file that cannot be read gives the reference's own fallback:
 * a uniform white sky (:126-131,187-192,226-229,293-294). */
class EnvironmentLight : public Light {
  public:
    EnvironmentLight(const char* map_filename) : filename(map_filename ? map_filename : "") {
        if (!load_rgbe(filename.c_str())) {
            std::cerr << "ERROR: Could not load HDR environment map: " << filename << std::endl;
            width = height = 0;
            hdr_data.clear();
            return;
        }
        is_light_probe = width > 0 && height > 0 && width == height; /* :137-140 */
        build_distribution();
    }
    bool is_infinite() const override { return true; }
    bool rtr_flatten(rtr_light& out, std::string&) const override {
        out = rtr_light{};
        out.type = RTR_LIGHT_ENV_UNIFORM;
        return true;
    }
    bool rtr_flatten(rtr_light& out, std::vector<uint8_t>& blob, std::string& why) const override {
        if (width == 0 || height == 0) return rtr_flatten(out, why);
        out = rtr_light{};
        out.type = RTR_LIGHT_ENV_MAP;
        auto append = [&blob](const void* p, size_t n) {
            const uint8_t* b = static_cast<const uint8_t*>(p);
            blob.insert(blob.end(), b, b + n);
        };
        while (blob.size() % 8) blob.push_back(0);
        out.f[0] = width, out.f[1] = height, out.f[2] = is_light_probe ? 1.0 : 0.0;
        out.f[3] = (double)blob.size();
        append(hdr_data.data(), hdr_data.size() * sizeof(float));
        while (blob.size() % 8) blob.push_back(0);
        out.f[4] = (double)blob.size();
        append(tables.data(), tables.size() * sizeof(double));
        return true;
    }
    std::string filename;
    std::vector<float> hdr_data;
    std::vector<double> tables; /* per row {func[w], cdf[w+1], func_int}, then the marginal {func[h], cdf[h+1], func_int} */
    int width = 0, height = 0;
    bool is_light_probe = false;
    double total_power = 0;

  private:
    /* Distribution1D::Distribution1D (:14-26), appended to `tables`; returns func_int */
    double push_distribution(const std::vector<double>& func) {
        const int n = (int)func.size();
        std::vector<double> cdf(n + 1);
        cdf[0] = 0;
        for (int i = 1; i <= n; ++i) cdf[i] = cdf[i - 1] + func[i - 1];
        const double func_int = cdf[n];
        if (func_int > 0)
            for (int i = 0; i <= n; ++i) cdf[i] /= func_int;
        tables.insert(tables.end(), func.begin(), func.end());
        tables.insert(tables.end(), cdf.begin(), cdf.end());
        tables.push_back(func_int);
        return func_int;
    }
    void build_distribution() { /* :147-179 */
        std::vector<double> marginal_func(height);
        total_power = 0;
        for (int v = 0; v < height; ++v) {
            const double sin_theta = sin(pi * (v + 0.5) / height);
            std::vector<double> row(width);
            for (int u = 0; u < width; ++u) {
                const int pixel_idx = (v * width + u) * 3;
                const double r = hdr_data[pixel_idx], g = hdr_data[pixel_idx + 1], b = hdr_data[pixel_idx + 2];
                const double lum = 0.2126 * r + 0.7152 * g + 0.0722 * b;
                row[u] = lum * sin_theta;
            }
            marginal_func[v] = push_distribution(row);
        }
        push_distribution(marginal_func);
    }
    static bool read_line(FILE* f, std::string& line) {
        line.clear();
        int ch;
        while ((ch = std::fgetc(f)) != EOF && ch != '\n') line.push_back((char)ch);
        return ch != EOF || !line.empty();
    }
    bool load_rgbe(const char* path) {
        FILE* f = std::fopen(path, "rb");
        if (!f) return false;
        bool ok = false;
        do {
            std::string line;
            if (!read_line(f, line) || (line != "#?RADIANCE" && line != "#?RGBE")) break;
            bool format_ok = false;
            while (read_line(f, line) && !line.empty())
                if (line == "FORMAT=32-bit_rle_rgbe") format_ok = true;
            if (!format_ok || !read_line(f, line)) break;
            int h = 0, w = 0;
            if (std::sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || w > 65536 || h > 65536) break;
            std::vector<unsigned char> rgbe((size_t)w * h * 4);
            bool flat = w < 8 || w >= 32768;
            size_t done_rows = 0;
            if (!flat) {
                std::vector<unsigned char> scan((size_t)w * 4);
                for (int j = 0; j < h; ++j) {
                    unsigned char hd[4];
                    if (std::fread(hd, 1, 4, f) != 4) goto fail;
                    if (hd[0] != 2 || hd[1] != 2 || (hd[2] & 0x80)) {
                        if (j != 0) goto fail;
                        /* not run-length encoded: these four bytes are the first texel of a flat file */
                        std::memcpy(rgbe.data(), hd, 4);
                        if (std::fread(rgbe.data() + 4, 1, rgbe.size() - 4, f) != rgbe.size() - 4) goto fail;
                        done_rows = (size_t)h;
                        break;
                    }
                    if (((hd[2] << 8) | hd[3]) != w) goto fail;
                    for (int comp = 0; comp < 4; ++comp) {
                        int i = 0;
                        while (i < w) {
                            int count = std::fgetc(f);
                            if (count == EOF) goto fail;
                            if (count > 128) {
                                const int value = std::fgetc(f);
                                count -= 128;
                                if (value == EOF || count == 0 || count > w - i) goto fail;
                                for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + comp] = (unsigned char)value;
                            } else {
                                if (count == 0 || count > w - i) goto fail;
                                for (int z = 0; z < count; ++z) {
                                    const int value = std::fgetc(f);
                                    if (value == EOF) goto fail;
                                    scan[(size_t)(i++) * 4 + comp] = (unsigned char)value;
                                }
                            }
                        }
                    }
                    std::memcpy(rgbe.data() + (size_t)j * w * 4, scan.data(), scan.size());
                    ++done_rows;
                }
            } else if (std::fread(rgbe.data(), 1, rgbe.size(), f) == rgbe.size()) {
                done_rows = (size_t)h;
            }
            if (done_rows != (size_t)h) break;
            hdr_data.resize((size_t)w * h * 3);
            for (size_t k = 0; k < (size_t)w * h; ++k) {
                const unsigned char* q = &rgbe[k * 4];
                if (q[3] != 0) {
                    const float f1 = (float)std::ldexp(1.0f, (int)q[3] - (128 + 8));
                    hdr_data[k * 3] = q[0] * f1, hdr_data[k * 3 + 1] = q[1] * f1, hdr_data[k * 3 + 2] = q[2] * f1;
                } else {
                    hdr_data[k * 3] = hdr_data[k * 3 + 1] = hdr_data[k * 3 + 2] = 0;
                }
            }
            width = w, height = h;
            ok = true;
        } while (false);
    fail:
        std::fclose(f);
        return ok;
    }
};

/* ---- renderer/camera.h:9-30 ------------------------------------------------------------------------------------------------ */
class camera {
  public:
    camera(point3 lookfrom, point3 lookat, point3 vup, double vfov, double aspect_ratio, double aperture,
           double focus_dist, double _time0 = 0.0, double _time1 = 0.0) {
        const double theta = degrees_to_radians(vfov);
        const double h = tan(theta / 2);
        const double viewport_height = 2.0 * h;
        const double viewport_width = aspect_ratio * viewport_height;
        w = unit_vector(lookfrom - lookat);
        u = unit_vector(cross(vup, w));
        v = cross(w, u);
        origin = lookfrom;
        horizontal = focus_dist * viewport_width * u;
        vertical = focus_dist * viewport_height * v;
        lower_left_corner = origin - horizontal / 2 - vertical / 2 - focus_dist * w;
        lens_radius = aperture / 2;
        time0 = _time0;
        time1 = _time1;
    }
    ray get_ray(double, double) const { rtr::device_only("camera::get_ray"); }
    rtr_camera rtr_flatten() const {
        rtr_camera c{};
        for (int k = 0; k < 3; ++k) {
            c.origin[k] = origin[k], c.lower_left_corner[k] = lower_left_corner[k];
            c.horizontal[k] = horizontal[k], c.vertical[k] = vertical[k];
            c.u[k] = u[k], c.v[k] = v[k], c.w[k] = w[k];
        }
        c.lens_radius = lens_radius, c.time0 = time0, c.time1 = time1;
        return c;
    }
    point3 origin, lower_left_corner;
    vec3 horizontal, vertical, u, v, w;
    double lens_radius, time0, time1;
};

/* ---- scene/scenes.h:11-24 ---------------------------------------------------------------------------------------------------- */
/* Shares the reference header's include guard: when the reference's scenes.cpp is compiled
 * against this layer it includes ITS OWN scenes.h first (same directory), which then defines
 * the identical struct. */
#ifndef SCENES_H
#define SCENES_H
struct SceneConfig {
    shared_ptr<hittable> world;
    std::vector<shared_ptr<Light>> lights;
    color background{0, 0, 0};
    point3 lookfrom{13, 2, 3};
    point3 lookat{0, 0, 0};
    vec3 vup{0, 1, 0};
    double vfov = 40.0;
    double aperture = 0.0;
    double focus_dist = 10.0;
    double aspect_ratio = 16.0 / 9.0;
    int image_width = 1280;
    int samples_per_pixel = 100;
};
SceneConfig select_scene(int scene_id);
#endif /* SCENES_H */

/* ---- flattening ------------------------------------------------------------------------------------------------------------------ */
namespace rtr {

/* Lowers the object graph to rtr_scene_storage.  Index order (it fixes the byte image):
 * nodes in DFS pre-order (bvh: left then right; list: in order; medium: phase material before
 * boundary), materials and textures at first encounter. */
struct Flattener {
    rtr_scene_storage out;
    std::string error;
    std::map<const hittable*, int> node_ix;
    std::map<const material*, int> mat_ix;
    std::map<const texture*, int> tex_ix;

    static void put3(double* f, const vec3& v) { f[0] = v.x(), f[1] = v.y(), f[2] = v.z(); }
    int fail(const std::string& m) {
        if (error.empty()) error = m;
        return -1;
    }
    int tex(const texture* t) {
        if (!t) return -1;
        auto it = tex_ix.find(t);
        if (it != tex_ix.end()) return it->second;
        const int ix = (int)out.textures.size();
        tex_ix[t] = ix;
        out.textures.push_back(rtr_texture{});
        t->rtr_flatten(*this);
        return ix;
    }
    int mat(const material* m) {
        auto it = mat_ix.find(m);
        if (it != mat_ix.end()) return it->second;
        const int ix = (int)out.materials.size();
        mat_ix[m] = ix;
        rtr_material r{};
        for (int k = 0; k < 4; ++k) r.tex[k] = -1;
        out.materials.push_back(r);
        m->rtr_flatten(*this);
        return ix;
    }
    int node(const hittable* h) {
        auto it = node_ix.find(h);
        if (it != node_ix.end()) return it->second;
        const int ix = (int)out.nodes.size();
        node_ix[h] = ix;
        out.nodes.push_back(rtr_node{});
        h->rtr_flatten(*this);
        return ix;
    }
    rtr_node& cur(const hittable* h) { return out.nodes[node_ix[h]]; }
    rtr_material& cur(const material* m) { return out.materials[mat_ix[m]]; }
    rtr_texture& cur(const texture* t) { return out.textures[tex_ix[t]]; }
    void list(const hittable* self, const hittable_list& l) {
        std::vector<int> kids;
        for (const auto& o : l.objects) kids.push_back(node(o.get()));
        rtr_node& r = cur(self);
        r.type = RTR_NODE_LIST;
        r.a = (int)out.list_children.size();
        r.b = (int)kids.size();
        out.list_children.insert(out.list_children.end(), kids.begin(), kids.end());
    }
};

/* world + lights + camera + background -> flattened scene; false (and `error`) if some object
 * has no device counterpart */
inline bool flatten(const hittable& world, const std::vector<shared_ptr<Light>>& lights, const camera& cam,
                    const color& background, rtr_scene_storage& out, std::string& error) {
    Flattener f;
    f.out.root = f.node(&world);
    for (const auto& l : lights) {
        rtr_light r{};
        std::string why;
        if (!l->rtr_flatten(r, f.out.image_bytes, why)) f.fail(why);
        f.out.lights.push_back(r);
    }
    f.out.camera = cam.rtr_flatten();
    Flattener::put3(f.out.background, background);
    if (!f.error.empty()) {
        error = f.error;
        return false;
    }
    out = std::move(f.out);
    return true;
}

} // namespace rtr

/* ---- rtr_flatten bodies --------------------------------------------------------------------------------------------------------------- */
inline int hittable::rtr_flatten(rtr::Flattener& f) const { return f.fail("hittable subclass without a device implementation"); }
inline int material::rtr_flatten(rtr::Flattener& f) const { return f.fail("material subclass without a device implementation"); }
inline int texture::rtr_flatten(rtr::Flattener& f) const { return f.fail("texture subclass without a device implementation"); }

inline int solid_color::rtr_flatten(rtr::Flattener& f) const {
    rtr_texture& r = f.cur(this);
    r.type = RTR_TEX_SOLID;
    rtr::Flattener::put3(r.f, color_value);
    return 0;
}
inline int checker_texture::rtr_flatten(rtr::Flattener& f) const {
    const int a = f.tex(even.get());
    const int b = f.tex(odd.get());
    rtr_texture& r = f.cur(this);
    r.type = RTR_TEX_CHECKER;
    r.a = a, r.b = b;
    return 0;
}
inline int noise_texture::rtr_flatten(rtr::Flattener& f) const {
    rtr_perlin p{};
    for (int i = 0; i < 256; ++i) {
        rtr::Flattener::put3(p.ranvec[i], noise.ranvec[i]);
        p.perm_x[i] = noise.perm_x[i], p.perm_y[i] = noise.perm_y[i], p.perm_z[i] = noise.perm_z[i];
    }
    rtr_texture& r = f.cur(this);
    r.type = RTR_TEX_NOISE;
    r.a = (int)f.out.perlin.size();
    r.f[0] = scale;
    f.out.perlin.push_back(p);
    return 0;
}
/* image files: binary PPM always; PNG (8/16-bit, grey / grey+alpha / RGB / RGBA / palette, non-interlaced)
 * when zlib is there.  Texels come out as the 3 x 8-bit RGB stb_image hands the reference for
 * req_comp = 3 (materials/texture.h:96-100): grey replicated, alpha dropped, 16-bit samples >> 8,
 * 1/2/4-bit grey scaled to 0..255.  JPEG and interlaced PNG are not decoded: such a file counts as
 * missing (cyan fallback, texture.h:101-105). */
namespace rtr {
inline bool read_file(const char* path, std::vector<unsigned char>& out) {
    FILE* fp = path ? std::fopen(path, "rb") : nullptr;
    if (!fp) return false;
    unsigned char buf[65536];
    size_t r;
    while ((r = std::fread(buf, 1, sizeof buf, fp)) > 0) out.insert(out.end(), buf, buf + r);
    std::fclose(fp);
    return true;
}
inline bool decode_ppm(const std::vector<unsigned char>& file, std::vector<unsigned char>& rgb, int& w, int& h) {
    if (file.size() < 2 || file[0] != 'P' || file[1] != '6') return false;
    size_t pos = 2;
    int vals[3];
    for (int k = 0; k < 3; ++k) {
        while (pos < file.size() && (std::isspace(file[pos]) || file[pos] == '#')) {
            if (file[pos] == '#')
                while (pos < file.size() && file[pos] != '\n') ++pos;
            else
                ++pos;
        }
        long v = 0;
        bool any = false;
        while (pos < file.size() && std::isdigit(file[pos])) v = v * 10 + (file[pos++] - '0'), any = true;
        if (!any || v <= 0 || v > 65536) return false;
        vals[k] = (int)v;
    }
    if (vals[2] != 255 || pos >= file.size()) return false;
    ++pos; /* the single whitespace after maxval */
    const size_t need = (size_t)vals[0] * vals[1] * 3;
    if (file.size() - pos < need) return false;
    rgb.assign(file.begin() + pos, file.begin() + pos + need);
    w = vals[0], h = vals[1];
    return true;
}
#if RTR_HAVE_ZLIB
inline bool decode_png(const std::vector<unsigned char>& file, std::vector<unsigned char>& rgb, int& w, int& h) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return false;
    auto be32 = [&](size_t o) { return ((uint32_t)file[o] << 24) | (file[o + 1] << 16) | (file[o + 2] << 8) | file[o + 3]; };
    size_t pos = 8;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<unsigned char> idat, plte;
    w = h = 0;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(pos);
        const unsigned char* tag = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) return false;
        const unsigned char* d = &file[pos + 8];
        if (!std::memcmp(tag, "IHDR", 4) && len == 13) {
            w = (int)be32(pos + 8), h = (int)be32(pos + 12);
            depth = d[8], ctype = d[9], interlace = d[12];
        } else if (!std::memcmp(tag, "PLTE", 4)) {
            plte.assign(d, d + len);
        } else if (!std::memcmp(tag, "IDAT", 4)) {
            idat.insert(idat.end(), d, d + len);
        } else if (!std::memcmp(tag, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (w <= 0 || h <= 0 || w > 65536 || h > 65536 || interlace != 0) return false;
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return false;
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return false;
    if (ctype == 3 && depth == 16) return false;
    const size_t bpp_bits = (size_t)channels * depth, stride = ((size_t)w * bpp_bits + 7) / 8;
    const size_t bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1; /* filter distance in bytes */
    std::vector<unsigned char> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return false;
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    rgb.resize((size_t)w * h * 3);
    for (int j = 0; j < h; ++j) {
        const unsigned char* line = &raw[(stride + 1) * (size_t)j];
        const int filter = line[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred;
            switch (filter) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: return false;
            }
            cur[i] = (unsigned char)(line[1 + i] + pred);
        }
        for (int i = 0; i < w; ++i) {
            unsigned char px[4] = {0, 0, 0, 0};
            for (int ch = 0; ch < channels; ++ch) {
                unsigned v;
                if (depth == 8) {
                    v = cur[(size_t)i * channels + ch];
                } else if (depth == 16) {
                    v = cur[((size_t)i * channels + ch) * 2]; /* high byte = sample >> 8 */
                } else {
                    const size_t bit = (size_t)i * depth;
                    v = (cur[bit / 8] >> (8 - depth - bit % 8)) & ((1u << depth) - 1);
                    if (ctype == 0) v *= depth == 1 ? 255 : (depth == 2 ? 85 : 17);
                }
                px[ch] = (unsigned char)v;
            }
            unsigned char* o = &rgb[((size_t)j * w + i) * 3];
            if (ctype == 3) {
                const size_t e = (size_t)px[0] * 3;
                if (e + 3 > plte.size()) return false;
                o[0] = plte[e], o[1] = plte[e + 1], o[2] = plte[e + 2];
            } else if (channels <= 2) {
                o[0] = o[1] = o[2] = px[0];
            } else {
                o[0] = px[0], o[1] = px[1], o[2] = px[2];
            }
        }
        prev.swap(cur);
    }
    return true;
}
#endif
} // namespace rtr

inline image_texture::image_texture(const char* filename) {
    std::vector<unsigned char> file;
    bool ok = rtr::read_file(filename, file) && rtr::decode_ppm(file, data, width, height);
#if RTR_HAVE_ZLIB
    if (!ok && !file.empty()) ok = rtr::decode_png(file, data, width, height);
#endif
    if (!ok) {
        data.clear();
        width = height = 0;
        std::cerr << "ERROR: Could not load texture image file '" << (filename ? filename : "") << "'.\n";
    }
}
inline int image_texture::rtr_flatten(rtr::Flattener& f) const {
    rtr_texture& r = f.cur(this);
    r.type = RTR_TEX_IMAGE;
    r.a = -1;
    if (!data.empty()) {
        rtr_image d{};
        d.width = width, d.height = height, d.offset = f.out.image_bytes.size();
        f.out.image_bytes.insert(f.out.image_bytes.end(), data.begin(), data.end());
        f.cur(this).a = (int)f.out.images.size();
        f.out.images.push_back(d);
    }
    return 0;
}

inline int lambertian::rtr_flatten(rtr::Flattener& f) const {
    const int t = f.tex(albedo.get());
    rtr_material& r = f.cur(this);
    r.type = RTR_MAT_LAMBERTIAN, r.tex[0] = t;
    return 0;
}
inline int metal::rtr_flatten(rtr::Flattener& f) const {
    rtr_material& r = f.cur(this);
    r.type = RTR_MAT_METAL;
    rtr::Flattener::put3(r.f, albedo);
    r.f[3] = fuzz;
    return 0;
}
inline int dielectric::rtr_flatten(rtr::Flattener& f) const {
    rtr_material& r = f.cur(this);
    r.type = RTR_MAT_DIELECTRIC, r.f[0] = ir;
    return 0;
}
inline int diffuse_light::rtr_flatten(rtr::Flattener& f) const {
    const int t = f.tex(emit.get());
    rtr_material& r = f.cur(this);
    r.type = RTR_MAT_DIFFUSE_LIGHT, r.tex[0] = t;
    return 0;
}
inline int PBRMaterial::rtr_flatten(rtr::Flattener& f) const {
    const int t0 = f.tex(albedo.get()), t1 = f.tex(roughness.get()), t2 = f.tex(metallic.get()),
              t3 = f.tex(normal_map.get());
    rtr_material& r = f.cur(this);
    r.type = RTR_MAT_PBR;
    r.tex[0] = t0, r.tex[1] = t1, r.tex[2] = t2, r.tex[3] = t3;
    return 0;
}
inline int isotropic::rtr_flatten(rtr::Flattener& f) const {
    const int t = f.tex(albedo.get());
    rtr_material& r = f.cur(this);
    r.type = RTR_MAT_ISOTROPIC, r.tex[0] = t;
    return 0;
}

inline int bvh_node::rtr_flatten(rtr::Flattener& f) const {
    const int l = f.node(left.get());
    const int r_ = f.node(right.get());
    rtr_node& r = f.cur(this);
    r.type = RTR_NODE_BVH;
    r.a = l, r.b = r_;
    rtr::Flattener::put3(r.f, box.minimum);
    rtr::Flattener::put3(r.f + 3, box.maximum);
    return 0;
}
inline int hittable_list::rtr_flatten(rtr::Flattener& f) const {
    f.list(this, *this);
    return 0;
}
inline int box::rtr_flatten(rtr::Flattener& f) const { /* box::hit == sides.hit (box.h:49-51) */
    f.list(this, sides);
    return 0;
}
inline int translate::rtr_flatten(rtr::Flattener& f) const {
    const int c = f.node(ptr.get());
    rtr_node& r = f.cur(this);
    r.type = RTR_NODE_TRANSLATE, r.a = c;
    rtr::Flattener::put3(r.f, offset);
    return 0;
}
inline int rotate_y::rtr_flatten(rtr::Flattener& f) const {
    const int c = f.node(ptr.get());
    rtr_node& r = f.cur(this);
    r.type = RTR_NODE_ROTATE_Y, r.a = c;
    r.f[0] = sin_theta, r.f[1] = cos_theta;
    return 0;
}
inline int flip_face::rtr_flatten(rtr::Flattener& f) const {
    const int c = f.node(ptr.get());
    rtr_node& r = f.cur(this);
    r.type = RTR_NODE_FLIP_FACE, r.a = c;
    return 0;
}
inline int constant_medium::rtr_flatten(rtr::Flattener& f) const {
    const int m = f.mat(phase_function.get());
    const int b = f.node(boundary.get());
    rtr_node& r = f.cur(this);
    r.type = RTR_NODE_MEDIUM, r.a = b, r.b = m;
    r.f[0] = neg_inv_density;
    return 0;
}
inline int sphere::rtr_flatten(rtr::Flattener& f) const {
    const int m = f.mat(mat_ptr.get());
    rtr_node& r = f.cur(this);
    r.type = RTR_NODE_SPHERE, r.a = m;
    rtr::Flattener::put3(r.f, center);
    r.f[3] = radius;
    return 0;
}
inline int moving_sphere::rtr_flatten(rtr::Flattener& f) const {
    const int m = f.mat(mat_ptr.get());
    rtr_node& r = f.cur(this);
    r.type = RTR_NODE_MOVING_SPHERE, r.a = m;
    rtr::Flattener::put3(r.f, center0);
    rtr::Flattener::put3(r.f + 3, center1);
    r.f[6] = time0, r.f[7] = time1, r.f[8] = radius;
    return 0;
}
inline int xy_rect::rtr_flatten(rtr::Flattener& f) const {
    const int m = f.mat(mp.get());
    rtr_node& r = f.cur(this);
    r.type = RTR_NODE_XY_RECT, r.a = m;
    r.f[0] = x0, r.f[1] = x1, r.f[2] = y0, r.f[3] = y1, r.f[4] = k;
    return 0;
}
inline int xz_rect::rtr_flatten(rtr::Flattener& f) const {
    const int m = f.mat(mp.get());
    rtr_node& r = f.cur(this);
    r.type = RTR_NODE_XZ_RECT, r.a = m;
    r.f[0] = x0, r.f[1] = x1, r.f[2] = z0, r.f[3] = z1, r.f[4] = k;
    return 0;
}
inline int yz_rect::rtr_flatten(rtr::Flattener& f) const {
    const int m = f.mat(mp.get());
    rtr_node& r = f.cur(this);
    r.type = RTR_NODE_YZ_RECT, r.a = m;
    r.f[0] = y0, r.f[1] = y1, r.f[2] = z0, r.f[3] = z1, r.f[4] = k;
    return 0;
}

#endif /* RTR_SCENE_API_H */
