/*
 * rtr_renderer.h -- host C++ mirror of the reference's render entry:
 *   RenderBuffer  (renderer/render_buffer.h:11-33)
 *   Integrator + the two integrators on the device path (renderer/integrator.h:9-21,
 *                 mis_path_integrator.h, rr_path_integrator.h) as id-carrying tags
 *   Renderer      (renderer/renderer.h:17-142): same methods, same blocking render() signature
 * render() flattens the object graph (rtr_scene_api.h) and drives the HIP library through the
 * C ABI of include/rtr_hip.h; it then applies the reference's own output stage
 * (renderer.h:126-140: scale by 1/spp happened on the device, here sqrt + clamp).
 * Link with -lrtr_hip (ray_tracing-rendering_amd/librtr_hip.so).  Errors the reference would
 * print to std::cerr are printed to std::cerr; render() never throws (renderer.h has no error
 * channel), last_status()/last_error() expose what the C ABI reported.
 */
#ifndef RTR_RENDERER_H
#define RTR_RENDERER_H

#include "rtr_scene_api.h"

#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>

#include <zlib.h>

class RenderBuffer {
  public:
    RenderBuffer(int width, int height) : m_width(width), m_height(height) {
        m_pixels.resize(height, std::vector<color>(width));
    }
    void set_pixel(int x, int y, const color& c) {
        if (x >= 0 && x < m_width && y >= 0 && y < m_height) m_pixels[y][x] = c;
    }
    const std::vector<std::vector<color>>& get_data() const { return m_pixels; }
    int width() const { return m_width; }
    int height() const { return m_height; }
    /* write_color_to_buffer (renderer.h:126-140) for rows [y0, y1) of linear mean radiance (the device already
     * applied scale = 1 / samples): sqrt gamma, clamp to [0, 1]; `lin` holds (y1 - y0) rows of `stride` pixels */
    void store_linear_rows(const double* lin, int y0, int y1, int stride) {
        for (int j = y0; j < y1; ++j)
            for (int i = 0; i < m_width; ++i) {
                const double* px = &lin[((size_t)(j - y0) * stride + i) * 3];
                set_pixel(i, j, color(clamp(sqrt(px[0]), 0.0, 1.0), clamp(sqrt(px[1]), 0.0, 1.0), clamp(sqrt(px[2]), 0.0, 1.0)));
            }
    }
    /* the bytes save_to_png hands to its encoder (render_buffer.h:36-51): Y flipped, uchar(c * 255) truncation */
    std::vector<unsigned char> to_rgb8() const {
        std::vector<unsigned char> out((size_t)m_width * m_height * 3);
        for (int j = 0; j < m_height; ++j)
            for (int i = 0; i < m_width; ++i) {
                const color& p = m_pixels[m_height - 1 - j][i];
                unsigned char* px = &out[((size_t)j * m_width + i) * 3];
                px[0] = static_cast<unsigned char>(p[0] * 255);
                px[1] = static_cast<unsigned char>(p[1] * 255);
                px[2] = static_cast<unsigned char>(p[2] * 255);
            }
        return out;
    }
    /* write_color_to_buffer for the pixels of one packed 16x16 tile (include/rtr_hip.h: rtr_render_tiles_host) that lie
     * inside rows [y0, y1): tile origin (tx0, ty0), `px` = 16 rows of 16 pixels of 3 doubles, lowest row first */
    void store_linear_tile(const double* px, int tx0, int ty0, int y0, int y1) {
        for (int r = 0; r < 16; ++r) {
            const int j = ty0 + r;
            if (j < y0 || j >= y1 || j >= m_height) continue;
            for (int q = 0; q < 16; ++q) {
                const int i = tx0 + q;
                if (i >= m_width) break;
                const double* v = &px[(r * 16 + q) * 3];
                m_pixels[j][i] = color(clamp(sqrt(v[0]), 0.0, 1.0), clamp(sqrt(v[1]), 0.0, 1.0), clamp(sqrt(v[2]), 0.0, 1.0));
            }
        }
    }
    /* render_buffer.h:35-55: the bytes of to_rgb8() as an 8-bit RGB PNG.  The reference hands them to stb_image_write;
     * here a plain encoder (filter 0 on every row, one zlib stream): another compressed byte stream, the same pixels
     * (tests/test_output_stage.py decodes the file and compares them with the reference's own PNG). */
    bool save_to_png(const std::string& filename) const {
        const std::vector<unsigned char> rgb = to_rgb8();
        std::vector<unsigned char> raw((size_t)m_height * (1 + (size_t)m_width * 3));
        for (int j = 0; j < m_height; ++j) {
            raw[(size_t)j * (1 + (size_t)m_width * 3)] = 0;
            std::memcpy(&raw[(size_t)j * (1 + (size_t)m_width * 3) + 1], &rgb[(size_t)j * m_width * 3], (size_t)m_width * 3);
        }
        uLongf zlen = compressBound((uLong)raw.size());
        std::vector<unsigned char> z(zlen);
        if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
        FILE* f = std::fopen(filename.c_str(), "wb");
        if (!f) return false;
        auto be32 = [](unsigned char* p, unsigned long v) { p[0] = v >> 24, p[1] = v >> 16, p[2] = v >> 8, p[3] = v; };
        bool ok = true;
        auto chunk = [&](const char* type, const unsigned char* data, size_t len) {
            unsigned char head[8];
            be32(head, (unsigned long)len);
            std::memcpy(head + 4, type, 4);
            unsigned long crc = crc32(0L, head + 4, 4);
            if (len) crc = crc32(crc, data, (uInt)len);
            unsigned char tail[4];
            be32(tail, crc);
            ok = ok && std::fwrite(head, 1, 8, f) == 8 && (len == 0 || std::fwrite(data, 1, len, f) == len) &&
                 std::fwrite(tail, 1, 4, f) == 4;
        };
        static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
        ok = std::fwrite(sig, 1, 8, f) == 8;
        unsigned char ihdr[13];
        be32(ihdr, (unsigned long)m_width), be32(ihdr + 4, (unsigned long)m_height);
        ihdr[8] = 8, ihdr[9] = 2, ihdr[10] = 0, ihdr[11] = 0, ihdr[12] = 0; /* 8 bits, RGB, deflate, adaptive filters, no interlace */
        chunk("IHDR", ihdr, 13);
        chunk("IDAT", z.data(), (size_t)zlen);
        chunk("IEND", nullptr, 0);
        return std::fclose(f) == 0 && ok;
    }
    /* render_buffer.h:58-80 writes a JPEG through stb_image_write's encoder, which is outside this path:
     * the method exists so the reference's callers compile; it reports failure like a file that cannot be written. */
    bool save_to_jpg(const std::string& filename, int quality = 90) const {
        (void)quality;
        std::cerr << "save_to_jpg(" << filename << "): no JPEG encoder in this build; use save_to_png\n";
        return false;
    }
    /* binary PPM of those bytes */
    bool save_to_ppm(const std::string& filename) const {
        FILE* f = std::fopen(filename.c_str(), "wb");
        if (!f) return false;
        const std::vector<unsigned char> rgb = to_rgb8();
        std::fprintf(f, "P6\n%d %d\n255\n", m_width, m_height);
        const bool ok = std::fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
        std::fclose(f);
        return ok;
    }

  private:
    int m_width, m_height;
    std::vector<std::vector<color>> m_pixels;
};

class Integrator {
  public:
    virtual ~Integrator() = default;
    virtual void set_max_depth(int depth) = 0;
    virtual int rtr_integrator_id() const = 0; /* reference CLI numbering, main.cpp:52 */
    virtual int rtr_max_depth() const = 0;
    virtual int rtr_rr_start() const = 0;
};
class MISPathIntegrator : public Integrator {
  public:
    void set_max_depth(int depth = 50) override { m_max_depth = depth; }
    void set_rr_start_depth(int depth) { m_rr_start_depth = depth; }
    int rtr_integrator_id() const override { return RTR_INTEGRATOR_MIS; }
    int rtr_max_depth() const override { return m_max_depth; }
    int rtr_rr_start() const override { return m_rr_start_depth; }

  private:
    int m_max_depth = 50, m_rr_start_depth = 3;
};
class RRPathInterator : public Integrator { /* sic: the reference's spelling */
  public:
    void set_max_depth(int depth = 50) override { m_max_depth = depth; }
    void set_rr_start_depth(int depth) { m_rr_start_depth = depth; }
    int rtr_integrator_id() const override { return RTR_INTEGRATOR_RR; }
    int rtr_max_depth() const override { return m_max_depth; }
    int rtr_rr_start() const override { return m_rr_start_depth; }

  private:
    int m_max_depth = 50, m_rr_start_depth = 3;
};

/* the other three integrators of the reference CLI (main.cpp:72-76), megakernel only */
class PathIntegrator : public Integrator { /* renderer/path_integrator.h */
  public:
    void set_max_depth(int depth = 50) override { m_max_depth = depth; }
    int rtr_integrator_id() const override { return RTR_INTEGRATOR_PATH; }
    int rtr_max_depth() const override { return m_max_depth; }
    int rtr_rr_start() const override { return 3; }

  private:
    int m_max_depth = 50;
};
class PBRPathIntegrator : public Integrator { /* renderer/pbr_path_integrator.h */
  public:
    void set_max_depth(int depth = 50) override { m_max_depth = depth; }
    void set_rr_start_depth(int depth) { m_rr_start_depth = depth; }
    int rtr_integrator_id() const override { return RTR_INTEGRATOR_PBR; }
    int rtr_max_depth() const override { return m_max_depth; }
    int rtr_rr_start() const override { return m_rr_start_depth; }

  private:
    int m_max_depth = 50, m_rr_start_depth = 3;
};
class DirectLightIntegrator : public Integrator { /* renderer/direct_light_integrator.h */
  public:
    void set_max_depth(int depth = 50) override { m_max_depth = depth; }
    void set_rr_start_depth(int depth) { m_rr_start_depth = depth; }
    int rtr_integrator_id() const override { return RTR_INTEGRATOR_NEE; }
    int rtr_max_depth() const override { return m_max_depth; }
    int rtr_rr_start() const override { return m_rr_start_depth; }

  private:
    int m_max_depth = 50, m_rr_start_depth = 3;
};

namespace rtr {
/* 16x16 tiles in the reference's dispatch numbering (renderer.h:40-44,61-62): tile 0 is the top-left one */
inline int tile_index_of_pixel(int W, int H, int i, int j) {
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    return ((tiles_y - 1) - j / 16) * tiles_x + i / 16;
}
/* which of `n` workers renders pixel (i, j): tiles are dealt round-robin, like index % tile_stride == tile_first
 * of rtr_render_params */
inline int tile_owner(int W, int H, int i, int j, int n) { return n > 1 ? tile_index_of_pixel(W, H, i, j) % n : 0; }
} // namespace rtr

class Renderer {
  public:
    struct Settings {
        int samples_per_pixel = 10;
    };
    /* one GPU */
    explicit Renderer(int device = 0) : Renderer(std::vector<int>{device}) {}
    /* One context and one host thread per entry, like the reference's one worker per hardware thread
     * (renderer.h:48-94): image tiles are dealt round-robin to the contexts (no exchange between them), each
     * fills its tiles of the one RenderBuffer.  An ordinal may repeat (two contexts on one GPU). */
    explicit Renderer(const std::vector<int>& devices) : m_is_rendering(false) {
        for (int d : devices) {
            rtr_context* c = nullptr;
            const int rc = rtr_create(d, &c);
            if (rc != RTR_OK) { /* render() refuses: an image from fewer GPUs than asked for would pass for the real one */
                m_create_status = rc;
                m_create_error = std::string("rtr_create(") + std::to_string(d) + "): " + rtr_last_error(nullptr);
                std::cerr << m_create_error << "\n";
                continue;
            }
            m_ctx.push_back(c);
        }
    }
    /* every GPU the process sees */
    static std::vector<int> all_devices() {
        std::vector<int> d;
        for (int k = 0; k < rtr_device_count(); ++k) d.push_back(k);
        return d;
    }
    ~Renderer() {
        for (rtr_context* c : m_ctx) rtr_destroy(c);
    }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    void set_integrator(std::shared_ptr<Integrator> integrator) { m_integrator = integrator; }
    void set_samples(int samples) { m_settings.samples_per_pixel = samples; }
    void set_max_depth(int depth) {
        if (m_integrator) m_integrator->set_max_depth(depth);
    }
    void set_seed(uint32_t seed) { m_seed = seed; } /* the reference has no seed control */
    /* The reference's workers store pixels as they finish tiles and main.cpp:124 polls
     * RenderBuffer::get_data() meanwhile.  Here every context renders the image in `n` horizontal bands of whole
     * tile rows, top band first like the reference's tile order (renderer.h:61-62), and stores the tiles it owns
     * as soon as they arrive; cancel() takes effect inside and between bands.  0 = pick by image
     * height (one band per 256 rows).  Per-sample seeds depend on (pixel, sample) only, so the
     * result does not depend on n. */
    void set_progress_bands(int n) { m_bands = n; }
    void cancel() {
        m_is_rendering = false;
        for (rtr_context* c : m_ctx) rtr_cancel(c);
    }
    bool is_rendering() const { return m_is_rendering; }
    int last_status() const { return m_status; }
    const char* last_error() const { return m_error.c_str(); }
    double last_seconds() const { return m_seconds; }
    int device_contexts() const { return (int)m_ctx.size(); }
    /* The flattened scene stays on the GPUs between render() calls with the same world / camera / lights objects
     * and background (the reference's scene graph is immutable once built; the Renderer holds the objects, so an
     * address cannot come back as another scene); call this after changing a scene object in place. */
    void invalidate_scene() { m_scene_valid = false; }
    int scene_uploads() const { return m_scene_uploads; }

    void render(shared_ptr<hittable> world, shared_ptr<camera> cam, const color& background,
                RenderBuffer& target_buffer, const std::vector<shared_ptr<Light>>& lights = {}) {
        m_is_rendering = true;
        const auto t0 = std::chrono::high_resolution_clock::now();
        /* the cache below compares object identities: holding the objects keeps their addresses from being reused */
        const bool same_scene = world == m_world && cam == m_cam && lights == m_lights &&
                                background[0] == m_scene_bg[0] && background[1] == m_scene_bg[1] && background[2] == m_scene_bg[2];
        m_status = render_impl(*world, *cam, background, target_buffer, lights, same_scene && m_scene_valid);
        if (m_scene_valid) m_world = world, m_cam = cam, m_lights = lights;
        m_seconds = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        m_is_rendering = false;
        if (m_status == RTR_OK)
            std::cout << "Rendering finished in " << m_seconds << " seconds." << std::endl;
        else
            std::cerr << "render failed (" << m_status << "): " << m_error << "\n";
    }

  private:
    int render_impl(const hittable& world, const camera& cam, const color& background, RenderBuffer& buf,
                    const std::vector<shared_ptr<Light>>& lights, bool scene_on_device) {
        if (m_create_status != RTR_OK) return m_error = m_create_error, m_create_status;
        if (m_ctx.empty()) return m_error = rtr_last_error(nullptr), RTR_ERR_DEVICE;
        if (!m_integrator) return m_error = "no integrator set", RTR_ERR_INVALID;
        const int n = (int)m_ctx.size();
        /* flatten + upload once per scene, not once per render() call */
        if (!scene_on_device) {
            m_scene_valid = false;
            rtr_scene_storage st;
            if (!rtr::flatten(world, lights, cam, background, st, m_error)) return RTR_ERR_UNSUPPORTED;
            rtr_scene_desc d = st.desc();
            for (rtr_context* c : m_ctx) {
                const int rc = rtr_upload_scene(c, &d);
                if (rc) return m_error = rtr_last_error(c), rc;
            }
            m_scene_bg[0] = background[0], m_scene_bg[1] = background[1], m_scene_bg[2] = background[2];
            m_scene_valid = true;
            ++m_scene_uploads;
        }
        const int W = buf.width(), H = buf.height();
        rtr_render_params p{};
        p.image_width = W, p.image_height = H;
        p.x0 = 0, p.y0 = 0, p.x1 = W, p.y1 = H;
        p.spp = m_settings.samples_per_pixel;
        p.max_depth = m_integrator->rtr_max_depth();
        p.rr_start_depth = m_integrator->rtr_rr_start();
        p.integrator = m_integrator->rtr_integrator_id();
        p.seed = m_seed;
        p.pipeline = RTR_PIPELINE_AUTO;
        p.spp_chunks = 0;
        int bands = m_bands > 0 ? m_bands : (H + 255) / 256;
        const int tile_rows = (H + 15) / 16, tiles_x = (W + 15) / 16;
        bands = std::max(1, std::min(bands, tile_rows));
        /* One worker per context, like the reference's one worker per hardware thread (renderer.h:48-94): worker k
         * walks the bands top to bottom on its own, renders the tiles index % n == k of each (nothing is uploaded, the
         * tiles it owns come back packed through pinned memory: rtr_render_tiles_host) and stores them into the
         * RenderBuffer itself -- distinct pixels, so no lock and no join per band; the polling UI (main.cpp:124) sees
         * tiles appear as they finish.  After a cancel the finished tiles of the band are stored, the others keep their
         * previous pixels (renderer.h:52-59). */
        std::vector<int> rcs(n, RTR_OK);
        std::vector<std::string> errs(n);
        auto work = [&](int k) {
            rtr_render_params q = p;
            q.tile_first = k, q.tile_stride = n;
            for (int b = 0; b < bands; ++b) { /* row 0 of the buffer is the bottom row; the top band goes first */
                const int r1 = tile_rows - (int)((long long)b * tile_rows / bands);
                const int r0 = tile_rows - (int)((long long)(b + 1) * tile_rows / bands);
                const int y0 = r0 * 16, y1 = std::min(H, r1 * 16);
                if (y0 >= y1) continue;
                if (!m_is_rendering) {
                    rcs[k] = RTR_ERR_CANCELLED, errs[k] = "render cancelled";
                    return;
                }
                q.y0 = y0, q.y1 = y1;
                const double* tiles = nullptr;
                const int32_t* ids = nullptr;
                const uint8_t* done = nullptr;
                int64_t n_tiles = 0;
                const int rc = rtr_render_tiles_host(m_ctx[k], &q, &tiles, &ids, &done, &n_tiles);
                if (rc != RTR_OK && rc != RTR_ERR_CANCELLED) {
                    rcs[k] = rc, errs[k] = rtr_last_error(m_ctx[k]);
                    return;
                }
                for (int64_t t = 0; t < n_tiles; ++t) {
                    if (!done[t]) continue;
                    const int ty = (tile_rows - 1) - ids[t] / tiles_x, tx = ids[t] % tiles_x; /* renderer.h:61-62 */
                    buf.store_linear_tile(tiles + t * 768, tx * 16, ty * 16, y0, y1);
                }
                if (rc == RTR_ERR_CANCELLED) {
                    rcs[k] = rc, errs[k] = "render cancelled";
                    return;
                }
            }
        };
        if (n == 1) {
            work(0);
        } else {
            std::vector<std::thread> th;
            for (int k = 0; k < n; ++k) th.emplace_back(work, k);
            for (auto& t : th) t.join();
        }
        for (int k = 0; k < n; ++k)
            if (rcs[k]) return m_error = errs[k], rcs[k];
        return RTR_OK;
    }

    Settings m_settings;
    std::atomic<bool> m_is_rendering;
    std::shared_ptr<Integrator> m_integrator;
    std::vector<rtr_context*> m_ctx;
    shared_ptr<hittable> m_world; /* the scene whose flattened form is on the GPUs */
    shared_ptr<camera> m_cam;
    std::vector<shared_ptr<Light>> m_lights;
    double m_scene_bg[3] = {0, 0, 0};
    bool m_scene_valid = false;
    int m_scene_uploads = 0;
    int m_status = RTR_OK;
    int m_create_status = RTR_OK; /* a context that could not be created: render() refuses */
    std::string m_create_error;
    uint32_t m_seed = 1;
    int m_bands = 0;
    double m_seconds = 0;
    std::string m_error;
};

#endif /* RTR_RENDERER_H */
