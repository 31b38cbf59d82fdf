// C-ABI entry points that never touch the GPU: error reporting, version, the XML front end.
// (Device entry points live in ../device/api_device.hip.)
#include "host_scene.h"
#include "api_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace lj {
thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
} // namespace lj

struct lj_host_scene { lj::HostScene *hs; };

extern "C" {

const char *lj_last_error(void) { return lj::g_last_error.c_str(); }
const char *lj_version(void) { return "lajolla_hip 0.1 (gfx950)"; }

int lj_parse_scene(const char *xml_path, lj_host_scene **out) {
    return lj::guard([&]() {
        if (!xml_path || !out) throw lj::LjError(LJ_ERR_INVALID_ARG, "lj_parse_scene: null argument");
        *out = nullptr;
        lj::HostScene *hs = lj::parse_scene_xml(xml_path);
        *out = new lj_host_scene{hs};
    });
}

const LjSceneDesc *lj_host_scene_desc(const lj_host_scene *hs) { return hs ? &hs->hs->desc : nullptr; }

void lj_host_scene_free(lj_host_scene *hs) {
    if (!hs) return;
    delete hs->hs;
    delete hs;
}

int lj_image_write(const char *filename, int32_t width, int32_t height, const float *rgb) {
    return lj::guard([&]() {
        if (!filename || !rgb) throw lj::LjError(LJ_ERR_INVALID_ARG, "lj_image_write: null argument");
        lj::write_image(filename, width, height, rgb);
    });
}

int lj_image_read(const char *filename, int32_t channels, int32_t *width, int32_t *height, float **data) {
    return lj::guard([&]() {
        if (!filename || !width || !height || !data || (channels != 1 && channels != 3)) throw lj::LjError(LJ_ERR_INVALID_ARG, "lj_image_read: null argument or channels not 1 / 3");
        *data = nullptr;
        lj::HostImage img = lj::read_image(filename, channels);
        float *out = (float *)malloc(std::max<size_t>(img.data.size(), 1) * sizeof(float));
        if (!out) throw std::bad_alloc();
        memcpy(out, img.data.data(), img.data.size() * sizeof(float));
        *width = img.width; *height = img.height; *data = out;
    });
}

void lj_image_free(float *data) { free(data); }

} // extern "C"
