/* lajolla_hip.h — C ABI of liblajolla_hip.so: the MI355X (gfx950) drop-in for lajolla's
 * per-pixel-sample hot path.
 *
 * The reference has no FFI layer.  Its de-facto boundary is
 *     Image3 render(const Scene &scene)                      (src/render.h:9, src/render.cpp:155-170)
 * fed by
 *     Scene parse_scene(const fs::path&, const RTCDevice&)   (src/parse_scene.h:9)
 * and, one level down, the two Embree-backed queries
 *     intersect(scene, ray, ray_diff) / occluded(scene, ray) (src/intersection.h:40-47).
 * Each entry point below names the reference interface it replaces.  Everything that crosses this
 * line is a plain pointer, size or POD struct: no C++ types, no torch types, no exceptions.
 *
 * Variant model.  The reference's std::variant alternatives (Shape shape.h:53, Material
 * material.h:102-110, Light light.h:34, Texture texture.h:108, Filter filter.h:45) become tagged PODs
 * with the same alternative order and the same field names, so a maintainer can fill an LjSceneDesc
 * from a reference `Scene` field by field.
 *
 * Numeric convention.  Host-side scene data is double (the reference's Real, lajolla.h:23); the
 * device narrows to float exactly where the reference narrows for Embree (intersection.cpp:15-24,
 * triangle_mesh.inl:11-14) and shades in float (tolerance stated in DESIGN.md).
 */
#ifndef LAJOLLA_HIP_H
#define LAJOLLA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- error codes (flexception.h:8-24 throws; we return) */
enum {
    LJ_OK = 0,
    LJ_ERR_INVALID_ARG = -1,
    LJ_ERR_PARSE = -2,       /* Error("...") sites of parse_scene.cpp / parse_obj.cpp / load_serialized.cpp */
    LJ_ERR_IO = -3,
    LJ_ERR_UNSUPPORTED = -4, /* a variant alternative / integrator the device path does not implement yet: loud, never a CPU fallback */
    LJ_ERR_DEVICE = -5,      /* HIP runtime error or no gfx950 device */
    LJ_ERR_INTERNAL = -6
};
/* Message for the last failing call on this thread ("" if none). */
const char *lj_last_error(void);
const char *lj_version(void);

/* ---------------------------------------------------------------- variant tags (alternative order == reference order) */
enum { LJ_FILTER_BOX = 0, LJ_FILTER_TENT = 1, LJ_FILTER_GAUSSIAN = 2 };                 /* filter.h:45 */
enum { LJ_SHAPE_SPHERE = 0, LJ_SHAPE_TRIMESH = 1 };                                      /* shape.h:53 */
enum { LJ_TEX_CONSTANT = 0, LJ_TEX_IMAGE = 1, LJ_TEX_CHECKERBOARD = 2 };                 /* texture.h:108 */
enum { LJ_LIGHT_AREA = 0, LJ_LIGHT_ENVMAP = 1 };                                         /* light.h:34 */
enum {                                                                                   /* material.h:102-110 */
    LJ_MAT_LAMBERTIAN = 0, LJ_MAT_ROUGHPLASTIC = 1, LJ_MAT_ROUGHDIELECTRIC = 2,
    LJ_MAT_DISNEYDIFFUSE = 3, LJ_MAT_DISNEYMETAL = 4, LJ_MAT_DISNEYGLASS = 5,
    LJ_MAT_DISNEYCLEARCOAT = 6, LJ_MAT_DISNEYSHEEN = 7, LJ_MAT_DISNEYBSDF = 8
};
enum {                                                                                   /* scene.h:14-22 */
    LJ_INTEGRATOR_DEPTH = 0, LJ_INTEGRATOR_SHADING_NORMAL = 1, LJ_INTEGRATOR_MEAN_CURVATURE = 2,
    LJ_INTEGRATOR_RAY_DIFFERENTIAL = 3, LJ_INTEGRATOR_MIPMAP_LEVEL = 4, LJ_INTEGRATOR_PATH = 5,
    LJ_INTEGRATOR_VOLPATH = 6
};

/* Texture<T> (texture.h:76-92).  A Texture<Real> uses value[0]/color1[0] only. */
typedef struct LjTexture {
    int32_t kind;        /* LJ_TEX_* */
    int32_t texture_id;  /* ImageTexture::texture_id: index into LjSceneDesc.images (3-channel) or images1 (1-channel) */
    double value[3];     /* ConstantTexture::value, or CheckerboardTexture::color0 */
    double color1[3];    /* CheckerboardTexture::color1 */
    double uscale, vscale, uoffset, voffset;
} LjTexture;

/* Material (material.h:12-98).  Texture slots, in the reference's field order:
 *   LAMBERTIAN       0 reflectance
 *   ROUGHPLASTIC     0 diffuse_reflectance  1 specular_reflectance   2 roughness        (+eta)
 *   ROUGHDIELECTRIC  0 specular_reflectance 1 specular_transmittance 2 roughness        (+eta)
 *   DISNEYDIFFUSE    0 base_color 1 roughness 2 subsurface
 *   DISNEYMETAL      0 base_color 1 roughness 2 anisotropic
 *   DISNEYGLASS      0 base_color 1 roughness 2 anisotropic                              (+eta)
 *   DISNEYCLEARCOAT  0 clearcoat_gloss
 *   DISNEYSHEEN      0 base_color 1 sheen_tint
 *   DISNEYBSDF       0 base_color 1 specular_transmission 2 metallic 3 subsurface 4 specular 5 roughness
 *                    6 specular_tint 7 anisotropic 8 sheen 9 sheen_tint 10 clearcoat 11 clearcoat_gloss (+eta) */
#define LJ_MAX_TEX_SLOTS 12
typedef struct LjMaterial {
    int32_t kind;  /* LJ_MAT_* */
    int32_t n_tex;
    double eta;    /* internal IOR / external IOR where the alternative has one */
    LjTexture tex[LJ_MAX_TEX_SLOTS];
} LjMaterial;

/* Shape = variant<Sphere, TriangleMesh> (shape.h:26-53).  Mesh arrays live in the scene-wide pools below;
 * indices are mesh-local like the reference's. */
typedef struct LjShape {
    int32_t kind;  /* LJ_SHAPE_* */
    int32_t material_id, area_light_id, interior_medium_id, exterior_medium_id;  /* ShapeBase, -1 = none */
    int32_t has_normals, has_uvs;   /* TriangleMesh::normals.size()>0 / uvs.size()>0 */
    int32_t _pad;
    int64_t first_vertex, n_vertices;     /* into positions / normals / uvs */
    int64_t first_triangle, n_triangles;  /* into indices */
    double position[3];  /* Sphere::position */
    double radius;       /* Sphere::radius */
} LjShape;

/* Light = variant<DiffuseAreaLight, Envmap> (light.h:15-34). */
typedef struct LjLight {
    int32_t kind;      /* LJ_LIGHT_* */
    int32_t shape_id;  /* DiffuseAreaLight::shape_id */
    double intensity[3];
    LjTexture values;  /* Envmap::values */
    double to_world[16], to_local[16];  /* row-major Matrix4x4 (matrix.h) */
    double scale;
} LjLight;

/* Level 0 of a TexturePool image (texture.h:13-19), as the decoders return it (float, image.cpp:44,96):
 * row-major, y=0 at the top, `channels` interleaved.  Mip chains (mipmap.h:25-48) are rebuilt by the consumer. */
typedef struct LjImage {
    int32_t width, height, channels, _pad;
    const float *data;
} LjImage;

/* Camera (camera.h:10-24) + RenderOptions (scene.h:24-31). */
typedef struct LjCamera {
    double cam_to_world[16], world_to_cam[16], sample_to_cam[16], cam_to_sample[16];
    int32_t width, height;
    int32_t filter_kind;  /* LJ_FILTER_* */
    int32_t medium_id;
    double filter_param;  /* Box::width / Tent::width / Gaussian::stddev */
} LjCamera;

/* VolumeSpectrum (volume.h:13-30): a constant, or a grid of RGB values interpolated trilinearly inside [p_min, p_max]. */
typedef enum LjVolumeKind { LJ_VOLUME_CONSTANT = 0, LJ_VOLUME_GRID = 1 } LjVolumeKind;
typedef struct LjVolume {
    int32_t kind;               /* LJ_VOLUME_* */
    int32_t resolution[3];      /* grid: x, y, z */
    double value[3];            /* constant (the medium's scale already applied, volume.h:105-107) */
    double p_min[3], p_max[3], max_data[3];
    double scale;               /* GridVolume::scale (set_scale, volume.h:109-111) */
    const float *data;          /* grid: 3 floats per voxel, x fastest; the file's float32 values (volume.cpp:62-98) */
} LjVolume;

/* Medium (medium.h:10-21) with its PhaseFunction (phase_function.h:10-17). */
typedef enum LjMediumKind { LJ_MEDIUM_HOMOGENEOUS = 0, LJ_MEDIUM_HETEROGENEOUS = 1 } LjMediumKind;
typedef enum LjPhaseKind { LJ_PHASE_ISOTROPIC = 0, LJ_PHASE_HG = 1 } LjPhaseKind;
typedef struct LjMedium {
    int32_t kind;               /* LJ_MEDIUM_* */
    int32_t phase_kind;         /* LJ_PHASE_* */
    double g;                   /* HenyeyGreenstein::g */
    double sigma_a[3], sigma_s[3];   /* homogeneous */
    LjVolume albedo, density;        /* heterogeneous */
} LjMedium;

typedef struct LjRenderOptions {
    int32_t integrator;  /* LJ_INTEGRATOR_* */
    int32_t samples_per_pixel, max_depth, rr_depth, vol_path_version, max_null_collisions;
} LjRenderOptions;

/* The flat mirror of the reference `Scene` *inputs* (scene.h:42-53 constructor arguments).  Derived tables —
 * bounds sphere, per-mesh triangle_sampler, envmap sampling_dist, light_dist (scene.cpp:30-52) — are NOT part of
 * the description: lj_scene_upload builds them, as Scene::Scene does. */
typedef struct LjSceneDesc {
    LjCamera camera;
    LjRenderOptions options;
    int32_t n_shapes, n_materials, n_lights, n_images3, n_images1, envmap_light_id;
    const LjShape *shapes;
    const LjMaterial *materials;
    const LjLight *lights;
    const LjImage *images3;  /* TexturePool::image3s level 0 */
    const LjImage *images1;  /* TexturePool::image1s level 0 */
    int64_t n_vertices, n_triangles;
    const double *positions;  /* 3 per vertex */
    const double *normals;    /* 3 per vertex; garbage where !has_normals */
    const double *uvs;        /* 2 per vertex; garbage where !has_uvs */
    const int32_t *indices;   /* 3 per triangle, mesh-local */
    const char *output_filename;
    int32_t n_media, _pad_media;
    const LjMedium *media;    /* Scene::media (scene.h:66); referenced by LjCamera::medium_id and LjShape::*_medium_id */
} LjSceneDesc;

/* ---------------------------------------------------------------- front end (host only)
 * Replaces parse_scene(path, embree_device) (parse_scene.cpp:1134-1149) up to, not including, Scene::Scene.
 * Same Mitsuba-0.x XML dialect, same std::stof number semantics, same material/shape/light/texture order. */
typedef struct lj_host_scene lj_host_scene;
int lj_parse_scene(const char *xml_path, lj_host_scene **out);
const LjSceneDesc *lj_host_scene_desc(const lj_host_scene *hs);
void lj_host_scene_free(lj_host_scene *hs);

/* ---------------------------------------------------------------- device side */
typedef struct lj_context lj_context;  /* one HIP device + stream + workspace; main.cpp:30 rtcNewDevice analogue */
typedef struct lj_scene lj_scene;      /* device-resident Scene: flattened BVH + tables; scene.cpp:3-53 analogue */

int lj_context_create(int device_id, lj_context **out);
/* Lifetime: scenes keep their context alive.  Destroying a context (or a device group, below) that still has scenes only marks it; the
 * last lj_scene_destroy (lj_group_scene_destroy) releases it — any destruction order is safe, every handle is destroyed exactly once. */
void lj_context_destroy(lj_context *ctx);

/* Replaces Scene::Scene (scene.cpp:3-53): builds the BVH (instead of rtcCommitScene), the bounds sphere,
 * sampling tables and the light table, and uploads everything.  The description may be freed afterwards. */
int lj_scene_upload(lj_context *ctx, const LjSceneDesc *desc, lj_scene **out);
void lj_scene_destroy(lj_scene *scene);

enum { LJ_RNG_SAMPLE = 0 /* one pcg32 stream per (pixel, sample): stream = (y*w + x)*spp + s */ };

typedef struct LjRenderArgs {
    int32_t spp;         /* <=0: RenderOptions::samples_per_pixel */
    int32_t max_depth;   /* INT32_MIN: RenderOptions::max_depth */
    int32_t rng_mode;    /* LJ_RNG_SAMPLE */
    int32_t rank, world_size; /* render only the 16x16 tiles t = ty*ntx+tx with t % world_size == rank (render.cpp:75-88);
                                 other pixels are written as 0 so a sum over ranks is the full image */
    int32_t crop_x0, crop_y0, crop_x1, crop_y1; /* all 0: full frame; else only pixels in [x0,x1)x[y0,y1) */
    uint32_t pool_paths; /* paths in flight (128 bytes of queue records each, allocated as far as a pass has samples); 0: default = 128 M */
    uint32_t flags;
    uint64_t seed;       /* 0: 0x853c49e6748fea9b (pcg.h:33) */
} LjRenderArgs;

/* Replaces `Image3 render(const Scene&)` (render.h:9; path_render render.cpp:71-101): fills rgb[h][w][3]
 * (row-major, y=0 top, image.h:28-34) with radiance / spp.  Blocking.  `rgb_host` is caller-owned. */
int lj_render(lj_scene *scene, const LjRenderArgs *args, float *rgb_host);
/* Same, but into caller-owned DEVICE memory, ordered on `hip_stream` (a hipStream_t; NULL: no ordering requested): the render
 * starts after everything `hip_stream` holds at the call and `hip_stream` waits for the finished frame before anything enqueued
 * on it later.  (The kernels themselves run on the context's own streams.)  Returns once the render has completed.  This is the
 * hand-off used for the RCCL framebuffer reduce. */
int lj_render_device(lj_scene *scene, const LjRenderArgs *args, float *rgb_device, void *hip_stream);

/* Per-sample radiance for the crop window: out[((y-y0)*(x1-x0) + (x-x0))*spp + s][3] — one path_tracing()
 * value (path_tracing.h:7-325) per entry, for path-by-path parity tests. */
int lj_render_samples(lj_scene *scene, const LjRenderArgs *args, float *radiance_host);

/* Batched replacements for intersect()/occluded() (intersection.cpp:7-85) on the flattened BVH, for parity tests
 * of the traversal alone.  Rays are float as Embree sees them (intersection.cpp:15-24). */
typedef struct LjRay { float org[3]; float tnear; float dir[3]; float tfar; } LjRay;
typedef struct LjHit { float t, u, v; int32_t shape_id, prim_id; } LjHit;  /* shape_id == -1: miss */
int lj_intersect(lj_scene *scene, int64_t n, const LjRay *rays_host, LjHit *hits_host);
int lj_occluded(lj_scene *scene, int64_t n, const LjRay *rays_host, uint8_t *occluded_host);

/* ---------------------------------------------------------------- per-object queries on the device, batched
 * The reference's free functions on Material / Light / Shape / Camera / Filter / pcg32, evaluated by the SAME float device
 * code the shade kernels run (device/dshade.h), one query per lane: what lets a test hold the HIP code value for value
 * against vectors produced by the reference's own functions (tests/golden) and re-run the reference's unit tests
 * (src/tests/materials.cpp, filter.cpp, frame.cpp, mipmap.cpp) on the GPU.  Host pointers in, host pointers out; blocking.
 *
 * `variant`: which compiled feature set of the device code answers — -1: the one lj_scene_upload chose for
 * this scene; 0..lj_shade_variant_count()-1: that one, LJ_ERR_INVALID_ARG if it does not cover what the scene holds. */
int lj_shade_variant_count(void);
int lj_scene_shade_variant(const lj_scene *scene);

/* The PathVertex fields a BSDF reads (intersection.h:15-35). */
typedef struct LjVertex {
    float position[3], geometry_normal[3];
    float frame_x[3], frame_y[3], frame_n[3];   /* shading_frame (frame.h:24-41) */
    float uv_screen_size, mean_curvature;
    double uv[2];                                /* double: tiled texture coordinates */
    int32_t material_id, light_id;               /* light_id: get_area_light_id(shape), -1 if not an emitter */
    int32_t shape_id, primitive_id;
} LjVertex;

/* eval(material, dir_in, dir_out, vertex, pool) (material.h:126), pdf_sample_bsdf (material.h:161) and
 * sample_bsdf(material, dir_in, vertex, pool, rnd_uv, rnd_w) (material.h:147), TransportDirection::TO_LIGHT. */
typedef struct LjBsdfQuery { LjVertex vertex; float dir_in[3], dir_out[3], rnd_uv[2], rnd_w; int32_t _pad; } LjBsdfQuery;
typedef struct LjBsdfResult { float eval[3], pdf; float sample_dir[3], sample_eta, sample_roughness; int32_t sample_valid; } LjBsdfResult;
int lj_bsdf_queries(lj_scene *scene, int variant, int64_t n, const LjBsdfQuery *queries_host, LjBsdfResult *results_host);

/* sample_point_on_light(light, ref, rnd_uv, rnd_w, scene), pdf_point_on_light of that point and
 * emission(light, view_dir, footprint, point, scene) (light.h:46-67); light_pmf(scene, id) (scene.cpp:77) beside them. */
typedef struct LjLightQuery { int32_t light_id; float ref[3], rnd_uv[2], rnd_w, view_dir[3]; } LjLightQuery;
typedef struct LjLightResult { float position[3], normal[3], pdf, emission[3], pmf; int32_t _pad; double position_d[3]; } LjLightResult;
int lj_light_queries(lj_scene *scene, int variant, int64_t n, const LjLightQuery *queries_host, LjLightResult *results_host);
/* sample_light(scene, u) (scene.cpp:73) */
int lj_sample_light_queries(lj_scene *scene, int64_t n, const float *u_host, int32_t *light_id_host);

/* compute_shading_info + the PathVertex assembly of intersect() (intersection.cpp:38-62, shape.h:92) for a hit record
 * (shape_id, primitive_id, t, u, v) of the ray (org, dir) whose differential has radius 0 and the given spread
 * (ray.h:27-42); `emission`: emission(vertex, -dir, scene) (path_tracing.h:58-61) when the shape is an emitter, else 0. */
typedef struct LjHitQuery { float org[3], dir[3], t, u, v, ray_spread; int32_t shape_id, primitive_id; } LjHitQuery;
typedef struct LjHitResult { LjVertex vertex; float emission[3]; int32_t _pad; } LjHitResult;
int lj_vertex_queries(lj_scene *scene, int variant, int64_t n, const LjHitQuery *queries_host, LjHitResult *results_host);

/* sample_primary(camera, screen_pos) (camera.cpp:23-47) for pixel (x, y) and the sub-pixel jitter (jx, jy). */
typedef struct LjPrimaryQuery { int32_t x, y; float jx, jy; } LjPrimaryQuery;
typedef struct LjPrimaryResult { float org[3], dir[3]; } LjPrimaryResult;
int lj_primary_ray_queries(lj_scene *scene, int64_t n, const LjPrimaryQuery *queries_host, LjPrimaryResult *results_host);

/* sample(filter, rnd) (filter.h:47, filters/{box,tent,gaussian}.inl); no scene needed. */
typedef struct LjFilterQuery { int32_t kind; float param, rnd[2]; } LjFilterQuery;
int lj_filter_queries(lj_context *ctx, int64_t n, const LjFilterQuery *queries_host, float *offsets_host /* 2 per query */);

/* init_pcg32(stream, seed) then `count` draws (pcg.h:22-68): u32_host[i*count + k] = next_pcg32, real_host = the device's
 * [0,1) float from the same word (NULL: not wanted).  seed 0: 0x853c49e6748fea9b. */
int lj_pcg32_queries(lj_context *ctx, int64_t n, const uint64_t *stream_ids_host, uint64_t seed, int32_t count, uint32_t *u32_host, float *real_host);

/* eval(texture, uv, footprint, pool) (texture.h:123-154, mipmap.h:52-89) of a caller-described texture against the scene's
 * TexturePool; spectrum != 0: Texture<Spectrum> (3 values), else Texture<Real> (value in out[0..2] replicated). */
typedef struct LjTextureQuery { LjTexture texture; double uv[2]; float footprint; int32_t spectrum; } LjTextureQuery;
int lj_texture_queries(lj_scene *scene, int64_t n, const LjTextureQuery *queries_host, float *rgb_host /* 3 per query */);

/* to_local(frame, v) / to_world(frame, v) / Frame(n) (frame.h:11-56) */
typedef struct LjFrameQuery { float n[3], v[3]; } LjFrameQuery;
typedef struct LjFrameResult { float x[3], y[3], to_local[3], to_world[3]; } LjFrameResult;
int lj_frame_queries(lj_context *ctx, int64_t n, const LjFrameQuery *queries_host, LjFrameResult *results_host);

/* Counters of the last lj_render* call. */
typedef struct LjStats {
    uint64_t samples;          /* camera samples traced */
    uint64_t bounce_iterations;/* sum over samples of executed iterations of the loop at path_tracing.h:66 (K) */
    uint64_t rays_closest, rays_shadow;
    uint64_t wavefront_steps;  /* host-side iterations of the extend/shade cycle */
    uint64_t queue_bytes;      /* algorithmic queue bytes moved */
    double render_ms;          /* device time of the timed region (HIP events on the render stream) */
    double extend_ms, shade_ms, generate_ms, resolve_ms; /* per-kernel sums, HIP events */
    uint64_t extend_launches, shade_launches;
    uint64_t extend_bytes, shade_bytes; /* algorithmic bytes of each kernel, summed over launches */
    /* tiny scenes run as one fused persistent launch per pass (k_mega) instead of extend / shade steps */
    double mega_ms; uint64_t mega_launches, mega_bytes, path_steps;
} LjStats;
int lj_get_stats(const lj_scene *scene, LjStats *out);

/* Scene facts the host needs (scene.h:58-81). */
typedef struct LjSceneInfo {
    int32_t width, height, spp, max_depth, rr_depth, integrator;
    int64_t n_triangles, n_spheres, n_bvh_nodes;
    double bounds_radius, bounds_center[3], shadow_epsilon;
} LjSceneInfo;
int lj_scene_info(const lj_scene *scene, LjSceneInfo *out);

/* ---------------------------------------------------------------- device groups: one process, N devices
 * The reference's data-parallel axis is its tile loop (render.cpp:75-98 on parallel.cpp:183-237: independent 16x16 tiles, disjoint
 * pixel writes).  A group shards that loop over devices from one host process: tile t = ty*ntx + tx goes to device t mod N, every
 * device renders its tiles at full spp into a zeroed full frame, one ncclReduce(sum, root 0) of the float frames over xGMI, one
 * copy to the host.  Each pixel has one non-zero contributor, so the image is bit-identical to a one-device render.
 * (The process-per-GPU route — LjRenderArgs.rank / world_size + the caller's own collective — stays available: bench.py.)
 *
 * device_ids: n_devices HIP device ids (NULL: 0..n-1).  Ids may repeat ("logical ranks" on one GPU: same sharding, frames summed by
 * a device kernel, since RCCL refuses duplicate devices).  RCCL (librccl.so) is bound at run time, only for >= 2 distinct devices. */
typedef struct lj_device_group lj_device_group;
typedef struct lj_group_scene lj_group_scene;
int lj_group_create(int n_devices, const int *device_ids, lj_device_group **out);
void lj_group_destroy(lj_device_group *group);
int lj_group_size(const lj_device_group *group);
lj_context *lj_group_context(lj_device_group *group, int i);     /* borrowed: device i's context */
int lj_group_uses_rccl(const lj_device_group *group);             /* 1: the frames are reduced by ncclReduce, 0: by a device-side sum */
/* Scene::Scene (scene.cpp:3-53) on every device of the group; the description may be freed afterwards. */
int lj_group_scene_upload(lj_device_group *group, const LjSceneDesc *desc, lj_group_scene **out);
void lj_group_scene_destroy(lj_group_scene *scene);
lj_scene *lj_group_scene_member(lj_group_scene *scene, int i);   /* borrowed: the scene on device i */
/* Replaces `Image3 render(const Scene&)` (render.h:9) across the group.  args->rank / world_size must be 0 (the group shards). */
int lj_group_render(lj_group_scene *scene, const LjRenderArgs *args, float *rgb_host);
int lj_group_get_stats(const lj_group_scene *scene, LjStats *out);   /* sums over the shares; render_ms is the slowest share's */

/* imwrite() (image.h:46, image.cpp:135-173), host only: writes a width x height RGB float image (row-major, y = 0 at the
 * top) by extension — ".pfm": "PF", "<w> <h>", "-1", then the rows top to bottom as little-endian float (the
 * reference's layout, image.cpp:141-149); ".exr": scan-line OpenEXR, HALF channels B, G, R (what the reference asks
 * tinyexr for with "write as fp16", image.cpp:161).  The reference ignores other extensions; here they are
 * LJ_ERR_UNSUPPORTED. */
int lj_image_write(const char *filename, int32_t width, int32_t height, const float *rgb);
/* imread3() / imread1() (image.h:41-45, image.cpp:28-133), host only: decodes a texture file the way the reference's loaders do (JPEG, PNG,
 * TGA, BMP, PSD, GIF, PIC and Radiance HDR with stb_image's conventions incl. its (float) pow(v / 255, 2.2) for 8-bit data; OpenEXR; PFM) into `channels`
 * (1 or 3) floats per pixel, row-major, y = 0 at the top.  *data is owned by the library: release it with lj_image_free. */
int lj_image_read(const char *filename, int32_t channels, int32_t *width, int32_t *height, float **data);
void lj_image_free(float *data);

#ifdef __cplusplus
}
#endif
#endif
