"""ctypes mirror of include/lajolla_hip.h (kept field-for-field in the header's order)."""
import ctypes as C

LJ_OK = 0
LJ_ERR_INVALID_ARG, LJ_ERR_PARSE, LJ_ERR_IO, LJ_ERR_UNSUPPORTED, LJ_ERR_DEVICE, LJ_ERR_INTERNAL = -1, -2, -3, -4, -5, -6

LJ_FILTER_BOX, LJ_FILTER_TENT, LJ_FILTER_GAUSSIAN = 0, 1, 2
LJ_SHAPE_SPHERE, LJ_SHAPE_TRIMESH = 0, 1
LJ_TEX_CONSTANT, LJ_TEX_IMAGE, LJ_TEX_CHECKERBOARD = 0, 1, 2
LJ_LIGHT_AREA, LJ_LIGHT_ENVMAP = 0, 1
MATERIAL_KINDS = ["lambertian", "roughplastic", "roughdielectric", "disneydiffuse", "disneymetal",
                  "disneyglass", "disneyclearcoat", "disneysheen", "disneybsdf"]
LJ_INTEGRATOR_PATH = 5
LJ_MAX_TEX_SLOTS = 12
INT32_MIN = -2**31

# texture slot names per material kind, in the reference's field order (material.h:12-98)
MATERIAL_SLOTS = {
    "lambertian": ["reflectance"],
    "roughplastic": ["diffuse_reflectance", "specular_reflectance", "roughness"],
    "roughdielectric": ["specular_reflectance", "specular_transmittance", "roughness"],
    "disneydiffuse": ["base_color", "roughness", "subsurface"],
    "disneymetal": ["base_color", "roughness", "anisotropic"],
    "disneyglass": ["base_color", "roughness", "anisotropic"],
    "disneyclearcoat": ["clearcoat_gloss"],
    "disneysheen": ["base_color", "sheen_tint"],
    "disneybsdf": ["base_color", "specular_transmission", "metallic", "subsurface", "specular", "roughness",
                   "specular_tint", "anisotropic", "sheen", "sheen_tint", "clearcoat", "clearcoat_gloss"],
}
# which slots are Texture<Spectrum> (others are Texture<Real>)
SPECTRUM_SLOTS = {"reflectance", "diffuse_reflectance", "specular_reflectance", "specular_transmittance", "base_color"}


class LjTexture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture_id", C.c_int32), ("value", C.c_double * 3), ("color1", C.c_double * 3),
                ("uscale", C.c_double), ("vscale", C.c_double), ("uoffset", C.c_double), ("voffset", C.c_double)]


class LjMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_tex", C.c_int32), ("eta", C.c_double), ("tex", LjTexture * LJ_MAX_TEX_SLOTS)]


class LjShape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material_id", C.c_int32), ("area_light_id", C.c_int32),
                ("interior_medium_id", C.c_int32), ("exterior_medium_id", C.c_int32),
                ("has_normals", C.c_int32), ("has_uvs", C.c_int32), ("_pad", C.c_int32),
                ("first_vertex", C.c_int64), ("n_vertices", C.c_int64),
                ("first_triangle", C.c_int64), ("n_triangles", C.c_int64),
                ("position", C.c_double * 3), ("radius", C.c_double)]


class LjLight(C.Structure):
    _fields_ = [("kind", C.c_int32), ("shape_id", C.c_int32), ("intensity", C.c_double * 3), ("values", LjTexture),
                ("to_world", C.c_double * 16), ("to_local", C.c_double * 16), ("scale", C.c_double)]


class LjImage(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("_pad", C.c_int32),
                ("data", C.POINTER(C.c_float))]


class LjCamera(C.Structure):
    _fields_ = [("cam_to_world", C.c_double * 16), ("world_to_cam", C.c_double * 16),
                ("sample_to_cam", C.c_double * 16), ("cam_to_sample", C.c_double * 16),
                ("width", C.c_int32), ("height", C.c_int32), ("filter_kind", C.c_int32), ("medium_id", C.c_int32),
                ("filter_param", C.c_double)]


class LjRenderOptions(C.Structure):
    _fields_ = [("integrator", C.c_int32), ("samples_per_pixel", C.c_int32), ("max_depth", C.c_int32),
                ("rr_depth", C.c_int32), ("vol_path_version", C.c_int32), ("max_null_collisions", C.c_int32)]


LJ_VOLUME_CONSTANT, LJ_VOLUME_GRID = 0, 1
LJ_MEDIUM_HOMOGENEOUS, LJ_MEDIUM_HETEROGENEOUS = 0, 1
LJ_PHASE_ISOTROPIC, LJ_PHASE_HG = 0, 1


class LjVolume(C.Structure):
    _fields_ = [("kind", C.c_int32), ("resolution", C.c_int32 * 3), ("value", C.c_double * 3),
                ("p_min", C.c_double * 3), ("p_max", C.c_double * 3), ("max_data", C.c_double * 3),
                ("scale", C.c_double), ("data", C.POINTER(C.c_float))]


class LjMedium(C.Structure):
    _fields_ = [("kind", C.c_int32), ("phase_kind", C.c_int32), ("g", C.c_double),
                ("sigma_a", C.c_double * 3), ("sigma_s", C.c_double * 3), ("albedo", LjVolume), ("density", LjVolume)]


class LjSceneDesc(C.Structure):
    _fields_ = [("camera", LjCamera), ("options", LjRenderOptions),
                ("n_shapes", C.c_int32), ("n_materials", C.c_int32), ("n_lights", C.c_int32),
                ("n_images3", C.c_int32), ("n_images1", C.c_int32), ("envmap_light_id", C.c_int32),
                ("shapes", C.POINTER(LjShape)), ("materials", C.POINTER(LjMaterial)), ("lights", C.POINTER(LjLight)),
                ("images3", C.POINTER(LjImage)), ("images1", C.POINTER(LjImage)),
                ("n_vertices", C.c_int64), ("n_triangles", C.c_int64),
                ("positions", C.POINTER(C.c_double)), ("normals", C.POINTER(C.c_double)),
                ("uvs", C.POINTER(C.c_double)), ("indices", C.POINTER(C.c_int32)),
                ("output_filename", C.c_char_p),
                ("n_media", C.c_int32), ("_pad_media", C.c_int32), ("media", C.POINTER(LjMedium))]


class LjRenderArgs(C.Structure):
    _fields_ = [("spp", C.c_int32), ("max_depth", C.c_int32), ("rng_mode", C.c_int32),
                ("rank", C.c_int32), ("world_size", C.c_int32),
                ("crop_x0", C.c_int32), ("crop_y0", C.c_int32), ("crop_x1", C.c_int32), ("crop_y1", C.c_int32),
                ("pool_paths", C.c_uint32), ("flags", C.c_uint32), ("seed", C.c_uint64)]


class LjRay(C.Structure):
    _fields_ = [("org", C.c_float * 3), ("tnear", C.c_float), ("dir", C.c_float * 3), ("tfar", C.c_float)]


class LjHit(C.Structure):
    _fields_ = [("t", C.c_float), ("u", C.c_float), ("v", C.c_float), ("shape_id", C.c_int32), ("prim_id", C.c_int32)]


class LjStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("bounce_iterations", C.c_uint64), ("rays_closest", C.c_uint64),
                ("rays_shadow", C.c_uint64), ("wavefront_steps", C.c_uint64), ("queue_bytes", C.c_uint64),
                ("render_ms", C.c_double), ("extend_ms", C.c_double), ("shade_ms", C.c_double),
                ("generate_ms", C.c_double), ("resolve_ms", C.c_double),
                ("extend_launches", C.c_uint64), ("shade_launches", C.c_uint64),
                ("extend_bytes", C.c_uint64), ("shade_bytes", C.c_uint64),
                ("mega_ms", C.c_double), ("mega_launches", C.c_uint64), ("mega_bytes", C.c_uint64), ("path_steps", C.c_uint64)]


class LjSceneInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("rr_depth", C.c_int32), ("integrator", C.c_int32),
                ("n_triangles", C.c_int64), ("n_spheres", C.c_int64), ("n_bvh_nodes", C.c_int64),
                ("bounds_radius", C.c_double), ("bounds_center", C.c_double * 3), ("shadow_epsilon", C.c_double)]


class LjVertex(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("geometry_normal", C.c_float * 3), ("frame_x", C.c_float * 3), ("frame_y", C.c_float * 3),
                ("frame_n", C.c_float * 3), ("uv_screen_size", C.c_float), ("mean_curvature", C.c_float), ("uv", C.c_double * 2),
                ("material_id", C.c_int32), ("light_id", C.c_int32), ("shape_id", C.c_int32), ("primitive_id", C.c_int32)]


class LjBsdfQuery(C.Structure):
    _fields_ = [("vertex", LjVertex), ("dir_in", C.c_float * 3), ("dir_out", C.c_float * 3), ("rnd_uv", C.c_float * 2), ("rnd_w", C.c_float), ("_pad", C.c_int32)]


class LjBsdfResult(C.Structure):
    _fields_ = [("eval", C.c_float * 3), ("pdf", C.c_float), ("sample_dir", C.c_float * 3), ("sample_eta", C.c_float), ("sample_roughness", C.c_float),
                ("sample_valid", C.c_int32)]


class LjLightQuery(C.Structure):
    _fields_ = [("light_id", C.c_int32), ("ref", C.c_float * 3), ("rnd_uv", C.c_float * 2), ("rnd_w", C.c_float), ("view_dir", C.c_float * 3)]


class LjLightResult(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("normal", C.c_float * 3), ("pdf", C.c_float), ("emission", C.c_float * 3), ("pmf", C.c_float), ("_pad", C.c_int32),
                ("position_d", C.c_double * 3)]


class LjHitQuery(C.Structure):
    _fields_ = [("org", C.c_float * 3), ("dir", C.c_float * 3), ("t", C.c_float), ("u", C.c_float), ("v", C.c_float), ("ray_spread", C.c_float),
                ("shape_id", C.c_int32), ("primitive_id", C.c_int32)]


class LjHitResult(C.Structure):
    _fields_ = [("vertex", LjVertex), ("emission", C.c_float * 3), ("_pad", C.c_int32)]


class LjPrimaryQuery(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("jx", C.c_float), ("jy", C.c_float)]


class LjPrimaryResult(C.Structure):
    _fields_ = [("org", C.c_float * 3), ("dir", C.c_float * 3)]


class LjFilterQuery(C.Structure):
    _fields_ = [("kind", C.c_int32), ("param", C.c_float), ("rnd", C.c_float * 2)]


class LjTextureQuery(C.Structure):
    _fields_ = [("texture", LjTexture), ("uv", C.c_double * 2), ("footprint", C.c_float), ("spectrum", C.c_int32)]


class LjFrameQuery(C.Structure):
    _fields_ = [("n", C.c_float * 3), ("v", C.c_float * 3)]


class LjFrameResult(C.Structure):
    _fields_ = [("x", C.c_float * 3), ("y", C.c_float * 3), ("to_local", C.c_float * 3), ("to_world", C.c_float * 3)]


# every symbol include/lajolla_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("lj_last_error", C.c_char_p, []),
    ("lj_version", C.c_char_p, []),
    ("lj_parse_scene", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    ("lj_host_scene_desc", C.POINTER(LjSceneDesc), [C.c_void_p]),
    ("lj_host_scene_free", None, [C.c_void_p]),
    ("lj_context_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("lj_context_destroy", None, [C.c_void_p]),
    ("lj_scene_upload", C.c_int, [C.c_void_p, C.POINTER(LjSceneDesc), C.POINTER(C.c_void_p)]),
    ("lj_scene_destroy", None, [C.c_void_p]),
    ("lj_render", C.c_int, [C.c_void_p, C.POINTER(LjRenderArgs), C.c_void_p]),
    ("lj_render_device", C.c_int, [C.c_void_p, C.POINTER(LjRenderArgs), C.c_void_p, C.c_void_p]),
    ("lj_render_samples", C.c_int, [C.c_void_p, C.POINTER(LjRenderArgs), C.c_void_p]),
    ("lj_intersect", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("lj_occluded", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("lj_get_stats", C.c_int, [C.c_void_p, C.POINTER(LjStats)]),
    ("lj_scene_info", C.c_int, [C.c_void_p, C.POINTER(LjSceneInfo)]),
    ("lj_image_write", C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("lj_image_read", C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_float))]),
    ("lj_image_free", None, [C.POINTER(C.c_float)]),
    ("lj_shade_variant_count", C.c_int, []),
    ("lj_scene_shade_variant", C.c_int, [C.c_void_p]),
    ("lj_bsdf_queries", C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    ("lj_light_queries", C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    ("lj_sample_light_queries", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("lj_vertex_queries", C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    ("lj_primary_ray_queries", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("lj_filter_queries", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("lj_pcg32_queries", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]),
    ("lj_texture_queries", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("lj_frame_queries", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("lj_group_create", C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    ("lj_group_destroy", None, [C.c_void_p]),
    ("lj_group_size", C.c_int, [C.c_void_p]),
    ("lj_group_context", C.c_void_p, [C.c_void_p, C.c_int]),
    ("lj_group_uses_rccl", C.c_int, [C.c_void_p]),
    ("lj_group_scene_upload", C.c_int, [C.c_void_p, C.POINTER(LjSceneDesc), C.POINTER(C.c_void_p)]),
    ("lj_group_scene_destroy", None, [C.c_void_p]),
    ("lj_group_scene_member", C.c_void_p, [C.c_void_p, C.c_int]),
    ("lj_group_render", C.c_int, [C.c_void_p, C.POINTER(LjRenderArgs), C.c_void_p]),
    ("lj_group_get_stats", C.c_int, [C.c_void_p, C.POINTER(LjStats)]),
]
