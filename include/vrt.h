/*
 * vrt.h — C ABI of libvrt.so, the MI355X-native tracing core for VisualRayTrace.
 *
 * Every entry point here stands in for a tracer-facing call of the Unity C# host
 * (paths relative to VisualRayTrace/Assets/Scripts/RayTracer/).  The reference has
 * no native boundary of its own; the seam is the data its host already owns:
 * camera (RayTracerUnity.cs:285, Info/ViewPortPlaneInformation.cs:86-99), scene
 * colliders/materials/lights (Tracer/WhittedTracer.cs:43-88,
 * Util/RayTraceUtility.cs:478-528), settings (RayTracerUnity.cs:102-168), and the
 * outputs it consumes: Color[] PixelColorData (Info/TextureInformation.cs:36) and
 * the ray polyline for the LineRenderer (RayTracerUnity.cs:620-654).
 *
 * Conventions: all structs are blittable (LayoutKind.Sequential, 4-byte floats and
 * ints, pointers as IntPtr); Vector3 = 3 floats, Color = 4 floats.  Functions
 * return 0 or a negative vrt_status and never throw.  Inputs are copied before the
 * call returns; output buffers are caller-owned and fully written on success.
 * One call at a time per vrt_ctx.
 */
#ifndef VRT_H
#define VRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRT_ABI_VERSION 5

typedef struct vrt_ctx vrt_ctx;

typedef enum vrt_status {
    VRT_OK = 0,
    VRT_ERR_INVALID = -1,   /* bad argument (null pointer, size out of range, bad id) */
    VRT_ERR_NO_DEVICE = -2, /* no HIP device / HIP runtime unavailable */
    VRT_ERR_HIP = -3,       /* a HIP call failed; see vrt_last_error */
    VRT_ERR_STATE = -4,     /* scene or camera not set */
    VRT_ERR_CAPACITY = -5   /* a fixed capacity was exceeded (lights, depth, records) */
} vrt_status;

/* Material classes; values follow RayTraceUtility.MaterialType
 * (Util/RayTraceUtility.cs:106).  The class is what DetermineMaterialType
 * (Util/RayTraceUtility.cs:139-152) derives from the Unity material name. */
enum {
    VRT_MAT_SOLID = 1, VRT_MAT_METAL = 2, VRT_MAT_DIELECTRIC = 3,
    /* The two materials the reference defines but never instantiates (their only users are
     * commented out): Material/MatteMaterial.cs:44-75 (ambient + Lambert, no shadow test, no
     * specular lobe) and Material/TransparentMaterial.cs:40-69 (Phong + perfect specular
     * reflection + BTDF/PerfectTransmitterBTDF.cs:32-69).  A host selects them by class like
     * the others; the kernels prune them at compile time for scenes without them. */
    VRT_MAT_MATTE = 4, VRT_MAT_TRANSPARENT = 5
};

/* Light kinds (Util/RayTraceUtility.cs:495-526). */
enum { VRT_LIGHT_DIRECTIONAL = 0, VRT_LIGHT_POINT = 1 };

/* Sampler ids = AASamplingStrategy (AntiAliasingStrategy.cs:8). */
enum {
    VRT_SAMPLER_REGULAR = 0, VRT_SAMPLER_RANDOM = 1, VRT_SAMPLER_JITTERED = 2,
    VRT_SAMPLER_NROOKS = 3, VRT_SAMPLER_MULTIJITTERED = 4, VRT_SAMPLER_HAMMERSLEY = 5
};

/* Behaviour switches.  0 reproduces the shipped C# including its quirks; each bit
 * replaces one quirk by the textbook behaviour. */
enum {
    VRT_FIX_PRIMARY_SKY = 1u << 0, /* primary misses return the sky colour instead of black
                                      (RayTracerUnity.cs:582-592) */
    VRT_FIX_AO_BASIS = 1u << 1,    /* ambient-occlusion ray always leaves along the normal; the
                                      reference poisons it with NaN for most normals
                                      (Light/AmbientOccluder.cs:66-73) */
    VRT_FIX_DIELECTRIC = 1u << 2,  /* Fresnel reflection + refraction with EtaIn/EtaOut instead
                                      of the always-total-internal-reflection wiring
                                      (Material/DielectricMaterial.cs:73, BTDF/FresnelTransmitterBTDF.cs:57-69) */
    /* Not a behaviour switch: output format of vrt_render_device only.  The frame buffer receives
       3 floats per pixel, the averaged radiance BEFORE MaxToOne (Util/RayTraceUtility.cs:409-422);
       vrt_tonemap_device turns it into the usual 4 floats.  Multi-GPU hosts gather strips in
       this form (12 instead of 16 bytes per pixel over xGMI) and tone-map once, after the gather. */
    VRT_OUT_LINEAR_RGB = 1u << 8,
    /* Not a behaviour switch either: output format of every frame call (vrt_render, vrt_render_begin,
       vrt_render_begin_parts, vrt_render_device).  The frame buffer receives 4 BYTES per pixel,
       r, g, b, a, each channel round(255 * clamp01(c)) of the tone-mapped colour (halves to even,
       NaN as 0): the Color32 a Texture2D of format RGBA32 holds after SetPixels + Apply
       (RayTracerUnity.cs:472-475), which is all UpdateRenderTexture ever shows.  rgba_out then is
       width*height*4 bytes, a quarter of what crosses PCIe per frame otherwise.  The float
       Color[] stays the default and the format parity is stated on. */
    VRT_OUT_RGBA8 = 1u << 9
};

/* Kernel organisation. */
enum {
    VRT_MODE_AUTO = 0,
    VRT_MODE_MEGAKERNEL = 1, /* one lane follows one pixel through all samples and bounces */
    VRT_MODE_WAVEFRONT = 2   /* per-bounce kernels over compacted SoA ray queues in HBM */
};

/* Ray classes, used for counters and path records.  One ray = one intersection query
 * (a Physics.Raycast call site).  PRECHECK is the any-hit test of
 * RayTracerUnity.cs:582, answered by the primary closest-hit query. */
enum {
    VRT_RAY_PRECHECK = 0, VRT_RAY_PRIMARY = 1, VRT_RAY_REFLECT = 2,
    VRT_RAY_TRANSMIT = 3, VRT_RAY_SHADOW = 4, VRT_RAY_AO = 5, VRT_RAY_CLASSES = 6
};

/* SphereCollider in world space. */
typedef struct vrt_sphere {
    float center[3];
    float radius;
    int32_t material; /* index into vrt_scene_desc.materials */
} vrt_sphere;

/* BoxCollider in world space: centre, three orthonormal axes (rows) and half extents. */
typedef struct vrt_box {
    float center[3];
    float axis[9];
    float half[3];
    int32_t material;
} vrt_box;

/* MeshRenderer.material as the tracer reads it (Tracer/WhittedTracer.cs:49-86). */
typedef struct vrt_material {
    int32_t type;   /* VRT_MAT_* */
    float color[4]; /* mat.color, passed to SetCD */
} vrt_material;

/* A scene light (Light/PointLight.cs, Light/DirectionalLight.cs). */
typedef struct vrt_light {
    int32_t type;         /* VRT_LIGHT_* */
    float vec[3];         /* point: position; directional: the vector GetDirection returns */
    float ls;             /* radiance factor (Light.intensity) */
    float color[4];
    int32_t cast_shadows; /* AbstractLight.CastShadows */
    float location[3];    /* directional only: _location used by its InShadow (never assigned
                             in the reference, so zero) */
} vrt_light;

/* Constants of the three singleton materials and the ambient occluder
 * (Util/RayTraceUtility.cs:15-76,481-490).  vrt_shading_defaults fills in the shipped values. */
typedef struct vrt_shading {
    float solid_ka, solid_kd, solid_ks, solid_exp;
    float metal_ka, metal_kd, metal_ks, metal_exp, metal_kr;
    float metal_cr[4];
    float glass_ka, glass_kd, glass_ks, glass_exp;
    float glass_eta_in, glass_eta_out; /* used only with VRT_FIX_DIELECTRIC */
    float ao_min_amount[4];
    float ao_ls;
    float ao_color[4];
    /* MatteMaterial: the two LambertianBRDF coefficients (BRDF/LambertianBRDF.cs:19). */
    float matte_ka, matte_kd;
    /* TransparentMaterial(rayDir, ks, exp, ior, kr, kt) (Material/TransparentMaterial.cs:10-25):
     * the reference never constructs one, so these are host-chosen constructor arguments
     * (vrt_shading_defaults: Suffern's glass, ks 0.5, exp 2000, ior 1.5, kr 0.1, kt 0.9);
     * ka/kd are the Phong base's Lambert coefficients.  transp_ray_dir is the rayDir the
     * constructor hands to PerfectTransmitterBTDF, which keeps it for good and evaluates
     * Tir() with it (BTDF/PerfectTransmitterBTDF.cs:17-20,57-69); the reference's material
     * singletons are all built with Vector3.zero (Util/RayTraceUtility.cs:15,27,59). */
    float transp_ka, transp_kd, transp_ks, transp_exp;
    float transp_ior, transp_kr, transp_kt;
    float transp_ray_dir[3];
} vrt_shading;

/* Material.mainTexture as the tracer reads it (Tracer/WhittedTracer.cs:55-63): a hit on a
 * textured material returns texture.GetPixel(floor(u*width), floor(v*height)) unshaded.
 * rgba = width*height*4 floats, row 0 first (Unity's texture origin is the lower left). */
typedef struct vrt_texture {
    int32_t width, height;
    int32_t wrap;        /* TextureWrapMode for out-of-range pixels: 0 = Repeat, 1 = Clamp */
    const float* rgba;
} vrt_texture;

typedef struct vrt_scene_desc {
    uint32_t n_spheres;
    const vrt_sphere* spheres;
    uint32_t n_boxes;
    const vrt_box* boxes;
    /* MeshColliders flattened to one indexed triangle list; single-sided, front face =
     * clockwise winding seen from outside (Unity convention), flat normals. */
    uint32_t n_vertices;
    const float* vertices;        /* n_vertices * 3 */
    uint32_t n_triangles;
    const uint32_t* indices;      /* n_triangles * 3 */
    const int32_t* tri_material;  /* n_triangles, or NULL for material 0 */
    uint32_t n_materials;
    const vrt_material* materials;
    uint32_t n_lights;
    const vrt_light* lights;      /* at most VRT_MAX_LIGHTS */
    vrt_shading shading;
    /* Textures (all optional). */
    uint32_t n_textures;
    const vrt_texture* textures;
    const int32_t* material_texture; /* n_materials entries: texture index or -1; NULL = none */
    const float* uvs;                /* n_vertices * 2 mesh texture coordinates, or NULL for (0,0).
                                        RaycastHit.textureCoord is (0,0) on sphere and box colliders. */
} vrt_scene_desc;

#define VRT_MAX_LIGHTS 8
#define VRT_MAX_DEPTH 15
#define VRT_MAX_GPUS 16                /* GPUs of one vrt_create_multi context */
#define VRT_MAX_PARTS 64               /* strip sets of vrt_render_begin_parts */
#define VRT_NOALLOC_MAX_OFFSETS 65536  /* (x,y) pairs of the sampler table vrt_render_device_noalloc accepts */

/* Primitive ids reported by the library: spheres first, then boxes, then triangles,
 * each in the order given; -1 = miss. */

/* Camera: rayOrigin, PlaneBorderPoints[0], PlaneXAxis, PlaneYAxis
 * (RayTracerUnity.cs:285,486-517).  width = pixels along X, height = pixels along Y;
 * the reference is square (TextureDimension). */
typedef struct vrt_camera {
    float origin[3];
    float p0[3];
    float x_axis[3];
    float y_axis[3];
    int32_t width;
    int32_t height;
} vrt_camera;

/* One recorded ray of the ray tree of a pixel (feeds RT_rec_points /
 * LineRenderer, RayTracerUnity.cs:620-654). */
typedef struct vrt_path_seg {
    float origin[3];
    float end[3];      /* hit point, or origin + direction * range on a miss */
    int32_t kind;      /* VRT_RAY_* (never PRECHECK) */
    int32_t depth;     /* recursion depth of the TraceRay call that cast it */
    int32_t hit_prim;  /* primitive id or -1 */
    int32_t sample;    /* anti-aliasing sample index */
} vrt_path_seg;

typedef struct vrt_params {
    int32_t spp;          /* SampleSize */
    int32_t n_sets;       /* SampleSetCount; rows of `offsets` */
    const float* offsets; /* n_sets * spp (x,y) pairs in access order, i.e.
                             samples[jump + shuffled[jump + i]] (Sampler/AbstractSampler.cs:163);
                             NULL = all zero */
    uint32_t seed;        /* selects the set per pixel: set = vrt_pixel_set(seed, pixel, n_sets) */
    int32_t max_depth;    /* GlobalWorld.MaxDepth */
    float range;          /* RayTrace_Range */
    uint32_t flags;       /* VRT_FIX_* | VRT_OUT_* */
    int32_t mode;         /* VRT_MODE_* */
    /* Tile sharding: the frame is cut into strips of tile_rows rows; this call renders
     * strips s with s % tile_count == tile_index and leaves the other rows untouched. */
    int32_t tile_index;
    int32_t tile_count;   /* <= 1: whole frame */
    int32_t tile_rows;
    int32_t compact_rows; /* != 0: the output buffers hold only this call's rows, packed in
                             ascending row order (local_rows * width entries) */
    /* Path recording for a bounded set of pixels (linear index h*width+v). */
    int32_t n_record_pixels;
    const int32_t* record_pixels;
    int32_t record_cap;           /* segments per recorded pixel */
    vrt_path_seg* record_out;     /* n_record_pixels * record_cap, host memory */
    int32_t* record_count_out;    /* n_record_pixels: segments produced (may exceed cap;
                                     only the first cap are stored) */
} vrt_params;

typedef struct vrt_stats {
    uint64_t rays[VRT_RAY_CLASSES]; /* intersection queries per class */
    uint64_t pixels;                /* pixels rendered by this call */
    float gpu_ms;                   /* device time of the render kernels (HIP events) */
    float bvh_build_ms;             /* device time of the last LBVH build (vrt_set_scene) */
    uint32_t bvh_nodes;
    int32_t mode_used;              /* VRT_MODE_* actually run */
    uint64_t nodes_visited;         /* LBVH nodes entered and primitives tested by the call, per lane; */
    uint64_t prims_tested;          /* counted only by a library built with -DVRT_COUNT_NODES=1, else 0 */
} vrt_stats;

/* Lifetime.  Replaces RayTracerUnity.Start (RayTracerUnity.cs:266-307), which builds the
 * world and tracer objects.  device = HIP device ordinal. */
int vrt_create(int device, vrt_ctx** out);
int vrt_destroy(vrt_ctx* ctx);
const char* vrt_last_error(vrt_ctx* ctx);

/* The same context over n_gpus GPUs of one node (devices = their HIP ordinals, NULL = 0..n_gpus-1).
 * The Unity host is one process with one main thread (RayTracerUnity.cs:312-352), so the split
 * lives behind the boundary: vrt_set_scene / vrt_set_camera replicate to every GPU, vrt_render and
 * vrt_render_begin deal 8-row strips of the frame round-robin to the GPUs (one host thread and
 * stream each, all samples of a strip on its GPU), and every GPU tone-maps and delivers the rows of
 * its strips to the caller's buffers itself, over its own PCIe link (stored there by its frame
 * kernel when the buffer comes from vrt_host_alloc).  vrt_render_poll reports the frame done when
 * every GPU's part is; the statistics are summed, gpu_ms is the longest GPU's.  vrt_trace_pixel,
 * vrt_trace_ray and path records run on the first GPU.  Option "multi_gather" = 1 gathers the
 * strips on the first GPU over xGMI instead (peer-to-peer copies; an RCCL send/recv group with
 * VRT_MULTI_RCCL=1 in the environment, a form no test has run on more than one GPU) and
 * assembles, tone-maps and reads the frame back there.  vrt_render_begin_parts and the
 * device-pointer calls are not offered on such a context. */
int vrt_create_multi(int n_gpus, const int* devices, vrt_ctx** out);

/* Study and test knobs by name ("sample_shift", "packet", "beam", "slab_fma", "quad_max", "stack_cap",
 * "wf_defer", "wf_refill", "wf_max_slots", "wf_any_cap", "wf_queue_cap", "builder", "collapse_parity",
 * "no_big", "xcd_remap", "block_compact", "zero_copy", "staged", "tile_sort", "multi_gather", "fail_after_build"); none changes a result.
 * builder / collapse_parity / no_big take effect at the next vrt_set_scene.  No counterpart in
 * the reference. */
int vrt_set_option(vrt_ctx* ctx, const char* name, long long value);

/* Page-locked host memory for the frame the caller reads back over and over (the Color[]
 * PixelColorData of Info/TextureInformation.cs:36,58-63, which a Unity host wraps in a NativeArray):
 * the frame kernels of vrt_render / vrt_render_begin / vrt_render_begin_parts store their pixels
 * straight into such a buffer (it is mapped into the device's address space), so the frame crosses
 * PCIe while it is being traced; hit ids are copied at PCIe speed instead of through the runtime's
 * staging copies of pageable memory.  Freed by vrt_host_free or vrt_destroy.
 * vrt_host_register page-locks a buffer the caller allocated itself; the operating system may
 * refuse (VRT_ERR_HIP), in which case the buffer still works, only slower. */
int vrt_host_alloc(vrt_ctx* ctx, uint64_t bytes, void** out);
int vrt_host_free(vrt_ctx* ctx, void* buffer);
int vrt_host_register(vrt_ctx* ctx, void* buffer, uint64_t bytes);
int vrt_host_unregister(vrt_ctx* ctx, void* buffer);

/* Fills the shipped constants (Util/RayTraceUtility.cs:43-47,73-76,484-489,
 * BRDF/LambertianBRDF.cs:19, BRDF/GlossySpecularBRDF.cs:17-20). */
void vrt_shading_defaults(vrt_shading* out);
void vrt_params_defaults(vrt_params* out);

/* Scene upload; builds the LBVH on the device when the primitive count calls for it.
 * Replaces the WorldInformation constructor (Util/RayTraceUtility.cs:478-528) and the
 * per-hit collider/material lookups (Tracer/WhittedTracer.cs:43-88). */
int vrt_set_scene(vrt_ctx* ctx, const vrt_scene_desc* scene);

/* Replaces InitPlane/InitTexture + rayOrigin (RayTracerUnity.cs:273-285,358-394). */
int vrt_set_camera(vrt_ctx* ctx, const vrt_camera* cam);

/* Whole frame: the loop Update → DoRayTraceArchitecture → UpdateScene over every pixel
 * (RayTracerUnity.cs:312-352,565-597,607-617).  rgba_out = width*height*4 floats
 * (PixelColorData layout, index h*width+v; width*height*4 bytes with VRT_OUT_RGBA8), hit_id_out = width*height int32 primary
 * primitive ids of sample 0 (may be NULL).  The buffers may be any host memory: a buffer of
 * vrt_host_alloc is written by the frame kernel itself, pageable memory (a pinned managed array)
 * is filled band by band from page-locked staging while the next band is traced. */
int vrt_render(vrt_ctx* ctx, const vrt_params* params, float* rgba_out,
               int32_t* hit_id_out, vrt_stats* stats_out);

/* The same frame without blocking the caller: begin enqueues the frame and the copies into
 * rgba_out / hit_id_out (which must stay valid, and should be page-locked for the copies to
 * overlap host work, until vrt_render_end), poll reports completion without waiting, end waits
 * and returns the statistics.  Lets the host keep rendering its own (VR) frames while a traced
 * frame is in flight instead of yielding once per sample (RayTracerUnity.cs:333-337,588); with
 * tile_index/tile_count the frame can also be delivered strip by strip.  No other call on the
 * context between begin and end except poll; path recording needs the blocking vrt_render. */
int vrt_render_begin(vrt_ctx* ctx, const vrt_params* params, float* rgba_out, int32_t* hit_id_out);
int vrt_render_poll(vrt_ctx* ctx, int* done);
int vrt_render_end(vrt_ctx* ctx, vrt_stats* stats_out);

/* The whole frame delivered in n_parts strip sets (1..VRT_MAX_PARTS): part k = the strips s of
 * params->tile_rows rows with s % n_parts == k, i.e. the rows r with (r / tile_rows) % n_parts == k.
 * Each part is rendered, only its rows are copied to rgba_out / hit_id_out, and its completion
 * is visible to vrt_render_poll_parts (parts_done = parts whose rows have landed; they land in
 * order), so the host uploads the dirty rows of a part as soon as it is done instead of
 * SetPixels+Apply of the whole texture per pixel (RayTracerUnity.cs:461-477).  params must
 * describe the whole frame (tile_count <= 1, compact_rows 0).  vrt_render_poll and
 * vrt_render_end work as after vrt_render_begin; the statistics are those of the whole frame. */
int vrt_render_begin_parts(vrt_ctx* ctx, const vrt_params* params, float* rgba_out, int32_t* hit_id_out,
                           int n_parts);
int vrt_render_poll_parts(vrt_ctx* ctx, int* parts_done);

/* Same, with the outputs left in device memory (rgba_dev, hit_id_dev are device pointers
 * on the context's GPU; hit_id_dev may be NULL).  stream = a hipStream_t or NULL.  The call
 * enqueues and returns; stats_out (may be NULL) is then complete only for gpu_ms = 0 and
 * ray counts are fetched by vrt_fetch_stats after the caller has synchronised. */
int vrt_render_device(vrt_ctx* ctx, const vrt_params* params, void* rgba_dev,
                      void* hit_id_dev, void* stream);
int vrt_fetch_stats(vrt_ctx* ctx, vrt_stats* stats_out);

/* vrt_render_device for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the
 * reference.  Same arguments, same frame, same counts (vrt_fetch_stats), but the call allocates
 * and frees nothing: the ray counters and the device copy of the sampler table live in a
 * static device array of the library (one copy per GPU, 16 KiB + 8 bytes x
 * VRT_NOALLOC_MAX_OFFSETS = 528 KiB, held until the process ends).  Megakernel only
 * (mode VRT_MODE_WAVEFRONT is refused: its ray queues need a workspace) and
 * n_sets * spp <= VRT_NOALLOC_MAX_OFFSETS (else VRT_ERR_CAPACITY).  A sampler table the GPU's
 * array already holds is not uploaded again: such a call enqueues one fill and the kernel, with
 * no host-side wait, and is what a capturing stream can take; a call with a new table under
 * capture is refused (VRT_ERR_STATE: upload it with one uncaptured call first).  Every call on
 * a GPU, from whichever context, shares that GPU's array: calls that overlap on one GPU race
 * (also a call that finds its table there while the upload by another stream is still in
 * flight), and the counts of a call are gone once the next one starts. */
int vrt_render_device_noalloc(vrt_ctx* ctx, const vrt_params* params, void* rgba_dev,
                              void* hit_id_dev, void* stream);

/* RayTracerUnity.ProcessColor (RayTracerUnity.cs:525-562) for n_pixels pixels already in device
 * memory: rgb_dev holds 3 floats per pixel as VRT_OUT_LINEAR_RGB leaves them, rgba_dev receives
 * 4 floats per pixel, bit for bit what the frame kernel writes without that flag.  Enqueues on
 * `stream` (a hipStream_t or NULL) and returns. */
int vrt_tonemap_device(vrt_ctx* ctx, const void* rgb_dev, void* rgba_dev, uint64_t n_pixels, void* stream);

/* Single-step VR mode: one pixel, optionally one sample (sample < 0 = all samples,
 * averaged and tone-mapped like a frame pixel).  Replaces one DoRayTraceArchitecture
 * coroutine and fills the polyline UpdateScene draws (RayTracerUnity.cs:565-597,620-654).
 * Also serves legacy AbstractTracer.TraceRay callers (Tracer/AbstractTracer.cs:7-11). */
int vrt_trace_pixel(vrt_ctx* ctx, const vrt_params* params, int h, int v, int sample,
                    vrt_path_seg* segs, int cap, int* n_out, float rgba[4],
                    int32_t* hit_id_out);

/* One ray through the tracer: AbstractTracer.TraceRay(Ray ray, int depth)
 * (Tracer/AbstractTracer.cs:9, Tracer/WhittedTracer.cs:33-104), for legacy per-ray callers and
 * material code that recurses by hand.  `direction` is normalised like the Ray constructor does;
 * the result is the raw radiance (no averaging, no tone mapping), alpha = 1.  depth >
 * max_depth returns black without casting anything. */
int vrt_trace_ray(vrt_ctx* ctx, const vrt_params* params, const float origin[3], const float direction[3],
                  int depth, float rgba[4], int32_t* hit_id_out);

/* Host-side sampler tables (Sampler/{Regular,Random,Jittered,NRooks,MultiJittered,Hammersley}Sampler.cs, Sampler/AbstractSampler.cs:37-52,143-182):
 * writes n_sets * n_samples (x,y) pairs in access order.  UnityEngine.Random is replaced
 * by PCG32 seeded with `seed`.  Needs no GPU. */
int vrt_sampler_table(int sampler, int n_samples, int n_sets, float h_step, float v_step,
                      uint64_t seed, float* offsets_out);

/* Sample-set choice per pixel (stands in for the shared `_count`/`_jump` state of
 * Sampler/AbstractSampler.cs:149-153, which is not reproducible). */
uint32_t vrt_pixel_set(uint32_t seed, uint32_t pixel, uint32_t n_sets);

int vrt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VRT_H */
