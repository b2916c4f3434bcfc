// C ABI of libvrt.so (include/vrt.h): context, scene/camera upload, render calls.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <utility>
#include <vector>

#include "vrt_launch.h"

// Scenes with at most this many primitives are traced by the brute-force loop whose
// primitive data stays in scalar registers; larger ones get the LBVH.
static const uint32_t kBruteMaxPrims = 48;

struct vrt_ctx {
    int device = 0;
    std::string err;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr;
    bool pending = false; // a vrt_render_begin frame is in flight
    // scene
    float4 *d_sph = nullptr, *d_box = nullptr, *d_tri = nullptr, *d_mat = nullptr, *d_bvh = nullptr, *d_bvh4 = nullptr, *d_bvh_top = nullptr;
    int* d_prim_mat = nullptr;
    DevLight* d_lights = nullptr;
    float4 *d_aab = nullptr, *d_cam_pre = nullptr, *d_tex_data = nullptr;
    int* d_mat_tex = nullptr; int4* d_tex_info = nullptr; float2* d_tri_uv = nullptr;
    std::vector<float4> h_sph, h_box; // host copies for the camera-origin terms
    DevScene scene;
    bool have_scene = false, use_bvh = false;
    float bvh_ms = 0.f;
    int bvh_depth = 0;
    // camera
    DevCam cam;
    bool have_cam = false;
    // per-call buffers
    float2* d_offsets = nullptr; size_t offsets_cap = 0;
    unsigned long long* d_ctr = nullptr;       // counter block of the multi-kernel paths, cleared per call
    // The megakernel path takes its counter block from a ring of kCtrRing blocks that is cleared
    // with one fill every kCtrRing frames instead of one fill launch (~5 us) per frame.
    unsigned long long* d_ctr_ring = nullptr;
    unsigned ring_next = 0;
    unsigned long long* last_ctr = nullptr;    // block the last call counted into
    float4* d_rgba = nullptr; int* d_hit = nullptr; size_t fb_cap = 0;
    int* d_rec_pixels = nullptr; vrt_path_seg* d_segs = nullptr; int* d_seg_counts = nullptr;
    size_t rec_px_cap = 0, rec_seg_cap = 0;
    float4* d_px_rgba = nullptr; int* d_px_hit = nullptr;
    // wavefront workspace
    WfWorkspace wf; bool wf_two = false; bool wf_short = false;
    // this device's copy of the library's static scratch array (vrt_render_device_noalloc), the
    // page-locked staging its sampler table passes through and the event that says the staging is free again
    unsigned char* static_scratch = nullptr;
    float* static_stage = nullptr;
    hipEvent_t ev_static = nullptr; bool static_used = false;
    // last call
    uint64_t last_pixels = 0; int last_mode = 0;
    bool timed = false;
};

static const unsigned kCtrRing = 64;
static const size_t kCtrBlock = VRT_CTR_SLOTS * VRT_CTR_STRIDE; // 64-bit words per counter block

// Device memory of vrt_render_device_noalloc: every GPU the library runs on holds one copy of
// this array from the moment the code object is loaded until the process ends.  Layout: one
// counter block, then the sampler offset table.  Host code never uses the array's name as an
// address; vrt_create asks the runtime for the current device's copy (hipGetSymbolAddress) and
// keeps that pointer in the context.
static const size_t kStaticCtrBytes = kCtrBlock * sizeof(unsigned long long);
static const size_t kStaticScratchBytes = kStaticCtrBytes + (size_t)VRT_NOALLOC_MAX_OFFSETS * sizeof(float2);
static __device__ __align__(256) unsigned char g_vrt_static_scratch[kStaticScratchBytes];

static int fail(vrt_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(c, expr)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(c, VRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" int vrt_abi_version(void) { return VRT_ABI_VERSION; }

extern "C" const char* vrt_last_error(vrt_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" void vrt_shading_defaults(vrt_shading* s)
{
    if (!s) return;
    s->solid_ka = 0.5f; s->solid_kd = 0.5f; s->solid_ks = 0.f; s->solid_exp = 1.f;
    s->metal_ka = 0.25f; s->metal_kd = 0.5f; s->metal_ks = 0.15f; s->metal_exp = 100.f; s->metal_kr = 0.75f;
    s->metal_cr[0] = s->metal_cr[1] = s->metal_cr[2] = s->metal_cr[3] = 1.f;
    s->glass_ka = 0.5f; s->glass_kd = 0.5f; s->glass_ks = 0.2f; s->glass_exp = 1.f;
    s->glass_eta_in = 1.5f; s->glass_eta_out = 1.f;
    s->ao_min_amount[0] = s->ao_min_amount[1] = s->ao_min_amount[2] = 0.05f; s->ao_min_amount[3] = 1.f;
    s->ao_ls = 1.f;
    s->ao_color[0] = s->ao_color[1] = s->ao_color[2] = s->ao_color[3] = 1.f;
}

extern "C" void vrt_params_defaults(vrt_params* p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->spp = 1; p->n_sets = 1; p->max_depth = 10; p->range = 30.f;
    p->mode = VRT_MODE_AUTO; p->tile_count = 1; p->tile_rows = 8;
}

extern "C" int vrt_create(int device, vrt_ctx** out)
{
    if (!out) return VRT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VRT_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return VRT_ERR_INVALID;
    vrt_ctx* c = new vrt_ctx();
    c->device = device;
    memset(&c->scene, 0, sizeof c->scene);
    memset(&c->cam, 0, sizeof c->cam);
    memset(&c->wf, 0, sizeof c->wf);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess ||
        hipMalloc(&c->d_ctr, VRT_CTR_SLOTS * VRT_CTR_STRIDE * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&c->d_ctr_ring, kCtrRing * kCtrBlock * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&c->d_px_rgba, sizeof(float4)) != hipSuccess ||
        hipMalloc(&c->d_px_hit, sizeof(int)) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_static, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void**)&c->static_stage, (size_t)VRT_NOALLOC_MAX_OFFSETS * sizeof(float2), hipHostMallocDefault) != hipSuccess) {
        delete c;
        return VRT_ERR_HIP;
    }
    {   // this device's copy of the static array, asked for with this device current
        void* scratch = nullptr;
        if (hipGetSymbolAddress(&scratch, HIP_SYMBOL(g_vrt_static_scratch)) != hipSuccess || !scratch) {
            delete c;
            return VRT_ERR_HIP;
        }
        c->static_scratch = (unsigned char*)scratch;
    }
    *out = c;
    return VRT_OK;
}

static void free_scene(vrt_ctx* c)
{
    hipFree(c->d_sph); hipFree(c->d_box); hipFree(c->d_tri); hipFree(c->d_mat); hipFree(c->d_bvh);
    hipFree(c->d_bvh4); hipFree(c->d_bvh_top); c->d_bvh_top = nullptr; hipFree(c->d_prim_mat); hipFree(c->d_lights);
    hipFree(c->d_aab); hipFree(c->d_cam_pre); hipFree(c->d_tex_data); hipFree(c->d_mat_tex);
    hipFree(c->d_tex_info); hipFree(c->d_tri_uv);
    c->d_aab = c->d_cam_pre = c->d_tex_data = nullptr;
    c->d_mat_tex = nullptr; c->d_tex_info = nullptr; c->d_tri_uv = nullptr;
    c->d_lights = nullptr;
    c->d_sph = c->d_box = c->d_tri = c->d_mat = c->d_bvh = c->d_bvh4 = nullptr;
    c->d_prim_mat = nullptr;
    c->have_scene = false;
}

static void free_wf(vrt_ctx* c)
{
    hipFree(c->wf.arena); hipFree(c->wf.lacc); hipFree(c->wf.counts);
    hipFree(c->wf.hits); hipFree(c->wf.any.ox); hipFree(c->wf.any_counts);
    memset(&c->wf, 0, sizeof c->wf);
}

extern "C" int vrt_destroy(vrt_ctx* c)
{
    if (!c) return VRT_ERR_INVALID;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    free_scene(c);
    free_wf(c);
    hipFree(c->d_offsets); hipFree(c->d_ctr); hipFree(c->d_ctr_ring); hipFree(c->d_rgba); hipFree(c->d_hit);
    hipFree(c->d_rec_pixels); hipFree(c->d_segs); hipFree(c->d_seg_counts);
    hipFree(c->d_px_rgba); hipFree(c->d_px_hit);
    hipEventDestroy(c->ev0); hipEventDestroy(c->ev1); hipEventDestroy(c->ev_done);
    if (c->ev_static) hipEventDestroy(c->ev_static);
    if (c->static_stage) hipHostFree(c->static_stage);
    hipStreamDestroy(c->stream);
    delete c;
    return VRT_OK;
}

template <typename T>
static hipError_t upload(T** dst, const std::vector<T>& src)
{
    *dst = nullptr;
    if (src.empty()) return hipSuccess;
    hipError_t e = hipMalloc(dst, src.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
}

// A box qualifies for the world-axis test when each of its axes is exactly +-e_j.  Returns the
// code hit_aab() expects (k_0 | k_1<<2 | k_2<<4 | flip_j<<(8+j)) and the half extents in world
// order, or -1.
static int aab_code(const vrt_box& b, float hw[3])
{
    int code = 0, seen = 0;
    for (int k = 0; k < 3; ++k) {
        int j = -1;
        for (int q = 0; q < 3; ++q) {
            float v = b.axis[3 * k + q];
            if (v == 1.f || v == -1.f) { if (j >= 0) return -1; j = q; }
            else if (v != 0.f) return -1;
        }
        if (j < 0 || (seen & (1 << j))) return -1;
        seen |= 1 << j;
        code |= k << (2 * j);
        if (b.axis[3 * k + j] < 0.f) code |= 1 << (8 + j);
        hw[j] = b.half[k];
    }
    return code;
}

// Terms of the brute-force tests that depend only on the ray origin, for rays leaving the
// camera: float operations and their order are those of hit_sphere / the box loop.
static int refresh_cam_pre(vrt_ctx* c)
{
    if (!c->have_scene || !c->have_cam || c->use_bvh) return VRT_OK;
    const DevScene& S = c->scene;
    std::vector<float4> pre((size_t)S.n_sph + S.n_box);
    const float ox = c->cam.ox, oy = c->cam.oy, oz = c->cam.oz;
    for (int i = 0; i < S.n_sph; ++i) {
        float4 s = c->h_sph[i];
        float x = ox - s.x, y = oy - s.y, z = oz - s.z;
        float cc = (x * x + y * y + z * z) - s.w * s.w;
        pre[i] = make_float4(x, y, z, cc);
    }
    for (int i = 0; i < S.n_box; ++i) {
        float4 b0 = c->h_box[4 * (size_t)i];
        pre[S.n_sph + i] = make_float4(ox - b0.x, oy - b0.y, oz - b0.z, 0.f);
    }
    if (pre.empty()) return VRT_OK;
    if (!c->d_cam_pre) HIPCHK(c, hipMalloc(&c->d_cam_pre, pre.size() * sizeof(float4)));
    HIPCHK(c, hipMemcpy(c->d_cam_pre, pre.data(), pre.size() * sizeof(float4), hipMemcpyHostToDevice));
    c->scene.cam_pre = c->d_cam_pre;
    return VRT_OK;
}

extern "C" int vrt_set_scene(vrt_ctx* c, const vrt_scene_desc* d)
{
    if (!c || !d) return fail(c, VRT_ERR_INVALID, "null argument");
    if (c->pending) return fail(c, VRT_ERR_STATE, "a frame started with vrt_render_begin is still in flight");
    if ((d->n_spheres && !d->spheres) || (d->n_boxes && !d->boxes) ||
        (d->n_triangles && (!d->indices || !d->vertices)) || !d->n_materials || !d->materials ||
        (d->n_lights && !d->lights))
        return fail(c, VRT_ERR_INVALID, "scene arrays missing");
    if (d->n_lights > VRT_MAX_LIGHTS) return fail(c, VRT_ERR_CAPACITY, "more than VRT_MAX_LIGHTS lights");
    uint64_t n_prims64 = (uint64_t)d->n_spheres + d->n_boxes + d->n_triangles;
    if (n_prims64 == 0 || n_prims64 > 0x3fffffffu) return fail(c, VRT_ERR_INVALID, "primitive count out of range");
    uint32_t n_prims = (uint32_t)n_prims64;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));

    std::vector<float4> sph(d->n_spheres), box(4 * (size_t)d->n_boxes), tri(3 * (size_t)d->n_triangles),
        mat(d->n_materials);
    std::vector<int> pm(n_prims);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const vrt_material& m = d->materials[i];
        if (m.type < VRT_MAT_SOLID || m.type > VRT_MAT_DIELECTRIC) return fail(c, VRT_ERR_INVALID, "bad material type");
        float4 v; v.x = m.color[0]; v.y = m.color[1]; v.z = m.color[2];
        int t = m.type; memcpy(&v.w, &t, 4);
        mat[i] = v;
    }
    auto mat_ok = [&](int32_t m) { return m >= 0 && (uint32_t)m < d->n_materials; };
    for (uint32_t i = 0; i < d->n_spheres; ++i) {
        const vrt_sphere& s = d->spheres[i];
        if (!mat_ok(s.material)) return fail(c, VRT_ERR_INVALID, "sphere material out of range");
        sph[i] = make_float4(s.center[0], s.center[1], s.center[2], s.radius);
        pm[i] = s.material;
    }
    for (uint32_t i = 0; i < d->n_boxes; ++i) {
        const vrt_box& b = d->boxes[i];
        if (!mat_ok(b.material)) return fail(c, VRT_ERR_INVALID, "box material out of range");
        box[4 * i + 0] = make_float4(b.center[0], b.center[1], b.center[2], b.half[0]);
        box[4 * i + 1] = make_float4(b.axis[0], b.axis[1], b.axis[2], b.half[1]);
        box[4 * i + 2] = make_float4(b.axis[3], b.axis[4], b.axis[5], b.half[2]);
        // .w = squared circumradius, inflated; used by the brute-force loop's conservative cull
        double r2 = ((double)b.half[0] * b.half[0] + (double)b.half[1] * b.half[1] + (double)b.half[2] * b.half[2]) * 1.001;
        box[4 * i + 3] = make_float4(b.axis[6], b.axis[7], b.axis[8], (float)r2);
        pm[d->n_spheres + i] = b.material;
    }
    for (uint32_t i = 0; i < d->n_triangles; ++i) {
        const uint32_t* ix = d->indices + 3 * (size_t)i;
        if (ix[0] >= d->n_vertices || ix[1] >= d->n_vertices || ix[2] >= d->n_vertices)
            return fail(c, VRT_ERR_INVALID, "triangle index out of range");
        const float* a = d->vertices + 3 * (size_t)ix[0];
        const float* b = d->vertices + 3 * (size_t)ix[1];
        const float* cc = d->vertices + 3 * (size_t)ix[2];
        int32_t m = d->tri_material ? d->tri_material[i] : 0;
        if (!mat_ok(m)) return fail(c, VRT_ERR_INVALID, "triangle material out of range");
        // edges in float32, one rounding each, exactly as a per-ray v1 - v0 would give
        tri[3 * (size_t)i + 0] = make_float4(a[0], a[1], a[2], 0.f);
        tri[3 * (size_t)i + 1] = make_float4(b[0] - a[0], b[1] - a[1], b[2] - a[2], 0.f);
        tri[3 * (size_t)i + 2] = make_float4(cc[0] - a[0], cc[1] - a[1], cc[2] - a[2], 0.f);
        pm[d->n_spheres + d->n_boxes + i] = m;
    }

    for (uint32_t i = 0; i < d->n_lights; ++i)   // everything is validated before the old scene is dropped
        if (d->lights[i].type != VRT_LIGHT_DIRECTIONAL && d->lights[i].type != VRT_LIGHT_POINT)
            return fail(c, VRT_ERR_INVALID, "bad light type");

    // textures (Tracer/WhittedTracer.cs:55-63)
    std::vector<int> mat_tex;
    std::vector<int4> tex_info;
    std::vector<float4> tex_data;
    std::vector<float2> tri_uv;
    if (d->n_textures && d->material_texture) {
        if (!d->textures) return fail(c, VRT_ERR_INVALID, "texture array missing");
        for (uint32_t i = 0; i < d->n_textures; ++i) {
            const vrt_texture& t = d->textures[i];
            if (t.width <= 0 || t.height <= 0 || !t.rgba || (t.wrap != 0 && t.wrap != 1) ||
                (uint64_t)t.width * t.height + tex_data.size() > 0x7fffffffull)
                return fail(c, VRT_ERR_INVALID, "bad texture");
            int4 info; info.x = (int)tex_data.size(); info.y = t.width; info.z = t.height; info.w = t.wrap;
            tex_info.push_back(info);
            size_t n = (size_t)t.width * t.height;
            for (size_t k = 0; k < n; ++k)
                tex_data.push_back(make_float4(t.rgba[4 * k], t.rgba[4 * k + 1], t.rgba[4 * k + 2], t.rgba[4 * k + 3]));
        }
        bool any = false;
        mat_tex.resize(d->n_materials);
        for (uint32_t i = 0; i < d->n_materials; ++i) {
            int tx = d->material_texture[i];
            if (tx >= (int)d->n_textures) return fail(c, VRT_ERR_INVALID, "material texture out of range");
            mat_tex[i] = tx < 0 ? -1 : tx;
            any = any || tx >= 0;
        }
        if (!any) mat_tex.clear();
        if (any && d->uvs && d->n_triangles) {
            tri_uv.resize(3 * (size_t)d->n_triangles);
            for (uint32_t i = 0; i < d->n_triangles; ++i)
                for (int k = 0; k < 3; ++k) {
                    const float* uv = d->uvs + 2 * (size_t)d->indices[3 * (size_t)i + k];
                    tri_uv[3 * (size_t)i + k] = make_float2(uv[0], uv[1]);
                }
        }
    }

    free_scene(c);
    HIPCHK(c, upload(&c->d_sph, sph));
    HIPCHK(c, upload(&c->d_box, box));
    HIPCHK(c, upload(&c->d_tri, tri));
    HIPCHK(c, upload(&c->d_mat, mat));
    HIPCHK(c, upload(&c->d_prim_mat, pm));

    DevScene& S = c->scene;
    memset(&S, 0, sizeof S);
    S.n_sph = (int)d->n_spheres; S.n_box = (int)d->n_boxes; S.n_tri = (int)d->n_triangles;
    S.n_prims = (int)n_prims;
    S.sph = c->d_sph; S.box = c->d_box; S.tri = c->d_tri; S.prim_mat = c->d_prim_mat; S.mat = c->d_mat;
    if (!mat_tex.empty()) {
        HIPCHK(c, upload(&c->d_mat_tex, mat_tex));
        HIPCHK(c, upload(&c->d_tex_info, tex_info));
        HIPCHK(c, upload(&c->d_tex_data, tex_data));
        HIPCHK(c, upload(&c->d_tri_uv, tri_uv));
        S.mat_tex = c->d_mat_tex; S.tex_info = c->d_tex_info; S.tex_data = c->d_tex_data; S.tri_uv = c->d_tri_uv;
    }
    S.n_lights = (int)d->n_lights;
    std::vector<DevLight> lights(d->n_lights ? d->n_lights : 1);
    memset(lights.data(), 0, lights.size() * sizeof(DevLight));
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        const vrt_light& l = d->lights[i];
        DevLight& o = lights[i];
        o.type = l.type; o.vx = l.vec[0]; o.vy = l.vec[1]; o.vz = l.vec[2]; o.ls = l.ls;
        o.cr = l.color[0]; o.cg = l.color[1]; o.cb = l.color[2]; o.cast = l.cast_shadows;
        o.lx = l.location[0]; o.ly = l.location[1]; o.lz = l.location[2];
    }
    HIPCHK(c, upload(&c->d_lights, lights));
    S.lights = c->d_lights;
    S.sh = d->shading;
    S.dbg = c->d_ctr;

    c->use_bvh = n_prims > kBruteMaxPrims;
    c->bvh_ms = 0.f; c->bvh_depth = 0;
    if (c->use_bvh) {
        // Up to VRT_MAX_BIG primitives whose box is at least half as long as the scene's stay
        // out of the tree (spheres and boxes only; a mesh is expected to be tessellated).
        // VRT_NO_BIG=1 keeps them in, for study.
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        std::vector<std::pair<double, int>> ext(d->n_spheres + d->n_boxes);
        auto grow = [&](const double l[3], const double h[3]) {
            for (int k = 0; k < 3; ++k) { if (l[k] < lo[k]) lo[k] = l[k]; if (h[k] > hi[k]) hi[k] = h[k]; }
        };
        for (uint32_t i = 0; i < d->n_spheres; ++i) {
            const vrt_sphere& s = d->spheres[i];
            double l[3], h[3];
            for (int k = 0; k < 3; ++k) { l[k] = s.center[k] - s.radius; h[k] = s.center[k] + s.radius; }
            grow(l, h);
            ext[i] = {2.0 * s.radius, (int)i};
        }
        for (uint32_t i = 0; i < d->n_boxes; ++i) {
            const vrt_box& b = d->boxes[i];
            double l[3], h[3], m = 0;
            for (int k = 0; k < 3; ++k) {
                double e = fabs(b.axis[k]) * b.half[0] + fabs(b.axis[3 + k]) * b.half[1] + fabs(b.axis[6 + k]) * b.half[2];
                l[k] = b.center[k] - e; h[k] = b.center[k] + e;
                if (2 * e > m) m = 2 * e;
            }
            grow(l, h);
            ext[d->n_spheres + i] = {m, (int)(d->n_spheres + i)};
        }
        for (uint32_t i = 0; i < d->n_vertices; ++i) {
            double v[3] = {d->vertices[3 * (size_t)i], d->vertices[3 * (size_t)i + 1], d->vertices[3 * (size_t)i + 2]};
            grow(v, v);
        }
        double scene_ext = 0;
        for (int k = 0; k < 3; ++k) if (hi[k] - lo[k] > scene_ext) scene_ext = hi[k] - lo[k];
        std::sort(ext.begin(), ext.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) {
            return a.first > b.first || (a.first == b.first && a.second < b.second);
        });
        const char* nb = getenv("VRT_NO_BIG");
        if (!(nb && atoi(nb)))
            for (size_t i = 0; i < ext.size() && S.n_big < VRT_MAX_BIG && ext[i].first >= 0.5 * scene_ext; ++i)
                S.big[S.n_big++] = ext[i].second;
        HIPCHK(c, hipMalloc(&c->d_bvh, 4 * (size_t)(n_prims - 1) * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_bvh4, 8 * (size_t)(n_prims - 1) * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_bvh_top, 8 * VRT_TOP_NODES * sizeof(float4)));
        S.bvh = c->d_bvh; S.bvh4 = c->d_bvh4; S.bvh_top = c->d_bvh_top;
        hipError_t e = vrt_build_lbvh(S, c->d_bvh, c->d_bvh4, c->d_bvh_top, &c->bvh_depth, &c->bvh_ms, c->stream);
        if (e != hipSuccess) return fail(c, VRT_ERR_HIP, std::string("LBVH build: ") + hipGetErrorString(e));
        if (c->bvh_depth > VRT_BVH_STACK - 2)
            return fail(c, VRT_ERR_CAPACITY, "LBVH deeper than the traversal stack");
    }
    if (!c->use_bvh) {
        c->h_sph = sph; c->h_box = box;
        std::vector<float4> aab(2 * (size_t)d->n_boxes);
        bool all = d->n_boxes > 0;
        for (uint32_t i = 0; i < d->n_boxes && all; ++i) {
            float hw[3];
            int code = aab_code(d->boxes[i], hw);
            if (code < 0) { all = false; break; }
            float4 a0 = box[4 * (size_t)i];
            memcpy(&a0.w, &code, 4);
            aab[2 * (size_t)i] = a0;
            aab[2 * (size_t)i + 1] = make_float4(hw[0], hw[1], hw[2], box[4 * (size_t)i + 3].w);
        }
        if (all) {
            HIPCHK(c, upload(&c->d_aab, aab));
            S.aab = c->d_aab; S.all_aab = 1;
        }
    }
    c->have_scene = true;
    return refresh_cam_pre(c);
}

extern "C" int vrt_set_camera(vrt_ctx* c, const vrt_camera* cam)
{
    if (!c || !cam) return fail(c, VRT_ERR_INVALID, "null argument");
    if (c->pending) return fail(c, VRT_ERR_STATE, "a frame started with vrt_render_begin is still in flight");
    if (cam->width <= 0 || cam->height <= 0 || (int64_t)cam->width * cam->height > (1ll << 30))
        return fail(c, VRT_ERR_INVALID, "frame size out of range");
    DevCam& d = c->cam;
    d.ox = cam->origin[0]; d.oy = cam->origin[1]; d.oz = cam->origin[2];
    // PlaneBorderPoints[0] - rayOrigin, rounded once like the C# Vector3 subtraction
    d.bx = cam->p0[0] - cam->origin[0]; d.by = cam->p0[1] - cam->origin[1]; d.bz = cam->p0[2] - cam->origin[2];
    d.xx = cam->x_axis[0]; d.xy = cam->x_axis[1]; d.xz = cam->x_axis[2];
    d.yx = cam->y_axis[0]; d.yy = cam->y_axis[1]; d.yz = cam->y_axis[2];
    d.width = cam->width; d.height = cam->height;
    c->have_cam = true;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize()); // frames in flight still read the old terms
    return refresh_cam_pre(c);
}

static int owned_rows(int H, int index, int count, int rows)
{
    if (count <= 1) return H;
    int n = 0;
    for (int s = index; s * rows < H; s += count) {
        int a = s * rows, b = a + rows < H ? a + rows : H;
        n += b - a;
    }
    return n;
}

// Validates params and fills DevParams (everything but the offset table pointer).
static int check_params(vrt_ctx* c, const vrt_params* p, DevParams* out)
{
    if (!c->have_scene || !c->have_cam) return fail(c, VRT_ERR_STATE, "scene or camera not set");
    if (p->spp < 1 || p->spp > 65536) return fail(c, VRT_ERR_INVALID, "spp out of range");
    if (p->max_depth < 0 || p->max_depth > VRT_MAX_DEPTH) return fail(c, VRT_ERR_CAPACITY, "max_depth out of range");
    if (!(p->range > 0.f)) return fail(c, VRT_ERR_INVALID, "range must be positive");
    if (p->offsets && p->n_sets < 1) return fail(c, VRT_ERR_INVALID, "n_sets must be >= 1");
    int tcount = p->tile_count > 1 ? p->tile_count : 1;
    if (tcount > 1 && (p->tile_index < 0 || p->tile_index >= tcount || p->tile_rows < 1))
        return fail(c, VRT_ERR_INVALID, "bad tile parameters");
    DevParams& P = *out;
    memset(&P, 0, sizeof P);
    P.spp = p->spp; P.n_sets = p->offsets ? p->n_sets : 1; P.seed = p->seed;
    P.max_depth = p->max_depth; P.range = p->range;
    P.flags = p->flags & ~(unsigned)VRT_OUT_LINEAR_RGB;   // an output format only vrt_render_device offers
    P.tile_index = tcount > 1 ? p->tile_index : 0; P.tile_count = tcount;
    P.tile_rows = p->tile_rows > 0 ? p->tile_rows : 8;
    P.local_rows = owned_rows(c->cam.height, P.tile_index, tcount, P.tile_rows);
    P.compact = p->compact_rows != 0;
    // Measured on MI355X: giving each XCD one contiguous band of the image costs 1.4-1.7x on
    // every workload (bands differ in work, the slowest XCD sets the frame time), so blocks
    // keep the hardware's round-robin deal; VRT_XCD_REMAP=1 re-enables the banding for study.
    P.xcd_remap = 0;
    if (const char* e = getenv("VRT_XCD_REMAP")) P.xcd_remap = atoi(e);
    // Megakernel lane split: the largest power of two <= min(spp, 64) sample lanes per pixel that
    // leaves at most 15 % of the lane-rounds idle; VRT_SAMPLE_SHIFT overrides it for study.
    P.sample_shift = 0;
    for (int sh = 6; sh >= 1; --sh) {
        int n = 1 << sh;
        if (n > p->spp) continue;
        int rounds = (p->spp + n - 1) / n;
        if ((double)rounds * n <= 1.15 * p->spp) { P.sample_shift = sh; break; }
    }
    if (const char* e = getenv("VRT_SAMPLE_SHIFT")) {
        int sh = atoi(e);
        if (sh >= 0 && sh <= 6) P.sample_shift = sh;
    }
    P.offsets = nullptr;
    return VRT_OK;
}

// check_params + the offset table uploaded to the context's buffer.
static int prepare(vrt_ctx* c, const vrt_params* p, DevParams* out)
{
    int rc = check_params(c, p, out);
    if (rc) return rc;
    DevParams& P = *out;
    if (p->offsets) {
        size_t n = (size_t)p->n_sets * p->spp;
        if (n > c->offsets_cap) {
            HIPCHK(c, hipFree(c->d_offsets));
            c->d_offsets = nullptr; c->offsets_cap = 0;
            HIPCHK(c, hipMalloc(&c->d_offsets, n * sizeof(float2)));
            c->offsets_cap = n;
        }
        HIPCHK(c, hipMemcpyAsync(c->d_offsets, p->offsets, n * sizeof(float2), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream)); // the caller's table may be reused right away
        P.offsets = c->d_offsets;
    }
    return VRT_OK;
}

// Sample slots the wavefront workspace holds at most: a band of rows of the frame is in flight
// at a time.  64 Mi slots are 1 GiB of radiance slots, 7.5 GiB of ray queues and 3.75 GiB for the
// deferred occlusion queries (queues twice that with two children per hit); a frame with fewer samples gets a workspace of its own size.
static const size_t kWfMaxSlots = 64ull << 20;

static int ensure_wf(vrt_ctx* c, size_t slots, bool two, hipStream_t stream)
{
    // VRT_WF_DEFER=0 (study, tests): occlusion queries of the bounces traced where they arise
    const char* e = getenv("VRT_WF_DEFER");
    const bool defer = !(e && e[0] == '0');
    if (c->wf.slots >= slots && (c->wf_two || !two) && (c->wf.hits != nullptr) == defer && !c->wf_short && !getenv("VRT_WF_ANY_CAP")) return VRT_OK;
    free_wf(c);
    c->wf_short = false;
    size_t cap = slots * (two ? 2 : 1) + VRT_WF_SLACK;
    HIPCHK(c, hipMalloc(&c->wf.arena, cap * 2 * VRT_QUEUE_ARRAYS * sizeof(float)));
    for (int k = 0; k < 2; ++k) {
        float* b = c->wf.arena + (size_t)k * cap * VRT_QUEUE_ARRAYS;
        RayQueue& q = c->wf.q[k];
        q.ox = b; q.oy = b + cap; q.oz = b + 2 * cap; q.dx = b + 3 * cap; q.dy = b + 4 * cap; q.dz = b + 5 * cap;
        q.tr = b + 6 * cap; q.tg = b + 7 * cap; q.tb = b + 8 * cap;
        q.lr = b + 9 * cap; q.lg = b + 10 * cap; q.lb = b + 11 * cap;
        q.pix = (int*)(b + 12 * cap); q.skip = (int*)(b + 13 * cap); q.meta = (int*)(b + 14 * cap);
        HIPCHK(c, hipMemsetAsync(q.pix, 0xff, cap * sizeof(int), stream));   // "unwritten" marks; consumers put them back
    }
    c->wf.cap = cap;
    HIPCHK(c, hipMalloc(&c->wf.lacc, slots * sizeof(float4)));
    HIPCHK(c, hipMalloc(&c->wf.counts, (VRT_MAX_DEPTH + 3) * sizeof(int)));
    if (defer) {
        // one flag word per queued ray, one query-queue entry per queued ray (a hit asks up to
        // 1 + n_lights questions but most ask fewer than one on average; the rest is traced in place)
        HIPCHK(c, hipMalloc(&c->wf.hits, cap * sizeof(int4)));
        float* b = nullptr;
        HIPCHK(c, hipMalloc(&b, cap * VRT_ANYQ_ARRAYS * sizeof(float)));
        AnyQueue& a = c->wf.any;
        a.ox = b; a.oy = b + cap; a.oz = b + 2 * cap; a.dx = b + 3 * cap; a.dy = b + 4 * cap; a.dz = b + 5 * cap;
        a.tlim = b + 6 * cap; a.maxd = b + 7 * cap;
        a.skip = (int*)(b + 8 * cap); a.target = (int*)(b + 9 * cap); a.bit = (int*)(b + 10 * cap);
        HIPCHK(c, hipMemsetAsync(a.target, 0xff, cap * sizeof(int), stream));
        c->wf.any_cap = cap;
        if (const char* v = getenv("VRT_WF_ANY_CAP")) {   // tests: a short queue forces the in-place path
            long long n = atoll(v);
            if (n >= 0 && (size_t)n < cap) { c->wf.any_cap = (size_t)n; c->wf_short = true; }
        }
        HIPCHK(c, hipMalloc(&c->wf.any_counts, 3 * (VRT_MAX_DEPTH + 2) * sizeof(int)));
        c->wf.cursors = c->wf.any_counts + (VRT_MAX_DEPTH + 2);
    }
    c->wf.slots = slots; c->wf_two = two;
    return VRT_OK;
}

// Enqueues one frame on `stream`.
// `ctr_in` = a counter block to count into, already clear (megakernel only), or null for the
// path's own block.
static int enqueue_frame(vrt_ctx* c, const vrt_params* p, const DevParams& P, float4* rgba, int* hit,
                         hipStream_t stream, unsigned long long* ctr_in = nullptr)
{
    bool two = (p->flags & VRT_FIX_DIELECTRIC) != 0;
    int mode = p->mode == VRT_MODE_WAVEFRONT ? VRT_MODE_WAVEFRONT : VRT_MODE_MEGAKERNEL;
    if (mode == VRT_MODE_WAVEFRONT) {
        // whole frame if it fits the slot budget, else bands of rows of at least one tile row
        size_t per_row = (size_t)c->cam.width * P.spp;
        size_t want = per_row * (size_t)(P.local_rows > 0 ? P.local_rows : 1);
        size_t min_band = per_row * 8;
        size_t max_slots = kWfMaxSlots;
        if (const char* e = getenv("VRT_WF_MAX_SLOTS")) { long long v = atoll(e); if (v > 0) max_slots = (size_t)v; } // tests: force bands
        if (want > max_slots) want = max_slots > min_band ? max_slots : min_band;
        if (want < min_band) want = min_band;
        if (want * (two ? 2 : 1) + 2 * VRT_WF_SLACK > 0x7fffffffull) return fail(c, VRT_ERR_CAPACITY, "frame row too long for the wavefront queues");
        int rc = ensure_wf(c, want, two, stream);   // its fills run on the frame's stream, ahead of the kernels
        if (rc) return rc;
        WfWorkspace wf = c->wf;
        wf.slots = want; // a workspace kept from a larger frame still runs this frame in bands of `want`
        const char* r = getenv("VRT_WF_REFILL");   // 0 (study, tests): queues traced one entry per lane and pass
        wf.refill = !(r && r[0] == '0');
        HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, kCtrBlock * sizeof(unsigned long long), stream));
        c->scene.dbg = c->d_ctr; c->last_ctr = c->d_ctr;
        HIPCHK(c, vrt_launch_wavefront(c->scene, c->cam, P, c->use_bvh, two, wf, rgba, hit, c->d_ctr, stream));
    } else {
        unsigned long long* ctr = ctr_in;
        if (!ctr) {
            const unsigned slot = c->ring_next % kCtrRing;
            if (slot == 0)   // frames on one stream run in order, so every block is clean when its turn comes
                HIPCHK(c, hipMemsetAsync(c->d_ctr_ring, 0, kCtrRing * kCtrBlock * sizeof(unsigned long long), stream));
            c->ring_next++;
            ctr = c->d_ctr_ring + (size_t)slot * kCtrBlock;
        }
        c->scene.dbg = ctr; c->last_ctr = ctr;
        HIPCHK(c, vrt_launch_mega(c->scene, c->cam, P, c->use_bvh, two, rgba, hit, ctr, stream));
    }
    c->last_pixels = (uint64_t)P.local_rows * c->cam.width;
    c->last_mode = mode;
    return VRT_OK;
}

static int read_stats(vrt_ctx* c, vrt_stats* s)
{
    std::vector<unsigned long long> h(VRT_CTR_SLOTS * VRT_CTR_STRIDE);
    HIPCHK(c, hipMemcpy(h.data(), c->last_ctr ? c->last_ctr : c->d_ctr, h.size() * sizeof(unsigned long long),
                        hipMemcpyDeviceToHost));
    memset(s, 0, sizeof *s);
    for (int i = 0; i < VRT_CTR_SLOTS; ++i)
        for (int k = 0; k < VRT_RAY_CLASSES; ++k) s->rays[k] += h[i * VRT_CTR_STRIDE + k];
    for (int i = 0; i < VRT_CTR_SLOTS; ++i) {
        s->nodes_visited += h[i * VRT_CTR_STRIDE + 6];
        s->prims_tested += h[i * VRT_CTR_STRIDE + 7];
    }
    s->pixels = c->last_pixels;
    s->bvh_build_ms = c->bvh_ms;
    s->bvh_nodes = c->use_bvh ? (uint32_t)(c->scene.n_prims - 1) : 0;
    s->mode_used = c->last_mode;
    s->gpu_ms = 0.f;
    if (c->last_mode == VRT_MODE_WAVEFRONT && c->wf.counts) {
        int overflow = 0;
        HIPCHK(c, hipMemcpy(&overflow, c->wf.counts + VRT_MAX_DEPTH + 2, sizeof(int), hipMemcpyDeviceToHost));
        if (overflow)
            return fail(c, VRT_ERR_CAPACITY, "wavefront ray queue overflowed (two-child refraction); "
                                             "use VRT_MODE_MEGAKERNEL for this scene");
    }
    if (c->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) s->gpu_ms = ms;
    }
    return VRT_OK;
}

static int run_records(vrt_ctx* c, const vrt_params* p, const DevParams& P, const int* pixels, int n, int cap,
                       int sample_sel, vrt_path_seg* out, int* counts_out, float4* rgba_dev, int* hit_dev,
                       unsigned long long* ctr)
{
    if (n <= 0) return VRT_OK;
    if (cap < 0) return fail(c, VRT_ERR_INVALID, "negative record capacity");
    size_t total = c->cam.width * (size_t)c->cam.height;
    for (int i = 0; i < n; ++i)
        if (pixels[i] < 0 || (size_t)pixels[i] >= total) return fail(c, VRT_ERR_INVALID, "record pixel out of range");
    if ((size_t)n > c->rec_px_cap) {
        hipFree(c->d_rec_pixels); hipFree(c->d_seg_counts);
        c->d_rec_pixels = nullptr; c->d_seg_counts = nullptr; c->rec_px_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_rec_pixels, n * sizeof(int)));
        HIPCHK(c, hipMalloc(&c->d_seg_counts, n * sizeof(int)));
        c->rec_px_cap = n;
    }
    size_t nseg = (size_t)n * (cap > 0 ? cap : 1);
    if (nseg > c->rec_seg_cap) {
        hipFree(c->d_segs); c->d_segs = nullptr; c->rec_seg_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_segs, nseg * sizeof(vrt_path_seg)));
        c->rec_seg_cap = nseg;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_rec_pixels, pixels, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    bool two = (p->flags & VRT_FIX_DIELECTRIC) != 0;
    HIPCHK(c, vrt_launch_pixels(c->scene, c->cam, P, c->use_bvh, two, c->d_rec_pixels, n, sample_sel, c->d_segs,
                                cap, c->d_seg_counts, rgba_dev, hit_dev, ctr, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (counts_out) HIPCHK(c, hipMemcpy(counts_out, c->d_seg_counts, n * sizeof(int), hipMemcpyDeviceToHost));
    if (out && cap > 0)
        HIPCHK(c, hipMemcpy(out, c->d_segs, (size_t)n * cap * sizeof(vrt_path_seg), hipMemcpyDeviceToHost));
    return VRT_OK;
}

// Enqueues one frame and its copies to the caller's buffers on the context's stream.
static int begin_frame(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out, DevParams* P_out)
{
    if (c->pending) return fail(c, VRT_ERR_STATE, "a frame started with vrt_render_begin is still in flight");
    if (p->flags & VRT_OUT_LINEAR_RGB) return fail(c, VRT_ERR_INVALID, "VRT_OUT_LINEAR_RGB is an output format of vrt_render_device");
    HIPCHK(c, hipSetDevice(c->device));
    DevParams& P = *P_out;
    int rc = prepare(c, p, &P);
    if (rc) return rc;
    size_t npix = (size_t)c->cam.width * c->cam.height;
    if (npix > c->fb_cap) {
        hipFree(c->d_rgba); hipFree(c->d_hit);
        c->d_rgba = nullptr; c->d_hit = nullptr; c->fb_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_rgba, npix * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_hit, npix * sizeof(int)));
        c->fb_cap = npix;
    }
    size_t nout = P.compact ? (size_t)P.local_rows * c->cam.width : npix;
    if (P.tile_count > 1 && !P.compact) { // rows of other ranks stay as the caller left them
        HIPCHK(c, hipMemcpyAsync(c->d_rgba, rgba_out, npix * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        if (hit_id_out)
            HIPCHK(c, hipMemcpyAsync(c->d_hit, hit_id_out, npix * sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    rc = enqueue_frame(c, p, P, c->d_rgba, c->d_hit, c->stream);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    HIPCHK(c, hipMemcpyAsync(rgba_out, c->d_rgba, nout * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (hit_id_out)
        HIPCHK(c, hipMemcpyAsync(hit_id_out, c->d_hit, nout * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    return VRT_OK;
}

extern "C" int vrt_render(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out, vrt_stats* stats_out)
{
    if (!c || !p || !rgba_out) return fail(c, VRT_ERR_INVALID, "null argument");
    if (p->n_record_pixels < 0 || (p->n_record_pixels > 0 && (!p->record_pixels || !p->record_count_out)))
        return fail(c, VRT_ERR_INVALID, "record arguments missing");
    DevParams P;
    int rc = begin_frame(c, p, rgba_out, hit_id_out, &P);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (stats_out) { rc = read_stats(c, stats_out); if (rc) return rc; }
    if (p->n_record_pixels > 0) {
        rc = run_records(c, p, P, p->record_pixels, p->n_record_pixels, p->record_cap, -1, p->record_out,
                         p->record_count_out, nullptr, nullptr, nullptr);
        if (rc) return rc;
    }
    return VRT_OK;
}

extern "C" int vrt_render_begin(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out)
{
    if (!c || !p || !rgba_out) return fail(c, VRT_ERR_INVALID, "null argument");
    if (p->n_record_pixels != 0) return fail(c, VRT_ERR_INVALID, "path recording needs the blocking vrt_render");
    DevParams P;
    int rc = begin_frame(c, p, rgba_out, hit_id_out, &P);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev_done, c->stream));
    c->pending = true;
    return VRT_OK;
}

extern "C" int vrt_render_poll(vrt_ctx* c, int* done)
{
    if (!c || !done) return fail(c, VRT_ERR_INVALID, "null argument");
    if (!c->pending) return fail(c, VRT_ERR_STATE, "no frame in flight");
    HIPCHK(c, hipSetDevice(c->device));
    hipError_t e = hipEventQuery(c->ev_done);
    if (e == hipSuccess) { *done = 1; return VRT_OK; }
    if (e == hipErrorNotReady) { *done = 0; (void)hipGetLastError(); return VRT_OK; }
    return fail(c, VRT_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
}

extern "C" int vrt_render_end(vrt_ctx* c, vrt_stats* stats_out)
{
    if (!c) return VRT_ERR_INVALID;
    if (!c->pending) return fail(c, VRT_ERR_STATE, "no frame in flight");
    HIPCHK(c, hipSetDevice(c->device));
    c->pending = false;
    HIPCHK(c, hipEventSynchronize(c->ev_done));
    if (stats_out) return read_stats(c, stats_out);
    return VRT_OK;
}

extern "C" int vrt_render_device(vrt_ctx* c, const vrt_params* p, void* rgba_dev, void* hit_id_dev, void* stream)
{
    if (!c || !p || !rgba_dev) return fail(c, VRT_ERR_INVALID, "null argument");
    if (c->pending) return fail(c, VRT_ERR_STATE, "a frame started with vrt_render_begin is still in flight");
    HIPCHK(c, hipSetDevice(c->device));
    DevParams P;
    int rc = prepare(c, p, &P);
    if (rc) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    c->timed = false;
    P.flags |= p->flags & VRT_OUT_LINEAR_RGB;
    return enqueue_frame(c, p, P, (float4*)rgba_dev, (int*)hit_id_dev, s);
}

// vrt_render_device for callers that cannot allocate: the megakernel counts into this device's
// copy of the library's static scratch array and samples with a copy of the table kept there
// too.  The array's address on this device was taken in vrt_create; nothing below allocates or frees.
extern "C" int vrt_render_device_noalloc(vrt_ctx* c, const vrt_params* p, void* rgba_dev, void* hit_id_dev, void* stream)
{
    if (!c || !p || !rgba_dev) return fail(c, VRT_ERR_INVALID, "null argument");
    if (c->pending) return fail(c, VRT_ERR_STATE, "a frame started with vrt_render_begin is still in flight");
    if (p->mode == VRT_MODE_WAVEFRONT)
        return fail(c, VRT_ERR_INVALID, "vrt_render_device_noalloc runs the megakernel; the wavefront queues need an allocated workspace");
    const size_t n_off = p->offsets ? (size_t)(p->n_sets > 0 ? p->n_sets : 0) * (size_t)(p->spp > 0 ? p->spp : 0) : 0;
    if (n_off > (size_t)VRT_NOALLOC_MAX_OFFSETS)
        return fail(c, VRT_ERR_CAPACITY, "sampler table larger than VRT_NOALLOC_MAX_OFFSETS entries (n_sets * spp)");
    HIPCHK(c, hipSetDevice(c->device));
    DevParams P;
    int rc = check_params(c, p, &P);
    if (rc) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    unsigned long long* ctr = (unsigned long long*)c->static_scratch;
    float2* offs = (float2*)(c->static_scratch + kStaticCtrBytes);
    if (n_off) {
        // through the context's page-locked staging, once the copy of the previous call has left it
        if (c->static_used) HIPCHK(c, hipEventSynchronize(c->ev_static));
        memcpy(c->static_stage, p->offsets, n_off * sizeof(float2));
        HIPCHK(c, hipMemcpyAsync(offs, c->static_stage, n_off * sizeof(float2), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipEventRecord(c->ev_static, s));
        c->static_used = true;
        P.offsets = offs;
    }
    HIPCHK(c, hipMemsetAsync(ctr, 0, kStaticCtrBytes, s));   // the array keeps the last call's counts
    c->timed = false;
    P.flags |= p->flags & VRT_OUT_LINEAR_RGB;
    return enqueue_frame(c, p, P, (float4*)rgba_dev, (int*)hit_id_dev, s, ctr);
}

extern "C" int vrt_tonemap_device(vrt_ctx* c, const void* rgb_dev, void* rgba_dev, uint64_t n_pixels, void* stream)
{
    if (!c || (n_pixels && (!rgb_dev || !rgba_dev))) return fail(c, VRT_ERR_INVALID, "null argument");
    if (n_pixels > (1ull << 40)) return fail(c, VRT_ERR_INVALID, "pixel count out of range");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    HIPCHK(c, vrt_launch_tonemap((const float*)rgb_dev, (float4*)rgba_dev, (size_t)n_pixels, s));
    return VRT_OK;
}

extern "C" int vrt_fetch_stats(vrt_ctx* c, vrt_stats* stats_out)
{
    if (!c || !stats_out) return fail(c, VRT_ERR_INVALID, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    return read_stats(c, stats_out);
}

extern "C" int vrt_trace_pixel(vrt_ctx* c, const vrt_params* p, int h, int v, int sample, vrt_path_seg* segs,
                               int cap, int* n_out, float rgba[4], int32_t* hit_id_out)
{
    if (!c || !p || !rgba) return fail(c, VRT_ERR_INVALID, "null argument");
    if (c->pending) return fail(c, VRT_ERR_STATE, "a frame started with vrt_render_begin is still in flight");
    HIPCHK(c, hipSetDevice(c->device));
    DevParams P;
    int rc = prepare(c, p, &P);
    if (rc) return rc;
    if (h < 0 || h >= c->cam.height || v < 0 || v >= c->cam.width) return fail(c, VRT_ERR_INVALID, "pixel out of range");
    if (sample >= p->spp) return fail(c, VRT_ERR_INVALID, "sample out of range");
    if (cap < 0 || (cap > 0 && !segs)) return fail(c, VRT_ERR_INVALID, "bad record buffer");
    int pix = h * c->cam.width + v, n = 0;
    HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, VRT_CTR_SLOTS * VRT_CTR_STRIDE * sizeof(unsigned long long), c->stream));
    c->scene.dbg = c->d_ctr; c->last_ctr = c->d_ctr;
    rc = run_records(c, p, P, &pix, 1, cap, sample < 0 ? -1 : sample, segs, &n, c->d_px_rgba, c->d_px_hit, c->d_ctr);
    if (rc) return rc;
    float4 col; int id;
    HIPCHK(c, hipMemcpy(&col, c->d_px_rgba, sizeof col, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&id, c->d_px_hit, sizeof id, hipMemcpyDeviceToHost));
    rgba[0] = col.x; rgba[1] = col.y; rgba[2] = col.z; rgba[3] = col.w;
    if (hit_id_out) *hit_id_out = id;
    if (n_out) *n_out = n;
    c->last_pixels = 1; c->last_mode = VRT_MODE_MEGAKERNEL; c->timed = false;
    return VRT_OK;
}

extern "C" int vrt_trace_ray(vrt_ctx* c, const vrt_params* p, const float origin[3], const float direction[3],
                             int depth, float rgba[4], int32_t* hit_id_out)
{
    if (!c || !p || !origin || !direction || !rgba) return fail(c, VRT_ERR_INVALID, "null argument");
    if (c->pending) return fail(c, VRT_ERR_STATE, "a frame started with vrt_render_begin is still in flight");
    if (depth < 0) return fail(c, VRT_ERR_INVALID, "negative depth");
    HIPCHK(c, hipSetDevice(c->device));
    DevParams P;
    int rc = prepare(c, p, &P);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, VRT_CTR_SLOTS * VRT_CTR_STRIDE * sizeof(unsigned long long), c->stream));
    bool two = (p->flags & VRT_FIX_DIELECTRIC) != 0;
    c->scene.dbg = c->d_ctr; c->last_ctr = c->d_ctr;
    HIPCHK(c, vrt_launch_ray(c->scene, P, c->use_bvh, two, origin, direction, depth, c->d_px_rgba, c->d_px_hit,
                             c->d_ctr, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float4 col; int id;
    HIPCHK(c, hipMemcpy(&col, c->d_px_rgba, sizeof col, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&id, c->d_px_hit, sizeof id, hipMemcpyDeviceToHost));
    rgba[0] = col.x; rgba[1] = col.y; rgba[2] = col.z; rgba[3] = col.w;
    if (hit_id_out) *hit_id_out = id;
    c->last_pixels = 0; c->last_mode = VRT_MODE_MEGAKERNEL; c->timed = false;
    return VRT_OK;
}

extern "C" uint32_t vrt_pixel_set(uint32_t seed, uint32_t pixel, uint32_t n_sets)
{
    if (n_sets <= 1u) return 0u;
    uint32_t x = pixel ^ seed;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x % n_sets;
}
