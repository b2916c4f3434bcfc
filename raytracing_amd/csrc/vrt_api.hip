// C ABI of libvrt.so (include/vrt.h): context, scene/camera upload, render calls.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <utility>
#include <thread>
#include <vector>

#include "vrt_ctx.h"

// Scenes with at most this many primitives are traced by the brute-force loop whose
// primitive data stays in scalar registers; larger ones get the LBVH.
static const uint32_t kBruteMaxPrims = 48;

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

// What each GPU's copy of the array holds as its sampler table, kept on the host for comparison: a
// call whose table is already there uploads nothing, and then consists of one fill and the kernel
// launch only (which is also all a stream under graph capture can take).  Per GPU, not per context:
// every context on a GPU shares that GPU's array.  The vectors get their full capacity once, in
// vrt_create, so that the entry point itself never allocates host memory either.
#define VRT_STATIC_MAX_DEVICES 64
#include <mutex>
struct StaticTable { std::mutex m; std::vector<float> host; size_t n = 0; };
static StaticTable g_static_table[VRT_STATIC_MAX_DEVICES];

// Clears the counter block of vrt_render_device_noalloc.  A kernel, not hipMemsetAsync: captured into
// a graph, the fill node of ROCm 7.2 cleared the block on the first replay and wrote other values on
// the following ones (the replayed frames were right, their ray counts were not); kernel nodes replay as recorded.
__global__ __launch_bounds__(256) void k_clear_words(unsigned long long* p, unsigned n)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) p[i] = 0ull;
}

int vrt_fail(vrt_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    return code;
}
static inline int fail(vrt_ctx* c, int code, const std::string& msg) { return vrt_fail(c, code, msg); }

extern "C" int vrt_abi_version(void) { return VRT_ABI_VERSION; }

extern "C" const char* vrt_last_error(vrt_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" void vrt_shading_defaults(vrt_shading* s)
{
    if (!s) return;
    memset(s, 0, sizeof *s);
    s->solid_ka = 0.5f; s->solid_kd = 0.5f; s->solid_ks = 0.f; s->solid_exp = 1.f;
    s->metal_ka = 0.25f; s->metal_kd = 0.5f; s->metal_ks = 0.15f; s->metal_exp = 100.f; s->metal_kr = 0.75f;
    s->metal_cr[0] = s->metal_cr[1] = s->metal_cr[2] = s->metal_cr[3] = 1.f;
    s->glass_ka = 0.5f; s->glass_kd = 0.5f; s->glass_ks = 0.2f; s->glass_exp = 1.f;
    s->glass_eta_in = 1.5f; s->glass_eta_out = 1.f;
    s->ao_min_amount[0] = s->ao_min_amount[1] = s->ao_min_amount[2] = 0.05f; s->ao_min_amount[3] = 1.f;
    s->ao_ls = 1.f;
    s->ao_color[0] = s->ao_color[1] = s->ao_color[2] = s->ao_color[3] = 1.f;
    // Material/MatteMaterial.cs:14-15: two default LambertianBRDFs (BRDF/LambertianBRDF.cs:19)
    s->matte_ka = 0.5f; s->matte_kd = 0.5f;
    // Material/TransparentMaterial.cs:10-25: Phong base with default Lambert coefficients; the
    // other five are constructor arguments the reference never supplies
    s->transp_ka = 0.5f; s->transp_kd = 0.5f; s->transp_ks = 0.5f; s->transp_exp = 2000.f;
    s->transp_ior = 1.5f; s->transp_kr = 0.1f; s->transp_kt = 0.9f;
    s->transp_ray_dir[0] = s->transp_ray_dir[1] = s->transp_ray_dir[2] = 0.f;
}

extern "C" void vrt_params_defaults(vrt_params* p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->spp = 1; p->n_sets = 1; p->max_depth = 10; p->range = 30.f;
    p->mode = VRT_MODE_AUTO; p->tile_count = 1; p->tile_rows = 8;
}

static void options_from_env(VrtOptions& o)
{
    auto geti = [](const char* n, long long dflt) { const char* e = getenv(n); return e && *e ? atoll(e) : dflt; };
    o.xcd_remap = (int)geti("VRT_XCD_REMAP", 0);
    o.sample_shift = (int)geti("VRT_SAMPLE_SHIFT", -1);
    o.wf_defer = (int)geti("VRT_WF_DEFER", 1);
    o.wf_refill = (int)geti("VRT_WF_REFILL", 1);
    o.wf_max_slots = geti("VRT_WF_MAX_SLOTS", 0);
    o.wf_any_cap = geti("VRT_WF_ANY_CAP", -1);
    const char* b = getenv("VRT_BUILDER");
    o.builder = !b || !*b ? -1 : (strcmp(b, "lbvh") == 0 ? 1 : (strcmp(b, "sah") == 0 ? 2 : (strcmp(b, "ploc") == 0 ? 0 : -1)));
    o.collapse_parity = (int)geti("VRT_COLLAPSE_PARITY", 0);
    o.no_big = (int)geti("VRT_NO_BIG", 0);
    o.packet = (int)geti("VRT_PACKET", -1);
    o.block_compact = (int)geti("VRT_BLOCK_COMPACT", 0);
    o.zero_copy = (int)geti("VRT_ZERO_COPY", 1);
    o.tile_sort = (int)geti("VRT_TILE_SORT", 1);
    o.staged = (int)geti("VRT_STAGED", 1);
    o.quad_max = (int)geti("VRT_QUAD_MAX", -1);
    o.beam = (int)geti("VRT_BEAM", -1);
    o.slab_fma = (int)geti("VRT_SLAB_FMA", 1);
    o.multi_gather = (int)geti("VRT_MULTI_GATHER", 0);
}

extern "C" int vrt_set_option(vrt_ctx* c, const char* name, long long value)
{
    if (!c || !name) return fail(c, VRT_ERR_INVALID, "null argument");
    if (c->pending) return fail(c, VRT_ERR_STATE, "a frame started with vrt_render_begin is still in flight");
    VrtOptions& o = c->opt;
    const std::string n(name);
    if (n == "xcd_remap") o.xcd_remap = value != 0;
    else if (n == "sample_shift") { if (value < -1 || value > 6) return fail(c, VRT_ERR_INVALID, "sample_shift: -1..6"); o.sample_shift = (int)value; }
    else if (n == "wf_defer") o.wf_defer = value != 0;
    else if (n == "wf_refill") o.wf_refill = value != 0;
    else if (n == "wf_max_slots") o.wf_max_slots = value > 0 ? value : 0;
    else if (n == "wf_any_cap") o.wf_any_cap = value;
    else if (n == "wf_queue_cap") o.wf_queue_cap = value > 0 ? value : 0;
    else if (n == "builder") { if (value < -1 || value > 3) return fail(c, VRT_ERR_INVALID, "builder: -1 by scene size, 0 PLOC, 1 radix tree, 2 SAH on the host, 3 SAH on the device"); o.builder = (int)value; }
    else if (n == "collapse_parity") o.collapse_parity = value != 0;
    else if (n == "no_big") o.no_big = value != 0;
    else if (n == "block_compact") o.block_compact = value != 0;
    else if (n == "zero_copy") o.zero_copy = value != 0;
    else if (n == "tile_sort") o.tile_sort = value != 0;
    else if (n == "staged") o.staged = value != 0;
    else if (n == "slab_fma") o.slab_fma = value != 0;
    else if (n == "stack_cap") { if (value < 0 || value > VRT_TRAV_STACK) return fail(c, VRT_ERR_INVALID, "stack_cap: 0..VRT_TRAV_STACK"); o.stack_cap = (int)value; }
    else if (n == "quad_max") { if (value < -1 || value > 64) return fail(c, VRT_ERR_INVALID, "quad_max: -1..64"); o.quad_max = (int)value; }
    else if (n == "fail_after_build") o.fail_after_build = value != 0;
    else if (n == "multi_gather") o.multi_gather = value != 0;
    else if (n == "beam") { if (value < -1 || value > 2) return fail(c, VRT_ERR_INVALID, "beam: -1, 0, 1 or 2"); o.beam = (int)value; }
    else if (n == "packet") { if (value < -1 || value > 2) return fail(c, VRT_ERR_INVALID, "packet: -1, 0, 1 or 2"); o.packet = (int)value; }
    else return fail(c, VRT_ERR_INVALID, "unknown option " + n);
    if (c->multi) return vrt_multi_set_option(c, name, value);
    return VRT_OK;
}

void SceneBufs::release()
{
    hipFree(d_sph); hipFree(d_box); hipFree(d_tri); hipFree(d_mat); hipFree(d_bvh); hipFree(d_bvh4);
    hipFree(d_prim_mat); hipFree(d_lights); hipFree(d_aab); hipFree(d_cam_pre); hipFree(d_tex_data); hipFree(d_mat_tex);
    hipFree(d_tex_info); hipFree(d_tri_uv);
    d_sph = d_box = d_tri = d_mat = d_bvh = d_bvh4 = d_aab = d_cam_pre = d_tex_data = nullptr;
    d_prim_mat = nullptr; d_lights = nullptr; d_mat_tex = nullptr; d_tex_info = nullptr; d_tri_uv = nullptr;
    h_sph.clear(); h_box.clear();
}

static void free_wf(vrt_ctx* c)
{
    hipFree(c->wf.arena); hipFree(c->wf.lacc); hipFree(c->wf.counts);
    hipFree(c->wf.hits); hipFree(c->wf.any.ox); hipFree(c->wf.any_counts);
    memset(&c->wf, 0, sizeof c->wf);
}

// The runtime page-locks whole pages: the range registered is the caller's buffer widened to page
// boundaries (what else lies in its first and last page is pinned with it, which is harmless).
static void page_range(void* p, size_t bytes, void** base, size_t* len)
{
    const uintptr_t page = 4096, a = (uintptr_t)p & ~(page - 1), b = ((uintptr_t)p + bytes + page - 1) & ~(page - 1);
    *base = (void*)a; *len = (size_t)(b - a);
}

// Releases everything a context owns; every handle may be null (a half-built context).
static void teardown(vrt_ctx* c)
{
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    c->sb.release();
    free_wf(c);
    for (OffsetSlot& s : c->off) { hipFree(s.d); s.d = nullptr; }
    hipFree(c->d_ctr); hipFree(c->d_ctr_ring); hipFree(c->d_rgba); hipFree(c->d_hit);
    hipFree(c->d_rec_pixels); hipFree(c->d_segs); hipFree(c->d_seg_counts);
    hipFree(c->d_px_rgba); hipFree(c->d_px_hit);
    hipFree(c->d_tile_cost); hipFree(c->d_tile_keys); hipFree(c->d_tile_iota); hipFree(c->d_tile_order); hipFree(c->d_tile_tmp);
    if (c->static_stage) hipHostFree(c->static_stage);
    for (auto& r : c->registered) { void* base; size_t len; page_range(r.first, r.second, &base, &len); hipHostUnregister(base); }
    c->registered.clear();
    delete c->copy_helper; c->copy_helper = nullptr;
    for (void* p : c->host_allocs) hipHostFree(p);
    c->host_allocs.clear();
    c->host_alloc_bytes.clear();
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->ev_done) hipEventDestroy(c->ev_done);
    if (c->ev_static) hipEventDestroy(c->ev_static);
    for (hipEvent_t e : c->part_ev) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    (void)hipGetLastError();
    delete c;
}

extern "C" int vrt_create(int device, vrt_ctx** out)
{
    if (!out) return VRT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VRT_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return VRT_ERR_INVALID;
    if (device < VRT_STATIC_MAX_DEVICES) {
        std::lock_guard<std::mutex> g(g_static_table[device].m);
        g_static_table[device].host.reserve(2 * (size_t)VRT_NOALLOC_MAX_OFFSETS);
    }
    vrt_ctx* c = new vrt_ctx();
    c->device = device;
    memset(&c->sb.scene, 0, sizeof c->sb.scene);
    memset(&c->cam, 0, sizeof c->cam);
    memset(&c->wf, 0, sizeof c->wf);
    options_from_env(c->opt);
    void* scratch = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_static, hipEventDisableTiming) != hipSuccess ||
        hipMalloc(&c->d_ctr, kCtrBlock * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&c->d_ctr_ring, kCtrRing * kCtrBlock * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&c->d_px_rgba, sizeof(float4)) != hipSuccess ||
        hipMalloc(&c->d_px_hit, sizeof(int)) != hipSuccess ||
        hipHostMalloc((void**)&c->static_stage, (size_t)VRT_NOALLOC_MAX_OFFSETS * sizeof(float2), hipHostMallocDefault) != hipSuccess ||
        // this device's copy of the static array, asked for with this device current
        hipGetSymbolAddress(&scratch, HIP_SYMBOL(g_vrt_static_scratch)) != hipSuccess || !scratch) {
        teardown(c);   // null handles are skipped
        return VRT_ERR_HIP;
    }
    c->static_scratch = (unsigned char*)scratch;
    *out = c;
    return VRT_OK;
}

extern "C" int vrt_create_multi(int n_gpus, const int* devices, vrt_ctx** out)
{
    if (!out) return VRT_ERR_INVALID;
    *out = nullptr;
    if (n_gpus < 1 || n_gpus > VRT_MAX_GPUS) return VRT_ERR_INVALID;
    vrt_ctx* c = nullptr;
    int rc = vrt_create(devices ? devices[0] : 0, &c);
    if (rc != VRT_OK) return rc;
    rc = vrt_multi_init(c, n_gpus, devices);
    if (rc != VRT_OK) { vrt_destroy(c); return rc; }
    *out = c;
    return VRT_OK;
}

extern "C" int vrt_destroy(vrt_ctx* c)
{
    if (!c) return VRT_ERR_INVALID;
    if (c->multi) vrt_multi_destroy(c);
    teardown(c);
    return VRT_OK;
}

extern "C" int vrt_host_alloc(vrt_ctx* c, uint64_t bytes, void** out)
{
    if (!c || !out || !bytes) return fail(c, VRT_ERR_INVALID, "null argument");
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    void* p = nullptr;
    // a buffer of a multi-GPU context is stored to by the frame kernels of all its GPUs: asked for as
    // portable and mapped explicitly, although the runtime maps page-locked memory for every device
    HIPCHK(c, hipHostMalloc(&p, (size_t)bytes, c->multi ? (hipHostMallocPortable | hipHostMallocMapped) : hipHostMallocDefault));
    c->host_allocs.push_back(p);
    c->host_alloc_bytes.push_back((size_t)bytes);
    *out = p;
    return VRT_OK;
}

extern "C" int vrt_host_free(vrt_ctx* c, void* p)
{
    if (!c || !p) return fail(c, VRT_ERR_INVALID, "null argument");
    if (c->pending) return fail(c, VRT_ERR_STATE, "a frame started with vrt_render_begin is still in flight");
    HIPCHK(c, hipSetDevice(c->device));
    for (size_t i = 0; i < c->host_allocs.size(); ++i)
        if (c->host_allocs[i] == p) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipHostFree(p));
            c->host_allocs.erase(c->host_allocs.begin() + i);
            c->host_alloc_bytes.erase(c->host_alloc_bytes.begin() + i);
            return VRT_OK;
        }
    return fail(c, VRT_ERR_INVALID, "buffer was not allocated with vrt_host_alloc on this context");
}

extern "C" int vrt_host_register(vrt_ctx* c, void* p, uint64_t bytes)
{
    if (!c || !p || !bytes) return fail(c, VRT_ERR_INVALID, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    for (auto& r : c->registered) if (r.first == p) return fail(c, VRT_ERR_STATE, "buffer is already registered");
    void* base; size_t len;
    page_range(p, (size_t)bytes, &base, &len);
    HIPCHK(c, hipHostRegister(base, len, hipHostRegisterDefault));
    c->registered.push_back({p, (size_t)bytes});
    return VRT_OK;
}

extern "C" int vrt_host_unregister(vrt_ctx* c, void* p)
{
    if (!c || !p) return fail(c, VRT_ERR_INVALID, "null argument");
    if (c->pending) return fail(c, VRT_ERR_STATE, "a frame started with vrt_render_begin is still in flight");
    HIPCHK(c, hipSetDevice(c->device));
    for (size_t i = 0; i < c->registered.size(); ++i)
        if (c->registered[i].first == p) {
            void* base; size_t len;
            page_range(p, c->registered[i].second, &base, &len);
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipHostUnregister(base));
            c->registered.erase(c->registered.begin() + i);
            return VRT_OK;
        }
    return fail(c, VRT_ERR_INVALID, "buffer was not registered with this context");
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
static int refresh_cam_pre(vrt_ctx* c, SceneBufs& sb)
{
    if (!c->have_cam || sb.use_bvh) return VRT_OK;
    const DevScene& S = sb.scene;
    // layout: the spheres, padded to a multiple of four with entries no ray passes (the loop of
    // intersect_brute takes a group of spheres per turn), then the boxes
    const size_t n_sph4 = ((size_t)S.n_sph + 3) & ~(size_t)3;
    std::vector<float4> pre(n_sph4 + S.n_box, make_float4(0.f, 0.f, 0.f, -INFINITY));
    const float ox = c->cam.ox, oy = c->cam.oy, oz = c->cam.oz;
    for (int i = 0; i < S.n_sph; ++i) {
        float4 s = sb.h_sph[i];
        float x = ox - s.x, y = oy - s.y, z = oz - s.z;
        float cc = (x * x + y * y + z * z) - s.w * s.w;
        pre[i] = make_float4(x, y, z, cc);
    }
    for (int i = 0; i < S.n_box; ++i) {
        float4 b0 = sb.h_box[4 * (size_t)i];
        pre[n_sph4 + i] = make_float4(ox - b0.x, oy - b0.y, oz - b0.z, 0.f);
    }
    if (pre.empty()) return VRT_OK;
    if (!sb.d_cam_pre) HIPCHK(c, hipMalloc(&sb.d_cam_pre, pre.size() * sizeof(float4)));
    HIPCHK(c, hipMemcpy(sb.d_cam_pre, pre.data(), pre.size() * sizeof(float4), hipMemcpyHostToDevice));
    sb.scene.cam_pre = sb.d_cam_pre;
    return VRT_OK;
}

// Uploads the scene into `sb` (empty on entry) and builds its tree.  On failure the caller
// releases `sb`; the context's current scene is not touched here.
static int build_scene(vrt_ctx* c, const vrt_scene_desc* d, SceneBufs& sb)
{
    const uint32_t n_prims = d->n_spheres + d->n_boxes + d->n_triangles;
    std::vector<float4> sph(d->n_spheres), box(4 * (size_t)d->n_boxes), tri(3 * (size_t)d->n_triangles),
        mat(d->n_materials);
    std::vector<int> pm(n_prims);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const vrt_material& m = d->materials[i];
        if (m.type < VRT_MAT_SOLID || m.type > VRT_MAT_TRANSPARENT) return fail(c, VRT_ERR_INVALID, "bad material type");
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
    for (uint32_t i = 0; i < d->n_lights; ++i)
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

    {   // padded to a multiple of four with spheres of infinite radius: |o - c|^2 - r^2 is -inf (or
        // NaN), never > 0, so no ray passes them (intersect_brute tests a group of spheres per turn)
        std::vector<float4> sph4 = sph;
        sph4.resize((sph.size() + 3) & ~(size_t)3, make_float4(0.f, 0.f, 0.f, INFINITY));
        HIPCHK(c, upload(&sb.d_sph, sph4));
    }
    HIPCHK(c, upload(&sb.d_box, box));
    HIPCHK(c, upload(&sb.d_tri, tri));
    HIPCHK(c, upload(&sb.d_mat, mat));
    HIPCHK(c, upload(&sb.d_prim_mat, pm));

    DevScene& S = sb.scene;
    memset(&S, 0, sizeof S);
    S.n_sph = (int)d->n_spheres; S.n_box = (int)d->n_boxes; S.n_tri = (int)d->n_triangles;
    S.n_prims = (int)n_prims;
    S.sph = sb.d_sph; S.box = sb.d_box; S.tri = sb.d_tri; S.prim_mat = sb.d_prim_mat; S.mat = sb.d_mat;
    if (!mat_tex.empty()) {
        HIPCHK(c, upload(&sb.d_mat_tex, mat_tex));
        HIPCHK(c, upload(&sb.d_tex_info, tex_info));
        HIPCHK(c, upload(&sb.d_tex_data, tex_data));
        HIPCHK(c, upload(&sb.d_tri_uv, tri_uv));
        S.mat_tex = sb.d_mat_tex; S.tex_info = sb.d_tex_info; S.tex_data = sb.d_tex_data; S.tri_uv = sb.d_tri_uv;
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
    HIPCHK(c, upload(&sb.d_lights, lights));
    S.lights = sb.d_lights;
    S.sh = d->shading;
    for (int k = 0; k < 3; ++k) {   // see DevScene: same float operations, in the order of the shading code
        const vrt_shading& sh = d->shading;
        S.ao_free[k] = sh.ao_color[k] * sh.ao_ls;
        const float mn_ls = sh.ao_min_amount[k] * sh.ao_ls;
        S.ao_occl[k] = mn_ls * sh.ao_color[k];
        S.metal_w[k] = sh.metal_cr[k] * sh.metal_kr;
    }
    S.dbg = c->d_ctr;
    S.stack_cap = VRT_TRAV_STACK;
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        if (d->materials[i].type == VRT_MAT_MATTE) S.has_matte = 1;
        if (d->materials[i].type == VRT_MAT_TRANSPARENT) S.has_transparent = 1;
    }

    sb.use_bvh = n_prims > kBruteMaxPrims;
    sb.bvh_ms = 0.f; sb.bvh_depth = 0;
    if (sb.use_bvh) {
        // Up to VRT_MAX_BIG primitives whose box is at least half as long as the scene's stay
        // out of the tree (spheres and boxes only; a mesh is expected to be tessellated).
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
        sb.max_abs = 0;
        for (int k = 0; k < 3; ++k) sb.max_abs = std::max(sb.max_abs, std::max(fabs(lo[k]), fabs(hi[k])));
        std::sort(ext.begin(), ext.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) {
            return a.first > b.first || (a.first == b.first && a.second < b.second);
        });
        if (!c->opt.no_big)
            for (size_t i = 0; i < ext.size() && S.n_big < VRT_MAX_BIG && ext[i].first >= 0.5 * scene_ext; ++i)
                S.big[S.n_big++] = ext[i].second;
        HIPCHK(c, hipMalloc(&sb.d_bvh, 4 * (size_t)(n_prims - 1) * sizeof(float4)));
        HIPCHK(c, hipMalloc(&sb.d_bvh4, 8 * (size_t)(n_prims - 1) * sizeof(float4)));
        S.bvh = sb.d_bvh; S.bvh4 = sb.d_bvh4;
        // Binned SAH (on the device, one wave per node and level: 7 ms for 10 k spheres, 0.12 s per
        // million triangles) buys 10 % of the frame time over PLOC (10 ms per million); beyond a few
        // million primitives the set-up time would be felt, so PLOC takes over.
        const int builder = c->opt.builder >= 0 ? c->opt.builder : (n_prims <= 4000000 ? 3 : 0);
        hipError_t e = vrt_build_lbvh(S, sb.d_bvh, sb.d_bvh4, &sb.bvh_depth, &sb.bvh_ms,
                                      builder, c->opt.collapse_parity != 0, c->stream);
        if (e != hipSuccess) return fail(c, VRT_ERR_HIP, std::string("LBVH build: ") + hipGetErrorString(e));
        if (sb.bvh_depth > VRT_MAX_TREE_DEPTH)
            return fail(c, VRT_ERR_CAPACITY, "LBVH deeper than VRT_MAX_TREE_DEPTH levels");
    } else {
        sb.h_sph = sph; sb.h_box = box;
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
            HIPCHK(c, upload(&sb.d_aab, aab));
            S.aab = sb.d_aab; S.all_aab = 1;
        }
    }
    return refresh_cam_pre(c, sb);
}

// Whether this frame's box tests may take the fused form (slab4 in vrt_trace.h): every ray starts
// at the camera (or at `origin`) or on a surface of the scene, and the fused form is proven to keep
// every candidate only while those points lie within +-1600 of the coordinate origin.
static bool fused_form(const vrt_ctx* c, const float* origin)
{
    if (!c->sb.use_bvh || c->opt.slab_fma == 0) return false;
    double m = c->sb.max_abs;
    const float cam[3] = {c->cam.ox, c->cam.oy, c->cam.oz};
    const float* o = origin ? origin : cam;
    for (int k = 0; k < 3; ++k) m = std::max(m, (double)fabsf(o[k]));
    return m <= 1600.0;
}

#define VRT_LAUNCH(fused, fn, ...) ((fused) ? vrt_fused::fn(__VA_ARGS__) : vrt_plain::fn(__VA_ARGS__))

// Assembles the scene beside the current one on this context's GPU (vrt_swap_scene_one makes it
// current).  On failure `nb` is released and the context still renders the scene it had.
int vrt_build_scene_one(vrt_ctx* c, const vrt_scene_desc* d, SceneBufs& nb)
{
    HIPCHK(c, hipSetDevice(c->device));
    int rc = build_scene(c, d, nb);
    if (rc == VRT_OK && c->opt.fail_after_build) rc = fail(c, VRT_ERR_HIP, "injected failure after the tree build (test option)");
    if (rc != VRT_OK) nb.release();
    return rc;
}

int vrt_swap_scene_one(vrt_ctx* c, SceneBufs& nb)
{
    hipError_t se = hipSetDevice(c->device);
    if (se == hipSuccess) se = hipDeviceSynchronize();   // frames in flight (any stream) still read the old buffers
    if (se != hipSuccess) { nb.release(); return fail(c, VRT_ERR_HIP, std::string("hipDeviceSynchronize: ") + hipGetErrorString(se)); }
    c->sb.release();
    c->sb = std::move(nb);
    nb = SceneBufs();
    c->have_scene = true;
    c->scene_gen++;
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
    if (c->multi) return vrt_multi_set_scene(c, d);
    // The new scene is assembled beside the old one and swapped in when every upload and the tree
    // build have succeeded: a failed call leaves the context rendering the scene it had.
    SceneBufs nb;
    int rc = vrt_build_scene_one(c, d, nb);
    if (rc != VRT_OK) return rc;
    return vrt_swap_scene_one(c, nb);
}

int vrt_set_camera_one(vrt_ctx* c, const vrt_camera* cam)
{
    DevCam& d = c->cam;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize()); // frames in flight still read the old terms
    d.ox = cam->origin[0]; d.oy = cam->origin[1]; d.oz = cam->origin[2];
    // PlaneBorderPoints[0] - rayOrigin, rounded once like the C# Vector3 subtraction
    d.bx = cam->p0[0] - cam->origin[0]; d.by = cam->p0[1] - cam->origin[1]; d.bz = cam->p0[2] - cam->origin[2];
    d.xx = cam->x_axis[0]; d.xy = cam->x_axis[1]; d.xz = cam->x_axis[2];
    d.yx = cam->y_axis[0]; d.yy = cam->y_axis[1]; d.yz = cam->y_axis[2];
    d.width = cam->width; d.height = cam->height;
    c->have_cam = true;
    return c->have_scene ? refresh_cam_pre(c, c->sb) : VRT_OK;
}

extern "C" int vrt_set_camera(vrt_ctx* c, const vrt_camera* cam)
{
    if (!c || !cam) return fail(c, VRT_ERR_INVALID, "null argument");
    if (c->pending) return fail(c, VRT_ERR_STATE, "a frame started with vrt_render_begin is still in flight");
    if (cam->width <= 0 || cam->height <= 0 || (int64_t)cam->width * cam->height > (1ll << 30))
        return fail(c, VRT_ERR_INVALID, "frame size out of range");
    if (c->multi) return vrt_multi_set_camera(c, cam);
    return vrt_set_camera_one(c, cam);
}

// Bytes per pixel of the frame buffer a call writes: the Color of PixelColorData, or its Color32.
static inline size_t frame_px_bytes(unsigned flags) { return (flags & VRT_OUT_RGBA8) ? 4 : sizeof(float4); }

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

// The device copy of the caller's sampler table.  Tables are cached by content in a few slots:
// an unchanged table costs one comparison and no copy; a new one goes to the slot used longest
// ago, after everything in flight on the device has finished, so no frame ever samples with
// another frame's table.
static int cached_offsets(vrt_ctx* c, const float* table, size_t n, const float2** out)
{
    for (OffsetSlot& s : c->off)
        if (s.d && s.n == n && memcmp(s.host.data(), table, n * sizeof(float2)) == 0) {
            s.last_use = ++c->off_clock;
            *out = s.d;
            return VRT_OK;
        }
    OffsetSlot* victim = &c->off[0];
    for (OffsetSlot& s : c->off) if (s.last_use < victim->last_use) victim = &s;
    if (victim->d) HIPCHK(c, hipDeviceSynchronize());
    if (n > victim->cap) {
        hipFree(victim->d); victim->d = nullptr; victim->cap = 0; victim->n = 0;
        HIPCHK(c, hipMalloc(&victim->d, n * sizeof(float2)));
        victim->cap = n;
    }
    victim->n = 0;
    HIPCHK(c, hipMemcpy(victim->d, table, n * sizeof(float2), hipMemcpyHostToDevice));
    victim->host.assign(table, table + 2 * n);
    victim->n = n;
    victim->last_use = ++c->off_clock;
    *out = victim->d;
    return VRT_OK;
}

// The general kernel instantiations (two children per hit, all material classes): refraction with
// VRT_FIX_DIELECTRIC, or a matte or transparent material in the scene.  Everything else runs the
// instantiations that know one child per hit and the three shipped material classes.
static bool two_children(const vrt_ctx* c, const vrt_params* p)
{
    return (p->flags & VRT_FIX_DIELECTRIC) != 0 || c->sb.scene.has_transparent != 0 || c->sb.scene.has_matte != 0;
}

// Validates params and fills DevParams (everything but the offset table pointer).
static int check_params(vrt_ctx* c, const vrt_params* p, DevParams* out)
{
    if (!c->have_scene || !c->have_cam) return fail(c, VRT_ERR_STATE, "scene or camera not set");
    if (p->spp < 1 || p->spp > 65536) return fail(c, VRT_ERR_INVALID, "spp out of range");
    if (p->max_depth < 0 || p->max_depth > VRT_MAX_DEPTH) return fail(c, VRT_ERR_CAPACITY, "max_depth out of range");
    if (!(p->range > 0.f)) return fail(c, VRT_ERR_INVALID, "range must be positive");
    if (p->offsets && p->n_sets < 1) return fail(c, VRT_ERR_INVALID, "n_sets must be >= 1");
    if ((p->flags & VRT_OUT_LINEAR_RGB) && (p->flags & VRT_OUT_RGBA8)) return fail(c, VRT_ERR_INVALID, "VRT_OUT_LINEAR_RGB and VRT_OUT_RGBA8 exclude each other");
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
    // keep the hardware's round-robin deal unless the xcd_remap option asks for the banding.
    P.xcd_remap = c->opt.xcd_remap;
    // Megakernel lane split: the largest power of two <= min(spp, 64) sample lanes per pixel that
    // leaves at most 15 % of the lane-rounds idle, unless the sample_shift option fixes it.
    P.sample_shift = 0;
    for (int sh = 6; sh >= 1; --sh) {
        int n = 1 << sh;
        if (n > p->spp) continue;
        int rounds = (p->spp + n - 1) / n;
        if ((double)rounds * n <= 1.15 * p->spp) { P.sample_shift = sh; break; }
    }
    if (c->opt.sample_shift >= 0) P.sample_shift = c->opt.sample_shift;
    // Wave-uniform traversal of the tree for the camera rays of a wave and the occlusion rays of
    // their hits: pays when the lanes of a wave belong to few pixels (measured +2.6 % at 16 lanes per
    // pixel on the 10k-sphere frame, +6 % at 4 lanes per pixel on the 1M-triangle frame).
    P.packet = c->opt.packet >= 0 ? c->opt.packet : (P.sample_shift >= 2 ? 1 : 0);
    P.quad_max = c->opt.quad_max >= 0 ? c->opt.quad_max : 16;
    P.beam = c->opt.beam >= 0 ? c->opt.beam : 1;
    P.general = two_children(c, p) ? 1 : 0;
    P.offsets = nullptr;
    return VRT_OK;
}

static int prepare(vrt_ctx* c, const vrt_params* p, DevParams* out)
{
    int rc = check_params(c, p, out);
    if (rc) return rc;
    if (p->offsets) {
        rc = cached_offsets(c, p->offsets, (size_t)p->n_sets * p->spp, &out->offsets);
        if (rc) return rc;
    }
    return VRT_OK;
}

// Sample slots the wavefront workspace holds at most: a band of rows of the frame is in flight
// at a time.  64 Mi slots are 1 GiB of radiance slots, 7.5 GiB of ray queues and 3.75 GiB for the
// deferred occlusion queries (queues twice that with two children per hit); a frame with fewer samples gets a workspace of its own size.
static const size_t kWfMaxSlots = 64ull << 20;

// Queue entries beyond the rays a band can hold: every wave of the persistent kernels may leave
// the last chunk it took partly unwritten, and the kernels run (live + 255) / 256 blocks of four
// waves, VRT_WF_MAX_BLOCKS at most.
static size_t wf_slack(size_t live)
{
    size_t blocks = (live + 255) / 256;
    if (blocks > VRT_WF_MAX_BLOCKS) blocks = VRT_WF_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    return blocks * 4 * VRT_WF_CHUNK;
}

static int ensure_wf(vrt_ctx* c, size_t slots, bool two, hipStream_t stream)
{
    const bool defer = c->opt.wf_defer != 0;
    const bool short_any = c->opt.wf_any_cap >= 0 || c->opt.wf_queue_cap > 0;
    if (c->wf.slots >= slots && (c->wf_two || !two) && (c->wf.hits != nullptr) == defer && !c->wf_short && !short_any) return VRT_OK;
    HIPCHK(c, hipDeviceSynchronize());   // a frame in flight may still use the old workspace
    free_wf(c);
    c->wf_short = false;
    const size_t live = slots * (two ? 2 : 1);
    size_t cap = live + wf_slack(live);
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
    if (c->opt.wf_queue_cap > 0 && (size_t)c->opt.wf_queue_cap < cap) {   // tests: ray queues that overflow
        c->wf.cap = (size_t)c->opt.wf_queue_cap; c->wf_short = true;
    }
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
        if (c->opt.wf_any_cap >= 0 && (size_t)c->opt.wf_any_cap < cap) {   // tests: a short queue forces the in-place path
            c->wf.any_cap = (size_t)c->opt.wf_any_cap; c->wf_short = true;
        }
        HIPCHK(c, hipMalloc(&c->wf.any_counts, 3 * (VRT_MAX_DEPTH + 2) * sizeof(int)));
        c->wf.cursors = c->wf.any_counts + (VRT_MAX_DEPTH + 2);
    }
    c->wf.slots = slots; c->wf_two = two;
    return VRT_OK;
}

// Enqueues one frame on `stream`.  `ctr` = the counter block to count into (already clear), or
// null for the path's own block, cleared here.
// Tree scenes (and the simple brute-force scenes): the megakernel's tiles are dispatched most
// expensive first.  A launch keeps 79-93 %
// of the chip's wave slots filled on the tree workloads; the rest is the tail, in which the last
// tiles dispatched run alone, and its length is the duration of whatever those tiles are.  Costs
// (clock ticks per tile) come from a frame of the same layout on the same scene: that frame runs
// in tile order and reports them, a radix sort behind it on the same stream turns them into the
// order (descending, ties by tile number), and the frames after it use the order; every 16th
// refreshes it (every 256th on the brute-force scenes, whose frames are short).  The camera is not
// part of the layout on purpose: a host that moves it a little every frame still dispatches by the
// costs of a frame ago.  Nothing about a frame's result depends on the order.  Returns with
// P.tile_order / P.tile_cost set (or left null) and *sort_after telling the caller to enqueue the
// sort behind the frame.
static int plan_tile_order(vrt_ctx* c, DevParams& P, hipStream_t stream, bool* sort_after, size_t* n_tiles)
{
    *sort_after = false;
    const int pshift = 6 - P.sample_shift, pw_shift = (pshift + 1) >> 1;
    const size_t bx = ((size_t)c->cam.width + ((size_t)1 << pw_shift) - 1) >> pw_shift;
    const size_t by = ((size_t)P.local_rows + ((size_t)1 << (pshift - pw_shift)) - 1) >> (pshift - pw_shift);
    const size_t n = bx * by;
    *n_tiles = n;
    if (n < 8192 || n > 0x7fffffffull) return VRT_OK;   // two rounds of waves: nothing to order
    const long long sig[10] = {(long long)c->scene_gen, (long long)n, (long long)bx, P.spp, P.max_depth, P.sample_shift,
                               ((long long)P.tile_index << 32) | (unsigned)P.tile_count, ((long long)P.tile_rows << 1) | (P.compact ? 1 : 0),
                               (long long)P.flags, ((long long)c->cam.width << 32) | (unsigned)c->cam.height};
    if (c->tile_stream && c->tile_stream != stream) {
        // frames of another stream may still read the order this one is about to replace
        HIPCHK(c, hipDeviceSynchronize());
        c->tile_valid = false;
    }
    if (n > c->tile_cap) {
        HIPCHK(c, hipDeviceSynchronize());
        hipFree(c->d_tile_cost); hipFree(c->d_tile_keys); hipFree(c->d_tile_iota); hipFree(c->d_tile_order); hipFree(c->d_tile_tmp);
        c->d_tile_cost = c->d_tile_keys = nullptr; c->d_tile_iota = c->d_tile_order = nullptr; c->d_tile_tmp = nullptr;
        c->tile_cap = 0; c->tile_valid = false;
        c->tile_tmp_bytes = vrt_tile_sort_bytes(n);
        if (!c->tile_tmp_bytes) return VRT_OK;   // no sort, no order
        HIPCHK(c, hipMalloc(&c->d_tile_cost, n * sizeof(unsigned)));
        HIPCHK(c, hipMalloc(&c->d_tile_keys, n * sizeof(unsigned)));
        HIPCHK(c, hipMalloc(&c->d_tile_iota, n * sizeof(int)));
        HIPCHK(c, hipMalloc(&c->d_tile_order, n * sizeof(int)));
        HIPCHK(c, hipMalloc(&c->d_tile_tmp, c->tile_tmp_bytes));
        HIPCHK(c, vrt_fill_iota(c->d_tile_iota, n, stream));
        c->tile_cap = n;
    }
    const bool same = c->tile_valid && memcmp(sig, c->tile_sig, sizeof sig) == 0;
    if (same) P.tile_order = c->d_tile_order;
    // the sort behind a recording frame is a third of a brute-force 1080p frame: those refresh rarely
    const unsigned period_mask = c->sb.use_bvh ? 15u : 255u;
    if (!same || (++c->tile_age & period_mask) == 0) {
        P.tile_cost = c->d_tile_cost;
        *sort_after = true;
        // The order on the device stops being valid here and becomes valid again when the sort
        // behind this frame has been enqueued: should the launch or the sort fail, the next frame
        // must not look its tiles up in an order made for another layout.
        c->tile_valid = false;
        memcpy(c->tile_sig, sig, sizeof sig);
    }
    c->tile_stream = stream;
    return VRT_OK;
}

static int enqueue_frame(vrt_ctx* c, const vrt_params* p, const DevParams& P_in, float4* rgba, int* hit,
                         hipStream_t stream, unsigned long long* ctr = nullptr, bool may_order = true)
{
    DevParams P = P_in;
    const bool two = two_children(c, p);
    int mode = p->mode == VRT_MODE_WAVEFRONT ? VRT_MODE_WAVEFRONT : VRT_MODE_MEGAKERNEL;
    DevScene& S = c->sb.scene;
    if (mode == VRT_MODE_WAVEFRONT) {
        // whole frame if it fits the slot budget, else bands of rows of at least one tile row
        size_t per_row = (size_t)c->cam.width * P.spp;
        size_t want = per_row * (size_t)(P.local_rows > 0 ? P.local_rows : 1);
        size_t min_band = per_row * 8;
        size_t max_slots = c->opt.wf_max_slots > 0 ? (size_t)c->opt.wf_max_slots : kWfMaxSlots;
        if (want > max_slots) want = max_slots > min_band ? max_slots : min_band;
        if (want < min_band) want = min_band;
        const size_t live = want * (two ? 2 : 1);
        if (live + 2 * wf_slack(live) > 0x7fffffffull) return fail(c, VRT_ERR_CAPACITY, "frame row too long for the wavefront queues");
        int rc = ensure_wf(c, want, two, stream);   // its fills run on the frame's stream, ahead of the kernels
        if (rc) return rc;
        WfWorkspace wf = c->wf;
        wf.slots = want; // a workspace kept from a larger frame still runs this frame in bands of `want`
        wf.refill = c->opt.wf_refill != 0;
        if (!ctr) {
            HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, kCtrBlock * sizeof(unsigned long long), stream));
            ctr = c->d_ctr;
        }
        S.dbg = ctr; c->last_ctr = ctr;
        S.stack_cap = c->opt.stack_cap > 0 ? c->opt.stack_cap : VRT_TRAV_STACK;
        HIPCHK(c, VRT_LAUNCH(fused_form(c, nullptr), vrt_launch_wavefront, S, c->cam, P, c->sb.use_bvh, two, wf, rgba, hit, ctr, stream));
    } else {
        if (!ctr) {
            // Blocks of the ring are clean when their turn comes only if the frames of a round ran
            // in order: a frame on another stream than its predecessor's starts a new round once
            // the device has drained.
            if (c->ring_stream != stream && c->ring_next % kCtrRing != 0) {
                HIPCHK(c, hipDeviceSynchronize());
                c->ring_next = 0;
            }
            c->ring_stream = stream;
            const unsigned slot = c->ring_next % kCtrRing;
            if (slot == 0)
                HIPCHK(c, hipMemsetAsync(c->d_ctr_ring, 0, kCtrRing * kCtrBlock * sizeof(unsigned long long), stream));
            c->ring_next++;
            ctr = c->d_ctr_ring + (size_t)slot * kCtrBlock;
        }
        S.dbg = ctr; c->last_ctr = ctr;
        S.stack_cap = c->opt.stack_cap > 0 ? c->opt.stack_cap : VRT_TRAV_STACK;
        const bool fused = fused_form(c, nullptr);
        if (c->sb.use_bvh && !two && !P.xcd_remap && c->opt.block_compact) {
            HIPCHK(c, VRT_LAUNCH(fused, vrt_launch_mega_blk, S, c->cam, P, rgba, hit, ctr, stream));
        } else {
            bool sort_after = false;
            size_t n_tiles = 0;
            // tree scenes, and the brute-force scenes whose kernel has instantiations with the order
            if (may_order && c->opt.tile_sort && !P.xcd_remap && (c->sb.use_bvh || (!two && vrt_scene_is_simple(S)))) {
                int rc = plan_tile_order(c, P, stream, &sort_after, &n_tiles);
                if (rc) return rc;
            }
            HIPCHK(c, VRT_LAUNCH(fused, vrt_launch_mega, S, c->cam, P, c->sb.use_bvh, two, rgba, hit, ctr, stream));
            if (sort_after) {
                HIPCHK(c, vrt_tile_sort(c->d_tile_cost, c->d_tile_keys, c->d_tile_iota, c->d_tile_order, n_tiles, c->d_tile_tmp,
                                        c->tile_tmp_bytes, stream));
                c->tile_valid = true;
            }
        }
    }
    c->last_pixels = (uint64_t)P.local_rows * c->cam.width;
    c->last_mode = mode;
    return VRT_OK;
}

// A wavefront frame whose ray tree outgrew the queues (two children per hit) has dropped
// branches: the frame is wrong and the call that produced it must say so.
static int check_wf_overflow(vrt_ctx* c)
{
    if (c->last_mode != VRT_MODE_WAVEFRONT || !c->wf.counts) return VRT_OK;
    int overflow = 0;
    HIPCHK(c, hipMemcpy(&overflow, c->wf.counts + VRT_MAX_DEPTH + 2, sizeof(int), hipMemcpyDeviceToHost));
    if (overflow)
        return fail(c, VRT_ERR_CAPACITY, "wavefront ray queue overflowed (two children per hit); "
                                         "use VRT_MODE_MEGAKERNEL for this scene");
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
    s->bvh_build_ms = c->sb.bvh_ms;
    s->bvh_nodes = c->sb.use_bvh ? (uint32_t)(c->sb.scene.n_prims - 1) : 0;
    s->mode_used = c->last_mode;
    s->gpu_ms = 0.f;
    int rc = check_wf_overflow(c);
    if (rc) return rc;
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
    const bool two = two_children(c, p);
    HIPCHK(c, VRT_LAUNCH(fused_form(c, nullptr), vrt_launch_pixels, c->sb.scene, c->cam, P, c->sb.use_bvh, two, c->d_rec_pixels, n, sample_sel, c->d_segs,
                                cap, c->d_seg_counts, rgba_dev, hit_dev, ctr, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (counts_out) HIPCHK(c, hipMemcpy(counts_out, c->d_seg_counts, n * sizeof(int), hipMemcpyDeviceToHost));
    if (out && cap > 0)
        HIPCHK(c, hipMemcpy(out, c->d_segs, (size_t)n * cap * sizeof(vrt_path_seg), hipMemcpyDeviceToHost));
    return VRT_OK;
}

static int ensure_fb(vrt_ctx* c)
{
    size_t npix = (size_t)c->cam.width * c->cam.height;
    if (npix > c->fb_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(c->d_rgba); hipFree(c->d_hit);
        c->d_rgba = nullptr; c->d_hit = nullptr; c->fb_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_rgba, npix * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_hit, npix * sizeof(int)));
        c->fb_cap = npix;
    }
    return VRT_OK;
}

// Copies the rows a frame call rendered from the context's frame buffer to the caller's: the
// whole buffer for a whole frame or packed rows, else only the strips this call owns (one strided
// copy for the full strips, one more for a ragged last strip), so that a frame delivered strip
// set by strip set moves every row over PCIe once and leaves the other rows as the caller has them.
static int copy_rows_out(vrt_ctx* c, const DevParams& P, float* rgba_out, int32_t* hit_id_out)
{
    const size_t W = (size_t)c->cam.width;
    const int H = c->cam.height;
    const size_t pxb = frame_px_bytes(P.flags);
    if (P.tile_count <= 1 || P.compact) {
        const size_t nout = P.compact ? (size_t)P.local_rows * W : W * H;
        if (nout == 0) return VRT_OK;
        if (rgba_out)   // null: the frame kernel has written the caller's buffer itself
            HIPCHK(c, hipMemcpyAsync(rgba_out, c->d_rgba, nout * pxb, hipMemcpyDeviceToHost, c->stream));
        if (hit_id_out)
            HIPCHK(c, hipMemcpyAsync(hit_id_out, c->d_hit, nout * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        return VRT_OK;
    }
    const int R = P.tile_rows, first = P.tile_index * R, period = P.tile_count * R;
    int n_full = 0, tail_row = -1, tail_rows = 0;
    for (int a = first; a < H; a += period) {
        if (a + R <= H) ++n_full; else { tail_row = a; tail_rows = H - a; }
    }
    auto strided = [&](void* dst, const void* src, size_t px_bytes) -> hipError_t {
        if (n_full > 0) {
            hipError_t e = hipMemcpy2DAsync((char*)dst + (size_t)first * W * px_bytes, (size_t)period * W * px_bytes,
                                            (const char*)src + (size_t)first * W * px_bytes, (size_t)period * W * px_bytes,
                                            (size_t)R * W * px_bytes, (size_t)n_full, hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) return e;
        }
        if (tail_rows > 0)
            return hipMemcpyAsync((char*)dst + (size_t)tail_row * W * px_bytes, (const char*)src + (size_t)tail_row * W * px_bytes,
                                  (size_t)tail_rows * W * px_bytes, hipMemcpyDeviceToHost, c->stream);
        return hipSuccess;
    };
    if (rgba_out) HIPCHK(c, strided(rgba_out, c->d_rgba, pxb));
    if (hit_id_out) HIPCHK(c, strided(hit_id_out, c->d_hit, sizeof(int)));
    return VRT_OK;
}

// The device's address of `p` when [p, p + bytes) lies inside a buffer of vrt_host_alloc, else null.
// Such a buffer is page-locked and mapped into the device's address space, so a kernel can store
// to it directly.
// (The buffers of a multi-GPU context belong to the context the caller holds; page-locked host
// memory is mapped into every GPU's address space, and each GPU asks for its own view of it with
// itself as the current device.)
static void* device_view_of_host_alloc(vrt_ctx* c, void* p, size_t bytes)
{
    const vrt_ctx* owner = c->host_owner ? c->host_owner : c;
    for (size_t i = 0; i < owner->host_allocs.size(); ++i) {
        char* b = (char*)owner->host_allocs[i];
        if ((char*)p >= b && (char*)p + bytes <= b + owner->host_alloc_bytes[i]) {
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, owner->host_allocs[i], 0) != hipSuccess || !d) { (void)hipGetLastError(); return nullptr; }
            return (char*)d + ((char*)p - b);
        }
    }
    return nullptr;
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
    rc = ensure_fb(c);
    if (rc) return rc;
    // A frame buffer from vrt_host_alloc is written by the frame kernel itself: its 128-byte row
    // segments cross PCIe while the rest of the frame is still being traced, instead of in a copy
    // that starts when the kernel has ended (1080p: 0.78 -> 0.72 ms; the device time of the frame
    // in vrt_stats then includes the transfer).  Hit ids, 32 bytes per row segment, are slower that
    // way than in one copy and keep going through the device buffer.
    const size_t rows_out = P.compact ? (size_t)P.local_rows : (size_t)c->cam.height;
    // (Not the 4-byte Color32 format: a wave's tile is 8 pixels wide at one sample per pixel, 32-byte
    // row segments cross PCIe worse than one copy after the kernel, 0.357 against 0.322 ms at 1080p.)
    const bool store_direct = c->opt.zero_copy && !(P.flags & VRT_OUT_RGBA8);
    float4* direct = store_direct ? (float4*)device_view_of_host_alloc(c, rgba_out, rows_out * c->cam.width * frame_px_bytes(P.flags)) : nullptr;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    rc = enqueue_frame(c, p, P, direct ? direct : c->d_rgba, hit_id_out ? c->d_hit : nullptr, c->stream);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    return copy_rows_out(c, P, direct ? nullptr : rgba_out, hit_id_out);
}

// vrt_render for a whole frame into memory the device cannot reach, i.e. pageable memory such as a
// managed Color[] pinned with GCHandle (the runtime copies into such memory through its own
// staging at about a third of the PCIe rate).  The frame is rendered in VRT_STAGED_BANDS bands of
// rows straight into page-locked staging of the context (the frame kernel stores there, see
// begin_frame), hit ids copied per band, and this thread copies a band to the caller's buffer as
// soon as its event has fired, while the next band is being traced.  Returns 1 when the call is
// not of that kind and the ordinary path has to take it.
#define VRT_STAGED_BANDS 8
#define VRT_STAGED_THREADS 2   // 4 bands: +3 %, 16: +12 %; more threads change nothing (1080p, MI355X host)
static int render_staged(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out, vrt_stats* stats_out)
{
    if (!c->opt.staged || !c->have_cam || p->tile_count > 1 || p->compact_rows || (p->flags & VRT_OUT_LINEAR_RGB)) return 1;
    const size_t W = (size_t)c->cam.width, H = (size_t)c->cam.height, px = W * H, pxb = frame_px_bytes(p->flags);
    if (px < (size_t)1 << 18) return 1;   // small frames: the bands' launches would cost more than they hide
    HIPCHK(c, hipSetDevice(c->device));
    hipPointerAttribute_t at;
    // memory the runtime knows (page-locked or device) is copied at PCIe speed as it is; an
    // unknown pointer is reported as an error by older runtimes and as "unregistered" by newer ones
    if (hipPointerGetAttributes(&at, rgba_out) == hipSuccess && at.type != hipMemoryTypeUnregistered) return 1;
    (void)hipGetLastError();
    if (px > c->stage_px) {
        if (c->stage_rgba) { vrt_host_free(c, c->stage_rgba); c->stage_rgba = nullptr; }
        if (c->stage_hit) { vrt_host_free(c, c->stage_hit); c->stage_hit = nullptr; }
        c->stage_px = 0;
        int rc = vrt_host_alloc(c, px * sizeof(float4), (void**)&c->stage_rgba);
        if (rc) return rc;
        rc = vrt_host_alloc(c, px * sizeof(int32_t), (void**)&c->stage_hit);
        if (rc) return rc;
        c->stage_px = px;
    }
    // a frame that is cheap to trace is cut into bands so that copying out overlaps tracing; the tree
    // kernels lose more in the tails of eight small launches than the overlap gives (1 M triangles
    // at 1080p: 5.7 ms in 8 bands, 2.9 ms in one), their frames are copied out after one launch
    const int n_bands = c->sb.use_bvh ? 1 : VRT_STAGED_BANDS, n_thr = VRT_STAGED_THREADS;
    const int band = (int)(((H + n_bands - 1) / n_bands + 7) & ~(size_t)7);
    const int n_parts = (int)((H + band - 1) / band);
    vrt_params q = *p;
    q.tile_rows = band; q.tile_index = 0; q.tile_count = 1;
    q.n_record_pixels = 0; q.record_pixels = nullptr;
    int rc = vrt_render_begin_parts(c, &q, c->stage_rgba, hit_id_out ? c->stage_hit : nullptr, n_parts);
    if (rc) return rc;
    // A band is traced in less time than one thread needs to copy it out, so a helper thread
    // copies with this one: each waits for a band's event and takes its share of the band's rows.
    hipError_t err[VRT_STAGED_THREADS] = {};
    auto copy_bands = [&](int me) {
        if (me >= 1 && hipSetDevice(c->device) != hipSuccess) { err[me] = hipErrorInvalidDevice; return; }
        for (int k = 0; k < n_parts; ++k) {
            const hipError_t e = hipEventSynchronize(c->part_ev[k]);
            if (e != hipSuccess) { err[me] = e; return; }
            const size_t b0 = (size_t)k * band, b1 = b0 + band < H ? b0 + band : H;
            const size_t r0 = b0 + (b1 - b0) * me / n_thr, r1 = b0 + (b1 - b0) * (me + 1) / n_thr;
            memcpy((char*)rgba_out + r0 * W * pxb, (const char*)c->stage_rgba + r0 * W * pxb, (r1 - r0) * W * pxb);
            if (hit_id_out) memcpy(hit_id_out + r0 * W, c->stage_hit + r0 * W, (r1 - r0) * W * sizeof(int32_t));
        }
    };
    {
        static_assert(VRT_STAGED_THREADS == 2, "one helper thread per context");
        if (!c->copy_helper) c->copy_helper = new HostHelper();
        c->copy_helper->post([&] { copy_bands(1); });
        copy_bands(0);
        c->copy_helper->wait();
    }
    for (int k = 0; k < VRT_STAGED_THREADS; ++k)
        if (err[k] != hipSuccess) {
            vrt_render_end(c, nullptr);
            return fail(c, VRT_ERR_HIP, std::string("staged frame: ") + hipGetErrorString(err[k]));
        }
    return vrt_render_end(c, stats_out);
}

extern "C" int vrt_render(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out, vrt_stats* stats_out)
{
    if (!c || !p || !rgba_out) return fail(c, VRT_ERR_INVALID, "null argument");
    if (p->n_record_pixels < 0 || (p->n_record_pixels > 0 && (!p->record_pixels || !p->record_count_out)))
        return fail(c, VRT_ERR_INVALID, "record arguments missing");
    if (c->multi) return vrt_multi_render(c, p, rgba_out, hit_id_out, stats_out);
    if (!c->pending) {
        const int rs = render_staged(c, p, rgba_out, hit_id_out, stats_out);
        if (rs < 0) return rs;
        if (rs == 0) {
            if (p->n_record_pixels > 0) {
                DevParams RP;
                int rr = prepare(c, p, &RP);
                if (rr) return rr;
                return run_records(c, p, RP, p->record_pixels, p->n_record_pixels, p->record_cap, -1, p->record_out,
                                   p->record_count_out, nullptr, nullptr, nullptr);
            }
            return VRT_OK;
        }
    }
    DevParams P;
    int rc = begin_frame(c, p, rgba_out, hit_id_out, &P);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (stats_out) rc = read_stats(c, stats_out); else rc = check_wf_overflow(c);
    if (rc) return rc;
    if (p->n_record_pixels > 0) {
        rc = run_records(c, p, P, p->record_pixels, p->n_record_pixels, p->record_cap, -1, p->record_out,
                         p->record_count_out, nullptr, nullptr, nullptr);
        if (rc) return rc;
    }
    return VRT_OK;
}

// vrt_render_begin on this context's own GPU: the frame (or the strips p names) and its copies to
// the caller's buffers are enqueued, ev_done marks their end.
int vrt_begin_one(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out)
{
    DevParams P;
    int rc = begin_frame(c, p, rgba_out, hit_id_out, &P);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev_done, c->stream));
    c->pending = true;
    c->parts_pending = 0;
    return VRT_OK;
}

extern "C" int vrt_render_begin(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out)
{
    if (!c || !p || !rgba_out) return fail(c, VRT_ERR_INVALID, "null argument");
    if (p->n_record_pixels != 0) return fail(c, VRT_ERR_INVALID, "path recording needs the blocking vrt_render");
    if (c->multi) return vrt_multi_begin(c, p, rgba_out, hit_id_out);
    return vrt_begin_one(c, p, rgba_out, hit_id_out);
}

// The frame in n_parts strip sets, each rendered, copied to the caller's buffers (its rows only)
// and marked with an event of its own.  All parts count into one counter block, so
// vrt_render_end reports the whole frame.
extern "C" int vrt_render_begin_parts(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out, int n_parts)
{
    if (!c || !p || !rgba_out) return fail(c, VRT_ERR_INVALID, "null argument");
    if (c->multi) return fail(c, VRT_ERR_INVALID, "a multi-GPU context renders with the blocking vrt_render");
    if (c->pending) return fail(c, VRT_ERR_STATE, "a frame started with vrt_render_begin is still in flight");
    if (p->n_record_pixels != 0) return fail(c, VRT_ERR_INVALID, "path recording needs the blocking vrt_render");
    if (p->flags & VRT_OUT_LINEAR_RGB) return fail(c, VRT_ERR_INVALID, "VRT_OUT_LINEAR_RGB is an output format of vrt_render_device");
    if (n_parts < 1 || n_parts > VRT_MAX_PARTS) return fail(c, VRT_ERR_INVALID, "n_parts out of range");
    if (p->tile_count > 1 || p->compact_rows) return fail(c, VRT_ERR_INVALID, "vrt_render_begin_parts cuts the whole frame itself");
    HIPCHK(c, hipSetDevice(c->device));
    while ((int)c->part_ev.size() < n_parts) {
        hipEvent_t e = nullptr;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->part_ev.push_back(e);
    }
    vrt_params q = *p;
    q.tile_count = n_parts; q.tile_rows = p->tile_rows > 0 ? p->tile_rows : 8; q.compact_rows = 0;
    int rc = ensure_fb(c);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_ctr, 0, kCtrBlock * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    // as in begin_frame: a buffer of vrt_host_alloc is written by the kernels themselves
    const bool store_direct = c->opt.zero_copy && !(p->flags & VRT_OUT_RGBA8);
    float4* direct = store_direct ? (float4*)device_view_of_host_alloc(c, rgba_out, (size_t)c->cam.height * c->cam.width * frame_px_bytes(p->flags)) : nullptr;
    uint64_t pixels = 0;
    for (int k = 0; k < n_parts; ++k) {
        q.tile_index = k;
        DevParams P;
        rc = prepare(c, &q, &P);
        if (rc) return rc;
        rc = enqueue_frame(c, &q, P, direct ? direct : c->d_rgba, hit_id_out ? c->d_hit : nullptr, c->stream, c->d_ctr, false);   // parts differ in layout from call to call
        if (rc) return rc;
        pixels += c->last_pixels;
        if (k == n_parts - 1) HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        rc = copy_rows_out(c, P, direct ? nullptr : rgba_out, hit_id_out);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(c->part_ev[k], c->stream));
    }
    c->last_pixels = pixels;
    c->timed = true;
    HIPCHK(c, hipEventRecord(c->ev_done, c->stream));
    c->pending = true;
    c->parts_pending = n_parts;
    return VRT_OK;
}

int vrt_poll_one(vrt_ctx* c, int* done)
{
    if (!c->pending) return fail(c, VRT_ERR_STATE, "no frame in flight");
    HIPCHK(c, hipSetDevice(c->device));
    hipError_t e = hipEventQuery(c->ev_done);
    if (e == hipSuccess) { *done = 1; return VRT_OK; }
    if (e == hipErrorNotReady) { *done = 0; (void)hipGetLastError(); return VRT_OK; }
    return fail(c, VRT_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
}

extern "C" int vrt_render_poll(vrt_ctx* c, int* done)
{
    if (!c || !done) return fail(c, VRT_ERR_INVALID, "null argument");
    if (!c->pending) return fail(c, VRT_ERR_STATE, "no frame in flight");
    if (c->multi) return vrt_multi_poll(c, done);
    return vrt_poll_one(c, done);
}

extern "C" int vrt_render_poll_parts(vrt_ctx* c, int* parts_done)
{
    if (!c || !parts_done) return fail(c, VRT_ERR_INVALID, "null argument");
    if (!c->pending || c->parts_pending <= 0) return fail(c, VRT_ERR_STATE, "no frame started with vrt_render_begin_parts in flight");
    HIPCHK(c, hipSetDevice(c->device));
    int n = 0;
    for (; n < c->parts_pending; ++n) {   // parts complete in order: they share one stream
        hipError_t e = hipEventQuery(c->part_ev[n]);
        if (e == hipErrorNotReady) { (void)hipGetLastError(); break; }
        if (e != hipSuccess) return fail(c, VRT_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
    }
    *parts_done = n;
    return VRT_OK;
}

int vrt_end_one(vrt_ctx* c, vrt_stats* stats_out)
{
    if (!c->pending) return fail(c, VRT_ERR_STATE, "no frame in flight");
    HIPCHK(c, hipSetDevice(c->device));
    c->pending = false;
    c->parts_pending = 0;
    HIPCHK(c, hipEventSynchronize(c->ev_done));
    if (stats_out) return read_stats(c, stats_out);
    return check_wf_overflow(c);
}

extern "C" int vrt_render_end(vrt_ctx* c, vrt_stats* stats_out)
{
    if (!c) return VRT_ERR_INVALID;
    if (!c->pending) return fail(c, VRT_ERR_STATE, "no frame in flight");
    if (c->multi) return vrt_multi_end(c, stats_out);
    return vrt_end_one(c, stats_out);
}

// vrt_render_device enqueues and returns: a wavefront frame whose queues overflowed is reported
// by the vrt_fetch_stats call that follows the caller's synchronisation, which is therefore the
// error check of this path.
int vrt_render_device_one(vrt_ctx* c, const vrt_params* p, void* rgba_dev, void* hit_id_dev, void* stream)
{
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

extern "C" int vrt_render_device(vrt_ctx* c, const vrt_params* p, void* rgba_dev, void* hit_id_dev, void* stream)
{
    if (!c || !p || !rgba_dev) return fail(c, VRT_ERR_INVALID, "null argument");
    if (c->multi) return fail(c, VRT_ERR_INVALID, "a multi-GPU context renders with vrt_render");
    return vrt_render_device_one(c, p, rgba_dev, hit_id_dev, stream);
}

// Path records of the listed pixels after a frame, on this context's own GPU.
int vrt_record_pixels_one(vrt_ctx* c, const vrt_params* p)
{
    HIPCHK(c, hipSetDevice(c->device));
    DevParams P;
    int rc = prepare(c, p, &P);
    if (rc) return rc;
    return run_records(c, p, P, p->record_pixels, p->n_record_pixels, p->record_cap, -1, p->record_out,
                       p->record_count_out, nullptr, nullptr, nullptr);
}

int vrt_read_stats_one(vrt_ctx* c, vrt_stats* s)
{
    HIPCHK(c, hipSetDevice(c->device));
    return read_stats(c, s);
}

// vrt_render_device for callers that cannot allocate: the megakernel counts into this device's
// copy of the library's static scratch array and samples with a copy of the table kept there
// too.  The array's address on this device was taken in vrt_create; nothing below allocates or frees.
extern "C" int vrt_render_device_noalloc(vrt_ctx* c, const vrt_params* p, void* rgba_dev, void* hit_id_dev, void* stream)
{
    if (!c || !p || !rgba_dev) return fail(c, VRT_ERR_INVALID, "null argument");
    if (c->multi) return fail(c, VRT_ERR_INVALID, "a multi-GPU context renders with vrt_render");
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
        if (c->device >= VRT_STATIC_MAX_DEVICES) return fail(c, VRT_ERR_CAPACITY, "device ordinal beyond VRT_STATIC_MAX_DEVICES");
        StaticTable& T = g_static_table[c->device];
        std::lock_guard<std::mutex> g(T.m);
        if (T.n != n_off || memcmp(T.host.data(), p->offsets, n_off * sizeof(float2)) != 0) {
            // A table the array does not hold yet goes through the context's page-locked staging, once
            // the copy of this context's previous upload has left it.  That wait is a host-side
            // synchronisation, which a capturing stream does not allow: under capture the table has
            // to be one an earlier, uncaptured call has put there.
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cap) != hipSuccess) (void)hipGetLastError();
            if (cap != hipStreamCaptureStatusNone)
                return fail(c, VRT_ERR_STATE, "vrt_render_device_noalloc under stream capture needs the sampler table of an earlier, "
                                              "uncaptured call on this GPU (a new table is uploaded through a host-side wait)");
            if (c->static_used) HIPCHK(c, hipEventSynchronize(c->ev_static));
            memcpy(c->static_stage, p->offsets, n_off * sizeof(float2));
            T.n = 0;   // not valid until the copy has been enqueued
            HIPCHK(c, hipMemcpyAsync(offs, c->static_stage, n_off * sizeof(float2), hipMemcpyHostToDevice, s));
            HIPCHK(c, hipEventRecord(c->ev_static, s));
            c->static_used = true;
            T.host.assign(p->offsets, p->offsets + 2 * n_off);   // within the capacity reserved by vrt_create
            T.n = n_off;
        }
        P.offsets = offs;
    }
    // the array keeps the last call's counts
    hipLaunchKernelGGL(k_clear_words, dim3((unsigned)((kCtrBlock + 255) / 256)), dim3(256), 0, s, ctr, (unsigned)kCtrBlock);
    HIPCHK(c, hipGetLastError());
    c->timed = false;
    P.flags |= p->flags & VRT_OUT_LINEAR_RGB;
    return enqueue_frame(c, p, P, (float4*)rgba_dev, (int*)hit_id_dev, s, ctr, false);   // ordering tiles would allocate
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
    c->sb.scene.dbg = c->d_ctr; c->last_ctr = c->d_ctr;
    c->sb.scene.stack_cap = c->opt.stack_cap > 0 ? c->opt.stack_cap : VRT_TRAV_STACK;
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
    const bool two = two_children(c, p);
    c->sb.scene.dbg = c->d_ctr; c->last_ctr = c->d_ctr;
    c->sb.scene.stack_cap = c->opt.stack_cap > 0 ? c->opt.stack_cap : VRT_TRAV_STACK;
    HIPCHK(c, VRT_LAUNCH(fused_form(c, origin), vrt_launch_ray, c->sb.scene, P, c->sb.use_bvh, two, origin, direction, depth, c->d_px_rgba, c->d_px_hit,
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
