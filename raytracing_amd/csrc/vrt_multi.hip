// Several GPUs of one node behind one vrt_ctx (vrt_create_multi): the Unity host is a single
// process with a single main thread (RayTracerUnity.cs:312-352), so the split over GPUs lives
// inside the library.
//
//   scene, camera   replicated: every GPU holds all primitives and builds its own tree.
//   frame           strips of tile_rows rows, strip s rendered by GPU s % n with all its samples
//                   (RayTracerUnity.cs:578-592 needs no exchange between pixels).
//   delivery        every GPU tone-maps its own pixels (Util/RayTraceUtility.cs:409-422 looks at one
//                   pixel at a time) and delivers the rows of its strips to the caller's buffer
//                   itself, over its own PCIe link: stored there by its frame kernel when the buffer
//                   comes from vrt_host_alloc, else copied strip by strip.  Nothing passes through
//                   the first GPU, so the frame's way to the host is n links wide, and the
//                   non-blocking calls (vrt_render_begin / poll / end) work as on one GPU.
//   gather          option multi_gather = 1 (study, and hosts that want the strips on one GPU):
//                   the GPUs write the averaged radiance of their strips, 3 floats per pixel, rows
//                   packed; the strips of GPUs 1..n-1 travel over their xGMI links into one buffer
//                   on the first GPU (peer-to-peer copies; one RCCL group of send/recv pairs when
//                   VRT_MULTI_RCCL=1 is set), one kernel there puts the rows back in frame order
//                   and tone-maps, then the frame is copied to the caller.
//
// One host thread per GPU enqueues that GPU's work.  RCCL is loaded only when asked for, so a host
// that renders on one GPU, or delivers per GPU, never loads it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "vrt_ctx.h"
#include "vrt_trace.h"

namespace {

// The few RCCL entry points used, resolved at run time.
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
        if (!r.lib) return;
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
        r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");
        r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.GetErrorString;
    });
    return r;
}

// A host thread that runs the jobs of one GPU in the order they are posted.
class Worker {
public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker()
    {
        { std::lock_guard<std::mutex> g(m_); quit_ = true; }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<int()> job)
    {
        { std::lock_guard<std::mutex> g(m_); job_ = std::move(job); busy_ = true; }
        cv_.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return !busy_; });
        return rc_;
    }
private:
    void loop()
    {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [this] { return quit_ || busy_; });
            if (quit_) return;
            std::function<int()> job = std::move(job_);
            g.unlock();
            const int rc = job();
            g.lock();
            rc_ = rc; busy_ = false;
            done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::function<int()> job_;
    bool busy_ = false, quit_ = false;
    int rc_ = 0;
    std::thread th_;   // last: started when everything above exists
};

struct StripLayout {
    int n, rows, width, height;
    unsigned long long first_px[VRT_MAX_GPUS];   // first pixel of GPU k's packed rows in the gathered buffer
};

// Frame row h, column v from the gathered strips: radiance -> process_color, and the hit id.
__global__ __launch_bounds__(256) void k_assemble(StripLayout L, unsigned out_flags, const float* __restrict__ rgb,
                                                  const int* __restrict__ hits, float4* __restrict__ rgba_out,
                                                  int* __restrict__ hit_out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)L.width * L.height) return;
    const int h = (int)(i / L.width), v = (int)(i % L.width);
    const int s = h / L.rows, k = s % L.n;
    const size_t src = L.first_px[k] + ((size_t)(s / L.n) * L.rows + (h % L.rows)) * L.width + v;
    DevParams P;
    P.flags = out_flags;   // the frame's own format: Color, or Color32 with VRT_OUT_RGBA8
    store_pixel(P, rgba_out, i, mk3(rgb[3 * src], rgb[3 * src + 1], rgb[3 * src + 2]));
    if (hit_out) hit_out[i] = hits[src];
}

int strip_rows_of(int H, int index, int count, int rows)
{
    int n = 0;
    for (int s = index; s * rows < H; s += count) n += (s * rows + rows < H ? rows : H - s * rows);
    return n;
}

} // namespace

struct VrtMulti {
    int n = 0;
    std::vector<int> dev;
    std::vector<vrt_ctx*> sub;            // sub[0] = the context the caller holds
    std::vector<Worker*> workers;         // workers[k] drives GPU k, k >= 1 (GPU 0 is driven by the calling thread)
    std::vector<ncclComm_t> comm;         // empty: peer-to-peer copies
    std::vector<float*> d_strip;          // GPUs k >= 1: packed radiance rows of their strips
    std::vector<int*> d_strip_hit;
    std::vector<hipEvent_t> ev_sent;      // GPUs k >= 1: their strips have left (peer-to-peer form)
    std::vector<size_t> strip_cap;        // pixels
    float* d_gather = nullptr; int* d_gather_hit = nullptr; size_t gather_cap = 0;   // first GPU
};

#define NCCLCHK(c, expr)                                                                              \
    do {                                                                                              \
        ncclResult_t r_ = (expr);                                                                     \
        if (r_ != ncclSuccess)                                                                        \
            return vrt_fail(c, VRT_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_));   \
    } while (0)

// Runs f(k) for every GPU: k >= 1 on its worker, k = 0 on the calling thread.  Returns the first failure.
template <class F>
static int on_all(VrtMulti* m, F f)
{
    for (int k = 1; k < m->n; ++k) m->workers[k]->post([f, k] { return f(k); });
    int rc = f(0);
    for (int k = 1; k < m->n; ++k) { int r = m->workers[k]->wait(); if (rc == VRT_OK && r != VRT_OK) rc = r; }
    return rc;
}

static void copy_error(vrt_ctx* c, VrtMulti* m)
{
    for (int k = 1; k < m->n; ++k)
        if (!m->sub[k]->err.empty()) { c->err = "GPU " + std::to_string(m->dev[k]) + ": " + m->sub[k]->err; m->sub[k]->err.clear(); return; }
}

int vrt_multi_init(vrt_ctx* c, int n_gpus, const int* devices)
{
    VrtMulti* m = new VrtMulti();
    m->n = n_gpus;
    m->dev.resize(n_gpus); m->sub.assign(n_gpus, nullptr); m->workers.assign(n_gpus, nullptr);
    m->d_strip.assign(n_gpus, nullptr); m->d_strip_hit.assign(n_gpus, nullptr); m->strip_cap.assign(n_gpus, 0);
    m->ev_sent.assign(n_gpus, nullptr);
    for (int k = 0; k < n_gpus; ++k) m->dev[k] = devices ? devices[k] : k;
    m->sub[0] = c;
    c->multi = m;   // from here on vrt_destroy releases what has been built
    bool distinct = true;
    for (int k = 0; k < n_gpus; ++k)
        for (int j = 0; j < k; ++j) if (m->dev[j] == m->dev[k]) distinct = false;
    for (int k = 1; k < n_gpus; ++k) {
        int rc = vrt_create(m->dev[k], &m->sub[k]);
        if (rc != VRT_OK) return vrt_fail(c, rc, "vrt_create failed on GPU " + std::to_string(m->dev[k]));
        m->sub[k]->opt = c->opt;
        m->sub[k]->host_owner = c;
        m->workers[k] = new Worker();
        HIPCHK(c, hipSetDevice(m->dev[k]));
        HIPCHK(c, hipEventCreateWithFlags(&m->ev_sent[k], hipEventDisableTiming));
        if (m->dev[k] != m->dev[0]) {   // direct loads/stores over the link for the peer-to-peer form
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, m->dev[k], m->dev[0]) == hipSuccess && can) {
                hipError_t e = hipDeviceEnablePeerAccess(m->dev[0], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
            (void)hipGetLastError();
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    // The gather's RCCL form has never run on more than one GPU (the pool this library is developed
    // on hands out one GPU at a time), so it has to be asked for; peer-to-peer copies are the
    // tested form.
    const char* want_rccl = getenv("VRT_MULTI_RCCL");
    if (n_gpus > 1 && distinct && want_rccl && atoi(want_rccl) && rccl().ok) {
        m->comm.assign(n_gpus, nullptr);
        ncclResult_t r = rccl().CommInitAll(m->comm.data(), n_gpus, m->dev.data());
        if (r != ncclSuccess) { m->comm.clear(); (void)hipGetLastError(); }   // peer-to-peer copies instead
        HIPCHK(c, hipSetDevice(c->device));
    }
    return VRT_OK;
}

void vrt_multi_destroy(vrt_ctx* c)
{
    VrtMulti* m = c->multi;
    if (!m) return;
    for (int k = 0; k < m->n; ++k) {
        if (!m->sub[k]) continue;
        hipSetDevice(m->dev[k]);
        hipDeviceSynchronize();
    }
    for (ncclComm_t cm : m->comm) if (cm) rccl().CommDestroy(cm);
    for (int k = 1; k < m->n; ++k) {
        delete m->workers[k];
        if (!m->sub[k]) continue;
        hipSetDevice(m->dev[k]);
        hipFree(m->d_strip[k]); hipFree(m->d_strip_hit[k]);
        if (m->ev_sent[k]) hipEventDestroy(m->ev_sent[k]);
        vrt_destroy(m->sub[k]);
    }
    hipSetDevice(c->device);
    hipFree(m->d_gather); hipFree(m->d_gather_hit);
    (void)hipGetLastError();
    c->multi = nullptr;
    delete m;
}

int vrt_multi_set_option(vrt_ctx* c, const char*, long long)
{
    VrtMulti* m = c->multi;
    for (int k = 1; k < m->n; ++k) m->sub[k]->opt = c->opt;
    return VRT_OK;
}

// Every GPU assembles the new scene beside its old one; the scenes are swapped in only when all
// GPUs have succeeded, so a failure on one of them leaves all of them on the scene they had.
int vrt_multi_set_scene(vrt_ctx* c, const vrt_scene_desc* d)
{
    VrtMulti* m = c->multi;
    std::vector<SceneBufs> nb(m->n);
    int rc = on_all(m, [&](int k) { return vrt_build_scene_one(m->sub[k], d, nb[k]); });
    if (rc != VRT_OK) {
        copy_error(c, m);
        for (SceneBufs& b : nb) b.release();   // those that did succeed (hipFree takes any device's pointer)
        return rc;
    }
    rc = on_all(m, [&](int k) { return vrt_swap_scene_one(m->sub[k], nb[k]); });
    if (rc != VRT_OK) copy_error(c, m);
    return rc;
}

int vrt_multi_set_camera(vrt_ctx* c, const vrt_camera* cam)
{
    VrtMulti* m = c->multi;
    int rc = on_all(m, [&](int k) { return vrt_set_camera_one(m->sub[k], cam); });
    if (rc != VRT_OK) copy_error(c, m);
    return rc;
}

// Waits for whatever the GPUs of the context have been given (an error path: a call that fails
// after some GPUs have enqueued their part must not return with work in flight on the caller's buffers).
static void drain_all(VrtMulti* m)
{
    for (int k = 0; k < m->n; ++k) {
        if (!m->sub[k]) continue;
        if (hipSetDevice(m->dev[k]) == hipSuccess) (void)hipStreamSynchronize(m->sub[k]->stream);
        m->sub[k]->pending = false; m->sub[k]->parts_pending = 0;
    }
    (void)hipGetLastError();
    (void)hipSetDevice(m->dev[0]);
}

static int multi_check(vrt_ctx* c, const vrt_params* p)
{
    if (!c->have_scene || !c->have_cam) return vrt_fail(c, VRT_ERR_STATE, "scene or camera not set");
    if (c->pending) return vrt_fail(c, VRT_ERR_STATE, "a frame started with vrt_render_begin is still in flight");
    if (p->tile_count > 1 || p->compact_rows) return vrt_fail(c, VRT_ERR_INVALID, "a multi-GPU context cuts the frame itself");
    if (p->flags & VRT_OUT_LINEAR_RGB) return vrt_fail(c, VRT_ERR_INVALID, "VRT_OUT_LINEAR_RGB is an output format of vrt_render_device");
    return VRT_OK;
}

// vrt_render_begin on every GPU: GPU k enqueues the strips s with s % n == k and their way to the
// caller's buffers on its own stream (vrt_begin_one with the strip parameters a rank of a
// multi-process job would pass).  The frame is complete when every GPU's is.
int vrt_multi_begin(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out)
{
    VrtMulti* m = c->multi;
    int rc = multi_check(c, p);
    if (rc != VRT_OK) return rc;
    const int n = m->n, rows = p->tile_rows > 0 ? p->tile_rows : 8;
    rc = on_all(m, [&](int k) -> int {
        vrt_params q = *p;
        q.tile_index = k; q.tile_count = n; q.tile_rows = rows; q.compact_rows = 0;
        q.n_record_pixels = 0; q.record_pixels = nullptr;
        return vrt_begin_one(m->sub[k], &q, rgba_out, hit_id_out);
    });
    if (rc != VRT_OK) { copy_error(c, m); drain_all(m); return rc; }
    (void)hipSetDevice(c->device);
    return VRT_OK;
}

int vrt_multi_poll(vrt_ctx* c, int* done)
{
    VrtMulti* m = c->multi;
    *done = 1;
    for (int k = 0; k < m->n; ++k) {
        int d = 0;
        int rc = vrt_poll_one(m->sub[k], &d);
        if (rc != VRT_OK) { if (k > 0) copy_error(c, m); (void)hipSetDevice(c->device); return rc; }
        if (!d) { *done = 0; break; }
    }
    (void)hipSetDevice(c->device);
    return VRT_OK;
}

// Statistics are the sum over the GPUs; gpu_ms is the longest of their device times, which run side by side.
int vrt_multi_end(vrt_ctx* c, vrt_stats* stats_out)
{
    VrtMulti* m = c->multi;
    vrt_stats sum;
    memset(&sum, 0, sizeof sum);
    int rc = VRT_OK;
    for (int k = 0; k < m->n; ++k) {
        vrt_stats s;
        memset(&s, 0, sizeof s);
        const int r = vrt_end_one(m->sub[k], &s);
        if (r != VRT_OK) { if (rc == VRT_OK) { rc = r; if (k > 0) copy_error(c, m); } continue; }
        for (int q = 0; q < VRT_RAY_CLASSES; ++q) sum.rays[q] += s.rays[q];
        sum.nodes_visited += s.nodes_visited; sum.prims_tested += s.prims_tested;
        sum.pixels += s.pixels;
        if (s.gpu_ms > sum.gpu_ms) sum.gpu_ms = s.gpu_ms;
        if (k == 0) { sum.bvh_build_ms = s.bvh_build_ms; sum.bvh_nodes = s.bvh_nodes; sum.mode_used = s.mode_used; }
    }
    (void)hipSetDevice(c->device);
    if (rc != VRT_OK) { drain_all(m); return rc; }
    c->last_pixels = sum.pixels;
    if (stats_out) *stats_out = sum;
    return VRT_OK;
}

static int multi_render_gather(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out, vrt_stats* stats_out);

int vrt_multi_render(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out, vrt_stats* stats_out)
{
    VrtMulti* m = c->multi;
    int rc = multi_check(c, p);
    if (rc != VRT_OK) return rc;
    if (c->opt.multi_gather) {
        rc = multi_render_gather(c, p, rgba_out, hit_id_out, stats_out);
        if (rc != VRT_OK) drain_all(m);
        return rc;
    }
    rc = vrt_multi_begin(c, p, rgba_out, hit_id_out);
    if (rc != VRT_OK) return rc;
    rc = vrt_multi_end(c, stats_out);
    if (rc != VRT_OK) return rc;
    if (p->n_record_pixels > 0) return vrt_record_pixels_one(c, p);   // the scene is replicated: the first GPU traces them
    return VRT_OK;
}

// The frame with its strips gathered on the first GPU (option multi_gather).
static int multi_render_gather(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out, vrt_stats* stats_out)
{
    VrtMulti* m = c->multi;
    const int W = c->cam.width, H = c->cam.height, n = m->n;
    const int rows = p->tile_rows > 0 ? p->tile_rows : 8;
    const bool want_hits = hit_id_out != nullptr;
    StripLayout L;
    memset(&L, 0, sizeof L);
    L.n = n; L.rows = rows; L.width = W; L.height = H;
    std::vector<size_t> px(n);
    size_t total = 0;
    for (int k = 0; k < n; ++k) { L.first_px[k] = total; px[k] = (size_t)strip_rows_of(H, k, n, rows) * W; total += px[k]; }

    // buffers: gathered strips, final frame and hit ids on the first GPU; a strip buffer on the others
    HIPCHK(c, hipSetDevice(c->device));
    if (total > m->gather_cap) {
        HIPCHK(c, hipDeviceSynchronize());
        hipFree(m->d_gather); hipFree(m->d_gather_hit); m->d_gather = nullptr; m->d_gather_hit = nullptr; m->gather_cap = 0;
        HIPCHK(c, hipMalloc(&m->d_gather, total * 3 * sizeof(float)));
        HIPCHK(c, hipMalloc(&m->d_gather_hit, total * sizeof(int)));
        m->gather_cap = total;
    }
    if ((size_t)W * H > c->fb_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(c->d_rgba); hipFree(c->d_hit); c->d_rgba = nullptr; c->d_hit = nullptr; c->fb_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_rgba, (size_t)W * H * sizeof(float4)));
        HIPCHK(c, hipMalloc(&c->d_hit, (size_t)W * H * sizeof(int)));
        c->fb_cap = (size_t)W * H;
    }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));

    // every GPU renders its strips on its own stream, enqueued by its own host thread
    int rc = on_all(m, [&](int k) -> int {
        vrt_ctx* s = m->sub[k];
        hipError_t e = hipSetDevice(m->dev[k]);
        if (e != hipSuccess) return vrt_fail(s, VRT_ERR_HIP, hipGetErrorString(e));
        if (px[k] == 0) { s->last_pixels = 0; return VRT_OK; }
        float* rgb = k == 0 ? m->d_gather + 3 * L.first_px[0] : nullptr;
        int* hit = k == 0 ? m->d_gather_hit + L.first_px[0] : nullptr;
        if (k > 0) {
            if (px[k] > m->strip_cap[k]) {
                if ((e = hipDeviceSynchronize()) != hipSuccess) return vrt_fail(s, VRT_ERR_HIP, hipGetErrorString(e));
                hipFree(m->d_strip[k]); hipFree(m->d_strip_hit[k]); m->d_strip[k] = nullptr; m->d_strip_hit[k] = nullptr; m->strip_cap[k] = 0;
                if ((e = hipMalloc(&m->d_strip[k], px[k] * 3 * sizeof(float))) != hipSuccess ||
                    (e = hipMalloc(&m->d_strip_hit[k], px[k] * sizeof(int))) != hipSuccess)
                    return vrt_fail(s, VRT_ERR_HIP, hipGetErrorString(e));
                m->strip_cap[k] = px[k];
            }
            rgb = m->d_strip[k]; hit = m->d_strip_hit[k];
        }
        vrt_params q = *p;
        q.tile_index = k; q.tile_count = n; q.tile_rows = rows; q.compact_rows = 1;
        q.flags = (q.flags & ~(unsigned)VRT_OUT_RGBA8) | VRT_OUT_LINEAR_RGB;   // strips travel as radiance; the frame's format is applied after the gather
        q.n_record_pixels = 0;
        int r = vrt_render_device_one(s, &q, rgb, want_hits ? hit : nullptr, s->stream);
        if (r != VRT_OK) return r;
        if (k > 0 && m->comm.empty()) {
            // peer-to-peer form: the strip leaves on this GPU's stream, right behind its rendering
            if ((e = hipMemcpyPeerAsync(m->d_gather + 3 * L.first_px[k], m->dev[0], rgb, m->dev[k], px[k] * 3 * sizeof(float), s->stream)) != hipSuccess)
                return vrt_fail(s, VRT_ERR_HIP, hipGetErrorString(e));
            if (want_hits && (e = hipMemcpyPeerAsync(m->d_gather_hit + L.first_px[k], m->dev[0], hit, m->dev[k], px[k] * sizeof(int), s->stream)) != hipSuccess)
                return vrt_fail(s, VRT_ERR_HIP, hipGetErrorString(e));
            if ((e = hipEventRecord(m->ev_sent[k], s->stream)) != hipSuccess) return vrt_fail(s, VRT_ERR_HIP, hipGetErrorString(e));
        }
        return VRT_OK;
    });
    if (rc != VRT_OK) { copy_error(c, m); return rc; }

    // gather to the first GPU
    HIPCHK(c, hipSetDevice(c->device));
    if (!m->comm.empty()) {
        Rccl& R = rccl();
        NCCLCHK(c, R.GroupStart());
        for (int k = 1; k < n; ++k) {
            if (px[k] == 0) continue;
            NCCLCHK(c, R.Send(m->d_strip[k], px[k] * 3, ncclFloat32, 0, m->comm[k], m->sub[k]->stream));
            NCCLCHK(c, R.Recv(m->d_gather + 3 * L.first_px[k], px[k] * 3, ncclFloat32, k, m->comm[0], c->stream));
            if (want_hits) {
                NCCLCHK(c, R.Send(m->d_strip_hit[k], px[k], ncclInt32, 0, m->comm[k], m->sub[k]->stream));
                NCCLCHK(c, R.Recv(m->d_gather_hit + L.first_px[k], px[k], ncclInt32, k, m->comm[0], c->stream));
            }
        }
        NCCLCHK(c, R.GroupEnd());
        HIPCHK(c, hipSetDevice(c->device));
    } else {
        for (int k = 1; k < n; ++k)
            if (px[k]) HIPCHK(c, hipStreamWaitEvent(c->stream, m->ev_sent[k], 0));
    }

    // rows back in frame order, tone mapping, read-back
    const size_t npix = (size_t)W * H;
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, c->stream, L, p->flags & VRT_OUT_RGBA8, m->d_gather,
                       want_hits ? m->d_gather_hit : nullptr, c->d_rgba, want_hits ? c->d_hit : nullptr);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipMemcpyAsync(rgba_out, c->d_rgba, npix * ((p->flags & VRT_OUT_RGBA8) ? 4 : sizeof(float4)), hipMemcpyDeviceToHost, c->stream));
    if (want_hits) HIPCHK(c, hipMemcpyAsync(hit_id_out, c->d_hit, npix * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int k = 1; k < n; ++k) {   // senders are done when the receiver is; make their streams say so too
        HIPCHK(c, hipSetDevice(m->dev[k]));
        HIPCHK(c, hipStreamSynchronize(m->sub[k]->stream));
    }
    HIPCHK(c, hipSetDevice(c->device));

    // statistics: the sum over the GPUs; gpu_ms = first GPU's stream from before its rendering to
    // the assembled frame (it waits for every strip, so this is the frame's device time)
    vrt_stats sum;
    memset(&sum, 0, sizeof sum);
    for (int k = 0; k < n; ++k) {
        if (px[k] == 0) continue;
        vrt_stats s;
        m->sub[k]->timed = false;
        rc = vrt_read_stats_one(m->sub[k], &s);
        if (rc != VRT_OK) { if (k > 0) copy_error(c, m); return rc; }
        for (int q = 0; q < VRT_RAY_CLASSES; ++q) sum.rays[q] += s.rays[q];
        sum.nodes_visited += s.nodes_visited; sum.prims_tested += s.prims_tested;
        sum.pixels += s.pixels;
        sum.bvh_build_ms = s.bvh_build_ms; sum.bvh_nodes = s.bvh_nodes; sum.mode_used = s.mode_used;
    }
    HIPCHK(c, hipSetDevice(c->device));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) sum.gpu_ms = ms;
    c->last_pixels = sum.pixels;
    if (stats_out) *stats_out = sum;
    if (p->n_record_pixels > 0) return vrt_record_pixels_one(c, p);
    return VRT_OK;
}
