// Private state behind the opaque vrt_ctx of include/vrt.h, shared by vrt_api.hip (one GPU) and
// vrt_multi.hip (several GPUs of one node behind the same entry points).
#pragma once
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "vrt_launch.h"

// Study and test knobs.  Environment variables of the same meaning (VRT_XCD_REMAP, VRT_SAMPLE_SHIFT,
// VRT_WF_DEFER, VRT_WF_REFILL, VRT_WF_MAX_SLOTS, VRT_WF_ANY_CAP, VRT_BUILDER, VRT_COLLAPSE_PARITY,
// VRT_NO_BIG, VRT_PACKET, VRT_BEAM, VRT_BLOCK_COMPACT, VRT_SLAB_FMA) are read once, when the context is created; vrt_set_option changes
// them afterwards.  Nothing on the per-frame path reads the environment.
struct VrtOptions {
    int xcd_remap = 0;          // 1: each XCD renders one contiguous band of tiles (measured slower)
    int sample_shift = -1;      // megakernel lanes per pixel = 1 << shift; -1 = chosen from spp
    int wf_defer = 1;           // wavefront: occlusion queries of the bounces in a queue of their own
    int wf_refill = 1;          // wavefront: queues traced with lanes refilled as they finish
    long long wf_max_slots = 0; // wavefront: sample slots per band (0 = library default)
    long long wf_queue_cap = 0; // wavefront, tests: ray queues cut to this many entries (0 = as allocated)
    long long wf_any_cap = -1;  // wavefront: capacity of the occlusion-query queue (-1 = full)
    int builder = -1;           // binary tree under the 4-wide collapse: 0 PLOC, 1 Karras' radix tree, 2 binned SAH built on the host, 3 the same on the device, -1 = 3 up to 4 M primitives, else PLOC
    int collapse_parity = 0;    // 1: collapse every second level instead of by surface area
    int no_big = 0;             // 1: scene-sized primitives stay in the tree
    int packet = -1;            // tree scenes: wave-uniform traversal for the rays of one pixel group; -1 = chosen from spp
    int beam = -1;              // tree scenes, megakernel: 1 = the camera rays of a wave, and the occlusion rays of their hits, collect their candidate leaves together; 2 = camera rays only; -1 = library default (1)
    int quad_max = -1;          // tree scenes: waves with at most this many queries trace them four lanes per ray; -1 = library default
    int slab_fma = 1;           // tree scenes: fused box test where the scene lies near enough to the origin for it (0: never)
    int stack_cap = 0;          // tests: traversal-stack entries per lane (0 = all the kernels have), forces the stackless finish
    int block_compact = 0;      // tree scenes, one child per hit: bounces compacted inside four-wave blocks (0: one wave per tile through all bounces)
    int staged = 1;             // vrt_render into pageable memory goes through page-locked staging in bands
    int tile_sort = 1;          // tree scenes, megakernel: tiles dispatched most expensive first, by the costs an earlier frame measured
    int zero_copy = 1;          // frames into a vrt_host_alloc buffer are stored there by the frame kernel itself
    int fail_after_build = 0;   // tests: vrt_set_scene reports a failure after uploads and tree build have succeeded
    int multi_gather = 0;       // multi-GPU context: 0 = every GPU delivers its strips to the caller's buffer itself, over its own PCIe link; 1 = strips gathered on the first GPU over xGMI and assembled there
};

// Everything vrt_set_scene uploads or builds.  A new scene is assembled in a fresh SceneBufs and
// swapped in only when every step has succeeded, so a failed call leaves the old scene in place.
struct SceneBufs {
    float4 *d_sph = nullptr, *d_box = nullptr, *d_tri = nullptr, *d_mat = nullptr, *d_bvh = nullptr, *d_bvh4 = nullptr;
    int* d_prim_mat = nullptr;
    DevLight* d_lights = nullptr;
    float4 *d_aab = nullptr, *d_cam_pre = nullptr, *d_tex_data = nullptr;
    int* d_mat_tex = nullptr; int4* d_tex_info = nullptr; float2* d_tri_uv = nullptr;
    std::vector<float4> h_sph, h_box; // host copies for the camera-origin terms
    DevScene scene;
    bool use_bvh = false;
    float bvh_ms = 0.f;
    int bvh_depth = 0;
    double max_abs = 0;   // tree scenes: largest |coordinate| of the scene's bounds
    void release();
};

// One cached upload of a sampler offset table.  Frames in flight keep reading the table they were
// enqueued with, so a changed table goes to another slot instead of overwriting the one in use.
struct OffsetSlot {
    float2* d = nullptr;
    size_t cap = 0, n = 0;
    std::vector<float> host;   // what the slot holds, for the comparison with the caller's table
    unsigned long long last_use = 0;
};

struct VrtMulti;   // vrt_multi.hip

// A host thread that runs one job at a time for its context (the second copier of frames staged
// for pageable memory): started on first use, joined when the context goes.
class HostHelper {
public:
    HostHelper() : th_([this] { loop(); }) {}
    ~HostHelper()
    {
        { std::lock_guard<std::mutex> g(m_); quit_ = true; }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<void()> job)
    {
        { std::lock_guard<std::mutex> g(m_); job_ = std::move(job); busy_ = true; }
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return !busy_; });
    }
private:
    void loop()
    {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [this] { return quit_ || busy_; });
            if (quit_) return;
            std::function<void()> job = std::move(job_);
            g.unlock();
            job();
            g.lock();
            busy_ = false;
            done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::function<void()> job_;
    bool busy_ = false, quit_ = false;
    std::thread th_;   // last: started when everything above exists
};

struct vrt_ctx {
    int device = 0;
    std::string err;
    VrtOptions opt;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr;
    bool pending = false; // a vrt_render_begin frame is in flight
    // frames delivered in parts (vrt_render_begin_parts)
    std::vector<hipEvent_t> part_ev;
    int parts_pending = 0;
    // scene
    SceneBufs sb;
    bool have_scene = false;
    // camera
    DevCam cam;
    bool have_cam = false;
    // sampler tables
    OffsetSlot off[4];
    unsigned long long off_clock = 0;
    unsigned long long* d_ctr = nullptr;       // counter block of the multi-kernel paths, cleared per call
    // The megakernel path takes its counter block from a ring of kCtrRing blocks that is cleared
    // with one fill every kCtrRing frames instead of one fill launch (~5 us) per frame.
    unsigned long long* d_ctr_ring = nullptr;
    unsigned ring_next = 0;
    hipStream_t ring_stream = nullptr;         // stream the frames of the current ring round ran on
    unsigned long long* last_ctr = nullptr;    // block the last call counted into
    float4* d_rgba = nullptr; int* d_hit = nullptr; size_t fb_cap = 0;
    int* d_rec_pixels = nullptr; vrt_path_seg* d_segs = nullptr; int* d_seg_counts = nullptr;
    size_t rec_px_cap = 0, rec_seg_cap = 0;
    float4* d_px_rgba = nullptr; int* d_px_hit = nullptr;
    // host buffers the caller asked us to page-lock (vrt_host_register)
    std::vector<std::pair<void*, size_t>> registered;
    std::vector<void*> host_allocs;            // page-locked buffers handed out by vrt_host_alloc
    std::vector<size_t> host_alloc_bytes;      // their sizes, same order
    // Dispatch order of the megakernel's tiles (tree scenes): costs a frame measured, the order
    // sorted from them, and the frame layout both belong to.
    unsigned* d_tile_cost = nullptr; unsigned* d_tile_keys = nullptr; int* d_tile_iota = nullptr; int* d_tile_order = nullptr;
    void* d_tile_tmp = nullptr; size_t tile_cap = 0, tile_tmp_bytes = 0;
    long long tile_sig[10] = {0}; bool tile_valid = false; hipStream_t tile_stream = nullptr; unsigned tile_age = 0;
    unsigned long long scene_gen = 0;          // bumped by every scene swap
    // page-locked staging of vrt_render for callers whose frame buffer the device cannot reach (pageable memory)
    float* stage_rgba = nullptr; int32_t* stage_hit = nullptr; size_t stage_px = 0;
    HostHelper* copy_helper = nullptr;
    // wavefront workspace
    WfWorkspace wf; bool wf_two = false; bool wf_short = false;
    // the device's copy of the library's static scratch array (vrt_render_device_noalloc) and the
    // page-locked staging its sampler table passes through
    unsigned char* static_scratch = nullptr;
    float* static_stage = nullptr;
    hipEvent_t ev_static = nullptr; bool static_used = false;
    // last call
    uint64_t last_pixels = 0; int last_mode = 0;
    bool timed = false;
    // several GPUs behind this context (vrt_create_multi); null for a one-GPU context
    VrtMulti* multi = nullptr;
    // the context whose vrt_host_alloc buffers this one may store frames into: the context the
    // caller holds, for the other GPUs' contexts of a multi-GPU context; null = its own
    vrt_ctx* host_owner = nullptr;
};

int vrt_fail(vrt_ctx* c, int code, const std::string& msg);

#define HIPCHK(c, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (void)hipGetLastError(); /* reported here: must not show up at the next launch */   \
            return vrt_fail(c, VRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                                       \
    } while (0)

// The single-GPU bodies of the entry points, on this context's own GPU whether or not it is the
// first GPU of a multi-GPU context.
int vrt_set_camera_one(vrt_ctx* c, const vrt_camera* cam);
int vrt_build_scene_one(vrt_ctx* c, const vrt_scene_desc* d, SceneBufs& nb);
int vrt_swap_scene_one(vrt_ctx* c, SceneBufs& nb);
int vrt_render_device_one(vrt_ctx* c, const vrt_params* p, void* rgba_dev, void* hit_id_dev, void* stream);
int vrt_record_pixels_one(vrt_ctx* c, const vrt_params* p);
int vrt_read_stats_one(vrt_ctx* c, vrt_stats* s);
int vrt_begin_one(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out);
int vrt_poll_one(vrt_ctx* c, int* done);
int vrt_end_one(vrt_ctx* c, vrt_stats* stats_out);

// vrt_multi.hip: the entry points of include/vrt.h on a context that spans several GPUs.
int vrt_multi_init(vrt_ctx* c, int n_gpus, const int* devices);
int vrt_multi_set_scene(vrt_ctx* c, const vrt_scene_desc* d);
int vrt_multi_set_camera(vrt_ctx* c, const vrt_camera* cam);
int vrt_multi_render(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out, vrt_stats* stats_out);
int vrt_multi_begin(vrt_ctx* c, const vrt_params* p, float* rgba_out, int32_t* hit_id_out);
int vrt_multi_poll(vrt_ctx* c, int* done);
int vrt_multi_end(vrt_ctx* c, vrt_stats* stats_out);
int vrt_multi_set_option(vrt_ctx* c, const char* name, long long value);
void vrt_multi_destroy(vrt_ctx* c);
