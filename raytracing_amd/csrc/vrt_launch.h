// Host-callable launchers implemented in vrt_kernels.hip / vrt_bvh.hip / vrt_wavefront.hip.
#pragma once
#include "vrt_device.h"

// The persistent kernels of the wavefront pipeline run at most VRT_WF_MAX_BLOCKS blocks of four
// waves; a wave takes queue entries VRT_WF_CHUNK at a time and may leave its last chunk partly
// unwritten, so queues are allocated VRT_WF_SLACK entries longer than the rays they must hold.
#ifndef VRT_WF_CHUNK
#define VRT_WF_CHUNK 256
#endif
#define VRT_WF_MAX_BLOCKS 4096
#define VRT_WF_SLACK ((size_t)VRT_WF_MAX_BLOCKS * 4 * VRT_WF_CHUNK)

// Wavefront pipeline workspace: ray queues (two, ping-pong), one radiance slot per sample of the
// band of rows in flight, queue counters.
struct WfWorkspace {
    RayQueue q[2];
    float* arena;       // backing store of both queues
    size_t cap;         // entries per queue; unwritten ones carry a negative slot (pix)
    size_t slots;       // sample slots (pixels of a band x samples per pixel)
    float4* lacc;       // per slot: radiance of that sample (xyz)
    int* counts;        // queue sizes per depth + overflow flag, VRT_MAX_DEPTH + 3 ints
    // deferred occlusion queries (null hits = the bounce kernel traces them on the spot)
    int4* hits;         // per ray of the queue being worked on: (t bits, primitive, face, answer bits)
    AnyQueue any;
    size_t any_cap;     // queries the queue holds; what does not fit is traced on the spot
    int* any_counts;    // queries per depth, VRT_MAX_DEPTH + 2 ints
    int* cursors;       // k_wf_trace: next unclaimed entry per depth, rays then queries, 2 * (VRT_MAX_DEPTH + 2) ints
                        // right behind any_counts (one fill clears all three)
    int refill;         // BVH scenes: trace queues with k_wf_trace (lanes refilled as they finish)
};

// The launchers exist twice, in namespaces vrt_plain and vrt_fused: the kernel sources are
// compiled once as they are and once with -DVRT_SLAB_FMA=1 (the 4-wide box test in its fused
// form, see slab4 in vrt_trace.h), and VRT_FORM names the namespace of the compilation at hand.
// The fused set serves tree scenes only (bvh = true; anything else is hipErrorInvalidValue).
#if defined(VRT_SLAB_FMA) && VRT_SLAB_FMA
#define VRT_FORM vrt_fused
#else
#define VRT_FORM vrt_plain
#endif
#define VRT_LAUNCHERS \
    hipError_t vrt_launch_mega(const DevScene& S, const DevCam& cam, const DevParams& P, bool bvh, bool twochild, \
                               float4* rgba, int* hit_id, unsigned long long* ctr, hipStream_t stream); \
    \
    hipError_t vrt_launch_mega_blk(const DevScene& S, const DevCam& cam, const DevParams& P, float4* rgba, int* hit_id, \
                                   unsigned long long* ctr, hipStream_t stream); \
    \
    hipError_t vrt_launch_pixels(const DevScene& S, const DevCam& cam, const DevParams& P, bool bvh, bool twochild, \
                                 const int* pixels, int n_pixels, int sample_sel, vrt_path_seg* segs, int cap, \
                                 int* seg_counts, float4* rgba_out, int* hit_out, unsigned long long* ctr, \
                                 hipStream_t stream); \
    \
    hipError_t vrt_launch_ray(const DevScene& S, const DevParams& P, bool bvh, bool twochild, const float o[3], \
                              const float d[3], int depth, float4* rgba_out, int* hit_out, unsigned long long* ctr, \
                              hipStream_t stream); \
    \
    hipError_t vrt_launch_wavefront(const DevScene& S, const DevCam& cam, const DevParams& P, bool bvh, \
                                    bool twochild, const WfWorkspace& W, float4* rgba, int* hit_id, \
                                    unsigned long long* ctr, hipStream_t stream); \

// Scenes of spheres and world-axis boxes without textures or the general materials take the
// megakernel's instantiation for them (k_mega<..., SIMPLE>).
static inline bool vrt_scene_is_simple(const DevScene& S)
{
    return S.n_tri == 0 && S.all_aab && !S.mat_tex && !S.has_matte && !S.has_transparent;
}

namespace vrt_plain { VRT_LAUNCHERS }
namespace vrt_fused { VRT_LAUNCHERS }
#undef VRT_LAUNCHERS

// LBVH build on the device.  prim arrays are those of DevScene; writes 4*(n_prims-1) float4
// binary nodes (with parent indices) to `nodes` and 8*(n_prims-1) float4 of 4-wide nodes to `wide`
// (slots of absorbed nodes stay unwritten).  Returns the binary tree depth in *depth_out.
// Synchronises the stream.
// builder: 0 PLOC clustering, 1 Karras' radix tree, 2 binned SAH built on the host, 3 binned SAH built on the device; parity_collapse: every second level
// becomes a wide node instead of the surface-area-guided collapse.
hipError_t vrt_build_lbvh(const DevScene& S, float4* nodes, float4* wide, int* depth_out, float* ms_out,
                          int builder, bool parity_collapse, hipStream_t stream);

// Dispatch order of the megakernel's tiles: order[k] = the tile with the k-th largest cost (ties by
// tile number).  iota = 0..n-1 (vrt_fill_iota), keys_tmp = n words of scratch, tmp = vrt_tile_sort_bytes(n) bytes.
size_t vrt_tile_sort_bytes(size_t n);
hipError_t vrt_fill_iota(int* iota, size_t n, hipStream_t stream);
hipError_t vrt_tile_sort(const unsigned* cost, unsigned* keys_tmp, const int* iota, int* order, size_t n, void* tmp,
                         size_t tmp_bytes, hipStream_t stream);

// process_color over n pixels: 3 floats in, 4 floats out.
hipError_t vrt_launch_tonemap(const float* rgb, float4* rgba, size_t n, hipStream_t stream);
