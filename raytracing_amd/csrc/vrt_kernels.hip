// Render kernels of libvrt (gfx950) and their launchers.
//
// k_mega      one lane = one pixel through all samples and bounces (single launch).
// k_pixels    the same per-pixel routine with path recording, for a short list of pixels.
// k_wf_*      wavefront pipeline: per-bounce kernels over compacted SoA ray queues.
#include "vrt_trace.h"
#include "vrt_launch.h"

// ------------------------------------------------------------------ megakernel

// Block = one wave.  The 64 lanes are split as (64 >> P.sample_shift) pixels x (1 << P.sample_shift)
// anti-aliasing samples: with one sample per pixel a wave owns an 8x8 pixel tile (eight full
// 128-byte framebuffer lines per store); with many samples per pixel the samples of a pixel sit
// in neighbouring lanes, where they follow almost the same path through the tree and end at the
// same depth, instead of being looped over by one lane while its neighbours wait for the
// deepest path of each round.  The per-pixel sum is still formed in sample order
// (RayTracerUnity.cs:580-592): the lanes' radiances are read back one by one with a wave shuffle.
// One wave per block measured 6 % faster than four (finer-grained tail).
// Every variant is compiled for 4 waves per SIMD (128 VGPRs).  The tree variant needs them for
// the packed slab arithmetic of the 4-wide traversal, whose operands live in register pairs: at
// 5 waves (96 VGPRs) it spills and loses 24 %, at 3 waves it loses 9 % to latency.  The general
// brute-force variants fit 106-124 registers and stay at 4 waves; see VRT_SIMPLE_WAVES for the others.
// P.xcd_remap (off by default, see vrt_api.hip) renumbers blocks so that the 8 blocks dealt
// round-robin to the 8 XCDs come from 8 separate bands of the image.
namespace VRT_FORM {

#define VRT_MEGA_THREADS 64
// VRT_KEEP_LDS=1: see KEEP_LDS in k_mega.
#ifndef VRT_KEEP_LDS
#define VRT_KEEP_LDS 1
#endif
#ifndef VRT_BVH_WAVES
#define VRT_BVH_WAVES 4
#endif
// The instantiations for simple brute-force scenes run at 6 waves per SIMD (80 VGPRs, six of them
// spilled) and keep the pixel's sum between rounds in LDS like the tree kernels (VRT_KEEP_LDS_SIMPLE):
// a wave of this kernel waits for the scalar loads of the primitives and for its own dependent chains,
// and every further wave fills some of the gaps, until the scalar unit, which issues one instruction
// per SIMD every four clocks and has 0.8 instructions to issue per vector instruction here, is the
// busier of the two.  Teaching scene at 1080p: 4 waves 0.1111 ms, 5 waves 0.1009 ms, 5 waves with the
// sum in LDS 0.0972 ms, 6 waves with the sum in LDS 0.0921 ms; 7 and 8 waves (with radiance and
// throughput of the path in LDS as well, which costs 1.5 % by itself) 0.0919 / 0.0921 ms.
#ifndef VRT_SIMPLE_WAVES
#define VRT_SIMPLE_WAVES 6
#endif
#ifndef VRT_KEEP_LDS_SIMPLE
#define VRT_KEEP_LDS_SIMPLE 1
#endif
// SIMPLE (brute-force scenes only) is the instantiation for scenes of spheres and world-axis boxes
// without triangles or textures, such as the shipped teaching scene: with those facts known at
// compile time the unused loops and branches, and the registers they keep alive, disappear.
// ORD (brute-force instantiations for simple scenes only; the tree kernels decide at run time):
// 0 tiles in block order, 1 tile looked up in P.tile_order, 2 as 1 when there is an order, and the
// wave reports its cost.  Separate instantiations because the frame kernel of the teaching scene
// loses 5 % to the two scalars the run-time form keeps alive.
template <bool BVH, bool TWOCHILD, bool SPLIT, bool SIMPLE, int ORD = 0>
__global__ __launch_bounds__(VRT_MEGA_THREADS, BVH ? VRT_BVH_WAVES : (SIMPLE ? VRT_SIMPLE_WAVES : 4)) void k_mega(DevScene S_in, DevCam cam, DevParams P, float4* __restrict__ rgba,
                                              int* __restrict__ hit_id, unsigned long long* ctr,
                                              int blocks_x, int n_blocks)
{
    DevScene S = S_in;
    // scenes with a matte or transparent material run the TWOCHILD instantiations (vrt_api.hip); P is
    // what carries the fact, because a modified copy of S, whose `big` array is indexed in a loop,
    // would live in scratch memory
    if (!TWOCHILD) P.general = 0;
    if (SIMPLE) { S.n_tri = 0; S.all_aab = 1; S.mat_tex = nullptr; S.n_big = 0; S.tri = nullptr; } // S.box stays: face normals
    unsigned bid = blockIdx.x;
    {   // bijective XCD remap: label = bid % 8 selects a contiguous band of block ids
        unsigned q = n_blocks / 8, r = n_blocks % 8, xcd = bid % 8, k = bid / 8;
        if (P.xcd_remap) bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    // Tree scenes: tiles are dispatched most expensive first when the host has an order from an
    // earlier frame (vrt_api.hip, tile_sort), and a wave reports the clock ticks it ran when the
    // host asks for the costs of this frame.  (Not in the brute-force kernels: their frame gains
    // 2 % from the order and loses 5 % to the two more live scalars.)
    long long t_start = 0ll;
    unsigned tile = bid;
    if constexpr (BVH || ORD == 2) {
        if (P.tile_cost) t_start = clock64();
        if (P.tile_order) bid = (unsigned)P.tile_order[bid];
        tile = bid;
    } else if constexpr (ORD == 1) {
        bid = (unsigned)P.tile_order[bid];
        tile = bid;
    }
    const int bx = bid % blocks_x, by = bid / blocks_x;
    // SPLIT = false is the instantiation for one lane per pixel (sample_shift 0)
    const int shift = !SPLIT ? 0 : P.sample_shift;
    const int n_sl = 1 << shift;                       // sample lanes per pixel
    const int pshift = 6 - shift;                      // log2(pixels per wave)
    const int pw_shift = (pshift + 1) >> 1;            // tile width 2^ceil(pshift/2), height 2^floor(pshift/2)
    // Tree kernels and the kernels for simple scenes (KEEP_LDS): what a lane knows about its pixel is
    // worked out again where it is needed (per round, and for the store at the end) from a lane number
    // the compiler cannot see through, and the pixel's sum waits in LDS while a round is traced.  In
    // the tree kernels values that live across the tracing code would otherwise be spilled to scratch
    // memory by every wave, 6 KB each, which at 512 resident waves per XCD is most of that XCD's L2
    // and ends up as gigabytes of write-backs per frame; in the kernels for simple scenes the
    // registers saved are what lets a sixth wave per SIMD in.
    constexpr bool KEEP_LDS = (BVH && VRT_KEEP_LDS) || (SIMPLE && VRT_KEEP_LDS_SIMPLE);
    int lane = threadIdx.x;
#define VRT_PIXEL_OF_LANE()                                                        \
    DevParams Pl = P;   /* divisors hidden too: the invariant halves of divisions by them would be hoisted and kept */ \
    if constexpr (KEEP_LDS) asm volatile("" : "+v"(lane), "+s"(Pl.tile_rows), "+s"(Pl.tile_count), "+s"(Pl.n_sets)); \
    const int pi = lane >> shift, si = lane & (n_sl - 1);                          \
    const int v = (bx << pw_shift) + (pi & ((1 << pw_shift) - 1));                 \
    const int r = (by << (pshift - pw_shift)) + (pi >> pw_shift);                  \
    const int h = local_to_global_row(Pl, r);                                      \
    const bool inside = v < cam.width && r < P.local_rows && h < cam.height;
    if (!BVH) stage_prims(S);
    stage_materials(S);
    Counters C = {0, 0, 0, 0, 0, 0};
#if VRT_LDS_COUNTERS
    if (threadIdx.x < VRT_RAY_CLASSES) g_lds_ctr[threadIdx.x] = 0;
    __syncthreads();
#endif
    Recorder R; R.segs = nullptr; R.cap = 0; R.n = 0; R.sample = 0;
    __shared__ float s_keep[KEEP_LDS ? 4 : 1][VRT_MEGA_THREADS];   // the pixel's sum and hit id between rounds
    // see VRT_USE_SM in vrt_trace.h
    constexpr bool USE_SM = VRT_USE_SM == 1 || (VRT_USE_SM == 2 && BVH);
    // paths a thinned-out wave finishes together in the state machine (VRT_TAIL_MAX in vrt_trace.h)
    constexpr bool TAIL = BVH && !TWOCHILD && !USE_SM && VRT_TAIL_MAX > 0;
    __shared__ LdsScene s_scene;
    if constexpr (USE_SM || TAIL) stage_scene(S, &s_scene);
    f3 sum = mk3(0.f, 0.f, 0.f);
    int id = -1;
    if constexpr (KEEP_LDS) {
        s_keep[0][threadIdx.x] = 0.f; s_keep[1][threadIdx.x] = 0.f; s_keep[2][threadIdx.x] = 0.f;
        s_keep[3][threadIdx.x] = __int_as_float(-1);
    }
    for (int base = 0; base < P.spp; base += n_sl) {   // wave-uniform trip count
        VRT_PIXEL_OF_LANE()
        const int i = base + si;
        f3 c = mk3(0.f, 0.f, 0.f);
        // per round, not hoisted: these would otherwise stay in registers through every traversal
        f3 raw = mk3(0.f, 0.f, 0.f);
        const float2* off = nullptr;
        if (inside && !USE_SM) {
            DevCam camr = cam;
            // (the image size hidden as well: the loop-invariant halves of the two divisions in
            // camera_raw_dir would be hoisted out of the loop and kept in vector registers)
            if constexpr (KEEP_LDS) asm volatile("" : "+s"(camr.width), "+s"(camr.height));
            raw = camera_raw_dir(camr, h, v);
            unsigned pixel = (unsigned)(h * cam.width + v);
            unsigned set = pixel_set_dev(P.seed, pixel, (unsigned)Pl.n_sets);
            off = P.offsets ? P.offsets + (size_t)set * P.spp : nullptr;
        }
        if constexpr (USE_SM) {
            // one sample per lane and round; lanes without a sample ride along
            const bool act = inside && i < P.spp;
            int fp;
            c = trace_samples_sm<BVH, false, TWOCHILD>(S, cam, P, &s_scene, act ? h : 0, act ? v : 0, act ? i : 0, act ? i + 1 : 0, fp, C, R);
            if (act && i == 0) id = fp;
        } else if (BVH) {
            // the whole wave goes in, so that the camera rays can be traced cooperatively
            const bool live = inside && i < P.spp;
            int fp;
            if constexpr (TAIL) {
                PathState ps;
                ps.pending = false;
                c = trace_sample_dev<BVH, false, TWOCHILD>(S, cam, P, raw, off, i, fp, C, R, live, true, &ps);
                if (__any(ps.pending)) {   // wave-uniform: all 64 lanes go in, those with a path to finish take it up
                    int unused;
                    const f3 t = trace_samples_sm<BVH, false, TWOCHILD>(S, cam, P, &s_scene, 0, 0, 0, ps.pending ? 1 : 0, unused, C, R, &ps);
                    if (ps.pending) c = t;
                }
            } else {
                c = trace_sample_dev<BVH, false, TWOCHILD>(S, cam, P, raw, off, i, fp, C, R, live, true);
            }
            if (live && i == 0) id = fp;
        } else if (inside && i < P.spp) {
            int fp;
            c = trace_sample_dev<BVH, false, TWOCHILD>(S, cam, P, raw, off, i, fp, C, R);
            if (i == 0) id = fp;
        }
        if constexpr (KEEP_LDS) {   // the sum so far comes back from LDS (the barrier keeps the load after the tracing code)
            asm volatile("" ::: "memory");
            sum = mk3(s_keep[0][threadIdx.x], s_keep[1][threadIdx.x], s_keep[2][threadIdx.x]);
        }
        if (shift == 0) {
            sum = sum + c;
        } else {
            // ordered sum: every lane of the pixel adds samples base, base+1, ... in turn
            const int first_lane = lane & ~(n_sl - 1);
            const int n_here = min(n_sl, P.spp - base);
            for (int k = 0; k < n_here; ++k) {
                f3 q = mk3(__shfl(c.x, first_lane + k, 64), __shfl(c.y, first_lane + k, 64),
                           __shfl(c.z, first_lane + k, 64));
                sum = sum + q;
            }
        }
        if constexpr (KEEP_LDS) {
            s_keep[0][threadIdx.x] = sum.x; s_keep[1][threadIdx.x] = sum.y; s_keep[2][threadIdx.x] = sum.z;
            if (id != -1) s_keep[3][threadIdx.x] = __int_as_float(id);   // set in the round of sample 0, by that sample's lane
            sum = mk3(0.f, 0.f, 0.f); id = -1;                           // nothing is carried in registers
            asm volatile("" ::: "memory");
        }
    }
    {
        VRT_PIXEL_OF_LANE()
        if constexpr (KEEP_LDS) {
            sum = mk3(s_keep[0][threadIdx.x], s_keep[1][threadIdx.x], s_keep[2][threadIdx.x]);
            id = __float_as_int(s_keep[3][threadIdx.x]);
        }
        if (inside && si == 0) {
            // x / 1 is x: one sample per pixel needs no IEEE divisions (a wave-uniform branch)
            f3 fin = P.spp == 1 ? sum : sum / (float)P.spp;
            size_t idx = (size_t)(P.compact ? r : h) * cam.width + v; // RayTracerUnity.cs:442
            store_pixel(P, rgba, idx, fin);
            if (hit_id) hit_id[idx] = id;
        }
    }
#undef VRT_PIXEL_OF_LANE
    flush_counters(C, ctr, blockIdx.x);
    if constexpr (BVH || ORD == 2) {
        if (P.tile_cost && threadIdx.x == 0) P.tile_cost[tile] = (unsigned)(clock64() - t_start);
    }
}

// ------------------------------------------------------------------ recorded pixels

// This kernel always runs the single-loop state-machine tracer (trace_pixel_sm) while k_mega
// defaults to the straight-line one, so every test run checks both against the oracle and
// against each other (vrt_trace_pixel must reproduce the frame's pixel bit for bit).

// One wave per listed pixel, one lane per anti-aliasing sample (rounds of 64 when there are more).
// The segments of a pixel must come out in (sample, cast) order in one array of `cap` entries, and a
// lane only knows where its sample's segments start once the samples before it have been counted:
// every round is therefore traced twice, first counting (nothing stored), then, after a prefix sum
// over the lanes in sample order, storing from each lane's own offset.  Twice the rays, but 64
// samples at a time instead of one after the other.  Rays are counted once (second pass only), and
// the pixel's colour is summed in sample order, so colour, hit id, counts and segments are those of
// the serial loop bit for bit.
template <bool BVH, bool TWOCHILD>
__global__ __launch_bounds__(64) void k_pixels(DevScene S, DevCam cam, DevParams P, const int* __restrict__ pixels,
                                               int n_pixels, int sample_sel, vrt_path_seg* __restrict__ segs,
                                               int cap, int* __restrict__ seg_counts, float4* __restrict__ rgba_out,
                                               int* __restrict__ hit_out, unsigned long long* ctr)
{
    const int pi = blockIdx.x;   // wave-uniform; the grid has exactly n_pixels blocks
    const int lane = threadIdx.x;
    if (!TWOCHILD) P.general = 0;
    if (!BVH) stage_prims(S);
    stage_materials(S);
    Counters C = {0, 0, 0, 0, 0, 0};
#if VRT_LDS_COUNTERS
    if (threadIdx.x < VRT_RAY_CLASSES) g_lds_ctr[threadIdx.x] = 0;
    __syncthreads();
#endif
    __shared__ LdsScene s_scene;
    stage_scene(S, &s_scene);
    const int pix = pi < n_pixels ? pixels[pi] : 0;
    const int h = pix / cam.width, v = pix % cam.width;
    const int s0 = sample_sel < 0 ? 0 : sample_sel, s1 = sample_sel < 0 ? P.spp : sample_sel + 1;
    f3 sum = mk3(0.f, 0.f, 0.f);
    int id = -1, n_before = 0;   // segments of the samples of earlier rounds (wave-uniform)
    for (int base = s0; base < s1; base += 64) {   // wave-uniform trip count
        const int i = base + lane;
        const bool act = i < s1;
        int n_mine = 0, fp = -1;
        f3 c = mk3(0.f, 0.f, 0.f);
        for (int pass = 0; pass < 2; ++pass) {
            // the counting pass's rays do not count: the counters are put back after it
            unsigned keep = 0;
            const Counters keepC = C;
#if VRT_LDS_COUNTERS
            if (pass == 0) { __syncthreads(); if (threadIdx.x < VRT_RAY_CLASSES) keep = g_lds_ctr[threadIdx.x]; }
#endif
            // where this lane's segments start: after those of the lanes before it, in sample order
            int off = n_before;
            if (pass == 1) {
                int incl = n_mine;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
                off = n_before + incl - n_mine;
                n_before += __shfl(incl, 63, 64);
            }
            Recorder R;
            R.n = 0; R.sample = 0;
            const int room = cap > off ? cap - off : 0;
            R.segs = (pass == 1 && act && room > 0) ? segs + (size_t)pi * cap + off : nullptr;
            R.cap = (pass == 1 && act) ? room : 0;
            c = trace_samples_sm<BVH, true, TWOCHILD>(S, cam, P, &s_scene, h, v, act ? i : 0, act ? i + 1 : 0, fp, C, R);
            n_mine = act ? R.n : 0;
            if (pass == 0) {
                C = keepC;
#if VRT_LDS_COUNTERS
                __syncthreads();
                if (threadIdx.x < VRT_RAY_CLASSES) g_lds_ctr[threadIdx.x] = keep;
                __syncthreads();
#endif
            }
        }
        if (base == 0 && lane == 0) id = fp;   // primary hit of sample 0
        // the pixel's sum in sample order (RayTracerUnity.cs:580-592)
        const int n_here = min(64, s1 - base);
        for (int k = 0; k < n_here; ++k)
            sum = sum + mk3(__shfl(c.x, k, 64), __shfl(c.y, k, 64), __shfl(c.z, k, 64));
    }
    if (lane == 0 && pi < n_pixels) {
        float4 col = process_color(sum / (float)(s1 - s0));
        seg_counts[pi] = n_before;
        if (rgba_out) rgba_out[pi] = col;
        if (hit_out) hit_out[pi] = id;
    }
    if (ctr) flush_counters(C, ctr, blockIdx.x);
}

// ------------------------------------------------------------------ single ray

// AbstractTracer.TraceRay(ray, depth) for one caller-supplied ray (Tracer/WhittedTracer.cs:33-104).
template <bool BVH, bool TWOCHILD>
__global__ __launch_bounds__(64) void k_ray(DevScene S, DevParams P, float ox, float oy, float oz, float dx, float dy,
                                            float dz, int depth, float4* __restrict__ rgba_out,
                                            int* __restrict__ hit_out, unsigned long long* ctr)
{
    if (!TWOCHILD) P.general = 0;
    if (!BVH) stage_prims(S);
    stage_materials(S);
    Counters C = {0, 0, 0, 0, 0, 0};
#if VRT_LDS_COUNTERS
    if (threadIdx.x < VRT_RAY_CLASSES) g_lds_ctr[threadIdx.x] = 0;
    __syncthreads();
#endif
    if (threadIdx.x == 0) {
        Recorder R; R.segs = nullptr; R.cap = 0; R.n = 0; R.sample = 0;
        f3 o = mk3(ox, oy, oz), d = normalize3(mk3(dx, dy, dz));
        f3 L = mk3(0.f, 0.f, 0.f);
        int id = -1;
        if (depth <= P.max_depth) {
            Hit first; first.prim = -1; first.t = INFINITY; first.aux = 0;
            if (!is_zero3(d)) {
                if (depth == 0) VRT_COUNT(C, primary, VRT_RAY_PRIMARY); else VRT_COUNT(C, reflect, VRT_RAY_REFLECT);
                first = intersect<BVH, false>(S, o, d, P.range, -1);
            }
            id = first.prim;
            L = first.prim >= 0 ? trace_path<BVH, false, TWOCHILD>(S, P, o, d, first, C, R, depth) : sky_color(d);
        }
        rgba_out[0] = make_float4(L.x, L.y, L.z, 1.f);
        hit_out[0] = id;
    }
    flush_counters(C, ctr, 0);
}

// ------------------------------------------------------------------ launchers

static inline int div_up(int a, int b) { return (a + b - 1) / b; }

hipError_t vrt_launch_mega(const DevScene& S, const DevCam& cam, const DevParams& P, bool bvh, bool twochild,
                           float4* rgba, int* hit_id, unsigned long long* ctr, hipStream_t stream)
{
    const int shift = P.sample_shift;
    const int pshift = 6 - shift, pw_shift = (pshift + 1) >> 1;
    int blocks_x = div_up(cam.width, 1 << pw_shift), blocks_y = div_up(P.local_rows, 1 << (pshift - pw_shift));
    int n_blocks = blocks_x * blocks_y;
    if (n_blocks <= 0) return hipSuccess;
    dim3 grid(n_blocks), block(VRT_MEGA_THREADS);
#if !VRT_SLAB_FMA
    const bool simple = !bvh && vrt_scene_is_simple(S);
    const int ord = P.tile_cost ? 2 : (P.tile_order ? 1 : 0);
#endif
#define VRT_GO3(B, T, SP, SI) hipLaunchKernelGGL((k_mega<B, T, SP, SI>), grid, block, 0, stream, S, cam, P, rgba, hit_id, ctr, blocks_x, n_blocks)
#define VRT_GO(B, T) do { if (shift == 0) VRT_GO3(B, T, false, false); else VRT_GO3(B, T, true, false); } while (0)
#define VRT_GOS(T) do { if (shift == 0) VRT_GO3(false, T, false, true); else VRT_GO3(false, T, true, true); } while (0)
#define VRT_GO5(SP, O) hipLaunchKernelGGL((k_mega<false, false, SP, true, O>), grid, block, 0, stream, S, cam, P, rgba, hit_id, ctr, blocks_x, n_blocks)
#define VRT_GOSO(O) do { if (shift == 0) VRT_GO5(false, O); else VRT_GO5(true, O); } while (0)
    if (bvh) { if (twochild) VRT_GO(true, true); else VRT_GO(true, false); }
#if VRT_SLAB_FMA
    else return hipErrorInvalidValue;   // the fused form only differs where a tree is walked
#else
    else if (simple && !twochild && ord == 1) VRT_GOSO(1);
    else if (simple && !twochild && ord == 2) VRT_GOSO(2);
    else if (simple) { if (twochild) VRT_GOS(true); else VRT_GOS(false); }
    else     { if (twochild) VRT_GO(false, true); else VRT_GO(false, false); }
#endif
#undef VRT_GOSO
#undef VRT_GO5
#undef VRT_GOS
#undef VRT_GO3
#undef VRT_GO
    return hipGetLastError();
}

hipError_t vrt_launch_pixels(const DevScene& S, const DevCam& cam, const DevParams& P, bool bvh, bool twochild,
                             const int* pixels, int n_pixels, int sample_sel, vrt_path_seg* segs, int cap,
                             int* seg_counts, float4* rgba_out, int* hit_out, unsigned long long* ctr,
                             hipStream_t stream)
{
    if (n_pixels <= 0) return hipSuccess;
    dim3 grid(n_pixels), block(64);   // one wave per pixel, its lanes are the pixel's samples
#define VRT_GO(B, T) hipLaunchKernelGGL((k_pixels<B, T>), grid, block, 0, stream, S, cam, P, pixels, n_pixels, sample_sel, segs, cap, seg_counts, rgba_out, hit_out, ctr)
    if (bvh) { if (twochild) VRT_GO(true, true); else VRT_GO(true, false); }
#if VRT_SLAB_FMA
    else return hipErrorInvalidValue;
#else
    else     { if (twochild) VRT_GO(false, true); else VRT_GO(false, false); }
#endif
#undef VRT_GO
    return hipGetLastError();
}

hipError_t vrt_launch_ray(const DevScene& S, const DevParams& P, bool bvh, bool twochild, const float o[3],
                          const float d[3], int depth, float4* rgba_out, int* hit_out, unsigned long long* ctr,
                          hipStream_t stream)
{
    dim3 grid(1), block(64);
#define VRT_GO(B, T) hipLaunchKernelGGL((k_ray<B, T>), grid, block, 0, stream, S, P, o[0], o[1], o[2], d[0], d[1], d[2], depth, rgba_out, hit_out, ctr)
    if (bvh) { if (twochild) VRT_GO(true, true); else VRT_GO(true, false); }
#if VRT_SLAB_FMA
    else return hipErrorInvalidValue;
#else
    else     { if (twochild) VRT_GO(false, true); else VRT_GO(false, false); }
#endif
#undef VRT_GO
    return hipGetLastError();
}

} // namespace VRT_FORM

#if !VRT_SLAB_FMA
// ------------------------------------------------------------------ tone mapping after a gather

__global__ __launch_bounds__(256) void k_tonemap(const float* __restrict__ rgb, float4* __restrict__ rgba, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    rgba[i] = process_color(mk3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]));
}

hipError_t vrt_launch_tonemap(const float* rgb, float4* rgba, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tonemap, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rgb, rgba, n);
    return hipGetLastError();
}
#endif
