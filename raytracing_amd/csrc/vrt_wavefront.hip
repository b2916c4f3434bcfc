// Wavefront pipeline (gfx950): paths that go on after their first hit leave the per-pixel
// lanes and are traced bounce by bounce from compacted struct-of-arrays ray queues in HBM, so
// that lanes whose path has ended do not ride along with live ones.
//
// The frame is worked through in bands of rows sized to a fixed number of sample slots
// (slot = pixel of the band * spp + sample).  Per band:
//   k_wf_primary  lanes = pixels x samples like the megakernel: camera ray, primary hit and its
//                 shading (ambient-occlusion and shadow rays included); a path that ends here
//                 writes its radiance to its slot, one that goes on is appended to queue 0
//                 together with the radiance gathered so far.
//   k_wf_bounce   depth d = 1..max_depth: a persistent grid walks queue d-1, traces and shades,
//                 and either finishes the path (slot write) or appends it to queue d.  Appends
//                 are compacted per wave: ballot -> mbcnt prefix -> one atomic per wave, which
//                 also keeps the samples of one pixel next to each other in the queue.
//   k_wf_resolve  per pixel: sum of its slots in sample order (RayTracerUnity.cs:580-592),
//                 average, tone-map, framebuffer write.
// A bounce step runs as three passes when the workspace has room for deferred occlusion queries
// (k_wf_bounce_a -> k_wf_any -> k_wf_bounce_b): the ambient-occlusion and shadow rays of scattered
// secondary hits are cast by few lanes of a wave each (an AO ray exists for a quarter of the
// normals, a shadow ray for half), so instead of tracing them where they arise pass a appends
// them to a queue, k_wf_any traces that queue with full waves and pass b shades with the answers.
// The answers, and with them every bit of the result, are those of the one-pass form.
// With one child per hit a path owns its slot and carries its radiance in the queue entry, so
// the arithmetic and its order are those of the megakernel's trace_path.  With two children per
// hit (VRT_FIX_DIELECTRIC) the branches of a path share the slot and add into it atomically.
#define VRT_STACK_LANES 256
#ifndef VRT_WF_BOUNCE_BLOCKS
#define VRT_WF_BOUNCE_BLOCKS 4   // blocks of 256 lanes per CU the bounce kernel is compiled for (4 waves per SIMD)
#endif
#include "vrt_trace.h"
#include "vrt_launch.h"

namespace VRT_FORM {
namespace {

// Slot for this lane's ray in the output queue, or -1.  Must be called by the whole wave.
__device__ __forceinline__ int wave_append(bool want, int* counter, int cap, int* overflow)
{
    unsigned long long m = __ballot(want);
    if (m == 0ull) return -1;
    int n = __popcll(m);
    int leader = __ffsll((long long)m) - 1;
    int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, n);
    base = __shfl(base, leader, 64);
    int prefix = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    int slot = base + prefix;
    if (!want) return -1;
    if (slot >= cap) { *overflow = 1; return -1; }
    return slot;
}

// Appends of the persistent kernels: a wave takes VRT_WF_CHUNK consecutive entries of a queue
// from the global counter with one atomic and hands them out to its lanes from a cursor in LDS
// (`st`: base of the current chunk, entries used; owned by this wave), so that the counter, which
// every wave of the chip shares, is touched once per chunk instead of once per call (an atomic on one address costs
// about 9 ns chip-wide on MI355X, whoever issues it).  `m` =
// the lanes that append, all of them active.  Entries at the end of a wave's last chunk stay
// unwritten; queues mark those with a negative slot / target, and every consumer restores the
// mark after use, so a queue is all marks whenever a producer starts on it.
__device__ __forceinline__ int chunk_slot(unsigned long long m, volatile int* st, int* counter)
{
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1, n = __popcll(m);
    int base = 0, used = 0, next = 0;
    if (lane == leader) {
        base = st[0]; used = st[1];
        if (used + n > VRT_WF_CHUNK) {
            next = atomicAdd(counter, VRT_WF_CHUNK);
            st[0] = next; st[1] = used + n - VRT_WF_CHUNK;
        } else {
            st[1] = used + n;
        }
    }
    base = __shfl(base, leader, 64); used = __shfl(used, leader, 64); next = __shfl(next, leader, 64);
    const int p = used + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    return p < VRT_WF_CHUNK ? base + p : next + (p - VRT_WF_CHUNK);
}

// Chunk cursors of the block's four waves: [wave][0..1] ray queue, [wave][2..3] query queue.
__device__ __forceinline__ volatile int* chunk_state_init(int (*s)[4])
{
    if ((threadIdx.x & 63) == 0) {
        int* w = s[threadIdx.x >> 6];
        w[0] = 0; w[1] = VRT_WF_CHUNK; w[2] = 0; w[3] = VRT_WF_CHUNK;   // "chunk used up": the first append takes one
    }
    __syncthreads();
    return s[threadIdx.x >> 6];
}

// wave_append through the wave's chunk cursor.  Must be called by the whole wave.
__device__ __forceinline__ int wave_append_chunked(bool want, volatile int* st, int* counter, int cap, int* overflow)
{
    unsigned long long m = __ballot(want);
    if (m == 0ull) return -1;
    int slot = chunk_slot(m, st, counter);
    if (!want) return -1;
    if (slot < 0 || slot >= cap) { *overflow = 1; return -1; }
    return slot;
}

__device__ __forceinline__ void queue_store(const RayQueue& q, int at, f3 o, f3 d, f3 T, f3 L, int slot, int skip, int meta)
{
    q.ox[at] = o.x; q.oy[at] = o.y; q.oz[at] = o.z;
    q.dx[at] = d.x; q.dy[at] = d.y; q.dz[at] = d.z;
    q.tr[at] = T.x; q.tg[at] = T.y; q.tb[at] = T.z;
    q.lr[at] = L.x; q.lg[at] = L.y; q.lb[at] = L.z;
    q.pix[at] = slot; q.skip[at] = skip; q.meta[at] = meta;
}

// Occlusion-query provider of pass a: appends the query and answers "free"; the shading result
// of that pass is discarded.  A query that does not fit the queue is traced on the spot.
struct OccEmit {
    static constexpr bool counts = true;
    AnyQueue q; int cap; int* count; volatile int* st; unsigned* word; int target;
    template <bool BVH>
    __device__ __forceinline__ bool occluded(const DevScene& S, int k, f3 o, f3 d, float maxd, float tlim, int skip)
    {
        // the active lanes of the wave (any subset) take consecutive entries
        const int at = chunk_slot(__ballot(1), st, count);
        if (at >= 0 && at < cap) {
            q.ox[at] = o.x; q.oy[at] = o.y; q.oz[at] = o.z;
            q.dx[at] = d.x; q.dy[at] = d.y; q.dz[at] = d.z;
            q.tlim[at] = tlim; q.maxd[at] = maxd;
            q.skip[at] = skip; q.target[at] = target; q.bit[at] = k;
        } else if (intersect<BVH, true>(S, o, d, maxd, skip, tlim).prim >= 0) {
            atomicOr(word, 1u << k);
        }
        return false;
    }
};

// Provider of pass b: the answers gathered in the ray's flag word.
struct OccLookup {
    static constexpr bool counts = false;
    unsigned bits;
    template <bool BVH>
    __device__ __forceinline__ bool occluded(const DevScene&, int k, f3, f3, float, float, int) { return (bits >> k) & 1u; }
};

// The part of trace_path's loop body that follows a closest-hit query, for a path at `depth`
// whose radiance so far is L and throughput T: adds the hit's (or the sky's) term, then either
// finishes the path into its slot or appends it (them) to the next queue.  `shaded_ray` is false
// for a primary miss that contributes nothing.  The whole wave must call it.
template <bool BVH, bool TWOCHILD, class OCC>
__device__ __forceinline__ void shade_and_spawn(const DevScene& S, const DevParams& P, bool active, bool shaded_ray,
                                                Hit h, f3 o, f3 d, f3 T, f3 L, int depth, int slot,
                                                const WfWorkspace& W, Counters& C, OCC& occ, volatile int* st)
{
    Bounce B; B.n_child = 0;
    Recorder R; R.segs = nullptr; R.cap = 0; R.n = 0; R.sample = 0;
    const bool live = active && shaded_ray;
    const bool hit = live && h.prim >= 0;
    f3 add = mk3(0.f, 0.f, 0.f);
    if (hit) {
        shade_hit<BVH, false>(S, P, h, o, d, depth, C, R, B, occ);
        add = T * B.A;
        if (depth + 1 > P.max_depth) B.n_child = 0;
    } else if (live) {
        add = T * sky_color(d);
    }
    if (live) L = L + add;
    const RayQueue& out = W.q[depth & 1];
    int* overflow = &W.counts[VRT_MAX_DEPTH + 2];
    // st = the wave's chunk cursor in a persistent kernel, null in the per-tile primary kernel
    int s0 = st ? wave_append_chunked(hit && B.n_child >= 1, st, &W.counts[depth], (int)W.cap, overflow)
                : wave_append(hit && B.n_child >= 1, &W.counts[depth], (int)W.cap, overflow);
    if (s0 >= 0) queue_store(out, s0, B.o, B.d0, T * B.w0, L, slot, B.prim, (depth + 1) | (VRT_RAY_REFLECT << 8));
    if (TWOCHILD) {
        int s1 = st ? wave_append_chunked(hit && B.n_child == 2, st, &W.counts[depth], (int)W.cap, overflow)
                    : wave_append(hit && B.n_child == 2, &W.counts[depth], (int)W.cap, overflow);
        if (s1 >= 0) queue_store(out, s1, B.o, B.d1, T * B.w1, L, slot, B.prim, (depth + 1) | (VRT_RAY_TRANSMIT << 8));
        if (live) {   // branches share the slot: every term is added where it arises
            float* a = (float*)&W.lacc[slot];
            atomicAdd(a, add.x); atomicAdd(a + 1, add.y); atomicAdd(a + 2, add.z);
        }
    } else if (active && !(hit && B.n_child >= 1)) {
        W.lacc[slot] = make_float4(L.x, L.y, L.z, 0.f);   // the path ends here
    }
}

// Four waves per block (the traversal stack in LDS is sized for 256 lanes); each wave owns one
// tile, lanes = pixels x samples exactly as in k_mega (same tile shapes).
template <bool BVH, bool TWOCHILD>
__global__ __launch_bounds__(256, 4) void k_wf_primary(DevScene S, DevCam cam, DevParams P, WfWorkspace W, int row0,
                                                       int band_rows, int* __restrict__ hit_id, unsigned long long* ctr,
                                                       int blocks_x, int n_tiles)
{
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int bx = tile % blocks_x, by = tile / blocks_x;
    const int lane = threadIdx.x & 63;
    const int shift = P.sample_shift;
    const int n_sl = 1 << shift;
    const int pshift = 6 - shift;
    const int pw_shift = (pshift + 1) >> 1;
    const int pi = lane >> shift, si = lane & (n_sl - 1);
    const int v = (bx << pw_shift) + (pi & ((1 << pw_shift) - 1));
    const int rb = (by << (pshift - pw_shift)) + (pi >> pw_shift); // row within the band
    const int r = row0 + rb;
    const int h = local_to_global_row(P, r);
    const bool inside = tile < n_tiles && v < cam.width && rb < band_rows && r < P.local_rows && h < cam.height;
    if (!BVH) stage_prims(S);
    stage_materials(S);
    Counters C = {0, 0, 0, 0, 0, 0};
#if VRT_LDS_COUNTERS
    if (threadIdx.x < VRT_RAY_CLASSES) g_lds_ctr[threadIdx.x] = 0;
    __syncthreads();
#endif
    const f3 O = mk3(cam.ox, cam.oy, cam.oz);
    f3 raw = mk3(0.f, 0.f, 0.f);
    const float2* off = nullptr;
    if (inside) {
        raw = camera_raw_dir(cam, h, v);
        unsigned pixel = (unsigned)(h * cam.width + v);
        unsigned set = pixel_set_dev(P.seed, pixel, (unsigned)P.n_sets);
        off = P.offsets ? P.offsets + (size_t)set * P.spp : nullptr;
    }
    const int slot0 = (rb * cam.width + v) * P.spp;
    const bool sky = (P.flags & VRT_FIX_PRIMARY_SKY) != 0;
    for (int base = 0; base < P.spp; base += n_sl) {   // wave-uniform trip count
        const int i = base + si;
        const bool active = inside && i < P.spp;
        f3 d = mk3(0.f, 0.f, 0.f);
        Hit first; first.prim = -1; first.t = INFINITY; first.aux = 0;
        if (active) {
            float ox = 0.f, oy = 0.f;
            if (off) { float2 q = off[i]; ox = q.x; oy = q.y; }
            d = normalize3(mk3(raw.x + ox, raw.y + oy, raw.z));
            if (!is_zero3(d)) {
                VRT_COUNT(C, precheck, VRT_RAY_PRECHECK);
                first = intersect<BVH, false, true>(S, O, d, P.range, -1);
            }
            if (first.prim >= 0) VRT_COUNT(C, primary, VRT_RAY_PRIMARY);
            if (i == 0 && hit_id) hit_id[(size_t)(P.compact ? r : h) * cam.width + v] = first.prim;
        }
        // RayTracerUnity.cs:582-586: a primary miss adds nothing (the sky with VRT_FIX_PRIMARY_SKY)
        OccInline occ;
        shade_and_spawn<BVH, TWOCHILD>(S, P, active, first.prim >= 0 || sky, first, O, d, mk3(1.f, 1.f, 1.f),
                                       mk3(0.f, 0.f, 0.f), 0, slot0 + i, W, C, occ, nullptr);
    }
    flush_counters(C, ctr, blockIdx.x);
}

template <bool BVH, bool TWOCHILD>
__global__ __launch_bounds__(256, VRT_WF_BOUNCE_BLOCKS) void k_wf_bounce(DevScene S, DevParams P, WfWorkspace W, int depth,
                                                   unsigned long long* ctr)
{
    const RayQueue& in = W.q[(depth - 1) & 1];
    int count = W.counts[depth - 1];
    if (count > (int)W.cap) count = (int)W.cap;
    if (!BVH) stage_prims(S);
    stage_materials(S);
    Counters C = {0, 0, 0, 0, 0, 0};
#if VRT_LDS_COUNTERS
    if (threadIdx.x < VRT_RAY_CLASSES) g_lds_ctr[threadIdx.x] = 0;
    __syncthreads();
#endif
    __shared__ int s_chunk[4][4];
    volatile int* st = chunk_state_init(s_chunk);
    int stride = gridDim.x * 256;
    // wave-uniform trip count: every lane of a wave runs the same number of iterations so
    // that the ballots in wave_append see the whole wave
    for (int base = blockIdx.x * 256 + (threadIdx.x & ~63); base < count; base += stride) {
        int i = base + (threadIdx.x & 63);
        int slot = i < count ? in.pix[i] : -1;
        bool active = slot >= 0;   // negative slot: entry of a chunk nobody filled
        f3 o = mk3(0.f, 0.f, 0.f), d = o, T = o, L = o;
        int skip = -1, meta = 0;
        Hit h; h.prim = -1; h.t = INFINITY; h.aux = 0;
        if (active) {
            in.pix[i] = -1;   // consumed: the mark goes back
            o = mk3(in.ox[i], in.oy[i], in.oz[i]);
            d = mk3(in.dx[i], in.dy[i], in.dz[i]);
            T = mk3(in.tr[i], in.tg[i], in.tb[i]);
            L = mk3(in.lr[i], in.lg[i], in.lb[i]);
            skip = in.skip[i]; meta = in.meta[i];
            if (!is_zero3(d)) {
                if ((meta >> 8) == VRT_RAY_REFLECT) VRT_COUNT(C, reflect, VRT_RAY_REFLECT); else VRT_COUNT(C, transmit, VRT_RAY_TRANSMIT);
                h = intersect<BVH, false>(S, o, d, P.range, skip);
            }
        }
        OccInline occ;
        shade_and_spawn<BVH, TWOCHILD>(S, P, active, true, h, o, d, T, L, depth, slot, W, C, occ, st);
    }
    flush_counters(C, ctr, blockIdx.x);
}

// Pass a of a bounce step with deferred occlusion queries: closest hit of every ray of queue
// depth-1, kept in W.hits, and the occlusion queries of its shading appended to W.any.
// TRACED: the closest hits are already in W.hits (k_wf_trace ran before).
template <bool BVH, bool TRACED>
__global__ __launch_bounds__(256, VRT_WF_BOUNCE_BLOCKS) void k_wf_bounce_a(DevScene S, DevParams P, WfWorkspace W, int depth,
                                                                           unsigned long long* ctr)
{
    const RayQueue& in = W.q[(depth - 1) & 1];
    int count = W.counts[depth - 1];
    if (count > (int)W.cap) count = (int)W.cap;
    if (!BVH) stage_prims(S);
    stage_materials(S);
    Counters C = {0, 0, 0, 0, 0, 0};
#if VRT_LDS_COUNTERS
    if (threadIdx.x < VRT_RAY_CLASSES) g_lds_ctr[threadIdx.x] = 0;
    __syncthreads();
#endif
    __shared__ int s_chunk[4][4];
    volatile int* st = chunk_state_init(s_chunk);
    const int stride = gridDim.x * 256;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        if (in.pix[i] < 0) continue;   // entry of a chunk nobody filled
        const f3 o = mk3(in.ox[i], in.oy[i], in.oz[i]);
        const f3 d = mk3(in.dx[i], in.dy[i], in.dz[i]);
        const int skip = in.skip[i], meta = in.meta[i];
        Hit h; h.prim = -1; h.t = INFINITY; h.aux = 0;
        if (!is_zero3(d)) {
            if ((meta >> 8) == VRT_RAY_REFLECT) VRT_COUNT(C, reflect, VRT_RAY_REFLECT); else VRT_COUNT(C, transmit, VRT_RAY_TRANSMIT);
            if (!TRACED) h = intersect<BVH, false>(S, o, d, P.range, skip);
        }
        if (TRACED) {
            const int4 r = W.hits[i];
            h.t = __int_as_float(r.x); h.prim = r.y; h.aux = r.z;
        } else {
            W.hits[i] = make_int4(__float_as_int(h.t), h.prim, h.aux, 0);
        }
        if (h.prim >= 0) {
            OccEmit occ;
            occ.q = W.any; occ.cap = (int)W.any_cap; occ.count = &W.any_counts[depth];
            occ.word = (unsigned*)&W.hits[i].w; occ.target = i; occ.st = st + 2;
            Bounce B;
            Recorder R; R.segs = nullptr; R.cap = 0; R.n = 0; R.sample = 0;
            shade_hit<BVH, false>(S, P, h, o, d, depth, C, R, B, occ);
        }
    }
    flush_counters(C, ctr, blockIdx.x);
}

// The queries pass a queued at this depth, one per lane.
template <bool BVH>
__global__ __launch_bounds__(256, VRT_WF_BOUNCE_BLOCKS) void k_wf_any(DevScene S, WfWorkspace W, int depth)
{
    int count = W.any_counts[depth];
    if (count > (int)W.any_cap) count = (int)W.any_cap;
    if (!BVH) stage_prims(S);
    stage_materials(S);
    const AnyQueue& q = W.any;
    const int stride = gridDim.x * 256;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < count; j += stride) {
        const int target = q.target[j];
        if (target < 0 || target >= (int)W.cap) continue;   // entry of a chunk nobody filled
        q.target[j] = -1;                                   // consumed: the mark goes back
        const f3 o = mk3(q.ox[j], q.oy[j], q.oz[j]);
        const f3 d = mk3(q.dx[j], q.dy[j], q.dz[j]);
        if (intersect<BVH, true>(S, o, d, q.maxd[j], q.skip[j], q.tlim[j]).prim >= 0)
            atomicOr((unsigned*)&W.hits[target].w, 1u << (q.bit[j] & 31));
    }
}

// Traversal of the 4-wide tree for a whole queue with lanes refilled as they finish: a lane
// whose ray is done does not wait for the longest traversal of its wave, it takes the next
// entry of the queue once VRT_WF_REFILL lanes of the wave are idle.  Queue entries are claimed
// VRT_WF_TRACE_CHUNK at a time per wave (one atomic on the cursor all waves share), and handed
// to idle lanes in order.
//   ANY = false: rays of queue depth-1; closest hit (t, primitive, face, 0) to W.hits[i].
//   ANY = true:  the occlusion queries of this depth; a hit nearer than tlim sets the answer bit.
// The node and leaf steps are those of intersect_bvh4 (same box test, same primitive tests, same
// ranking of candidates by (t, id)), so hits and answers are the same whatever the order.
#ifndef VRT_WF_REFILL
#define VRT_WF_REFILL 16
#endif
#define VRT_WF_TRACE_CHUNK 256
template <bool ANY>
__global__ __launch_bounds__(256, VRT_WF_BOUNCE_BLOCKS) void k_wf_trace(DevScene S, DevParams P, WfWorkspace W, int depth)
{
    unsigned long long* my_lds = g_lds_stack + threadIdx.x;
    const RayQueue& in = W.q[(depth - 1) & 1];
    const AnyQueue& q = W.any;
    int count = ANY ? W.any_counts[depth] : W.counts[depth - 1];
    const int cap = ANY ? (int)W.any_cap : (int)W.cap;
    if (count > cap) count = cap;
    int* cursor = W.cursors + (ANY ? VRT_MAX_DEPTH + 2 : 0) + depth;
    const int lane = threadIdx.x & 63;
    int next = 0, end = 0;            // the wave's claimed entries [next, end); wave-uniform
    bool exhausted = count <= 0;      // wave-uniform: the cursor has passed the end of the queue
    bool first = true;                // wave-uniform: no chunk taken yet
    bool have = false;                // this lane is in the middle of a traversal
    Trav T;
    trav_begin(T, mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 1.f), 0.f, 0.f, -1);
    int entry = 0, bit = 0, steps_left = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!have);
        if (!exhausted && __popcll(idle) >= VRT_WF_REFILL) {
            if (next >= end) {
                // The first chunk of a wave is the one with its own number, taken without an atomic
                // (atomics on one address are served one after the other chip-wide, about 9 ns each:
                // a claim per wave at the start would cost a short queue more than its traversal);
                // the cursor hands out what lies beyond those.
                int b;
                if (first) {
                    b = (int)((blockIdx.x * 4 + (threadIdx.x >> 6)) * VRT_WF_TRACE_CHUNK);
                    first = false;
                } else {
                    b = 0;
                    if (lane == 0) b = (int)(gridDim.x * 4 * VRT_WF_TRACE_CHUNK) + atomicAdd(cursor, VRT_WF_TRACE_CHUNK);
                    b = __builtin_amdgcn_readfirstlane(b);
                }
                if (b < 0 || b >= count) exhausted = true;
                else { next = b; end = b + VRT_WF_TRACE_CHUNK < count ? b + VRT_WF_TRACE_CHUNK : count; }
            }
            if (!exhausted) {
                const int n_idle = __popcll(idle);
                const int take = n_idle < end - next ? n_idle : end - next;
                const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
                if (!have && r < take) {
                    entry = next + r;
                    bool valid;
                    f3 o = mk3(0.f, 0.f, 0.f), d = o;
                    float maxd = 0.f, tlim = 0.f;
                    int skip = -1;
                    if (ANY) {
                        const int target = q.target[entry];
                        valid = target >= 0 && target < (int)W.cap;   // else: entry of a chunk nobody filled
                        if (valid) {
                            q.target[entry] = -1;                     // consumed: the mark goes back
                            o = mk3(q.ox[entry], q.oy[entry], q.oz[entry]);
                            d = mk3(q.dx[entry], q.dy[entry], q.dz[entry]);
                            maxd = q.maxd[entry]; tlim = q.tlim[entry]; skip = q.skip[entry]; bit = q.bit[entry] & 31;
                            entry = target;
                        }
                    } else {
                        valid = in.pix[entry] >= 0;
                        if (valid) {
                            o = mk3(in.ox[entry], in.oy[entry], in.oz[entry]);
                            d = mk3(in.dx[entry], in.dy[entry], in.dz[entry]);
                            maxd = P.range; tlim = INFINITY; skip = in.skip[entry];
                        }
                    }
                    if (valid) {
                        trav_begin(T, o, d, maxd, tlim, skip);
                        bool found = false;
                        have = !is_zero3(d);
                        if (have) {
                            found = trav_big(S, T, ANY);
                            steps_left = 2 * S.n_prims + 8;
                            if (found) have = false;
                        }
                        if (!have) {
                            if (ANY) { if (found) atomicOr((unsigned*)&W.hits[entry].w, 1u << bit); }
                            else W.hits[entry] = make_int4(__float_as_int(T.best.t), T.best.prim, T.best.aux, 0);
                        }
                    }
                }
                next += take;
            }
        }
        if (!__any(have)) {
            if (exhausted) break;
            continue;
        }
        if (have) {
            // one step of intersect_bvh4's loop
            bool pop, finished = false, found = false;
            if (T.node >= 0) pop = trav_inner(S, T, my_lds, ANY);
            else { found = finished = trav_leaf(S, T, ANY); pop = true; }
            if (!finished && pop && !trav_pop(T, my_lds)) finished = true;
            if (--steps_left <= 0) finished = true;   // only with a corrupt tree
            if (finished) {
                if (!found && T.ovf) found = trav_stackless(S, T, ANY);
                if (ANY) { if (found) atomicOr((unsigned*)&W.hits[entry].w, 1u << bit); }
                else W.hits[entry] = make_int4(__float_as_int(T.best.t), T.best.prim, T.best.aux, 0);
                have = false;
            }
        }
    }
}

// Pass b: shading of every ray of queue depth-1 with the answers of k_wf_any, then as k_wf_bounce.
template <bool BVH, bool TWOCHILD>
__global__ __launch_bounds__(256, VRT_WF_BOUNCE_BLOCKS) void k_wf_bounce_b(DevScene S, DevParams P, WfWorkspace W, int depth)
{
    const RayQueue& in = W.q[(depth - 1) & 1];
    int count = W.counts[depth - 1];
    if (count > (int)W.cap) count = (int)W.cap;
    Counters C = {0, 0, 0, 0, 0, 0};   // rays were counted in pass a
    stage_materials(S);
    __shared__ int s_chunk[4][4];
    volatile int* st = chunk_state_init(s_chunk);
    const int stride = gridDim.x * 256;
    // wave-uniform trip count, as in k_wf_bounce
    for (int base = blockIdx.x * 256 + (threadIdx.x & ~63); base < count; base += stride) {
        const int i = base + (threadIdx.x & 63);
        const int slot = i < count ? in.pix[i] : -1;
        const bool active = slot >= 0;   // negative slot: entry of a chunk nobody filled
        f3 o = mk3(0.f, 0.f, 0.f), d = o, T = o, L = o;
        Hit h; h.prim = -1; h.t = INFINITY; h.aux = 0;
        OccLookup occ; occ.bits = 0u;
        if (active) {
            in.pix[i] = -1;   // consumed: the mark goes back
            o = mk3(in.ox[i], in.oy[i], in.oz[i]);
            d = mk3(in.dx[i], in.dy[i], in.dz[i]);
            T = mk3(in.tr[i], in.tg[i], in.tb[i]);
            L = mk3(in.lr[i], in.lg[i], in.lb[i]);
            const int4 r = W.hits[i];
            h.t = __int_as_float(r.x); h.prim = r.y; h.aux = r.z; occ.bits = (unsigned)r.w;
        }
        shade_and_spawn<BVH, TWOCHILD>(S, P, active, true, h, o, d, T, L, depth, slot, W, C, occ, st);
    }
}

// Lanes = pixels x samples with the megakernel's split (P.sample_shift): every lane reads its
// own slot, so a wave reads consecutive float4s, and lane 0 of each pixel's group adds the
// group's radiances in sample order through wave shuffles (RayTracerUnity.cs:580-592).
__global__ __launch_bounds__(256) void k_wf_resolve(DevCam cam, DevParams P, WfWorkspace W, int row0, int band_rows,
                                                    float4* __restrict__ rgba)
{
    const int shift = P.sample_shift, n_sl = 1 << shift;
    const int lane = threadIdx.x & 63, si = lane & (n_sl - 1);
    const long long lp = ((long long)blockIdx.x * 256 + threadIdx.x) >> shift;   // pixel of the band
    const bool inside = lp < (long long)band_rows * cam.width;
    const int rb = inside ? (int)(lp / cam.width) : 0, v = inside ? (int)(lp % cam.width) : 0;
    const int r = row0 + rb;
    const int h = local_to_global_row(P, r);
    const bool live = inside && r < P.local_rows && h < cam.height;
    const float4* a = W.lacc + (size_t)(inside ? lp : 0) * P.spp;
    f3 sum = mk3(0.f, 0.f, 0.f);
    for (int base = 0; base < P.spp; base += n_sl) {   // wave-uniform trip count
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && base + si < P.spp) q = a[base + si];
        for (int k = 0; k < n_sl; ++k) {
            const int src = (lane & ~(n_sl - 1)) + k;
            const float x = __shfl(q.x, src, 64), y = __shfl(q.y, src, 64), z = __shfl(q.z, src, 64);
            if (base + k < P.spp) sum = sum + mk3(x, y, z);
        }
    }
    if (live && si == 0) {
        f3 fin = sum / (float)P.spp;
        const size_t idx = (size_t)(P.compact ? r : h) * cam.width + v;
        store_pixel(P, rgba, idx, fin);
    }
}

template <bool BVH, bool TWOCHILD>
hipError_t run(const DevScene& S, const DevCam& cam, const DevParams& P, const WfWorkspace& W, float4* rgba, int* hit_id,
               unsigned long long* ctr, hipStream_t stream)
{
    if (P.local_rows <= 0) return hipSuccess;
    const int pshift = 6 - P.sample_shift, pw_shift = (pshift + 1) >> 1, ph_shift = pshift - pw_shift;
    const int blocks_x = (cam.width + (1 << pw_shift) - 1) >> pw_shift;
    // rows per band: as many as the slot budget holds, a multiple of the tile height
    const size_t per_row = (size_t)cam.width * P.spp;
    long long band = (long long)(W.slots / per_row);
    band = (band >> ph_shift) << ph_shift;
    if (band < (1 << ph_shift)) return hipErrorInvalidValue; // the workspace holds one tile row at least
    for (int row0 = 0; row0 < P.local_rows; row0 += (int)band) {
        const int rows = P.local_rows - row0 < band ? P.local_rows - row0 : (int)band;
        const int blocks_y = (rows + (1 << ph_shift) - 1) >> ph_shift;
        hipError_t e = hipMemsetAsync(W.counts, 0, (VRT_MAX_DEPTH + 2) * sizeof(int), stream);
        if (e != hipSuccess) return e;
        if (TWOCHILD) {
            e = hipMemsetAsync(W.lacc, 0, (size_t)rows * per_row * sizeof(float4), stream);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((k_wf_primary<BVH, TWOCHILD>), dim3((blocks_x * blocks_y + 3) / 4), dim3(256), 0, stream, S, cam,
                           P, W, row0, rows, hit_id, ctr, blocks_x, blocks_x * blocks_y);
        size_t live = (size_t)rows * per_row * (TWOCHILD ? 2 : 1);
        int bounce_blocks = (int)((live + 255) / 256 < VRT_WF_MAX_BLOCKS ? (live + 255) / 256 : VRT_WF_MAX_BLOCKS);
        if (W.hits) {
            e = hipMemsetAsync(W.any_counts, 0, 3 * (VRT_MAX_DEPTH + 2) * sizeof(int), stream);   // counts and both cursors
            if (e != hipSuccess) return e;
        }
        for (int d = 1; d <= P.max_depth; ++d) {
            if (W.hits && BVH && W.refill) {
                hipLaunchKernelGGL((k_wf_trace<false>), dim3(bounce_blocks), dim3(256), 0, stream, S, P, W, d);
                hipLaunchKernelGGL((k_wf_bounce_a<BVH, true>), dim3(bounce_blocks), dim3(256), 0, stream, S, P, W, d, ctr);
                hipLaunchKernelGGL((k_wf_trace<true>), dim3(bounce_blocks), dim3(256), 0, stream, S, P, W, d);
                hipLaunchKernelGGL((k_wf_bounce_b<BVH, TWOCHILD>), dim3(bounce_blocks), dim3(256), 0, stream, S, P, W, d);
            } else if (W.hits) {
                hipLaunchKernelGGL((k_wf_bounce_a<BVH, false>), dim3(bounce_blocks), dim3(256), 0, stream, S, P, W, d, ctr);
                hipLaunchKernelGGL((k_wf_any<BVH>), dim3(bounce_blocks), dim3(256), 0, stream, S, W, d);
                hipLaunchKernelGGL((k_wf_bounce_b<BVH, TWOCHILD>), dim3(bounce_blocks), dim3(256), 0, stream, S, P, W, d);
            } else {
                hipLaunchKernelGGL((k_wf_bounce<BVH, TWOCHILD>), dim3(bounce_blocks), dim3(256), 0, stream, S, P, W, d, ctr);
            }
        }
        hipLaunchKernelGGL(k_wf_resolve, dim3((unsigned)((((size_t)rows * cam.width << P.sample_shift) + 255) / 256)), dim3(256), 0, stream,
                           cam, P, W, row0, rows, rgba);
    }
    return hipGetLastError();
}

} // namespace

hipError_t vrt_launch_wavefront(const DevScene& S, const DevCam& cam, const DevParams& P, bool bvh, bool twochild,
                                const WfWorkspace& W, float4* rgba, int* hit_id, unsigned long long* ctr,
                                hipStream_t stream)
{
    // the overflow word is cleared once per frame, the per-depth counts once per band
    hipError_t e = hipMemsetAsync(W.counts + VRT_MAX_DEPTH + 2, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    if (bvh) return twochild ? run<true, true>(S, cam, P, W, rgba, hit_id, ctr, stream)
                             : run<true, false>(S, cam, P, W, rgba, hit_id, ctr, stream);
#if VRT_SLAB_FMA
    return hipErrorInvalidValue;   // the fused form only differs where a tree is walked
#else
    return twochild ? run<false, true>(S, cam, P, W, rgba, hit_id, ctr, stream)
                    : run<false, false>(S, cam, P, W, rgba, hit_id, ctr, stream);
#endif
}

} // namespace VRT_FORM
