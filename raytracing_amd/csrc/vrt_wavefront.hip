// Wavefront pipeline (gfx950): one kernel per bounce over compacted struct-of-arrays ray
// queues in HBM, so that lanes whose path has ended do not ride along with live ones.
//
// Per anti-aliasing sample i:
//   k_wf_primary  one lane per pixel: folds the previous sample's radiance into the running
//                 sum, generates the camera ray, finds and shades the primary hit and appends
//                 the child ray(s) to queue 0.
//   k_wf_bounce   depth d = 1..max_depth: a persistent grid walks queue d-1, traces and shades,
//                 adds throughput * direct light to the pixel's accumulator and appends the
//                 survivors to queue d.  Appends are compacted per wave: ballot -> mbcnt prefix
//                 -> one atomic per wave.
//   k_wf_resolve  after the last sample: average, tone-map, write the framebuffer.
// Samples are traced one after the other so that the per-pixel sum is formed in sample order
// (RayTracerUnity.cs:580-592).
#define VRT_STACK_LANES 256
#include "vrt_trace.h"
#include "vrt_launch.h"

namespace {

__device__ __forceinline__ int local_to_global_row(const DevParams& P, int r)
{
    if (P.tile_count <= 1) return r;
    return ((r / P.tile_rows) * P.tile_count + P.tile_index) * P.tile_rows + (r % P.tile_rows);
}

__device__ __forceinline__ unsigned wave_sum(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void flush_counters(const Counters& C, unsigned long long* ctr, unsigned block_linear)
{
    __shared__ unsigned s_ctr[VRT_RAY_CLASSES];
    if (threadIdx.x < VRT_RAY_CLASSES) s_ctr[threadIdx.x] = 0;
    __syncthreads();
    unsigned v[VRT_RAY_CLASSES] = {C.precheck, C.primary, C.reflect, C.transmit, C.shadow, C.ao};
#pragma unroll
    for (int k = 0; k < VRT_RAY_CLASSES; ++k) {
        unsigned s = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&s_ctr[k], s);
    }
    __syncthreads();
    if (threadIdx.x < VRT_RAY_CLASSES && s_ctr[threadIdx.x])
        atomicAdd(&ctr[(block_linear % VRT_CTR_SLOTS) * VRT_CTR_STRIDE + threadIdx.x],
                  (unsigned long long)s_ctr[threadIdx.x]);
}

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

__device__ __forceinline__ void queue_store(const RayQueue& q, int slot, f3 o, f3 d, f3 T, int pix, int skip, int meta)
{
    q.ox[slot] = o.x; q.oy[slot] = o.y; q.oz[slot] = o.z;
    q.dx[slot] = d.x; q.dy[slot] = d.y; q.dz[slot] = d.z;
    q.tr[slot] = T.x; q.tg[slot] = T.y; q.tb[slot] = T.z;
    q.pix[slot] = pix; q.skip[slot] = skip; q.meta[slot] = meta;
}

// Shades hit `h` of ray (o, d) at `depth`, returns T*A and appends the children.
template <bool BVH, bool TWOCHILD>
__device__ __forceinline__ f3 shade_and_spawn(const DevScene& S, const DevParams& P, bool active, Hit h, f3 o, f3 d,
                                              f3 T, int depth, int pix, const RayQueue& out, int* out_count, int cap,
                                              int* overflow, Counters& C)
{
    f3 add = mk3(0.f, 0.f, 0.f);
    Bounce B; B.n_child = 0;
    Recorder R; R.segs = nullptr; R.cap = 0; R.n = 0; R.sample = 0;
    bool hit = active && h.prim >= 0;
    if (hit) {
        shade_hit<BVH, false>(S, P, h, o, d, depth, C, R, B);
        add = T * B.A;
        if (depth + 1 > P.max_depth) B.n_child = 0;
    }
    int s0 = wave_append(hit && B.n_child >= 1, out_count, cap, overflow);
    if (s0 >= 0) queue_store(out, s0, B.o, B.d0, T * B.w0, pix, B.prim, (depth + 1) | (VRT_RAY_REFLECT << 8));
    if (TWOCHILD) {
        int s1 = wave_append(hit && B.n_child == 2, out_count, cap, overflow);
        if (s1 >= 0) queue_store(out, s1, B.o, B.d1, T * B.w1, pix, B.prim, (depth + 1) | (VRT_RAY_TRANSMIT << 8));
    }
    return add;
}

template <bool BVH, bool TWOCHILD>
__global__ __launch_bounds__(256) void k_wf_primary(DevScene S, DevCam cam, DevParams P, WfWorkspace W, int sample,
                                                    int* __restrict__ hit_id, unsigned long long* ctr, int blocks_x,
                                                    int n_blocks)
{
    unsigned bid = blockIdx.x;
    {
        unsigned q = n_blocks / 8, r = n_blocks % 8, xcd = bid % 8, k = bid / 8;
        if (P.xcd_remap) bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    int bx = bid % blocks_x, by = bid / blocks_x;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int v = bx * 32 + wave * 8 + (lane & 7);
    int r = by * 8 + (lane >> 3);
    int h = local_to_global_row(P, r);
    bool active = v < cam.width && r < P.local_rows && h < cam.height;
    int lp = r * cam.width + v;
    Counters C = {0, 0, 0, 0, 0, 0};
    f3 O = mk3(cam.ox, cam.oy, cam.oz);
    f3 d = mk3(0.f, 0.f, 0.f);
    Hit first; first.prim = -1; first.t = INFINITY; first.aux = 0;
    f3 L = mk3(0.f, 0.f, 0.f);
    if (active) {
        if (sample == 0) W.sum[lp] = make_float4(0.f, 0.f, 0.f, 0.f);
        else {
            float4 s = W.sum[lp], a = W.lacc[lp];
            W.sum[lp] = make_float4(s.x + a.x, s.y + a.y, s.z + a.z, 0.f);
        }
        f3 raw = camera_raw_dir(cam, h, v);
        unsigned pixel = (unsigned)(h * cam.width + v);
        unsigned set = pixel_set_dev(P.seed, pixel, (unsigned)P.n_sets);
        float ox = 0.f, oy = 0.f;
        if (P.offsets) { float2 q = P.offsets[(size_t)set * P.spp + sample]; ox = q.x; oy = q.y; }
        d = normalize3(mk3(raw.x + ox, raw.y + oy, raw.z));
        if (!is_zero3(d)) {
            C.precheck++;
            first = intersect<BVH, false, true>(S, O, d, P.range, -1);
        }
        if (first.prim >= 0) C.primary++;
        else if (P.flags & VRT_FIX_PRIMARY_SKY) L = sky_color(d);
        if (sample == 0 && hit_id) hit_id[(size_t)(P.compact ? r : h) * cam.width + v] = first.prim;
    }
    f3 add = shade_and_spawn<BVH, TWOCHILD>(S, P, active, first, O, d, mk3(1.f, 1.f, 1.f), 0, lp, W.q[0], &W.counts[0],
                                            (int)W.cap, &W.counts[VRT_MAX_DEPTH + 2], C);
    if (active) {
        if (first.prim >= 0) L = add;
        W.lacc[lp] = make_float4(L.x, L.y, L.z, 0.f);
    }
    flush_counters(C, ctr, blockIdx.x);
}

template <bool BVH, bool TWOCHILD>
__global__ __launch_bounds__(256) void k_wf_bounce(DevScene S, DevParams P, WfWorkspace W, int depth,
                                                   unsigned long long* ctr)
{
    const RayQueue& in = W.q[(depth - 1) & 1];
    const RayQueue& out = W.q[depth & 1];
    int count = W.counts[depth - 1];
    if (count > (int)W.cap) count = (int)W.cap;
    Counters C = {0, 0, 0, 0, 0, 0};
    int stride = gridDim.x * 256;
    // wave-uniform trip count: every lane of a wave runs the same number of iterations so
    // that the ballots in wave_append see the whole wave
    for (int base = blockIdx.x * 256 + (threadIdx.x & ~63); base < count; base += stride) {
        int i = base + (threadIdx.x & 63);
        bool active = i < count;
        f3 o = mk3(0.f, 0.f, 0.f), d = o, T = o;
        int pix = 0, skip = -1, meta = 0;
        Hit h; h.prim = -1; h.t = INFINITY; h.aux = 0;
        if (active) {
            o = mk3(in.ox[i], in.oy[i], in.oz[i]);
            d = mk3(in.dx[i], in.dy[i], in.dz[i]);
            T = mk3(in.tr[i], in.tg[i], in.tb[i]);
            pix = in.pix[i]; skip = in.skip[i]; meta = in.meta[i];
            if (!is_zero3(d)) {
                if ((meta >> 8) == VRT_RAY_REFLECT) C.reflect++; else C.transmit++;
                h = intersect<BVH, false>(S, o, d, P.range, skip);
            }
        }
        f3 add = shade_and_spawn<BVH, TWOCHILD>(S, P, active, h, o, d, T, depth, pix, out, &W.counts[depth], (int)W.cap,
                                                &W.counts[VRT_MAX_DEPTH + 2], C);
        if (active) {
            if (h.prim < 0) add = T * sky_color(d);
            if (TWOCHILD) {
                float* a = (float*)&W.lacc[pix];
                atomicAdd(a, add.x); atomicAdd(a + 1, add.y); atomicAdd(a + 2, add.z);
            } else {
                float4 a = W.lacc[pix];
                W.lacc[pix] = make_float4(a.x + add.x, a.y + add.y, a.z + add.z, 0.f);
            }
        }
    }
    flush_counters(C, ctr, blockIdx.x);
}

__global__ __launch_bounds__(256) void k_wf_resolve(DevCam cam, DevParams P, WfWorkspace W, float4* __restrict__ rgba)
{
    int lp = blockIdx.x * 256 + threadIdx.x;
    if (lp >= P.local_rows * cam.width) return;
    int r = lp / cam.width, v = lp % cam.width;
    int h = local_to_global_row(P, r);
    if (h >= cam.height) return;
    float4 s = W.sum[lp], a = W.lacc[lp];
    f3 sum = mk3(s.x + a.x, s.y + a.y, s.z + a.z);
    f3 fin = sum / (float)P.spp;
    rgba[(size_t)(P.compact ? r : h) * cam.width + v] = process_color(fin);
}

template <bool BVH, bool TWOCHILD>
hipError_t run(const DevScene& S, const DevCam& cam, const DevParams& P, const WfWorkspace& W, float4* rgba, int* hit_id,
               unsigned long long* ctr, hipStream_t stream)
{
    int blocks_x = (cam.width + 31) / 32, blocks_y = (P.local_rows + 7) / 8;
    int n_blocks = blocks_x * blocks_y;
    if (n_blocks <= 0) return hipSuccess;
    size_t pixels = (size_t)P.local_rows * cam.width;
    int bounce_blocks = (int)((W.cap + 255) / 256);
    if (bounce_blocks > 2048) bounce_blocks = 2048;
    for (int s = 0; s < P.spp; ++s) {
        hipError_t e = hipMemsetAsync(W.counts, 0, (VRT_MAX_DEPTH + 2) * sizeof(int), stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_wf_primary<BVH, TWOCHILD>), dim3(n_blocks), dim3(256), 0, stream, S, cam, P, W, s, hit_id,
                           ctr, blocks_x, n_blocks);
        for (int d = 1; d <= P.max_depth; ++d)
            hipLaunchKernelGGL((k_wf_bounce<BVH, TWOCHILD>), dim3(bounce_blocks), dim3(256), 0, stream, S, P, W, d, ctr);
    }
    hipLaunchKernelGGL(k_wf_resolve, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, stream, cam, P, W, rgba);
    return hipGetLastError();
}

} // namespace

hipError_t vrt_launch_wavefront(const DevScene& S, const DevCam& cam, const DevParams& P, bool bvh, bool twochild,
                                const WfWorkspace& W, float4* rgba, int* hit_id, unsigned long long* ctr,
                                hipStream_t stream)
{
    // the overflow word is cleared once per frame, the per-depth counts once per sample
    hipError_t e = hipMemsetAsync(W.counts + VRT_MAX_DEPTH + 2, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    if (bvh) return twochild ? run<true, true>(S, cam, P, W, rgba, hit_id, ctr, stream)
                             : run<true, false>(S, cam, P, W, rgba, hit_id, ctr, stream);
    return twochild ? run<false, true>(S, cam, P, W, rgba, hit_id, ctr, stream)
                    : run<false, false>(S, cam, P, W, rgba, hit_id, ctr, stream);
}
