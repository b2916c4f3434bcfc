// Megakernel with the bounces compacted inside the block (gfx950), for scenes traced through the tree.
//
// In k_mega a lane follows its sample through all bounces, and the wave lasts as long as its
// deepest path: on the 10k-sphere frame 30 % of the samples go on after the first hit, 6 % after
// the second, 1 % after the third, and the vector ALUs run the later bounces with 11 %, 5 % and
// 2 % of their lanes.  Here a block is four waves (four of k_mega's tiles).  Camera ray, first hit
// and its shading run as in k_mega, lanes = pixels x samples.  After every bounce the paths that
// go on are appended to a queue in LDS (origin, direction, throughput, owner, primitive left) and
// handed out again densely: ray j of the queue goes to lane j of the block, so the rays of a
// depth fill the first waves and the other waves wait at the barrier without taking issue
// slots.  The radiance of a sample is accumulated in LDS under its owner's lane number, term by
// term in path order, so the arithmetic and its order per sample are those of trace_path and the
// frame is k_mega's bit for bit whichever lane traces which ray.
//
// LDS per block: the traversal stacks (16 entries x 256 lanes x 8 B = 32 KiB), which double as
// the ray queue between traversals (11 x 256 x 4 B), 3 KiB of radiance and the counters: four
// blocks per CU, 4 waves per SIMD as in k_mega.
#define VRT_STACK_LANES 256
#include "vrt_trace.h"
#include "vrt_launch.h"

namespace VRT_FORM {
namespace {

__shared__ float s_lacc[3][256];              // radiance of the block's samples by owner lane
__shared__ int s_qcount[VRT_MAX_DEPTH + 2];   // rays queued per depth

// Queue slot of this lane's ray or -1; the whole wave calls it.
__device__ __forceinline__ int block_append(bool want, int* counter)
{
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return -1;
    const int leader = __ffsll((long long)m) - 1, lane = threadIdx.x & 63;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader, 64);
    const int prefix = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    return want ? base + prefix : -1;
}

template <bool SPLIT>
__global__ __launch_bounds__(256, 4) void k_mega_blk(DevScene S, DevCam cam, DevParams P, float4* __restrict__ rgba,
                                                     int* __restrict__ hit_id, unsigned long long* ctr, int blocks_x,
                                                     int n_tiles)
{
    constexpr bool BVH = true;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tile = blockIdx.x * 4 + (tid >> 6);
    const int bx = tile % blocks_x, by = tile / blocks_x;
    const int shift = SPLIT ? P.sample_shift : 0;
    const int n_sl = 1 << shift;
    const int pshift = 6 - shift, pw_shift = (pshift + 1) >> 1;
    const int pi = lane >> shift, si = lane & (n_sl - 1);
    const int v = (bx << pw_shift) + (pi & ((1 << pw_shift) - 1));
    const int r = (by << (pshift - pw_shift)) + (pi >> pw_shift);
    const int h = local_to_global_row(P, r);
    Counters C = {0, 0, 0, 0, 0, 0};
    if (tid < VRT_RAY_CLASSES) g_lds_ctr[tid] = 0;
    stage_materials(S);
    Recorder R; R.segs = nullptr; R.cap = 0; R.n = 0; R.sample = 0;
    const bool inside = tile < n_tiles && v < cam.width && r < P.local_rows && h < cam.height;
    const f3 O = mk3(cam.ox, cam.oy, cam.oz);
    f3 raw = mk3(0.f, 0.f, 0.f);
    const float2* off = nullptr;
    if (inside) {
        raw = camera_raw_dir(cam, h, v);
        unsigned pixel = (unsigned)(h * cam.width + v);
        unsigned set = pixel_set_dev(P.seed, pixel, (unsigned)P.n_sets);
        off = P.offsets ? P.offsets + (size_t)set * P.spp : nullptr;
    }
    float* q = (float*)g_lds_stack;   // the queue, valid between two barriers while no lane traverses
    f3 sum = mk3(0.f, 0.f, 0.f);
    int id = -1;
    for (int base = 0; base < P.spp; base += n_sl) {   // block-uniform trip count
        if (tid < VRT_MAX_DEPTH + 2) s_qcount[tid] = 0;
        const int i = base + si;
        const bool active = inside && i < P.spp;
        // ---- camera ray, first hit and its shading (trace_sample_dev)
        bool alive = false;
        f3 ro = O, rd = mk3(0.f, 0.f, 0.f), rT = mk3(1.f, 1.f, 1.f);
        int rskip = -1, owner = tid;
        f3 L = mk3(0.f, 0.f, 0.f);
        if (active) {
            float ox = 0.f, oy = 0.f;
            if (off) { float2 w = off[i]; ox = w.x; oy = w.y; }
            const f3 d = normalize3(mk3(raw.x + ox, raw.y + oy, raw.z));
            Hit first; first.prim = -1; first.t = INFINITY; first.aux = 0;
            if (!is_zero3(d)) {
                VRT_COUNT(C, precheck, VRT_RAY_PRECHECK);
                first = intersect<BVH, false, true>(S, O, d, P.range, -1, INFINITY, P.packet != 0);
            }
            if (i == 0) id = first.prim;
            if (first.prim >= 0) {
                VRT_COUNT(C, primary, VRT_RAY_PRIMARY);
                Bounce B;
                shade_hit<BVH, false>(S, P, first, O, d, 0, C, R, B);
                L = L + rT * B.A;
                if (B.n_child > 0 && 1 <= P.max_depth) {
                    alive = true; ro = B.o; rd = B.d0; rT = rT * B.w0; rskip = B.prim;
                }
            } else if (P.flags & VRT_FIX_PRIMARY_SKY) {
                L = sky_color(d);
            }
        }
        s_lacc[0][tid] = L.x; s_lacc[1][tid] = L.y; s_lacc[2][tid] = L.z;
        // ---- bounces, compacted over the block
        for (int depth = 1; depth <= P.max_depth; ++depth) {
            __syncthreads();   // every traversal of the previous step is over: the stacks become the queue
            const int at = block_append(alive, &s_qcount[depth]);
            if (at >= 0) {
                q[0 * 256 + at] = ro.x; q[1 * 256 + at] = ro.y; q[2 * 256 + at] = ro.z;
                q[3 * 256 + at] = rd.x; q[4 * 256 + at] = rd.y; q[5 * 256 + at] = rd.z;
                q[6 * 256 + at] = rT.x; q[7 * 256 + at] = rT.y; q[8 * 256 + at] = rT.z;
                q[9 * 256 + at] = __int_as_float(owner); q[10 * 256 + at] = __int_as_float(rskip);
            }
            __syncthreads();
            const int n = s_qcount[depth];
            if (n == 0) break;   // block-uniform
            alive = tid < n;
            if (alive) {
                ro = mk3(q[0 * 256 + tid], q[1 * 256 + tid], q[2 * 256 + tid]);
                rd = mk3(q[3 * 256 + tid], q[4 * 256 + tid], q[5 * 256 + tid]);
                rT = mk3(q[6 * 256 + tid], q[7 * 256 + tid], q[8 * 256 + tid]);
                owner = __float_as_int(q[9 * 256 + tid]); rskip = __float_as_int(q[10 * 256 + tid]);
            }
            __syncthreads();   // the queue has been read: the stacks are stacks again
            if (alive) {
                // one turn of trace_path's loop for the ray (ro, rd) at `depth`
                alive = false;
                Hit hh; hh.prim = -1; hh.t = INFINITY; hh.aux = 0;
                if (!is_zero3(rd)) {
                    VRT_COUNT(C, reflect, VRT_RAY_REFLECT);
                    hh = intersect<BVH, false>(S, ro, rd, P.range, rskip);
                }
                f3 acc = mk3(s_lacc[0][owner], s_lacc[1][owner], s_lacc[2][owner]);
                if (hh.prim < 0) {
                    acc = acc + rT * sky_color(rd);
                } else {
                    Bounce B;
                    shade_hit<BVH, false>(S, P, hh, ro, rd, depth, C, R, B);
                    acc = acc + rT * B.A;
                    if (B.n_child > 0 && depth + 1 <= P.max_depth) {
                        alive = true; ro = B.o; rd = B.d0; rT = rT * B.w0; rskip = B.prim;
                    }
                }
                s_lacc[0][owner] = acc.x; s_lacc[1][owner] = acc.y; s_lacc[2][owner] = acc.z;
            }
        }
        __syncthreads();
        // ---- per-pixel sum in sample order (RayTracerUnity.cs:580-592), as in k_mega
        f3 c = mk3(0.f, 0.f, 0.f);
        if (active) c = mk3(s_lacc[0][tid], s_lacc[1][tid], s_lacc[2][tid]);
        if (shift == 0) {
            sum = sum + c;
        } else {
            const int first_lane = lane & ~(n_sl - 1);
            const int n_here = min(n_sl, P.spp - base);
            for (int k = 0; k < n_here; ++k) {
                f3 w = mk3(__shfl(c.x, first_lane + k, 64), __shfl(c.y, first_lane + k, 64),
                           __shfl(c.z, first_lane + k, 64));
                sum = sum + w;
            }
        }
        __syncthreads();   // s_lacc and s_qcount are rewritten by the next round
    }
    if (inside && si == 0) {
        f3 fin = sum / (float)P.spp;
        size_t idx = (size_t)(P.compact ? r : h) * cam.width + v; // RayTracerUnity.cs:442
        store_pixel(P, rgba, idx, fin);
        if (hit_id) hit_id[idx] = id;
    }
    flush_counters(C, ctr, blockIdx.x);
}

} // namespace

hipError_t vrt_launch_mega_blk(const DevScene& S, const DevCam& cam, const DevParams& P, float4* rgba, int* hit_id,
                               unsigned long long* ctr, hipStream_t stream)
{
    const int shift = P.sample_shift;
    const int pshift = 6 - shift, pw_shift = (pshift + 1) >> 1;
    const int blocks_x = (cam.width + (1 << pw_shift) - 1) >> pw_shift;
    const int blocks_y = (P.local_rows + (1 << (pshift - pw_shift)) - 1) >> (pshift - pw_shift);
    const int n_tiles = blocks_x * blocks_y;
    if (n_tiles <= 0) return hipSuccess;
    dim3 grid((n_tiles + 3) / 4), block(256);
    if (shift == 0)
        hipLaunchKernelGGL((k_mega_blk<false>), grid, block, 0, stream, S, cam, P, rgba, hit_id, ctr, blocks_x, n_tiles);
    else
        hipLaunchKernelGGL((k_mega_blk<true>), grid, block, 0, stream, S, cam, P, rgba, hit_id, ctr, blocks_x, n_tiles);
    return hipGetLastError();
}

} // namespace VRT_FORM
