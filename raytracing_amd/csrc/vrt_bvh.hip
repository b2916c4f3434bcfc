// LBVH construction on the device (gfx950): Morton codes of primitive centroids -> radix sort
// -> Karras' binary radix tree -> bottom-up fit of child boxes -> collapse of every second
// level into 4-wide nodes.  The outputs are the 64-byte "two child boxes per node" binary
// layout and the 128-byte 4-wide layout described in vrt_device.h.
//
// Closest-hit results do not depend on the tree (candidates are ranked by (t, id)), so the
// builder only has to be conservative: primitive boxes are padded by a relative 1e-4.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>
#include <rocprim/rocprim.hpp>

#include "vrt_launch.h"

namespace {

__device__ __forceinline__ unsigned enc_f(float f)
{
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float dec_f(unsigned e)
{
    unsigned b = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
    float f;
#ifdef __HIP_DEVICE_COMPILE__
    f = __uint_as_float(b);
#else
    memcpy(&f, &b, 4);
#endif
    return f;
}

__device__ __forceinline__ void prim_box(const DevScene& S, int id, float lo[3], float hi[3])
{
    if (id < S.n_sph) {
        float4 s = S.sph[id];
        lo[0] = s.x - s.w; lo[1] = s.y - s.w; lo[2] = s.z - s.w;
        hi[0] = s.x + s.w; hi[1] = s.y + s.w; hi[2] = s.z + s.w;
    } else if (id < S.n_sph + S.n_box) {
        const float4* b = S.box + 4 * (id - S.n_sph);
        float4 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
        float c[3] = {b0.x, b0.y, b0.z};
        float a0[3] = {b1.x, b1.y, b1.z}, a1[3] = {b2.x, b2.y, b2.z}, a2[3] = {b3.x, b3.y, b3.z};
        for (int k = 0; k < 3; ++k) {
            float e = fabsf(a0[k]) * b0.w + fabsf(a1[k]) * b1.w + fabsf(a2[k]) * b2.w;
            lo[k] = c[k] - e; hi[k] = c[k] + e;
        }
    } else {
        const float4* t = S.tri + 3 * (size_t)(id - S.n_sph - S.n_box);
        float4 v0 = t[0], e1 = t[1], e2 = t[2];
        float p0[3] = {v0.x, v0.y, v0.z};
        float p1[3] = {v0.x + e1.x, v0.y + e1.y, v0.z + e1.z};
        float p2[3] = {v0.x + e2.x, v0.y + e2.y, v0.z + e2.z};
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(p0[k], fminf(p1[k], p2[k]));
            hi[k] = fmaxf(p0[k], fmaxf(p1[k], p2[k]));
        }
    }
    for (int b = 0; b < S.n_big; ++b)
        if (S.big[b] == id)  // tested ahead of the traversal: only a point in the tree
            for (int k = 0; k < 3; ++k) lo[k] = hi[k] = 0.5f * (lo[k] + hi[k]);
    for (int k = 0; k < 3; ++k) {
        float e = 1e-4f * (1.f + fmaxf(fabsf(lo[k]), fabsf(hi[k])));
        lo[k] -= e; hi[k] += e;
    }
}

// Primitive boxes and the bounds of their centroids (6 ordered-uint atomics per block).
__global__ __launch_bounds__(256) void k_prim_bounds(DevScene S, float4* __restrict__ pbox, unsigned* __restrict__ cb)
{
    __shared__ unsigned s_mn[3], s_mx[3];
    if (threadIdx.x < 3) { s_mn[threadIdx.x] = 0xffffffffu; s_mx[threadIdx.x] = 0u; }
    __syncthreads();
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < S.n_prims) {
        float lo[3], hi[3];
        prim_box(S, i, lo, hi);
        pbox[2 * (size_t)i] = make_float4(lo[0], lo[1], lo[2], 0.f);
        pbox[2 * (size_t)i + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
        for (int k = 0; k < 3; ++k) {
            unsigned e = enc_f(0.5f * (lo[k] + hi[k]));
            atomicMin(&s_mn[k], e);
            atomicMax(&s_mx[k], e);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        atomicMin(&cb[threadIdx.x], s_mn[threadIdx.x]);
        atomicMax(&cb[3 + threadIdx.x], s_mx[threadIdx.x]);
    }
}

__device__ __forceinline__ unsigned long long spread21(unsigned long long v)
{
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

__global__ __launch_bounds__(256) void k_morton(int n, const float4* __restrict__ pbox, const unsigned* __restrict__ cb,
                                                unsigned long long* __restrict__ keys, unsigned* __restrict__ vals)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 lo = pbox[2 * (size_t)i], hi = pbox[2 * (size_t)i + 1];
    float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
    unsigned long long code = 0;
    for (int k = 0; k < 3; ++k) {
        float mn = dec_f(cb[k]), mx = dec_f(cb[3 + k]);
        float ext = mx - mn;
        float u = ext > 0.f ? (c[k] - mn) / ext : 0.f;
        u = fminf(fmaxf(u, 0.f), 1.f);
        unsigned long long q = (unsigned long long)(u * 2097151.f);
        code |= spread21(q) << (2 - k);
    }
    keys[i] = code;
    vals[i] = (unsigned)i;
}

__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// Karras 2012, one lane per internal node.  Children are stored as: >= 0 internal node index,
// < 0 leaf with ~(sorted position).
__global__ __launch_bounds__(256) void k_hierarchy(int n, const unsigned long long* __restrict__ keys,
                                                   int2* __restrict__ child, int* __restrict__ parent,
                                                   int* __restrict__ leaf_parent)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    int c0, c1;
    if (lo == gamma) { c0 = ~gamma; leaf_parent[gamma] = i; } else { c0 = gamma; parent[gamma] = i; }
    if (hi == gamma + 1) { c1 = ~(gamma + 1); leaf_parent[gamma + 1] = i; } else { c1 = gamma + 1; parent[gamma + 1] = i; }
    child[i] = make_int2(c0, c1);
    if (i == 0) parent[0] = -1;
}

// Bottom-up fit, one lane per leaf.  The first lane to reach a node parks its box and leaves;
// the second reads it and carries the union upward.  The parked box crosses CUs/XCDs inside
// one launch, so it is published with an agent-scope release before the arrival counter and
// read after an agent-scope acquire.
__global__ __launch_bounds__(256) void k_fit(int n, const unsigned* __restrict__ vals, const float4* __restrict__ pbox,
                                             const int2* __restrict__ child, const int* __restrict__ parent,
                                             const int* __restrict__ leaf_parent, unsigned* __restrict__ arrived,
                                             float4* __restrict__ parked, float4* __restrict__ nodes)
{
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    int prim = (int)vals[p];
    float4 lo = pbox[2 * (size_t)prim], hi = pbox[2 * (size_t)prim + 1];
    int me = ~p;            // code under which the parent knows this subtree
    int out_code = ~prim;   // code written to the final node (leaves refer to primitive ids)
    int node = leaf_parent[p];
    for (int guard = 0; node >= 0 && guard < 4096; ++guard) {
        int2 ch = child[node];
        int side = ch.x == me ? 0 : 1;
        float4* slot = parked + 4 * (size_t)node + 2 * side;
        lo.w = __int_as_float(out_code);
        slot[0] = lo; slot[1] = hi;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned old = atomicAdd(&arrived[node], 1u);
        if (old == 0u) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float4* other = parked + 4 * (size_t)node + 2 * (1 - side);
        float4 olo = other[0], ohi = other[1];
        float4 lo0 = side == 0 ? lo : olo, hi0 = side == 0 ? hi : ohi;
        float4 lo1 = side == 0 ? olo : lo, hi1 = side == 0 ? ohi : hi;
        float4* nd = nodes + 4 * (size_t)node;
        nd[0] = make_float4(lo0.x, lo0.y, lo0.z, hi0.x);
        nd[1] = make_float4(hi0.y, hi0.z, lo1.x, lo1.y);
        nd[2] = make_float4(lo1.z, hi1.x, hi1.y, hi1.z);
        nd[3] = make_float4(lo0.w, lo1.w, 0.f, 0.f);
        lo = make_float4(fminf(lo.x, olo.x), fminf(lo.y, olo.y), fminf(lo.z, olo.z), 0.f);
        hi = make_float4(fmaxf(hi.x, ohi.x), fmaxf(hi.y, ohi.y), fmaxf(hi.z, ohi.z), 0.f);
        me = node; out_code = node;
        node = parent[node];
    }
}

__global__ __launch_bounds__(256) void k_depth(int n, const int* __restrict__ parent, const int* __restrict__ leaf_parent,
                                               int* __restrict__ depth)
{
    int p = blockIdx.x * 256 + threadIdx.x;
    int d = 0;
    if (p < n) {
        int node = leaf_parent[p];
        while (node >= 0 && d < 4096) { ++d; node = parent[node]; }
    }
    for (int o = 32; o > 0; o >>= 1) d = max(d, __shfl_xor(d, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(depth, d);
}

// 4-wide collapse, one lane per binary inner node.  Nodes at even depth survive and adopt
// their grandchildren (a leaf child is adopted as it is); nodes at odd depth are absorbed.  A
// surviving node keeps its binary index, so child codes need no translation.
__global__ __launch_bounds__(256) void k_collapse4(int n_inner, const float4* __restrict__ nodes,
                                                   const int* __restrict__ parent, float4* __restrict__ wide)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_inner) return;
    int depth = 0;
    for (int p = parent[i]; p >= 0 && depth < 4096; p = parent[p]) ++depth;
    if (depth & 1) return;
    float lo[4][3], hi[4][3];
    int code[4];
    int m = 0;
    const float4* nd = nodes + 4 * (size_t)i;
    float4 a0 = nd[0], a1 = nd[1], a2 = nd[2], a3 = nd[3];
    const float blo[2][3] = {{a0.x, a0.y, a0.z}, {a1.z, a1.w, a2.x}};
    const float bhi[2][3] = {{a0.w, a1.x, a1.y}, {a2.y, a2.z, a2.w}};
    const int bc[2] = {__float_as_int(a3.x), __float_as_int(a3.y)};
    for (int side = 0; side < 2; ++side) {
        if (bc[side] < 0) {
            for (int k = 0; k < 3; ++k) { lo[m][k] = blo[side][k]; hi[m][k] = bhi[side][k]; }
            code[m++] = bc[side];
        } else {
            const float4* cd = nodes + 4 * (size_t)bc[side];
            float4 c0 = cd[0], c1 = cd[1], c2 = cd[2], c3 = cd[3];
            lo[m][0] = c0.x; lo[m][1] = c0.y; lo[m][2] = c0.z; hi[m][0] = c0.w; hi[m][1] = c1.x; hi[m][2] = c1.y;
            code[m++] = __float_as_int(c3.x);
            lo[m][0] = c1.z; lo[m][1] = c1.w; lo[m][2] = c2.x; hi[m][0] = c2.y; hi[m][1] = c2.z; hi[m][2] = c2.w;
            code[m++] = __float_as_int(c3.y);
        }
    }
    for (; m < 4; ++m) {
        // lo = hi = +inf: (inf - o) * inv is +inf or -inf on both planes, which fails the slab
        // test for every ray, so the traversal needs no separate check for unused slots
        for (int k = 0; k < 3; ++k) { lo[m][k] = INFINITY; hi[m][k] = INFINITY; }
        code[m] = VRT_BVH4_EMPTY;
    }
    float4* w = wide + 8 * (size_t)i;
    w[0] = make_float4(lo[0][0], lo[1][0], lo[2][0], lo[3][0]);
    w[1] = make_float4(lo[0][1], lo[1][1], lo[2][1], lo[3][1]);
    w[2] = make_float4(lo[0][2], lo[1][2], lo[2][2], lo[3][2]);
    w[3] = make_float4(hi[0][0], hi[1][0], hi[2][0], hi[3][0]);
    w[4] = make_float4(hi[0][1], hi[1][1], hi[2][1], hi[3][1]);
    w[5] = make_float4(hi[0][2], hi[1][2], hi[2][2], hi[3][2]);
    w[6] = make_float4(__int_as_float(code[0]), __int_as_float(code[1]), __int_as_float(code[2]),
                       __int_as_float(code[3]));
    w[7] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------ PLOC
// Parallel locally-ordered clustering (Meister & Bittner 2017) on the Morton-sorted primitives:
// every round each cluster looks at its VRT_PLOC_RADIUS neighbours on either side for the one
// whose union box has the smallest surface area; clusters that chose each other merge into a new
// inner node, the rest carry over, and the survivors are compacted in order.  The result is a
// binary tree in the same node layout as the LBVH's, usually with less overlap.  Node ids are
// handed out downwards so that the last merge, the root, is node 0.
#ifndef VRT_PLOC_RADIUS
#define VRT_PLOC_RADIUS 12
#endif

__global__ __launch_bounds__(256) void k_ploc_init(int n, const unsigned* __restrict__ vals, const float4* __restrict__ pbox,
                                                   int* __restrict__ code, float4* __restrict__ lo, float4* __restrict__ hi)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int prim = (int)vals[i];
    code[i] = ~prim;
    lo[i] = pbox[2 * (size_t)prim];
    hi[i] = pbox[2 * (size_t)prim + 1];
}

__global__ __launch_bounds__(256) void k_ploc_nn(int m, const float4* __restrict__ lo, const float4* __restrict__ hi,
                                                 int* __restrict__ nn)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    float4 a = lo[i], b = hi[i];
    int best = -1;
    float best_area = INFINITY;
    int j0 = max(0, i - VRT_PLOC_RADIUS), j1 = min(m - 1, i + VRT_PLOC_RADIUS);
    for (int j = j0; j <= j1; ++j) {
        if (j == i) continue;
        float4 c = lo[j], d = hi[j];
        float dx = fmaxf(b.x, d.x) - fminf(a.x, c.x);
        float dy = fmaxf(b.y, d.y) - fminf(a.y, c.y);
        float dz = fmaxf(b.z, d.z) - fminf(a.z, c.z);
        float area = dx * dy + dy * dz + dz * dx;
        if (area < best_area) { best_area = area; best = j; } // ties: the lower index, so a mutual pair always exists
    }
    nn[i] = best;
}

// flags[i] = valid | merge << 32 (summed by one 64-bit scan)
__global__ __launch_bounds__(256) void k_ploc_flags(int m, const int* __restrict__ nn, unsigned long long* __restrict__ flags)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    int j = nn[i];
    bool mutual = j >= 0 && nn[j] == i;
    unsigned long long f;
    if (mutual && i < j) f = 1ull | (1ull << 32);
    else if (mutual) f = 0ull;
    else f = 1ull;
    flags[i] = f;
}

__global__ __launch_bounds__(256) void k_ploc_emit(int m, int left, const int* __restrict__ nn,
                                                   const unsigned long long* __restrict__ flags,
                                                   const unsigned long long* __restrict__ scan,
                                                   const int* __restrict__ code, const float4* __restrict__ lo,
                                                   const float4* __restrict__ hi, int* __restrict__ code_out,
                                                   float4* __restrict__ lo_out, float4* __restrict__ hi_out,
                                                   float4* __restrict__ nodes, int* __restrict__ parent,
                                                   int* __restrict__ leaf_parent)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    unsigned long long f = flags[i];
    if (!(f & 1ull)) return;
    unsigned long long sc = scan[i];
    int pos = (int)(unsigned)sc, rank = (int)(unsigned)(sc >> 32);
    int c = code[i];
    float4 a = lo[i], b = hi[i];
    if (f >> 32) {
        int j = nn[i];
        int id = left - 1 - rank;
        int cj = code[j];
        float4 cl = lo[j], ch = hi[j];
        float4* nd = nodes + 4 * (size_t)id;
        nd[0] = make_float4(a.x, a.y, a.z, b.x);
        nd[1] = make_float4(b.y, b.z, cl.x, cl.y);
        nd[2] = make_float4(cl.z, ch.x, ch.y, ch.z);
        nd[3] = make_float4(__int_as_float(c), __int_as_float(cj), 0.f, 0.f);
        if (c >= 0) parent[c] = id; else leaf_parent[~c] = id;
        if (cj >= 0) parent[cj] = id; else leaf_parent[~cj] = id;
        if (id == 0) parent[0] = -1;
        a = make_float4(fminf(a.x, cl.x), fminf(a.y, cl.y), fminf(a.z, cl.z), 0.f);
        b = make_float4(fmaxf(b.x, ch.x), fmaxf(b.y, ch.y), fmaxf(b.z, ch.z), 0.f);
        c = id;
    }
    code_out[pos] = c; lo_out[pos] = a; hi_out[pos] = b;
}

// Surface-area-guided 4-wide collapse, one lane per node of the current wide level: the node
// starts from its two binary children and, while it has a free slot, opens the inner child with
// the largest box (Wald et al.'s greedy collapse).  Inner children that are left become the
// wide nodes of the next level (appended to out_q).  A wide node keeps its binary index.
__global__ __launch_bounds__(256) void k_collapse4_level(const float4* __restrict__ nodes, float4* __restrict__ wide,
                                                         const int* __restrict__ in_q, int n_in,
                                                         int* __restrict__ out_q, int* __restrict__ out_count)
{
    int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_in) return;
    const int x = in_q[t];
    float lo[4][3], hi[4][3];
    int code[4];
    int m = 2;
    {
        const float4* nd = nodes + 4 * (size_t)x;
        float4 a0 = nd[0], a1 = nd[1], a2 = nd[2], a3 = nd[3];
        lo[0][0] = a0.x; lo[0][1] = a0.y; lo[0][2] = a0.z; hi[0][0] = a0.w; hi[0][1] = a1.x; hi[0][2] = a1.y;
        lo[1][0] = a1.z; lo[1][1] = a1.w; lo[1][2] = a2.x; hi[1][0] = a2.y; hi[1][1] = a2.z; hi[1][2] = a2.w;
        code[0] = __float_as_int(a3.x); code[1] = __float_as_int(a3.y);
    }
    while (m < 4) {
        int best = -1;
        float best_area = -1.f;
        for (int k = 0; k < m; ++k) {
            if (code[k] < 0) continue;
            float dx = hi[k][0] - lo[k][0], dy = hi[k][1] - lo[k][1], dz = hi[k][2] - lo[k][2];
            float area = dx * dy + dy * dz + dz * dx;
            if (area > best_area) { best_area = area; best = k; }
        }
        if (best < 0) break;
        const float4* cd = nodes + 4 * (size_t)code[best];
        float4 c0 = cd[0], c1 = cd[1], c2 = cd[2], c3 = cd[3];
        lo[best][0] = c0.x; lo[best][1] = c0.y; lo[best][2] = c0.z; hi[best][0] = c0.w; hi[best][1] = c1.x; hi[best][2] = c1.y;
        code[best] = __float_as_int(c3.x);
        lo[m][0] = c1.z; lo[m][1] = c1.w; lo[m][2] = c2.x; hi[m][0] = c2.y; hi[m][1] = c2.z; hi[m][2] = c2.w;
        code[m] = __float_as_int(c3.y);
        ++m;
    }
    for (int k = 0; k < m; ++k)
        if (code[k] >= 0) out_q[atomicAdd(out_count, 1)] = code[k];
    for (; m < 4; ++m) {
        for (int k = 0; k < 3; ++k) { lo[m][k] = INFINITY; hi[m][k] = INFINITY; } // see k_collapse4
        code[m] = VRT_BVH4_EMPTY;
    }
    float4* w = wide + 8 * (size_t)x;
    w[0] = make_float4(lo[0][0], lo[1][0], lo[2][0], lo[3][0]);
    w[1] = make_float4(lo[0][1], lo[1][1], lo[2][1], lo[3][1]);
    w[2] = make_float4(lo[0][2], lo[1][2], lo[2][2], lo[3][2]);
    w[3] = make_float4(hi[0][0], hi[1][0], hi[2][0], hi[3][0]);
    w[4] = make_float4(hi[0][1], hi[1][1], hi[2][1], hi[3][1]);
    w[5] = make_float4(hi[0][2], hi[1][2], hi[2][2], hi[3][2]);
    w[6] = make_float4(__int_as_float(code[0]), __int_as_float(code[1]), __int_as_float(code[2]),
                       __int_as_float(code[3]));
    w[7] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------ binned SAH build on the device
//
// The same top-down build as build_sah_host (16 bins per axis over the centroids, one primitive
// per leaf), level by level: one wave per node of the level.  The wave reads its primitives
// (contiguous in `in`), reduces the centroid bounds, bins into its part of LDS (ordered-uint
// min/max, integer counts), three lanes sweep one axis each, and a stable two-sided partition
// writes the primitives to the same range of `out`, accumulating both children's boxes on the
// way, so the node is complete when the wave is.  A subtree over m primitives has m - 1 inner
// nodes, hence the children's preorder numbers are idx + 1 and idx + n_left: no counter, and the
// tree does not depend on scheduling.  Children of two or more primitives join the next level's
// queue.  Beyond level 48 ranges are halved by index (depth bound for contrived scenes).
struct SahTask { int b, e, idx, parent; };
#define VRT_SAH_NB 16

// Minimum / maximum over the G lanes of a group (G = 64: the wave).
template <int G>
__device__ __forceinline__ float sah_group_min(float v)
{
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int G>
__device__ __forceinline__ float sah_group_max(float v)
{
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float sah_wave_min(float v) { return sah_group_min<64>(v); }
__device__ __forceinline__ float sah_wave_max(float v) { return sah_group_max<64>(v); }
__device__ __forceinline__ float sah_area(const float lo[3], const float hi[3])
{
    const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return x < 0.f ? 0.f : x * y + y * z + z * x;
}

// Tasks of at most VRT_SAH_SMALL primitives are split by groups of 16 lanes, four to a wave: near
// the leaves a level of the 1 M-triangle build has 250 000 nodes of two to eight primitives, and
// a wave each left 56 and more of its lanes idle (5 of the build's 11 ms).
#define VRT_SAH_SMALL 32

// Appends a child task to the queue of its size class.  counts: [0] wave-sized, [3] small.
// (One atomic per call on a counter the whole chip shares: fine for the few large tasks of
// k_big_finish; k_sah_level claims its slots once per block.)
__device__ __forceinline__ void sah_push_child(int b, int e, int idx, int parent, SahTask* __restrict__ out_q,
                                               SahTask* __restrict__ out_small, int* __restrict__ counts)
{
    const bool small = e - b <= VRT_SAH_SMALL;
    SahTask* q = (small ? out_small : out_q) + atomicAdd(&counts[small ? 3 : 0], 1);
    q->b = b; q->e = e; q->idx = idx; q->parent = parent;
}

// G lanes per task (64 or 16), 256 / G tasks per block.
template <int G>
__global__ __launch_bounds__(256) void k_sah_level(int level, const float4* __restrict__ in, float4* __restrict__ out,
                                                   const SahTask* __restrict__ in_q, int n_in,
                                                   SahTask* __restrict__ out_q, SahTask* __restrict__ out_small,
                                                   int* __restrict__ counts,
                                                   float4* __restrict__ nodes, int* __restrict__ parent,
                                                   int* __restrict__ leaf_parent)
{
    // per group: 3 axes x 16 bins x (count, lo[3], hi[3]) and two words for the ids of one-primitive children
    __shared__ unsigned s_bins[256 / G][3 * VRT_SAH_NB * 7 + 2];
    // Queue slots for the children are claimed once per block: an atomic on one address is served
    // in about 9 ns whoever issues it, and a level near the leaves has 250 000 tasks.
    __shared__ int s_want[2], s_base[2];   // [0] wave-sized children, [1] small ones
    const int grp = threadIdx.x / G, lane = threadIdx.x % G;
    const int shift = (threadIdx.x & 63) & ~(G - 1);                 // first lane of the group within its wave
    const unsigned long long gmask = G == 64 ? ~0ull : (((1ull << (G & 63)) - 1ull) << shift);
    const int t = blockIdx.x * (256 / G) + grp;
    if (threadIdx.x < 2) s_want[threadIdx.x] = 0;
    __syncthreads();
    const bool have = t < n_in;   // groups without a task only take part in the block's barriers
    SahTask task = {0, 0, 0, -1};
    if (have) task = in_q[t];
    const int b = task.b, e = task.e, idx = task.idx, m = e - b;
    unsigned* bins = s_bins[grp];
    if (have && lane == 0) parent[idx] = task.parent;

    // 1. centroid bounds
    float cl[3] = {INFINITY, INFINITY, INFINITY}, ch[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = b + lane; i < e; i += G) {
        const float4 lo = in[2 * (size_t)i], hi = in[2 * (size_t)i + 1];
        const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
        for (int k = 0; k < 3; ++k) { cl[k] = fminf(cl[k], c[k]); ch[k] = fmaxf(ch[k], c[k]); }
    }
    for (int k = 0; k < 3; ++k) { cl[k] = sah_group_min<G>(cl[k]); ch[k] = sah_group_max<G>(ch[k]); }

    // 2. bins
    int best_axis = -1, best_bin = 0;
    if (level < 48 && (ch[0] > cl[0] || ch[1] > cl[1] || ch[2] > cl[2])) {
        for (int w = lane; w < 3 * VRT_SAH_NB * 7; w += G) {
            const int f = w % 7;
            bins[w] = f == 0 ? 0u : (f <= 3 ? 0xffffffffu : 0u);   // count, lo (min), hi (max) as ordered uints
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float scale[3];
        for (int k = 0; k < 3; ++k) scale[k] = ch[k] > cl[k] ? VRT_SAH_NB / (ch[k] - cl[k]) : 0.f;
        for (int i = b + lane; i < e; i += G) {
            const float4 lo = in[2 * (size_t)i], hi = in[2 * (size_t)i + 1];
            const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
            const float l3[3] = {lo.x, lo.y, lo.z}, h3[3] = {hi.x, hi.y, hi.z};
            for (int k = 0; k < 3; ++k) {
                if (!(ch[k] > cl[k])) continue;
                int q = (int)((c[k] - cl[k]) * scale[k]);
                q = q < 0 ? 0 : (q >= VRT_SAH_NB ? VRT_SAH_NB - 1 : q);
                unsigned* bin = bins + (k * VRT_SAH_NB + q) * 7;
                atomicAdd(&bin[0], 1u);
                for (int a = 0; a < 3; ++a) { atomicMin(&bin[1 + a], enc_f(l3[a])); atomicMax(&bin[4 + a], enc_f(h3[a])); }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 3. lanes 0..2 of the group sweep one axis each
        float my_cost = INFINITY;
        int my_bin = 0;
        if (lane < 3 && ch[lane] > cl[lane]) {
            const unsigned* ax = bins + lane * VRT_SAH_NB * 7;
            float right_area[VRT_SAH_NB];
            int right_cnt[VRT_SAH_NB];
            float alo[3] = {INFINITY, INFINITY, INFINITY}, ahi[3] = {-INFINITY, -INFINITY, -INFINITY};
            int n_acc = 0;
            for (int q = VRT_SAH_NB - 1; q > 0; --q) {
                const unsigned* bin = ax + q * 7;
                if (bin[0]) for (int a = 0; a < 3; ++a) { alo[a] = fminf(alo[a], dec_f(bin[1 + a])); ahi[a] = fmaxf(ahi[a], dec_f(bin[4 + a])); }
                n_acc += (int)bin[0];
                right_area[q] = sah_area(alo, ahi); right_cnt[q] = n_acc;
            }
            float llo[3] = {INFINITY, INFINITY, INFINITY}, lhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            int n_left = 0;
            for (int q = 0; q < VRT_SAH_NB - 1; ++q) {
                const unsigned* bin = ax + q * 7;
                if (bin[0]) for (int a = 0; a < 3; ++a) { llo[a] = fminf(llo[a], dec_f(bin[1 + a])); lhi[a] = fmaxf(lhi[a], dec_f(bin[4 + a])); }
                n_left += (int)bin[0];
                if (n_left == 0 || right_cnt[q + 1] == 0) continue;
                const float cost = sah_area(llo, lhi) * n_left + right_area[q + 1] * right_cnt[q + 1];
                if (cost < my_cost) { my_cost = cost; my_bin = q; }
            }
        }
        // the first axis with the smallest cost, as the host build picks it
        float best = INFINITY;
        for (int k = 0; k < 3; ++k) {
            const float ck = __shfl(my_cost, shift + k, 64);
            const int bk = __shfl(my_bin, shift + k, 64);
            if (ck < best) { best = ck; best_axis = k; best_bin = bk; }
        }
    }

    // 4. stable partition into `out`, the children's boxes on the way
    float scale_k = 0.f, base_k = 0.f;
    if (best_axis >= 0) { base_k = cl[best_axis]; scale_k = VRT_SAH_NB / (ch[best_axis] - cl[best_axis]); }
    auto goes_left = [&](const float4& lo, const float4& hi, int i) {
        if (best_axis < 0) return false;
        const float c = best_axis == 0 ? 0.5f * (lo.x + hi.x) : (best_axis == 1 ? 0.5f * (lo.y + hi.y) : 0.5f * (lo.z + hi.z));
        int q = (int)((c - base_k) * scale_k);
        q = q < 0 ? 0 : (q >= VRT_SAH_NB ? VRT_SAH_NB - 1 : q);
        return q <= best_bin;
    };
    int n_left = 0;
    if (best_axis >= 0) {
        for (int base = 0; base < m; base += G) {
            const int i = b + base + lane;
            bool p = false;
            if (i < e) p = goes_left(in[2 * (size_t)i], in[2 * (size_t)i + 1], i);
            n_left += __popcll(__ballot(p) & gmask);
        }
    }
    const bool by_index = best_axis < 0 || n_left == 0 || n_left == m;   // no usable split: halve by index
    if (by_index) n_left = m / 2;
    float llo[3] = {INFINITY, INFINITY, INFINITY}, lhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float rlo[3] = {INFINITY, INFINITY, INFINITY}, rhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    int off_l = 0, off_r = 0;
    for (int base = 0; base < m; base += G) {
        const int i = b + base + lane;
        const bool valid = i < e;
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
        bool p = false;
        if (valid) {
            lo = in[2 * (size_t)i]; hi = in[2 * (size_t)i + 1];
            p = by_index ? (i - b) < n_left : goes_left(lo, hi, i);
        }
        const unsigned long long ml = __ballot(valid && p) & gmask, mr = __ballot(valid && !p) & gmask;
        if (valid) {
            const unsigned long long mine = (p ? ml : mr) >> shift;   // the group's lanes from bit 0
            const int rank = __popcll(mine & ((1ull << lane) - 1ull));
            const int dest = p ? b + off_l + rank : b + n_left + off_r + rank;
            out[2 * (size_t)dest] = lo; out[2 * (size_t)dest + 1] = hi;
            const float l3[3] = {lo.x, lo.y, lo.z}, h3[3] = {hi.x, hi.y, hi.z};
            if (p) for (int a = 0; a < 3; ++a) { llo[a] = fminf(llo[a], l3[a]); lhi[a] = fmaxf(lhi[a], h3[a]); }
            else   for (int a = 0; a < 3; ++a) { rlo[a] = fminf(rlo[a], l3[a]); rhi[a] = fmaxf(rhi[a], h3[a]); }
            // a child of one primitive is a leaf: its id is the fourth word of the primitive's lo
            if (p && n_left == 1) bins[3 * VRT_SAH_NB * 7] = __float_as_uint(lo.w);
            if (!p && m - n_left == 1) bins[3 * VRT_SAH_NB * 7 + 1] = __float_as_uint(lo.w);
        }
        off_l += __popcll(ml); off_r += __popcll(mr);
    }
    for (int a = 0; a < 3; ++a) {
        llo[a] = sah_group_min<G>(llo[a]); lhi[a] = sah_group_max<G>(lhi[a]);
        rlo[a] = sah_group_min<G>(rlo[a]); rhi[a] = sah_group_max<G>(rhi[a]);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // 5. the node and its children
    const int n_right = m - n_left;
    const int idx_l = idx + 1, idx_r = idx + n_left;
    const bool small_l = n_left <= VRT_SAH_SMALL, small_r = n_right <= VRT_SAH_SMALL;
    int at_l = 0, at_r = 0;   // the children's places among the block's new tasks of their class
    if (have && lane == 0) {
        if (n_left > 1) at_l = atomicAdd(&s_want[small_l ? 1 : 0], 1);
        if (n_right > 1) at_r = atomicAdd(&s_want[small_r ? 1 : 0], 1);
    }
    __syncthreads();
    if (threadIdx.x < 2) s_base[threadIdx.x] = s_want[threadIdx.x] ? atomicAdd(&counts[threadIdx.x ? 3 : 0], s_want[threadIdx.x]) : 0;
    __syncthreads();
    if (have && lane == 0) {
        int c0, c1;
        if (n_left == 1) { const int id = (int)bins[3 * VRT_SAH_NB * 7]; c0 = ~id; leaf_parent[id] = idx; } else c0 = idx_l;
        if (n_right == 1) { const int id = (int)bins[3 * VRT_SAH_NB * 7 + 1]; c1 = ~id; leaf_parent[id] = idx; } else c1 = idx_r;
        float4* nd = nodes + 4 * (size_t)idx;
        nd[0] = make_float4(llo[0], llo[1], llo[2], lhi[0]);
        nd[1] = make_float4(lhi[1], lhi[2], rlo[0], rlo[1]);
        nd[2] = make_float4(rlo[2], rhi[0], rhi[1], rhi[2]);
        nd[3] = make_float4(__int_as_float(c0), __int_as_float(c1), 0.f, 0.f);
        if (n_left > 1) {
            const SahTask q = {b, b + n_left, idx_l, idx};
            (small_l ? out_small : out_q)[s_base[small_l ? 1 : 0] + at_l] = q;
        }
        if (n_right > 1) {
            const SahTask q = {b + n_left, e, idx_r, idx};
            (small_r ? out_small : out_q)[s_base[small_r ? 1 : 0] + at_r] = q;
        }
    }
}

// ------------------------------------------------------------------ large nodes of the SAH build
//
// One wave per node leaves the chip idle while the tree is still narrow: the root of a 1 M
// triangle scene is one wave walking a million primitives three times.  Nodes of more than
// VRT_SAH_BIG primitives are therefore split by many blocks, VRT_SAH_CHUNK primitives each, in
// four launches per level: k_big_bin (LDS bins per block, merged into the node's bins with integer
// atomics), k_big_count (every block repeats the node's sweep, counts the primitives of its chunk
// that go left), k_big_scatter (stable partition: a chunk's offsets are the counts of the chunks
// before it; the children's boxes and centroid bounds are gathered with ordered-uint atomics, so
// no pass over the children is needed to start on them) and k_big_finish (node record, children
// queued as large or small tasks).  Bins, counts, minima and maxima are exact whatever the order of
// the atomics, and the partition is stable, so the tree is the one the one-wave build produces.
// Minima are kept as maxima of the complemented code, which lets one memset of zeros reset a level.
#ifndef VRT_SAH_BIG
#define VRT_SAH_BIG 4096
#endif
#ifndef VRT_SAH_CHUNK
#define VRT_SAH_CHUNK 1024
#endif
#define VRT_SAH_BIN_WORDS (3 * VRT_SAH_NB * 7)

// Decodes a minimum kept as the maximum of complemented codes: dec_fc(v).  (Written on the
// complemented word directly; hipcc 7.2 crashes in instruction selection on dec_fc(v) for gfx950.)
__device__ __forceinline__ float dec_fc(unsigned v)
{
    return __uint_as_float((v & 0x80000000u) ? v : (v ^ 0x7fffffffu));
}

struct BigTask {
    int b, e, idx, parent;
    int chunk0, n_chunks;
    unsigned cb[6];        // centroid bounds as ordered uints: minima x y z, maxima x y z
};
struct BigSplit { int axis, bin, n_left, by_index; float base, scale; };
struct BigOut {
    unsigned box[12];      // left lo[3] (complemented), left hi[3], right lo[3] (complemented), right hi[3]
    unsigned cen[12];      // the same for the children's centroids
    unsigned leaf[2];      // primitive id of a child of one primitive
    unsigned pad[2];
};

// The node's split from its bins (the sweep of k_sah_level, step 3) and from them the number of
// primitives that go left.  Called by one whole wave; the result is valid in every lane.
__device__ __forceinline__ BigSplit big_split(int level, const BigTask& T, const unsigned* __restrict__ bins, int lane)
{
    float cl[3], ch[3];
    for (int k = 0; k < 3; ++k) { cl[k] = dec_f(T.cb[k]); ch[k] = dec_f(T.cb[3 + k]); }
    const int m = T.e - T.b;
    int best_axis = -1, best_bin = 0;
    if (level < 48 && (ch[0] > cl[0] || ch[1] > cl[1] || ch[2] > cl[2])) {
        float my_cost = INFINITY;
        int my_bin = 0;
        if (lane < 3 && ch[lane] > cl[lane]) {
            const unsigned* ax = bins + lane * VRT_SAH_NB * 7;
            float right_area[VRT_SAH_NB];
            int right_cnt[VRT_SAH_NB];
            float alo[3] = {INFINITY, INFINITY, INFINITY}, ahi[3] = {-INFINITY, -INFINITY, -INFINITY};
            int n_acc = 0;
            for (int q = VRT_SAH_NB - 1; q > 0; --q) {
                const unsigned* bin = ax + q * 7;
                if (bin[0]) for (int a = 0; a < 3; ++a) { alo[a] = fminf(alo[a], dec_fc(bin[1 + a])); ahi[a] = fmaxf(ahi[a], dec_f(bin[4 + a])); }
                n_acc += (int)bin[0];
                right_area[q] = sah_area(alo, ahi); right_cnt[q] = n_acc;
            }
            float llo[3] = {INFINITY, INFINITY, INFINITY}, lhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            int n_left = 0;
            for (int q = 0; q < VRT_SAH_NB - 1; ++q) {
                const unsigned* bin = ax + q * 7;
                if (bin[0]) for (int a = 0; a < 3; ++a) { llo[a] = fminf(llo[a], dec_fc(bin[1 + a])); lhi[a] = fmaxf(lhi[a], dec_f(bin[4 + a])); }
                n_left += (int)bin[0];
                if (n_left == 0 || right_cnt[q + 1] == 0) continue;
                const float cost = sah_area(llo, lhi) * n_left + right_area[q + 1] * right_cnt[q + 1];
                if (cost < my_cost) { my_cost = cost; my_bin = q; }
            }
        }
        float best = INFINITY;
        for (int k = 0; k < 3; ++k) {
            const float ck = __shfl(my_cost, k, 64);
            const int bk = __shfl(my_bin, k, 64);
            if (ck < best) { best = ck; best_axis = k; best_bin = bk; }
        }
    }
    BigSplit s;
    s.axis = best_axis; s.bin = best_bin; s.n_left = 0; s.base = 0.f; s.scale = 0.f;
    if (best_axis >= 0) {
        s.base = cl[best_axis]; s.scale = VRT_SAH_NB / (ch[best_axis] - cl[best_axis]);
        for (int q = 0; q <= best_bin; ++q) s.n_left += (int)bins[(best_axis * VRT_SAH_NB + q) * 7];
    }
    s.by_index = best_axis < 0 || s.n_left == 0 || s.n_left == m;
    if (s.by_index) s.n_left = m / 2;
    return s;
}

__device__ __forceinline__ bool big_goes_left(const BigSplit& s, const float4& lo, const float4& hi, int i_rel)
{
    if (s.by_index) return i_rel < s.n_left;
    const float c = s.axis == 0 ? 0.5f * (lo.x + hi.x) : (s.axis == 1 ? 0.5f * (lo.y + hi.y) : 0.5f * (lo.z + hi.z));
    int q = (int)((c - s.base) * s.scale);
    q = q < 0 ? 0 : (q >= VRT_SAH_NB ? VRT_SAH_NB - 1 : q);
    return q <= s.bin;
}

// The root as a large task: its centroid bounds are those k_prim_bounds gathered.
__global__ __launch_bounds__(256) void k_big_root(int n, const unsigned* __restrict__ cb, BigTask* __restrict__ tasks,
                                                  int* __restrict__ chunk_task)
{
    const int n_chunks = (n + VRT_SAH_CHUNK - 1) / VRT_SAH_CHUNK;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        BigTask T;
        T.b = 0; T.e = n; T.idx = 0; T.parent = -1; T.chunk0 = 0; T.n_chunks = n_chunks;
        for (int k = 0; k < 6; ++k) T.cb[k] = cb[k];
        tasks[0] = T;
    }
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n_chunks; j += gridDim.x * 256) chunk_task[j] = 0;
}

__global__ __launch_bounds__(256) void k_big_bin(int level, const float4* __restrict__ in, const BigTask* __restrict__ tasks,
                                                 const int* __restrict__ chunk_task, unsigned* __restrict__ bins)
{
    __shared__ unsigned s_bins[VRT_SAH_BIN_WORDS];
    const int t = chunk_task[blockIdx.x];
    const BigTask T = tasks[t];
    float cl[3], ch[3], scale[3];
    for (int k = 0; k < 3; ++k) {
        cl[k] = dec_f(T.cb[k]); ch[k] = dec_f(T.cb[3 + k]);
        scale[k] = ch[k] > cl[k] ? VRT_SAH_NB / (ch[k] - cl[k]) : 0.f;
    }
    if (!(level < 48 && (ch[0] > cl[0] || ch[1] > cl[1] || ch[2] > cl[2]))) return;   // the whole task: halved by index
    for (int w = threadIdx.x; w < VRT_SAH_BIN_WORDS; w += 256) s_bins[w] = 0u;
    __syncthreads();
    const int first = T.b + (blockIdx.x - T.chunk0) * VRT_SAH_CHUNK;
    const int last = first + VRT_SAH_CHUNK < T.e ? first + VRT_SAH_CHUNK : T.e;
    for (int i = first + threadIdx.x; i < last; i += 256) {
        const float4 lo = in[2 * (size_t)i], hi = in[2 * (size_t)i + 1];
        const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
        const float l3[3] = {lo.x, lo.y, lo.z}, h3[3] = {hi.x, hi.y, hi.z};
        for (int k = 0; k < 3; ++k) {
            if (!(ch[k] > cl[k])) continue;
            int q = (int)((c[k] - cl[k]) * scale[k]);
            q = q < 0 ? 0 : (q >= VRT_SAH_NB ? VRT_SAH_NB - 1 : q);
            unsigned* bin = s_bins + (k * VRT_SAH_NB + q) * 7;
            atomicAdd(&bin[0], 1u);
            for (int a = 0; a < 3; ++a) { atomicMax(&bin[1 + a], ~enc_f(l3[a])); atomicMax(&bin[4 + a], enc_f(h3[a])); }
        }
    }
    __syncthreads();
    unsigned* g = bins + (size_t)t * VRT_SAH_BIN_WORDS;
    for (int w = threadIdx.x; w < VRT_SAH_BIN_WORDS; w += 256) {
        const unsigned v = s_bins[w];
        if (!v) continue;
        if (w % 7 == 0) atomicAdd(&g[w], v); else atomicMax(&g[w], v);
    }
}

__global__ __launch_bounds__(256) void k_big_count(int level, const float4* __restrict__ in, const BigTask* __restrict__ tasks,
                                                   const int* __restrict__ chunk_task, const unsigned* __restrict__ bins,
                                                   BigSplit* __restrict__ split, int* __restrict__ left_count)
{
    __shared__ BigSplit s_split;
    __shared__ int s_cnt;
    const int t = chunk_task[blockIdx.x];
    const BigTask T = tasks[t];
    const int c = blockIdx.x - T.chunk0;
    if (threadIdx.x < 64) {
        const BigSplit s = big_split(level, T, bins + (size_t)t * VRT_SAH_BIN_WORDS, threadIdx.x);
        if (threadIdx.x == 0) { s_split = s; s_cnt = 0; if (c == 0) split[t] = s; }
    }
    __syncthreads();
    const BigSplit s = s_split;
    const int first = T.b + c * VRT_SAH_CHUNK;
    const int last = first + VRT_SAH_CHUNK < T.e ? first + VRT_SAH_CHUNK : T.e;
    int mine = 0;
    for (int i = first + threadIdx.x; i < last; i += 256)
        mine += big_goes_left(s, in[2 * (size_t)i], in[2 * (size_t)i + 1], i - T.b) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) left_count[blockIdx.x] = s_cnt;
}

__global__ __launch_bounds__(256) void k_big_scatter(const float4* __restrict__ in, float4* __restrict__ out,
                                                     const BigTask* __restrict__ tasks, const int* __restrict__ chunk_task,
                                                     const BigSplit* __restrict__ split, const int* __restrict__ left_count,
                                                     BigOut* __restrict__ outs)
{
    __shared__ int s_before;
    __shared__ int s_wl[4], s_wr[4];
    const int t = chunk_task[blockIdx.x];
    const BigTask T = tasks[t];
    const BigSplit s = split[t];
    const int c = blockIdx.x - T.chunk0;
    const int wave = threadIdx.x >> 6;
    const int m = T.e - T.b;
    if (threadIdx.x == 0) s_before = 0;
    __syncthreads();
    {   // primitives of the chunks before this one that go left
        int acc = 0;
        for (int j = threadIdx.x; j < c; j += 256) acc += left_count[T.chunk0 + j];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&s_before, acc);
    }
    __syncthreads();
    int run_l = s_before, run_r = c * VRT_SAH_CHUNK - s_before;
    float llo[3] = {INFINITY, INFINITY, INFINITY}, lhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float rlo[3] = {INFINITY, INFINITY, INFINITY}, rhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float lcl[3] = {INFINITY, INFINITY, INFINITY}, lch[3] = {-INFINITY, -INFINITY, -INFINITY};
    float rcl[3] = {INFINITY, INFINITY, INFINITY}, rch[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool any_l = false, any_r = false;
    const int first = T.b + c * VRT_SAH_CHUNK;
    for (int base = 0; base < VRT_SAH_CHUNK; base += 256) {   // block-uniform trip count
        const int i = first + base + threadIdx.x;
        const bool valid = i < T.e;
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
        bool p = false;
        if (valid) {
            lo = in[2 * (size_t)i]; hi = in[2 * (size_t)i + 1];
            p = big_goes_left(s, lo, hi, i - T.b);
        }
        const unsigned long long ml = __ballot(valid && p), mr = __ballot(valid && !p);
        if ((threadIdx.x & 63) == 0) { s_wl[wave] = __popcll(ml); s_wr[wave] = __popcll(mr); }
        __syncthreads();
        int pre_l = 0, pre_r = 0, tot_l = 0, tot_r = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) { pre_l += s_wl[w]; pre_r += s_wr[w]; }
            tot_l += s_wl[w]; tot_r += s_wr[w];
        }
        if (valid) {
            const unsigned long long mine = p ? ml : mr;
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mine >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mine, 0u));
            const int dest = p ? T.b + run_l + pre_l + rank : T.b + s.n_left + run_r + pre_r + rank;
            out[2 * (size_t)dest] = lo; out[2 * (size_t)dest + 1] = hi;
            const float l3[3] = {lo.x, lo.y, lo.z}, h3[3] = {hi.x, hi.y, hi.z};
            const float c3[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
            if (p) {
                any_l = true;
                for (int a = 0; a < 3; ++a) {
                    llo[a] = fminf(llo[a], l3[a]); lhi[a] = fmaxf(lhi[a], h3[a]);
                    lcl[a] = fminf(lcl[a], c3[a]); lch[a] = fmaxf(lch[a], c3[a]);
                }
                if (s.n_left == 1) outs[t].leaf[0] = __float_as_uint(lo.w);
            } else {
                any_r = true;
                for (int a = 0; a < 3; ++a) {
                    rlo[a] = fminf(rlo[a], l3[a]); rhi[a] = fmaxf(rhi[a], h3[a]);
                    rcl[a] = fminf(rcl[a], c3[a]); rch[a] = fmaxf(rch[a], c3[a]);
                }
                if (m - s.n_left == 1) outs[t].leaf[1] = __float_as_uint(lo.w);
            }
        }
        run_l += tot_l; run_r += tot_r;
        __syncthreads();
    }
    const bool wave_l = __any(any_l), wave_r = __any(any_r);
    for (int a = 0; a < 3; ++a) {
        llo[a] = sah_wave_min(llo[a]); lhi[a] = sah_wave_max(lhi[a]); rlo[a] = sah_wave_min(rlo[a]); rhi[a] = sah_wave_max(rhi[a]);
        lcl[a] = sah_wave_min(lcl[a]); lch[a] = sah_wave_max(lch[a]); rcl[a] = sah_wave_min(rcl[a]); rch[a] = sah_wave_max(rch[a]);
    }
    // the block's four waves meet in LDS first: the 24 words of a task share two cache lines, and
    // atomics on one line are served one after the other for the whole chip
    __shared__ unsigned s_out[24];
    if (threadIdx.x < 24) s_out[threadIdx.x] = 0u;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        if (wave_l) for (int a = 0; a < 3; ++a) {
            atomicMax(&s_out[a], ~enc_f(llo[a])); atomicMax(&s_out[3 + a], enc_f(lhi[a]));
            atomicMax(&s_out[12 + a], ~enc_f(lcl[a])); atomicMax(&s_out[15 + a], enc_f(lch[a]));
        }
        if (wave_r) for (int a = 0; a < 3; ++a) {
            atomicMax(&s_out[6 + a], ~enc_f(rlo[a])); atomicMax(&s_out[9 + a], enc_f(rhi[a]));
            atomicMax(&s_out[18 + a], ~enc_f(rcl[a])); atomicMax(&s_out[21 + a], enc_f(rch[a]));
        }
    }
    __syncthreads();
    if (threadIdx.x < 24 && s_out[threadIdx.x]) {
        BigOut& O = outs[t];
        atomicMax(threadIdx.x < 12 ? &O.box[threadIdx.x] : &O.cen[threadIdx.x - 12], s_out[threadIdx.x]);
    }
}

// One lane per large task: the node record (step 5 of k_sah_level) and the children as tasks of
// the next level, large ones with their chunk numbers and centroid bounds.
__global__ __launch_bounds__(64) void k_big_finish(int n_big, const BigTask* __restrict__ tasks, const BigSplit* __restrict__ split,
                                                   const BigOut* __restrict__ outs, float4* __restrict__ nodes,
                                                   int* __restrict__ parent, int* __restrict__ leaf_parent,
                                                   SahTask* __restrict__ out_q, SahTask* __restrict__ out_small,
                                                   int* __restrict__ counts,
                                                   BigTask* __restrict__ next_tasks, int* __restrict__ next_chunk_task)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n_big) return;
    const BigTask T = tasks[t];
    const unsigned* __restrict__ box = outs[t].box;   // read in place: a copy indexed by `side` would live in scratch
    const unsigned* __restrict__ cen = outs[t].cen;
    const int n_left = split[t].n_left, m = T.e - T.b, n_right = m - n_left;
    const int idx = T.idx, idx_l = idx + 1, idx_r = idx + n_left;
    parent[idx] = T.parent;
    int c0, c1;
    if (n_left == 1) { const int id = (int)outs[t].leaf[0]; c0 = ~id; leaf_parent[id] = idx; } else c0 = idx_l;
    if (n_right == 1) { const int id = (int)outs[t].leaf[1]; c1 = ~id; leaf_parent[id] = idx; } else c1 = idx_r;
    float4* nd = nodes + 4 * (size_t)idx;
    nd[0] = make_float4(dec_fc(box[0]), dec_fc(box[1]), dec_fc(box[2]), dec_f(box[3]));
    nd[1] = make_float4(dec_f(box[4]), dec_f(box[5]), dec_fc(box[6]), dec_fc(box[7]));
    nd[2] = make_float4(dec_fc(box[8]), dec_f(box[9]), dec_f(box[10]), dec_f(box[11]));
    nd[3] = make_float4(__int_as_float(c0), __int_as_float(c1), 0.f, 0.f);
    for (int side = 0; side < 2; ++side) {
        const int b = side ? T.b + n_left : T.b, e = side ? T.e : T.b + n_left, size = e - b;
        const int child_idx = side ? idx_r : idx_l;
        if (size <= 1) continue;
        if (size > VRT_SAH_BIG) {
            const int n_chunks = (size + VRT_SAH_CHUNK - 1) / VRT_SAH_CHUNK;
            const int chunk0 = atomicAdd(&counts[2], n_chunks);
            const int nt = atomicAdd(&counts[1], 1);
            BigTask* N = next_tasks + nt;
            N->b = b; N->e = e; N->idx = child_idx; N->parent = idx; N->chunk0 = chunk0; N->n_chunks = n_chunks;
            const unsigned* cs = cen + 6 * side;
            N->cb[0] = ~cs[0]; N->cb[1] = ~cs[1]; N->cb[2] = ~cs[2];
            N->cb[3] = cs[3]; N->cb[4] = cs[4]; N->cb[5] = cs[5];
            for (int j = 0; j < n_chunks; ++j) next_chunk_task[chunk0 + j] = nt;
        } else {
            sah_push_child(b, e, child_idx, idx, out_q, out_small, counts);
        }
    }
}

// Primitive boxes with the primitive id in the fourth word of lo: the records the build moves around.
__global__ __launch_bounds__(256) void k_sah_refs(int n, const float4* __restrict__ pbox, float4* __restrict__ refs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 lo = pbox[2 * (size_t)i], hi = pbox[2 * (size_t)i + 1];
    lo.w = __int_as_float(i);   // (the id's bit pattern; never used as a number)
    refs[2 * (size_t)i] = lo; refs[2 * (size_t)i + 1] = hi;
}

// Writes every binary node's parent index into word 2 of its fourth float4 (-1 at the root): the
// stackless walk that finishes a query whose traversal stack overflowed climbs with it.
__global__ __launch_bounds__(256) void k_store_parents(int n_inner, const int* __restrict__ parent, float4* __restrict__ nodes)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_inner) return;
    nodes[4 * (size_t)i + 3].z = __int_as_float(parent[i]);
}

} // namespace

// ------------------------------------------------------------------ binned SAH build on the host
//
// Top-down build by the surface-area heuristic with 16 bins per axis over the primitives'
// centroids (Wald 2007), one primitive per leaf, for scenes whose tree is built once and traced
// many times (large scenes on up to 16 host threads): the padded primitive boxes come from k_prim_bounds, the binary nodes, parent and
// leaf-parent arrays go back to the device in the layout k_ploc_emit writes, and the 4-wide
// collapse runs on them as on any other binary tree.  Node 0 is the root (preorder numbering).
static inline float int_bits_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
#ifndef VRT_SAH_BINS
#define VRT_SAH_BINS 16
#endif
struct SahRef { float lo[3], hi[3], c[3]; int id; };
struct SahBox {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    void grow(const float* l, const float* h) { for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], l[k]); hi[k] = fmaxf(hi[k], h[k]); } }
    float area() const
    {
        const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x < 0.f ? 0.f : x * y + y * z + z * x;
    }
};
struct SahBuild {
    std::vector<SahRef> refs;
    std::vector<float4> nodes;
    std::vector<int> parent, leaf_parent;

    // Builds refs[b, e) into the inner nodes idx .. idx + (e - b) - 2 (a subtree over m primitives
    // has m - 1 inner nodes, so preorder numbers follow from the range and need no shared
    // counter: subtrees can be built by different threads and the tree is the same whoever
    // builds which).  Returns the child code and the box.  spawn: levels that may still fork.
    int build(int b, int e, int parent_idx, int idx, SahBox& box, int spawn, int depth = 0)
    {
        if (e - b == 1) {
            const SahRef& r = refs[b];
            leaf_parent[r.id] = parent_idx;
            box.grow(r.lo, r.hi);
            return ~r.id;
        }
        parent[idx] = parent_idx;
        float cl[3] = {INFINITY, INFINITY, INFINITY}, ch[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = b; i < e; ++i)
            for (int k = 0; k < 3; ++k) { cl[k] = fminf(cl[k], refs[i].c[k]); ch[k] = fmaxf(ch[k], refs[i].c[k]); }
        constexpr int NB = VRT_SAH_BINS;
        float best = INFINITY;
        int best_axis = -1, best_bin = 0;
        // (beyond 48 levels only median splits: a contrived scene cannot make the tree, and the
        // recursion, deeper than 48 + log2 n)
        for (int k = 0; k < 3 && depth < 48; ++k) {
            const float ext = ch[k] - cl[k];
            if (!(ext > 0.f)) continue;
            const float scale = NB / ext;
            SahBox bins[NB];
            int cnt[NB] = {0};
            for (int i = b; i < e; ++i) {
                int q = (int)((refs[i].c[k] - cl[k]) * scale);
                q = q < 0 ? 0 : (q >= NB ? NB - 1 : q);
                bins[q].grow(refs[i].lo, refs[i].hi);
                cnt[q]++;
            }
            float right_area[NB];
            int right_cnt[NB];
            SahBox acc;
            int n_acc = 0;
            for (int q = NB - 1; q > 0; --q) {
                if (cnt[q]) acc.grow(bins[q].lo, bins[q].hi);
                n_acc += cnt[q];
                right_area[q] = acc.area(); right_cnt[q] = n_acc;
            }
            SahBox left;
            int n_left = 0;
            for (int q = 0; q < NB - 1; ++q) {
                if (cnt[q]) left.grow(bins[q].lo, bins[q].hi);
                n_left += cnt[q];
                if (n_left == 0 || right_cnt[q + 1] == 0) continue;
                const float cost = left.area() * n_left + right_area[q + 1] * right_cnt[q + 1];
                if (cost < best) { best = cost; best_axis = k; best_bin = q; }
            }
        }
        int mid;
        if (best_axis < 0) {
            // all centroids coincide, or the depth bound: halve the range along its longest axis
            int k = 0;
            if (ch[1] - cl[1] > ch[k] - cl[k]) k = 1;
            if (ch[2] - cl[2] > ch[k] - cl[k]) k = 2;
            mid = (b + e) / 2;
            std::nth_element(refs.begin() + b, refs.begin() + mid, refs.begin() + e, [k](const SahRef& x, const SahRef& y) {
                return x.c[k] < y.c[k] || (x.c[k] == y.c[k] && x.id < y.id);
            });
        } else {
            const int k = best_axis;
            const float scale = NB / (ch[k] - cl[k]);
            const float base = cl[k];
            auto it = std::partition(refs.begin() + b, refs.begin() + e, [=](const SahRef& r) {
                int q = (int)((r.c[k] - base) * scale);
                q = q < 0 ? 0 : (q >= NB ? NB - 1 : q);
                return q <= best_bin;
            });
            mid = (int)(it - refs.begin());
            if (mid == b || mid == e) mid = (b + e) / 2;
        }
        SahBox lb, rb;
        int c0, c1;
        const int left_idx = idx + 1, right_idx = idx + (mid - b);
        if (spawn > 0 && e - b > 32768) {
            std::thread t([&] { c0 = build(b, mid, idx, left_idx, lb, spawn - 1, depth + 1); });
            c1 = build(mid, e, idx, right_idx, rb, spawn - 1, depth + 1);
            t.join();
        } else {
            c0 = build(b, mid, idx, left_idx, lb, 0, depth + 1);
            c1 = build(mid, e, idx, right_idx, rb, 0, depth + 1);
        }
        float4* nd = nodes.data() + 4 * (size_t)idx;
        nd[0] = make_float4(lb.lo[0], lb.lo[1], lb.lo[2], lb.hi[0]);
        nd[1] = make_float4(lb.hi[1], lb.hi[2], rb.lo[0], rb.lo[1]);
        nd[2] = make_float4(rb.lo[2], rb.hi[0], rb.hi[1], rb.hi[2]);
        nd[3] = make_float4(int_bits_as_float(c0), int_bits_as_float(c1), 0.f, 0.f);
        box.grow(lb.lo, lb.hi); box.grow(rb.lo, rb.hi);
        return idx;
    }
};

static hipError_t build_sah_host(int n, const float4* d_pbox, float4* d_nodes, int* d_parent, int* d_leaf_parent,
                                 hipStream_t stream)
{
    std::vector<float4> pbox(2 * (size_t)n);
    hipError_t e = hipMemcpyAsync(pbox.data(), d_pbox, pbox.size() * sizeof(float4), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    SahBuild B;
    B.refs.resize(n);
    for (int i = 0; i < n; ++i) {
        SahRef& r = B.refs[i];
        const float4 lo = pbox[2 * (size_t)i], hi = pbox[2 * (size_t)i + 1];
        r.lo[0] = lo.x; r.lo[1] = lo.y; r.lo[2] = lo.z; r.hi[0] = hi.x; r.hi[1] = hi.y; r.hi[2] = hi.z;
        for (int k = 0; k < 3; ++k) r.c[k] = 0.5f * (r.lo[k] + r.hi[k]);
        r.id = i;
    }
    B.nodes.resize(4 * (size_t)(n - 1));
    B.parent.assign(n - 1, -1);
    B.leaf_parent.assign(n, -1);
    SahBox root;
    B.build(0, n, -1, 0, root, 4);   // up to 16 threads near the root
    if ((e = hipMemcpyAsync(d_nodes, B.nodes.data(), B.nodes.size() * sizeof(float4), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(d_parent, B.parent.data(), (size_t)(n - 1) * 4, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(d_leaf_parent, B.leaf_parent.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    return hipStreamSynchronize(stream);   // the host vectors go out of scope
}

static hipError_t build_sah_device(int n, const float4* d_pbox, const unsigned* d_cb, float4* d_nodes, int* d_parent,
                                   int* d_leaf_parent, hipStream_t stream)
{
    float4 *ref_a = nullptr, *ref_b = nullptr;
    SahTask *q_a = nullptr, *q_b = nullptr;     // tasks a wave splits
    SahTask *qs_a = nullptr, *qs_b = nullptr;   // tasks of at most VRT_SAH_SMALL primitives, four to a wave
    int* counts = nullptr;   // next level: wave-sized tasks, large tasks, chunks of the large tasks, small tasks
    // large tasks (more than VRT_SAH_BIG primitives, so at most n / VRT_SAH_BIG per level) and their chunks
    const int max_big = n / VRT_SAH_BIG + 2, max_chunks = n / VRT_SAH_CHUNK + max_big + 2;
    BigTask *big_a = nullptr, *big_b = nullptr;
    int *chunk_a = nullptr, *chunk_b = nullptr, *left_count = nullptr;
    BigSplit* split = nullptr;
    unsigned* level_state = nullptr;   // per large task: bins, then the children's boxes; zeroed per level
    const size_t bins_words = (size_t)max_big * VRT_SAH_BIN_WORDS;
    const size_t state_bytes = bins_words * sizeof(unsigned) + (size_t)max_big * sizeof(BigOut);
    hipError_t e = hipSuccess;
#define SCHK(expr) do { e = (expr); if (e != hipSuccess) goto sah_done; } while (0)
    SCHK(hipMalloc(&ref_a, 2 * (size_t)n * sizeof(float4)));
    SCHK(hipMalloc(&ref_b, 2 * (size_t)n * sizeof(float4)));
    SCHK(hipMalloc(&q_a, (size_t)n * sizeof(SahTask)));
    SCHK(hipMalloc(&q_b, (size_t)n * sizeof(SahTask)));
    SCHK(hipMalloc(&qs_a, (size_t)n * sizeof(SahTask)));
    SCHK(hipMalloc(&qs_b, (size_t)n * sizeof(SahTask)));
    SCHK(hipMalloc(&counts, 4 * sizeof(int)));
    SCHK(hipMalloc(&big_a, (size_t)max_big * sizeof(BigTask)));
    SCHK(hipMalloc(&big_b, (size_t)max_big * sizeof(BigTask)));
    SCHK(hipMalloc(&chunk_a, (size_t)max_chunks * sizeof(int)));
    SCHK(hipMalloc(&chunk_b, (size_t)max_chunks * sizeof(int)));
    SCHK(hipMalloc(&left_count, (size_t)max_chunks * sizeof(int)));
    SCHK(hipMalloc(&split, (size_t)max_big * sizeof(BigSplit)));
    SCHK(hipMalloc(&level_state, state_bytes));
    hipLaunchKernelGGL(k_sah_refs, dim3((n + 255) / 256), dim3(256), 0, stream, n, d_pbox, ref_a);
    SCHK(hipGetLastError());
    {
        int n_in = 0, n_big = 0, n_chunks = 0, n_small = 0;
        if (n > VRT_SAH_BIG) {
            n_big = 1; n_chunks = (n + VRT_SAH_CHUNK - 1) / VRT_SAH_CHUNK;
            hipLaunchKernelGGL(k_big_root, dim3((n_chunks + 255) / 256), dim3(256), 0, stream, n, d_cb, big_a, chunk_a);
            SCHK(hipGetLastError());
        } else {
            const SahTask root = {0, n, 0, -1};
            SCHK(hipMemcpyAsync(n <= VRT_SAH_SMALL ? qs_a : q_a, &root, sizeof root, hipMemcpyHostToDevice, stream));
            if (n <= VRT_SAH_SMALL) n_small = 1; else n_in = 1;
        }
        for (int level = 0; (n_in > 0 || n_big > 0 || n_small > 0) && level < 4096; ++level) {
            int h[4] = {0, 0, 0, 0};
            SCHK(hipMemsetAsync(counts, 0, 4 * sizeof(int), stream));
            if (n_big > 0) {
                if (n_big > max_big || n_chunks > max_chunks) { e = hipErrorUnknown; goto sah_done; }   // cannot happen: tasks are disjoint ranges
                unsigned* bins = level_state;
                BigOut* outs = (BigOut*)(level_state + bins_words);
                SCHK(hipMemsetAsync(level_state, 0, state_bytes, stream));
                // a launch that fails (a grid too large, say) must end the build here: the host would
                // otherwise read the counts of a level that never ran
                hipLaunchKernelGGL(k_big_bin, dim3(n_chunks), dim3(256), 0, stream, level, ref_a, big_a, chunk_a, bins);
                SCHK(hipGetLastError());
                hipLaunchKernelGGL(k_big_count, dim3(n_chunks), dim3(256), 0, stream, level, ref_a, big_a, chunk_a, bins, split,
                                   left_count);
                SCHK(hipGetLastError());
                hipLaunchKernelGGL(k_big_scatter, dim3(n_chunks), dim3(256), 0, stream, ref_a, ref_b, big_a, chunk_a, split,
                                   left_count, outs);
                SCHK(hipGetLastError());
                hipLaunchKernelGGL(k_big_finish, dim3((n_big + 63) / 64), dim3(64), 0, stream, n_big, big_a, split, outs, d_nodes,
                                   d_parent, d_leaf_parent, q_b, qs_b, counts, big_b, chunk_b);
                SCHK(hipGetLastError());
            }
            if (n_in > 0) {
                hipLaunchKernelGGL((k_sah_level<64>), dim3((n_in + 3) / 4), dim3(256), 0, stream, level, ref_a, ref_b, q_a, n_in, q_b,
                                   qs_b, counts, d_nodes, d_parent, d_leaf_parent);
                SCHK(hipGetLastError());
            }
            if (n_small > 0) {
                hipLaunchKernelGGL((k_sah_level<16>), dim3((n_small + 15) / 16), dim3(256), 0, stream, level, ref_a, ref_b, qs_a,
                                   n_small, q_b, qs_b, counts, d_nodes, d_parent, d_leaf_parent);
                SCHK(hipGetLastError());
            }
            SCHK(hipMemcpyAsync(h, counts, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
            SCHK(hipStreamSynchronize(stream));
            n_in = h[0]; n_big = h[1]; n_chunks = h[2]; n_small = h[3];
            float4* tr = ref_a; ref_a = ref_b; ref_b = tr;
            SahTask* tq = q_a; q_a = q_b; q_b = tq;
            tq = qs_a; qs_a = qs_b; qs_b = tq;
            BigTask* tb = big_a; big_a = big_b; big_b = tb;
            int* tc = chunk_a; chunk_a = chunk_b; chunk_b = tc;
        }
        if (n_in > 0 || n_big > 0 || n_small > 0) e = hipErrorUnknown;
    }
sah_done:
#undef SCHK
    hipFree(ref_a); hipFree(ref_b); hipFree(q_a); hipFree(q_b); hipFree(qs_a); hipFree(qs_b); hipFree(counts);
    hipFree(big_a); hipFree(big_b); hipFree(chunk_a); hipFree(chunk_b); hipFree(left_count); hipFree(split); hipFree(level_state);
    return e;
}

// ------------------------------------------------------------------ dispatch order of the frame kernel's tiles

namespace {
__global__ __launch_bounds__(256) void k_iota(int* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int)i;
}
} // namespace

size_t vrt_tile_sort_bytes(size_t n)
{
    size_t bytes = 0;
    unsigned* k = nullptr; int* v = nullptr;
    if (rocprim::radix_sort_pairs_desc(nullptr, bytes, k, k, v, v, n, 0, 32, (hipStream_t)0) != hipSuccess) return 0;
    return bytes ? bytes : 16;
}

hipError_t vrt_fill_iota(int* iota, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, iota, n);
    return hipGetLastError();
}

hipError_t vrt_tile_sort(const unsigned* cost, unsigned* keys_tmp, const int* iota, int* order, size_t n, void* tmp,
                         size_t tmp_bytes, hipStream_t stream)
{
    // a stable sort: tiles of equal cost keep their ascending order, so the result is deterministic
    return rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, cost, keys_tmp, iota, order, n, 0, 32, stream);
}

#define BCHK(expr) do { e = (expr); if (e != hipSuccess) goto done; } while (0)

hipError_t vrt_build_lbvh(const DevScene& S, float4* nodes, float4* wide, int* depth_out, float* ms_out,
                          int builder, bool parity_collapse, hipStream_t stream)
{
    const int n = S.n_prims;
    hipError_t e = hipSuccess;
    float4 *pbox = nullptr, *parked = nullptr;
    unsigned *cb = nullptr, *vals = nullptr, *vals2 = nullptr, *arrived = nullptr;
    unsigned long long *keys = nullptr, *keys2 = nullptr;
    int2* child = nullptr;
    int *parent = nullptr, *leaf_parent = nullptr, *depth = nullptr;
    void* tmp = nullptr; size_t tmp_bytes = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const int blocks = (n + 255) / 256;
    const unsigned cb_init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    if (n < 2) return hipErrorInvalidValue;

    BCHK(hipEventCreate(&e0)); BCHK(hipEventCreate(&e1));
    BCHK(hipMalloc(&pbox, 2 * (size_t)n * sizeof(float4)));
    BCHK(hipMalloc(&parked, 4 * (size_t)n * sizeof(float4)));
    BCHK(hipMalloc(&cb, 6 * sizeof(unsigned)));
    BCHK(hipMalloc(&keys, (size_t)n * 8)); BCHK(hipMalloc(&keys2, (size_t)n * 8));
    BCHK(hipMalloc(&vals, (size_t)n * 4)); BCHK(hipMalloc(&vals2, (size_t)n * 4));
    BCHK(hipMalloc(&arrived, (size_t)(n - 1) * 4));
    BCHK(hipMalloc(&child, (size_t)(n - 1) * sizeof(int2)));
    BCHK(hipMalloc(&parent, (size_t)(n - 1) * 4));
    BCHK(hipMalloc(&leaf_parent, (size_t)n * 4));
    BCHK(hipMalloc(&depth, 4));
    BCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys2, vals, vals2, (size_t)n, 0, 63, stream));
    BCHK(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));

    BCHK(hipMemcpyAsync(cb, cb_init, sizeof cb_init, hipMemcpyHostToDevice, stream));
    BCHK(hipMemsetAsync(arrived, 0, (size_t)(n - 1) * 4, stream));
    BCHK(hipMemsetAsync(depth, 0, 4, stream));
    BCHK(hipEventRecord(e0, stream));
    hipLaunchKernelGGL(k_prim_bounds, dim3(blocks), dim3(256), 0, stream, S, pbox, cb);
    if (builder == 2) {
        BCHK(build_sah_host(n, pbox, nodes, parent, leaf_parent, stream));
    } else if (builder == 3) {
        BCHK(build_sah_device(n, pbox, cb, nodes, parent, leaf_parent, stream));
    } else {
    hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(256), 0, stream, n, pbox, cb, keys, vals);
    BCHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (size_t)n, 0, 63, stream));
    {
        const bool ploc = builder != 1;
        if (ploc) {
            // cluster arrays, ping-pong; `parked` (4 float4 per node) holds the four box arrays
            int *code_a = nullptr, *code_b = nullptr, *nn = nullptr;
            unsigned long long *flags = nullptr, *scan = nullptr;
            void* stmp = nullptr; size_t stmp_bytes = 0;
            float4 *lo_a = parked, *hi_a = parked + n, *lo_b = parked + 2 * (size_t)n, *hi_b = parked + 3 * (size_t)n;
            hipError_t pe = hipSuccess;
#define PCHK(expr) do { pe = (expr); if (pe != hipSuccess) goto ploc_done; } while (0)
            PCHK(hipMalloc(&code_a, (size_t)n * 4)); PCHK(hipMalloc(&code_b, (size_t)n * 4)); PCHK(hipMalloc(&nn, (size_t)n * 4));
            PCHK(hipMalloc(&flags, (size_t)n * 8)); PCHK(hipMalloc(&scan, (size_t)n * 8));
            PCHK(rocprim::exclusive_scan(nullptr, stmp_bytes, flags, scan, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), stream));
            PCHK(hipMalloc(&stmp, stmp_bytes ? stmp_bytes : 16));
            hipLaunchKernelGGL(k_ploc_init, dim3(blocks), dim3(256), 0, stream, n, vals2, pbox, code_a, lo_a, hi_a);
            {
                int m = n, left = n - 1;
                while (m > 1) {
                    int mb = (m + 255) / 256;
                    hipLaunchKernelGGL(k_ploc_nn, dim3(mb), dim3(256), 0, stream, m, lo_a, hi_a, nn);
                    hipLaunchKernelGGL(k_ploc_flags, dim3(mb), dim3(256), 0, stream, m, nn, flags);
                    PCHK(rocprim::exclusive_scan(stmp, stmp_bytes, flags, scan, 0ull, (size_t)m,
                                                 rocprim::plus<unsigned long long>(), stream));
                    hipLaunchKernelGGL(k_ploc_emit, dim3(mb), dim3(256), 0, stream, m, left, nn, flags, scan, code_a, lo_a, hi_a,
                                       code_b, lo_b, hi_b, nodes, parent, leaf_parent);
                    unsigned long long last_f = 0, last_s = 0;
                    PCHK(hipMemcpyAsync(&last_f, flags + (m - 1), 8, hipMemcpyDeviceToHost, stream));
                    PCHK(hipMemcpyAsync(&last_s, scan + (m - 1), 8, hipMemcpyDeviceToHost, stream));
                    PCHK(hipStreamSynchronize(stream));
                    unsigned long long tot = last_f + last_s;
                    int m2 = (int)(unsigned)tot, merges = (int)(unsigned)(tot >> 32);
                    if (merges <= 0 || m2 != m - merges) { pe = hipErrorUnknown; goto ploc_done; } // cannot happen: a mutual pair always exists
                    m = m2; left -= merges;
                    int* tc = code_a; code_a = code_b; code_b = tc;
                    float4* t4 = lo_a; lo_a = lo_b; lo_b = t4;
                    t4 = hi_a; hi_a = hi_b; hi_b = t4;
                }
            }
ploc_done:
#undef PCHK
            hipFree(code_a); hipFree(code_b); hipFree(nn); hipFree(flags); hipFree(scan); hipFree(stmp);
            BCHK(pe);
        } else {
            hipLaunchKernelGGL(k_hierarchy, dim3(blocks), dim3(256), 0, stream, n, keys2, child, parent, leaf_parent);
            hipLaunchKernelGGL(k_fit, dim3(blocks), dim3(256), 0, stream, n, vals2, pbox, child, parent, leaf_parent, arrived,
                               parked, nodes);
        }
    }
    }
    hipLaunchKernelGGL(k_depth, dim3(blocks), dim3(256), 0, stream, n, parent, leaf_parent, depth);
    {
        // Greedy collapse level by level; `child`/`arrived` are free by now and serve as the two
        // level queues.  (The every-second-level collapse remains as a study option.)
        int* q0 = (int*)child;
        int* q1 = q0 + (n - 1);
        int* qcount = (int*)arrived;
        int n_in = 1, root = 0;
        bool greedy = !parity_collapse;
        BCHK(hipMemcpyAsync(q0, &root, sizeof(int), hipMemcpyHostToDevice, stream));
        while (greedy && n_in > 0) {
            BCHK(hipMemsetAsync(qcount, 0, sizeof(int), stream));
            hipLaunchKernelGGL(k_collapse4_level, dim3((n_in + 255) / 256), dim3(256), 0, stream, nodes, wide, q0, n_in, q1,
                               qcount);
            BCHK(hipMemcpyAsync(&n_in, qcount, sizeof(int), hipMemcpyDeviceToHost, stream));
            BCHK(hipStreamSynchronize(stream));
            int* tq = q0; q0 = q1; q1 = tq;
        }
        if (!greedy)
            hipLaunchKernelGGL(k_collapse4, dim3((n - 1 + 255) / 256), dim3(256), 0, stream, n - 1, nodes, parent, wide);
        hipLaunchKernelGGL(k_store_parents, dim3((n - 1 + 255) / 256), dim3(256), 0, stream, n - 1, parent, nodes);
    }
    BCHK(hipEventRecord(e1, stream));
    BCHK(hipGetLastError());
    BCHK(hipStreamSynchronize(stream));
    BCHK(hipMemcpy(depth_out, depth, 4, hipMemcpyDeviceToHost));
    BCHK(hipEventElapsedTime(ms_out, e0, e1));
done:
    hipFree(pbox); hipFree(parked); hipFree(cb); hipFree(keys); hipFree(keys2); hipFree(vals); hipFree(vals2);
    hipFree(arrived); hipFree(child); hipFree(parent); hipFree(leaf_parent); hipFree(depth); hipFree(tmp);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    return e;
}
