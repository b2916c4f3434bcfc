// Device functions of the tracing loop (gfx950, wave64).  Compiled with -ffp-contract=off:
// every multiply and add rounds separately, in the order written, because hit decisions must
// be bit-identical to the reference's float arithmetic.
//
// Citations: /root/reference/VisualRayTrace/Assets/Scripts/RayTracer/<file>:<lines>.
#pragma once
#include "vrt_device.h"
#include "vrt_exact_math.h"

// VRT_USE_SM selects the tracer of the frame kernel: 0 the straight-line tracer (trace_sample_dev),
// 1 the single-loop state machine (trace_samples_sm), 2 the state machine for tree scenes only.
// Measured on MI355X with one sample per lane and round in both: the state machine, whose one
// traversal site lets sparse waves trace four lanes per ray, runs 62 % of the lanes per vector
// instruction against 51 % on the 10k-sphere frame but executes 8 % more instructions: 43.7 ms
// against 41.9 ms (4.14 against 3.64 ms on the 1M-triangle frame), and it is 55 % slower on the
// brute-force teaching scene.  The frame kernel therefore stays on the straight-line tracer; the
// recorded-pixel kernel runs the state machine.
#ifndef VRT_USE_SM
#define VRT_USE_SM 0
#endif


typedef float v2f __attribute__((ext_vector_type(2))); // maps to v_pk_add_f32 / v_pk_mul_f32
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
#ifndef VRT_STUDY_FAST_DIV
#define VRT_STUDY_FAST_DIV 0
#endif
// VRT_SHORT_MATH=1: square roots, reciprocals and divisions of the vector helpers take the short forms
// of vrt_exact_math.h (the compiler's own expansions without their scaling and fix-up steps: 9 instead
// of 16, 3 instead of 11 and 5 instead of 11 instructions, the reciprocal shared by the components
// of a vector) whenever no operand of any lane of the wave needs those steps, which a handful of
// instructions decide; otherwise the wave takes the compiler's form.  Same values either way (all GPU
// tests pass with it), but not faster on MI355X: teaching scene 0.1112 -> 0.1116 ms, 10k spheres 24.6 ->
// 25.1 ms, 1M triangles 2.07 -> 2.10 ms; the test and the second code path cost what the shorter
// sequences save (with approximate forms and no test the bound is -13.6 % / -4.1 %).  A study switch, off.
#ifndef VRT_SHORT_MATH
#define VRT_SHORT_MATH 0
#endif
#if VRT_STUDY_FAST_DIV   // timing study only (results differ): what the IEEE divisions and square roots of the vector helpers cost
__device__ __forceinline__ f3 operator/(f3 a, float s) { const float r = __builtin_amdgcn_rcpf(s); return mk3(a.x * r, a.y * r, a.z * r); }
#elif VRT_SHORT_MATH
// a.x / s, a.y / s, a.z / s, correctly rounded.  Short form when, for every lane that is here, s lies in
// [2^-40, 2^40] and every component of a is zero or lies in that range too: then the division needs
// no scaling, and one refined reciprocal of s serves all three.  (w maps |a| to 2|a|-2 as integers, which
// sends zero to the top of the range: a component is too small iff 0 < |a| < 2^-40.)
__device__ __forceinline__ f3 operator/(f3 a, float s)
{
    const unsigned wx = (__float_as_uint(a.x) << 1) - 2u, wy = (__float_as_uint(a.y) << 1) - 2u, wz = (__float_as_uint(a.z) << 1) - 2u;
    const unsigned wmin = min(min(wx, wy), wz);
    const float amax = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fabsf(a.z));
    const bool ok = wmin >= ((87u << 23) << 1) - 2u && amax <= 0x1p40f && __builtin_amdgcn_fmed3f(s, 0x1p-40f, 0x1p40f) == s;
    if (__all(ok)) {
        const float r = vrt_rcp_short(s);
        return mk3(vrt_div_short(a.x, s, r), vrt_div_short(a.y, s, r), vrt_div_short(a.z, s, r));
    }
    return mk3(a.x / s, a.y / s, a.z / s);
}
#else
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
#endif
// UnityEngine.Vector3.Dot / Cross / Magnitude / Normalize
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// sqrtf(x): the short form where x needs no scaling in any lane that is here.
__device__ __forceinline__ float sqrt_ieee(float x)
{
#if VRT_SHORT_MATH && !VRT_STUDY_FAST_DIV
    if (__all(__builtin_amdgcn_fmed3f(x, 0x1p-80f, 0x1p80f) == x)) return vrt_sqrt_short(x);
#endif
    return sqrtf(x);
}
// 1 / x per component, correctly rounded; the short form where every component of every lane lies in +-[2^-40, 2^40].
__device__ __forceinline__ f3 rcp3_ieee(f3 d)
{
#if VRT_SHORT_MATH && !VRT_STUDY_FAST_DIV
    const float lo = fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z)), hi = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
    if (__all(lo >= 0x1p-40f && hi <= 0x1p40f)) return mk3(vrt_rcp_short(d.x), vrt_rcp_short(d.y), vrt_rcp_short(d.z));
#endif
    return mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);
}
#if VRT_STUDY_FAST_DIV
__device__ __forceinline__ float mag3(f3 a) { return __builtin_amdgcn_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
#else
__device__ __forceinline__ float mag3(f3 a) { return sqrt_ieee(a.x * a.x + a.y * a.y + a.z * a.z); }
#endif
__device__ __forceinline__ f3 normalize3(f3 a)
{
    float m = mag3(a);
    if (m > 1e-5f) return a / m;
    return mk3(0.f, 0.f, 0.f);
}
__device__ __forceinline__ bool is_zero3(f3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }
__device__ __forceinline__ f3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }

// Load of scene data (primitives, tree nodes, materials, lights: written by the host before the
// launch, never by a kernel) through the constant address space.  A wave-uniform address then
// always becomes a scalar load; from the global address space it only does while the compiler can
// prove that nothing in the kernel may write that memory, which a single call or a store through a
// generic pointer anywhere in the kernel ends (every node of the wave-uniform traversal and every
// primitive of the brute-force loop then comes through the vector memory path, +10 % frame time).
#define VRT_CONST_AS __attribute__((address_space(4)))
__device__ __forceinline__ float4 ldc(const float4* p)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = *(const VRT_CONST_AS v4f*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int ldc(const int* p) { return *(const VRT_CONST_AS int*)p; }
__device__ __forceinline__ DevLight ldc(const DevLight* p)
{
    static_assert(sizeof(DevLight) % 4 == 0, "DevLight is copied word by word");
    DevLight l;
    int* w = (int*)&l;
    const VRT_CONST_AS int* q = (const VRT_CONST_AS int*)p;
#pragma unroll
    for (int k = 0; k < (int)(sizeof(DevLight) / 4); ++k) w[k] = q[k];
    return l;
}

// Same mixing function as vrt_pixel_set on the host.
__device__ __forceinline__ unsigned pixel_set_dev(unsigned seed, unsigned pixel, unsigned n_sets)
{
    if (n_sets <= 1u) return 0u;
    unsigned x = pixel ^ seed;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x % n_sets;
}

// ------------------------------------------------------------------ colliders
// Semantics (Unity Physics.Raycast as documented): closest collider with 0 <= t <= maxd; a
// collider containing the origin is not reported; triangles are single sided; equal t goes to
// the lowest primitive id.

__device__ __forceinline__ bool hit_sphere(float4 s, f3 o, f3 d, float maxd, float& t)
{
    f3 oc = o - xyz(s);
    float b = dot3(oc, d);
    float cc = dot3(oc, oc) - s.w * s.w;
    float disc = b * b - cc;
    if (!(cc > 0.f && b < 0.f && disc >= 0.f)) return false;
    t = -b - sqrtf(disc);
    return t >= 0.f && t <= maxd;
}

// The same test for a ray leaving the camera origin: oc = origin - centre and
// cc = oc.oc - r*r do not depend on the pixel and come precomputed (same float operations,
// same order, done once on the host).
__device__ __forceinline__ bool hit_sphere_pre(float4 pre, f3 d, float maxd, float& t)
{
    float b = dot3(xyz(pre), d);
    float cc = pre.w;
    float disc = b * b - cc;
    if (!(cc > 0.f && b < 0.f && disc >= 0.f)) return false;
    t = -b - sqrtf(disc);
    return t >= 0.f && t <= maxd;
}

// Box whose axes are signed world axes: local axis k_j = +-e_j.  The local quantities of the
// general test are then exact sign flips of world ones (p.A = +-p_j, d.A = +-d_j,
// 1/(d.A) = +-(1/d_j)), so the slabs can be evaluated on world coordinates with one reciprocal
// per ray component, bit for bit like box_slabs: a sign flip swaps t1 and t2 together with the
// sign of the direction, which leaves the near/far values unchanged.  Only the face code needs
// the local view: aux = k_j | (entered through the +axis face ? 4 : 0), and among equal
// near values the lowest local axis wins as in box_slabs' ascending loop.
// p = origin - centre, inv = 1/d per component, fl = k_0 | k_1<<2 | k_2<<4 | flip_j<<(8+j).
__device__ __forceinline__ bool hit_aab(f3 p, f3 d, f3 inv, float h0, float h1, float h2, int fl, float maxd,
                                        float& t, int& aux)
{
    bool inside = (fabsf(p.x) <= h0) && (fabsf(p.y) <= h1) && (fabsf(p.z) <= h2);
    if (inside) return false;
    float tn = -INFINITY, tf = INFINITY;
    int ax = -1, kb = -1;
#define VRT_WSLAB(J, PJ, DJ, INV, H)                                                   \
    if (DJ == 0.f) { if (fabsf(PJ) > H) return false; }                                \
    else {                                                                             \
        const int kj = (fl >> (2 * J)) & 3;                                            \
        const bool flip = ((fl >> (8 + J)) & 1) != 0;                                  \
        float t1 = (-H - PJ) * INV, t2 = (H - PJ) * INV;                               \
        bool pos = DJ > 0.f;                                                           \
        float nr = pos ? t1 : t2, fr = pos ? t2 : t1;                                  \
        if (nr > tn || (nr == tn && kj < kb)) { tn = nr; kb = kj; ax = kj | ((pos == flip) ? 4 : 0); } \
        if (fr < tf) tf = fr;                                                          \
    }
    VRT_WSLAB(0, p.x, d.x, inv.x, h0)
    VRT_WSLAB(1, p.y, d.y, inv.y, h1)
    VRT_WSLAB(2, p.z, d.z, inv.z, h2)
#undef VRT_WSLAB
    if (ax < 0) return false;
    if (!(tn <= tf)) return false;
    if (!(tn >= 0.f && tn <= maxd)) return false;
    t = tn; aux = ax;
    return true;
}

// hit_aab in two parts for the brute-force loop, which needs the entry distance of every box it tests
// but the entry face only of the one box that ends up as the closest hit (and never for an
// occlusion query).  aab_slabs: the same slabs, same operations, without the face; it accepts
// exactly the rays hit_aab accepts (an axis updates tn in hit_aab iff nr > tn here, or on a tie,
// which leaves tn as it is; a ray for which no axis ever updates tn ends with tn = -inf and is
// rejected by tn >= 0 as hit_aab rejects it by its face staying unset).
__device__ __forceinline__ bool aab_slabs(f3 p, f3 d, f3 inv, float h0, float h1, float h2, float maxd, float& t)
{
    // (a form without the early exits, all selects, was measured 1 % slower on the teaching scene: it
    // needs 22 more scalar registers, which are spilled)
    bool inside = (fabsf(p.x) <= h0) && (fabsf(p.y) <= h1) && (fabsf(p.z) <= h2);
    if (inside) return false;
    float tn = -INFINITY, tf = INFINITY;
#define VRT_WSLAB(PJ, DJ, INV, H)                                                      \
    if (DJ == 0.f) { if (fabsf(PJ) > H) return false; }                                \
    else {                                                                             \
        float t1 = (-H - PJ) * INV, t2 = (H - PJ) * INV;                               \
        bool pos = DJ > 0.f;                                                           \
        float nr = pos ? t1 : t2, fr = pos ? t2 : t1;                                  \
        if (nr > tn) tn = nr;                                                          \
        if (fr < tf) tf = fr;                                                          \
    }
    VRT_WSLAB(p.x, d.x, inv.x, h0)
    VRT_WSLAB(p.y, d.y, inv.y, h1)
    VRT_WSLAB(p.z, d.z, inv.z, h2)
#undef VRT_WSLAB
    if (!(tn <= tf)) return false;
    if (!(tn >= 0.f && tn <= maxd)) return false;
    t = tn;
    return true;
}

// The face code hit_aab reports for a box the ray is known to hit: the axis of the largest entry
// distance, the lowest local axis among equal ones.
__device__ __forceinline__ int aab_face(f3 p, f3 d, f3 inv, float h0, float h1, float h2, int fl)
{
    float tn = -INFINITY;
    int ax = 0, kb = -1;
#define VRT_WFACE(J, PJ, DJ, INV, H)                                                   \
    if (DJ != 0.f) {                                                                   \
        const int kj = (fl >> (2 * J)) & 3;                                            \
        const bool flip = ((fl >> (8 + J)) & 1) != 0;                                  \
        const bool pos = DJ > 0.f;                                                     \
        const float nr = (pos ? -H - PJ : H - PJ) * INV;                               \
        if (nr > tn || (nr == tn && kj < kb)) { tn = nr; kb = kj; ax = kj | ((pos == flip) ? 4 : 0); } \
    }
    VRT_WFACE(0, p.x, d.x, inv.x, h0)
    VRT_WFACE(1, p.y, d.y, inv.y, h1)
    VRT_WFACE(2, p.z, d.z, inv.z, h2)
#undef VRT_WFACE
    return ax;
}

// Slab core of the box test on box-local coordinates: lo = origin, ld = direction, inv = 1/ld.
// aux = axis | (entered through the +axis face ? 4 : 0)
__device__ __forceinline__ bool box_slabs(float lo0, float lo1, float lo2, float ld0, float ld1, float ld2,
                                          float inv0, float inv1, float inv2, float h0, float h1, float h2,
                                          float maxd, float& t, int& aux)
{
    bool inside = (fabsf(lo0) <= h0) && (fabsf(lo1) <= h1) && (fabsf(lo2) <= h2);
    if (inside) return false;
    float tn = -INFINITY, tf = INFINITY;
    int ax = -1;
#define VRT_SLAB(K, LO, LD, INV, H)                                         \
    if (LD == 0.f) { if (fabsf(LO) > H) return false; }                     \
    else {                                                                  \
        float t1 = (-H - LO) * INV, t2 = (H - LO) * INV;                    \
        float nr = LD > 0.f ? t1 : t2, fr = LD > 0.f ? t2 : t1;             \
        if (nr > tn) { tn = nr; ax = K | (LD > 0.f ? 0 : 4); }              \
        if (fr < tf) tf = fr;                                               \
    }
    VRT_SLAB(0, lo0, ld0, inv0, h0)
    VRT_SLAB(1, lo1, ld1, inv1, h1)
    VRT_SLAB(2, lo2, ld2, inv2, h2)
#undef VRT_SLAB
    if (ax < 0) return false;
    if (!(tn <= tf)) return false;
    if (!(tn >= 0.f && tn <= maxd)) return false;
    t = tn; aux = ax;
    return true;
}

// Oriented box: centre b0.xyz, axes b1..b3.xyz, half extents b0.w, b1.w, b2.w.
__device__ __forceinline__ bool hit_box(float4 b0, float4 b1, float4 b2, float4 b3, f3 o, f3 d,
                                        float maxd, float& t, int& aux)
{
    f3 p = o - xyz(b0);
    float lo0 = dot3(p, xyz(b1)), lo1 = dot3(p, xyz(b2)), lo2 = dot3(p, xyz(b3));
    float ld0 = dot3(d, xyz(b1)), ld1 = dot3(d, xyz(b2)), ld2 = dot3(d, xyz(b3));
    return box_slabs(lo0, lo1, lo2, ld0, ld1, ld2, 1.f / ld0, 1.f / ld1, 1.f / ld2, b0.w, b1.w, b2.w, maxd, t, aux);
}

__device__ __forceinline__ bool hit_tri(f3 v0, f3 e1, f3 e2, f3 o, f3 d, float maxd, float& t)
{
    // One exit ahead of the division instead of one per rejection test: the lanes of a wave test
    // different triangles, so the early exits almost never skip anything for the wave, and each
    // costs exec-mask work and a branch on the scalar unit.  Same operations, same comparisons
    // (a NaN u or v passes its range test and fails the last one, as it always did).
    const f3 pvec = cross3(d, e2);
    const float det = dot3(e1, pvec);
    const f3 tvec = o - v0;
    const float u = dot3(tvec, pvec);
    const f3 qvec = cross3(tvec, e1);
    const float v = dot3(d, qvec);
    if (!(det > 0.f) || (u < 0.f || u > det) || (v < 0.f || u + v > det)) return false;
    t = dot3(e2, qvec) / det;
    return t >= 0.f && t <= maxd;
}

struct Hit {
    float t;
    int prim; // -1 = miss
    int aux;  // box face
};

// Tests primitive `id` and keeps it if it beats `best` by (t, id).
#ifndef VRT_KEY_ACCEPT
#define VRT_KEY_ACCEPT 1
#endif
// (skip: the primitive the ray was spawned on is tested like any other and refused here, which is one
// region of exec-mask instructions less than a test around the call)
__device__ __forceinline__ void test_prim(const DevScene& S, int id, f3 o, f3 d, float maxd, Hit& best, int skip = -1)
{
    float t; int aux = 0; bool h;
    if (id < S.n_sph) {
        h = hit_sphere(ldc(S.sph + id), o, d, maxd, t);
    } else if (id < S.n_sph + S.n_box) {
        const float4* b = S.box + 4 * (id - S.n_sph);
        h = hit_box(ldc(b), ldc(b + 1), ldc(b + 2), ldc(b + 3), o, d, maxd, t, aux);
    } else {
        const float4* tr = S.tri + 3 * (size_t)(id - S.n_sph - S.n_box);
        h = hit_tri(xyz(ldc(tr)), xyz(ldc(tr + 1)), xyz(ldc(tr + 2)), o, d, maxd, t);
    }
#if VRT_KEY_ACCEPT
    // (t, id) against (best.t, best.prim) as one 64-bit unsigned comparison instead of five float and
    // integer comparisons joined on the scalar unit: distances of hits are >= 0 and not NaN, so they
    // order like their bit patterns once -0 is made +0 (x + 0); "no hit yet" is (+inf, -1), the
    // largest key there is, and best.prim < 0 never occurs with another distance.
    if (h && id != skip) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(t + 0.f) << 32) | (unsigned)id;
        const unsigned long long bkey = ((unsigned long long)__float_as_uint(best.t + 0.f) << 32) | (unsigned)best.prim;
        if (key < bkey) { best.t = t; best.prim = id; best.aux = aux; }
    }
#else
    if (h && id != skip && (t < best.t || (t == best.t && id < best.prim) || best.prim < 0)) {
        best.t = t; best.prim = id; best.aux = aux;
    }
#endif
}

// Brute force over all primitives in id order.  Primitive data is wave-uniform, so the loads go
// through the scalar cache and the operands sit in SGPRs.  (Fetching primitive i+1 while
// testing i, and a division-free path for axis-aligned boxes, were both measured slower on
// MI355X and are not used.)
// ANY: stop at the first primitive hit with t < tlim (and t <= maxd).
// VRT_LDS_PRIMS=1 (study build) reads the primitives of the brute-force loop from a copy in LDS
// (whole block calls stage_prims once) instead of through the scalar cache into SGPRs.
#ifndef VRT_LDS_PRIMS
#define VRT_LDS_PRIMS 0
#endif
#define VRT_BRUTE_MAX 48
#if VRT_LDS_PRIMS
__shared__ float4 g_lds_sph[VRT_BRUTE_MAX], g_lds_aab[2 * VRT_BRUTE_MAX], g_lds_pre[VRT_BRUTE_MAX + 4];
__device__ __forceinline__ void stage_prims(const DevScene& S)
{
    for (int k = threadIdx.x; k < ((S.n_sph + 3) & ~3); k += blockDim.x) g_lds_sph[k] = S.sph[k];
    if (S.all_aab) for (int k = threadIdx.x; k < 2 * S.n_box; k += blockDim.x) g_lds_aab[k] = S.aab[k];
    if (S.cam_pre) for (int k = threadIdx.x; k < ((S.n_sph + 3) & ~3) + S.n_box; k += blockDim.x) g_lds_pre[k] = S.cam_pre[k];
    __syncthreads();
}
#define VRT_SPH(i) g_lds_sph[i]
#define VRT_AAB(i) g_lds_aab[i]
#define VRT_PRE(i) g_lds_pre[i]
#else
__device__ __forceinline__ void stage_prims(const DevScene&) {}
#define VRT_SPH(i) ldc(S.sph + (i))
#define VRT_AAB(i) ldc(S.aab + (i))
#define VRT_PRE(i) ldc(S.cam_pre + (i))
#endif

// CAM: the ray starts at the camera origin and S.cam_pre holds the per-primitive terms that
// depend only on the origin.
#ifndef VRT_AAB_CULL
#define VRT_AAB_CULL 1
#endif
#ifndef VRT_ANY_UNIFORM_EXIT
#define VRT_ANY_UNIFORM_EXIT 1
#endif
#ifndef VRT_SPH_GROUP
#define VRT_SPH_GROUP 2   // 1, 2 or 4
#endif
template <bool CAM>
__device__ __forceinline__ Hit intersect_brute(const DevScene& S, f3 o, f3 d, float maxd, float tlim, int skip, bool ANY)
{
    Hit best; best.t = INFINITY; best.prim = -1; best.aux = 0;
    if (is_zero3(d)) return best;
    // Spheres in groups of VRT_SPH_GROUP (the host pads the arrays to a multiple of four with spheres of
    // infinite radius, which fail cc > 0 for every ray): the loop's scalar work - counter, address, load,
    // wait, the lane-mask logic of the rejection test and its branch - is what this kernel is short
    // of (the scalar unit issues one instruction per SIMD every four clocks), and it is paid once per
    // group instead of once per sphere.
    // hit_sphere / hit_sphere_pre in two steps: most spheres are missed by every ray of a wave
    // (neighbouring pixels, or the occlusion rays of their hits), and the square root, which IEEE
    // rounding makes 16 instructions, is only taken when some lane's ray passes the rejection test.
    const int n_sph4 = (S.n_sph + 3) & ~3;
    constexpr int G = VRT_SPH_GROUP;
    for (int i = 0; i < n_sph4; i += G) {
        float b[G], disc[G];
        bool pass[G], some = false;
#pragma unroll
        for (int k = 0; k < G; ++k) {
            float cc;
            if (CAM) { const float4 pre = VRT_PRE(i + k); b[k] = dot3(xyz(pre), d); cc = pre.w; }
            else { const float4 sp = VRT_SPH(i + k); const f3 oc = o - xyz(sp); b[k] = dot3(oc, d); cc = dot3(oc, oc) - sp.w * sp.w; }
            disc[k] = b[k] * b[k] - cc;
            pass[k] = i + k != skip && cc > 0.f && b[k] < 0.f && disc[k] >= 0.f;
            some = some || pass[k];
        }
        if (!__any(some)) continue;
#pragma unroll
        for (int k = 0; k < G; ++k) {   // in id order: among equal distances the lowest id is kept
            if (G > 1 && !__any(pass[k])) continue;
            const float t = -b[k] - sqrtf(disc[k]);
#if VRT_ANY_UNIFORM_EXIT
            // an occlusion query keeps its first hit and rides along until every lane of the wave has
            // one (a lane that leaves a loop on its own costs the loop exec-mask bookkeeping every turn)
            if (pass[k] && t >= 0.f && t <= maxd && t < best.t && t < tlim && (!ANY || best.prim < 0)) {
                best.t = t; best.prim = i + k;
            }
        }
        if (ANY && !__any(best.prim < 0)) return best;
#else
            if (pass[k] && t >= 0.f && t <= maxd && t < best.t && t < tlim) {
                best.t = t; best.prim = i + k;
                if (ANY) return best;
            }
        }
#endif
    }
    if (S.all_aab) {
        // every box is aligned with the world axes: one reciprocal per ray component serves all
        const f3 inv = rcp3_ieee(d);
        // (boxes two per turn, like the spheres, measured slower: 0.0819 -> 0.0880 ms on the teaching
        // scene, whose five boxes become six and whose waves usually have to look at one of a pair)
        for (int i = 0; i < S.n_box; ++i) {
            float t;
            const float4 a0 = VRT_AAB(2 * i), a1 = VRT_AAB(2 * i + 1);
            const f3 p = CAM ? xyz(VRT_PRE(n_sph4 + i)) : o - xyz(a0);
#if VRT_AAB_CULL
            float oo = dot3(p, p), bb = dot3(p, d);
            float cs = oo - a1.w, slack = 4e-6f * oo;
            bool may = !((bb * bb - cs + slack < 0.f) || (cs > slack && bb > 0.f));
            if (!__any(may)) continue;
#endif
#if VRT_ANY_UNIFORM_EXIT
            if (i + S.n_sph != skip && (!ANY || best.prim < 0) && aab_slabs(p, d, inv, a1.x, a1.y, a1.z, maxd, t) && t < best.t && t < tlim) {
                best.t = t; best.prim = i + S.n_sph;
            }
            if (ANY && !__any(best.prim < 0)) return best;
#else
            if (i + S.n_sph != skip && aab_slabs(p, d, inv, a1.x, a1.y, a1.z, maxd, t) && t < best.t && t < tlim) {
                best.t = t; best.prim = i + S.n_sph;
                if (ANY) return best;
            }
#endif
        }
        // the entry face of the closest box, once per query instead of once per box tested
        if (!ANY && best.prim >= S.n_sph) {
            const int bi = best.prim - S.n_sph;
            const float4 a0 = S.aab[2 * bi], a1 = S.aab[2 * bi + 1];
            const f3 p = CAM ? xyz(S.cam_pre[n_sph4 + bi]) : o - xyz(a0);
            best.aux = aab_face(p, d, inv, a1.x, a1.y, a1.z, __float_as_int(a0.w));
        }
    } else {
#if VRT_ANY_UNIFORM_EXIT
        if (ANY && best.prim >= 0) return best;   // answered among the spheres: the first hit in id order stands
#endif
        for (int i = 0; i < S.n_box; ++i) {
            float t; int aux;
            const float4* b = S.box + 4 * i;
            // Conservative cull: b[3].w = squared circumradius of the box.  A ray that clears the
            // circumsphere (with 32 ulp of |oc|^2 slack for rounding) cannot touch the box; when
            // no lane of the wave can, the slab test with its three divisions is skipped.
            float4 bc = ldc(b);
            f3 oc = o - xyz(bc);
            float oo = dot3(oc, oc), bb = dot3(oc, d);
            float cs = oo - ldc(b + 3).w, slack = 4e-6f * oo;
            bool may = !((bb * bb - cs + slack < 0.f) || (cs > slack && bb > 0.f));
            if (!__any(may)) continue;
            if (i + S.n_sph != skip && hit_box(ldc(b), ldc(b + 1), ldc(b + 2), ldc(b + 3), o, d, maxd, t, aux) && t < best.t && t < tlim) {
                best.t = t; best.prim = i + S.n_sph; best.aux = aux;
                if (ANY) return best;
            }
        }
    }
#if VRT_ANY_UNIFORM_EXIT
    if (ANY && best.prim >= 0) return best;   // answered among the spheres or boxes
#endif
    for (int i = 0; i < S.n_tri; ++i) {
        float t;
        const float4* tr = S.tri + 3 * (size_t)i;
        if (i + S.n_sph + S.n_box != skip &&
            hit_tri(xyz(ldc(tr)), xyz(ldc(tr + 1)), xyz(ldc(tr + 2)), o, d, maxd, t) && t < best.t && t < tlim) {
            best.t = t; best.prim = i + S.n_sph + S.n_box;
            if (ANY) return best;
        }
    }
    return best;
}

// Conservative ray/box test for BVH nodes: NaNs from 0*inf are dropped by fminf/fmaxf and
// the far side is widened by 2 ulp so rounding can only add candidates, never lose one.
__device__ __forceinline__ bool slab_test(f3 lo, f3 hi, f3 o, f3 inv, float tmax, float& tnear)
{
    float ax = (lo.x - o.x) * inv.x, bx = (hi.x - o.x) * inv.x;
    float ay = (lo.y - o.y) * inv.y, by = (hi.y - o.y) * inv.y;
    float az = (lo.z - o.z) * inv.z, bz = (hi.z - o.z) * inv.z;
    float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
    float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    tnear = t0;
    return t0 <= t1 * 1.0000004f;
}

// Lanes per block of the translation unit that includes this header, and traversal-stack entries
// per lane: together they size the one traversal stack of a block in LDS, [entry][lane], 8-byte
// entries (conflict-free).  16 entries x 64 lanes = 8 KiB per one-wave block, so 16 such blocks
// fit a CU beside their counters.
#ifndef VRT_STACK_LANES
#define VRT_STACK_LANES 64
#endif
// VRT_CC_EARLY=1 keeps the load of a node's child codes with the loads of its planes instead of
// inside the "some child is hit" branch, where it is a second memory round trip on the way to the
// next node.  It needs four registers more; since the frame kernel stopped spilling it has them:
// 10k spheres 23.92 -> 23.73 ms, 1M triangles 2.010 -> 1.980 ms.
#ifndef VRT_CC_EARLY
#define VRT_CC_EARLY 1
#endif
// VRT_PUSH_BRANCHLESS (1: any-hit traversals, 2: closest-hit ones as well): trav_inner leaves
// its children on the stack without branches; every push is a store, and the stores that are not to
// happen go to a row past the stack (one more row of LDS).  Fewer exec-mask instructions and
// branches for more vector and LDS instructions.  Slower while the compiler's packed arithmetic kept
// the vector side busy with moves (10k spheres 24.6 -> 25.1 / 25.0 ms); since the scalar unit is the
// busier one, form 1 wins: 10k spheres 23.50 -> 23.21 ms, 1M triangles 1.935 -> 1.927 ms (form 2:
// 23.56 / 1.950 ms).
#ifndef VRT_PUSH_BRANCHLESS
#define VRT_PUSH_BRANCHLESS 1
#endif
// What T.node holds when there is nothing to enter: the code of an unused child slot, which no
// traversal ever enters or pushes.
#define VRT_NODE_NONE ((int)0x80000000)
#ifndef VRT_TRAV_VALUE_FLOW
#define VRT_TRAV_VALUE_FLOW 1
#endif
__shared__ unsigned long long g_lds_stack[(VRT_TRAV_STACK + (VRT_PUSH_BRANCHLESS ? 1 : 0)) * VRT_STACK_LANES];

// Orders two (tnear, code) candidates, nearer first.
__device__ __forceinline__ void order2(float& ta, int& ca, float& tb, int& cb)
{
    bool s = tb < ta;
    float t = s ? tb : ta; tb = s ? ta : tb; ta = t;
    int c = s ? cb : ca; cb = s ? ca : cb; ca = c;
}

// Debug build only (-DVRT_COUNT_NODES=1): adds the number of active lanes to counter word `k`
// (6: inner nodes entered, 7: primitives tested).
#ifndef VRT_COUNT_NODES
#define VRT_COUNT_NODES 0
#endif
__device__ __forceinline__ void count_lanes(const DevScene& S, int k, bool mine = true)
{
#if VRT_COUNT_NODES
    unsigned long long m = __ballot(mine);
    if (!m) return;
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1)
        atomicAdd(&S.dbg[(blockIdx.x % VRT_CTR_SLOTS) * VRT_CTR_STRIDE + k], (unsigned long long)__popcll(m));
#endif
}

// One ray on its way through the tree.
struct Trav {
    f3 o, d, inv;
    f3 oi;                  // what slab4 subtracts: the origin, or -(origin * inv) in the fused form
    float maxd, tlim, lim0; // range of the query, "nearer than" bound of an any-hit query, min of both
    int skip;               // primitive the ray was spawned on
    Hit best;
    int node;               // what to look at next: >= 0 inner node, < 0 leaf ~primitive
    int sp;                 // entries on the lane's stack
    int ovf;                // entries that did not fit the stack and were dropped (a count, not a flag: a flag
                            // that lives across divergent branches is kept as a lane mask and merged on the scalar unit)
};

__device__ __forceinline__ void trav_begin(Trav& T, f3 o, f3 d, float maxd, float tlim, int skip)
{
    T.o = o; T.d = d;
#if VRT_SLAB_FMA
    // box tests only produce candidates, so the reciprocal need not be the correctly rounded one
    T.inv = mk3(__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -1e30f, 1e30f),
                __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -1e30f, 1e30f),
                __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -1e30f, 1e30f));
    T.oi = mk3(-(o.x * T.inv.x), -(o.y * T.inv.y), -(o.z * T.inv.z));
#else
    T.inv = mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);
    T.oi = o;
#endif
    T.maxd = maxd; T.tlim = tlim; T.lim0 = fminf(maxd, tlim); T.skip = skip;
    T.best.t = INFINITY; T.best.prim = -1; T.best.aux = 0;
    T.node = 0; T.sp = 0; T.ovf = 0;
}

// Distance beyond which nothing can matter to the query any more.
__device__ __forceinline__ float trav_limit(const Trav& T) { return T.best.prim >= 0 ? T.best.t : T.lim0; }

// The four slab tests of slab_test() on one 128-byte node, children (0,1) and (2,3) as packed
// pairs (v_pk_add_f32 / v_pk_mul_f32).  Unused child slots hold boxes at +inf, which no ray passes.
struct Slab4 { float t0, t1, t2, t3; bool h0, h1, h2, h3; };
// Every kernel that walks the tree is compiled twice (see VRT_FORM in vrt_launch.h).  With
// VRT_SLAB_FMA the box test is plane * inv - origin * inv in one fused step per pair of planes
// (v_pk_fma_f32, `o` then holds -(origin * inv)) instead of (plane - origin) * inv: 12 instead of
// 24 packed instructions per node.  The fused form rounds origin * inv before the subtraction, an
// absolute error of 2^-24 |origin| |inv|; the boxes are padded by 1e-4 (1 + |plane|), so a
// candidate cannot be lost while ray origins stay within +-1600 of the coordinate origin, which
// the host checks per frame before it picks the fused kernels.  inv is clamped to +-1e30 in that
// form (trav_begin): a zero direction component would otherwise give inf - inf for a slab the ray
// runs inside of.
#ifndef VRT_SLAB_FMA
#define VRT_SLAB_FMA 0
#endif
#if VRT_SLAB_FMA
#define VRT_PLANE(P0, P1, I, C) __builtin_elementwise_fma(v2f{P0, P1}, I, C)
#else
#define VRT_PLANE(P0, P1, I, C) ((v2f{P0, P1} - C) * I)
#endif
__device__ __forceinline__ Slab4 slab4(float4 lx, float4 ly, float4 lz, float4 hx, float4 hy, float4 hz, f3 o, f3 inv,
                                       float lim)
{
    Slab4 r;
    const v2f ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
    const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
    const v2f ax01 = VRT_PLANE(lx.x, lx.y, ix, ox), ax23 = VRT_PLANE(lx.z, lx.w, ix, ox);
    const v2f bx01 = VRT_PLANE(hx.x, hx.y, ix, ox), bx23 = VRT_PLANE(hx.z, hx.w, ix, ox);
    const v2f ay01 = VRT_PLANE(ly.x, ly.y, iy, oy), ay23 = VRT_PLANE(ly.z, ly.w, iy, oy);
    const v2f by01 = VRT_PLANE(hy.x, hy.y, iy, oy), by23 = VRT_PLANE(hy.z, hy.w, iy, oy);
    const v2f az01 = VRT_PLANE(lz.x, lz.y, iz, oz), az23 = VRT_PLANE(lz.z, lz.w, iz, oz);
    const v2f bz01 = VRT_PLANE(hz.x, hz.y, iz, oz), bz23 = VRT_PLANE(hz.z, hz.w, iz, oz);
#undef VRT_PLANE
#define VRT_SLAB4(T, H, AX, BX, AY, BY, AZ, BZ)                                                       \
    {                                                                                                 \
        float n_ = fmaxf(fmaxf(fminf(AX, BX), fminf(AY, BY)), fmaxf(fminf(AZ, BZ), 0.f));             \
        float f_ = fminf(fminf(fmaxf(AX, BX), fmaxf(AY, BY)), fminf(fmaxf(AZ, BZ), lim));             \
        T = n_; H = n_ <= f_ * 1.0000004f;                                                            \
    }
    VRT_SLAB4(r.t0, r.h0, ax01.x, bx01.x, ay01.x, by01.x, az01.x, bz01.x)
    VRT_SLAB4(r.t1, r.h1, ax01.y, bx01.y, ay01.y, by01.y, az01.y, bz01.y)
    VRT_SLAB4(r.t2, r.h2, ax23.x, bx23.x, ay23.x, by23.x, az23.x, bz23.x)
    VRT_SLAB4(r.t3, r.h3, ax23.y, bx23.y, ay23.y, by23.y, az23.y, bz23.y)
#undef VRT_SLAB4
    return r;
}

#if VRT_SLAB_FMA
// slab4 with the entry planes (n*) and exit planes (f*) of the four children already told apart.
__device__ __forceinline__ Slab4 slab4_signed(float4 nx, float4 ny, float4 nz, float4 fx, float4 fy, float4 fz, f3 c,
                                              f3 inv, float lim)
{
    Slab4 r;
    const v2f ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
    const v2f cx = {c.x, c.x}, cy = {c.y, c.y}, cz = {c.z, c.z};
#define VRT_PLANE(P0, P1, I, C) __builtin_elementwise_fma(v2f{P0, P1}, I, C)
    const v2f ax01 = VRT_PLANE(nx.x, nx.y, ix, cx), ax23 = VRT_PLANE(nx.z, nx.w, ix, cx);
    const v2f bx01 = VRT_PLANE(fx.x, fx.y, ix, cx), bx23 = VRT_PLANE(fx.z, fx.w, ix, cx);
    const v2f ay01 = VRT_PLANE(ny.x, ny.y, iy, cy), ay23 = VRT_PLANE(ny.z, ny.w, iy, cy);
    const v2f by01 = VRT_PLANE(fy.x, fy.y, iy, cy), by23 = VRT_PLANE(fy.z, fy.w, iy, cy);
    const v2f az01 = VRT_PLANE(nz.x, nz.y, iz, cz), az23 = VRT_PLANE(nz.z, nz.w, iz, cz);
    const v2f bz01 = VRT_PLANE(fz.x, fz.y, iz, cz), bz23 = VRT_PLANE(fz.z, fz.w, iz, cz);
#undef VRT_PLANE
#define VRT_SLAB4(T, H, AX, BX, AY, BY, AZ, BZ)                       \
    {                                                                 \
        float n_ = fmaxf(fmaxf(AX, AY), fmaxf(AZ, 0.f));              \
        float f_ = fminf(fminf(BX, BY), fminf(BZ, lim));              \
        T = n_; H = n_ <= f_ * 1.0000004f;                            \
    }
    VRT_SLAB4(r.t0, r.h0, ax01.x, bx01.x, ay01.x, by01.x, az01.x, bz01.x)
    VRT_SLAB4(r.t1, r.h1, ax01.y, bx01.y, ay01.y, by01.y, az01.y, bz01.y)
    VRT_SLAB4(r.t2, r.h2, ax23.x, bx23.x, ay23.x, by23.x, az23.x, bz23.x)
    VRT_SLAB4(r.t3, r.h3, ax23.y, bx23.y, ay23.y, by23.y, az23.y, bz23.y)
#undef VRT_SLAB4
    return r;
}
#endif

__device__ __forceinline__ void trav_push(const DevScene& S, Trav& T, unsigned long long* my_lds, float tnear, int code)
{
#if VRT_PUSH_BRANCHLESS
    // no branch on "is there room": an entry that does not fit is written to the row past the stack
    // and dropped; the query is then finished by trav_stackless()
    const bool room = T.sp < S.stack_cap;   // VRT_TRAV_STACK unless a test shortens it
    my_lds[(room ? T.sp : VRT_TRAV_STACK) * VRT_STACK_LANES] = ((unsigned long long)__float_as_uint(tnear) << 32) | (unsigned)code;
    T.sp += room ? 1 : 0;
    T.ovf += room ? 0 : 1;
#else
    if (T.sp < S.stack_cap) {   // VRT_TRAV_STACK unless a test shortens it
        my_lds[T.sp * VRT_STACK_LANES] = ((unsigned long long)__float_as_uint(tnear) << 32) | (unsigned)code;
        T.sp++;
    } else {
        T.ovf = 1;   // dropped: the query is finished by trav_stackless()
    }
#endif
}

// The same push without a branch: `valid` says whether there is an entry at all.  An entry that is
// not to be stored (no entry, or no room: it is dropped and the query finished by trav_stackless()
// as above) is written to the row past the stack instead.  The nested branches of the four children
// cost the any-hit traversal loop more scalar mask instructions than it has vector instructions.
__device__ __forceinline__ void trav_push_if(int cap, Trav& T, unsigned long long* my_lds, bool valid, float tnear, int code)
{
    const bool store = valid && T.sp < cap;
    const int row = store ? T.sp : VRT_TRAV_STACK;
    my_lds[row * VRT_STACK_LANES] = ((unsigned long long)__float_as_uint(tnear) << 32) | (unsigned)code;
    T.ovf += (valid && !store) ? 1 : 0;
    T.sp += store ? 1 : 0;
}

// Inner-node step of the per-lane traversal of the 4-wide tree: loads the node T.node, tests its
// four child boxes, enters the nearest hit child and leaves the others on the stack as
// (tnear, code) pairs (an any-hit query enters the first hit child and keeps the others with
// distance 0).  Returns true when no child was hit, i.e. the caller has to pop; T.node is then
// VRT_NODE_NONE.
__device__ __forceinline__ bool trav_inner(const DevScene& S, Trav& T, unsigned long long* my_lds, bool any)
{
    count_lanes(S, 6);
#if VRT_SLAB_FMA
    // Entry and exit planes picked by the sign of the direction at load time, instead of loading
    // both and taking min and max: a node is lo.x, lo.y, lo.z, hi.x, hi.y, hi.z (16 bytes each, four
    // children wide) and the entry plane of an axis is lo for a positive direction, hi for a
    // negative one, 48 bytes further on.  No NaN can arise in the fused form (planes are finite or
    // +inf, inv is finite and not zero), so the distances are those of the min/max form bit for bit.
    const char* base = (const char*)S.bvh4;
    const unsigned at = (unsigned)T.node << 7;
    const unsigned ex = at + (__float_as_uint(T.inv.x) >> 31) * 48u;
    const unsigned ey = at + 16u + (__float_as_uint(T.inv.y) >> 31) * 48u;
    const unsigned ez = at + 32u + (__float_as_uint(T.inv.z) >> 31) * 48u;
    const float4 nx = *(const float4*)(base + ex), ny = *(const float4*)(base + ey), nz = *(const float4*)(base + ez);
    const float4 fx = *(const float4*)(base + (ex ^ 48u)), fy = *(const float4*)(base + (ey ^ 80u)),
                 fz = *(const float4*)(base + (ez ^ 112u));
    const float4 cc = *(const float4*)(base + (at + 96u));
    int c0 = __float_as_int(cc.x), c1 = __float_as_int(cc.y), c2 = __float_as_int(cc.z), c3 = __float_as_int(cc.w);
#if VRT_CC_EARLY
    // keeps the load of the child codes with the loads of the planes: sunk into the "some child is
    // hit" branch it is a second memory round trip on the way to the next node
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
#endif
    Slab4 r = slab4_signed(nx, ny, nz, fx, fy, fz, T.oi, T.inv, trav_limit(T));
#else
    const float4* n = S.bvh4 + 8 * (size_t)T.node;
    const float4 lx = n[0], ly = n[1], lz = n[2], hx = n[3], hy = n[4], hz = n[5], cc = n[6];
    int c0 = __float_as_int(cc.x), c1 = __float_as_int(cc.y), c2 = __float_as_int(cc.z), c3 = __float_as_int(cc.w);
    Slab4 r = slab4(lx, ly, lz, hx, hy, hz, T.oi, T.inv, trav_limit(T));
#endif
    if (any) {
#if VRT_PUSH_BRANCHLESS
        // the first hit child is entered, the hit children after it wait with distance 0, in child order
        const int cap = S.stack_cap;
        const bool b1 = r.h0, b2 = b1 || r.h1, b3 = b2 || r.h2, some = b3 || r.h3;
        T.node = r.h0 ? c0 : r.h1 ? c1 : r.h2 ? c2 : r.h3 ? c3 : VRT_NODE_NONE;
        trav_push_if(cap, T, my_lds, r.h1 && b1, 0.f, c1);
        trav_push_if(cap, T, my_lds, r.h2 && b2, 0.f, c2);
        trav_push_if(cap, T, my_lds, r.h3 && b3, 0.f, c3);
        return !some;
#else
        bool taken = false;
#define VRT_VISIT_ANY(H, C) if (H) { if (!taken) { T.node = C; taken = true; } else trav_push(S, T, my_lds, 0.f, C); }
        VRT_VISIT_ANY(r.h0, c0) VRT_VISIT_ANY(r.h1, c1) VRT_VISIT_ANY(r.h2, c2) VRT_VISIT_ANY(r.h3, c3)
#undef VRT_VISIT_ANY
        if (!taken) T.node = VRT_NODE_NONE;
        return !taken;
#endif
    }
    float t0 = r.h0 ? r.t0 : INFINITY, t1 = r.h1 ? r.t1 : INFINITY, t2 = r.h2 ? r.t2 : INFINITY, t3 = r.h3 ? r.t3 : INFINITY;
    order2(t0, c0, t1, c1); order2(t2, c2, t3, c3);
    order2(t0, c0, t2, c2); order2(t1, c1, t3, c3);
    order2(t1, c1, t2, c2);
#if VRT_PUSH_BRANCHLESS == 2
    const int cap = S.stack_cap;
    const bool some = t0 < INFINITY;   // t1..t3 are infinite too when t0 is
    T.node = some ? c0 : T.node;
    trav_push_if(cap, T, my_lds, t3 < INFINITY, t3, c3);
    trav_push_if(cap, T, my_lds, t2 < INFINITY, t2, c2);
    trav_push_if(cap, T, my_lds, t1 < INFINITY, t1, c1);
    return !some;
#else
#if VRT_TRAV_VALUE_FLOW
    // no exit for "no child hit": t1..t3 are infinite too when t0 is, so nothing is pushed then
    const bool some = t0 < INFINITY;
    T.node = some ? c0 : VRT_NODE_NONE;
    if (t3 < INFINITY) trav_push(S, T, my_lds, t3, c3);
    if (t2 < INFINITY) trav_push(S, T, my_lds, t2, c2);
    if (t1 < INFINITY) trav_push(S, T, my_lds, t1, c1);
    return !some;
#else
    if (!(t0 < INFINITY)) { T.node = VRT_NODE_NONE; return true; }
    T.node = c0;
    if (t3 < INFINITY) trav_push(S, T, my_lds, t3, c3);
    if (t2 < INFINITY) trav_push(S, T, my_lds, t2, c2);
    if (t1 < INFINITY) trav_push(S, T, my_lds, t1, c1);
    return false;
#endif
#endif
}

// Leaf step: tests primitive ~T.node.  Returns true when an any-hit query has its answer.
__device__ __forceinline__ bool trav_leaf(const DevScene& S, Trav& T, bool any, bool counted = true)
{
    const int id = ~T.node;
    count_lanes(S, 7, counted);
    test_prim(S, id, T.o, T.d, T.maxd, T.best, T.skip);
    if (any && T.best.prim >= 0) {
        if (T.best.t < T.tlim) return true;
        T.best.prim = -1; T.best.t = INFINITY;
    }
    return false;
}

// Takes the next entry that can still matter off the stack.  Returns false when the stack is empty.
__device__ __forceinline__ bool trav_pop(Trav& T, const unsigned long long* my_lds)
{
    const float lim = trav_limit(T);
    while (T.sp > 0) {
        --T.sp;
        const unsigned long long e = my_lds[T.sp * VRT_STACK_LANES];
        if (!(__uint_as_float((unsigned)(e >> 32)) > lim)) { T.node = (int)(unsigned)e; return true; }
    }
    return false;
}

// Primitives kept out of the tree (DevScene.big), tested ahead of the traversal.  Returns true
// when an any-hit query has its answer.
__device__ __forceinline__ bool trav_big(const DevScene& S, Trav& T, bool any, bool counted = true)
{
    for (int k = 0; k < S.n_big; ++k) {
        const int id = S.big[k];
        count_lanes(S, 7, counted);
        if (id != T.skip) test_prim(S, id, T.o, T.d, T.maxd, T.best);
        if (any && T.best.prim >= 0) {
            if (T.best.t < T.tlim) return true;
            T.best.prim = -1; T.best.t = INFINITY;
        }
    }
    return false;
}

// Finishes a query whose stack overflowed: a walk of the whole binary tree that needs no stack,
// only the parent index every binary node carries (word 2 of its fourth float4).  State = the
// inner node `cur` and which of its two children comes next (k = 2: both done, go up); a
// subtree is entered when the ray passes its box nearer than the best hit so far.  Every
// primitive the ray can reach is tested (again), so with candidates ranked by (t, id) the result
// is that of a traversal with an unbounded stack.  Slow (fixed child order, every way up
// re-reads nodes) and meant for the rare ray only.  Returns true when an any-hit query has its answer.
__device__ __forceinline__ bool trav_stackless(const DevScene& S, Trav& T, bool any)
{
    int cur = 0, k = 0;
    for (long long guard = 6ll * S.n_prims + 16; guard > 0; --guard) {
        const float4* n = S.bvh + 4 * (size_t)cur;
        if (k == 2) {
            if (cur == 0) break;
            const int p = __float_as_int(n[3].z);
            k = __float_as_int(S.bvh[4 * (size_t)p + 3].y) == cur ? 2 : 1;
            cur = p;
            continue;
        }
        const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
        const f3 lo = k ? mk3(n1.z, n1.w, n2.x) : mk3(n0.x, n0.y, n0.z);
        const f3 hi = k ? mk3(n2.y, n2.z, n2.w) : mk3(n0.w, n1.x, n1.y);
        const int code = __float_as_int(k ? n3.y : n3.x);
        float tn;
        count_lanes(S, 6);
        if (slab_test(lo, hi, T.o, T.inv, trav_limit(T), tn)) {
            if (code < 0) {
                T.node = code;
                if (trav_leaf(S, T, any)) return true;
                ++k;
            } else {
                cur = code; k = 0;
            }
        } else {
            ++k;
        }
    }
    return false;
}

// Stack-based traversal of the 4-wide tree, one ray per lane.  One inner step loads one 128-byte
// node and tests its four child boxes; the hit children are visited nearest first, the others
// wait on the lane's stack in LDS and are dropped at pop time when a closer hit has been found
// since.  The result does not depend on the visiting order because candidates are ranked by (t, id).
__device__ __forceinline__ bool intersect_bvh4_packet(const DevScene& S, Trav& T, bool live, bool any);

// The tree part of a query that has been through trav_begin and trav_big.
__device__ __forceinline__ void intersect_bvh4_tree(const DevScene& S, Trav& T, bool any, bool packet)
{
    unsigned long long* my_lds = g_lds_stack + threadIdx.x;
    if (packet && intersect_bvh4_packet(S, T, true, any)) return;
#if VRT_TRAV_VALUE_FLOW
    // The state of the walk is carried in vector registers only: T.node is the next thing to look at
    // or VRT_NODE_NONE ("take the next entry off the stack"; with an empty stack: done), an any-hit
    // query that has its answer empties its stack, and the entry a pop settles on is a value, not a
    // flag.  Flags that live across divergent branches (pop / found / overflowed) are lane masks in
    // scalar registers, and every merge of one is two or three instructions on the scalar unit, the
    // busier unit of these kernels.
    // every node is entered at most once; the bound only guards against a corrupt tree
    for (int guard = 2 * S.n_prims + 8; guard > 0 && T.node != VRT_NODE_NONE; --guard) {
        if (T.node >= 0) {
            trav_inner(S, T, my_lds, any);
        } else {
            const bool done = trav_leaf(S, T, any);
            T.node = VRT_NODE_NONE;
            T.sp = done ? 0 : T.sp;
            T.ovf = done ? 0 : T.ovf;
        }
        if (T.node == VRT_NODE_NONE && T.sp > 0) {
            const float lim = trav_limit(T);
            int cand;
            do {
                --T.sp;
                const unsigned long long e = my_lds[T.sp * VRT_STACK_LANES];
                cand = !(__uint_as_float((unsigned)(e >> 32)) > lim) ? (int)(unsigned)e : VRT_NODE_NONE;
            } while (cand == VRT_NODE_NONE && T.sp > 0);
            T.node = cand;
        }
    }
#else
    // every node is entered at most once; the bound only guards against a corrupt tree
    for (int guard = 2 * S.n_prims + 8; guard > 0; --guard) {
        bool pop;
        if (T.node >= 0) pop = trav_inner(S, T, my_lds, any);
        else { if (trav_leaf(S, T, any)) return; pop = true; }
        if (pop && !trav_pop(T, my_lds)) break;
    }
#endif
    if (T.ovf) trav_stackless(S, T, any);
}

// packet: the calling lanes' rays are expected to stay close to each other and try the
// wave-uniform traversal first (intersect_bvh4_packet).
__device__ __forceinline__ Hit intersect_bvh4(const DevScene& S, f3 o, f3 d, float maxd, float tlim, int skip, bool any,
                                              bool packet = false)
{
    Trav T;
    trav_begin(T, o, d, maxd, tlim, skip);
    if (is_zero3(d)) return T.best;
    if (trav_big(S, T, any)) return T.best;
    intersect_bvh4_tree(S, T, any, packet);
    return T.best;
}

// Wave-uniform traversal of the 4-wide tree for rays a wave casts together and that stay close
// to each other (the camera rays of one or a few pixels, the shadow and ambient-occlusion rays of
// their hits): all calling lanes walk the same nodes.  The node index is wave-uniform, so a node
// comes through the scalar cache into scalar registers (no vector-memory traffic at all), the
// choice of the next node is scalar arithmetic, and a leaf's primitive is loaded once for the
// wave.  A child is entered when ANY lane's ray passes its box; the stack holds, per lane, that
// lane's entry distance next to the (uniform) child code, so an entry is skipped at pop time
// once no lane can still gain from it.  Lanes test every primitive the wave reaches and rank
// candidates by (t, id), so each lane ends with exactly the hit its own traversal would find.
// Returns false (T untouched apart from the stack) when the wave's stack overflowed; the caller
// then runs the per-lane traversal.
__device__ __forceinline__ bool intersect_bvh4_packet(const DevScene& S, Trav& T, bool live, bool any)
{
    unsigned long long* my_lds = g_lds_stack + threadIdx.x;
    Hit best = T.best;
    bool found = !live;          // an any-hit lane that has its answer, or a lane without a ray
    int sp = 0;                  // wave-uniform
    int node = 0;                // wave-uniform
    for (int guard = 4 * S.n_prims + 16; guard > 0; --guard) {
        bool pop = true;
        if (node >= 0) {
            count_lanes(S, 6);
            // (readfirstlane: the index is wave-uniform by construction; said again so that the node
            // comes through scalar loads whatever the compiler's uniformity analysis makes of the loop)
            const float4* n = S.bvh4 + 8 * (size_t)__builtin_amdgcn_readfirstlane(node);
            const float4 lx = ldc(n), ly = ldc(n + 1), lz = ldc(n + 2), hx = ldc(n + 3), hy = ldc(n + 4), hz = ldc(n + 5), cc = ldc(n + 6);
            const float lim = found ? -INFINITY : (best.prim >= 0 ? best.t : T.lim0);
            const Slab4 r = slab4(lx, ly, lz, hx, hy, hz, T.oi, T.inv, lim);
            const unsigned long long m0 = __ballot(r.h0), m1 = __ballot(r.h1), m2 = __ballot(r.h2), m3 = __ballot(r.h3);
            // order key of a child: entry distance of the first lane that passes its box (bits of a
            // non-negative float compare like integers); no lane: never entered
            const int k0 = m0 ? __builtin_amdgcn_readlane(__float_as_int(r.t0), __ffsll((long long)m0) - 1) : 0x7fffffff;
            const int k1 = m1 ? __builtin_amdgcn_readlane(__float_as_int(r.t1), __ffsll((long long)m1) - 1) : 0x7fffffff;
            const int k2 = m2 ? __builtin_amdgcn_readlane(__float_as_int(r.t2), __ffsll((long long)m2) - 1) : 0x7fffffff;
            const int k3 = m3 ? __builtin_amdgcn_readlane(__float_as_int(r.t3), __ffsll((long long)m3) - 1) : 0x7fffffff;
            int first = 0, kmin = k0;
            if (k1 < kmin) { kmin = k1; first = 1; }
            if (k2 < kmin) { kmin = k2; first = 2; }
            if (k3 < kmin) { kmin = k3; first = 3; }
            if (kmin != 0x7fffffff) {
                pop = false;
                const int n_push = (m0 != 0) + (m1 != 0) + (m2 != 0) + (m3 != 0) - 1;
                if (sp + n_push > S.stack_cap) return false;
#define VRT_PPUSH(I, M, H, TT, C)                                                                           \
                if (first != I && M) {                                                                      \
                    my_lds[sp * VRT_STACK_LANES] =                                                          \
                        ((unsigned long long)__float_as_uint(H ? TT : INFINITY) << 32) | (unsigned)__float_as_int(C); \
                    ++sp;                                                                                   \
                }
                VRT_PPUSH(3, m3, r.h3, r.t3, cc.w) VRT_PPUSH(2, m2, r.h2, r.t2, cc.z)
                VRT_PPUSH(1, m1, r.h1, r.t1, cc.y) VRT_PPUSH(0, m0, r.h0, r.t0, cc.x)
#undef VRT_PPUSH
                const int cf = first == 0 ? __float_as_int(cc.x) : first == 1 ? __float_as_int(cc.y)
                             : first == 2 ? __float_as_int(cc.z) : __float_as_int(cc.w);
                node = __builtin_amdgcn_readfirstlane(cf);
            }
        } else {
            const int id = __builtin_amdgcn_readfirstlane(~node);
            count_lanes(S, 7);
            if (!found && id != T.skip) test_prim(S, id, T.o, T.d, T.maxd, best);
            if (any && !found && best.prim >= 0) {
                if (best.t < T.tlim) found = true;
                else { best.prim = -1; best.t = INFINITY; }
            }
            if (any && !__any(!found)) break;   // every lane has its answer
        }
        if (pop) {
            const float lim = found ? -INFINITY : (best.prim >= 0 ? best.t : T.lim0);
            bool got = false;
            while (sp > 0) {
                --sp;
                const unsigned long long e = my_lds[sp * VRT_STACK_LANES];
                if (__any(!(__uint_as_float((unsigned)(e >> 32)) > lim))) {
                    node = __builtin_amdgcn_readfirstlane((int)(unsigned)e);
                    got = true;
                    break;
                }
            }
            if (!got) break;
        }
    }
    T.best = best;
    return true;
}


// ------------------------------------------------------------------ beam: candidates once per wave
//
// Camera rays of a wave (one pixel x 64 samples, or a few neighbouring pixels) leave one point in
// almost one direction.  Instead of every lane testing the same four boxes with its own ray at
// every node, the wave walks the tree ONCE with a conservative bound of all its rays, and uses its
// lanes for different boxes: 16 nodes of a frontier at a time, lane 4k+c testing child c of node
// k.  The bound is interval arithmetic on the slab test: with the reciprocal direction of every
// live lane inside [ilo, ihi] per axis (same sign, else the axis does not constrain) and the
// common origin, a child is entered when max_axis(lower bound of the entry distance) <=
// min_axis(upper bound of the exit distance), which holds for every box some ray of the wave
// passes.  Inner children go back to the frontier, leaves to a candidate list; afterwards every
// lane tests its own ray against every candidate, and the ranking by (t, id) makes the result the
// one of the lane's own traversal.  There is no front-to-back order and no pruning by hits found
// on the way (only by the range, and by hits on the primitives kept out of the tree), so a beam
// collects more leaves than a ray visits (5-8 against 3 on the 10 k-sphere frame) but takes 3-5
// frontier steps instead of 16 node steps.  All 64 lanes must call; live = this lane holds a ray.
// Returns false, with nothing changed, when frontier or candidate list would overflow (a wide
// beam): the caller then traverses as usual.
#ifndef VRT_BEAM_FRONT
#define VRT_BEAM_FRONT 160
#endif
// VRT_BEAM_BOUNCE=1: the first reflections of the hits of a wave's camera rays are cast together too
// (intersect_bounce_wave).  A compile-time choice: as a run-time switch, with both paths in the
// kernel, it cost the tree kernels 3-8 % in register allocation.
#ifndef VRT_BEAM_BOUNCE
#define VRT_BEAM_BOUNCE 1
#endif
#ifndef VRT_BEAM_CAND
#define VRT_BEAM_CAND 96
#endif
// Frontier and candidate list live in the wave's columns of the traversal stack, which is idle
// while the beam runs: slot k = row k / 64, column k % 64 of the wave.
__device__ __forceinline__ int* beam_slot(int k)
{
    return (int*)&g_lds_stack[(k >> 6) * VRT_STACK_LANES + (threadIdx.x & ~63) + (k & 63)];
}

__device__ __forceinline__ void wave_lds_fence_()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// Minimum / maximum over the 64 lanes of a wave, all of which must be active: six DPP steps in the
// vector ALU (shifts within rows of 16 lanes, then the row broadcasts of GFX9), the result read
// from lane 63 into a scalar register.  A lane whose source lies outside its row keeps its value.
// (The s_nop are the wait states a DPP read needs after a vector write of the same register; the
// compiler cannot see into the asm to insert them.)
#define VRT_WAVE_REDUCE(OP)                                                                       \
    asm("s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                   \
        "s_nop 1\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                   \
        "s_nop 1\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                   \
        "s_nop 1\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                   \
        "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                \
        "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                \
        "s_nop 1" : "+v"(v))
__device__ __forceinline__ float wave_min_f(float v)
{
    VRT_WAVE_REDUCE("v_min_f32_dpp");
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f(float v)
{
    VRT_WAVE_REDUCE("v_max_f32_dpp");
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#undef VRT_WAVE_REDUCE
// Minimum and maximum of three values at once: the six reductions take their steps in turn, so the
// five instructions between two steps of one reduction are the wait states its DPP read needs, and
// no s_nop stands between them (13 of a reduction's 19 issue slots are s_nop when it runs alone).
#define VRT_STEP6(CTRL)                                            \
    "v_min_f32_dpp %0, %0, %0 " CTRL "\n\tv_max_f32_dpp %1, %1, %1 " CTRL "\n\t" \
    "v_min_f32_dpp %2, %2, %2 " CTRL "\n\tv_max_f32_dpp %3, %3, %3 " CTRL "\n\t" \
    "v_min_f32_dpp %4, %4, %4 " CTRL "\n\tv_max_f32_dpp %5, %5, %5 " CTRL "\n\t"
__device__ __forceinline__ void wave_minmax3(float& a_lo, float& a_hi, float& b_lo, float& b_hi, float& c_lo, float& c_hi)
{
    asm("s_nop 1\n\t"
        VRT_STEP6("row_shr:1 row_mask:0xf bank_mask:0xf")
        VRT_STEP6("row_shr:2 row_mask:0xf bank_mask:0xf")
        VRT_STEP6("row_shr:4 row_mask:0xf bank_mask:0xf")
        VRT_STEP6("row_shr:8 row_mask:0xf bank_mask:0xf")
        VRT_STEP6("row_bcast:15 row_mask:0xa bank_mask:0xf")
        VRT_STEP6("row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 1"
        : "+v"(a_lo), "+v"(a_hi), "+v"(b_lo), "+v"(b_hi), "+v"(c_lo), "+v"(c_hi));
    a_lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_lo), 63));
    a_hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_hi), 63));
    b_lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b_lo), 63));
    b_hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b_hi), 63));
    c_lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_lo), 63));
    c_hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_hi), 63));
}
#undef VRT_STEP6

// The collection pass.  Not inlined on purpose: compiled into the frame kernel it changed the
// register allocation of every traversal loop there (10 % on frames that never take the beam).
// Takes what it needs by value (a reference to the scene would force the scene into memory).
// Returns the number of candidates, now in beam_slot(VRT_BEAM_FRONT + k), or -1 on overflow.
__device__ __noinline__ int beam_collect(const float4* bvh4, int n_prims, float ix, float iy, float iz, float o_x,
                                         float o_y, float o_z, float lim, int live_i, int one_origin,
                                         unsigned long long* dbg)
{
    // The wave-wide reductions below read lane 63 and DPP sources in every row: with a lane of the
    // wave switched off they would return stale registers, so a partial wave does not take the beam
    // (the caller traverses per lane instead).
    if (__builtin_amdgcn_read_exec() != ~0ull) return -1;
    const bool live = live_i != 0;
    const unsigned long long lm = __ballot(live);
    if (lm == 0ull) return 0;
    const int lane = threadIdx.x & 63;
    // reciprocal-direction interval of the live lanes per axis
    float ilx = live ? ix : INFINITY, ihx = live ? ix : -INFINITY, ily = live ? iy : INFINITY, ihy = live ? iy : -INFINITY,
          ilz = live ? iz : INFINITY, ihz = live ? iz : -INFINITY;
    wave_minmax3(ilx, ihx, ily, ihy, ilz, ihz);
    // nothing beyond the farthest limit of a lane (its range, its "nearer than" bound, a hit it has) matters
    const float far = wave_max_f(live ? lim : 0.f);
    // origin interval: one point for camera rays, the lanes' hit points for occlusion rays
    float olx, ohx, oly, ohy, olz, ohz;
    if (one_origin) {
        const int src = __ffsll((long long)lm) - 1;
        olx = ohx = __shfl(o_x, src, 64); oly = ohy = __shfl(o_y, src, 64); olz = ohz = __shfl(o_z, src, 64);
    } else {
        olx = live ? o_x : INFINITY; ohx = live ? o_x : -INFINITY; oly = live ? o_y : INFINITY; ohy = live ? o_y : -INFINITY;
        olz = live ? o_z : INFINITY; ohz = live ? o_z : -INFINITY;
        wave_minmax3(olx, ohx, oly, ohy, olz, ohz);
    }
    // an axis constrains when all reciprocals are finite and of one sign
    const bool px = ilx > 0.f, nx = ihx < 0.f, cx = (px || nx) && fabsf(ilx) < 1e37f && fabsf(ihx) < 1e37f;
    const bool py = ily > 0.f, ny = ihy < 0.f, cy = (py || ny) && fabsf(ily) < 1e37f && fabsf(ihy) < 1e37f;
    const bool pz = ilz > 0.f, nz = ihz < 0.f, cz = (pz || nz) && fabsf(ilz) < 1e37f && fabsf(ihz) < 1e37f;
    int n_front = 0, n_cand = 0;   // wave-uniform
    const int slot = lane >> 2, c = lane & 3;
    {
        // The first two levels without box tests: the frontier starts with the children of the
        // root's children (lane 4k+c: child c of the root's child k), which fill one step of 16
        // nodes; testing the root and then its children first would be two steps that use 4 and
        // at most 16 lanes.  A leaf among them becomes a candidate untested.
        int code = VRT_BVH4_EMPTY;
        if (lane < 16) {
            const int c1 = ((const VRT_CONST_AS int*)(bvh4 + 6))[slot];
            if (c1 >= 0) code = ((const VRT_CONST_AS int*)(bvh4 + 8 * (size_t)c1 + 6))[c];
            else if (c == 0) code = c1;
        }
        const bool inner = code >= 0, leaf = code < 0 && code != VRT_BVH4_EMPTY;
        const unsigned long long mi = __ballot(inner), ml = __ballot(leaf);
        if (inner) *beam_slot(__builtin_amdgcn_mbcnt_hi((unsigned)(mi >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mi, 0u))) = code;
        if (leaf) *beam_slot(VRT_BEAM_FRONT + __builtin_amdgcn_mbcnt_hi((unsigned)(ml >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ml, 0u))) = ~code;
        n_front = __popcll(mi); n_cand = __popcll(ml);
    }
    wave_lds_fence_();
    for (int guard = n_prims + 8; n_front > 0 && guard > 0; --guard) {
        const int take = n_front < 16 ? n_front : 16;
        const int node = slot < take ? *beam_slot(n_front - 1 - slot) : -1;
        n_front -= take;
        bool hit = false;
        int code = 0;
#if VRT_COUNT_NODES   // a frontier node counts once: its four boxes are tested once, by four lanes
        {
            const unsigned long long m = __ballot(node >= 0 && c == 0);
            if (m && lane == __ffsll((long long)m) - 1)
                atomicAdd(&dbg[(blockIdx.x % VRT_CTR_SLOTS) * VRT_CTR_STRIDE + 6], (unsigned long long)__popcll(m));
        }
#endif
        if ((unsigned)node < (unsigned)(n_prims - 1)) {   // a tree over n primitives has n - 1 inner nodes
            const VRT_CONST_AS float* nb = (const VRT_CONST_AS float*)(bvh4 + 8 * (size_t)node) + c;
            const float lox = nb[0], loy = nb[4], loz = nb[8], hix = nb[12], hiy = nb[16], hiz = nb[20];
            code = __float_as_int(nb[24]);
            float tn = 0.f, tf = far;
            // entry distance (plane - origin) * inv is smallest, over origins and reciprocals, for the
            // largest origin when inv > 0 and the smallest when inv < 0; the exit distance is largest
            // for the other one
#define VRT_BEAM_AXIS(CON, POS, LO, HI, OL, OH, IL, IH)                                     \
            if (CON) {                                                                      \
                const float sn = POS ? LO - OH : HI - OL, sf = POS ? HI - OL : LO - OH;     \
                tn = fmaxf(tn, fminf(sn * IL, sn * IH));                                    \
                tf = fminf(tf, fmaxf(sf * IL, sf * IH));                                    \
            }
            VRT_BEAM_AXIS(cx, px, lox, hix, olx, ohx, ilx, ihx)
            VRT_BEAM_AXIS(cy, py, loy, hiy, oly, ohy, ily, ihy)
            VRT_BEAM_AXIS(cz, pz, loz, hiz, olz, ohz, ilz, ihz)
#undef VRT_BEAM_AXIS
            // An unused child slot holds a box at +inf, which every constrained axis rejects (its
            // entry distance is +inf or its exit distance -inf).  When the live rays' directions
            // differ in sign, or are not finite, on all three axes nothing constrains, and the
            // slot has to be told by its code: as a "leaf" it would name primitive 0x7fffffff.
            hit = tn <= tf * 1.00001f + 1e-6f && code != VRT_BVH4_EMPTY;
        }
        const bool inner = hit && code >= 0, leaf = hit && code < 0;
        const unsigned long long mi = __ballot(inner), ml = __ballot(leaf);
        if (n_front + __popcll(mi) > VRT_BEAM_FRONT || n_cand + __popcll(ml) > VRT_BEAM_CAND) return -1;
        wave_lds_fence_();   // the frontier entries of this step have been read
        if (inner) *beam_slot(n_front + __builtin_amdgcn_mbcnt_hi((unsigned)(mi >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mi, 0u))) = code;
        if (leaf) *beam_slot(VRT_BEAM_FRONT + n_cand + __builtin_amdgcn_mbcnt_hi((unsigned)(ml >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ml, 0u))) = ~code;
        n_front += __popcll(mi); n_cand += __popcll(ml);
        wave_lds_fence_();
    }
    return n_front > 0 ? -1 : n_cand;
}

__device__ __forceinline__ bool intersect_bvh4_beam(const DevScene& S, Trav& T, bool live)
{
    const int n_cand = __builtin_amdgcn_readfirstlane(
        beam_collect(S.bvh4, S.n_prims, T.inv.x, T.inv.y, T.inv.z, T.o.x, T.o.y, T.o.z, trav_limit(T), live ? 1 : 0, 1, S.dbg));
    if (n_cand < 0) return false;
    for (int k = 0; k < n_cand; ++k) {
        const int id = __builtin_amdgcn_readfirstlane(*beam_slot(VRT_BEAM_FRONT + k));
        if ((unsigned)id >= (unsigned)S.n_prims) continue;   // cannot happen; a wild id must not become an address
        count_lanes(S, 7, live);
        if (live && id != T.skip) test_prim(S, id, T.o, T.d, T.maxd, T.best);
    }
    return true;
}

// Closest hit of the camera rays of a wave (tree scenes): every lane of the wave calls, cast = this
// lane has a ray.  beam: try the candidate collection above first.
__device__ __forceinline__ Hit intersect_primary_bvh(const DevScene& S, f3 o, f3 d, float maxd, bool cast, bool packet,
                                                     bool beam)
{
    // the beam works on a traversal record of its own: sharing the one of the ordinary traversal
    // below cost that traversal 10 % (register allocation)
    if (beam) {
        Trav T;
        trav_begin(T, o, d, maxd, INFINITY, -1);
        if (cast) trav_big(S, T, false);
        if (intersect_bvh4_beam(S, T, cast)) return T.best;
    }
    Hit h; h.prim = -1; h.t = INFINITY; h.aux = 0;
    if (cast) h = intersect_bvh4(S, o, d, maxd, INFINITY, -1, false, packet);
    return h;
}

// Occlusion queries a wave asks together (the ambient-occlusion rays, or the shadow rays towards
// one light, of the hits of its camera rays): is there a collider nearer than tlim?  Every lane
// of the wave calls, want = this lane has a query.  The beam takes the lanes' origins as an
// interval as well; an any-hit query needs no order among the candidates.
__device__ __forceinline__ bool occluded_wave(const DevScene& S, bool want, f3 o, f3 d, float maxd, float tlim, int skip,
                                              bool packet)
{
    Trav T;
    trav_begin(T, o, d, maxd, tlim, skip);
    const bool live = want && !is_zero3(d);
    bool found = false;
    if (live) found = trav_big(S, T, true);
    const bool need = live && !found;
    const int n_cand = __builtin_amdgcn_readfirstlane(
        beam_collect(S.bvh4, S.n_prims, T.inv.x, T.inv.y, T.inv.z, o.x, o.y, o.z, T.lim0, need ? 1 : 0, 0, S.dbg));
    if (n_cand < 0) {
        if (need) found = intersect_bvh4(S, o, d, maxd, tlim, skip, true, packet).prim >= 0;
        return found;
    }
    for (int k = 0; k < n_cand; ++k) {
        if (!__any(need && !found)) break;
        const int id = __builtin_amdgcn_readfirstlane(*beam_slot(VRT_BEAM_FRONT + k));
        if ((unsigned)id >= (unsigned)S.n_prims) continue;   // as in intersect_bvh4_beam
        count_lanes(S, 7, need && !found);
        if (need && !found && id != skip) {
            test_prim(S, id, T.o, T.d, T.maxd, T.best);
            if (T.best.prim >= 0) {
                if (T.best.t < tlim) found = true;
                else { T.best.prim = -1; T.best.t = INFINITY; }
            }
        }
    }
    return found;
}

// Closest hits of rays a wave casts together from different points (the first reflections of the
// hits of its camera rays: neighbouring pixels or samples of one pixel mirror in neighbouring
// surface points).  Every lane of the wave calls, want = this lane has a ray (direction not zero).
// The candidates are collected once for the wave, with the origins as an interval, and ranked per
// lane by (t, id), so a lane's hit is the one its own traversal finds; a beam too wide for the
// lists leaves every lane to that traversal.
__device__ __forceinline__ Hit intersect_bounce_wave(const DevScene& S, bool want, f3 o, f3 d, float maxd, int skip,
                                                     bool packet)
{
    Trav T;
    trav_begin(T, o, d, maxd, INFINITY, skip);
    if (want) trav_big(S, T, false);
    const int n_cand = __builtin_amdgcn_readfirstlane(
        beam_collect(S.bvh4, S.n_prims, T.inv.x, T.inv.y, T.inv.z, o.x, o.y, o.z, trav_limit(T), want ? 1 : 0, 0, S.dbg));
    if (n_cand < 0) {
        Hit h; h.prim = -1; h.t = INFINITY; h.aux = 0;
        if (want) h = intersect_bvh4(S, o, d, maxd, INFINITY, skip, false, packet);
        return h;
    }
    for (int k = 0; k < n_cand; ++k) {
        const int id = __builtin_amdgcn_readfirstlane(*beam_slot(VRT_BEAM_FRONT + k));
        if ((unsigned)id >= (unsigned)S.n_prims) continue;   // as in intersect_bvh4_beam
        count_lanes(S, 7, want);
        if (want && id != skip) test_prim(S, id, T.o, T.d, T.maxd, T.best);
    }
    return T.best;
}

// ------------------------------------------------------------------ four lanes per ray
//
// Traversal for a wave in which few lanes hold a query (the later bounces: on the 10k-sphere frame
// 11 % of the lanes are still on a path after two hits, 5 % after three).  A per-lane step costs
// the wave the same ~120 vector instructions whether 4 or 64 lanes take it, so here the wave is
// re-cut: up to 16 queries per pass, each given to a quad of four neighbouring lanes, and lane c
// of a quad tests child c of the quad's node (one box instead of four per lane, 7 dwords instead
// of 7 x 16 bytes loaded).  The four entry distances are ranked with quad permutes (DPP), the
// nearest hit child is entered, the others go to the quad's stack, which is the four lanes'
// stack columns taken together (4 x VRT_TRAV_STACK entries).  Leaves are tested by the four
// lanes alike, so the quad's state stays replicated without broadcasts.  Box test, primitive
// tests and the ranking of candidates by (t, id) are those of the per-lane traversal, hence so
// are the hits.  All 64 lanes of the wave must call; want = this lane holds a query.
#ifndef VRT_QUAD_MAX
#define VRT_QUAD_MAX 16
#endif
__shared__ int g_lds_quad[VRT_STACK_LANES];   // per wave: 16 x (t bits, primitive, face, source lane)

#define VRT_DPP_ROT1 0x39   // quad_perm [1,2,3,0]
#define VRT_DPP_ROT2 0x4E   // quad_perm [2,3,0,1]
#define VRT_DPP_ROT3 0x93   // quad_perm [3,0,1,2]
__device__ __forceinline__ unsigned quad_rot(unsigned v, int which)
{
    if (which == 1) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, VRT_DPP_ROT1, 0xF, 0xF, true);
    if (which == 2) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, VRT_DPP_ROT2, 0xF, 0xF, true);
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, VRT_DPP_ROT3, 0xF, 0xF, true);
}

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Entry e of the stack of the quad whose first lane is column qcol of the block.
__device__ __forceinline__ unsigned long long* quad_slot(int qcol, int e)
{
    return g_lds_stack + (e >> 2) * VRT_STACK_LANES + qcol + (e & 3);
}

// Inner-node step of a quad; c = lane & 3.  Returns true when no child was hit (the quad pops).
__device__ __forceinline__ bool quad_inner(const DevScene& S, Trav& T, int c, int qcol)
{
    count_lanes(S, 6, c == 0);
    const float* nf = (const float*)S.bvh4 + 32 * (size_t)T.node + c;
    const float lx = nf[0], ly = nf[4], lz = nf[8], hx = nf[12], hy = nf[16], hz = nf[20];
    const int code = __float_as_int(nf[24]);
    const float lim = trav_limit(T);
    // slab_test() on this lane's child
    const float ax = (lx - T.o.x) * T.inv.x, bx = (hx - T.o.x) * T.inv.x;
    const float ay = (ly - T.o.y) * T.inv.y, by = (hy - T.o.y) * T.inv.y;
    const float az = (lz - T.o.z) * T.inv.z, bz = (hz - T.o.z) * T.inv.z;
    const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
    const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), lim));
    const bool hit = t0 <= t1 * 1.0000004f;
    // order key: entry distance (non-negative, so its bits order like the float) with the child
    // slot in the two lowest bits, which makes the keys of a quad distinct
    const unsigned key = hit ? ((__float_as_uint(t0) & ~3u) | (unsigned)c) : 0xffffffffu;
    const unsigned k1 = quad_rot(key, 1), k2 = quad_rot(key, 2), k3 = quad_rot(key, 3);
    const int n_hit = (int)(key != 0xffffffffu) + (int)(k1 != 0xffffffffu) + (int)(k2 != 0xffffffffu) + (int)(k3 != 0xffffffffu);
    if (n_hit == 0) return true;
    const int rank = (int)(k1 < key) + (int)(k2 < key) + (int)(k3 < key);
    // the nearest child's code, or-ed together over the quad
    const unsigned mine = (hit && rank == 0) ? (unsigned)code : 0u;
    const unsigned next = mine | quad_rot(mine, 1) | quad_rot(mine, 2) | quad_rot(mine, 3);
    // the other hit children wait on the quad's stack, the nearer the higher
    const int n_push = n_hit - 1;
    if (hit && rank >= 1) {
        const int e = T.sp + (n_push - rank);
        if (e < 4 * S.stack_cap) *quad_slot(qcol, e) = ((unsigned long long)__float_as_uint(t0) << 32) | (unsigned)code;
    }
    T.sp += n_push;
    if (T.sp > 4 * S.stack_cap) { T.sp = 4 * S.stack_cap; T.ovf = 1; }   // finished by trav_stackless()
    T.node = (int)next;
    return false;
}

__device__ __forceinline__ bool quad_pop(Trav& T, int qcol)
{
    const float lim = trav_limit(T);
    while (T.sp > 0) {
        --T.sp;
        const unsigned long long e = *quad_slot(qcol, T.sp);
        if (!(__uint_as_float((unsigned)(e >> 32)) > lim)) { T.node = (int)(unsigned)e; return true; }
    }
    return false;
}

__device__ __forceinline__ void intersect_bvh4_quads(const DevScene& S, bool want, f3 o, f3 d, float maxd, float tlim,
                                                     int skip, bool any, Hit& out)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(want);
    const int n = __popcll(m);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    int* qs = g_lds_quad + (threadIdx.x & ~63);
    const int q = lane >> 2, c = lane & 3, qcol = threadIdx.x & ~3;
    for (int p0 = 0; p0 < n; p0 += 16) {   // wave-uniform
        const bool mine = want && rank >= p0 && rank < p0 + 16;
        if (mine) qs[(rank - p0) * 4 + 3] = lane;
        wave_lds_fence();
        const bool has = p0 + q < n;
        const int src = has ? qs[q * 4 + 3] : lane;
        Trav T;
        {
            const f3 qo = mk3(__shfl(o.x, src, 64), __shfl(o.y, src, 64), __shfl(o.z, src, 64));
            const f3 qd = mk3(__shfl(d.x, src, 64), __shfl(d.y, src, 64), __shfl(d.z, src, 64));
            trav_begin(T, qo, qd, __shfl(maxd, src, 64), __shfl(tlim, src, 64), __shfl(skip, src, 64));
        }
        const bool qany = __shfl((int)any, src, 64) != 0;
        bool done = !has, found = false;
        if (!done && trav_big(S, T, qany, c == 0)) { done = true; found = true; }
        int guard = 2 * S.n_prims + 8;   // only a corrupt tree gets there
        while (__any(!done)) {
            if (!done) {
                bool pop;
                if (T.node >= 0) pop = quad_inner(S, T, c, qcol);
                else { if (trav_leaf(S, T, qany, c == 0)) { done = true; found = true; } pop = true; }
                if (!done && pop && !quad_pop(T, qcol)) done = true;
                if (--guard <= 0) done = true;
            }
        }
        if (has && !found && T.ovf) trav_stackless(S, T, qany);
        if (has && c == 0) { qs[q * 4] = __float_as_int(T.best.t); qs[q * 4 + 1] = T.best.prim; qs[q * 4 + 2] = T.best.aux; }
        wave_lds_fence();
        if (mine) {
            const int* r = qs + (rank - p0) * 4;
            out.t = __int_as_float(r[0]); out.prim = r[1]; out.aux = r[2];
        }
        wave_lds_fence();
    }
}

// One query per lane of a whole wave (all 64 lanes call; want = this lane has one): four lanes per
// ray when few lanes ask, else the per-lane traversal, wave-uniform first when the queries are
// known to stay close together (coherent, wave-uniform).
__device__ __forceinline__ void intersect_bvh4_wave(const DevScene& S, bool want, f3 o, f3 d, float maxd, float tlim,
                                                    int skip, bool any, bool coherent, int quad_max, Hit& out)
{
    const int n = __popcll(__ballot(want));
    if (n <= quad_max) intersect_bvh4_quads(S, want, o, d, maxd, tlim, skip, any, out);
    else if (want) out = intersect_bvh4(S, o, d, maxd, tlim, skip, any, coherent);
}

// Closest hit (ANY = false) or "some hit with t < tlim" (ANY = true) within maxd.
template <bool BVH, bool ANY, bool CAM = false>
__device__ __forceinline__ Hit intersect(const DevScene& S, f3 o, f3 d, float maxd, int skip, float tlim = INFINITY,
                                         bool packet = false)
{
    if (BVH) return intersect_bvh4(S, o, d, maxd, tlim, skip, ANY, packet);
    return intersect_brute<CAM>(S, o, d, maxd, tlim, skip, ANY);
}

// The same with the any-hit choice made per lane at run time.
template <bool BVH>
__device__ __forceinline__ Hit intersect_rt(const DevScene& S, f3 o, f3 d, float maxd, int skip, float tlim, bool any)
{
    if (BVH) return intersect_bvh4(S, o, d, maxd, tlim, skip, any);
    return intersect_brute<false>(S, o, d, maxd, tlim, skip, any);
}

// hit.point = origin + direction * distance; hit.normal per collider type.
__device__ __forceinline__ void hit_geometry(const DevScene& S, Hit h, f3 o, f3 d, f3& point, f3& normal)
{
    point = o + d * h.t;
    if (h.prim < S.n_sph) {
        normal = normalize3(point - xyz(ldc(S.sph + h.prim)));
    } else if (h.prim < S.n_sph + S.n_box) {
        float4 a = ldc(S.box + (4 * (h.prim - S.n_sph) + 1 + (h.aux & 3)));
        float s = (h.aux & 4) ? 1.f : -1.f;
        normal = xyz(a) * s;
    } else {
        const float4* tr = S.tri + 3 * (size_t)(h.prim - S.n_sph - S.n_box);
        normal = normalize3(cross3(xyz(ldc(tr + 1)), xyz(ldc(tr + 2))));
    }
}

// ------------------------------------------------------------------ recording / counting

struct Counters { unsigned precheck, primary, reflect, transmit, shadow, ao; };

// VRT_LDS_COUNTERS=1 counts rays in LDS (one ds_add per wave and event, the wave's active lanes
// counted with a ballot) instead of in six registers per lane.
#ifndef VRT_LDS_COUNTERS
#define VRT_LDS_COUNTERS 1
#endif
#ifndef VRT_STUDY_NO_ANYHIT
#define VRT_STUDY_NO_ANYHIT 0
#endif
// VRT_STUDY_LIVE_HIST=1 (study build): the counter words report, instead of rays per class, how many
// turns of the bounce loop ran with 1, 2, 3-4, 5-8, 9-16, 17-32, 33-48, 49-64 lanes of the wave on a path.
#ifndef VRT_STUDY_LIVE_HIST
#define VRT_STUDY_LIVE_HIST 0
#endif
#if VRT_LDS_COUNTERS
__shared__ unsigned g_lds_ctr[VRT_RAY_CLASSES];
__device__ __forceinline__ void count_event(int k)
{
#if VRT_STUDY_LIVE_HIST
    return;
#endif
    unsigned long long m = __ballot(1);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&g_lds_ctr[k], (unsigned)__popcll(m));
}
#define VRT_COUNT(C, FIELD, K) count_event(K)
#else
#define VRT_COUNT(C, FIELD, K) ((C).FIELD++)
#endif

// Global row of local row r of a strip-sharded frame (vrt_params.tile_*).
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

// Adds this block's counters into one of VRT_CTR_SLOTS slots (one atomic per class per block).
__device__ __forceinline__ void flush_counters(const Counters& C, unsigned long long* ctr, unsigned block_linear)
{
#if VRT_LDS_COUNTERS
    __syncthreads();   // counts were added to g_lds_ctr as the rays were cast
    if (threadIdx.x < VRT_RAY_CLASSES && g_lds_ctr[threadIdx.x])
        atomicAdd(&ctr[(block_linear % VRT_CTR_SLOTS) * VRT_CTR_STRIDE + threadIdx.x],
                  (unsigned long long)g_lds_ctr[threadIdx.x]);
#else
    __shared__ unsigned s_ctr[VRT_RAY_CLASSES];
    if (threadIdx.x < VRT_RAY_CLASSES) s_ctr[threadIdx.x] = 0;
    __syncthreads();
    // one field at a time: an indexed local array of the six counts would live in scratch
#define VRT_FLUSH(K, FIELD)                                               \
    {                                                                     \
        unsigned s = wave_sum(C.FIELD);                                   \
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&s_ctr[K], s);        \
    }
    VRT_FLUSH(VRT_RAY_PRECHECK, precheck) VRT_FLUSH(VRT_RAY_PRIMARY, primary) VRT_FLUSH(VRT_RAY_REFLECT, reflect)
    VRT_FLUSH(VRT_RAY_TRANSMIT, transmit) VRT_FLUSH(VRT_RAY_SHADOW, shadow) VRT_FLUSH(VRT_RAY_AO, ao)
#undef VRT_FLUSH
    __syncthreads();
    if (threadIdx.x < VRT_RAY_CLASSES && s_ctr[threadIdx.x])
        atomicAdd(&ctr[(block_linear % VRT_CTR_SLOTS) * VRT_CTR_STRIDE + threadIdx.x],
                  (unsigned long long)s_ctr[threadIdx.x]);
#endif
}

struct Recorder {
    vrt_path_seg* segs; // null = off
    int cap;
    int n;
    int sample;
};

template <bool RECORD>
__device__ __forceinline__ void record_seg(Recorder& R, f3 o, f3 d, Hit h, int kind, int depth, float range)
{
    if (!RECORD) return;
    if (R.n < R.cap) {
        vrt_path_seg* s = R.segs + R.n;
        f3 e = o + d * (h.prim >= 0 ? h.t : range);
        s->origin[0] = o.x; s->origin[1] = o.y; s->origin[2] = o.z;
        s->end[0] = e.x; s->end[1] = e.y; s->end[2] = e.z;
        s->kind = kind; s->depth = depth; s->hit_prim = h.prim; s->sample = R.sample;
    }
    R.n++;
}

// ------------------------------------------------------------------ shading

// Phong coefficients (ka, kd, ks, exp) by material class (VRT_MAT_*), one copy per block in LDS.
// The whole block calls stage_materials() once before it shades anything.
__shared__ float4 g_lds_matk[8];
__device__ __forceinline__ void stage_materials(const DevScene& S)
{
    if (threadIdx.x == 0) {
        const vrt_shading& sh = S.sh;
        const float4 solid = make_float4(sh.solid_ka, sh.solid_kd, sh.solid_ks, sh.solid_exp);
        g_lds_matk[0] = solid;
        g_lds_matk[VRT_MAT_SOLID] = solid;
        g_lds_matk[VRT_MAT_METAL] = make_float4(sh.metal_ka, sh.metal_kd, sh.metal_ks, sh.metal_exp);
        g_lds_matk[VRT_MAT_DIELECTRIC] = make_float4(sh.glass_ka, sh.glass_kd, sh.glass_ks, sh.glass_exp);
        g_lds_matk[VRT_MAT_MATTE] = make_float4(sh.matte_ka, sh.matte_kd, 0.f, 1.f);
        g_lds_matk[VRT_MAT_TRANSPARENT] = make_float4(sh.transp_ka, sh.transp_kd, sh.transp_ks, sh.transp_exp);
        g_lds_matk[6] = solid; g_lds_matk[7] = solid;
    }
    __syncthreads();
}

// Util/RayTraceUtility.cs:123-131
__device__ __forceinline__ f3 sky_color(f3 direction)
{
    f3 u = normalize3(direction);
    float t = 0.5f * (u.y + 1.f);
    f3 a = mk3(1.f, 1.f, 1.f) * (1.f - t);
    f3 b = mk3(.5f, .7f, 1.f) * t;
    return a + b;
}

// How shading gets the answer to "is there a collider along this ray nearer than tlim":
// query k = 0 is the ambient-occlusion ray of a hit, k = 1 + i the shadow ray towards light i.
// OccInline traverses on the spot.  The wavefront pipeline has two more providers: one that
// queues the query for a later, densely packed pass and one that reads the answers back.
struct OccInline {
    static constexpr bool counts = true;   // the shading pass that uses this provider counts the rays
    bool packet = false;                   // the queries of the calling lanes stay close together (hits of camera rays)
    template <bool BVH>
    __device__ __forceinline__ bool occluded(const DevScene& S, int, f3 o, f3 d, float maxd, float tlim, int skip)
    {
#if VRT_STUDY_NO_ANYHIT   // timing study only: every query answered "free" without a traversal
        return false;
#else
        return intersect<BVH, true>(S, o, d, maxd, skip, tlim, packet).prim >= 0;
#endif
    }
};

// Provider of the answers gathered in a flag word (bit k = query k is occluded).
struct OccBits {
    static constexpr bool counts = false;
    unsigned bits;
    template <bool BVH>
    __device__ __forceinline__ bool occluded(const DevScene&, int k, f3, f3, float, float, int) { return (bits >> k) & 1u; }
};

// Provider of a shading pass whose only purpose is to find out which queries the hit asks: the
// ambient-occlusion ray (k = 0) and the shadow ray towards the first light (k = 1) are kept for the
// wave to trace together (occluded_wave), those towards further lights are traced on the spot and
// their answers noted in `bits`.  The pass counts the rays; its shading result is discarded.
struct OccCapture {
    static constexpr bool counts = true;
    bool packet = false;
    unsigned has = 0, bits = 0;
    f3 d0 = {0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f};   // zero direction: no query
    float tlim0 = 0.f, tlim1 = 0.f, maxd0 = 0.f, maxd1 = 0.f;
    template <bool BVH>
    __device__ __forceinline__ bool occluded(const DevScene& S, int k, f3 o, f3 d, float maxd, float tlim, int skip)
    {
        if (k == 0) { has |= 1u; d0 = d; tlim0 = tlim; maxd0 = maxd; }
        else if (k == 1) { has |= 2u; d1 = d; tlim1 = tlim; maxd1 = maxd; }
        else if (intersect<BVH, true>(S, o, d, maxd, skip, tlim, packet).prim >= 0) bits |= 1u << k;
        return false;
    }
};

// Light/AmbientOccluder.cs:56-85.  The reference builds u and v with Mathf.Pow of the normal's
// components and shoots sp.x*u + sp.y*v + sp.z*w with the hemisphere sample sp = (0,0,1) (its
// multi-jittered table is all zeros, Sampler/MultiJitteredSampler.cs:23-27,39).  A negative base
// or Pow(0, negative) makes u or v NaN/Inf, 0*Inf poisons the direction, the Ray constructor
// normalises it to zero and no ray is cast.  Working that through: the direction degenerates
// iff n.x < 0 or n.z < 0 or (n.y < 0 and n.y != -1); otherwise it equals n.
template <bool BVH, bool RECORD, class OCC>
__device__ __forceinline__ f3 ambient_L(const DevScene& S, const DevParams& P, f3 point, f3 n, int prim,
                                        int depth, Counters& C, Recorder& R, OCC& occ)
{
    bool cast = true;
    if (!(P.flags & VRT_FIX_AO_BASIS))
        cast = !(n.x < 0.f || n.z < 0.f || (n.y < 0.f && n.y != -1.f));
    bool occluded = false;
    if (cast) {
        f3 dir = normalize3(mk3(n.x + 0.f, n.y + 0.f, n.z + 0.f));
        if (!is_zero3(dir)) {
            if (OCC::counts) VRT_COUNT(C, ao, VRT_RAY_AO);
            if constexpr (RECORD) {
                Hit h = intersect<BVH, false>(S, point, dir, P.range, prim);
                record_seg<RECORD>(R, point, dir, h, VRT_RAY_AO, depth, P.range);
                occluded = h.prim >= 0;
            } else {
                occluded = occ.template occluded<BVH>(S, 0, point, dir, P.range, INFINITY, prim);
            }
        }
    }
    // (min_amount * ls) * color when occluded, else color * ls.  The tree kernels take the products the
    // host formed (DevScene): formed here they are hoisted to the kernel's start and spilled by every
    // wave.  The brute-force kernels have the registers, and nine more scalars would cost them 1 %.
    if constexpr (BVH) {
        if (occluded) return mk3(S.ao_occl[0], S.ao_occl[1], S.ao_occl[2]);
        return mk3(S.ao_free[0], S.ao_free[1], S.ao_free[2]);
    }
    const vrt_shading& sh = S.sh;
    f3 lc = mk3(sh.ao_color[0], sh.ao_color[1], sh.ao_color[2]);
    if (occluded) {
        f3 mn = mk3(sh.ao_min_amount[0], sh.ao_min_amount[1], sh.ao_min_amount[2]);
        return (mn * sh.ao_ls) * lc;
    }
    return lc * sh.ao_ls;
}

// Material/PhongMaterial.cs:53-89 with BRDF/LambertianBRDF.cs:23-26,54-57,
// BRDF/GlossySpecularBRDF.cs:40-63, Light/PointLight.cs:23-51, Light/DirectionalLight.cs:22-42.
// matte: Material/MatteMaterial.cs:44-75 instead: the same ambient term, no shadow test, no specular
// lobe, and the light term associated as (F * L) * ndotwi.
template <bool BVH, bool RECORD, class OCC>
__device__ __forceinline__ f3 phong_shade(const DevScene& S, const DevParams& P, f3 point, f3 n, int prim,
                                          f3 raydir, int depth, float ka, float kd, float ks, float ex,
                                          f3 cd, Counters& C, Recorder& R, OCC& occ, bool matte = false)
{
    f3 wo = -raydir;
    f3 L = (cd * ka) * ambient_L<BVH, RECORD>(S, P, point, n, prim, depth, C, R, occ);
    for (int i = 0; i < S.n_lights; ++i) {
        const DevLight l = ldc(S.lights + i);
        f3 lp = mk3(l.vx, l.vy, l.vz);
        f3 wi;
        if (l.type == VRT_LIGHT_POINT) {
            f3 tmp = lp - point;
            float len = sqrt_ieee(tmp.x * tmp.x + tmp.y * tmp.y + tmp.z * tmp.z);
            wi = tmp / len;
        } else {
            wi = lp;
        }
        float ndotwi = dot3(n, wi);
        if (ndotwi > 0.f) {
            bool in_shadow = false;
            if (l.cast && !matte) {
                f3 sd = normalize3(wi);
                f3 loc = l.type == VRT_LIGHT_POINT ? lp : mk3(l.lx, l.ly, l.lz);
                float dist = mag3(loc - point);
                if (!is_zero3(sd)) {
                    if (OCC::counts) VRT_COUNT(C, shadow, VRT_RAY_SHADOW);
                    // occluded iff the closest hit is nearer than the light, i.e. iff any
                    // collider is hit at t < dist; recording needs the closest hit itself
                    if constexpr (RECORD) {
                        Hit h = intersect<BVH, false>(S, point, sd, 30.f, prim);
                        record_seg<RECORD>(R, point, sd, h, VRT_RAY_SHADOW, depth, 30.f);
                        if (h.prim >= 0) in_shadow = h.t < dist;
                    } else {
                        in_shadow = occ.template occluded<BVH>(S, 1 + i, point, sd, 30.f, dist, prim);
                    }
                }
            }
            if (matte) {
                f3 f = (cd * kd) * 0.3183098861837906715f;
                L = L + (f * (mk3(l.cr, l.cg, l.cb) * l.ls)) * ndotwi;
            } else if (!in_shadow) {
                f3 diff = (cd * kd) * 0.3183098861837906715f;
                float spec = 0.f;
                // ks = 0 (the solid material) makes the lobe 0 * pow = 0 for every finite pow,
                // which a non-negative exponent guarantees here (rdb <= 1 up to rounding)
                if (!(ks == 0.f && ex >= 0.f)) {
                    float nda = dot3(n, wo);
                    f3 r = mk3(-wo.x + 2.f * n.x * nda, -wo.y + 2.f * n.y * nda, -wo.z + 2.f * n.z * nda);
                    float rdb = dot3(r, wi);
                    if (rdb > 0.f) spec = (1.f * ks) * powf(rdb, ex);
                }
                f3 lam = mk3(diff.x + spec, diff.y + spec, diff.z + spec);
                f3 world = (mk3(l.cr, l.cg, l.cb) * l.ls) * ndotwi;
                L = L + lam * world;
            }
        }
    }
    return L;
}

// Material/TransparentMaterial.cs:44-66 after the Phong term: the reflected ray and, unless
// PerfectTransmitterBTDF.Tir says total internal reflection, the transmitted ray with their weights.
// Tir (BTDF/PerfectTransmitterBTDF.cs:57-69) looks at the direction the BTDF was constructed with
// (sh.transp_ray_dir), SampleF (:32-49) at the ray being shaded.
__device__ __forceinline__ int transparent_children(const vrt_shading& sh, f3 n, f3 d, f3& d0, f3& w0, f3& d1, f3& w1)
{
    const f3 wo = -d;
    const float ndotwo = dot3(n, wo);
    const f3 wi = (-wo) + (n * 2.f) * ndotwo;
    d0 = normalize3(wi);
    const f3 bwo = -mk3(sh.transp_ray_dir[0], sh.transp_ray_dir[1], sh.transp_ray_dir[2]);
    const float bci = dot3(n, bwo);
    float beta = sh.transp_ior;
    if (bci < 0.f) beta = 1.f / beta;
    if ((1.f - (1.f - bci * bci) / (beta * beta)) < 0.f) { w0 = mk3(1.f, 1.f, 1.f); return 1; }
    const float fr = sh.transp_kr / fabsf(dot3(n, wi));
    f3 nn = n;
    float ci = dot3(nn, wo);
    float eta = sh.transp_ior;
    if (ci < 0.f) { ci = -ci; nn = -nn; eta = 1.f / eta; }
    const float temp = 1.f - (1.f - ci * ci) / (eta * eta);
    const float c2 = sqrtf(temp);
    const f3 wt = ((-wo) / eta) - nn * (c2 - ci / eta);
    const float ft = sh.transp_kt / (eta * eta) / fabsf(dot3(n, wt));
    const float wr = fr * fabsf(dot3(n, wi)), wtt = ft * fabsf(dot3(n, wt));
    w0 = mk3(wr, wr, wr);
    d1 = normalize3(wt); w1 = mk3(wtt, wtt, wtt);
    return 2;
}

// BRDF/FresnelReflectorBRDF.cs:14-39 (only with VRT_FIX_DIELECTRIC)
__device__ __forceinline__ float fresnel_kr(f3 normal, f3 direction, float eta_in, float eta_out)
{
    float ndotd = dot3(-normal, direction);
    float eta;
    if (ndotd < 0.f) { normal = -normal; eta = eta_out / eta_in; }
    else eta = eta_in / eta_out;
    float ci = dot3(-normal, direction);
    float ct = sqrtf(1.f - (1.f - ci * ci) / (eta * eta));
    float rpar = (eta * ci - ct) / (eta * ci + ct);
    float rper = (ci - eta * ct) / (ci + eta * ct);
    return 0.5f * (rpar * rpar + rper * rper);
}

// Texture2D.GetPixel with out-of-range coordinates handled by the wrap mode (0 repeat, 1 clamp).
__device__ __forceinline__ int wrap_pixel(int p, int n, int wrap)
{
    if (wrap == 1) return p < 0 ? 0 : (p >= n ? n - 1 : p);
    p %= n;
    return p < 0 ? p + n : p;
}

// Tracer/WhittedTracer.cs:55-63: the texel under hit.textureCoord, which is (0,0) on sphere and
// box colliders and the barycentric mix of the vertex coordinates on a mesh; the barycentrics
// are those of the triangle test (u/det, v/det), recomputed with the same operations.
__device__ __forceinline__ f3 texel_at_hit(const DevScene& S, int tx, Hit h, f3 o, f3 d)
{
    float tu = 0.f, tv = 0.f;
    const int first_tri = S.n_sph + S.n_box;
    if (h.prim >= first_tri && S.tri_uv) {
        const size_t ti = (size_t)(h.prim - first_tri);
        const float4* tr = S.tri + 3 * ti;
        f3 v0 = xyz(tr[0]), e1 = xyz(tr[1]), e2 = xyz(tr[2]);
        f3 pvec = cross3(d, e2);
        float det = dot3(e1, pvec);
        f3 tvec = o - v0;
        float u = dot3(tvec, pvec);
        f3 qvec = cross3(tvec, e1);
        float v = dot3(d, qvec);
        float bu = u / det, bv = v / det;
        float bw = 1.f - bu - bv;
        float2 a = S.tri_uv[3 * ti], b = S.tri_uv[3 * ti + 1], c = S.tri_uv[3 * ti + 2];
        tu = a.x * bw + b.x * bu + c.x * bv;
        tv = a.y * bw + b.y * bu + c.y * bv;
    }
    const int4 info = S.tex_info[tx];
    float px = tu * (float)info.y, py = tv * (float)info.z;
    int ix = wrap_pixel((int)floorf(px), info.y, info.w);
    int iy = wrap_pixel((int)floorf(py), info.z, info.w);
    float4 t = S.tex_data[(size_t)info.x + (size_t)iy * info.y + ix];
    return mk3(t.x, t.y, t.z);
}

// One surface interaction: direct lighting A and up to two child rays with their weights.
struct Bounce {
    f3 A;         // Phong term of this hit
    int n_child;  // 0, 1 or 2
    f3 o;         // child origin = hit point
    f3 d0, w0;    // first child (reflection): unit direction, weight
    f3 d1, w1;    // second child (transmission)
    int prim;
};

// Tracer/WhittedTracer.cs:49-88 (material dispatch) + Material/ReflectiveMaterial.cs:25-42 +
// Material/DielectricMaterial.cs:56-146 + BRDF/PerfectSpecularBRDF.cs:29-39.
template <bool BVH, bool RECORD, class OCC>
__device__ __forceinline__ void shade_hit(const DevScene& S, const DevParams& P, Hit h, f3 o, f3 d, int depth,
                                          Counters& C, Recorder& R, Bounce& B, OCC& occ)
{
    f3 point, n;
    hit_geometry(S, h, o, d, point, n);
    const int mi = ldc(S.prim_mat + h.prim);
    if (S.mat_tex) {   // a textured material returns its texel unshaded and ends the path
        const int tx = S.mat_tex[mi];
        if (tx >= 0) {
            B.prim = h.prim; B.o = point; B.n_child = 0; B.A = texel_at_hit(S, tx, h, o, d);
            B.d0 = B.w0 = B.d1 = B.w1 = mk3(0.f, 0.f, 0.f);
            return;
        }
    }
    float4 m = ldc(S.mat + mi);
    int mtype = __float_as_int(m.w);
    f3 cd = mk3(m.x, m.y, m.z);
    const vrt_shading& sh = S.sh;
    int n_child = 0;
    f3 d0 = mk3(0.f, 0.f, 0.f), w0 = d0, d1 = d0, w1 = d0;
    // One Phong site for every material class: the lanes of a wave shade hits on different
    // materials, and with a call per class their ambient-occlusion and shadow traversals would run
    // one class after the other with a part of the lanes each.
    const bool metal = mtype == VRT_MAT_METAL, glass = mtype == VRT_MAT_DIELECTRIC;
    const bool matte = P.general && mtype == VRT_MAT_MATTE, transp = P.general && mtype == VRT_MAT_TRANSPARENT;
    // (ka, kd, ks, exp) of the hit's class from the block's table in LDS (stage_materials): the class
    // differs per lane, and a chain of selects over kernel arguments ends up as a table in scratch
    const float4 k = g_lds_matk[(matte || transp || metal || glass) ? (mtype & 7) : VRT_MAT_SOLID];
    const f3 A = phong_shade<BVH, RECORD>(S, P, point, n, h.prim, d, depth, k.x, k.y, k.z, k.w, cd, C, R, occ, matte);
    if (metal) {
        f3 wo = -d;
        float ndotwo = dot3(n, wo);
        f3 wi = (-wo) + (n * 2.f) * ndotwo;
        float nd = dot3(n, wi);
        // (cr * kr) / |n.wi|, the product from the host in the tree kernels (see ambient_L)
        f3 fr = (BVH ? mk3(S.metal_w[0], S.metal_w[1], S.metal_w[2])
                     : mk3(sh.metal_cr[0], sh.metal_cr[1], sh.metal_cr[2]) * sh.metal_kr) / fabsf(nd);
        n_child = 1; d0 = normalize3(wi); w0 = fr * nd;
    } else if (glass) {
        f3 wo = -d;
        float ndotwo = dot3(n, wo);
        f3 wi = (-wo) + (n * 2.f) * ndotwo;
        n_child = 1; d0 = normalize3(wi); w0 = mk3(1.f, 1.f, 1.f);
        // (P.general: the flag sends a frame to the kernels that follow two children per hit, and the
        // others need not carry this code, nor its wave-uniform eta in a register)
        if (P.general && (P.flags & VRT_FIX_DIELECTRIC)) {
            float eta = sh.glass_eta_in / sh.glass_eta_out;
            float ci = dot3(n, wo);
            f3 nn = n;
            if (ci < 0.f) { ci = -ci; nn = -n; eta = 1.f / eta; }
            float temp = 1.f - (1.f - ci * ci) / (eta * eta);
            if (!(temp < 0.f)) {
                float kr = fresnel_kr(n, d, sh.glass_eta_in, sh.glass_eta_out);
                float kt = 1.f - kr;
                float c2 = sqrtf(temp);
                f3 wt = ((-wo) / eta) - nn * (c2 - ci / eta);
                float ndotwi = dot3(n, wi), ndotwt = dot3(n, wt);
                float wr = (kr / fabsf(ndotwi)) * fabsf(ndotwi);
                float wtt = (kt / (eta * eta) / fabsf(ndotwt)) * fabsf(ndotwt);
                w0 = mk3(wr, wr, wr);
                n_child = 2; d1 = normalize3(wt); w1 = mk3(wtt, wtt, wtt);
            }
        }
    } else if (transp) {
        n_child = transparent_children(sh, n, d, d0, w0, d1, w1);
    }
    B.prim = h.prim; B.o = point; B.n_child = n_child; B.A = A; B.d0 = d0; B.w0 = w0; B.d1 = d1; B.w1 = w1;
}

template <bool BVH, bool RECORD>
__device__ __forceinline__ void shade_hit(const DevScene& S, const DevParams& P, Hit h, f3 o, f3 d, int depth,
                                          Counters& C, Recorder& R, Bounce& B)
{
    OccInline occ;
    occ.packet = BVH && (P.packet == 2 || (P.packet == 1 && depth == 0));
    shade_hit<BVH, RECORD>(S, P, h, o, d, depth, C, R, B, occ);
}

// RayTracerUnity.cs:486-517 + AntiAliasingStrategy.cs:89-90
__device__ __forceinline__ f3 camera_raw_dir(const DevCam& cam, int h, int v)
{
    float hScale = (float)v / (float)cam.width;
    float vScale = (float)h / (float)cam.height;
    f3 ho = mk3(cam.xx, cam.xy, cam.xz) * hScale;
    f3 vo = mk3(cam.yx, cam.yy, cam.yz) * vScale;
    return (mk3(cam.bx, cam.by, cam.bz) + ho) + vo;
}

// Util/RayTraceUtility.cs:409-422 through RayTracerUnity.cs:525-562 (the gamma step there is
// computed on a local and dropped).
__device__ __forceinline__ float4 process_color(f3 c)
{
    float m = fmaxf(c.x, fmaxf(c.y, c.z));
    if (m > 1.f) return make_float4(c.x / m, c.y / m, c.z / m, 1.f / m);
    return make_float4(c.x, c.y, c.z, 1.f);
}

// One channel as a Color32 holds it: round(255 * clamp01(c)), halves to even, NaN as 0.
__device__ __forceinline__ unsigned channel8(float c)
{
    return (unsigned)__builtin_rintf(fminf(fmaxf(c, 0.f), 1.f) * 255.f);   // fmaxf drops a NaN
}

// Stores the averaged radiance `fin` of pixel idx in the format the call asked for: the tone-mapped
// Color (4 floats), its Color32 (4 bytes, VRT_OUT_RGBA8), or the radiance itself (3 floats,
// VRT_OUT_LINEAR_RGB, tone mapping left to vrt_tonemap_device).
__device__ __forceinline__ void store_pixel(const DevParams& P, float4* rgba, size_t idx, f3 fin)
{
    if (P.flags & VRT_OUT_LINEAR_RGB) {
        float* o = (float*)rgba + 3 * idx;
        o[0] = fin.x; o[1] = fin.y; o[2] = fin.z;
    } else if (P.flags & VRT_OUT_RGBA8) {
        const float4 c = process_color(fin);
        ((unsigned*)rgba)[idx] = channel8(c.x) | (channel8(c.y) << 8) | (channel8(c.z) << 16) | (channel8(c.w) << 24);
    } else {
        rgba[idx] = process_color(fin);
    }
}

struct PendingRay { f3 o, d, T; int depth, skip, kind; };

// A path handed over before its next ray is cast: origin and direction of that ray, throughput and
// radiance so far, the depth the ray is cast at and the primitive it leaves.
struct PathState { f3 o, d, T, L; int depth, skip; bool pending; };

// VRT_TAIL_MAX > 0 (tree scenes, one child per hit): once at most that many lanes of a wave are still
// on a path, they leave the per-lane bounce loop with their paths unfinished, and the wave finishes
// them together in the single-loop state machine, whose one traversal site gives every query four
// lanes (intersect_bvh4_quads).  A per-lane traversal step costs the wave the same ~120 vector
// instructions whether 1 or 64 lanes take it; on the 10k-sphere frame 41 % of the turns of the bounce
// loop run with 16 lanes or fewer.  Built, bit-identical, and measured slower on MI355X whatever the
// threshold (10k spheres 4K x 16 spp: 24.6 ms without, 29.2 / 29.5 / 29.1 ms with 1 / 4 / 16; 1M
// triangles 2.07 -> 2.46 ms): the state machine's registers come on top of the frame kernel's (87
// spilled VGPRs instead of 22), and handing 41 % of the turns to it instead of none changes 0.1 ms.
// A study switch, 0 = off.
#ifndef VRT_TAIL_MAX
#define VRT_TAIL_MAX 0
#endif

// Tracer/WhittedTracer.cs:33-104 as an iterative loop: radiance is accumulated as
// sum(throughput * direct term) instead of unwinding a recursion; the second child of a
// refracting hit waits on a bounded per-lane stack.
template <bool BVH, bool RECORD, bool TWOCHILD, bool TAIL = false>
// shaded: the first hit has been shaded already (its Bounce), the loop starts at its children.
// TAIL: see VRT_TAIL_MAX; *ps receives the unfinished path (the function then returns early).
// L0, T0: radiance gathered and throughput reached before (o, d) was cast (a path taken up after its
// first bounce); the sums go on in the order of the path traced from the camera, so the bits agree.
__device__ __forceinline__ f3 trace_path(const DevScene& S, const DevParams& P, f3 o, f3 d, Hit first,
                                         Counters& C, Recorder& R, int depth0 = 0, const Bounce* shaded = nullptr,
                                         f3 L0 = {0.f, 0.f, 0.f}, f3 T0 = {1.f, 1.f, 1.f}, PathState* ps = nullptr)
{
    f3 L = L0;
    f3 T = T0;
    int depth = depth0, skip = -1, kind = VRT_RAY_PRIMARY;
    Hit h = first;
    bool have_first = true;
    bool have_B = shaded != nullptr;
    Bounce B;
    if (have_B) B = *shaded;
    PendingRay stack[TWOCHILD ? VRT_MAX_DEPTH + 2 : 1];
    int sp = 0;
    for (;;) {
#if VRT_STUDY_LIVE_HIST   // study build: how many lanes of the wave are still on a path, per turn of this loop
        {
            const unsigned long long m = __ballot(1);
            const int n = __popcll(m);
            const int bucket = n <= 1 ? 0 : n <= 2 ? 1 : n <= 4 ? 2 : n <= 8 ? 3 : n <= 16 ? 4 : n <= 32 ? 5 : n <= 48 ? 6 : 7;
            if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1)
                atomicAdd(&S.dbg[(blockIdx.x % VRT_CTR_SLOTS) * VRT_CTR_STRIDE + bucket], 1ull);
        }
#endif
        if constexpr (TAIL && !TWOCHILD) {
            // the lanes still in this loop are the active ones here
            if (!have_B && !have_first && __popcll(__ballot(1)) <= VRT_TAIL_MAX) {
                ps->o = o; ps->d = d; ps->T = T; ps->L = L; ps->depth = depth; ps->skip = skip; ps->pending = true;
                return L;
            }
        }
        if (!have_B) {
            if (!have_first) {
                h.prim = -1; h.t = INFINITY; h.aux = 0;
                if (!is_zero3(d)) {
                    if (kind == VRT_RAY_REFLECT) VRT_COUNT(C, reflect, VRT_RAY_REFLECT); else VRT_COUNT(C, transmit, VRT_RAY_TRANSMIT);
                    h = intersect<BVH, false>(S, o, d, P.range, skip, INFINITY, BVH && P.packet == 2);
                    record_seg<RECORD>(R, o, d, h, kind, depth, P.range);
                }
            }
            if (h.prim < 0) L = L + T * sky_color(d);
            else { shade_hit<BVH, RECORD>(S, P, h, o, d, depth, C, R, B); have_B = true; }
        }
        have_first = false;
        bool cont = false;
        if (have_B) {
            have_B = false;
            L = L + T * B.A;
            if (B.n_child > 0 && depth + 1 <= P.max_depth) {
                if constexpr (TWOCHILD) if (B.n_child == 2 && sp < VRT_MAX_DEPTH + 2) {
                    PendingRay& q = stack[sp++];
                    q.o = B.o; q.d = B.d1; q.T = T * B.w1; q.depth = depth + 1; q.skip = B.prim;
                    q.kind = VRT_RAY_TRANSMIT;
                }
                o = B.o; d = B.d0; T = T * B.w0; depth = depth + 1; skip = B.prim; kind = VRT_RAY_REFLECT;
                cont = true;
            }
        }
        if (!cont) {
            if constexpr (TWOCHILD) {
                if (sp == 0) break;
                PendingRay& q = stack[--sp];
                o = q.o; d = q.d; T = q.T; depth = q.depth; skip = q.skip; kind = q.kind;
            } else {
                break;
            }
        }
    }
    return L;
}

// RayTracerUnity.cs:578-587: anti-aliasing sample i of pixel (h, v): its radiance, or zero when
// the primary any-hit pre-check fails (sky with VRT_FIX_PRIMARY_SKY).  `off` = the pixel's row
// of the offset table or null.
template <bool BVH, bool RECORD, bool TWOCHILD>
__device__ __forceinline__ f3 trace_sample_dev(const DevScene& S, const DevCam& cam, const DevParams& P, f3 raw,
                                               const float2* off, int i, int& first_prim, Counters& C, Recorder& R,
                                               bool live = true, bool wave = false, PathState* ps = nullptr)
{
    // wave: all 64 lanes of the wave are here (live tells which hold a sample), so the camera rays
    // can be traced with the wave-cooperative forms
    f3 O = mk3(cam.ox, cam.oy, cam.oz);
    float ox = 0.f, oy = 0.f;
    if (off && live) { float2 q = off[i]; ox = q.x; oy = q.y; }
    f3 d = normalize3(mk3(raw.x + ox, raw.y + oy, raw.z));
    R.sample = i;
    Hit first; first.prim = -1; first.t = INFINITY; first.aux = 0;
    const bool cast = live && !is_zero3(d);
    // one closest-hit query answers both the any-hit pre-check (RayTracerUnity.cs:582)
    // and the primary TraceRay (:584); they are counted as the two queries they replace
    if (cast) VRT_COUNT(C, precheck, VRT_RAY_PRECHECK);
    if (BVH && wave) first = intersect_primary_bvh(S, O, d, P.range, cast, P.packet != 0, P.beam != 0);
    else if (cast) first = intersect<BVH, false, true>(S, O, d, P.range, -1, INFINITY, BVH && P.packet);
    first_prim = first.prim;
    if constexpr (BVH && !RECORD) {
        if (wave && P.beam == 1) {
            // The hits' ambient-occlusion rays and their shadow rays towards the first light are
            // traced by the wave together: one shading pass only asks which queries there are (and
            // counts them), the wave answers them, a second pass shades with the answers.
            const bool hit = live && first.prim >= 0;
            OccCapture cap;
            cap.packet = P.packet != 0;
            f3 point = O;
            if (hit) {
                VRT_COUNT(C, primary, VRT_RAY_PRIMARY);
                Bounce unused;
                shade_hit<BVH, false>(S, P, first, O, d, 0, C, R, unused, cap);
                point = unused.o;
            }
            unsigned bits = cap.bits;
            if (occluded_wave(S, hit && (cap.has & 1u), point, cap.d0, cap.maxd0, cap.tlim0, first.prim, P.packet != 0)) bits |= 1u;
            if (occluded_wave(S, hit && (cap.has & 2u), point, cap.d1, cap.maxd1, cap.tlim1, first.prim, P.packet != 0)) bits |= 2u;
            if constexpr (VRT_BEAM_BOUNCE && !TWOCHILD) {
                // The first reflections of the wave's hits are cast together as well (neighbouring
                // surface points mirror into neighbouring directions), and each lane takes its path
                // up from there.
                Bounce B;
                B.n_child = 0; B.o = O; B.d0 = d; B.prim = -1;
                if (hit) {
                    OccBits answers; answers.bits = bits;
                    shade_hit<BVH, false>(S, P, first, O, d, 0, C, R, B, answers);
                }
                const bool go = hit && B.n_child > 0 && 1 <= P.max_depth;
                const bool cast1 = go && !is_zero3(B.d0);
                if (cast1) VRT_COUNT(C, reflect, VRT_RAY_REFLECT);
                const Hit h1 = intersect_bounce_wave(S, cast1, B.o, B.d0, P.range, B.prim, P.packet == 2);
                if (hit) {
                    const f3 L0 = mk3(0.f, 0.f, 0.f) + mk3(1.f, 1.f, 1.f) * B.A;
                    if (!go) return L0;
                    if constexpr (VRT_TAIL_MAX > 0) {
                        if (ps) return trace_path<BVH, false, TWOCHILD, true>(S, P, B.o, B.d0, h1, C, R, 1, nullptr, L0, mk3(1.f, 1.f, 1.f) * B.w0, ps);
                    }
                    return trace_path<BVH, false, TWOCHILD>(S, P, B.o, B.d0, h1, C, R, 1, nullptr, L0, mk3(1.f, 1.f, 1.f) * B.w0);
                }
                if (!live) return mk3(0.f, 0.f, 0.f);
                if (P.flags & VRT_FIX_PRIMARY_SKY) return sky_color(d);
                return mk3(0.f, 0.f, 0.f);
            }
            if (!live) return mk3(0.f, 0.f, 0.f);
            if (hit) {
                OccBits answers; answers.bits = bits;
                Bounce B;
                shade_hit<BVH, false>(S, P, first, O, d, 0, C, R, B, answers);
                return trace_path<BVH, false, TWOCHILD>(S, P, O, d, first, C, R, 0, &B);
            }
            if (P.flags & VRT_FIX_PRIMARY_SKY) return sky_color(d);
            return mk3(0.f, 0.f, 0.f);
        }
    }
    if (!live) return mk3(0.f, 0.f, 0.f);
    if (first.prim >= 0) {
        VRT_COUNT(C, primary, VRT_RAY_PRIMARY);
        record_seg<RECORD>(R, O, d, first, VRT_RAY_PRIMARY, 0, P.range);
        return trace_path<BVH, RECORD, TWOCHILD>(S, P, O, d, first, C, R);
    }
    if (P.flags & VRT_FIX_PRIMARY_SKY) return sky_color(d);
    return mk3(0.f, 0.f, 0.f);
}

// RayTracerUnity.cs:565-597: all samples of one pixel, averaged and tone-mapped.
// sample_sel < 0: every sample; otherwise just that one (single-step mode).
template <bool BVH, bool RECORD, bool TWOCHILD>
__device__ __forceinline__ float4 trace_pixel_dev(const DevScene& S, const DevCam& cam, const DevParams& P,
                                                  int h, int v, int sample_sel, int& hit_id,
                                                  Counters& C, Recorder& R)
{
    f3 raw = camera_raw_dir(cam, h, v);
    unsigned pixel = (unsigned)(h * cam.width + v);
    unsigned set = pixel_set_dev(P.seed, pixel, (unsigned)P.n_sets);
    const float2* off = P.offsets ? P.offsets + (size_t)set * P.spp : nullptr;
    f3 sum = mk3(0.f, 0.f, 0.f);
    hit_id = -1;
    int i0 = sample_sel < 0 ? 0 : sample_sel, i1 = sample_sel < 0 ? P.spp : sample_sel + 1;
    for (int i = i0; i < i1; ++i) {
        int fp;
        f3 c = trace_sample_dev<BVH, RECORD, TWOCHILD>(S, cam, P, raw, off, i, fp, C, R);
        if (i == 0) hit_id = fp;
        sum = sum + c;
    }
    f3 fin = sum / (float)(i1 - i0);
    return process_color(fin);
}

// ------------------------------------------------------------------ single-loop tracer
//
// The same per-pixel routine as trace_pixel_dev, organised as a per-lane state machine around
// ONE intersection call site: primary, reflection, transmission, ambient-occlusion and shadow
// queries of the 64 lanes of a wave all run through the same primitive loop / BVH traversal,
// whatever stage each lane's pixel is in.  That keeps the kernel small (one copy of the
// intersection code) and lets lanes in different stages share the expensive part instead of
// serialising it.  Arithmetic and its order per lane are those of trace_pixel_dev.

// Per-lane-indexed scene data staged in LDS: lanes of one wave can be at different lights and
// on different material classes, which scalar kernel arguments cannot serve.
struct LdsScene {
    DevLight lights[VRT_MAX_LIGHTS];
    float4 matk[8]; // (ka, kd, ks, exp) by VRT_MAT_* class
};

// Fills the LDS copy: lights one dword per lane from device memory, material constants from
// the kernel arguments (whole block must call it).
__device__ __forceinline__ void stage_scene(const DevScene& S, LdsScene* lds)
{
    const int words = S.n_lights * (int)(sizeof(DevLight) / 4);
    const unsigned* src = (const unsigned*)S.lights;
    unsigned* dst = (unsigned*)lds->lights;
    for (int w = threadIdx.x; w < words; w += blockDim.x) dst[w] = src[w];
    if (threadIdx.x == 0) {
        const vrt_shading& sh = S.sh;
        lds->matk[0] = make_float4(0.f, 0.f, 0.f, 1.f);
        lds->matk[VRT_MAT_SOLID] = make_float4(sh.solid_ka, sh.solid_kd, sh.solid_ks, sh.solid_exp);
        lds->matk[VRT_MAT_METAL] = make_float4(sh.metal_ka, sh.metal_kd, sh.metal_ks, sh.metal_exp);
        lds->matk[VRT_MAT_DIELECTRIC] = make_float4(sh.glass_ka, sh.glass_kd, sh.glass_ks, sh.glass_exp);
        lds->matk[VRT_MAT_MATTE] = make_float4(sh.matte_ka, sh.matte_kd, 0.f, 1.f);
        lds->matk[VRT_MAT_TRANSPARENT] = make_float4(sh.transp_ka, sh.transp_kd, sh.transp_ks, sh.transp_exp);
        lds->matk[6] = lds->matk[7] = lds->matk[0];
    }
    __syncthreads();
}

enum { ST_SAMPLE = 0, ST_CLOSEST = 1, ST_AO = 2, ST_LIGHTS = 3, ST_SHADOW = 4, ST_BOUNCE = 5, ST_NEXT = 6 };

struct MatK { float ka, kd, ks, ex; };
__device__ __forceinline__ MatK material_consts(const LdsScene* lds, int mtype)
{
    float4 v = lds->matk[mtype & 7];
    MatK k; k.ka = v.x; k.kd = v.y; k.ks = v.z; k.ex = v.w;
    return k;
}

// Direction to light `l` from `point` (Light/PointLight.cs:23-34, Light/DirectionalLight.cs:22-25).
__device__ __forceinline__ f3 light_wi(const DevLight& l, f3 point)
{
    f3 lp = mk3(l.vx, l.vy, l.vz);
    if (l.type == VRT_LIGHT_POINT) {
        f3 tmp = lp - point;
        float len = sqrt_ieee(tmp.x * tmp.x + tmp.y * tmp.y + tmp.z * tmp.z);
        return tmp / len;
    }
    return lp;
}

// One unshadowed light's term of Material/PhongMaterial.cs:77-84.
__device__ __forceinline__ f3 light_term(const DevLight& l, f3 wi, float ndotwi, f3 n, f3 wo, f3 cd, MatK k, bool matte = false)
{
    f3 diff = (cd * k.kd) * 0.3183098861837906715f;
    if (matte) return (diff * (mk3(l.cr, l.cg, l.cb) * l.ls)) * ndotwi;   // Material/MatteMaterial.cs:69
    float spec = 0.f;
    if (!(k.ks == 0.f && k.ex >= 0.f)) { // see phong_shade
        float nda = dot3(n, wo);
        f3 r = mk3(-wo.x + 2.f * n.x * nda, -wo.y + 2.f * n.y * nda, -wo.z + 2.f * n.z * nda);
        float rdb = dot3(r, wi);
        if (rdb > 0.f) spec = (1.f * k.ks) * powf(rdb, k.ex);
    }
    f3 lam = mk3(diff.x + spec, diff.y + spec, diff.z + spec);
    f3 world = (mk3(l.cr, l.cg, l.cb) * l.ls) * ndotwi;
    return lam * world;
}

// Samples i0 .. i1-1 of pixel (ph, pv), their radiances added up in order.  `lds` = the block's
// LdsScene filled by stage_scene().  Lanes with i0 >= i1 ride along without a sample.
template <bool BVH, bool RECORD, bool TWOCHILD>
__device__ __forceinline__ f3 trace_samples_sm(const DevScene& S, const DevCam& cam, const DevParams& P,
                                               const LdsScene* lds, int ph, int pv, int i0, int i1,
                                               int& hit_id, Counters& C, Recorder& R, const PathState* resume = nullptr)
{
    // resume: instead of samples, a lane with resume->pending (and i1 = i0 + 1) takes up the path
    // handed over there, at the ray that is cast next; its radiance is what the function returns
    const vrt_shading& sh = S.sh;
    const f3 O = mk3(cam.ox, cam.oy, cam.oz);
    const f3 raw = camera_raw_dir(cam, ph, pv);
    const unsigned pixel = (unsigned)(ph * cam.width + pv);
    const unsigned set = pixel_set_dev(P.seed, pixel, (unsigned)P.n_sets);
    const float2* off = P.offsets ? P.offsets + (size_t)set * P.spp : nullptr;
    f3 sum = mk3(0.f, 0.f, 0.f);
    hit_id = -1;
    int i = i0;
    // path state
    f3 T = mk3(1.f, 1.f, 1.f), L = mk3(0.f, 0.f, 0.f);
    int depth = 0, kind = VRT_RAY_PRIMARY;
    // pending query; after a hit qo stays at the hit point for the shading queries and the child rays
    f3 qo = O, qd = mk3(0.f, 0.f, 0.f), rd = qd; // rd = direction of the ray being shaded
    float qmax = P.range, qlim = INFINITY, dist = 0.f;
    int qskip = -1;
    bool qany = false;
    // shading context of the current hit
    f3 n = qd, cd = qd, A = qd;
    int mtype = 0, prim = -1, light = 0;
    Hit h; h.prim = -1; h.t = INFINITY; h.aux = 0;
    PendingRay stack[TWOCHILD ? VRT_MAX_DEPTH + 2 : 1];
    int sp = 0;
    int st = ST_SAMPLE;
    bool done = false;
    bool seeded = false;   // the first turn starts with a query already waiting
    if (resume != nullptr && i0 < i1) {
        T = resume->T; L = resume->L; depth = resume->depth; kind = VRT_RAY_REFLECT;
        qo = resume->o; qd = resume->d; qskip = resume->skip;
        st = ST_CLOSEST;
        if (!is_zero3(qd)) { VRT_COUNT(C, reflect, VRT_RAY_REFLECT); seeded = true; }
    }

    for (;;) {
        bool query = seeded;
        seeded = false;
        while (!query && !done) {
            if (st == ST_SAMPLE) {
                if (i >= i1) { done = true; continue; }
                float ox = 0.f, oy = 0.f;
                if (off) { float2 q = off[i]; ox = q.x; oy = q.y; }
                qo = O; qd = normalize3(mk3(raw.x + ox, raw.y + oy, raw.z));
                R.sample = i;
                T = mk3(1.f, 1.f, 1.f); L = mk3(0.f, 0.f, 0.f); depth = 0; kind = VRT_RAY_PRIMARY;
                qmax = P.range; qlim = INFINITY; qskip = -1; qany = false;
                h.prim = -1; h.t = INFINITY; h.aux = 0;
                st = ST_CLOSEST;
                if (!is_zero3(qd)) { VRT_COUNT(C, precheck, VRT_RAY_PRECHECK); query = true; }
            } else if (st == ST_CLOSEST) {
                // result of the closest-hit query for ray (qo, qd) at `depth`
                // (Tracer/WhittedTracer.cs:33-104; at depth 0 also RayTracerUnity.cs:582-586)
                if (depth == 0) {
                    if (i == i0 && i == 0) hit_id = h.prim;
                    if (h.prim >= 0) { VRT_COUNT(C, primary, VRT_RAY_PRIMARY); record_seg<RECORD>(R, qo, qd, h, VRT_RAY_PRIMARY, 0, P.range); }
                } else if (!is_zero3(qd)) {
                    record_seg<RECORD>(R, qo, qd, h, kind, depth, P.range);
                }
                if (h.prim < 0) {
                    if (depth == 0) { if (P.flags & VRT_FIX_PRIMARY_SKY) L = sky_color(qd); }
                    else L = L + T * sky_color(qd);
                    st = ST_NEXT;
                } else {
                    f3 point;
                    hit_geometry(S, h, qo, qd, point, n);
                    const int mi = S.prim_mat[h.prim];
                    const int tx = S.mat_tex ? S.mat_tex[mi] : -1;
                    if (tx >= 0) {   // Tracer/WhittedTracer.cs:55-63: texel, unshaded, path ends
                        L = L + T * texel_at_hit(S, tx, h, qo, qd);
                        st = ST_NEXT;
                        continue;
                    }
                    float4 m = S.mat[mi];
                    mtype = __float_as_int(m.w);
                    cd = mk3(m.x, m.y, m.z);
                    prim = h.prim; rd = qd; qo = point;
                    // Light/AmbientOccluder.cs:56-85, see ambient_L
                    bool cast = true;
                    if (!(P.flags & VRT_FIX_AO_BASIS))
                        cast = !(n.x < 0.f || n.z < 0.f || (n.y < 0.f && n.y != -1.f));
                    f3 dir = normalize3(mk3(n.x + 0.f, n.y + 0.f, n.z + 0.f));
                    if (cast && !is_zero3(dir)) {
                        VRT_COUNT(C, ao, VRT_RAY_AO);
                        qd = dir; qmax = P.range; qlim = INFINITY; qskip = prim; qany = !RECORD;
                        st = ST_AO; query = true;
                    } else {
                        MatK k = material_consts(lds, mtype);
                        A = (cd * k.ka) * mk3(S.ao_free[0], S.ao_free[1], S.ao_free[2]);
                        light = 0; st = ST_LIGHTS;
                    }
                }
            } else if (st == ST_AO) {
                record_seg<RECORD>(R, qo, qd, h, VRT_RAY_AO, depth, P.range);
                MatK k = material_consts(lds, mtype);
                f3 aoL = mk3(S.ao_free[0], S.ao_free[1], S.ao_free[2]);
                if (h.prim >= 0) aoL = mk3(S.ao_occl[0], S.ao_occl[1], S.ao_occl[2]);
                A = (cd * k.ka) * aoL;
                light = 0; st = ST_LIGHTS;
            } else if (st == ST_LIGHTS) {
                // Material/PhongMaterial.cs:61-86, one light per step
                if (light >= S.n_lights) { st = ST_BOUNCE; continue; }
                const DevLight& l = lds->lights[light];
                f3 wi = light_wi(l, qo);
                float ndotwi = dot3(n, wi);
                bool shadow_query = false;
                if (ndotwi > 0.f) {
                    if (l.cast && mtype != VRT_MAT_MATTE) {   // the matte material asks for no shadow test
                        f3 sd = normalize3(wi);
                        f3 loc = l.type == VRT_LIGHT_POINT ? mk3(l.vx, l.vy, l.vz) : mk3(l.lx, l.ly, l.lz);
                        dist = mag3(loc - qo);
                        if (!is_zero3(sd)) {
                            VRT_COUNT(C, shadow, VRT_RAY_SHADOW);
                            qd = sd; qmax = 30.f; qlim = RECORD ? INFINITY : dist; qskip = prim; qany = !RECORD;
                            st = ST_SHADOW; query = true; shadow_query = true;
                        }
                    }
                    if (!shadow_query)
                        A = A + light_term(l, wi, ndotwi, n, -rd, cd, material_consts(lds, mtype), mtype == VRT_MAT_MATTE);
                }
                if (!shadow_query) light++;
            } else if (st == ST_SHADOW) {
                record_seg<RECORD>(R, qo, qd, h, VRT_RAY_SHADOW, depth, 30.f);
                bool in_shadow = h.prim >= 0 && h.t < dist;
                if (!in_shadow) {
                    const DevLight& l = lds->lights[light];
                    f3 wi = light_wi(l, qo);
                    A = A + light_term(l, wi, dot3(n, wi), n, -rd, cd, material_consts(lds, mtype));
                }
                light++; st = ST_LIGHTS;
            } else if (st == ST_BOUNCE) {
                // Material/ReflectiveMaterial.cs:25-42, Material/DielectricMaterial.cs:56-146
                L = L + T * A;
                int n_child = 0;
                f3 d0 = mk3(0.f, 0.f, 0.f), w0 = d0, d1 = d0, w1 = d0;
                if (TWOCHILD && mtype == VRT_MAT_TRANSPARENT) {
                    n_child = transparent_children(sh, n, rd, d0, w0, d1, w1);
                } else if (mtype == VRT_MAT_METAL || mtype == VRT_MAT_DIELECTRIC) {
                    f3 wo = -rd;
                    float ndotwo = dot3(n, wo);
                    f3 wi = (-wo) + (n * 2.f) * ndotwo;
                    n_child = 1; d0 = normalize3(wi);
                    if (mtype == VRT_MAT_METAL) {
                        float nd = dot3(n, wi);
                        f3 fr = mk3(S.metal_w[0], S.metal_w[1], S.metal_w[2]) / fabsf(nd);
                        w0 = fr * nd;
                    } else {
                        w0 = mk3(1.f, 1.f, 1.f);
                        if (TWOCHILD && (P.flags & VRT_FIX_DIELECTRIC)) {
                            float eta = sh.glass_eta_in / sh.glass_eta_out;
                            float ci = dot3(n, wo);
                            f3 nn = n;
                            if (ci < 0.f) { ci = -ci; nn = -n; eta = 1.f / eta; }
                            float temp = 1.f - (1.f - ci * ci) / (eta * eta);
                            if (!(temp < 0.f)) {
                                float kr = fresnel_kr(n, rd, sh.glass_eta_in, sh.glass_eta_out);
                                float kt = 1.f - kr;
                                float c2 = sqrtf(temp);
                                f3 wt = ((-wo) / eta) - nn * (c2 - ci / eta);
                                float ndotwi = dot3(n, wi), ndotwt = dot3(n, wt);
                                float wr = (kr / fabsf(ndotwi)) * fabsf(ndotwi);
                                float wtt = (kt / (eta * eta) / fabsf(ndotwt)) * fabsf(ndotwt);
                                w0 = mk3(wr, wr, wr);
                                n_child = 2; d1 = normalize3(wt); w1 = mk3(wtt, wtt, wtt);
                            }
                        }
                    }
                }
                if (n_child > 0 && depth + 1 <= P.max_depth) {
                    if (TWOCHILD && n_child == 2 && sp < VRT_MAX_DEPTH + 2) {
                        PendingRay& q = stack[sp++];
                        q.o = qo; q.d = d1; q.T = T * w1; q.depth = depth + 1; q.skip = prim; q.kind = VRT_RAY_TRANSMIT;
                    }
                    qd = d0; T = T * w0; depth = depth + 1; qskip = prim; kind = VRT_RAY_REFLECT;
                    qmax = P.range; qlim = INFINITY; qany = false;
                    h.prim = -1; h.t = INFINITY; h.aux = 0;
                    st = ST_CLOSEST;
                    if (!is_zero3(qd)) { VRT_COUNT(C, reflect, VRT_RAY_REFLECT); query = true; }
                } else {
                    st = ST_NEXT;
                }
            } else { // ST_NEXT: this ray's subtree is finished
                if (TWOCHILD && sp > 0) {
                    PendingRay& q = stack[--sp];
                    qo = q.o; qd = q.d; T = q.T; depth = q.depth; qskip = q.skip; kind = q.kind;
                    qmax = P.range; qlim = INFINITY; qany = false;
                    h.prim = -1; h.t = INFINITY; h.aux = 0;
                    st = ST_CLOSEST;
                    if (!is_zero3(qd)) { if (kind == VRT_RAY_REFLECT) VRT_COUNT(C, reflect, VRT_RAY_REFLECT); else VRT_COUNT(C, transmit, VRT_RAY_TRANSMIT); query = true; }
                } else {
                    sum = sum + L;
                    i++; st = ST_SAMPLE;
                }
            }
        }
        if (!__any(query)) break;   // wave-uniform: every lane has finished its samples
        if constexpr (BVH) {
            const bool coherent = P.packet != 0 && __all(!query || depth == 0);
            intersect_bvh4_wave(S, query, qo, qd, qmax, qlim, qskip, qany, coherent, P.quad_max, h);
        } else {
            if (query) h = intersect_rt<BVH>(S, qo, qd, qmax, qskip, qlim, qany);
        }
    }
    return sum;
}

// RayTracerUnity.cs:565-597 on the state machine: all samples of one pixel (sample_sel < 0) or
// just that one, averaged and tone-mapped.
template <bool BVH, bool RECORD, bool TWOCHILD>
__device__ __forceinline__ float4 trace_pixel_sm(const DevScene& S, const DevCam& cam, const DevParams& P,
                                                 const LdsScene* lds, int ph, int pv, int sample_sel,
                                                 int& hit_id, Counters& C, Recorder& R)
{
    const int i0 = sample_sel < 0 ? 0 : sample_sel, i1 = sample_sel < 0 ? P.spp : sample_sel + 1;
    f3 sum = trace_samples_sm<BVH, RECORD, TWOCHILD>(S, cam, P, lds, ph, pv, i0, i1, hit_id, C, R);
    f3 fin = sum / (float)(i1 - i0);
    return process_color(fin);
}
