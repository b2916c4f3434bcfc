/*
 * vrt_oracle.c — CPU restatement of VisualRayTrace's per-pixel tracing loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (libvrt.so, raytracing_amd/) links,
 * loads or calls this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker / CPU baseline.
 *
 * PARITY UNPINNED at the intersection boundary: the reference resolves every ray with
 * Unity's closed Physics.Raycast (PhysX) and ships no golden vectors, so the analytic
 * sphere / box / triangle tests below are this oracle's own statement of Unity's
 * documented raycast semantics.  Everything above that boundary follows the cited C#
 * line by line, in IEEE float32 with no FMA contraction (build with -ffp-contract=off).
 * The one known answer the reference's tests hold (sky colour of the zero vector,
 * Assets/Scripts/Tests/RayTracerTests.cs:51-54) is checked in tests/.
 *
 * Citations are relative to VisualRayTrace/Assets/Scripts/RayTracer/.
 * It is written recursively, like the C#, on purpose: the GPU code is iterative.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/vrt.h"

/* ------------------------------------------------------------------ vector / colour */

typedef struct { float x, y, z; } v3;
typedef struct { float r, g, b, a; } col;

static v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static col C(float r, float g, float b, float a) { col c = {r, g, b, a}; return c; }
static v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static v3 vmulf(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static v3 vdivf(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
/* UnityEngine.Vector3.Dot / Cross / Magnitude / Normalize (public C# reference source). */
static float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 vcross(v3 a, v3 b)
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static float vmag(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static v3 vnormalize(v3 a)
{
    float m = vmag(a);
    if (m > 1e-5f) return vdivf(a, m);
    return V(0.f, 0.f, 0.f);
}
static col cadd(col a, col b) { return C(a.r + b.r, a.g + b.g, a.b + b.b, a.a + b.a); }
static col cmul(col a, col b) { return C(a.r * b.r, a.g * b.g, a.b * b.b, a.a * b.a); }
static col cmulf(col a, float s) { return C(a.r * s, a.g * s, a.b * s, a.a * s); }
static col cdivf(col a, float s) { return C(a.r / s, a.g / s, a.b / s, a.a / s); }
/* Mathf.Pow = (float)Math.Pow((double)f, (double)p) */
static float mpow(float f, float p) { return (float)pow((double)f, (double)p); }

/* ------------------------------------------------------------------ scene */

typedef struct { float lo[3], hi[3]; int left, right; int first, count; } bnode;

typedef struct {
    const vrt_scene_desc* d;
    uint32_t n_prims;
    /* CPU BVH (median split); used when n_prims > BVH_THRESHOLD */
    bnode* nodes;
    int n_nodes;
    int* order;
    int use_bvh;
} oscene;

typedef struct { int hit; float t; int prim; v3 point; v3 normal; } ohit;

/* A ray as UnityEngine.Ray stores it: direction normalised by the constructor. */
typedef struct { v3 o, d; } oray;
static oray make_ray(v3 o, v3 d) { oray r; r.o = o; r.d = vnormalize(d); return r; }

/* ---- analytic colliders.  Unity semantics stated here: the closest collider with
 * 0 <= distance <= maxDistance is reported; a collider that contains the ray origin is
 * not; mesh triangles are single sided (ProjectSettings/DynamicsManager.asset:14
 * m_QueriesHitBackfaces 0); ties go to the lowest primitive id. */

static int hit_sphere(const vrt_sphere* s, oray r, float maxd, float* t_out)
{
    v3 c = V(s->center[0], s->center[1], s->center[2]);
    v3 oc = vsub(r.o, c);
    float b = vdot(oc, r.d);
    float cc = vdot(oc, oc) - s->radius * s->radius;
    if (!(cc > 0.f)) return 0;
    if (!(b < 0.f)) return 0;
    float disc = b * b - cc;
    if (!(disc >= 0.f)) return 0;
    float t = -b - sqrtf(disc);
    if (!(t >= 0.f && t <= maxd)) return 0;
    *t_out = t;
    return 1;
}

static int hit_box(const vrt_box* bx, oray r, float maxd, float* t_out, int* axis_out, float* sgn_out)
{
    v3 c = V(bx->center[0], bx->center[1], bx->center[2]);
    v3 p = vsub(r.o, c);
    float lo[3], ld[3];
    int inside = 1;
    for (int k = 0; k < 3; ++k) {
        v3 a = V(bx->axis[3 * k], bx->axis[3 * k + 1], bx->axis[3 * k + 2]);
        lo[k] = vdot(p, a);
        ld[k] = vdot(r.d, a);
        if (!(fabsf(lo[k]) <= bx->half[k])) inside = 0;
    }
    if (inside) return 0;
    float tn = -INFINITY, tf = INFINITY, sgn = 0.f;
    int axis = -1;
    for (int k = 0; k < 3; ++k) {
        float h = bx->half[k];
        if (ld[k] == 0.f) {
            if (fabsf(lo[k]) > h) return 0;
            continue;
        }
        float inv = 1.f / ld[k];
        float t1 = (-h - lo[k]) * inv;
        float t2 = (h - lo[k]) * inv;
        float nr, fr, s;
        if (ld[k] > 0.f) { nr = t1; fr = t2; s = -1.f; } else { nr = t2; fr = t1; s = 1.f; }
        if (nr > tn) { tn = nr; axis = k; sgn = s; }
        if (fr < tf) tf = fr;
    }
    if (axis < 0) return 0;
    if (!(tn <= tf)) return 0;
    if (!(tn >= 0.f && tn <= maxd)) return 0;
    *t_out = tn; *axis_out = axis; *sgn_out = sgn;
    return 1;
}

static void tri_verts(const vrt_scene_desc* d, uint32_t ti, v3* v0, v3* v1, v3* v2)
{
    const uint32_t* ix = d->indices + 3 * (size_t)ti;
    const float* a = d->vertices + 3 * (size_t)ix[0];
    const float* b = d->vertices + 3 * (size_t)ix[1];
    const float* c = d->vertices + 3 * (size_t)ix[2];
    *v0 = V(a[0], a[1], a[2]); *v1 = V(b[0], b[1], b[2]); *v2 = V(c[0], c[1], c[2]);
}

static int hit_tri(v3 v0, v3 v1, v3 v2, oray r, float maxd, float* t_out)
{
    v3 e1 = vsub(v1, v0), e2 = vsub(v2, v0);
    v3 pvec = vcross(r.d, e2);
    float det = vdot(e1, pvec);
    if (!(det > 0.f)) return 0;
    v3 tvec = vsub(r.o, v0);
    float u = vdot(tvec, pvec);
    if (u < 0.f || u > det) return 0;
    v3 qvec = vcross(tvec, e1);
    float v = vdot(r.d, qvec);
    if (v < 0.f || u + v > det) return 0;
    float t = vdot(e2, qvec) / det;
    if (!(t >= 0.f && t <= maxd)) return 0;
    *t_out = t;
    return 1;
}

/* Tests primitive `id`; on a hit strictly better than *best (smaller t, or equal t and
 * smaller id) updates it. */
static void test_prim(const oscene* s, int id, oray r, float maxd, ohit* best)
{
    const vrt_scene_desc* d = s->d;
    float t;
    if ((uint32_t)id < d->n_spheres) {
        const vrt_sphere* sp = &d->spheres[id];
        if (!hit_sphere(sp, r, maxd, &t)) return;
        if (best->hit && !(t < best->t || (t == best->t && id < best->prim))) return;
        best->hit = 1; best->t = t; best->prim = id;
        best->point = vadd(r.o, vmulf(r.d, t));
        best->normal = vnormalize(vsub(best->point, V(sp->center[0], sp->center[1], sp->center[2])));
        return;
    }
    uint32_t bi = (uint32_t)id - d->n_spheres;
    if (bi < d->n_boxes) {
        const vrt_box* bx = &d->boxes[bi];
        int axis; float sgn;
        if (!hit_box(bx, r, maxd, &t, &axis, &sgn)) return;
        if (best->hit && !(t < best->t || (t == best->t && id < best->prim))) return;
        best->hit = 1; best->t = t; best->prim = id;
        best->point = vadd(r.o, vmulf(r.d, t));
        best->normal = vmulf(V(bx->axis[3 * axis], bx->axis[3 * axis + 1], bx->axis[3 * axis + 2]), sgn);
        return;
    }
    uint32_t ti = bi - d->n_boxes;
    v3 v0, v1, v2;
    tri_verts(d, ti, &v0, &v1, &v2);
    if (!hit_tri(v0, v1, v2, r, maxd, &t)) return;
    if (best->hit && !(t < best->t || (t == best->t && id < best->prim))) return;
    best->hit = 1; best->t = t; best->prim = id;
    best->point = vadd(r.o, vmulf(r.d, t));
    best->normal = vnormalize(vcross(vsub(v1, v0), vsub(v2, v0)));
}

/* ---- CPU BVH: top-down median split, conservative slab test.  Only an accelerator:
 * the result equals the brute-force loop because acceptance is by (t, id). */

#define BVH_THRESHOLD 64
#define BVH_LEAF 4

static void prim_bounds(const vrt_scene_desc* d, int id, float lo[3], float hi[3])
{
    if ((uint32_t)id < d->n_spheres) {
        const vrt_sphere* s = &d->spheres[id];
        for (int k = 0; k < 3; ++k) { lo[k] = s->center[k] - s->radius; hi[k] = s->center[k] + s->radius; }
    } else if ((uint32_t)id - d->n_spheres < d->n_boxes) {
        const vrt_box* b = &d->boxes[id - d->n_spheres];
        for (int k = 0; k < 3; ++k) {
            float e = fabsf(b->axis[k]) * b->half[0] + fabsf(b->axis[3 + k]) * b->half[1] +
                      fabsf(b->axis[6 + k]) * b->half[2];
            lo[k] = b->center[k] - e; hi[k] = b->center[k] + e;
        }
    } else {
        v3 a, b, c;
        tri_verts(d, id - d->n_spheres - d->n_boxes, &a, &b, &c);
        float p[3][3] = {{a.x, a.y, a.z}, {b.x, b.y, b.z}, {c.x, c.y, c.z}};
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(p[0][k], fminf(p[1][k], p[2][k]));
            hi[k] = fmaxf(p[0][k], fmaxf(p[1][k], p[2][k]));
        }
    }
    for (int k = 0; k < 3; ++k) { /* pad against rounding in the exact tests */
        float e = 1e-4f * (1.f + fmaxf(fabsf(lo[k]), fabsf(hi[k])));
        lo[k] -= e; hi[k] += e;
    }
}

typedef struct { float c[3]; float lo[3], hi[3]; int id; } bprim;
static int g_axis;
static int cmp_bprim(const void* a, const void* b)
{
    float x = ((const bprim*)a)->c[g_axis], y = ((const bprim*)b)->c[g_axis];
    if (x < y) return -1;
    if (x > y) return 1;
    return ((const bprim*)a)->id - ((const bprim*)b)->id;
}

static int bvh_build_rec(oscene* s, bprim* p, int first, int count)
{
    int ni = s->n_nodes++;
    bnode* n = &s->nodes[ni];
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < 3; ++k) { n->lo[k] = INFINITY; n->hi[k] = -INFINITY; }
    for (int i = first; i < first + count; ++i)
        for (int k = 0; k < 3; ++k) {
            n->lo[k] = fminf(n->lo[k], p[i].lo[k]); n->hi[k] = fmaxf(n->hi[k], p[i].hi[k]);
            clo[k] = fminf(clo[k], p[i].c[k]); chi[k] = fmaxf(chi[k], p[i].c[k]);
        }
    n->first = first; n->count = count; n->left = n->right = -1;
    if (count <= BVH_LEAF) return ni;
    int ax = 0;
    if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
    if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
    g_axis = ax;
    qsort(p + first, count, sizeof(bprim), cmp_bprim);
    int half = count / 2;
    int l = bvh_build_rec(s, p, first, half);
    int r = bvh_build_rec(s, p, first + half, count - half);
    s->nodes[ni].left = l; s->nodes[ni].right = r; s->nodes[ni].count = 0;
    return ni;
}

static void scene_init(oscene* s, const vrt_scene_desc* d)
{
    memset(s, 0, sizeof *s);
    s->d = d;
    s->n_prims = d->n_spheres + d->n_boxes + d->n_triangles;
    s->use_bvh = s->n_prims > BVH_THRESHOLD;
    if (!s->use_bvh) return;
    bprim* p = (bprim*)malloc(sizeof(bprim) * s->n_prims);
    for (uint32_t i = 0; i < s->n_prims; ++i) {
        prim_bounds(d, (int)i, p[i].lo, p[i].hi);
        for (int k = 0; k < 3; ++k) p[i].c[k] = 0.5f * (p[i].lo[k] + p[i].hi[k]);
        p[i].id = (int)i;
    }
    s->nodes = (bnode*)malloc(sizeof(bnode) * 2 * (size_t)s->n_prims);
    s->n_nodes = 0;
    bvh_build_rec(s, p, 0, (int)s->n_prims);
    s->order = (int*)malloc(sizeof(int) * s->n_prims);
    for (uint32_t i = 0; i < s->n_prims; ++i) s->order[i] = p[i].id;
    free(p);
}

static void scene_free(oscene* s) { free(s->nodes); free(s->order); }

static int slab(const bnode* n, oray r, const float inv[3], float tmax)
{
    const float o[3] = {r.o.x, r.o.y, r.o.z};
    float t0 = 0.f, t1 = tmax;
    for (int k = 0; k < 3; ++k) {
        float a = (n->lo[k] - o[k]) * inv[k], b = (n->hi[k] - o[k]) * inv[k];
        float mn = fminf(a, b), mx = fmaxf(a, b);
        t0 = fmaxf(t0, mn);           /* fminf/fmaxf drop NaN (0*inf) */
        t1 = fminf(t1, mx * 1.0000004f);
    }
    return t0 <= t1;
}

/* Closest hit (any = 0) or "is there any hit" (any = 1) within maxd, ignoring primitive
 * `skip`, the primitive a secondary ray was spawned on.  The reference starts those rays exactly
 * on the surface, without an epsilon: an outward ray cannot re-hit a convex collider, and for an
 * inward one Unity does not report a collider that contains the origin, so the source is never
 * the answer; naming it here replaces PhysX's behaviour at t ~ 0, which is rounding noise. */
static ohit intersect(const oscene* s, oray r, float maxd, int skip, int any)
{
    ohit best; memset(&best, 0, sizeof best); best.prim = -1;
    if (r.d.x == 0.f && r.d.y == 0.f && r.d.z == 0.f) return best; /* Unity: zero direction → false */
    if (!s->use_bvh) {
        for (uint32_t i = 0; i < s->n_prims; ++i) {
            if ((int)i == skip) continue;
            test_prim(s, (int)i, r, maxd, &best);
            if (any && best.hit) return best;
        }
        return best;
    }
    float inv[3] = {1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z};
    int stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const bnode* n = &s->nodes[stack[--sp]];
        float lim = best.hit ? best.t : maxd;
        if (!slab(n, r, inv, lim)) continue;
        if (n->left < 0) {
            for (int i = n->first; i < n->first + n->count; ++i) {
                if (s->order[i] == skip) continue;
                test_prim(s, s->order[i], r, maxd, &best);
                if (any && best.hit) return best;
            }
        } else { stack[sp++] = n->left; stack[sp++] = n->right; }
    }
    return best;
}

/* ------------------------------------------------------------------ tracer state */

typedef struct {
    oscene* scene;
    const vrt_params* prm;
    uint64_t rays[VRT_RAY_CLASSES];
    /* optional path recording */
    vrt_path_seg* segs; int seg_cap; int seg_n; int cur_sample;
} octx;

static void record(octx* c, oray r, ohit h, int kind, int depth, float range)
{
    if (!c->segs) return;
    if (c->seg_n < c->seg_cap) {
        vrt_path_seg* s = &c->segs[c->seg_n];
        v3 e = h.hit ? h.point : vadd(r.o, vmulf(r.d, range));
        s->origin[0] = r.o.x; s->origin[1] = r.o.y; s->origin[2] = r.o.z;
        s->end[0] = e.x; s->end[1] = e.y; s->end[2] = e.z;
        s->kind = kind; s->depth = depth; s->hit_prim = h.hit ? h.prim : -1; s->sample = c->cur_sample;
    }
    c->seg_n++;
}

static col trace_ray(octx* c, oray ray, int depth, int skip, int kind);

/* Util/RayTraceUtility.cs:123-131 */
static col sky_color(v3 direction)
{
    v3 u = vnormalize(direction);
    float t = 0.5f * (u.y + 1.f);
    v3 a = vmulf(V(1.f, 1.f, 1.f), 1.f - t);
    v3 b = vmulf(V(.5f, .7f, 1.f), t);
    v3 s = vadd(a, b);
    return C(s.x, s.y, s.z, 1.f);
}

/* Light/AmbientOccluder.cs:56-85 with the hemisphere sample of
 * Sampler/AbstractSampler.cs:123-140 taken from the all-zero multi-jittered table
 * (Sampler/MultiJitteredSampler.cs:23-27,39: every Set() hits a struct copy). */
static col ambient_occluder_L(octx* c, ohit hit, int depth)
{
    const vrt_shading* sh = &c->scene->d->shading;
    v3 w = hit.normal, u, v, dir;
    if (c->prm->flags & VRT_FIX_AO_BASIS) {
        dir = w;
    } else {
        v3 wCrossUp = vcross(w, V(0.f, 1.f, 0.f));
        v = vdivf(wCrossUp, vmag(wCrossUp));
        u = vcross(v, w);
        v.x = mpow(w.x, 0.0072f);
        v.y = mpow(w.y, 1.f);
        v.z = mpow(w.z, 0.0034f);
        v = vnormalize(v);
        u.x = mpow(v.x, w.x);
        u.y = mpow(v.y, w.y);
        u.z = mpow(v.z, w.z);
        /* hemisphere sample for unit-square sample (0,0), e = 1 */
        float sx = 0.f, sy = 0.f, e = 1.f;
        float cos_phi = (float)cos((double)(2.f * 3.14159274f * sx));
        float sin_phi = (float)sin((double)(2.f * 3.14159274f * sx));
        float cos_theta = mpow(1.f - sy, 1.f / (e + 1.f));
        float sin_theta = sqrtf(1.f - cos_theta * cos_theta);
        v3 sp = V(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
        dir = vadd(vadd(vmulf(u, sp.x), vmulf(v, sp.y)), vmulf(w, sp.z));
    }
    oray shadow = make_ray(hit.point, dir);
    int occluded = 0;
    if (!(shadow.d.x == 0.f && shadow.d.y == 0.f && shadow.d.z == 0.f)) {
        c->rays[VRT_RAY_AO]++;
        ohit h = intersect(c->scene, shadow, c->prm->range, hit.prim, c->segs ? 0 : 1);
        record(c, shadow, h, VRT_RAY_AO, depth, c->prm->range);
        occluded = h.hit;
    }
    col lc = C(sh->ao_color[0], sh->ao_color[1], sh->ao_color[2], sh->ao_color[3]);
    if (occluded) {
        col mn = C(sh->ao_min_amount[0], sh->ao_min_amount[1], sh->ao_min_amount[2], sh->ao_min_amount[3]);
        return cmul(cmulf(mn, sh->ao_ls), lc);
    }
    return cmulf(lc, sh->ao_ls);
}

/* Material/PhongMaterial.cs:53-89 */
static col phong_shade(octx* c, ohit hit, v3 raydir, int depth, float ka, float kd, float ks,
                       float ex, col cd)
{
    const vrt_scene_desc* d = c->scene->d;
    v3 wo = vneg(raydir);
    col L = cmul(cmulf(cd, ka), ambient_occluder_L(c, hit, depth)); /* LambertianBRDF.Rho :54-57 */
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        const vrt_light* l = &d->lights[i];
        v3 wi; float dist_to_light;
        v3 lp = V(l->vec[0], l->vec[1], l->vec[2]);
        if (l->type == VRT_LIGHT_POINT) { /* Light/PointLight.cs:23-34 */
            v3 tmp = vsub(lp, hit.point);
            float len = sqrtf(tmp.x * tmp.x + tmp.y * tmp.y + tmp.z * tmp.z);
            tmp.x /= len; tmp.y /= len; tmp.z /= len;
            wi = tmp;
        } else { /* Light/DirectionalLight.cs:22-25 */
            wi = lp;
        }
        float ndotwi = vdot(hit.normal, wi);
        if (ndotwi > 0.f) {
            int in_shadow = 0;
            if (l->cast_shadows) { /* PointLight.cs:36-46 / DirectionalLight.cs:27-37 */
                oray sr = make_ray(hit.point, wi);
                v3 loc = l->type == VRT_LIGHT_POINT ? lp : V(l->location[0], l->location[1], l->location[2]);
                dist_to_light = vmag(vsub(loc, sr.o));
                if (!(sr.d.x == 0.f && sr.d.y == 0.f && sr.d.z == 0.f)) {
                    c->rays[VRT_RAY_SHADOW]++;
                    ohit h = intersect(c->scene, sr, 30.f, hit.prim, 0);
                    record(c, sr, h, VRT_RAY_SHADOW, depth, 30.f);
                    if (h.hit) in_shadow = h.t < dist_to_light;
                }
            }
            if (!in_shadow) {
                col diff = cmulf(cmulf(cd, kd), 0.3183098861837906715f); /* LambertianBRDF.F :23-26 */
                /* GlossySpecularBRDF.F :40-63, called with (wo, wi) into parameters named (wi, wo) */
                col spec = C(0.f, 0.f, 0.f, 0.f);
                float nda = vdot(hit.normal, wo);
                v3 r;
                r.x = -wo.x + 2.f * hit.normal.x * nda;
                r.y = -wo.y + 2.f * hit.normal.y * nda;
                r.z = -wo.z + 2.f * hit.normal.z * nda;
                float rdb = vdot(r, wi);
                if (rdb > 0.0) spec = cmulf(cmulf(C(1.f, 1.f, 1.f, 1.f), ks), mpow(rdb, ex));
                col lam = cadd(diff, spec);
                col lc = C(l->color[0], l->color[1], l->color[2], l->color[3]);
                col world = cmulf(cmulf(lc, l->ls), ndotwi);
                L = cadd(L, cmul(lam, world));
            }
        }
    }
    return L;
}

/* BRDF/FresnelReflectorBRDF.cs:14-39 (only reached with VRT_FIX_DIELECTRIC) */
static float fresnel_kr(v3 normal, v3 direction, float eta_in, float eta_out)
{
    float ndotd = vdot(vneg(normal), direction);
    float eta;
    if (ndotd < 0.f) { normal = vneg(normal); eta = eta_out / eta_in; }
    else eta = eta_in / eta_out;
    float ci = vdot(vneg(normal), direction);
    float ct = sqrtf(1.f - (1.f - ci * ci) / (eta * eta));
    float rpar = (eta * ci - ct) / (eta * ci + ct);
    float rper = (ci - eta * ct) / (ci + eta * ct);
    return 0.5f * (rpar * rpar + rper * rper);
}

/* Texture2D.GetPixel with out-of-range coordinates handled by the wrap mode. */
static int wrap_pixel(int p, int n, int wrap)
{
    if (wrap == 1) return p < 0 ? 0 : (p >= n ? n - 1 : p);
    p %= n;
    return p < 0 ? p + n : p;
}

/* Tracer/WhittedTracer.cs:55-63: the texel under hit.textureCoord.  textureCoord is (0,0) on
 * sphere and box colliders; on a mesh it is the barycentric mix of the vertex coordinates, the
 * barycentrics being those of the triangle test (u/det, v/det). */
static col texel_at_hit(const vrt_scene_desc* d, const vrt_texture* tx, ohit hit, oray ray)
{
    float tu = 0.f, tv = 0.f;
    uint32_t first_tri = d->n_spheres + d->n_boxes;
    if ((uint32_t)hit.prim >= first_tri && d->uvs) {
        uint32_t ti = (uint32_t)hit.prim - first_tri;
        v3 v0, v1, v2;
        tri_verts(d, ti, &v0, &v1, &v2);
        v3 e1 = vsub(v1, v0), e2 = vsub(v2, v0);
        v3 pvec = vcross(ray.d, e2);
        float det = vdot(e1, pvec);
        v3 tvec = vsub(ray.o, v0);
        float u = vdot(tvec, pvec);
        v3 qvec = vcross(tvec, e1);
        float v = vdot(ray.d, qvec);
        float bu = u / det, bv = v / det;
        float bw = 1.f - bu - bv;
        const uint32_t* ix = d->indices + 3 * (size_t)ti;
        const float* a = d->uvs + 2 * (size_t)ix[0];
        const float* b = d->uvs + 2 * (size_t)ix[1];
        const float* cc = d->uvs + 2 * (size_t)ix[2];
        tu = a[0] * bw + b[0] * bu + cc[0] * bv;
        tv = a[1] * bw + b[1] * bu + cc[1] * bv;
    }
    float px = tu * (float)tx->width, py = tv * (float)tx->height; /* :59-61 */
    int ix_ = wrap_pixel((int)floorf(px), tx->width, tx->wrap);
    int iy_ = wrap_pixel((int)floorf(py), tx->height, tx->wrap);
    const float* t = tx->rgba + 4 * ((size_t)iy_ * tx->width + ix_);
    return C(t[0], t[1], t[2], t[3]);
}

static col shade(octx* c, ohit hit, oray ray, int depth)
{
    const vrt_scene_desc* d = c->scene->d;
    const vrt_shading* sh = &d->shading;
    int mi;
    if ((uint32_t)hit.prim < d->n_spheres) mi = d->spheres[hit.prim].material;
    else if ((uint32_t)hit.prim - d->n_spheres < d->n_boxes) mi = d->boxes[hit.prim - d->n_spheres].material;
    else mi = d->tri_material ? d->tri_material[hit.prim - d->n_spheres - d->n_boxes] : 0;
    const vrt_material* m = &d->materials[mi];
    if (d->material_texture && d->n_textures && d->material_texture[mi] >= 0 &&
        (uint32_t)d->material_texture[mi] < d->n_textures)
        return texel_at_hit(d, &d->textures[d->material_texture[mi]], hit, ray);
    col cd = C(m->color[0], m->color[1], m->color[2], m->color[3]);
    v3 n = hit.normal;

    if (m->type == VRT_MAT_METAL) { /* Material/ReflectiveMaterial.cs:25-42 */
        col L = phong_shade(c, hit, ray.d, depth, sh->metal_ka, sh->metal_kd, sh->metal_ks, sh->metal_exp, cd);
        v3 wo = vneg(ray.d);
        float ndotwo = vdot(n, wo); /* BRDF/PerfectSpecularBRDF.cs:29-39 */
        v3 wi = vadd(vneg(wo), vmulf(vmulf(n, 2.f), ndotwo));
        col cr = C(sh->metal_cr[0], sh->metal_cr[1], sh->metal_cr[2], sh->metal_cr[3]);
        col fr = cdivf(cmulf(cr, sh->metal_kr), fabsf(vdot(n, wi)));
        oray rr = make_ray(hit.point, wi);
        col child = trace_ray(c, rr, depth + 1, hit.prim, VRT_RAY_REFLECT);
        return cadd(L, cmulf(cmul(fr, child), vdot(n, wi)));
    }
    if (m->type == VRT_MAT_DIELECTRIC) { /* Material/DielectricMaterial.cs:56-146 */
        col L = phong_shade(c, hit, ray.d, depth, sh->glass_ka, sh->glass_kd, sh->glass_ks, sh->glass_exp, cd);
        v3 wo = vneg(ray.d);
        float ndotwo = vdot(n, wo); /* BRDF/FresnelReflectorBRDF.cs:51-61 */
        v3 wi = vadd(vneg(wo), vmulf(vmulf(n, 2.f), ndotwo));
        oray rr = make_ray(hit.point, wi);
        float ndotwi = vdot(n, wi);
        if (!(c->prm->flags & VRT_FIX_DIELECTRIC)) {
            /* BTDF _ior = 0 and _rayDir = 0 (never assigned, DielectricMaterial.cs:23,
             * FresnelTransmitterBTDF.cs:17-20): Tir = 1 - (1 - 0)/(0*0) = -inf < 0, always true.
             * Filter colours are white and t = 0: Pow(Cf, 0) = 1. */
            float cos_theta_i = vdot(n, V(-0.f, -0.f, -0.f));
            float eta = 0.f;
            if (cos_theta_i < 0.f) eta = 1.f / eta;
            int tir = (1.f - (1.f - cos_theta_i * cos_theta_i) / (eta * eta)) < 0.f;
            if (tir) {
                col Lr = trace_ray(c, rr, depth + 1, hit.prim, VRT_RAY_REFLECT);
                float f = mpow(1.f, 0.f);
                L.r += f * Lr.r; L.g += f * Lr.g; L.b += f * Lr.b;
                return L;
            }
            return L; /* unreachable */
        }
        /* Textbook wiring: kr from Fresnel(), kt = 1 - kr, ior = EtaIn/EtaOut. */
        float eta = sh->glass_eta_in / sh->glass_eta_out;
        float ci = vdot(n, wo);
        v3 nn = n;
        if (ci < 0.f) { ci = -ci; nn = vneg(n); eta = 1.f / eta; }
        float temp = 1.f - (1.f - ci * ci) / (eta * eta);
        if (temp < 0.f) { /* total internal reflection */
            col Lr = trace_ray(c, rr, depth + 1, hit.prim, VRT_RAY_REFLECT);
            L.r += Lr.r; L.g += Lr.g; L.b += Lr.b;
            return L;
        }
        float kr = fresnel_kr(n, ray.d, sh->glass_eta_in, sh->glass_eta_out);
        float kt = 1.f - kr;
        float c2 = sqrtf(temp); /* BTDF/FresnelTransmitterBTDF.cs:32-50 */
        v3 wt = vsub(vdivf(vneg(wo), eta), vmulf(nn, c2 - ci / eta));
        float ndotwt = vdot(n, wt);
        float frs = kr / fabsf(ndotwi);
        float fts = kt / (eta * eta) / fabsf(ndotwt);
        oray tr = make_ray(hit.point, wt);
        col Lr = trace_ray(c, rr, depth + 1, hit.prim, VRT_RAY_REFLECT);
        float wr = frs * fabsf(ndotwi);
        L.r += wr * Lr.r; L.g += wr * Lr.g; L.b += wr * Lr.b;
        col Lt = trace_ray(c, tr, depth + 1, hit.prim, VRT_RAY_TRANSMIT);
        float wt_ = fts * fabsf(ndotwt);
        L.r += wt_ * Lt.r; L.g += wt_ * Lt.g; L.b += wt_ * Lt.b;
        return L;
    }
    if (m->type == VRT_MAT_MATTE) { /* Material/MatteMaterial.cs:44-75: ambient + Lambert, no shadow test */
        col L = cmul(cmulf(cd, sh->matte_ka), ambient_occluder_L(c, hit, depth)); /* :49-53 */
        for (uint32_t i = 0; i < d->n_lights; ++i) {
            const vrt_light* l = &d->lights[i];
            v3 lp = V(l->vec[0], l->vec[1], l->vec[2]);
            v3 wi;
            if (l->type == VRT_LIGHT_POINT) { /* Light/PointLight.cs:23-34 */
                v3 tmp = vsub(lp, hit.point);
                float len = sqrtf(tmp.x * tmp.x + tmp.y * tmp.y + tmp.z * tmp.z);
                tmp.x /= len; tmp.y /= len; tmp.z /= len;
                wi = tmp;
            } else {
                wi = lp;
            }
            float ndotwi = vdot(n, wi);
            if (ndotwi > 0.f) { /* :69: F * lights[i].L(hit) * ndotwi, left to right */
                col f = cmulf(cmulf(cd, sh->matte_kd), 0.3183098861837906715f);
                col lc = C(l->color[0], l->color[1], l->color[2], l->color[3]);
                L = cadd(L, cmulf(cmul(f, cmulf(lc, l->ls)), ndotwi));
            }
        }
        return L;
    }
    if (m->type == VRT_MAT_TRANSPARENT) { /* Material/TransparentMaterial.cs:40-69 */
        col L = phong_shade(c, hit, ray.d, depth, sh->transp_ka, sh->transp_kd, sh->transp_ks, sh->transp_exp, cd);
        v3 wo = vneg(ray.d);
        float ndotwo = vdot(n, wo); /* BRDF/PerfectSpecularBRDF.cs:29-39 */
        v3 wi = vadd(vneg(wo), vmulf(vmulf(n, 2.f), ndotwo));
        float fr = sh->transp_kr / fabsf(vdot(n, wi));
        oray rr = make_ray(hit.point, wi);
        /* BTDF/PerfectTransmitterBTDF.cs:57-69: Tir() looks at the direction the BTDF was constructed
         * with, not at the ray being shaded */
        v3 bwo = vneg(V(sh->transp_ray_dir[0], sh->transp_ray_dir[1], sh->transp_ray_dir[2]));
        float bci = vdot(n, bwo);
        float beta = sh->transp_ior;
        if (bci < 0.f) beta = (float)(1.f / beta);
        if ((1.f - (1.f - bci * bci) / (beta * beta)) < 0.f) {
            col Lr = trace_ray(c, rr, depth + 1, hit.prim, VRT_RAY_REFLECT);
            L.r += Lr.r; L.g += Lr.g; L.b += Lr.b;
            return L;
        }
        /* PerfectTransmitterBTDF.SampleF :32-49 */
        v3 nn = n;
        float ci = vdot(nn, wo);
        float eta = sh->transp_ior;
        if (ci < 0.f) { ci = -ci; nn = vneg(nn); eta = (float)1.f / eta; }
        float temp = 1.f - (1.f - ci * ci) / (eta * eta);
        float c2 = sqrtf(temp);
        v3 wt = vsub(vdivf(vneg(wo), eta), vmulf(nn, c2 - ci / eta));
        float ft = sh->transp_kt / (eta * eta) / fabsf(vdot(n, wt));
        oray tr = make_ray(hit.point, wt);
        col Lr = trace_ray(c, rr, depth + 1, hit.prim, VRT_RAY_REFLECT);
        float ar = fabsf(vdot(n, wi));
        L.r += fr * Lr.r * ar; L.g += fr * Lr.g * ar; L.b += fr * Lr.b * ar;
        col Lt = trace_ray(c, tr, depth + 1, hit.prim, VRT_RAY_TRANSMIT);
        float at = fabsf(vdot(n, wt));
        L.r += ft * Lt.r * at; L.g += ft * Lt.g * at; L.b += ft * Lt.b * at;
        return L;
    }
    return phong_shade(c, hit, ray.d, depth, sh->solid_ka, sh->solid_kd, sh->solid_ks, sh->solid_exp, cd);
}

/* Tracer/WhittedTracer.cs:33-104 */
static col trace_ray(octx* c, oray ray, int depth, int skip, int kind)
{
    if (depth > c->prm->max_depth) return C(0.f, 0.f, 0.f, 1.f);
    ohit hit; memset(&hit, 0, sizeof hit); hit.prim = -1;
    if (!(ray.d.x == 0.f && ray.d.y == 0.f && ray.d.z == 0.f)) {
        c->rays[kind]++;
        hit = intersect(c->scene, ray, c->prm->range, skip, 0);
        record(c, ray, hit, kind, depth, c->prm->range);
    }
    if (!hit.hit) return sky_color(ray.d);
    return shade(c, hit, ray, depth);
}

/* ------------------------------------------------------------------ pixel loop */

/* Stand-in for the irreproducible shared sampler state: which sample set a pixel uses. */
static uint32_t pixel_set(uint32_t seed, uint32_t pixel, uint32_t n_sets)
{
    if (n_sets <= 1) return 0;
    uint32_t x = pixel ^ seed;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x % n_sets;
}
uint32_t vrto_pixel_set(uint32_t seed, uint32_t pixel, uint32_t n_sets) { return pixel_set(seed, pixel, n_sets); }

/* RayTracerUnity.cs:486-517 */
static v3 camera_dir(const vrt_camera* cam, int h, int v)
{
    v3 O = V(cam->origin[0], cam->origin[1], cam->origin[2]);
    v3 P0 = V(cam->p0[0], cam->p0[1], cam->p0[2]);
    v3 X = V(cam->x_axis[0], cam->x_axis[1], cam->x_axis[2]);
    v3 Y = V(cam->y_axis[0], cam->y_axis[1], cam->y_axis[2]);
    float hScale = (float)v / (float)cam->width;
    float vScale = (float)h / (float)cam->height;
    v3 ho = vmulf(X, hScale), vo = vmulf(Y, vScale);
    return vadd(vadd(vsub(P0, O), ho), vo);
}

/* Util/RayTraceUtility.cs:409-422 via RayTracerUnity.cs:525-562 (gamma is computed and dropped) */
static col process_color(v3 rgb)
{
    col c = C(rgb.x, rgb.y, rgb.z, 1.f);
    float m = fmaxf(c.r, fmaxf(c.g, c.b));
    if (m > 1.f) return cdivf(c, m);
    return c;
}

/* RayTracerUnity.cs:565-597.  sample < 0: all samples. */
static col trace_pixel(octx* c, const vrt_camera* cam, int h, int v, int sample, int32_t* hit_id)
{
    const vrt_params* p = c->prm;
    v3 O = V(cam->origin[0], cam->origin[1], cam->origin[2]);
    v3 raydir = camera_dir(cam, h, v);
    int spp = p->spp > 0 ? p->spp : 1;
    uint32_t pixel = (uint32_t)(h * cam->width + v);
    uint32_t set = pixel_set(p->seed, pixel, (uint32_t)(p->n_sets > 0 ? p->n_sets : 1));
    v3 sum = V(0.f, 0.f, 0.f);
    if (hit_id) *hit_id = -1;
    int i0 = sample < 0 ? 0 : sample, i1 = sample < 0 ? spp : sample + 1;
    for (int i = i0; i < i1; ++i) {
        float ox = 0.f, oy = 0.f;
        if (p->offsets) { ox = p->offsets[2 * ((size_t)set * spp + i)]; oy = p->offsets[2 * ((size_t)set * spp + i) + 1]; }
        v3 d = V(raydir.x + ox, raydir.y + oy, raydir.z); /* AntiAliasingStrategy.cs:89-90 */
        oray ray = make_ray(O, d);
        c->cur_sample = i;
        /* pre-check, RayTracerUnity.cs:582 */
        int pre = 0;
        if (!(ray.d.x == 0.f && ray.d.y == 0.f && ray.d.z == 0.f)) {
            c->rays[VRT_RAY_PRECHECK]++;
            pre = intersect(c->scene, ray, p->range, -1, 1).hit;
        }
        if (pre) {
            /* find the sample-0 primary id without disturbing the counters */
            if (i == 0 && hit_id) *hit_id = intersect(c->scene, ray, p->range, -1, 0).prim;
            col t = trace_ray(c, ray, 0, -1, VRT_RAY_PRIMARY);
            sum = vadd(sum, V(t.r, t.g, t.b));
        } else if (p->flags & VRT_FIX_PRIMARY_SKY) {
            col t = sky_color(ray.d);
            sum = vadd(sum, V(t.r, t.g, t.b));
        }
    }
    v3 fin = vdivf(sum, (float)(i1 - i0 == spp ? spp : 1));
    return process_color(fin);
}

/* ------------------------------------------------------------------ exported API */

int vrto_render(const vrt_scene_desc* scene, const vrt_camera* cam, const vrt_params* prm,
                float* rgba, int32_t* hit_id, uint64_t* rays_out, int n_threads)
{
    oscene s; scene_init(&s, scene);
    uint64_t tot[VRT_RAY_CLASSES] = {0};
    int W = cam->width, H = cam->height;
    int tcount = prm->tile_count > 1 ? prm->tile_count : 1;
    int trows = prm->tile_rows > 0 ? prm->tile_rows : 8;
    (void)n_threads;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel
#endif
    {
        octx c; memset(&c, 0, sizeof c); c.scene = &s; c.prm = prm;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int h = 0; h < H; ++h) {
            if (tcount > 1 && (h / trows) % tcount != prm->tile_index) continue;
            for (int v = 0; v < W; ++v) {
                int32_t id;
                col px = trace_pixel(&c, cam, h, v, -1, &id);
                size_t idx = (size_t)h * W + v; /* RayTracerUnity.cs:442 */
                rgba[4 * idx] = px.r; rgba[4 * idx + 1] = px.g; rgba[4 * idx + 2] = px.b; rgba[4 * idx + 3] = px.a;
                if (hit_id) hit_id[idx] = id;
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        for (int k = 0; k < VRT_RAY_CLASSES; ++k) tot[k] += c.rays[k];
    }
    if (rays_out) memcpy(rays_out, tot, sizeof tot);
    scene_free(&s);
    return 0;
}

int vrto_trace_pixel(const vrt_scene_desc* scene, const vrt_camera* cam, const vrt_params* prm,
                     int h, int v, int sample, vrt_path_seg* segs, int cap, int* n_out,
                     float rgba[4], int32_t* hit_id, uint64_t* rays_out)
{
    oscene s; scene_init(&s, scene);
    octx c; memset(&c, 0, sizeof c); c.scene = &s; c.prm = prm;
    vrt_path_seg dummy;
    c.segs = segs ? segs : (cap == 0 ? NULL : &dummy); c.seg_cap = segs ? cap : 0;
    col px = trace_pixel(&c, cam, h, v, sample, hit_id);
    rgba[0] = px.r; rgba[1] = px.g; rgba[2] = px.b; rgba[3] = px.a;
    if (n_out) *n_out = c.seg_n;
    if (rays_out) memcpy(rays_out, c.rays, sizeof c.rays);
    scene_free(&s);
    return 0;
}

/* AbstractTracer.TraceRay(ray, depth) for one ray (Tracer/WhittedTracer.cs:33-104). */
int vrto_trace_ray(const vrt_scene_desc* scene, const vrt_params* prm, const float o[3], const float d[3],
                   int depth, float rgba[4], int32_t* hit_id, uint64_t* rays_out)
{
    oscene s; scene_init(&s, scene);
    octx c; memset(&c, 0, sizeof c); c.scene = &s; c.prm = prm;
    oray r = make_ray(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]));
    if (hit_id) {
        *hit_id = -1;
        if (depth <= prm->max_depth) *hit_id = intersect(&s, r, prm->range, -1, 0).prim;
    }
    col px = trace_ray(&c, r, depth, -1, depth == 0 ? VRT_RAY_PRIMARY : VRT_RAY_REFLECT);
    rgba[0] = px.r; rgba[1] = px.g; rgba[2] = px.b; rgba[3] = px.a;
    if (rays_out) memcpy(rays_out, c.rays, sizeof c.rays);
    scene_free(&s);
    return 0;
}

/* Single queries, for unit tests of the pieces. */
int vrto_intersect(const vrt_scene_desc* scene, const float o[3], const float d[3], float maxd,
                   int skip, int any, int force_brute, float* t, int32_t* prim, float point[3], float normal[3])
{
    oscene s; scene_init(&s, scene);
    if (force_brute) s.use_bvh = 0;
    oray r = make_ray(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]));
    ohit h = intersect(&s, r, maxd, skip, any);
    *t = h.t; *prim = h.hit ? h.prim : -1;
    point[0] = h.point.x; point[1] = h.point.y; point[2] = h.point.z;
    normal[0] = h.normal.x; normal[1] = h.normal.y; normal[2] = h.normal.z;
    scene_free(&s);
    return h.hit;
}

void vrto_sky_color(const float d[3], float out[4])
{
    col c = sky_color(V(d[0], d[1], d[2]));
    out[0] = c.r; out[1] = c.g; out[2] = c.b; out[3] = c.a;
}

void vrto_process_color(const float rgb[3], float out[4])
{
    col c = process_color(V(rgb[0], rgb[1], rgb[2]));
    out[0] = c.r; out[1] = c.g; out[2] = c.b; out[3] = c.a;
}

void vrto_camera_dir(const vrt_camera* cam, int h, int v, float out[3])
{
    v3 d = camera_dir(cam, h, v);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}

/* Direction the ambient occluder would shoot for a given normal under the reference's
 * arithmetic; returns 0 when the direction degenerates (NaN/zero) and no ray is cast. */
int vrto_ao_direction(const float n[3], float out[3])
{
    v3 w = V(n[0], n[1], n[2]), u, v;
    v.x = mpow(w.x, 0.0072f); v.y = mpow(w.y, 1.f); v.z = mpow(w.z, 0.0034f);
    v = vnormalize(v);
    u.x = mpow(v.x, w.x); u.y = mpow(v.y, w.y); u.z = mpow(v.z, w.z);
    v3 dir = vadd(vadd(vmulf(u, 0.f), vmulf(v, 0.f)), vmulf(w, 1.f));
    v3 nd = vnormalize(dir);
    out[0] = nd.x; out[1] = nd.y; out[2] = nd.z;
    return !(nd.x == 0.f && nd.y == 0.f && nd.z == 0.f);
}

/* ------------------------------------------------------------------ sampler tables
 * Sampler/{Regular,Random,Jittered,NRooks,MultiJittered,Hammersley}Sampler.cs with UnityEngine.Random replaced by PCG32 (XSH-RR, 64-bit state).
 * Random.Range(int a, int b) -> a + next() % (b - a); Random.Range(float a, float b) ->
 * a + (b - a) * ((next() >> 8) * 2^-24). */

typedef struct { uint64_t state, inc; } pcg;
static uint32_t pcg_next(pcg* g)
{
    uint64_t old = g->state;
    g->state = old * 6364136223846793005ULL + g->inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
}
static void pcg_seed(pcg* g, uint64_t seed)
{
    g->state = 0; g->inc = (54u << 1) | 1u;
    pcg_next(g); g->state += seed; pcg_next(g);
}
static int rnd_int(pcg* g, int a, int b) { return b > a ? a + (int)(pcg_next(g) % (uint32_t)(b - a)) : a; }
static float rnd_float(pcg* g, float a, float b) { return a + (b - a) * ((float)(pcg_next(g) >> 8) * (1.f / 16777216.f)); }

/* Sampler/HammersleySampler.cs:22-35 */
static float hammersley_phi(float j, float vStep)
{
    float x = 0.f, f = vStep * 0.5f;
    while (j >= 0.5f) {
        x += f * (float)fmodf(j * vStep, 2.f);
        j *= 0.5f;
        f *= 0.5f;
    }
    return x;
}

int vrto_sampler_table(int sampler, int n, int sets, float hStep, float vStep, uint64_t seed, float* out)
{
    if (n <= 0 || sets <= 0) return -1;
    pcg g; pcg_seed(&g, seed);
    size_t total = (size_t)n * sets;
    float* sx = (float*)calloc(total, sizeof(float));
    float* sy = (float*)calloc(total, sizeof(float));
    size_t cnt = 0; /* samples actually generated (Regular/Jittered produce floor(sqrt n)^2 per set) */
    int rt = (int)sqrt((double)(float)n);
    /* base-class constructor runs first: shuffled indices (AbstractSampler.cs:37-52,264-277) */
    int* shuf = (int*)malloc(sizeof(int) * total);
    int* idx = (int*)malloc(sizeof(int) * n);
    for (int j = 0; j < n; ++j) idx[j] = j;
    for (int p = 0; p < sets; ++p) {
        int m = n;
        while (m > 1) { m--; int k = rnd_int(&g, 0, m + 1); int t = idx[k]; idx[k] = idx[m]; idx[m] = t; }
        for (int j = 0; j < n; ++j) shuf[(size_t)p * n + j] = idx[j];
    }
    switch (sampler) {
    case VRT_SAMPLER_REGULAR: /* RegularSampler.cs:12-30 */
        for (int j = 0; j < sets; ++j)
            for (int p = 0; p < rt; ++p)
                for (int q = 0; q < rt; ++q) { sx[cnt] = (q * hStep) / (float)rt; sy[cnt] = (p * vStep) / (float)rt; cnt++; }
        break;
    case VRT_SAMPLER_RANDOM: /* RandomSampler.cs:11-23 */
        for (int p = 0; p < sets; ++p)
            for (int i = 0; i < n; ++i) { sx[cnt] = rnd_float(&g, 0.f, hStep - 1e-5f); sy[cnt] = rnd_float(&g, 0.f, vStep - 1e-5f); cnt++; }
        break;
    case VRT_SAMPLER_JITTERED: /* JitteredSampler.cs:12-30 */
        for (int p = 0; p < sets; ++p)
            for (int j = 0; j < rt; ++j)
                for (int k = 0; k < rt; ++k) {
                    float hr = rnd_float(&g, 0.f, hStep - 1e-5f), vr = rnd_float(&g, 0.f, vStep - 1e-5f);
                    sx[cnt] = (k * hStep + hr) / (float)rt; sy[cnt] = (j * vStep + vr) / (float)rt; cnt++;
                }
        break;
    case VRT_SAMPLER_NROOKS: /* NRooksSampler.cs:11-31; the shuffles write to struct copies (:52,55,77,80)
                                and only consume random numbers */
        for (int p = 0; p < sets; ++p)
            for (int j = 0; j < n; ++j) {
                float hr = rnd_float(&g, 0.f, hStep - 1e-5f), vr = rnd_float(&g, 0.f, vStep - 1e-5f);
                sx[cnt] = (j * hStep + hr) / (float)n; sy[cnt] = (j * vStep + vr) / (float)n; cnt++;
            }
        break;
    case VRT_SAMPLER_MULTIJITTERED: /* MultiJitteredSampler.cs:14-83: list pre-filled with (0,0),
                                       every Set() mutates a copy */
        cnt = total;
        break;
    case VRT_SAMPLER_HAMMERSLEY: /* HammersleySampler.cs:12-20 */
        for (int p = 0; p < sets; ++p)
            for (int j = 0; j < n; ++j) { sx[cnt] = (float)j * hStep / (float)n; sy[cnt] = hammersley_phi(j * vStep, vStep); cnt++; }
        break;
    default:
        free(sx); free(sy); free(shuf); free(idx); return -1;
    }
    /* access order: samples[jump + shuffled[jump + i]] (AbstractSampler.cs:163); an index past
     * the generated count throws in C# and leaves the zero vector (:145,165). */
    for (int p = 0; p < sets; ++p)
        for (int i = 0; i < n; ++i) {
            size_t k = (size_t)p * n + shuf[(size_t)p * n + i];
            float x = 0.f, y = 0.f;
            if (k < cnt) { x = sx[k]; y = sy[k]; }
            out[2 * ((size_t)p * n + i)] = x; out[2 * ((size_t)p * n + i) + 1] = y;
        }
    free(sx); free(sy); free(shuf); free(idx);
    return 0;
}

/* A list of pixels of one frame (linear index h*width+v): scene set-up once, then the
 * per-pixel routine for each.  Used to spot-check full-size frames. */
int vrto_trace_pixels(const vrt_scene_desc* scene, const vrt_camera* cam, const vrt_params* prm,
                      int n, const int32_t* pixels, float* rgba, int32_t* hit_id, uint64_t* rays_out)
{
    oscene s; scene_init(&s, scene);
    uint64_t tot[VRT_RAY_CLASSES] = {0};
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        octx c; memset(&c, 0, sizeof c); c.scene = &s; c.prm = prm;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int i = 0; i < n; ++i) {
            int32_t id;
            col px = trace_pixel(&c, cam, pixels[i] / cam->width, pixels[i] % cam->width, -1, &id);
            rgba[4 * i] = px.r; rgba[4 * i + 1] = px.g; rgba[4 * i + 2] = px.b; rgba[4 * i + 3] = px.a;
            hit_id[i] = id;
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        for (int k = 0; k < VRT_RAY_CLASSES; ++k) tot[k] += c.rays[k];
    }
    if (rays_out) memcpy(rays_out, tot, sizeof tot);
    scene_free(&s);
    return 0;
}
