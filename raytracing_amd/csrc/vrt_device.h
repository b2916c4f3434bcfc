// Device-side data layout of libvrt (gfx950).  Shared by the kernels and the C-ABI host code.
//
// Layout in HBM:
//   sph   float4[n_sph]        (cx, cy, cz, r)
//   box   float4[4*n_box]      (c.xyz, h0) (A0.xyz, h1) (A1.xyz, h2) (A2.xyz, 0)
//   tri   float4[3*n_tri]      (v0.xyz, 0) (e1.xyz, 0) (e2.xyz, 0), e1 = v1-v0, e2 = v2-v0
//   prim_mat int[n_prims]      material index per primitive id (spheres, boxes, triangles)
//   mat   float4[n_mat]        (r, g, b, type-as-int-bits)
//   mat_tex, tex_info, tex_data, tri_uv   textured materials, see DevScene
//   bvh   float4[4*(n_prims-1)] binary tree (PLOC over Morton order), one 64-byte node per internal node holding the
//                              boxes of both children:
//                              (lo0.xyz, hi0.x) (hi0.yz, lo1.xy) (lo1.z, hi1.xyz) (c0, c1, parent, -)
//                              child >= 0: internal node index, child < 0: leaf, primitive = ~child;
//                              parent = index of the node above, -1 at the root
//   bvh4  float4[8*(n_prims-1)] the same tree collapsed to four children per node (greedily by
//                              box area, see k_collapse4_level): one 128-byte node per
//                              surviving internal node (index = its binary index)
//                              holding up to four child boxes, struct of arrays:
//                              lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] code[4] pad[4];
//                              code as above, VRT_BVH4_EMPTY for an unused slot
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vrt.h"

#define VRT_MAX_BIG 8

// Traversal-stack entries per lane (LDS, 8 bytes each): 16 entries x 64 lanes = 8 KiB per one-wave
// block, so 16 such blocks fit a CU beside their counters.  Measured on MI355X: 12 entries cost the
// 1M-triangle frame 12 % (stackless finishes), 24 entries cost both tree workloads 20 % (occupancy).
#ifndef VRT_TRAV_STACK
#define VRT_TRAV_STACK 16
#endif

struct DevLight {
    int type;
    float vx, vy, vz;
    float ls;
    float cr, cg, cb;
    int cast;
    float lx, ly, lz;
};

struct DevScene {
    int n_sph, n_box, n_tri, n_prims;
    const float4* sph;
    const float4* box;
    const float4* tri;
    const int* prim_mat;
    const float4* mat;
    const float4* bvh;
    const float4* bvh4;
    // brute-force scenes only:
    const float4* aab;     // 2 per box when all_aab: (centre.xyz, face/flip code) (world half extents, r^2)
    int all_aab;           // every box has signed world axes as its axes
    const float4* cam_pre; // per sphere (o-c, |o-c|^2 - r^2), per box (o-c, 0) for o = camera origin
    // textures (null mat_tex = the scene has none)
    const int* mat_tex;      // per material: texture index or -1
    const int4* tex_info;    // per texture: (first texel in tex_data, width, height, wrap)
    const float4* tex_data;  // texels of all textures, row 0 first
    const float2* tri_uv;    // 3 per triangle (vertex order of the index list) or null
    int n_lights;
    const DevLight* lights; // n_lights entries in device memory
    // Primitives whose box spans a large part of the scene (a ground box under 10k spheres) would
    // inflate every LBVH node above their leaf; they enter the tree as a point at their centre
    // and are tested ahead of the traversal instead.
    int n_big;
    int big[VRT_MAX_BIG];
    unsigned long long* dbg; // the context's counter block (slots 6, 7: nodes entered, primitives tested)
    int stack_cap;           // traversal-stack entries per lane in use: VRT_TRAV_STACK (vrt_trace.h) unless a test shortens it
    int has_matte, has_transparent; // some material is of class VRT_MAT_MATTE / VRT_MAT_TRANSPARENT
    // Products of shading constants, formed once on the host with the float operations the shading code
    // would use (one rounding per multiply): the ambient occluder's radiance unoccluded (color * ls) and
    // occluded ((min_amount * ls) * color), the mirror's cr * kr.  Wave-uniform products formed in the
    // kernel are hoisted to its start, kept in vector registers and spilled by every wave.
    float ao_free[3], ao_occl[3], metal_w[3];
    vrt_shading sh;
};

struct DevCam {
    float ox, oy, oz;
    float bx, by, bz; // P0 - O
    float xx, xy, xz;
    float yx, yy, yz;
    int width, height;
};

struct DevParams {
    int spp, n_sets;
    const float2* offsets; // n_sets*spp or null
    unsigned seed;
    int max_depth;
    float range;
    unsigned flags;
    // strip sharding: local row r -> global row
    int tile_index, tile_count, tile_rows;
    int local_rows; // rows this call renders
    int compact;    // outputs indexed by local row instead of global row
    int xcd_remap;  // renumber blocks so that each XCD works on one contiguous band of tiles
    int sample_shift; // megakernel: log2 of the lanes that share one pixel's samples (0..6)
    int packet;       // tree scenes: rays a wave casts together take the wave-uniform traversal first
    int general;      // the scene or the flags need the general kernels: two children per hit, matte / transparent materials
    int quad_max;     // tree scenes: a wave with at most this many queries traces them four lanes per ray (0 = never)
    int beam;         // tree scenes, megakernel: the camera rays of a wave collect their candidate leaves together first
    const int* tile_order;   // megakernel: block b renders tile tile_order[b] (null: tile b)
    unsigned* tile_cost;     // megakernel: per tile, the clock ticks its wave ran (null: not recorded)
};

// Per-call ray counters: 256 slots x 8 words to spread atomic traffic, summed by the host.
#define VRT_CTR_SLOTS 256
#define VRT_CTR_STRIDE 8

// Ray queue for the wavefront pipeline, struct of arrays, one entry per live ray.
struct RayQueue {
    float *ox, *oy, *oz, *dx, *dy, *dz; // origin, unit direction
    float *tr, *tg, *tb;                // path throughput
    float *lr, *lg, *lb;                // radiance the path has gathered so far
    int *pix;                           // sample slot = pixel of the band * spp + sample
    int *skip;                          // primitive the ray was spawned on
    int *meta;                          // depth | kind << 8
};
#define VRT_QUEUE_ARRAYS 15

// Occlusion queries (ambient-occlusion and shadow rays) the wavefront pipeline traces in a pass
// of their own; the answer of entry j is or-ed as bit `bit[j]` into the flag word of ray `target[j]`.
struct AnyQueue {
    float *ox, *oy, *oz, *dx, *dy, *dz; // origin, unit direction
    float *tlim, *maxd;                 // occluded iff some collider is hit at t < tlim within maxd
    int *skip, *target, *bit;
};
#define VRT_ANYQ_ARRAYS 11

// Levels a tree may have.  The traversal stacks live in LDS and are short; a ray that overflows
// its stack finishes on the stackless walk of the binary tree, so depth costs time, not correctness.
#define VRT_MAX_TREE_DEPTH 2048
#define VRT_BVH4_EMPTY ((int)0x80000000)
