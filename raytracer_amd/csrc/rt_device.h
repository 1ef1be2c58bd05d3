// rt_device.h - device-side scene layout shared by the host re-layout code
// (rt_host.cpp) and the gfx950 kernels (rt_kernels.hip).
//
// The reference shader reads 18 separate storage buffers (raytrace.comp:83-171).
// For the GPU they are re-laid out at upload time into five arrays:
//
//  DevObject  one per ParentBVHNode, in scene order (wave-uniform reads, so the
//             compiler fetches them with scalar loads): world box, the 3x4 part
//             of inverseTransform, the 3x4 part of its inverse (the shader's
//             per-ray `inverse(mat4)`, raytrace.comp:485, hoisted to upload
//             time with the same arithmetic), material, root reference.
//  DevNode    one 64-byte record per *inner* BVH node holding BOTH children's
//             boxes and references, so that one node visit is one aligned
//             64-byte fetch instead of two dependent 40-byte visits.
//             `first` is the child the reference pops first: the LEFT child of
//             an inner node (raytrace.comp:603-604) but the RIGHT child of a
//             ParentBVHNode (raytrace.comp:525-526).
//  DevTri     48 bytes per triangle in leaf order: v0, edge1 = v1-v0,
//             edge2 = v2-v0 (raytrace.comp:288-289 hoisted), plus the number of
//             triangles left in the leaf and the original triangle index.
//  DevNrm     48 bytes per original triangle: the three vertex normals, read
//             only for the one winning hit of a GetHit call.
//  materials  GPUMaterial verbatim.
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <stdint.h>

#define RT_REF_ABSENT ((int32_t)0x80000000)      /* child index was -1: not counted, not visited */
#define RT_REF_EMPTY  ((int32_t)0x80000001)      /* leaf with zero triangles: counted, nothing to test */
#define RT_REF_DEAD   ((int32_t)0x80000002)      /* child whose box has min > max on an axis: counted, no ray can enter it */
#define RT_REF_LIVE_MIN ((int32_t)0x80000002)    /* references above this value can be entered */
#define RT_REF_INSTANCE ((int32_t)0x40000000)    /* four-child records over the objects: RT_REF_INSTANCE + k = object k */
#define RT_MAX_STACK  32                         /* the shader's `int stack[32]` (raytrace.comp:523) */

struct DevObject {
    float   bmin[3];
    int32_t root;          /* >=0: DevNode index; <0: leaf, ~root = first DevTri; or RT_REF_EMPTY */
    float   bmax[3];
    int32_t materialIndex;
    float   inv[12];       /* inverseTransform, row r = inv[4r..4r+3] = (c0r,c1r,c2r,c3r) */
    float   fwd[12];       /* inverse(inverseTransform), same row layout */
};

static_assert(sizeof(DevObject) == 128, "DevObject: two 16-byte box words and two 3x4 matrices");

struct DevNode {
    float   lo[3][2];           /* lo[axis][c]: box minimum of the child visited first (c = 0) / second (c = 1) */
    float   hi[3][2];           /* box maximum; the two children sit side by side so that one packed f32
                                   instruction works on both boxes */
    int32_t first;              /* >=0 DevNode, <0 leaf (~first = DevTri index), RT_REF_ABSENT / RT_REF_EMPTY */
    int32_t second;
    int32_t pad[2];
};

/* Four-child record of the scenes whose hits do not depend on the visiting order (DevScene::nodes4):
 * a two-child tree of rt_bvh.cpp folded so that one visit tests four boxes and a root-to-leaf path
 * is half as many dependent fetches. Plane by plane, the four children side by side (packed f32
 * instructions work on two of them at a time). An unused slot holds a box far outside any scene
 * (all planes at RT_FAR_PLANE), which no ray interval can reach, so the walk needs no marker test. */
#define RT_FAR_PLANE 1e30f
struct DevNode4 {
    float   lo[3][4];           /* lo[axis][child] */
    float   hi[3][4];
    int32_t ref[4];             /* >= 0 DevNode4 index, < 0 leaf (~ref = first DevTri) */
    int32_t pad[4];
};

/* Scenes with at least this many BVH parents get a hierarchy over the parents' boxes (DevScene::tlasRoot); with
 * fewer, the parents are found by testing their boxes one after the other (measured: the default scene with 2 and
 * the height field with 1 are 5-8 % faster that way, the scene with 64 parents 18 % slower). */
#define RT_TLAS_MIN_PARENTS 8

struct DevTri {
    float   v0[3];  int32_t remaining;  /* triangles left in this leaf including this one */
    float   e1[3];  int32_t orig;       /* index into the caller's triangle arrays (and DevNrm) */
    float   e2[3];  int32_t pad;
};

static_assert(sizeof(DevNode4) == 128, "DevNode4: one 128-byte line per record");

#ifndef RT_WIDE_LDS_LEVELS
#define RT_WIDE_LDS_LEVELS 8      /* traversal-stack entries of the four-child walk kept in LDS (16, 12 and 8 measured: same time) */
#endif
#define RT_WIDE_DEEP_LEVELS 56    /* further entries, in the lane's private memory */

struct DevNrm {
    float n0[4], n1[4], n2[4];
};

struct DevMaterial {   /* GPUMaterial, RaytracerTypes.h:127-138 */
    float colour[3]; float smoothness;
    float fuzziness; float emission; float refractionIndex; float pad1;
    float absorbtion[3]; float pad3;
};

struct DevScene {
    const DevObject*   objects;
    const DevNode*     nodes;
    const DevTri*      tris;
    const DevNrm*      nrms;
    const DevMaterial* materials;
    /* Visiting order for rays that only ask "is anything hit" (shadow rays): leaf parents
     * first, BVH parents after. anyOrder[p] = object at position p; nextAny[j] = object that
     * follows object j (objectCount at the end); anyFirst = anyOrder[0]; anyLeafCount = number
     * of leaf parents = position of the first BVH parent in anyOrder. */
    const int32_t*     anyOrder;
    const int32_t*     nextAny;
    int32_t            anyFirst;
    int32_t            anyLeafCount;
    int32_t            objectCount;   /* number of DevObject records uploaded */
    int32_t            stackDepth;    /* entries of traversal stack a lane can need (<= RT_MAX_STACK) */
    /* Two-child records with an index >= nearFirstBase belong to trees rebuilt at upload (rt_bvh.cpp):
     * inside an object whose root is such a record the walk enters the nearer child first. INT32_MAX
     * in the reference-order scene. */
    int32_t            nearFirstBase;
    /* 1: the hit GetHit returns does not depend on the order in which the parents are visited (the
     * near-first scene when no object is scaled below 1), so parents may be tested concurrently and their
     * hits merged by distance. 0 in the reference-order scene. */
    int32_t            orderFree;
    /* Four-child records (null unless this is the wide twin of an order-free near-first scene: then the
     * object roots index this array and `nodes` is unused). wideLdsLevels = traversal-stack entries a lane
     * keeps in LDS; deeper ones go to the lane's private memory. */
    const DevNode4*    nodes4;
    int32_t            wideLdsLevels;   /* (unused) */
    int32_t            wideStackNeed;   /* entries a lane can need in the worst case */
    /* Runs of consecutive BVH parents: runLen[p] = number of BVH parents (root >= 0) that follow one another
     * from position p of the scene order on, at most 64 (0 where position p holds none); runLenAny the same
     * for positions of anyOrder. */
    const int32_t*     runLen;
    const int32_t*     runLenAny;
    /* Wide scenes only: the objects are stored leaf parents first, BVH parents after them. */
    /* Wide scenes: a hierarchy of four-child records over the world boxes of the BVH parents, stored behind the
     * objects' own records in nodes4 (record indices >= tlasBase; their boxes are in world space, their child
     * references records of the hierarchy or RT_REF_INSTANCE + object index). tlasRoot: its root reference
     * (an instance reference when the scene has one BVH parent); RT_REF_EMPTY: the scene has none. */
    int32_t            tlasRoot;
    int32_t            tlasBase;
};

#if defined(__HIPCC__)
#define RT_HOST_DEVICE __host__ __device__
#else
#define RT_HOST_DEVICE
#endif

/* Tiles of the partition are numbered along a serpentine path: row 0 left to right, row 1 right to
 * left, and so on, so that consecutive numbers are neighbouring tiles and dealing them out round-robin
 * gives every rank a sample of every part of the image. */
static inline RT_HOST_DEVICE void rt_tile_of_position(int p, int tilesX, int* tx, int* ty)
{
    const int row = p / tilesX, col = p - row * tilesX;
    *ty = row;
    *tx = (row & 1) ? tilesX - 1 - col : col;
}

struct DevLaunch {
    int32_t width, height;
    int32_t tileSize;      /* partition tile edge in pixels, multiple of 8 */
    int32_t tilesX, tilesY;
    int32_t tileRank, tileWorld;
    int32_t ownedTiles;    /* number of tiles t (serpentine numbers) with t % tileWorld == tileRank */
    /* round selection of the wave-scheduled kernel: the kind with the largest (waiting lanes -
     * bias) runs; a BVH round ends when fewer than max(quorumMin, started/quorumDiv) lanes walk */
    int32_t biasObject, biasTrav, biasShade, quorumMin, quorumDiv;
    /* lock-step kernel, path tracing: consecutive frames rendered by one launch (>= 1) and the step
     * of the frame index between them */
    int32_t frames, frameStride;
    /* Two-samples-per-lane kernel, launches in pieces (partial != null): the launch's frames are cut into pieces of
     * pieceFrames frames, block b renders piece b % pieces of the 8x8 pixels of wave b / pieces and stores each
     * frame's contribution (the value a dispatch adds to the accumulation pixel) at partial[frame in launch][pixel];
     * rt_sum_pieces then adds them to the accumulation image in frame order - the additions a launch in one piece
     * performs, in the same order - and writes the draw image. More and shorter waves: a launch lasts as long as its
     * slowest wave, which matters when a launch holds little more than the waves the chip runs at one time. */
    int32_t pieces, pieceFrames;
};

/* Buffers of the wavefront pipeline. A ray that leaves the path kernel becomes a job: a record
 * spread over RT_WF_JOB_ARRAYS arrays of 16-byte parts, indexed by its position in the round's job
 * list, so that consecutive lanes write and read consecutive addresses:
 *   0 ray origin, pixel slot (-1: unused entry)   1 ray direction, object to walk | start << 30 | shadow << 31
 *   2 closest, hitT, u, v                         3 best object + 1, best triangle, node tests, triangle tests
 *   4 throughput, sample                          5 result, bounce
 *   6 pending sun term   7 pending origin   8 pending direction           (shadow rays)
 *   9 pixel sum, GetHit calls                                             (raysPerPixel > 1, counting)
 * The walk kernel reads parts 0-3 and writes its answer back into parts 2 and 3 (part 3's first word
 * becomes (best object + 1) | next object << 16); the path kernel of the next round reads the list
 * in order. Job buffers alternate between rounds; a round's list holds at most one job per pixel
 * plus one partly used block per resident wave. */
#define RT_WF_JOB_ARRAYS 10
#define RT_WF_MAX_ROUNDS 30
#define RT_WF_WORDS_PER_ROUND 8   /* 0 jobs, 1 job head (walk kernel), 4 chunk head (path kernel of the next round) */
struct DevWavefront {
    void*     jobs[2];      /* RT_WF_JOB_ARRAYS * listCap float4 each; round r uses jobs[r & 1] */
    uint32_t* counts;       /* (RT_WF_MAX_ROUNDS + 2) * RT_WF_WORDS_PER_ROUND words, zeroed per dispatch */
    void*     deepStack;    /* per walking wave: the traversal stack levels that do not fit its LDS share */
    int32_t   listCap;
    int32_t   refillMin;    /* a walking wave fetches new jobs when this many lanes are free */
};

struct DevCounters {
    unsigned long long rays, nodeTests, triTests, samples;
};

#endif
