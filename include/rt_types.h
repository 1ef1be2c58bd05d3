/*
 * rt_types.h - byte-exact C mirrors of the data contract between the reference
 * renderer's host code and its ray-tracing compute shader.
 *
 * Every struct here has the same size and field offsets as the C++/glm struct
 * the reference uploads (src/Renderers/Hardware/RaytracerTypes.h:22-149) and as
 * the GLSL block that reads it (src/Shaders/Compute/raytrace.comp:53-203), so a
 * caller can pass its existing std::vector<...>::data() pointers unchanged.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* RaytracePushConstants: RaytracerTypes.h:22-52, raytrace.comp:173-203.
 * 144 bytes on the host, the shader reads the first 140. */
typedef struct rt_push_constants {
    float   pixel00Location[3];
    int32_t raysPerPixel;          /* default 1 */
    float   pixelDeltaU[3];
    int32_t maxBounces;            /* default 3 */
    float   pixelDeltaV[3];
    float   defocusAngle;
    float   cameraPosition[3];
    int32_t parentBVHCount;
    float   defocusDiskU[3];
    int32_t accumulateFrames;      /* default 1 */
    float   defocusDiskV[3];
    int32_t frame;                 /* default 0 */
    float   sunDirection[3];       /* default (-0.5,-1,-0.3), used un-normalised */
    float   sunIntensity;          /* default 5 */
    float   sunColour[3];          /* default (1,0.95,0.85) */
    int32_t renderMode;            /* 0 path trace, 1 boxes, 2 depth, 3 tri heat, 4 node heat */
    int32_t triangleTestThreshold; /* default 150 */
    int32_t bvhNodeTestThreshold;  /* default 120 */
    float   depthDebugScale;       /* default 50 */
    int32_t padding2;
} rt_push_constants;

/* AABB: RaytracerTypes.h:54-79 (vec3 + pad, vec3 + pad). */
typedef struct rt_aabb {
    float min[3];
    float padding2;
    float max[3];
    float padding;
} rt_aabb;

/* BVHNode: RaytracerTypes.h:81-89. */
typedef struct rt_bvh_node {
    rt_aabb aabb;
    int32_t leftChild;
    int32_t rightChild;
    int32_t triangleStartIndex;
    int32_t triangleCount;
} rt_bvh_node;

/* ParentBVHNode: RaytracerTypes.h:91-98; std140 stride 64 (raytrace.comp:68-72,83-86).
 * The box is in world space, child indices are global node indices. */
typedef struct rt_parent_bvh_node {
    rt_bvh_node node;
    int32_t objectIndex;
    int32_t padding1;
    int32_t padding2;
    int32_t padding3;
} rt_parent_bvh_node;

/* GPUObject: RaytracerTypes.h:117-125; std140 stride 80 (raytrace.comp:74-81,88-91).
 * inverseTransform is column-major (glm::mat4). */
typedef struct rt_gpu_object {
    int32_t triangleStartIndex;
    int32_t triangleCount;
    int32_t materialIndex;
    float   padding;
    float   inverseTransform[16];
} rt_gpu_object;

/* GPUMaterial: RaytracerTypes.h:127-138; std430 stride 48 (raytrace.comp:17-25,93-96). */
typedef struct rt_gpu_material {
    float albedo[3];
    float smoothness;
    float fuzziness;
    float emission;
    float refractiveIndex;
    float padding1;
    float absorbtion[3];
    float padding3;
} rt_gpu_material;

/* CameraSettings: HardwareRenderer.h:68-75. */
typedef struct rt_camera_settings {
    float cameraPosition[3];      /* default (0,1,5) */
    float cameraLookDirection[3]; /* default (0,0,-1) */
    float cameraFov;              /* default 90 */
    float focusDistance;          /* default 10 */
    float defocusAngle;           /* default 0 */
} rt_camera_settings;

/* One storage-buffer binding: pointer + element count (not bytes). */
typedef struct rt_buffer_view {
    const void* data;
    size_t      count;
} rt_buffer_view;

/* The 18 storage buffers of descriptor set 1, in binding order
 * (raytrace.comp:83-171, HardwareRenderer.cpp:751-770). Element types:
 *   parentBVH   rt_parent_bvh_node          objects   rt_gpu_object
 *   materials   rt_gpu_material             triangle* float[4] (xyz used)
 *   aabbMins/aabbMaxs float[4] (xyz used)   bvh*      int32_t
 * The three UV bindings are never read by the shader; they may be {NULL,0}. */
typedef struct rt_scene_buffers {
    rt_buffer_view parentBVH;                    /* binding 0  */
    rt_buffer_view objects;                      /* binding 1  */
    rt_buffer_view materials;                    /* binding 2  */
    rt_buffer_view triangleV0s;                  /* binding 3  */
    rt_buffer_view triangleV1s;                  /* binding 4  */
    rt_buffer_view triangleV2s;                  /* binding 5  */
    rt_buffer_view triangleN0s;                  /* binding 6  */
    rt_buffer_view triangleN1s;                  /* binding 7  */
    rt_buffer_view triangleN2s;                  /* binding 8  */
    rt_buffer_view triangleUV0s;                 /* binding 9  */
    rt_buffer_view triangleUV1s;                 /* binding 10 */
    rt_buffer_view triangleUV2s;                 /* binding 11 */
    rt_buffer_view aabbMins;                     /* binding 12 */
    rt_buffer_view aabbMaxs;                     /* binding 13 */
    rt_buffer_view bvhNodeLeftChildren;          /* binding 14 */
    rt_buffer_view bvhNodeRightChildren;         /* binding 15 */
    rt_buffer_view bvhNodeTriangleStartIndices;  /* binding 16 */
    rt_buffer_view bvhNodeTriangleCounts;        /* binding 17 */
} rt_scene_buffers;

/* Work counters summed over every pixel of one dispatch. bvhNodeTests and
 * triangleTests are the shader's own counters (raytrace.comp:463,492,533,549);
 * rays counts GetHit invocations (raytrace.comp:699,733,845). */
typedef struct rt_frame_stats {
    uint64_t rays;
    uint64_t bvhNodeTests;
    uint64_t triangleTests;
    uint64_t samples;
} rt_frame_stats;

#ifdef __cplusplus
}
#endif

#if defined(__cplusplus)
static_assert(sizeof(rt_push_constants) == 144, "push constants are 144 bytes");
static_assert(offsetof(rt_push_constants, raysPerPixel) == 12, "layout");
static_assert(offsetof(rt_push_constants, sunDirection) == 96, "layout");
static_assert(offsetof(rt_push_constants, renderMode) == 124, "layout");
static_assert(offsetof(rt_push_constants, depthDebugScale) == 136, "layout");
static_assert(sizeof(rt_aabb) == 32, "AABB is 32 bytes");
static_assert(sizeof(rt_bvh_node) == 48, "BVHNode is 48 bytes");
static_assert(sizeof(rt_parent_bvh_node) == 64, "ParentBVHNode is 64 bytes");
static_assert(offsetof(rt_parent_bvh_node, objectIndex) == 48, "layout");
static_assert(sizeof(rt_gpu_object) == 80, "GPUObject is 80 bytes");
static_assert(offsetof(rt_gpu_object, inverseTransform) == 16, "layout");
static_assert(sizeof(rt_gpu_material) == 48, "GPUMaterial is 48 bytes");
static_assert(offsetof(rt_gpu_material, absorbtion) == 32, "layout");
#else
_Static_assert(sizeof(rt_push_constants) == 144, "push constants are 144 bytes");
_Static_assert(sizeof(rt_parent_bvh_node) == 64, "ParentBVHNode is 64 bytes");
_Static_assert(sizeof(rt_gpu_object) == 80, "GPUObject is 80 bytes");
_Static_assert(sizeof(rt_gpu_material) == 48, "GPUMaterial is 48 bytes");
#endif

#endif /* RT_TYPES_H */
