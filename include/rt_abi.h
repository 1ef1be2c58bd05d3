/*
 * rt_abi.h - C ABI of librt_amd.so, the MI355X (gfx950, HIP) replacement for the
 * reference renderer's Vulkan compute dispatch of raytrace.comp.
 *
 * The reference has no plugin/FFI layer: its ray tracer sits behind private
 * methods of HardwareRenderer. The boundary reproduced here is the Vulkan
 * binding contract between that class and its shader: two W x H rgba32f storage
 * images, 18 storage buffers, one 144-byte push-constant block, one dispatch per
 * frame. Each export names the reference call site it replaces.
 *
 * Conventions: every function returns RT_OK (0) or a negative rt_status; no C++
 * exception crosses the ABI; rt_last_error() gives the text of the last failure
 * on that context (or of the last context-less failure when ctx is NULL).
 * One rt_ctx is externally synchronised (not re-entrant), like the reference's
 * single render thread; distinct contexts are independent. All device work of a
 * context is enqueued on one HIP stream; calls that hand data back to the host
 * (rt_read_image, rt_frame_stats_*) synchronise that stream themselves.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;
typedef struct rt_scene rt_scene;

typedef enum rt_status {
    RT_OK                 = 0,
    RT_ERR_INVALID_ARG    = -1,
    RT_ERR_NO_DEVICE      = -2,
    RT_ERR_HIP            = -3,
    RT_ERR_NO_SCENE       = -4,
    RT_ERR_NO_IMAGES      = -5,
    RT_ERR_SCENE_INVALID  = -6,
    RT_ERR_IO             = -7,
    RT_ERR_UNSUPPORTED    = -8,
    RT_ERR_NOMEM          = -9      /* host memory exhausted (std::bad_alloc caught at the boundary) */
} rt_status;

enum { RT_IMAGE_DRAW = 0, RT_IMAGE_ACCUM = 1 };

/* Options for rt_set_option. */
enum {
    /* 0 = strict IEEE f32 arithmetic, no fused multiply-add (the flavour that is
     * compared value-for-value with the CPU oracle); 1 = same algorithm with FMA
     * contraction allowed. Default 0. */
    RT_OPT_FAST_MATH   = 1,
    /* Kernel that renders modes 0/3/4 (default 3). 0 = lock-step megakernel: one lane per pixel,
     * the 64 lanes of a wave walk the same GetHit call together. 1 = wave-scheduled persistent
     * megakernel: every lane runs its own state machine, the wave votes whether to run a BVH
     * round (only the lanes inside a BVH walk step) or an object/shade iteration (everything
     * else) next, finished lanes pull new pixels from a global counter. Same arithmetic, same
     * results. 2 = wavefront: a ray that enters the box of a BVH parent is suspended (state in
     * HBM) and a second kernel, in which every lane takes the next ray as soon as its walk is
     * done, finishes its GetHit; further rounds of the path kernel pick the paths up again.
     * Needs about 430 bytes of device memory per pixel. Same arithmetic, same results. 3 = lock-step
     * with two samples per lane: when an rt_render_frames launch holds at least two samples of a
     * pixel (frames per launch x raysPerPixel), every lane traces two of them together and walks
     * its second ray while other lanes of the wave are still on their first, so the wave waits for
     * the largest sum of two walks instead of twice for the largest walk. Each ray sees objects,
     * nodes and triangles in the reference's order and the samples are added in sequence order:
     * same results. A launch that holds one sample per pixel (rt_render_frame with 1 ray per pixel)
     * pairs two vertically adjacent pixels per lane instead (tile sizes that are multiples of 16);
     * debug views and counters use kernel 0. Runs of consecutive BVH parents are gone through lane by
     * lane in kernels 0 and 3 (every lane visits the parents its own rays can enter). The default. 4 = path waves + walker waves: a
     * workgroup of 16 waves per compute unit; its path waves (12) run kernel 0's code except the walks of
     * BVH parents, which they hand - one job per ray, through a ring in LDS - to the workgroup's walker
     * waves (4), whose lanes take the next job as soon as they are free. Same arithmetic, same results. */
    RT_OPT_PIPELINE    = 2,
    /* Tile partition for multi-GPU tile split: this context renders only the
     * tiles t with t % RT_OPT_TILE_WORLD == RT_OPT_TILE_RANK (tiles are
     * RT_OPT_TILE_SIZE x RT_OPT_TILE_SIZE pixels, numbered along a serpentine path:
     * row 0 left to right, row 1 right to left, ...). Pixels of other tiles are left
     * untouched. Defaults: rank 0, world 1, size 64. */
    RT_OPT_TILE_RANK   = 3,
    RT_OPT_TILE_WORLD  = 4,
    RT_OPT_TILE_SIZE   = 5,
    /* Resident 64-lane waves per compute unit launched by pipeline 1 (default 12). */
    RT_OPT_WAVES_PER_CU = 6,
    /* Round selection of pipeline 1 (tuning knobs, results do not depend on them): a BVH round
     * runs when (walking lanes - BIAS_TRAV) >= (other lanes - BIAS_SHADE); it ends when fewer
     * than max(min(QUORUM_MIN, lanes_at_start), lanes_at_start / QUORUM_DIV) lanes are still
     * walking. BIAS_OBJECT is reserved. */
    RT_OPT_SCHED_BIAS_OBJECT = 7, RT_OPT_SCHED_BIAS_TRAV = 8, RT_OPT_SCHED_BIAS_SHADE = 9,
    RT_OPT_SCHED_QUORUM_MIN = 10, RT_OPT_SCHED_QUORUM_DIV = 11,
    /* Pipeline 2: number of suspend / walk / resume rounds before the remaining paths are
     * finished with in-place walks (default 5; 0 = in-place walks only), and the number of
     * free lanes at which a walking wave fetches new rays (default 16), and the number of resident
     * waves per compute unit of the walk kernel (default 12). WF_PARTS (default 1, at most 3): the
     * tiles are dealt out to that many partitions, each rendered by its own chain of kernels on
     * its own stream, so that the thin end of one chain overlaps with the bulk of another. Every
     * chain needs a hardware queue of its own beside the context's (a process has four): more are
     * refused, because chains that share a queue serialise behind each other's fork / join events;
     * two processes that share one GPU should stay at 1. */
    RT_OPT_WF_ROUNDS = 12, RT_OPT_WF_REFILL = 13, RT_OPT_WF_WALK_WAVES = 14, RT_OPT_WF_PARTS = 15,
    /* Pipeline 0, rt_render_frames in path-tracing mode: this many consecutive frames of the call are
     * rendered by one kernel launch, every pixel keeping its accumulation value in registers from
     * one frame to the next - the additions a sequence of single-frame dispatches performs, in the
     * same order, so both images are the same bit for bit; the draw image is written for the last
     * frame of a launch only. Saves the launch ramps and draw-image writes. Default 32 (fewer where the
     * pieces of RT_OPT_LAUNCH_PIECES would take more than 4 GiB); 1 = one launch per frame. rt_render_frame
     * (one frame) is not affected. */
    RT_OPT_FRAMES_PER_LAUNCH = 16,
    /* Traversal of the BVH parents by the lock-step kernels (pipelines 0 and 3) in path-tracing mode.
     * 0 (default) = the reference's trees in the reference's order (left child first,
     * raytrace.comp:603-604): results bit-identical to the CPU oracle. 1 = near-first: every object whose
     * transform shrinks nothing walks a tree rebuilt at first use (binned surface-area heuristic, tight
     * boxes, at most RT_OPT_SAH_MAX_LEAF triangles per leaf, default 2), the nearer child first, children
     * behind the closest hit dropped; every object gets a tight world box. The hit GetHit returns is the
     * triangle with the smallest world-space distance in either order, so the images differ only where two
     * triangles tie exactly (shared edges): within the stated tolerance (>= 99.5 % of pixels within
     * 1e-4 + 1e-4*|ref| at 1 spp, PSNR >= 45 dB converged), not bit for bit. Objects scaled below 1 keep
     * the reference's tree and order (their result depends on it, raytrace.comp:553-572). The counting
     * builds, the debug views, pipelines 1 and 2 and dispatches with parentBVHCount below the uploaded
     * object count always use 0. */
    RT_OPT_TRAVERSAL = 17, RT_OPT_SAH_MAX_LEAF = 18,
    /* Pipeline 4: walker waves per workgroup of 16 waves (default 4; the other waves are path waves). */
    RT_OPT_TEAM_WALKERS = 19,
    /* Near-first traversal on pipeline 3, scenes in which no object keeps the reference's order: 1 (default)
     * = the rebuilt trees are folded into four-child 128-byte records (same boxes, same leaves), the four
     * children visited nearest first - half as many dependent fetches on a root-to-leaf path - and, in scenes
     * with 8 or more objects that own a tree, a hierarchy of such records over the objects' world boxes lets
     * a ray find the objects it crosses nearest first instead of testing every object's box (the reference
     * loops over all of them, raytrace.comp:458-471; the hit is the same, only the order of the tests is
     * not); 2 = four-child records without that hierarchy; 0 = two-child records. In reference order
     * (RT_OPT_TRAVERSAL 0) pipeline 3 walks the reference's own trees folded into four-child records whose
     * slots stand in the reference's visiting order - the same leaves in the same order, bit-identical
     * images - unless this option is 0. Other pipelines and near-first scenes in which some object keeps the
     * reference's order always walk two-child records. */
    RT_OPT_WIDE_RECORDS = 20,
    /* Pipeline 3, rt_render_frames launches of more than two frames: 1 = a launch is rendered in pieces of two
     * frames - four times as many, shorter waves, each storing its frames' contributions, which a second small
     * kernel adds to the accumulation image in frame order (the same additions in the same order: same images) -
     * so that a launch that holds little more than the waves the chip runs at one time (a rank's share of a tile
     * split, a small image) does not last as long as all the frames of its most expensive tile, and faster on
     * every scene measured (the default scene 1.5 %, the 1M-triangle scene 12 %); the default. 0 = one piece;
     * 2 = in pieces only when the launch holds fewer than three times the resident waves. Needs 16 bytes of
     * device memory per pixel and frame of a launch. */
    RT_OPT_LAUNCH_PIECES = 21,
    /* Read-only (rt_get_option) facts about the uploaded scene: traversal stack entries a lane can
     * need (longest chain of two-child records), number of two-child records, and number of
     * triangle records in leaf order (a triangle that several leaves share counts once per leaf). */
    RT_INFO_STACK_DEPTH = 100, RT_INFO_RECORDS = 101, RT_INFO_TRIANGLES = 102,
    /* The same for the near-first scene (built on first use): objects that walk a rebuilt tree, two-child
     * records and triangle records in its arrays, traversal stack entries. */
    RT_INFO_FAST_OBJECTS = 103, RT_INFO_FAST_RECORDS = 104, RT_INFO_FAST_TRIANGLES = 105, RT_INFO_FAST_STACK_DEPTH = 106,
    /* Four-child records of the near-first scene (0 when the scene has none, see RT_OPT_WIDE_RECORDS) and the
     * traversal stack entries a lane can need while walking them. */
    RT_INFO_WIDE_RECORDS = 107, RT_INFO_WIDE_STACK_NEED = 108,
    /* Records of the hierarchy over the objects' boxes among them (0: the scene has none, see RT_OPT_WIDE_RECORDS). */
    RT_INFO_TLAS_RECORDS = 109
};

/* ---- device context ------------------------------------------------------ */

/* Replaces InitializeVulkan's device/queue/pipeline bring-up
 * (HardwareRenderer.cpp:28-129,441-472). device_id is a HIP device ordinal. */
int rt_create(int device_id, rt_ctx** out_ctx);
void rt_destroy(rt_ctx* ctx);

/* Use an existing HIP stream (hipStream_t passed as void*) instead of the
 * context's own; NULL restores the context's stream. */
int rt_set_stream(rt_ctx* ctx, void* hip_stream);

int rt_set_option(rt_ctx* ctx, int option, int value);
int rt_get_option(rt_ctx* ctx, int option, int* value);

/* Replaces BufferSceneData / RebufferSceneData (HardwareRenderer.cpp:651-771,
 * 787-791): takes the 18 host arrays exactly as the reference uploads them,
 * validates them, re-lays them out for the GPU and copies them to HBM. The
 * library copies; the caller may free its arrays after return. Re-callable. */
int rt_upload_scene(rt_ctx* ctx, const rt_scene_buffers* buffers);

/* Replaces InitializeSwapchain / RecreateSwapchain image creation
 * (HardwareRenderer.cpp:131-158,232-245): (re)allocates the W x H rgba32f draw
 * and accumulation images and clears both. */
int rt_resize(rt_ctx* ctx, int width, int height);

/* Same, but renders into caller-owned device memory (two W*H*16-byte buffers,
 * row 0 = top row). Nothing is cleared. Pass NULLs to return to owned images. */
int rt_bind_images(rt_ctx* ctx, void* draw_dev, void* accum_dev, int width, int height);

/* Replaces RefreshAccumulation (HardwareRenderer.cpp:840-848): zero the
 * accumulation image (all four channels). The caller owns the frame counter. */
int rt_clear_accumulation(rt_ctx* ctx);

/* Replaces DispatchRayTracingCommands' vkCmdPushConstants + vkCmdDispatch
 * (HardwareRenderer.cpp:833-837): one dispatch of the shader's main() over every
 * pixel with these push constants. Asynchronous on the context's stream. */
int rt_render_frame(rt_ctx* ctx, const rt_push_constants* pc);

/* Replaces ProduceRender's loop (HardwareRenderer.cpp:1039-1043): `count`
 * dispatches with pc->frame = first_frame, first_frame + stride, ... A stride
 * above 1 is the sample split used across GPUs (rank r renders frames r, r+N..). */
int rt_render_frames(rt_ctx* ctx, const rt_push_constants* pc,
                     int first_frame, int stride, int count);

/* Replaces the per-frame vkDeviceWaitIdle (HardwareRenderer.cpp:969). */
int rt_sync(rt_ctx* ctx);

/* Replaces WriteDrawImageToFile's image->staging copy (HardwareRenderer.cpp:
 * 1063-1104): tightly packed row-major RGBA f32, row 0 = top. `bytes` must be
 * exactly W*H*16. Synchronises. */
int rt_read_image(rt_ctx* ctx, int which, float* dst_rgba, size_t bytes);

/* Loads an accumulation image (resume, or a merged multi-GPU sum). */
int rt_write_accum(rt_ctx* ctx, const float* src_rgba, size_t bytes);

/* draw = gamma(accum.rgb * 1/total_frames), draw.a = accum.a: the output half of
 * Pathtrace (raytrace.comp:794-801) applied to an accumulation image that was
 * merged from several GPUs. */
int rt_resolve(rt_ctx* ctx, int total_frames);

/* ---- several GPUs: one process per GPU, RCCL over xGMI ---------------------------------------
 * The reference renders on one GPU; these calls have no counterpart in it. They spread its progressive
 * accumulation (ProduceRender, HardwareRenderer.cpp:1030-1061) over the GPUs of a node:
 *   sample split: rank r of W renders frames r, r+W, ... (rt_render_frames(ctx, pc, r, W, n)) into its own
 *     accumulation image, rt_merge_accum sums the images of all ranks (one all-reduce of width*height*4
 *     floats), rt_resolve(ctx, n*W) produces the draw image;
 *   tile split: rank r renders all frames of the tiles it owns (RT_OPT_TILE_RANK = r, RT_OPT_TILE_WORLD = W),
 *     rt_gather_tiles hands every rank's tiles of both images to every other rank (one all-gather in which
 *     each rank sends its own 1/W of the pixels): bit-identical to a single-GPU render.
 * Rank 0 calls rt_comm_unique_id and passes the 128 bytes to the other ranks by any channel; every rank
 * calls rt_comm_init (collective, like ncclCommInitRank). librccl is loaded at that moment. The merges are
 * enqueued on the context's stream behind the render kernels; they do not synchronise the host. */
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(unsigned char id[RT_COMM_ID_BYTES]);
int rt_comm_init(rt_ctx* ctx, const unsigned char id[RT_COMM_ID_BYTES], int rank, int world);
int rt_comm_destroy(rt_ctx* ctx);
/* rank and size of the context's communicator (world = 0: none). */
int rt_comm_info(rt_ctx* ctx, int* rank, int* world);
int rt_merge_accum(rt_ctx* ctx);
int rt_gather_tiles(rt_ctx* ctx);
/* The same two merges among n contexts of ONE process (on one device or several) by device copies, no
 * communicator needed: context i is rank i of n. Synchronise. rt_merge_accum_local adds in rank order. */
int rt_merge_accum_local(rt_ctx** ctxs, int n);
int rt_gather_tiles_local(rt_ctx** ctxs, int n);

/* Runs the same dispatch as rt_render_frame on a counting build of the kernel
 * that writes no image and returns the work counters summed over all pixels
 * (the shader's bvhNodeTests/triangleTests in reference traversal order, plus
 * the number of GetHit calls). Synchronises. */
int rt_frame_stats_compute(rt_ctx* ctx, const rt_push_constants* pc, rt_frame_stats* out);

/* rt_frame_stats_compute for callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc. It has no counterpart in the reference. The call itself
 * allocates nothing, on the device or through the HIP runtime (only a failure builds its message on
 * the host heap): every wave of the counting kernel stores its counters into its own slot of a
 * file-scope device array that is part of the library's code object (4 MiB per GPU, held until the
 * process ends), and a second kernel adds the slots up into *out_dev. Differences from
 * rt_frame_stats_compute: always the lock-step counting kernel, whatever RT_OPT_PIPELINE says (same
 * counters); out_dev must be memory the GPU can write (device memory or pinned host memory); the call
 * only enqueues on the context's stream and does not synchronise (read *out_dev after rt_sync);
 * a dispatch of more than 131 072 waves of 8x8 pixels (8 388 608 pixels, e.g. above 3840x2160) is
 * refused with RT_ERR_UNSUPPORTED. All contexts on one GPU share that GPU's array: calls that overlap
 * on one GPU (different contexts or streams) race; calls on one stream are ordered. */
int rt_frame_stats_noalloc(rt_ctx* ctx, const rt_push_constants* pc, rt_frame_stats* out_dev);

/* Diagnostics of the wave-scheduled kernel (no counterpart in the reference): renders the
 * dispatch without writing images and returns, for k = 0 object/shade iterations and k = 1
 * BVH-round steps: out[2k] = wave-level executions, out[2k+1] = lanes that took part
 * (out[2k+1] / (64*out[2k]) is the lane participation); out[8] = waves. */
int rt_schedule_stats(rt_ctx* ctx, const rt_push_constants* pc, uint64_t out[16]);

/* Self-test hook (no counterpart in the reference): applies the kernels' exp2 (which = 0), exp (1),
 * log2 (2), acos (3) - the functions of rt_math.h that stand in for the GLSL built-ins - or the
 * random generator's k / 10000 (4, for k = 0 ... 9999) to n host floats on the device. A host that
 * includes rt_math.h must get the same bits. Synchronises. */
int rt_math_apply(rt_ctx* ctx, int which, const float* x, float* y, size_t n);

/* Milliseconds the last rt_render_frame(s) call spent on the device, measured with HIP events on
 * the stream the kernels were launched on: the total, and the total divided by the number of frames
 * the call rendered (one frame is one dispatch of the reference; rt_render_frames may put several
 * frames into one kernel launch, see RT_OPT_FRAMES_PER_LAUNCH). Synchronises. */
int rt_last_render_ms(rt_ctx* ctx, float* ms_total, float* ms_per_frame);

int rt_image_size(rt_ctx* ctx, int* width, int* height);
const char* rt_last_error(rt_ctx* ctx);
const char* rt_version(void);

/* ---- host helpers that restate the reference's CPU-side feeders ------------ */

/* Camera -> push constants, the arithmetic of DispatchRayTracingCommands
 * (HardwareRenderer.cpp:795-824). Fills pixel00Location, pixelDeltaU/V,
 * cameraPosition, defocusAngle, defocusDiskU/V; leaves every other field. */
int rt_camera_to_push_constants(const rt_camera_settings* cam, int width, int height,
                                rt_push_constants* pc);

/* Default-initialised push constants / camera / material
 * (RaytracerTypes.h:22-52,127-138, HardwareRenderer.h:68-75). */
void rt_default_push_constants(rt_push_constants* pc);
void rt_default_camera(rt_camera_settings* cam);
void rt_default_material(rt_gpu_material* m);

/* Scene builder with the semantics of HardwareRenderer's scene half. */
rt_scene* rt_scene_create(void);
void rt_scene_destroy(rt_scene* s);
const char* rt_scene_last_error(rt_scene* s);

/* m_sceneMaterials.push_back (HardwareRenderer.cpp:576-627). Returns the index. */
int rt_scene_add_material(rt_scene* s, const rt_gpu_material* m);
/* GetMaterialIndex (HardwareRenderer.cpp:479-488): first equal material or -1. */
int rt_scene_material_index(rt_scene* s, const rt_gpu_material* m);
/* LoadModel (HardwareRenderer.cpp:1157-1314): OBJ -> triangles -> midpoint BVH,
 * appended to the global arrays; cached by path. */
int rt_scene_load_model(rt_scene* s, const char* obj_path);
/* AddSceneObject (HardwareRenderer.cpp:490-500); rotation in degrees.
 * Returns the object index. */
int rt_scene_add_object(rt_scene* s, const char* model_path, const float position[3],
                        const float rotation_deg[3], const float scale[3],
                        const rt_gpu_material* material);
/* Scene-editor edits (SceneEditorUI.hpp:23-97). */
int rt_scene_set_object(rt_scene* s, int index, const float position[3],
                        const float rotation_deg[3], const float scale[3], int material_index);
int rt_scene_set_material(rt_scene* s, int index, const rt_gpu_material* m);
/* Empty the object list / the material list (the loaded models stay cached), so that a front end
 * whose own lists shrank can hand them over again from scratch: the list half of ClearSceneData
 * (HardwareRenderer.cpp:773-785), which RebufferSceneData runs before every BufferSceneData. */
int rt_scene_clear_objects(rt_scene* s);
int rt_scene_clear_materials(rt_scene* s);
/* What the scene editor reads back (SceneEditorUI.hpp:23-97 walks m_sceneObjects and
 * m_sceneMaterials): the number of objects / materials and one entry of either list. model_path
 * may be NULL; otherwise it receives the object's model path (model_path_cap bytes available). */
int rt_scene_object_count(rt_scene* s);
int rt_scene_material_count(rt_scene* s);
int rt_scene_get_object(rt_scene* s, int index, char* model_path, size_t model_path_cap, float position[3],
                        float rotation_deg[3], float scale[3], int* material_index);
int rt_scene_get_material(rt_scene* s, int index, rt_gpu_material* out);
/* BufferSceneData's CPU half (HardwareRenderer.cpp:653-657): run
 * ConvertSceneObjectToGPUObject over every object, rebuilding the parent/object
 * arrays. Must be called before rt_scene_get_buffers. */
int rt_scene_buffer_data(rt_scene* s);
/* Views into the scene's arrays (valid until the scene is modified/destroyed). */
int rt_scene_get_buffers(rt_scene* s, rt_scene_buffers* out);
/* InitializeScene (HardwareRenderer.cpp:574-649): 9 materials, 3 models, 8
 * objects. models_dir holds flat_quad.obj, icosphere.obj and dragon_file (the
 * reference names dragon.obj; that file is not distributed, dragon_lowres.obj is). */
int rt_scene_init_default(rt_scene* s, const char* models_dir, const char* dragon_file);
/* Per-model structure facts: triangles, BVH nodes (excluding the parent), depth. */
int rt_scene_model_info(rt_scene* s, const char* model_path, int* triangles, int* nodes,
                        int* max_depth, int* max_leaf);

/* WriteDrawImageToFile's float -> 8-bit conversion (HardwareRenderer.cpp:1099-
 * 1125): clamp to [0,1], x*255+0.5 truncated, alpha forced to 255. */
void rt_rgba32f_to_rgba8(const float* src, uint8_t* dst, size_t pixel_count);

/* stbi_write_png's role in WriteDrawImageToFile (HardwareRenderer.cpp:1127-1134): an 8-bit RGBA
 * PNG of width x height tightly packed pixels, row 0 = top. The file is not compressed; a decoder
 * returns exactly the bytes passed in. */
int rt_write_png_rgba8(const char* path, const uint8_t* rgba, int width, int height);

/* WriteDrawImageToFile (HardwareRenderer.cpp:1063-1139): read the draw image back, convert it with
 * rt_rgba32f_to_rgba8 and write it to `path` (the caller chooses the name; the reference uses
 * Renders\\render_<date>.png). Synchronises. */
int rt_save_png(rt_ctx* ctx, const char* path);

/* ProduceRender (HardwareRenderer.cpp:1029-1060): dispatches frames pc->frame .. render_frames-1,
 * advancing pc->frame as the reference advances m_pushConstants.frame, then writes the PNG
 * (png_path may be NULL: no file) and returns in *seconds the wall-clock time of the whole call,
 * the number the reference prints as "Render took ... seconds". */
int rt_produce_render(rt_ctx* ctx, rt_push_constants* pc, int render_frames, const char* png_path, double* seconds);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
