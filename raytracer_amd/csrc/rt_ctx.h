// rt_ctx.h - the context behind the opaque rt_ctx of include/rt_abi.h, shared by the host files of
// librt_amd.so (rt_host.cpp: scene, images, launches; rt_comm.hip: multi-GPU merge).
#ifndef RT_CTX_H
#define RT_CTX_H

#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_device.h"

struct rt_ctx {
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    bool timed = false;
    int lastFrames = 0;

    // scene in HBM
    DevScene scene{};
    void* dObjects = nullptr;
    void* dNodes = nullptr;
    void* dTris = nullptr;
    void* dNrms = nullptr;
    void* dMaterials = nullptr;
    void* dAnyOrder = nullptr;
    bool haveScene = false;
    size_t sceneBytes = 0;
    size_t nodeCount = 0, triCount = 0;

    // host copies of the re-laid-out scene (the caller's arrays may be gone after rt_upload_scene) and the
    // near-first scene built from them on first use (RT_OPT_TRAVERSAL = 1): its own object records
    // (tight world boxes, roots in the rebuilt trees), and record / triangle arrays that hold the
    // rebuilt trees (behind a copy of the reference-order records when some object has to keep those)
    std::vector<DevObject> hObjects, hFastObjects;     // host copies of the object records (reference scene, near-first twin)
    float anyOrderSun[3] = { NAN, NAN, NAN };           // sun direction the any-hit order tables are sorted for
    std::vector<DevNode> hNodes;
    std::vector<DevTri> hTris;
    int hStackDepth = 0;
    DevScene fastScene{};
    void* dFastObjects = nullptr;
    void* dFastNodes = nullptr;
    void* dFastTris = nullptr;
    void* dFastAnyOrder = nullptr;
    bool haveFast = false;
    int fastObjects = 0;          // objects that walk a rebuilt tree
    size_t fastNodeCount = 0, fastTriCount = 0;
    int optTraversal = 0, sahMaxLeaf = 2;
    // the wide twin of an order-free near-first scene (four-child records, pipeline 3 only): same object
    // boxes, triangles and any-hit tables as fastScene, object roots in dWideNodes
    DevScene wideScene{};
    void* dWideObjects = nullptr;
    void* dWideNodes = nullptr;
    void* dWideAnyOrder = nullptr;      // its own order tables: the wide scene stores the objects in another order
    std::vector<DevObject> hWideObjects;
    bool haveWide = false;
    size_t wideNodeCount = 0;
    int optWide = 1;
    int optTlas = 1;              // wide scenes: hierarchy over the BVH parents' boxes
    int teamWalkers = 4;          // pipeline 4: walker waves per 16-wave workgroup

    // the reference-order scene once more with its trees folded into four-child records in the reference's visiting order
    // (pipeline 3 only): same objects in the same order, triangles and tables of `scene`, object roots in dRefWideNodes
    DevScene refWideScene{};
    void* dRefWideObjects = nullptr;
    void* dRefWideNodes = nullptr;
    bool haveRefWide = false;
    size_t refWideNodeCount = 0;

    // images
    int width = 0, height = 0;
    void* ownDraw = nullptr;
    void* ownAccum = nullptr;
    void* draw = nullptr;
    void* accum = nullptr;

    DevCounters* dCounters = nullptr;
    unsigned long long* dQueue = nullptr;   // pixel counter of the wave-scheduled kernel
    int computeUnits = 0;
    int wavesPerCU = 12;
    int policy[5] = { 0, 0, 0, 4, 8 };   // biasObject, biasTrav, biasShade, quorumMin, quorumDiv

    // wavefront pipeline: the image is dealt out to wfParts independent partitions (tile split inside
    // one GPU), each with its own job buffers and stream, so that the thin end of one partition's
    // kernels overlaps with the bulk of another's
    static const int kMaxParts = 8;
    DevWavefront wf[kMaxParts]{};
    void* wfMem[kMaxParts] = {};
    hipStream_t wfStream[kMaxParts] = {};
    hipEvent_t wfFork = nullptr, wfJoin[kMaxParts] = {};
    size_t wfKey[6] = {};
    int wfRounds = 5, wfRefill = 16, wfWalkWaves = 12, wfParts = 1;
    int framesPerLaunch = 32;
    int optPieces = 1;            // RT_OPT_LAUNCH_PIECES
    void* dPartial = nullptr;     // per-frame contributions of a launch in pieces: frames x width x height float4
    size_t partialBytes = 0;

    int optFast = 0, optPipeline = 3, tileRank = 0, tileWorld = 1, tileSize = 64;
    // multi-GPU merge (rt_comm.hip): RCCL communicator of this context and the tile-exchange buffer
    void* comm = nullptr;
    int commRank = 0, commWorld = 1;
    void* dGather = nullptr;
    size_t gatherBytes = 0;

    std::string error;
};

// Records the message on the context (and as the last context-less error) and returns `code`.
int rt_fail(rt_ctx* ctx, int code, const std::string& msg);
int rt_fail_hip(rt_ctx* ctx, hipError_t e, const char* what);
const char* rt_global_error();
// For `catch (...)` handlers of exported functions: classifies the exception in flight, records a message.
int rt_caught(rt_ctx* ctx) noexcept;

#define RT_HIP(ctx, call)                                         \
    do {                                                          \
        hipError_t e_ = (call);                                   \
        if (e_ != hipSuccess) return rt_fail_hip(ctx, e_, #call); \
    } while (0)

#endif
