"""raytracer_amd - Python access to librt_amd.so, the MI355X (gfx950) implementation of
the reference renderer's per-pixel path-tracing dispatch.

The package is a thin ctypes layer over the C ABI declared in include/rt_abi.h; all
rendering happens in hand-written HIP kernels inside the shared library. There is no
Python or CPU rendering path: if the library is missing, or no HIP device is present,
the calls raise.
"""
from .api import (  # noqa: F401
    LIB_PATH, MODELS_DIR, PushConstants, CameraSettings, Material, FrameStats, SceneBuffers, RtError,
    load_library, Scene, Renderer, default_push_constants, default_camera, camera_to_push_constants,
    rgba32f_to_rgba8, comm_unique_id, merge_accum_local, gather_tiles_local, RT_IMAGE_DRAW, RT_IMAGE_ACCUM, RT_OPT_FAST_MATH, RT_OPT_PIPELINE, RT_OPT_TILE_RANK,
    RT_OPT_TILE_WORLD, RT_OPT_TILE_SIZE, RT_OPT_WAVES_PER_CU, RT_OPT_SCHED_BIAS_OBJECT, RT_OPT_SCHED_BIAS_TRAV,
    RT_OPT_SCHED_BIAS_SHADE, RT_OPT_SCHED_QUORUM_MIN, RT_OPT_SCHED_QUORUM_DIV, RT_OPT_WF_ROUNDS, RT_OPT_WF_REFILL, RT_OPT_WF_WALK_WAVES, RT_OPT_WF_PARTS, RT_OPT_FRAMES_PER_LAUNCH, RT_INFO_STACK_DEPTH, RT_INFO_RECORDS, RT_INFO_TRIANGLES,
    RT_OPT_TRAVERSAL, RT_OPT_SAH_MAX_LEAF, RT_OPT_TEAM_WALKERS, RT_OPT_WIDE_RECORDS, RT_OPT_LAUNCH_PIECES, RT_INFO_WIDE_RECORDS, RT_INFO_WIDE_STACK_NEED, RT_INFO_TLAS_RECORDS, RT_INFO_FAST_OBJECTS, RT_INFO_FAST_RECORDS, RT_INFO_FAST_TRIANGLES, RT_INFO_FAST_STACK_DEPTH,
)
