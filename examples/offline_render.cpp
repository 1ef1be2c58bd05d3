// offline_render.cpp - a C++ host that uses librt_amd.so through its C ABI only (no Python, no
// torch): builds the reference's default scene, renders N accumulated frames and writes the PNG,
// i.e. what HardwareRenderer::ProduceRender does (HardwareRenderer.cpp:1029-1060).
//
//   offline_render <models_dir> <out.png> [width height frames pipeline]
//
// Prints the reference's timing line and a 64-bit checksum of the draw image.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/rt_abi.h"

static int die(const char* what, rt_ctx* ctx)
{
    std::fprintf(stderr, "%s: %s\n", what, rt_last_error(ctx));
    return 1;
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <models_dir> <out.png> [width height frames pipeline]\n", argv[0]);
        return 2;
    }
    const int width = argc > 3 ? std::atoi(argv[3]) : 1280, height = argc > 4 ? std::atoi(argv[4]) : 720;
    const int frames = argc > 5 ? std::atoi(argv[5]) : 16, pipeline = argc > 6 ? std::atoi(argv[6]) : 0;

    rt_scene* scene = rt_scene_create();
    if (rt_scene_init_default(scene, argv[1], "dragon_lowres.obj") != RT_OK || rt_scene_buffer_data(scene) != RT_OK) {
        std::fprintf(stderr, "scene: %s\n", rt_scene_last_error(scene));
        return 1;
    }
    rt_scene_buffers buffers;
    rt_scene_get_buffers(scene, &buffers);

    rt_ctx* ctx = nullptr;
    if (rt_create(0, &ctx) != RT_OK) return die("rt_create", nullptr);
    if (rt_set_option(ctx, RT_OPT_PIPELINE, pipeline) != RT_OK) return die("rt_set_option", ctx);
    if (rt_upload_scene(ctx, &buffers) != RT_OK) return die("rt_upload_scene", ctx);
    if (rt_resize(ctx, width, height) != RT_OK) return die("rt_resize", ctx);

    rt_push_constants pc;
    rt_camera_settings cam;
    rt_default_push_constants(&pc);
    rt_default_camera(&cam);
    rt_camera_to_push_constants(&cam, width, height, &pc);
    pc.parentBVHCount = (int)buffers.parentBVH.count;

    double seconds = 0.0;
    if (rt_produce_render(ctx, &pc, frames, argv[2], &seconds) != RT_OK) return die("rt_produce_render", ctx);
    std::printf("Render took %g seconds.\n", seconds);
    std::printf("Wrote render to %s\n", argv[2]);

    std::vector<float> img((size_t)width * height * 4);
    if (rt_read_image(ctx, RT_IMAGE_DRAW, img.data(), img.size() * sizeof(float)) != RT_OK) return die("rt_read_image", ctx);
    uint64_t h = 1469598103934665603ull;   // FNV-1a over the image bytes
    const unsigned char* p = reinterpret_cast<const unsigned char*>(img.data());
    for (size_t i = 0; i < img.size() * sizeof(float); ++i) h = (h ^ p[i]) * 1099511628211ull;
    float ms_total = 0.0f, ms_per = 0.0f;
    rt_last_render_ms(ctx, &ms_total, &ms_per);
    std::printf("frames %d  frame counter %d  device ms/frame %.3f  image fnv1a %016llx\n", frames, pc.frame, ms_per,
                (unsigned long long)h);
    rt_destroy(ctx);
    rt_scene_destroy(scene);
    return 0;
}
