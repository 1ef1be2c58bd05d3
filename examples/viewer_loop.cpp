// viewer_loop.cpp - drives rtamd::HardwareRenderer (include/rt_renderer.hpp) the way the reference's
// MainLoop drives its renderer (HardwareRenderer.cpp:973-1011), with a scripted input sequence in
// place of the SDL event loop: idle frames, mouse lock, W / D movement, a mouse turn, a window
// resize, a scene edit (SceneEditorUI's pattern: edit the lists - change, remove, append -, RebufferSceneData,
// SetRefreshAccumulation) and an offline render request (SetRenderFrames + SetDoRender).
//
//   viewer_loop <models_dir> <out.png>      full session on HIP device 0
//   viewer_loop --camera-only               camera controller only, no device needed
//
// Prints one line per step: step, frame counter after the step, camera position and direction.
#include <cstdio>
#include <cstring>
#include <string>

#include "../include/rt_renderer.hpp"

using rtamd::FrameInput;
using rtamd::HardwareRenderer;

static void report(int step, const HardwareRenderer& r)
{
    const rtamd::CameraSettings& c = r.m_camera;
    std::printf("step %d frame %d pos %.9g %.9g %.9g dir %.9g %.9g %.9g locked %d refresh %d\n", step, r.m_pushConstants.frame,
                c.cameraPosition.x, c.cameraPosition.y, c.cameraPosition.z, c.cameraLookDirection.x, c.cameraLookDirection.y,
                c.cameraLookDirection.z, r.m_cameraController.m_bLockedMouse ? 1 : 0, r.m_bRefreshAccumulation ? 1 : 0);
}

int main(int argc, char** argv)
{
    const bool cameraOnly = argc > 1 && std::strcmp(argv[1], "--camera-only") == 0;
    if (!cameraOnly && argc < 3) {
        std::fprintf(stderr, "usage: %s <models_dir> <out.png> | --camera-only\n", argv[0]);
        return 2;
    }
    try {
        HardwareRenderer r;
        if (!cameraOnly) {
            r.InitializeRenderer(320, 180, argv[1]);
            r.m_renderPath = argv[2];
        }
        int step = 0;
        FrameInput idle;
        idle.deltaT = 0.016;
        for (int k = 0; k < 3; ++k) { r.Step(idle); report(++step, r); }          // three accumulated frames

        FrameInput in = idle;
        in.keyW = true;                                                            // mouse not locked: W is ignored
        r.Step(in); report(++step, r);

        in = FrameInput();
        in.escapePressed = true; in.keyW = true; in.deltaT = 0.5;                  // lock, and W moves in the same frame
        r.Step(in); report(++step, r);

        in = FrameInput();
        in.mouseDeltaX = 1.8f; in.deltaT = 1.0;                                    // yaw -90 -> 0: look along +x
        r.Step(in); report(++step, r);

        in = FrameInput();
        in.keyD = true; in.deltaT = 0.25;                                          // strafe along cross(dir, up) = +z
        r.Step(in); report(++step, r);

        if (!cameraOnly) {
            in = idle;
            in.windowWidth = 256; in.windowHeight = 144;                           // resize: new images, accumulation restarts
            r.Step(in); report(++step, r);
            r.Step(idle); report(++step, r);
            r.Step(idle); report(++step, r);

            r.m_sceneMaterials[1].albedo[0] = 0.9f; r.m_sceneMaterials[1].albedo[1] = 0.2f; r.m_sceneMaterials[1].albedo[2] = 0.2f;
            r.m_sceneObjects[6].position.y += 0.5f;
            // the lists are the front end's own: remove the last wall, append a material and a cube that uses it
            r.m_sceneObjects.erase(r.m_sceneObjects.begin() + 7);
            rtamd::GPUMaterial blue;
            rt_default_material(&blue);
            blue.albedo[0] = 0.2f; blue.albedo[1] = 0.3f; blue.albedo[2] = 0.9f;
            r.m_sceneMaterials.push_back(blue);
            const std::string cube = std::string(argv[1]) + "/cube.obj";
            r.LoadModel(cube);
            r.AddSceneObject(cube, rtamd::Vec3(3.0f, 1.0f, 2.0f), rtamd::Vec3(0.0f, 30.0f, 0.0f), rtamd::Vec3(1.0f, 1.0f, 1.0f), blue);
            r.RebufferSceneData();
            r.SetRefreshAccumulation();
            r.SetRenderFrames(8);
            r.SetDoRender();
            r.Step(idle); report(++step, r);                                       // renders frames 0..7 and writes the PNG
            std::printf("size %d %d objects %zu materials %zu\n", r.GetWidth(), r.GetHeight(), r.m_sceneObjects.size(), r.m_sceneMaterials.size());
        }
    } catch (const rtamd::Error& e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
