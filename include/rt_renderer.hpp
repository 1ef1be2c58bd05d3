// rt_renderer.hpp - a HardwareRenderer-shaped C++ front over the C ABI of rt_abi.h.
//
// The reference's SDL/ImGui front end talks to one class (src/Renderers/Hardware/HardwareRenderer.h:94-309):
// its panels edit m_pushConstants, m_camera, m_sceneObjects and m_sceneMaterials, call
// SetRenderFrames / SetDoRender / SetRefreshAccumulation and RebufferSceneData
// (src/Interface/RaytracerSettingsUI.hpp:66-116, SceneEditorUI.hpp:23-97), and its CameraController
// moves the camera from keyboard and mouse state (CameraController.cpp:11-101). This header keeps
// that surface - same member names, same frame-loop semantics - over a HIP context, so a window
// layer only has to (1) fill a FrameInput from its event loop, (2) call Step(), (3) blit
// ReadDrawImage() to the screen. Nothing here needs SDL, ImGui, Vulkan or glm.
//
// What one MainLoop iteration does (HardwareRenderer.cpp:973-1011), and Step() does in the same order:
//   camera controller update -> window size check (RecreateSwapchain on change, :232-245) ->
//   RenderFrame (refresh accumulation if flagged :921-922, dispatch :924, wait :969) ->
//   m_pushConstants.frame++ (:1002) -> ProduceRender if requested (:1004-1005).
//
// Errors: the reference throws std::exception; every failing rt_* call here throws rtamd::Error
// (a std::runtime_error) carrying rt_last_error().
#ifndef RT_RENDERER_HPP
#define RT_RENDERER_HPP

#include <chrono>
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_abi.h"

namespace rtamd {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// glm::vec3 stand-in with exactly the operations the camera code needs; every operator is the
// component-wise binary32 operation glm performs.
struct Vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    Vec3 operator+(Vec3 o) const { return { x + o.x, y + o.y, z + o.z }; }
    Vec3 operator-(Vec3 o) const { return { x - o.x, y - o.y, z - o.z }; }
    Vec3 operator*(float s) const { return { x * s, y * s, z * s }; }
    Vec3& operator+=(Vec3 o) { x += o.x; y += o.y; z += o.z; return *this; }
    Vec3& operator-=(Vec3 o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    bool operator!=(Vec3 o) const { return x != o.x || y != o.y || z != o.z; }
};
inline Vec3 operator*(float s, Vec3 v) { return { s * v.x, s * v.y, s * v.z }; }
inline Vec3 cross(Vec3 a, Vec3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
inline Vec3 normalize(Vec3 v) { const float inv = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); return v * inv; }
inline float radians(float degrees) { return degrees * 0.01745329251994329576923690768489f; }

// HardwareRenderer.h:68-75
struct CameraSettings {
    Vec3 cameraPosition = Vec3(0, 1, 5);
    Vec3 cameraLookDirection = Vec3(0, 0, -1);
    float cameraFov = 90.0f;
    float focusDistance = 10.0f;
    float defocusAngle = 0.0f;
};

// HardwareRenderer.h:85-92
struct SceneObject {
    std::string modelName;
    Vec3 position = Vec3(0, 0, 0);
    Vec3 rotation = Vec3(0, 0, 0);   // degrees
    Vec3 scale = Vec3(1, 1, 1);
    int materialIndex = -1;
};

using GPUMaterial = rt_gpu_material;                 // RaytracerTypes.h:127-138
using RaytracePushConstants = rt_push_constants;     // RaytracerTypes.h:22-52

// What InputManager, PerformanceStats and Window hand to one MainLoop iteration
// (CameraController.cpp:13-14,18,29-60,78-79; Window.cpp:20-50). A window layer fills it per frame.
struct FrameInput {
    bool keyW = false, keyS = false, keyA = false, keyD = false, keySpace = false, keyX = false;   // GetKey: held
    bool escapePressed = false;     // GetKeyDown(SDLK_ESCAPE): went down this frame
    float mouseDeltaX = 0.0f, mouseDeltaY = 0.0f;
    double deltaT = 0.0;            // PerformanceStats::GetDeltaT, seconds
    int windowWidth = 0, windowHeight = 0;   // 0 x 0: keep the current size
    bool minimized = false;         // MainLoop skips the iteration (HardwareRenderer.cpp:982-983)
};

class HardwareRenderer;

// CameraController.h:7-29, CameraController.cpp:11-101
class CameraController {
public:
    float m_fMoveSpeed = 16.0f;
    float m_fSensitivity = 50.0f;
    bool m_bFirstMouse = false;
    float pitch = 0.0f, yaw = -90.0f;
    bool m_bLockedMouse = false;

    void Update(HardwareRenderer* renderer, const FrameInput& in);
    void UpdateCameraRotation(HardwareRenderer* renderer, const FrameInput& in);
    float GetSensitivity() const { return m_fSensitivity; }
    void SetSensitivity(float sensitivity) { m_fSensitivity = sensitivity; }
};

class HardwareRenderer {
public:
    // the state the reference's panels and camera controller edit directly
    CameraSettings m_camera;
    CameraController m_cameraController;
    RaytracePushConstants m_pushConstants;
    std::vector<SceneObject> m_sceneObjects;
    std::vector<GPUMaterial> m_sceneMaterials;

    bool m_bRefreshAccumulation = true;
    bool m_bDoRender = false;
    int m_iRenderFrames = 0;

    HardwareRenderer() { rt_default_push_constants(&m_pushConstants); }
    HardwareRenderer(const HardwareRenderer&) = delete;
    HardwareRenderer& operator=(const HardwareRenderer&) = delete;
    ~HardwareRenderer()
    {
        if (m_ctx) rt_destroy(m_ctx);
        if (m_scene) rt_scene_destroy(m_scene);
    }

    // InitializeRenderer (HardwareRenderer.cpp:1146-1155): device, images, default scene, buffers.
    // modelsDir must hold flat_quad.obj, icosphere.obj and dragonFile.
    void InitializeRenderer(int width, int height, const std::string& modelsDir, const std::string& dragonFile = "dragon_lowres.obj",
                            int device = 0)
    {
        InitializeDevice(width, height, device);
        InitializeScene(modelsDir, dragonFile);
        BufferSceneData();
    }

    // The device half alone, for callers that build their own scene.
    void InitializeDevice(int width, int height, int device = 0)
    {
        if (m_ctx) throw Error(RT_ERR_INVALID_ARG, "renderer already initialised");
        check(rt_create(device, &m_ctx), nullptr);
        m_scene = rt_scene_create();
        if (!m_scene) throw Error(RT_ERR_HIP, "out of host memory");
        RecreateSwapchain(width, height);
        m_bInitialized = true;
    }

    // RecreateSwapchain (HardwareRenderer.cpp:232-245): new draw / accumulation images, refresh flagged.
    void RecreateSwapchain(int width, int height)
    {
        check(rt_resize(m_ctx, width, height), m_ctx);
        m_width = width;
        m_height = height;
        m_bRefreshAccumulation = true;
    }

    // LoadModel (HardwareRenderer.cpp:1157-1314)
    void LoadModel(const std::string& filePath) { checkScene(rt_scene_load_model(m_scene, filePath.c_str())); }

    // GetMaterialIndex (HardwareRenderer.cpp:479-488): first material equal in every field, else -1
    int GetMaterialIndex(const GPUMaterial& material) const
    {
        for (size_t i = 0; i < m_sceneMaterials.size(); ++i) {
            const GPUMaterial& m = m_sceneMaterials[i];
            if (m.albedo[0] == material.albedo[0] && m.albedo[1] == material.albedo[1] && m.albedo[2] == material.albedo[2] &&
                m.smoothness == material.smoothness && m.fuzziness == material.fuzziness && m.emission == material.emission &&
                m.refractiveIndex == material.refractiveIndex && m.absorbtion[0] == material.absorbtion[0] &&
                m.absorbtion[1] == material.absorbtion[1] && m.absorbtion[2] == material.absorbtion[2])
                return (int)i;
        }
        return -1;
    }

    // AddSceneObject (HardwareRenderer.cpp:490-500)
    void AddSceneObject(const std::string& modelPath, Vec3 position, Vec3 rotation, Vec3 scale, const GPUMaterial& material)
    {
        SceneObject o;
        o.modelName = modelPath;
        o.position = position;
        o.rotation = rotation;
        o.scale = scale;
        o.materialIndex = GetMaterialIndex(material);
        m_sceneObjects.push_back(o);
    }

    // InitializeScene (HardwareRenderer.cpp:574-649): the hard-coded scene, then the editable lists
    // are read back from it.
    void InitializeScene(const std::string& modelsDir, const std::string& dragonFile = "dragon_lowres.obj")
    {
        checkScene(rt_scene_init_default(m_scene, modelsDir.c_str(), dragonFile.c_str()));
        m_sceneMaterials.resize((size_t)rt_scene_material_count(m_scene));
        for (size_t i = 0; i < m_sceneMaterials.size(); ++i) checkScene(rt_scene_get_material(m_scene, (int)i, &m_sceneMaterials[i]));
        m_sceneObjects.resize((size_t)rt_scene_object_count(m_scene));
        std::vector<char> name(4096);
        for (size_t i = 0; i < m_sceneObjects.size(); ++i) {
            SceneObject& o = m_sceneObjects[i];
            float p[3], r[3], s[3];
            checkScene(rt_scene_get_object(m_scene, (int)i, name.data(), name.size(), p, r, s, &o.materialIndex));
            o.modelName = name.data();
            o.position = Vec3(p[0], p[1], p[2]);
            o.rotation = Vec3(r[0], r[1], r[2]);
            o.scale = Vec3(s[0], s[1], s[2]);
        }
    }

    // BufferSceneData (HardwareRenderer.cpp:651-771): objects -> GPU objects and parent boxes on the
    // host, then all 18 arrays to the device. The editable lists are the truth, whatever was done to
    // them (entries edited, appended, removed, reordered): as RebufferSceneData does in the reference
    // (ClearSceneData, then BufferSceneData over the member vectors), the scene's lists are emptied and
    // filled again from them; the loaded models stay cached.
    void BufferSceneData()
    {
        checkScene(rt_scene_clear_materials(m_scene));
        for (const GPUMaterial& m : m_sceneMaterials) checkScene(rt_scene_add_material(m_scene, &m));
        checkScene(rt_scene_clear_objects(m_scene));
        for (size_t i = 0; i < m_sceneObjects.size(); ++i) {
            const SceneObject& o = m_sceneObjects[i];
            const float p[3] = { o.position.x, o.position.y, o.position.z };
            const float r[3] = { o.rotation.x, o.rotation.y, o.rotation.z };
            const float s[3] = { o.scale.x, o.scale.y, o.scale.z };
            checkScene(rt_scene_add_object(m_scene, o.modelName.c_str(), p, r, s, nullptr));
            checkScene(rt_scene_set_object(m_scene, (int)i, p, r, s, o.materialIndex));
        }
        checkScene(rt_scene_buffer_data(m_scene));
        rt_scene_buffers buffers;
        checkScene(rt_scene_get_buffers(m_scene, &buffers));
        check(rt_upload_scene(m_ctx, &buffers), m_ctx);
        m_parentCount = (int)buffers.parentBVH.count;
    }

    // RebufferSceneData (HardwareRenderer.cpp:787-791)
    void RebufferSceneData() { BufferSceneData(); }

    void SetDoRender() { m_bDoRender = true; }
    void SetRenderFrames(int frames) { m_iRenderFrames = frames; }
    void SetRefreshAccumulation() { m_bRefreshAccumulation = true; }

    // RenderFrame (HardwareRenderer.cpp:864-971) without the swapchain: refresh if flagged, fill the
    // camera part of the push constants (DispatchRayTracingCommands :795-824), dispatch, wait.
    void RenderFrame()
    {
        if (!m_bInitialized) return;
        if (m_bRefreshAccumulation) RefreshAccumulation();
        rt_camera_settings cam;
        cam.cameraPosition[0] = m_camera.cameraPosition.x; cam.cameraPosition[1] = m_camera.cameraPosition.y; cam.cameraPosition[2] = m_camera.cameraPosition.z;
        cam.cameraLookDirection[0] = m_camera.cameraLookDirection.x; cam.cameraLookDirection[1] = m_camera.cameraLookDirection.y;
        cam.cameraLookDirection[2] = m_camera.cameraLookDirection.z;
        cam.cameraFov = m_camera.cameraFov;
        cam.focusDistance = m_camera.focusDistance;
        cam.defocusAngle = m_camera.defocusAngle;
        check(rt_camera_to_push_constants(&cam, m_width, m_height, &m_pushConstants), nullptr);
        m_pushConstants.parentBVHCount = m_parentCount;
        check(rt_render_frame(m_ctx, &m_pushConstants), m_ctx);
        check(rt_sync(m_ctx), m_ctx);
    }

    // One MainLoop iteration (HardwareRenderer.cpp:977-1010). Returns false when the iteration was
    // skipped because the window is minimised.
    bool Step(const FrameInput& in)
    {
        if (in.minimized) return false;
        m_cameraController.Update(this, in);
        if (in.windowWidth > 0 && in.windowHeight > 0 && (in.windowWidth != m_width || in.windowHeight != m_height))
            RecreateSwapchain(in.windowWidth, in.windowHeight);
        RenderFrame();
        m_pushConstants.frame++;
        if (m_bDoRender) ProduceRender();
        return true;
    }

    // ProduceRender (HardwareRenderer.cpp:1030-1061): frames m_pushConstants.frame .. m_iRenderFrames-1,
    // then the PNG, then the wall-clock line. The file goes to m_renderPath (the reference builds
    // Renders\render_<date>.png; the caller of this class chooses).
    void ProduceRender()
    {
        const auto start = std::chrono::system_clock::now();
        for (int i = m_pushConstants.frame; i < m_iRenderFrames; ++i) {
            RenderFrame();
            m_pushConstants.frame++;
        }
        WriteDrawImageToFile();
        m_bDoRender = false;
        m_lastRenderSeconds = std::chrono::duration<double>(std::chrono::system_clock::now() - start).count();
        if (m_bPrintRenderTime) std::printf("Render took %g seconds.\n", m_lastRenderSeconds);
    }

    // WriteDrawImageToFile (HardwareRenderer.cpp:1063-1144)
    void WriteDrawImageToFile()
    {
        if (m_renderPath.empty()) return;
        check(rt_save_png(m_ctx, m_renderPath.c_str()), m_ctx);
        if (m_bPrintRenderTime) std::printf("Wrote render to %s\n", m_renderPath.c_str());
    }

    // The draw image as the swapchain blit sees it (HardwareRenderer.cpp:928): W x H RGBA f32, row 0 on top.
    const std::vector<float>& ReadDrawImage()
    {
        m_hostImage.resize((size_t)m_width * (size_t)m_height * 4);
        check(rt_read_image(m_ctx, RT_IMAGE_DRAW, m_hostImage.data(), m_hostImage.size() * sizeof(float)), m_ctx);
        return m_hostImage;
    }

    int GetWidth() const { return m_width; }
    int GetHeight() const { return m_height; }
    rt_ctx* GetContext() { return m_ctx; }
    rt_scene* GetScene() { return m_scene; }
    double GetLastRenderSeconds() const { return m_lastRenderSeconds; }

    std::string m_renderPath = "render.png";
    bool m_bPrintRenderTime = true;

private:
    // RefreshAccumulation (HardwareRenderer.cpp:840-848)
    void RefreshAccumulation()
    {
        check(rt_clear_accumulation(m_ctx), m_ctx);
        m_pushConstants.frame = 0;
        m_bRefreshAccumulation = false;
    }

    void check(int rc, rt_ctx* ctx) const
    {
        if (rc < 0) throw Error(rc, rt_last_error(ctx));
    }
    void checkScene(int rc) const
    {
        if (rc < 0) throw Error(rc, rt_scene_last_error(m_scene));
    }

    rt_ctx* m_ctx = nullptr;
    rt_scene* m_scene = nullptr;
    bool m_bInitialized = false;
    int m_width = 0, m_height = 0, m_parentCount = 0;
    double m_lastRenderSeconds = 0.0;
    std::vector<float> m_hostImage;
};

// CameraController::Update (CameraController.cpp:11-66). The six movement keys are one table of
// (held?, direction, sign); each held key displaces the camera by speed * direction * dt, applied key by
// key in the reference's order W, S, A, D, Space, X (binary32, so the order is part of the result).
inline void CameraController::Update(HardwareRenderer* renderer, const FrameInput& in)
{
    UpdateCameraRotation(renderer, in);
    if (in.escapePressed) m_bLockedMouse = !m_bLockedMouse;
    if (!m_bLockedMouse) return;

    CameraSettings& cam = renderer->m_camera;
    const float dt = (float)in.deltaT;
    const Vec3 up(0, 1, 0);
    const Vec3 ahead = cam.cameraLookDirection, sideways = cross(cam.cameraLookDirection, up);
    const struct { bool held; Vec3 direction; bool forward; } keys[6] = {
        { in.keyW, ahead, true }, { in.keyS, ahead, false }, { in.keyA, sideways, false },
        { in.keyD, sideways, true }, { in.keySpace, up, true }, { in.keyX, up, false } };
    int pressed = 0;
    for (const auto& k : keys) {
        if (!k.held) continue;
        const Vec3 displacement = m_fMoveSpeed * k.direction * dt;
        if (k.forward) cam.cameraPosition += displacement; else cam.cameraPosition -= displacement;
        ++pressed;
    }
    if (pressed > 0) renderer->m_bRefreshAccumulation = true;
}

// CameraController::UpdateCameraRotation (CameraController.cpp:68-101): yaw / pitch from the mouse
// deltas scaled by sensitivity * dt, pitch kept within +-89 degrees, look direction from the two
// angles; a changed direction asks for a fresh accumulation. While the mouse is free nothing turns
// and the next locked frame counts as the first.
inline void CameraController::UpdateCameraRotation(HardwareRenderer* renderer, const FrameInput& in)
{
    if (!m_bLockedMouse) { m_bFirstMouse = true; return; }
    Vec3& look = renderer->m_camera.cameraLookDirection;
    const Vec3 before = look;
    const float scale = m_fSensitivity * (float)in.deltaT;
    yaw += in.mouseDeltaX * scale;
    pitch += -in.mouseDeltaY * scale;
    pitch = std::fmin(std::fmax(pitch, -89.0f), 89.0f);
    const float cp = std::cos(radians(pitch));
    look = normalize(Vec3(std::cos(radians(yaw)) * cp, std::sin(radians(pitch)), std::sin(radians(yaw)) * cp));
    if (before != look) renderer->m_bRefreshAccumulation = true;
}

} // namespace rtamd

#endif // RT_RENDERER_HPP
