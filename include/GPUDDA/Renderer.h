// GPUDDA/Renderer.h -- the reference's VoxelRT/Renderer.cuh API surface over the C ABI (include/vxrt.h).
#pragma once

#include "VolumeRaytracer.h"

#include <cstdint>

namespace GPUDDA {
namespace Graphics {

struct BGRA8888 {  // Renderer.cuh:29-31; same byte order as SDLRenderer.h:8-11 PixelData
    uint8_t b, g, r, a;
};

struct Environment {  // Renderer.cuh:33-37
    float3 LightDirection;
    float3 LightColor;
    float3 AmbientColor;
};

// run-time forms of the reference's compile-time switches; defaults = the checked-in values
// (`#define DEBUG_VIEW`, ENABLE_CHECKERBOARD_RENDER = true, shadow call commented out, samples = 0:
// Renderer.cu:4-5,102,123)
struct RenderSwitches {
    bool DebugView = true;
    bool Checkerboard = true;
    bool Ortho = false;
    bool ShadowRay = false;
    int BounceSamples = 0;
    bool BounceAllHits = false;
    int BounceDepth = 1;  // 2: second bounce, an extension beyond the reference (include/vxrt.h)
};
void SetRenderSwitches(const RenderSwitches& s);

void GetDirections(float3 eularAngles, float3* forwad, float3* up, float3* right);
void SetEnvironment(const Environment& env);
void SetFOV(float fov);
void SetOrthoWindowSize(float2 windowSize);
// d_screen_texture: device BGRA8 buffer of screen_width*screen_height pixels (hipMalloc'ed by the caller,
// VoxelApp/main.cu:68).  Synchronous like the reference (Renderer.cu:327).
void RenderScreen(VoxelRaytracer3D* rt, uint32_t screen_width, uint32_t screen_height, void* d_screen_texture, float3 origin,
                  float3 camera_fwd, float3 camera_up, float3 camera_right);

// This build's addition: several RenderScreen views in one launch (vxrt_render_views, include/vxrt.h) -- the GPU does
// not idle at the end of every frame, about 1.35x the rays/s at 1080p.  Each view is the frame RenderScreen would have
// produced in its place (the views take successive FrameNumbers).
struct ScreenView {
    void* d_screen_texture;
    float3 origin, camera_fwd, camera_up, camera_right;
};
void RenderScreens(VoxelRaytracer3D* rt, uint32_t screen_width, uint32_t screen_height, const ScreenView* views, uint32_t count);

}  // namespace Graphics
}  // namespace GPUDDA
