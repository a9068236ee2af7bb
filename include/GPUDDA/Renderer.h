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

// This build's addition for interactive callers (their next camera depends on input, so they cannot pre-batch views):
// RenderScreen WITHOUT the closing device synchronisation.  Frames alternate between two internal streams, so up to two
// frames are in flight and the wavefronts of frame k+1 fill the SIMD slots that the last, longest rays of frame k leave
// (a single 1080p frame keeps the GPU full for only ~70 % of its launch; measured 3.3 -> 4.6 Grays/s).  The frame is exactly
// the frame RenderScreen produces.  Give consecutive frames different device buffers; WaitFrame(ticket) returns once that
// frame is complete (then the device->host copy of VoxelApp/main.cu:167 may read it).  The call never blocks: frame t queues
// behind frame t-2 on their common stream, so at most two frames execute at once however far ahead the caller runs; a caller
// paces itself with WaitFrame (which waits for the newest frame launched on the ticket's stream, a later one of the same
// parity included).
typedef uint64_t FrameTicket;
FrameTicket RenderScreenAsync(VoxelRaytracer3D* rt, uint32_t screen_width, uint32_t screen_height, void* d_screen_texture,
                              float3 origin, float3 camera_fwd, float3 camera_up, float3 camera_right);
void WaitFrame(FrameTicket ticket);
// the stream a ticket's frame was rendered on (a hipStream_t), e.g. to enqueue the device->host copy behind it
void* FrameStream(FrameTicket ticket);

}  // namespace Graphics
}  // namespace GPUDDA
