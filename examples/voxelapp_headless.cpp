// voxelapp_headless.cpp -- the call sequence of the reference's VoxelApp/main.cu against the GPUDDA facade,
// without SDL2: build the world, build the brickmap, upload, set the environment, then per frame
// GetDirections + RenderScreen + device->host copy of the framebuffer (main.cu:18-69,165-167).  The fly camera
// is a scripted path instead of keyboard/mouse input; the last frame is written as raw BGRA and as a PPM.
//
//   voxelapp_headless [world_edge=256] [frames=2] [out_prefix=frame] [width=320] [height=180] [shaded=0]
//                     [camera_path_file] [dump_every_frame=0] [views_per_launch=1] [frames_in_flight=1] [device_world=XxYxZ]
//
// shaded: 0 = the checked-in debug view, 1 = shaded with shadow + 1 bounce sample, checkerboard on, 2 = the same with the
// checkerboard off (whole frames).  frames_in_flight=2 renders through Graphics::RenderScreenAsync / WaitFrame: frame k+1 is
// launched before frame k is copied to the host, on two alternating device buffers (use shaded=2: under checkerboard a
// frame needs its predecessor's buffer).  device_world=8192x512x8192 builds the brickmap on the device in one step
// (VoxelRaytracer3D::BuildProceduralWorld) instead of CreateVoxels + GenerateLowresVoxelBuffer, whose dense bit array
// the reference cannot index beyond 2^32 voxels; `world_edge` then only scales the default camera.  A '-' skips
// camera_path_file.  The run ends with the rays traced and the Mrays/s over the frame loop.

// camera_path_file replaces the fixed camera: one frame per line, "x y z eulerX eulerY eulerZ" (voxels, radians;
// '#' starts a comment); `frames` is then the number of lines.  With dump_every_frame=1 each frame is also written
// as <out_prefix>_NNNN.ppm (under checkerboard rendering a frame keeps the other half of the previous one).
// views_per_launch > 1 renders that many poses per launch through Graphics::RenderScreens (no checkerboard then).
#include "../include/GPUDDA/Renderer.h"
#include "../include/GPUDDA/VoxelWorldBuilder.h"
#include "../include/vxrt.h"

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

using namespace GPUDDA;
using namespace GPUDDA::Graphics;

int main(int argc, char** argv)
{
    const unsigned edge = argc > 1 ? (unsigned)atoi(argv[1]) : 256u;
    const int frames = argc > 2 ? atoi(argv[2]) : 2;
    const std::string prefix = argc > 3 ? argv[3] : "frame";
    const uint32_t width = argc > 4 ? (uint32_t)atoi(argv[4]) : 320u, height = argc > 5 ? (uint32_t)atoi(argv[5]) : 180u;
    const bool shaded = argc > 6 && atoi(argv[6]) != 0;
    const int shade_mode = argc > 6 ? atoi(argv[6]) : 0;
    const std::string path_file = (argc > 7 && std::string(argv[7]) != "-") ? argv[7] : "";
    const bool dump_all = argc > 8 && atoi(argv[8]) != 0;
    const int batch = argc > 9 ? atoi(argv[9]) : 1;
    const int in_flight = argc > 10 ? atoi(argv[10]) : 1;
    unsigned wx = 0, wy = 0, wz = 0;
    if (argc > 11 && std::sscanf(argv[11], "%ux%ux%u", &wx, &wy, &wz) != 3) {
        std::cerr << "device_world must look like 8192x512x8192" << std::endl;
        return 2;
    }

    struct Pose {
        float3 pos, euler;
    };
    std::vector<Pose> path;
    if (!path_file.empty()) {
        std::ifstream in(path_file);
        if (!in) {
            std::cerr << "cannot open camera path " << path_file << std::endl;
            return 2;
        }
        std::string line;
        while (std::getline(in, line)) {
            const size_t hash = line.find('#');
            if (hash != std::string::npos)
                line.resize(hash);
            float v[6];
            if (std::sscanf(line.c_str(), "%f %f %f %f %f %f", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) == 6)
                path.push_back({make_float3(v[0], v[1], v[2]), make_float3(v[3], v[4], v[5])});
        }
        if (path.empty()) {
            std::cerr << "camera path " << path_file << " holds no poses" << std::endl;
            return 2;
        }
    }

    int factor = 32;
    VoxelRaytracer3D* raytracer = new VoxelRaytracer3D(1);
    auto t0 = std::chrono::high_resolution_clock::now();
    if (wx) {
        raytracer->BuildProceduralWorld(make_uint3(wx, wy, wz), factor);
        auto t1 = std::chrono::high_resolution_clock::now();
        std::cout << "World built on the device: " << std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count() << "ms" << std::endl;
    } else {
        auto buffer = CreateVoxels(make_uint3(edge, edge, edge));
        auto t1 = std::chrono::high_resolution_clock::now();
        std::cout << "Voxel generation time: " << std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count() << "ms" << std::endl;

        auto buffers = GenerateLowresVoxelBuffer(buffer, factor);
        auto t2 = std::chrono::high_resolution_clock::now();
        std::cout << "Buffer generation time: " << std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count() << "ms" << std::endl;
        delete[] buffer.grid.Raw();

        auto low_res_buffer = std::get<0>(buffers);
        auto low_res_grid_data = std::get<1>(buffers);
        auto bounds = std::get<2>(buffers);
        auto count = (size_t)low_res_buffer.dimensions[0] * low_res_buffer.dimensions[1] * low_res_buffer.dimensions[2];
        raytracer->UploadVoxelBuffer(low_res_buffer);
        raytracer->UploadVoxelBufferDatas(low_res_grid_data, count);
        raytracer->UploadVoxelBufferDataBounds(bounds, count);
        raytracer->SetFactor(factor);
    }

    float3 cam_pos = wx ? make_float3(wx * 0.5f, wy * 0.9f, wz * 0.5f) : make_float3(edge * 0.25f, edge * 0.9f, edge * 0.25f);
    float3 cam_up, cam_right, cam_forward;
    float3 cam_eular = make_float3(-0.45f, 0.7f, 0.0f);

    Environment env;
    const float inv = 1.0f / std::sqrt(3.0f);
    env.LightDirection = make_float3(1.0f * inv, 1.0f * inv, 1.0f * inv);
    env.LightColor = make_float3(2, 2, 2);
    env.AmbientColor = make_float3(0.5f, 0.5f, 0.5f);
    SetEnvironment(env);
    SetFOV(90);
    SetOrthoWindowSize(make_float2(10, 10));
    if (shaded) {  // the README screenshots' configuration; default = the checked-in debug view
        RenderSwitches s;
        s.DebugView = false;
        s.ShadowRay = true;
        s.BounceSamples = 1;
        s.Checkerboard = shade_mode != 2;
        SetRenderSwitches(s);
    }

    void* d_pixels = nullptr;
    if (hipMalloc(&d_pixels, (size_t)width * height * sizeof(BGRA8888)) != hipSuccess)
        return 1;
    (void)hipMemset(d_pixels, 255, (size_t)width * height * sizeof(BGRA8888));
    std::vector<BGRA8888> pixels((size_t)width * height);

    auto write_ppm = [&](const std::string& name) {
        std::ofstream ppm(name, std::ios::binary);
        ppm << "P6\n" << width << " " << height << "\n255\n";
        for (const auto& p : pixels) {
            const char rgb[3] = {(char)p.r, (char)p.g, (char)p.b};
            ppm.write(rgb, 3);
        }
    };

    double avgFrameTime = 0.0;
    const int nframes = path.empty() ? frames : (int)path.size();
    // buffers of the two-frames-in-flight mode (allocated before the timed frame loop, like d_pixels)
    void* d_ring[3] = {d_pixels, nullptr, nullptr};
    BGRA8888* h_ring[3] = {nullptr, nullptr, nullptr};
    hipEvent_t rendered[3], copied[3];
    hipStream_t copy_stream;
    const size_t bytes = (size_t)width * height * sizeof(BGRA8888);
    if (hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess)
        return 1;
    for (int k = 0; k < 3 && in_flight >= 2 && batch <= 1; ++k) {
        if ((k > 0 && hipMalloc(&d_ring[k], bytes) != hipSuccess) || hipHostMalloc((void**)&h_ring[k], bytes) != hipSuccess ||
            hipEventCreateWithFlags(&rendered[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&copied[k], hipEventDisableTiming) != hipSuccess)
            return 1;
        (void)hipMemset(d_ring[k], 255, bytes);
    }
    vxrt_frame_stats rays_before{};
    (void)vxrt_frame_stats_get(raytracer->Context(), &rays_before);  // start the ray counters of the frame loop from zero
    const auto loop0 = std::chrono::high_resolution_clock::now();
    double dump_ms = 0.0;
    auto timed_dump = [&](int i) {
        if (!dump_all)
            return;
        const auto d0 = std::chrono::high_resolution_clock::now();
        char name[32];
        std::snprintf(name, sizeof(name), "_%04d.ppm", i);
        write_ppm(prefix + name);
        dump_ms += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - d0).count() / 1000.0;
    };
    if (batch <= 1 && in_flight >= 2) {
        // Two frames in flight (Graphics::RenderScreenAsync): launch frame i, queue its device->host copy behind it on the
        // frame's stream, THEN take delivery of frame i-2 -- the per-frame sequence of VoxelApp/main.cu:165-167 with the
        // host two frames ahead of the pixels.  The camera of frame i may depend on anything the host knows when it launches
        // it; nothing is batched ahead.  Three device buffers and three pinned host buffers rotate.
        for (int i = 0; i < nframes + 2; ++i) {
            auto f0 = std::chrono::high_resolution_clock::now();
            if (i < nframes) {
                if (!path.empty()) {
                    cam_pos = path[(size_t)i].pos;
                    cam_eular = path[(size_t)i].euler;
                }
                GetDirections(cam_eular, &cam_forward, &cam_up, &cam_right);
                const FrameTicket t = RenderScreenAsync(raytracer, width, height, d_ring[i % 3], cam_pos, cam_forward, cam_up, cam_right);
                // the copy runs on its own stream behind the frame, so the render streams never wait for the copy engine
                (void)hipEventRecord(rendered[i % 3], (hipStream_t)FrameStream(t));
                (void)hipStreamWaitEvent(copy_stream, rendered[i % 3], 0);
                (void)hipMemcpyAsync(h_ring[i % 3], d_ring[i % 3], bytes, hipMemcpyDeviceToHost, copy_stream);
                (void)hipEventRecord(copied[i % 3], copy_stream);
            }
            if (i >= 2) {  // frame i-2 is delivered (its device buffer is the one frame i+1 will render into)
                (void)hipEventSynchronize(copied[(i - 2) % 3]);
                if (dump_all || i - 2 == nframes - 1)
                    std::memcpy(pixels.data(), h_ring[(i - 2) % 3], bytes);
                timed_dump(i - 2);
            }
            auto f1 = std::chrono::high_resolution_clock::now();
            double td = std::chrono::duration_cast<std::chrono::microseconds>(f1 - f0).count() / 1000.0;
            avgFrameTime = i == 0 ? td : avgFrameTime * 0.9 + td * 0.1;
        }
    }
    if (batch > 1) {
        // several poses per launch (Graphics::RenderScreens): every view has its own framebuffer, so this mode runs
        // without the checkerboard's frame-to-frame history
        RenderSwitches s;
        s.DebugView = !shaded;
        s.Checkerboard = false;
        s.ShadowRay = shaded;
        s.BounceSamples = shaded ? 1 : 0;
        SetRenderSwitches(s);
        std::vector<void*> d_views((size_t)batch, nullptr);
        for (auto& p : d_views)
            if (hipMalloc(&p, (size_t)width * height * sizeof(BGRA8888)) != hipSuccess)
                return 1;
        for (int first = 0; first < nframes; first += batch) {
            const int n = nframes - first < batch ? nframes - first : batch;
            std::vector<ScreenView> views((size_t)n);
            auto f0 = std::chrono::high_resolution_clock::now();
            for (int j = 0; j < n; ++j) {
                if (!path.empty()) {
                    cam_pos = path[(size_t)(first + j)].pos;
                    cam_eular = path[(size_t)(first + j)].euler;
                }
                GetDirections(cam_eular, &cam_forward, &cam_up, &cam_right);
                views[(size_t)j] = ScreenView{d_views[(size_t)j], cam_pos, cam_forward, cam_up, cam_right};
            }
            RenderScreens(raytracer, width, height, views.data(), (uint32_t)n);
            auto f1 = std::chrono::high_resolution_clock::now();
            double td = std::chrono::duration_cast<std::chrono::microseconds>(f1 - f0).count() / 1000.0 / n;
            avgFrameTime = first == 0 ? td : avgFrameTime * 0.9 + td * 0.1;
            for (int j = 0; j < n; ++j) {
                (void)hipMemcpy(pixels.data(), d_views[(size_t)j], pixels.size() * sizeof(BGRA8888), hipMemcpyDeviceToHost);
                timed_dump(first + j);
            }
        }
        for (auto p : d_views)
            (void)hipFree(p);
    }
    for (int i = 0; batch <= 1 && in_flight < 2 && i < nframes; ++i) {
        if (!path.empty()) {
            cam_pos = path[(size_t)i].pos;
            cam_eular = path[(size_t)i].euler;
        }
        auto f0 = std::chrono::high_resolution_clock::now();
        GetDirections(cam_eular, &cam_forward, &cam_up, &cam_right);
        RenderScreen(raytracer, width, height, d_pixels, cam_pos, cam_forward, cam_up, cam_right);
        (void)hipMemcpy(pixels.data(), d_pixels, pixels.size() * sizeof(BGRA8888), hipMemcpyDeviceToHost);
        auto f1 = std::chrono::high_resolution_clock::now();
        double td = std::chrono::duration_cast<std::chrono::microseconds>(f1 - f0).count() / 1000.0;
        avgFrameTime = i == 0 ? td : avgFrameTime * 0.9 + td * 0.1;
        timed_dump(i);
    }
    std::cout << "Avg FPS: " << 1000.0 / avgFrameTime << std::endl;
    {
        vxrt_frame_stats st{};
        (void)vxrt_frame_stats_get(raytracer->Context(), &st);  // synchronises the device
        const double loop_ms = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - loop0).count() / 1000.0 - dump_ms;
        const unsigned long long rays = st.primary_rays + st.shadow_rays + st.bounce_rays;
        std::printf("Frame loop: %d frames, %llu rays in %.3f ms (device->host copy of every frame included) = %.1f Mrays/s\n", nframes, rays,
                    loop_ms, loop_ms > 0 ? rays / loop_ms / 1e3 : 0.0);
    }

    std::ofstream raw(prefix + ".bgra", std::ios::binary);
    raw.write(reinterpret_cast<const char*>(pixels.data()), (std::streamsize)(pixels.size() * sizeof(BGRA8888)));
    write_ppm(prefix + ".ppm");
    // a few batch queries through VoxelRaytracer3D::Raytrace
    std::vector<float3> o(4, cam_pos), d = {make_float3(0, -1, 0), make_float3(1, -1, 0), make_float3(0, 1, 0), cam_forward};
    auto res = raytracer->Raytrace(o, d);
    for (int i = 0; i < 4; ++i)
        std::printf("ray %d valid=%d steps=%d voxel=%d\n", i, (int)res.valid[i], res.steps[i], res.voxelIndex[i]);
    for (int k = 0; k < 3 && in_flight >= 2 && batch <= 1; ++k) {
        if (k > 0)
            (void)hipFree(d_ring[k]);
        (void)hipHostFree(h_ring[k]);
        (void)hipEventDestroy(rendered[k]);
        (void)hipEventDestroy(copied[k]);
    }
    (void)hipStreamDestroy(copy_stream);
    (void)hipFree(d_pixels);
    delete raytracer;
    return 0;
}
