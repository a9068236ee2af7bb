// gpudda_facade.cpp -- namespace GPUDDA (the reference's VoxelRT C++ API) implemented on the C ABI of libvxrt.so.
// Host code only: every pixel and ray comes from the HIP kernels behind vxrt_render / vxrt_trace_batch.
#include "../../include/GPUDDA/Renderer.h"
#include "../../include/GPUDDA/VoxelWorldBuilder.h"
#include "../../include/vxrt.h"

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>

namespace {

[[noreturn]] void die(const char* what)
{
    // the reference prints and exits on device errors (CUDA_SAFE_CALL, Renderer.cuh:15-23)
    std::cerr << "GPUDDA error: " << what << ": " << vxrt_last_error() << std::endl;
    std::exit(EXIT_FAILURE);
}
void ok(int rc, const char* what)
{
    if (rc != VXRT_OK)
        die(what);
}

// process-wide renderer state, like hFrameInfo / g_env in the reference (Renderer.cu:24-25,89)
GPUDDA::Graphics::Environment g_env{};
float g_fov = 90.0f;
float2 g_ortho = {10.0f, 10.0f};
GPUDDA::Graphics::RenderSwitches g_switches{};

}  // namespace

namespace GPUDDA {

uint32_t GetSampleIndex(uint32_t x, uint32_t y, uint32_t z, uint32_t width, uint32_t height)
{
    const uint32_t tiles_w = width >> 3, tiles_h = height >> 3;
    const uint32_t tile = (x >> 3) + (y >> 3) * tiles_w + (z >> 3) * tiles_w * tiles_h;
    return tile * 512u + (x & 7u) + ((y & 7u) << 3) + ((z & 7u) << 6);
}

void GetPositionFromSampleIndex(uint32_t index, uint32_t width, uint32_t height, uint32_t& x, uint32_t& y, uint32_t& z)
{
    const uint32_t tiles_w = width >> 3, tiles_h = height >> 3;
    const uint32_t tile = index >> 9, in = index & 511u;
    x = ((tile % tiles_w) << 3) + (in & 7u);
    y = (((tile / tiles_w) % tiles_h) << 3) + ((in >> 3) & 7u);
    z = ((tile / (tiles_w * tiles_h)) << 3) + (in >> 6);
}

// ---- BitRef / BitArray ---------------------------------------------------------------------------

BitRef::operator bool() const { return (*byte >> index) & 1u; }

BitRef& BitRef::operator=(bool value)
{
    auto* word = reinterpret_cast<std::atomic<uint32_t>*>(byte);
    const uint32_t mask = 1u << (index & 31);
    if (value)
        word->fetch_or(mask, std::memory_order_relaxed);
    else
        word->fetch_and(~mask, std::memory_order_relaxed);
    return *this;
}

BitArray::BitArray() = default;

static uint32_t* alloc_words(size_t nbits, bool on_device)
{
    const size_t bytes = (nbits + 31) / 32 * sizeof(uint32_t);
    if (!on_device)
        return new uint32_t[(nbits + 31) / 32];
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) {
        std::cerr << "GPUDDA error: hipMalloc failed" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    return static_cast<uint32_t*>(p);
}

BitArray::BitArray(size_t num_bits, bool isGPU) : size(num_bits), data(alloc_words(num_bits, isGPU)) {}

BitArray::BitArray(const BitArray& other, bool isGPU) : size(other.size), data(alloc_words(other.size, isGPU))
{
    const size_t bytes = (size + 31) / 32 * sizeof(uint32_t);
    if (isGPU)
        (void)hipMemcpy(data, other.data, bytes, hipMemcpyHostToDevice);
    else
        std::memcpy(data, other.data, bytes);
}

bool BitArray::operator[](size_t index) const
{
    if (index >= size)
        return false;
    return (data[index / 32] >> (index % 32)) & 1u;
}
BitRef BitArray::operator[](size_t index) { return BitRef{&data[index / 32], index % 32}; }
uint32_t* BitArray::Raw() { return data; }
const uint32_t* BitArray::Raw() const { return data; }
size_t BitArray::BitSize() const { return size; }
size_t BitArray::ByteSize() const { return (size + 31) / 32 * sizeof(uint32_t); }
std::ostream& operator<<(std::ostream& os, const BitArray& bits)
{
    for (size_t i = 0; i < bits.BitSize(); ++i)
        os << bits[i];
    return os;
}

// ---- VoxelRaytracer3D ------------------------------------------------------------------------------

static int g_live_raytracers = 0;  // RenderScreenAsync's streams live as long as some raytracer does (see FrameSlots)

VoxelRaytracer3D::VoxelRaytracer3D(size_t /*count*/)
{
    int device = 0;
    (void)hipGetDevice(&device);
    ok(vxrt_create(device, &ctx), "vxrt_create");
    ++g_live_raytracers;
}

namespace Graphics {
static void release_frame_slots();
}

VoxelRaytracer3D::~VoxelRaytracer3D()
{
    Free();
    if (--g_live_raytracers == 0)
        Graphics::release_frame_slots();  // while the HIP runtime is certainly still up: never from a static destructor
}

void VoxelRaytracer3D::Free()
{
    if (ctx)
        vxrt_destroy(ctx);
    ctx = nullptr;
}

void VoxelRaytracer3D::SetFactor(int f)
{
    factor = f;
    dirty = true;
}

void VoxelRaytracer3D::UploadVoxelBuffer(const VoxelBuffer3D& buff)
{
    for (int a = 0; a < 3; ++a)
        cdims[a] = buff.dimensions[a];
    const size_t n = (size_t)cdims[0] * cdims[1] * cdims[2];
    coarse_bits.assign(buff.grid.Raw(), buff.grid.Raw() + (n + 31) / 32);
    have_coarse = true;
    dirty = true;
}

void VoxelRaytracer3D::UploadVoxelBufferDatas(VoxelBuffer3D* buff, size_t count)
{
    // descriptors with their own bit arrays -> slot table + one pool
    brick_slot.assign(count, VXRT_EMPTY_SLOT);
    pool.clear();
    uint32_t next = 0;
    for (size_t i = 0; i < count; ++i) {
        const size_t f = buff[i].dimensions[0];
        if (f == 0)
            continue;  // empty brick: the reference frees its bits (VolumeRaytracer.cuh:461-464)
        const size_t words = f * f * f / 32;
        pool.insert(pool.end(), buff[i].grid.Raw(), buff[i].grid.Raw() + words);
        brick_slot[i] = next++;
    }
    have_bricks = true;
    dirty = true;
}

void VoxelRaytracer3D::UploadVoxelBufferDataBounds(Bounds3Df* b, size_t count)
{
    bounds.resize(count * 6);
    for (size_t i = 0; i < count; ++i) {
        float* o = &bounds[i * 6];
        o[0] = b[i].min.x; o[1] = b[i].min.y; o[2] = b[i].min.z;
        o[3] = b[i].max.x; o[4] = b[i].max.y; o[5] = b[i].max.z;
    }
    have_bounds = true;
    dirty = true;
}

void VoxelRaytracer3D::Flush()
{
    if (!dirty || !(have_coarse && have_bricks && have_bounds))
        return;
    vxrt_world_desc d{};
    d.struct_size = sizeof(d);
    d.factor = factor;
    for (int a = 0; a < 3; ++a)
        d.cdims[a] = cdims[a];
    d.nslots = pool.size() / ((size_t)factor * factor * factor / 32);
    d.coarse_bits = coarse_bits.data();
    d.brick_slot = brick_slot.data();
    d.bounds = bounds.data();
    d.pool = pool.data();
    ok(vxrt_upload_world(ctx, &d), "vxrt_upload_world");
    dirty = false;
}

vxrt_ctx* VoxelRaytracer3D::Context()
{
    Flush();
    return ctx;
}

void VoxelRaytracer3D::BuildProceduralWorld(uint3 size, int f, int generator)
{
    ok(vxrt_build_world_procedural(ctx, generator, (int)size.x, (int)size.y, (int)size.z, f), "vxrt_build_world_procedural");
    factor = f;
    dirty = false;
}

RayTraceResults<float3> VoxelRaytracer3D::Raytrace(std::vector<float3> origin, std::vector<float3> ray)
{
    const size_t n = origin.size();
    RayTraceResults<float3> r(n);
    if (n == 0)
        return r;
    static_assert(sizeof(float3) == 12, "float3 must be three packed floats");
    std::vector<uint8_t> hit(n);
    std::vector<int64_t> vox(n);
    const auto t0 = std::chrono::high_resolution_clock::now();
    ok(vxrt_trace_batch_host(Context(), &origin[0].x, &ray[0].x, n, &r.hitPoint[0].x, &r.normal[0].x, r.steps.get(),
                             hit.data(), vox.data(), nullptr),
       "vxrt_trace_batch_host");
    const auto t1 = std::chrono::high_resolution_clock::now();
    std::cout << "Raytracing time: " << std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count() / 1000.0f
              << " ms" << std::endl;  // the reference reports the same line (VolumeRaytracer.cu:595)
    for (size_t i = 0; i < n; ++i) {
        const float3 p = r.hitPoint[i];
        r.valid[i] = p.x != FLT_INF && p.y != FLT_INF && p.z != FLT_INF;
        const float dx = origin[i].x - p.x, dy = origin[i].y - p.y, dz = origin[i].z - p.z;
        r.distance[i] = std::sqrt(dx * dx + dy * dy + dz * dz);
        r.voxelIndex[i] = (vox[i] >= 0 && vox[i] <= 0x7FFFFFFF) ? (int)vox[i] : -1;
    }
    return r;
}

// ---- GenerateLowresVoxelBuffer ---------------------------------------------------------------------

std::tuple<VoxelBuffer3D, VoxelBuffer3D*, Bounds3Df*> GenerateLowresVoxelBuffer(const VoxelBuffer3D& src, int factor)
{
    const size_t f = (size_t)factor;
    const size_t cols = src.dimensions[0] / f, rows = src.dimensions[1] / f, slices = src.dimensions[2] / f;
    const size_t cells = cols * rows * slices;
    auto* bricks = new VoxelBuffer3D[cells]{};
    auto* extents = new Bounds3Df[cells]{};
    std::vector<uint8_t> occupied(cells, 0);
    const uint32_t* dense = src.grid.Raw();
    const uint64_t X = src.dimensions[0], Y = src.dimensions[1];
    auto dense_index = [&](uint64_t x, uint64_t y, uint64_t z) {  // 64-bit: worlds past 2^32 bits do not wrap
        return ((x >> 3) + (y >> 3) * (X >> 3) + (z >> 3) * (X >> 3) * (Y >> 3)) * 512u + (x & 7u) + ((y & 7u) << 3) + ((z & 7u) << 6);
    };
    auto work = [&](size_t begin, size_t end) {
        for (size_t cell = begin; cell < end; ++cell) {
            uint32_t bx, by, bz;
            GetPositionFromSampleIndex((uint32_t)cell, (uint32_t)cols, (uint32_t)rows, bx, by, bz);
            VoxelBuffer3D& out = bricks[cell];
            out.grid = BitArray(f * f * f, false);
            std::memset(out.grid.Raw(), 0, out.grid.ByteSize());
            int mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
            bool any = false;
            for (size_t dz = 0; dz < f; ++dz)
                for (size_t dy = 0; dy < f; ++dy)
                    for (size_t dx = 0; dx < f; ++dx) {
                        const uint64_t hi = dense_index(dx + f * bx, dy + f * by, dz + f * bz);
                        if (!((dense[hi >> 5] >> (hi & 31)) & 1u))
                            continue;
                        const uint32_t lo = GetSampleIndex((uint32_t)dx, (uint32_t)dy, (uint32_t)dz, (uint32_t)f, (uint32_t)f);
                        out.grid.Raw()[lo >> 5] |= 1u << (lo & 31);
                        any = true;
                        mn[0] = std::min(mn[0], (int)dx); mn[1] = std::min(mn[1], (int)dy); mn[2] = std::min(mn[2], (int)dz);
                        mx[0] = std::max(mx[0], (int)dx); mx[1] = std::max(mx[1], (int)dy); mx[2] = std::max(mx[2], (int)dz);
                    }
            if (any) {
                out.dimensions[0] = out.dimensions[1] = out.dimensions[2] = (uint16_t)f;
            } else {
                mn[0] = mn[1] = mn[2] = 0;
                mx[0] = mx[1] = mx[2] = -1;
                delete[] out.grid.Raw();
                out.grid = BitArray();
            }
            extents[cell].min = make_float3((float)mn[0], (float)mn[1], (float)mn[2]);
            extents[cell].max = make_float3((float)mx[0], (float)mx[1], (float)mx[2]);
            occupied[cell] = any;
        }
    };
    const size_t nthreads = std::max<size_t>(1, std::thread::hardware_concurrency());
    std::vector<std::thread> pool_threads;
    for (size_t t = 0; t < nthreads; ++t)
        pool_threads.emplace_back(work, cells * t / nthreads, cells * (t + 1) / nthreads);
    for (auto& th : pool_threads)
        th.join();

    VoxelBuffer3D coarse;
    coarse.grid = BitArray(cells, false);
    std::memset(coarse.grid.Raw(), 0, coarse.grid.ByteSize());
    for (size_t i = 0; i < cells; ++i)
        if (occupied[i])
            coarse.grid.Raw()[i >> 5] |= 1u << (i & 31);
    coarse.dimensions[0] = (uint16_t)cols;
    coarse.dimensions[1] = (uint16_t)rows;
    coarse.dimensions[2] = (uint16_t)slices;
    return std::make_tuple(coarse, bricks, extents);
}

// ---- Graphics --------------------------------------------------------------------------------------

namespace Graphics {

void GetDirections(float3 e, float3* forwad, float3* up, float3* right)
{
    const float in[3] = {e.x, e.y, e.z};
    float f[3], u[3], r[3];
    vxrt_get_directions(in, f, u, r);
    *forwad = make_float3(f[0], f[1], f[2]);
    *up = make_float3(u[0], u[1], u[2]);
    *right = make_float3(r[0], r[1], r[2]);
}

void SetEnvironment(const Environment& env) { g_env = env; }
void SetFOV(float fov) { g_fov = fov; }
void SetOrthoWindowSize(float2 s) { g_ortho = s; }
void SetRenderSwitches(const RenderSwitches& s) { g_switches = s; }

// globals -> context state and launch flags (what RenderScreen copies to dFrameInfo / reads from g_env)
static vxrt_ctx* prepare_launch(VoxelRaytracer3D* rt, vxrt_render_flags& fl)
{
    vxrt_ctx* c = rt->Context();
    const float L[3] = {g_env.LightDirection.x, g_env.LightDirection.y, g_env.LightDirection.z};
    const float C[3] = {g_env.LightColor.x, g_env.LightColor.y, g_env.LightColor.z};
    const float A[3] = {g_env.AmbientColor.x, g_env.AmbientColor.y, g_env.AmbientColor.z};
    ok(vxrt_set_environment(c, L, C, A), "vxrt_set_environment");
    ok(vxrt_set_fov(c, g_fov), "vxrt_set_fov");
    ok(vxrt_set_ortho_window_size(c, g_ortho.x, g_ortho.y), "vxrt_set_ortho_window_size");
    vxrt_render_flags_default(&fl);
    fl.mode = g_switches.DebugView ? VXRT_MODE_DEBUG : VXRT_MODE_SHADED;
    fl.checkerboard = g_switches.Checkerboard;
    fl.ortho = g_switches.Ortho;
    fl.shadow = g_switches.ShadowRay;
    fl.bounce_samples = g_switches.BounceSamples;
    fl.bounce_all_hits = g_switches.BounceAllHits;
    fl.bounce_depth = g_switches.BounceDepth;
    fl.frame_number = -1;  // the context's counter: copy, then increment (Renderer.cu:310,322)
    return c;
}

void RenderScreen(VoxelRaytracer3D* rt, uint32_t w, uint32_t h, void* d_screen_texture, float3 origin, float3 fwd, float3 up,
                  float3 right)
{
    vxrt_render_flags fl;
    vxrt_ctx* c = prepare_launch(rt, fl);
    const float o[3] = {origin.x, origin.y, origin.z}, f[3] = {fwd.x, fwd.y, fwd.z}, u[3] = {up.x, up.y, up.z},
                r[3] = {right.x, right.y, right.z};
    ok(vxrt_render(c, w, h, d_screen_texture, o, f, u, r, &fl), "vxrt_render");
    ok(vxrt_synchronize(c), "vxrt_synchronize");  // RenderScreen returns with the frame finished (Renderer.cu:327)
}

// ---- two frames in flight (RenderScreenAsync / WaitFrame) ---------------------------------------------------------
namespace {
// The two streams of RenderScreenAsync.  Process-wide like the reference's Graphics state (Renderer.cu:24-25,89), but tied to
// a DEVICE (streams of one device must not carry another device's launches: a raytracer on another device gets fresh ones)
// and released with the last VoxelRaytracer3D, not by a static destructor (that would run after the HIP runtime's own
// teardown).
struct FrameSlots {
    hipStream_t stream[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    int device = -1;
    FrameTicket issued = 0;  // tickets count from 1; ticket t uses slot t % 2
    void release()
    {
        for (int i = 0; i < 2; ++i) {
            if (done[i]) (void)hipEventDestroy(done[i]);
            if (stream[i]) (void)hipStreamDestroy(stream[i]);
            done[i] = nullptr;
            stream[i] = nullptr;
        }
        device = -1;
    }
};
FrameSlots g_frames;

void hip_ok(hipError_t e, const char* what)
{
    if (e != hipSuccess) {
        std::fprintf(stderr, "GPUDDA: %s: %s\n", what, hipGetErrorString(e));
        std::exit(EXIT_FAILURE);  // CUDA_SAFE_CALL's behaviour (Renderer.cuh:15-23)
    }
}
}  // namespace

FrameTicket RenderScreenAsync(VoxelRaytracer3D* rt, uint32_t w, uint32_t h, void* d_screen_texture, float3 origin, float3 fwd,
                              float3 up, float3 right)
{
    vxrt_render_flags fl;
    vxrt_ctx* c = prepare_launch(rt, fl);
    int device = 0;
    hip_ok(hipGetDevice(&device), "hipGetDevice");
    if (g_frames.device != device) {  // first use, or a raytracer on another device: finish and drop the old device's pair
        for (int i = 0; i < 2; ++i)
            if (g_frames.done[i])
                (void)hipEventSynchronize(g_frames.done[i]);
        g_frames.release();
        g_frames.device = device;
    }
    const FrameTicket t = ++g_frames.issued;
    const int s = (int)(t % 2);
    if (!g_frames.stream[s]) {
        // non-blocking: the two streams overlap each other and do not join the caller's default stream
        hip_ok(hipStreamCreateWithFlags(&g_frames.stream[s], hipStreamNonBlocking), "hipStreamCreateWithFlags");
        hip_ok(hipEventCreateWithFlags(&g_frames.done[s], hipEventDisableTiming), "hipEventCreateWithFlags");
    }
    // (no host wait here: frame t queues behind frame t-2 on their common stream, so at most two frames EXECUTE at once
    // however far ahead the caller runs; callers pace themselves with WaitFrame)
    fl.stream = g_frames.stream[s];
    const float o[3] = {origin.x, origin.y, origin.z}, f[3] = {fwd.x, fwd.y, fwd.z}, u[3] = {up.x, up.y, up.z},
                r[3] = {right.x, right.y, right.z};
    ok(vxrt_render(c, w, h, d_screen_texture, o, f, u, r, &fl), "vxrt_render");
    hip_ok(hipEventRecord(g_frames.done[s], g_frames.stream[s]), "hipEventRecord");
    return t;
}

void WaitFrame(FrameTicket t)
{
    if (t == 0 || t > g_frames.issued || !g_frames.done[t % 2])
        return;  // never issued (or its streams went with the last raytracer, which waited for them)
    // the event of the newest frame on t's stream: frames of one parity complete in order, so this covers frame t (and
    // waits for a later frame of the same parity if the caller has already launched one)
    hip_ok(hipEventSynchronize(g_frames.done[t % 2]), "hipEventSynchronize");
}

void* FrameStream(FrameTicket t) { return g_frames.stream[t % 2]; }

static void release_frame_slots()
{
    for (int i = 0; i < 2; ++i)
        if (g_frames.done[i])
            (void)hipEventSynchronize(g_frames.done[i]);
    g_frames.release();
}

void RenderScreens(VoxelRaytracer3D* rt, uint32_t w, uint32_t h, const ScreenView* views, uint32_t count)
{
    vxrt_render_flags fl;
    vxrt_ctx* c = prepare_launch(rt, fl);
    std::vector<vxrt_view> v(count);
    for (uint32_t i = 0; i < count; ++i) {
        std::memset(&v[i], 0, sizeof(vxrt_view));
        v[i].d_fb = views[i].d_screen_texture;
        const float3 in[4] = {views[i].origin, views[i].camera_fwd, views[i].camera_up, views[i].camera_right};
        float* out[4] = {v[i].origin, v[i].fwd, v[i].up, v[i].right};
        for (int k = 0; k < 4; ++k) {
            out[k][0] = in[k].x;
            out[k][1] = in[k].y;
            out[k][2] = in[k].z;
        }
        v[i].frame_number = -1;  // each view takes the next FrameNumber, as successive RenderScreen calls would
    }
    for (uint32_t first = 0; first < count; first += 16) {  // one launch takes up to 16 views
        const uint32_t n = count - first < 16u ? count - first : 16u;
        ok(vxrt_render_views(c, w, h, n, v.data() + first, &fl), "vxrt_render_views");
    }
    ok(vxrt_synchronize(c), "vxrt_synchronize");
}

}  // namespace Graphics
}  // namespace GPUDDA

// ---- CreateVoxels ------------------------------------------------------------------------------------

GPUDDA::VoxelBuffer3D CreateVoxels(uint3 size)
{
    using namespace GPUDDA;
    VoxelBuffer3D voxels;
    voxels.dimensions[0] = (uint16_t)size.x;
    voxels.dimensions[1] = (uint16_t)size.y;
    voxels.dimensions[2] = (uint16_t)size.z;
    const size_t nbits = (size_t)size.x * size.y * size.z;
    voxels.grid = BitArray(nbits, false);
    std::memset(voxels.grid.Raw(), 0, voxels.grid.ByteSize());
    // evaluate on the device as 8^3 bricks (one tile each), then scatter tiles into the dense tiled-linear array
    int device = 0;
    (void)hipGetDevice(&device);
    vxrt_ctx* c = nullptr;
    ok(vxrt_create(device, &c), "vxrt_create");
    ok(vxrt_build_world_procedural(c, VXRT_GEN_PERLIN_REF, (int)size.x, (int)size.y, (int)size.z, 8), "vxrt_build_world_procedural");
    vxrt_world_info info{};
    ok(vxrt_world_info_get(c, &info), "vxrt_world_info_get");
    std::vector<uint32_t> coarse((info.ncells + 31) / 32), slot(info.ncells), pool(info.nslots * 16);
    std::vector<float> bnd(info.ncells * 6);
    ok(vxrt_download_world(c, coarse.data(), slot.data(), bnd.data(), pool.data()), "vxrt_download_world");
    vxrt_destroy(c);
    const uint32_t tw = size.x / 8, th = size.y / 8;
    for (uint64_t cell = 0; cell < info.ncells; ++cell) {
        if (slot[cell] == VXRT_EMPTY_SLOT)
            continue;
        uint32_t bx, by, bz;
        GetPositionFromSampleIndex((uint32_t)cell, (uint32_t)info.cdims[0], (uint32_t)info.cdims[1], bx, by, bz);
        const uint64_t tile = (uint64_t)bx + (uint64_t)by * tw + (uint64_t)bz * tw * th;  // brick (bx,by,bz) is dense tile (bx,by,bz)
        std::memcpy(voxels.grid.Raw() + tile * 16, pool.data() + (uint64_t)slot[cell] * 16, 64);
    }
    return voxels;
}
