// vxrt_api.hip -- host side of the C ABI declared in include/vxrt.h.
// Owns the HBM-resident world tables, the camera/lighting state the reference keeps in
// process globals (hFrameInfo / g_env, VoxelRT/Renderer.cu:24-25,89) and the launches.
#include "../../include/vxrt.h"
#include "vxrt_kernels.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <utility>
#include <vector>

namespace vxrt {
hipError_t launch_render(const RenderArgs& A, bool stats, int variant, hipStream_t stream);
int resolve_render_variant(const RenderArgs& A, int variant);
hipError_t launch_trace_batch(const BatchArgs& B, bool stats, int variant, hipStream_t stream);
void launch_deinterleave(const void* shards, unsigned long long shard_stride_bytes, void* fb, uint32_t width,
                         uint32_t height, uint32_t strip_rows, uint32_t strip_count, hipStream_t stream, uint32_t n_views = 1,
                         unsigned long long view_stride_bytes = 0, unsigned long long fb_stride_bytes = 0);
int build_world_on_device(struct ::vxrt_ctx* ctx, int generator, int X, int Y, int Z, int factor);
// re-ordering between the reference's tiled bit order (the C ABI's and the file's) and the HBM order (vxrt_worldgen.hip)
hipError_t layout_bits(const uint32_t* src, uint32_t* dst, const int cd[3], bool to_hbm);
hipError_t layout_meta(const uint2* src, uint2* dst, const int cd[3], bool to_hbm);
hipError_t layout_bricks(const uint32_t* src, uint32_t* dst, uint64_t nbricks, int f, bool to_hbm);  // src == dst: in place
hipError_t chunk_tables(uint2* meta, uint32_t* coarse, const uint2* d_chunk_meta, int tx, int ty, int tz, int cx, int cz);
}  // namespace vxrt

static thread_local std::string g_last_error = "";

static int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define VX_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(VXRT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));              \
    } while (0)

struct vxrt_ctx {
    int device = 0;
    // world
    bool has_world = false;
    vxrt::WorldView view{};
    uint32_t* d_coarse = nullptr;   // (inside coarse_alloc, with slack before and behind: vxrt_wave2.hpp)
    void* coarse_alloc = nullptr;
    void* pool_alloc = nullptr;
    uint2* d_meta = nullptr;
    uint32_t* d_pool = nullptr;
    uint64_t ncells = 0, nslots = 0, pool_capacity_slots = 0;
    uint64_t coarse_alloc_bytes = 0, pool_alloc_bytes = 0;  // what is addressable around the two bit tables (load guard)
    bool guard_no_slack = false;  // vxrt_debug_guard_pretend_no_slack
    // state the reference keeps in globals
    float light_dir[3] = {0, 0, 0};  // g_env is a zero-initialised device global until SetEnvironment (Renderer.cu:89)
    float light_color[3] = {0, 0, 0};
    float ambient[3] = {0, 0, 0};
    float fov = 90.0f;               // hFrameInfo initial value, Renderer.cu:25
    float ortho[2] = {10.0f, 10.0f};
    uint32_t frame_counter = 0;
    int kernel_variant = 4;          // 4 = default (= 7: the persistent kernels on the tracer of vxrt_wave2.hpp), 1 = straightforward loops
    unsigned persistent_waves = 4096;
    unsigned cus = 256;
    unsigned long long* d_stats = nullptr;
    unsigned int* d_queues = nullptr;  // kTileCounterRing queue heads of vxrt::kQueueWords words (one per launch in flight)
    // Counters only ever grow on the device (atomics from any stream); "read and clear" is a host-side snapshot that the
    // next read subtracts, so nothing clears device memory under running kernels.
    unsigned long long stats_base[vxrt::kStatCount] = {};
    // Rings: queue heads (render tile counters / batch tickets) and per-view argument slots of multi-view launches.  A ring
    // entry carries the event of the launch that used it last; taking an entry that is still in flight waits for that launch
    // (launch 65 of 64 in flight, multi-view launch 17 of 16), so an entry is never shared by two live launches.
    std::atomic<unsigned> launch_seq{0};
    hipEvent_t counter_busy[64] = {};
    vxrt::ViewArgs* d_views = nullptr;
    std::atomic<unsigned> view_seq{0};
    hipEvent_t views_busy[16] = {};
    int batch_max_steps = vxrt::kMaxSteps;  // Raytrace's maxSteps for the batch API (vxrt_set_batch_max_steps)
    struct StreamState* stream = nullptr;   // chunk streaming (vxrt_stream_*), or NULL
};
constexpr unsigned kViewSlots = 16;  // multi-view launches that may be in flight at once on one context
constexpr unsigned kTileCounterRing = 64;  // render launches that may be in flight at once on one context
static_assert(kTileCounterRing == sizeof(vxrt_ctx::counter_busy) / sizeof(hipEvent_t), "ring size");
static_assert(kViewSlots == sizeof(vxrt_ctx::views_busy) / sizeof(hipEvent_t), "ring size");

namespace vxrt {

void stream_drop(vxrt_ctx* c);  // defined with the chunk streaming code below

static void free_world(vxrt_ctx* c)
{
    stream_drop(c);
    if (c->coarse_alloc) (void)hipFree(c->coarse_alloc);
    if (c->d_meta) (void)hipFree(c->d_meta);
    if (c->pool_alloc) (void)hipFree(c->pool_alloc);
    c->coarse_alloc = c->pool_alloc = nullptr;
    c->d_coarse = nullptr;
    c->d_meta = nullptr;
    c->d_pool = nullptr;
    c->has_world = false;
    c->ncells = c->nslots = c->pool_capacity_slots = 0;
}

// shared by upload and the device builder
int check_shape(int factor, const int cd[3])
{
    if (!(factor == 8 || factor == 16 || factor == 32))
        return fail(VXRT_ERR_INVALID, "factor must be 8, 16 or 32");
    for (int a = 0; a < 3; ++a)
        if (cd[a] <= 0 || cd[a] % 8 != 0 || cd[a] > 65535)
            return fail(VXRT_ERR_INVALID, "coarse dimensions must be positive multiples of 8 (the tables' tiled order)");
    // cell_index(): x + cx * (z + cz * y) by two 24-bit multiply-adds, 32-bit bit indices biased by one x-z slice, and a
    // lane that has just left the grid may look one more slice ahead (vxrt_wave2.hpp)
    const uint64_t slice = (uint64_t)cd[0] * (uint64_t)cd[2];
    if ((uint64_t)cd[1] * (uint64_t)cd[2] >= (1ull << 24) || slice * (uint64_t)cd[1] + 2ull * slice + 64ull >= (1ull << 32))
        return fail(VXRT_ERR_INVALID, "coarse grid too large for 32-bit cell indices (cy * cz must stay below 2^24, cx * cz * (cy + 2) below 2^32)");
    return VXRT_OK;
}

void fill_view(vxrt_ctx* c, int factor, const int cd[3])
{
    WorldView& v = c->view;
    v.coarse_bits = c->d_coarse;
    v.cell_meta = c->d_meta;
    v.pool = c->d_pool;
    v.cx = cd[0];
    v.cy = cd[1];
    v.cz = cd[2];
    v.c_row = cd[0];
    v.c_slice = cd[0] * cd[2];
    v.f = factor;
    v.f_row = factor;
    v.f_slice = factor * factor;
    v.brick_words = (uint32_t)(factor * factor * factor / 32);
    v.ff = (float)factor;
    v.inv_f = 1.0f / (float)factor;
    v.wmax_x = (float)((double)cd[0] - 1e-6);  // dims - FLT_EPS_DDA in double, VolumeRaytracer.cu:375-376
    v.wmax_y = (float)((double)cd[1] - 1e-6);
    v.wmax_z = (float)((double)cd[2] - 1e-6);
    v.X = cd[0] * factor;
    v.Y = cd[1] * factor;
    v.c_wide = grid_is_wide(cd[0], cd[1], cd[2]) ? 1 : 0;
    // load guard of the probe-counting kernels: the tables proper, and the allocations around them
    v.coarse_end = c->d_coarse + (c->ncells + 31) / 32;
    v.pool_end = c->d_pool + (c->pool_capacity_slots ? c->pool_capacity_slots : 1) * (uint64_t)v.brick_words;
    v.coarse_lo = c->guard_no_slack ? v.coarse_bits : static_cast<const uint32_t*>(c->coarse_alloc);
    v.coarse_hi = c->guard_no_slack ? v.coarse_end : v.coarse_lo + c->coarse_alloc_bytes / 4;
    v.pool_lo = c->guard_no_slack ? v.pool : static_cast<const uint32_t*>(c->pool_alloc);
    v.pool_hi = c->guard_no_slack ? v.pool_end : v.pool_lo + c->pool_alloc_bytes / 4;
}

int alloc_world(vxrt_ctx* c, int factor, const int cd[3], uint64_t pool_slots)
{
    free_world(c);
    c->ncells = (uint64_t)cd[0] * cd[1] * cd[2];
    uint64_t bw = (uint64_t)factor * factor * factor / 32;
    // The tracer of vxrt_wave2.hpp lets a lane that has just left the grid (or a brick) issue one more load, one x-z slice
    // (one brick row-plane) beyond the table at most: both tables sit inside allocations with that much addressable slack
    // before and behind them.  The slack is never written and its bits are never used.
    const uint64_t coarse_bytes = ((c->ncells + 31) / 32) * sizeof(uint32_t);
    const uint64_t coarse_slack = (((uint64_t)cd[0] * cd[2] / 8 + 64) + 255) / 256 * 256;
    VX_HIP(hipMalloc(&c->coarse_alloc, coarse_bytes + 2 * coarse_slack));
    c->coarse_alloc_bytes = coarse_bytes + 2 * coarse_slack;
    c->d_coarse = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(c->coarse_alloc) + coarse_slack);
    VX_HIP(hipMalloc((void**)&c->d_meta, c->ncells * sizeof(uint2)));
    const uint64_t pool_bytes = (pool_slots ? pool_slots : 1) * bw * sizeof(uint32_t);
    const uint64_t pool_slack = (bw * sizeof(uint32_t) + 255) / 256 * 256;
    VX_HIP(hipMalloc(&c->pool_alloc, pool_bytes + 2 * pool_slack));
    c->pool_alloc_bytes = pool_bytes + 2 * pool_slack;
    c->d_pool = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(c->pool_alloc) + pool_slack);
    c->pool_capacity_slots = pool_slots;
    return VXRT_OK;
}

int adopt_world(vxrt_ctx* c, int factor, const int cd[3], uint64_t nslots, uint32_t** d_coarse, uint2** d_meta,
                uint32_t** d_pool)
{
    int rc = alloc_world(c, factor, cd, nslots);
    if (rc)
        return rc;
    c->nslots = nslots;
    fill_view(c, factor, cd);
    c->has_world = true;
    *d_coarse = c->d_coarse;
    *d_meta = c->d_meta;
    *d_pool = c->d_pool;
    return VXRT_OK;
}

int set_error(int code, const char* msg) { return fail(code, msg); }
void abandon_world(vxrt_ctx* c) { free_world(c); }

// Take ring entry `i`: if the launch that used it last has not finished, wait for it (the documented in-flight limits are
// enforced here instead of silently sharing a queue head).  Inside a stream capture nothing can be waited for or
// recorded: the capturing caller keeps within the limits itself.
static hipError_t ring_acquire(hipEvent_t& ev, hipStream_t stream, bool& capturing)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cs) != hipSuccess)
        (void)hipGetLastError();
    capturing = cs != hipStreamCaptureStatusNone;
    if (capturing || !ev)
        return hipSuccess;
    hipError_t e = hipEventQuery(ev);
    if (e == hipErrorNotReady)
        e = hipEventSynchronize(ev);
    return e;
}

static hipError_t ring_release(hipEvent_t& ev, hipStream_t stream, bool capturing)
{
    if (capturing)
        return hipSuccess;
    if (!ev) {
        hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess)
            return e;
    }
    return hipEventRecord(ev, stream);
}

// Default hand-out order of the persistent kernel's tile queue: expected-longest ray chains first, so that what is
// still in flight when the queue runs dry is cheap.  The cost proxy needs the camera only: the elevation of the
// centre ray of each 8-pixel tile row in a Y-up world -- rays just below the horizon travel farthest, rays
// pointing up leave the grid at once.  A few hundred flops on the host per frame; scheduling only.
static void schedule_tile_rows(const RenderArgs& A, const f3& fwd, const f3& up, uint16_t* order, uint32_t& order_n)
{
    const unsigned nty = (A.launch_rows + 7u) / 8u;
    order_n = 0;
    if (nty < 2 || nty > kMaxScheduledTileRows)
        return;
    std::vector<std::pair<float, uint16_t>> key(nty);
    for (unsigned j = 0; j < nty; ++j) {
        unsigned row = j * 8u + 4u < A.launch_rows ? j * 8u + 4u : A.launch_rows - 1u;
        unsigned y = row;  // launch row -> frame row (pixel_coords in vxrt_persist2.hpp)
        if (A.checkerboard)
            y = 2u * row;
        else if (A.strip_count > 1)
            y = ((row / (unsigned)A.strip_rows) * (unsigned)A.strip_count + (unsigned)A.strip_index) * (unsigned)A.strip_rows +
                row % (unsigned)A.strip_rows;
        const float sv = ((float)y / (float)A.height) * 2.0f - 1.0f;
        float dy = fwd.y;
        if (!A.ortho) {
            const float dx = fwd.x + sv * A.ky * up.x, dz = fwd.z + sv * A.ky * up.z;
            dy = fwd.y + sv * A.ky * up.y;
            const float len = sqrtf(dx * dx + dy * dy + dz * dz);
            dy = len > 0.0f ? dy / len : dy;
        }
        key[j] = {dy < 0.0f ? -dy : 2.0f + dy, (uint16_t)j};  // grazing-down first ... straight down, then up
    }
    std::stable_sort(key.begin(), key.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    for (unsigned j = 0; j < nty; ++j)
        order[j] = key[j].second;
    order_n = nty;
}

}  // namespace vxrt

extern "C" {

int vxrt_abi_version(void) { return VXRT_ABI_VERSION; }

const char* vxrt_last_error(void) { return g_last_error.c_str(); }

int vxrt_create(int device, vxrt_ctx** out)
{
    if (!out)
        return fail(VXRT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    VX_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(VXRT_ERR_INVALID, "no such HIP device");
    VX_HIP(hipSetDevice(device));
    vxrt_ctx* c = new (std::nothrow) vxrt_ctx();
    if (!c)
        return fail(VXRT_ERR_NOMEM, "out of host memory");
    c->device = device;
    // counters, and a ring of queue heads for the persistent kernels: launches on different streams may be in flight
    // together (frame k+1 fills the SIMD slots frame k's last waves leave), each needs its own queue head
    const size_t stat_words = (size_t)vxrt::kStatRows * vxrt::kStatRowStride;  // (vxrt_kernels.hpp: rows of counters)
    hipError_t e = hipMalloc((void**)&c->d_stats, stat_words * sizeof(unsigned long long));
    if (e == hipSuccess)
        e = hipMemset(c->d_stats, 0, stat_words * sizeof(unsigned long long));
    if (e == hipSuccess)
        e = hipMalloc((void**)&c->d_queues, sizeof(unsigned int) * vxrt::kQueueWords * kTileCounterRing);
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device);
        c->persistent_waves = (unsigned)prop.multiProcessorCount * 16u;  // 4 waves per SIMD at <= 128 VGPRs
        c->cus = (unsigned)prop.multiProcessorCount;
#ifdef VXRT_EXPERIMENTS  // A/B builds only (make libvxrt_exp.so): the product library reads no environment variable
        if (const char* e = getenv("VXRT_WAVES_PER_CU"))                  // occupancy / grid experiments
            if (atoi(e) > 0 && atoi(e) <= 32)
                c->persistent_waves = (unsigned)prop.multiProcessorCount * (unsigned)atoi(e);
#endif
    }
    if (e != hipSuccess) {
        if (c->d_stats) (void)hipFree(c->d_stats);
        if (c->d_queues) (void)hipFree(c->d_queues);
        delete c;
        return fail(VXRT_ERR_HIP, std::string("context setup: ") + hipGetErrorString(e));
    }
    *out = c;
    return VXRT_OK;
}

int vxrt_destroy(vxrt_ctx* c)
{
    if (!c)
        return VXRT_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    vxrt::free_world(c);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->d_queues) (void)hipFree(c->d_queues);
    if (c->d_views) (void)hipFree(c->d_views);
    for (hipEvent_t& e : c->counter_busy)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t& e : c->views_busy)
        if (e) (void)hipEventDestroy(e);
    delete c;
    return VXRT_OK;
}

int vxrt_kernel_for_launch(const vxrt_ctx* c, uint32_t width, uint32_t height, const vxrt_render_flags* fl, uint32_t nviews)
{
    if (!c || !fl || fl->struct_size != sizeof(vxrt_render_flags))
        return -1;
    vxrt::RenderArgs A;
    memset(&A, 0, sizeof(A));
    if (c->has_world)
        A.W = c->view;
    A.width = width;
    A.shadow = fl->shadow ? 1 : 0;
    A.bounce_samples = fl->bounce_samples;
    A.nviews = nviews;
    // the launch grid's rows, as vxrt_render works them out
    if (fl->checkerboard)
        A.launch_rows = height >> 1;
    else if (fl->strip_count > 1)
        A.launch_rows = vxrt_compact_rows(height, fl->strip_rows > 0 ? fl->strip_rows : 16, fl->strip_count, fl->strip_index);
    else
        A.launch_rows = height;
    return vxrt::resolve_render_variant(A, c->kernel_variant);
}

int vxrt_set_persistent_waves_per_cu(vxrt_ctx* c, int waves_per_cu)
{
    if (!c || waves_per_cu < 0 || waves_per_cu > 32)
        return fail(VXRT_ERR_INVALID, "waves_per_cu must be in [1, 32], or 0 for the default");
    c->persistent_waves = c->cus * (unsigned)(waves_per_cu == 0 ? 16 : waves_per_cu);
    return VXRT_OK;
}

int vxrt_has_experiments(void)
{
#ifdef VXRT_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

int vxrt_debug_guard_pretend_no_slack(vxrt_ctx* c, int on)
{
    if (!c)
        return fail(VXRT_ERR_INVALID, "NULL context");
    c->guard_no_slack = on != 0;
    if (c->has_world) {
        const int cd[3] = {c->view.cx, c->view.cy, c->view.cz};
        vxrt::fill_view(c, c->view.f, cd);
    }
    return VXRT_OK;
}

int vxrt_set_kernel_variant(vxrt_ctx* c, int variant)
{
    if (!c || !(variant == 1 || variant == 4 || variant == 7))
        return fail(VXRT_ERR_INVALID, "variant must be 4 (default), 7 (the persistent kernels on the wave-level tracer: what the default runs) or 1 (straightforward per-lane loops, the cross-check)");
    c->kernel_variant = variant;
    return VXRT_OK;
}

int vxrt_synchronize(vxrt_ctx* c)
{
    if (!c)
        return fail(VXRT_ERR_INVALID, "ctx is NULL");
    VX_HIP(hipSetDevice(c->device));
    VX_HIP(hipDeviceSynchronize());
    return VXRT_OK;
}

int vxrt_upload_world(vxrt_ctx* c, const vxrt_world_desc* d)
{
    if (!c || !d || d->struct_size != sizeof(vxrt_world_desc))
        return fail(VXRT_ERR_INVALID, "bad ctx/desc");
    if (!d->coarse_bits || !d->brick_slot || !d->bounds || (d->nslots && !d->pool))
        return fail(VXRT_ERR_INVALID, "NULL table");
    int cd[3] = {d->cdims[0], d->cdims[1], d->cdims[2]};
    int rc = vxrt::check_shape(d->factor, cd);
    if (rc)
        return rc;
    VX_HIP(hipSetDevice(c->device));
    const int f = d->factor;
    const uint64_t ncells = (uint64_t)cd[0] * cd[1] * cd[2];
    // flatten {descriptor, bounds} pairs into 8-byte cell_meta records
    std::vector<uint2> meta(ncells);
    for (uint64_t i = 0; i < ncells; ++i) {
        bool bit = (d->coarse_bits[i >> 5] >> (i & 31)) & 1u;
        uint32_t slot = d->brick_slot[i];
        const float* b = d->bounds + i * 6;
        uint32_t packed = 0;
        if (bit) {
            if (slot == VXRT_EMPTY_SLOT || slot >= d->nslots)
                return fail(VXRT_ERR_INVALID, "occupied coarse cell without a valid brick slot");
            for (int k = 0; k < 6; ++k) {
                float v = b[k];
                if (!(v >= 0.0f && v <= (float)(f - 1) && v == (float)(int)v))
                    return fail(VXRT_ERR_INVALID, "brick extents must be integers in [0, factor-1]");
                packed |= (uint32_t)(int)v << (5 * k);
            }
        } else {
            slot = VXRT_EMPTY_SLOT;
        }
        meta[i] = make_uint2(slot, packed);
    }
    rc = vxrt::alloc_world(c, f, cd, d->nslots);
    if (rc)
        return rc;
    const uint64_t bw = (uint64_t)f * f * f / 32;
    // the tables arrive in the reference's tiled order and are re-ordered on the device into the HBM order
    const uint64_t coarse_bytes = ((ncells + 31) / 32) * sizeof(uint32_t);
    uint32_t* t_coarse = nullptr;
    uint2* t_meta = nullptr;
    hipError_t e = hipMalloc((void**)&t_coarse, coarse_bytes);
    if (e == hipSuccess)
        e = hipMalloc((void**)&t_meta, ncells * sizeof(uint2));
    if (e == hipSuccess)
        e = hipMemcpy(t_coarse, d->coarse_bits, coarse_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(t_meta, meta.data(), ncells * sizeof(uint2), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = vxrt::layout_bits(t_coarse, c->d_coarse, cd, true);
    if (e == hipSuccess)
        e = vxrt::layout_meta(t_meta, c->d_meta, cd, true);
    if (e == hipSuccess && d->nslots)
        e = hipMemcpy(c->d_pool, d->pool, d->nslots * bw * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && d->nslots)
        e = vxrt::layout_bricks(c->d_pool, c->d_pool, d->nslots, f, true);
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    (void)hipFree(t_coarse);
    (void)hipFree(t_meta);
    if (e != hipSuccess) {
        vxrt::free_world(c);
        return fail(VXRT_ERR_HIP, std::string("world upload: ") + hipGetErrorString(e));
    }
    c->nslots = d->nslots;
    vxrt::fill_view(c, f, cd);
    c->has_world = true;
    return VXRT_OK;
}

int vxrt_build_world_procedural(vxrt_ctx* c, int generator, int X, int Y, int Z, int factor)
{
    if (!c)
        return fail(VXRT_ERR_INVALID, "ctx is NULL");
    VX_HIP(hipSetDevice(c->device));
    return vxrt::build_world_on_device(c, generator, X, Y, Z, factor);
}

int vxrt_world_info_get(vxrt_ctx* c, vxrt_world_info* out)
{
    if (!c || !out)
        return fail(VXRT_ERR_INVALID, "NULL argument");
    if (!c->has_world)
        return fail(VXRT_ERR_NO_WORLD, "no world resident");
    out->factor = c->view.f;
    out->cdims[0] = c->view.cx;
    out->cdims[1] = c->view.cy;
    out->cdims[2] = c->view.cz;
    out->ncells = c->ncells;
    out->nslots = c->nslots;
    out->hbm_bytes = ((c->ncells + 31) / 32) * 4 + c->ncells * sizeof(uint2) + c->nslots * c->view.brick_words * 4ull;
    return VXRT_OK;
}

int vxrt_download_world(vxrt_ctx* c, uint32_t* coarse_bits, uint32_t* brick_slot, float* bounds, uint32_t* pool)
{
    if (!c || !coarse_bits || !brick_slot || !bounds)
        return fail(VXRT_ERR_INVALID, "NULL argument");
    if (!c->has_world)
        return fail(VXRT_ERR_NO_WORLD, "no world resident");
    VX_HIP(hipSetDevice(c->device));
    VX_HIP(hipDeviceSynchronize());
    // back into the reference's tiled order, through device temporaries (the pool in pieces of at most 64 MiB)
    const int cd[3] = {c->view.cx, c->view.cy, c->view.cz};
    const uint64_t coarse_bytes = ((c->ncells + 31) / 32) * sizeof(uint32_t);
    const uint64_t bw = c->view.brick_words;
    const uint64_t piece = std::max<uint64_t>(1, (64ull << 20) / (bw * 4));  // bricks per piece
    uint32_t *t_coarse = nullptr, *t_pool = nullptr;
    uint2* t_meta = nullptr;
    std::vector<uint2> meta(c->ncells);
    hipError_t e = hipMalloc((void**)&t_coarse, coarse_bytes);
    if (e == hipSuccess)
        e = hipMalloc((void**)&t_meta, c->ncells * sizeof(uint2));
    if (e == hipSuccess)
        e = vxrt::layout_bits(c->d_coarse, t_coarse, cd, false);
    if (e == hipSuccess)
        e = vxrt::layout_meta(c->d_meta, t_meta, cd, false);
    if (e == hipSuccess)
        e = hipMemcpy(coarse_bits, t_coarse, coarse_bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        e = hipMemcpy(meta.data(), t_meta, c->ncells * sizeof(uint2), hipMemcpyDeviceToHost);
    if (e == hipSuccess && pool && c->nslots) {
        e = hipMalloc((void**)&t_pool, std::min<uint64_t>(piece, c->nslots) * bw * 4);
        for (uint64_t at = 0; e == hipSuccess && at < c->nslots; at += piece) {
            const uint64_t n = std::min<uint64_t>(piece, c->nslots - at);
            e = vxrt::layout_bricks(c->d_pool + at * bw, t_pool, n, c->view.f, false);
            if (e == hipSuccess)
                e = hipMemcpy(pool + at * bw, t_pool, n * bw * 4, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(t_coarse);
    (void)hipFree(t_meta);
    (void)hipFree(t_pool);
    if (e != hipSuccess)
        return fail(VXRT_ERR_HIP, std::string("world download: ") + hipGetErrorString(e));
    for (uint64_t i = 0; i < c->ncells; ++i) {
        brick_slot[i] = meta[i].x;
        float* b = bounds + i * 6;
        if (meta[i].x == VXRT_EMPTY_SLOT) {
            b[0] = b[1] = b[2] = 0.0f;
            b[3] = b[4] = b[5] = -1.0f;  // VolumeRaytracer.cuh:454-467
        } else {
            for (int k = 0; k < 6; ++k)
                b[k] = (float)((meta[i].y >> (5 * k)) & 31u);
        }
    }
    return VXRT_OK;
}

int vxrt_set_environment(vxrt_ctx* c, const float light_dir[3], const float light_color[3], const float ambient[3])
{
    if (!c || !light_dir || !light_color || !ambient)
        return fail(VXRT_ERR_INVALID, "NULL argument");
    memcpy(c->light_dir, light_dir, sizeof(c->light_dir));
    memcpy(c->light_color, light_color, sizeof(c->light_color));
    memcpy(c->ambient, ambient, sizeof(c->ambient));
    return VXRT_OK;
}

int vxrt_set_fov(vxrt_ctx* c, float fov_degrees)
{
    if (!c)
        return fail(VXRT_ERR_INVALID, "ctx is NULL");
    c->fov = fov_degrees;
    return VXRT_OK;
}

int vxrt_set_ortho_window_size(vxrt_ctx* c, float sx, float sy)
{
    if (!c)
        return fail(VXRT_ERR_INVALID, "ctx is NULL");
    c->ortho[0] = sx;
    c->ortho[1] = sy;
    return VXRT_OK;
}

// GetDirections (Renderer.cu:27-42): float cos/sin, forward and up negated on return
void vxrt_get_directions(const float euler[3], float fwd[3], float up[3], float right[3])
{
    float fx = cosf(euler[0]) * sinf(euler[1]);
    float fy = -sinf(euler[0]);
    float fz = cosf(euler[0]) * cosf(euler[1]);
    float rx = cosf(euler[1]), ry = 0.0f, rz = -sinf(euler[1]);
    float ux = fy * rz - fz * ry, uy = fz * rx - fx * rz, uz = fx * ry - fy * rx;
    fwd[0] = fx * -1;
    fwd[1] = fy * -1;
    fwd[2] = fz * -1;
    up[0] = ux * -1;
    up[1] = uy * -1;
    up[2] = uz * -1;
    right[0] = rx;
    right[1] = ry;
    right[2] = rz;
}

void vxrt_render_flags_default(vxrt_render_flags* f)
{
    if (!f)
        return;
    memset(f, 0, sizeof(*f));
    f->struct_size = sizeof(*f);
    f->mode = VXRT_MODE_SHADED;
    f->frame_number = -1;
    f->strip_rows = 16;
    f->strip_count = 1;
    f->tile_schedule = 1;
}

uint32_t vxrt_compact_rows(uint32_t height, int32_t strip_rows, int32_t strip_count, int32_t strip_index)
{
    if (strip_rows <= 0 || strip_count <= 1)
        return height;
    uint32_t rows = 0;
    uint32_t nstrips = (height + (uint32_t)strip_rows - 1) / (uint32_t)strip_rows;
    for (uint32_t s = (uint32_t)strip_index; s < nstrips; s += (uint32_t)strip_count) {
        uint32_t begin = s * (uint32_t)strip_rows;
        uint32_t end = begin + (uint32_t)strip_rows < height ? begin + (uint32_t)strip_rows : height;
        rows += end - begin;
    }
    return rows;
}

// one launch for `nviews` views; nviews == 0: the single view `views[0]` through the single-view kernel arguments
static int render_launch(vxrt_ctx* c, uint32_t width, uint32_t height, unsigned nviews, const vxrt_view* views,
                         const vxrt_render_flags* fl)
{
    vxrt_render_flags def;
    if (!fl) {
        vxrt_render_flags_default(&def);
        fl = &def;
    }
    if (fl->struct_size != sizeof(vxrt_render_flags))
        return fail(VXRT_ERR_INVALID, "vxrt_render_flags size mismatch");
    if (!c->has_world)
        return fail(VXRT_ERR_NO_WORLD, "no world resident");
    if (width == 0 || height == 0 || height > 65535u)
        return fail(VXRT_ERR_INVALID, "empty frame, or more than 65535 rows");
    if (fl->strip_count > 1 && (fl->strip_rows <= 0 || fl->strip_index < 0 || fl->strip_index >= fl->strip_count))
        return fail(VXRT_ERR_INVALID, "bad strip sharding");
    const unsigned n = nviews ? nviews : 1u;
    for (unsigned v = 0; v < n; ++v) {
        if (!views[v].d_fb)
            return fail(VXRT_ERR_INVALID, "a view has no framebuffer");
        // ray validity (include/vxrt.h): a camera with a non-finite component would hand every pixel an invalid ray
        for (int a = 0; a < 3; ++a)
            if (!std::isfinite(views[v].origin[a]) || !std::isfinite(views[v].fwd[a]) || !std::isfinite(views[v].up[a]) ||
                !std::isfinite(views[v].right[a]))
                return fail(VXRT_ERR_INVALID, "camera origin / forward / up / right must be finite numbers");
    }
    if (fl->d_accum && nviews != 0)
        return fail(VXRT_ERR_INVALID, "temporal accumulation (d_accum) is per view: use vxrt_render");
    if (fl->d_accum && (reinterpret_cast<uintptr_t>(fl->d_accum) & 15u))
        return fail(VXRT_ERR_INVALID, "d_accum must be 16-byte aligned");
    VX_HIP(hipSetDevice(c->device));
    hipStream_t stream = (hipStream_t)fl->stream;

    vxrt::RenderArgs A;
    memset(&A, 0, sizeof(A));
    A.W = c->view;
    A.width = width;
    A.height = height;
    {   // getRayDirection's per-pixel constants (Renderer.cu:46,50-52), hoisted to the host
        float aspect = (float)width / (float)height;
        float fov = (float)((double)c->fov * 3.1415 / 180.0);
        A.kx = tanf(fov / 2.0f) * aspect;
        A.ky = tanf(fov / 2.0f);
        A.ratio = (float)width / (float)height;
        A.ortho_x = c->ortho[0];
        A.ortho_y = c->ortho[1];
    }
    A.light_dir = vxrt::f3{c->light_dir[0], c->light_dir[1], c->light_dir[2]};
    {   // unit3() of vxrt_device.hpp on the host: v * (1 / sqrt(dot(v, v))), binary32 throughout, no contraction
        const float lx = c->light_dir[0], ly = c->light_dir[1], lz = c->light_dir[2];
        const float dd = lx * lx + ly * ly + lz * lz;
        const float inv = 1.0f / sqrtf(dd);
        A.light_unit = vxrt::f3{lx * inv, ly * inv, lz * inv};
        A.light_step = vxrt::f3{A.light_unit.x * 0.01f, A.light_unit.y * 0.01f, A.light_unit.z * 0.01f};
    }
    A.light_color = vxrt::f3{c->light_color[0], c->light_color[1], c->light_color[2]};
    A.ambient = vxrt::f3{c->ambient[0], c->ambient[1], c->ambient[2]};
    A.mode = fl->mode;
    A.checkerboard = fl->checkerboard ? 1 : 0;
    A.shadow = fl->shadow ? 1 : 0;
    A.bounce_samples = fl->bounce_samples < 0 ? 0 : fl->bounce_samples;
    A.bounce_samples_f = (float)A.bounce_samples;
    A.inv_width = 1.0f / (float)(int)width;
    A.inv_height = 1.0f / (float)(int)height;
    A.inv_bounce_samples = A.bounce_samples > 0 ? 1.0f / A.bounce_samples_f : 0.0f;
    A.bounce_all_hits = fl->bounce_all_hits ? 1 : 0;
    A.bounce_depth = fl->bounce_depth >= 2 ? 2 : 1;
    A.ortho = fl->ortho ? 1 : 0;
    A.strip_rows = fl->strip_rows > 0 ? fl->strip_rows : 16;
    A.strip_count = fl->strip_count > 1 ? fl->strip_count : 1;
    A.strip_index = fl->strip_index;
    A.compact = fl->compact ? 1 : 0;
    A.accum = reinterpret_cast<float4*>(fl->d_accum);
    A.accum_reset = fl->accum_reset ? 1 : 0;
    A.strip_shift = -1;
    for (int b = 0; b < 31; ++b)
        if (A.strip_rows == (1 << b))
            A.strip_shift = b;
    // launch shape: RenderScreen halves the rows under checkerboard (Renderer.cu:311-316); a shard
    // without checkerboard launches only its own rows
    if (A.checkerboard)
        A.launch_rows = height >> 1;
    else if (A.strip_count > 1)
        A.launch_rows = vxrt_compact_rows(height, A.strip_rows, A.strip_count, A.strip_index);
    else
        A.launch_rows = height;
    A.stats = c->d_stats;  // counters accumulate until vxrt_frame_stats_get reads and clears them
    A.persistent_waves = c->persistent_waves;
    const bool persistent = c->kernel_variant != 1;
    const bool schedule = fl->tile_schedule && persistent;

    auto frame_number_of = [&](const vxrt_view& v) -> uint32_t {
        if (v.frame_number >= 0)
            return (uint32_t)v.frame_number;
        return c->frame_counter++;  // the copy precedes the increment, Renderer.cu:310,322
    };
    auto f3_of = [](const float* p) { return vxrt::f3{p[0], p[1], p[2]}; };

    if (nviews == 0 || !persistent) {  // single-view kernel arguments; variant 1 takes the views one by one
        for (unsigned v = 0; v < n; ++v) {
            A.frame_number = frame_number_of(views[v]);
            A.origin = f3_of(views[v].origin);
            A.fwd = f3_of(views[v].fwd);
            A.up = f3_of(views[v].up);
            A.right = f3_of(views[v].right);
            A.fb = (uint8_t*)views[v].d_fb;
            A.color_aov = views[v].d_color_aov;
            A.hit_aov = (long long*)views[v].d_hit_aov;
            A.want_hit_aov = A.hit_aov != nullptr;
            A.tile_order = nviews == 0 ? fl->d_tile_order : nullptr;
            A.row_order_n = 0;
            if (schedule && !A.tile_order)
                vxrt::schedule_tile_rows(A, A.fwd, A.up, A.row_order, A.row_order_n);
            const unsigned slot = c->launch_seq.fetch_add(1u) % kTileCounterRing;
            bool capturing = false;
            VX_HIP(vxrt::ring_acquire(c->counter_busy[slot], stream, capturing));
            A.tile_counter = c->d_queues + (size_t)slot * vxrt::kQueueWords;
            VX_HIP(vxrt::launch_render(A, fl->collect_stats != 0, c->kernel_variant, stream));
            VX_HIP(hipGetLastError());
            VX_HIP(vxrt::ring_release(c->counter_busy[slot], stream, capturing));
        }
        return VXRT_OK;
    }

    // multi-view launch: the per-view arguments travel through a ring of device slots (stream-ordered copy)
    if (!c->d_views)
        VX_HIP(hipMalloc((void**)&c->d_views, sizeof(vxrt::ViewArgs) * vxrt::kMaxViews * kViewSlots));
    std::vector<vxrt::ViewArgs> host(n);
    for (unsigned v = 0; v < n; ++v) {
        vxrt::ViewArgs& S = host[v];
        memset(&S, 0, sizeof(S));
        S.origin = f3_of(views[v].origin);
        S.fwd = f3_of(views[v].fwd);
        S.up = f3_of(views[v].up);
        S.right = f3_of(views[v].right);
        S.frame_number = frame_number_of(views[v]);
        S.fb = (uint8_t*)views[v].d_fb;
        S.color_aov = views[v].d_color_aov;
        S.hit_aov = (long long*)views[v].d_hit_aov;
        A.want_hit_aov |= S.hit_aov != nullptr;
        if (schedule)
            vxrt::schedule_tile_rows(A, S.fwd, S.up, S.row_order, S.row_order_n);
    }
    const unsigned vslot = c->view_seq.fetch_add(1u) % kViewSlots, cslot = c->launch_seq.fetch_add(1u) % kTileCounterRing;
    bool capturing = false, capturing2 = false;
    VX_HIP(vxrt::ring_acquire(c->views_busy[vslot], stream, capturing));
    VX_HIP(vxrt::ring_acquire(c->counter_busy[cslot], stream, capturing2));
    vxrt::ViewArgs* slot = c->d_views + (size_t)vslot * vxrt::kMaxViews;
    VX_HIP(hipMemcpyAsync(slot, host.data(), sizeof(vxrt::ViewArgs) * n, hipMemcpyHostToDevice, stream));
    A.views = slot;
    A.nviews = n;
    A.tile_counter = c->d_queues + (size_t)cslot * vxrt::kQueueWords;
    VX_HIP(vxrt::launch_render(A, fl->collect_stats != 0, c->kernel_variant, stream));
    VX_HIP(hipGetLastError());
    VX_HIP(vxrt::ring_release(c->views_busy[vslot], stream, capturing));
    VX_HIP(vxrt::ring_release(c->counter_busy[cslot], stream, capturing));
    return VXRT_OK;
}

int vxrt_render(vxrt_ctx* c, uint32_t width, uint32_t height, void* d_fb, const float origin[3], const float fwd[3],
                const float up[3], const float right[3], const vxrt_render_flags* fl)
{
    if (!c || !d_fb || !origin || !fwd || !up || !right)
        return fail(VXRT_ERR_INVALID, "NULL argument");
    vxrt_view v;
    memset(&v, 0, sizeof(v));
    v.d_fb = d_fb;
    for (int a = 0; a < 3; ++a) {
        v.origin[a] = origin[a];
        v.fwd[a] = fwd[a];
        v.up[a] = up[a];
        v.right[a] = right[a];
    }
    v.frame_number = fl ? fl->frame_number : -1;
    v.d_color_aov = fl ? fl->d_color_aov : nullptr;
    v.d_hit_aov = fl ? fl->d_hit_aov : nullptr;
    return render_launch(c, width, height, 0, &v, fl);
}

int vxrt_render_views(vxrt_ctx* c, uint32_t width, uint32_t height, uint32_t n_views, const vxrt_view* views,
                      const vxrt_render_flags* fl)
{
    if (!c || !views)
        return fail(VXRT_ERR_INVALID, "NULL argument");
    if (n_views == 0 || n_views > vxrt::kMaxViews)
        return fail(VXRT_ERR_INVALID, "between 1 and 16 views per launch");
    return render_launch(c, width, height, n_views, views, fl);
}

int vxrt_frame_stats_get(vxrt_ctx* c, vxrt_frame_stats* out)
{
    if (!c || !out)
        return fail(VXRT_ERR_INVALID, "NULL argument");
    VX_HIP(hipSetDevice(c->device));
    VX_HIP(hipDeviceSynchronize());  // every stream of the device, non-blocking ones included
    unsigned long long now[vxrt::kStatCount] = {}, h[vxrt::kStatCount];
    std::vector<unsigned long long> rows((size_t)vxrt::kStatRows * vxrt::kStatRowStride);
    VX_HIP(hipMemcpy(rows.data(), c->d_stats, rows.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (unsigned r = 0; r < vxrt::kStatRows; ++r)
        for (int i = 0; i < vxrt::kStatCount; ++i)
            now[i] += rows[(size_t)r * vxrt::kStatRowStride + i];
    for (int i = 0; i < vxrt::kStatCount; ++i) {  // what was added since the previous read; the device copy only grows
        h[i] = now[i] - c->stats_base[i];
        c->stats_base[i] = now[i];
    }
    out->primary_rays = h[vxrt::kStatPrimary];
    out->shadow_rays = h[vxrt::kStatShadow];
    out->bounce_rays = h[vxrt::kStatBounce];
    out->primary_hits = h[vxrt::kStatPrimaryHits];
    out->coarse_probes = h[vxrt::kStatCoarseProbes];
    out->brick_entries = h[vxrt::kStatBrickEntries];
    out->fine_probes = h[vxrt::kStatFineProbes];
    out->dbg[0] = h[vxrt::kStatDbgIters];
    out->dbg[1] = h[vxrt::kStatDbgWalkLanes];
    out->dbg[2] = h[vxrt::kStatDbgEndRuns];
    out->dbg[3] = h[vxrt::kStatDbgBoxRuns];
    out->dbg[4] = h[vxrt::kStatDbgNextRuns];
    out->dbg[5] = h[vxrt::kStatDbgEndLanes];
    out->dbg[6] = h[vxrt::kStatDbgBoxLanes];
    out->dbg[7] = h[vxrt::kStatDbgNextLanes];
    out->dbg[8] = h[vxrt::kStatDbgLifetime];
    out->dbg[9] = h[vxrt::kStatDbgDrained];
    out->dbg[10] = h[vxrt::kStatDbgNextTicks];
    out->dbg[11] = h[vxrt::kStatDbgParkTicks];
    out->guard_slack_loads = h[vxrt::kStatGuardSlack];
    out->guard_stray_loads = h[vxrt::kStatGuardStray];
    return VXRT_OK;
}

int vxrt_deinterleave_strips(vxrt_ctx* c, uint32_t width, uint32_t height, int32_t strip_rows, int32_t strip_count,
                             const void* d_shards, uint64_t shard_stride_bytes, void* d_fb, void* stream)
{
    if (!c || !d_shards || !d_fb)
        return fail(VXRT_ERR_INVALID, "NULL argument");
    if (width % 4 != 0 || shard_stride_bytes % 16 != 0 || strip_rows <= 0 || strip_count <= 0)
        return fail(VXRT_ERR_INVALID, "width must be a multiple of 4 pixels and the shard stride of 16 bytes");
    VX_HIP(hipSetDevice(c->device));
    vxrt::launch_deinterleave(d_shards, shard_stride_bytes, d_fb, width, height, (uint32_t)strip_rows,
                              (uint32_t)strip_count, (hipStream_t)stream);
    VX_HIP(hipGetLastError());
    return VXRT_OK;
}

int vxrt_deinterleave_views(vxrt_ctx* c, uint32_t width, uint32_t height, int32_t strip_rows, int32_t strip_count,
                            const void* d_shards, uint64_t shard_stride_bytes, uint64_t view_stride_bytes, uint32_t n_views,
                            void* d_fb, uint64_t fb_stride_bytes, void* stream)
{
    if (!c || !d_shards || !d_fb)
        return fail(VXRT_ERR_INVALID, "NULL argument");
    if (width % 4 != 0 || shard_stride_bytes % 16 != 0 || view_stride_bytes % 16 != 0 || fb_stride_bytes % 16 != 0 ||
        strip_rows <= 0 || strip_count <= 0)
        return fail(VXRT_ERR_INVALID, "width must be a multiple of 4 pixels and the strides of 16 bytes");
    if (n_views > 65535u)
        return fail(VXRT_ERR_INVALID, "too many views");
    VX_HIP(hipSetDevice(c->device));
    vxrt::launch_deinterleave(d_shards, shard_stride_bytes, d_fb, width, height, (uint32_t)strip_rows, (uint32_t)strip_count,
                              (hipStream_t)stream, n_views, view_stride_bytes, fb_stride_bytes);
    VX_HIP(hipGetLastError());
    return VXRT_OK;
}

int vxrt_trace_batch(vxrt_ctx* c, const float* d_origins, const float* d_dirs, uint64_t n, float* d_pos,
                     float* d_normal, int32_t* d_steps, uint8_t* d_hit, int64_t* d_voxel, vxrt_frame_stats* stats,
                     void* stream_)
{
    if (!c || (n && (!d_origins || !d_dirs || !d_pos || !d_normal || !d_steps)))
        return fail(VXRT_ERR_INVALID, "NULL argument");
    if (!c->has_world)
        return fail(VXRT_ERR_NO_WORLD, "no world resident");
    VX_HIP(hipSetDevice(c->device));
    hipStream_t stream = (hipStream_t)stream_;
    vxrt::BatchArgs B;
    memset(&B, 0, sizeof(B));
    B.W = c->view;
    B.origins = d_origins;
    B.dirs = d_dirs;
    B.n = n;
    B.pos = d_pos;
    B.normal = d_normal;
    B.steps = d_steps;
    B.hit = d_hit;
    B.voxel = (long long*)d_voxel;
    B.stats = c->d_stats;
    const unsigned tslot = c->launch_seq.fetch_add(1u) % kTileCounterRing;
    bool capturing = false;
    VX_HIP(vxrt::ring_acquire(c->counter_busy[tslot], stream, capturing));
    B.ticket = c->d_queues + (size_t)tslot * vxrt::kQueueWords;
    B.persistent_waves = c->persistent_waves;
    B.max_steps = c->batch_max_steps;
    if (stats) {  // a stats request reports what ran between two device-wide syncs: this batch alone if nothing else is submitted
        vxrt_frame_stats drop;
        int rc = vxrt_frame_stats_get(c, &drop);
        if (rc)
            return rc;
    }
    VX_HIP(vxrt::launch_trace_batch(B, stats != nullptr, c->kernel_variant, stream));
    VX_HIP(hipGetLastError());
    VX_HIP(vxrt::ring_release(c->counter_busy[tslot], stream, capturing));
    if (stats) {
        VX_HIP(hipStreamSynchronize(stream));
        return vxrt_frame_stats_get(c, stats);
    }
    return VXRT_OK;
}

int vxrt_set_batch_max_steps(vxrt_ctx* c, int32_t max_steps)
{
    if (!c || max_steps < 1 || max_steps > vxrt::kMaxSteps)
        return fail(VXRT_ERR_INVALID, "max_steps must be in [1, 2048]");
    c->batch_max_steps = max_steps;
    return VXRT_OK;
}

int vxrt_trace_batch_host(vxrt_ctx* c, const float* origins, const float* dirs, uint64_t n, float* pos, float* normal,
                          int32_t* steps, uint8_t* hit, int64_t* voxel, vxrt_frame_stats* stats)
{
    if (!c || (n && (!origins || !dirs || !pos || !normal || !steps)))
        return fail(VXRT_ERR_INVALID, "NULL argument");
    if (!c->has_world)
        return fail(VXRT_ERR_NO_WORLD, "no world resident");
    if (n == 0)
        return VXRT_OK;
    VX_HIP(hipSetDevice(c->device));
    float *d_o = nullptr, *d_d = nullptr, *d_p = nullptr, *d_n = nullptr;
    int32_t* d_s = nullptr;
    uint8_t* d_h = nullptr;
    int64_t* d_v = nullptr;
    int rc = VXRT_OK;
    auto cleanup = [&]() {
        (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_p); (void)hipFree(d_n);
        (void)hipFree(d_s); (void)hipFree(d_h); (void)hipFree(d_v);
    };
#define VX_TRY(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            cleanup();                                                                     \
            return fail(VXRT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));  \
        }                                                                                  \
    } while (0)
    VX_TRY(hipMalloc((void**)&d_o, n * 12));
    VX_TRY(hipMalloc((void**)&d_d, n * 12));
    VX_TRY(hipMalloc((void**)&d_p, n * 12));
    VX_TRY(hipMalloc((void**)&d_n, n * 12));
    VX_TRY(hipMalloc((void**)&d_s, n * 4));
    VX_TRY(hipMalloc((void**)&d_h, n));
    VX_TRY(hipMalloc((void**)&d_v, n * 8));
    VX_TRY(hipMemcpy(d_o, origins, n * 12, hipMemcpyHostToDevice));
    VX_TRY(hipMemcpy(d_d, dirs, n * 12, hipMemcpyHostToDevice));
    rc = vxrt_trace_batch(c, d_o, d_d, n, d_p, d_n, d_s, d_h, d_v, stats, nullptr);
    if (rc == VXRT_OK) {
        VX_TRY(hipDeviceSynchronize());
        VX_TRY(hipMemcpy(pos, d_p, n * 12, hipMemcpyDeviceToHost));
        VX_TRY(hipMemcpy(normal, d_n, n * 12, hipMemcpyDeviceToHost));
        VX_TRY(hipMemcpy(steps, d_s, n * 4, hipMemcpyDeviceToHost));
        if (hit) VX_TRY(hipMemcpy(hit, d_h, n, hipMemcpyDeviceToHost));
        if (voxel) VX_TRY(hipMemcpy(voxel, d_v, n * 8, hipMemcpyDeviceToHost));
    }
#undef VX_TRY
    cleanup();
    return rc;
}

// ---- brickmap file: the resident tables as they lie in HBM, behind a versioned header ------
namespace {

constexpr char kFileMagic[8] = {'V', 'X', 'B', 'R', 'K', 'M', 'A', 'P'};
constexpr uint32_t kFileVersion = 2;  // 2: position-sensitive stream checksums (sum + sum of running sums)
constexpr size_t kFileChunk = 64u << 20;  // staging buffer for the table streams

struct FileHeader {  // 120 bytes, little endian
    char magic[8];
    uint32_t version, header_bytes;
    int32_t factor, cdims[3];
    uint64_t ncells, nslots;
    uint64_t coarse_bytes, meta_bytes, pool_bytes;  // the three table streams, in this order after the header
    uint64_t sum[3];                                 // per stream: a = sum of its 32-bit words, mod 2^64
    uint64_t sum2[3];                                // per stream: b = sum of the running values of a (Fletcher style):
                                                     // unlike a alone, it changes when words are swapped or moved
};
static_assert(sizeof(FileHeader) == 120, "file header layout");

struct StreamSum {
    uint64_t a = 0, b = 0;
    void add(const void* p, size_t bytes)
    {
        const uint32_t* w = static_cast<const uint32_t*>(p);
        uint64_t a_ = a, b_ = b;
        for (size_t i = 0; i < bytes / 4; ++i) {
            a_ += w[i];
            b_ += a_;
        }
        a = a_;
        b = b_;
    }
};

struct FileCloser {
    FILE* f;
    ~FileCloser() { if (f) fclose(f); }
};

// tight extents of an occupied cell's record: six 5-bit fields {min x,y,z, max x,y,z}, each inside the brick, min <= max,
// nothing above them (shared by vxrt_load_world and vxrt_stream_open)
bool extents_valid(uint32_t packed, int factor)
{
    if ((packed >> 30) != 0u)
        return false;
    for (int a = 0; a < 3; ++a) {
        const uint32_t lo = (packed >> (5 * a)) & 31u, hi = (packed >> (5 * (a + 3))) & 31u;
        if (hi >= (uint32_t)factor || lo > hi)
            return false;
    }
    return true;
}

int read_header(FILE* f, const char* path, FileHeader& h)
{
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, kFileMagic, 8) != 0)
        return fail(VXRT_ERR_INVALID, std::string(path) + ": not a brickmap file");
    if (h.version != kFileVersion || h.header_bytes != sizeof(FileHeader))
        return fail(VXRT_ERR_INVALID, std::string(path) + ": unsupported brickmap file version");
    int cd[3] = {h.cdims[0], h.cdims[1], h.cdims[2]};
    int rc = vxrt::check_shape(h.factor, cd);
    if (rc)
        return rc;
    const uint64_t ncells = (uint64_t)cd[0] * cd[1] * cd[2], bw = (uint64_t)h.factor * h.factor * h.factor / 32;
    if (h.ncells != ncells || h.nslots > ncells || h.coarse_bytes != ((ncells + 31) / 32) * 4 ||
        h.meta_bytes != ncells * sizeof(uint2) || h.pool_bytes != h.nslots * bw * 4)
        return fail(VXRT_ERR_INVALID, std::string(path) + ": header sizes are inconsistent");
    return VXRT_OK;
}

}  // namespace

int vxrt_world_file_info(const char* path, vxrt_world_info* out)
{
    if (!path || !out)
        return fail(VXRT_ERR_INVALID, "NULL argument");
    FileCloser fc{fopen(path, "rb")};
    if (!fc.f)
        return fail(VXRT_ERR_INVALID, std::string(path) + ": cannot open");
    FileHeader h;
    int rc = read_header(fc.f, path, h);
    if (rc)
        return rc;
    out->factor = h.factor;
    for (int a = 0; a < 3; ++a)
        out->cdims[a] = h.cdims[a];
    out->ncells = h.ncells;
    out->nslots = h.nslots;
    out->hbm_bytes = h.coarse_bytes + h.meta_bytes + h.pool_bytes;
    return VXRT_OK;
}

int vxrt_save_world(vxrt_ctx* c, const char* path)
{
    if (!c || !path)
        return fail(VXRT_ERR_INVALID, "NULL argument");
    if (!c->has_world)
        return fail(VXRT_ERR_NO_WORLD, "no world resident");
    VX_HIP(hipSetDevice(c->device));
    VX_HIP(hipDeviceSynchronize());
    FileHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, kFileMagic, 8);
    h.version = kFileVersion;
    h.header_bytes = sizeof(FileHeader);
    h.factor = c->view.f;
    h.cdims[0] = c->view.cx;
    h.cdims[1] = c->view.cy;
    h.cdims[2] = c->view.cz;
    h.ncells = c->ncells;
    h.nslots = c->nslots;
    h.coarse_bytes = ((c->ncells + 31) / 32) * 4;
    h.meta_bytes = c->ncells * sizeof(uint2);
    h.pool_bytes = c->nslots * (uint64_t)c->view.brick_words * 4;
    FileCloser fc{fopen(path, "wb")};
    if (!fc.f)
        return fail(VXRT_ERR_INVALID, std::string(path) + ": cannot create");
    if (fwrite(&h, sizeof(h), 1, fc.f) != 1)
        return fail(VXRT_ERR_INVALID, std::string(path) + ": write failed");
    // the file holds the tables in the reference's tiled order: re-ordered on the device, then streamed out (the two
    // cell tables through whole-table temporaries, the pool through a staging buffer of kFileChunk bytes)
    std::vector<unsigned char> stage(kFileChunk);
    const int cd[3] = {c->view.cx, c->view.cy, c->view.cz};
    struct Temps {
        uint32_t *coarse = nullptr, *pool = nullptr;
        uint2* meta = nullptr;
        ~Temps() { (void)hipFree(coarse); (void)hipFree(meta); (void)hipFree(pool); }
    } T;
    VX_HIP(hipMalloc((void**)&T.coarse, h.coarse_bytes));
    VX_HIP(hipMalloc((void**)&T.meta, h.meta_bytes));
    VX_HIP(hipMalloc((void**)&T.pool, std::max<uint64_t>(4, std::min<uint64_t>(kFileChunk, h.pool_bytes))));
    VX_HIP(vxrt::layout_bits(c->d_coarse, T.coarse, cd, false));
    VX_HIP(vxrt::layout_meta(c->d_meta, T.meta, cd, false));
    const void* src[3] = {T.coarse, T.meta, nullptr};
    const uint64_t bytes[3] = {h.coarse_bytes, h.meta_bytes, h.pool_bytes};
    const uint64_t brick_bytes = (uint64_t)c->view.brick_words * 4;
    for (int t = 0; t < 3; ++t) {
        StreamSum cs;
        for (uint64_t off = 0; off < bytes[t]; off += kFileChunk) {
            const size_t n = (size_t)std::min<uint64_t>(kFileChunk, bytes[t] - off);
            if (t == 2) {  // kFileChunk is a whole number of bricks
                VX_HIP(vxrt::layout_bricks(c->d_pool + off / 4, T.pool, n / brick_bytes, c->view.f, false));
                VX_HIP(hipMemcpy(stage.data(), T.pool, n, hipMemcpyDeviceToHost));
            } else {
                VX_HIP(hipMemcpy(stage.data(), static_cast<const unsigned char*>(src[t]) + off, n, hipMemcpyDeviceToHost));
            }
            cs.add(stage.data(), n);
            if (fwrite(stage.data(), 1, n, fc.f) != n)
                return fail(VXRT_ERR_INVALID, std::string(path) + ": write failed (disk full?)");
        }
        h.sum[t] = cs.a;
        h.sum2[t] = cs.b;
    }
    if (fseek(fc.f, 0, SEEK_SET) != 0 || fwrite(&h, sizeof(h), 1, fc.f) != 1 || fflush(fc.f) != 0)
        return fail(VXRT_ERR_INVALID, std::string(path) + ": write failed");
    return VXRT_OK;
}

int vxrt_load_world(vxrt_ctx* c, const char* path)
{
    if (!c || !path)
        return fail(VXRT_ERR_INVALID, "NULL argument");
    FileCloser fc{fopen(path, "rb")};
    if (!fc.f)
        return fail(VXRT_ERR_INVALID, std::string(path) + ": cannot open");
    FileHeader h;
    int rc = read_header(fc.f, path, h);
    if (rc)
        return rc;
    VX_HIP(hipSetDevice(c->device));
    int cd[3] = {h.cdims[0], h.cdims[1], h.cdims[2]};
    rc = vxrt::alloc_world(c, h.factor, cd, h.nslots);
    if (rc)
        return rc;
    // stream the tables into HBM, checking on the way what vxrt_upload_world checks: every occupied coarse cell
    // owns a brick inside the pool, every empty one owns none
    std::vector<unsigned char> stage(kFileChunk);
    std::vector<uint32_t> coarse(h.coarse_bytes / 4);
    // the file's tables are in the reference's tiled order: the two cell tables land in device temporaries and are
    // re-ordered into the HBM order when their stream is complete, the pool piece by piece in place
    struct Temps {
        uint32_t* coarse = nullptr;
        uint2* meta = nullptr;
        ~Temps() { (void)hipFree(coarse); (void)hipFree(meta); }
    } T;
    auto bad = [&](const std::string& why) {
        vxrt::free_world(c);
        return fail(VXRT_ERR_INVALID, std::string(path) + ": " + why);
    };
    auto bad_hip = [&](hipError_t e) {
        vxrt::free_world(c);
        return fail(VXRT_ERR_HIP, std::string("world load: ") + hipGetErrorString(e));
    };
    hipError_t e = hipMalloc((void**)&T.coarse, h.coarse_bytes);
    if (e == hipSuccess)
        e = hipMalloc((void**)&T.meta, h.meta_bytes);
    if (e != hipSuccess)
        return bad_hip(e);
    void* dst[3] = {T.coarse, T.meta, c->d_pool};
    const uint64_t bytes[3] = {h.coarse_bytes, h.meta_bytes, h.pool_bytes};
    const uint64_t brick_bytes = (uint64_t)h.factor * h.factor * h.factor / 8;
    for (int t = 0; t < 3; ++t) {
        StreamSum cs;
        for (uint64_t off = 0; off < bytes[t]; off += kFileChunk) {
            const size_t n = (size_t)std::min<uint64_t>(kFileChunk, bytes[t] - off);
            if (fread(stage.data(), 1, n, fc.f) != n)
                return bad("file is truncated");
            cs.add(stage.data(), n);
            if (t == 0)
                memcpy(reinterpret_cast<unsigned char*>(coarse.data()) + off, stage.data(), n);
            if (t == 1) {
                const uint2* m = reinterpret_cast<const uint2*>(stage.data());
                const uint64_t first = off / sizeof(uint2);
                for (size_t i = 0; i < n / sizeof(uint2); ++i) {
                    const uint64_t cell = first + i;
                    const bool bit = (coarse[cell >> 5] >> (cell & 31)) & 1u;
                    if (bit ? m[i].x >= h.nslots : m[i].x != VXRT_EMPTY_SLOT)
                        return bad("cell table does not match the coarse bits / pool size");
                    if (bit && !extents_valid(m[i].y, h.factor))
                        return bad("brick extents outside the brick");
                }
            }
            e = hipMemcpy(static_cast<unsigned char*>(dst[t]) + off, stage.data(), n, hipMemcpyHostToDevice);
            if (e == hipSuccess && t == 2)  // kFileChunk is a whole number of bricks
                e = vxrt::layout_bricks(c->d_pool + off / 4, c->d_pool + off / 4, n / brick_bytes, h.factor, true);
            if (e != hipSuccess)
                return bad_hip(e);
        }
        if (cs.a != h.sum[t] || cs.b != h.sum2[t])
            return bad("checksum mismatch (corrupt file)");
        if (t == 0)
            e = vxrt::layout_bits(T.coarse, c->d_coarse, cd, true);
        if (t == 1)
            e = vxrt::layout_meta(T.meta, c->d_meta, cd, true);
        if (e != hipSuccess)
            return bad_hip(e);
    }
    e = hipDeviceSynchronize();
    if (e != hipSuccess)
        return bad_hip(e);
    c->nslots = h.nslots;
    vxrt::fill_view(c, h.factor, cd);
    c->has_world = true;
    return VXRT_OK;
}

}  // extern "C"

// ---- chunk streaming (include/vxrt.h): coarse tables of the whole world resident, brick data only near the focus --------
struct StreamState {
    FILE* f = nullptr;
    FileHeader h{};
    uint64_t pool_off = 0, brick_bytes = 0, nchunks = 0;
    std::vector<uint32_t> coarse;   // the whole world's coarse bits
    std::vector<uint2> meta;        // ... and cell records, with the FILE's slot numbers
    struct Chunk {
        uint32_t first_slot = 0, nbricks = 0;   // its run of bricks in the file
        int64_t base = -1;                      // first brick of its range in the device pool, or -1 = not resident
        float lo[3], hi[3];                     // its box in voxels
    };
    std::vector<Chunk> chunks;
    std::map<uint64_t, uint64_t> free_ranges;   // device pool: start -> length, in bricks
    uint2* d_chunk_meta = nullptr;              // device staging for one chunk's 512 cell records
    uint64_t capacity = 0, bricks_resident = 0, chunks_resident = 0, chunks_occupied = 0;

    bool alloc(uint64_t n, uint64_t& start)
    {
        for (auto it = free_ranges.begin(); it != free_ranges.end(); ++it)
            if (it->second >= n) {  // first fit
                start = it->first;
                const uint64_t rest = it->second - n, at = it->first + n;
                free_ranges.erase(it);
                if (rest)
                    free_ranges[at] = rest;
                return true;
            }
        return false;
    }
    void release(uint64_t start, uint64_t n)
    {
        auto next = free_ranges.lower_bound(start);
        if (next != free_ranges.begin()) {  // merge with the range that ends where this one starts
            auto prev = std::prev(next);
            if (prev->first + prev->second == start) {
                start = prev->first;
                n += prev->second;
                free_ranges.erase(prev);
            }
        }
        if (next != free_ranges.end() && start + n == next->first) {
            n += next->second;
            free_ranges.erase(next);
        }
        free_ranges[start] = n;
    }
};

namespace vxrt {
void stream_drop(vxrt_ctx* c)
{
    if (!c->stream)
        return;
    if (c->stream->f)
        fclose(c->stream->f);
    (void)hipFree(c->stream->d_chunk_meta);
    delete c->stream;
    c->stream = nullptr;
}
}  // namespace vxrt

extern "C" {

int vxrt_stream_open(vxrt_ctx* c, const char* path, uint64_t pool_capacity_bricks)
{
    if (!c || !path || pool_capacity_bricks == 0)
        return fail(VXRT_ERR_INVALID, "NULL argument or empty pool");
    FILE* f = fopen(path, "rb");
    if (!f)
        return fail(VXRT_ERR_INVALID, std::string(path) + ": cannot open");
    StreamState* S = new (std::nothrow) StreamState();
    if (!S) {
        fclose(f);
        return fail(VXRT_ERR_NOMEM, "out of host memory");
    }
    S->f = f;
    auto bad = [&](int code, const std::string& why) {
        fclose(S->f);
        delete S;
        return fail(code, why);
    };
    int rc = read_header(f, path, S->h);
    if (rc) {
        fclose(S->f);
        delete S;
        return rc;
    }
    const FileHeader& h = S->h;
    S->coarse.resize(h.coarse_bytes / 4);
    S->meta.resize(h.ncells);
    if (fread(S->coarse.data(), 1, h.coarse_bytes, f) != h.coarse_bytes || fread(S->meta.data(), 1, h.meta_bytes, f) != h.meta_bytes)
        return bad(VXRT_ERR_INVALID, std::string(path) + ": file is truncated");
    {   // the two table streams are checked like vxrt_load_world checks them (the pool is read chunk by chunk later)
        StreamSum a, b;
        a.add(S->coarse.data(), h.coarse_bytes);
        b.add(S->meta.data(), h.meta_bytes);
        if (a.a != h.sum[0] || a.b != h.sum2[0] || b.a != h.sum[1] || b.b != h.sum2[1])
            return bad(VXRT_ERR_INVALID, std::string(path) + ": checksum mismatch (corrupt file)");
    }
    S->pool_off = sizeof(FileHeader) + h.coarse_bytes + h.meta_bytes;
    S->brick_bytes = (uint64_t)h.factor * h.factor * h.factor / 8;
    S->nchunks = h.ncells / 512;
    S->chunks.resize(S->nchunks);
    const int tw = h.cdims[0] / 8, th = h.cdims[1] / 8;
    uint32_t next_slot = 0;
    for (uint64_t ch = 0; ch < S->nchunks; ++ch) {
        StreamState::Chunk& C = S->chunks[ch];
        C.first_slot = next_slot;
        for (uint64_t i = ch * 512; i < ch * 512 + 512; ++i) {
            const bool bit = (S->coarse[i >> 5] >> (i & 31)) & 1u;
            // slots run through the file in cell order (vxrt_save_world writes what the builders produce): a chunk's
            // bricks are ONE contiguous run
            if (bit ? S->meta[i].x != next_slot : S->meta[i].x != VXRT_EMPTY_SLOT)
                return bad(VXRT_ERR_INVALID, std::string(path) + ": brick slots are not in cell order");
            if (bit && !extents_valid(S->meta[i].y, h.factor))  // (what vxrt_load_world checks of a cell record)
                return bad(VXRT_ERR_INVALID, std::string(path) + ": brick extents outside the brick");
            if (bit) {
                ++next_slot;
                ++C.nbricks;
            }
        }
        const int tx = (int)(ch % tw), ty = (int)((ch / tw) % th), tz = (int)(ch / ((uint64_t)tw * th));
        const float e = 8.0f * (float)h.factor;
        C.lo[0] = tx * e; C.lo[1] = ty * e; C.lo[2] = tz * e;
        C.hi[0] = C.lo[0] + e; C.hi[1] = C.lo[1] + e; C.hi[2] = C.lo[2] + e;
        if (C.nbricks)
            S->chunks_occupied += 1;
    }
    if (next_slot != h.nslots)
        return bad(VXRT_ERR_INVALID, std::string(path) + ": brick count does not match the coarse bits");
    // device: the whole world's tables (all empty for now) + a pool of the requested capacity
    int cd[3] = {h.cdims[0], h.cdims[1], h.cdims[2]};
    if (hipSetDevice(c->device) != hipSuccess)
        return bad(VXRT_ERR_HIP, "hipSetDevice");
    (void)hipDeviceSynchronize();
    rc = vxrt::alloc_world(c, h.factor, cd, pool_capacity_bricks);  // (frees a previous world, streamed or not)
    if (rc) {
        fclose(S->f);
        delete S;
        return rc;
    }
    std::vector<uint2> empty(h.ncells, make_uint2(VXRT_EMPTY_SLOT, 0u));
    hipError_t e = hipMemset(c->d_coarse, 0, h.coarse_bytes);
    if (e == hipSuccess)  // the cache pool starts as empty space, not as whatever the allocation held
        e = hipMemset(c->d_pool, 0, pool_capacity_bricks * S->brick_bytes);
    if (e == hipSuccess)
        e = hipMemcpy(c->d_meta, empty.data(), h.meta_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMalloc((void**)&S->d_chunk_meta, 512 * sizeof(uint2));
    if (e != hipSuccess) {
        vxrt::free_world(c);
        return bad(VXRT_ERR_HIP, std::string("stream tables: ") + hipGetErrorString(e));
    }
    S->capacity = pool_capacity_bricks;
    S->free_ranges[0] = pool_capacity_bricks;
    c->nslots = pool_capacity_bricks;
    vxrt::fill_view(c, h.factor, cd);
    c->has_world = true;
    c->stream = S;
    return VXRT_OK;
}

int vxrt_stream_focus(vxrt_ctx* c, const float focus[3], float radius, vxrt_stream_stats* out)
{
    if (!c || !focus || !(radius >= 0.0f))
        return fail(VXRT_ERR_INVALID, "NULL argument or negative radius");
    StreamState* S = c->stream;
    if (!S)
        return fail(VXRT_ERR_NO_WORLD, "no streamed world (vxrt_stream_open)");
    VX_HIP(hipSetDevice(c->device));
    VX_HIP(hipDeviceSynchronize());  // no launch may read the tables while chunks come and go
    // squared distance of the focus to every occupied chunk's box
    std::vector<std::pair<float, uint32_t>> order;
    order.reserve(S->chunks_occupied);
    for (uint64_t ch = 0; ch < S->nchunks; ++ch) {
        const StreamState::Chunk& C = S->chunks[ch];
        if (!C.nbricks)
            continue;
        float d2 = 0.0f;
        for (int a = 0; a < 3; ++a) {
            const float d = focus[a] < C.lo[a] ? C.lo[a] - focus[a] : (focus[a] > C.hi[a] ? focus[a] - C.hi[a] : 0.0f);
            d2 += d * d;
        }
        order.emplace_back(d2, (uint32_t)ch);
    }
    std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    const float r2 = radius * radius;
    uint64_t loaded = 0, evicted = 0, missing = 0, bytes = 0;
    size_t far = order.size();  // eviction candidates: from the far end of the order, outside the radius only
    std::vector<unsigned char> stage;
    std::vector<uint2> meta(512);
    auto write_tables = [&](uint32_t ch, bool resident) -> hipError_t {
        const StreamState::Chunk& C = S->chunks[ch];
        uint32_t k = 0;
        for (uint64_t i = 0; i < 512; ++i) {
            const uint2 m = S->meta[(uint64_t)ch * 512 + i];
            meta[i] = (resident && m.x != VXRT_EMPTY_SLOT) ? make_uint2((uint32_t)(C.base + k++), m.y) : make_uint2(VXRT_EMPTY_SLOT, 0u);
        }
        // the chunk's 512 records (file order: the reference's tiled order, one chunk = one tile) go to their places in
        // the HBM tables -- 64 rows of 8 cells -- with their coarse bits
        hipError_t e = hipMemcpy(S->d_chunk_meta, meta.data(), 512 * sizeof(uint2), hipMemcpyHostToDevice);
        if (e != hipSuccess)
            return e;
        const int tw = S->h.cdims[0] / 8, th = S->h.cdims[1] / 8;
        return vxrt::chunk_tables(c->d_meta, c->d_coarse, S->d_chunk_meta, (int)(ch % (uint32_t)tw), (int)((ch / (uint32_t)tw) % (uint32_t)th),
                                  (int)(ch / ((uint32_t)tw * (uint32_t)th)), S->h.cdims[0], S->h.cdims[2]);
    };
    for (size_t k = 0; k < order.size() && order[k].first <= r2; ++k) {
        const uint32_t ch = order[k].second;
        StreamState::Chunk& C = S->chunks[ch];
        if (C.base >= 0)
            continue;
        uint64_t start = 0;
        bool ok = S->alloc(C.nbricks, start);
        while (!ok && far > 0) {  // make room: the farthest resident chunk outside the radius goes
            --far;
            if (order[far].first <= r2)
                break;
            StreamState::Chunk& V = S->chunks[order[far].second];
            if (V.base < 0)
                continue;
            VX_HIP(write_tables(order[far].second, false));
            S->release((uint64_t)V.base, V.nbricks);
            S->bricks_resident -= V.nbricks;
            S->chunks_resident -= 1;
            V.base = -1;
            evicted += 1;
            ok = S->alloc(C.nbricks, start);
        }
        if (!ok) {
            missing += 1;
            continue;
        }
        const uint64_t nbytes = (uint64_t)C.nbricks * S->brick_bytes;
        stage.resize(nbytes);
        if (fseek(S->f, (long)(S->pool_off + (uint64_t)C.first_slot * S->brick_bytes), SEEK_SET) != 0 ||
            fread(stage.data(), 1, nbytes, S->f) != nbytes) {
            S->release(start, C.nbricks);
            return fail(VXRT_ERR_INVALID, "brickmap file: chunk read failed");
        }
        bytes += nbytes;
        hipError_t e = hipMemcpy(reinterpret_cast<unsigned char*>(c->d_pool) + start * S->brick_bytes, stage.data(), nbytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) {  // the file's bricks are in the reference's tiled bit order: into the HBM order, in place
            uint32_t* at = c->d_pool + start * (S->brick_bytes / 4);
            e = vxrt::layout_bricks(at, at, C.nbricks, S->h.factor, true);
        }
        if (e != hipSuccess) {
            S->release(start, C.nbricks);
            return fail(VXRT_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
        }
        C.base = (int64_t)start;
        VX_HIP(write_tables(ch, true));
        S->bricks_resident += C.nbricks;
        S->chunks_resident += 1;
        loaded += 1;
    }
    VX_HIP(hipDeviceSynchronize());  // the table updates and re-ordered bricks are in place before any launch reads them
    if (out) {
        out->chunks_total = S->nchunks;
        out->chunks_occupied = S->chunks_occupied;
        out->chunks_resident = S->chunks_resident;
        out->bricks_resident = S->bricks_resident;
        out->chunks_loaded = loaded;
        out->chunks_evicted = evicted;
        out->chunks_missing = missing;
        out->bytes_read = bytes;
    }
    return VXRT_OK;
}

int vxrt_stream_resident(vxrt_ctx* c, uint8_t* flags, uint64_t n_chunks)
{
    if (!c || !flags)
        return fail(VXRT_ERR_INVALID, "NULL argument");
    if (!c->stream)
        return fail(VXRT_ERR_NO_WORLD, "no streamed world (vxrt_stream_open)");
    if (n_chunks != c->stream->nchunks)
        return fail(VXRT_ERR_INVALID, "one flag per 8x8x8 tile of the coarse grid");
    for (uint64_t ch = 0; ch < n_chunks; ++ch)
        flags[ch] = c->stream->chunks[ch].base >= 0 ? 1 : 0;
    return VXRT_OK;
}

int vxrt_stream_close(vxrt_ctx* c)
{
    if (!c)
        return fail(VXRT_ERR_INVALID, "ctx is NULL");
    if (!c->stream)
        return VXRT_OK;
    VX_HIP(hipSetDevice(c->device));
    VX_HIP(hipDeviceSynchronize());
    vxrt::free_world(c);  // drops the stream state with the tables
    return VXRT_OK;
}

}  // extern "C"
