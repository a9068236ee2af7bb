// vxrt_worldgen.hip -- on-device brickmap construction for procedural worlds.
//
// Replaces CreateVoxels + PopulateVoxels (VoxelRT/VoxelWorldBuilder.cuh:12-32, .cu:10-35: one thread
// per voxel, one atomic RMW per bit into a dense X*Y*Z bit array copied to the host) followed by the
// host-threaded GenerateLowresVoxelBuffer (VoxelRT/VolumeRaytracer.cuh:379-516).  Here one 256-thread
// workgroup owns one brick: every lane evaluates whole 32-bit words of the brick's tiled-linear bit
// image (no atomics on bits, no dense intermediate, 64-bit addressing), extents are reduced through LDS,
// and a second pass packs the non-empty bricks into the pool in coarse tiled-index order.
#include "../../include/vxrt.h"
#include "vxrt_kernels.hpp"

#include <string>
#include <vector>

struct vxrt_ctx;

namespace vxrt {

int check_shape(int factor, const int cd[3]);
int alloc_world(vxrt_ctx* c, int factor, const int cd[3], uint64_t pool_slots);
void fill_view(vxrt_ctx* c, int factor, const int cd[3]);

// ---- generators (must match oracle/vxo_world.c bit for bit; all float ops are exactly specified) ----

__device__ __forceinline__ uint32_t g_hash32(uint32_t s)  // cuda_noise.cuh:44-54
{
    s = (s + 0x7ed55d16u) + (s << 12);
    s = (s ^ 0xc761c23cu) ^ (s >> 19);
    s = (s + 0x165667b1u) + (s << 5);
    s = (s + 0xd3a2646cu) ^ (s << 9);
    s = (s + 0xfd7046c5u) + (s << 3);
    s = (s ^ 0xb55a4f09u) ^ (s >> 16);
    return s;
}
__device__ __forceinline__ uint32_t g_hash2(uint32_t a, uint32_t b, uint32_t seed)
{
    return g_hash32(a * 73856093u ^ b * 19349663u ^ seed);
}
// saturating float -> u32 (what the GPU conversion does; made explicit so host and device agree)
__device__ __forceinline__ uint32_t g_sat_u32(float v)
{
    if (!(v > 0.0f))
        return 0u;
    if (v >= 4294967296.0f)
        return 0xFFFFFFFFu;
    return (uint32_t)v;
}
__device__ __forceinline__ uint32_t g_lattice(float x, float y, float z, float seed)  // randomIntGrid, cuda_noise.cuh:118-121
{
    return g_hash32(g_sat_u32(x * 1723.0f + y * 93241.0f + z * 149812.0f + 3824.0f + seed));
}
__device__ __forceinline__ float g_grad(uint32_t h, float x, float y, float z)  // cuda_noise.cuh:174-196
{
    switch (h & 0xFu) {
    case 0x0: return x + y;
    case 0x1: return -x + y;
    case 0x2: return x - y;
    case 0x3: return -x - y;
    case 0x4: return x + z;
    case 0x5: return -x + z;
    case 0x6: return x - z;
    case 0x7: return -x - z;
    case 0x8: return y + z;
    case 0x9: return -y + z;
    case 0xA: return y - z;
    case 0xB: return -y - z;
    case 0xC: return y + x;
    case 0xD: return -y + z;
    case 0xE: return y - x;
    default: return -y - z;
    }
}
__device__ __forceinline__ float g_fade(float t) { return t * t * t * (t * (t * 6.0f - 15.0f) + 10.0f); }
__device__ __forceinline__ float g_mix(float a, float b, float r) { return a * (1.0f - r) + b * r; }

__device__ float g_perlin3(float px, float py, float pz, int seed)  // perlinNoise, cuda_noise.cuh:565-607
{
    float fseed = (float)seed;
    float ix = floorf(px), iy = floorf(py), iz = floorf(pz);
    px -= ix;
    py -= iy;
    pz -= iz;
    float u = g_fade(px), v = g_fade(py), w = g_fade(pz);
    float g000 = g_grad(g_lattice(ix, iy, iz, fseed), px, py, pz);
    float g100 = g_grad(g_lattice(ix + 1.0f, iy, iz, fseed), px - 1.0f, py, pz);
    float g010 = g_grad(g_lattice(ix, iy + 1.0f, iz, fseed), px, py - 1.0f, pz);
    float g110 = g_grad(g_lattice(ix + 1.0f, iy + 1.0f, iz, fseed), px - 1.0f, py - 1.0f, pz);
    float g001 = g_grad(g_lattice(ix, iy, iz + 1.0f, fseed), px, py, pz - 1.0f);
    float g101 = g_grad(g_lattice(ix + 1.0f, iy, iz + 1.0f, fseed), px - 1.0f, py, pz - 1.0f);
    float g011 = g_grad(g_lattice(ix, iy + 1.0f, iz + 1.0f, fseed), px, py - 1.0f, pz - 1.0f);
    float g111 = g_grad(g_lattice(ix + 1.0f, iy + 1.0f, iz + 1.0f, fseed), px - 1.0f, py - 1.0f, pz - 1.0f);
    float x00 = g_mix(g000, g100, u), x10 = g_mix(g010, g110, u);
    float x01 = g_mix(g001, g101, u), x11 = g_mix(g011, g111, u);
    return g_mix(g_mix(x00, x10, v), g_mix(x01, x11, v), w);
}

__device__ float g_fbm(float x, float y, float z)  // repeaterPerlin(pos,1,_,32,2,0.5), cuda_noise.cuh:612-628
{
    float acc = 0.0f, amp = 1.0f, scale = 1.0f;
    for (int i = 0; i < 32; ++i) {
        int seed = (int)((uint32_t)(i + 38) * 27389482u);
        // `scale` is an exact power of two, so once all three scaled coordinates are integers they stay integers
        // in every later octave; there every fraction and fade weight is 0 and the octave adds exactly +-0.
        // Stopping here gives the bits of the full 32-octave sum (oracle/vxo_world.c runs all 32) at about half
        // the work for coordinates in the thousands.
        const float sx = x * scale, sy = y * scale, sz = z * scale;
        if (sx == floorf(sx) && sy == floorf(sy) && sz == floorf(sz))
            break;
        acc += g_perlin3(sx, sy, sz, seed) * amp;
        scale *= 2.0f;
        amp *= 0.5f;
    }
    return acc;
}

template <int GEN>
__device__ __forceinline__ bool g_solid(int x, int y, int z, int Y)
{
    if (GEN == VXRT_GEN_HASH_HEIGHTFIELD) {
        uint32_t base = (uint32_t)(3 * Y / 16), range = (uint32_t)(3 * Y / 8);
        if (range == 0)
            range = 1;
        uint32_t h = base + g_hash2((uint32_t)x >> 3, (uint32_t)z >> 3, 1u) % range;
        return (uint32_t)y < h;
    } else if (GEN == VXRT_GEN_PERLIN_REF) {  // PopulateVoxels, VoxelWorldBuilder.cu:19-33
        const float scale = 0.005f;
        float t = g_fbm((float)x * scale, (float)y * scale, (float)z * scale) * 1000.0f;
        t = t > 0.0f ? t : 0.0f;
        return !((float)y > t);
    } else {
        unsigned long long h = (unsigned long long)Y / 8u;
        for (int o = 0; o < 4; ++o) {
            int k = 8 - o;
            unsigned long long S = 1ull << k;
            unsigned long long amp = ((unsigned long long)Y / 2u) >> o;
            if (amp == 0)
                break;
            uint32_t cx = (uint32_t)x >> k, cz = (uint32_t)z >> k;
            unsigned long long fx = (unsigned long long)x & (S - 1), fz = (unsigned long long)z & (S - 1);
            unsigned long long v00 = g_hash2(cx, cz, 7u + (uint32_t)o) % amp;
            unsigned long long v10 = g_hash2(cx + 1u, cz, 7u + (uint32_t)o) % amp;
            unsigned long long v01 = g_hash2(cx, cz + 1u, 7u + (uint32_t)o) % amp;
            unsigned long long v11 = g_hash2(cx + 1u, cz + 1u, 7u + (uint32_t)o) % amp;
            unsigned long long top = v00 * (S - fx) + v10 * fx;
            unsigned long long bot = v01 * (S - fx) + v11 * fx;
            h += (top * (S - fz) + bot * fz) >> (2 * k);
        }
        return (unsigned long long)y < h;
    }
}

// For the column generators (solid depends on x,z through a height only) a brick's word needs one height
// per (x,z); the generic path below simply evaluates every voxel, which is what PERLIN_REF (a true 3-D
// field) requires anyway.

// one workgroup per brick cell, cells enumerated in the reference's tiled order (slot numbers follow that order: it is
// the order of the brickmap file, where the bricks of an 8x8x8-cell chunk are one contiguous run); writes the brick's
// bit image -- in HBM order, x-fastest linear -- to scratch[cell], its packed extents to ext[cell] and any[cell]
template <int GEN>
__global__ __launch_bounds__(256) void k_fill_bricks(uint32_t* __restrict__ scratch, uint32_t* __restrict__ ext,
                                                     uint8_t* __restrict__ any, int ctw, int cth, int f, int Y,
                                                     uint32_t ncells)
{
    __shared__ int red[7];  // min xyz, max xyz, any
    const uint32_t cell = blockIdx.x + blockIdx.y * gridDim.x;  // 2-D grid: more cells than one grid axis holds
    if (cell >= ncells)
        return;
    int bx, by, bz;  // inverse of the tiled index for the brick cell (GetPositionFromSampleIndex, VolumeRaytracer.cuh:138-171)
    ref_tiled_cell(cell, ctw, cth, bx, by, bz);
    if (threadIdx.x < 3)
        red[threadIdx.x] = 0x7FFFFFFF;
    else if (threadIdx.x < 6)
        red[threadIdx.x] = -1;
    else if (threadIdx.x == 6)
        red[6] = 0;
    __syncthreads();

    // One voxel per lane: a wave's 64 consecutive bits are 64 / f whole x-rows of the brick (HBM order: x, then z, then y), and the ballot mask IS
    // that uint64 of the bit image.  Every lane is busy for every brick edge (f = 8: 512 voxels on 256 threads; the
    // earlier one-word-per-lane form left 240 of them idle there).
    const int fshift = f == 32 ? 5 : (f == 16 ? 4 : 3);
    const uint32_t words = (uint32_t)(f * f * f) >> 5, nbits = words << 5;
    int mnx = 0x7FFFFFFF, mny = 0x7FFFFFFF, mnz = 0x7FFFFFFF, mxx = -1, mxy = -1, mxz = -1;
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(scratch + (size_t)cell * words);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t o = threadIdx.x; o < nbits; o += blockDim.x) {  // nbits is a multiple of 512: whole waves iterate
        const int lx = (int)(o & (uint32_t)(f - 1)), lz = (int)((o >> fshift) & (uint32_t)(f - 1)), ly = (int)(o >> (2 * fshift));  // HBM order: x, z, y
        const bool solid = g_solid<GEN>(bx * f + lx, by * f + ly, bz * f + lz, Y);
        const unsigned long long mask = __ballot(solid);
        if (lane == 0)
            dst[o >> 6] = mask;
        if (solid) {
            mnx = min(mnx, lx); mny = min(mny, ly); mnz = min(mnz, lz);
            mxx = max(mxx, lx); mxy = max(mxy, ly); mxz = max(mxz, lz);
        }
    }
    if (mxx >= 0) {
        atomicMin(&red[0], mnx); atomicMin(&red[1], mny); atomicMin(&red[2], mnz);
        atomicMax(&red[3], mxx); atomicMax(&red[4], mxy); atomicMax(&red[5], mxz);
        red[6] = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t packed = 0;
        if (red[6])
            packed = (uint32_t)red[0] | ((uint32_t)red[1] << 5) | ((uint32_t)red[2] << 10) | ((uint32_t)red[3] << 15) |
                     ((uint32_t)red[4] << 20) | ((uint32_t)red[5] << 25);
        ext[cell] = packed;
        any[cell] = (uint8_t)red[6];
    }
}

// pack non-empty bricks into the pool, 16 bytes per lane, and write the cell_meta records (at the cell's HBM index)
__global__ __launch_bounds__(256) void k_pack_bricks(const uint4* __restrict__ scratch, const uint32_t* __restrict__ slot,
                                                     const uint32_t* __restrict__ ext, uint4* __restrict__ pool,
                                                     uint2* __restrict__ meta, uint32_t vecs_per_brick, uint32_t ncells,
                                                     int cx, int cy, int cz)
{
    const uint32_t cell = blockIdx.x + blockIdx.y * gridDim.x;
    if (cell >= ncells)
        return;
    const uint32_t s = slot[cell];
    if (threadIdx.x == 0) {
        int bx, by, bz;
        ref_tiled_cell(cell, cx / 8, cy / 8, bx, by, bz);
        meta[hbm_index(bx, by, bz, cx, cz)] = make_uint2(s, ext[cell]);
    }
    if (s == kEmptySlot)
        return;
    const uint4* src = scratch + (size_t)cell * vecs_per_brick;
    uint4* dst = pool + (size_t)s * vecs_per_brick;
    for (uint32_t i = threadIdx.x; i < vecs_per_brick; i += blockDim.x)
        dst[i] = src[i];
}

// coarse bit = brick non-empty (VolumeRaytracer.cuh:504-507), one word of the HBM order per lane; any[] is in the
// builder's tiled cell order
__global__ void k_coarse_bits(const uint8_t* __restrict__ any, uint32_t* __restrict__ coarse, int cx, int cy, int cz)
{
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t ncells = (uint64_t)cx * cy * cz, nwords = (ncells + 31) / 32;
    if (w >= nwords)
        return;
    uint32_t bits = 0;
    for (uint32_t k = 0; k < 32u; ++k) {
        const uint64_t i = w * 32u + k;
        if (i >= ncells)
            break;
        int x, y, z;
        hbm_cell(i, cx, cz, x, y, z);
        if (any[ref_tiled_index(x, y, z, cx / 8, cy / 8)])
            bits |= 1u << k;
    }
    coarse[w] = bits;
}

// ---- re-ordering between the reference's tiled bit order (C ABI, brickmap file) and the HBM order ---------------------
// TO_HBM: src is tiled, dst linear; else the inverse.  One destination word per lane, 32 gathered bits.
template <bool TO_HBM>
__global__ void k_layout_bits(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int dx, int dy, int dz)
{
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = (uint64_t)dx * dy * dz;
    if (w * 32u >= n)
        return;
    uint32_t bits = 0;
    for (uint32_t k = 0; k < 32u; ++k) {
        const uint64_t i = w * 32u + k;
        if (i >= n)
            break;
        uint64_t from;
        if (TO_HBM) {
            int x, y, z;
            hbm_cell(i, dx, dz, x, y, z);
            from = ref_tiled_index(x, y, z, dx / 8, dy / 8);
        } else {
            int x, y, z;
            ref_tiled_cell((uint32_t)i, dx / 8, dy / 8, x, y, z);
            from = hbm_index(x, y, z, dx, dz);
        }
        bits |= ((src[from >> 5] >> (from & 31u)) & 1u) << k;
    }
    dst[w] = bits;
}

template <bool TO_HBM>
__global__ void k_layout_meta(const uint2* __restrict__ src, uint2* __restrict__ dst, int dx, int dy, int dz)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)dx * dy * dz)
        return;
    int x, y, z;
    if (TO_HBM) {
        hbm_cell(i, dx, dz, x, y, z);
        dst[i] = src[ref_tiled_index(x, y, z, dx / 8, dy / 8)];
    } else {
        ref_tiled_cell((uint32_t)i, dx / 8, dy / 8, x, y, z);
        dst[i] = src[hbm_index(x, y, z, dx, dz)];
    }
}

// one workgroup per brick, staged through LDS: src and dst may be the same brick (in place)
template <bool TO_HBM>
__global__ __launch_bounds__(256) void k_layout_bricks(const uint32_t* src, uint32_t* dst, int f)
{
    __shared__ uint32_t old_words[1024];  // f <= 32
    const uint32_t words = (uint32_t)(f * f * f) / 32u;
    const uint32_t* in = src + (size_t)blockIdx.x * words;
    uint32_t* out = dst + (size_t)blockIdx.x * words;
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x)
        old_words[i] = in[i];
    __syncthreads();
    for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) {
        uint32_t bits = 0;
        for (uint32_t k = 0; k < 32u; ++k) {
            const uint32_t i = w * 32u + k;
            uint32_t from;
            if (TO_HBM) {
                int x, y, z;
                hbm_cell(i, f, f, x, y, z);
                from = ref_tiled_index(x, y, z, f / 8, f / 8);
            } else {
                int x, y, z;
                ref_tiled_cell(i, f / 8, f / 8, x, y, z);
                from = (uint32_t)hbm_index(x, y, z, f, f);
            }
            bits |= ((old_words[from >> 5] >> (from & 31u)) & 1u) << k;
        }
        out[w] = bits;
    }
}

// chunk streaming: the cell records and coarse bits of ONE 8x8x8-cell chunk (chunk_meta: its 512 records in the file's
// tiled order, slots already translated) go to their places in the HBM tables; 8 cells of an x-row share a coarse word
__global__ __launch_bounds__(512) void k_chunk_tables(uint2* __restrict__ meta, uint32_t* __restrict__ coarse,
                                                      const uint2* __restrict__ chunk_meta, int tx, int ty, int tz, int cx, int cz)
{
    const uint32_t i = threadIdx.x;
    const int x = tx * 8 + (int)(i & 7u), y = ty * 8 + (int)((i >> 3) & 7u), z = tz * 8 + (int)(i >> 6);
    const uint64_t at = hbm_index(x, y, z, cx, cz);
    const uint2 m = chunk_meta[i];
    meta[at] = m;
    if (m.x != kEmptySlot)
        atomicOr(&coarse[at >> 5], 1u << (at & 31u));
    else
        atomicAnd(&coarse[at >> 5], ~(1u << (at & 31u)));
}

// host entry points of the above (vxrt_api.hip)
hipError_t layout_bits(const uint32_t* src, uint32_t* dst, const int cd[3], bool to_hbm)
{
    const uint64_t nwords = ((uint64_t)cd[0] * cd[1] * cd[2] + 31) / 32;
    const dim3 g((unsigned)((nwords + 255) / 256)), b(256);
    if (to_hbm)
        hipLaunchKernelGGL(k_layout_bits<true>, g, b, 0, 0, src, dst, cd[0], cd[1], cd[2]);
    else
        hipLaunchKernelGGL(k_layout_bits<false>, g, b, 0, 0, src, dst, cd[0], cd[1], cd[2]);
    return hipGetLastError();
}
hipError_t layout_meta(const uint2* src, uint2* dst, const int cd[3], bool to_hbm)
{
    const uint64_t n = (uint64_t)cd[0] * cd[1] * cd[2];
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    if (to_hbm)
        hipLaunchKernelGGL(k_layout_meta<true>, g, b, 0, 0, src, dst, cd[0], cd[1], cd[2]);
    else
        hipLaunchKernelGGL(k_layout_meta<false>, g, b, 0, 0, src, dst, cd[0], cd[1], cd[2]);
    return hipGetLastError();
}
hipError_t layout_bricks(const uint32_t* src, uint32_t* dst, uint64_t nbricks, int f, bool to_hbm)
{
    const uint32_t words = (uint32_t)(f * f * f) / 32u;
    for (uint64_t done = 0; done < nbricks;) {  // a grid axis holds 2^31 - 1 workgroups
        const uint64_t n = nbricks - done < (1ull << 30) ? nbricks - done : (1ull << 30);
        if (to_hbm)
            hipLaunchKernelGGL(k_layout_bricks<true>, dim3((unsigned)n), dim3(256), 0, 0, src + done * words, dst + done * words, f);
        else
            hipLaunchKernelGGL(k_layout_bricks<false>, dim3((unsigned)n), dim3(256), 0, 0, src + done * words, dst + done * words, f);
        done += n;
    }
    return hipGetLastError();
}
hipError_t chunk_tables(uint2* meta, uint32_t* coarse, const uint2* d_chunk_meta, int tx, int ty, int tz, int cx, int cz)
{
    hipLaunchKernelGGL(k_chunk_tables, dim3(1), dim3(512), 0, 0, meta, coarse, d_chunk_meta, tx, ty, tz, cx, cz);
    return hipGetLastError();
}

}  // namespace vxrt

namespace vxrt {

// accessors into the opaque context (defined in vxrt_api.hip)
int adopt_world(vxrt_ctx* c, int factor, const int cd[3], uint64_t nslots, uint32_t** d_coarse, uint2** d_meta,
                uint32_t** d_pool);
int set_error(int code, const char* msg);
void abandon_world(vxrt_ctx* c);  // frees the tables of a world whose build failed half way

#define WG_HIP(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            cleanup();                                                                             \
            return set_error(VXRT_ERR_HIP, (std::string(#call) + ": " + hipGetErrorString(e_)).c_str()); \
        }                                                                                          \
    } while (0)

int build_world_on_device(vxrt_ctx* c, int generator, int X, int Y, int Z, int factor)
{
    if (generator < 0 || generator > 2)
        return set_error(VXRT_ERR_INVALID, "unknown generator");
    if (factor <= 0 || X <= 0 || Y <= 0 || Z <= 0 || X % factor || Y % factor || Z % factor)
        return set_error(VXRT_ERR_INVALID, "world size must be a multiple of the brick edge");
    int cd[3] = {X / factor, Y / factor, Z / factor};
    int rc = check_shape(factor, cd);
    if (rc)
        return rc;
    const uint64_t ncells = (uint64_t)cd[0] * cd[1] * cd[2];
    const uint64_t bw = (uint64_t)factor * factor * factor / 32;
    uint32_t *d_scratch = nullptr, *d_ext = nullptr, *d_slot = nullptr;
    uint8_t* d_any = nullptr;
    bool adopted = false, done = false;
    auto cleanup = [&]() {
        (void)hipFree(d_scratch); (void)hipFree(d_ext); (void)hipFree(d_slot); (void)hipFree(d_any);
        if (adopted && !done)
            abandon_world(c);  // a failed pack / coarse-bit pass must not leave a half-built world flagged resident
    };
    WG_HIP(hipMalloc((void**)&d_scratch, ncells * bw * sizeof(uint32_t)));
    WG_HIP(hipMalloc((void**)&d_ext, ncells * sizeof(uint32_t)));
    WG_HIP(hipMalloc((void**)&d_any, ncells));
    const unsigned gx = ncells > (1u << 20) ? (1u << 20) : (unsigned)ncells;
    dim3 grid(gx, (unsigned)((ncells + gx - 1) / gx)), block(256);
    if (generator == VXRT_GEN_HASH_HEIGHTFIELD)
        hipLaunchKernelGGL(k_fill_bricks<VXRT_GEN_HASH_HEIGHTFIELD>, grid, block, 0, 0, d_scratch, d_ext, d_any, cd[0] / 8,
                           cd[1] / 8, factor, Y, (uint32_t)ncells);
    else if (generator == VXRT_GEN_PERLIN_REF)
        hipLaunchKernelGGL(k_fill_bricks<VXRT_GEN_PERLIN_REF>, grid, block, 0, 0, d_scratch, d_ext, d_any, cd[0] / 8,
                           cd[1] / 8, factor, Y, (uint32_t)ncells);
    else
        hipLaunchKernelGGL(k_fill_bricks<VXRT_GEN_INT_TERRAIN>, grid, block, 0, 0, d_scratch, d_ext, d_any, cd[0] / 8,
                           cd[1] / 8, factor, Y, (uint32_t)ncells);
    WG_HIP(hipGetLastError());
    WG_HIP(hipDeviceSynchronize());

    // slot numbers in the reference's tiled cell order (host scan of one byte per cell)
    std::vector<uint8_t> any(ncells);
    WG_HIP(hipMemcpy(any.data(), d_any, ncells, hipMemcpyDeviceToHost));
    std::vector<uint32_t> slot(ncells);
    uint64_t nslots = 0;
    for (uint64_t i = 0; i < ncells; ++i)
        slot[i] = any[i] ? (uint32_t)nslots++ : kEmptySlot;
    WG_HIP(hipMalloc((void**)&d_slot, ncells * sizeof(uint32_t)));
    WG_HIP(hipMemcpy(d_slot, slot.data(), ncells * sizeof(uint32_t), hipMemcpyHostToDevice));

    uint32_t* d_coarse = nullptr;
    uint2* d_meta = nullptr;
    uint32_t* d_pool = nullptr;
    rc = adopt_world(c, factor, cd, nslots, &d_coarse, &d_meta, &d_pool);
    if (rc) {
        cleanup();
        return rc;
    }
    adopted = true;
    hipLaunchKernelGGL(k_pack_bricks, grid, block, 0, 0, (const uint4*)d_scratch, d_slot, d_ext, (uint4*)d_pool, d_meta,
                       (uint32_t)(bw / 4), (uint32_t)ncells, cd[0], cd[1], cd[2]);
    WG_HIP(hipGetLastError());
    uint64_t nwords = (ncells + 31) / 32;
    hipLaunchKernelGGL(k_coarse_bits, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, 0, d_any, d_coarse, cd[0], cd[1], cd[2]);
    WG_HIP(hipGetLastError());
    WG_HIP(hipDeviceSynchronize());
    done = true;
    cleanup();
    return VXRT_OK;
}

}  // namespace vxrt
