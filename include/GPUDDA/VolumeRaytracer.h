// GPUDDA/VolumeRaytracer.h -- the reference's VoxelRT/VolumeRaytracer.cuh API surface, re-implemented for
// MI355X as a thin C++ facade over the C ABI in include/vxrt.h (libvxrt.so).  Same namespace, type and member
// names and argument meaning as the reference so VoxelApp-style callers compile against it; nothing here is
// CUDA and no reference source is reused.  float3/uint3 come from <hip/hip_vector_types.h>.
//
// Deliberate differences: device tables are one flat pool instead of one allocation per brick
// (VolumeRaytracer.cu:552-565); Get*() accessors return opaque handles; Raytrace() sizes its result buffers from
// the request instead of the constructor count (VolumeRaytracer.cuh:318-331); voxelIndex is the global voxel
// index of the hit (x + X*(y + Y*z), -1 on a miss or if it does not fit an int) instead of the reference's
// float expression on coarse dimensions (VolumeRaytracer.cu:611-612).
#pragma once

#include <hip/hip_vector_types.h>

#include <cstddef>
#include <cstdint>
#include <ostream>
#include <limits>
#include <memory>
#include <tuple>
#include <vector>

struct vxrt_ctx;

constexpr auto FLT_EPS_DDA = 1e-6;  // VolumeRaytracer.cuh:20 (a double)
constexpr auto FLT_INF = std::numeric_limits<float>::infinity();
constexpr auto FLT_EPS = std::numeric_limits<float>::epsilon();

namespace GPUDDA {

// 8x8x8 tiled-linear bit address and its inverse (VolumeRaytracer.cuh:107-171)
uint32_t GetSampleIndex(uint32_t x, uint32_t y, uint32_t z, uint32_t width, uint32_t height);
void GetPositionFromSampleIndex(uint32_t index, uint32_t width, uint32_t height, uint32_t& x, uint32_t& y, uint32_t& z);

template <class T>
struct Bounds {
    T min;
    T max;
};

template <typename T>
class RayTraceResults {  // VolumeRaytracer.cuh:179-202
public:
    std::shared_ptr<bool[]> valid{};
    std::shared_ptr<T[]> hitPoint{};
    std::shared_ptr<T[]> normal{};
    std::shared_ptr<float[]> distance{};
    std::shared_ptr<int[]> voxelIndex{};
    std::shared_ptr<int[]> steps{};
    explicit RayTraceResults(size_t count)
    {
        if (count == 0)
            return;
        valid = std::shared_ptr<bool[]>(new bool[count]());
        hitPoint = std::shared_ptr<T[]>(new T[count]());
        normal = std::shared_ptr<T[]>(new T[count]());
        distance = std::shared_ptr<float[]>(new float[count]());
        voxelIndex = std::shared_ptr<int[]>(new int[count]());
        steps = std::shared_ptr<int[]>(new int[count]());
    }
};

struct BitRef {  // VolumeRaytracer.cuh:204-209; writes are atomic RMW like the reference's
    uint32_t* byte = nullptr;
    size_t index = 0;
    operator bool() const;
    BitRef& operator=(bool value);
};

struct BitArray {  // VolumeRaytracer.cuh:210-223: LSB-first bits in u32 words
private:
    size_t size = 0;
    uint32_t* data = nullptr;

public:
    BitArray();
    BitArray(const BitArray& other, bool isGPU);  // deep copy to host (false) or device (true) memory
    BitArray(size_t num_bits, bool isGPU);
    BitArray(const BitArray&) = default;           // shallow, like the reference's implicit copy
    BitArray& operator=(const BitArray&) = default;
    bool operator[](size_t index) const;
    BitRef operator[](size_t index);
    uint32_t* Raw();
    const uint32_t* Raw() const;
    size_t BitSize() const;
    size_t ByteSize() const;
};
// prints every bit as 0/1, first bit first (VolumeRaytracer.cu:86-93); host arrays only
std::ostream& operator<<(std::ostream& os, const BitArray& bits);


template <size_t D>
struct VoxelBuffer {
    BitArray grid{};
    uint16_t dimensions[D]{};
};
typedef VoxelBuffer<3> VoxelBuffer3D;
typedef Bounds<float3> Bounds3Df;

constexpr size_t MAX_STEPS = 2048;

class VoxelRaytracer3D {  // VolumeRaytracer.cuh:291-377
    VoxelRaytracer3D(const VoxelRaytracer3D&) = delete;
    VoxelRaytracer3D& operator=(const VoxelRaytracer3D&) = delete;

public:
    explicit VoxelRaytracer3D(size_t count);
    ~VoxelRaytracer3D();
    void Free();

    void UploadVoxelBuffer(const VoxelBuffer3D& buff);
    void UploadVoxelBufferDatas(VoxelBuffer3D* buff, size_t count);
    void UploadVoxelBufferDataBounds(Bounds3Df* bounds, size_t count);
    int GetFactor() const { return factor; }
    void SetFactor(int f);
    // opaque: the device tables live behind the C ABI
    VoxelBuffer3D* GetVoxelBuffer() { return nullptr; }
    VoxelBuffer3D* GetVoxelBufferDatas() { return nullptr; }
    Bounds3Df* GetVoxelBufferDataBounds() { return nullptr; }

    RayTraceResults<float3> Raytrace(std::vector<float3> origin, std::vector<float3> ray);

    // facade extras
    vxrt_ctx* Context();        // uploads pending tables first
    // CreateVoxels + GenerateLowresVoxelBuffer + Upload* in one on-device step (no dense intermediate)
    void BuildProceduralWorld(uint3 size, int factor, int generator = 1);

private:
    void Flush();
    vxrt_ctx* ctx = nullptr;
    int factor = 1;
    bool dirty = false;
    uint16_t cdims[3] = {0, 0, 0};
    std::vector<uint32_t> coarse_bits, brick_slot, pool;
    std::vector<float> bounds;
    bool have_coarse = false, have_bricks = false, have_bounds = false;
};

// Brickmap build on host threads (VolumeRaytracer.cuh:379-516).  Returned arrays are heap-owned by the caller,
// as in the reference; empty bricks have dimensions 0 and no bits.
std::tuple<VoxelBuffer3D, VoxelBuffer3D*, Bounds3Df*> GenerateLowresVoxelBuffer(const VoxelBuffer3D& originalData, int factor);

}  // namespace GPUDDA
