// GPUDDA/VoxelWorldBuilder.h -- CreateVoxels (VoxelRT/VoxelWorldBuilder.cuh:12-32) over the C ABI.
#pragma once

#include "VolumeRaytracer.h"

// Dense X*Y*Z procedural world as host bits in tiled-linear order, like the reference's CreateVoxels; the
// voxels are evaluated brick by brick on the device (PopulateVoxels' formula) and scattered into the dense
// array.  Each size component must be a multiple of 64.  For large worlds prefer
// GPUDDA::VoxelRaytracer3D::BuildProceduralWorld, which never materialises the dense array.
GPUDDA::VoxelBuffer3D CreateVoxels(uint3 size);
