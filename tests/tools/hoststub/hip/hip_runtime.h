// Host stand-in for the few HIP device builtins vxrt_device.hpp / vxrt_wave2.hpp use, so the traversal code can be
// compiled for the CPU with ONE lane per "wave" and stepped in a debugger / compared with the oracle.
// Debug tooling only (tests/tools/host_wave_check.cpp); never part of the product build.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __global__
#define __launch_bounds__(x)
struct uint2 { uint32_t x, y; };
static inline uint2 make_uint2(uint32_t a, uint32_t b) { return uint2{a, b}; }
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
typedef unsigned long long lanemask_t;  // one lane on the host: bit 0
static inline lanemask_t lane_mask(bool p) { return p ? 1ull : 0ull; }
static inline bool lane_test(lanemask_t m) { return (m & 1ull) != 0ull; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
#define VXRT_HOST_CHECK 1
static inline uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu) + c; }  // v_mad_u32_u24
static inline int __float2int_rz(float v) { if (!(v == v)) return 0; if (v >= 2147483648.0f) return 2147483647; if (v <= -2147483648.0f) return (-2147483647 - 1); return (int)v; }
using std::min;
using std::max;
