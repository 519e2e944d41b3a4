// Shared device/host helpers for librelation_detr_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "relation_detr_amd.h"

namespace rdetr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kXcds = 8;         // MI355X: 8 XCDs, each with a private 4 MiB L2

// Workgroups are dealt round-robin over the XCDs (block b and b+8 share an L2).  Map the hardware
// block id to a logical id so that every XCD owns one contiguous chunk of the logical range:
// neighbouring queries then hit the same L2.  Bijective for any grid size.  Speed only -- no
// result depends on the placement.
__device__ __forceinline__ int xcd_contiguous_block(int bid, int nblk)
{
    const int q = nblk / kXcds, r = nblk % kXcds;
    const int xcd = bid % kXcds, idx = bid / kXcds;
    return xcd < r ? xcd * (q + 1) + idx : r * (q + 1) + (xcd - r) * q + idx;
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned int lo16)
{
    return __builtin_bit_cast(float, lo16 << 16);
}

// Round-to-nearest-even f32 -> bf16 through the hardware convert (keeps NaN a NaN).
__device__ __forceinline__ unsigned int f32_to_bf16_bits(float f)
{
    return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)f);
}

// Two f32 -> one register of two bf16 (a in the low half), round-to-nearest-even: ONE v_cvt_pk_bf16_f32
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack_bf16x2(float a, float b)
{
    return __builtin_bit_cast(unsigned int, bf16x2_t{(__bf16)a, (__bf16)b});
}
// max(x, 0) on both bf16 halves: a negative bf16 is a negative int16 (v_pk_max_i16); -0 -> +0, NaN with the sign bit -> 0
__device__ __forceinline__ unsigned int relu_bf16x2(unsigned int packed)
{
    return __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, packed), s16x2_t{0, 0}));
}

inline int launch_status()
{
    return hipGetLastError() == hipSuccess ? RDETR_OK : RDETR_ERR_LAUNCH;
}

}  // namespace rdetr
