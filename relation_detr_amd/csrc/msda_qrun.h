// Shared pieces of the query-run MSDA forward kernels (msda_fwd.hip: one tile per workgroup through the texture path;
// msda_res.hip: persistent workgroups with the coarse levels resident in LDS): constants, value / query-side load helpers,
// the bf16 weight split and the per-point matrix-core step.
#pragma once

#include "common.h"

namespace rdetr {

constexpr int kHeads = 8;
constexpr int kHeadDim = 32;
constexpr int kPoints = 4;
constexpr int kMaxLevels = 8;
constexpr int kWavesPerBlock = 4;
constexpr unsigned kInvalidOffset = 0x80000000u;   // >= num_records for every supported tensor

struct LevelTable {
    int h[kMaxLevels];
    int w[kMaxLevels];
    int start[kMaxLevels];
};

template <typename T> struct ValueIO;

// Query-side streams (sampling locations / weights or raw offsets / logits, reference points) and the output rows are touched
// ONCE per launch, the value rows dozens of times.  RDETR_NT_STREAMS (development A/B): mark the streams
// non-temporal so that they do not push value rows out of the XCD's 4-MiB L2.
#ifdef RDETR_NT_STREAMS
template <typename V> __device__ __forceinline__ V qload(const V *p) { return __builtin_nontemporal_load(p); }
template <typename V> __device__ __forceinline__ void qstore(V *p, V v) { __builtin_nontemporal_store(v, p); }
#else
template <typename V> __device__ __forceinline__ V qload(const V *p) { return *p; }
template <typename V> __device__ __forceinline__ void qstore(V *p, V v) { *p = v; }
#endif

template <> struct ValueIO<float> {
    static constexpr int kRunSub = 8, kRunCh = 4;                    // 8 lanes x 4 channels (16 B) per 128-byte head row
    static constexpr unsigned kHeadBytes = kHeadDim * 4;             // 128 B: one cache line per head row
    static constexpr unsigned kPixelBytes = kHeads * kHeadBytes;     // 1 KiB per pixel
    static __device__ __forceinline__ void load_run(__amdgpu_buffer_rsrc_t rsrc, unsigned off, float (&v)[4])
    {
        const f32x4 r = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
    static __device__ __forceinline__ void store_run(float *p, const float (&a)[4])
    {
        qstore(reinterpret_cast<f32x4 *>(p), f32x4{a[0], a[1], a[2], a[3]});
    }
};

template <> struct ValueIO<uint16_t> {                               // bf16 storage, fp32 math
    static constexpr int kRunSub = 4, kRunCh = 8;                    // 4 lanes x 8 channels (16 B) per 64-byte head row
    static constexpr unsigned kHeadBytes = kHeadDim * 2;
    static constexpr unsigned kPixelBytes = kHeads * kHeadBytes;     // 512 B per pixel
    static __device__ __forceinline__ void load_run(__amdgpu_buffer_rsrc_t rsrc, unsigned off, float (&v)[8])
    {
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        v[0] = __builtin_bit_cast(float, r.x << 16); v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, r.y << 16); v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
        v[4] = __builtin_bit_cast(float, r.z << 16); v[5] = __builtin_bit_cast(float, r.z & 0xffff0000u);
        v[6] = __builtin_bit_cast(float, r.w << 16); v[7] = __builtin_bit_cast(float, r.w & 0xffff0000u);
    }
    static __device__ __forceinline__ void store_run(uint16_t *p, const float (&a)[8])
    {
        u32x4 o;
        o.x = f32_to_bf16_bits(a[0]) | (f32_to_bf16_bits(a[1]) << 16);
        o.y = f32_to_bf16_bits(a[2]) | (f32_to_bf16_bits(a[3]) << 16);
        o.z = f32_to_bf16_bits(a[4]) | (f32_to_bf16_bits(a[5]) << 16);
        o.w = f32_to_bf16_bits(a[6]) | (f32_to_bf16_bits(a[7]) << 16);
        qstore(reinterpret_cast<u32x4 *>(p), o);
    }
};

// Query-side scalar loads: the producer inputs (sampling offsets / attention logits) arrive in the
// dtype of the projection that made them: fp32, or bf16 under autocast.
template <typename Q> __device__ __forceinline__ float load_q(const Q *p);
template <> __device__ __forceinline__ float load_q<float>(const float *p) { return qload(p); }
template <> __device__ __forceinline__ float load_q<uint16_t>(const uint16_t *p) { return bf16_bits_to_f32(qload(p)); }
template <typename Q> __device__ __forceinline__ f32x2 load_q2(const Q *p);
template <> __device__ __forceinline__ f32x2 load_q2<float>(const float *p) { return qload(reinterpret_cast<const f32x2 *>(p)); }
template <> __device__ __forceinline__ f32x2 load_q2<uint16_t>(const uint16_t *p)
{
    const unsigned u = qload(reinterpret_cast<const unsigned *>(p));
    return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}

template <int WIDTH> __device__ __forceinline__ float group_max(float v)
{
#pragma unroll
    for (int o = 1; o < WIDTH; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int WIDTH> __device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int o = 1; o < WIDTH; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// N consecutive query-side values with one load (16-byte alignment for 16 bytes and more, natural alignment below)
template <typename Q, int N> struct LoadQ;
template <> struct LoadQ<float, 2> {
    static __device__ __forceinline__ void run(const float *p, float (&v)[2])
    {
        const f32x2 r = qload(reinterpret_cast<const f32x2 *>(p));
        v[0] = r.x; v[1] = r.y;
    }
};
template <> struct LoadQ<float, 4> {
    static __device__ __forceinline__ void run(const float *p, float (&v)[4])
    {
        const f32x4 r = qload(reinterpret_cast<const f32x4 *>(p));
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
};
template <> struct LoadQ<float, 8> {
    static __device__ __forceinline__ void run(const float *p, float (&v)[8])
    {
        const f32x4 a = qload(reinterpret_cast<const f32x4 *>(p)), b = qload(reinterpret_cast<const f32x4 *>(p + 4));
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};
template <> struct LoadQ<uint16_t, 4> {
    static __device__ __forceinline__ void run(const uint16_t *p, float (&v)[4])
    {
        const u32x2 r = qload(reinterpret_cast<const u32x2 *>(p));
        v[0] = __builtin_bit_cast(float, r.x << 16); v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, r.y << 16); v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
    }
};
template <> struct LoadQ<uint16_t, 8> {
    static __device__ __forceinline__ void run(const uint16_t *p, float (&v)[8])
    {
        const u32x4 r = qload(reinterpret_cast<const u32x4 *>(p));
        v[0] = __builtin_bit_cast(float, r.x << 16); v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, r.y << 16); v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
        v[4] = __builtin_bit_cast(float, r.z << 16); v[5] = __builtin_bit_cast(float, r.z & 0xffff0000u);
        v[6] = __builtin_bit_cast(float, r.w << 16); v[7] = __builtin_bit_cast(float, r.w & 0xffff0000u);
    }
};

// 5 points per lane (5 levels x 4 points over the 4 lanes of a bf16 head row): runs of 5 / 10 values that start at a multiple
// of their own size only -- the alignment is stated and the compiler chooses the widest legal loads (dwordx4 + dword, ...)
template <> struct LoadQ<uint16_t, 10> {
    static __device__ __forceinline__ void run(const uint16_t *p, float (&v)[10])
    {
        unsigned r[5];
#ifdef RDETR_NT_STREAMS
        typedef u32x4 __attribute__((aligned(4))) u32x4_a4;
        const u32x4 q4 = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a4 *>(p));
        r[0] = q4.x; r[1] = q4.y; r[2] = q4.z; r[3] = q4.w;
        r[4] = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(p) + 4);
#else
        __builtin_memcpy(r, __builtin_assume_aligned(p, 4), 20);
#endif
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            v[2 * i] = __builtin_bit_cast(float, r[i] << 16);
            v[2 * i + 1] = __builtin_bit_cast(float, r[i] & 0xffff0000u);
        }
    }
};
template <> struct LoadQ<uint16_t, 5> {
    static __device__ __forceinline__ void run(const uint16_t *p, float (&v)[5])
    {
        uint16_t r[5];
#ifdef RDETR_NT_STREAMS
#pragma unroll
        for (int i = 0; i < 5; ++i) r[i] = __builtin_nontemporal_load(p + i);
#else
        __builtin_memcpy(r, __builtin_assume_aligned(p, 2), 10);
#endif
#pragma unroll
        for (int i = 0; i < 5; ++i) v[i] = bf16_bits_to_f32(r[i]);
    }
};
template <> struct LoadQ<float, 10> {
    static __device__ __forceinline__ void run(const float *p, float (&v)[10])
    {
#ifdef RDETR_NT_STREAMS
        typedef f32x4 __attribute__((aligned(8))) f32x4_a8;
        const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_a8 *>(p));
        const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4_a8 *>(p + 4));
        const f32x2 c = __builtin_nontemporal_load(reinterpret_cast<const f32x2 *>(p + 8));
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = c.x; v[9] = c.y;
#else
        __builtin_memcpy(v, __builtin_assume_aligned(p, 8), 40);
#endif
    }
};
template <> struct LoadQ<float, 5> {
    static __device__ __forceinline__ void run(const float *p, float (&v)[5])
    {
#ifdef RDETR_NT_STREAMS
        typedef f32x4 __attribute__((aligned(4))) f32x4_a4;
        const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_a4 *>(p));
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = __builtin_nontemporal_load(p + 4);
#else
        __builtin_memcpy(v, __builtin_assume_aligned(p, 4), 20);
#endif
    }
};

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// bf16 high parts (round to nearest even) and low parts of two fp32 weights, packed (a in the low half): w = hi + lo up to 2^-17 |w|
__device__ __forceinline__ void split2_bf16(float a, float b, unsigned &hi, unsigned &lo)
{
    hi = __builtin_bit_cast(unsigned, bf16x2_t{(__bf16)a, (__bf16)b});
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, bf16x2_t{(__bf16)ra, (__bf16)rb});
}

// One sampling point of 16 queries x 4 lanes on the matrix cores (bf16 value): v_mfma_f32_4x4x4_16b_bf16 is 16 independent
// 4x4x4 products, one per query (lanes 4q .. 4q+3).  K = the 4 corners.  B[k][j] = corner k of channel c of lane j: the lane's
// four 16-byte loads re-paired by v_perm_b32 (2 per channel pair and corner pair) -- no bf16 -> fp32 unpacking.  A[i][k] = the
// corner weights, row 0 their bf16 high parts, row 1 the low parts (rows 2, 3 repeat them), so D[0][j] + D[1][j] is the
// fp32-weighted sum of lane j's channel c up to 2^-17 per weight; products are exact, accumulation fp32.  Half the vector-ALU
// instructions of unpack + v_pk_fma_f32 (20 vs 52 per point).  r00..r11: the loads; wq: this lane's A row.
__device__ __forceinline__ void mfma_point(const u32x4 &r00, const u32x4 &r01, const u32x4 &r10, const u32x4 &r11, const u32x2 &wq,
                                           f32x4 (&acc)[8])
{
    const s16x4 a = __builtin_bit_cast(s16x4, wq);
    const unsigned t0[4] = {r00.x, r00.y, r00.z, r00.w}, t1[4] = {r01.x, r01.y, r01.z, r01.w};
    const unsigned b0[4] = {r10.x, r10.y, r10.z, r10.w}, b1[4] = {r11.x, r11.y, r11.z, r11.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32x2 even = {__builtin_amdgcn_perm(t1[j], t0[j], 0x05040100u), __builtin_amdgcn_perm(b1[j], b0[j], 0x05040100u)};
        const u32x2 odd = {__builtin_amdgcn_perm(t1[j], t0[j], 0x07060302u), __builtin_amdgcn_perm(b1[j], b0[j], 0x07060302u)};
        acc[2 * j] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, __builtin_bit_cast(s16x4, even), acc[2 * j], 0, 0, 0);
        acc[2 * j + 1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, __builtin_bit_cast(s16x4, odd), acc[2 * j + 1], 0, 0, 0);
    }
}

}  // namespace rdetr
