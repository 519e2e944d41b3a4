// Multi-scale deformable attention, forward -- "hybrid" kernel for large query counts on gfx950 (MI355X).
//
// The direct gather (msda_fwd.hip) is bound by the texture addresser: every 16-byte lane load costs TA time
// and 64 x the value tensor has to come through it.  The coarsest pyramid levels are
// tiny (R50: level 3 = 273 pixels, level 2 = 1,050 pixels per image) yet receive the same number of samples
// as the large ones, so this kernel keeps the (image, head) planes of the R coarsest levels RESIDENT IN LDS
// (64 B per pixel: the whole head in bf16, one 16-channel half in fp32 -- fp32 runs two workgroups per head)
// and gathers them with ds_read_b128 (LDS pipe, 256 B/clk/CU) while the fine levels still come through
// buffer_load_dwordx4 (TA pipe).  With R = 2 of L = 4 levels resident the TA load is halved.
//
//   workgroup  = 1024 threads = 16 waves, persistent over the query tiles of one (image, head[, half], split);
//                planes are loaded once per workgroup, then there is no barrier at all.
//   wave pass  = 16 queries x 4 lanes; lane `sub` prepares point `sub` of every level (pixel coordinates,
//                4 corner offsets, 4 corner weights x attention weight) and stages TWO LEVELS (8 points) at a
//                time in a 4 KiB per-wave LDS area (64 KiB for 16 waves); the rest of LDS (~95 KiB) holds the
//                planes.  The gather itself is a real loop over the staged points (a fully unrolled body makes
//                hipcc hoist all 64 row loads and spill).
//   corners outside a level: global path -> byte offset 0x80000000 (range-checked load returns 0);
//                LDS path -> offset of a zeroed row.  Same arithmetic, same results as msda_fwd.hip.
//   FUSED: softmax over L*P and the location arithmetic in the set-up phase (see msda_fwd.hip).
#include <algorithm>

#include "common.h"

namespace rdetr {

constexpr int kHyThreads = 1024;
constexpr int kHyWaves = kHyThreads / kWave;
constexpr int kHyHeads = 8, kHyHeadDim = 32, kHyPoints = 4;
constexpr int kHyRowBytes = 64;
constexpr int kHyQueriesPerBlockPass = kHyWaves * 16;
constexpr unsigned kHyInvalid = 0x80000000u;
constexpr int kHyZeroOff = 0;                  // 128 zero bytes at the start of LDS
constexpr int kHyPlaneBase = 128;
constexpr int kHyStagePoints = 2 * kHyPoints;                   // two levels per staging round
constexpr int kHyStageBytes = kHyWaves * kHyStagePoints * 16 * 32;   // 8 points x 16 slots x 32 B per wave = 4 KiB
constexpr int kHyLdsBudget = 160 * 1024 - kHyStageBytes - 1024;   // what is left for the resident planes

template <int P> __device__ __forceinline__ unsigned hy_bcast(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, P * 0x55, 0xf, 0xf, false);     // quad_perm [P,P,P,P]
}
template <int P> __device__ __forceinline__ float hy_bcast(float v)
{
    return __builtin_bit_cast(float, hy_bcast<P>(__builtin_bit_cast(unsigned, v)));
}
__device__ __forceinline__ float hy_quad_max(float v)
{
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false)));
    return v;
}
__device__ __forceinline__ float hy_quad_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));
    return v;
}

template <typename T> struct HybridIO;
template <> struct HybridIO<float> {
    static constexpr int kHalves = 2, kCh = 4;
    static __device__ __forceinline__ void unpack(u32x4 r, float (&v)[4])
    {
        const f32x4 f = __builtin_bit_cast(f32x4, r);      // never bit_cast a vector COMPONENT (clang miscompile)
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    }
    static __device__ __forceinline__ u32x4 pack(const float (&a)[4])
    {
        return __builtin_bit_cast(u32x4, f32x4{a[0], a[1], a[2], a[3]});
    }
    static __device__ __forceinline__ float q(const float *p) { return *p; }
    static __device__ __forceinline__ f32x2 q2(const float *p) { return *reinterpret_cast<const f32x2 *>(p); }
};
template <> struct HybridIO<uint16_t> {
    static constexpr int kHalves = 1, kCh = 8;
    static __device__ __forceinline__ void unpack(u32x4 r, float (&v)[8])
    {
        v[0] = __builtin_bit_cast(float, r.x << 16); v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, r.y << 16); v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
        v[4] = __builtin_bit_cast(float, r.z << 16); v[5] = __builtin_bit_cast(float, r.z & 0xffff0000u);
        v[6] = __builtin_bit_cast(float, r.w << 16); v[7] = __builtin_bit_cast(float, r.w & 0xffff0000u);
    }
    static __device__ __forceinline__ u32x4 pack(const float (&a)[8])
    {
        u32x4 o;
        o.x = f32_to_bf16_bits(a[0]) | (f32_to_bf16_bits(a[1]) << 16);
        o.y = f32_to_bf16_bits(a[2]) | (f32_to_bf16_bits(a[3]) << 16);
        o.z = f32_to_bf16_bits(a[4]) | (f32_to_bf16_bits(a[5]) << 16);
        o.w = f32_to_bf16_bits(a[6]) | (f32_to_bf16_bits(a[7]) << 16);
        return o;
    }
    static __device__ __forceinline__ float q(const uint16_t *p) { return bf16_bits_to_f32(*p); }
    static __device__ __forceinline__ f32x2 q2(const uint16_t *p)
    {
        const unsigned u = *reinterpret_cast<const unsigned *>(p);
        return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
    }
};

struct HybridLevels {
    int h[8], w[8], start[8], lds_base[8];
};

// LT = number of levels (compile time), RES = how many of the coarsest levels are resident in LDS.
template <typename T, int LT, int RES, bool FUSED>
__global__ __launch_bounds__(kHyThreads) void msda_fwd_hybrid_kernel(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const void *__restrict__ src_a, const void *__restrict__ src_b, const float *__restrict__ ref, int ref_dim, int S,
    int Nq, int splits, int nblk, int stage_base, T *__restrict__ out)
{
    using IO = HybridIO<T>;
    constexpr int NH = IO::kHalves, CH = IO::kCh;
    constexpr unsigned PIXB = kHyHeads * kHyHeadDim * sizeof(T), HEADB = kHyHeadDim * sizeof(T);
    constexpr int LP = LT * kHyPoints;
    constexpr int FIRST_RES = LT - RES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ HybridLevels lvl;

    const int tid = threadIdx.x;
    if (tid == 0) {
        int base = kHyPlaneBase;
        for (int l = 0; l < LT; ++l) {
            lvl.h[l] = (int)shapes[2 * l];
            lvl.w[l] = (int)shapes[2 * l + 1];
            lvl.start[l] = (int)level_start[l];
            lvl.lds_base[l] = base;
            if (l >= FIRST_RES) base += lvl.h[l] * lvl.w[l] * kHyRowBytes;
        }
    }
    if (tid < 32) reinterpret_cast<unsigned *>(lds + kHyZeroOff)[tid] = 0u;
    __syncthreads();

    int id = xcd_contiguous_block(blockIdx.x, nblk);
    const int split = id % splits; id /= splits;
    const int half = id % NH;      id /= NH;
    const int m = id % kHyHeads;
    const int b = id / kHyHeads;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T *>(value) + (size_t)b * S * (kHyHeads * kHyHeadDim), 0, (unsigned)S * PIXB, 0x00020000);
    const unsigned chan_off = (unsigned)m * HEADB + (unsigned)half * (NH == 2 ? kHyRowBytes : 0);

    // ---- resident planes: every pixel of levels >= FIRST_RES, 64 B each, for this (image, head[, half]) -------
#pragma unroll
    for (int l = FIRST_RES; l < LT; ++l) {
        const int n = lvl.h[l] * lvl.w[l] * 4;
        for (int u = tid; u < n; u += kHyThreads) {
            const int px = u >> 2, s16 = u & 3;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                rsrc, (unsigned)(lvl.start[l] + px) * PIXB + chan_off + (unsigned)s16 * 16u, 0, 0);
            *reinterpret_cast<u32x4 *>(lds + lvl.lds_base[l] + px * kHyRowBytes + s16 * 16) = v;
        }
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int slot = lane >> 2, sub = lane & 3;
    const int ntiles = (Nq + kHyQueriesPerBlockPass - 1) / kHyQueriesPerBlockPass;
    // per-wave staging of the prepared points: [point][query slot] -> {4 corner offsets, 4 corner weights}
    u32x4 *soff = reinterpret_cast<u32x4 *>(lds + stage_base) + wave * (kHyStagePoints * 16);
    f32x4 *swgt = reinterpret_cast<f32x4 *>(lds + stage_base + kHyWaves * kHyStagePoints * 16 * 16) + wave * (kHyStagePoints * 16);

    for (int tile = split; tile < ntiles; tile += splits) {
        const int q = tile * kHyQueriesPerBlockPass + wave * 16 + slot;
        const bool qok = q < Nq;
        const size_t row = (size_t)b * Nq + (qok ? q : 0);
        const size_t hrow = (row * kHyHeads + m) * (size_t)LP;

        // ---- set-up: lane `sub` owns point `sub` of every level ------------------------------------------
        f32x2 pxy[LT];
        float pa[LT];
        if constexpr (FUSED) {
            const T *off_q = static_cast<const T *>(src_a) + hrow * 2;
            const T *lg_q = static_cast<const T *>(src_b) + hrow;
            float mx = -__builtin_inff();
#pragma unroll
            for (int k = 0; k < LT; ++k) {
                pa[k] = IO::q(lg_q + k * kHyPoints + sub);
                pxy[k] = IO::q2(off_q + 2 * (k * kHyPoints + sub));
                mx = fmaxf(mx, pa[k]);
            }
            mx = hy_quad_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < LT; ++k) {
                pa[k] = expf(pa[k] - mx);
                sum += pa[k];
            }
            sum = hy_quad_sum(sum);
#pragma unroll
            for (int k = 0; k < LT; ++k) {
                const float *rp = ref + (row * LT + k) * (size_t)ref_dim;
                pa[k] = pa[k] / sum;
                if (ref_dim == 2) {
                    pxy[k].x = rp[0] + pxy[k].x / (float)lvl.w[k];
                    pxy[k].y = rp[1] + pxy[k].y / (float)lvl.h[k];
                } else {
                    pxy[k].x = rp[0] + pxy[k].x * (1.0f / kHyPoints) * rp[2] * 0.5f;
                    pxy[k].y = rp[1] + pxy[k].y * (1.0f / kHyPoints) * rp[3] * 0.5f;
                }
            }
        } else {
            const float *loc_q = static_cast<const float *>(src_a) + hrow * 2;
            const float *att_q = static_cast<const float *>(src_b) + hrow;
#pragma unroll
            for (int k = 0; k < LT; ++k) {
                pxy[k] = *reinterpret_cast<const f32x2 *>(loc_q + 2 * (k * kHyPoints + sub));
                pa[k] = att_q[k * kHyPoints + sub];
            }
        }
        float acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = 0.f;
        const unsigned lane_off = (unsigned)sub * 16u;
        constexpr int kUnroll = RES == 2 ? 4 : 2;   // points per unrolled body (16 / 8 row loads in flight per lane)

#pragma unroll
        for (int round = 0; round < LT / 2; ++round) {
            // ---- stage the 8 points of levels 2*round and 2*round+1 ---------------------------------------
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int k = 2 * round + kk;
                const bool resident = k >= FIRST_RES;               // compile-time after unrolling
                const int h = lvl.h[k], w = lvl.w[k];
                const float x = pxy[k].x * (float)w - 0.5f;
                const float y = pxy[k].y * (float)h - 0.5f;
                const bool inside = qok && (y > -1.f) && (x > -1.f) && (y < (float)h) && (x < (float)w);   // false for NaN
                const float xf = floorf(x), yf = floorf(y);
                const int x0 = inside ? (int)xf : 0, y0 = inside ? (int)yf : 0;
                const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
                const float a = pa[k];
                const bool okx0 = inside && x0 >= 0, okx1 = inside && x0 + 1 <= w - 1;
                const bool oky0 = y0 >= 0, oky1 = y0 + 1 <= h - 1;
                u32x4 o;
                if (resident) {
                    const unsigned base = (unsigned)lvl.lds_base[k] + (unsigned)(y0 * w + x0) * kHyRowBytes;
                    const unsigned rowb = (unsigned)w * kHyRowBytes;
                    o.x = (okx0 && oky0) ? base : (unsigned)kHyZeroOff;
                    o.y = (okx1 && oky0) ? base + kHyRowBytes : (unsigned)kHyZeroOff;
                    o.z = (okx0 && oky1) ? base + rowb : (unsigned)kHyZeroOff;
                    o.w = (okx1 && oky1) ? base + rowb + kHyRowBytes : (unsigned)kHyZeroOff;
                } else {
                    const unsigned base = (unsigned)(lvl.start[k] + y0 * w + x0) * PIXB + chan_off;
                    const unsigned rowb = (unsigned)w * PIXB;
                    o.x = (okx0 && oky0) ? base : kHyInvalid;
                    o.y = (okx1 && oky0) ? base + PIXB : kHyInvalid;
                    o.z = (okx0 && oky1) ? base + rowb : kHyInvalid;
                    o.w = (okx1 && oky1) ? base + rowb + PIXB : kHyInvalid;
                }
                f32x4 wt;
                wt.x = inside ? hy * hx * a : 0.f;
                wt.y = inside ? hy * lx * a : 0.f;
                wt.z = inside ? ly * hx * a : 0.f;
                wt.w = inside ? ly * lx * a : 0.f;
                soff[(kk * kHyPoints + sub) * 16 + slot] = o;
                swgt[(kk * kHyPoints + sub) * 16 + slot] = wt;
            }
            // the staging area is private to this wave and LDS ops of one wave complete in order (wave-level
            // fences, no s_barrier)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---- gather: points of non-resident levels through the texture path, resident ones from LDS ---------
            const int n_global = (2 * round + 2 <= FIRST_RES) ? 2 * kHyPoints
                                 : ((2 * round + 1 <= FIRST_RES) ? kHyPoints : 0);      // constant after unrolling
            for (int pt0 = 0; pt0 < n_global; pt0 += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const u32x4 o = soff[(pt0 + u) * 16 + slot];
                    const f32x4 wt = swgt[(pt0 + u) * 16 + slot];
                    float v00[CH], v01[CH], v10[CH], v11[CH];
                    IO::unpack(__builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + lane_off, 0, 0), v00);
                    IO::unpack(__builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + lane_off, 0, 0), v01);
                    IO::unpack(__builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + lane_off, 0, 0), v10);
                    IO::unpack(__builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + lane_off, 0, 0), v11);
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        acc[c] += wt.x * v00[c];
                        acc[c] += wt.y * v01[c];
                        acc[c] += wt.z * v10[c];
                        acc[c] += wt.w * v11[c];
                    }
                }
                // ask the scheduler for: the 8 staging reads of the body first, then ALL 16 row loads, then the FMAs
                // (left alone, hipcc issues 4 loads, waits, computes -- 4 loads in flight per wave)
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);     // DS read
                __builtin_amdgcn_sched_group_barrier(0x020, 16, 0);    // VMEM read
            }
#pragma unroll kUnroll
            for (int pt = n_global; pt < kHyStagePoints; ++pt) {
                const u32x4 o = soff[pt * 16 + slot];
                const f32x4 wt = swgt[pt * 16 + slot];
                float v00[CH], v01[CH], v10[CH], v11[CH];
                IO::unpack(*reinterpret_cast<const u32x4 *>(lds + o.x + lane_off), v00);
                IO::unpack(*reinterpret_cast<const u32x4 *>(lds + o.y + lane_off), v01);
                IO::unpack(*reinterpret_cast<const u32x4 *>(lds + o.z + lane_off), v10);
                IO::unpack(*reinterpret_cast<const u32x4 *>(lds + o.w + lane_off), v11);
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    acc[c] += wt.x * v00[c];
                    acc[c] += wt.y * v01[c];
                    acc[c] += wt.z * v10[c];
                    acc[c] += wt.w * v11[c];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // reads done before the next round's staging writes
            __builtin_amdgcn_wave_barrier();
        }
        if (qok) {
            T *o = out + row * (kHyHeads * kHyHeadDim) + m * kHyHeadDim + half * (NH == 2 ? 16 : 0) + sub * CH;
            *reinterpret_cast<u32x4 *>(o) = IO::pack(acc);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
// Decide how many of the coarsest levels fit the LDS budget (needs a HOST copy of the shape table).
static int hybrid_resident_levels(const int64_t *host_shapes, int L, long long *plane_bytes)
{
    long long bytes = 0;
    int res = 0;
    for (int l = L - 1; l >= 1; --l) {               // never all levels: the finest one stays on the texture path
        const long long b = host_shapes[2 * l] * host_shapes[2 * l + 1] * kHyRowBytes;
        if (bytes + b > kHyLdsBudget || res == 2) break;
        bytes += b;
        ++res;
    }
    *plane_bytes = bytes;
    return res;
}

template <typename T, int LT, int RES, bool FUSED>
static int launch_hybrid(const T *value, const int64_t *shapes, const int64_t *level_start, const void *src_a,
                         const void *src_b, const float *ref, int ref_dim, int B, int S, int Nq, long long plane_bytes,
                         T *out, hipStream_t stream)
{
    auto kern = msda_fwd_hybrid_kernel<T, LT, RES, FUSED>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);   // static __shared__ (level table) takes the rest
    if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
    const int ntiles = (Nq + kHyQueriesPerBlockPass - 1) / kHyQueriesPerBlockPass;
    const long long per_split = (long long)B * kHyHeads * HybridIO<T>::kHalves;
    // one resident workgroup per CU: aim at ~2 rounds of 256 workgroups, each amortising its plane load over >= 4 tiles
    long long splits = (512 + per_split - 1) / per_split;
    splits = std::max<long long>(1, std::min<long long>(splits, std::max(1, ntiles / 4)));
    const long long nblk = per_split * splits;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    const int stage_base = (int)((kHyPlaneBase + plane_bytes + 15) / 16 * 16);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(kHyThreads), (size_t)(stage_base + kHyStageBytes), stream, value,
                       shapes, level_start, src_a, src_b, ref, ref_dim, S, Nq, (int)splits, (int)nblk, stage_base, out);
    return launch_status();
}

// Returns RDETR_ERR_UNSUPPORTED when the shape is not served (callers then use the direct kernel).
template <typename T, bool FUSED>
int msda_hybrid_forward(const T *value, const int64_t *shapes, const int64_t *level_start, const int64_t *host_shapes,
                        const void *src_a, const void *src_b, const float *ref, int ref_dim, int B, int S, int L, int Nq,
                        T *out, hipStream_t stream)
{
    if (!host_shapes || L != 4 || Nq < 4 * kHyQueriesPerBlockPass) return RDETR_ERR_UNSUPPORTED;
    auto al16 = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (!al16(value) || !al16(out) || reinterpret_cast<uintptr_t>(src_a) % 8 != 0) return RDETR_ERR_UNSUPPORTED;
    if ((long long)S * kHyHeads * kHyHeadDim * (long long)sizeof(T) >= (1ll << 31)) return RDETR_ERR_UNSUPPORTED;
    long long plane_bytes = 0;
    const int res = hybrid_resident_levels(host_shapes, L, &plane_bytes);
    if (res == 2)
        return launch_hybrid<T, 4, 2, FUSED>(value, shapes, level_start, src_a, src_b, ref, ref_dim, B, S, Nq, plane_bytes,
                                             out, stream);
    if (res == 1)
        return launch_hybrid<T, 4, 1, FUSED>(value, shapes, level_start, src_a, src_b, ref, ref_dim, B, S, Nq, plane_bytes,
                                             out, stream);
    return RDETR_ERR_UNSUPPORTED;
}

template int msda_hybrid_forward<float, false>(const float *, const int64_t *, const int64_t *, const int64_t *, const void *,
                                               const void *, const float *, int, int, int, int, int, float *, hipStream_t);
template int msda_hybrid_forward<float, true>(const float *, const int64_t *, const int64_t *, const int64_t *, const void *,
                                              const void *, const float *, int, int, int, int, int, float *, hipStream_t);
template int msda_hybrid_forward<uint16_t, false>(const uint16_t *, const int64_t *, const int64_t *, const int64_t *,
                                                  const void *, const void *, const float *, int, int, int, int, int,
                                                  uint16_t *, hipStream_t);
template int msda_hybrid_forward<uint16_t, true>(const uint16_t *, const int64_t *, const int64_t *, const int64_t *,
                                                 const void *, const void *, const float *, int, int, int, int, int,
                                                 uint16_t *, hipStream_t);

}  // namespace rdetr
