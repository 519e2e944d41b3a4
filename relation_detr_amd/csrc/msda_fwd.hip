// Multi-scale deformable attention, forward -- hand-written for gfx950 (MI355X).
//
// Replaces the reference's ms_deformable_im2col_gpu_kernel
// (models/bricks/ops/cuda/ms_deform_im2col_cuda.cuh:226-288: one thread per output scalar, every
// thread re-reading the 16 (loc, weight) triples of its head and issuing 64 scattered 4-byte loads).
//
// Design (wave-per-query kernel, H = 8 heads x D = 32 channels, P = 4 points, L <= 8 levels):
//   * one 64-lane wavefront owns one query (b, q); lane = head*8 + sub, the lane accumulates the
//     4 channels [4*sub, 4*sub+4) of its head in registers, so a wave's gather instruction reads
//     eight full 128-byte head rows (fp32) -- one per head -- of the level-packed value tensor;
//   * the per-(head, point) bilinear set-up (pixel coords, 4 corner byte offsets, 4 corner weights
//     already multiplied by the attention weight) is computed ONCE by one lane and staged in LDS,
//     then broadcast to the 8 lanes of the head with two conflict-free ds_read_b128 per point;
//   * corners outside the level get the byte offset 0x80000000: the buffer descriptor's range
//     check returns 0 for them without a memory access, which is exactly the zero-padding rule of
//     ms_deform_im2col_cuda.cuh:44-67, so the inner loop has no branches;
//   * the value tensor of image b is addressed through one wave-uniform buffer descriptor with
//     32-bit byte offsets (S*H*D*sizeof(T) < 2^31 is checked on the host);
//   * hardware block ids are remapped so that each XCD (private L2) owns a contiguous range of
//     queries (common.h: xcd_contiguous_block).
// Any other (H, D, P) runs msda_fwd_generic_kernel (one thread per output, 64-bit indexing).
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

template <> struct ValueIO<float> {
    static constexpr unsigned kLaneBytes = 16;                       // 4 channels x fp32
    static constexpr unsigned kHeadBytes = kHeadDim * 4;             // 128 B: one cache line per head row
    static constexpr unsigned kPixelBytes = kHeads * kHeadBytes;     // 1 KiB per pixel
    static __device__ __forceinline__ f32x4 load(__amdgpu_buffer_rsrc_t rsrc, unsigned off)
    {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
    static __device__ __forceinline__ void store(float *row, int lane, f32x4 acc)
    {
        reinterpret_cast<f32x4 *>(row)[lane] = acc;
    }
};

template <> struct ValueIO<uint16_t> {                               // bf16 storage, fp32 math
    static constexpr unsigned kLaneBytes = 8;
    static constexpr unsigned kHeadBytes = kHeadDim * 2;
    static constexpr unsigned kPixelBytes = kHeads * kHeadBytes;     // 512 B per pixel
    static __device__ __forceinline__ f32x4 load(__amdgpu_buffer_rsrc_t rsrc, unsigned off)
    {
        const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);
        f32x4 v;
        v.x = __builtin_bit_cast(float, r.x << 16);
        v.y = __builtin_bit_cast(float, r.x & 0xffff0000u);
        v.z = __builtin_bit_cast(float, r.y << 16);
        v.w = __builtin_bit_cast(float, r.y & 0xffff0000u);
        return v;
    }
    static __device__ __forceinline__ void store(uint16_t *row, int lane, f32x4 acc)
    {
        u32x2 p;
        p.x = f32_to_bf16_bits(acc.x) | (f32_to_bf16_bits(acc.y) << 16);
        p.y = f32_to_bf16_bits(acc.z) | (f32_to_bf16_bits(acc.w) << 16);
        reinterpret_cast<u32x2 *>(row)[lane] = p;
    }
};

// LT = compile-time level count (4, 5) or 0 for a run-time L in [1, 8].
template <typename T, int LT>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void msda_fwd_wave_kernel(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, int S, int L_rt, int Nq, int tiles_per_image,
    int queries_per_wave, int nblk, T *__restrict__ out)
{
    using IO = ValueIO<T>;
    const int L = LT ? LT : L_rt;
    const int LP = L * kPoints;

    __shared__ LevelTable lvl;
    // staging per wave: [point][head] -> {4 corner byte offsets, 4 corner weights}
    __shared__ u32x4 stage_off[kWavesPerBlock][kMaxLevels * kPoints * kHeads];
    __shared__ f32x4 stage_wgt[kWavesPerBlock][kMaxLevels * kPoints * kHeads];

    const int tid = threadIdx.x;
    if (tid < L) {
        lvl.h[tid] = (int)shapes[2 * tid];
        lvl.w[tid] = (int)shapes[2 * tid + 1];
        lvl.start[tid] = (int)level_start[tid];
    }
    __syncthreads();

    const int logical = xcd_contiguous_block(blockIdx.x, nblk);
    const int b = logical / tiles_per_image;
    const int tile = logical - b * tiles_per_image;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int m = lane >> 3;          // head
    const int sub = lane & 7;         // which 4-channel slice of the head / which staged points

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T *>(value) + (size_t)b * S * (kHeads * kHeadDim), 0, (unsigned)S * IO::kPixelBytes, 0x00020000);
    const unsigned lane_off = (unsigned)m * IO::kHeadBytes + (unsigned)sub * IO::kLaneBytes;

    u32x4 *soff = stage_off[wave];
    f32x4 *swgt = stage_wgt[wave];

    const int q_begin = (tile * kWavesPerBlock + wave) * queries_per_wave;
    for (int qi = 0; qi < queries_per_wave; ++qi) {
        const int q = q_begin + qi;
        if (q >= Nq) break;                                   // wave-uniform
        const size_t row = (size_t)b * Nq + q;
        const float *loc_q = loc + (row * kHeads + m) * (size_t)LP * 2;
        const float *att_q = attn + (row * kHeads + m) * (size_t)LP;

        // ---- set-up: each lane prepares points sub, sub+8, ... of its head -----------------------
        for (int pt = sub; pt < LP; pt += 8) {
            const f32x2 xy = *reinterpret_cast<const f32x2 *>(loc_q + 2 * pt);
            const float a = att_q[pt];
            const int l = pt / kPoints;
            const int h = lvl.h[l], w = lvl.w[l];
            const float x = xy.x * (float)w - 0.5f;
            const float y = xy.y * (float)h - 0.5f;
            const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)h) && (x < (float)w);   // false for NaN
            const float xf = floorf(x), yf = floorf(y);
            const int x0 = inside ? (int)xf : 0, y0 = inside ? (int)yf : 0;
            const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
            const bool okx0 = inside && x0 >= 0, okx1 = inside && x0 + 1 <= w - 1;
            const bool oky0 = y0 >= 0, oky1 = y0 + 1 <= h - 1;
            const unsigned base = (unsigned)(lvl.start[l] + y0 * w + x0) * IO::kPixelBytes;
            const unsigned rowb = (unsigned)w * IO::kPixelBytes;
            u32x4 o;
            o.x = (okx0 && oky0) ? base : kInvalidOffset;
            o.y = (okx1 && oky0) ? base + IO::kPixelBytes : kInvalidOffset;
            o.z = (okx0 && oky1) ? base + rowb : kInvalidOffset;
            o.w = (okx1 && oky1) ? base + rowb + IO::kPixelBytes : kInvalidOffset;
            f32x4 wt;
            wt.x = inside ? hy * hx * a : 0.f;
            wt.y = inside ? hy * lx * a : 0.f;
            wt.z = inside ? ly * hx * a : 0.f;
            wt.w = inside ? ly * lx * a : 0.f;
            soff[pt * kHeads + m] = o;
            swgt[pt * kHeads + m] = wt;
        }
        // staging is private to this wave: LDS ops of one wave complete in order, so a wave-level
        // fence (no s_barrier) is all that is needed between the writes above and the reads below.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- gather + weighted sum --------------------------------------------------------------
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int pt = 0; pt < LP; ++pt) {
            const u32x4 o = soff[pt * kHeads + m];
            const f32x4 wt = swgt[pt * kHeads + m];
            const f32x4 v00 = IO::load(rsrc, o.x + lane_off);
            const f32x4 v01 = IO::load(rsrc, o.y + lane_off);
            const f32x4 v10 = IO::load(rsrc, o.z + lane_off);
            const f32x4 v11 = IO::load(rsrc, o.w + lane_off);
            acc += wt.x * v00;
            acc += wt.y * v01;
            acc += wt.z * v10;
            acc += wt.w * v11;
        }
        IO::store(out + row * (kHeads * kHeadDim), lane, acc);

        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // reads done before the next query's writes
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// Generic fallback: one thread per output scalar, any (H, D, L, P), 64-bit indexing.
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<uint16_t>(uint16_t v) { return bf16_bits_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ uint16_t from_f32<uint16_t>(float v) { return (uint16_t)f32_to_bf16_bits(v); }

template <typename T>
__global__ __launch_bounds__(256) void msda_fwd_generic_kernel(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, int S, int H, int D, int L, int Nq, int P,
    long long total, T *__restrict__ out)
{
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % D);
        const long long r = idx / D;              // (b*Nq + q)*H + m
        const int m = (int)(r % H);
        const long long b = r / H / Nq;
        const long long pix = (long long)H * D;
        const T *vb = value + b * S * pix + (long long)m * D + c;
        const float *lp = loc + r * L * P * 2;
        const float *ap = attn + r * L * P;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) {
            const int h = (int)shapes[2 * l], w = (int)shapes[2 * l + 1];
            const T *vl = vb + level_start[l] * pix;
            for (int p = 0; p < P; ++p) {
                const float x = lp[(l * P + p) * 2] * (float)w - 0.5f;
                const float y = lp[(l * P + p) * 2 + 1] * (float)h - 0.5f;
                if (!((y > -1.f) && (x > -1.f) && (y < (float)h) && (x < (float)w))) continue;
                const float xf = floorf(x), yf = floorf(y);
                const int x0 = (int)xf, y0 = (int)yf;
                const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
                const float a = ap[l * P + p];
                float s = 0.f;
                if (y0 >= 0 && x0 >= 0) s += hy * hx * to_f32<T>(vl[((long long)y0 * w + x0) * pix]);
                if (y0 >= 0 && x0 + 1 <= w - 1) s += hy * lx * to_f32<T>(vl[((long long)y0 * w + x0 + 1) * pix]);
                if (y0 + 1 <= h - 1 && x0 >= 0) s += ly * hx * to_f32<T>(vl[((long long)(y0 + 1) * w + x0) * pix]);
                if (y0 + 1 <= h - 1 && x0 + 1 <= w - 1)
                    s += ly * lx * to_f32<T>(vl[((long long)(y0 + 1) * w + x0 + 1) * pix]);
                acc += s * a;
            }
        }
        out[idx] = from_f32<T>(acc);
    }
}

static bool fast_path(int H, int D, int L, int P)
{
    return H == kHeads && D == kHeadDim && P == kPoints && L >= 1 && L <= kMaxLevels;
}

template <typename T>
static int msda_forward(const T *value, const int64_t *shapes, const int64_t *level_start, const float *loc,
                        const float *attn, int B, int S, int H, int D, int L, int Nq, int P, T *out,
                        hipStream_t stream)
{
    if (B < 0 || S < 0 || Nq < 0 || H <= 0 || D <= 0 || L <= 0 || P <= 0) return RDETR_ERR_INVALID_ARG;
    if (B == 0 || Nq == 0) return RDETR_OK;
    if (!value || !shapes || !level_start || !loc || !attn || !out) return RDETR_ERR_INVALID_ARG;
    if (S == 0) return RDETR_ERR_INVALID_ARG;

    const long long pixel_bytes = (long long)H * D * (long long)sizeof(T);
    const bool aligned = (reinterpret_cast<uintptr_t>(value) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0) &&
                         (reinterpret_cast<uintptr_t>(loc) % 8 == 0);
    if (fast_path(H, D, L, P) && aligned && (long long)S * pixel_bytes < (1ll << 31)) {
        // 4 queries per wave amortise the level-table load and give each block 16 neighbouring
        // queries; small problems (decoder, Nq = 300..900) drop to 1 so the grid still covers the chip.
        int qpw = 4;
        while (qpw > 1 && (long long)B * ((Nq + kWavesPerBlock * qpw - 1) / (kWavesPerBlock * qpw)) < 2048) qpw >>= 1;
        const int qpb = kWavesPerBlock * qpw;
        const int tiles = (Nq + qpb - 1) / qpb;
        const long long nblk = (long long)B * tiles;
        if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
        dim3 grid((unsigned)nblk), block(kWavesPerBlock * kWave);
        if (L == 4)
            hipLaunchKernelGGL((msda_fwd_wave_kernel<T, 4>), grid, block, 0, stream, value, shapes, level_start, loc,
                               attn, S, L, Nq, tiles, qpw, (int)nblk, out);
        else if (L == 5)
            hipLaunchKernelGGL((msda_fwd_wave_kernel<T, 5>), grid, block, 0, stream, value, shapes, level_start, loc,
                               attn, S, L, Nq, tiles, qpw, (int)nblk, out);
        else
            hipLaunchKernelGGL((msda_fwd_wave_kernel<T, 0>), grid, block, 0, stream, value, shapes, level_start, loc,
                               attn, S, L, Nq, tiles, qpw, (int)nblk, out);
        return launch_status();
    }
    const long long total = (long long)B * Nq * H * D;
    const long long want = (total + 255) / 256;
    dim3 grid((unsigned)(want < 16384 ? want : 16384)), block(256);
    hipLaunchKernelGGL((msda_fwd_generic_kernel<T>), grid, block, 0, stream, value, shapes, level_start, loc, attn, S,
                       H, D, L, Nq, P, total, out);
    return launch_status();
}

}  // namespace rdetr

extern "C" int rdetr_msda_fast_path(int H, int D, int L, int P) { return rdetr::fast_path(H, D, L, P) ? 1 : 0; }

extern "C" int rdetr_msda_forward_f32(const float *value, const int64_t *spatial_shapes,
                                      const int64_t *level_start_index, const float *sampling_loc,
                                      const float *attn_weight, int B, int S, int H, int D, int L, int Nq, int P,
                                      float *out, void *stream)
{
    return rdetr::msda_forward<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, B, S, H, D,
                                      L, Nq, P, out, static_cast<hipStream_t>(stream));
}

extern "C" int rdetr_msda_forward_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                                       const int64_t *level_start_index, const float *sampling_loc,
                                       const float *attn_weight, int B, int S, int H, int D, int L, int Nq, int P,
                                       uint16_t *out, void *stream)
{
    return rdetr::msda_forward<uint16_t>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, B, S, H,
                                         D, L, Nq, P, out, static_cast<hipStream_t>(stream));
}
