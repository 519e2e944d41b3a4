// Multi-scale deformable attention, backward (fp32) -- hand-written for gfx950 (MI355X).
//
// Replaces ms_deformable_col2im_cuda and, for head_dim 32, its kernel
// ms_deformable_col2im_gpu_kernel_shm_blocksize_aware_reduce_v1<T,32>
// (models/bricks/ops/cuda/ms_deform_im2col_cuda.cuh:290-392,1129-1150: block = 32 threads = one
// (b,q,head), 4 atomicAdd per thread per point, thread 0 serially sums 32 partials for the
// location / weight gradients).
//
// One wavefront owns one head and two consecutive queries, one channel per lane: the 4 corner rows are re-gathered
// (needed for d/dloc and d/dweight), `w_corner * g * attn` is scattered into grad_value with hardware fp32 atomics
// whose wave instruction covers two full 128-byte head rows (out-of-level corners carry offset 0x80000000 and are
// dropped by the buffer range check), and the per-head sums over D = 32 channels for grad_loc / grad_attn are
// 32-lane xor-shuffle reductions instead of a shared-memory pass.
//
// Float atomics make grad_value's summation order run-dependent (as in the reference).
//
// DETERMINISTIC mode (rdetr_msda_backward_det_f32; SURVEY section 8 f4 "deterministic alternative to atomics"): the same kernel
// writes one (pixel-row key, weight) record per sample corner instead of adding; the records are sorted by key with a stable
// radix sort (hipCUB / rocPRIM, caller-provided temporary storage) and `msda_bwd_segment_sum_kernel` adds each value row's
// records in sorted = original sample order -- a fixed order, the same bits from run to run.  grad_loc / grad_attn are
// shuffle-tree sums inside a wave in both modes.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace rdetr {

constexpr int kBH = 8, kBD = 32, kBP = 4, kBMaxL = 8, kBWaves = 4;
constexpr unsigned kBInvalid = 0x80000000u;
constexpr unsigned kBPixelBytes = kBH * kBD * 4, kBHeadBytes = kBD * 4;

struct BwdLevels {
    int h[kBMaxL], w[kBMaxL], start[kBMaxL];
};

// sum over the 32 lanes of one (query, head) row (lanes 0-31 / 32-63 reduce independently)
__device__ __forceinline__ float sum32(float v)
{
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    return v;
}

// One wavefront = ONE head x TWO consecutive queries; lane = pair*32 + channel, one channel per lane.
// Every atomic wave instruction therefore adds two full 128-byte head rows (the access shape that runs at the
// chip-wide float-atomic rate, MI355X_MICROARCH.md "Global float atomics"; the first version, 4 channels per lane
// x 8 heads, added 8 x 32 sparse bytes per instruction and reached 0.36 TB/s).
template <bool DET>
__global__ __launch_bounds__(kBWaves *kWave) void msda_bwd_wave_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, const float *__restrict__ grad_out, int S, int L,
    int Nq, int tiles_per_image, int nblk, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn, unsigned *__restrict__ rec_key, unsigned *__restrict__ rec_id, float *__restrict__ rec_w)
{
    const int LP = L * kBP;
    __shared__ BwdLevels lvl;
    // per wave, [point][pair]: corner offsets | {hy, hx, ly, lx} | {attn (0 if outside), W_l, H_l, inside}
    __shared__ u32x4 st_off[kBWaves][kBMaxL * kBP * 2];
    __shared__ f32x4 st_frac[kBWaves][kBMaxL * kBP * 2];
    __shared__ f32x4 st_misc[kBWaves][kBMaxL * kBP * 2];

    const int tid = threadIdx.x;
    if (tid < L) {
        lvl.h[tid] = (int)shapes[2 * tid];
        lvl.w[tid] = (int)shapes[2 * tid + 1];
        lvl.start[tid] = (int)level_start[tid];
    }
    __syncthreads();

    // logical block -> (image b, head m, tile of 2*kBWaves consecutive queries)
    const int logical = xcd_contiguous_block(blockIdx.x, nblk);
    const int bm = logical / tiles_per_image;
    const int tile = logical - bm * tiles_per_image;
    const int b = bm / kBH, m = bm - b * kBH;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, pair = lane >> 5, c = lane & 31;
    const int q = (tile * kBWaves + wave) * 2 + pair;
    const bool qok = q < Nq;

    const size_t plane = (size_t)b * S * (kBH * kBD) + (size_t)m * kBD;
    const unsigned nrec = (unsigned)S * kBPixelBytes - (unsigned)m * kBHeadBytes;
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(value) + plane, 0, nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(grad_value + plane, 0, nrec, 0x00020000);
    const unsigned lane_off = (unsigned)c * 4u;

    const size_t row = (size_t)b * Nq + (qok ? q : 0);
    const size_t hrow = (row * kBH + m) * (size_t)LP;
    u32x4 *soff = st_off[wave];
    f32x4 *sfrac = st_frac[wave];
    f32x4 *smisc = st_misc[wave];

    // ---- set-up: lane (pair, c) prepares point c (and c + 32 for L > 8 ... LP <= 32) of its query ---------------
    if (c < LP) {
        const int pt = c;
        const f32x2 xy = *reinterpret_cast<const f32x2 *>(loc + (hrow + pt) * 2);
        const float a = attn[hrow + pt];
        const int l = pt / kBP;
        const int h = lvl.h[l], w = lvl.w[l];
        const float x = xy.x * (float)w - 0.5f, y = xy.y * (float)h - 0.5f;
        const bool inside = qok && (y > -1.f) && (x > -1.f) && (y < (float)h) && (x < (float)w);
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = inside ? (int)xf : 0, y0 = inside ? (int)yf : 0;
        const float lx = inside ? x - xf : 0.f, ly = inside ? y - yf : 0.f;   // NaN-safe
        const bool okx0 = inside && x0 >= 0, okx1 = inside && x0 + 1 <= w - 1;
        const bool oky0 = y0 >= 0, oky1 = y0 + 1 <= h - 1;
        const unsigned base = (unsigned)(lvl.start[l] + y0 * w + x0) * kBPixelBytes;
        const unsigned rowb = (unsigned)w * kBPixelBytes;
        u32x4 o;
        o.x = (okx0 && oky0) ? base : kBInvalid;
        o.y = (okx1 && oky0) ? base + kBPixelBytes : kBInvalid;
        o.z = (okx0 && oky1) ? base + rowb : kBInvalid;
        o.w = (okx1 && oky1) ? base + rowb + kBPixelBytes : kBInvalid;
        soff[pt * 2 + pair] = o;
        sfrac[pt * 2 + pair] = f32x4{1.f - ly, 1.f - lx, ly, lx};
        smisc[pt * 2 + pair] = f32x4{inside ? a : 0.f, (float)w, (float)h, inside ? 1.f : 0.f};
        if constexpr (DET) {
            // one record per corner: key = the (image, pixel, head) row of grad_value it adds to (all ones: no row), weight =
            // bilinear weight * attention weight; the record's position IS its identity (query, head, point, corner)
            if (qok) {
                const unsigned rbase = (unsigned)((size_t)b * S) * kBH + (unsigned)m;
                const float hy = 1.f - ly, hx = 1.f - lx, aa = inside ? a : 0.f;
                auto key = [&](unsigned off) { return off == kBInvalid ? 0xffffffffu : rbase + (off / kBPixelBytes) * kBH; };
                const size_t rec = (hrow + pt) * 4;
                *reinterpret_cast<u32x4 *>(rec_key + rec) = u32x4{key(o.x), key(o.y), key(o.z), key(o.w)};
                *reinterpret_cast<u32x4 *>(rec_id + rec) = u32x4{(unsigned)rec, (unsigned)rec + 1u, (unsigned)rec + 2u, (unsigned)rec + 3u};
                *reinterpret_cast<f32x4 *>(rec_w + rec) = f32x4{(hy * hx) * aa, (hy * lx) * aa, (ly * hx) * aa, (ly * lx) * aa};
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const float g = grad_out[row * (kBH * kBD) + m * kBD + c];
    float my_ga = 0.f, my_gx = 0.f, my_gy = 0.f;                  // results of point `c`, kept by lane c of the pair

#pragma unroll 2
    for (int pt = 0; pt < LP; ++pt) {
        const u32x4 o = soff[pt * 2 + pair];
        const f32x4 fr = sfrac[pt * 2 + pair];      // hy, hx, ly, lx
        const f32x4 mi = smisc[pt * 2 + pair];      // attn (0 if outside), W, H, inside
        const float v00 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_v, o.x + lane_off, 0, 0));
        const float v01 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_v, o.y + lane_off, 0, 0));
        const float v10 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_v, o.z + lane_off, 0, 0));
        const float v11 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_v, o.w + lane_off, 0, 0));
        const float hy = fr.x, hx = fr.y, ly = fr.z, lx = fr.w;
        const float ga = g * mi.x;                                   // top_grad * attn_weight
        if constexpr (!DET) {
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((hy * hx) * ga, rs_g, o.x + lane_off, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((hy * lx) * ga, rs_g, o.y + lane_off, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((ly * hx) * ga, rs_g, o.z + lane_off, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((ly * lx) * ga, rs_g, o.w + lane_off, 0, 0);
        }
        // d/dx and d/dy of the bilinear sample (ms_deform_im2col_cuda.cuh:102-141)
        const float dxs = hy * (v01 - v00) + ly * (v11 - v10);
        const float dys = hx * (v10 - v00) + lx * (v11 - v01);
        const float smp = (hy * hx) * v00 + (hy * lx) * v01 + (ly * hx) * v10 + (ly * lx) * v11;
        const float ga_w = sum32(g * smp) * mi.w;                    // grad wrt attention weight
        const float gx = sum32(ga * dxs) * mi.y;                     // * W_l
        const float gy = sum32(ga * dys) * mi.z;                     // * H_l
        if (c == pt) {
            my_ga = ga_w;
            my_gx = gx;
            my_gy = gy;
        }
    }
    if (qok && c < LP) {                                              // 16-20 contiguous floats per query-head
        grad_attn[hrow + c] = my_ga;
        *reinterpret_cast<f32x2 *>(grad_loc + (hrow + c) * 2) = f32x2{my_gx, my_gy};
    }
}

// Deterministic mode, second half: grad_value row r = (image, pixel, head) is the sum of ITS records in sorted order.  Half a wave
// per row (lane = channel); the segment [lo, hi) of the sorted keys by binary search (uniform per half wave); the record gives the
// weight and, through its position, the (image, query, head) row of grad_out.  Every row is written (empty segment: zeros), so
// grad_value needs no zero-initialisation in this mode.
__global__ __launch_bounds__(256) void msda_bwd_segment_sum_kernel(const unsigned *__restrict__ skey, const unsigned *__restrict__ sid,
                                                                  const float *__restrict__ rec_w, const float *__restrict__ grad_out,
                                                                  long long nrec, long long nrows, int LP, float *__restrict__ grad_value)
{
    const long long r = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int c = threadIdx.x & 31;
    if (r >= nrows) return;
    auto lower = [&](unsigned k) {                                              // first i with skey[i] >= k
        long long lo = 0, hi = nrec;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (skey[mid] < k) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const long long lo = lower((unsigned)r), hi = lower((unsigned)r + 1u);
    float acc = 0.f;
    for (long long i = lo; i < hi; ++i) {
        const unsigned id = sid[i];
        const unsigned rowhm = (id >> 2) / (unsigned)LP;                       // (image * Nq + query) * 8 + head
        acc = __builtin_fmaf(rec_w[id], grad_out[(size_t)rowhm * kBD + c], acc);
    }
    grad_value[(size_t)r * kBD + c] = acc;
}

// Generic fallback: one thread per (b, q, head, point), loops over D; any (H, D, L, P).
__global__ __launch_bounds__(256) void msda_bwd_generic_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, const float *__restrict__ grad_out, int S, int H,
    int D, int L, int Nq, int P, long long total, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < total;
         k += (long long)gridDim.x * blockDim.x) {
        const int l = (int)((k / P) % L);
        const long long r = k / ((long long)L * P);          // (b*Nq + q)*H + m
        const int m = (int)(r % H);
        const long long bq = r / H;
        const long long b = bq / Nq;
        const long long pix = (long long)H * D;
        const int h = (int)shapes[2 * l], w = (int)shapes[2 * l + 1];
        const float x = loc[2 * k] * (float)w - 0.5f, y = loc[2 * k + 1] * (float)h - 0.5f;
        float g_a = 0.f, g_x = 0.f, g_y = 0.f;
        if ((y > -1.f) && (x > -1.f) && (y < (float)h) && (x < (float)w)) {
            const float a = attn[k];
            const float xf = floorf(x), yf = floorf(y);
            const int x0 = (int)xf, y0 = (int)yf;
            const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
            const bool k00 = y0 >= 0 && x0 >= 0, k01 = y0 >= 0 && x0 + 1 <= w - 1;
            const bool k10 = y0 + 1 <= h - 1 && x0 >= 0, k11 = y0 + 1 <= h - 1 && x0 + 1 <= w - 1;
            const long long o00 = b * S * pix + (level_start[l] + (long long)y0 * w + x0) * pix + (long long)m * D;
            const long long o01 = o00 + pix, o10 = o00 + (long long)w * pix, o11 = o10 + pix;
            const float *go = grad_out + bq * pix + (long long)m * D;
            for (int c = 0; c < D; ++c) {
                const float gc = go[c], ga = gc * a;
                const float v00 = k00 ? value[o00 + c] : 0.f, v01 = k01 ? value[o01 + c] : 0.f;
                const float v10 = k10 ? value[o10 + c] : 0.f, v11 = k11 ? value[o11 + c] : 0.f;
                if (k00) atomicAdd(grad_value + o00 + c, hy * hx * ga);
                if (k01) atomicAdd(grad_value + o01 + c, hy * lx * ga);
                if (k10) atomicAdd(grad_value + o10 + c, ly * hx * ga);
                if (k11) atomicAdd(grad_value + o11 + c, ly * lx * ga);
                g_a += gc * (hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11);
                g_x += ga * (hy * (v01 - v00) + ly * (v11 - v10));
                g_y += ga * (hx * (v10 - v00) + lx * (v11 - v01));
            }
            g_x *= (float)w;
            g_y *= (float)h;
        }
        grad_attn[k] = g_a;
        grad_loc[2 * k] = g_x;
        grad_loc[2 * k + 1] = g_y;
    }
}

}  // namespace rdetr

using namespace rdetr;

extern "C" int rdetr_msda_backward_f32(const float *value, const int64_t *spatial_shapes,
                                       const int64_t *level_start_index, const float *sampling_loc,
                                       const float *attn_weight, const float *grad_out, int B, int S, int H, int D,
                                       int L, int Nq, int P, float *grad_value, float *grad_sampling_loc,
                                       float *grad_attn_weight, void *stream)
{
    if (B < 0 || S < 0 || Nq < 0 || H <= 0 || D <= 0 || L <= 0 || P <= 0) return RDETR_ERR_INVALID_ARG;
    if (B == 0 || Nq == 0) return RDETR_OK;
    if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !grad_out || !grad_value ||
        !grad_sampling_loc || !grad_attn_weight || S == 0)
        return RDETR_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto al16 = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    const bool fast = H == kBH && D == kBD && P == kBP && L <= kBMaxL && al16(value) && al16(grad_out) &&
                      al16(grad_value) && reinterpret_cast<uintptr_t>(sampling_loc) % 8 == 0 &&
                      reinterpret_cast<uintptr_t>(grad_sampling_loc) % 8 == 0 &&
                      (long long)S * kBPixelBytes < (1ll << 31);
    if (fast) {
        const int tiles = (Nq + 2 * kBWaves - 1) / (2 * kBWaves);
        const long long nblk = (long long)B * kBH * tiles;
        if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(msda_bwd_wave_kernel<false>, dim3((unsigned)nblk), dim3(kBWaves * kWave), 0, st, value,
                           spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_out, S, L, Nq, tiles,
                           (int)nblk, grad_value, grad_sampling_loc, grad_attn_weight, nullptr, nullptr, nullptr);
        return launch_status();
    }
    const long long total = (long long)B * Nq * H * L * P;
    const long long want = (total + 255) / 256;
    hipLaunchKernelGGL(msda_bwd_generic_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, st, value,
                       spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_out, S, H, D, L, Nq, P, total,
                       grad_value, grad_sampling_loc, grad_attn_weight);
    return launch_status();
}

// ---- deterministic mode --------------------------------------------------------------------------------------------------------
namespace {
struct DetLayout {
    long long nrec, nrows;
    size_t off_key, off_id, off_w, off_skey, off_sid, off_tmp, tmp_bytes, total;
    int key_bits;
};
// workspace = rec_key | rec_id | rec_w | sorted_key | sorted_id | radix-sort temporary storage (sizes from hipCUB itself: a
// host-side query, nothing is launched)
bool det_layout(int B, int S, int L, int Nq, DetLayout &d)
{
    d.nrec = (long long)B * Nq * kBH * L * kBP * 4;
    d.nrows = (long long)B * S * kBH;
    if (d.nrec >= (1ll << 31) || d.nrows >= 0xffffffffll) return false;
    d.key_bits = 32;                                                            // the all-ones "no row" key needs every bit
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t n4 = up((size_t)d.nrec * 4);
    d.off_key = 0; d.off_id = n4; d.off_w = 2 * n4; d.off_skey = 3 * n4; d.off_sid = 4 * n4; d.off_tmp = 5 * n4;
    d.tmp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, d.tmp_bytes, (const unsigned *)nullptr, (unsigned *)nullptr, (const unsigned *)nullptr,
                                           (unsigned *)nullptr, (int)d.nrec, 0, d.key_bits, (hipStream_t)0) != hipSuccess)
        return false;
    d.total = d.off_tmp + up(d.tmp_bytes);
    return true;
}
}  // namespace

extern "C" long long rdetr_msda_backward_det_workspace_bytes(int B, int S, int H, int D, int L, int Nq, int P)
{
    if (B <= 0 || S <= 0 || Nq <= 0 || H != kBH || D != kBD || P != kBP || L <= 0 || L > kBMaxL) return 0;
    DetLayout d;
    return det_layout(B, S, L, Nq, d) ? (long long)d.total : -1;
}

extern "C" int rdetr_msda_backward_det_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                           const float *sampling_loc, const float *attn_weight, const float *grad_out, int B, int S,
                                           int H, int D, int L, int Nq, int P, void *workspace, long long workspace_bytes,
                                           float *grad_value, float *grad_sampling_loc, float *grad_attn_weight, void *stream)
{
    if (B < 0 || S < 0 || Nq < 0 || H <= 0 || D <= 0 || L <= 0 || P <= 0) return RDETR_ERR_INVALID_ARG;
    if (H != kBH || D != kBD || P != kBP || L > kBMaxL) return RDETR_ERR_UNSUPPORTED;
    if (!grad_value) return RDETR_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (B == 0 || S == 0) return RDETR_OK;
    if (Nq == 0) {                                                              // no records: every row of grad_value is zero
        hipLaunchKernelGGL(msda_bwd_segment_sum_kernel, dim3((unsigned)(((long long)B * S * kBH + 7) / 8)), dim3(256), 0, st, nullptr, nullptr,
                           nullptr, nullptr, 0ll, (long long)B * S * kBH, L * kBP, grad_value);
        return launch_status();
    }
    if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !grad_out || !grad_sampling_loc ||
        !grad_attn_weight || !workspace)
        return RDETR_ERR_INVALID_ARG;
    auto al16 = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (!(al16(value) && al16(grad_out) && al16(grad_value) && al16(workspace) && reinterpret_cast<uintptr_t>(sampling_loc) % 8 == 0 &&
          reinterpret_cast<uintptr_t>(grad_sampling_loc) % 8 == 0 && (long long)S * kBPixelBytes < (1ll << 31)))
        return RDETR_ERR_UNSUPPORTED;
    DetLayout d;
    if (!det_layout(B, S, L, Nq, d)) return RDETR_ERR_UNSUPPORTED;
    if (workspace_bytes < (long long)d.total) return RDETR_ERR_INVALID_ARG;
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    unsigned *rec_key = reinterpret_cast<unsigned *>(ws + d.off_key), *rec_id = reinterpret_cast<unsigned *>(ws + d.off_id);
    float *rec_w = reinterpret_cast<float *>(ws + d.off_w);
    unsigned *skey = reinterpret_cast<unsigned *>(ws + d.off_skey), *sid = reinterpret_cast<unsigned *>(ws + d.off_sid);
    const int tiles = (Nq + 2 * kBWaves - 1) / (2 * kBWaves);
    const long long nblk = (long long)B * kBH * tiles;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(msda_bwd_wave_kernel<true>, dim3((unsigned)nblk), dim3(kBWaves * kWave), 0, st, value, spatial_shapes,
                       level_start_index, sampling_loc, attn_weight, grad_out, S, L, Nq, tiles, (int)nblk, grad_value, grad_sampling_loc,
                       grad_attn_weight, rec_key, rec_id, rec_w);
    if (launch_status() != RDETR_OK) return RDETR_ERR_LAUNCH;
    size_t tmp = d.tmp_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(ws + d.off_tmp, tmp, rec_key, skey, rec_id, sid, (int)d.nrec, 0, d.key_bits, st) != hipSuccess)
        return RDETR_ERR_LAUNCH;
    hipLaunchKernelGGL(msda_bwd_segment_sum_kernel, dim3((unsigned)((d.nrows + 7) / 8)), dim3(256), 0, st, skey, sid, rec_w, grad_out, d.nrec,
                       d.nrows, L * kBP, grad_value);
    return launch_status();
}
