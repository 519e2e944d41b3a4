// Position-relation bias, BACKWARD with respect to the 1x1 projection -- hand-written for gfx950 (MI355X).
//
// The reference differentiates  pos_proj(get_sine_pos_embed(box_rel_encoding(src, tgt)))  through autograd
// (models/bricks/relation_transformer.py:527-532): the boxes carry no gradient (torch.no_grad, :527-529), the Conv2d(64, 8, 1)
// + ReLU does, so per image it keeps the [N1, N2, 64] sine features (207 MB at N = 900) for the backward GEMM
//     grad_weight[h][ch] = sum_{b,i,j} g[b,h,i,j] * feat[b,i,j,ch],   grad_bias[h] = sum g,   g = grad_out * (out > 0).
// Here the features are REGENERATED from the boxes (the forward kernel's arithmetic, csrc/relation.hip: logf, the IEEE quotient
// by dim_t, Cody-Waite sin / cos) and reduced on the fly: neither the features nor g ever exist in HBM beyond grad_out itself.
//
//   work split   blockIdx.z = image * 8 + (coordinate c, frequency half): a block owns the 8 channels  c*16 + 2k + {sin, cos},
//                k = 4*half .. 4*half + 3, of a tile of 32 rows x 256 key columns; a thread owns one key column, walks the
//                rows, and keeps 8 heads x 8 channels = 64 partial sums in registers (one log + four sin/cos per pair).
//   reduction    DETERMINISTIC: wave shuffles -> LDS across the 4 waves -> one 72-float record per block in a workspace ->
//                a second kernel adds the records in a fixed order.  No float atomics, no zero-initialised output
//                (SURVEY section 8 f4 asks for a deterministic alternative to atomics on the training path).
#include "common.h"

namespace rdetr {

namespace {

__device__ __forceinline__ void rb_sincos(float a, float &s, float &c)
{
    // csrc/relation.hip::sincos_cw (Cody-Waite pi/2 reduction + minimax polynomials, |a| < 2^15)
    const float n = __builtin_rintf(a * 0.63661977236758134308f);
    float r = __builtin_fmaf(n, -1.57079637050628662109375f, a);
    r = __builtin_fmaf(n, 4.37113900018624283e-8f, r);
    const float r2 = r * r;
    float ps = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, r2, -1.6666654611e-1f);
    ps = __builtin_fmaf(ps * r2, r, r);
    float pc = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, r2, 4.166664568298827e-2f);
    pc = __builtin_fmaf(pc * r2, r2, __builtin_fmaf(r2, -0.5f, 1.0f));
    const int q = (int)n;
    const float sv = (q & 1) ? pc : ps;
    const float cv = (q & 1) ? ps : pc;
    s = (q & 2) ? -sv : sv;
    c = ((q + 1) & 2) ? -cv : cv;
}

struct RbDimT {
    float v[8];        // temperature^(2k/16), k < 8
    float inv[8];      // correctly rounded reciprocals (relation.hip::div_by_const)
};

constexpr int kRbThreads = 256, kRbRows = 32, kRbHeads = 8, kRbK = 4;        // 4 frequencies x {sin, cos} = 8 channels per block
constexpr int kRbRecord = kRbHeads * 2 * kRbK + kRbHeads;                     // 64 weight partials + 8 bias partials

__global__ __launch_bounds__(kRbThreads) void relation_bias_bwd_kernel(
    const float *__restrict__ src, const float *__restrict__ tgt, const float *__restrict__ grad_out,
    const unsigned char *__restrict__ active, int N1, int N2, float scale, float eps, RbDimT dim_t, float *__restrict__ ws)
{
    __shared__ float red[kRbThreads / kWave][kRbRecord];
    const int zb = blockIdx.z >> 3, part = blockIdx.z & 7;                      // image, (coordinate, frequency half)
    const int c = part >> 1, k0 = (part & 1) * kRbK;
    const int j = blockIdx.x * kRbThreads + threadIdx.x;
    const int i0 = blockIdx.y * kRbRows;
    const bool jok = j < N2;
    const f32x4 t = *reinterpret_cast<const f32x4 *>(tgt + ((size_t)zb * N2 + (jok ? j : N2 - 1)) * 4);

    float acc[kRbK][2][kRbHeads];
    float gb[kRbHeads];
#pragma unroll
    for (int k = 0; k < kRbK; ++k)
#pragma unroll
        for (int sc = 0; sc < 2; ++sc)
#pragma unroll
            for (int h = 0; h < kRbHeads; ++h) acc[k][sc][h] = 0.f;
#pragma unroll
    for (int h = 0; h < kRbHeads; ++h) gb[h] = 0.f;

    const int rows = (N1 - i0 < kRbRows) ? N1 - i0 : kRbRows;
#pragma unroll 1
    for (int r = 0; r < rows; ++r) {
        const int i = i0 + r;
        const float *sb = src + ((size_t)zb * N1 + i) * 4;                      // uniform: scalar loads
        // the forward's encoding of coordinate c (relation_transformer.py:481-490; eps on both sizes in the ratio)
        float e;
        if (c < 2) {
            const float sz = sb[2 + c] + eps;
            e = logf(__builtin_fabsf(sb[c] - (c == 0 ? t.x : t.y)) / sz + 1.0f) * scale;
        } else {
            e = logf((sb[c] + eps) / ((c == 2 ? t.z : t.w) + eps)) * scale;
        }
        float g[kRbHeads];
#pragma unroll
        for (int h = 0; h < kRbHeads; ++h) {
            const size_t at = (((size_t)zb * kRbHeads + h) * N1 + i) * (size_t)N2 + (jok ? j : 0);
            g[h] = (jok && active[at]) ? grad_out[at] : 0.f;                    // ReLU': the forward's out > 0
            gb[h] += g[h];
        }
#pragma unroll
        for (int k = 0; k < kRbK; ++k) {
            const float d = dim_t.v[k0 + k], rd = dim_t.inv[k0 + k];
            const float q = e * rd;                                             // IEEE e / d (Markstein, relation.hip)
            const float ang = __builtin_fmaf(__builtin_fmaf(-q, d, e), rd, q);
            float s, co;
            rb_sincos(ang, s, co);
#pragma unroll
            for (int h = 0; h < kRbHeads; ++h) {
                acc[k][0][h] = __builtin_fmaf(g[h], s, acc[k][0][h]);
                acc[k][1][h] = __builtin_fmaf(g[h], co, acc[k][1][h]);
            }
        }
    }

    // wave reduction (fixed order), then the 4 waves through LDS, then the block's record
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < kRbK; ++k)
#pragma unroll
        for (int sc = 0; sc < 2; ++sc)
#pragma unroll
            for (int h = 0; h < kRbHeads; ++h) {
                float v = acc[k][sc][h];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) red[wave][(k * 2 + sc) * kRbHeads + h] = v;
            }
#pragma unroll
    for (int h = 0; h < kRbHeads; ++h) {
        float v = gb[h];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][kRbHeads * 2 * kRbK + h] = v;
    }
    __syncthreads();
    if (threadIdx.x < kRbRecord) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        ws[blk * kRbRecord + threadIdx.x] = v;
    }
}

// grad_weight[h][ch] / grad_bias[h] = the sum of the block records that carry it, in block order.  One wave per output, lanes
// stride over the records, fixed-order shuffle tree: the same bits from run to run.
__global__ __launch_bounds__(kWave) void relation_bias_bwd_reduce_kernel(const float *__restrict__ ws, int images, int blocks_xy,
                                                                        float *__restrict__ grad_weight, float *__restrict__ grad_bias)
{
    const int o = blockIdx.x;                                                   // 0 .. 511: (h, ch);  512 .. 519: bias h
    const int lane = threadIdx.x;
    int part, slot;
    if (o < 512) {
        const int h = o >> 6, ch = o & 63;
        const int c = ch >> 4, k = (ch & 15) >> 1, sc = ch & 1;
        part = c * 2 + (k >> 2);
        slot = ((k & 3) * 2 + sc) * kRbHeads + h;
    } else {
        part = 0;                                                               // every part sums the same g: take part 0's
        slot = kRbHeads * 2 * kRbK + (o - 512);
    }
    float v = 0.f;
    for (int b = 0; b < images; ++b) {
        const float *base = ws + ((size_t)(b * 8 + part) * blocks_xy) * kRbRecord + slot;
        for (int x = lane; x < blocks_xy; x += kWave) v += base[(size_t)x * kRbRecord];
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    if (lane == 0) {
        if (o < 512) grad_weight[o] = v;
        else if (grad_bias) grad_bias[o - 512] = v;
    }
}

}  // namespace

}  // namespace rdetr

using namespace rdetr;

static long long rb_blocks_xy(int N1, int N2)
{
    return (long long)((N2 + kRbThreads - 1) / kRbThreads) * ((N1 + kRbRows - 1) / kRbRows);
}

extern "C" long long rdetr_relation_bias_backward_workspace_bytes(int B, int N1, int N2)
{
    if (B <= 0 || N1 <= 0 || N2 <= 0) return 0;
    return (long long)B * 8 * rb_blocks_xy(N1, N2) * kRbRecord * (long long)sizeof(float);
}

extern "C" int rdetr_relation_bias_backward_f32(const float *src, const float *tgt, const float *grad_out, const uint8_t *active,
                                                int B, int N1, int N2, int Hh, int F, float scale, float temperature, float eps,
                                                float *workspace, float *grad_weight, float *grad_bias, void *stream)
{
    if (B < 0 || N1 < 0 || N2 < 0 || Hh <= 0 || F <= 0) return RDETR_ERR_INVALID_ARG;
    if (Hh != kRbHeads || F != 16) return RDETR_ERR_UNSUPPORTED;          // the model's configuration (relation_transformer.py:301)
    if (!grad_weight) return RDETR_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (B == 0 || N1 == 0 || N2 == 0) {                                        // empty sum: the reduction over no records writes zeros
        hipLaunchKernelGGL(relation_bias_bwd_reduce_kernel, dim3(512 + kRbHeads), dim3(kWave), 0, st, workspace, 0, 0, grad_weight, grad_bias);
        return launch_status();
    }
    if (!src || !tgt || !grad_out || !active || !workspace) return RDETR_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(tgt) % 16 != 0) return RDETR_ERR_INVALID_ARG;
    const long long bxy = rb_blocks_xy(N1, N2);
    const int gx = (N2 + kRbThreads - 1) / kRbThreads, gy = (N1 + kRbRows - 1) / kRbRows;
    if ((long long)B * 8 > 65535 || gy > 65535 || bxy > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    RbDimT dt;
    for (int k = 0; k < 8; ++k) {                                              // get_dim_t (position_encoding.py:101-105), fp32
        dt.v[k] = powf(temperature, (float)k * 2.0f / (float)F);
        dt.inv[k] = (float)(1.0 / (double)dt.v[k]);
    }
    hipLaunchKernelGGL(relation_bias_bwd_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)(B * 8)), dim3(kRbThreads), 0, st, src, tgt,
                       grad_out, active, N1, N2, scale, eps, dt, workspace);
    if (launch_status() != RDETR_OK) return RDETR_ERR_LAUNCH;
    hipLaunchKernelGGL(relation_bias_bwd_reduce_kernel, dim3(512 + kRbHeads), dim3(kWave), 0, st, workspace, B, (int)bxy, grad_weight,
                       grad_bias);
    return launch_status();
}
