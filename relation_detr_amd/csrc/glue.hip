// Small fused elementwise kernels of the decoder's box bookkeeping (gfx950): each replaces a chain of ~8 torch launches
// on a [B, N, 4] tensor -- inside a HIP graph every launch still costs a few microseconds of dependency latency.
//
//   box_refine      sigmoid(delta + inverse_sigmoid(ref))                 models/bricks/relation_transformer.py:363-381
//                   with inverse_sigmoid(x) = log(clamp(x,0,1).clamp(min=eps) / (1 - clamp(x,0,1)).clamp(min=eps)), eps = 1e-3
//                   (util/misc.py:31-35); delta in fp32 or bf16, ref / out fp32
//   sine_pos_embed  get_sine_pos_embed(pos, num_pos_feats = F, temperature, scale = 2*pi, exchange_xy = True)
//                   (models/bricks/position_encoding.py:115-138): [rows, n] -> [rows, n * F], coordinate order (y, x, rest),
//                   channel = coord * F + 2k + {sin, cos};  (pos * scale) / dim_t[k] in fp32 as the reference rounds it
// Bound: launch latency (tens of KB of traffic).
#include "common.h"

namespace rdetr {

template <typename TD> __device__ __forceinline__ float glue_load(const TD *p);
template <> __device__ __forceinline__ float glue_load<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float glue_load<uint16_t>(const uint16_t *p) { return bf16_bits_to_f32(*p); }

template <typename TD>
__global__ __launch_bounds__(256) void box_refine_kernel(const TD *__restrict__ delta, const float *__restrict__ ref, long long n,
                                                         float eps, float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = ref[i];
    x = fminf(fmaxf(x, 0.f), 1.f);                       // NaN -> 0 like torch.clamp? torch propagates NaN: keep it
    if (ref[i] != ref[i]) x = ref[i];
    const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
    const float z = glue_load<TD>(delta + i) + logf(x1 / x2);
    out[i] = 1.f / (1.f + expf(-z));
}

template <typename TO> __device__ __forceinline__ void glue_store(TO *p, float v);
template <> __device__ __forceinline__ void glue_store<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void glue_store<uint16_t>(uint16_t *p, float v) { *p = (uint16_t)f32_to_bf16_bits(v); }

struct GlueDimT {
    float v[64];
};

template <typename TO>
__global__ __launch_bounds__(256) void sine_pos_embed_kernel(const float *__restrict__ pos, long long rows, int n, int F, float scale,
                                                             GlueDimT dim_t, TO *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (row, coord, k): a sin/cos pair
    const int half = F / 2;
    const long long total = rows * n * half;
    if (idx >= total) return;
    const int k = (int)(idx % half);
    const long long rc = idx / half;
    const int c = (int)(rc % n);
    const long long row = rc / n;
    const int src = c == 0 ? 1 : (c == 1 ? 0 : c);       // exchange_xy: output coordinate 0 is y, 1 is x
    const float a = (pos[row * n + src] * scale) / dim_t.v[k];
    TO *o = out + (row * n + c) * F + 2 * k;
    glue_store<TO>(o, sinf(a));
    glue_store<TO>(o + 1, cosf(a));
}

}  // namespace rdetr

using namespace rdetr;

extern "C" int rdetr_box_refine_f32(const void *delta, int delta_is_bf16, const float *ref, long long n, float eps, float *out,
                                    void *stream)
{
    if (n < 0) return RDETR_ERR_INVALID_ARG;
    if (n == 0) return RDETR_OK;
    if (!delta || !ref || !out) return RDETR_ERR_INVALID_ARG;
    const long long nblk = (n + 255) / 256;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (delta_is_bf16)
        hipLaunchKernelGGL((box_refine_kernel<uint16_t>), dim3((unsigned)nblk), dim3(256), 0, st, static_cast<const uint16_t *>(delta),
                           ref, n, eps, out);
    else
        hipLaunchKernelGGL((box_refine_kernel<float>), dim3((unsigned)nblk), dim3(256), 0, st, static_cast<const float *>(delta), ref, n,
                           eps, out);
    return launch_status();
}

extern "C" int rdetr_sine_pos_embed(const float *pos, long long rows, int n, int F, float temperature, float scale, void *out,
                                    int out_is_bf16, void *stream)
{
    if (rows < 0 || n <= 0 || F <= 0) return RDETR_ERR_INVALID_ARG;
    if ((F & 1) || F > 128) return RDETR_ERR_UNSUPPORTED;
    if (rows == 0) return RDETR_OK;
    if (!pos || !out) return RDETR_ERR_INVALID_ARG;
    GlueDimT dt;
    // get_dim_t: temperature ** (2 * (arange(F) // 2) / F), fp32 (position_encoding.py:101-105)
    for (int k = 0; k < F / 2; ++k) dt.v[k] = powf(temperature, 2.0f * (float)k / (float)F);
    const long long total = rows * n * (F / 2), nblk = (total + 255) / 256;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (out_is_bf16)
        hipLaunchKernelGGL((sine_pos_embed_kernel<uint16_t>), dim3((unsigned)nblk), dim3(256), 0, st, pos, rows, n, F, scale, dt,
                           static_cast<uint16_t *>(out));
    else
        hipLaunchKernelGGL((sine_pos_embed_kernel<float>), dim3((unsigned)nblk), dim3(256), 0, st, pos, rows, n, F, scale, dt,
                           static_cast<float *>(out));
    return launch_status();
}
