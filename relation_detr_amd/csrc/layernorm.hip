// Residual add + LayerNorm in one pass, for the rows either side of the hot-path kernels (gfx950).
//
// Replaces the pairs  `x + sublayer(x)` -> nn.LayerNorm  of the reference's encoder / decoder layers
// (models/bricks/relation_transformer.py:262-276 encoder layer, :452-478 decoder layer, :360 decoder norm):
// two elementwise passes + a normalisation pass (5 tensor traversals) become one read of each operand and one write.
//
//   C = 256 (the model's embed_dim): half a wavefront per row, a lane owns 8 consecutive channels (16-byte bf16 / 2 x 16-byte
//   fp32 accesses), four rows per wave in flight; any other C <= 8192: one wavefront per row, strided scalar loop.
//   Statistics in fp32, two-pass in registers (mean, then the variance of the centred values), biased variance and
//   1/sqrt(var + eps) as torch.nn.functional.layer_norm; the sum x + r is NOT rounded to the storage type first.
// Bound: HBM (3 x rows x C x sizeof(T) bytes per call).
#include "common.h"

namespace rdetr {

template <typename T> struct LnIO;
template <> struct LnIO<float> {
    static __device__ __forceinline__ void load4(const float *p, float (&v)[4])
    {
        const f32x4 r = *reinterpret_cast<const f32x4 *>(p);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
    static __device__ __forceinline__ void store4(float *p, const float (&v)[4])
    {
        *reinterpret_cast<f32x4 *>(p) = f32x4{v[0], v[1], v[2], v[3]};
    }
    static __device__ __forceinline__ float load1(const float *p) { return *p; }
    static __device__ __forceinline__ void store1(float *p, float v) { *p = v; }
};
template <> struct LnIO<uint16_t> {
    static __device__ __forceinline__ void load4(const uint16_t *p, float (&v)[4])
    {
        const u32x2 r = *reinterpret_cast<const u32x2 *>(p);
        v[0] = __builtin_bit_cast(float, r.x << 16); v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, r.y << 16); v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void store4(uint16_t *p, const float (&v)[4])
    {
        u32x2 o;
        o.x = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
        o.y = f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16);
        *reinterpret_cast<u32x2 *>(p) = o;
    }
    static __device__ __forceinline__ float load1(const uint16_t *p) { return bf16_bits_to_f32(*p); }
    static __device__ __forceinline__ void store1(uint16_t *p, float v) { *p = (uint16_t)f32_to_bf16_bits(v); }
};

__device__ __forceinline__ float ln_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T> __device__ __forceinline__ float ln_round(float v);
template <> __device__ __forceinline__ float ln_round<float>(float v) { return v; }
template <> __device__ __forceinline__ float ln_round<uint16_t>(float v) { return bf16_bits_to_f32((uint16_t)f32_to_bf16_bits(v)); }

constexpr int kLnWaves = 4;
constexpr int kLnRowsPerWave = 4;          // C = 256 kernel: half a wave per row (32 lanes x 8 channels), 2 rows per half

// sum over the 32 lanes of a half wave
__device__ __forceinline__ float ln_half_sum(float v)
{
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T> struct LnIO8;
template <> struct LnIO8<float> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[8])
    {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(p), b = *reinterpret_cast<const f32x4 *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[8])
    {
        *reinterpret_cast<f32x4 *>(p) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4 *>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
};
template <> struct LnIO8<uint16_t> {
    static __device__ __forceinline__ void load(const uint16_t *p, float (&v)[8])
    {
        const u32x4 r = *reinterpret_cast<const u32x4 *>(p);
        v[0] = __builtin_bit_cast(float, r.x << 16); v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, r.y << 16); v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
        v[4] = __builtin_bit_cast(float, r.z << 16); v[5] = __builtin_bit_cast(float, r.z & 0xffff0000u);
        v[6] = __builtin_bit_cast(float, r.w << 16); v[7] = __builtin_bit_cast(float, r.w & 0xffff0000u);
    }
    static __device__ __forceinline__ void store(uint16_t *p, const float (&v)[8])
    {
        u32x4 o;
        o.x = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
        o.y = f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16);
        o.z = f32_to_bf16_bits(v[4]) | (f32_to_bf16_bits(v[5]) << 16);
        o.w = f32_to_bf16_bits(v[6]) | (f32_to_bf16_bits(v[7]) << 16);
        *reinterpret_cast<u32x4 *>(p) = o;
    }
};

// C == 256: a half wave owns a row (a lane 8 consecutive channels = one 16-byte bf16 access), a wave works on 4 rows with
// all of their loads issued before the first use (the one-row-per-wave version kept 1.5 KB in flight per wave and reached
// 3.7 TB/s; HBM needs more outstanding bytes than that).
template <typename T>
__global__ __launch_bounds__(kLnWaves *kWave) void add_layernorm256_kernel(const T *__restrict__ x, const T *__restrict__ r,
                                                                           const T *__restrict__ gamma,
                                                                           const T *__restrict__ beta, long long rows,
                                                                           long long ldx, long long ldr, long long ldo,
                                                                           float eps, T *__restrict__ out,
                                                                           const T *__restrict__ pos, long long ldp,
                                                                           T *__restrict__ out2, long long ldo2)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, c = (lane & 31) * 8;
    const long long row0 = ((long long)blockIdx.x * kLnWaves + (threadIdx.x >> 6)) * kLnRowsPerWave + half;
    float v[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long row = row0 + 2 * i;
        if (row < rows) {
            LnIO8<T>::load(x + row * ldx + c, v[i]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[i][k] = 0.f;
        }
    }
    if (r) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long row = row0 + 2 * i;
            if (row < rows) {
                float t[8];
                LnIO8<T>::load(r + row * ldr + c, t);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[i][k] += t[k];
            }
        }
    }
    float g[8], b[8];
    LnIO8<T>::load(gamma + c, g);
    LnIO8<T>::load(beta + c, b);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long row = row0 + 2 * i;
        // same summation tree per lane as the 4-channel version would not be required: statistics are fp32 either way
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[i][k];
        const float mean = ln_half_sum(s) * (1.0f / 256.0f);
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[i][k] -= mean;
            sq += v[i][k] * v[i][k];
        }
        const float rstd = 1.0f / sqrtf(ln_half_sum(sq) * (1.0f / 256.0f) + eps);
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = v[i][k] * rstd * g[k] + b[k];
        if (row < rows) {
            LnIO8<T>::store(out + row * ldo + c, y);
            if (out2) {                      // out2 = out + pos, from the STORED (rounded) normalised values: the bits of a separate add
                float pv[8];
                LnIO8<T>::load(pos + row * ldp + c, pv);
#pragma unroll
                for (int k = 0; k < 8; ++k) pv[k] += ln_round<T>(y[k]);
                LnIO8<T>::store(out2 + row * ldo2 + c, pv);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kLnWaves *kWave) void add_layernorm_generic_kernel(const T *__restrict__ x, const T *__restrict__ r,
                                                                                const T *__restrict__ gamma,
                                                                                const T *__restrict__ beta, long long rows,
                                                                                int C, long long ldx, long long ldr,
                                                                                long long ldo, float eps, T *__restrict__ out,
                                                                                const T *__restrict__ pos, long long ldp,
                                                                                T *__restrict__ out2, long long ldo2)
{
    const long long row = (long long)blockIdx.x * kLnWaves + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const T *xr = x + row * ldx, *rr = r ? r + row * ldr : nullptr;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += LnIO<T>::load1(xr + c) + (rr ? LnIO<T>::load1(rr + c) : 0.f);
    const float mean = ln_wave_sum(s) / (float)C;
    float sq = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = LnIO<T>::load1(xr + c) + (rr ? LnIO<T>::load1(rr + c) : 0.f) - mean;
        sq += d * d;
    }
    const float rstd = 1.0f / sqrtf(ln_wave_sum(sq) / (float)C + eps);
    for (int c = lane; c < C; c += 64) {
        const float d = LnIO<T>::load1(xr + c) + (rr ? LnIO<T>::load1(rr + c) : 0.f) - mean;
        const float y = d * rstd * LnIO<T>::load1(gamma + c) + LnIO<T>::load1(beta + c);
        LnIO<T>::store1(out + row * ldo + c, y);
        if (out2) LnIO<T>::store1(out2 + row * ldo2 + c, ln_round<T>(y) + LnIO<T>::load1(pos + row * ldp + c));
    }
}

template <typename T>
static int add_layernorm(const T *x, const T *r, const T *gamma, const T *beta, long long rows, int C, long long ldx,
                         long long ldr, long long ldo, float eps, T *out, hipStream_t stream, const T *pos = nullptr,
                         long long ldp = 0, T *out2 = nullptr, long long ldo2 = 0)
{
    if (rows < 0 || C <= 0 || C > 8192 || ldx < C || ldo < C || (r && ldr < C)) return RDETR_ERR_INVALID_ARG;
    if ((out2 != nullptr) != (pos != nullptr) || (out2 && (ldp < C || ldo2 < C))) return RDETR_ERR_INVALID_ARG;
    if (rows == 0) return RDETR_OK;
    if (!x || !gamma || !beta || !out) return RDETR_ERR_INVALID_ARG;
    const long long nblk = (rows + kLnWaves - 1) / kLnWaves;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    const long long nblk256 = (rows + kLnWaves * kLnRowsPerWave - 1) / (kLnWaves * kLnRowsPerWave);
    auto al16 = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    const long long a16 = 16 / (long long)sizeof(T);
    if (C == 256 && al16(x) && al16(out) && al16(gamma) && al16(beta) && (!r || al16(r)) && ldx % a16 == 0 && ldo % a16 == 0 &&
        (!r || ldr % a16 == 0) && (!out2 || (al16(pos) && al16(out2) && ldp % a16 == 0 && ldo2 % a16 == 0)))
        hipLaunchKernelGGL((add_layernorm256_kernel<T>), dim3((unsigned)nblk256), dim3(kLnWaves * kWave), 0, stream, x, r, gamma,
                           beta, rows, ldx, ldr, ldo, eps, out, pos, ldp, out2, ldo2);
    else
        hipLaunchKernelGGL((add_layernorm_generic_kernel<T>), dim3((unsigned)nblk), dim3(kLnWaves * kWave), 0, stream, x, r,
                           gamma, beta, rows, C, ldx, ldr, ldo, eps, out, pos, ldp, out2, ldo2);
    return launch_status();
}

}  // namespace rdetr

extern "C" int rdetr_add_layernorm_f32(const float *x, const float *residual, const float *gamma, const float *beta,
                                       long long rows, int C, float eps, float *out, void *stream)
{
    return rdetr::add_layernorm<float>(x, residual, gamma, beta, rows, C, C, C, C, eps, out, static_cast<hipStream_t>(stream));
}

extern "C" int rdetr_add_layernorm_strided_f32(const float *x, const float *residual, const float *gamma, const float *beta,
                                               long long rows, int C, long long ldx, long long ldr, long long ldo, float eps,
                                               float *out, void *stream)
{
    return rdetr::add_layernorm<float>(x, residual, gamma, beta, rows, C, ldx, ldr, ldo, eps, out, static_cast<hipStream_t>(stream));
}

extern "C" int rdetr_add_layernorm_bf16(const uint16_t *x, const uint16_t *residual, const uint16_t *gamma,
                                        const uint16_t *beta, long long rows, int C, float eps, uint16_t *out, void *stream)
{
    return rdetr::add_layernorm<uint16_t>(x, residual, gamma, beta, rows, C, C, C, C, eps, out, static_cast<hipStream_t>(stream));
}

extern "C" int rdetr_add_layernorm_strided_bf16(const uint16_t *x, const uint16_t *residual, const uint16_t *gamma,
                                                const uint16_t *beta, long long rows, int C, long long ldx, long long ldr,
                                                long long ldo, float eps, uint16_t *out, void *stream)
{
    return rdetr::add_layernorm<uint16_t>(x, residual, gamma, beta, rows, C, ldx, ldr, ldo, eps, out,
                                          static_cast<hipStream_t>(stream));
}

extern "C" int rdetr_add_layernorm_pos_f32(const float *x, const float *residual, const float *gamma, const float *beta,
                                           const float *pos, long long rows, int C, long long ldx, long long ldr, long long ldo,
                                           long long ldp, long long ldo2, float eps, float *out, float *out2, void *stream)
{
    if (!pos || !out2) return RDETR_ERR_INVALID_ARG;
    return rdetr::add_layernorm<float>(x, residual, gamma, beta, rows, C, ldx, ldr, ldo, eps, out, static_cast<hipStream_t>(stream),
                                       pos, ldp, out2, ldo2);
}

extern "C" int rdetr_add_layernorm_pos_bf16(const uint16_t *x, const uint16_t *residual, const uint16_t *gamma,
                                            const uint16_t *beta, const uint16_t *pos, long long rows, int C, long long ldx,
                                            long long ldr, long long ldo, long long ldp, long long ldo2, float eps, uint16_t *out,
                                            uint16_t *out2, void *stream)
{
    if (!pos || !out2) return RDETR_ERR_INVALID_ARG;
    return rdetr::add_layernorm<uint16_t>(x, residual, gamma, beta, rows, C, ldx, ldr, ldo, eps, out,
                                          static_cast<hipStream_t>(stream), pos, ldp, out2, ldo2);
}
