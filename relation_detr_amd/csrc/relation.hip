// Position-relation bias and bias+softmax kernels -- hand-written for gfx950 (MI355X).
//
// relation_bias: replaces PositionRelationEmbedding.forward (models/bricks/relation_transformer.py:520-532).
// The reference materialises [B,N,N,4] -> [B,N,N,64] (207 MB at N=900) -> 1x1 conv -> [B,8,N,N];
// here one lane owns one key column j and R query rows i, keeps the 4 log-encodings, the 64 sin/cos
// features and the 8 head accumulators in registers, and only the [B,8,N1,N2] result touches HBM
// (coalesced along j).  The 8x64 projection matrix is transposed once per block into LDS and read
// with broadcast ds_read_b128 (same address in every lane), each read feeding R*8 FMAs.
//
// Numerics follow the reference's fp32 op order: e = log(...), a = (e*scale)/dim_t[k] with an
// IEEE-rounded division (position_encoding.py:133), sin/cos of |a| up to ~1.2e3 rad through a
// Cody-Waite pi/2 reduction (two FMA steps) and minimax polynomials -- the fast hardware sin/cos
// are NOT accurate enough at these magnitudes (hard parts).
//
// bias_softmax: softmax(scores + bias) over rows of the decoder self-attention score matrix
// (nn.MultiheadAttention with a float attn_mask, relation_transformer.py:452-459), one wavefront
// per row, row kept in registers between the max / sum / normalise passes (one HBM read of scores
// and bias, one write), wave reductions by cross-lane shuffles.
#include "common.h"

namespace rdetr {

// ------------------------------------------------------------------------------------------- sincos
// sin and cos of a, |a| < 2^15.  r = a - n*pi/2 by FMA Cody-Waite (pi/2 = HI + LO, |n| < 2^15 so the
// neglected third term n*1.7e-15 is < 1e-10), then degree-7 / degree-8 minimax polynomials on
// [-pi/4, pi/4].  Max abs error ~1.5e-7 over |a| <= 2e3 (checked against fp64 in tests).
__device__ __forceinline__ void sincos_cw(float a, float &s, float &c)
{
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

struct DimT {
    float v[16];       // temperature^(2k/F), k < F/2 <= 16, computed on the host in fp32
    float inv[16];     // RN(1 / v[k]): the correctly rounded reciprocal (from the double quotient)
};

// a / d rounded to nearest, for a divisor whose correctly rounded reciprocal rd is known: q = RN(a * rd),
// r = a - q * d (exact in an FMA), RN(q + r * rd).  Markstein's theorem: the result equals the IEEE quotient whenever
// nothing over- or underflows (the one excluded divisor pattern, an all-ones significand, does not occur among
// temperature^(2k/F)); 3 instructions instead of the ~10 of the division sequence, same bits.
__device__ __forceinline__ float div_by_const(float a, float d, float rd)
{
    const float q = a * rd;
    const float r = __builtin_fmaf(-q, d, a);
    return __builtin_fmaf(r, rd, q);
}

constexpr int kRelRows = 4;        // query rows per lane
constexpr int kRelWaves = 4;

// Fast path: F = 16 (8 frequencies), HH = 8 heads.
// TABLES = true: the two size-ratio coordinates  log(w_i / w_j), log(h_i / h_j)  are differences of per-box terms, so their
// sine features come from per-box tables (relation_tables_kernel) by the angle-difference identities -- 4 multiply-adds
// instead of a division, an argument reduction and two polynomials per (pair, frequency, coordinate).
template <int F, int HH, bool TABLES>
__global__ __launch_bounds__(kRelWaves *kWave) void relation_bias_kernel(
    const float *__restrict__ src, const float *__restrict__ tgt, const float *__restrict__ Wp,
    const float *__restrict__ bp, int N1, int N2, float scale, float eps, DimT dim_t, float *__restrict__ out,
    const float *__restrict__ src_tab, const float *__restrict__ tgt_tab)
{
    constexpr int K = F / 2, CH = 4 * F;
    __shared__ f32x4 wT[CH * HH / 4];                      // [ch][h] transposed weights
    float *wTf = reinterpret_cast<float *>(wT);
    for (int t = threadIdx.x; t < CH * HH; t += blockDim.x) {
        const int h = t / CH, ch = t - h * CH;             // coalesced read of Wp[h][ch]
        wTf[ch * HH + h] = Wp[t];
    }
    __syncthreads();

    const int b = blockIdx.z;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * kWave + lane;
    const int i0 = (blockIdx.y * kRelWaves + wave) * kRelRows;
    if (i0 >= N1) return;
    const bool jok = j < N2;

    // tail lanes re-read the last box (never stored); no `bool ? vec : vec` (not a whole-vector select in clang)
    const f32x4 t = *reinterpret_cast<const f32x4 *>(tgt + ((size_t)b * N2 + (jok ? j : N2 - 1)) * 4);
    const float tw = t.z + eps, th = t.w + eps;

    float es[kRelRows][4];
    f32x2 acc[kRelRows][HH / 2];                           // head pairs: the projection runs on v_pk_fma_f32
#pragma unroll
    for (int r = 0; r < kRelRows; ++r) {
        const int i = (i0 + r < N1) ? i0 + r : N1 - 1;     // clamp: tail rows recompute the last row, not stored
        const float *sb = src + ((size_t)b * N1 + i) * 4;  // wave-uniform -> scalar loads
        const float sx = sb[0], sy = sb[1], sw = sb[2] + eps, sh = sb[3] + eps;
        es[r][0] = logf(__builtin_fabsf(sx - t.x) / sw + 1.0f) * scale;
        es[r][1] = logf(__builtin_fabsf(sy - t.y) / sh + 1.0f) * scale;
        es[r][2] = TABLES ? 0.f : logf(sw / tw) * scale;
        es[r][3] = TABLES ? 0.f : logf(sh / th) * scale;
#pragma unroll
        for (int h = 0; h < HH / 2; ++h) acc[r][h] = bp ? f32x2{bp[2 * h], bp[2 * h + 1]} : f32x2{0.f, 0.f};
    }

    // k is a real loop (it only indexes the kernarg table and LDS); c and r are unrolled so that
    // es[][] / acc[][] stay in registers (a fully unrolled body spills: 256 VGPRs + scratch).
    // per-box tables: [box][coordinate w / h][k][sin, cos] = 32 floats per box (F = 16)
    const float *ttab = TABLES ? tgt_tab + ((size_t)b * N2 + (jok ? j : N2 - 1)) * (2 * F) : nullptr;
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
        const float d = dim_t.v[k], rd = dim_t.inv[k];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float sn[kRelRows], cs[kRelRows];
            if (TABLES && c >= 2) {
                const f32x2 tsc = *reinterpret_cast<const f32x2 *>(ttab + ((c - 2) * K + k) * 2);       // sin, cos of the key box's term
#pragma unroll
                for (int r = 0; r < kRelRows; ++r) {
                    const int i = (i0 + r < N1) ? i0 + r : N1 - 1;
                    const float *st = src_tab + ((size_t)b * N1 + i) * (2 * F) + ((c - 2) * K + k) * 2;  // wave-uniform: scalar loads
                    const float ss = st[0], sc = st[1];
                    sn[r] = __builtin_fmaf(ss, tsc.y, -(sc * tsc.x));          // sin(a - b) = sin a cos b - cos a sin b
                    cs[r] = __builtin_fmaf(sc, tsc.y, ss * tsc.x);             // cos(a - b) = cos a cos b + sin a sin b
                }
            } else {
#pragma unroll
                for (int r = 0; r < kRelRows; ++r) sincos_cw(div_by_const(es[r][c], d, rd), sn[r], cs[r]);
            }
            const int ch = c * F + 2 * k;
            f32x2 ws[HH / 2], wc[HH / 2];
#pragma unroll
            for (int h4 = 0; h4 < HH / 4; ++h4) {
                const f32x4 a = wT[(ch * HH) / 4 + h4];
                const f32x4 bq = wT[((ch + 1) * HH) / 4 + h4];
                ws[2 * h4] = f32x2{a.x, a.y};  ws[2 * h4 + 1] = f32x2{a.z, a.w};
                wc[2 * h4] = f32x2{bq.x, bq.y}; wc[2 * h4 + 1] = f32x2{bq.z, bq.w};
            }
#pragma unroll
            for (int r = 0; r < kRelRows; ++r) {
                const f32x2 s2 = {sn[r], sn[r]}, c2 = {cs[r], cs[r]};
#pragma unroll
                for (int h = 0; h < HH / 2; ++h) {
                    acc[r][h] = __builtin_elementwise_fma(ws[h], s2, acc[r][h]);
                    acc[r][h] = __builtin_elementwise_fma(wc[h], c2, acc[r][h]);
                }
            }
        }
    }

    if (!jok) return;
#pragma unroll
    for (int r = 0; r < kRelRows; ++r) {
        if (i0 + r >= N1) break;
#pragma unroll
        for (int h = 0; h < HH / 2; ++h) {
            const f32x2 a = acc[r][h];
            out[(((size_t)b * HH + 2 * h) * N1 + (i0 + r)) * N2 + j] = a.x > 0.f ? a.x : 0.f;
            out[(((size_t)b * HH + 2 * h + 1) * N1 + (i0 + r)) * N2 + j] = a.y > 0.f ? a.y : 0.f;
        }
    }
}

// Per-box tables for TABLES = true: thread = (box, coordinate, k); the angle (log(size + eps) * scale) / dim_t[k] and its sine /
// cosine in DOUBLE precision, rounded once -- the fp32 reference rounds log(s1 / s2) * scale / dim_t at |angle| up to 1e3,
// i.e. to ~3e-5 rad; the table route is within that of it and closer to the exact value (tests hold both to 1e-4 of the
// fp32 and of the fp64 evaluation of the reference formula).  size + eps is formed in fp32 as the reference does.
struct DimTD {
    double v[16];
};
__global__ __launch_bounds__(256) void relation_tables_kernel(const float *__restrict__ src, long long nsrc, const float *__restrict__ tgt,
                                                             long long nboxes, int K, float scale, float eps, DimTD dim_t,
                                                             float *__restrict__ tab)
{
    // boxes 0 .. nsrc-1 come from `src`, nsrc .. nboxes-1 from `tgt`: one launch fills both tables (they are adjacent in `tab`)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nboxes * 2 * K) return;
    const int k = (int)(idx % K);
    const int c = (int)((idx / K) % 2);
    const long long box = idx / (2 * K);
    const float sz = (box < nsrc ? src[box * 4 + 2 + c] : tgt[(box - nsrc) * 4 + 2 + c]) + eps;
    const double a = log((double)sz) * (double)scale / dim_t.v[k];
    tab[idx * 2] = (float)sin(a);
    tab[idx * 2 + 1] = (float)cos(a);
}

// Generic fallback: any even F <= 32, Hh <= 16; one thread per (i, j).
__global__ __launch_bounds__(256) void relation_bias_generic_kernel(
    const float *__restrict__ src, const float *__restrict__ tgt, const float *__restrict__ Wp,
    const float *__restrict__ bp, int N1, int N2, int Hh, int F, float scale, float eps, DimT dim_t,
    float *__restrict__ out)
{
    const int b = blockIdx.z;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= N2) return;
    const float *sb = src + ((size_t)b * N1 + i) * 4;
    const float *tb = tgt + ((size_t)b * N2 + j) * 4;
    const float sw = sb[2] + eps, sh = sb[3] + eps;
    float e[4];
    e[0] = logf(__builtin_fabsf(sb[0] - tb[0]) / sw + 1.0f) * scale;
    e[1] = logf(__builtin_fabsf(sb[1] - tb[1]) / sh + 1.0f) * scale;
    e[2] = logf(sw / (tb[2] + eps)) * scale;
    e[3] = logf(sh / (tb[3] + eps)) * scale;
    float acc[16];
    for (int h = 0; h < Hh; ++h) acc[h] = bp ? bp[h] : 0.f;
    const int K = F / 2, CH = 4 * F;
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < K; ++k) {
            float s, co;
            sincos_cw(e[c] / dim_t.v[k], s, co);
            const int ch = c * F + 2 * k;
            for (int h = 0; h < Hh; ++h) {
                acc[h] = __builtin_fmaf(Wp[h * CH + ch], s, acc[h]);
                acc[h] = __builtin_fmaf(Wp[h * CH + ch + 1], co, acc[h]);
            }
        }
    for (int h = 0; h < Hh; ++h) out[(((size_t)b * Hh + h) * N1 + i) * N2 + j] = acc[h] > 0.f ? acc[h] : 0.f;
}

// ------------------------------------------------------------------------------------------- softmax
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr float kLog2e = 1.44269504088896340736f;
constexpr int kSmWaves = 4;

// VEC = 4: N2 % 4 == 0 and 16-byte aligned rows (float4 / uchar4 accesses); VEC = 1 otherwise.
// ITERS = ceil(N2 / (64*VEC)) rounded up to the instantiated size; the row lives in registers.
template <int VEC, int ITERS>
__global__ __launch_bounds__(kSmWaves *kWave) void bias_softmax_kernel(float *__restrict__ scores,
                                                                       const float *__restrict__ bias,
                                                                       const uint8_t *__restrict__ mask,
                                                                       long long rows, int N1, int N2)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * kSmWaves + wave;
    if (row >= rows) return;
    float *srow = scores + row * N2;
    const float *brow = bias ? bias + row * N2 : nullptr;
    const uint8_t *mrow = mask ? mask + (row % N1) * (long long)N2 : nullptr;
    const float ninf = -__builtin_inff();

    float v[ITERS][VEC];
    float mx = ninf;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int j = (it * kWave + lane) * VEC;
        if (j < N2) {
            if constexpr (VEC == 4) {
                f32x4 s = *reinterpret_cast<const f32x4 *>(srow + j);
                if (brow) s += *reinterpret_cast<const f32x4 *>(brow + j);
                if (mrow) {
                    const unsigned mm = *reinterpret_cast<const unsigned *>(mrow + j);
                    if (mm & 0x000000ffu) s.x = ninf;
                    if (mm & 0x0000ff00u) s.y = ninf;
                    if (mm & 0x00ff0000u) s.z = ninf;
                    if (mm & 0xff000000u) s.w = ninf;
                }
                v[it][0] = s.x; v[it][1] = s.y; v[it][2] = s.z; v[it][3] = s.w;
            } else {
                float s = srow[j];
                if (brow) s += brow[j];
                if (mrow && mrow[j]) s = ninf;
                v[it][0] = s;
            }
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[it][e] = ninf;
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) mx = fmaxf(mx, v[it][e]);
    }
    mx = wave_max(mx);
    // a fully masked row has mx = -inf: (-inf) - (-inf) = NaN propagates, as in torch.softmax
    const float mxs = mx * kLog2e;
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int j = (it * kWave + lane) * VEC + e;
            const float p = (j < N2) ? __builtin_amdgcn_exp2f(__builtin_fmaf(v[it][e], kLog2e, -mxs)) : 0.f;
            v[it][e] = p;
            sum += p;
        }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int j = (it * kWave + lane) * VEC;
        if (j < N2) {
            if constexpr (VEC == 4) {
                f32x4 p = {v[it][0] * inv, v[it][1] * inv, v[it][2] * inv, v[it][3] * inv};
                *reinterpret_cast<f32x4 *>(srow + j) = p;
            } else {
                srow[j] = v[it][0] * inv;
            }
        }
    }
}

// Rows too long for registers: three passes over memory (max, exp+sum, scale), still one wave per row.
__global__ __launch_bounds__(kSmWaves *kWave) void bias_softmax_long_kernel(float *__restrict__ scores,
                                                                            const float *__restrict__ bias,
                                                                            const uint8_t *__restrict__ mask,
                                                                            long long rows, int N1, int N2)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * kSmWaves + wave;
    if (row >= rows) return;
    float *srow = scores + row * N2;
    const float *brow = bias ? bias + row * N2 : nullptr;
    const uint8_t *mrow = mask ? mask + (row % N1) * (long long)N2 : nullptr;
    const float ninf = -__builtin_inff();
    float mx = ninf;
    for (int j = lane; j < N2; j += kWave) {
        float s = srow[j] + (brow ? brow[j] : 0.f);
        if (mrow && mrow[j]) s = ninf;
        srow[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    const float mxs = mx * kLog2e;
    float sum = 0.f;
    for (int j = lane; j < N2; j += kWave) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(srow[j], kLog2e, -mxs));
        srow[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < N2; j += kWave) srow[j] *= inv;
}

template <int VEC>
static void launch_softmax(int iters, dim3 grid, dim3 block, hipStream_t st, float *s, const float *b, const uint8_t *m,
                           long long rows, int N1, int N2)
{
    if (iters <= 1) hipLaunchKernelGGL((bias_softmax_kernel<VEC, 1>), grid, block, 0, st, s, b, m, rows, N1, N2);
    else if (iters <= 2) hipLaunchKernelGGL((bias_softmax_kernel<VEC, 2>), grid, block, 0, st, s, b, m, rows, N1, N2);
    else if (iters <= 4) hipLaunchKernelGGL((bias_softmax_kernel<VEC, 4>), grid, block, 0, st, s, b, m, rows, N1, N2);
    else if (iters <= 8) hipLaunchKernelGGL((bias_softmax_kernel<VEC, 8>), grid, block, 0, st, s, b, m, rows, N1, N2);
    else hipLaunchKernelGGL((bias_softmax_kernel<VEC, 16>), grid, block, 0, st, s, b, m, rows, N1, N2);
}

}  // namespace rdetr

using namespace rdetr;

extern "C" int rdetr_relation_bias_f32(const float *src, const float *tgt, const float *proj_weight,
                                       const float *proj_bias, int B, int N1, int N2, int Hh, int F, float scale,
                                       float temperature, float eps, float *out, void *stream)
{
    if (B < 0 || N1 < 0 || N2 < 0 || Hh <= 0 || F <= 0) return RDETR_ERR_INVALID_ARG;
    if ((F & 1) || F > 32 || Hh > 16) return RDETR_ERR_UNSUPPORTED;
    if (B == 0 || N1 == 0 || N2 == 0) return RDETR_OK;
    if (!src || !tgt || !proj_weight || !out) return RDETR_ERR_INVALID_ARG;
    if (B > 65535) return RDETR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DimT dt;
    for (int k = 0; k < 16; ++k) dt.v[k] = dt.inv[k] = 1.f;
    // get_dim_t (position_encoding.py:101-105): temperature ** (arange(F/2) * 2 / F), all in fp32
    for (int k = 0; k < F / 2; ++k) {
        dt.v[k] = powf(temperature, (float)k * 2.0f / (float)F);
        dt.inv[k] = (float)(1.0 / (double)dt.v[k]);
    }
    const bool aligned = reinterpret_cast<uintptr_t>(tgt) % 16 == 0;
    if (F == 16 && Hh == 8 && aligned) {
        const int rows_per_block = kRelWaves * kRelRows;
        dim3 grid((N2 + kWave - 1) / kWave, (N1 + rows_per_block - 1) / rows_per_block, B), block(kRelWaves * kWave);
        if (grid.y > 65535) return RDETR_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((relation_bias_kernel<16, 8, false>), grid, block, 0, st, src, tgt, proj_weight, proj_bias, N1, N2,
                           scale, eps, dt, out, nullptr, nullptr);
    } else {
        if (N1 > 65535) return RDETR_ERR_UNSUPPORTED;
        dim3 grid((N2 + 255) / 256, N1, B), block(256);
        hipLaunchKernelGGL(relation_bias_generic_kernel, grid, block, 0, st, src, tgt, proj_weight, proj_bias, N1, N2,
                           Hh, F, scale, eps, dt, out);
    }
    return launch_status();
}

namespace rdetr {
// Per-box sine tables of the two size coordinates for both box sets: workspace = [B*N1 + B*N2][2][F/2][sin, cos] floats.
int launch_relation_tables(const float *src, const float *tgt, int B, int N1, int N2, int F, float scale, float temperature, float eps,
                           float *workspace, hipStream_t st)
{
    if (F <= 0 || (F & 1) || F > 32) return RDETR_ERR_UNSUPPORTED;
    DimTD dtd;
    for (int k = 0; k < 16; ++k) dtd.v[k] = 1.0;
    for (int k = 0; k < F / 2; ++k) dtd.v[k] = (double)powf(temperature, (float)k * 2.0f / (float)F);
    const long long n1 = (long long)B * N1, n2 = (long long)B * N2;
    hipLaunchKernelGGL(relation_tables_kernel, dim3((unsigned)(((n1 + n2) * F + 255) / 256)), dim3(256), 0, st, src, n1, tgt, n1 + n2, F / 2,
                       scale, eps, dtd, workspace);
    return launch_status();
}
}  // namespace rdetr

extern "C" int rdetr_relation_bias_ws_f32(const float *src, const float *tgt, const float *proj_weight, const float *proj_bias,
                                          int B, int N1, int N2, int Hh, int F, float scale, float temperature, float eps,
                                          float *workspace, float *out, void *stream)
{
    // the table route serves the model's configuration only; anything else is the plain entry point's business
    const bool aligned = reinterpret_cast<uintptr_t>(tgt) % 16 == 0 && reinterpret_cast<uintptr_t>(workspace) % 8 == 0;
    if (!(F == 16 && Hh == 8 && aligned && workspace) || B <= 0 || N1 <= 0 || N2 <= 0 || B > 65535)
        return rdetr_relation_bias_f32(src, tgt, proj_weight, proj_bias, B, N1, N2, Hh, F, scale, temperature, eps, out, stream);
    if (!src || !tgt || !proj_weight || !out) return RDETR_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DimT dt;
    for (int k = 0; k < 16; ++k) dt.v[k] = dt.inv[k] = 1.f;
    for (int k = 0; k < F / 2; ++k) {
        dt.v[k] = powf(temperature, (float)k * 2.0f / (float)F);
        dt.inv[k] = (float)(1.0 / (double)dt.v[k]);
    }
    const int rc = launch_relation_tables(src, tgt, B, N1, N2, F, scale, temperature, eps, workspace, st);
    if (rc != RDETR_OK) return rc;
    const float *src_tab = workspace, *tgt_tab = workspace + (size_t)B * N1 * 2 * F;
    const int rows_per_block = kRelWaves * kRelRows;
    dim3 grid((N2 + kWave - 1) / kWave, (N1 + rows_per_block - 1) / rows_per_block, B), block(kRelWaves * kWave);
    if (grid.y > 65535) return RDETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((relation_bias_kernel<16, 8, true>), grid, block, 0, st, src, tgt, proj_weight, proj_bias, N1, N2, scale, eps,
                       dt, out, src_tab, tgt_tab);
    return launch_status();
}

extern "C" int rdetr_bias_softmax_f32(float *scores, const float *bias, const uint8_t *mask, int BH, int N1, int N2,
                                      void *stream)
{
    if (BH < 0 || N1 < 0 || N2 < 0) return RDETR_ERR_INVALID_ARG;
    if (BH == 0 || N1 == 0 || N2 == 0) return RDETR_OK;
    if (!scores) return RDETR_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long rows = (long long)BH * N1;
    const long long nblk = (rows + kSmWaves - 1) / kSmWaves;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    dim3 grid((unsigned)nblk), block(kSmWaves * kWave);
    const bool vec = (N2 % 4 == 0) && reinterpret_cast<uintptr_t>(scores) % 16 == 0 &&
                     (!bias || reinterpret_cast<uintptr_t>(bias) % 16 == 0) &&
                     (!mask || reinterpret_cast<uintptr_t>(mask) % 4 == 0);
    if (vec && N2 <= 16 * kWave * 4)
        launch_softmax<4>((N2 + kWave * 4 - 1) / (kWave * 4), grid, block, st, scores, bias, mask, rows, N1, N2);
    else if (N2 <= 16 * kWave)
        launch_softmax<1>((N2 + kWave - 1) / kWave, grid, block, st, scores, bias, mask, rows, N1, N2);
    else
        hipLaunchKernelGGL(bias_softmax_long_kernel, grid, block, 0, st, scores, bias, mask, rows, N1, N2);
    return launch_status();
}
