// Dense projection with K = 256 (the model's embed_dim) for bf16 activations on gfx950:  out = act(X W^T + b).
//
// Replaces the library GEMM behind the path's K = 256 nn.Linear layers -- MSDA value_proj / output_proj / the merged
// sampling_offsets + attention_weights projection (models/bricks/ms_deform_attn.py:259-262) and the FFN's linear1 + ReLU
// (models/bricks/relation_transformer.py:226-233) -- for tall inputs (tens of thousands of rows).  With K this short the
// library kernels spend their time in tile prologues / epilogues (200-570 TFLOP/s, 2.3 TB/s of output for the FFN); the
// shape is really a streaming problem: read X once, write out once, and the whole weight slice fits in LDS.
//
//   workgroup   512 threads = 8 waves, persistent over row tiles; owns a slice of NT * 16 <= 256 output columns whose weights
//               (<= 128 KiB bf16) sit in LDS for the whole kernel, in MFMA-fragment order (every A-operand read is one
//               contiguous, conflict-free ds_read_b128 per lane)
//   wave        32 rows per step: X^T is the B operand straight from global memory (a lane's 8 consecutive k = 16 contiguous
//               bytes of its row; the next step's rows are prefetched into a second register set), the TRANSPOSED product
//               out^T = W X^T leaves lane (row, g) with 4 consecutive output columns per 16 x 16 tile; the slice's columns are
//               permuted over the tiles so that two tiles give 8 consecutive columns = one 16-byte bf16 store
//   math        v_mfma_f32_16x16x32_bf16, fp32 accumulators initialised with the bias, one rounding to bf16 at the end
// Bound: HBM (X in + out; the weights come from L2 once per workgroup).
#include <cstdlib>

#include "common.h"

namespace rdetr {

typedef __bf16 ln_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kLinK = 256;
constexpr int kLinThreads = 512;           // 8 waves = 2 per SIMD: one wave's LDS reads / packing overlap the other's MFMAs
constexpr int kLinRows = 32;               // rows per wave step

template <int NT, bool RELU>
__global__ __launch_bounds__(kLinThreads) void linear_k256_kernel(const uint16_t *__restrict__ x, long long ldx,
                                                                  const uint16_t *__restrict__ w, const uint16_t *__restrict__ bias,
                                                                  long long M, int N, uint16_t *__restrict__ out, long long ldo, int dbg,
                                                                  const unsigned char *__restrict__ row_mask, int hm_S)
{
    // hm_S != 0 (N == 256 = 8 heads x 32): `out` is HEAD-MAJOR [B, 8, hm_S, 32] -- row r = image r / hm_S, position r % hm_S --
    // and the rows whose `row_mask` byte is set are written as zeros: MSDA's value projection with the padding zero-fill
    // (ms_deform_attn.py:316-321) and the re-layout the gather kernels want folded into the store.  Tile pair u is head u and a
    // lane's 8 columns are 16 bytes of that head's row; the 16 rows x 4 lanes of a tile write 1 KiB contiguous.
    extern __shared__ __attribute__((aligned(16))) unsigned char lin_lds[];
    u32x4 *wl = reinterpret_cast<u32x4 *>(lin_lds);                           // [NT][8 k-steps][64 lanes] 16-byte fragments
    float *bl = reinterpret_cast<float *>(lin_lds + NT * 8 * 64 * 16);        // [NT * 16] bias, natural column order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * (NT * 16);                                    // first column of this workgroup's slice

    // column of the slice carried by row m of tile t:  32 (t >> 1) + 8 (m >> 2) + 4 (t & 1) + (m & 3)
    {
        // weights -> LDS: rows are read whole (32 lanes x 16 B = one 512-byte row, coalesced) and scattered into fragment order:
        // piece p of slice column c goes to fragment (tile t, k-step p >> 2, lane 16 (p & 3) + m)
        constexpr int kPer = NT * 16 * 32 / kLinThreads;                      // 16-byte pieces per thread (16 / 12), all in flight
        u32x4 piece[kPer];
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int idx = i * kLinThreads + tid, c = idx >> 5, p = idx & 31;
            piece[i] = n0 + c < N ? *reinterpret_cast<const u32x4 *>(w + (size_t)(n0 + c) * kLinK + 8 * p) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int idx = i * kLinThreads + tid, c = idx >> 5, p = idx & 31;
            const int t = 2 * (c >> 5) + ((c >> 2) & 1), m = 4 * ((c >> 3) & 3) + (c & 3);
            wl[(t * 8 + (p >> 2)) * 64 + 16 * (p & 3) + m] = piece[i];
        }
    }
    for (int i = tid; i < NT * 16; i += kLinThreads) bl[i] = (bias && n0 + i < N) ? bf16_bits_to_f32(bias[n0 + i]) : 0.f;
    __syncthreads();

    const long long nsteps = (M + kLinRows - 1) / kLinRows;
    constexpr int kWaves = kLinThreads / 64;
    const long long stride = (long long)gridDim.x * kWaves;
    long long step = (long long)blockIdx.x * kWaves + wave;

    u32x4 xc[2][8], xn[2][8];
    auto load_rows = [&](long long st, u32x4 (&dst)[2][8]) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const long long row = st * kLinRows + cb * 16 + col;
            const uint16_t *p = x + row * ldx + 8 * g;
#pragma unroll
            for (int s = 0; s < 8; ++s)
                dst[cb][s] = (st < nsteps && row < M) ? *reinterpret_cast<const u32x4 *>(p + 32 * s) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    load_rows(step, xc);
    for (; step < nsteps; step += stride) {
        load_rows(step + stride, xn);                                         // prefetch (all zeros past the end)
        const long long row_a = step * kLinRows + col, row_b = row_a + 16;
        uint16_t *oa = out + row_a * ldo + n0 + 8 * g, *ob = out + row_b * ldo + n0 + 8 * g;
        bool zero_a = false, zero_b = false;
        if (hm_S) {
            const unsigned ia = (unsigned)row_a / (unsigned)hm_S, ib = (unsigned)row_b / (unsigned)hm_S;
            oa = out + ((long long)ia * 8 * hm_S + (row_a - (long long)ia * hm_S)) * 32 + 8 * g;           // + head * hm_S * 32
            ob = out + ((long long)ib * 8 * hm_S + (row_b - (long long)ib * hm_S)) * 32 + 8 * g;
            if (row_mask) {
                zero_a = row_a < M && row_mask[row_a] != 0;
                zero_b = row_b < M && row_mask[row_b] != 0;
            }
        }
        // two tiles (= 8 consecutive output columns per lane) at a time: 4 independent accumulator chains, stored as soon as they
        // are complete, so that only 16 accumulator registers are live and two waves fit a SIMD
#pragma unroll
        for (int u = 0; u < NT / 2; ++u) {
            f32x4 acc[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bl + 32 * u + 8 * g + 4 * e);
                acc[e][0] = b4;
                acc[e][1] = b4;
            }
            if (!(dbg & 2))
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const ln_bf16x8 a = __builtin_bit_cast(ln_bf16x8, wl[((2 * u + e) * 8 + s) * 64 + lane]);
                    acc[e][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(ln_bf16x8, xc[0][s]), acc[e][0], 0, 0, 0);
                    acc[e][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(ln_bf16x8, xc[1][s]), acc[e][1], 0, 0, 0);
                }
            }
            if (n0 + 32 * u + 8 * g < N) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    f32x4 lo = acc[0][cb], hi = acc[1][cb];
                    if (RELU) {
                        lo.x = fmaxf(lo.x, 0.f); lo.y = fmaxf(lo.y, 0.f); lo.z = fmaxf(lo.z, 0.f); lo.w = fmaxf(lo.w, 0.f);
                        hi.x = fmaxf(hi.x, 0.f); hi.y = fmaxf(hi.y, 0.f); hi.z = fmaxf(hi.z, 0.f); hi.w = fmaxf(hi.w, 0.f);
                    }
                    u32x4 pk;
                    pk.x = f32_to_bf16_bits(lo.x) | (f32_to_bf16_bits(lo.y) << 16);
                    pk.y = f32_to_bf16_bits(lo.z) | (f32_to_bf16_bits(lo.w) << 16);
                    pk.z = f32_to_bf16_bits(hi.x) | (f32_to_bf16_bits(hi.y) << 16);
                    pk.w = f32_to_bf16_bits(hi.z) | (f32_to_bf16_bits(hi.w) << 16);
                    if ((cb ? zero_b : zero_a)) pk = u32x4{0u, 0u, 0u, 0u};
                    const long long col_off = hm_S ? (long long)u * hm_S * 32 : 32 * u;
                    if ((cb ? row_b : row_a) < M && !((dbg & 1) && pk.x != 0x12345u)) *reinterpret_cast<u32x4 *>((cb ? ob : oa) + col_off) = pk;
                }
            }
            __builtin_amdgcn_sched_barrier(0);                                // keep the next pair's LDS reads from piling up here
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int s = 0; s < 8; ++s) xc[cb][s] = xn[cb][s];
    }
}

template <int NT, bool RELU>
static int linear_launch(const uint16_t *x, long long ldx, const uint16_t *w, const uint16_t *bias, long long M, int N,
                         uint16_t *out, long long ldo, int chunks, hipStream_t st, const unsigned char *row_mask = nullptr, int hm_S = 0)
{
    auto kern = linear_k256_kernel<NT, RELU>;
    constexpr int lds = NT * 8 * 64 * 16 + NT * 16 * 4;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
    const long long steps = (M + kLinRows - 1) / kLinRows;
    long long gx = (steps + kLinThreads / 64 - 1) / (kLinThreads / 64);
    const long long cap = 256 / chunks > 0 ? 256 / chunks : 1;                // about one resident workgroup per CU
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)chunks), dim3(kLinThreads), (size_t)lds, st, x, ldx, w, bias, M, N, out,
                       ldo, 0, row_mask, hm_S);
    return launch_status();
}


// ---- N = 256 projection with the layer's residual + LayerNorm in the epilogue ------------------------------------------------
// out = LayerNorm(residual + (X W^T + b)): MSDA's output_proj followed by norm1(query + attn) of the encoder layer
// (models/bricks/ms_deform_attn.py:372-376 + relation_transformer.py:262-271).  A row's 256 outputs sit in the 4 lanes
// (row, g = 0..3), 64 values each, so the statistics cost two xor-shuffles.  The weights come PACKED in fragment order
// (linear_pack_kernel) and are brought to LDS by 128 coalesced 1-KiB LDS-DMA instructions.
constexpr int kLnlThreads = 512;
constexpr int kLnlWaves = kLnlThreads / 64;

__global__ __launch_bounds__(256) void linear_pack_kernel(const uint16_t *__restrict__ w, u32x4 *__restrict__ packed)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;                           // [16 tiles][8 k-steps][64 lanes] 16-byte pieces
    if (idx >= 16 * 8 * 64) return;
    const int t = idx >> 9, s = (idx >> 6) & 7, l = idx & 63, m = l & 15, kb = l >> 4;
    const int n = 32 * (t >> 1) + 8 * (m >> 2) + 4 * (t & 1) + (m & 3);
    packed[idx] = *reinterpret_cast<const u32x4 *>(w + (size_t)n * kLinK + 32 * s + 8 * kb);
}

__global__ __launch_bounds__(kLnlThreads) void linear_ln_k256_kernel(const uint16_t *__restrict__ x, long long ldx,
                                                                     const uint16_t *__restrict__ packed,
                                                                     const uint16_t *__restrict__ bias,
                                                                     const uint16_t *__restrict__ res, long long ldr,
                                                                     const uint16_t *__restrict__ gamma,
                                                                     const uint16_t *__restrict__ beta, float eps, long long M,
                                                                     uint16_t *__restrict__ out, long long ldo)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lin_lds[];
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(lin_lds);               // 128 KiB of fragments
    float *bl = reinterpret_cast<float *>(lin_lds + 16 * 8 * 64 * 16);        // bias | gamma | beta, 256 each
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    {
        const unsigned lane_off = (unsigned)lane * 16u;
#pragma unroll 1
        for (int i = 0; i < 128 / kLnlWaves; ++i) {
            const int f = wave * (128 / kLnlWaves) + i;                       // uniform
            const unsigned m0v = (unsigned)f * 1024u;
            const unsigned char *src = reinterpret_cast<const unsigned char *>(packed) + f * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(m0v), "v"(lane_off), "s"(src) : "memory", "m0");
        }
    }
    if (tid < 256) {
        bl[tid] = bias ? bf16_bits_to_f32(bias[tid]) : 0.f;
        bl[256 + tid] = bf16_bits_to_f32(gamma[tid]);
        bl[512 + tid] = bf16_bits_to_f32(beta[tid]);
    }
    const long long nsteps = (M + kLinRows - 1) / kLinRows;
    long long step = (long long)blockIdx.x * kLnlWaves + wave;
    const long long stride = (long long)gridDim.x * kLnlWaves;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (; step < nsteps; step += stride) {
        const long long row_a = step * kLinRows + col, row_b = row_a + 16;
        u32x4 xc[2][8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            xc[0][s] = row_a < M ? *reinterpret_cast<const u32x4 *>(x + row_a * ldx + 32 * s + 8 * g) : u32x4{0u, 0u, 0u, 0u};
            xc[1][s] = row_b < M ? *reinterpret_cast<const u32x4 *>(x + row_b * ldx + 32 * s + 8 * g) : u32x4{0u, 0u, 0u, 0u};
        }
        f32x4 acc[16][2];
        {
            // one stream of 128 weight fragments (k-step i >> 4, tile i & 15), two MFMAs each, read two fragments ahead through a
            // ring of three registers with counted LDS waits (csrc/ffn.hip has the story); accumulators start from the inline
            // constant 0, the bias is added in the epilogue
            auto frag = [&](int i) -> u32x4 { return wl[((i & 15) * 8 + (i >> 4)) * 64 + lane]; };
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            u32x4 ring[3];
            ring[0] = frag(0);
            ring[1] = frag(1);
#pragma unroll
            for (int i = 0; i < 128; ++i) {
                if (i + 2 < 128) ring[(i + 2) % 3] = frag(i + 2);
                __builtin_amdgcn_sched_barrier(0);
                const int s = i >> 4, t = i & 15;
                const ln_bf16x8 a = __builtin_bit_cast(ln_bf16x8, ring[i % 3]);
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(ln_bf16x8, xc[0][s]), s ? acc[t][0] : zero4, 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(ln_bf16x8, xc[1][s]), s ? acc[t][1] : zero4, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const long long row = cb ? row_b : row_a;
            const bool ok = row < M;
            // v = round_bf16(projection) + residual (the unfused path stores the projection in bf16 first); fp32 two-pass statistics
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const u32x4 r = ok ? *reinterpret_cast<const u32x4 *>(res + row * ldr + 32 * u + 8 * g) : u32x4{0u, 0u, 0u, 0u};
                f32x4 &lo = acc[2 * u][cb], &hi = acc[2 * u + 1][cb];
                lo += *reinterpret_cast<const f32x4 *>(bl + 32 * u + 8 * g);
                hi += *reinterpret_cast<const f32x4 *>(bl + 32 * u + 8 * g + 4);
                const unsigned p0 = pack_bf16x2(lo.x, lo.y), p1 = pack_bf16x2(lo.z, lo.w), p2 = pack_bf16x2(hi.x, hi.y), p3 = pack_bf16x2(hi.z, hi.w);
                lo.x = __builtin_bit_cast(float, p0 << 16) + __builtin_bit_cast(float, r.x << 16);
                lo.y = __builtin_bit_cast(float, p0 & 0xffff0000u) + __builtin_bit_cast(float, r.x & 0xffff0000u);
                lo.z = __builtin_bit_cast(float, p1 << 16) + __builtin_bit_cast(float, r.y << 16);
                lo.w = __builtin_bit_cast(float, p1 & 0xffff0000u) + __builtin_bit_cast(float, r.y & 0xffff0000u);
                hi.x = __builtin_bit_cast(float, p2 << 16) + __builtin_bit_cast(float, r.z << 16);
                hi.y = __builtin_bit_cast(float, p2 & 0xffff0000u) + __builtin_bit_cast(float, r.z & 0xffff0000u);
                hi.z = __builtin_bit_cast(float, p3 << 16) + __builtin_bit_cast(float, r.w << 16);
                hi.w = __builtin_bit_cast(float, p3 & 0xffff0000u) + __builtin_bit_cast(float, r.w & 0xffff0000u);
                sum += ((lo.x + lo.y) + (lo.z + lo.w)) + ((hi.x + hi.y) + (hi.z + hi.w));
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / 256.0f);
            float sq = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                f32x4 &lo = acc[2 * u][cb], &hi = acc[2 * u + 1][cb];
                lo.x -= mean; lo.y -= mean; lo.z -= mean; lo.w -= mean;
                hi.x -= mean; hi.y -= mean; hi.z -= mean; hi.w -= mean;
                sq += ((lo.x * lo.x + lo.y * lo.y) + (lo.z * lo.z + lo.w * lo.w)) + ((hi.x * hi.x + hi.y * hi.y) + (hi.z * hi.z + hi.w * hi.w));
            }
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = 1.0f / sqrtf(sq * (1.0f / 256.0f) + eps);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 lo = acc[2 * u][cb], hi = acc[2 * u + 1][cb];
                const float *gp = bl + 256 + 32 * u + 8 * g, *bp = bl + 512 + 32 * u + 8 * g;
                const f32x4 g0 = *reinterpret_cast<const f32x4 *>(gp), g1 = *reinterpret_cast<const f32x4 *>(gp + 4);
                const f32x4 c0 = *reinterpret_cast<const f32x4 *>(bp), c1 = *reinterpret_cast<const f32x4 *>(bp + 4);
                u32x4 pk;
                pk.x = pack_bf16x2(lo.x * rstd * g0.x + c0.x, lo.y * rstd * g0.y + c0.y);
                pk.y = pack_bf16x2(lo.z * rstd * g0.z + c0.z, lo.w * rstd * g0.w + c0.w);
                pk.z = pack_bf16x2(hi.x * rstd * g1.x + c1.x, hi.y * rstd * g1.y + c1.y);
                pk.w = pack_bf16x2(hi.z * rstd * g1.z + c1.z, hi.w * rstd * g1.w + c1.w);
                if (ok) *reinterpret_cast<u32x4 *>(out + row * ldo + 32 * u + 8 * g) = pk;
            }
        }
    }
}

}  // namespace rdetr

using namespace rdetr;

// out[M, N] = act(x[M, 256] w[N, 256]^T + bias[N]); bf16 storage, fp32 accumulation.  N a multiple of 32; x / out rows ldx / ldo
// elements apart (multiples of 8, 16-byte aligned bases); bias nullable; relu: 0 | 1.
extern "C" int rdetr_linear_k256_bf16(const uint16_t *x, long long ldx, const uint16_t *w, const uint16_t *bias, long long M,
                                      int N, int relu, uint16_t *out, long long ldo, void *stream)
{
    if (M < 0 || N <= 0 || ldx < kLinK || ldo < N) return RDETR_ERR_INVALID_ARG;
    if ((N & 31) || (ldx & 7) || (ldo & 7)) return RDETR_ERR_UNSUPPORTED;
    if (M == 0) return RDETR_OK;
    if (!x || !w || !out) return RDETR_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(out)) & 15) return RDETR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // slices of 256 columns (16 tiles), or of 192 when that divides N and 256 does not (N = 384)
    if (N % 256 == 0 || N < 192 || (N % 192 != 0)) {
        const int chunks = (N + 255) / 256;
        return relu ? linear_launch<16, true>(x, ldx, w, bias, M, N, out, ldo, chunks, st)
                    : linear_launch<16, false>(x, ldx, w, bias, M, N, out, ldo, chunks, st);
    }
    const int chunks = N / 192;
    return relu ? linear_launch<12, true>(x, ldx, w, bias, M, N, out, ldo, chunks, st)
                : linear_launch<12, false>(x, ldx, w, bias, M, N, out, ldo, chunks, st);
}

// MSDA value projection straight into the head-major layout: out_hm [B, 8, S, 32] <- x [B*S, 256] w[256, 256]^T + bias, rows of
// padded positions (row_mask u8 [B*S], nullable) written as zeros.  rdetr_value_to_head_major_bf16 applied to
// rdetr_linear_k256_bf16's output gives the same bits.
extern "C" int rdetr_linear_k256_hm_bf16(const uint16_t *x, long long ldx, const uint16_t *w, const uint16_t *bias,
                                         const uint8_t *row_mask, int B, int S, uint16_t *out_hm, void *stream)
{
    if (B < 0 || S < 0 || ldx < kLinK) return RDETR_ERR_INVALID_ARG;
    if (ldx & 7) return RDETR_ERR_UNSUPPORTED;
    if (B == 0 || S == 0) return RDETR_OK;
    if ((long long)B * S >= (1ll << 31)) return RDETR_ERR_UNSUPPORTED;
    if (!x || !w || !out_hm) return RDETR_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(out_hm)) & 15) return RDETR_ERR_UNSUPPORTED;
    return linear_launch<16, false>(x, ldx, w, bias, (long long)B * S, 256, out_hm, 256, 1, static_cast<hipStream_t>(stream), row_mask, S);
}

// packed <- w [256, 256] in the fragment order rdetr_linear_ln_k256_bf16 reads (65,536 bf16 elements, 16-byte aligned)
extern "C" int rdetr_linear_pack_k256_bf16(const uint16_t *w, uint16_t *packed, void *stream)
{
    if (!w || !packed) return RDETR_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(packed)) & 15) return RDETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(linear_pack_kernel, dim3(32), dim3(256), 0, static_cast<hipStream_t>(stream), w, reinterpret_cast<u32x4 *>(packed));
    return launch_status();
}

// out[M, 256] = LayerNorm(residual + (x[M, 256] w^T + bias)) with w packed by rdetr_linear_pack_k256_bf16; gamma / beta [256];
// the projection is rounded to bf16 before the residual is added (as the unfused path stores it); fp32 statistics.
extern "C" int rdetr_linear_ln_k256_bf16(const uint16_t *x, long long ldx, const uint16_t *packed, const uint16_t *bias,
                                         const uint16_t *residual, long long ldr, const uint16_t *gamma, const uint16_t *beta,
                                         float eps, long long M, uint16_t *out, long long ldo, void *stream)
{
    if (M < 0 || ldx < kLinK || ldr < kLinK || ldo < kLinK) return RDETR_ERR_INVALID_ARG;
    if ((ldx & 7) || (ldr & 7) || (ldo & 7)) return RDETR_ERR_UNSUPPORTED;
    if (M == 0) return RDETR_OK;
    if (!x || !packed || !residual || !gamma || !beta || !out) return RDETR_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(residual) |
         reinterpret_cast<uintptr_t>(out)) & 15)
        return RDETR_ERR_UNSUPPORTED;
    constexpr int lds = 16 * 8 * 64 * 16 + 3 * 256 * 4;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(linear_ln_k256_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
    const long long steps = (M + kLinRows - 1) / kLinRows;
    long long gx = (steps + kLnlWaves - 1) / kLnlWaves;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(linear_ln_k256_kernel, dim3((unsigned)gx), dim3(kLnlThreads), (size_t)lds, static_cast<hipStream_t>(stream), x,
                       ldx, packed, bias, residual, ldr, gamma, beta, eps, M, out, ldo);
    return launch_status();
}
