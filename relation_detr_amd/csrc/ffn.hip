// Fused feed-forward block for bf16 activations with embed_dim 256 on gfx950:
//     out = (relu(X W1^T + b1)) W2^T + b2          X [M, 256], W1 [F, 256], W2 [256, F], F = d_ffn (a multiple of 64)
// i.e. linear2(relu(linear1(x))) of the encoder / decoder layers (models/bricks/relation_transformer.py:226-233, 272-275) without
// the [M, F] round trip through HBM (732 MB per encoder layer at B = 4: written by one library GEMM, read by the next).
//
//   workgroup   512 threads = 8 waves, persistent over tiles of 256 rows; a wave owns 32 rows for the whole block
//   hidden dim  walked in chunks of 64 units; the chunk's slices of W1 (64 x 256) and W2 (256 x 64) are brought L2 -> LDS by
//               LDS-DMA in MFMA-fragment order (one instruction = one 1-KiB A fragment), double buffered, one barrier per chunk
//   GEMM 1      H^T[hidden x rows] = W1c X^T: A = W1 fragments (LDS), B = X^T from registers (loaded once per row tile).  The
//               chunk's hidden units are permuted over the tiles so that after two tiles lane (row, g) holds hidden units
//               32 u + 8 g .. + 7 of its row -- which, after bias + ReLU + rounding to bf16, IS the B operand of
//   GEMM 2      out^T[256 x rows] += W2c H^T with k = those 32 hidden units: nothing moves between lanes, H never leaves
//               registers.  The 256 outputs are permuted over the 16 tiles the same way, so a lane ends with 8 consecutive
//               outputs per tile pair = one 16-byte store.
//   math        v_mfma_f32_16x16x32_bf16, fp32 accumulation; H is rounded to bf16 exactly where the unfused path stores it
// Bound: MFMA (4096 per 32 rows and wave); LDS reads at half their peak beside it.
#include <cstdlib>

#include "common.h"

namespace rdetr {

typedef __bf16 ffn_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kFfnK = 256;                 // embed_dim: K of GEMM 1, N of GEMM 2
constexpr int kFfnThreads = 512;
constexpr int kFfnWaves = kFfnThreads / 64;
constexpr int kFfnRows = 32;               // rows per wave
constexpr int kFfnHC = 64;                 // hidden units per chunk
constexpr int kFfnW1Bytes = kFfnHC * kFfnK * 2;          // 32 KiB: [4 tiles][8 k-steps][64 lanes] x 16 B
constexpr int kFfnW2Bytes = kFfnK * kFfnHC * 2;          // 32 KiB: [16 out tiles][2 k-steps][64 lanes] x 16 B
constexpr int kFfnBufBytes = kFfnW1Bytes + kFfnW2Bytes;

// slot of a 16-wide tile sequence that carries index i (i = 32 u + 8 g + 4 e + r  <->  tile 2u + e, row 4g + r)
__device__ __forceinline__ int ffn_index(int tile, int m) { return 32 * (tile >> 1) + 8 * (m >> 2) + 4 * (tile & 1) + (m & 3); }

__global__ __launch_bounds__(kFfnThreads) void ffn_k256_kernel(const uint16_t *__restrict__ x, long long ldx,
                                                               const uint16_t *__restrict__ packed, const uint16_t *__restrict__ b1,
                                                               const uint16_t *__restrict__ b2,
                                                               long long M, int F, uint16_t *__restrict__ out, long long ldo, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ffn_lds[];
    float *b1l = reinterpret_cast<float *>(ffn_lds + 2 * kFfnBufBytes);      // [F]
    float *b2l = b1l + F;                                                     // [256]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    for (int i = tid; i < F; i += kFfnThreads) b1l[i] = bf16_bits_to_f32(b1[i]);
    if (tid < kFfnK) b2l[tid] = bf16_bits_to_f32(b2[tid]);

    // LDS-DMA of chunk c into buffer c & 1: the chunk's 64 fragments of 1 KiB are one contiguous 64-KiB slab of the PACKED
    // weights (ffn_pack_kernel below), 8 instructions per wave, each a fully coalesced 1-KiB read
    const int nchunks = F / kFfnHC;
    auto issue_chunk = [&](int c) {
        const unsigned buf = (unsigned)((c & 1) * kFfnBufBytes);
        const unsigned char *slab = reinterpret_cast<const unsigned char *>(packed) + (size_t)c * kFfnBufBytes;
        const unsigned lane_off = (unsigned)lane * 16u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = wave * 8 + i;                                       // uniform
            const unsigned m0v = buf + (unsigned)f * 1024u;
            const unsigned char *src = slab + f * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(m0v), "v"(lane_off), "s"(src) : "memory", "m0");
        }
    };

    const long long ntiles = (M + kFfnWaves * kFfnRows - 1) / (kFfnWaves * kFfnRows);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row_a = (tile * kFfnWaves + wave) * kFfnRows + col, row_b = row_a + 16;
        u32x4 xr[2][8];                                                       // X^T fragments: B operand of GEMM 1, all of K
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            xr[0][s] = row_a < M ? *reinterpret_cast<const u32x4 *>(x + row_a * ldx + 32 * s + 8 * g) : u32x4{0u, 0u, 0u, 0u};
            xr[1][s] = row_b < M ? *reinterpret_cast<const u32x4 *>(x + row_b * ldx + 32 * s + 8 * g) : u32x4{0u, 0u, 0u, 0u};
        }
        f32x4 acc2[16][2];                                                    // out^T: tile ot, column block cb
        __syncthreads();                                                      // biases visible; previous tile's last chunk consumed
#pragma unroll
        for (int ot = 0; ot < 16; ++ot) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(b2l + 32 * (ot >> 1) + 8 * g + 4 * (ot & 1));
            acc2[ot][0] = b4;
            acc2[ot][1] = b4;
        }
        if (!(dbg & 1)) issue_chunk(0);
        for (int c = 0; c < nchunks; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's fragments of chunk c have landed
            if (!(dbg & 2)) __syncthreads();                                  // ... everyone's; and chunk c - 1 is consumed
            if (c + 1 < nchunks && !(dbg & 1)) issue_chunk(c + 1);
            const u32x4 *w1l = reinterpret_cast<const u32x4 *>(ffn_lds + (c & 1) * kFfnBufBytes);
            const u32x4 *w2l = reinterpret_cast<const u32x4 *>(ffn_lds + (c & 1) * kFfnBufBytes + kFfnW1Bytes);
#pragma unroll
            for (int u = 0; u < 2; ++u) {                                     // tile pair = 32 hidden units
                f32x4 acc1[2][2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(b1l + c * kFfnHC + 32 * u + 8 * g + 4 * e);
                    acc1[e][0] = b4;
                    acc1[e][1] = b4;
                }
                if (!(dbg & 8))
#pragma unroll
                for (int s = 0; s < 8; ++s) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const ffn_bf16x8 a = __builtin_bit_cast(ffn_bf16x8, w1l[((2 * u + e) * 8 + s) * 64 + lane]);
                        acc1[e][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(ffn_bf16x8, xr[0][s]), acc1[e][0], 0, 0, 0);
                        acc1[e][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(ffn_bf16x8, xr[1][s]), acc1[e][1], 0, 0, 0);
                    }
                }
                u32x4 h[2];                                                   // relu, round to bf16: B operand of GEMM 2, k = 8 g + j
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const f32x4 lo = acc1[0][cb], hi = acc1[1][cb];
                    h[cb].x = f32_to_bf16_bits(fmaxf(lo.x, 0.f)) | (f32_to_bf16_bits(fmaxf(lo.y, 0.f)) << 16);
                    h[cb].y = f32_to_bf16_bits(fmaxf(lo.z, 0.f)) | (f32_to_bf16_bits(fmaxf(lo.w, 0.f)) << 16);
                    h[cb].z = f32_to_bf16_bits(fmaxf(hi.x, 0.f)) | (f32_to_bf16_bits(fmaxf(hi.y, 0.f)) << 16);
                    h[cb].w = f32_to_bf16_bits(fmaxf(hi.z, 0.f)) | (f32_to_bf16_bits(fmaxf(hi.w, 0.f)) << 16);
                }
                if (!(dbg & 4))
#pragma unroll
                for (int ot = 0; ot < 16; ++ot) {
                    const ffn_bf16x8 a = __builtin_bit_cast(ffn_bf16x8, w2l[(ot * 2 + u) * 64 + lane]);
                    acc2[ot][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(ffn_bf16x8, h[0]), acc2[ot][0], 0, 0, 0);
                    acc2[ot][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(ffn_bf16x8, h[1]), acc2[ot][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const long long row = cb ? row_b : row_a;
            if (row < M) {
                uint16_t *o = out + row * ldo + 8 * g;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const f32x4 lo = acc2[2 * u][cb], hi = acc2[2 * u + 1][cb];
                    u32x4 pk;
                    pk.x = f32_to_bf16_bits(lo.x) | (f32_to_bf16_bits(lo.y) << 16);
                    pk.y = f32_to_bf16_bits(lo.z) | (f32_to_bf16_bits(lo.w) << 16);
                    pk.z = f32_to_bf16_bits(hi.x) | (f32_to_bf16_bits(hi.y) << 16);
                    pk.w = f32_to_bf16_bits(hi.z) | (f32_to_bf16_bits(hi.w) << 16);
                    *reinterpret_cast<u32x4 *>(o + 32 * u) = pk;
                }
            }
        }
    }
}

// Weights -> fragment order, once per weight update: packed[chunk][fragment f][lane][8 bf16] with, for lane (m = lane & 15,
// kb = lane >> 4):  f < 32: W1[64 chunk + index(f >> 3, m)][32 (f & 7) + 8 kb ..]   (GEMM 1: tile f >> 3, k-step f & 7)
//                   f >= 32: W2[index((f - 32) >> 1, m)][64 chunk + 32 (f & 1) + 8 kb ..]   (GEMM 2: out tile, k-step)
__global__ __launch_bounds__(256) void ffn_pack_kernel(const uint16_t *__restrict__ w1, const uint16_t *__restrict__ w2, int F,
                                                       u32x4 *__restrict__ packed)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;                           // one 16-byte piece
    const int total = (F / kFfnHC) * 64 * 64;
    if (idx >= total) return;
    const int c = idx >> 12, f = (idx >> 6) & 63, l = idx & 63, m = l & 15, kb = l >> 4;
    const uint16_t *src;
    if (f < 32) src = w1 + (size_t)(c * kFfnHC + ffn_index(f >> 3, m)) * kFfnK + 32 * (f & 7) + 8 * kb;
    else src = w2 + (size_t)ffn_index((f - 32) >> 1, m) * F + c * kFfnHC + 32 * (f & 1) + 8 * kb;
    packed[idx] = *reinterpret_cast<const u32x4 *>(src);
}

}  // namespace rdetr

using namespace rdetr;

// packed <- (w1 [F, 256], w2 [256, F]) in the fragment order rdetr_ffn_k256_bf16 streams (2 * 256 * F bf16 elements)
extern "C" int rdetr_ffn_k256_pack_bf16(const uint16_t *w1, const uint16_t *w2, int F, uint16_t *packed, void *stream)
{
    if (F <= 0) return RDETR_ERR_INVALID_ARG;
    if ((F % kFfnHC) || F > 4096) return RDETR_ERR_UNSUPPORTED;
    if (!w1 || !w2 || !packed) return RDETR_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) | reinterpret_cast<uintptr_t>(packed)) & 15) return RDETR_ERR_UNSUPPORTED;
    const int total = (F / kFfnHC) * 64 * 64;
    hipLaunchKernelGGL(ffn_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w1, w2, F,
                       reinterpret_cast<u32x4 *>(packed));
    return launch_status();
}

// out[M, 256] = relu(x[M, 256] w1[F, 256]^T + b1[F]) w2[256, F]^T + b2[256] with (w1, w2) packed by rdetr_ffn_k256_pack_bf16; bf16
// storage, fp32 accumulation, the hidden activations rounded to bf16 (as the unfused path stores them).  F % 64 == 0, <= 4096.
extern "C" int rdetr_ffn_k256_bf16(const uint16_t *x, long long ldx, const uint16_t *packed, const uint16_t *b1, const uint16_t *b2,
                                   long long M, int F, uint16_t *out, long long ldo, void *stream)
{
    if (M < 0 || F <= 0 || ldx < kFfnK || ldo < kFfnK) return RDETR_ERR_INVALID_ARG;
    if ((F % kFfnHC) || F > 4096 || (ldx & 7) || (ldo & 7)) return RDETR_ERR_UNSUPPORTED;
    if (M == 0) return RDETR_OK;
    if (!x || !packed || !b1 || !b2 || !out) return RDETR_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out)) & 15)
        return RDETR_ERR_UNSUPPORTED;
    const int lds = 2 * kFfnBufBytes + (F + kFfnK) * 4;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(ffn_k256_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kFfnBufBytes + (4096 + kFfnK) * 4);
    if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
    const long long ntiles = (M + kFfnWaves * kFfnRows - 1) / (kFfnWaves * kFfnRows);
    const long long gx = ntiles < 256 ? ntiles : 256;
    static const int dbg = []() { const char *e = getenv("RDETR_FFN_DBG"); return e ? atoi(e) : 0; }();   // timing experiments only
    hipLaunchKernelGGL(ffn_k256_kernel, dim3((unsigned)gx), dim3(kFfnThreads), (size_t)lds, static_cast<hipStream_t>(stream), x, ldx,
                       packed, b1, b2, M, F, out, ldo, dbg);
    return launch_status();
}
