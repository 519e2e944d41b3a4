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

template <bool LN>
__global__ __launch_bounds__(kFfnThreads) void ffn_k256_kernel(const uint16_t *__restrict__ x, long long ldx,
                                                               const uint16_t *__restrict__ packed, const uint16_t *__restrict__ b1,
                                                               const uint16_t *__restrict__ b2,
                                                               long long M, int F, uint16_t *__restrict__ out, long long ldo, int dbg_arg,
                                                               const uint16_t *__restrict__ gamma, const uint16_t *__restrict__ beta,
                                                               float eps, const uint16_t *__restrict__ pos, long long ldp,
                                                               uint16_t *__restrict__ out2, long long ldo2)
{
#ifdef RDETR_DEV
    const int dbg = dbg_arg;                 // development builds: component-timing mask (1 no weight stream, 2 no barrier, 4 / 8 no GEMM 2 / 1)
#else
    constexpr int dbg = 0;                   // product build: no branches inside the MFMA loop (they end the scheduling regions)
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char ffn_lds[];
    float *b1l = reinterpret_cast<float *>(ffn_lds + 2 * kFfnBufBytes);      // [F]
    float *b2l = b1l + F;                                                     // [256]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    for (int i = tid; i < F; i += kFfnThreads) b1l[i] = bf16_bits_to_f32(b1[i]);
    if (tid < kFfnK) b2l[tid] = bf16_bits_to_f32(b2[tid]);
    float *gml = b2l + kFfnK, *btl = gml + kFfnK;                             // LayerNorm weight / bias (gamma != nullptr)
    if (LN && tid < kFfnK) {
        gml[tid] = bf16_bits_to_f32(gamma[tid]);
        btl[tid] = bf16_bits_to_f32(beta[tid]);
    }

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
            const u32x4 *w1l = reinterpret_cast<const u32x4 *>(ffn_lds + (c & 1) * kFfnBufBytes);
            const u32x4 *w2l = reinterpret_cast<const u32x4 *>(ffn_lds + (c & 1) * kFfnBufBytes + kFfnW1Bytes);
            // The chunk is ONE stream of 64 A fragments (32 of W1, 32 of W2), each feeding two MFMAs (the wave's two 16-row blocks):
            //   t =  0..15  GEMM 1 of tile pair 0                      (k-step t >> 1, tile t & 1)          -> acc1a
            //   t = 16..47  per k-step s: GEMM 1 of pair 1 (2 fragments -> acc1b), GEMM 2 of pair 0 (out tiles 2s, 2s + 1; B = h0)
            //   t = 48..63  GEMM 2 of pair 1 (out tile t - 48; B = h1)
            // read through a ring of three registers, two fragments AHEAD of their use: left to itself the compiler issued every
            // ds_read_b128 right before its MFMAs and waited lgkmcnt(0) -- the LDS latency once per pair of MFMAs, the matrix pipe
            // 56 % busy.  The scheduling barriers pin the order; the counted waits follow from it.  The hidden bias is added when a
            // pair is packed (accumulators start from the inline constant 0: no initialising moves), h0 / h1 are packed two steps
            // after their last MFMA was issued, behind MFMAs that do not need them.
            auto frag = [&](int t) -> u32x4 {
                if (t < 16) return w1l[((t & 1) * 8 + (t >> 1)) * 64 + lane];
                if (t < 48) {
                    const int s = (t - 16) >> 2, r = (t - 16) & 3;
                    return r < 2 ? w1l[((2 + r) * 8 + s) * 64 + lane] : w2l[((2 * s + (r - 2)) * 2) * 64 + lane];
                }
                return w2l[((t - 48) * 2 + 1) * 64 + lane];
            };
            f32x4 acc1a[2][2], acc1b[2][2];
            u32x4 h0[2], h1[2];
            auto mm = [&](const u32x4 &a, const u32x4 &bq, const f32x4 &cacc) {
                return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ffn_bf16x8, a), __builtin_bit_cast(ffn_bf16x8, bq), cacc, 0, 0, 0);
            };
            auto activate = [&](int u, const f32x4 (&acc1)[2][2], u32x4 (&h)[2]) {   // + bias, round to bf16, relu: B operand of GEMM 2
                const f32x4 blo = *reinterpret_cast<const f32x4 *>(b1l + c * kFfnHC + 32 * u + 8 * g);
                const f32x4 bhi = *reinterpret_cast<const f32x4 *>(b1l + c * kFfnHC + 32 * u + 8 * g + 4);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const f32x4 lo = acc1[0][cb] + blo, hi = acc1[1][cb] + bhi;
                    h[cb].x = relu_bf16x2(pack_bf16x2(lo.x, lo.y));          // round, then relu on the packed pair: 2 instructions
                    h[cb].y = relu_bf16x2(pack_bf16x2(lo.z, lo.w));
                    h[cb].z = relu_bf16x2(pack_bf16x2(hi.x, hi.y));
                    h[cb].w = relu_bf16x2(pack_bf16x2(hi.z, hi.w));
                }
            };
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            auto apply = [&](int t, const u32x4 &a) {
                if (t < 16) {
                    const int s = t >> 1, e = t & 1;
                    acc1a[e][0] = mm(a, xr[0][s], s ? acc1a[e][0] : zero4);
                    acc1a[e][1] = mm(a, xr[1][s], s ? acc1a[e][1] : zero4);
                } else if (t < 48) {
                    const int s = (t - 16) >> 2, r = (t - 16) & 3;
                    if (r < 2) {
                        acc1b[r][0] = mm(a, xr[0][s], s ? acc1b[r][0] : zero4);
                        acc1b[r][1] = mm(a, xr[1][s], s ? acc1b[r][1] : zero4);
                    } else {
                        const int ot = 2 * s + (r - 2);
                        acc2[ot][0] = mm(a, h0[0], acc2[ot][0]);
                        acc2[ot][1] = mm(a, h0[1], acc2[ot][1]);
                    }
                } else {
                    const int ot = t - 48;
                    acc2[ot][0] = mm(a, h1[0], acc2[ot][0]);
                    acc2[ot][1] = mm(a, h1[1], acc2[ot][1]);
                }
                if (t == 17) activate(0, acc1a, h0);                          // first needed at t = 18
                if (t == 47) activate(1, acc1b, h1);                          // acc1b complete since t = 45; first needed at t = 48
            };
            if (!(dbg & 12)) {
                u32x4 ring[3];
                ring[0] = frag(0);
                ring[1] = frag(1);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    ring[(t + 2) % 3] = frag(t + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    apply(t, ring[t % 3]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (c + 1 < nchunks && !(dbg & 1)) issue_chunk(c + 1);        // behind the first MFMAs: the pipe starts at once
#pragma unroll
                for (int t = 4; t < 64; ++t) {
                    if (t + 2 < 64) ring[(t + 2) % 3] = frag(t + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    apply(t, ring[t % 3]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the epilogue's row pointers (out, pos, out2) are derived from values the compiler cannot see before this point: hoisted
        // above the chunk loop they would occupy 12 registers there and spill the loop
        unsigned col_e = (unsigned)col;
        asm volatile("" : "+v"(col_e));
        const long long row_e = (tile * kFfnWaves + wave) * kFfnRows + col_e;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const long long row = row_e + 16 * cb;
            if constexpr (LN) {
                // out = LayerNorm(x + ffn(x)) (relation_transformer.py:272-276): the residual is the X^T fragment of k-step u
                // (x[row][32 u + 8 g ..] -- the very columns this lane holds of tile pair u); the row is spread over the 4 lanes
                // l, l ^ 16, l ^ 32, l ^ 48.  ffn(x) is rounded to bf16 first, as the unfused path stores it; fp32 two-pass statistics
                float sum = 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const u32x4 r = xr[cb][u];
                    f32x4 &lo = acc2[2 * u][cb], &hi = acc2[2 * u + 1][cb];
                    lo.x = bf16_bits_to_f32((uint16_t)f32_to_bf16_bits(lo.x)) + __builtin_bit_cast(float, r.x << 16);
                    lo.y = bf16_bits_to_f32((uint16_t)f32_to_bf16_bits(lo.y)) + __builtin_bit_cast(float, r.x & 0xffff0000u);
                    lo.z = bf16_bits_to_f32((uint16_t)f32_to_bf16_bits(lo.z)) + __builtin_bit_cast(float, r.y << 16);
                    lo.w = bf16_bits_to_f32((uint16_t)f32_to_bf16_bits(lo.w)) + __builtin_bit_cast(float, r.y & 0xffff0000u);
                    hi.x = bf16_bits_to_f32((uint16_t)f32_to_bf16_bits(hi.x)) + __builtin_bit_cast(float, r.z << 16);
                    hi.y = bf16_bits_to_f32((uint16_t)f32_to_bf16_bits(hi.y)) + __builtin_bit_cast(float, r.z & 0xffff0000u);
                    hi.z = bf16_bits_to_f32((uint16_t)f32_to_bf16_bits(hi.z)) + __builtin_bit_cast(float, r.w << 16);
                    hi.w = bf16_bits_to_f32((uint16_t)f32_to_bf16_bits(hi.w)) + __builtin_bit_cast(float, r.w & 0xffff0000u);
                    sum += ((lo.x + lo.y) + (lo.z + lo.w)) + ((hi.x + hi.y) + (hi.z + hi.w));
                }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float mean = sum * (1.0f / kFfnK);
                float sq = 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    f32x4 &lo = acc2[2 * u][cb], &hi = acc2[2 * u + 1][cb];
                    lo.x -= mean; lo.y -= mean; lo.z -= mean; lo.w -= mean;
                    hi.x -= mean; hi.y -= mean; hi.z -= mean; hi.w -= mean;
                    sq += ((lo.x * lo.x + lo.y * lo.y) + (lo.z * lo.z + lo.w * lo.w)) + ((hi.x * hi.x + hi.y * hi.y) + (hi.z * hi.z + hi.w * hi.w));
                }
                sq += __shfl_xor(sq, 16, 64);
                sq += __shfl_xor(sq, 32, 64);
                const float rstd = 1.0f / sqrtf(sq * (1.0f / kFfnK) + eps);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    f32x4 &lo = acc2[2 * u][cb], &hi = acc2[2 * u + 1][cb];
                    const f32x4 g0 = *reinterpret_cast<const f32x4 *>(gml + 32 * u + 8 * g), g1 = *reinterpret_cast<const f32x4 *>(gml + 32 * u + 8 * g + 4);
                    const f32x4 c0 = *reinterpret_cast<const f32x4 *>(btl + 32 * u + 8 * g), c1 = *reinterpret_cast<const f32x4 *>(btl + 32 * u + 8 * g + 4);
                    lo.x = lo.x * rstd * g0.x + c0.x; lo.y = lo.y * rstd * g0.y + c0.y; lo.z = lo.z * rstd * g0.z + c0.z; lo.w = lo.w * rstd * g0.w + c0.w;
                    hi.x = hi.x * rstd * g1.x + c1.x; hi.y = hi.y * rstd * g1.y + c1.y; hi.z = hi.z * rstd * g1.z + c1.z; hi.w = hi.w * rstd * g1.w + c1.w;
                }
            }
            if (row < M) {
                uint16_t *o = out + row * ldo + 8 * g;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const f32x4 lo = acc2[2 * u][cb], hi = acc2[2 * u + 1][cb];
                    u32x4 pk;
                    pk.x = pack_bf16x2(lo.x, lo.y);
                    pk.y = pack_bf16x2(lo.z, lo.w);
                    pk.z = pack_bf16x2(hi.x, hi.y);
                    pk.w = pack_bf16x2(hi.z, hi.w);
                    *reinterpret_cast<u32x4 *>(o + 32 * u) = pk;
                    if (LN && out2) {            // out2 = out + pos from the STORED values: the next layer's query + query_pos
                        const u32x4 pv = *reinterpret_cast<const u32x4 *>(pos + row * ldp + 8 * g + 32 * u);
                        u32x4 q;
                        q.x = f32_to_bf16_bits(__builtin_bit_cast(float, pk.x << 16) + __builtin_bit_cast(float, pv.x << 16)) |
                              (f32_to_bf16_bits(__builtin_bit_cast(float, pk.x & 0xffff0000u) + __builtin_bit_cast(float, pv.x & 0xffff0000u)) << 16);
                        q.y = f32_to_bf16_bits(__builtin_bit_cast(float, pk.y << 16) + __builtin_bit_cast(float, pv.y << 16)) |
                              (f32_to_bf16_bits(__builtin_bit_cast(float, pk.y & 0xffff0000u) + __builtin_bit_cast(float, pv.y & 0xffff0000u)) << 16);
                        q.z = f32_to_bf16_bits(__builtin_bit_cast(float, pk.z << 16) + __builtin_bit_cast(float, pv.z << 16)) |
                              (f32_to_bf16_bits(__builtin_bit_cast(float, pk.z & 0xffff0000u) + __builtin_bit_cast(float, pv.z & 0xffff0000u)) << 16);
                        q.w = f32_to_bf16_bits(__builtin_bit_cast(float, pk.w << 16) + __builtin_bit_cast(float, pv.w << 16)) |
                              (f32_to_bf16_bits(__builtin_bit_cast(float, pk.w & 0xffff0000u) + __builtin_bit_cast(float, pv.w & 0xffff0000u)) << 16);
                        *reinterpret_cast<u32x4 *>(out2 + row * ldo2 + 8 * g + 32 * u) = q;
                    }
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

#ifdef RDETR_DEV
// development builds only (make dev): component-timing mask of ffn_k256_kernel (WRONG results)
static int g_ffn_dbg = 0;
extern "C" void rdetr_dev_set_ffn_dbg(int v) { g_ffn_dbg = v; }
#endif

// out[M, 256] = relu(x[M, 256] w1[F, 256]^T + b1[F]) w2[256, F]^T + b2[256] with (w1, w2) packed by rdetr_ffn_k256_pack_bf16; bf16
// storage, fp32 accumulation, the hidden activations rounded to bf16 (as the unfused path stores them).  F % 64 == 0, <= 4096.
static int ffn_launch(const uint16_t *x, long long ldx, const uint16_t *packed, const uint16_t *b1, const uint16_t *b2, long long M,
                      int F, uint16_t *out, long long ldo, const uint16_t *gamma, const uint16_t *beta, float eps, const uint16_t *pos,
                      long long ldp, uint16_t *out2, long long ldo2, void *stream)
{
    if (M < 0 || F <= 0 || ldx < kFfnK || ldo < kFfnK) return RDETR_ERR_INVALID_ARG;
    if ((F % kFfnHC) || F > 4096 || (ldx & 7) || (ldo & 7)) return RDETR_ERR_UNSUPPORTED;
    if (M == 0) return RDETR_OK;
    if (!x || !packed || !b1 || !b2 || !out) return RDETR_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out)) & 15)
        return RDETR_ERR_UNSUPPORTED;
    const int lds = 2 * kFfnBufBytes + (F + 3 * kFfnK) * 4;
    static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void *>(ffn_k256_kernel<false>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kFfnBufBytes + (4096 + 3 * kFfnK) * 4);
    static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void *>(ffn_k256_kernel<true>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kFfnBufBytes + (4096 + 3 * kFfnK) * 4);
    if (attr0 != hipSuccess || attr1 != hipSuccess) return RDETR_ERR_LAUNCH;
    const long long ntiles = (M + kFfnWaves * kFfnRows - 1) / (kFfnWaves * kFfnRows);
    const long long gx = ntiles < 256 ? ntiles : 256;
#ifdef RDETR_DEV
    const int dbg = g_ffn_dbg;
#else
    const int dbg = 0;
#endif
    if (gamma)
        hipLaunchKernelGGL(ffn_k256_kernel<true>, dim3((unsigned)gx), dim3(kFfnThreads), (size_t)lds, static_cast<hipStream_t>(stream), x,
                           ldx, packed, b1, b2, M, F, out, ldo, dbg, gamma, beta, eps, pos, ldp, out2, ldo2);
    else
        hipLaunchKernelGGL(ffn_k256_kernel<false>, dim3((unsigned)gx), dim3(kFfnThreads), (size_t)lds, static_cast<hipStream_t>(stream), x,
                           ldx, packed, b1, b2, M, F, out, ldo, dbg, gamma, beta, eps, pos, ldp, out2, ldo2);
    return launch_status();
}

extern "C" int rdetr_ffn_k256_bf16(const uint16_t *x, long long ldx, const uint16_t *packed, const uint16_t *b1, const uint16_t *b2,
                                   long long M, int F, uint16_t *out, long long ldo, void *stream)
{
    return ffn_launch(x, ldx, packed, b1, b2, M, F, out, ldo, nullptr, nullptr, 0.f, nullptr, 0, nullptr, 0, stream);
}

// out = LayerNorm(x + ffn(x)) (gamma, beta [256], eps) from the same kernel -- the end of an encoder / decoder layer
// (relation_transformer.py:272-276); with pos / out2 (both or neither): out2 = out + pos, the next layer's query + query_pos.
extern "C" int rdetr_ffn_ln_k256_bf16(const uint16_t *x, long long ldx, const uint16_t *packed, const uint16_t *b1,
                                      const uint16_t *b2, const uint16_t *gamma, const uint16_t *beta, float eps,
                                      const uint16_t *pos, long long ldp, long long M, int F, uint16_t *out, long long ldo,
                                      uint16_t *out2, long long ldo2, void *stream)
{
    if (!gamma || !beta || (pos != nullptr) != (out2 != nullptr)) return RDETR_ERR_INVALID_ARG;
    if (pos && (ldp < kFfnK || ldo2 < kFfnK)) return RDETR_ERR_INVALID_ARG;
    if (pos && ((ldp & 7) || (ldo2 & 7) || ((reinterpret_cast<uintptr_t>(pos) | reinterpret_cast<uintptr_t>(out2)) & 15)))
        return RDETR_ERR_UNSUPPORTED;
    return ffn_launch(x, ldx, packed, b1, b2, M, F, out, ldo, gamma, beta, eps, pos, ldp, out2, ldo2, stream);
}
