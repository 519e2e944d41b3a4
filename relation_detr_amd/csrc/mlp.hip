// The decoder's box head as ONE kernel (bf16, embed_dim 256, gfx950):
//     delta = W3 relu(W2 relu(W1 x + b1) + b2) + b3                        MLP(256, 256, 4, 3)   models/bricks/basic.py:6-24
//     out   = sigmoid(delta + inverse_sigmoid(reference))                                        relation_transformer.py:363-381
// for the two inputs a decoder layer feeds it -- the normalised layer output (the layer's boxes) and the layer output itself (the
// next layer's reference points) -- i.e. six library GEMMs of 1,800 rows and two refine launches of the decoder's dependency chain
// (~8 us each whatever their size) in one launch.
//
//   workgroup  512 threads = 8 waves x 32 rows; rows [0, M) are input A, [M, 2M) input B
//   layers     chained INSIDE the wave: with the output permutation of csrc/ffn.hip a lane (row, g) ends a tile pair u holding
//              outputs 32 u + 8 g .. + 7 of its row, which -- biased, ReLU'd, rounded to bf16 as the unfused path stores them --
//              ARE the B operand of the next layer's k-step u.  Nothing changes lanes, nothing goes through LDS.
//   weights    W1, W2 packed in fragment order (rdetr_linear_pack_k256_bf16), streamed L2 -> LDS by LDS-DMA in two 64-KiB halves
//              (output tiles 0-7 / 8-15): W2's halves overwrite W1's as soon as every wave is done with them, behind the other
//              half's MFMAs.  W3 [4, 256] is turned into one zero-padded tile of fragments by the kernel itself.
//   last layer one 16 x 16 tile; lanes g == 0 hold the row's 4 outputs, add the bias, round to bf16 (as the library GEMM stores
//              them), refine against the fp32 reference box and store fp32.
#include "common.h"

namespace rdetr {

typedef __bf16 mlp_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kMlpThreads = 512, kMlpWaves = 8, kMlpRows = 32;
constexpr int kMlpHalf = 8 * 8 * 64 * 16;                 // 64 KiB: 8 tiles x 8 k-steps of 1-KiB fragments
constexpr int kMlpLdsW3 = 2 * kMlpHalf;                    // 8 KiB: the last layer's single tile
constexpr int kMlpLdsBias = kMlpLdsW3 + 8 * 64 * 16;       // b1 | b2 (fp32, 256 each) | b3 (4)
constexpr int kMlpLdsBytes = kMlpLdsBias + (2 * 256 + 4) * 4;

__global__ __launch_bounds__(kMlpThreads) void box_head_k256_kernel(
    const uint16_t *__restrict__ xa, long long lda, const uint16_t *__restrict__ xb, long long ldb, const uint16_t *__restrict__ pw1,
    const uint16_t *__restrict__ b1, const uint16_t *__restrict__ pw2, const uint16_t *__restrict__ b2, const uint16_t *__restrict__ w3,
    const uint16_t *__restrict__ b3, const float *__restrict__ ref, int ref_is_logit, float eps, long long M,
    float *__restrict__ out_a, float *__restrict__ out_b)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char mlp_lds[];
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(mlp_lds);
    u32x4 *w3l = reinterpret_cast<u32x4 *>(mlp_lds + kMlpLdsW3);
    float *bl = reinterpret_cast<float *>(mlp_lds + kMlpLdsBias);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const long long total = xb ? 2 * M : M;

    // half h (0 | 1) of a packed [256, 256] weight -> LDS half h: 64 fragments, 8 per wave
    auto issue_half = [&](const uint16_t *packed, int h) {
        const unsigned lane_off = (unsigned)lane * 16u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = wave * 8 + i;                                       // uniform
            const unsigned m0v = (unsigned)(h * kMlpHalf + f * 1024);
            const unsigned char *src = reinterpret_cast<const unsigned char *>(packed) + (size_t)h * kMlpHalf + f * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(m0v), "v"(lane_off), "s"(src) : "memory", "m0");
        }
    };
    issue_half(pw1, 0);
    issue_half(pw1, 1);
    {   // last layer's tile: fragment (k-step s, lane (m, kb)) = W3[m][32 s + 8 kb ..] for m < 4, zeros otherwise
        const int s = tid >> 6, m = lane & 15, kb = lane >> 4;
        w3l[tid] = m < 4 ? *reinterpret_cast<const u32x4 *>(w3 + m * 256 + 32 * s + 8 * kb) : u32x4{0u, 0u, 0u, 0u};
    }
    if (tid < 256) {
        bl[tid] = bf16_bits_to_f32(b1[tid]);
        bl[256 + tid] = bf16_bits_to_f32(b2[tid]);
    }
    if (tid < 4) bl[512 + tid] = bf16_bits_to_f32(b3[tid]);

    const long long row0 = ((long long)blockIdx.x * kMlpWaves + wave) * kMlpRows + col;
    u32x4 x[2][8];                                                            // the layer's input: B operand, k-step s
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const long long r = row0 + 16 * cb;
        const uint16_t *p = r < M ? xa + r * lda : xb + (r - M) * ldb;
#pragma unroll
        for (int s = 0; s < 8; ++s) x[cb][s] = r < total ? *reinterpret_cast<const u32x4 *>(p + 32 * s + 8 * g) : u32x4{0u, 0u, 0u, 0u};
    }

    auto mm = [&](const u32x4 &a, const u32x4 &bq, const f32x4 &c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mlp_bf16x8, a), __builtin_bit_cast(mlp_bf16x8, bq), c, 0, 0, 0);
    };
    // one hidden layer, tile pairs u0 .. u0 + 3 (one weight half): y[cb][u] = relu(W x + b) as the next layer's operand
    auto half_layer = [&](int h, const float *bias, const u32x4 (&xin)[2][8], u32x4 (&y)[2][8]) {
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const int u = 4 * h + uu;
            f32x4 acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const u32x4 a = wl[((2 * u + e) * 8 + s) * 64 + lane];
                    acc[e][0] = mm(a, xin[0][s], acc[e][0]);
                    acc[e][1] = mm(a, xin[1][s], acc[e][1]);
                }
            const f32x4 blo = *reinterpret_cast<const f32x4 *>(bias + 32 * u + 8 * g), bhi = *reinterpret_cast<const f32x4 *>(bias + 32 * u + 8 * g + 4);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const f32x4 lo = acc[0][cb] + blo, hi = acc[1][cb] + bhi;
                y[cb][u] = u32x4{relu_bf16x2(pack_bf16x2(lo.x, lo.y)), relu_bf16x2(pack_bf16x2(lo.z, lo.w)),
                                 relu_bf16x2(pack_bf16x2(hi.x, hi.y)), relu_bf16x2(pack_bf16x2(hi.z, hi.w))};
            }
        }
    };

    u32x4 y1[2][8], y2[2][8];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                          // W1, W3 tile, biases in LDS
    half_layer(0, bl, x, y1);
    __syncthreads();                                                          // every wave is done with W1's first half
    issue_half(pw2, 0);                                                       // ... W2's first half lands behind the second half's MFMAs
    half_layer(1, bl, x, y1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                          // W2 half 0 landed; W1 half 1 consumed
    issue_half(pw2, 1);
    half_layer(0, bl + 256, y1, y2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    half_layer(1, bl + 256, y1, y2);

    // last layer: one tile, k = the 256 hidden units; lane (row, g = 0) holds outputs 0 .. 3
    f32x4 d[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const u32x4 a = w3l[s * 64 + lane];
        d[0] = mm(a, y2[0][s], d[0]);
        d[1] = mm(a, y2[1][s], d[1]);
    }
    if (g == 0) {
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(bl + 512);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const long long r = row0 + 16 * cb;
            if (r >= total) continue;
            const long long q = r < M ? r : r - M;
            const f32x4 rf = *reinterpret_cast<const f32x4 *>(ref + q * 4);
            const float dl[4] = {d[cb].x + bb.x, d[cb].y + bb.y, d[cb].z + bb.z, d[cb].w + bb.w};
            const float rv[4] = {rf.x, rf.y, rf.z, rf.w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float lg = rv[k];                                             // the reference already in logit space ...
                if (!ref_is_logit) {                                          // ... or a box: inverse_sigmoid (util/misc.py:31-35)
                    float xx = fminf(fmaxf(rv[k], 0.f), 1.f);
                    if (rv[k] != rv[k]) xx = rv[k];
                    const float x1 = fmaxf(xx, eps), x2 = fmaxf(1.f - xx, eps);
                    lg = logf(x1 / x2);
                }
                const float z = bf16_bits_to_f32(f32_to_bf16_bits(dl[k])) + lg;   // delta as the bf16 the GEMM would store
                o[k] = 1.f / (1.f + expf(-z));
            }
            *reinterpret_cast<f32x4 *>((r < M ? out_a : out_b) + q * 4) = f32x4{o[0], o[1], o[2], o[3]};
        }
    }
}

}  // namespace rdetr

using namespace rdetr;

// out_a [M, 4] (and out_b [M, 4] when xb is given) = sigmoid(MLP3(x) + inverse_sigmoid(reference [M, 4])): the decoder's box head
// and box refinement for one or two [M, 256] bf16 inputs (row strides lda / ldb in elements).  pw1 / pw2: the two [256, 256]
// hidden weights packed by rdetr_linear_pack_k256_bf16; w3 [4, 256], b1 / b2 [256], b3 [4] bf16; reference / outputs fp32.
// reference_is_logit != 0: `reference` is added as it is (the two-stage proposals, relation_transformer.py:89-90, come as logits).
extern "C" int rdetr_box_head_k256_bf16(const uint16_t *xa, long long lda, const uint16_t *xb, long long ldb, const uint16_t *pw1,
                                        const uint16_t *b1, const uint16_t *pw2, const uint16_t *b2, const uint16_t *w3, const uint16_t *b3,
                                        const float *reference, int reference_is_logit, float eps, long long M, float *out_a, float *out_b,
                                        void *stream)
{
    if (M < 0 || lda < 256 || (xb && ldb < 256)) return RDETR_ERR_INVALID_ARG;
    if ((lda & 7) || (xb && (ldb & 7))) return RDETR_ERR_UNSUPPORTED;
    if (M == 0) return RDETR_OK;
    if (!xa || !pw1 || !b1 || !pw2 || !b2 || !w3 || !b3 || !reference || !out_a || (xb && !out_b)) return RDETR_ERR_INVALID_ARG;
    auto al = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (!al(xa) || (xb && !al(xb)) || !al(pw1) || !al(pw2) || !al(w3) || !al(reference) || !al(out_a) || (xb && !al(out_b)))
        return RDETR_ERR_UNSUPPORTED;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(box_head_k256_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMlpLdsBytes);
    if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
    const long long total = xb ? 2 * M : M, nblk = (total + kMlpWaves * kMlpRows - 1) / (kMlpWaves * kMlpRows);
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(box_head_k256_kernel, dim3((unsigned)nblk), dim3(kMlpThreads), kMlpLdsBytes, static_cast<hipStream_t>(stream), xa,
                       lda, xb, ldb, pw1, b1, pw2, b2, w3, b3, reference, reference_is_logit, eps, M, out_a, out_b);
    return launch_status();
}
