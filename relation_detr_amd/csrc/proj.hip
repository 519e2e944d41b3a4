// The encoder layer's three input projections of MultiScaleDeformableAttention as ONE kernel (bf16, embed_dim 256, gfx950):
//     value    = value_proj(x)                 [rows, 256] -> HEAD-MAJOR [B, 8, S, 32], rows of padded positions zero
//                                              (models/bricks/ms_deform_attn.py:315-321: value_proj, then masked_fill)
//     offsets | logits = [sampling_offsets ; attention_weights](x + pos)      [rows, 256] -> [rows, 384 | 480] raw projection outputs
//                                              (ms_deform_attn.py:322-327; the fused gather reads the two column slices in place;
//                                              384 columns with 4 feature levels, 480 with 5)
// i.e. the hand-written value projection (csrc/linear.hip, 21.6 us per image group) and the N = 384 library GEMM (25.2 us) of
// every encoder layer: two ~20-us launches of 5.8 + 8.8 GFLOP whose duration is mostly their own start-up chain (launch, weight
// fill, first rows) -- one launch, one chain.
//
//   workgroup  w waves, one or two 16-row blocks per wave (CB): 8 x 32 rows = 256-row tiles, or thin waves (9-12 x 16 rows)
//              sized so that ONE round of tiles covers all 256 CUs (44,646 rows = 254 tiles of 176 rows)
//   phases     five or six independent 128-column output blocks: value heads 0-3, heads 4-7 (B operand = x), then the three or
//              four blocks of the query projection (B operand = x + pos; of the fourth only its first 96 columns exist).  A block's 64 weight fragments (64 KiB, packed by
//              rdetr_linear_pack_k256_bf16) are fetched into registers behind the previous block's MFMAs and written to the one
//              LDS buffer when every wave has left it (csrc/qpos.hip's scheme); the fragments are read as one software-pipelined
//              stream; each block's outputs are biased, rounded and stored as soon as it is done (8 consecutive columns per lane
//              = one 16-byte store; for the value: column block u = head u of the head-major plane).
#include <type_traits>

#include "common.h"

namespace rdetr {

namespace {

typedef __bf16 pj_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kPjHalf = 8 * 8 * 64 * 16;                  // 64 KiB: 8 tiles x 8 k-steps of 1-KiB fragments = 128 output columns
constexpr int kPjLdsBias = kPjHalf;                       // value bias [256] | query bias [QCOLS <= 512] as fp32
constexpr int kPjLdsBytes = kPjLdsBias + (256 + 512) * 4;

// QCOLS = output columns of the query projection: 384 (4 levels) or 480 (5 levels)
template <int CB, int THREADS, int QCOLS>
__global__ __launch_bounds__(THREADS) void encoder_proj_k256_kernel(
    const uint16_t *__restrict__ x, long long ldx, const uint16_t *__restrict__ xq, long long ldq, const uint16_t *__restrict__ pwv,
    const uint16_t *__restrict__ bv, const uint16_t *__restrict__ pwq, const uint16_t *__restrict__ bq,
    const unsigned char *__restrict__ row_mask, int S, long long M, uint16_t *__restrict__ out_hm, uint16_t *__restrict__ out_q)
{
    constexpr int kMaxFrags = CB == 2 ? 8 : 8;            // fragments per wave and block: ceil(64 / waves), waves >= 8
    extern __shared__ __attribute__((aligned(16))) unsigned char pj_lds[];
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(pj_lds);
    float *bl = reinterpret_cast<float *>(pj_lds + kPjLdsBias);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = (int)(blockDim.x >> 6);
    const int col = lane & 15, g = lane >> 4;

    // this wave's share of a block's 64 fragments: f = wave, wave + nwaves, ...  (registers now, LDS at the next commit)
    u32x4 stg[kMaxFrags];
    auto fetch = [&](const uint16_t *packed, int h) {
#pragma unroll
        for (int i = 0; i < kMaxFrags; ++i) {
            const int f = wave + i * nwaves;                                  // uniform
            if (f < 64) stg[i] = reinterpret_cast<const u32x4 *>(packed)[((size_t)h * kPjHalf + (size_t)f * 1024) / 16 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);                                    // the loads are issued HERE, ahead of the block's MFMAs
    };
    auto commit = [&]() {
        __syncthreads();                                                      // every wave has left the buffer
        u32x4 *dst = reinterpret_cast<u32x4 *>(pj_lds);
#pragma unroll
        for (int i = 0; i < kMaxFrags; ++i) {
            const int f = wave + i * nwaves;
            if (f < 64) dst[f * 64 + lane] = stg[i];
        }
        __syncthreads();                                                      // ... and sees the new block
    };

    fetch(pwv, 0);
    for (int i = tid; i < 256; i += (int)blockDim.x) bl[i] = bv ? bf16_bits_to_f32(bv[i]) : 0.f;
    for (int i = tid; i < 512; i += (int)blockDim.x) bl[256 + i] = (bq && i < QCOLS) ? bf16_bits_to_f32(bq[i]) : 0.f;

    const long long row0 = ((long long)blockIdx.x * nwaves + wave) * (16 * CB) + col;
    u32x4 xr[CB][8], qr[CB][8];                                               // B operands: k-step s = columns 32 s + 8 g .. + 7 of the lane's rows
    bool rok[CB], zero[CB];
    uint16_t *ohm[CB], *oq[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const long long row = row0 + 16 * cb;
        rok[cb] = row < M;
        const long long r = rok[cb] ? row : 0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            xr[cb][s] = rok[cb] ? *reinterpret_cast<const u32x4 *>(x + r * ldx + 32 * s + 8 * g) : u32x4{0u, 0u, 0u, 0u};
            qr[cb][s] = rok[cb] ? *reinterpret_cast<const u32x4 *>(xq + r * ldq + 32 * s + 8 * g) : u32x4{0u, 0u, 0u, 0u};
        }
        zero[cb] = rok[cb] && row_mask && row_mask[r] != 0;
        const long long img = r / S, pos = r - img * S;
        ohm[cb] = out_hm + ((img * 8) * (long long)S + pos) * 32 + 8 * g;     // + head * S * 32
        oq[cb] = out_q + r * QCOLS + 8 * g;
    }

    auto mm = [&](const u32x4 &a, const u32x4 &bq_, const f32x4 &c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pj_bf16x8, a), __builtin_bit_cast(pj_bf16x8, bq_), c, 0, 0, 0);
    };
    // one 128-column block: acc[uu][e][cb] over 4 tile pairs x 8 k-steps x 2 tiles = 64 fragments, read through a ring of four
    // registers three fragments ahead of their MFMAs
    auto block = [&](const u32x4 (&xin)[CB][8], f32x4 (&acc)[4][2][CB], auto nfrags) {
        constexpr int kFrags = decltype(nfrags)::value, kAhead = 3;      // 64, or 48 for a block whose last 32 columns do not exist
        u32x4 ring[4];
        auto frag = [&](int f) { return wl[(((2 * (f >> 4) + (f & 1)) * 8 + ((f >> 1) & 7)) * 64) + lane]; };
#pragma unroll
        for (int f = 0; f < kAhead; ++f) ring[f & 3] = frag(f);
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < kFrags; ++f) {
            if (f + kAhead < kFrags) ring[(f + kAhead) & 3] = frag(f + kAhead);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                acc[f >> 4][f & 1][cb] = mm(ring[f & 3], xin[cb][(f >> 1) & 7], ((f >> 1) & 7) ? acc[f >> 4][f & 1][cb] : zero4);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto packed_out = [&](const f32x4 (&acc)[4][2][CB], const float *bias, int uu, int cb) {
        const f32x4 lo = acc[uu][0][cb] + *reinterpret_cast<const f32x4 *>(bias + 32 * uu + 8 * g);
        const f32x4 hi = acc[uu][1][cb] + *reinterpret_cast<const f32x4 *>(bias + 32 * uu + 8 * g + 4);
        return u32x4{pack_bf16x2(lo.x, lo.y), pack_bf16x2(lo.z, lo.w), pack_bf16x2(hi.x, hi.y), pack_bf16x2(hi.z, hi.w)};
    };

    constexpr int kBlocks = 2 + (QCOLS + 127) / 128;                          // 5 or 6
    constexpr int kLastUnits = (QCOLS - 128 * (kBlocks - 3)) / 32;            // 32-column units of the last block: 4 or 3
    f32x4 acc[4][2][CB];
    commit();                                                                 // value heads 0-3 (and the biases)
#pragma unroll
    for (int p = 0; p < kBlocks; ++p) {
        if (p == 0) fetch(pwv, 1);
        else if (p < kBlocks - 1) fetch(pwq, p - 1);
        if (p < 2) {
            block(xr, acc, std::integral_constant<int, 64>{});
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                if (rok[cb]) {
#pragma unroll
                    for (int uu = 0; uu < 4; ++uu) {                          // column block 32 (4 p + uu) = head 4 p + uu
                        const u32x4 v = zero[cb] ? u32x4{0u, 0u, 0u, 0u} : packed_out(acc, bl + 128 * p, uu, cb);
                        *reinterpret_cast<u32x4 *>(ohm[cb] + (long long)(4 * p + uu) * S * 32) = v;
                    }
                }
        } else {
            constexpr int kUnitsOf[2] = {4, kLastUnits};
            const int units = kUnitsOf[p == kBlocks - 1];
            if (p == kBlocks - 1) block(qr, acc, std::integral_constant<int, 16 * kLastUnits>{});
            else block(qr, acc, std::integral_constant<int, 64>{});
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                if (rok[cb]) {
#pragma unroll
                    for (int uu = 0; uu < 4; ++uu)
                        if (uu < units)
                            *reinterpret_cast<u32x4 *>(oq[cb] + 128 * (p - 2) + 32 * uu) = packed_out(acc, bl + 256 + 128 * (p - 2), uu, cb);
                }
        }
        if (p < kBlocks - 1) commit();
    }
}

}  // namespace

}  // namespace rdetr

using namespace rdetr;

// out_hm [B, 8, S, 32] = head-major(value_proj(x)) with the rows of padded positions zero; out_q [B*S, q_cols] = xq Wq^T + bq,
// q_cols = 384 (4 feature levels) or 480 (5).
//   x, xq  [B*S, 256] bf16 (row strides ldx / ldq elements, multiples of 8; 16-byte aligned): the layer input and input + pos
//   pwv    value_proj.weight [256, 256] packed by rdetr_linear_pack_k256_bf16; bv [256] bf16 or NULL
//   pwq    [sampling_offsets.weight ; attention_weights.weight ; zero rows] = [512, 256] as TWO packed [256, 256] blocks, one
//          after the other (only the first q_cols output columns are stored); bq [q_cols] bf16 or NULL
//   row_mask  key_padding_mask u8 [B*S] or NULL
extern "C" int rdetr_encoder_proj_k256_bf16(const uint16_t *x, long long ldx, const uint16_t *xq, long long ldq, const uint16_t *pwv,
                                            const uint16_t *bv, const uint16_t *pwq, const uint16_t *bq, const uint8_t *row_mask, int B, int S,
                                            int q_cols, uint16_t *out_hm, uint16_t *out_q, void *stream)
{
    if (B < 0 || S < 0 || ldx < 256 || ldq < 256) return RDETR_ERR_INVALID_ARG;
    if (q_cols != 384 && q_cols != 480) return RDETR_ERR_UNSUPPORTED;
    if ((ldx & 7) || (ldq & 7)) return RDETR_ERR_UNSUPPORTED;
    if (B == 0 || S == 0) return RDETR_OK;
    if (!x || !xq || !pwv || !pwq || !out_hm || !out_q) return RDETR_ERR_INVALID_ARG;
    auto al = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (!al(x) || !al(xq) || !al(pwv) || !al(pwq) || !al(out_hm) || !al(out_q)) return RDETR_ERR_UNSUPPORTED;
    const long long M = (long long)B * S;
    static const hipError_t attr = [] {
        hipError_t e = hipSuccess;
        const void *fns[4] = {reinterpret_cast<const void *>(encoder_proj_k256_kernel<2, 512, 384>),
                              reinterpret_cast<const void *>(encoder_proj_k256_kernel<1, 768, 384>),
                              reinterpret_cast<const void *>(encoder_proj_k256_kernel<2, 512, 480>),
                              reinterpret_cast<const void *>(encoder_proj_k256_kernel<1, 768, 480>)};
        for (const void *fn : fns) {
            const hipError_t r = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kPjLdsBytes);
            if (r != hipSuccess) e = r;
        }
        return e;
    }();
    if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
    // tile shape as for the thin feed-forward experiment: rounds of tiles over 256 CUs x (waves per SIMD x blocks per wave)
    const auto rounds = [&](long long rows) { return ((M + rows - 1) / rows + 255) / 256; };
    long long best = 4 * rounds(256);
    int thin = 0;
#ifndef RDETR_PROJ_THIN
#define RDETR_PROJ_THIN 1
#endif
    if (RDETR_PROJ_THIN)
        for (int w = 12; w >= 9; --w) {
            const long long cost = 3 * rounds(16 * w);
            if (cost < best || (cost == best && thin)) {
                best = cost;
                thin = w;
            }
        }
    const long long tile_rows = thin ? 16 * thin : 256;
    const long long ntiles = (M + tile_rows - 1) / tile_rows;
    if (ntiles > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto launch = [&](auto kernel, unsigned threads) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)ntiles), dim3(threads), kPjLdsBytes, st, x, ldx, xq, ldq, pwv, bv, pwq, bq, row_mask, S, M,
                           out_hm, out_q);
    };
    if (q_cols == 384) {
        if (thin) launch(encoder_proj_k256_kernel<1, 768, 384>, 64u * (unsigned)thin);
        else launch(encoder_proj_k256_kernel<2, 512, 384>, 512u);
    } else {
        if (thin) launch(encoder_proj_k256_kernel<1, 768, 480>, 64u * (unsigned)thin);
        else launch(encoder_proj_k256_kernel<2, 512, 480>, 512u);
    }
    return launch_status();
}
