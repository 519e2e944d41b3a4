// Decoder self-attention with the position-relation bias, fused: softmax(Q K^T * scale + bias) V in one kernel (gfx950).
//
// Replaces, for bf16 activations, the chain behind the reference's nn.MultiheadAttention call with a float attn_mask
// (models/bricks/relation_transformer.py:452-461, bias from :369-374): QK^T GEMM -> fp32 copy -> bias add + softmax ->
// bf16 copy -> PV GEMM (five launches and four [B*H,N,N] round trips) by one flash-style pass that reads the bias once.
// SURVEY.md section 8f rank 1 (the bias itself still comes from rdetr_relation_bias_f32: its sine features are shared
// by the 8 heads, which a per-head attention kernel would recompute 8 times).
//
//   grid      = (ceil(N / 64) query tiles, B * H);  workgroup = 4 waves, wave w owns queries 16w .. 16w+15 of the tile
//   keys      = chunks of 64; K and V chunks staged in LDS once per workgroup (register double buffer for the next one)
//   MFMA      = v_mfma_f32_16x16x32_bf16, head dim 32 = one K step.  The products are taken TRANSPOSED so that nothing
//               ever has to change lanes:
//                 S^T[key][q] = K[key][:] . Q[q][:]      A = K rows (ds_read_b128), B = Q rows (registers, loaded once)
//                               -> lane (q = lane & 15, g = lane >> 4) holds keys 16 kb + 4 g + r of ITS query
//                 O^T[d][q]   = sum_key V[key][d] P[q][key]   B = the lane's own P values (bf16), A = V^T through
//                               ds_read_b64_tr_b16 with the SAME key permutation  k = 8 g + j  <->  key 16 (2 pair + (j >> 2))
//                               + 4 g + (j & 3)  (a sum over keys does not care about their order)
//               so the soft-max statistics of a query live in the 4 lanes {q, q+16, q+32, q+48}: two xor-shuffles per chunk.
//   softmax   = online (running max / sum in fp32, exp2 with the log2e fold), P rounded to bf16 for the PV product, fp32
//               accumulation; a fully masked row yields NaN like torch.softmax.
//   bias      = fp32 [B*H, N, M], read as 16-byte pieces (4 consecutive keys of one query per lane); bool mask [N, M].
#include "common.h"

namespace rdetr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kAtD = 32;                 // head dim
constexpr int kAtTileQ = 64, kAtChunk = 64, kAtWaves = 4;
constexpr int kAtKS = 80, kAtVS = 96;    // LDS row strides in bytes (K: ds_read_b128 rows; V: conflict-free transposed reads)

__global__ __launch_bounds__(kAtWaves *kWave) void relation_attention_kernel(
    const uint16_t *__restrict__ q, const uint16_t *__restrict__ k, const uint16_t *__restrict__ v, int ldq, int ldk, int ldv,
    const float *__restrict__ bias, const unsigned char *__restrict__ mask, int H, int N, int M, float scale_log2e,
    uint16_t *__restrict__ out, int ldo)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds_k[kAtChunk * kAtKS];
    __shared__ __attribute__((aligned(16))) unsigned char lds_v[kAtChunk * kAtVS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int q0 = blockIdx.x * kAtTileQ + wave * 16;
    const int qi = q0 + ql;
    const bool qok = qi < N;
    const int qc = qok ? qi : N - 1;
    constexpr float kLog2e = 1.4426950408889634f;

    // Q fragment: B operand of the S^T product, B[k = d = 8 g + j][col = q]
    const u32x4 qfrag = *reinterpret_cast<const u32x4 *>(q + ((size_t)b * N + qc) * ldq + h * kAtD + g * 8);
    const float *bias_row = bias ? bias + ((size_t)bh * N + qc) * M : nullptr;
    const unsigned char *mask_row = mask ? mask + (size_t)qc * M : nullptr;
    const bool vec_bias = (M % 4 == 0) && (reinterpret_cast<uintptr_t>(bias) % 16 == 0);

    // staging of one K / V chunk: thread -> (row = tid >> 2, 16-byte piece = tid & 3)
    const int srow = tid >> 2, spiece = tid & 3;
    const uint16_t *kbase = k + (size_t)b * M * ldk + h * kAtD + spiece * 8;
    const uint16_t *vbase = v + (size_t)b * M * ldv + h * kAtD + spiece * 8;
    auto load_chunk = [&](int key0, u32x4 &kr, u32x4 &vr) {
        const int key = key0 + srow;
        if (key < M) {
            kr = *reinterpret_cast<const u32x4 *>(kbase + (size_t)key * ldk);
            vr = *reinterpret_cast<const u32x4 *>(vbase + (size_t)key * ldv);
        } else {
            kr = u32x4{0, 0, 0, 0};
            vr = u32x4{0, 0, 0, 0};
        }
    };
    auto store_chunk = [&](const u32x4 &kr, const u32x4 &vr) {
        *reinterpret_cast<u32x4 *>(lds_k + srow * kAtKS + spiece * 16) = kr;
        *reinterpret_cast<u32x4 *>(lds_v + srow * kAtVS + spiece * 16) = vr;
    };

    float m_run = -__builtin_inff(), l_run = 0.f;
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};            // O^T[d = 16 cb + 4 g + r][q]

    u32x4 kr, vr;
    load_chunk(0, kr, vr);
    store_chunk(kr, vr);
    __syncthreads();

    const int nchunks = (M + kAtChunk - 1) / kAtChunk;
    for (int c = 0; c < nchunks; ++c) {
        const int key0 = c * kAtChunk;
        if (c + 1 < nchunks) load_chunk(key0 + kAtChunk, kr, vr);

        // bias / mask of this lane's 16 (query, key) pairs: keys key0 + 16 kb + 4 g + r
        f32x4 bz[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int kk = key0 + 16 * kb + 4 * g;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (bias_row) {
                if (vec_bias) {
                    if (kk < M) t = *reinterpret_cast<const f32x4 *>(bias_row + kk);
                } else {
                    t.x = kk + 0 < M ? bias_row[kk + 0] : 0.f;
                    t.y = kk + 1 < M ? bias_row[kk + 1] : 0.f;
                    t.z = kk + 2 < M ? bias_row[kk + 2] : 0.f;
                    t.w = kk + 3 < M ? bias_row[kk + 3] : 0.f;
                }
            }
            if (mask_row) {
                if (kk + 0 < M && mask_row[kk + 0]) t.x = -__builtin_inff();
                if (kk + 1 < M && mask_row[kk + 1]) t.y = -__builtin_inff();
                if (kk + 2 < M && mask_row[kk + 2]) t.z = -__builtin_inff();
                if (kk + 3 < M && mask_row[kk + 3]) t.w = -__builtin_inff();
            }
            if (kk + 0 >= M) t.x = -__builtin_inff();          // keys past the end never take part
            if (kk + 1 >= M) t.y = -__builtin_inff();
            if (kk + 2 >= M) t.z = -__builtin_inff();
            if (kk + 3 >= M) t.w = -__builtin_inff();
            bz[kb] = t;
        }

        // S^T = K Q^T for the chunk's four 16-key blocks
        f32x4 s[4];
        float m_loc = -__builtin_inff();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const u32x4 kf = *reinterpret_cast<const u32x4 *>(lds_k + (16 * kb + ql) * kAtKS + g * 16);
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qfrag), z, 0, 0, 0);
            // work in the log2 domain: (s * scale + bias) * log2(e)
            z.x = z.x * scale_log2e + bz[kb].x * kLog2e;
            z.y = z.y * scale_log2e + bz[kb].y * kLog2e;
            z.z = z.z * scale_log2e + bz[kb].z * kLog2e;
            z.w = z.w * scale_log2e + bz[kb].w * kLog2e;
            s[kb] = z;
            m_loc = fmaxf(m_loc, fmaxf(fmaxf(z.x, z.y), fmaxf(z.z, z.w)));
        }
        m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 16, 64));
        m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 32, 64));
        const float m_new = fmaxf(m_run, m_loc);
        const float m_safe = (m_new == -__builtin_inff()) ? 0.f : m_new;       // all keys masked so far: keep exp2(-inf) = 0
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
        m_run = m_new;
        float l_loc = 0.f;
        u32x4 pf[2];                                        // B operands of the PV product: P^T[k = 8 g + j][q], bf16
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
            float p[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x4 z = s[2 * pair + half];
                p[4 * half + 0] = __builtin_amdgcn_exp2f(z.x - m_safe);
                p[4 * half + 1] = __builtin_amdgcn_exp2f(z.y - m_safe);
                p[4 * half + 2] = __builtin_amdgcn_exp2f(z.z - m_safe);
                p[4 * half + 3] = __builtin_amdgcn_exp2f(z.w - m_safe);
            }
            u32x4 o;
            o.x = f32_to_bf16_bits(p[0]) | (f32_to_bf16_bits(p[1]) << 16);
            o.y = f32_to_bf16_bits(p[2]) | (f32_to_bf16_bits(p[3]) << 16);
            o.z = f32_to_bf16_bits(p[4]) | (f32_to_bf16_bits(p[5]) << 16);
            o.w = f32_to_bf16_bits(p[6]) | (f32_to_bf16_bits(p[7]) << 16);
            pf[pair] = o;
#pragma unroll
            for (int i = 0; i < 8; ++i) l_loc += p[i];
        }
        l_run = l_run * alpha + l_loc;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            acc[cb].x *= alpha; acc[cb].y *= alpha; acc[cb].z *= alpha; acc[cb].w *= alpha;
        }
        // O^T += V^T P^T : A[d][k] = V[key(g, j)][16 cb + d] through the transposed read
        //   lane 4 q' + p of its 16-lane group supplies row key = 32 pair (+16) + 4 g + q', columns 16 cb + 4 p .. + 3
        const int tq = (lane >> 2) & 3, tp = lane & 3;
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const unsigned char *a0 = lds_v + (32 * pair + 4 * g + tq) * kAtVS + cb * 32 + tp * 8;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a0 + 16 * kAtVS));
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                const u32x4 vf = {l2.x, l2.y, h2.x, h2.y};
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf[pair]),
                                                                  acc[cb], 0, 0, 0);
            }
        }
        __syncthreads();                                    // every wave is done with this chunk's LDS image
        if (c + 1 < nchunks) store_chunk(kr, vr);
        __syncthreads();
    }

    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (qok) {
        const float inv = 1.0f / l_run;                     // 0 / 0 = NaN for a fully masked row, as torch.softmax
        uint16_t *o = out + ((size_t)b * N + qi) * ldo + h * kAtD + 4 * g;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            u32x2 w;
            w.x = f32_to_bf16_bits(acc[cb].x * inv) | (f32_to_bf16_bits(acc[cb].y * inv) << 16);
            w.y = f32_to_bf16_bits(acc[cb].z * inv) | (f32_to_bf16_bits(acc[cb].w * inv) << 16);
            *reinterpret_cast<u32x2 *>(o + 16 * cb) = w;
        }
    }
}

}  // namespace rdetr

extern "C" int rdetr_relation_attention_bf16(const uint16_t *q, const uint16_t *k, const uint16_t *v, int ldq, int ldk,
                                             int ldv, const float *bias, const uint8_t *bool_mask, int B, int H, int D,
                                             int N, int M, float scale, uint16_t *out, int ldo, void *stream)
{
    using namespace rdetr;
    if (B < 0 || H <= 0 || N < 0 || M < 0) return RDETR_ERR_INVALID_ARG;
    if (D != kAtD) return RDETR_ERR_UNSUPPORTED;
    if (B == 0 || N == 0) return RDETR_OK;
    if (M == 0) return RDETR_ERR_INVALID_ARG;
    if (!q || !k || !v || !out) return RDETR_ERR_INVALID_ARG;
    auto al = [](const void *p, unsigned a) { return reinterpret_cast<uintptr_t>(p) % a == 0; };
    if (!al(q, 16) || !al(k, 16) || !al(v, 16) || !al(out, 8) || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4 || (bias && !al(bias, 4)))
        return RDETR_ERR_UNSUPPORTED;
    const long long bh = (long long)B * H;
    if (bh > 65535) return RDETR_ERR_UNSUPPORTED;
    dim3 grid((unsigned)((N + kAtTileQ - 1) / kAtTileQ), (unsigned)bh);
    hipLaunchKernelGGL(relation_attention_kernel, grid, dim3(kAtWaves * kWave), 0, static_cast<hipStream_t>(stream), q, k, v, ldq,
                       ldk, ldv, bias, bool_mask, H, N, M, scale * 1.4426950408889634f, out, ldo);
    return launch_status();
}
