// Decoder self-attention with the position-relation bias, fused: softmax(Q K^T * scale + bias) V in one kernel (gfx950).
//
// Replaces, for bf16 activations, the chain behind the reference's nn.MultiheadAttention call with a float attn_mask
// (models/bricks/relation_transformer.py:452-461, bias from :369-374): QK^T GEMM -> fp32 copy -> bias add + softmax ->
// bf16 copy -> PV GEMM (five launches and four [B*H,N,N] round trips) by one flash-style pass that reads the bias once.
// SURVEY.md section 8f rank 1 (the bias itself still comes from rdetr_relation_bias_f32: its sine features are shared
// by the 8 heads, which a per-head attention kernel would recompute 8 times).
//
//   grid      = (ceil(N / 16) query tiles, B * H);  workgroup = 4 waves that share 16 queries and SPLIT THE KEYS: wave w takes the
//               64-key chunks w, w + 4, ... with its own online soft-max state; the four partial results are merged through LDS
//   keys      = chunks of 64; a wave stages its chunk of K and V in its own LDS image (register double buffer for the next one,
//               bias of its next 4 chunks in flight): no workgroup barrier inside the loop
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
#include <cstdlib>

#include "common.h"

namespace rdetr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kAtD = 32;                 // head dim
constexpr int kAtTileQ = 16, kAtChunk = 64, kAtWaves = 4;
constexpr int kAtKS = 80, kAtVS = 96;    // LDS row strides in bytes (K: ds_read_b128 rows; V: conflict-free transposed reads)
constexpr int kAtWaveLds = kAtChunk * (kAtKS + kAtVS);      // one wave's private K / V chunk image (11 KiB)
constexpr int kAtAhead = 4;              // chunks of bias a wave keeps in flight

// One workgroup = 16 queries of one (image, head); its 4 waves SPLIT THE KEYS: wave w takes chunks w, w + 4, w + 8, ... of 64 keys
// with its own online-softmax state, its own LDS image of the chunk (staged by the wave itself: no workgroup barrier in the loop)
// and the bias of its next chunks already in flight; the four partial results are merged through LDS at the end.
// (The first version gave each wave 16 queries and ALL keys: at the decoder's size -- 900 keys = 15 chunks -- every wave ran a serial
// chain of 15 x ~2.6 us whatever the occupancy, and the kernel took 40-50 us with the chip nearly idle.)
template <int S>          // S = waves that share 16 queries and split their keys (1, 2, 4): more for fewer queries in the launch
__global__ __launch_bounds__(kAtWaves *kWave) void relation_attention_kernel(
    const uint16_t *__restrict__ q, const uint16_t *__restrict__ k, const uint16_t *__restrict__ v, int ldq, int ldk, int ldv,
    const float *__restrict__ bias, const unsigned char *__restrict__ mask, int H, int N, int M, float scale_log2e,
    uint16_t *__restrict__ out, int ldo)
{
    __shared__ __attribute__((aligned(16))) unsigned char at_lds[kAtWaves * kAtWaveLds];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int qgrp = wave / S, part = wave % S;            // query group of the workgroup, key-split index inside it
    const int qi = (blockIdx.x * (kAtWaves / S) + qgrp) * kAtTileQ + ql;
    const bool qok = qi < N;
    const int qc = qok ? qi : N - 1;
    constexpr float kLog2e = 1.4426950408889634f;
    unsigned char *lds_k = at_lds + wave * kAtWaveLds, *lds_v = lds_k + kAtChunk * kAtKS;

    // Q fragment: B operand of the S^T product, B[k = d = 8 g + j][col = q]
    const u32x4 qfrag = *reinterpret_cast<const u32x4 *>(q + ((size_t)b * N + qc) * ldq + h * kAtD + g * 8);
    const float *bias_row = bias ? bias + ((size_t)bh * N + qc) * M : nullptr;
    const unsigned char *mask_row = mask ? mask + (size_t)qc * M : nullptr;
    const bool vec_bias = (M % 4 == 0) && (reinterpret_cast<uintptr_t>(bias) % 16 == 0);

    // staging of one K / V chunk by ONE wave: piece idx = lane + 64 i -> (row = idx >> 2, 16-byte piece = idx & 3)
    const uint16_t *kbase = k + (size_t)b * M * ldk + h * kAtD;
    const uint16_t *vbase = v + (size_t)b * M * ldv + h * kAtD;
    auto load_chunk = [&](int key0, u32x4 (&kr)[4], u32x4 (&vr)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i, key = key0 + (idx >> 2), piece = idx & 3;
            if (key < M) {
                kr[i] = *reinterpret_cast<const u32x4 *>(kbase + (size_t)key * ldk + piece * 8);
                vr[i] = *reinterpret_cast<const u32x4 *>(vbase + (size_t)key * ldv + piece * 8);
            } else {
                kr[i] = u32x4{0, 0, 0, 0};
                vr[i] = u32x4{0, 0, 0, 0};
            }
        }
    };
    auto store_chunk = [&](const u32x4 (&kr)[4], const u32x4 (&vr)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i, srow = idx >> 2, spiece = idx & 3;
            *reinterpret_cast<u32x4 *>(lds_k + srow * kAtKS + spiece * 16) = kr[i];
            *reinterpret_cast<u32x4 *>(lds_v + srow * kAtVS + spiece * 16) = vr[i];
        }
    };
    auto wave_sync = [] {                                   // the LDS image is private to the wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // bias / mask of this lane's 16 (query, key) pairs of a chunk: keys key0 + 16 kb + 4 g + r
    auto load_bias = [&](int key0, f32x4 (&bz)[4]) {
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
    };

    float m_run = -__builtin_inff(), l_run = 0.f;
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};            // O^T[d = 16 cb + 4 g + r][q]

    const int nchunks = (M + kAtChunk - 1) / kAtChunk;
    const int mine = part < nchunks ? (nchunks - part + S - 1) / S : 0;                    // this wave's chunks: part + S i
    f32x4 ring[kAtAhead][4];
#pragma unroll
    for (int j = 0; j < kAtAhead; ++j)
        if (j < mine) load_bias((part + S * j) * kAtChunk, ring[j]);
    u32x4 kr[4], vr[4];
    if (mine > 0) {
        load_chunk(part * kAtChunk, kr, vr);
        store_chunk(kr, vr);
    }
    wave_sync();

    auto chunk_body = [&](int i, f32x4 (&slot)[4]) {                 // i-th chunk of this wave
        const int key0 = (part + S * i) * kAtChunk;
        if (i + 1 < mine) load_chunk(key0 + S * kAtChunk, kr, vr);
        f32x4 bz[4] = {slot[0], slot[1], slot[2], slot[3]};
        if (i + kAtAhead < mine) load_bias(key0 + kAtAhead * S * kAtChunk, slot);

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
            o.x = pack_bf16x2(p[0], p[1]);
            o.y = pack_bf16x2(p[2], p[3]);
            o.z = pack_bf16x2(p[4], p[5]);
            o.w = pack_bf16x2(p[6], p[7]);
            pf[pair] = o;
#pragma unroll
            for (int j = 0; j < 8; ++j) l_loc += p[j];
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
        wave_sync();                                        // the wave is done with this chunk's LDS image
        if (i + 1 < mine) store_chunk(kr, vr);
        wave_sync();
    };
    for (int i0 = 0; i0 < mine; i0 += kAtAhead) {           // uniform per wave
#pragma unroll
        for (int j = 0; j < kAtAhead; ++j)
            if (i0 + j < mine) chunk_body(i0 + j, ring[j]);
    }

    // merge the four waves' partial soft-max states: lane (q, g) of every wave holds (m, l partial over its keys, acc[2])
    __syncthreads();                                        // every wave is done with its K / V image: re-use it
    float *mg = reinterpret_cast<float *>(at_lds);          // [wave][10][64 lanes]
    {
        float *w = mg + wave * 640 + lane;
        w[0] = m_run; w[64] = l_run;
        w[128] = acc[0].x; w[192] = acc[0].y; w[256] = acc[0].z; w[320] = acc[0].w;
        w[384] = acc[1].x; w[448] = acc[1].y; w[512] = acc[1].z; w[576] = acc[1].w;
    }
    __syncthreads();
    if (part == 0) {
        float m_all = -__builtin_inff();
#pragma unroll
        for (int w = 0; w < S; ++w) m_all = fmaxf(m_all, mg[(qgrp * S + w) * 640 + lane]);
        const float m_safe = (m_all == -__builtin_inff()) ? 0.f : m_all;
        float l = 0.f;
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < S; ++w) {
            const float *r = mg + (qgrp * S + w) * 640 + lane;
            const float f = __builtin_amdgcn_exp2f(r[0] - m_safe);             // exp2(-inf - m) = 0 for a wave without keys
            l += r[64] * f;
            o0.x += r[128] * f; o0.y += r[192] * f; o0.z += r[256] * f; o0.w += r[320] * f;
            o1.x += r[384] * f; o1.y += r[448] * f; o1.z += r[512] * f; o1.w += r[576] * f;
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (qok) {
            const float inv = 1.0f / l;                     // 0 / 0 = NaN for a fully masked row, as torch.softmax
            uint16_t *o = out + ((size_t)b * N + qi) * ldo + h * kAtD + 4 * g;
            *reinterpret_cast<u32x2 *>(o) = u32x2{pack_bf16x2(o0.x * inv, o0.y * inv), pack_bf16x2(o0.z * inv, o0.w * inv)};
            *reinterpret_cast<u32x2 *>(o + 16) = u32x2{pack_bf16x2(o1.x * inv, o1.y * inv), pack_bf16x2(o1.z * inv, o1.w * inv)};
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
    // waves per 16 queries: the fewer queries the launch has, the more ways their keys are split (a wave's pass over 64 keys is a
    // ~2.6-us latency chain: 900 keys in one wave are 40 us however idle the chip is)
    const long long groups = ((long long)N + kAtTileQ - 1) / kAtTileQ, total = groups * bh;
    const int split = total <= 512 ? 4 : (total <= 1024 ? 2 : 1);       // measured at N = M = 900: B = 2 (912
                                                                // groups): 40.8 / 26.3 / 28.3 us for 1 / 2 / 4; B = 4: 46.4 / 49.1 / 52.4
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float sl = scale * 1.4426950408889634f;
    if (split == 4)
        hipLaunchKernelGGL(relation_attention_kernel<4>, dim3((unsigned)groups, (unsigned)bh), dim3(kAtWaves * kWave), 0, st, q, k, v, ldq,
                           ldk, ldv, bias, bool_mask, H, N, M, sl, out, ldo);
    else if (split == 2)
        hipLaunchKernelGGL(relation_attention_kernel<2>, dim3((unsigned)((groups + 1) / 2), (unsigned)bh), dim3(kAtWaves * kWave), 0, st, q,
                           k, v, ldq, ldk, ldv, bias, bool_mask, H, N, M, sl, out, ldo);
    else
        hipLaunchKernelGGL(relation_attention_kernel<1>, dim3((unsigned)((groups + 3) / 4), (unsigned)bh), dim3(kAtWaves * kWave), 0, st, q,
                           k, v, ldq, ldk, ldv, bias, bool_mask, H, N, M, sl, out, ldo);
    return launch_status();
}
