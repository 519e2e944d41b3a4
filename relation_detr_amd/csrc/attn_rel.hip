// Decoder self-attention that GENERATES its position-relation bias: softmax(Q K^T * scale + relu(W . feat(box_q, box_k) + b)) V
// in one kernel (gfx950) -- SURVEY.md section 8 f1 as written.
//
// Replaces, for bf16 inference, PositionRelationEmbedding.forward (models/bricks/relation_transformer.py:520-532) followed by the
// nn.MultiheadAttention call that takes its result as a float attn_mask (:369-374, :452-461): the [B, 8, N, N] fp32 bias (26 MB per
// image and layer at N = 900) is never written to or read from HBM, and the 64 -> 8 projection of the sine features moves from 512
// VALU FMAs per (query, key) pair onto the matrix cores.
//
//   grid      = (ceil(N / 16) query tiles, B images); workgroup = 16 waves that own 16 queries for ALL 8 heads, so that the sine
//               features of a (query, key) pair are computed once and shared by the heads
//   keys      = chunks of 64.  Per chunk the 16 x 64 pairs are 64 "pair sets" (one query x 16 consecutive keys), each projected by
//               MFMAs  bias^T[key][head] = A[key][:] . B[:][head]  (v_mfma_f32_16x16x32_bf16, B in hi + lo bf16 parts):
//                 distance coordinates (x, y; 32 features): lane (key = lane & 15, g = lane >> 4) computes the 8 features the A
//                           fragment wants from it -- coordinate g >> 1, frequencies 4 (g & 1) .. + 3: one log2, 4 x (v_sin, v_cos);
//                           B = W[:, 0:32]
//                 size-ratio coordinates (w, h; 32 features): log(w_q / w_k) = a_q - a_k, and sin / cos (a_q - a_k) are bilinear in
//                           per-box terms, so their projection is the dot product of the KEY's 32 values (sin a_k, cos a_k) (A,
//                           computed once per chunk) with per-(query, head) coefficients (B, built once per workgroup): no
//                           per-pair arithmetic at all
//               accumulator initialised with the projection bias, ReLU, written to LDS as [head][query][key] fp32, everything
//               pre-multiplied by log2(e) for the soft-max
//               waves 8-15 compute 8 pair sets each per chunk; waves 0-7 run the attention of head = wave for the chunk whose bias
//               tile the feature waves finished one barrier earlier (two bias tiles in LDS: one barrier per chunk).  Disjoint
//               roles in two separate loops keep either path inside the 128 VGPRs a 16-wave workgroup has
//   attention = as csrc/attn.hip (transposed products, online soft-max in the log2 domain, P rounded to bf16, V^T through
//               ds_read_b64_tr_b16), one wave per head over all keys; K fragments come straight from global memory into registers,
//               V through a wave-private LDS image; the soft-max denominator is the MFMA product ones . P^T of the rounded P
//   numerics  = the features / table entries are rounded to bf16 (2^-9) for the projection, the weights are not (hi + lo split);
//               the angles use the hardware log2 / sin / cos (|angle| < 256 revolutions for any box of size >= 1e-5).  This is the
//               bf16 inference path: its results are held to the same 2^-7 bound against the fp32 oracle as the materialised-bias
//               attention kernel.  fp32 runs and training keep rdetr_relation_bias_f32 (reference op order, IEEE division,
//               Cody-Waite sin / cos).
//   cost      = VALU-bound (one wave-instruction per clock and CU): ~50 issue slots per pair set, 36 of them the quarter-rate
//               log2 / sin / cos, + ~160 per head and chunk for the soft-max
#include <type_traits>

#include "common.h"

namespace rdetr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kArD = 32, kArH = 8, kArTileQ = 16, kArChunk = 64, kArWaves = 16, kArF = 16;
constexpr int kArQStride = 68;                                  // floats per (head, query) row of a bias tile: 64 keys + 4 (banks)
constexpr int kArHeadStride = kArTileQ * kArQStride + 4;        // floats per head (+4: the 8 heads' b128 writes hit 8 bank groups)
constexpr int kArBiasBuf = kArH * kArHeadStride;                // floats per bias tile
constexpr int kArVS = 96;                                       // V image row stride in bytes (conflict-free transposed reads)
constexpr int kArVImg = kArChunk * kArVS;                       // one attention wave's V chunk image
constexpr int kArQC = 4;                                        // floats per query: x, y, 1/(w+eps), 1/(h+eps)
constexpr int kArLdsBias = 0, kArLdsV = 2 * kArBiasBuf * 4, kArLdsQC = kArLdsV + kArH * kArVImg,
              kArLdsW = kArLdsQC + kArTileQ * kArQC * 4, kArLdsU = kArLdsW + 2 * 64 * 16,
              kArLdsBytes = kArLdsU + kArTileQ * 2 * 64 * 16;   // 150.5 KiB

struct RelFreq {
    float cf[8];        // ln 2 * scale / (temperature^(2k/F) * 2 pi): log2 of the encoding -> revolutions
};

__global__ __launch_bounds__(kArWaves *kWave) void relation_attention_boxes_kernel(
    const uint16_t *__restrict__ q, const uint16_t *__restrict__ k, const uint16_t *__restrict__ v, int ldq, int ldk, int ldv,
    const float *__restrict__ src, const float *__restrict__ tgt,
    const float *__restrict__ Wp, const float *__restrict__ bp, const unsigned char *__restrict__ mask, int N, int M,
    float scale_log2e, float eps, RelFreq fr, uint16_t *__restrict__ out, int ldo, int dbg_arg)
{
#ifdef RDETR_DEV
    const int dbg = dbg_arg;                 // development builds: component-timing mask (tools/attn_rel_components.py)
#else
    constexpr int dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char ar_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, q0 = blockIdx.x * kArTileQ;
    float *bias_lds = reinterpret_cast<float *>(ar_lds + kArLdsBias);
    float *qc = reinterpret_cast<float *>(ar_lds + kArLdsQC);
    u32x4 *wfr = reinterpret_cast<u32x4 *>(ar_lds + kArLdsW);
    u32x4 *ufr = reinterpret_cast<u32x4 *>(ar_lds + kArLdsU);
    constexpr float kLog2e = 1.4426950408889634f;

    // ---- prologue.  Everything that feeds the soft-max is produced in the log2 domain: W, b scaled by log2(e) (ReLU commutes) ----
    if (tid < kArTileQ) {                                   // per-query constants of the two distance coordinates
        const int qi = q0 + tid < N ? q0 + tid : N - 1;
        const f32x4 s = *reinterpret_cast<const f32x4 *>(src + ((size_t)b * N + qi) * 4);
        float *r = qc + tid * kArQC;
        r[0] = s.x; r[1] = s.y; r[2] = 1.0f / (s.z + eps); r[3] = 1.0f / (s.w + eps);
    }
    if (tid < 128) {                                        // W[:, 0:32] (distance features) as MFMA B fragments, hi / lo bf16 parts
        const int part = tid >> 6, l = tid & 63, head = l & 15, gg = l >> 4;
        unsigned int o[4] = {0u, 0u, 0u, 0u};
        if (head < kArH) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float w0 = Wp[head * 64 + 8 * gg + 2 * j] * kLog2e, w1 = Wp[head * 64 + 8 * gg + 2 * j + 1] * kLog2e;
                if (part) {
                    w0 -= bf16_bits_to_f32(f32_to_bf16_bits(w0));
                    w1 -= bf16_bits_to_f32(f32_to_bf16_bits(w1));
                }
                o[j] = pack_bf16x2(w0, w1);
            }
        }
        wfr[tid] = u32x4{o[0], o[1], o[2], o[3]};
    }
    {
        // Size-ratio coordinates: sin / cos (a_q - a_k) are bilinear in the per-box tables, so their projection is a 32-term dot
        // product of the KEY's table entries (sin a_k, cos a_k) with per-(query, head) coefficients
        //   U[sin entry] = -W_sin cos a_q + W_cos sin a_q,   U[cos entry] = W_sin sin a_q + W_cos cos a_q
        // -> one more MFMA K-step whose B operand belongs to the query; thread = (query, fragment lane), hi and lo parts
        const int qq = tid >> 6, l = tid & 63, head = l & 15, gg = l >> 4, cc = gg >> 1, k0 = 4 * (gg & 1);
        const int qi = q0 + qq < N ? q0 + qq : N - 1;
        unsigned int hi[4] = {0u, 0u, 0u, 0u}, lo[4] = {0u, 0u, 0u, 0u};
        if (head < kArH) {
            const float l2 = __builtin_amdgcn_logf(src[((size_t)b * N + qi) * 4 + 2 + cc] + eps);      // angle a_q = log(size + eps) * scale / dim_t
            const float *wrow = Wp + head * 64 + 32 + 16 * cc + 2 * k0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = l2 * fr.cf[k0 + j];
                const float ss = __builtin_amdgcn_sinf(x), sc = __builtin_amdgcn_cosf(x);
                const float ws = wrow[2 * j] * kLog2e, wc = wrow[2 * j + 1] * kLog2e;
                const float us = __builtin_fmaf(wc, ss, -(ws * sc)), uc = __builtin_fmaf(ws, ss, wc * sc);
                hi[j] = pack_bf16x2(us, uc);
                lo[j] = pack_bf16x2(us - bf16_bits_to_f32(hi[j] & 0xffffu), uc - bf16_bits_to_f32(hi[j] >> 16));
            }
        }
        ufr[(qq * 2 + 0) * 64 + l] = u32x4{hi[0], hi[1], hi[2], hi[3]};
        ufr[(qq * 2 + 1) * 64 + l] = u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
    __syncthreads();

    // ---- roles ----
    const bool att = wave < kArH;
    const int kbw = (wave & 7) >> 1;                  // feature waves: the 16-key block of a chunk this wave computes features for
    const int qbase = (wave & 1) * 8;                 // ... for these 8 queries of the tile
    const int c01 = g >> 1, fsel = g & 1;
    const float cf0 = fr.cf[4 * fsel + 0], cf1 = fr.cf[4 * fsel + 1], cf2 = fr.cf[4 * fsel + 2], cf3 = fr.cf[4 * fsel + 3];
    const float bph = (bp && ql < kArH) ? bp[ql] * kLog2e : 0.f;
    const int nch = (M + kArChunk - 1) / kArChunk;

    // this lane's key of a chunk (fetched one chunk ahead): its box -> the distance coordinate, and (sin a_k, cos a_k) of this lane's
    // size coordinate and frequencies = the A fragment of the table K-step
    auto load_key = [&](int chunk) {
        const int key = chunk * kArChunk + 16 * kbw + ql < M ? chunk * kArChunk + 16 * kbw + ql : M - 1;
        return *reinterpret_cast<const f32x4 *>(tgt + ((size_t)b * M + key) * 4);
    };
    auto pair_sets = [&](auto nps_c, int buf, const f32x4 &kbox) {
        constexpr int NPS = decltype(nps_c)::value;
        const u32x4 w0h = wfr[lane], w0l = wfr[64 + lane];
        float *dst = bias_lds + buf * kArBiasBuf + ql * kArHeadStride + 16 * kbw + 4 * g;
        const float tcoord = c01 ? kbox.y : kbox.x;
        const float l2k = __builtin_amdgcn_logf((c01 ? kbox.w : kbox.z) + eps);
        const float y0 = l2k * cf0, y1 = l2k * cf1, y2 = l2k * cf2, y3 = l2k * cf3;
        const u32x4 a1 = {pack_bf16x2(__builtin_amdgcn_sinf(y0), __builtin_amdgcn_cosf(y0)),
                          pack_bf16x2(__builtin_amdgcn_sinf(y1), __builtin_amdgcn_cosf(y1)),
                          pack_bf16x2(__builtin_amdgcn_sinf(y2), __builtin_amdgcn_cosf(y2)),
                          pack_bf16x2(__builtin_amdgcn_sinf(y3), __builtin_amdgcn_cosf(y3))};
#pragma unroll 1
        for (int i0 = 0; i0 < NPS; i0 += 4)
#pragma unroll
        for (int i1 = 0; i1 < 4; ++i1) {                                            // four independent chains in flight
            const int qq = qbase + i0 + i1;
            const float *qr = qc + qq * kArQC;
            const float sc = qr[c01], inv = qr[2 + c01];
            const u32x4 uh = ufr[(qq * 2 + 0) * 64 + lane], ul = ufr[(qq * 2 + 1) * 64 + lane];
            // distance coordinate: log(|c_q - c_k| / (size_q + eps) + 1), as log2; the frequency factors carry ln 2 * scale / dim_t
            const float e2 = __builtin_amdgcn_logf(__builtin_fmaf(__builtin_fabsf(sc - tcoord), inv, 1.0f));
            const float x0 = e2 * cf0, x1 = e2 * cf1, x2 = e2 * cf2, x3 = e2 * cf3;
            u32x4 a0;
            a0.x = pack_bf16x2(__builtin_amdgcn_sinf(x0), __builtin_amdgcn_cosf(x0));
            a0.y = pack_bf16x2(__builtin_amdgcn_sinf(x1), __builtin_amdgcn_cosf(x1));
            a0.z = pack_bf16x2(__builtin_amdgcn_sinf(x2), __builtin_amdgcn_cosf(x2));
            a0.w = pack_bf16x2(__builtin_amdgcn_sinf(x3), __builtin_amdgcn_cosf(x3));
            f32x4 acc = {bph, bph, bph, bph};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, uh), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, ul), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, w0h), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, w0l), acc, 0, 0, 0);
            // lane (head = lane & 15, g) holds keys 4 g .. 4 g + 3 of the pair set: relu(bias) * log2(e)
            acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
            if (ql < kArH) *reinterpret_cast<f32x4 *>(dst + qq * kArQStride) = acc;
        }
    };
    auto features = [&](int buf, const f32x4 &kbox) { pair_sets(std::integral_constant<int, 8>{}, buf, kbox); };

    // ---- pipeline: the feature waves produce the bias tile of chunk c + 1 while the heads consume chunk c.  Two loops with the
    //      same number of workgroup barriers (the branch is wave-uniform): a wave's registers hold only its own role's state ----
    if (!att) {
        f32x4 kc = load_key(0);
        features(0, kc);
        if (nch > 1) kc = load_key(1);
        __syncthreads();
        for (int c = 0; c < nch; ++c) {
            if (c + 1 < nch && !(dbg & 1)) features((c + 1) & 1, kc);
            if (c + 2 < nch) kc = load_key(c + 2);              // in flight across the barrier
            __syncthreads();
        }
        return;
    }

    // ---- attention state (waves 0-7: head = wave) ----
    const int h = wave;
    const int qi = q0 + ql;
    const bool qok = qi < N;
    const int qcl = qok ? qi : N - 1;
    unsigned char *lds_v = ar_lds + kArLdsV + wave * kArVImg;
    const uint16_t *kbase = k + (size_t)b * M * ldk + h * kArD;
    const uint16_t *vbase = v + (size_t)b * M * ldv + h * kArD;
    const unsigned char *mask_row = mask ? mask + (size_t)qcl * M : nullptr;
    u32x4 qfrag = {0, 0, 0, 0}, kf[4], vr[4];
    float m_run = -__builtin_inff(), l_run = 0.f;
    f32x4 oacc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};

    // K / V rows through buffer loads: one per-lane offset for all chunks, the chunk's position in the scalar offset, and rows past
    // the last key come back as zeros from the descriptor's range check (no address arithmetic or bounds tests in the loop)
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(kbase), 0,
                                                                         (unsigned)((M - 1) * ldk + kArD) * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(vbase), 0,
                                                                         (unsigned)((M - 1) * ldv + kArD) * 2u, 0x00020000);
    const unsigned kvo = (unsigned)(ql * ldk + g * 8) * 2u, vvo = (unsigned)((lane >> 2) * ldv + (lane & 3) * 8) * 2u;
    auto load_k = [&](int key0) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
            kf[kb] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvo, (unsigned)((key0 + 16 * kb) * ldk) * 2u, 0);
    };
    auto load_v = [&](int key0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            vr[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvo, (unsigned)((key0 + 16 * i) * ldv) * 2u, 0);
    };
    auto store_v = [&] {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i;
            *reinterpret_cast<u32x4 *>(lds_v + (idx >> 2) * kArVS + (idx & 3) * 16) = vr[i];
        }
    };
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    auto attention = [&](int c, int buf) {
        const int key0 = c * kArChunk;
        const bool more = c + 1 < nch;
        if (more) load_v(key0 + kArChunk);
        const float *brow = bias_lds + buf * kArBiasBuf + h * kArHeadStride + ql * kArQStride + 4 * g;
        f32x4 s[4];
        float m_loc = -__builtin_inff();
        const bool edge = mask_row != nullptr || key0 + kArChunk > M;              // wave-uniform: masks and the key tail are rare
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[kb]), __builtin_bit_cast(bf16x8, qfrag), z, 0, 0, 0);
            f32x4 t = *reinterpret_cast<const f32x4 *>(brow + 16 * kb);             // already in the log2 domain
            if (edge) {
                const int kk = key0 + 16 * kb + 4 * g;
                if (mask_row) {
                    if (kk + 0 < M && mask_row[kk + 0]) t.x = -__builtin_inff();
                    if (kk + 1 < M && mask_row[kk + 1]) t.y = -__builtin_inff();
                    if (kk + 2 < M && mask_row[kk + 2]) t.z = -__builtin_inff();
                    if (kk + 3 < M && mask_row[kk + 3]) t.w = -__builtin_inff();
                }
                if (kk + 0 >= M) t.x = -__builtin_inff();       // keys past the end never take part
                if (kk + 1 >= M) t.y = -__builtin_inff();
                if (kk + 2 >= M) t.z = -__builtin_inff();
                if (kk + 3 >= M) t.w = -__builtin_inff();
            }
            z.x = __builtin_fmaf(z.x, scale_log2e, t.x);
            z.y = __builtin_fmaf(z.y, scale_log2e, t.y);
            z.z = __builtin_fmaf(z.z, scale_log2e, t.z);
            z.w = __builtin_fmaf(z.w, scale_log2e, t.w);
            s[kb] = z;
            m_loc = fmaxf(m_loc, fmaxf(fmaxf(z.x, z.y), fmaxf(z.z, z.w)));
        }
        if (more) load_k(key0 + kArChunk);                      // the S^T products have consumed this chunk's fragments
        m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 16, 64));
        m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 32, 64));
        const float m_new = fmaxf(m_run, m_loc);
        const float m_safe = (m_new == -__builtin_inff()) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
        m_run = m_new;
        u32x4 pf[2];
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
            pf[pair] = u32x4{pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]), pack_bf16x2(p[6], p[7])};
        }
        // the row sums of P (as rounded for the PV product) on the matrix core: ones . P^T -> every accumulator row holds the sum
        {
            const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
            f32x4 ls = {0.f, 0.f, 0.f, 0.f};
            ls = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, pf[0]), ls, 0, 0, 0);
            ls = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, pf[1]), ls, 0, 0, 0);
            l_run = __builtin_fmaf(l_run, alpha, ls.x);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            oacc[cb].x *= alpha; oacc[cb].y *= alpha; oacc[cb].z *= alpha; oacc[cb].w *= alpha;
        }
        const int tq = (lane >> 2) & 3, tp = lane & 3;
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const unsigned char *a0 = lds_v + (32 * pair + 4 * g + tq) * kArVS + cb * 32 + tp * 8;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a0 + 16 * kArVS));
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                const u32x4 vf = {l2.x, l2.y, h2.x, h2.y};
                oacc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf[pair]),
                                                                   oacc[cb], 0, 0, 0);
            }
        }
        wave_sync();                                            // this wave is done with its V image
        if (more) store_v();
        wave_sync();
    };

    qfrag = *reinterpret_cast<const u32x4 *>(q + ((size_t)b * N + qcl) * ldq + h * kArD + g * 8);
    load_k(0);
    load_v(0);
    store_v();
    wave_sync();
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        if (!(dbg & 2)) attention(c, c & 1);
        __syncthreads();
    }

    if (qok) {
        const float inv = 1.0f / l_run;                            // 0 / 0 = NaN for a fully masked row, as torch.softmax
        uint16_t *o = out + ((size_t)b * N + qi) * ldo + h * kArD + 4 * g;
        *reinterpret_cast<u32x2 *>(o) = u32x2{pack_bf16x2(oacc[0].x * inv, oacc[0].y * inv), pack_bf16x2(oacc[0].z * inv, oacc[0].w * inv)};
        *reinterpret_cast<u32x2 *>(o + 16) = u32x2{pack_bf16x2(oacc[1].x * inv, oacc[1].y * inv), pack_bf16x2(oacc[1].z * inv, oacc[1].w * inv)};
    }
}

}  // namespace rdetr

#ifdef RDETR_DEV
// development builds only (make dev): 1 = feature waves idle, 2 = attention waves idle (WRONG results, component timing)
static int g_ar_dbg = 0;
extern "C" void rdetr_dev_set_attn_rel_dbg(int v) { g_ar_dbg = v; }
#define RDETR_AR_DBG g_ar_dbg
#else
#define RDETR_AR_DBG 0
#endif

extern "C" int rdetr_relation_attention_boxes_bf16(const uint16_t *q, const uint16_t *k, const uint16_t *v, int ldq, int ldk, int ldv,
                                                   const float *src_boxes, const float *tgt_boxes, const float *proj_weight,
                                                   const float *proj_bias, const uint8_t *bool_mask, int B, int H,
                                                   int D, int N, int M, int F, float rel_scale, float temperature, float eps,
                                                   float attn_scale, uint16_t *out, int ldo, void *stream)
{
    using namespace rdetr;
    if (B < 0 || H <= 0 || N < 0 || M < 0 || F <= 0) return RDETR_ERR_INVALID_ARG;
    if (D != kArD || H != kArH || F != kArF) return RDETR_ERR_UNSUPPORTED;
    if (B == 0 || N == 0) return RDETR_OK;
    if (M == 0) return RDETR_ERR_INVALID_ARG;
    if (!q || !k || !v || !out || !src_boxes || !tgt_boxes || !proj_weight) return RDETR_ERR_INVALID_ARG;
    auto al = [](const void *p, unsigned a) { return reinterpret_cast<uintptr_t>(p) % a == 0; };
    if (!al(q, 16) || !al(k, 16) || !al(v, 16) || !al(out, 8) || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4 || !al(src_boxes, 16) ||
        !al(tgt_boxes, 16))
        return RDETR_ERR_UNSUPPORTED;
    if (B > 65535 || (long long)M * (ldk > ldv ? ldk : ldv) * 2 >= (1ll << 31)) return RDETR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    RelFreq fr;
    for (int i = 0; i < F / 2; ++i) {
        const double dim_t = (double)powf(temperature, (float)i * 2.0f / (float)F);       // get_dim_t, position_encoding.py:101-105
        fr.cf[i] = (float)(0.6931471805599453 * (double)rel_scale / (dim_t * 6.283185307179586));
    }
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(relation_attention_boxes_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kArLdsBytes);
    if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
    hipLaunchKernelGGL(relation_attention_boxes_kernel, dim3((unsigned)((N + kArTileQ - 1) / kArTileQ), (unsigned)B),
                       dim3(kArWaves * kWave), kArLdsBytes, st, q, k, v, ldq, ldk, ldv, src_boxes, tgt_boxes,
                       proj_weight, proj_bias, bool_mask, N, M, attn_scale * 1.4426950408889634f, eps, fr, out, ldo, RDETR_AR_DBG);
    return launch_status();
}
