// The decoder layer's query position as ONE kernel (bf16, embed_dim 256, gfx950):
//     query_pos = ref_point_head(sine_embed)                 MLP(512, 256, 256, 2)      relation_transformer.py:294, 343-344
//     query_pos = query_pos * query_scale(query)             MLP(256, 256, 256, 2)      relation_transformer.py:345-347 (layers >= 1)
//     qpp       = query + query_pos                          what the layer's self-attention takes as q = k  (:452-455)
// i.e. four library GEMMs of 1,800 rows + the product / sum launch of the decoder's dependency chain (~6 us each whatever their
// size, and the two MLPs are independent branches a single stream serialises) in one launch.
//
//   workgroup  4 waves x 16 rows (rows are the unit of parallelism: a row needs all 256 hidden units; every wave reads the whole
//              weight half from LDS per phase, so more waves per workgroup only queue on the LDS: 8 waves 28.5 us, 4 waves see
//              tools/time_qpos.py)
//   layers     chained INSIDE the wave with the output permutation of csrc/ffn.hip / csrc/mlp.hip: after tile pair u lane
//              (row, g) holds outputs 32 u + 8 g .. + 7 of its row, which -- biased, ReLU'd, rounded to bf16 as the unfused path
//              stores them -- ARE the B operand of the next layer's k-step u.  Nothing changes lanes.
//   weights    five [256, 256] blocks packed in fragment order (rdetr_linear_pack_k256_bf16; the 512-input layer as its two
//              K halves), streamed in 64-KiB halves: the half for phase i + 1 is loaded into REGISTERS (16 coalesced 1-KiB loads
//              per wave) behind phase i's MFMAs and written to the one LDS buffer once every wave has left it.
//   measured   (tools/time_qpos.py, 1,800 rows, graph replay) 21.7 us against 22.6 us for the five launches it replaces; component
//              builds: without the MFMAs 21.5, without the LDS fragment reads 17.4, without the weight stream 14.1 -- a workgroup
//              pulls all 640 KB of weights through its CU's 64 B/clk vector-memory path (4.7 us) and through LDS twice, phase by
//              phase, so the chain cannot shrink much further; +1 % images/s in the stack (profiles/r03/ab_stack_query_pos.txt).
//   rounding   every intermediate is rounded where the unfused bf16 path stores it (hidden activations, query_pos, the scale, their
//              product), so the result is the unfused sequence's up to the summation order inside a dot product.
#include "common.h"

namespace rdetr {

namespace {

typedef __bf16 qp_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kQpWaves = 4, kQpThreads = kQpWaves * 64, kQpRows = 16;        // 64 rows per workgroup: 29 workgroups at 1,800 rows
constexpr int kQpFragsPerWave = 64 / kQpWaves;                                 // LDS-DMA instructions per wave and weight half
constexpr int kQpHalf = 8 * 8 * 64 * 16;                  // 64 KiB: 8 tiles x 8 k-steps of 1-KiB fragments
constexpr int kQpLdsBias = kQpHalf;                       // one weight half, then b1 | b2 | c1 | c2 as fp32
constexpr int kQpLdsBytes = kQpLdsBias + 4 * 256 * 4;

__global__ __launch_bounds__(kQpThreads) void query_pos_k256_kernel(
    const uint16_t *__restrict__ emb, long long lde, const uint16_t *__restrict__ query, long long ldq,
    const uint16_t *__restrict__ pw1a, const uint16_t *__restrict__ pw1b, const uint16_t *__restrict__ b1, const uint16_t *__restrict__ pw2,
    const uint16_t *__restrict__ b2, const uint16_t *__restrict__ pv1, const uint16_t *__restrict__ c1, const uint16_t *__restrict__ pv2,
    const uint16_t *__restrict__ c2, long long M, uint16_t *__restrict__ out_pos, uint16_t *__restrict__ out_qpp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qp_lds[];
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(qp_lds);
    float *bl = reinterpret_cast<float *>(qp_lds + kQpLdsBias);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const bool scaled = pv1 != nullptr;                                       // uniform: layers >= 1

    // half h (output tiles 8 h .. 8 h + 7) of a packed [256, 256] block: 64 fragments of 1 KiB dealt over the waves.  `fetch`
    // brings this wave's share into REGISTERS (coalesced 1-KiB loads, in flight behind the current phase's MFMAs); `commit` moves
    // it into the LDS buffer once every wave has left that buffer.  (LDS-DMA would save the registers, but each DMA instruction
    // costs 100-350 cycles of issue behind its M0 write: with 16 of them per wave and phase the kernel took 25-28 us.)
    // One register set, fetched one phase ahead (a second set, two phases ahead, measured the same and spilled).
    u32x4 stg[kQpFragsPerWave];
    auto fetch = [&](const uint16_t *packed, int h) {
#pragma unroll
        for (int i = 0; i < kQpFragsPerWave; ++i)
            stg[i] = reinterpret_cast<const u32x4 *>(packed)[((size_t)h * kQpHalf + (size_t)(wave * kQpFragsPerWave + i) * 1024) / 16 + lane];
        __builtin_amdgcn_sched_barrier(0);                                    // the loads are issued HERE, ahead of the phase's MFMAs
    };
    auto commit = [&]() {
        __syncthreads();                                                      // every wave has left the buffer
        u32x4 *dst = reinterpret_cast<u32x4 *>(qp_lds);
#pragma unroll
        for (int i = 0; i < kQpFragsPerWave; ++i) dst[(wave * kQpFragsPerWave + i) * 64 + lane] = stg[i];
        __syncthreads();                                                      // ... and sees the new half
    };

    fetch(pw1a, 0);
    for (int i = tid; i < 256; i += kQpThreads) {
        bl[i] = bf16_bits_to_f32(b1[i]);
        bl[256 + i] = bf16_bits_to_f32(b2[i]);
        bl[512 + i] = scaled ? bf16_bits_to_f32(c1[i]) : 0.f;
        bl[768 + i] = scaled ? bf16_bits_to_f32(c2[i]) : 0.f;
    }
    const long long row = ((long long)blockIdx.x * kQpWaves + wave) * kQpRows + col;
    const bool rok = row < M;
    u32x4 xe[16], xq[8];                                                      // B operands: k-step s = columns 32 s + 8 g .. + 7 of the lane's row
#pragma unroll
    for (int s = 0; s < 16; ++s) xe[s] = rok ? *reinterpret_cast<const u32x4 *>(emb + row * lde + 32 * s + 8 * g) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int s = 0; s < 8; ++s) xq[s] = rok ? *reinterpret_cast<const u32x4 *>(query + row * ldq + 32 * s + 8 * g) : u32x4{0u, 0u, 0u, 0u};

    auto mm = [&](const u32x4 &a, const u32x4 &bq, const f32x4 &c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(qp_bf16x8, a), __builtin_bit_cast(qp_bf16x8, bq), c, 0, 0, 0);
    };
    // acc[uu][e] += (tile pair uu of the half in LDS) x xin: 4 pairs x 8 k-steps x 2 tiles = 64 fragments, read as ONE stream
    // through a ring of four registers, three fragments ahead of their MFMA (left to itself hipcc reads every fragment into the
    // same register right before its MFMA and waits lgkmcnt(0): the full LDS latency 64 times per phase -- 2.9 us per phase)
    auto pairs = [&](int, const u32x4 *xin, f32x4 (&acc)[4][2]) {
        constexpr int kFrags = 64, kAhead = 3;          // deeper (7 ahead) measured the same
        u32x4 ring[4];
        auto frag = [&](int f) { return wl[(((2 * (f >> 4) + (f & 1)) * 8 + ((f >> 1) & 7)) * 64) + lane]; };
#pragma unroll
        for (int f = 0; f < kAhead; ++f) ring[f & 3] = frag(f);
#pragma unroll
        for (int f = 0; f < kFrags; ++f) {
            if (f + kAhead < kFrags) ring[(f + kAhead) & 3] = frag(f + kAhead);
            __builtin_amdgcn_sched_barrier(0);
            acc[f >> 4][f & 1] = mm(ring[f & 3], xin[(f >> 1) & 7], acc[f >> 4][f & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto zero = [&](f32x4 (&acc)[4][2]) {
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) acc[uu][0] = acc[uu][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // biased (+ ReLU'd) and rounded to bf16: outputs 32 u + 8 g .. + 7 of the lane's row, u = 4 h + uu
    auto finish = [&](const f32x4 (&acc)[4][2], const float *bias, int h, bool relu, u32x4 *y) {
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const int u = 4 * h + uu;
            const f32x4 lo = acc[uu][0] + *reinterpret_cast<const f32x4 *>(bias + 32 * u + 8 * g);
            const f32x4 hi = acc[uu][1] + *reinterpret_cast<const f32x4 *>(bias + 32 * u + 8 * g + 4);
            u32x4 p = {pack_bf16x2(lo.x, lo.y), pack_bf16x2(lo.z, lo.w), pack_bf16x2(hi.x, hi.y), pack_bf16x2(hi.z, hi.w)};
            if (relu) p = u32x4{relu_bf16x2(p.x), relu_bf16x2(p.y), relu_bf16x2(p.z), relu_bf16x2(p.w)};
            y[u] = p;
        }
    };

    f32x4 acc[4][2];
    u32x4 y1[8], pos[8];
    // ---- layer 1 of ref_point_head: K = 512 = two packed blocks ------------------------------------------------------------
    // phase i: fetch the half of phase i + 1 | MFMAs on the half in LDS | commit
    commit();                                                                 // W1a.h0 (and the biases)
    fetch(pw1b, 0);
    zero(acc);
    pairs(0, xe, acc);
    commit();                                                                 // W1b.h0
    fetch(pw1a, 1);
    pairs(0, xe + 8, acc);
    finish(acc, bl, 0, true, y1);
    commit();                                                                 // W1a.h1
    fetch(pw1b, 1);
    zero(acc);
    pairs(0, xe, acc);
    commit();                                                                 // W1b.h1
    fetch(pw2, 0);
    pairs(0, xe + 8, acc);
    finish(acc, bl, 1, true, y1);
    commit();                                                                 // W2.h0
    // ---- layer 2 of ref_point_head ---------------------------------------------------------------------------------------
    fetch(pw2, 1);
    zero(acc);
    pairs(0, y1, acc);
    finish(acc, bl + 256, 0, false, pos);
    commit();                                                                 // W2.h1
    if (scaled) fetch(pv1, 0);
    zero(acc);
    pairs(0, y1, acc);
    finish(acc, bl + 256, 1, false, pos);
    if (scaled) {
        // ---- query_scale(query), then the product -----------------------------------------------------------------------
        u32x4 sc[8];
        commit();                                                             // V1.h0
        fetch(pv1, 1);
        zero(acc);
        pairs(0, xq, acc);
        finish(acc, bl + 512, 0, true, y1);
        commit();                                                             // V1.h1
        fetch(pv2, 0);
        zero(acc);
        pairs(0, xq, acc);
        finish(acc, bl + 512, 1, true, y1);
        commit();                                                             // V2.h0
        fetch(pv2, 1);
        zero(acc);
        pairs(0, y1, acc);
        finish(acc, bl + 768, 0, false, sc);
        commit();                                                             // V2.h1
        zero(acc);
        pairs(0, y1, acc);
        finish(acc, bl + 768, 1, false, sc);
        // query_pos * scale, rounded to bf16 as torch's bf16 multiply (fp32 product of the two bf16 values, one rounding)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned a[4] = {pos[u].x, pos[u].y, pos[u].z, pos[u].w}, b[4] = {sc[u].x, sc[u].y, sc[u].z, sc[u].w};
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = pack_bf16x2(bf16_bits_to_f32(a[k] & 0xffffu) * bf16_bits_to_f32(b[k] & 0xffffu),
                                   __builtin_bit_cast(float, a[k] & 0xffff0000u) * __builtin_bit_cast(float, b[k] & 0xffff0000u));
            pos[u] = u32x4{o[0], o[1], o[2], o[3]};
        }
    }
    if (rok) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            *reinterpret_cast<u32x4 *>(out_pos + row * 256 + 32 * u + 8 * g) = pos[u];
            const unsigned a[4] = {pos[u].x, pos[u].y, pos[u].z, pos[u].w}, q[4] = {xq[u].x, xq[u].y, xq[u].z, xq[u].w};
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = pack_bf16x2(bf16_bits_to_f32(q[k] & 0xffffu) + bf16_bits_to_f32(a[k] & 0xffffu),
                                   __builtin_bit_cast(float, q[k] & 0xffff0000u) + __builtin_bit_cast(float, a[k] & 0xffff0000u));
            *reinterpret_cast<u32x4 *>(out_qpp + row * 256 + 32 * u + 8 * g) = u32x4{o[0], o[1], o[2], o[3]};
        }
    }
}

}  // namespace

}  // namespace rdetr

using namespace rdetr;

// out_pos [M, 256] = MLP2(emb [M, 512]) (* MLP2'(query [M, 256]) when pv1 / c1 / pv2 / c2 are given), out_qpp = query + out_pos; bf16.
// pw1a / pw1b: the K halves [:, :256] / [:, 256:] of ref_point_head.layers[0].weight [256, 512]; pw2, pv1, pv2: [256, 256] weights;
// all five packed by rdetr_linear_pack_k256_bf16.  b1, b2, c1, c2: bf16 [256].  Row strides lde / ldq in elements.
extern "C" int rdetr_query_pos_k256_bf16(const uint16_t *emb, long long lde, const uint16_t *query, long long ldq, const uint16_t *pw1a,
                                         const uint16_t *pw1b, const uint16_t *b1, const uint16_t *pw2, const uint16_t *b2, const uint16_t *pv1,
                                         const uint16_t *c1, const uint16_t *pv2, const uint16_t *c2, long long M, uint16_t *out_pos,
                                         uint16_t *out_qpp, void *stream)
{
    if (M < 0 || lde < 512 || ldq < 256) return RDETR_ERR_INVALID_ARG;
    if ((lde & 7) || (ldq & 7)) return RDETR_ERR_UNSUPPORTED;
    if (M == 0) return RDETR_OK;
    if (!emb || !query || !pw1a || !pw1b || !b1 || !pw2 || !b2 || !out_pos || !out_qpp) return RDETR_ERR_INVALID_ARG;
    const bool any = pv1 || c1 || pv2 || c2, all = pv1 && c1 && pv2 && c2;
    if (any && !all) return RDETR_ERR_INVALID_ARG;
    auto al = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (!al(emb) || !al(query) || !al(pw1a) || !al(pw1b) || !al(pw2) || (all && (!al(pv1) || !al(pv2))) || !al(out_pos) || !al(out_qpp))
        return RDETR_ERR_UNSUPPORTED;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(query_pos_k256_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kQpLdsBytes);
    if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
    const long long per = kQpWaves * kQpRows, nblk = (M + per - 1) / per;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(query_pos_k256_kernel, dim3((unsigned)nblk), dim3(kQpThreads), kQpLdsBytes, static_cast<hipStream_t>(stream), emb, lde,
                       query, ldq, pw1a, pw1b, b1, pw2, b2, pv1, c1, pv2, c2, M, out_pos, out_qpp);
    return launch_status();
}
