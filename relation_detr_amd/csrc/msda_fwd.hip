// Multi-scale deformable attention, forward -- hand-written for gfx950 (MI355X).
//
// Replaces the reference's ms_deformable_im2col_gpu_kernel
// (models/bricks/ops/cuda/ms_deform_im2col_cuda.cuh:226-288: one thread per output scalar, every
// thread re-reading the 16 (loc, weight) triples of its head and issuing 64 scattered 4-byte loads).
//
// Design ("query-run" kernel, H = 8 heads x D = 32 channels, P = 4 points, L <= 8 levels):
//   * one 64-lane wavefront owns ONE head and a run of consecutive queries: 8 queries x 8 lanes (fp32,
//     a lane holds 4 channels = 16 B of the 128-byte head row) or 16 queries x 4 lanes (bf16, 8 channels
//     = 16 B of the 64-byte head row).  Every gather is a 16-byte-per-lane buffer_load_dwordx4: the
//     texture addresser processes 4 lanes per clock whatever the width, so narrower loads only waste it.
//     In the encoder consecutive queries are neighbouring pixels, so the rows one instruction reads are
//     neighbouring rows of one head plane and the x0 / x0+1 corner instructions re-use each other's lines
//     (measured 12 % faster than a one-query-x-eight-heads mapping, which reads eight unrelated rows);
//   * the per-(query, point) bilinear set-up (pixel coords, 4 corner byte offsets, 4 corner weights
//     already multiplied by the attention weight) is computed ONCE by one lane and staged in LDS,
//     then broadcast to the lanes of the row with two LDS reads per point.  With 4 levels a lane prepares
//     4 consecutive points (one level), so its locations / weights arrive with 3 vector loads (every
//     vector-memory instruction costs 12-16 clocks in the texture addresser, whatever it moves), and
//     the 4 levels' constants come by scalar loads: no LDS table, no barrier in front of the inputs;
//   * bf16: the weighted sum runs on the matrix cores -- v_mfma_f32_4x4x4_16b_bf16 is 16 independent
//     4x4x4 products, one per query; the four 16-byte loads of a lane are re-paired by v_perm_b32 into
//     the B operand, the corner weights (bf16 high + low parts) are the A operand (mfma_point below);
//   * corners outside the level get the byte offset 0x80000000: the buffer descriptor's range
//     check returns 0 for them without a memory access, which is exactly the zero-padding rule of
//     ms_deform_im2col_cuda.cuh:44-67, so the inner loop has no branches;
//   * the (image, head) plane is addressed through one wave-uniform buffer descriptor with
//     32-bit byte offsets (S*H*D*sizeof(T) < 2^31 is checked on the host);
//   * hardware block ids are remapped so that each XCD (private L2) owns a contiguous range of
//     (image, head, query-tile) blocks (common.h: xcd_contiguous_block).
// Any other (H, D, P) runs msda_fwd_generic_kernel (one thread per output, 64-bit indexing).
#include <cstdlib>

#include "common.h"
#include "msda_qrun.h"

namespace rdetr {

// LT = compile-time level count (4, 5) or 0 for a run-time L in [1, 8].
// FUSED = false: `src_a` = sampling locations fp32 [B,Nq,H,L,P,2], `src_b` = soft-maxed weights fp32 [B,Nq,H,L,P]
//                (the reference operator's inputs, ms_deform_attn_cuda.cu:12-19).
// FUSED = true : `src_a` = raw sampling offsets [B,Nq,H,L,P,2], `src_b` = raw attention logits [B,Nq,H,L*P], both in
//                value's dtype, `ref` = reference points fp32 [B,Nq,L,ref_dim]; the softmax over L*P and
//                loc = ref + off/(W,H)  |  ref_xy + off/P * ref_wh * 0.5  (ms_deform_attn.py:326-349) happen in the
//                set-up phase, so neither locations nor weights ever exist in HBM.
template <typename T, int LT, bool FUSED>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void msda_fwd_qrun_kernel(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const void *__restrict__ src_a, const void *__restrict__ src_b, const float *__restrict__ ref, int ref_dim, int S,
    int L_rt, int Nq, int tiles_per_image, int nblk, T *__restrict__ out,
    const unsigned char *__restrict__ pad_mask, int ld_a, int ld_b, int head_major, int value_pix_bytes, int order)
{
    using IO = ValueIO<T>;
    constexpr int kSub = IO::kRunSub;            // lanes per head row
    constexpr int kCh = IO::kRunCh;              // channels per lane (16 bytes)
    constexpr int kSlots = kWave / kSub;         // queries per wave (8 fp32, 16 bf16)
    constexpr int kPtsPerLane = LT ? (LT * kPoints + kSub - 1) / kSub : (kMaxLevels * kPoints) / kSub;
    // LT == 4: a lane prepares kPtsPerLane CONSECUTIVE points (4 = one level for bf16, 2 for fp32), so that its share of the
    // locations / weights (or raw offsets / logits / reference point) arrives with 3 vector loads instead of 8-12 narrow ones
    // -- the texture addresser's instruction rate is one of the kernel's three co-limiters.  Otherwise points sub, sub + kSub, ...
    // LT == 5, bf16: 5 consecutive points per lane -- lane `sub` holds the last 4 - sub points of level sub and the first sub + 1
    // of level sub + 1 (kSpan: two sets of level constants / reference points per lane), 5 vector loads instead of 15.
    constexpr bool kSpan = LT == 5 && kSub == 4;
    constexpr bool kConsec = LT == 4 || kSpan;
    constexpr bool kMM = sizeof(T) == 2;         // bf16: the weighted sum on the matrix cores (mfma_point)
    constexpr bool kFast = FUSED && sizeof(T) == 2;      // bf16 producer inputs: see the softmax below
    const int L = LT ? LT : L_rt;
    const int LP = L * kPoints;

    __shared__ LevelTable lvl;
    constexpr int kStageLevels = LT ? LT : kMaxLevels;      // LDS per block: 16 KiB (fp32) / 32 KiB (bf16) at L = 4
    __shared__ u32x4 stage_off[kWavesPerBlock][kStageLevels * kPoints * kSlots];
    __shared__ f32x4 stage_wgt[kWavesPerBlock][kStageLevels * kPoints * kSlots];

    const int tid = threadIdx.x;
    if constexpr (!kConsec) {
        if (tid < L) {
            lvl.h[tid] = (int)shapes[2 * tid];
            lvl.w[tid] = (int)shapes[2 * tid + 1];
            lvl.start[tid] = (int)level_start[tid];
        }
        __syncthreads();
    }
    // kConsec: no table in LDS and no barrier -- a wave starts on its inputs at once.  The 4 levels' constants come by scalar
    // loads (uniform addresses) and a lane selects those of ITS level; a memory round trip + barrier in front of every
    // workgroup's input loads was a third of the launch.
    int my_h = 0, my_w = 0, my_start = 0, my_h1 = 0, my_w1 = 0, my_start1 = 0;       // ..1: the lane's second level (kSpan)
    if constexpr (kConsec) {
        const int my_level = ((int)(threadIdx.x % kSub) * kPtsPerLane) / kPoints;
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            const int h = (int)shapes[2 * l], w = (int)shapes[2 * l + 1], st = (int)level_start[l];
            my_h = my_level == l ? h : my_h;
            my_w = my_level == l ? w : my_w;
            my_start = my_level == l ? st : my_start;
            if constexpr (kSpan) {
                my_h1 = my_level + 1 == l ? h : my_h1;
                my_w1 = my_level + 1 == l ? w : my_w1;
                my_start1 = my_level + 1 == l ? st : my_start1;
            }
        }
    }
    // k = index of the point among the lane's own (kSpan: its points k >= 4 - sub belong to the lane's second level)
    auto second_level = [&](int k) { return kSpan && k >= kPoints - (int)(threadIdx.x % kSub); };
    auto level_h = [&](int l, int k) { return kConsec ? (second_level(k) ? my_h1 : my_h) : lvl.h[l]; };
    auto level_w = [&](int l, int k) { return kConsec ? (second_level(k) ? my_w1 : my_w) : lvl.w[l]; };
    auto level_start_of = [&](int l, int k) { return kConsec ? (second_level(k) ? my_start1 : my_start) : lvl.start[l]; };

    // logical block -> (image b, head m, tile of consecutive queries).  `order` bit 0: band-interleaved tiles (below); bits 1-2 =
    // log2 G of the HEAD GROUP: the G heads of a group take turns over the same tile (consecutive logical blocks = consecutive
    // dispatches on one XCD), the tiles of one (image, head group) are consecutive.  The query-side rows hold a query's 8 heads
    // side by side -- per head 128 B of locations + 64 B of weights at L = 4 (160 + 80 at L = 5; 64 + 32 B of raw offsets / logits
    // in the FUSED form) -- so with one head per XCD every L2 fetches the 128-byte lines of its neighbours' heads as well
    // (measured: 250 MB read for 183 MB distinct at R50, 1.71 GB for 0.99 GB at FocalNet); a group's heads share those lines in
    // ONE L2, at the price of G value planes' rows competing for it.
    const int logical = xcd_contiguous_block(blockIdx.x, nblk);
    const int hg = (order >> 1) & 3, interleave = order & 1;
    static_assert(kHeads == 8, "head-group decode assumes 8 heads");
    // Encoder shape (queries = the pyramid's pixels, `interleave`): the tiles of one (image, head) are taken in BANDS -- ~16-31
    // tiles of level 0 followed by the proportional share of every coarser level's tiles -- instead of level after level.  In
    // query order every level's queries sweep the WHOLE value plane once more (5 sweeps of 13 MB per plane at the FocalNet
    // size, where an XCD's L2 holds 4 MiB: 2.2 GB leave the L2s for 0.99 GB of distinct bytes, profiles/r04/
    // pmc_ea_sizes_direct_kernel.txt); interleaved, the queries of all levels over one part of the image run together and share
    // its rows.  A pure re-ordering of the blocks (a bijection on the tiles for any level table; identity unless the level
    // starts are non-decreasing): results are bit-identical.
    auto decode = [&](int lg, int &b_, int &m_, int &tile_) {
        const int per_group = tiles_per_image << hg;
        const int grp = lg / per_group, r_in = lg - grp * per_group;
        int tile = r_in >> hg;
        b_ = grp >> (3 - hg);
        m_ = ((grp & ((kHeads >> hg) - 1)) << hg) | (r_in & ((1 << hg) - 1));
        if constexpr (LT != 0) {
            if (interleave) {
                constexpr int kQpb = kWavesPerBlock * kSlots;
                int A[LT + 1];
                bool mono = true;
                A[0] = 0;
#pragma unroll
                for (int l = 1; l < LT; ++l) {
                    const int t = ((int)level_start[l] + kQpb - 1) / kQpb;
                    A[l] = t < tiles_per_image ? t : tiles_per_image;
                    mono = mono && A[l] >= A[l - 1];
                }
                A[LT] = tiles_per_image;
                const int T0 = A[1] - A[0];
                if (mono && T0 >= 64) {
                    const int k = 27 - __builtin_clz((unsigned)T0);                  // 2^k bands of 16..31 level-0 tiles
                    const int NB = 1 << k;
                    auto prefix = [&](int bnd) {
                        int sum = 0;
#pragma unroll
                        for (int l = 0; l < LT; ++l) sum += (bnd * (A[l + 1] - A[l])) >> k;
                        return sum;
                    };
                    int bnd = (int)((float)tile * (float)NB / (float)tiles_per_image);
                    bnd = bnd < 0 ? 0 : (bnd > NB - 1 ? NB - 1 : bnd);
                    while (bnd + 1 < NB && prefix(bnd + 1) <= tile) ++bnd;
                    while (bnd > 0 && prefix(bnd) > tile) --bnd;
                    int r = tile - prefix(bnd);
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        const int Tl = A[l + 1] - A[l], lo = (bnd * Tl) >> k, c = (((bnd + 1) * Tl) >> k) - lo;
                        if (r >= 0 && r < c) tile = A[l] + lo + r;
                        r = r >= 0 && r < c ? -1 : r - c;
                    }
                }
            }
        }
        tile_ = tile;
    };
    int b, m, tile;
    decode(logical, b, m, tile);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int qs = lane / kSub, sub = lane % kSub;
    const int q = (tile * kWavesPerBlock + wave) * kSlots + qs;
    const bool qok = q < Nq;

    // value [B,S,H,D] (the reference operator's layout: a pixel's heads side by side) or, head_major, [B,H,S,D]: either way
    // the (image, head) plane sits behind one wave-uniform buffer descriptor and a pixel step is `pixb` bytes
    // (`value_pix_bytes` = bytes from one pixel to the next in [B,S,H,D]: H*D*sizeof(T) for a dense tensor, more when the rows
    // are a column slice of a wider projection output -- the decoder's six cross-attention value projections as ONE GEMM)
    const unsigned pixb = head_major ? IO::kHeadBytes : (unsigned)value_pix_bytes;
    const __amdgpu_buffer_rsrc_t rsrc =
        head_major ? __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(value) + ((size_t)b * kHeads + m) * (size_t)S * kHeadDim, 0,
                                                       (unsigned)S * IO::kHeadBytes, 0x00020000)
                   : __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char *>(const_cast<T *>(value)) +
                                                           (size_t)b * S * (size_t)pixb + (size_t)m * IO::kHeadBytes,
                                                       0, (unsigned)S * pixb - (unsigned)m * IO::kHeadBytes, 0x00020000);
    const unsigned lane_off = (unsigned)sub * 16u;

    u32x4 *soff = stage_off[wave];
    f32x4 *swgt = stage_wgt[wave];
    const size_t row = (size_t)b * Nq + (qok ? q : 0);
    const size_t hrow = (row * kHeads + m) * (size_t)LP;

    // ---- set-up: lane (qs, sub) prepares points sub, sub+kSub, ... (kConsec: sub*kPtsPerLane, +1, ...) of query qs -------
    auto point_of = [&](int k) { return kConsec ? sub * kPtsPerLane + k : sub + k * kSub; };
    float pa[kPtsPerLane];
    f32x2 pxy[kPtsPerLane];
    if constexpr (FUSED) {
        // row strides (elements) of the two projection outputs: they may be column slices of ONE [rows, 3*H*L*P] GEMM output
        const T *off_q = static_cast<const T *>(src_a) + (ld_a ? row * (size_t)ld_a + (size_t)m * LP * 2 : hrow * 2);
        const T *lg_q = static_cast<const T *>(src_b) + (ld_b ? row * (size_t)ld_b + (size_t)m * LP : hrow);
        float mx = -__builtin_inff();
        if constexpr (kConsec) {
            float o2[2 * kPtsPerLane];
            LoadQ<T, 2 * kPtsPerLane>::run(off_q + 2 * point_of(0), o2);
            LoadQ<T, kPtsPerLane>::run(lg_q + point_of(0), pa);
#pragma unroll
            for (int k = 0; k < kPtsPerLane; ++k) {
                pxy[k] = f32x2{o2[2 * k], o2[2 * k + 1]};
                mx = fmaxf(mx, pa[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < kPtsPerLane; ++k) {
                const int pt = point_of(k);
                const bool ok = pt < LP;
                pa[k] = ok ? load_q<T>(lg_q + pt) : -__builtin_inff();
                pxy[k] = ok ? load_q2<T>(off_q + 2 * pt) : f32x2{0.f, 0.f};
                mx = fmaxf(mx, pa[k]);
            }
        }
        mx = group_max<kSub>(mx);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < kPtsPerLane; ++k) {
            // exp(-inf) = 0 for the padding slots.  bf16 producer inputs (kFast): the hardware exponential and, below, products
            // with reciprocals instead of 12 fp32 divisions per lane -- a tenth of the wave's vector-ALU instructions for
            // differences of an ulp of fp32 under inputs that carry 8 bits; the fp32 kernel keeps the reference's operations
            pa[k] = kFast ? __builtin_amdgcn_exp2f((pa[k] - mx) * 1.44269504088896341f) : expf(pa[k] - mx);
            sum += pa[k];
        }
        sum = group_sum<kSub>(sum);
        const float inv_sum = 1.0f / sum;
        // kConsec: the lane's points share a level -> one reference point (kSpan: two levels -> two, rc1 for the second)
        f32x4 rc = {0.f, 0.f, 0.f, 0.f}, rc0 = {0.f, 0.f, 0.f, 0.f}, rc1 = {0.f, 0.f, 0.f, 0.f};
        const float inv_w0 = kConsec ? 1.0f / (float)my_w : 0.f, inv_h0 = kConsec ? 1.0f / (float)my_h : 0.f;
        const float inv_w1 = kSpan ? 1.0f / (float)my_w1 : 0.f, inv_h1 = kSpan ? 1.0f / (float)my_h1 : 0.f;
        if constexpr (kConsec) {
            const float *rp = ref + (row * L + point_of(0) / kPoints) * (size_t)ref_dim;
            if (ref_dim == 2) {
                if constexpr (kSpan) {                 // levels sub, sub + 1: four floats, 8-byte aligned
                    float r4[4];
                    __builtin_memcpy(r4, __builtin_assume_aligned(rp, 8), 16);
                    rc0 = f32x4{r4[0], r4[1], 0.f, 0.f};
                    rc1 = f32x4{r4[2], r4[3], 0.f, 0.f};
                } else {
                    const f32x2 r2 = *reinterpret_cast<const f32x2 *>(rp);
                    rc0 = f32x4{r2.x, r2.y, 0.f, 0.f};
                }
            } else {
                rc0 = *reinterpret_cast<const f32x4 *>(rp);
                if constexpr (kSpan) rc1 = *reinterpret_cast<const f32x4 *>(rp + 4);
            }
        }
#pragma unroll
        for (int k = 0; k < kPtsPerLane; ++k) {
            const int pt = point_of(k);
            const int l = (pt < LP ? pt : 0) / kPoints;
            if constexpr (!kConsec) {
                const float *rp = ref + (row * L + l) * (size_t)ref_dim;
                rc = ref_dim == 2 ? f32x4{rp[0], rp[1], 0.f, 0.f} : f32x4{rp[0], rp[1], rp[2], rp[3]};
            } else {
                rc = second_level(k) ? rc1 : rc0;
            }
            pa[k] = kFast ? pa[k] * inv_sum : pa[k] / sum;
            if (ref_dim == 2) {
                if (kFast && kConsec) {
                    pxy[k].x = rc.x + pxy[k].x * (second_level(k) ? inv_w1 : inv_w0);
                    pxy[k].y = rc.y + pxy[k].y * (second_level(k) ? inv_h1 : inv_h0);
                } else {
                    pxy[k].x = rc.x + pxy[k].x / (float)level_w(l, k);
                    pxy[k].y = rc.y + pxy[k].y / (float)level_h(l, k);
                }
            } else {
                pxy[k].x = rc.x + pxy[k].x * (1.0f / kPoints) * rc.z * 0.5f;
                pxy[k].y = rc.y + pxy[k].y * (1.0f / kPoints) * rc.w * 0.5f;
            }
        }
    } else {
        const float *loc_q = static_cast<const float *>(src_a) + hrow * 2;
        const float *att_q = static_cast<const float *>(src_b) + hrow;
        if constexpr (kConsec) {
            float l2[2 * kPtsPerLane];
            LoadQ<float, 2 * kPtsPerLane>::run(loc_q + 2 * point_of(0), l2);
            LoadQ<float, kPtsPerLane>::run(att_q + point_of(0), pa);
#pragma unroll
            for (int k = 0; k < kPtsPerLane; ++k) pxy[k] = f32x2{l2[2 * k], l2[2 * k + 1]};
        } else {
#pragma unroll
            for (int k = 0; k < kPtsPerLane; ++k) {
                const int pt = point_of(k);
                const bool ok = pt < LP;
                pxy[k] = ok ? qload(reinterpret_cast<const f32x2 *>(loc_q + 2 * pt)) : f32x2{0.f, 0.f};
                pa[k] = ok ? qload(att_q + pt) : 0.f;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kPtsPerLane; ++k) {
        const int pt = point_of(k);
        if (pt < LP) {
            const f32x2 xy = pxy[k];
            const float a = pa[k];
            const int l = pt / kPoints;
            const int h = level_h(l, k), w = level_w(l, k);
            const float x = xy.x * (float)w - 0.5f;
            const float y = xy.y * (float)h - 0.5f;
            const bool inside = qok && (y > -1.f) && (x > -1.f) && (y < (float)h) && (x < (float)w);   // false for NaN
            const float xf = floorf(x), yf = floorf(y);
            const int x0 = inside ? (int)xf : 0, y0 = inside ? (int)yf : 0;
            const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
            const bool okx0 = inside && x0 >= 0, okx1 = inside && x0 + 1 <= w - 1;
            const bool oky0 = y0 >= 0, oky1 = y0 + 1 <= h - 1;
            const unsigned base = (unsigned)(level_start_of(l, k) + y0 * w + x0) * pixb;
            const unsigned rowb = (unsigned)w * pixb;
            u32x4 o;
            o.x = (okx0 && oky0) ? base : kInvalidOffset;
            o.y = (okx1 && oky0) ? base + pixb : kInvalidOffset;
            o.z = (okx0 && oky1) ? base + rowb : kInvalidOffset;
            o.w = (okx1 && oky1) ? base + rowb + pixb : kInvalidOffset;
            if (pad_mask) {                  // key_padding_mask: a padded pixel's projected value row counts as zero
                const unsigned char *mp = pad_mask + (size_t)b * S + (level_start_of(l, k) + y0 * w + x0);     // (ms_deform_attn.py:316-319)
                if (okx0 && oky0 && mp[0]) o.x = kInvalidOffset;
                if (okx1 && oky0 && mp[1]) o.y = kInvalidOffset;
                if (okx0 && oky1 && mp[w]) o.z = kInvalidOffset;
                if (okx1 && oky1 && mp[w + 1]) o.w = kInvalidOffset;
            }
            f32x4 wt;
            wt.x = inside ? hy * hx * a : 0.f;
            wt.y = inside ? hy * lx * a : 0.f;
            wt.z = inside ? ly * hx * a : 0.f;
            wt.w = inside ? ly * lx * a : 0.f;
            soff[pt * kSlots + qs] = o;
            if constexpr (kMM) {
                unsigned h01, l01, h23, l23;                // {hi(00,01), hi(10,11), lo(00,01), lo(10,11)}: A rows 0 and 1 of mfma_point
                split2_bf16(wt.x, wt.y, h01, l01);
                split2_bf16(wt.z, wt.w, h23, l23);
                swgt[pt * kSlots + qs] = __builtin_bit_cast(f32x4, u32x4{h01, h23, l01, l23});
            } else {
                swgt[pt * kSlots + qs] = wt;
            }
        }
    }
    // staging is private to this wave: LDS ops of one wave complete in order, so a wave-level
    // fence (no s_barrier) is all that is needed between the writes above and the reads below.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    if constexpr (kMM) {
        f32x4 accm[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) accm[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned wsel = (unsigned)(sub & 1) * 8u;
#pragma unroll 2
        for (int pt = 0; pt < LP; ++pt) {        // 8 loads of 16 B in flight per lane
            const u32x4 o = soff[pt * kSlots + qs];
            const u32x2 wq = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const unsigned char *>(swgt + pt * kSlots + qs) + wsel);
            const u32x4 r00 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.x + lane_off, 0, 0);
            const u32x4 r01 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.y + lane_off, 0, 0);
            const u32x4 r10 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.z + lane_off, 0, 0);
            const u32x4 r11 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o.w + lane_off, 0, 0);
            mfma_point(r00, r01, r10, r11, wq, accm);
        }
        float res[kCh];
#pragma unroll
        for (int c = 0; c < kCh; ++c) res[c] = accm[c % 8].x + accm[c % 8].y;
        if (qok) IO::store_run(out + row * (kHeads * kHeadDim) + m * kHeadDim + sub * kCh, res);
        return;
    }
    float acc[kCh];
#pragma unroll
    for (int c = 0; c < kCh; ++c) acc[c] = 0.f;
    constexpr int kUnroll = 16 / kCh;            // 16 loads in flight per lane for fp32
#pragma unroll kUnroll
    for (int pt = 0; pt < LP; ++pt) {
        const u32x4 o = soff[pt * kSlots + qs];
        const f32x4 wt = swgt[pt * kSlots + qs];
        float v00[kCh], v01[kCh], v10[kCh], v11[kCh];
        IO::load_run(rsrc, o.x + lane_off, v00);
        IO::load_run(rsrc, o.y + lane_off, v01);
        IO::load_run(rsrc, o.z + lane_off, v10);
        IO::load_run(rsrc, o.w + lane_off, v11);
        // explicit FMAs (the file is built with -ffp-contract=off): the loop is VALU-bound at 1 wave-instruction/clk/CU
        // (tools/microbench/valu_rate.hip), and mul + add costs twice the issue slots of v_pk_fma_f32
#pragma unroll
        for (int c = 0; c < kCh; ++c) {
            acc[c] = __builtin_fmaf(wt.x, v00[c], acc[c]);
            acc[c] = __builtin_fmaf(wt.y, v01[c], acc[c]);
            acc[c] = __builtin_fmaf(wt.z, v10[c], acc[c]);
            acc[c] = __builtin_fmaf(wt.w, v11[c], acc[c]);
        }
    }
    if (qok) IO::store_run(out + row * (kHeads * kHeadDim) + m * kHeadDim + sub * kCh, acc);
}

// ---------------------------------------------------------------------------------------------
// Generic fallback: one thread per output scalar, any (H, D, L, P), 64-bit indexing.
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<uint16_t>(uint16_t v) { return bf16_bits_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ uint16_t from_f32<uint16_t>(float v) { return (uint16_t)f32_to_bf16_bits(v); }

template <typename T>
__global__ __launch_bounds__(256) void msda_fwd_generic_kernel(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, int S, int H, int D, int L, int Nq, int P,
    long long total, T *__restrict__ out)
{
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % D);
        const long long r = idx / D;              // (b*Nq + q)*H + m
        const int m = (int)(r % H);
        const long long b = r / H / Nq;
        const long long pix = (long long)H * D;
        const T *vb = value + b * S * pix + (long long)m * D + c;
        const float *lp = loc + r * L * P * 2;
        const float *ap = attn + r * L * P;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) {
            const int h = (int)shapes[2 * l], w = (int)shapes[2 * l + 1];
            const T *vl = vb + level_start[l] * pix;
            for (int p = 0; p < P; ++p) {
                const float x = lp[(l * P + p) * 2] * (float)w - 0.5f;
                const float y = lp[(l * P + p) * 2 + 1] * (float)h - 0.5f;
                if (!((y > -1.f) && (x > -1.f) && (y < (float)h) && (x < (float)w))) continue;
                const float xf = floorf(x), yf = floorf(y);
                const int x0 = (int)xf, y0 = (int)yf;
                const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
                const float a = ap[l * P + p];
                float s = 0.f;
                if (y0 >= 0 && x0 >= 0) s += hy * hx * to_f32<T>(vl[((long long)y0 * w + x0) * pix]);
                if (y0 >= 0 && x0 + 1 <= w - 1) s += hy * lx * to_f32<T>(vl[((long long)y0 * w + x0 + 1) * pix]);
                if (y0 + 1 <= h - 1 && x0 >= 0) s += ly * hx * to_f32<T>(vl[((long long)(y0 + 1) * w + x0) * pix]);
                if (y0 + 1 <= h - 1 && x0 + 1 <= w - 1)
                    s += ly * lx * to_f32<T>(vl[((long long)(y0 + 1) * w + x0 + 1) * pix]);
                acc += s * a;
            }
        }
        out[idx] = from_f32<T>(acc);
    }
}

static bool fast_path(int H, int D, int L, int P)
{
    return H == kHeads && D == kHeadDim && P == kPoints && L >= 1 && L <= kMaxLevels;
}

// csrc/msda_win.hip: LDS-window MFMA kernel for the encoder shape (bf16, L == 4, Nq == S).
template <bool FUSED, bool HM>
int msda_win_forward(const uint16_t *value, const int64_t *shapes, const int64_t *level_start, const void *src_a,
                     const void *src_b, const float *ref, int ref_dim, int B, int S, int L, int Nq, int ld_a, int ld_b,
                     uint16_t *out, hipStream_t stream);

// development A/B of the block order (make dev: rdetr_dev_set_msda_identity_order); the product library always interleaves
struct MsdaOrder {
#ifdef RDETR_DEV
    static inline bool identity = false;
    static inline int head_group_log2 = -1;          // -1 = the product's choice
#else
    static constexpr bool identity = false;
    static constexpr int head_group_log2 = -1;
#endif
};

template <typename T, bool FUSED>
static void launch_qrun(dim3 grid, dim3 block, hipStream_t stream, const T *value, const int64_t *shapes,
                        const int64_t *level_start, const void *src_a, const void *src_b, const float *ref, int ref_dim,
                        int S, int L, int Nq, int tiles, int nblk, T *out, const unsigned char *pad_mask, int ld_a, int ld_b,
                        int head_major, int value_pix_bytes)
{
    // encoder shape with an (image, head) value plane that does not fit an XCD's 4-MiB L2: band-interleaved block order.
    // Same-box A/B (profiles/r04/ab_block_order_direct_kernel.txt): FocalNet-L 1200 x 2000 (13-MB planes) 872 -> 840 us and
    // 2.19 -> 1.71 GB of L2 -> fabric reads; R50 800 x 1333 (1.4-MB planes, they stay in the L2 between the sweeps) 109 -> 113 us:
    // there the query order keeps the better L1 locality.
    const int interleave = (Nq == S && (long long)S * (long long)(kHeadDim * sizeof(T)) > (4ll << 20) && !MsdaOrder::identity) ? 1 : 0;
    // head pairs: same-box A/B over G = 1, 2, 4, 8 (profiles/r04/ab_head_group_direct_kernel.txt): FocalNet-L operator form 838 ->
    // 805 us and 13.4 M -> 11.1 M read requests with G = 2 (G = 4: 823 us, 17.7 M -- four planes' rows no longer fit; G = 8: 862 us,
    // 39 M), fused form 787 -> 747 us; [B,S,H,D] value (a pair's rows = one 128-byte line) 1179 -> 1014 us, at R50 127 -> 122 us;
    // R50 head-major: equal within the boxes' clock noise, 1.95 M -> 1.82 M requests
    const int hg = MsdaOrder::head_group_log2 >= 0 ? MsdaOrder::head_group_log2 : 1;
    const int order = interleave | (hg << 1);
    // the 4-level kernel reads a lane's share of the query-side inputs as 16-byte vectors
    const bool vec_ok = reinterpret_cast<uintptr_t>(src_a) % 16 == 0 && reinterpret_cast<uintptr_t>(src_b) % 16 == 0 &&
                        (!FUSED || (reinterpret_cast<uintptr_t>(ref) % 16 == 0 && (ld_a * (int)sizeof(T)) % 16 == 0 &&
                                    (ld_b * (int)sizeof(T)) % 16 == 0));
    // the 5-level bf16 kernel reads runs of 5 / 10 values at their natural alignment and two reference points at once
    const bool vec5_ok = sizeof(T) == 4 ||
                         (reinterpret_cast<uintptr_t>(src_a) % 8 == 0 && reinterpret_cast<uintptr_t>(src_b) % 4 == 0 &&
                          (!FUSED || (reinterpret_cast<uintptr_t>(ref) % 16 == 0 && (ld_a * (int)sizeof(T)) % 4 == 0)));
    if (L == 4 && vec_ok)
        hipLaunchKernelGGL((msda_fwd_qrun_kernel<T, 4, FUSED>), grid, block, 0, stream, value, shapes, level_start, src_a,
                           src_b, ref, ref_dim, S, L, Nq, tiles, nblk, out, pad_mask, ld_a, ld_b, head_major, value_pix_bytes, order);
    else if (L == 5 && vec5_ok)
        hipLaunchKernelGGL((msda_fwd_qrun_kernel<T, 5, FUSED>), grid, block, 0, stream, value, shapes, level_start, src_a,
                           src_b, ref, ref_dim, S, L, Nq, tiles, nblk, out, pad_mask, ld_a, ld_b, head_major, value_pix_bytes, order);
    else
        hipLaunchKernelGGL((msda_fwd_qrun_kernel<T, 0, FUSED>), grid, block, 0, stream, value, shapes, level_start, src_a,
                           src_b, ref, ref_dim, S, L, Nq, tiles, nblk, out, pad_mask, ld_a, ld_b, head_major, value_pix_bytes, order);
}

// FUSED = false: src_a / src_b = sampling locations / soft-maxed weights (fp32).
// FUSED = true : src_a / src_b = raw offsets / logits in T, ref = reference points; fast-path shapes only.
// layout: RDETR_VALUE_BSHD | RDETR_VALUE_BHSD.  algo: RDETR_MSDA_AUTO | _DIRECT | _WINDOW (explicit arguments -- the library
// reads no environment and keeps no state).
template <typename T, bool FUSED>
static int msda_forward(const T *value, int layout, const int64_t *shapes, const int64_t *level_start, const void *src_a,
                        const void *src_b, const float *ref, int ref_dim, int B, int S, int H, int D, int L, int Nq,
                        int P, T *out, hipStream_t stream, int algo = RDETR_MSDA_AUTO, const unsigned char *pad_mask = nullptr,
                        int ld_a = 0, int ld_b = 0, long long value_ld = 0)
{
    if (B < 0 || S < 0 || Nq < 0 || H <= 0 || D <= 0 || L <= 0 || P <= 0) return RDETR_ERR_INVALID_ARG;
    if (layout != RDETR_VALUE_BSHD && layout != RDETR_VALUE_BHSD) return RDETR_ERR_INVALID_ARG;
    if (algo != RDETR_MSDA_AUTO && algo != RDETR_MSDA_DIRECT && algo != RDETR_MSDA_WINDOW && algo != RDETR_MSDA_AUTO_PACKED)
        return RDETR_ERR_INVALID_ARG;
    if (FUSED && ref_dim != 2 && ref_dim != 4) return RDETR_ERR_INVALID_ARG;
    if (B == 0 || Nq == 0) return RDETR_OK;
    if (!value || !shapes || !level_start || !src_a || !src_b || !out || (FUSED && !ref)) return RDETR_ERR_INVALID_ARG;
    if (S == 0) return RDETR_ERR_INVALID_ARG;
    const bool hm = layout == RDETR_VALUE_BHSD;

    // value_ld (elements, [B,S,H,D] only): row stride of a value tensor whose pixels are column slices of a wider buffer; 0 = dense
    if (value_ld && (hm || value_ld < (long long)H * D || (value_ld * (long long)sizeof(T)) % 16 != 0)) return RDETR_ERR_INVALID_ARG;
    const long long pixel_bytes = (value_ld ? value_ld : (long long)H * D) * (long long)sizeof(T);
    const bool aligned = (reinterpret_cast<uintptr_t>(value) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0) &&
                         (reinterpret_cast<uintptr_t>(src_a) % 8 == 0) && (reinterpret_cast<uintptr_t>(src_b) % 4 == 0);
    if (fast_path(H, D, L, P) && aligned && (long long)S * pixel_bytes < (1ll << 31)) {
        if constexpr (sizeof(T) == 2) {
            // encoder shape (queries = the pyramid's own pixels): the LDS-window MFMA kernel.  It enumerates its queries as the
            // pixels of the levels, so it is only correct for a level table that tiles [0, S) -- which lives in device memory
            // and cannot be checked here: plain AUTO therefore never takes it.  AUTO_PACKED (the caller has checked the table,
            // rdetr_msda_levels_window_ok) takes it for the reference operator's layout (measured at BASELINE.json configs[1]:
            // 127-132 us vs 141 us direct) and the direct kernel for the head-major one (118 us direct vs 127 us window).
            if ((algo == RDETR_MSDA_WINDOW || (algo == RDETR_MSDA_AUTO_PACKED && !hm)) && !pad_mask && !value_ld) {
                const int st = hm ? msda_win_forward<FUSED, true>(value, shapes, level_start, src_a, src_b, ref, ref_dim, B, S,
                                                                  L, Nq, ld_a, ld_b, out, stream)
                                  : msda_win_forward<FUSED, false>(value, shapes, level_start, src_a, src_b, ref, ref_dim, B,
                                                                   S, L, Nq, ld_a, ld_b, out, stream);
                if (st != RDETR_ERR_UNSUPPORTED) return st;
            }
        }
        if (algo == RDETR_MSDA_WINDOW) return RDETR_ERR_UNSUPPORTED;
        const int slots = kWave / ValueIO<T>::kRunSub;                       // queries per wave (8 fp32 / 16 bf16)
        const int qpb = kWavesPerBlock * slots;
        const long long tiles = (Nq + qpb - 1) / qpb;
        const long long nblk = (long long)B * H * tiles;
        if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
        launch_qrun<T, FUSED>(dim3((unsigned)nblk), dim3(kWavesPerBlock * kWave), stream, value, shapes, level_start,
                              src_a, src_b, ref, ref_dim, S, L, Nq, (int)tiles, (int)nblk, out, pad_mask, ld_a, ld_b, hm ? 1 : 0,
                              (int)pixel_bytes);
        return launch_status();
    }
    if (FUSED || pad_mask || ld_a || ld_b || hm || value_ld || algo == RDETR_MSDA_WINDOW)
        return RDETR_ERR_UNSUPPORTED;      // callers fall back to producing loc / weights themselves
    const long long total = (long long)B * Nq * H * D;
    const long long want = (total + 255) / 256;
    dim3 grid((unsigned)(want < 16384 ? want : 16384)), block(256);
    hipLaunchKernelGGL((msda_fwd_generic_kernel<T>), grid, block, 0, stream, value, shapes, level_start,
                       static_cast<const float *>(src_a), static_cast<const float *>(src_b), S, H, D, L, Nq, P, total, out);
    return launch_status();
}

// ---------------------------------------------------------------------------------------------
// [B,S,H*D] rows (row stride `ld` elements) -> head-major [B,H,S,D], rows of padded positions zeroed on the way
// (ms_deform_attn.py:316-319): the layout the window kernel fills its LDS windows from at the contiguous-row rate.
// One workgroup = 64 positions: coalesced 512-byte row reads, LDS transpose, coalesced 4-KiB head runs out.
__global__ __launch_bounds__(256) void value_to_head_major_kernel(const uint16_t *__restrict__ src, long long ld,
                                                                  const unsigned char *__restrict__ mask, int S,
                                                                  uint16_t *__restrict__ dst)
{
    __shared__ u32x4 tile[64 * 32 + 64];                                   // 64 positions x 32 chunks of 16 B (+ skew)
    const int b = blockIdx.y, s0 = blockIdx.x * 64, tid = threadIdx.x;
    const int n = S - s0 < 64 ? S - s0 : 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + i * 256, r = e >> 5, c = e & 31;               // position r, 16-byte chunk c (head c >> 2)
        if (r < n) {
            u32x4 v = *reinterpret_cast<const u32x4 *>(src + ((size_t)b * S + s0 + r) * (size_t)ld + c * 8);
            if (mask && mask[(size_t)b * S + s0 + r]) v = u32x4{0u, 0u, 0u, 0u};
            tile[r * 32 + (c ^ (r & 31))] = v;                             // skew: the transposed read below is conflict-free
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + i * 256, h = e >> 8, r = (e >> 2) & 63, k = e & 3;      // head h: 64 positions x 4 chunks, contiguous
        if (r < n) {
            const int c = h * 4 + k;
            *reinterpret_cast<u32x4 *>(dst + (((size_t)b * kHeads + h) * (size_t)S + s0 + r) * kHeadDim + k * 8) = tile[r * 32 + (c ^ (r & 31))];
        }
    }
}

}  // namespace rdetr

extern "C" int rdetr_msda_fast_path(int H, int D, int L, int P) { return rdetr::fast_path(H, D, L, P) ? 1 : 0; }

extern "C" int rdetr_msda_levels_window_ok(const int64_t *shapes, const int64_t *level_start, int L, long long S)
{
    if (!shapes || !level_start || L <= 0) return 0;
    long long run = 0;
    for (int l = 0; l < L; ++l) {
        const long long h = shapes[2 * l], w = shapes[2 * l + 1];
        if (h <= 0 || w <= 0 || level_start[l] != run) return 0;
        if (l > 0 && (h > shapes[0] || w > shapes[1])) return 0;        // the tile tables assume no level outgrows level 0
        run += h * w;
    }
    return run == S ? 1 : 0;
}

extern "C" int rdetr_msda_forward_f32(const float *value, const int64_t *spatial_shapes,
                                      const int64_t *level_start_index, const float *sampling_loc,
                                      const float *attn_weight, int B, int S, int H, int D, int L, int Nq, int P,
                                      float *out, void *stream)
{
    return rdetr::msda_forward<float, false>(value, RDETR_VALUE_BSHD, spatial_shapes, level_start_index, sampling_loc,
                                             attn_weight, nullptr, 0, B, S, H, D, L, Nq, P, out, static_cast<hipStream_t>(stream));
}

extern "C" int rdetr_msda_forward_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                                       const int64_t *level_start_index, const float *sampling_loc,
                                       const float *attn_weight, int B, int S, int H, int D, int L, int Nq, int P,
                                       uint16_t *out, void *stream)
{
    return rdetr::msda_forward<uint16_t, false>(value, RDETR_VALUE_BSHD, spatial_shapes, level_start_index, sampling_loc,
                                                attn_weight, nullptr, 0, B, S, H, D, L, Nq, P, out,
                                                static_cast<hipStream_t>(stream));
}

extern "C" int rdetr_msda_forward_fused_f32(const float *value, const int64_t *spatial_shapes,
                                            const int64_t *level_start_index, const float *sampling_offsets,
                                            const float *attn_logits, const float *reference_points, int ref_dim, int B,
                                            int S, int H, int D, int L, int Nq, int P, float *out, void *stream)
{
    return rdetr::msda_forward<float, true>(value, RDETR_VALUE_BSHD, spatial_shapes, level_start_index, sampling_offsets,
                                            attn_logits, reference_points, ref_dim, B, S, H, D, L, Nq, P, out,
                                            static_cast<hipStream_t>(stream));
}

extern "C" int rdetr_msda_forward_fused_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                                             const int64_t *level_start_index, const uint16_t *sampling_offsets,
                                             const uint16_t *attn_logits, const float *reference_points, int ref_dim,
                                             int B, int S, int H, int D, int L, int Nq, int P, uint16_t *out, void *stream)
{
    return rdetr::msda_forward<uint16_t, true>(value, RDETR_VALUE_BSHD, spatial_shapes, level_start_index, sampling_offsets,
                                               attn_logits, reference_points, ref_dim, B, S, H, D, L, Nq, P, out,
                                               static_cast<hipStream_t>(stream));
}

// Fused-producer form, general: optional padding mask (applied inside the gather: no fill pass over the projected value) and
// row strides of the two projection outputs.
extern "C" int rdetr_msda_forward_fused_ex_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                               const float *sampling_offsets, int ld_offsets, const float *attn_logits,
                                               int ld_logits, const float *reference_points, int ref_dim,
                                               const uint8_t *key_padding_mask, int B, int S, int H, int D, int L, int Nq, int P,
                                               float *out, void *stream)
{
    if (ld_offsets < 0 || ld_logits < 0 || (ld_offsets && ld_offsets < H * L * P * 2) || (ld_logits && ld_logits < H * L * P) ||
        ld_offsets % 2 != 0)
        return RDETR_ERR_INVALID_ARG;
    return rdetr::msda_forward<float, true>(value, RDETR_VALUE_BSHD, spatial_shapes, level_start_index, sampling_offsets,
                                            attn_logits, reference_points, ref_dim, B, S, H, D, L, Nq, P, out,
                                            static_cast<hipStream_t>(stream), RDETR_MSDA_AUTO, key_padding_mask, ld_offsets,
                                            ld_logits);
}

extern "C" int rdetr_msda_forward_fused_ex_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                                                const int64_t *level_start_index, const uint16_t *sampling_offsets,
                                                int ld_offsets, const uint16_t *attn_logits, int ld_logits,
                                                const float *reference_points, int ref_dim, const uint8_t *key_padding_mask,
                                                int B, int S, int H, int D, int L, int Nq, int P, uint16_t *out, void *stream)
{
    if (ld_offsets < 0 || ld_logits < 0 || (ld_offsets && ld_offsets < H * L * P * 2) || (ld_logits && ld_logits < H * L * P) ||
        ld_offsets % 2 != 0)
        return RDETR_ERR_INVALID_ARG;
    return rdetr::msda_forward<uint16_t, true>(value, RDETR_VALUE_BSHD, spatial_shapes, level_start_index, sampling_offsets,
                                               attn_logits, reference_points, ref_dim, B, S, H, D, L, Nq, P, out,
                                               static_cast<hipStream_t>(stream), RDETR_MSDA_AUTO, key_padding_mask, ld_offsets,
                                               ld_logits);
}

// bf16 operator with the value layout and the kernel choice as explicit arguments (tests, A/B timing, and the module path,
// whose value projection writes the head-major layout).
extern "C" int rdetr_msda_forward_opt_bf16(const uint16_t *value, int value_layout, const int64_t *spatial_shapes,
                                           const int64_t *level_start_index, const float *sampling_loc,
                                           const float *attn_weight, int B, int S, int H, int D, int L, int Nq, int P,
                                           int algo, uint16_t *out, void *stream)
{
    return rdetr::msda_forward<uint16_t, false>(value, value_layout, spatial_shapes, level_start_index, sampling_loc,
                                                attn_weight, nullptr, 0, B, S, H, D, L, Nq, P, out,
                                                static_cast<hipStream_t>(stream), algo);
}

extern "C" int rdetr_msda_forward_fused_opt_bf16(const uint16_t *value, int value_layout, const int64_t *spatial_shapes,
                                                 const int64_t *level_start_index, const uint16_t *sampling_offsets,
                                                 int ld_offsets, const uint16_t *attn_logits, int ld_logits,
                                                 const float *reference_points, int ref_dim, const uint8_t *key_padding_mask,
                                                 int B, int S, int H, int D, int L, int Nq, int P, int algo, uint16_t *out,
                                                 void *stream)
{
    if (ld_offsets < 0 || ld_logits < 0 || (ld_offsets && ld_offsets < H * L * P * 2) || (ld_logits && ld_logits < H * L * P) ||
        ld_offsets % 2 != 0)
        return RDETR_ERR_INVALID_ARG;
    return rdetr::msda_forward<uint16_t, true>(value, value_layout, spatial_shapes, level_start_index, sampling_offsets,
                                               attn_logits, reference_points, ref_dim, B, S, H, D, L, Nq, P, out,
                                               static_cast<hipStream_t>(stream), algo, key_padding_mask, ld_offsets, ld_logits);
}

// Fused-producer form on a ROW-STRIDED value [B,S,H,D]: pixel s of image b starts at value + (b * S + s) * value_ld elements
// (value_ld >= H*D, 16-byte multiple).  Direct kernel.
extern "C" int rdetr_msda_forward_fused_strided_bf16(const uint16_t *value, long long value_ld, const int64_t *spatial_shapes,
                                                     const int64_t *level_start_index, const uint16_t *sampling_offsets, int ld_offsets,
                                                     const uint16_t *attn_logits, int ld_logits, const float *reference_points,
                                                     int ref_dim, const uint8_t *key_padding_mask, int B, int S, int H, int D, int L, int Nq,
                                                     int P, uint16_t *out, void *stream)
{
    if (ld_offsets < 0 || ld_logits < 0 || (ld_offsets && ld_offsets < H * L * P * 2) || (ld_logits && ld_logits < H * L * P) ||
        ld_offsets % 2 != 0 || value_ld < 0)
        return RDETR_ERR_INVALID_ARG;
    return rdetr::msda_forward<uint16_t, true>(value, RDETR_VALUE_BSHD, spatial_shapes, level_start_index, sampling_offsets,
                                               attn_logits, reference_points, ref_dim, B, S, H, D, L, Nq, P, out,
                                               static_cast<hipStream_t>(stream), RDETR_MSDA_DIRECT, key_padding_mask, ld_offsets,
                                               ld_logits, value_ld);
}

extern "C" int rdetr_value_to_head_major_bf16(const uint16_t *src, long long ld, const uint8_t *key_padding_mask, int B, int S,
                                              int H, int D, uint16_t *dst, void *stream)
{
    if (B < 0 || S < 0 || H != rdetr::kHeads || D != rdetr::kHeadDim || ld < H * D || ld % 8) return RDETR_ERR_INVALID_ARG;
    if (B == 0 || S == 0) return RDETR_OK;
    if (!src || !dst || reinterpret_cast<uintptr_t>(src) % 16 || reinterpret_cast<uintptr_t>(dst) % 16) return RDETR_ERR_INVALID_ARG;
    if (B > 65535) return RDETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(rdetr::value_to_head_major_kernel, dim3((unsigned)((S + 63) / 64), (unsigned)B), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, ld, key_padding_mask, S, dst);
    return rdetr::launch_status();
}

#ifdef RDETR_DEV
extern "C" void rdetr_dev_set_msda_identity_order(int v) { rdetr::MsdaOrder::identity = v != 0; }
extern "C" void rdetr_dev_set_msda_head_group_log2(int v) { rdetr::MsdaOrder::head_group_log2 = v > 3 ? 3 : v; }
#endif
