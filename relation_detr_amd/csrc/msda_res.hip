// Multi-scale deformable attention, forward, bf16, head-major value [B,H,S,D] -- "resident coarse levels" kernel for gfx950.
//
// Same operator as msda_fwd.hip (reference: ms_deform_im2col_cuda.cuh:226-288) and the same arithmetic per sampling point; the points
// of a query are accumulated in another order, so results agree with msda_fwd_qrun_kernel's to fp32 re-association (the last bit of
// a bf16 output now and then: 4e-5 of the outputs at the R50 shape).  What changes is WHERE the corner rows come from.
//
// The query-run kernel brings every corner row through the texture path, whose addresser retires one 64-lane x 16-byte
// instruction per ~17 clocks and CU: 64 such gathers per wave and run of 16 queries at 4 levels -- that kernel's ceiling
// (DESIGN 4.1).  But the levels are not alike: at the R50 encoder shape levels 2 and 3 together are 1,323 pixels = 85 KB of
// an (image, head) plane and receive HALF of all samples.  So:
//   * one persistent 12-wave workgroup per CU serves ONE (image, head) plane at a time and keeps that plane's coarse levels
//     LR .. L-1 -- the last one or two, as many as fit beside the staging area -- RESIDENT in LDS: one contiguous copy, since the
//     levels are packed along S;
//   * a sample on a resident level reads its corner rows with ds_read_b128 from that copy (a corner outside the level reads a
//     64-byte row of zeros at LDS offset 0: the zero padding of .cuh:44-67), a sample on a fine level goes through the buffer
//     descriptor as before.  No windows, no halo, no flagged samples: a resident level is resident as a whole.  The texture path
//     and the LDS are independent units (tools/microbench/ta_lds_concurrency.hip: mixed traffic takes the maximum, not the sum);
//   * a lane prepares two points of levels 0 / 1 and two (three) coarse points; the four corner OFFSETS of a point stay in its
//     registers and reach the query's other three lanes inside the address add (v_add_u32_dpp quad_perm: no LDS round trip, no
//     extra instruction), only the split corner WEIGHTS -- the matrix-core A operand, which differs from lane to lane -- are
//     staged in LDS (4 KiB per wave);
//   * eight software-pipelined steps per run: ask for the plane rows of fine point i + 1, work off a coarse point from LDS, take
//     in fine point i; the next run's inputs are asked for behind the run's last plane rows (loads return in order);
//   * encoder shape (Nq == S): a run is a 4 x 4 TILE of one level instead of 16 consecutive pixels (a third fewer distinct rows
//     per gather instruction: L1 -> L2 requests 10.6 M -> 6.6 M per launch);
//   * an XCD (private L2) serves CONSECUTIVE planes -- neighbouring heads of one image, which share the 128-byte lines of the
//     query-side rows -- with its workgroups split into teams, a team per plane in flight, the workgroups of a team sweeping the
//     plane's runs together (run r of workgroup g, wave w: r = k * 12 G + 12 g + w).
// Measured at BASELINE.json configs[1] (B = 4, S = Nq = 22,323): 91-96 us against the query-run kernel's 106-110 (0.30 vs 0.26 of
// the HBM roofline on SURVEY 8d's bytes); the vector ALU's share is ~56 us, the texture path's ~46, the query-side streams' ~41
// (profiles/r04/components_msda_res.txt, pmc_msda_res_B4_encoder.txt).  Tried and not kept: 8 and 16 waves, deeper pipelines, a
// paired-wave form (fine and coarse points in partner waves) -- profiles/r04/README.md.
// The level table is a HOST argument here (the resident set is sized on the host); callers that only have the device tensors
// use the query-run kernel.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "msda_qrun.h"

namespace rdetr {
namespace {

constexpr int kRQ = 16;                          // queries per wave and run (bf16: 4 lanes x 16 B per head row)
constexpr unsigned kResBase = 128;               // LDS: [0, 64) zero row, [128, 128 + res_bytes) resident levels, then staging
constexpr int kLdsBytes = 160 * 1024;

struct ResArgs {
    const uint16_t *value;
    const void *src_a;
    const void *src_b;
    const float *ref;
    uint16_t *out;
    int h[5], w[5], start[5];
    int S, Nq, B, ld_a, ld_b;
    int lr;                                      // first resident level
    int start_lr;                                // its first pixel
    int res_bytes;                               // bytes of levels lr .. L-1 of one plane (multiple of 64)
    int stage_base;                              // LDS offset of the staging area (multiple of 128)
    // Nq == S (the queries are the pyramid's pixels): a run is a 4 x 4 TILE of one level instead of 16 consecutive pixels
    int tiled, runs;                             // runs = tiles of all levels | ceil(Nq / 16)
    int plane_major;                             // development A/B: 0 = plane p on XCD p % 8
    int max_teams;                               // planes of an XCD in flight at a time
    int tpre[5], tw[5];                          // tiles before level l, tiles per tile row
    float inv_tw[5];
};

// N consecutive query-side values at a stated alignment (the compiler picks the widest legal loads)
template <int N, int ALIGN> __device__ __forceinline__ void load_f32(const float *p, float (&v)[N])
{
    __builtin_memcpy(v, __builtin_assume_aligned(p, ALIGN), N * 4);
}
template <int N, int ALIGN> __device__ __forceinline__ void load_bf16(const uint16_t *p, float (&v)[N])
{
    uint16_t r[N];
    __builtin_memcpy(r, __builtin_assume_aligned(p, ALIGN), N * 2);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = bf16_bits_to_f32(r[i]);
}

// What a lane needs to know about one level: size, the pixel index whose byte offset (x 64) addresses the level's pixel 0 -- in
// the plane (buffer path) or in the resident copy (LDS path) -- and the offset that stands for "outside" on that path.
struct LaneLevel {
    int h, w, st;
    unsigned inv;
};

// RW = waves per workgroup, LR = first resident level (LT - 2 or LT - 1): which path a point takes is known at compile time, so
// the rounds are straight-line code (with a run-time branch per point hipcc kept two sets of accumulators: 32 registers)
// DBG (development builds: component timing, wrong results): 1 = the fine points read the LDS zero row instead of the plane (no
// gathers on the texture path), 2 = the coarse points read the zero row (no bank conflicts), 4 = no set-up arithmetic (constant
// offsets and weights), 8 = no matrix-core steps
template <int LT, bool FUSED, int RW, int LR, int DBG = 0>
__global__ __launch_bounds__(RW *kWave) void msda_fwd_res_kernel(const ResArgs a)
{
    constexpr int kRW = RW;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    constexpr int LP = LT * kPoints;
    // A lane prepares two consecutive points of levels 0 / 1 (points 0..7: 2 sub, 2 sub + 1) and two (three) consecutive points of
    // the coarse levels (points 8..L*P-1: 8 + kB sub ..): every lane has work in both groups, and its loads stay vectors.
    constexpr int kFirstB = 2 * kPoints;
    constexpr int kB = (LP - kFirstB) / 4;       // 2 (L = 4) or 3 (L = 5)
    constexpr unsigned kStageBytes = LP * kRQ * 16;          // per wave: the split corner weights of all L*P points of 16 queries
    static_assert(LT == 4 || LT == 5, "4 or 5 levels");
    using IO = ValueIO<uint16_t>;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qs = lane >> 2, sub = lane & 3;
    const unsigned lane_off = (unsigned)sub * 16u;

    // ---- which planes, which share of their runs
    const int nx = (int)gridDim.x >> 3;                              // workgroups per XCD
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int planes = a.B * kHeads;
    // An XCD's planes are CONSECUTIVE: planes = 8 B is a multiple of 8, so XCD x serves planes x B .. x B + B - 1 -- neighbouring
    // heads of one image.  The query-side rows hold a query's 8 heads side by side (per head 128 B of locations + 64 B of weights;
    // fused form 64 B of raw offsets + 32 B of logits at L = 4): with head m on XCD m every L2 fetched its neighbours' heads' halves
    // of each 128-byte line as well; the planes of one XCD now sweep the same queries at the same pace and share those lines.
    const int Jx = planes >> 3;
    if (Jx <= 0) return;
    // a.max_teams planes of an XCD are in flight at a time (their fine levels' bands share its 4-MiB L2 with the query-side streams);
    // a team of G workgroups takes planes team, team + teams, ... one after the other
    const int want = Jx < a.max_teams ? Jx : a.max_teams;
    const int teams = want < nx ? want : nx;
    const int G = nx / teams;                                        // workgroups per plane
    const int team = slot / G, g = slot - team * G;
    if (team >= teams) return;                                       // nx not a multiple of Jx: the remainder idles

    // ---- the lane's level constants, once per wave
    auto level_of = [&](int lvl) {
        LaneLevel c{1, 1, 0, kInvalidOffset};
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            const bool res = l >= LR;
            const int st = res ? (int)(kResBase / 64) + a.start[l] - a.start_lr : a.start[l];
            if (lvl == l) c = LaneLevel{a.h[l], a.w[l], st, res ? 0u : kInvalidOffset};
        }
        return c;
    };
    const int pA = 2 * sub, pB = kFirstB + kB * sub;                 // the lane's first point of each half
    const int lvA = pA >> 2, lvB = pB >> 2;
    const int lvB1 = lvB + 1 < LT ? lvB + 1 : LT - 1;                // L = 5: a lane's three points may span two levels
    const LaneLevel cA = level_of(lvA), cB0 = level_of(lvB), cB1 = level_of(lvB1);
    const int nB0 = kPoints - (pB & 3);                              // points k < nB0 of half B lie on level lvB
    constexpr int lr4 = LR * kPoints;
    static_assert(LR >= 2 && LR < LT, "levels 0 and 1 always come through the buffer path");
    const float iwA = 1.0f / (float)cA.w, ihA = 1.0f / (float)cA.h;
    const float iwB0 = 1.0f / (float)cB0.w, ihB0 = 1.0f / (float)cB0.h, iwB1 = 1.0f / (float)cB1.w, ihB1 = 1.0f / (float)cB1.h;

    f32x4 *swgt = reinterpret_cast<f32x4 *>(smem + a.stage_base + wave * kStageBytes);
    const unsigned wsel = (unsigned)(sub & 1) * 8u;
    const int runs = a.runs;

    for (int j = team; j < Jx; j += teams) {
        const int p = (a.plane_major ? xcd * Jx + j : xcd + 8 * j), b = p >> 3, m = p & 7;
        const uint16_t *plane = a.value + ((size_t)b * kHeads + m) * (size_t)a.S * kHeadDim;
        __syncthreads();                                             // the previous plane's readers are done
        if (tid < 4) *reinterpret_cast<u32x4 *>(smem + tid * 16) = u32x4{0u, 0u, 0u, 0u};
        {
            const u32x4 *src = reinterpret_cast<const u32x4 *>(plane + (size_t)a.start_lr * kHeadDim);
            const int n16 = a.res_bytes >> 4;
#pragma unroll 4
            for (int i = tid; i < n16; i += kRW * kWave) *reinterpret_cast<u32x4 *>(smem + kResBase + i * 16) = src[i];
        }
        __syncthreads();
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(plane), 0, (unsigned)a.S * IO::kHeadBytes, 0x00020000);

        // the lane's query of run r.  Tiled (encoder): 16 consecutive pixels of a row sample a strip of (16 + spread) x (1 + spread)
        // pixels of their own level, a 4 x 4 tile (4 + spread)^2 -- a third fewer distinct rows per gather instruction, and the 8-12
        // tiles a workgroup's waves hold at a time form one patch whose rows largely stay in the CU's L1 (the texture path's time
        // per instruction grows with the lines it misses).  The order of the queries changes nothing in any query's result.
        auto query_of = [&](int r, bool &ok) -> int {
            if (!a.tiled) {
                const int q = r * kRQ + qs;
                ok = q < a.Nq;
                return q;
            }
            int tp = 0, tw = a.tw[0], w = a.w[0], h = a.h[0], st = a.start[0];
            float itw = a.inv_tw[0];
#pragma unroll
            for (int l = 1; l < LT; ++l) {
                const bool ge = r >= a.tpre[l];
                tp = ge ? a.tpre[l] : tp; tw = ge ? a.tw[l] : tw; w = ge ? a.w[l] : w; h = ge ? a.h[l] : h; st = ge ? a.start[l] : st;
                itw = ge ? a.inv_tw[l] : itw;
            }
            const int u = r - tp;
            const int ty = (int)(((float)u + 0.5f) * itw), tx = u - ty * tw;      // exact: u < 2^21
            const int y = 4 * ty + (qs >> 2), x = 4 * tx + (qs & 3);
            ok = y < h && x < w;
            return st + y * w + x;
        };
        // the lane's share of a run's query-side inputs: locations / weights, or raw offsets / logits (as fp32) + reference points
        struct Inputs {
            float lA[4], lB[2 * kB], aA[2], aB[kB];         // operator form: locations (x, y) and weights
            unsigned oA[2], oB[kB], gA, gB[(kB + 1) / 2];    // fused form: raw offsets and logits, two bf16 per register
            f32x2 rA, rB0, rB1;                              // fused form: reference points (x, y) of the lane's levels
        };
        auto load_inputs = [&](int r, Inputs &in) {
            bool q_ok;
            const int q = query_of(r, q_ok);
            const size_t row = (size_t)b * a.Nq + (q_ok ? q : 0);
            const size_t hrow = (row * kHeads + m) * (size_t)LP;
            if constexpr (FUSED) {
                const uint16_t *off_q = static_cast<const uint16_t *>(a.src_a) + (a.ld_a ? row * (size_t)a.ld_a + (size_t)m * LP * 2 : hrow * 2);
                const uint16_t *lg_q = static_cast<const uint16_t *>(a.src_b) + (a.ld_b ? row * (size_t)a.ld_b + (size_t)m * LP : hrow);
                __builtin_memcpy(in.oA, __builtin_assume_aligned(off_q + 2 * pA, (LT == 4 ? 8 : 4)), 8);
                __builtin_memcpy(in.oB, __builtin_assume_aligned(off_q + 2 * pB, (LT == 4 ? 8 : 4)), 4 * kB);
                __builtin_memcpy(&in.gA, __builtin_assume_aligned(lg_q + pA, (LT == 4 ? 4 : 2)), 4);
                in.gB[(kB + 1) / 2 - 1] = 0u;
                __builtin_memcpy(in.gB, __builtin_assume_aligned(lg_q + pB, (LT == 4 ? 4 : 2)), 2 * kB);
                in.rA = *reinterpret_cast<const f32x2 *>(a.ref + (row * LT + lvA) * 2);
                in.rB0 = *reinterpret_cast<const f32x2 *>(a.ref + (row * LT + lvB) * 2);
                in.rB1 = f32x2{0.f, 0.f};
                if constexpr (LT == 5) in.rB1 = *reinterpret_cast<const f32x2 *>(a.ref + (row * LT + lvB1) * 2);
            } else {
                const float *loc_q = static_cast<const float *>(a.src_a) + hrow * 2;
                const float *att_q = static_cast<const float *>(a.src_b) + hrow;
                load_f32<4, (LT == 4 ? 16 : 8)>(loc_q + 2 * pA, in.lA);
                load_f32<2 * kB, (LT == 4 ? 16 : 8)>(loc_q + 2 * pB, in.lB);
                load_f32<2, (LT == 4 ? 8 : 4)>(att_q + pA, in.aA);
                load_f32<kB, (LT == 4 ? 8 : 4)>(att_q + pB, in.aB);
            }
        };
        const int step = G * kRW;
        int r = g * kRW + wave;
        Inputs cur;
        if (r < runs) load_inputs(r, cur);
        for (; r < runs; r += step) {
            bool qok;
            const int q = query_of(r, qok);
            const size_t row = (size_t)b * a.Nq + (qok ? q : 0);

            // ---- inputs of the lane's 2 + kB points (loaded by the previous run, after its buffer-path gathers were consumed)
            float atA[2], atB[kB];
            f32x2 xyA[2], xyB[kB];
            if constexpr (FUSED) {
                // raw offsets / logits (bf16) and reference points: softmax over all L*P logits of the (query, head), then
                // loc = ref + off / (W_l, H_l)   (ms_deform_attn.py:326-336: 2-d reference points, the encoder's; the 4-d form runs on
                // the query-run kernel)
                auto lo16 = [](unsigned u) { return __builtin_bit_cast(float, u << 16); };
                auto hi16 = [](unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); };
                cur.aA[0] = lo16(cur.gA); cur.aA[1] = hi16(cur.gA);
#pragma unroll
                for (int k = 0; k < kB; ++k) cur.aB[k] = (k & 1) ? hi16(cur.gB[k / 2]) : lo16(cur.gB[k / 2]);
#pragma unroll
                for (int k = 0; k < 2; ++k) { cur.lA[2 * k] = lo16(cur.oA[k]); cur.lA[2 * k + 1] = hi16(cur.oA[k]); }
#pragma unroll
                for (int k = 0; k < kB; ++k) { cur.lB[2 * k] = lo16(cur.oB[k]); cur.lB[2 * k + 1] = hi16(cur.oB[k]); }
                float mx = fmaxf(cur.aA[0], cur.aA[1]);
#pragma unroll
                for (int k = 0; k < kB; ++k) mx = fmaxf(mx, cur.aB[k]);
                mx = group_max<4>(mx);
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    atA[k] = __builtin_amdgcn_exp2f((cur.aA[k] - mx) * 1.44269504088896341f);
                    sum += atA[k];
                }
#pragma unroll
                for (int k = 0; k < kB; ++k) {
                    atB[k] = __builtin_amdgcn_exp2f((cur.aB[k] - mx) * 1.44269504088896341f);
                    sum += atB[k];
                }
                sum = group_sum<4>(sum);
                const float inv_sum = 1.0f / sum;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    atA[k] *= inv_sum;
                    xyA[k] = f32x2{cur.rA.x + cur.lA[2 * k] * iwA, cur.rA.y + cur.lA[2 * k + 1] * ihA};
                }
#pragma unroll
                for (int k = 0; k < kB; ++k) {
                    const bool second = LT == 5 && k >= nB0;
                    const f32x2 rc = second ? cur.rB1 : cur.rB0;
                    atB[k] *= inv_sum;
                    xyB[k] = f32x2{rc.x + cur.lB[2 * k] * (second ? iwB1 : iwB0), rc.y + cur.lB[2 * k + 1] * (second ? ihB1 : ihB0)};
                }
            } else {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    xyA[k] = f32x2{cur.lA[2 * k], cur.lA[2 * k + 1]};
                    atA[k] = cur.aA[k];
                }
#pragma unroll
                for (int k = 0; k < kB; ++k) {
                    xyB[k] = f32x2{cur.lB[2 * k], cur.lB[2 * k + 1]};
                    atB[k] = cur.aB[k];
                }
            }

            // corner offsets (bytes in the plane, or LDS addresses in the resident copy) of one point -> returned (they stay in the
            // lane's registers); its split weights -> the wave's staging slot of that point
            auto setup_point = [&](const f32x2 xy, const float at, const LaneLevel &c, int pt) -> u32x4 {
                if constexpr ((DBG & 4) != 0) {
                    swgt[pt * kRQ + qs] = f32x4{xy.x, xy.y, at, at};
                    return u32x4{c.inv, c.inv, c.inv, c.inv};
                }
                const int h = c.h, w = c.w;
                const float x = xy.x * (float)w - 0.5f;
                const float y = xy.y * (float)h - 0.5f;
                const bool inside = qok && (y > -1.f) && (x > -1.f) && (y < (float)h) && (x < (float)w);   // false for NaN
                const float xf = floorf(x), yf = floorf(y);
                const int x0 = inside ? (int)xf : 0, y0 = inside ? (int)yf : 0;
                const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
                const bool okx0 = inside && x0 >= 0, okx1 = inside && x0 + 1 <= w - 1;
                const bool oky0 = y0 >= 0, oky1 = y0 + 1 <= h - 1;
                const unsigned base = (unsigned)(c.st + y0 * w + x0) * IO::kHeadBytes;
                const unsigned rowb = (unsigned)w * IO::kHeadBytes;
                u32x4 o;
                o.x = (okx0 && oky0) ? base : c.inv;
                o.y = (okx1 && oky0) ? base + IO::kHeadBytes : c.inv;
                o.z = (okx0 && oky1) ? base + rowb : c.inv;
                o.w = (okx1 && oky1) ? base + rowb + IO::kHeadBytes : c.inv;
                const float w00 = inside ? hy * hx * at : 0.f, w01 = inside ? hy * lx * at : 0.f;
                const float w10 = inside ? ly * hx * at : 0.f, w11 = inside ? ly * lx * at : 0.f;
                unsigned h01, l01, h23, l23;                 // {hi(00,01), hi(10,11), lo(00,01), lo(10,11)}: A rows 0 and 1 of mfma_point
                split2_bf16(w00, w01, h01, l01);
                split2_bf16(w10, w11, h23, l23);
                swgt[pt * kRQ + qs] = __builtin_bit_cast(f32x4, u32x4{h01, h23, l01, l23});
                return o;
            };
            u32x4 oA[2], oB[kB];
#pragma unroll
            for (int k = 0; k < 2; ++k) oA[k] = setup_point(xyA[k], atA[k], cA, pA + k);
#pragma unroll
            for (int k = 0; k < kB; ++k) oB[k] = setup_point(xyB[k], atB[k], (LT == 5 && k >= nB0) ? cB1 : cB0, pB + k);
            // the weights are private to the wave and LDS operations of one wave complete in order: a wave-level fence, no s_barrier
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            f32x4 accm[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) accm[c] = f32x4{0.f, 0.f, 0.f, 0.f};

            // Point PT (compile time) of the 16 queries: its four corner addresses = the owner lane's offsets broadcast over the
            // query's 4 lanes inside the address add (DPP quad_perm: no LDS round trip, no extra instruction) + the lane's 16 bytes
            auto corner_addr = [&](auto PT) -> u32x4 {
                constexpr int pt = decltype(PT)::value;
                constexpr bool in_a = pt < kFirstB;
                constexpr int own = in_a ? pt / 2 : (pt - kFirstB) / kB, k = in_a ? pt % 2 : (pt - kFirstB) % kB;
                constexpr int ctrl = own * 0x55;             // quad_perm:[own, own, own, own]
                const u32x4 o = in_a ? oA[k] : oB[k];
                return u32x4{(unsigned)__builtin_amdgcn_update_dpp(0, (int)o.x, ctrl, 0xf, 0xf, true) + lane_off,
                             (unsigned)__builtin_amdgcn_update_dpp(0, (int)o.y, ctrl, 0xf, 0xf, true) + lane_off,
                             (unsigned)__builtin_amdgcn_update_dpp(0, (int)o.z, ctrl, 0xf, 0xf, true) + lane_off,
                             (unsigned)__builtin_amdgcn_update_dpp(0, (int)o.w, ctrl, 0xf, 0xf, true) + lane_off};
            };
            auto weights_of = [&](int pt) {
                return *reinterpret_cast<const u32x2 *>(reinterpret_cast<const unsigned char *>(swgt + pt * kRQ + qs) + wsel);
            };
            struct Rows { u32x4 r00, r01, r10, r11; };
            auto from_plane = [&](const u32x4 ad) {
                if constexpr ((DBG & 1) != 0) return Rows{*reinterpret_cast<const u32x4 *>(smem + lane_off), *reinterpret_cast<const u32x4 *>(smem + lane_off + (ad.x & 0u)),
                                                        *reinterpret_cast<const u32x4 *>(smem + lane_off + (ad.y & 0u)), *reinterpret_cast<const u32x4 *>(smem + lane_off + (ad.z & 0u))};
                return Rows{__builtin_amdgcn_raw_buffer_load_b128(rsrc, ad.x, 0, 0), __builtin_amdgcn_raw_buffer_load_b128(rsrc, ad.y, 0, 0),
                            __builtin_amdgcn_raw_buffer_load_b128(rsrc, ad.z, 0, 0), __builtin_amdgcn_raw_buffer_load_b128(rsrc, ad.w, 0, 0)};
            };
            auto from_lds = [&](const u32x4 ad) {
                if constexpr ((DBG & 2) != 0) return Rows{*reinterpret_cast<const u32x4 *>(smem + lane_off), *reinterpret_cast<const u32x4 *>(smem + lane_off + (ad.x & 0u)),
                                                        *reinterpret_cast<const u32x4 *>(smem + lane_off + (ad.y & 0u)), *reinterpret_cast<const u32x4 *>(smem + lane_off + (ad.z & 0u))};
                return Rows{*reinterpret_cast<const u32x4 *>(smem + ad.x), *reinterpret_cast<const u32x4 *>(smem + ad.y),
                            *reinterpret_cast<const u32x4 *>(smem + ad.z), *reinterpret_cast<const u32x4 *>(smem + ad.w)};
            };
            // Eight steps, one point of levels 0 / 1 (the plane, through the buffer descriptor) each.  Step i: ask for the corner rows of
            // point i + 1, work off this step's share of the coarse points (resident: LDS reads; else the plane), then take in point
            // i's rows, which have been travelling since the step before.  The accumulation order is therefore 8, 0, 9, 1, ... --
            // not the query-run kernel's 0 .. 15: results agree to fp32 re-association (the last bit of a bf16 output now and then).
            constexpr int kCoarse = LP - kFirstB;            // 8 or 12
            Inputs nxt;
            auto consume = [&](const Rows &rw, const u32x2 wq) {
                if constexpr ((DBG & 128) != 0) {            // the matrix-core steps without the v_perm re-pairing (upper bound of a transposing read)
                    const s16x4 am = __builtin_bit_cast(s16x4, wq);
                    const unsigned t0[4] = {rw.r00.x, rw.r00.y, rw.r00.z, rw.r00.w}, t1[4] = {rw.r01.x, rw.r01.y, rw.r01.z, rw.r01.w};
                    const unsigned b0[4] = {rw.r10.x, rw.r10.y, rw.r10.z, rw.r10.w}, b1[4] = {rw.r11.x, rw.r11.y, rw.r11.z, rw.r11.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        accm[2 * j] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(am, __builtin_bit_cast(s16x4, u32x2{t0[j], b0[j]}), accm[2 * j], 0, 0, 0);
                        accm[2 * j + 1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(am, __builtin_bit_cast(s16x4, u32x2{t1[j], b1[j]}), accm[2 * j + 1], 0, 0, 0);
                    }
                } else if constexpr ((DBG & 8) != 0) {
                    accm[0].x += __builtin_bit_cast(float, rw.r00.x ^ rw.r01.y ^ rw.r10.z ^ rw.r11.w ^ wq.x);
                } else {
                    mfma_point(rw.r00, rw.r01, rw.r10, rw.r11, wq, accm);
                }
            };
            auto consume_noperm = [&](const Rows &rw, const u32x2 wq) {
                const s16x4 am = __builtin_bit_cast(s16x4, wq);
                const unsigned t0[4] = {rw.r00.x, rw.r00.y, rw.r00.z, rw.r00.w}, t1[4] = {rw.r01.x, rw.r01.y, rw.r01.z, rw.r01.w};
                const unsigned b0[4] = {rw.r10.x, rw.r10.y, rw.r10.z, rw.r10.w}, b1[4] = {rw.r11.x, rw.r11.y, rw.r11.z, rw.r11.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    accm[2 * j] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(am, __builtin_bit_cast(s16x4, u32x2{t0[j], b0[j]}), accm[2 * j], 0, 0, 0);
                    accm[2 * j + 1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(am, __builtin_bit_cast(s16x4, u32x2{t1[j], b1[j]}), accm[2 * j + 1], 0, 0, 0);
                }
            };
            auto coarse = [&](auto PT) {
                constexpr int pt = decltype(PT)::value;
                const u32x4 ad = corner_addr(PT);
                const u32x2 wq = weights_of(pt);
                if constexpr (pt >= lr4) {
                    const Rows b = from_lds(ad);
                    if constexpr ((DBG & 256) != 0) consume_noperm(b, wq); else consume(b, wq);
                } else {
                    const Rows b = from_plane(ad);
                    consume(b, wq);
                }
            };
            // kAhead = how many steps ahead a fine point's rows are asked for; kLoose: no fences inside a step (two waves per SIMD have
            // 256 registers each: let the compiler overlap the coarse points' LDS round trips)
            constexpr int kAhead = RW == 8 ? ((DBG & 16) ? 3 : (DBG & 32) ? 4 : (DBG & 64) ? 1 : 2) : 1;
            constexpr bool kLoose = RW == 8;
            if constexpr (RW == 8 && LT == 4 && LR == 2 && (DBG & 512) != 0) {
                // explicit pipeline (two waves per SIMD, 256 registers): all weights up front, fine rows two steps ahead, the coarse
                // rows (LDS) one step ahead -- no instruction waits on a round trip it has just started
                u32x2 wq[LP];
#pragma unroll
                for (int pt = 0; pt < LP; ++pt) wq[pt] = weights_of(pt);
                Rows fine[kFirstB], cz[kCoarse];
                fine[0] = from_plane(corner_addr(std::integral_constant<int, 0>{}));
                fine[1] = from_plane(corner_addr(std::integral_constant<int, 1>{}));
                cz[0] = from_lds(corner_addr(std::integral_constant<int, kFirstB>{}));
                auto pstep = [&](auto I) {
                    constexpr int i = decltype(I)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (i + 2 < kFirstB) fine[i + 2] = from_plane(corner_addr(std::integral_constant<int, i + 2>{}));
                    if constexpr (i + 3 == kFirstB) {
                        if (r + step < runs) load_inputs(r + step, nxt);
                    }
                    if constexpr (i + 1 < kCoarse) cz[i + 1] = from_lds(corner_addr(std::integral_constant<int, kFirstB + i + 1>{}));
                    __builtin_amdgcn_sched_barrier(0);
                    consume(cz[i], wq[kFirstB + i]);
                    consume(fine[i], wq[i]);
                };
                pstep(std::integral_constant<int, 0>{});
                pstep(std::integral_constant<int, 1>{});
                pstep(std::integral_constant<int, 2>{});
                pstep(std::integral_constant<int, 3>{});
                pstep(std::integral_constant<int, 4>{});
                pstep(std::integral_constant<int, 5>{});
                pstep(std::integral_constant<int, 6>{});
                pstep(std::integral_constant<int, 7>{});
            } else {
            Rows fine[kFirstB];
            fine[0] = from_plane(corner_addr(std::integral_constant<int, 0>{}));
            if constexpr (kAhead >= 2) fine[1] = from_plane(corner_addr(std::integral_constant<int, 1>{}));
            if constexpr (kAhead >= 3) fine[2] = from_plane(corner_addr(std::integral_constant<int, 2>{}));
            if constexpr (kAhead >= 4) fine[3] = from_plane(corner_addr(std::integral_constant<int, 3>{}));
            auto step_fn = [&](auto I) {
                constexpr int i = decltype(I)::value;
                __builtin_amdgcn_sched_barrier(0);           // a step's addresses are formed in the step (registers)
                if constexpr (i + kAhead < kFirstB) fine[i + kAhead] = from_plane(corner_addr(std::integral_constant<int, i + kAhead>{}));
                if constexpr (i + kAhead + 1 == kFirstB) {
                    // the next run's inputs: behind this run's last plane rows (loads return in order: asked for earlier they would
                    // hold up every step), in flight over the last steps, the store and the loop top
                    if (r + step < runs) load_inputs(r + step, nxt);
                }
                constexpr int c0 = kFirstB + (i * kCoarse) / kFirstB, c1 = kFirstB + ((i + 1) * kCoarse) / kFirstB;
                if constexpr (!kLoose) __builtin_amdgcn_sched_barrier(0);
                coarse(std::integral_constant<int, c0>{});
                if constexpr (c1 - c0 == 2) {
                    if constexpr (!kLoose) __builtin_amdgcn_sched_barrier(0);       // one coarse point at a time: 16 registers of rows
                    coarse(std::integral_constant<int, c0 + 1>{});
                }
                if constexpr (!kLoose) __builtin_amdgcn_sched_barrier(0);
                consume(fine[i], weights_of(i));
            };
            step_fn(std::integral_constant<int, 0>{});
            step_fn(std::integral_constant<int, 1>{});
            step_fn(std::integral_constant<int, 2>{});
            step_fn(std::integral_constant<int, 3>{});
            step_fn(std::integral_constant<int, 4>{});
            step_fn(std::integral_constant<int, 5>{});
            step_fn(std::integral_constant<int, 6>{});
            step_fn(std::integral_constant<int, 7>{});
            }
            static_assert(kFirstB == 8, "eight steps");

            float res[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) res[c] = accm[c].x + accm[c].y;
            if (qok) IO::store_run(a.out + row * (kHeads * kHeadDim) + m * kHeadDim + sub * 8, res);
            cur = nxt;
        }
    }
}

}  // namespace

// development A/B: waves per workgroup (make dev: rdetr_dev_set_res_waves)
struct ResVariant {
#ifdef RDETR_DEV
    static inline int waves = 12;
    static inline int dbg = 0;
    static inline bool tiled = true;
    static inline bool plane_major = true;
    static inline int max_teams = 0;                 // 0 = the product's choice
#else
    static constexpr int max_teams = 0;
    static constexpr bool tiled = true;
    static constexpr bool plane_major = true;
    static constexpr int waves = 12;
#endif
};

// Host side.  Returns RDETR_ERR_UNSUPPORTED for everything the kernel does not cover (the caller then runs the query-run kernel).
template <bool FUSED>
static int msda_res_forward(const uint16_t *value, const int64_t *shapes, const int64_t *level_start, const void *src_a,
                            const void *src_b, const float *ref, int ref_dim, int B, int S, int L, int Nq, int ld_a, int ld_b,
                            uint16_t *out, hipStream_t stream)
{
    if (L != 4 && L != 5) return RDETR_ERR_UNSUPPORTED;
    if (FUSED && ref_dim != 2) return RDETR_ERR_UNSUPPORTED;           // 4-d reference points (decoder): the query-run kernel
    if ((long long)S * 64 >= (1ll << 31)) return RDETR_ERR_UNSUPPORTED;
    ResArgs a{};
    long long at = 0;
    for (int l = 0; l < L; ++l) {                                        // the levels must tile [0, S): the resident copy is plane[start_lr .. S)
        const long long h = shapes[2 * l], w = shapes[2 * l + 1];
        if (h <= 0 || w <= 0 || h > 32768 || w > 32768 || level_start[l] != at) return RDETR_ERR_UNSUPPORTED;
        a.h[l] = (int)h; a.w[l] = (int)w; a.start[l] = (int)at;
        at += h * w;
    }
    if (at != S) return RDETR_ERR_UNSUPPORTED;
    // alignment of the vector loads (msda_fwd.hip: vec_ok / vec5_ok)
    const auto al = [](const void *p, uintptr_t n) { return reinterpret_cast<uintptr_t>(p) % n == 0; };
    if (!al(value, 16) || !al(out, 16)) return RDETR_ERR_UNSUPPORTED;
    if (L == 4) {
        if (!al(src_a, 16) || !al(src_b, 16)) return RDETR_ERR_UNSUPPORTED;
        if (FUSED && (!al(ref, 16) || (ld_a * 2) % 16 != 0 || (ld_b * 2) % 16 != 0)) return RDETR_ERR_UNSUPPORTED;
    } else {
        if (!al(src_a, 8) || !al(src_b, 4)) return RDETR_ERR_UNSUPPORTED;
        if (FUSED && (!al(ref, 16) || (ld_a * 2) % 4 != 0)) return RDETR_ERR_UNSUPPORTED;
    }
    const int RW = ResVariant::waves;
    const int stage = L * kPoints * kRQ * 16 * RW;                               // weights of all points, per wave
    const int budget = kLdsBytes - (int)kResBase - stage - 128;
    int lr = L;
    while (lr > L - 2 && ((long long)S - a.start[lr - 1]) * 64 <= budget) --lr;  // one or two resident levels (the kernel's LR)
    if (lr == L) return RDETR_ERR_UNSUPPORTED;                           // not even the coarsest level fits
    a.value = value; a.src_a = src_a; a.src_b = src_b; a.ref = ref; a.out = out;
    a.S = S; a.Nq = Nq; a.B = B; a.ld_a = ld_a; a.ld_b = ld_b;
    a.lr = lr; a.start_lr = a.start[lr]; a.res_bytes = (S - a.start[lr]) * 64;
    a.stage_base = (int)((kResBase + a.res_bytes + 127) / 128 * 128);
    a.plane_major = ResVariant::plane_major ? 1 : 0;
    // operator form: two planes of an XCD in flight (a head PAIR: the weights' 128-byte lines are shared) -- 2.56 M -> 1.76 M L2 -> fabric
    // read requests at B = 4 against all four (225 MB for 183 MB of distinct bytes), same time; fused form: all of them (its raw
    // logits share a line between FOUR heads; 94-97 us against 97-103) -- profiles/r04/ab_res_planes_in_flight.txt
    a.max_teams = ResVariant::max_teams > 0 ? ResVariant::max_teams : (FUSED ? (1 << 20) : 2);
    a.tiled = (Nq == S && ResVariant::tiled) ? 1 : 0;
    a.runs = (Nq + kRQ - 1) / kRQ;
    if (a.tiled) {
        long long t = 0;
        for (int l = 0; l < L; ++l) {
            a.tpre[l] = (int)t;
            a.tw[l] = (a.w[l] + 3) / 4;
            a.inv_tw[l] = 1.0f / (float)a.tw[l];
            t += (long long)a.tw[l] * ((a.h[l] + 3) / 4);
        }
        if (t >= (1ll << 21)) return RDETR_ERR_UNSUPPORTED;
        a.runs = (int)t;
    }
    const int lds = a.stage_base + stage;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return RDETR_ERR_LAUNCH;
        cus = prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 * 8 : 256;
    }
    const dim3 grid((unsigned)cus);
    auto launch = [&](auto kern, int waves) -> int {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, grid, dim3(waves * kWave), (size_t)lds, stream, a);
        return launch_status();
    };
    const bool two = lr == L - 2;
#ifdef RDETR_DEV
    if (ResVariant::dbg && L == 4 && !FUSED && two) {
        if constexpr (!FUSED) {
            switch (ResVariant::dbg) {
            case 512: return launch(msda_fwd_res_kernel<4, false, 8, 2, 512>, 8);
            case 128: return launch(msda_fwd_res_kernel<4, false, 12, 2, 128>, 12);
            case 256: return launch(msda_fwd_res_kernel<4, false, 12, 2, 256>, 12);
            case 16: return launch(msda_fwd_res_kernel<4, false, 8, 2, 16>, 8);
            case 32: return launch(msda_fwd_res_kernel<4, false, 8, 2, 32>, 8);
            case 64: return launch(msda_fwd_res_kernel<4, false, 8, 2, 64>, 8);
            case 1: return launch(msda_fwd_res_kernel<4, false, 12, 2, 1>, 12);
            case 2: return launch(msda_fwd_res_kernel<4, false, 12, 2, 2>, 12);
            case 3: return launch(msda_fwd_res_kernel<4, false, 12, 2, 3>, 12);
            case 4: return launch(msda_fwd_res_kernel<4, false, 12, 2, 4>, 12);
            case 7: return launch(msda_fwd_res_kernel<4, false, 12, 2, 7>, 12);
            case 8: return launch(msda_fwd_res_kernel<4, false, 12, 2, 8>, 12);
            case 9: return launch(msda_fwd_res_kernel<4, false, 12, 2, 9>, 12);
            case 11: return launch(msda_fwd_res_kernel<4, false, 12, 2, 11>, 12);
            case 12: return launch(msda_fwd_res_kernel<4, false, 12, 2, 12>, 12);
            case 15: return launch(msda_fwd_res_kernel<4, false, 12, 2, 15>, 12);
            default: break;
            }
        }
    }
    if (RW == 16) {
        if (L == 4) return two ? launch(msda_fwd_res_kernel<4, FUSED, 16, 2>, 16) : launch(msda_fwd_res_kernel<4, FUSED, 16, 3>, 16);
        return two ? launch(msda_fwd_res_kernel<5, FUSED, 16, 3>, 16) : launch(msda_fwd_res_kernel<5, FUSED, 16, 4>, 16);
    }
    if (RW == 8) {
        if (L == 4) return two ? launch(msda_fwd_res_kernel<4, FUSED, 8, 2>, 8) : launch(msda_fwd_res_kernel<4, FUSED, 8, 3>, 8);
        return two ? launch(msda_fwd_res_kernel<5, FUSED, 8, 3>, 8) : launch(msda_fwd_res_kernel<5, FUSED, 8, 4>, 8);
    }
#endif
    // 12 waves = 3 per SIMD = 168 registers each: the rounds need ~160 (16 waves: spills, 170 us; 8 waves: the same time as 12)
    if (L == 4) return two ? launch(msda_fwd_res_kernel<4, FUSED, 12, 2>, 12) : launch(msda_fwd_res_kernel<4, FUSED, 12, 3>, 12);
    return two ? launch(msda_fwd_res_kernel<5, FUSED, 12, 3>, 12) : launch(msda_fwd_res_kernel<5, FUSED, 12, 4>, 12);
}

}  // namespace rdetr

static int res_common_checks(const void *value, const int64_t *hs, const int64_t *hl, const void *a, const void *b, const void *out, int B,
                             int S, int H, int D, int L, int Nq, int P)
{
    if (B < 0 || S < 0 || Nq < 0 || H <= 0 || D <= 0 || L <= 0 || P <= 0) return RDETR_ERR_INVALID_ARG;
    if (B == 0 || Nq == 0) return RDETR_OK;
    if (!value || !hs || !hl || !a || !b || !out || S == 0) return RDETR_ERR_INVALID_ARG;
    if (H != rdetr::kHeads || D != rdetr::kHeadDim || P != rdetr::kPoints) return RDETR_ERR_UNSUPPORTED;
    return 1;                                                            // go on
}

extern "C" int rdetr_msda_forward_resident_bf16(const uint16_t *value, const int64_t *host_spatial_shapes,
                                                const int64_t *host_level_start_index, const float *sampling_loc,
                                                const float *attn_weight, int B, int S, int H, int D, int L, int Nq, int P,
                                                uint16_t *out, void *stream)
{
    const int st = res_common_checks(value, host_spatial_shapes, host_level_start_index, sampling_loc, attn_weight, out, B, S, H, D, L, Nq, P);
    if (st != 1) return st;
    return rdetr::msda_res_forward<false>(value, host_spatial_shapes, host_level_start_index, sampling_loc, attn_weight, nullptr, 0, B, S,
                                          L, Nq, 0, 0, out, static_cast<hipStream_t>(stream));
}

extern "C" int rdetr_msda_forward_fused_resident_bf16(const uint16_t *value, const int64_t *host_spatial_shapes,
                                                      const int64_t *host_level_start_index, const uint16_t *sampling_offsets,
                                                      int ld_offsets, const uint16_t *attn_logits, int ld_logits,
                                                      const float *reference_points, int ref_dim, int B, int S, int H, int D, int L,
                                                      int Nq, int P, uint16_t *out, void *stream)
{
    const int st = res_common_checks(value, host_spatial_shapes, host_level_start_index, sampling_offsets, attn_logits, out, B, S, H, D, L, Nq, P);
    if (st != 1) return st;
    if (!reference_points || (ref_dim != 2 && ref_dim != 4)) return RDETR_ERR_INVALID_ARG;
    if (ld_offsets < 0 || ld_logits < 0 || (ld_offsets && ld_offsets < H * L * P * 2) || (ld_logits && ld_logits < H * L * P) ||
        ld_offsets % 2 != 0)
        return RDETR_ERR_INVALID_ARG;
    return rdetr::msda_res_forward<true>(value, host_spatial_shapes, host_level_start_index, sampling_offsets, attn_logits,
                                         reference_points, ref_dim, B, S, L, Nq, ld_offsets, ld_logits, out,
                                         static_cast<hipStream_t>(stream));
}

#ifdef RDETR_DEV
extern "C" void rdetr_dev_set_res_max_teams(int v) { rdetr::ResVariant::max_teams = v; }           // 0 = the product's choice
extern "C" void rdetr_dev_set_res_plane_major(int v) { rdetr::ResVariant::plane_major = v != 0; }
extern "C" void rdetr_dev_set_res_tiled(int v) { rdetr::ResVariant::tiled = v != 0; }
extern "C" void rdetr_dev_set_res_dbg(int v) { rdetr::ResVariant::dbg = v; }
extern "C" void rdetr_dev_set_res_waves(int v) { rdetr::ResVariant::waves = (v == 8 || v == 16) ? v : 12; }
#endif
