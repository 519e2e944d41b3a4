// Multi-scale deformable attention, forward -- SWEEP kernel for the ENCODER shape (queries = the pyramid's own pixels,
// Nq == S, L == 4, bf16 value, materialised sampling locations / attention weights = the reference operator's inputs,
// ms_deform_attn_cuda.cu:12-72) on gfx950 (MI355X).  Round 4.
//
// What the earlier LDS-sourced kernels of this repository taught (csrc/msda_win.hip, csrc/msda_tile.hip): the data path --
// corner rows gathered from LDS windows by ds_read_b64_tr_b16 straight into v_mfma_f32_16x16x32_bf16 -- runs at the LDS rate
// (~45 us per launch at BASELINE.json configs[1]), but a window per (tile, level) costs a fill, a barrier pair and a latency
// that nothing hides when LDS holds only two such windows per CU, and every tile pays its geometry again.  This kernel slides:
//
//   workgroup  = 512 threads (8 waves), ONE per CU, persistent over a contiguous range of STEPS.  A step = 8 columns x TH rows
//                (TH <= 6, one BAND of level-0 rows) of one (image, head) plane plus the pixels of the coarser levels whose
//                centres fall into it: one OCTET (8 queries x 16 samples) per wave -- waves 0..TH-1 a level-0 row each, the
//                remaining waves the coarser pixels.
//   rings      the windows of ALL FOUR levels are resident at once, as ring buffers of COLUMNS (column-major: a column =
//                consecutive rows, 64 B each): level l keeps columns [cl(s) - 8, cl(s + 1) + 7] for step s, cl(s) =
//                floor(8 s W_l / W_0), rows [floor(Y0 H_l / H_0) - 8, floor((Y1 - 1) H_l / H_0) + 8] of the band.  Going from
//                step s to s + 1 costs only the NEW columns (8 + 4 + 2 + 1 of them at a /2 pyramid): two LDS-DMA
//                instructions per column (16 rows each, range checked: what lies outside the level arrives as zeros, so the
//                zero padding of ms_deform_im2col_cuda.cuh:44-67 is in the data), issued a whole step ahead into the ring
//                slots the previous step vacated.  One barrier per step.
//   set-up     lane = (query, level, point pair): 16-byte loads of locations, 8-byte loads of weights, issued TWO steps
//                ahead; per sample the pixel coordinates, the four corner addresses in the ring (no corner is a constant
//                away from another: columns wrap) and the four corner weights split into bf16 high + low parts.
//   gather     per level 4 MFMA steps per wave (K = 8 samples x 4 corners; B = the gathered rows, A = block-diagonal
//                weights with high / low parts in separate rows, fp32 accumulation) from a 1-KiB wave-private staging area.
//   flagged    a sample whose corners are not all inside its ring window gets its four rows fetched by a range-checked
//                LDS-DMA into a patch cell and goes through an extra round of the same MFMA steps after the regular ones
//                (its DMA has landed by then): results never depend on the windows.
// Per corner the arithmetic is msda_fwd.hip's (same weights); the summation order differs and each weight carries a 2^-17
// relative representation error (the bf16 output rounds at 2^-9).
#include <type_traits>

#include "common.h"

namespace rdetr {

typedef __bf16 sw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sw_bf16x2 __attribute__((ext_vector_type(2)));
typedef short sw_s16x4 __attribute__((ext_vector_type(4)));

constexpr int kSwThreads = 1024;
constexpr int kSwWaves = kSwThreads / kWave;                  // 16 = quartets per step
constexpr int kSwQ = 4;                                       // queries per wave and step
constexpr int kSwTW = 8;                                      // level-0 columns per step
constexpr int kSwMaxTH = 6;                                   // level-0 rows per band
constexpr int kSwHeads = 8, kSwHeadDim = 32, kSwPoints = 4, kSwLevels = 4;
constexpr int kSwMargin = 8;
constexpr int kSwFlagCap = 4;                                 // flagged samples a wave keeps in flight per step
// rings: columns (even, so that a wrapped neighbour column keeps the bank phase) and rows per column (== 2 mod 4: the column
// stride is == 128 mod 256 bytes, the four corners of a sample fall on four bank groups)
constexpr int kSwRW0 = 32, kSwRW1 = 24, kSwRW2 = 20, kSwRW3 = 18;
constexpr int kSwCR0 = 22, kSwCR1 = 22, kSwCR2 = 22, kSwCR3 = 18;
__host__ __device__ constexpr int sw_rw(int l) { return l == 0 ? kSwRW0 : l == 1 ? kSwRW1 : l == 2 ? kSwRW2 : kSwRW3; }
__host__ __device__ constexpr int sw_cr(int l) { return l == 0 ? kSwCR0 : l == 1 ? kSwCR1 : l == 2 ? kSwCR2 : kSwCR3; }

// ---- LDS map ------------------------------------------------------------------------------------------------------
constexpr int kSwZeroOff = 0;                                 // 1 KiB of zeros: the zero sample (TL 0, BL 64, TR 128, BR 192) and what
constexpr int kSwZeroKOff = 512 + 32;                         // idle A-operand lanes read
constexpr int kSwWaveOff = 1024;                              // per-wave area:
constexpr int kSwStageW = 0;                                  //   [0, 1024)     W[query 4][part 2][level 4][point 4][corner 4] bf16
constexpr int kSwStageO = 1024;                               //   [1024, 1536)  O[side 2: left, right][query 4][level 4][point 4] u32:
                                                              //                 LDS address of the sample's top corner on that side
constexpr int kSwFgo = 1536;                                  //   [1536, 1792)  flagged samples in flight x {pixel, slot, 4 weights, -, -}
constexpr int kSwWaveBytes = 1792;
constexpr int kSwRing0 = kSwWaveOff + kSwWaves * kSwWaveBytes;                       // 29696
constexpr int kSwRing1 = kSwRing0 + kSwRW0 * kSwCR0 * 64;
constexpr int kSwRing2 = kSwRing1 + kSwRW1 * kSwCR1 * 64;
constexpr int kSwRing3 = kSwRing2 + kSwRW2 * kSwCR2 * 64;
constexpr int kSwLdsBytes = kSwRing3 + kSwRW3 * kSwCR3 * 64;
__host__ __device__ constexpr int sw_ring(int l) { return l == 0 ? kSwRing0 : l == 1 ? kSwRing1 : l == 2 ? kSwRing2 : kSwRing3; }
static_assert(kSwRing0 % 256 == 0 && kSwWaveBytes % 64 == 0, "sample bases are multiples of 64");
static_assert(kSwLdsBytes <= 160 * 1024, "LDS map exceeds 160 KiB");

struct SweepLevels { int h[kSwLevels], w[kSwLevels], start[kSwLevels]; };

// a / b for a < 2^24, 0 < b < 2^24 without the integer division sequence
__device__ __forceinline__ unsigned sw_div(unsigned a, unsigned b)
{
    unsigned q = (unsigned)((float)a * __builtin_amdgcn_rcpf((float)b));
    int r = (int)a - (int)(q * b);
    if (r < 0) { --q; r += (int)b; }
    if (r >= (int)b) { ++q; }
    return q;
}
// cl(s): first column of level (width n) under level-0 column 8 s (level-0 width n0); == n from the last step on
__device__ __forceinline__ int sw_cl(int s, int n, int n0)
{
    const unsigned x0 = (unsigned)(s * kSwTW);
    return x0 >= (unsigned)n0 ? n : (int)sw_div(x0 * (unsigned)n, (unsigned)n0);
}
static long long sweep_cl(long long s, long long n, long long n0) { return s * kSwTW >= n0 ? n : s * kSwTW * n / n0; }
// first pixel coordinate of a level (size n) whose centre lies at level-0 coordinate `c0` or beyond (level-0 size n0):
// the smallest x with (2x + 1) * n0 >= 2 * c0 * n, clipped to n
__device__ __forceinline__ int sw_first(int c0, int n, int n0)
{
    const unsigned v = 2u * (unsigned)c0 * (unsigned)n;                      // < 2^24: checked on the host
    const int x = (int)(sw_div(v + (unsigned)n0 - 1u, (unsigned)n0) >> 1);
    return x < n ? x : n;
}
static long long sweep_first(long long c0, long long n, long long n0)
{
    const long long x = ((2 * c0 * n + n0 - 1) / n0) >> 1;
    return x < n ? x : n;
}

// bf16 high parts (round to nearest even) and low parts of two fp32 weights, packed (a in the low half)
__device__ __forceinline__ void sw_split2(float a, float b, unsigned &hi, unsigned &lo)
{
    hi = __builtin_bit_cast(unsigned, sw_bf16x2{(__bf16)a, (__bf16)b});
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, sw_bf16x2{(__bf16)ra, (__bf16)rb});
}
__device__ __forceinline__ void sw_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// x + (x rotated by N lanes inside its row of 16 lanes)
template <int N> __device__ __forceinline__ float sw_row_ror_add(float x)
{
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, false));
}

// HM = false: value [B,S,H,D] (the reference operator's layout); HM = true: value [B,H,S,D] (head-major).
template <bool HM>
__global__ __launch_bounds__(kSwThreads, 4) void msda_fwd_sweep_kernel(
    const uint16_t *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attn, const SweepLevels lv,
    int S, int nb, int steps_x, int total_steps, int nblk, int dbg, uint16_t *__restrict__ out)
{
    constexpr unsigned kGPixB = HM ? 64u : (unsigned)(kSwHeads * kSwHeadDim * 2);   // global bytes from one pixel to the next
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int Nq = S;
    // the level table as twelve scalars (an array of them indexed inside lambdas ended up in scratch memory, and every scratch
    // reload waits for ALL vector-memory operations in flight)
    const int W0 = lv.w[0], W1 = lv.w[1], W2 = lv.w[2], W3 = lv.w[3];
    const int H0 = lv.h[0], H1 = lv.h[1], H2 = lv.h[2], H3 = lv.h[3];
    const int T0 = lv.start[0], T1 = lv.start[1], T2 = lv.start[2], T3 = lv.start[3];
    auto LW = [&](int l) { return l == 0 ? W0 : l == 1 ? W1 : l == 2 ? W2 : W3; };
    auto LH = [&](int l) { return l == 0 ? H0 : l == 1 ? H1 : l == 2 ? H2 : H3; };
    auto LS = [&](int l) { return l == 0 ? T0 : l == 1 ? T1 : l == 2 ? T2 : T3; };
    // ... and a selection by a RUN-TIME level as masks (a chain of selects over run-time values becomes a look-up table in
    // scratch memory too)
    auto pick = [](int l, int a0, int a1, int a2, int a3) {
        return (a0 & -(int)(l == 0)) | (a1 & -(int)(l == 1)) | (a2 & -(int)(l == 2)) | (a3 & -(int)(l == 3));
    };

    // this workgroup's steps [g0, g1) of the global list (plane-major, then band, then column step)
    const int logical = xcd_contiguous_block(blockIdx.x, nblk);
    const int g0 = (int)((long long)logical * total_steps / nblk), g1 = (int)((long long)(logical + 1) * total_steps / nblk);

    if (tid < 256) reinterpret_cast<unsigned *>(lds + kSwZeroOff)[tid] = 0u;      // published by the first barrier

    // ---- lane roles ---------------------------------------------------------------------------------------------------------
    // set-up: ONE sample per lane -- query qi of the wave's quartet, level sl, point pp
    const int qi = lane >> 4, sl = (lane >> 2) & 3, pp = lane & 3;
    // gather: K-group kg (= query kg), corner tq / piece tp of a transposed read; as an A-operand lane: row am = lane & 15 =
    // 2 * (query ag) + (0 = bf16 high part, 1 = low part), rows 8..15 unused
    const int kg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int am = lane & 15, ag = am >> 1, apart = am & 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;     // 0 in practice
    const unsigned wave_off = (unsigned)(kSwWaveOff + wave * kSwWaveBytes);
    unsigned char *const wreg = lds + wave_off;
    // level constants of the set-up role (level sl)
    const int myW = pick(sl, W0, W1, W2, W3), myH = pick(sl, H0, H1, H2, H3);
    const float myWf = (float)myW, myHf = (float)myH, myWc = (float)(myW + 1), myHc = (float)(myH + 1);
    const unsigned myRW = sl == 0 ? kSwRW0 : sl == 1 ? kSwRW1 : sl == 2 ? kSwRW2 : kSwRW3;
    const int myCR = sl == 3 ? kSwCR3 : kSwCR0;
    static_assert(kSwCR0 == kSwCR1 && kSwCR1 == kSwCR2, "myCR");
    const unsigned myCS = (unsigned)myCR * 64u;
    const unsigned par32 = (unsigned)(qi & 1) * 32u;              // odd queries read the other channel half first: the two
                                                                  // K-groups of a 32-lane half never share a bank group
    const unsigned myRB = lds0 + (unsigned)(sl == 0 ? kSwRing0 : sl == 1 ? kSwRing1 : sl == 2 ? kSwRing2 : kSwRing3) + par32;
    const unsigned o_zero = lds0 + (unsigned)kSwZeroOff + par32;
    // One MFMA step = level l, point pair j: K-group kg carries the two samples (points 2j, 2j + 1 of level l) of query kg;
    // its lane (corner tq, piece tp) reads 8 bytes of that corner's row of each: O[side tq & 1] + 64 (tq >> 1) + 8 tp
    const unsigned cd = (unsigned)tp * 8u + (unsigned)(tq >> 1) * 64u;
    const unsigned o_rd = lds0 + wave_off + (unsigned)kSwStageO + (unsigned)(tq & 1) * 256u + (unsigned)kg * 64u;   // + l * 16
    const unsigned w_rd = (am < 8 && ag == kg) ? lds0 + wave_off + (unsigned)kSwStageW + (unsigned)(ag * 256 + apart * 128)
                                               : lds0 + (unsigned)kSwZeroKOff;                                      // + l * 32 + j * 16
    // staging writes of the set-up role
    unsigned char *const st_o = wreg + kSwStageO + qi * 64 + sl * 16 + pp * 4;    // + side * 256
    unsigned char *const st_w = wreg + kSwStageW + qi * 256 + sl * 32 + pp * 8;   // + part * 128
    // geometry role: lanes 0..3 compute cl_l(s + 3), lanes 4..6 xa_l(s + 4), l = lane - 3
    const int geoW = lane < 4 ? pick(lane, W0, W1, W2, W3) : pick(lane - 3, W0, W1, W2, W3);

    auto lds_b128 = [](unsigned a) { return *(__attribute__((address_space(3))) const u32x4 *)a; };
    auto lds_tr = [](unsigned a) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sw_s16x4 *)a));
    };

    // The inputs of a step -- location and weight of the lane's sample -- are loaded TWO steps ahead by inline assembly into one
    // of three register sets used in rotation: hipcc does not know these loads, so it neither waits for them nor drains the
    // window fills in flight when it meets their first use (tracked loads cost a vmcnt(0) per step here: a trip to HBM).  A set
    // is complete when the step after the one that issued it ends (its counted wait names the set, so no use is scheduled above).
    struct Inputs { f32x2 xy; float a; };
    Inputs r0, r1, r2;
    r0.xy = r1.xy = r2.xy = f32x2{0.f, 0.f};
    r0.a = r1.a = r2.a = 0.f;

    for (int gs = g0; gs < g1;) {
        // ---- segment: consecutive steps of one band ----------------------------------------------------------------------
        const int per_plane = nb * steps_x;
        const int plane_i = (int)sw_div((unsigned)gs, (unsigned)per_plane), rem = gs - plane_i * per_plane;
        const int band = __builtin_amdgcn_readfirstlane((int)sw_div((unsigned)rem, (unsigned)steps_x));
        const int sx0 = __builtin_amdgcn_readfirstlane(rem - band * steps_x);
        const int b = __builtin_amdgcn_readfirstlane(plane_i >> 3), m = __builtin_amdgcn_readfirstlane(plane_i & 7);
        const int left = g1 - gs, room = steps_x - sx0;
        const int sx_end = sx0 + (left < room ? left : room);
        const int Y0 = __builtin_amdgcn_readfirstlane((int)sw_div((unsigned)(band * H0), (unsigned)nb));
        const int Y1 = __builtin_amdgcn_readfirstlane((int)sw_div((unsigned)((band + 1) * H0), (unsigned)nb));
        const int th2 = 2 * (Y1 - Y0);                                // waves [0, th2): level-0 half rows; [th2, 16): coarser pixels

        // the (image, head) value plane behind one wave-uniform buffer descriptor; byte offsets inside it are 32-bit
        const unsigned char *plane = reinterpret_cast<const unsigned char *>(value) +
                                     (HM ? ((size_t)b * kSwHeads + m) * (size_t)S * 64u
                                         : (size_t)b * S * kGPixB + (size_t)m * 64u);
        const unsigned plane_bytes = HM ? (unsigned)S * 64u : (unsigned)S * kGPixB - (unsigned)m * 64u;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(plane), 0, plane_bytes, 0x00020000);
        // wave-uniform bases of this (image, head); the per-lane parts are 32-bit byte offsets (one image's tensors are < 4 GB)
        const float *loc_b = loc + ((size_t)b * Nq * kSwHeads + (size_t)m) * (kSwLevels * kSwPoints * 2);
        const float *att_b = attn + ((size_t)b * Nq * kSwHeads + (size_t)m) * (kSwLevels * kSwPoints);
        uint16_t *out_b = out + (size_t)b * Nq * (kSwHeads * kSwHeadDim) + m * kSwHeadDim;

        // rows of the band's windows, per level: [rlo, rlo + nrows); rows of the coarser pixels inside the band: [ya, ya + ny)
        int RLO[kSwLevels], NR[kSwLevels];
        int YAv = 0, NYv = 0;                           // lane l = 1..3: first row / row count of the coarser pixels inside the band
        // ... of the lane's own level (set-up role): the same formulas on per-lane operands
        const int my_rlo = (int)sw_div((unsigned)(Y0 * myH), (unsigned)H0) - kSwMargin;
        const int my_rhi = (int)sw_div((unsigned)((Y1 - 1) * myH), (unsigned)H0) + kSwMargin;
        const int my_nr1 = (my_rhi - my_rlo + 1 > myCR ? myCR : my_rhi - my_rlo + 1) - 1;
#pragma unroll
        for (int l = 0; l < kSwLevels; ++l) {
            const int rlo = (l == 0 ? Y0 : (int)sw_div((unsigned)(Y0 * LH(l)), (unsigned)H0)) - kSwMargin;
            const int rhi = (l == 0 ? Y1 - 1 : (int)sw_div((unsigned)((Y1 - 1) * LH(l)), (unsigned)H0)) + kSwMargin;
            int nr = rhi - rlo + 1;
            nr = nr > sw_cr(l) ? sw_cr(l) : nr;
            RLO[l] = __builtin_amdgcn_readfirstlane(rlo);
            NR[l] = __builtin_amdgcn_readfirstlane(nr);
            if (l > 0) {
                const int ya = sw_first(Y0, LH(l), H0), yb = sw_first(Y1, LH(l), H0);
                YAv = lane == l ? ya : YAv;
                NYv = lane == l ? yb - ya : NYv;
            }
        }

        // column geometry: cl_l(s) and xa_l(s) = first column of level l whose centre lies at level-0 column 8 s or beyond (the
        // coarser pixels of a step are [xa(s), xa(s + 1))).  One lane-parallel evaluation gives all seven values of a step.
        auto geometry = [&](int s) -> int {                                  // lanes 0..3: cl_lane(s); lanes 4..6: xa_(lane-3)(s)
            const bool is_cl = lane < 4;
            const unsigned c0 = (unsigned)(s * kSwTW);
            const unsigned num = is_cl ? c0 * (unsigned)geoW : 2u * c0 * (unsigned)geoW + (unsigned)W0 - 1u;
            unsigned v = sw_div(num, (unsigned)W0);
            v = is_cl ? v : v >> 1;
            v = (c0 >= (unsigned)W0 || v > (unsigned)geoW) ? (unsigned)geoW : v;
            return (int)v;
        };

        // this lane's query in a step: waves [0, th2) = half a level-0 row of the step each, waves [th2, 16) = the coarser pixels
        // [xa_l, xb_l) x [YA_l, YA_l + NY_l), l = 1..3
        // (ga / gb = the geometry vectors of step s and s + 1)
        auto query_of = [&](int s, int ga, int gb) -> int {
            if (wave < th2) {
                const int x = s * kSwTW + (wave & 1) * kSwQ + qi;
                return x < W0 ? T0 + (Y0 + (wave >> 1)) * W0 + x : -1;
            }
            int j = (wave - th2) * kSwQ + qi, q = -1;
#pragma unroll
            for (int l = 1; l < kSwLevels; ++l) {
                const int xa = __builtin_amdgcn_readlane(ga, l + 3), nx = __builtin_amdgcn_readlane(gb, l + 3) - xa;
                const int ya = __builtin_amdgcn_readlane(YAv, l), n = nx * __builtin_amdgcn_readlane(NYv, l);
                if (q < 0 && j >= 0 && j < n) {
                    const int yy = (int)sw_div((unsigned)j, (unsigned)nx);
                    q = LS(l) + (ya + yy) * LW(l) + xa + (j - yy * nx);
                }
                j -= n;
            }
            return q;
        };
        auto issue_loads = [&](Inputs &r, int q) {
            const unsigned e = (unsigned)(q >= 0 ? q : 0) * (kSwHeads * kSwLevels * kSwPoints) + (unsigned)(lane & 15);   // element sl * 4 + pp
            const unsigned ol = e * 8u, oa = e * 4u;
            asm volatile("global_load_dwordx2 %0, %2, %4\n\tglobal_load_dword %1, %3, %5"
                         : "=&v"(r.xy), "=&v"(r.a)
                         : "v"(ol), "v"(oa), "s"(loc_b), "s"(att_b)
                         : "memory");
        };
        // s_waitcnt vmcnt(n) for registers hipcc does not know to be in flight: a wait-only statement (naming the registers as
        // operands made hipcc COPY them into the statement's operand registers ahead of the wait -- unlanded data) followed by a
        // scheduling barrier, so that no use is moved above it
        auto wait_vm = [&](int n) {
            switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;                  // more than expected in flight: drain
            }
            __builtin_amdgcn_sched_barrier(0);
        };

        // ---- LDS-DMA of one 16-row piece of one column of level l into ring slot `slot` ---------------------------------------
        auto fill_piece = [&](int l, int c, int slot, int second) {
            // fill role: lane = (row lane >> 2 of the 16-row piece, 16-byte chunk lane & 3); the column's first piece = rows 0..15
            // of the window, its second = rows nr - 16 .. nr - 1
            const int y = RLO[l] + (second ? NR[l] - 16 : 0) + (lane >> 2);
            const bool ok = c >= 0 && c < LW(l) && y >= 0 && y < LH(l);
            const unsigned voff = ok ? (unsigned)(LS(l) + y * LW(l) + c) * kGPixB + (unsigned)(lane & 3) * 16u : 0x80000000u;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)sw_ring(l) + (unsigned)slot * (unsigned)(sw_cr(l) * 64) +
                                                                (second ? (unsigned)(NR[l] - 16) * 64u : 0u));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                         :
                         : "s"(m0v), "v"(voff), "s"(rsrc)
                         : "memory", "m0");
        };
        // the columns [cf[l], cf[l] + n[l]) of every level, ring slot of the first = sf[l]; the (column, piece) items are dealt
        // round-robin over the waves.  Returns the number of instructions THIS wave issued.
        auto fill_columns = [&](const int (&cf)[kSwLevels], const int (&n)[kSwLevels], const int (&sf)[kSwLevels]) -> int {
            int issued = 0, base = 0;
#pragma unroll
            for (int l = 0; l < kSwLevels; ++l) {
                const int items = 2 * n[l];
                for (int i = (wave - base) & (kSwWaves - 1); i < items; i += kSwWaves) {          // uniform
                    const int col = i >> 1;
                    int slot = sf[l] + col;
                    slot = slot >= sw_rw(l) ? slot - sw_rw(l) : slot;
                    fill_piece(l, cf[l] + col, slot, i & 1);
                    ++issued;
                }
                base = (base + items) & (kSwWaves - 1);
            }
            return issued;
        };

        // ---- segment start: geometry of the first steps, inputs of the first two, warm-up fill of the whole window of sx0 -----
        int V1, V2, V3;                                                      // geometry vectors of steps s + 1, s + 2, s + 3
        int slot_new[kSwLevels];                                             // scalars: ring slot of the first new column
        int my_ca, my_cb, my_cc;                                             // per lane (level sl): cl(s), cl(s + 1), cl(s + 2)
        unsigned my_slo = 0;                                                 // per lane: ring slot of column cl(s) - 8 of level sl
        int q0, q1;
        {
            const int V0 = geometry(sx0);
            V1 = geometry(sx0 + 1); V2 = geometry(sx0 + 2); V3 = geometry(sx0 + 3);
            int cf[kSwLevels], nn[kSwLevels], sf[kSwLevels];
#pragma unroll
            for (int l = 0; l < kSwLevels; ++l) {
                const int ca = __builtin_amdgcn_readlane(V0, l);
                cf[l] = ca - kSwMargin;
                nn[l] = __builtin_amdgcn_readlane(V1, l) - ca + 2 * kSwMargin;
                sf[l] = 0;
                slot_new[l] = nn[l] >= sw_rw(l) ? nn[l] - sw_rw(l) : nn[l];
            }
            my_ca = __builtin_amdgcn_ds_bpermute(sl * 4, V0);                // lane l holds cl_l
            my_cb = __builtin_amdgcn_ds_bpermute(sl * 4, V1);
            my_cc = __builtin_amdgcn_ds_bpermute(sl * 4, V2);
            q0 = query_of(sx0, V0, V1);
            q1 = sx0 + 1 < sx_end ? query_of(sx0 + 1, V1, V2) : -1;
            issue_loads(r0, q0);
            issue_loads(r1, q1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                    // the previous segment's last gathers are done
            if (!(dbg & 1)) fill_columns(cf, nn, sf);
        }
        wait_vm(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        // ---- one step: `cur` = this step's inputs, `nxt` = the next step's (issued one step ago), `nn` = loaded here for sx + 2 --
        auto step = [&](int sx, Inputs &cur, Inputs &nxt, Inputs &nn) {
            // geometry three / four steps ahead, the query of step sx + 2 (its inputs are loaded below)
            const int V4 = geometry(sx + 4);
            const int q2 = sx + 2 < sx_end ? query_of(sx + 2, V2, V3) : -1;
            if (sx >= sx_end) { issue_loads(nn, -1); return; }              // keeps the rotation; uniform
            const bool busy = __ballot(q0 >= 0) != 0ull;                     // uniform: does this wave have queries in this step?
            const bool next = sx + 1 < sx_end;

            // ---- set-up of this lane's sample (query qi, level sl, point pp): msda_fwd.hip's arithmetic
            // (ms_deform_im2col_cuda.cuh:22-73, 274-277) ----------------------------------------------------------------
            float w00, w01, w10, w11;
            unsigned pk;
            bool pend;
            {
                float x = cur.xy.x * myWf - 0.5f, y = cur.xy.y * myHf - 0.5f;
                x = fminf(fmaxf(x, -2.f), myWc);                              // NaN -> -2: outside
                y = fminf(fmaxf(y, -2.f), myHc);
                const float xf = floorf(x), yf = floorf(y);
                const int x0 = (int)xf, y0 = (int)yf;
                const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
                // -1 < x < W  <=>  x0 in [-1, W - 1] (x == -1 exactly weighs the outside column: contributes zero either way)
                const bool valid = q0 >= 0 && (unsigned)(x0 + 1) <= (unsigned)myW && (unsigned)(y0 + 1) <= (unsigned)myH;
                w00 = hy * hx * cur.a; w01 = hy * lx * cur.a; w10 = ly * hx * cur.a; w11 = ly * lx * cur.a;
                const int cx = x0 - (my_ca - kSwMargin), cy = y0 - my_rlo;
                const bool in_win = (unsigned)cx < (unsigned)(my_cb - my_ca + 2 * kSwMargin - 1) && (unsigned)cy < (unsigned)my_nr1;
                const bool live = valid && in_win && !(dbg & 8);
                pend = valid && !in_win && !(dbg & 8);
                pk = ((unsigned)sl << 30) | ((unsigned)(y0 + 2) << 15) | (unsigned)(x0 + 2);
                unsigned s0 = my_slo + (unsigned)cx;
                s0 = s0 < s0 - myRW ? s0 : s0 - myRW;                         // wrap (s0 < 2 RW)
                unsigned s1 = s0 + 1u;
                s1 = s1 < s1 - myRW ? s1 : s1 - myRW;
                const unsigned rowb = myRB + (unsigned)cy * 64u;
                // the staged set-up: top corners' rows on the left / right side (the bottom ones 64 B further on) and the weights
                *reinterpret_cast<unsigned *>(st_o) = live ? rowb + s0 * myCS : o_zero;
                *reinterpret_cast<unsigned *>(st_o + 256) = live ? rowb + s1 * myCS : o_zero + 128u;
                unsigned h01, l01, h23, l23;
                sw_split2(w00, w01, h01, l01);
                sw_split2(w10, w11, h23, l23);
                *reinterpret_cast<u32x2 *>(st_w) = u32x2{h01, h23};
                *reinterpret_cast<u32x2 *>(st_w + 128) = u32x2{l01, l23};
            }

            // ---- flagged samples (corners not all inside the ring window): their four corner rows come straight from global
            // memory into REGISTERS, 4 samples per load instruction (lane = sample lane >> 4, corner (lane >> 2) & 3: TL BL TR BR,
            // 16-byte chunk lane & 3), and are added in fp32 at the end of the step ----------------------------------------------
            const unsigned long long fm = __ballot(pend);
            const int nflag = __builtin_popcountll(fm);                      // uniform
            int *const fgo = reinterpret_cast<int *>(wreg + kSwFgo);
            u32x4 F0 = {0u, 0u, 0u, 0u};
            auto flag_offset = [&](int slot_i, int count) -> unsigned {      // this lane's byte offset for flagged sample slot_i
                const int c4 = (lane >> 2) & 3;
                const unsigned p = (unsigned)fgo[slot_i * 8];
                const int pl = (int)(p >> 30);
                const int pw = pick(pl, W0, W1, W2, W3), ph = pick(pl, H0, H1, H2, H3), ps = pick(pl, T0, T1, T2, T3);
                const int xx = (int)(p & 0x7fffu) - 2 + (c4 >> 1), yy = (int)((p >> 15) & 0x7fffu) - 2 + (c4 & 1);
                const bool ok = slot_i < count && (unsigned)xx < (unsigned)pw && (unsigned)yy < (unsigned)ph;
                return ok ? (unsigned)(ps + yy * pw + xx) * kGPixB + (unsigned)(lane & 3) * 16u : 0x80000000u;
            };
            auto flag_publish = [&](int first) {                             // pending samples of rank first .. first + 3 -> fgo
                const unsigned long long f = __ballot(pend);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(f >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)f, 0)) - first;
                if (pend && rank >= 0 && rank < kSwFlagCap) {
                    int *e = fgo + rank * 8;
                    e[0] = (int)pk; e[1] = qi;
                    e[2] = __builtin_bit_cast(int, w00); e[3] = __builtin_bit_cast(int, w10);      // corner order of the loads: TL BL TR BR
                    e[4] = __builtin_bit_cast(int, w01); e[5] = __builtin_bit_cast(int, w11);
                }
                sw_wave_sync();
            };
            int vm_after_flag = 2;                                           // vector-memory instructions issued after the flagged load
            if (nflag) {                                                     // uniform
                flag_publish(0);
                const unsigned oa = flag_offset(lane >> 4, nflag);
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(F0) : "v"(oa), "s"(rsrc) : "memory");
            }

            // ---- prefetch: the new ring columns of step sx + 1, then the inputs of step sx + 2 ----------------------------------
            if (next && !(dbg & 1)) {
                int cf[kSwLevels], nn_[kSwLevels];
#pragma unroll
                for (int l = 0; l < kSwLevels; ++l) {
                    const int cb = __builtin_amdgcn_readlane(V1, l);
                    cf[l] = cb + kSwMargin;
                    nn_[l] = __builtin_amdgcn_readlane(V2, l) - cb;
                }
                vm_after_flag += fill_columns(cf, nn_, slot_new);
            }
            issue_loads(nn, q2);

            // ---- the MFMA steps: 4 levels x 2 point pairs, software-pipelined (the operands of step s + 1 are on their way while
            // the MFMAs of step s run) -----------------------------------------------------------------------------------------
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // D rows 4 g' + r of a lane = query 2 g' + (r >> 1), part r & 1,
                                                                             // channel (lane & 15) + 16 ((r >> 1) ^ X); g' < 2
            sw_wave_sync();
            if (busy && !(dbg & 2)) {
                struct Operands { u32x4 af; u32x2 x0, x1, y0, y1; };
                auto fetch = [&](unsigned wa, unsigned oa, unsigned ob) {
                    Operands r;
                    r.af = lds_b128(wa);
                    r.x0 = lds_tr(oa); r.x1 = lds_tr(ob); r.y0 = lds_tr(oa ^ 32u); r.y1 = lds_tr(ob ^ 32u);
                    return r;
                };
                auto fma2 = [&](const Operands &r) {
                    const u32x4 b0 = {r.x0.x, r.x0.y, r.x1.x, r.x1.y}, b1 = {r.y0.x, r.y0.y, r.y1.x, r.y1.y};
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sw_bf16x8, r.af), __builtin_bit_cast(sw_bf16x8, b0), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sw_bf16x8, r.af), __builtin_bit_cast(sw_bf16x8, b1), acc1, 0, 0, 0);
                };
                u32x4 so0 = lds_b128(o_rd), so1 = lds_b128(o_rd + 16u);
                Operands ra = fetch(w_rd, so0.x + cd, so0.y + cd);
                Operands rb = fetch(w_rd + 16, so0.z + cd, so0.w + cd);
                so0 = lds_b128(o_rd + 32u);
                fma2(ra);
                ra = fetch(w_rd + 32, so1.x + cd, so1.y + cd);
                fma2(rb);
                rb = fetch(w_rd + 48, so1.z + cd, so1.w + cd);
                so1 = lds_b128(o_rd + 48u);
                fma2(ra);
                ra = fetch(w_rd + 64, so0.x + cd, so0.y + cd);
                fma2(rb);
                rb = fetch(w_rd + 80, so0.z + cd, so0.w + cd);
                fma2(ra);
                ra = fetch(w_rd + 96, so1.x + cd, so1.y + cd);
                fma2(rb);
                rb = fetch(w_rd + 112, so1.z + cd, so1.w + cd);
                fma2(ra);
                fma2(rb);
            }
            sw_wave_sync();

            // ---- out[query][channel] = D[hi row] + D[lo row] (+ the flagged samples), transposed through the wave's staging area
            // so that a lane stores 16 bytes ------------------------------------------------------------------------------------
            if (busy && !(dbg & 4)) {
                float *tr = reinterpret_cast<float *>(wreg + kSwStageW);     // 512 B: 4 queries x 32 channels
                if (lane < 32) {
                    tr[(2 * kg) * 32 + (lane & 15)] = acc0.x + acc0.y;
                    tr[(2 * kg + 1) * 32 + (lane & 15) + 16] = acc0.z + acc0.w;
                    tr[(2 * kg) * 32 + (lane & 15) + 16] = acc1.x + acc1.y;
                    tr[(2 * kg + 1) * 32 + (lane & 15)] = acc1.z + acc1.w;
                }
                sw_wave_sync();
                if (nflag) {                                                 // uniform
                    // the loads have landed by now (counted wait: the fills and loads issued after them stay in flight)
                    wait_vm(vm_after_flag);
                    auto flag_add = [&](const u32x4 &F, int slot_i) {        // this lane: one corner, 8 channels of flagged sample slot_i
                        const int c4 = (lane >> 2) & 3;
                        const float wv = __builtin_bit_cast(float, fgo[slot_i * 8 + 2 + c4]);
                        const int qs = fgo[slot_i * 8 + 1];
                        float p[8];
                        p[0] = bf16_bits_to_f32(F.x & 0xffffu) * wv; p[1] = __builtin_bit_cast(float, F.x & 0xffff0000u) * wv;
                        p[2] = bf16_bits_to_f32(F.y & 0xffffu) * wv; p[3] = __builtin_bit_cast(float, F.y & 0xffff0000u) * wv;
                        p[4] = bf16_bits_to_f32(F.z & 0xffffu) * wv; p[5] = __builtin_bit_cast(float, F.z & 0xffff0000u) * wv;
                        p[6] = bf16_bits_to_f32(F.w & 0xffffu) * wv; p[7] = __builtin_bit_cast(float, F.w & 0xffff0000u) * wv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) p[e] = sw_row_ror_add<8>(sw_row_ror_add<4>(p[e]));      // over the four corners
                        if (c4 == 0) {
                            float *dst = tr + qs * 32 + (lane & 3) * 8;
#pragma unroll
                            for (int e = 0; e < 8; ++e) __hip_atomic_fetch_add(dst + e, p[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                    };
                    int done = 0;
                    while (true) {                                           // uniform; further trips only with more than 4 flagged
                        if ((lane >> 4) < nflag - done) flag_add(F0, lane >> 4);
                        done += kSwFlagCap;
                        if (done >= nflag) break;
                        sw_wave_sync();
                        flag_publish(done);
                        const unsigned oa = flag_offset(lane >> 4, nflag - done);
                        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=&v"(F0) : "v"(oa), "s"(rsrc) : "memory");
                    }
                    sw_wave_sync();
                }
                // lane (query slot (lane >> 2) & 3, chunk lane & 3) stores 8 channels; the query index of slot k lives in lane 16 k
                const int sq = __builtin_amdgcn_ds_bpermute(((lane >> 2) & 3) * 64, q0);
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(tr + ((lane >> 2) & 3) * 32 + (lane & 3) * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(tr + ((lane >> 2) & 3) * 32 + (lane & 3) * 8 + 4);
                u32x4 w;
                w.x = pack_bf16x2(lo.x, lo.y);
                w.y = pack_bf16x2(lo.z, lo.w);
                w.z = pack_bf16x2(hi.x, hi.y);
                w.w = pack_bf16x2(hi.z, hi.w);
                const unsigned od = (unsigned)(sq >= 0 ? sq : 0) * (kSwHeads * kSwHeadDim * 2) + (unsigned)(lane & 3) * 16u;
                if (lane < 16 && sq >= 0)
                    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(od), "v"(w), "s"(out_b) : "memory");
                sw_wave_sync();
                // end of the step: the fills of step sx + 1 have landed and the NEXT step's inputs are complete (this step's loads and
                // the store may stay in flight)
                wait_vm(3);
            } else {
                wait_vm(2);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();

            // shift the geometry pipeline
#pragma unroll
            for (int l = 0; l < kSwLevels; ++l) {
                int s = slot_new[l] + __builtin_amdgcn_readlane(V2, l) - __builtin_amdgcn_readlane(V1, l);
                slot_new[l] = s >= sw_rw(l) ? s - sw_rw(l) : s;
            }
            my_slo += (unsigned)(my_cb - my_ca);
            my_slo = my_slo < my_slo - myRW ? my_slo : my_slo - myRW;
            my_ca = my_cb;
            my_cb = my_cc;
            my_cc = __builtin_amdgcn_ds_bpermute(sl * 4, V3);
            V1 = V2; V2 = V3; V3 = V4;
            q0 = q1;
            q1 = q2;
        };
        for (int sx = sx0; sx < sx_end; sx += 3) {
            step(sx, r0, r1, r2);
            step(sx + 1, r1, r2, r0);
            step(sx + 2, r2, r0, r1);
        }
        gs += sx_end - sx0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Band height for a level table (HOST copy): the largest number of level-0 rows per band, at most kSwMaxTH, such that the
// coarser pixels of every step fit the remaining waves' octets and the rings hold two steps' columns.  0 = cannot be served.
static int sweep_band_height(const int64_t *shapes, int *nb_out, int *steps_x_out)
{
    const long long H0 = shapes[0], W0 = shapes[1];
    const long long steps_x = (W0 + kSwTW - 1) / kSwTW;
    for (int l = 0; l < kSwLevels; ++l) {
        const long long W = shapes[2 * l + 1];
        for (long long s = 0; s < steps_x; ++s)
            if (sweep_cl(s + 2, W, W0) - sweep_cl(s, W, W0) + 2 * kSwMargin > sw_rw(l)) return 0;
    }
    for (int th = kSwMaxTH; th >= 1; --th) {
        const long long nb = (H0 + th - 1) / th;
        bool ok = true;
        for (long long b = 0; b < nb && ok; ++b) {
            const long long Y0 = b * H0 / nb, Y1 = (b + 1) * H0 / nb;
            if (Y1 - Y0 > th) { ok = false; break; }
            for (long long s = 0; s < steps_x && ok; ++s) {
                long long n = 0;
                for (int l = 1; l < kSwLevels; ++l) {
                    const long long W = shapes[2 * l + 1], H = shapes[2 * l];
                    const long long xa = s * kSwTW >= W0 ? W : sweep_first(s * kSwTW, W, W0);
                    const long long xb = (s + 1) * kSwTW >= W0 ? W : sweep_first((s + 1) * kSwTW, W, W0);
                    n += (xb - xa) * (sweep_first(Y1, H, H0) - sweep_first(Y0, H, H0));
                }
                if (n > kSwQ * (kSwWaves - 2 * (Y1 - Y0))) ok = false;
            }
        }
        if (ok) {
            *nb_out = (int)nb;
            *steps_x_out = (int)steps_x;
            return th;
        }
    }
    return 0;
}

// Returns RDETR_ERR_UNSUPPORTED when the shape is not served (callers then use the direct kernel).  `shapes` / `level_start`
// are HOST pointers.
template <bool HM>
int msda_sweep_forward(const uint16_t *value, const int64_t *shapes, const int64_t *level_start, const float *loc,
                       const float *attn, int B, int S, int L, int Nq, int dbg, uint16_t *out, hipStream_t stream)
{
    if (L != kSwLevels || Nq != S) return RDETR_ERR_UNSUPPORTED;
    if (!rdetr_msda_levels_window_ok(shapes, level_start, L, S)) return RDETR_ERR_UNSUPPORTED;
    const long long gpix = HM ? 64 : 512;
    if ((long long)S * gpix >= (1ll << 31)) return RDETR_ERR_UNSUPPORTED;
    SweepLevels lv;
    for (int l = 0; l < kSwLevels; ++l) {
        const long long h = shapes[2 * l], w = shapes[2 * l + 1];
        if (h > 2048 || w > 2048) return RDETR_ERR_UNSUPPORTED;             // geometry arithmetic: 2 * 2048 * 2048 < 2^24
        lv.h[l] = (int)h; lv.w[l] = (int)w; lv.start[l] = (int)level_start[l];
    }
    int nb = 0, steps_x = 0;
    const int th = sweep_band_height(shapes, &nb, &steps_x);
    if (th == 0) return RDETR_ERR_UNSUPPORTED;
    const long long total = (long long)B * kSwHeads * nb * steps_x;
    if (total >= (1ll << 24)) return RDETR_ERR_UNSUPPORTED;                  // sw_div on the step id
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return RDETR_ERR_LAUNCH;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    auto kern = msda_fwd_sweep_kernel<HM>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kSwLdsBytes);
    if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
    const long long nblk = total < cus ? total : cus;                        // one persistent workgroup per CU
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(kSwThreads), (size_t)kSwLdsBytes, stream, value, loc, attn, lv, S, nb,
                       steps_x, (int)total, (int)nblk, dbg, out);
    return launch_status();
}

}  // namespace rdetr

#ifdef RDETR_DEV
static int g_sweep_dbg = 0;
extern "C" void rdetr_dev_set_sweep_dbg(int v) { g_sweep_dbg = v; }
#define RDETR_SWEEP_DBG g_sweep_dbg
#else
#define RDETR_SWEEP_DBG 0
#endif

extern "C" int rdetr_msda_forward_sweep_bf16(const uint16_t *value, int value_layout, const int64_t *host_spatial_shapes,
                                             const int64_t *host_level_start_index, const float *sampling_loc,
                                             const float *attn_weight, int B, int S, int H, int D, int L, int Nq, int P,
                                             uint16_t *out, void *stream)
{
    if (B < 0 || S < 0 || Nq < 0 || H <= 0 || D <= 0 || L <= 0 || P <= 0) return RDETR_ERR_INVALID_ARG;
    if (value_layout != RDETR_VALUE_BSHD && value_layout != RDETR_VALUE_BHSD) return RDETR_ERR_INVALID_ARG;
    if (B == 0 || Nq == 0) return RDETR_OK;
    if (!value || !host_spatial_shapes || !host_level_start_index || !sampling_loc || !attn_weight || !out) return RDETR_ERR_INVALID_ARG;
    if (S == 0) return RDETR_ERR_INVALID_ARG;
    if (H != rdetr::kSwHeads || D != rdetr::kSwHeadDim || P != rdetr::kSwPoints) return RDETR_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(value) % 16 || reinterpret_cast<uintptr_t>(out) % 16 ||
        reinterpret_cast<uintptr_t>(sampling_loc) % 16 || reinterpret_cast<uintptr_t>(attn_weight) % 8)
        return RDETR_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return value_layout == RDETR_VALUE_BHSD
               ? rdetr::msda_sweep_forward<true>(value, host_spatial_shapes, host_level_start_index, sampling_loc, attn_weight, B,
                                                 S, L, Nq, RDETR_SWEEP_DBG, out, s)
               : rdetr::msda_sweep_forward<false>(value, host_spatial_shapes, host_level_start_index, sampling_loc, attn_weight, B,
                                                  S, L, Nq, RDETR_SWEEP_DBG, out, s);
}
