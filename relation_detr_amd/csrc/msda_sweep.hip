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

constexpr int kSwThreads = 512;
constexpr int kSwWaves = kSwThreads / kWave;                  // 8 = octets per step
constexpr int kSwTW = 8;                                      // level-0 columns per step
constexpr int kSwMaxTH = 6;                                   // level-0 rows per band
constexpr int kSwHeads = 8, kSwHeadDim = 32, kSwPoints = 4, kSwLevels = 4;
constexpr int kSwMargin = 8;
// rings: columns (even, so that a wrapped neighbour column keeps the bank phase) and rows per column (== 2 mod 4: the column
// stride is == 128 mod 256 bytes, the four corners of a sample fall on four bank groups)
constexpr int kSwRW0 = 32, kSwRW1 = 26, kSwRW2 = 22, kSwRW3 = 20;
constexpr int kSwCR0 = 22, kSwCR1 = 22, kSwCR2 = 22, kSwCR3 = 18;
__host__ __device__ constexpr int sw_rw(int l) { return l == 0 ? kSwRW0 : l == 1 ? kSwRW1 : l == 2 ? kSwRW2 : kSwRW3; }
__host__ __device__ constexpr int sw_cr(int l) { return l == 0 ? kSwCR0 : l == 1 ? kSwCR1 : l == 2 ? kSwCR2 : kSwCR3; }

// ---- LDS map ------------------------------------------------------------------------------------------------------
constexpr int kSwZeroOff = 0;                                 // 1 KiB of zeros: the zero sample (TL 0, BL 64, TR 128, BR 192) and what
constexpr int kSwZeroKOff = 512 + 32;                         // idle A-operand lanes read
constexpr int kSwWaveOff = 1024;                              // per-wave area:
constexpr int kSwStageW = 0;                                  //   [0, 512)      W[query 8][part 2][point 4][corner 4] bf16
constexpr int kSwStageO = 512;                                //   [512, 1024)   O[corner 4: TL TR BL BR][query 8][point 4] u32: LDS address of that corner's row
constexpr int kSwPatch = 1024;                                //   [1024, 2048)  4 patch cells of 256 B: [TL 64][BL 64][TR 64][BR 64]
constexpr int kSwFgo = 2048;                                  //   [2048, 2112)  packed pixels of the flagged samples in flight
constexpr int kSwWaveBytes = 2176;
constexpr int kSwRing0 = kSwWaveOff + kSwWaves * kSwWaveBytes;                       // 18432
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
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ __forceinline__ void sw_wait_vm(int n)
{
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;        // more than expected in flight: drain (always safe)
    }
}

// HM = false: value [B,S,H,D] (the reference operator's layout); HM = true: value [B,H,S,D] (head-major).
template <bool HM>
__global__ __launch_bounds__(kSwThreads, 2) void msda_fwd_sweep_kernel(
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
    // set-up: query qi of the octet, level sl, point pair hf (points 2 hf, 2 hf + 1)
    const int qi = lane >> 3, sl = (lane >> 1) & 3, hf = lane & 1;
    // gather: K-group kg, corner tq / piece tp of a transposed read; as an A-operand lane: row am = lane & 15 =
    // 8 * (quad half ah) + 2 * (K-group ag) + (0 = bf16 high part, 1 = low part)
    const int kg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int am = lane & 15, ah = am >> 3, ag = (am >> 1) & 3, apart = am & 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;     // 0 in practice
    const unsigned wave_off = (unsigned)(kSwWaveOff + wave * kSwWaveBytes);
    unsigned char *const wreg = lds + wave_off;
    int *const fgo = reinterpret_cast<int *>(wreg + kSwFgo);
    const unsigned cell0 = __builtin_amdgcn_readfirstlane(lds0 + wave_off + (unsigned)kSwPatch);
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
    // One MFMA step = quad half h, point pair j: K-group kg carries the two samples (points 2j, 2j + 1) of query 4 h + kg; its
    // lane (corner tq, piece tp) reads 8 bytes of that corner's row of each
    const unsigned cd = (unsigned)tp * 8u;
    const unsigned o_rd = lds0 + wave_off + (unsigned)kSwStageO + (unsigned)tq * 128u + (unsigned)kg * 16u;   // + h * 64
    const unsigned w_real = lds0 + wave_off + (unsigned)kSwStageW + (unsigned)((4 * ah + kg) * 64 + apart * 32);
    const unsigned w_rd0 = (ag == kg && ah == 0) ? w_real : lds0 + (unsigned)kSwZeroKOff;
    const unsigned w_rd1 = (ag == kg && ah == 1) ? w_real : lds0 + (unsigned)kSwZeroKOff;
    // staging writes of the set-up role
    unsigned char *const st_o = wreg + kSwStageO + qi * 16 + hf * 8;              // + corner * 128
    unsigned char *const st_w = wreg + kSwStageW + qi * 64 + hf * 16;             // + part * 32

    auto lds_b128 = [](unsigned a) { return *(__attribute__((address_space(3))) const u32x4 *)a; };
    auto lds_tr = [](unsigned a) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sw_s16x4 *)a));
    };

    // The inputs of a step -- locations of the lane's two points (x0, y0, x1, y1) and their weights -- are loaded TWO steps
    // ahead by inline assembly into one of three register sets used in rotation: hipcc does not know these loads, so it
    // neither waits for them nor drains the window fills in flight when it meets their first use (tracked loads cost a
    // vmcnt(0) per step here: a full trip to HBM).  A set is complete when the step after the one that issued it ends (its
    // counted wait names the set, so no use is scheduled above it).
    struct Inputs { f32x4 xy; f32x2 a; };
    Inputs r0, r1, r2;
    r0.xy = r1.xy = r2.xy = f32x4{0.f, 0.f, 0.f, 0.f};
    r0.a = r1.a = r2.a = f32x2{0.f, 0.f};

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
        const int th = Y1 - Y0;                                       // <= kSwMaxTH

        // the (image, head) value plane behind one wave-uniform buffer descriptor; byte offsets inside it are 32-bit
        const unsigned char *plane = reinterpret_cast<const unsigned char *>(value) +
                                     (HM ? ((size_t)b * kSwHeads + m) * (size_t)S * 64u
                                         : (size_t)b * S * kGPixB + (size_t)m * 64u);
        const unsigned plane_bytes = HM ? (unsigned)S * 64u : (unsigned)S * kGPixB - (unsigned)m * 64u;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(plane), 0, plane_bytes, 0x00020000);
        const float *loc_b = loc + ((size_t)b * Nq * kSwHeads + (size_t)m) * (kSwLevels * kSwPoints * 2) + sl * 8 + hf * 4;
        const float *att_b = attn + ((size_t)b * Nq * kSwHeads + (size_t)m) * (kSwLevels * kSwPoints) + sl * 4 + hf * 2;
        uint16_t *out_b = out + (size_t)b * Nq * (kSwHeads * kSwHeadDim) + m * kSwHeadDim + (lane & 3) * 8;

        // rows of the band's windows, per level: [rlo, rlo + nrows); rows of the coarser pixels inside the band: [ya, ya + ny)
        int NR[kSwLevels], YA[kSwLevels], NY[kSwLevels];
        unsigned ROWA[kSwLevels], ROWB[kSwLevels];     // fill role: lane = (row lane >> 2 of a 16-row piece, 16-byte chunk lane & 3):
                                                       // byte offset of the lane's row for the column's first piece (rows 0..15)
                                                       // and its second (rows nr - 16 .. nr - 1), or "out of range"
        // ... of the lane's own level (set-up role): the same formulas on per-lane operands
        const int my_rlo = (int)sw_div((unsigned)(Y0 * myH), (unsigned)H0) - kSwMargin;
        const int my_rhi = (int)sw_div((unsigned)((Y1 - 1) * myH), (unsigned)H0) + kSwMargin;
        const int my_nr = my_rhi - my_rlo + 1 > myCR ? myCR : my_rhi - my_rlo + 1;
#pragma unroll
        for (int l = 0; l < kSwLevels; ++l) {
            const int rlo = (l == 0 ? Y0 : (int)sw_div((unsigned)(Y0 * LH(l)), (unsigned)H0)) - kSwMargin;
            const int rhi = (l == 0 ? Y1 - 1 : (int)sw_div((unsigned)((Y1 - 1) * LH(l)), (unsigned)H0)) + kSwMargin;
            int nr = rhi - rlo + 1;
            nr = nr > sw_cr(l) ? sw_cr(l) : nr;
            NR[l] = __builtin_amdgcn_readfirstlane(nr);
            const int ya = l == 0 ? Y0 : sw_first(Y0, LH(l), H0), yb = l == 0 ? Y1 : sw_first(Y1, LH(l), H0);
            YA[l] = __builtin_amdgcn_readfirstlane(ya);
            NY[l] = __builtin_amdgcn_readfirstlane(yb - ya);
            const int fa = rlo + (lane >> 2), fb = rlo + nr - 16 + (lane >> 2);
            ROWA[l] = (fa >= 0 && fa < LH(l)) ? (unsigned)(LS(l) + fa * LW(l)) * kGPixB + (unsigned)(lane & 3) * 16u : 0x80000000u;
            ROWB[l] = (fb >= 0 && fb < LH(l)) ? (unsigned)(LS(l) + fb * LW(l)) * kGPixB + (unsigned)(lane & 3) * 16u : 0x80000000u;
        }

        // column geometry: cl(s) per level, xa(s) = first column whose centre lies at level-0 column 8 s or beyond (the coarser
        // pixels of a step are [xa(s), xa(s + 1))).  Computed PER LANE for the lane's level sl (lane 2 l = level l), read as
        // scalars by the other roles.
        auto cl_of = [&](int s) { return sw_cl(s, myW, W0); };
        auto xa_of = [&](int s) { const int c0 = s * kSwTW; return c0 >= W0 ? myW : sw_first(c0, myW, W0); };
        auto lvl = [&](int v, int l) { return __builtin_amdgcn_readlane(v, 2 * l); };

        // this lane's query in a step: waves [0, th) = the level-0 rows of the band, waves [th, 8) = the coarser pixels
        // [xa_l, xb_l) x [YA_l, YA_l + NY_l), l = 1..3
        auto query_of = [&](int s, const int (&xa)[kSwLevels], const int (&xb)[kSwLevels]) -> int {
            if (wave < th) {
                const int x = s * kSwTW + qi;
                return x < W0 ? T0 + (Y0 + wave) * W0 + x : -1;
            }
            int j = (wave - th) * 8 + qi, q = -1;
#pragma unroll
            for (int l = 1; l < kSwLevels; ++l) {
                const int nx = xb[l] - xa[l], n = nx * NY[l];
                if (q < 0 && j >= 0 && j < n) {
                    const int yy = (int)sw_div((unsigned)j, (unsigned)nx);
                    q = LS(l) + (YA[l] + yy) * LW(l) + xa[l] + (j - yy * nx);
                }
                j -= n;
            }
            return q;
        };
        auto issue_loads = [&](Inputs &r, int q) {
            const unsigned e = (unsigned)(q >= 0 ? q : 0) * (kSwHeads * kSwLevels * kSwPoints);
            const float *pl = loc_b + 2u * e, *pa = att_b + e;
            asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx2 %1, %3, off"
                         : "=&v"(r.xy), "=&v"(r.a)
                         : "v"(pl), "v"(pa)
                         : "memory");
        };
        // s_waitcnt vmcnt(n) that also names a register set: nothing that uses the set is scheduled above the wait
        auto wait_set = [&](int n, Inputs &r) {
            switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.xy), "+v"(r.a) : : "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" : "+v"(r.xy), "+v"(r.a) : : "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" : "+v"(r.xy), "+v"(r.a) : : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(3)" : "+v"(r.xy), "+v"(r.a) : : "memory"); break;
            }
        };

        // ---- LDS-DMA of one 16-row piece of one column of level l into ring slot `slot` ---------------------------------------
        auto fill_piece = [&](int l, int c, int slot, int second) {
            const unsigned rowoff = second ? ROWB[l] : ROWA[l];
            const bool colok = c >= 0 && c < LW(l);
            const unsigned voff = colok ? rowoff + (unsigned)c * kGPixB : 0x80000000u;
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
        int cl0 = cl_of(sx0), cl1 = cl_of(sx0 + 1), cl2 = cl_of(sx0 + 2);    // per lane (level sl): cl(s), cl(s + 1), cl(s + 2)
        unsigned my_slo = 0;                                                 // per lane: ring slot of column cl(s) - 8 of level sl
        int C1[kSwLevels], C2[kSwLevels];                                    // scalars: cl(s + 1), cl(s + 2)
        int XA2[kSwLevels], XA3[kSwLevels];                                  // scalars: xa(s + 2), xa(s + 3)
        int slot_new[kSwLevels];                                             // scalars: ring slot of the first new column
        int q0, q1;
        {
            const int c2 = cl2, x0v = xa_of(sx0), x1v = xa_of(sx0 + 1), x2v = xa_of(sx0 + 2), x3v = xa_of(sx0 + 3);
            int XA0[kSwLevels], XA1[kSwLevels], cf[kSwLevels], nn[kSwLevels], sf[kSwLevels];
#pragma unroll
            for (int l = 0; l < kSwLevels; ++l) {
                const int c0s = lvl(cl0, l);
                C1[l] = lvl(cl1, l);
                C2[l] = lvl(c2, l);
                XA0[l] = lvl(x0v, l); XA1[l] = lvl(x1v, l); XA2[l] = lvl(x2v, l); XA3[l] = lvl(x3v, l);
                cf[l] = c0s - kSwMargin;
                nn[l] = C1[l] - c0s + 2 * kSwMargin;
                sf[l] = 0;
                slot_new[l] = nn[l] >= sw_rw(l) ? nn[l] - sw_rw(l) : nn[l];
            }
            q0 = query_of(sx0, XA0, XA1);
            q1 = sx0 + 1 < sx_end ? query_of(sx0 + 1, XA1, XA2) : -1;
            issue_loads(r0, q0);
            issue_loads(r1, q1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                    // the previous segment's last gathers are done
            if (!(dbg & 1)) fill_columns(cf, nn, sf);
        }
        wait_set(0, r0);
        wait_set(0, r1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        // ---- one step: `cur` = this step's inputs, `nxt` = the next step's (issued one step ago), `nn` = loaded here for sx + 2 --
        auto step = [&](int sx, Inputs &cur, Inputs &nxt, Inputs &nn) {
            // geometry two / three steps ahead (per lane), the query of step sx + 2, its inputs on their way
            const int cl3v = cl_of(sx + 3), xa4v = xa_of(sx + 4);
            int XA4[kSwLevels];
#pragma unroll
            for (int l = 1; l < kSwLevels; ++l) XA4[l] = lvl(xa4v, l);
            XA4[0] = 0;
            const int q2 = sx + 2 < sx_end ? query_of(sx + 2, XA2, XA3) : -1;
            if (sx >= sx_end) { issue_loads(nn, -1); return; }              // keeps the rotation; uniform
            const bool octet = __ballot(q0 >= 0) != 0ull;                    // uniform: does this wave have queries in this step?
            const bool next = sx + 1 < sx_end;

            // ---- set-up of this lane's two samples (level sl, points 2 hf, 2 hf + 1 of query qi): msda_fwd.hip's arithmetic
            // (ms_deform_im2col_cuda.cuh:22-73, 274-277) ----------------------------------------------------------------
            unsigned A_[2][4];                 // LDS addresses of the four corner rows (TL, TR, BL, BR), + par32
            unsigned WH01[2], WH23[2], WL01[2], WL23[2];
            unsigned PK[2];
            bool pend[2];
            {
                const int c_lo = cl0 - kSwMargin, span1 = cl1 - cl0 + 2 * kSwMargin - 1;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float lxn = i ? cur.xy.z : cur.xy.x, lyn = i ? cur.xy.w : cur.xy.y, aw = i ? cur.a.y : cur.a.x;
                    float x = lxn * myWf - 0.5f, y = lyn * myHf - 0.5f;
                    x = fminf(fmaxf(x, -2.f), myWc);                          // NaN -> -2: outside
                    y = fminf(fmaxf(y, -2.f), myHc);
                    const float xf = floorf(x), yf = floorf(y);
                    const int x0 = (int)xf, y0 = (int)yf;
                    const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
                    // -1 < x < W  <=>  x0 in [-1, W - 1] (x == -1 exactly weighs the outside column: contributes zero either way)
                    const bool valid = q0 >= 0 && (unsigned)(x0 + 1) <= (unsigned)myW && (unsigned)(y0 + 1) <= (unsigned)myH;
                    const float w00 = hy * hx * aw, w01 = hy * lx * aw, w10 = ly * hx * aw, w11 = ly * lx * aw;
                    const int cx = x0 - c_lo, cy = y0 - my_rlo;
                    const bool in_win = (unsigned)cx < (unsigned)span1 && (unsigned)cy < (unsigned)(my_nr - 1);
                    const bool live = valid && in_win && !(dbg & 8);
                    pend[i] = valid && !in_win && !(dbg & 8);
                    PK[i] = ((unsigned)sl << 30) | ((unsigned)(y0 + 2) << 15) | (unsigned)(x0 + 2);
                    unsigned s0 = my_slo + (unsigned)cx;
                    s0 = s0 < s0 - myRW ? s0 : s0 - myRW;                     // wrap (s0 < 2 RW)
                    unsigned s1 = s0 + 1u;
                    s1 = s1 < s1 - myRW ? s1 : s1 - myRW;
                    const unsigned rowb = myRB + (unsigned)cy * 64u;
                    const unsigned tl = rowb + s0 * myCS, tr = rowb + s1 * myCS;
                    A_[i][0] = live ? tl : o_zero;                          // corner order of the weights: TL, TR, BL, BR
                    A_[i][1] = live ? tr : o_zero + 128u;
                    A_[i][2] = live ? tl + 64u : o_zero + 64u;
                    A_[i][3] = live ? tr + 64u : o_zero + 192u;
                    sw_split2(w00, w01, WH01[i], WL01[i]);
                    sw_split2(w10, w11, WH23[i], WL23[i]);
                }
            }

            // ---- flagged samples: the first four of the wave get a patch cell each, their rows one LDS-DMA instruction --------
            int cellof[2] = {-1, -1};
            const bool any_flag = (__ballot(pend[0]) | __ballot(pend[1])) != 0ull;      // uniform
            auto patch_dma = [&](int n) {                                    // cells [0, n) <- fgo[0 .. n)
                sw_wave_sync();
                const int k = lane >> 4, c4 = (lane >> 2) & 3;               // cell, corner (TL, BL, TR, BR), chunk lane & 3
                const unsigned pk = (unsigned)fgo[k];
                const int pl = (int)(pk >> 30);
                const int pw = pick(pl, W0, W1, W2, W3), ph = pick(pl, H0, H1, H2, H3), ps = pick(pl, T0, T1, T2, T3);
                const int xx = (int)(pk & 0x7fffu) - 2 + (c4 >> 1), yy = (int)((pk >> 15) & 0x7fffu) - 2 + (c4 & 1);
                const bool ok = k < n && (unsigned)xx < (unsigned)pw && (unsigned)yy < (unsigned)ph;
                const unsigned go = ok ? (unsigned)(ps + yy * pw + xx) * kGPixB + (unsigned)(lane & 3) * 16u : 0x80000000u;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                             :
                             : "s"(cell0), "v"(go), "s"(rsrc)
                             : "memory", "m0");
            };
            auto assign_cells = [&]() -> int {                               // up to four pending samples -> cells; returns how many
                const unsigned long long f0 = __ballot(pend[0]), f1 = __ballot(pend[1]);
                const int below0 = __builtin_amdgcn_mbcnt_hi((unsigned)(f0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)f0, 0));
                const int below1 = __builtin_amdgcn_mbcnt_hi((unsigned)(f1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)f1, 0));
                const int ra = below0 + below1, rb = ra + (pend[0] ? 1 : 0);
                cellof[0] = cellof[1] = -1;
                if (pend[0] && ra < 4) { cellof[0] = ra; fgo[ra] = (int)PK[0]; pend[0] = false; }
                if (pend[1] && rb < 4) { cellof[1] = rb; fgo[rb] = (int)PK[1]; pend[1] = false; }
                const int n = __builtin_popcountll(f0) + __builtin_popcountll(f1);
                return n < 4 ? n : 4;
            };
            int vm_after_patch = 2;                                          // vector-memory instructions issued after the patch DMA
            if (any_flag) patch_dma(assign_cells());

            // ---- prefetch: the new ring columns of step sx + 1, then the inputs of step sx + 2 ----------------------------------
            if (next && !(dbg & 1)) {
                int cf[kSwLevels], nn_[kSwLevels];
#pragma unroll
                for (int l = 0; l < kSwLevels; ++l) { cf[l] = C1[l] + kSwMargin; nn_[l] = C2[l] - C1[l]; }
                vm_after_patch += fill_columns(cf, nn_, slot_new);
            }
            issue_loads(nn, q2);

            // ---- staging + MFMA rounds ---------------------------------------------------------------------------------------
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // D rows 4 g' + r of a lane = query 2 g' + (r >> 1), part r & 1,
                                                                             // channel (lane & 15) + 16 ((r >> 1) ^ X)
            auto stage = [&](bool only_patched) {
                // lanes of the round's level write their two samples: O[corner][qi][point], W[qi][part][point][corner]
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned a0 = A_[0][c], a1 = A_[1][c];
                    if (only_patched) {
                        const unsigned co = (unsigned)(c & 1) * 128u + (unsigned)(c >> 1) * 64u;       // cell / zero block: TL 0, BL 64, TR 128, BR 192
                        a0 = cellof[0] >= 0 ? cell0 + (unsigned)cellof[0] * 256u + co + par32 : o_zero + co;
                        a1 = cellof[1] >= 0 ? cell0 + (unsigned)cellof[1] * 256u + co + par32 : o_zero + co;
                    }
                    *reinterpret_cast<u32x2 *>(st_o + c * 128) = u32x2{a0, a1};
                }
                *reinterpret_cast<u32x4 *>(st_w) = u32x4{WH01[0], WH23[0], WH01[1], WH23[1]};
                *reinterpret_cast<u32x4 *>(st_w + 32) = u32x4{WL01[0], WL23[0], WL01[1], WL23[1]};
            };
            auto gather_round = [&]() {
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
                const u32x4 so0 = lds_b128(o_rd), so1 = lds_b128(o_rd + 64u);
                Operands ra = fetch(w_rd0, so0.x + cd, so0.y + cd);
                Operands rb = fetch(w_rd0 + 16, so0.z + cd, so0.w + cd);
                fma2(ra);
                ra = fetch(w_rd1, so1.x + cd, so1.y + cd);
                fma2(rb);
                rb = fetch(w_rd1 + 16, so1.z + cd, so1.w + cd);
                fma2(ra);
                fma2(rb);
            };
            if (octet && !(dbg & 2)) {
#pragma unroll 1
                for (int l = 0; l < kSwLevels; ++l) {
                    if (sl == l) stage(false);
                    sw_wave_sync();
                    gather_round();
                    sw_wave_sync();
                }
                // flagged samples: their rows have landed by now (counted wait: the fills and loads issued after the patch DMA
                // stay in flight); one extra round per level that has any
                bool first = true;
                while (any_flag) {                                           // uniform; a second trip only with more than four flagged
                    if (first) sw_wait_vm(vm_after_patch); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    first = false;
#pragma unroll 1
                    for (int l = 0; l < kSwLevels; ++l) {
                        const bool has = __ballot(sl == l && (cellof[0] >= 0 || cellof[1] >= 0)) != 0ull;
                        if (has) {                                           // uniform
                            if (sl == l) stage(true);
                            sw_wave_sync();
                            gather_round();
                            sw_wave_sync();
                        }
                    }
                    if ((__ballot(pend[0]) | __ballot(pend[1])) == 0ull) break;
                    patch_dma(assign_cells());
                }
            }

            // ---- out[query][channel] = D[hi row] + D[lo row], transposed through the wave's staging area so that a lane stores
            // 16 bytes -------------------------------------------------------------------------------------------------------
            if (octet && !(dbg & 4)) {
                float *tr = reinterpret_cast<float *>(wreg + kSwStageW);     // 1 KiB: 8 queries x 32 channels
                tr[(2 * kg) * 32 + (lane & 15)] = acc0.x + acc0.y;
                tr[(2 * kg + 1) * 32 + (lane & 15) + 16] = acc0.z + acc0.w;
                tr[(2 * kg) * 32 + (lane & 15) + 16] = acc1.x + acc1.y;
                tr[(2 * kg + 1) * 32 + (lane & 15)] = acc1.z + acc1.w;
                sw_wave_sync();
                // lane (query slot (lane >> 2) & 7, chunk lane & 3) stores 8 channels; the query index of slot k lives in lane 8 k
                const int sq = __builtin_amdgcn_ds_bpermute(((lane >> 2) & 7) * 32, q0);
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(tr + ((lane >> 2) & 7) * 32 + (lane & 3) * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(tr + ((lane >> 2) & 7) * 32 + (lane & 3) * 8 + 4);
                u32x4 w;
                w.x = pack_bf16x2(lo.x, lo.y);
                w.y = pack_bf16x2(lo.z, lo.w);
                w.z = pack_bf16x2(hi.x, hi.y);
                w.w = pack_bf16x2(hi.z, hi.w);
                uint16_t *dst = out_b + (size_t)(sq >= 0 ? sq : 0) * (kSwHeads * kSwHeadDim);
                if (lane < 32 && sq >= 0)
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(dst), "v"(w) : "memory");
                sw_wave_sync();
                // end of the step: the fills of step sx + 1 have landed and the NEXT step's inputs are complete (this step's loads and
                // the store may stay in flight)
                wait_set(3, nxt);
            } else {
                wait_set(2, nxt);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();

            // shift the geometry pipeline
#pragma unroll
            for (int l = 0; l < kSwLevels; ++l) {
                const int c3 = lvl(cl3v, l);
                int s = slot_new[l] + C2[l] - C1[l];
                slot_new[l] = s >= sw_rw(l) ? s - sw_rw(l) : s;
                C1[l] = C2[l];
                C2[l] = c3;
                XA2[l] = XA3[l];
                XA3[l] = XA4[l];
            }
            my_slo += (unsigned)(cl1 - cl0);
            my_slo = my_slo < my_slo - myRW ? my_slo : my_slo - myRW;
            cl0 = cl1;
            cl1 = cl2;
            cl2 = cl3v;
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
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0.xy), "+v"(r0.a), "+v"(r1.xy), "+v"(r1.a), "+v"(r2.xy), "+v"(r2.a) : : "memory");
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
                if (n > 8 * (kSwWaves - (Y1 - Y0))) ok = false;
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
