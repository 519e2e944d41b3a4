// Multi-scale deformable attention, forward -- TILE kernel for the ENCODER shape (queries = the pyramid's own pixels,
// Nq == S, L == 4, bf16 value, materialised sampling locations / attention weights = the reference operator's inputs,
// ms_deform_attn_cuda.cu:12-72) on gfx950 (MI355X).  Third generation of the LDS-sourced gather (round 4).
//
// The gather of ms_deformable_im2col_gpu_kernel (ms_deform_im2col_cuda.cuh:226-288) reads 4 corner rows of 64 B per sample:
// 2.93 GB per launch at BASELINE.json configs[1].  Through the vector-memory path that is one wave instruction per KiB at
// ~16.5 clocks each (profiles/r03/microbench_ta_instruction_cost_vs_live_lanes.txt) -- the ceiling of msda_fwd.hip.  The LDS
// serves the same rows at 256 B/clk/CU (MI355X_MICROARCH.md, LDS table), four times the rate, and the matrix cores do the
// weighted sum.  csrc/msda_win.hip had that data path but one 16-wave workgroup per CU running a five-stage software pipeline
// behind one barrier per pass: a wave spent half its cycles parked.  This kernel keeps the data path and drops the pipeline:
//
//   workgroup  = 512 threads (8 waves) = ONE spatial tile of one (image, head): 16 x TH pixels of level 0 (TH <= 11, chosen on
//                the host so that the rows of level 0 split evenly) plus the pixels of the coarser levels whose centres fall into
//                it -- at most 256 queries = 2 groups of 16 per wave.  ~80 KB of LDS, so TWO workgroups share a CU and the
//                hardware overlaps one's window fill / set-up with the other's gather; nothing is software-pipelined across
//                passes, nothing persists from tile to tile.
//   pass       = one sampled level: (1) the 32-pixel-wide window of the level around the tile's footprint is copied L2 -> LDS
//                by range-checked LDS-DMA, in PADDED coordinates: whatever lies outside the level arrives as zeros, so a sample
//                has one LDS offset and its corners are constants from it (the zero padding of ms_deform_im2col_cuda.cuh:44-67
//                is in the data); (2) meanwhile every lane sets up ONE sample per group (query = lane / 4, point = lane % 4):
//                pixel coordinates, LDS offset, the four corner weights split into bf16 high + low parts; (3) barrier;
//                (4) per group 8 MFMA steps: v_mfma_f32_16x16x32_bf16, K = 8 samples x 4 corners, B operand = the gathered
//                rows read straight from the window by ds_read_b64_tr_b16 (per-lane row addresses), A operand = the corner
//                weights, block diagonal, rows carrying the high and low parts (w = hi + lo to 2^-17, fp32 accumulation).
//                Two barriers per pass; between them the waves run free (staging is wave-private).
//   flagged    nothing is assumed about the sampling locations.  A sample whose corners are not all inside its window gets
//                its four corner rows fetched by one range-checked LDS-DMA into a 256-byte patch cell (two cells per group in
//                the pass's first round, sharing ONE instruction; further flagged samples take extra rounds of four) and goes
//                through the same MFMA steps: results never depend on the windows.
//   inputs     the level table comes from the HOST (kernel arguments): no device-side table set-up, the grid is sized from it.
// Per corner the arithmetic is msda_fwd.hip's (same weights); the summation order differs and each weight carries a 2^-17
// relative representation error (the bf16 output rounds at 2^-9).
#include <type_traits>

#include "common.h"

namespace rdetr {

typedef __bf16 tl_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tl_bf16x2 __attribute__((ext_vector_type(2)));
typedef short tl_s16x4 __attribute__((ext_vector_type(4)));

constexpr int kTlThreads = 512;
constexpr int kTlWaves = kTlThreads / kWave;                  // 8
constexpr int kTlGroups = 2;                                  // groups of 16 queries per wave
constexpr int kTlMaxQueries = kTlWaves * kTlGroups * 16;      // 256 query slots per tile
constexpr int kTlTW = 16;                                     // tile width, level-0 pixels: one group = one tile row
constexpr int kTlHeads = 8, kTlHeadDim = 32, kTlPoints = 4, kTlLevels = 4;
constexpr unsigned kTlPixB = 64;                              // LDS bytes per pixel (one bf16 head row)
constexpr int kTlWinW = 32;                                   // window width in pixels = two DMA instructions per row
constexpr unsigned kTlPitchB = (kTlWinW + 2) * kTlPixB;       // 2176 B == 128 (mod 256): the four corners on four bank groups
constexpr int kTlWinRows = 27;                                // window rows
constexpr int kTlMargin = 8;                                  // rows / columns of margin around a footprint
constexpr int kTlMaxTH = kTlWinRows - 2 * kTlMargin;          // 11

// ---- LDS map ------------------------------------------------------------------------------------------------------
constexpr int kTlZeroOff = 0;                                 // 1 KiB of zeros: the "zero sample" (top corners at 0, bottom corners
                                                              // 128 B further on) and what idle A-operand lanes read (their step
constexpr int kTlZeroKOff = kTlZeroOff + 128 + 32;            // offsets 0 .. 543 included)
constexpr int kTlWaveOff = 1024;                              // per-wave area:
constexpr int kTlStageW = 0;                                  //   [0, 1024)     W[query][part][point][corner] bf16
constexpr int kTlStageOT = 1024;                              //   [1024, 1280)  Otop[query][point] u32: LDS address of the sample's top corners
constexpr int kTlStageOB = 1280;                              //   [1280, 1536)  Obot[query][point]: ... of its bottom corners
constexpr int kTlPatch = 1536;                                //   [1536, 2560)  4 patch cells of 256 B: [TL 64][TR 64][BL 64][BR 64]
constexpr int kTlFgo = 2560;                                  //   [2560, 2624)  pixels of the flagged samples in flight
constexpr int kTlWaveBytes = 2624;
constexpr int kTlBufOff = kTlWaveOff + kTlWaves * kTlWaveBytes;
constexpr int kTlLdsBytes = kTlBufOff + kTlWinRows * (int)kTlPitchB;
static_assert(kTlBufOff % 64 == 0 && kTlWaveBytes % 64 == 0 && kTlPatch % 64 == 0 && kTlWaveOff % 64 == 0, "sample bases are multiples of 64");
static_assert(kTlZeroKOff + 528 + 16 <= kTlWaveOff, "idle A-operand reads stay inside the zero block");
static_assert(2 * kTlLdsBytes <= 160 * 1024, "two workgroups per CU");

struct TileLevels { int h[kTlLevels], w[kTlLevels], start[kTlLevels]; };

// a / b for a < 2^24, 0 < b < 2^24 without the integer division sequence
__device__ __forceinline__ unsigned tl_div(unsigned a, unsigned b)
{
    unsigned q = (unsigned)((float)a * __builtin_amdgcn_rcpf((float)b));
    int r = (int)a - (int)(q * b);
    if (r < 0) { --q; r += (int)b; }
    if (r >= (int)b) { ++q; }
    return q;
}
// first pixel coordinate of a level of size `n` whose centre lies in region `r` or beyond (regions of `reg` level-0 pixels,
// level-0 size n0): the smallest x with (2x + 1) * n0 >= 2 * reg * n * r, clipped to n.  tile_region_begin() on the host.
__device__ __forceinline__ int tl_region_begin(int r, int reg, int n, int n0)
{
    const unsigned v = 2u * (unsigned)reg * (unsigned)n * (unsigned)r;      // < 2^24: checked on the host
    const unsigned c = tl_div(v + (unsigned)n0 - 1u, (unsigned)n0);
    const int x = (int)(c >> 1);
    return x < n ? x : n;
}
static long long tile_region_begin(long long r, long long reg, long long n, long long n0)
{
    const long long c = (2 * reg * n * r + n0 - 1) / n0;
    const long long x = c >> 1;
    return x < n ? x : n;
}

// bf16 high parts (round to nearest even) and low parts of two fp32 weights, packed (a in the low half)
__device__ __forceinline__ void tl_split2(float a, float b, unsigned &hi, unsigned &lo)
{
    hi = __builtin_bit_cast(unsigned, tl_bf16x2{(__bf16)a, (__bf16)b});
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, tl_bf16x2{(__bf16)ra, (__bf16)rb});
}

__device__ __forceinline__ void tl_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// HM = false: value [B,S,H,D] (the reference operator's layout); HM = true: value [B,H,S,D] (head-major).
template <bool HM>
__global__ __launch_bounds__(kTlThreads, 4) void msda_fwd_tile_kernel(
    const uint16_t *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attn, const TileLevels lv,
    int S, int tiles_x, int tiles_y, int th, int nblk, int dbg, uint16_t *__restrict__ out)
{
    constexpr unsigned kGPixB = HM ? kTlPixB : (unsigned)(kTlHeads * kTlHeadDim * 2);   // global bytes from one pixel to the next
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int Nq = S;
    int LW[kTlLevels], LH[kTlLevels], LS[kTlLevels];
#pragma unroll
    for (int l = 0; l < kTlLevels; ++l) { LW[l] = lv.w[l]; LH[l] = lv.h[l]; LS[l] = lv.start[l]; }

    const int logical = xcd_contiguous_block(blockIdx.x, nblk);
    const int tpp = tiles_x * tiles_y;
    const int plane_i = (int)tl_div((unsigned)logical, (unsigned)tpp), tile = logical - plane_i * tpp;
    const int ty = __builtin_amdgcn_readfirstlane((int)tl_div((unsigned)tile, (unsigned)tiles_x));
    const int tx = __builtin_amdgcn_readfirstlane(tile - ty * tiles_x);
    const int b = __builtin_amdgcn_readfirstlane(plane_i >> 3), m = __builtin_amdgcn_readfirstlane(plane_i & 7);

    if (tid < 256) reinterpret_cast<unsigned *>(lds + kTlZeroOff)[tid] = 0u;      // published by the first pass's barrier

    // pixels of the coarser levels whose centres fall into the tile: [xa, xa + nx) x [ya, ya + ny)
    int XA[kTlLevels], YA[kTlLevels], NX[kTlLevels], NY[kTlLevels];
    XA[0] = tx * kTlTW; YA[0] = ty * th; NX[0] = kTlTW; NY[0] = th;
#pragma unroll
    for (int l = 1; l < kTlLevels; ++l) {
        const int xa = tl_region_begin(tx, kTlTW, LW[l], LW[0]), xb = tl_region_begin(tx + 1, kTlTW, LW[l], LW[0]);
        const int ya = tl_region_begin(ty, th, LH[l], LH[0]), yb = tl_region_begin(ty + 1, th, LH[l], LH[0]);
        XA[l] = __builtin_amdgcn_readfirstlane(xa);
        YA[l] = __builtin_amdgcn_readfirstlane(ya);
        NX[l] = __builtin_amdgcn_readfirstlane(xb - xa);
        NY[l] = __builtin_amdgcn_readfirstlane(yb - ya);
    }

    // the (image, head) value plane behind one wave-uniform buffer descriptor; byte offsets inside it are 32-bit
    const unsigned char *plane = reinterpret_cast<const unsigned char *>(value) +
                                 (HM ? ((size_t)b * kTlHeads + m) * (size_t)S * kTlPixB
                                     : (size_t)b * S * kGPixB + (size_t)m * kTlPixB);
    const unsigned plane_bytes = HM ? (unsigned)S * kTlPixB : (unsigned)S * kGPixB - (unsigned)m * kTlPixB;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(plane), 0, plane_bytes, 0x00020000);

    // set-up role: query qx of the group, point pp.   gather role: K-group g, corner tq / piece tp of a transposed read;
    // as an A-operand lane: row am = lane & 15 = 8 * (quad half ah) + 2 * (K-group ag) + (0 = bf16 high part, 1 = low part)
    const int qx = lane >> 2, pp = lane & 3;
    const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int am = lane & 15, ah = am >> 3, ag = (am >> 1) & 3, apart = am & 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;     // 0 in practice
    const unsigned wave_off = (unsigned)(kTlWaveOff + wave * kTlWaveBytes);
    unsigned char *const wreg = lds + wave_off;
    int *const fgo = reinterpret_cast<int *>(wreg + kTlFgo);
    const unsigned cell0 = __builtin_amdgcn_readfirstlane(lds0 + wave_off + (unsigned)kTlPatch);
    // One MFMA step = octet o', quad half h, point pair j: K-group g carries the two samples (points 2j, 2j + 1) of query
    // 8 o' + 4 h + g; its lane (corner tq, piece tp) reads 8 bytes of row (tq >> 1 ? bottom : top) + cd of each
    const unsigned cd = (unsigned)(tq & 1) * kTlPixB + (unsigned)tp * 8u;
    const unsigned o_rd = lds0 + wave_off + (unsigned)((tq >> 1) ? kTlStageOB : kTlStageOT) + (unsigned)g * 16u;   // + (8 o' + 4 h) * 16
    // A operand: lane (row am, K-group g) is live only in the steps of its own quad half and only if its row's query is the
    // K-group's -- then it reads that query's 2 x 4 weights (16 B); otherwise 16 B of zeros.  + o' * 512 + j * 16
    const unsigned w_real = lds0 + wave_off + (unsigned)kTlStageW + (unsigned)((4 * ah + g) * 64 + apart * 32);
    const unsigned w_rd0 = (ag == g && ah == 0) ? w_real : lds0 + (unsigned)kTlZeroKOff;
    const unsigned w_rd1 = (ag == g && ah == 1) ? w_real : lds0 + (unsigned)kTlZeroKOff;
    const unsigned par32 = (unsigned)(qx & 1) * 32u;              // odd queries read the other channel half first: the two
                                                                  // K-groups of a 32-lane half never share a bank group
    const unsigned o_zero = lds0 + (unsigned)kTlZeroOff + par32;  // the zero sample (its bottom corners: + 128)

    // ---- the wave's queries: slot = (wave + 8 gi) * 16 + qx; slots [0, 16 th) = the tile's level-0 pixels, row by row, then
    // the pixels of levels 1, 2, 3 in the tile ------------------------------------------------------------------------------
    auto query_of = [&](int gi) -> int {
        const int idx = (wave + kTlWaves * gi) * 16 + qx;
        const int n0 = kTlTW * th;
        if (idx < n0) {
            const int x = tx * kTlTW + (idx & 15), y = ty * th + (idx >> 4);
            return (x < LW[0] && y < LH[0]) ? LS[0] + y * LW[0] + x : -1;
        }
        int j = idx - n0, q = -1;
#pragma unroll
        for (int l = 1; l < kTlLevels; ++l) {
            const int n = NX[l] * NY[l];
            if (q < 0 && j >= 0 && j < n) {
                const int yy = (int)tl_div((unsigned)j, (unsigned)NX[l]);
                q = LS[l] + (YA[l] + yy) * LW[l] + XA[l] + (j - yy * NX[l]);
            }
            j -= n;
        }
        return q;
    };
    int q[kTlGroups];
    bool busy[kTlGroups];
#pragma unroll
    for (int gi = 0; gi < kTlGroups; ++gi) {
        q[gi] = query_of(gi);
        busy[gi] = __ballot(q[gi] >= 0) != 0ull;                  // uniform
    }

    // (location, soft-maxed weight) of this lane's point in level l: the reference operator's inputs
    struct LevelData { f32x2 xy; float a; };
    const float *loc_b = loc + (size_t)b * Nq * (kTlHeads * kTlLevels * kTlPoints * 2);
    const float *att_b = attn + (size_t)b * Nq * (kTlHeads * kTlLevels * kTlPoints);
    auto load_level = [&](int qq, int l) -> LevelData {
        LevelData d;
        if (dbg & 16) { d.xy = f32x2{0.5f, 0.5f}; d.a = 0.0625f; return d; }
        const unsigned e = ((unsigned)(qq >= 0 ? qq : 0) * kTlHeads + (unsigned)m) * (kTlLevels * kTlPoints) + (unsigned)(l * kTlPoints + pp);
        d.xy = *reinterpret_cast<const f32x2 *>(loc_b + 2u * e);
        d.a = att_b[e];
        return d;
    };

    struct Window { int wx0, wy0, rh; };
    // window of level l: 32 columns x (footprint + 2 * margin, at most the buffer's) rows centred on the tile's footprint and
    // kept inside the level's padded frame [-1, size].  A speed heuristic only: what a window misses is flagged and patched.
    auto window_of = [&](int l) -> Window {
        const int fx0 = XA[l], fx1 = fx0 + NX[l], fy0 = YA[l], fy1 = fy0 + NY[l];
        const int W = LW[l], H = LH[l];
        int rh = fy1 - fy0 + 2 * kTlMargin;
        rh = rh > kTlWinRows ? kTlWinRows : rh;
        rh = rh > H + 2 ? H + 2 : rh;
        int wx0 = (fx0 + fx1 - kTlWinW) >> 1, wy0 = (fy0 + fy1 - rh) >> 1;
        const int mx = W + 1 - kTlWinW, my = H + 1 - rh;      // last origin that still ends inside the padded frame
        wx0 = wx0 > mx ? mx : wx0; wx0 = wx0 < -1 ? -1 : wx0;
        wy0 = wy0 > my ? my : wy0; wy0 = wy0 < -1 ? -1 : wy0;
        return Window{wx0, wy0, rh};
    };

    // DMA the window of level l into the buffer.  Instruction i = row i >> 1, column half i & 1; the waves deal the instructions
    // round-robin.  Per instruction: scalar row base (soffset) and LDS destination (M0); the per-lane offset is one of two
    // constants of the fill (column * pixel pitch + 16-byte chunk, or "out of range" -> the lane writes zeros).  Inline
    // assembly on purpose: hipcc tracks the builtin as an LDS write and would drain it before the next LDS read.
    auto fill = [&](int l, const Window &wd) {
        const int W = LW[l], H = LH[l], st = LS[l];
        const int c0 = wd.wx0 + (lane >> 2), c1 = c0 + 16;                    // this lane's column in either half
        const unsigned chunk = (unsigned)(lane & 3) * 16u;
        const unsigned v0 = (c0 >= 0 && c0 < W) ? (unsigned)c0 * kGPixB + chunk : 0x80000000u;
        const unsigned v1 = (c1 >= 0 && c1 < W) ? (unsigned)c1 * kGPixB + chunk : 0x80000000u;
        const int n = 2 * wd.rh;
        for (int i = wave; i < n; i += kTlWaves) {                           // uniform
            const int r = i >> 1, j = i & 1;
            const int y = wd.wy0 + r;
            const bool rowok = y >= 0 && y < H;
            const unsigned soff = __builtin_amdgcn_readfirstlane(rowok ? (unsigned)(st + y * W) * kGPixB : 0u);
            const unsigned voff = rowok ? (j ? v1 : v0) : 0x80000000u;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)kTlBufOff + (unsigned)r * kTlPitchB + (unsigned)j * 1024u);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :
                         : "s"(m0v), "v"(voff), "s"(rsrc), "s"(soff)
                         : "memory", "m0");
        }
    };

    f32x4 acc[kTlGroups][2][2];            // [group][octet o'][X]: D rows 4g + r of a lane = query 8 o' + 2g + (r >> 1), part r & 1,
                                           // channel (lane & 15) + 16 ((r >> 1) ^ X)
#pragma unroll
    for (int gi = 0; gi < kTlGroups; ++gi)
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
            for (int X = 0; X < 2; ++X) acc[gi][op][X] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto lds_b128 = [](unsigned a) { return *(__attribute__((address_space(3))) const u32x4 *)a; };
    auto lds_tr = [](unsigned a) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tl_s16x4 *)a));
    };

    // ---- set-up of this lane's sample (query qx, point pp) in level l: msda_fwd.hip's arithmetic
    // (ms_deform_im2col_cuda.cuh:22-73, 274-277) ---------------------------------------------------------------------------
    struct Staged {
        unsigned ot, ob;                   // LDS addresses of the sample's top / bottom corner pair (+ par32)
        unsigned h01, h23, l01, l23;       // bf16 high / low parts of the four corner weights
        unsigned pk;                       // top-left pixel (y0 + 1) << 15 | (x0 + 1)
        bool live, pend;                   // live: goes through this round's MFMA steps; pend: flagged, rows not fetched yet
    };
    auto setup = [&](int l, const Window &wd, const LevelData &d, bool qok) -> Staged {
        const int W = LW[l], H = LH[l];
        const float x = d.xy.x * (float)W - 0.5f;
        const float y = d.xy.y * (float)H - 0.5f;
        const bool inside = qok && (y > -1.f) && (x > -1.f) && (y < (float)H) && (x < (float)W);      // false for NaN
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = inside ? (int)xf : 0, y0 = inside ? (int)yf : 0;       // in [-1, size - 1]
        const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
        // corners outside the level read zeros (window border / range-checked patch loads): no per-corner masks
        const float w00 = inside ? hy * hx * d.a : 0.f, w01 = inside ? hy * lx * d.a : 0.f;
        const float w10 = inside ? ly * hx * d.a : 0.f, w11 = inside ? ly * lx * d.a : 0.f;
        const int cx = x0 - wd.wx0, cy = y0 - wd.wy0;
        const bool in_win = (unsigned)cx < (unsigned)(kTlWinW - 1) && (unsigned)cy < (unsigned)(wd.rh - 1);
        Staged st;
        st.live = inside && in_win;
        st.pend = inside && !in_win;
        st.pk = ((unsigned)(y0 + 1) << 15) | (unsigned)(x0 + 1);             // levels up to 32766 pixels a side (host check)
        st.ot = lds0 + (unsigned)kTlBufOff + (unsigned)cy * kTlPitchB + (unsigned)cx * kTlPixB + par32;
        st.ob = st.ot + kTlPitchB;
        tl_split2(w00, w01, st.h01, st.l01);
        tl_split2(w10, w11, st.h23, st.l23);
        return st;
    };
    // top-left pixel `pk` -> this lane's byte offset in the value plane for corner (dx, dy), 16-byte chunk c
    // (corners outside the level: out of range -> the load returns zeros and makes no request)
    auto corner_offset = [&](int l, int pk, bool have, int dx, int dy, int c) -> unsigned {
        const int xx = (pk & 0x7fff) - 1 + dx, yy = (pk >> 15) - 1 + dy;
        const bool ok = have && (unsigned)xx < (unsigned)LW[l] && (unsigned)yy < (unsigned)LH[l];
        return ok ? (unsigned)(LS[l] + yy * LW[l] + xx) * kGPixB + (unsigned)c * 16u : 0x80000000u;
    };
    // one LDS-DMA instruction fetches the corner rows of up to four flagged samples into the wave's patch cells:
    // lane = (cell k = lane >> 4, corner (lane >> 2) & 3, 16-byte chunk lane & 3) -> LDS cell0 + 16 * lane
    auto patch_dma = [&](int l, int n_a, int n_b) {            // cells 0,1 <- fgo[0..n_a), cells 2,3 <- fgo[2..2+n_b)
        tl_wave_sync();
        const int k = lane >> 4, c = (lane >> 2) & 3;
        const bool have = k < 2 ? k < n_a : (k - 2) < n_b;
        const unsigned go = corner_offset(l, fgo[k], have, c & 1, c >> 1, lane & 3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                     :
                     : "s"(cell0), "v"(go), "s"(rsrc)
                     : "memory", "m0");
    };
    // the staged set-up of a group goes to the wave's LDS area (O, W); samples that are not live become the zero sample
    auto stage = [&](const Staged &st) {
        reinterpret_cast<unsigned *>(wreg + kTlStageOT)[qx * 4 + pp] = st.live ? st.ot : o_zero;
        reinterpret_cast<unsigned *>(wreg + kTlStageOB)[qx * 4 + pp] = st.live ? st.ob : o_zero + 128u;
        u32x2 *sw = reinterpret_cast<u32x2 *>(wreg + kTlStageW + qx * 64 + pp * 8);              // W[query][part][point][corner]
        sw[0] = st.live ? u32x2{st.h01, st.h23} : u32x2{0u, 0u};
        sw[4] = st.live ? u32x2{st.l01, st.l23} : u32x2{0u, 0u};
        tl_wave_sync();
    };

    // gather: the MFMA loop over the staged samples of one group -- per (octet, quad half) one 16-byte read brings the row
    // addresses of all four points, then two steps; the operands of step s + 1 are on their way while the MFMAs of step s run
    auto gather = [&](f32x4 (&ac)[2][2]) {
        struct Operands { u32x4 af; u32x2 x0, x1, y0, y1; };
        auto fetch = [&](unsigned wa, unsigned oa, unsigned ob) {
            Operands r;
            r.af = lds_b128(wa);
            r.x0 = lds_tr(oa); r.x1 = lds_tr(ob); r.y0 = lds_tr(oa ^ 32u); r.y1 = lds_tr(ob ^ 32u);
            return r;
        };
        auto fma2 = [&](const Operands &r, f32x4 &d0, f32x4 &d1) {
            const u32x4 b0 = {r.x0.x, r.x0.y, r.x1.x, r.x1.y}, b1 = {r.y0.x, r.y0.y, r.y1.x, r.y1.y};
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tl_bf16x8, r.af), __builtin_bit_cast(tl_bf16x8, b0), d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tl_bf16x8, r.af), __builtin_bit_cast(tl_bf16x8, b1), d1, 0, 0, 0);
        };
        const u32x4 so0 = lds_b128(o_rd), so1 = lds_b128(o_rd + 64u), so2 = lds_b128(o_rd + 128u), so3 = lds_b128(o_rd + 192u);
        // step (o', h, j): A operand at (h ? w_rd1 : w_rd0) + o' * 512 + j * 16; rows of points 2j, 2j + 1 of query 8 o' + 4 h + g
        Operands ra = fetch(w_rd0, so0.x + cd, so0.y + cd);
        Operands rb = fetch(w_rd0 + 16, so0.z + cd, so0.w + cd);
        fma2(ra, ac[0][0], ac[0][1]);
        ra = fetch(w_rd1, so1.x + cd, so1.y + cd);
        fma2(rb, ac[0][0], ac[0][1]);
        rb = fetch(w_rd1 + 16, so1.z + cd, so1.w + cd);
        fma2(ra, ac[0][0], ac[0][1]);
        ra = fetch(w_rd0 + 512, so2.x + cd, so2.y + cd);
        fma2(rb, ac[0][0], ac[0][1]);
        rb = fetch(w_rd0 + 528, so2.z + cd, so2.w + cd);
        fma2(ra, ac[1][0], ac[1][1]);
        ra = fetch(w_rd1 + 512, so3.x + cd, so3.y + cd);
        fma2(rb, ac[1][0], ac[1][1]);
        rb = fetch(w_rd1 + 528, so3.z + cd, so3.w + cd);
        fma2(ra, ac[1][0], ac[1][1]);
        fma2(rb, ac[1][0], ac[1][1]);
        tl_wave_sync();                    // the reads of the staging come before whatever overwrites it
    };

    // flagged samples beyond the first round's two per group (rare): four at a time through all four cells, one full round of
    // MFMA steps each, every other sample of the group the zero sample
    auto extra_rounds = [&](int l, Staged &st, f32x4 (&ac)[2][2]) {
        unsigned long long fm = __ballot(st.pend);
        while (fm != 0ull) {                                                 // uniform
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0));
            const int n = __builtin_popcountll(fm);
            const bool mine = st.pend && rank < 4;
            if (mine) fgo[rank] = (int)st.pk;
            patch_dma(l, n < 2 ? n : 2, n - 2 < 2 ? n - 2 : 2);
            st.live = mine;
            if (mine) {
                st.ot = cell0 + (unsigned)rank * 256u + par32;
                st.ob = st.ot + 128u;
                st.pend = false;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stage(st);
            gather(ac);
            fm = __ballot(st.pend);
        }
    };

    // out[query][channel] = D[hi row] + D[lo row]; transposed through the wave's W area, one octet at a time, so that a lane
    // stores 16 bytes
    auto store_group = [&](int sq, f32x4 (&ac)[2][2]) {
        float *tr = reinterpret_cast<float *>(wreg + kTlStageW);             // 1 KiB: 8 queries x 32 channels
#pragma unroll
        for (int op = 0; op < 2; ++op) {
#pragma unroll
            for (int X = 0; X < 2; ++X) {
                const f32x4 d = ac[op][X];
                tr[(2 * g) * 32 + (lane & 15) + 16 * X] = d.x + d.y;
                tr[(2 * g + 1) * 32 + (lane & 15) + 16 * (X ^ 1)] = d.z + d.w;
            }
            tl_wave_sync();
            // lane (qx, pp) stores channels 8 pp .. 8 pp + 7 of query qx: the octet's queries are qx = 8 op .. 8 op + 7
            const int ql = qx & 7;
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(tr + ql * 32 + pp * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(tr + ql * 32 + pp * 8 + 4);
            if (sq >= 0 && (qx >> 3) == op) {
                u32x4 w;
                w.x = pack_bf16x2(lo.x, lo.y);
                w.y = pack_bf16x2(lo.z, lo.w);
                w.z = pack_bf16x2(hi.x, hi.y);
                w.w = pack_bf16x2(hi.z, hi.w);
                *reinterpret_cast<u32x4 *>(out + ((size_t)b * Nq + sq) * (kTlHeads * kTlHeadDim) + m * kTlHeadDim + pp * 8) = w;
            }
            tl_wave_sync();
        }
    };

    // ---- the four passes --------------------------------------------------------------------------------------------------
    LevelData d[kTlGroups];
#pragma unroll
    for (int gi = 0; gi < kTlGroups; ++gi) d[gi] = load_level(q[gi], 0);

    auto one_pass = [&](auto lc) {
        constexpr int l = decltype(lc)::value;
        const Window wd = window_of(l);
        if (l > 0) {                                             // every wave is done reading the previous window
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(dbg & 32)) __builtin_amdgcn_s_barrier();
        }
        if (!(dbg & 1)) fill(l, wd);
        LevelData dn[kTlGroups];
        if (l + 1 < kTlLevels) {
#pragma unroll
            for (int gi = 0; gi < kTlGroups; ++gi) dn[gi] = load_level(q[gi], l + 1);
        }
        Staged st[kTlGroups];
#pragma unroll
        for (int gi = 0; gi < kTlGroups; ++gi) {
            if (dbg & 8) {
                st[gi] = Staged{o_zero, o_zero + 128u, __builtin_bit_cast(unsigned, d[gi].xy.x), 0u, __builtin_bit_cast(unsigned, d[gi].a), 0u, 0u, true, false};
            } else {
                st[gi] = setup(l, wd, d[gi], q[gi] >= 0);
            }
        }
        // first round: up to two flagged samples per group share ONE patch instruction
        const unsigned long long fm0 = __ballot(st[0].pend), fm1 = __ballot(st[1].pend);
        if ((fm0 | fm1) != 0ull) {                               // uniform
            const int r0 = __builtin_amdgcn_mbcnt_hi((unsigned)(fm0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm0, 0));
            const int r1 = __builtin_amdgcn_mbcnt_hi((unsigned)(fm1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm1, 0));
            const int n0 = __builtin_popcountll(fm0), n1 = __builtin_popcountll(fm1);
            const bool mine0 = st[0].pend && r0 < 2, mine1 = st[1].pend && r1 < 2;
            if (mine0) fgo[r0] = (int)st[0].pk;
            if (mine1) fgo[2 + r1] = (int)st[1].pk;
            patch_dma(l, n0 < 2 ? n0 : 2, n1 < 2 ? n1 : 2);
            if (mine0) { st[0].ot = cell0 + (unsigned)r0 * 256u + par32; st[0].ob = st[0].ot + 128u; st[0].live = true; st[0].pend = false; }
            if (mine1) { st[1].ot = cell0 + (unsigned)(2 + r1) * 256u + par32; st[1].ob = st[1].ot + 128u; st[1].live = true; st[1].pend = false; }
        }
        if (busy[0]) stage(st[0]);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // the window, the patch rows (and the next level's inputs)
        if (!(dbg & 32)) __builtin_amdgcn_s_barrier();
        if (!(dbg & 2)) {
            if (busy[0]) {
                gather(acc[0]);
                extra_rounds(l, st[0], acc[0]);
            }
            if (busy[1]) {
                stage(st[1]);
                gather(acc[1]);
                extra_rounds(l, st[1], acc[1]);
            }
        }
        if (l + 1 < kTlLevels) {
#pragma unroll
            for (int gi = 0; gi < kTlGroups; ++gi) d[gi] = dn[gi];
        }
    };
    one_pass(std::integral_constant<int, 0>{});
    one_pass(std::integral_constant<int, 1>{});
    one_pass(std::integral_constant<int, 2>{});
    one_pass(std::integral_constant<int, 3>{});
    if (!(dbg & 4)) {
#pragma unroll
        for (int gi = 0; gi < kTlGroups; ++gi)
            if (busy[gi]) store_group(q[gi], acc[gi]);
    }
}

// Tile height for a level table (HOST copy): the largest number of level-0 rows per tile, at most kTlMaxTH, such that no tile
// holds more than 256 queries, preferring the fewest tile rows and then the most even split.  0 = the table cannot be served.
static int tile_height(const int64_t *shapes, int *tiles_x_out, int *tiles_y_out)
{
    const long long H0 = shapes[0], W0 = shapes[1];
    const long long tiles_x = (W0 + kTlTW - 1) / kTlTW;
    int best = 0;
    long long best_ty = 0;
    for (int th = kTlMaxTH; th >= 1; --th) {
        const long long tiles_y = (H0 + th - 1) / th;
        if (best && tiles_y > best_ty) break;
        long long worst = 0;
        for (long long ty = 0; ty < tiles_y && worst <= kTlMaxQueries; ++ty)
            for (long long tx = 0; tx < tiles_x; ++tx) {
                long long n = (long long)kTlTW * th;
                for (int l = 1; l < kTlLevels; ++l) {
                    const long long nx = tile_region_begin(tx + 1, kTlTW, shapes[2 * l + 1], W0) - tile_region_begin(tx, kTlTW, shapes[2 * l + 1], W0);
                    const long long ny = tile_region_begin(ty + 1, th, shapes[2 * l], H0) - tile_region_begin(ty, th, shapes[2 * l], H0);
                    n += nx * ny;
                }
                worst = n > worst ? n : worst;
            }
        if (worst > kTlMaxQueries) continue;
        best = th;                                               // same (or the first) tile-row count, more even split
        best_ty = tiles_y;
    }
    *tiles_x_out = (int)tiles_x;
    *tiles_y_out = (int)best_ty;
    return best;
}

// Returns RDETR_ERR_UNSUPPORTED when the shape is not served (callers then use the direct kernel).  `shapes` / `level_start`
// are HOST pointers.
template <bool HM>
int msda_tile_forward(const uint16_t *value, const int64_t *shapes, const int64_t *level_start, const float *loc,
                      const float *attn, int B, int S, int L, int Nq, int dbg, uint16_t *out, hipStream_t stream)
{
    if (L != kTlLevels || Nq != S) return RDETR_ERR_UNSUPPORTED;
    if (!rdetr_msda_levels_window_ok(shapes, level_start, L, S)) return RDETR_ERR_UNSUPPORTED;
    const long long gpix = HM ? 64 : 512;
    if ((long long)S * gpix >= (1ll << 31)) return RDETR_ERR_UNSUPPORTED;
    TileLevels lv;
    for (int l = 0; l < kTlLevels; ++l) {
        const long long h = shapes[2 * l], w = shapes[2 * l + 1];
        if (h > 2896 || w > 2896) return RDETR_ERR_UNSUPPORTED;             // tile geometry: 2 * 16 * w * (w / 16 + 1) < 2^24
        lv.h[l] = (int)h; lv.w[l] = (int)w; lv.start[l] = (int)level_start[l];
    }
    int tiles_x = 0, tiles_y = 0;
    const int th = tile_height(shapes, &tiles_x, &tiles_y);
    if (th == 0) return RDETR_ERR_UNSUPPORTED;
    auto kern = msda_fwd_tile_kernel<HM>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kTlLdsBytes);
    if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
    const long long nblk = (long long)B * kTlHeads * tiles_x * tiles_y;
    if (nblk > 0xffffffll) return RDETR_ERR_UNSUPPORTED;                     // tl_div on the block id
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(kTlThreads), (size_t)kTlLdsBytes, stream, value, loc, attn, lv, S,
                       tiles_x, tiles_y, th, (int)nblk, dbg, out);
    return launch_status();
}

}  // namespace rdetr

#ifdef RDETR_DEV
static int g_tile_dbg = 0;
extern "C" void rdetr_dev_set_tile_dbg(int v) { g_tile_dbg = v; }
#define RDETR_TILE_DBG g_tile_dbg
#else
#define RDETR_TILE_DBG 0
#endif

extern "C" int rdetr_msda_forward_tile_bf16(const uint16_t *value, int value_layout, const int64_t *host_spatial_shapes,
                                            const int64_t *host_level_start_index, const float *sampling_loc,
                                            const float *attn_weight, int B, int S, int H, int D, int L, int Nq, int P,
                                            uint16_t *out, void *stream)
{
    if (B < 0 || S < 0 || Nq < 0 || H <= 0 || D <= 0 || L <= 0 || P <= 0) return RDETR_ERR_INVALID_ARG;
    if (value_layout != RDETR_VALUE_BSHD && value_layout != RDETR_VALUE_BHSD) return RDETR_ERR_INVALID_ARG;
    if (B == 0 || Nq == 0) return RDETR_OK;
    if (!value || !host_spatial_shapes || !host_level_start_index || !sampling_loc || !attn_weight || !out) return RDETR_ERR_INVALID_ARG;
    if (S == 0) return RDETR_ERR_INVALID_ARG;
    if (H != rdetr::kTlHeads || D != rdetr::kTlHeadDim || P != rdetr::kTlPoints) return RDETR_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(value) % 16 || reinterpret_cast<uintptr_t>(out) % 16 ||
        reinterpret_cast<uintptr_t>(sampling_loc) % 8 || reinterpret_cast<uintptr_t>(attn_weight) % 4)
        return RDETR_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return value_layout == RDETR_VALUE_BHSD
               ? rdetr::msda_tile_forward<true>(value, host_spatial_shapes, host_level_start_index, sampling_loc, attn_weight, B, S,
                                                L, Nq, RDETR_TILE_DBG, out, s)
               : rdetr::msda_tile_forward<false>(value, host_spatial_shapes, host_level_start_index, sampling_loc, attn_weight, B, S,
                                                 L, Nq, RDETR_TILE_DBG, out, s);
}
