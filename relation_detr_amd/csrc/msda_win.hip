// Multi-scale deformable attention, forward -- LDS-window MFMA kernel for the ENCODER shape (queries = the pyramid's own
// pixels, Nq == S, L == 4, bf16 value) on gfx950 (MI355X).  Second generation of the round-1 tile kernel.
//
// Why a window kernel at all: a gather through the texture path delivers ~30 B/clk/CU whatever the layout (measured:
// tools/microbench/l1_gather_bw.hip, 8 x 128-byte and 16 x 64-byte segments per instruction alike, L1 hits included),
// and one launch at BASELINE.json configs[1] gathers 2.93 GB of corner rows: >= 125 us.  LDS serves the same rows at
// 3.5x that rate, and a spatial tile of queries re-uses every row it stages about six times.
//
//   tile      a 16 x 12 pixel region of level 0 plus the pixels of the coarser levels whose centres fall into it
//             (<= 192 + 64 queries); waves 0..11 own the region's rows, waves 12..15 its coarser pixels.
//   windows   per tile and sampled level a 32-pixel-wide window of the (image, head) value plane around the tile's
//             footprint in that level (28 rows for levels 0 / 1, 19 for levels 2 / 3), IN PADDED COORDINATES: the window
//             may start at pixel -1 and end at pixel W, and whatever lies outside the level arrives as zeros -- the
//             LDS-DMA (buffer_load_dwordx4 ... lds) is range checked per lane, out-of-range lanes write zeros
//             (tools/microbench/dma_oob_check.hip).  So a sample has ONE LDS offset (its top-left corner); the other
//             three corners are the constants +64, +pitch, +pitch+64, no corner is ever clamped or masked, and the
//             zero padding of ms_deform_im2col_cuda.cuh:44-67 is literally in the data.
//   fills     one DMA instruction = 16 pixels of one window row.  Row base and LDS destination are scalar arithmetic
//             (soffset / M0), the per-lane part is one of two constants per fill -- no vector arithmetic per
//             instruction.  The LDS row pitch is 34 pixels (== 128 B mod 256) so that the corners of a sample fall on
//             four different bank groups.  Value layout: [B,S,H,D] (the reference operator's, 64-byte pieces at a
//             512-byte pitch) or [B,H,S,D] (head-major: what rdetr_value_to_head_major_bf16 / the value projection of
//             the module path write; contiguous rows, 1.6x the fill rate).
//   math      the bilinear gather-and-weighted-sum runs on the matrix cores, v_mfma_f32_16x16x32_bf16 with
//             K = 8 samples x 4 corners: B[k][n] = value row k, channel n, read from the window by ds_read_b64_tr_b16
//             (the transposed read takes a PER-LANE row address: the 32 rows of an operand are the gathered rows
//             themselves); A[m][k] = corner weights, block diagonal, rows carrying the bf16 HIGH and LOW parts of the
//             fp32 weights (w = hi + lo to 2^-17, products exact, fp32 accumulate); D rows of one lane add up to a
//             query's channels.  One MFMA serves 8 samples x 16 channels.
//   flagged   nothing is assumed about the sampling locations.  A sample whose corners are not all inside its window
//             gets its four rows fetched by range-checked LDS-DMA into a patch cell laid out like a piece of window
//             (same corner constants: the patch cells form a pseudo-window of their own) while the previous pass
//             runs, so the MFMA loop is the same 8 steps whatever the locations; more than four such samples per
//             wave and level take extra steps (a decoder-like scatter merely runs slowly).  The result never depends
//             on the windows.
//
//   workgroup = 1024 threads = 16 waves (one 16-query group each; a wave issues one instruction per ~5 clocks whatever
//               its instruction-level parallelism, so it takes four waves per SIMD to keep the vector ALU busy), persistent
//               over a contiguous range of the tiles of ONE (image, head).
//   passes    = one per sampled level, order L0 (A), L2 (B), L1 (A), L3 (B): the two window buffers alternate, the fill of
//               the next pass is issued at the top of the current one; one barrier per pass.  A pass GATHERS its level
//               (LDS reads + MFMA) and SETS UP the next pass (VALU; results kept in registers, staged to LDS at the end):
//               the two are independent, and the two waves of a SIMD run them in opposite order, so that the LDS and
//               the vector ALU work at the same time (in lockstep they took turns: set-up 27 us + gather 46 us of a
//               142-us kernel, nothing overlapping).
// Per corner the arithmetic is msda_fwd.hip's (same weights); the summation order differs and each weight carries a
// 2^-17 relative representation error (the bf16 output rounds at 2^-9).
#include <type_traits>

#include "common.h"

namespace rdetr {

typedef __bf16 wn_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wn_bf16x2 __attribute__((ext_vector_type(2)));
typedef short wn_s16x4 __attribute__((ext_vector_type(4)));

constexpr int kWnThreads = 1024;
constexpr int kWnWaves = kWnThreads / kWave;                  // 16 = groups of 16 queries per tile, one per wave
constexpr int kWnRegW = 16, kWnRegH = 12;                     // level-0 pixels of a spatial tile: one row per wave 0..11
constexpr int kWnCoarseWave0 = kWnRegH;                       // waves 12..15: the region's coarser-level queries
constexpr int kWnCoarseSlots = (kWnWaves - kWnCoarseWave0) * 16;
constexpr int kWnHeads = 8, kWnHeadDim = 32, kWnPoints = 4, kWnLevels = 4;
constexpr unsigned kWnPixB = 64;                              // LDS bytes per pixel (one bf16 head row)
constexpr int kWnWinW = 32;                                   // window width in pixels = two DMA instructions per row
constexpr unsigned kWnPitchB = (kWnWinW + 2) * kWnPixB;       // 2176 B: == 128 (mod 256)
constexpr int kWnRowsA = 27, kWnRowsB = 18;                   // window rows: buffer A (levels 0 / 1), buffer B (levels 2 / 3)
constexpr int kWnMargin = 8;                                  // rows / columns of margin around a footprint
constexpr int kWnRing = 16;                                   // tiles whose geometry / window tables are kept (ring)

// ---- LDS map ------------------------------------------------------------------------------------------------------
constexpr int kWnFgoOff = 512;                                // 16 waves x 64 B: [0..3] pixels of the flagged samples in flight, [8] overflow flag
constexpr int kWnGeoOff = 1536;                               // int geo[kWnRing][20]
constexpr int kWnDescOff = kWnGeoOff + kWnRing * 80;          // int desc[kWnRing][4 levels][4]
constexpr int kWnZeroOff = 4096;                              // 1 KiB of zeros: idle A-operand lanes; its first 128 B = the
                                                              // top corners of the "zero sample", whose bottom corners are
constexpr int kWnZeroBotOff = kWnZeroOff + (int)kWnPitchB;    // 128 B of zeros one window pitch further on
constexpr int kWnZeroKOff = kWnZeroOff + 128 + 32;            // what idle A lanes read; == 32 (mod 64) keeps it off the live lanes' banks
constexpr int kWnMiscBytes = kWnZeroBotOff + 128;             // 6400
static_assert(kWnDescOff + kWnRing * 64 <= kWnZeroOff, "tables overlap the zero block");
// per-wave staging
constexpr int kWnWaveOff = kWnMiscBytes;
constexpr int kWnStageW = 0;                                  // [0, 1024)    W[query][part][point][corner] bf16
constexpr int kWnStageO = 1024;                               // [1024, 1280) O[query][point] u32: LDS offset of the sample's top-left corner
constexpr int kWnStageV = 1280;                               // [1280, 1536) overflow list: per set-up lane, bit 31 | top-left pixel, or 0
constexpr int kWnWaveBytes = 1536;
// patch cells: the rows of the flagged samples, WINDOW-SHAPED -- a sample's top corners side by side (128 B), its bottom
// corners one window pitch further on -- so that window, patch and zero samples share the corner constants.  A cell =
// four samples = 512 B of one row + 512 B of the next; four cells per row pair, two cells (passes p, p + 1) per wave
constexpr int kWnPatchOff = kWnWaveOff + kWnWaves * kWnWaveBytes;
constexpr int kWnPatchBytes = (2 * kWnWaves / 4) * 2 * (int)kWnPitchB;
constexpr int kWnBufAOff = kWnPatchOff + kWnPatchBytes;
constexpr int kWnBufBOff = kWnBufAOff + kWnRowsA * (int)kWnPitchB;
constexpr int kWnLdsBytes = kWnBufBOff + kWnRowsB * (int)kWnPitchB;
static_assert(kWnLdsBytes <= 160 * 1024, "LDS map exceeds 160 KiB");
static_assert(kWnPatchOff % 256 == 0 && kWnBufAOff % 256 == 0, "window-shaped areas must keep the bank phase of the pitch");

struct WinShared {
    int h[kWnLevels], w[kWnLevels], start[kWnLevels];
    int regions_x, regions_y, chunks;      // spatial tiles of level 0; coarse-query chunks per region (1 unless > 64 coarse)
    int max_coarse;
};
static_assert(sizeof(WinShared) <= kWnFgoOff, "tables overlap");
// ring tables (kWnGeoOff / kWnDescOff), entry = tile & (kWnRing - 1):
//   geo[20]      rx, ry, chunk, -, then per coarser level: xa, ya, nx, n, 2^16 / nx
//   desc[4][4]   per level: window origin x, y in pixel coordinates (>= -1), rows, -

__device__ __forceinline__ float wn_quad_max(float v)
{
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));
    return v;
}
__device__ __forceinline__ float wn_quad_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    return v;
}
// value of lane (quad, L) for every lane of the quad (DPP quad_perm broadcast)
template <int L> __device__ __forceinline__ float wn_quad_bcast(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), L * 0x55, 0xf, 0xf, false));
}

// a / b for a < 2^24, 0 < b < 2^24 without the integer division sequence: the float quotient is within one of the exact
// one, two compare-and-adjust steps fix it
__device__ __forceinline__ unsigned wn_div(unsigned a, unsigned b)
{
    unsigned q = (unsigned)((float)a * __builtin_amdgcn_rcpf((float)b));
    int r = (int)a - (int)(q * b);
    if (r < 0) { --q; r += (int)b; }
    if (r >= (int)b) { ++q; }
    return q;
}

// first pixel coordinate of a level of size `n` whose centre lies in region `r` or beyond (regions of `reg` level-0
// pixels, level-0 size n0): the smallest x with (2x + 1) * n0 >= 2 * reg * n * r, clipped to n
__device__ __forceinline__ int wn_region_begin(int r, int reg, int n, int n0)
{
    const unsigned v = 2u * (unsigned)reg * (unsigned)n * (unsigned)r;      // < 2^24: checked on the host
    const unsigned c = wn_div(v + (unsigned)n0 - 1u, (unsigned)n0);
    const int x = (int)(c >> 1);
    return x < n ? x : n;
}

// bf16 high parts (round to nearest even) and low parts of two fp32 weights, packed (a in the low half):
// w = hi + lo up to 2^-17 |w|
__device__ __forceinline__ void wn_split2(float a, float b, unsigned &hi, unsigned &lo)
{
    hi = __builtin_bit_cast(unsigned, wn_bf16x2{(__bf16)a, (__bf16)b});
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, wn_bf16x2{(__bf16)ra, (__bf16)rb});
}

// A copy of `v` the optimiser cannot see through: address arithmetic derived from it is redone where it is used instead of
// being hoisted out of the tile loop as yet another long-lived register (the kernel runs at the 128-register limit of a
// 1024-thread workgroup; a spilled loop invariant costs a memory round trip per tile)
__device__ __forceinline__ int wn_opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// retire this wave's LDS-DMA before the barrier that publishes the buffers (the compiler does not know about it)
__device__ __forceinline__ void wn_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ... when the wave's two youngest vector-memory operations are the tile's output stores (vmcnt counts in issue order, stores
// included): the DMA is retired, the stores stay in flight across the barrier instead of holding the whole workgroup for
// their round trip to memory
__device__ __forceinline__ void wn_dma_wait_keep2() { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }

// HM = false: value [B,S,H,D] (pixel-major, the reference operator's layout); HM = true: value [B,H,S,D] (head-major).
template <bool FUSED, bool HM>
__global__ __launch_bounds__(kWnThreads) void msda_fwd_win_kernel(
    const uint16_t *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const void *__restrict__ src_a, const void *__restrict__ src_b, const float *__restrict__ ref, int ref_dim, int S,
    int splits, int nblk, int dbg_arg, int ld_a, int ld_b, uint16_t *__restrict__ out)
{
#ifdef RDETR_DEV
    const int dbg = dbg_arg;                 // development builds: component-timing mask (tools/win_components.py)
#else
    constexpr int dbg = 0;
#endif
    constexpr unsigned kGPixB = HM ? kWnPixB : (unsigned)(kWnHeads * kWnHeadDim * 2);   // global bytes from one pixel to the next
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    WinShared &sh = *reinterpret_cast<WinShared *>(lds);
    int *const geo_tab = reinterpret_cast<int *>(lds + kWnGeoOff);
    int *const desc_tab = reinterpret_cast<int *>(lds + kWnDescOff);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int Nq = S;

    if (tid == 0) {
        for (int l = 0; l < kWnLevels; ++l) {
            sh.h[l] = (int)shapes[2 * l];
            sh.w[l] = (int)shapes[2 * l + 1];
            sh.start[l] = (int)level_start[l];
        }
        sh.regions_x = (sh.w[0] + kWnRegW - 1) / kWnRegW;
        sh.regions_y = (sh.h[0] + kWnRegH - 1) / kWnRegH;
        sh.max_coarse = 0;
    }
    if (tid < 256) reinterpret_cast<unsigned *>(lds + kWnZeroOff)[tid] = 0u;
    if (tid < 32) reinterpret_cast<unsigned *>(lds + kWnZeroBotOff)[tid] = 0u;
    __syncthreads();

    // coarser-level pixels per region: level l contributes [xa, xb) x [ya, yb), the pixels whose centres fall in the region
    auto coarse_count = [&](int rx, int ry) {
        int n = 0;
#pragma unroll
        for (int l = 1; l < kWnLevels; ++l) {
            const int nx = wn_region_begin(rx + 1, kWnRegW, sh.w[l], sh.w[0]) - wn_region_begin(rx, kWnRegW, sh.w[l], sh.w[0]);
            const int ny = wn_region_begin(ry + 1, kWnRegH, sh.h[l], sh.h[0]) - wn_region_begin(ry, kWnRegH, sh.h[l], sh.h[0]);
            n += nx * ny;
        }
        return n;
    };
    {
        const int nreg = sh.regions_x * sh.regions_y;
        int mx = 0;
        for (int r = tid; r < nreg; r += kWnThreads) {
            const int ry = r / sh.regions_x;
            const int n = coarse_count(r - ry * sh.regions_x, ry);
            mx = n > mx ? n : mx;
        }
        if (mx > 0) atomicMax(&sh.max_coarse, mx);
    }
    __syncthreads();
    if (tid == 0) sh.chunks = sh.max_coarse <= kWnCoarseSlots ? 1 : (sh.max_coarse + kWnCoarseSlots - 1) / kWnCoarseSlots;
    __syncthreads();

    // the tables are wave-uniform: keep them in SGPRs
    int LW[kWnLevels], LH[kWnLevels], LS[kWnLevels];
#pragma unroll
    for (int l = 0; l < kWnLevels; ++l) {
        LW[l] = __builtin_amdgcn_readfirstlane(sh.w[l]);
        LH[l] = __builtin_amdgcn_readfirstlane(sh.h[l]);
        LS[l] = __builtin_amdgcn_readfirstlane(sh.start[l]);
    }
    const int regions_x = __builtin_amdgcn_readfirstlane(sh.regions_x);
    const int regions_y = __builtin_amdgcn_readfirstlane(sh.regions_y);

    const int logical = xcd_contiguous_block(blockIdx.x, nblk);
    const int pair = logical / splits, split = logical - pair * splits;
    const int b = pair / kWnHeads, m = pair - b * kWnHeads;
    const int chunks = __builtin_amdgcn_readfirstlane(sh.chunks);
    const int ntiles = regions_x * regions_y * chunks;
    const int t0 = (int)((long long)split * ntiles / splits), t1 = (int)((long long)(split + 1) * ntiles / splits);
    if (t0 >= t1) return;                                   // uniform for the workgroup

    // the (image, head) value plane behind one wave-uniform buffer descriptor; byte offsets inside it are 32-bit
    const unsigned char *plane = reinterpret_cast<const unsigned char *>(value) +
                                 (HM ? ((size_t)b * kWnHeads + m) * (size_t)S * kWnPixB
                                     : (size_t)b * S * kGPixB + (size_t)m * kWnPixB);
    const unsigned plane_bytes = HM ? (unsigned)S * kWnPixB : (unsigned)S * kGPixB - (unsigned)m * kWnPixB;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(plane), 0, plane_bytes, 0x00020000);

    // set-up role: query qx of the wave, point pp.   gather role: K-group g, corner tq / piece tp of a transposed read;
    // as an A-operand lane: row am = lane & 15 = 8 * (quad half ah) + 2 * (K-group ag) + (0 = bf16 high part, 1 = low part)
    const int qx = lane >> 2, pp = lane & 3;
    const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int am = lane & 15, ah = am >> 3, ag = (am >> 1) & 3, apart = am & 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;     // 0 in practice
    const unsigned wave_off = (unsigned)(kWnWaveOff + wave * kWnWaveBytes);
    unsigned char *wreg = lds + wave_off;
    int *fgo = reinterpret_cast<int *>(lds + kWnFgoOff + wave * 64);
    // patch cell of this wave for pass parity pb: its four samples' top corners; the bottom corners sit one pitch further on
    auto cell_off = [&](int pb) {
        const int c = 2 * wave + pb;
        return (unsigned)(kWnPatchOff + (c >> 2) * 2 * (int)kWnPitchB + (c & 3) * 512);
    };
    // One MFMA step = octet o', quad half h, point pair j: K-group g carries the two samples (points 2j, 2j + 1) of
    // query 8 o' + 4 h + g; its lane (corner tq, piece tp) reads row `top-left + cd` of each -- window, patch and zero
    // samples alike
    const unsigned cd = (unsigned)(tq & 1) * kWnPixB + (unsigned)(tq >> 1) * kWnPitchB + (unsigned)tp * 8u;
    const unsigned o_rd = lds0 + wave_off + (unsigned)kWnStageO + (unsigned)g * 16u;                   // + (8 o' + 4 h) * 16
    // A operand: lane (row am, K-group g) is live only in the steps of its own quad half and only if its row's query is the
    // K-group's -- then it reads that query's 2 x 4 weights (16 B); otherwise 16 B of zeros.  + o' * 512 + j * 16
    const unsigned w_real = lds0 + wave_off + (unsigned)kWnStageW + (unsigned)((4 * ah + g) * 64 + apart * 32);
    const unsigned w_rd0 = (ag == g && ah == 0) ? w_real : lds0 + (unsigned)kWnZeroKOff;
    const unsigned w_rd1 = (ag == g && ah == 1) ? w_real : lds0 + (unsigned)kWnZeroKOff;
    const unsigned par32 = (unsigned)(qx & 1) * 32u;              // odd queries read the other channel half first: the two
                                                                  // K-groups of a 32-lane half never share a bank group
    const unsigned o_zero = lds0 + (unsigned)kWnZeroOff + par32;  // the zero sample

    // ---- helpers -----------------------------------------------------------------------------------------------
    // Tile geometry (which region, and which pixels of the coarser levels have their centres in it) costs a dozen integer
    // divisions: ONE lane per value computes it, once per tile, into the ring; every wave then reads it as scalars.
    auto compute_geometry = [&](int t, int k) {        // k = lane of the computing group, 0 .. 15
        const int region = (int)wn_div((unsigned)t, (unsigned)chunks), chunk = t - region * chunks;
        const int ry = (int)wn_div((unsigned)region, (unsigned)regions_x), rx = region - ry * regions_x;
        int *geo = geo_tab + (t & (kWnRing - 1)) * 20;
        if (k == 0) { geo[0] = rx; geo[1] = ry; geo[2] = chunk; }
        if (k >= 1 && k < kWnLevels) {
            const int l = k;
            const int xa = wn_region_begin(rx, kWnRegW, LW[l], LW[0]), xb = wn_region_begin(rx + 1, kWnRegW, LW[l], LW[0]);
            const int ya = wn_region_begin(ry, kWnRegH, LH[l], LH[0]), yb = wn_region_begin(ry + 1, kWnRegH, LH[l], LH[0]);
            const int nx = xb - xa;
            geo[4 + 5 * (l - 1) + 0] = xa;
            geo[4 + 5 * (l - 1) + 1] = ya;
            geo[4 + 5 * (l - 1) + 2] = nx;
            geo[4 + 5 * (l - 1) + 3] = nx * (yb - ya);
            geo[4 + 5 * (l - 1) + 4] = nx > 0 ? (int)wn_div(65536u + (unsigned)nx - 1u, (unsigned)nx) : 0;   // j / nx for j < 2^16 / nx
        }
    };

    // query owned by lane >> 2 of this wave in tile t (-1 = none)
    auto query_of = [&](int t) -> int {
        const int qx = wn_opaque(lane) >> 2;
        const int *geo = geo_tab + (t & (kWnRing - 1)) * 20;
        const int rx = __builtin_amdgcn_readfirstlane(geo[0]), ry = __builtin_amdgcn_readfirstlane(geo[1]);
        const int chunk = __builtin_amdgcn_readfirstlane(geo[2]);
        if (wave < kWnCoarseWave0) {
            const int x = rx * kWnRegW + qx, y = ry * kWnRegH + wave;
            return (chunk == 0 && x < LW[0] && y < LH[0]) ? LS[0] + y * LW[0] + x : -1;
        }
        int j = chunk * kWnCoarseSlots + (wave - kWnCoarseWave0) * 16 + qx;
        int q = -1;
#pragma unroll
        for (int l = 1; l < kWnLevels; ++l) {
            const int xa = __builtin_amdgcn_readfirstlane(geo[4 + 5 * (l - 1) + 0]);
            const int ya = __builtin_amdgcn_readfirstlane(geo[4 + 5 * (l - 1) + 1]);
            const int nx = __builtin_amdgcn_readfirstlane(geo[4 + 5 * (l - 1) + 2]);
            const int n = __builtin_amdgcn_readfirstlane(geo[4 + 5 * (l - 1) + 3]);
            const unsigned inv = (unsigned)__builtin_amdgcn_readfirstlane(geo[4 + 5 * (l - 1) + 4]);
            if (q < 0 && j >= 0 && j < n) {
                const int yy = (int)(((unsigned)j * inv) >> 16);
                q = LS[l] + (ya + yy) * LW[l] + xa + (j - yy * nx);
            }
            j -= n;
        }
        return q;
    };

    // ---- sample data: loaded ONE LEVEL AT A TIME, one pass before the set-up that consumes it --------------------------------
    // !FUSED: (location, soft-maxed weight) of this lane's point in level l (the reference operator's inputs).
    // FUSED : raw sampling offset of the point (packed bf16 x, y); the logits of all levels and the reference points are loaded
    //         once per tile (tile_inputs), soft-maxed / applied in the set-up.
    struct LevelData { f32x2 xy; float a; };               // FUSED: xy.x = the packed raw offsets (bits), a unused
    auto load_level = [&](int q, int l) -> LevelData {
        LevelData d;
        const int pp = wn_opaque(lane) & 3;
        const unsigned qq = (unsigned)(q >= 0 ? q : 0);
        if constexpr (FUSED) {
            const size_t row = (size_t)b * Nq + qq;
            const uint16_t *off_q = static_cast<const uint16_t *>(src_a) +
                                    (ld_a ? row * (size_t)ld_a + (size_t)m * (kWnLevels * kWnPoints * 2)
                                          : (row * kWnHeads + m) * (size_t)(kWnLevels * kWnPoints * 2));
            d.xy.x = __builtin_bit_cast(float, *reinterpret_cast<const unsigned *>(off_q + 2 * (l * kWnPoints + pp)));
            d.xy.y = 0.f;
            d.a = 0.f;
        } else {
            // 32-bit element offsets from the image's (uniform) base: B * Nq * 8 * 16 * 2 floats can exceed 2^32, one image cannot
            const float *loc_b = static_cast<const float *>(src_a) + (size_t)b * Nq * (kWnHeads * kWnLevels * kWnPoints * 2);
            const float *att_b = static_cast<const float *>(src_b) + (size_t)b * Nq * (kWnHeads * kWnLevels * kWnPoints);
            const unsigned e = (qq * kWnHeads + (unsigned)m) * (kWnLevels * kWnPoints) + (unsigned)(l * kWnPoints + pp);
            d.xy = *reinterpret_cast<const f32x2 *>(loc_b + 2u * e);
            d.a = att_b[e];
        }
        return d;
    };
    struct TileInputs { float lg[kWnLevels]; f32x4 rf; };  // FUSED: raw logits of the lane's point in each level; reference point of level `pp`
    auto load_tile_inputs = [&](int q) -> TileInputs {
        TileInputs ti;
        const int pp = wn_opaque(lane) & 3;
#pragma unroll
        for (int l = 0; l < kWnLevels; ++l) ti.lg[l] = 0.f;
        ti.rf = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (FUSED) {
            const size_t row = (size_t)b * Nq + (unsigned)(q >= 0 ? q : 0);
            const uint16_t *lg_q = static_cast<const uint16_t *>(src_b) +
                                   (ld_b ? row * (size_t)ld_b + (size_t)m * (kWnLevels * kWnPoints)
                                         : (row * kWnHeads + m) * (size_t)(kWnLevels * kWnPoints));
#pragma unroll
            for (int l = 0; l < kWnLevels; ++l) ti.lg[l] = bf16_bits_to_f32(lg_q[l * kWnPoints + pp]);
            const float *rp = ref + (row * kWnLevels + pp) * (size_t)ref_dim;        // lane pp: the reference point of level pp
            if (ref_dim == 2) {
                const f32x2 r2 = *reinterpret_cast<const f32x2 *>(rp);
                ti.rf = f32x4{r2.x, r2.y, 0.f, 0.f};
            } else {
                ti.rf = *reinterpret_cast<const f32x4 *>(rp);
            }
        }
        return ti;
    };
    // FUSED: softmax over the L * P logits of a (query, head) (ms_deform_attn.py:326-336) -> the four weights of this lane's point
    auto softmax_levels = [&](const TileInputs &ti, float (&aw)[kWnLevels]) {
        float mx = fmaxf(fmaxf(ti.lg[0], ti.lg[1]), fmaxf(ti.lg[2], ti.lg[3]));
        mx = wn_quad_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int l = 0; l < kWnLevels; ++l) {
            aw[l] = expf(ti.lg[l] - mx);
            sum += aw[l];
        }
        sum = wn_quad_sum(sum);
#pragma unroll
        for (int l = 0; l < kWnLevels; ++l) aw[l] = aw[l] / sum;
    };
    // FUSED: loc = ref + off / (W, H)  |  ref_xy + off / P * ref_wh * 0.5  (ms_deform_attn.py:339-349), the reference's operation order
    auto fused_location = [&](auto lc, float packed, const f32x4 &rf) -> f32x2 {
        constexpr int l = decltype(lc)::value;
        const unsigned u = __builtin_bit_cast(unsigned, packed);
        const float ox = __builtin_bit_cast(float, u << 16), oy = __builtin_bit_cast(float, u & 0xffff0000u);
        const float rx = wn_quad_bcast<l>(rf.x), ry = wn_quad_bcast<l>(rf.y);
        if (ref_dim == 2) return f32x2{rx + ox / (float)LW[l], ry + oy / (float)LH[l]};
        const float rw = wn_quad_bcast<l>(rf.z), rh = wn_quad_bcast<l>(rf.w);
        return f32x2{rx + ox * (1.0f / kWnPoints) * rw * 0.5f, ry + oy * (1.0f / kWnPoints) * rh * 0.5f};
    };

    // thread l: the window of level l for tile t -- 32 columns x (footprint + 2 * margin, at most the buffer's) rows centred
    // on the tile's footprint in that level and kept inside the level's padded frame [-1, size].  A speed heuristic only:
    // whatever a window misses is flagged and patched.
    auto window_desc = [&](int l, int t) {
        const int *geo = geo_tab + (t & (kWnRing - 1)) * 20;
        int fx0, fx1, fy0, fy1;                        // footprint [fx0, fx1) x [fy0, fy1)
        if (l == 0) {
            fx0 = geo[0] * kWnRegW; fx1 = fx0 + kWnRegW;
            fy0 = geo[1] * kWnRegH; fy1 = fy0 + kWnRegH;
        } else {
            const int nx = geo[4 + 5 * (l - 1) + 2], n = geo[4 + 5 * (l - 1) + 3];
            fx0 = geo[4 + 5 * (l - 1)]; fx1 = fx0 + nx;
            fy0 = geo[4 + 5 * (l - 1) + 1]; fy1 = fy0 + (nx > 0 ? (int)wn_div((unsigned)n, (unsigned)nx) : 0);
        }
        const int W = LW[l], H = LH[l];
        const int cap = l < 2 ? kWnRowsA : kWnRowsB;
        int rh = fy1 - fy0 + 2 * kWnMargin;
        rh = rh > cap ? cap : rh;
        rh = rh > H + 2 ? H + 2 : rh;
        int wx0 = (fx0 + fx1 - kWnWinW) >> 1, wy0 = (fy0 + fy1 - rh) >> 1;
        const int mx = W + 1 - kWnWinW, my = H + 1 - rh;      // last origin that still ends inside the padded frame
        wx0 = wx0 > mx ? mx : wx0; wx0 = wx0 < -1 ? -1 : wx0;
        wy0 = wy0 > my ? my : wy0; wy0 = wy0 < -1 ? -1 : wy0;
        int *d = desc_tab + (t & (kWnRing - 1)) * 16 + l * 4;
        d[0] = wx0;
        d[1] = wy0;
        d[2] = rh;
        d[3] = 0;
    };

    // DMA the window of level l of tile t into its buffer.  Instruction i of the fill = row i >> 1, column half i & 1; the
    // waves deal the instructions round-robin.  Per instruction: scalar row base (soffset) and LDS destination (M0); the
    // per-lane offset is one of two constants of the fill (column * pixel pitch + 16-byte chunk, or "out of range").
    // Issued through inline assembly ON PURPOSE: hipcc tracks the builtin as an LDS write and drains it (s_waitcnt
    // vmcnt(0)) before the pass's first LDS read, which would serialise fill and gather; untracked, it stays in flight
    // behind the pass and is retired by wn_dma_wait() before the barrier that hands the buffer over.
    auto fill = [&](int l, int t) {
        const int *dsc = desc_tab + (t & (kWnRing - 1)) * 16 + l * 4;
        const int wx0 = __builtin_amdgcn_readfirstlane(dsc[0]);
        const int wy0 = __builtin_amdgcn_readfirstlane(dsc[1]);
        const int rh = __builtin_amdgcn_readfirstlane(dsc[2]);
        const int W = LW[l], H = LH[l], st = LS[l];
        const unsigned buf = l < 2 ? (unsigned)kWnBufAOff : (unsigned)kWnBufBOff;
        const int ln = wn_opaque(lane);
        const int c0 = wx0 + (ln >> 2), c1 = c0 + 16;                         // this lane's column in either half
        const unsigned chunk = (unsigned)(ln & 3) * 16u;
        const unsigned v0 = (c0 >= 0 && c0 < W) ? (unsigned)c0 * kGPixB + chunk : 0x80000000u;
        const unsigned v1 = (c1 >= 0 && c1 < W) ? (unsigned)c1 * kGPixB + chunk : 0x80000000u;
        const int n = 2 * rh;
        for (int i = wave; i < n; i += kWnWaves) {                           // uniform
            const int r = i >> 1, j = i & 1;
            const int y = wy0 + r;
            const bool rowok = y >= 0 && y < H;
            const unsigned soff = __builtin_amdgcn_readfirstlane(rowok ? (unsigned)(st + y * W) * kGPixB : 0u);
            const unsigned voff = rowok ? (j ? v1 : v0) : 0x80000000u;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + buf + (unsigned)r * kWnPitchB + (unsigned)j * 1024u);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :
                         : "s"(m0v), "v"(voff), "s"(rsrc), "s"(soff)
                         : "memory", "m0");
        }
    };

    f32x4 acc[2][2];                       // [octet o'][X]: D rows 4g + r of a lane = query 8 o' + 2g + (r >> 1), part r & 1,
                                           // channel (lane & 15) + 16 ((r >> 1) ^ X)

    auto lds_b128 = [](unsigned a) { return *(__attribute__((address_space(3))) const u32x4 *)a; };
    auto lds_tr = [](unsigned a) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wn_s16x4 *)a));
    };
    // one MFMA step: 4 queries x 2 points (8 samples) x 32 channels.  `wa` = LDS address of this lane's 16 bytes of the
    // A operand; oa / ob = LDS addresses of this lane's row (corner tq, piece tp) of its K-group's two samples
    auto mfma_step = [&](unsigned wa, unsigned oa, unsigned ob, f32x4 &d0, f32x4 &d1) {
        const u32x4 af = lds_b128(wa);
        const u32x2 x0 = lds_tr(oa), x1 = lds_tr(ob), y0 = lds_tr(oa ^ 32u), y1 = lds_tr(ob ^ 32u);
        const u32x4 b0 = {x0.x, x0.y, x1.x, x1.y}, b1 = {y0.x, y0.y, y1.x, y1.y};
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wn_bf16x8, af), __builtin_bit_cast(wn_bf16x8, b0), d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wn_bf16x8, af), __builtin_bit_cast(wn_bf16x8, b1), d1, 0, 0, 0);
    };

    // ---- one level of the wave's 16 queries: set-up into REGISTERS, staged to LDS at the end of the previous pass -----------
    struct Staged {                        // what a lane carries from its set-up to the end of the pass
        unsigned o, h01, h23, l01, l23;    // LDS offset of the sample's top-left corner (bit 31 set: an OVERFLOW sample --
    };                                     // flagged beyond the wave's first four -- and the low bits are its top-left pixel,
                                           // (y0 + 1) << 15 | (x0 + 1)); bf16 high / low parts of the four corner weights
    // top-left pixel `pk` -> this lane's byte offset in the value plane for corner (dx, dy), 16-byte chunk c
    // (corners outside the level: out of range -> the load returns zeros and makes no request)
    auto corner_offset = [&](int l, int pk, bool have, int dx, int dy, int c) -> unsigned {
        const int xx = (pk & 0x7fff) - 1 + dx, yy = (pk >> 15) - 1 + dy;
        const bool ok = have && (unsigned)xx < (unsigned)LW[l] && (unsigned)yy < (unsigned)LH[l];
        return ok ? (unsigned)(LS[l] + yy * LW[l] + xx) * kGPixB + (unsigned)c * 16u : 0x80000000u;
    };
    // set-up of this lane's sample (query qx, point pp) in level l of tile t: msda_fwd.hip's arithmetic
    // (ms_deform_im2col_cuda.cuh:22-73); writes nothing to LDS except the coordinates of the flagged samples, whose rows it
    // sends on their way into patch cell `pb`
    auto setup = [&](int l, int t, f32x2 sxy, float a, bool qok, int pb) -> Staged {
        const int *dsc = desc_tab + (t & (kWnRing - 1)) * 16 + l * 4;
        const int wx0 = __builtin_amdgcn_readfirstlane(dsc[0]);
        const int wy0 = __builtin_amdgcn_readfirstlane(dsc[1]);
        const int rh = __builtin_amdgcn_readfirstlane(dsc[2]);
        const int W = LW[l], H = LH[l];
        const unsigned buf = l < 2 ? (unsigned)kWnBufAOff : (unsigned)kWnBufBOff;
        const float x = sxy.x * (float)W - 0.5f;
        const float y = sxy.y * (float)H - 0.5f;
        const bool inside = qok && (y > -1.f) && (x > -1.f) && (y < (float)H) && (x < (float)W);      // false for NaN
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = inside ? (int)xf : 0, y0 = inside ? (int)yf : 0;       // in [-1, size - 1]
        const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
        // corners outside the level read zeros (window border / range-checked patch loads): no per-corner masks
        const float w00 = inside ? hy * hx * a : 0.f, w01 = inside ? hy * lx * a : 0.f;
        const float w10 = inside ? ly * hx * a : 0.f, w11 = inside ? ly * lx * a : 0.f;
        const int cx = x0 - wx0, cy = y0 - wy0;
        const bool in_win = (unsigned)cx < (unsigned)(kWnWinW - 1) && (unsigned)cy < (unsigned)(rh - 1);
        const bool flagged = inside && !in_win;
        const unsigned long long fmask = __ballot(flagged);
        Staged st;
        st.o = o_zero;                                                        // a sample outside the level: the zero sample
        if (inside && in_win) st.o = lds0 + buf + (unsigned)cy * kWnPitchB + (unsigned)cx * kWnPixB + par32;
        if (fmask != 0ull) {                                                  // uniform
            const int frank = __builtin_amdgcn_mbcnt_hi((unsigned)(fmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fmask, 0));
            const int nflag = __builtin_popcountll(fmask);
            const unsigned pk = ((unsigned)(y0 + 1) << 15) | (unsigned)(x0 + 1);          // levels up to 32766 pixels a side (host check)
            const unsigned cell = lds0 + cell_off(pb);
            if (flagged) st.o = frank < 4 ? cell + (unsigned)frank * 128u + par32 : 0x80000000u | pk;
            if (flagged && frank < 4) fgo[frank] = (int)pk;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // the rows of the first four flagged samples go straight into the patch cell by LDS-DMA (no registers; retired with
            // the window fills before the barrier that ends the pass): lanes 0..31 the top corners (sample k, corner x, chunk),
            // lanes 32..63 the bottom corners one pitch further on
            const int lnp = wn_opaque(lane), pl = lnp & 31, k = pl >> 3;
            const unsigned go = corner_offset(l, fgo[k], k < nflag, (pl >> 2) & 1, lnp >> 5, pl & 3);
            const unsigned m0t = __builtin_amdgcn_readfirstlane(cell);
            const unsigned m0b = __builtin_amdgcn_readfirstlane(cell + kWnPitchB - 512u);
            if (lnp < 32) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                             :
                             : "s"(m0t), "v"(go), "s"(rsrc)
                             : "memory", "m0");
            } else {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                             :
                             : "s"(m0b), "v"(go), "s"(rsrc)
                             : "memory", "m0");
            }
        }
        wn_split2(w00, w01, st.h01, st.l01);
        wn_split2(w10, w11, st.h23, st.l23);
        return st;
    };
    // end of a pass: the staged set-up of the next one goes to LDS (O, W, overflow list) -- this wave's gather, which read the
    // old staging, is done
    auto stage = [&](const Staged &st) {
        const int ln = wn_opaque(lane), qx = ln >> 2, pp = ln & 3;
        unsigned char *wreg = lds + wave_off;
        const bool overflow = (int)st.o < 0;
        const bool any = __ballot(overflow) != 0ull;                          // uniform
        reinterpret_cast<unsigned *>(wreg + kWnStageO)[qx * 4 + pp] = overflow ? o_zero : st.o;   // O[query][point]
        u32x2 *sw = reinterpret_cast<u32x2 *>(wreg + kWnStageW + qx * 64 + pp * 8);              // W[query][part][point][corner]
        sw[0] = u32x2{st.h01, st.h23};
        sw[4] = u32x2{st.l01, st.l23};
        if (any) reinterpret_cast<unsigned *>(wreg + kWnStageV)[ln] = overflow ? st.o : 0u;
        if (ln == 0) fgo[8] = any ? 1 : 0;
    };

    // gather: the MFMA loop over the staged samples -- per (octet, quad half) one 16-byte read brings the top-left offsets
    // of all four points, then two steps
    auto gather = [&](int l, int pb) {
        if (!(dbg & 32)) {
            // Software-pipelined by hand: all four offset reads first, then the operands of step s + 1 are on their way while
            // the MFMAs of step s run (hipcc serialises read -> wait -> MFMA per step: twelve exposed LDS round trips per pass)
            struct Operands { u32x4 af; u32x2 x0, x1, y0, y1; };
            auto fetch = [&](unsigned wa, unsigned oa, unsigned ob) {
                Operands r;
                r.af = lds_b128(wa);
                r.x0 = lds_tr(oa); r.x1 = lds_tr(ob); r.y0 = lds_tr(oa ^ 32u); r.y1 = lds_tr(ob ^ 32u);
                return r;
            };
            auto fma2 = [&](const Operands &r, f32x4 &d0, f32x4 &d1) {
                const u32x4 b0 = {r.x0.x, r.x0.y, r.x1.x, r.x1.y}, b1 = {r.y0.x, r.y0.y, r.y1.x, r.y1.y};
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wn_bf16x8, r.af), __builtin_bit_cast(wn_bf16x8, b0), d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wn_bf16x8, r.af), __builtin_bit_cast(wn_bf16x8, b1), d1, 0, 0, 0);
            };
            const u32x4 so0 = lds_b128(o_rd), so1 = lds_b128(o_rd + 64u), so2 = lds_b128(o_rd + 128u), so3 = lds_b128(o_rd + 192u);
            // step (o', h, j): A operand at (h ? w_rd1 : w_rd0) + o' * 512 + j * 16; rows of points 2j, 2j + 1 of query 8 o' + 4 h + g
            Operands ra = fetch(w_rd0, so0.x + cd, so0.y + cd);
            Operands rb = fetch(w_rd0 + 16, so0.z + cd, so0.w + cd);
            fma2(ra, acc[0][0], acc[0][1]);
            ra = fetch(w_rd1, so1.x + cd, so1.y + cd);
            fma2(rb, acc[0][0], acc[0][1]);
            rb = fetch(w_rd1 + 16, so1.z + cd, so1.w + cd);
            fma2(ra, acc[0][0], acc[0][1]);
            ra = fetch(w_rd0 + 512, so2.x + cd, so2.y + cd);
            fma2(rb, acc[0][0], acc[0][1]);
            rb = fetch(w_rd0 + 528, so2.z + cd, so2.w + cd);
            fma2(ra, acc[1][0], acc[1][1]);
            ra = fetch(w_rd1 + 512, so3.x + cd, so3.y + cd);
            fma2(rb, acc[1][0], acc[1][1]);
            rb = fetch(w_rd1 + 528, so3.z + cd, so3.w + cd);
            fma2(ra, acc[1][0], acc[1][1]);
            fma2(rb, acc[1][0], acc[1][1]);
        }
        // overflow samples (rare): four at a time through the wave's (now consumed) patch cell, one MFMA step per sample with
        // every other row of the operand pointing at the zero sample
        if (__builtin_amdgcn_readfirstlane(fgo[8]) != 0) {
            const unsigned mine = reinterpret_cast<const unsigned *>(wreg + kWnStageV)[lane];     // as set-up lane
            unsigned long long fmask = __ballot((int)mine < 0);
            const int frank = (int)mine < 0 ? __builtin_amdgcn_mbcnt_hi((unsigned)(fmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fmask, 0)) : -1;
            int left = __builtin_popcountll(fmask), done = 0;
            const unsigned cell = cell_off(pb);
            while (fmask != 0ull) {                                  // uniform
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (frank >= done && frank < done + 4) fgo[frank - done] = (int)(mine & 0x3fffffffu);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // lane = (sample g, corner tq, chunk tp): row -> registers -> the cell, window-shaped
                const u32x4 pre = __builtin_amdgcn_raw_buffer_load_b128(rsrc, corner_offset(l, fgo[g], g < left, tq & 1, tq >> 1, tp), 0, 0);
                *reinterpret_cast<u32x4 *>(lds + cell + g * 128 + (tq & 1) * 64 + (tq >> 1) * (int)kWnPitchB + tp * 16) = pre;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (fmask == 0ull) break;                        // uniform
                    const int id = __builtin_ctzll(fmask);           // set-up lane = query * 4 + point
                    fmask &= fmask - 1;
                    const int fq = id >> 2, fp = id & 3;               // its step: octet fq >> 3, quad half (fq >> 2) & 1, pair fp >> 1
                    const unsigned gp32 = (unsigned)(g & 1) * 32u;
                    const unsigned zr = lds0 + (unsigned)kWnZeroOff + gp32 + cd;
                    const unsigned prow = lds0 + cell + (unsigned)k * 128u + gp32 + cd;
                    const unsigned oa = (g == (fq & 3) && !(fp & 1)) ? prow : zr;
                    const unsigned ob = (g == (fq & 3) && (fp & 1)) ? prow : zr;
                    const unsigned wa = ((fq & 4) ? w_rd1 : w_rd0) + (unsigned)((fp >> 1) * 16);
                    if (fq < 8) mfma_step(wa, oa, ob, acc[0][0], acc[0][1]);
                    else mfma_step(wa + 512, oa, ob, acc[1][0], acc[1][1]);
                }
                done += 4;
                left -= 4;
            }
        }
    };

    // out[query][channel] = D[hi row] + D[lo row]; transposed through the wave's W area (dead after the last gather of the
    // tile), one octet at a time, so that a lane stores 16 bytes
    auto store_tile = [&](int sq) {
        const int ln = wn_opaque(lane), qx = ln >> 2, pp = ln & 3, g = ln >> 4;
        float *tr = reinterpret_cast<float *>(lds + wave_off + kWnStageW);    // 1 KiB: 8 queries x 32 channels
#pragma unroll
        for (int op = 0; op < 2; ++op) {
#pragma unroll
            for (int X = 0; X < 2; ++X) {
                const f32x4 d = acc[op][X];
                tr[(2 * g) * 32 + (ln & 15) + 16 * X] = d.x + d.y;
                tr[(2 * g + 1) * 32 + (ln & 15) + 16 * (X ^ 1)] = d.z + d.w;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
                *reinterpret_cast<u32x4 *>(out + ((size_t)b * Nq + sq) * (kWnHeads * kWnHeadDim) + m * kWnHeadDim + pp * 8) = w;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };

    // ---- pipeline ----------------------------------------------------------------------------------------------
    // Tile geometry and window tables are computed for 8 tiles at a time, 16 lanes per tile
    auto tables_for = [&](int t, int k) {                    // 16 consecutive lanes of one wave per tile
        if (t < t1) compute_geometry(t, k);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (t < t1 && k < kWnLevels) window_desc(k, t);
    };
    if (tid < kWnRing * 16) tables_for(t0 + (tid >> 4), tid & 15);
    __syncthreads();

    // Pass P = 4 (t - t0) + p gathers level LV[p] of tile t and sets up pass P + 1; the sample data of pass P + 2 is loaded
    // at its top (one level at a time: a lane carries two levels' worth of samples, not four).  LV = 0, 2, 1, 3.
    int q_cur = query_of(t0), q_nxt = -1;                    // this lane's query in the tile being gathered / the next tile
    float aw[kWnLevels] = {0.f, 0.f, 0.f, 0.f};              // FUSED: soft-maxed weights of this lane's point, current tile
    f32x4 rf = {0.f, 0.f, 0.f, 0.f};                         // FUSED: reference point of level `pp`, current tile
    TileInputs ti_nxt;                                       // FUSED: the next tile's, in flight
    LevelData d_now, d_nxt;                                  // sample data for the coming set-up / the one after, in flight
    auto sample_of = [&](auto lc, const LevelData &d, const float (&w)[kWnLevels], const f32x4 &r, f32x2 &xy, float &a) {
        constexpr int l = decltype(lc)::value;
        if constexpr (FUSED) { xy = fused_location(lc, d.xy.x, r); a = w[l]; }
        else { xy = d.xy; a = d.a; }
    };
    Staged sn;
    fill(0, t0);
    {
        ti_nxt = load_tile_inputs(q_cur);
        d_now = load_level(q_cur, 0);
        d_nxt = load_level(q_cur, 2);
        if constexpr (FUSED) { softmax_levels(ti_nxt, aw); rf = ti_nxt.rf; }
        f32x2 xy; float a;
        sample_of(std::integral_constant<int, 0>{}, d_now, aw, rf, xy, a);
        sn = setup(0, t0, xy, a, q_cur >= 0, 0);             // set-up of the first pass
        stage(sn);
        d_now = d_nxt;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    wn_dma_wait();
    __syncthreads();

    const bool gather_first = wave < kWnWaves / 2;           // the two halves of the workgroup in opposite order
    if (dbg & 64) return;
    for (int t = t0; t < t1; ++t) {
        const bool has_next = t + 1 < t1;
        const bool busy = __ballot(q_cur >= 0) != 0ull;      // any query in this wave?
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
            for (int X = 0; X < 2; ++X) acc[op][X] = f32x4{0.f, 0.f, 0.f, 0.f};

        // pass p: gather level l from its window | set up level ln (of tile tn) = the next pass | window (fl, ft) -> the idle
        // buffer | load the sample data of level ll (tile tl) for the set-up of the pass after
        auto one_pass = [&](auto pc, auto lnc, int l, int tn, bool do_setup, int fl, int ft, bool do_fill, int ll, bool next_tile_load,
                            bool do_load) {
            constexpr int p = decltype(pc)::value;
            constexpr int ln = decltype(lnc)::value;
            // tracked loads first, untracked DMA after them: the compiler's own waits for the former (vmcnt counts in order)
            // then never cover the window fill
            if (p == 2 && has_next && !(dbg & 1024)) {
                q_nxt = query_of(t + 1);
                if (!(dbg & 16)) ti_nxt = load_tile_inputs(q_nxt);
            }
            if (do_load && !(dbg & 16)) d_nxt = load_level(next_tile_load ? q_nxt : q_cur, ll);
            if (do_fill && !(dbg & 2)) fill(fl, ft);
            if constexpr (FUSED && p == 3) {
                if (has_next) { softmax_levels(ti_nxt, aw); rf = ti_nxt.rf; }        // the current tile's last set-up was in pass 2
            }
            const int q_set = p == 3 ? q_nxt : q_cur;
            const bool sbusy = do_setup && __ballot(q_set >= 0) != 0ull;
            // the two halves of the workgroup in opposite order (two copies of the gather rather than a two-trip loop around
            // both: a loop makes the 16 accumulators loop-carried through both arms, and hipcc shuffles them every trip)
            if (gather_first && busy && !(dbg & 4)) gather(l, p & 1);
            if (sbusy && !(dbg & 4)) {
                f32x2 xy; float a;
                LevelData dd = d_now;
                asm volatile("" : "+v"(dd.xy.x), "+v"(dd.xy.y), "+v"(dd.a));     // keeps the set-up arithmetic HERE (the compiler
                sample_of(lnc, dd, aw, rf, xy, a);                               // would run it up front, ahead of the gather)
                sn = setup(ln, tn, xy, a, q_set >= 0, (p + 1) & 1);
            }
            if (!gather_first && busy && !(dbg & 4)) gather(l, p & 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the gather's reads of the staging before the new staging
            __builtin_amdgcn_wave_barrier();
            if (p == 3 && busy && !(dbg & 8)) store_tile(q_cur);
            if (sbusy && !(dbg & 4)) stage(sn);
            if (p == 3 && has_next) q_cur = q_nxt;
            d_now = d_nxt;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (!(dbg & 256)) {
                if (p == 3 && busy && !(dbg & 8)) wn_dma_wait_keep2();      // store_tile's two stores are the youngest operations
                else wn_dma_wait();
                __syncthreads();
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        // gather L0 (A) | set up L2 | L2 -> B | load L1
        one_pass(I0{}, I2{}, 0, t, true, 2, t, true, 1, false, true);
        // gather L2 (B) | set up L1 | L1 -> A | load L3
        one_pass(I1{}, I1{}, 2, t, true, 1, t, true, 3, false, true);
        if (t > t0 && ((t - t0) & 7) == 0 && tid < 128 && !(dbg & 128)) tables_for(t + 8 + (tid >> 4), tid & 15);   // entries of tiles t - 8 .. t - 1 are dead
        // gather L1 (A) | set up L3 | L3 -> B | load the next tile's L0 (and its logits / reference points)
        one_pass(I2{}, I3{}, 1, t, true, 3, t, true, 0, true, has_next);
        // gather L3 (B) | set up the next tile's L0 | its window -> A | load the next tile's L2
        one_pass(I3{}, I0{}, 3, t + 1, has_next, 0, t + 1, has_next, 2, true, has_next);
    }
}

#ifdef RDETR_DEV
// development builds only (make dev -> librelation_detr_amd_dev.so): a mask that switches parts of the kernel off for
// component timing (WRONG results).  The product library is built without it and keeps no state.
static int g_win_dbg = 0;
extern "C" void rdetr_dev_set_win_dbg(int v) { g_win_dbg = v; }
#define RDETR_WIN_DBG g_win_dbg
#else
#define RDETR_WIN_DBG 0
#endif

// Returns RDETR_ERR_UNSUPPORTED when the shape is not served (callers then use the direct kernel).
template <bool FUSED, bool HM>
int msda_win_forward(const uint16_t *value, const int64_t *shapes, const int64_t *level_start, const void *src_a,
                     const void *src_b, const float *ref, int ref_dim, int B, int S, int L, int Nq, int ld_a, int ld_b,
                     uint16_t *out, hipStream_t stream)
{
    if (L != kWnLevels || Nq != S || S < 4096) return RDETR_ERR_UNSUPPORTED;
    const long long gpix = HM ? 64 : 512;
    if ((long long)S * gpix >= (1ll << 31) || S > (1 << 21)) return RDETR_ERR_UNSUPPORTED;     // tile geometry: 2 * 16 * w * regions < 2^24
    auto kern = msda_fwd_win_kernel<FUSED, HM>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kWnLdsBytes);
    if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
    const long long pairs = (long long)B * kWnHeads;
    long long splits = 256 / pairs;                 // one resident workgroup per CU
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    const long long nblk = pairs * splits;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(kWnThreads), (size_t)kWnLdsBytes, stream, value, shapes,
                       level_start, src_a, src_b, ref, ref_dim, S, (int)splits, (int)nblk, RDETR_WIN_DBG, ld_a, ld_b, out);
    return launch_status();
}

#define RDETR_WIN_INST(F, H)                                                                                              \
    template int msda_win_forward<F, H>(const uint16_t *, const int64_t *, const int64_t *, const void *, const void *,   \
                                        const float *, int, int, int, int, int, int, int, uint16_t *, hipStream_t);
RDETR_WIN_INST(false, false)
RDETR_WIN_INST(false, true)
RDETR_WIN_INST(true, false)
RDETR_WIN_INST(true, true)
#undef RDETR_WIN_INST

}  // namespace rdetr
