// Multi-scale deformable attention, forward -- LDS-tiled MFMA kernel for the ENCODER shape (queries = the pyramid's own
// pixels, Nq == S, L == 4, bf16 value) on gfx950 (MI355X).
//
// Why: the direct gather (msda_fwd.hip) is bound twice over -- by the ~0.23 L2 requests/clk/CU a missing L1 sustains (the
// footprint a wave samples is several times the L1) and by the VALU (1 wave-instruction/clk/CU: unpacking bf16 and the
// FMAs cost 3 instructions per sample).  This kernel removes both:
//   memory   per SPATIAL tile (a 16 x 12 pixel region of level 0 and the pixels of the coarser levels whose centres fall
//            into it: <= 192 + 64 queries) and per sampled level, the window ("rect") of the value plane the tile's
//            queries sample -- the region's footprint in that level grown to a fixed size (+-11 x +-8 pixels around a
//            region at level 0) -- is copied L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, coalesced, no VGPRs) one pass ahead of its use.  Tiles are
//            spatial so that every query level of a region shares one window per sampled level (a 16 x 16 block of
//            level-2 QUERIES would need a 64 x 64 window of level 0).
//   math     the bilinear gather-and-weighted-sum runs on the matrix cores: v_mfma_f32_16x16x32_bf16 with
//              K = 8 samples x 4 corners,   B[k][n] = value row k, channel n   read from the LDS window with
//              ds_read_b64_tr_b16 -- the transposed read takes a PER-LANE row address, so the 32 rows of an operand are
//              the gathered rows themselves (odd lane groups read the other channel half: conflict-free),
//              A[m][k] = corner weights, block diagonal: row 4g+r carries the bf16 HIGH (r < 2) or LOW (r >= 2) part of
//              the fp32 weights of sample 2g + (r & 1)  (w = hi + lo to 2^-17, products exact, fp32 accumulate),
//              D[4g+r][n] -> out[query 2g + (r & 1)][n] = D[r] + D[r+2]   (same lane, no shuffle).
//            One MFMA serves 8 samples x 16 channels; the VALU only adds the staged row offsets (about 1 instruction
//            per sample instead of 3.25).
// Nothing is assumed about the sampling locations: a sample whose corners are not all inside its window ("flagged") gets
// its four rows fetched from global memory into a per-wave LDS patch (issued before the pass's MFMA loop, consumed after
// it), so the result never depends on the windows; a decoder-like scatter merely runs slowly.
//
//   workgroup = 1024 threads = 16 waves, persistent over a contiguous range of the tiles of ONE (image, head);
//               waves 0..11 own the 12 rows of the region's level-0 pixels, waves 12..15 its coarser-level pixels
//   LDS 160 KiB = 3 KiB tables | 16 x 2 KiB per-wave staging (row offsets + bf16 weights; re-used as the patch and as the
//               output transpose) | buffer A 1344 px | buffer B 656 px  (64 B / pixel = one bf16 head row)
//   passes    = one per sampled level, order L0 (A), L2 (B), L1 (A), L3 (B): the buffers alternate, the DMA of the next
//               pass is issued before the MFMA loop of the current one; the next tile's locations are loaded one tile
//               ahead.  One barrier per pass.
// Per corner the arithmetic is msda_fwd.hip's (same weights, same zero padding); the summation order differs and each
// weight carries a 2^-17 relative representation error (the bf16 output rounds at 2^-9).
#include <cstdlib>

#include "common.h"

namespace rdetr {

typedef __bf16 tl_bf16x8 __attribute__((ext_vector_type(8)));
typedef short tl_s16x4 __attribute__((ext_vector_type(4)));

constexpr int kTlThreads = 1024;
constexpr int kTlWaves = kTlThreads / kWave;
constexpr int kTlRegW = 16, kTlRegH = 12;                     // level-0 pixels of a spatial tile: one row per wave 0..11
constexpr int kTlCoarseWave0 = kTlRegH;                       // waves 12..15: the region's coarser-level queries
constexpr int kTlCoarseSlots = (kTlWaves - kTlCoarseWave0) * 16;
constexpr int kTlHeads = 8, kTlHeadDim = 32, kTlPoints = 4, kTlLevels = 4;
constexpr unsigned kTlPixB = 64;                              // LDS bytes per pixel (one bf16 head row)
constexpr unsigned kTlGPixB = kTlHeads * kTlHeadDim * 2;      // global bytes per pixel (512)
constexpr int kTlRing = 16;                                // tiles whose geometry / window tables are kept (ring)
constexpr int kTlGeoOff = 3072;                             // int geo[kTlRing][20]
constexpr int kTlDescOff = kTlGeoOff + kTlRing * 80;        // int desc[kTlRing][4 levels][4]
constexpr int kTlMiscBytes = kTlDescOff + kTlRing * 64;     // 5376
constexpr int kTlZeroOff = 512;                               // 64 zero bytes (inside the misc area): the "zero row"
constexpr int kTlFgoOff = 1024;                               // 16 waves x 64 B: global row offsets of the flagged samples in flight
constexpr int kTlZeroKOff = 2048 + 32;                        // ~1 KiB of zeros (from 2048): what the idle lanes of an A operand read;
                                                              // == 32 (mod 256) keeps it off the banks of the active lanes' weights
constexpr int kTlStageOff = kTlMiscBytes;
constexpr int kTlStagePerWave = 2048;                         // [0,1K) row offsets O[query][point][corner], [1K,2K) weights
constexpr int kTlBufAOff = kTlStageOff + kTlWaves * kTlStagePerWave;
constexpr int kTlCapA = 1344, kTlSqWA = 38, kTlSqHA = 28;     // buffer capacity in pixels; window of level 0 (level 1: 30 x 28)
constexpr int kTlBufBOff = kTlBufAOff + kTlCapA * (int)kTlPixB;
constexpr int kTlCapB = 620, kTlSqWB = 22, kTlSqHB = 21;      // window of levels 2 / 3 (a level that fits is taken whole)
constexpr int kTlLdsBytes = kTlBufBOff + kTlCapB * (int)kTlPixB;
static_assert(kTlLdsBytes <= 160 * 1024 && kTlLdsBytes > 160 * 1024 - 64, "LDS map must fill exactly 160 KiB");

struct TileShared {
    int h[kTlLevels], w[kTlLevels], start[kTlLevels];
    int regions_x, regions_y, chunks;      // spatial tiles of level 0; coarse-query chunks per region (1 unless > 64 coarse)
    int max_coarse;
};
// ring tables (kTlGeoOff / kTlDescOff), entry = tile & (kTlRing - 1):
//   geo[20]      rx, ry, chunk, -, then per coarser level: xa, ya, nx, n, 2^16 / nx
//   desc[4][4]   per level: rect x, y, width (== 2 mod 4), height (0 = none)
static_assert(sizeof(TileShared) <= kTlZeroOff, "tables overlap the zero row");

struct TileSamples {                       // lane (query = lane >> 2, point = lane & 3): its sample in each level
    f32x2 xy[kTlLevels];
    float a[kTlLevels];
    int q;                                 // query index of lane >> 2, -1 = none
};

__device__ __forceinline__ float tl_quad_max(float v)
{
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));
    return v;
}
__device__ __forceinline__ float tl_quad_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    return v;
}

// Pixel geometry of one sample in level (W, H): clamped corner columns / rows and the four corner weights.
struct TileCorner {
    int xa, xb, ya, yb;        // columns / rows of the corners, clamped into the level (valid pixels)
    float w00, w01, w10, w11;  // corner weights x attention weight; 0 for corners outside the level
    bool inside;
};
__device__ __forceinline__ TileCorner tl_corners(f32x2 xy, float a, bool qok, int W, int H)
{
    TileCorner c;
    const float x = xy.x * (float)W - 0.5f;
    const float y = xy.y * (float)H - 0.5f;
    c.inside = qok && (y > -1.f) && (x > -1.f) && (y < (float)H) && (x < (float)W);      // false for NaN
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = c.inside ? (int)xf : 0, y0 = c.inside ? (int)yf : 0;
    const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
    const bool okx0 = x0 >= 0, okx1 = x0 + 1 <= W - 1, oky0 = y0 >= 0, oky1 = y0 + 1 <= H - 1;
    c.xa = okx0 ? x0 : x0 + 1;
    c.xb = okx1 ? x0 + 1 : x0;
    c.ya = oky0 ? y0 : y0 + 1;
    c.yb = oky1 ? y0 + 1 : y0;
    c.w00 = (c.inside && okx0 && oky0) ? hy * hx * a : 0.f;
    c.w01 = (c.inside && okx1 && oky0) ? hy * lx * a : 0.f;
    c.w10 = (c.inside && okx0 && oky1) ? ly * hx * a : 0.f;
    c.w11 = (c.inside && okx1 && oky1) ? ly * lx * a : 0.f;
    return c;
}

// a / b for a < 2^24, 0 < b < 2^24, without the ~40-instruction integer division sequence: the float quotient is within
// one of the exact one, two compare-and-adjust steps fix it
__device__ __forceinline__ unsigned tl_div(unsigned a, unsigned b)
{
    unsigned q = (unsigned)((float)a * __builtin_amdgcn_rcpf((float)b));
    int r = (int)a - (int)(q * b);
    if (r < 0) { --q; r += (int)b; }
    if (r >= (int)b) { ++q; }
    return q;
}

// first pixel coordinate of a level of size `n` whose centre lies in region `r` or beyond (regions of `reg` level-0
// pixels, level-0 size n0): the smallest x with (2x + 1) * n0 >= 2 * reg * n * r, clipped to n
__device__ __forceinline__ int tl_region_begin(int r, int reg, int n, int n0)
{
    const unsigned v = 2u * (unsigned)reg * (unsigned)n * (unsigned)r;      // < 2^24: checked on the host
    const unsigned c = tl_div(v + (unsigned)n0 - 1u, (unsigned)n0);
    const int x = (int)(c >> 1);
    return x < n ? x : n;
}

// bf16 high parts (round to nearest even) and low parts of two fp32 weights, packed (a in the low half):
// w = hi + lo up to 2^-17 |w|
typedef __bf16 tl_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tl_split2(float a, float b, unsigned &hi, unsigned &lo)
{
    hi = __builtin_bit_cast(unsigned, tl_bf16x2{(__bf16)a, (__bf16)b});
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, tl_bf16x2{(__bf16)ra, (__bf16)rb});
}

// retire this wave's LDS-DMA before the barrier that publishes the buffer (the compiler does not know about it)
__device__ __forceinline__ void tl_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <bool FUSED>
__global__ __launch_bounds__(kTlThreads) void msda_fwd_tile_kernel(
    const uint16_t *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const void *__restrict__ src_a, const void *__restrict__ src_b, const float *__restrict__ ref, int ref_dim, int S,
    int splits, int nblk, int dbg, int rect_cfg, uint16_t *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(64))) unsigned char lds[];
    TileShared &sh = *reinterpret_cast<TileShared *>(lds);
    int *const geo_tab = reinterpret_cast<int *>(lds + kTlGeoOff);
    int *const desc_tab = reinterpret_cast<int *>(lds + kTlDescOff);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int Nq = S;

    if (tid == 0) {
        for (int l = 0; l < kTlLevels; ++l) {
            sh.h[l] = (int)shapes[2 * l];
            sh.w[l] = (int)shapes[2 * l + 1];
            sh.start[l] = (int)level_start[l];
        }
        sh.regions_x = (sh.w[0] + kTlRegW - 1) / kTlRegW;
        sh.regions_y = (sh.h[0] + kTlRegH - 1) / kTlRegH;
        sh.max_coarse = 0;
    }
    if (tid < 16) reinterpret_cast<unsigned *>(lds + kTlZeroOff)[tid] = 0u;
    if (tid < 256) reinterpret_cast<unsigned *>(lds + 2048)[tid] = 0u;
    __syncthreads();

    // coarser-level pixels per region: level l contributes [xa, xb) x [ya, yb), the pixels whose centres fall in the region
    auto coarse_count = [&](int rx, int ry) {
        int n = 0;
#pragma unroll
        for (int l = 1; l < kTlLevels; ++l) {
            const int nx = tl_region_begin(rx + 1, kTlRegW, sh.w[l], sh.w[0]) - tl_region_begin(rx, kTlRegW, sh.w[l], sh.w[0]);
            const int ny = tl_region_begin(ry + 1, kTlRegH, sh.h[l], sh.h[0]) - tl_region_begin(ry, kTlRegH, sh.h[l], sh.h[0]);
            n += nx * ny;
        }
        return n;
    };
    {
        const int nreg = sh.regions_x * sh.regions_y;
        int mx = 0;
        for (int r = tid; r < nreg; r += kTlThreads) {
            const int ry = r / sh.regions_x;
            const int n = coarse_count(r - ry * sh.regions_x, ry);
            mx = n > mx ? n : mx;
        }
        if (mx > 0) atomicMax(&sh.max_coarse, mx);
    }
    __syncthreads();
    if (tid == 0) sh.chunks = sh.max_coarse <= kTlCoarseSlots ? 1 : (sh.max_coarse + kTlCoarseSlots - 1) / kTlCoarseSlots;
    __syncthreads();

    // the tables are wave-uniform: keep them in SGPRs (values read from LDS live in VGPRs, and every index computation
    // on them -- including the integer divisions of the tile geometry -- would run on the VALU)
    int LW[kTlLevels], LH[kTlLevels], LS[kTlLevels];
#pragma unroll
    for (int l = 0; l < kTlLevels; ++l) {
        LW[l] = __builtin_amdgcn_readfirstlane(sh.w[l]);
        LH[l] = __builtin_amdgcn_readfirstlane(sh.h[l]);
        LS[l] = __builtin_amdgcn_readfirstlane(sh.start[l]);
    }
    const int regions_x = __builtin_amdgcn_readfirstlane(sh.regions_x);
    const int regions_y = __builtin_amdgcn_readfirstlane(sh.regions_y);

    const int logical = xcd_contiguous_block(blockIdx.x, nblk);
    const int pair = logical / splits, split = logical - pair * splits;
    const int b = pair / kTlHeads, m = pair - b * kTlHeads;
    const int chunks = __builtin_amdgcn_readfirstlane(sh.chunks);
    const int ntiles = regions_x * regions_y * chunks;
    const int t0 = (int)((long long)split * ntiles / splits), t1 = (int)((long long)(split + 1) * ntiles / splits);
    if (t0 >= t1) return;                                   // uniform for the workgroup

    const unsigned char *plane = reinterpret_cast<const unsigned char *>(value) + (size_t)b * S * kTlGPixB + (size_t)m * kTlPixB;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(plane), 0, (unsigned)S * kTlGPixB - (unsigned)m * kTlPixB, 0x00020000);

    // set-up role: query qx of the wave, point pp.   gather role: K-group g, row tq / piece tp of a transposed read;
    // as an A-operand lane: row am = lane & 15 = 8 * (quad half ah) + 2 * (K-group ag) + (0 = bf16 high part, 1 = low part)
    const int qx = lane >> 2, pp = lane & 3;
    const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int am = lane & 15, ah = am >> 3, ag = (am >> 1) & 3, apart = am & 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;     // 0 in practice
    const unsigned stage_off = (unsigned)(kTlStageOff + wave * kTlStagePerWave);
    unsigned char *stage = lds + stage_off;
    // staging of one pass: [0, 1K) O[query][corner][point] u32 LDS row offsets (the +32 of the odd queries' channel-half
    // swizzle included), [1K, 2K) W[query][part][point][corner] bf16 weights.  One MFMA step = octet o', quad half h, point
    // pair j: K-group g carries the two samples (points 2j, 2j + 1) of query 8 o' + 4 h + g.
    const unsigned o_rd = lds0 + stage_off + (unsigned)(g * 64 + tq * 16);                 // + o' * 512 + h * 256: all 4 points
    // A operand: lane (row am, K-group g) is live only in the steps of its own quad half and only if its row's query is the
    // K-group's -- then it reads that query's 2 x 4 weights (16 B); otherwise 16 B of zeros.  + o' * 512 + j * 16
    const unsigned w_real = lds0 + stage_off + 1024u + (unsigned)((4 * ah + g) * 64 + apart * 32);
    const unsigned w_rd0 = (ag == g && ah == 0) ? w_real : lds0 + (unsigned)kTlZeroKOff;
    const unsigned w_rd1 = (ag == g && ah == 1) ? w_real : lds0 + (unsigned)kTlZeroKOff;
    const unsigned c0 = lds0 + (unsigned)(tp * 8);
    unsigned char *fgo = lds + kTlFgoOff + wave * 64;

    // ---- helpers -----------------------------------------------------------------------------------------------
    // query owned by lane >> 2 of this wave in tile t (-1 = none)
    // Tile geometry (which region, and which pixels of the coarser levels have their centres in it) costs a dozen integer
    // divisions: ONE lane per value computes it, once per tile, into sh.geo[t & 1]; every wave then reads it as scalars.
    auto compute_geometry = [&](int t, int k) {        // k = lane of the computing wave, 0 .. 15
        const int region = (int)tl_div((unsigned)t, (unsigned)chunks), chunk = t - region * chunks;
        const int ry = (int)tl_div((unsigned)region, (unsigned)regions_x), rx = region - ry * regions_x;
        int *geo = geo_tab + (t & (kTlRing - 1)) * 20;
        if (k == 0) { geo[0] = rx; geo[1] = ry; geo[2] = chunk; }
        if (k >= 1 && k < kTlLevels) {
            const int l = k;
            const int xa = tl_region_begin(rx, kTlRegW, LW[l], LW[0]), xb = tl_region_begin(rx + 1, kTlRegW, LW[l], LW[0]);
            const int ya = tl_region_begin(ry, kTlRegH, LH[l], LH[0]), yb = tl_region_begin(ry + 1, kTlRegH, LH[l], LH[0]);
            const int nx = xb - xa;
            geo[4 + 5 * (l - 1) + 0] = xa;
            geo[4 + 5 * (l - 1) + 1] = ya;
            geo[4 + 5 * (l - 1) + 2] = nx;
            geo[4 + 5 * (l - 1) + 3] = nx * (yb - ya);
            geo[4 + 5 * (l - 1) + 4] = nx > 0 ? (int)tl_div(65536u + (unsigned)nx - 1u, (unsigned)nx) : 0;   // j / nx for j < 2^16 / nx
        }
    };

    // query owned by lane >> 2 of this wave in tile t (-1 = none)
    auto query_of = [&](int t) -> int {
        const int *geo = geo_tab + (t & (kTlRing - 1)) * 20;
        const int rx = __builtin_amdgcn_readfirstlane(geo[0]), ry = __builtin_amdgcn_readfirstlane(geo[1]);
        const int chunk = __builtin_amdgcn_readfirstlane(geo[2]);
        if (wave < kTlCoarseWave0) {
            const int x = rx * kTlRegW + qx, y = ry * kTlRegH + wave;
            return (chunk == 0 && x < LW[0] && y < LH[0]) ? LS[0] + y * LW[0] + x : -1;
        }
        int j = chunk * kTlCoarseSlots + (wave - kTlCoarseWave0) * 16 + qx;
        int q = -1;
#pragma unroll
        for (int l = 1; l < kTlLevels; ++l) {
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

    // sampling locations / attention weights of this lane's (query, point) in every level
    auto load_samples = [&](int t, TileSamples &sm) {
        sm.q = query_of(t);
        const size_t row = (size_t)b * Nq + (sm.q >= 0 ? sm.q : 0);
        const size_t hrow = (row * kTlHeads + m) * (size_t)(kTlLevels * kTlPoints);
        if constexpr (FUSED) {
            const uint16_t *off_q = static_cast<const uint16_t *>(src_a) + hrow * 2;
            const uint16_t *lg_q = static_cast<const uint16_t *>(src_b) + hrow;
            float mx = -__builtin_inff();
#pragma unroll
            for (int l = 0; l < kTlLevels; ++l) {
                const int pt = l * kTlPoints + pp;
                sm.a[l] = bf16_bits_to_f32(lg_q[pt]);
                const unsigned u = *reinterpret_cast<const unsigned *>(off_q + 2 * pt);
                sm.xy[l] = f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
                mx = fmaxf(mx, sm.a[l]);
            }
            mx = tl_quad_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int l = 0; l < kTlLevels; ++l) {
                sm.a[l] = expf(sm.a[l] - mx);
                sum += sm.a[l];
            }
            sum = tl_quad_sum(sum);
#pragma unroll
            for (int l = 0; l < kTlLevels; ++l) {
                const float *rp = ref + (row * kTlLevels + l) * (size_t)ref_dim;
                sm.a[l] = sm.a[l] / sum;
                if (ref_dim == 2) {
                    sm.xy[l].x = rp[0] + sm.xy[l].x / (float)LW[l];
                    sm.xy[l].y = rp[1] + sm.xy[l].y / (float)LH[l];
                } else {
                    sm.xy[l].x = rp[0] + sm.xy[l].x * (1.0f / kTlPoints) * rp[2] * 0.5f;
                    sm.xy[l].y = rp[1] + sm.xy[l].y * (1.0f / kTlPoints) * rp[3] * 0.5f;
                }
            }
        } else {
            // 32-bit element offsets from the image's (uniform) base: B * Nq * 8 * 16 * 2 floats can exceed 2^32, one image cannot
            const float *loc_b = static_cast<const float *>(src_a) + (size_t)b * Nq * (kTlHeads * kTlLevels * kTlPoints * 2);
            const float *att_b = static_cast<const float *>(src_b) + (size_t)b * Nq * (kTlHeads * kTlLevels * kTlPoints);
            const unsigned e = ((unsigned)(sm.q >= 0 ? sm.q : 0) * kTlHeads + (unsigned)m) * (kTlLevels * kTlPoints) + (unsigned)pp;
#pragma unroll
            for (int l = 0; l < kTlLevels; ++l) {
                sm.xy[l] = *reinterpret_cast<const f32x2 *>(loc_b + 2u * (e + (unsigned)(l * kTlPoints)));
                sm.a[l] = att_b[e + (unsigned)(l * kTlPoints)];
            }
        }
    };

    // thread l: the window of level l for tile t = the region's footprint in that level (the pixels whose centres lie in
    // it) grown symmetrically to a fixed size: 38 x 28 / 30 x 28 in A (levels 0 / 1: +-11 x +-8 pixels around a region), 22 x 21
    // in B (levels 2 / 3; a level that fits entirely is taken whole) -- the sizes that measured fastest on SURVEY 8d's spread.  Fixed-size windows cost nothing to determine (a
    // data-driven bounding box per tile was a quarter of the kernel's time and came out clipped to these sizes anyway);
    // whatever a window misses is flagged and patched, so this is a speed heuristic only.
    auto fixed_desc = [&](int l, int t) {
        const int *geo = geo_tab + (t & (kTlRing - 1)) * 20;
        int fx0, fx1, fy0, fy1;                        // footprint [fx0, fx1) x [fy0, fy1)
        if (l == 0) {
            fx0 = geo[0] * kTlRegW; fx1 = fx0 + kTlRegW;
            fy0 = geo[1] * kTlRegH; fy1 = fy0 + kTlRegH;
        } else {
            const int nx = geo[4 + 5 * (l - 1) + 2], n = geo[4 + 5 * (l - 1) + 3];
            fx0 = geo[4 + 5 * (l - 1)]; fx1 = fx0 + nx;
            fy0 = geo[4 + 5 * (l - 1) + 1]; fy1 = fy0 + (nx > 0 ? (int)tl_div((unsigned)n, (unsigned)nx) : 0);
        }
        const int W = LW[l], H = LH[l];
        const int cap = l < 2 ? kTlCapA : kTlCapB;
        int rw = l == 0 ? kTlSqWA : (l == 1 ? kTlSqWA - 8 : kTlSqWB), rh = l < 2 ? kTlSqHA : kTlSqHB;
        if (rect_cfg) {                                 // tuning aid (RDETR_TILE_RECT): windows grown by 4 * (rect_cfg & 15) columns
            rw += 4 * (rect_cfg & 15);                  // and (rect_cfg >> 4) rows, as far as the buffers allow
            rh += rect_cfg >> 4;
            while (rw * rh > cap) --rh;
        }
        const int wfull = ((W + 1) & ~3) + 2;           // smallest width >= W that is == 2 (mod 4)
        if (wfull <= rw) { rw = wfull; const int hmax = (int)tl_div((unsigned)cap, (unsigned)rw); rh = H < hmax ? H : hmax; }
        if (H < rh) rh = H;
        int rx = (fx0 + fx1 - rw) / 2, ry = (fy0 + fy1 - rh) / 2;              // centred on the footprint (floor for negatives is irrelevant: clamped)
        const int mx = W - rw, my = H - rh;
        rx = rx > mx ? mx : rx; rx = rx < 0 ? 0 : rx;
        ry = ry > my ? my : ry; ry = ry < 0 ? 0 : ry;
        int *d = desc_tab + (t & (kTlRing - 1)) * 16 + l * 4;
        d[0] = rx;
        d[1] = ry;
        d[2] = rw;
        d[3] = rh;
    };

    // DMA the rect of level l (tile parity par) into its buffer: lane = (pixel, 16-byte chunk), 16 pixels per instruction
    auto fill = [&](int l, int par) {
        const int *dsc = desc_tab + (par & (kTlRing - 1)) * 16 + l * 4;        // par = the tile
        const int rx = __builtin_amdgcn_readfirstlane(dsc[0]);
        const int ry = __builtin_amdgcn_readfirstlane(dsc[1]);
        const int rw = __builtin_amdgcn_readfirstlane(dsc[2]);
        const int rh = __builtin_amdgcn_readfirstlane(dsc[3]);
        const int W = LW[l], st = LS[l];
        const unsigned buf = l < 2 ? (unsigned)kTlBufAOff : (unsigned)kTlBufBOff;
        const int n4 = rw * rh * 4;
        // pixel tid >> 2 of the rect -> (row, column); one division up front (the +0.5 makes the approximate reciprocal
        // exact for these small integers), then every further pixel of this thread is 256 pixels on: a constant step
        const float inv = __builtin_amdgcn_rcpf((float)rw);
        const int r0 = (int)(((float)(tid >> 2) + 0.5f) * inv);
        int cx = (tid >> 2) - r0 * rw;
        const int dr = (int)tl_div(256u, (unsigned)rw), dc = 256 - dr * rw;      // uniform
        int gp = st + (ry + r0) * W + rx + cx;
        const int step = dr * W + dc, wrap = W - rw;
        const unsigned lane16 = (unsigned)(tid & 3) * 16u;
        for (int i = tid; i < n4; i += kTlThreads) {
            const int gc = gp < S ? gp : S - 1;                   // columns past the level's edge (width rounding) read
            // valid-but-unused pixels.  The DMA is issued through inline assembly ON PURPOSE: hipcc tracks the builtin as
            // an LDS write and drains it (s_waitcnt vmcnt(0)) before the pass's first LDS read, which serialises fill and
            // gather; untracked, it stays in flight behind the MFMA loop and is retired by tl_dma_wait() before the barrier
            // that hands the buffer over.  (M0 = LDS destination of the wave's lane 0, lanes land 16 bytes apart.)
            const unsigned voff = (unsigned)gc * kTlGPixB + lane16;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(buf + (unsigned)(i & ~63) * 16u);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :
                         : "s"(m0v), "v"(voff), "s"(plane)
                         : "memory", "m0");
            cx += dc;
            const bool w = cx >= rw;
            cx -= w ? rw : 0;
            gp += step + (w ? wrap : 0);
        }
    };

    f32x4 acc[2][2];                       // [octet o'][X]: D rows 4g + r of a lane = query 8 o' + 2g + (r >> 1), part r & 1,
                                           // channel (lane & 15) + 16 ((r >> 1) ^ X)

    auto lds_b128 = [](unsigned a) { return *(__attribute__((address_space(3))) const u32x4 *)a; };
    auto lds_tr = [](unsigned a) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tl_s16x4 *)a));
    };
    // one MFMA step: 4 queries x 2 points (8 samples) x 32 channels.  `wa` = LDS address of this lane's 16 bytes of the
    // A operand; oa / ob = LDS offsets of this lane's row (corner tq) of its K-group's two samples
    auto mfma_step = [&](unsigned wa, unsigned oa, unsigned ob, f32x4 &d0, f32x4 &d1) {
        const u32x4 af = lds_b128(wa);
        const unsigned ba = oa + c0, bb = ob + c0;
        const u32x2 x0 = lds_tr(ba), x1 = lds_tr(bb), y0 = lds_tr(ba ^ 32u), y1 = lds_tr(bb ^ 32u);
        const u32x4 b0 = {x0.x, x0.y, x1.x, x1.y}, b1 = {y0.x, y0.y, y1.x, y1.y};
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tl_bf16x8, af), __builtin_bit_cast(tl_bf16x8, b0), d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tl_bf16x8, af), __builtin_bit_cast(tl_bf16x8, b1), d1, 0, 0, 0);
    };

    // one level of the wave's 16 queries: set-up (lane = query x point) -> staging -> MFMA loop -> patch steps for flagged samples
    auto pass = [&](int l, int par, const TileSamples &sm, auto &&after_setup) {
        const int *dsc = desc_tab + (par & (kTlRing - 1)) * 16 + l * 4;        // par = the tile
        const int rx = __builtin_amdgcn_readfirstlane(dsc[0]);
        const int ry = __builtin_amdgcn_readfirstlane(dsc[1]);
        const int rw = __builtin_amdgcn_readfirstlane(dsc[2]);
        const int rh = __builtin_amdgcn_readfirstlane(dsc[3]);
        const int W = LW[l], H = LH[l];
        const unsigned buf = l < 2 ? (unsigned)kTlBufAOff : (unsigned)kTlBufBOff;
        const TileCorner c = tl_corners(sm.xy[l], sm.a[l], sm.q >= 0, W, H);
        const bool in_rect = c.xa >= rx && c.xb < rx + rw && c.ya >= ry && c.yb < ry + rh;
        const bool flagged = c.inside && !in_rect;
        const unsigned zrow = (unsigned)kTlZeroOff + (unsigned)(qx & 1) * 32u;       // odd queries read the other channel half first
        u32x4 o = {zrow, zrow, zrow, zrow};
        if (c.inside && in_rect) {           // LDS byte offsets inside the rect
            const unsigned o00 = buf + (unsigned)(qx & 1) * 32u + (unsigned)((c.ya - ry) * rw + (c.xa - rx)) * kTlPixB;
            const unsigned dx = (unsigned)(c.xb - c.xa) * kTlPixB, dy = (unsigned)(c.yb - c.ya) * (unsigned)rw * kTlPixB;
            o = u32x4{o00, o00 + dx, o00 + dy, o00 + dy + dx};
        }
        unsigned h01, h23, l01, l23;
        tl_split2(c.w00, c.w01, h01, l01);
        tl_split2(c.w10, c.w11, h23, l23);
        {
            unsigned *so = reinterpret_cast<unsigned *>(stage + qx * 64 + pp * 4);       // O[query][corner][point]
            so[0] = o.x; so[4] = o.y; so[8] = o.z; so[12] = o.w;
            u32x2 *sw = reinterpret_cast<u32x2 *>(stage + 1024 + qx * 64 + pp * 8);     // W[query][part][point][corner]
            sw[0] = u32x2{h01, h23};
            sw[4] = u32x2{l01, l23};
        }
        unsigned long long fmask = __ballot(flagged);          // remaining flagged samples (bit = set-up lane = query * 4 + point)

        // flagged samples: publish the global byte offsets of four of them, start their row loads (lane = sample g,
        // corner tq, 16-byte chunk tp) -- the loads land behind the pass's DMA fill, i.e. by the time the MFMA loop is done
        u32x4 pre = {0u, 0u, 0u, 0u};
        const unsigned long long fmask0 = fmask;
        auto issue_patch_loads = [&](int first_rank) {
            const int frank = __builtin_amdgcn_mbcnt_hi((unsigned)(fmask0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fmask0, 0));
            if (flagged && frank >= first_rank && frank < first_rank + 4) {
                const unsigned g00 = (unsigned)(LS[l] + c.ya * W + c.xa) * kTlGPixB;
                const unsigned gdx = (unsigned)(c.xb - c.xa) * kTlGPixB, gdy = (unsigned)(c.yb - c.ya) * (unsigned)W * kTlGPixB;
                *reinterpret_cast<u32x4 *>(fgo + (frank - first_rank) * 16) = u32x4{g00, g00 + gdx, g00 + gdy, g00 + gdy + gdx};
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int have = __builtin_popcountll(fmask);                           // uniform: samples still to serve
            const unsigned go = g < have ? reinterpret_cast<const unsigned *>(fgo)[g * 4 + tq] + (unsigned)tp * 16u : 0x80000000u;
            pre = __builtin_amdgcn_raw_buffer_load_b128(rsrc, go, 0, 0);           // out of range -> zeros, no request
        };
        if (fmask != 0ull) issue_patch_loads(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // staging is private to the wave: wave-level ordering suffices
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        after_setup();                       // the locations are consumed: pass 3 starts loading the next tile's here

        // per (octet, quad half): one 16-byte read brings the row offsets of all four points, then two steps
        if (!(dbg & 32))
#pragma unroll
        for (int op = 0; op < 2; ++op) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x4 so = lds_b128(o_rd + op * 512 + h * 256);
                const unsigned wa = (h ? w_rd1 : w_rd0) + op * 512;
                mfma_step(wa, so.x, so.y, acc[op][0], acc[op][1]);
                mfma_step(wa + 16, so.z, so.w, acc[op][0], acc[op][1]);
            }
        }

        // patch steps: the rows of up to four flagged samples at a time go into the (now dead) offset area, then one MFMA
        // step per sample with every other row of the operand pointing at the zero row
        int done = 0;
        while (fmask != 0ull) {                                  // uniform
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<u32x4 *>(stage + lane * 16) = pre;                   // row (sample g, corner tq) at g * 256 + tq * 64
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (fmask == 0ull) break;                        // uniform
                const int id = __builtin_ctzll(fmask);           // set-up lane = query * 4 + point
                fmask &= fmask - 1;
                const int fq = id >> 2, fp = id & 3;               // its step: octet fq >> 3, quad half (fq >> 2) & 1, pair fp >> 1
                const unsigned zr = (unsigned)kTlZeroOff + (unsigned)(g & 1) * 32u;
                const unsigned prow = stage_off + (unsigned)(k * 256 + tq * 64) + (unsigned)(g & 1) * 32u;
                const unsigned oa = (g == (fq & 3) && !(fp & 1)) ? prow : zr;
                const unsigned ob = (g == (fq & 3) && (fp & 1)) ? prow : zr;
                const unsigned wa = ((fq & 4) ? w_rd1 : w_rd0) + (unsigned)((fp >> 1) * 16);
                if (fq < 8) mfma_step(wa, oa, ob, acc[0][0], acc[0][1]);
                else mfma_step(wa + 512, oa, ob, acc[1][0], acc[1][1]);
            }
            done += 4;
            if (fmask != 0ull) issue_patch_loads(done);          // more than four: next batch (its latency is exposed; rare)
        }
        // retire the patch load for the compiler's bookkeeping at the END of the pass (otherwise it parks its wait on the
        // first instruction of the next pass that re-uses these registers -- behind that pass's DMA)
        asm volatile("" ::"v"(pre.x), "v"(pre.y), "v"(pre.z), "v"(pre.w));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // reads before the next pass's staging writes
        __builtin_amdgcn_wave_barrier();
    };

    // out[query][channel] = D[hi row] + D[lo row]; transposed through the wave's staging area so that a lane stores 16 bytes
    auto store_tile = [&](int sq) {
        float *tr = reinterpret_cast<float *>(stage);
#pragma unroll
        for (int op = 0; op < 2; ++op) {
#pragma unroll
            for (int X = 0; X < 2; ++X) {
                const f32x4 d = acc[op][X];
                tr[(8 * op + 2 * g) * 32 + (lane & 15) + 16 * X] = d.x + d.y;
                tr[(8 * op + 2 * g + 1) * 32 + (lane & 15) + 16 * (X ^ 1)] = d.z + d.w;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(tr + qx * 32 + pp * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(tr + qx * 32 + pp * 8 + 4);
        if (sq >= 0) {
            u32x4 w;
            w.x = f32_to_bf16_bits(lo.x) | (f32_to_bf16_bits(lo.y) << 16);
            w.y = f32_to_bf16_bits(lo.z) | (f32_to_bf16_bits(lo.w) << 16);
            w.z = f32_to_bf16_bits(hi.x) | (f32_to_bf16_bits(hi.y) << 16);
            w.w = f32_to_bf16_bits(hi.z) | (f32_to_bf16_bits(hi.w) << 16);
            *reinterpret_cast<u32x4 *>(out + ((size_t)b * Nq + sq) * (kTlHeads * kTlHeadDim) + m * kTlHeadDim + pp * 8) = w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // ---- pipeline ----------------------------------------------------------------------------------------------
    // Tile geometry and window tables are computed for 8 tiles at a time, 16 lanes per tile (the arithmetic is a serial
    // chain of a few hundred instructions: done per tile by one wave it delayed every barrier of the tile by ~1.5 us)
    auto tables_for = [&](int t, int k) {                    // 16 consecutive lanes of one wave per tile
        if (t < t1) compute_geometry(t, k);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (t < t1 && k < kTlLevels) fixed_desc(k, t);
    };
    if (tid < kTlRing * 16) tables_for(t0 + (tid >> 4), tid & 15);
    __syncthreads();
    TileSamples cur;                       // ONE register set: the next tile's locations are loaded in pass 3, after the
    load_samples(t0, cur);                 // last set-up of the current tile has consumed them (a second set spills)
#pragma unroll
    for (int l = 0; l < kTlLevels; ++l) asm volatile("" ::"v"(cur.xy[l].x), "v"(cur.xy[l].y), "v"(cur.a[l]));   // retire the loads (see pass 3)
    fill(0, t0);
    tl_dma_wait();
    __syncthreads();

    for (int t = t0; t < t1; ++t) {
        const int par = t;
        const bool has_next = t + 1 < t1;
        const bool busy = __ballot(cur.q >= 0) != 0ull;  // any query in this wave?
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
            for (int X = 0; X < 2; ++X) acc[op][X] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (!(dbg & 2)) fill(2, par);                                   // pass 0: level 0 from A   | level 2 -> B in flight
        if (busy && !(dbg & 4)) pass(0, par, cur, [] {});
        tl_dma_wait();
        __syncthreads();

        if (!(dbg & 2)) fill(1, par);                                   // pass 1: level 2 from B   | level 1 -> A in flight
        if (busy && !(dbg & 4)) pass(2, par, cur, [] {});
        tl_dma_wait();
        __syncthreads();

        if (!(dbg & 2)) fill(3, par);                                   // pass 2: level 1 from A   | level 3 -> B in flight
        if (busy && !(dbg & 4)) pass(1, par, cur, [] {});
        tl_dma_wait();
        __syncthreads();

        if (t > t0 && ((t - t0) & 7) == 0 && tid < 128) tables_for(t + 8 + (tid >> 4), tid & 15);      // entries of tiles t - 8 .. t - 1 are dead
        if (has_next && !(dbg & 2)) fill(0, t + 1);                 // pass 3: level 3 from B   | next tile's level 0 -> A in flight
        const int sq = cur.q;
        bool loaded = false;
        if (busy && !(dbg & 4)) {
            pass(3, par, cur, [&] {            // after the last set-up of the tile the locations are dead: fetch the next tile's
                if (has_next && !(dbg & 16)) load_samples(t + 1, cur);     // behind the MFMA loop and the store
            });
            loaded = true;
        }
        if (!loaded && has_next && !(dbg & 16)) load_samples(t + 1, cur);
        if (busy && !(dbg & 8)) store_tile(sq);
        // make the compiler retire the location loads HERE (it waits lazily, at the first use -- which would be inside the
        // next pass, behind that pass's untracked DMA, and would drain it)
#pragma unroll
        for (int l = 0; l < kTlLevels; ++l) asm volatile("" ::"v"(cur.xy[l].x), "v"(cur.xy[l].y), "v"(cur.a[l]));
        tl_dma_wait();
        __syncthreads();
    }
}

// Returns RDETR_ERR_UNSUPPORTED when the shape is not served (callers then use the direct kernel).
template <bool FUSED>
int msda_tile_forward(const uint16_t *value, const int64_t *shapes, const int64_t *level_start, const void *src_a,
                      const void *src_b, const float *ref, int ref_dim, int B, int S, int L, int Nq, uint16_t *out,
                      hipStream_t stream)
{
    if (L != kTlLevels || Nq != S || S < 4096) return RDETR_ERR_UNSUPPORTED;
    if ((long long)S * kTlGPixB >= (1ll << 31) || S > (1 << 21)) return RDETR_ERR_UNSUPPORTED;     // tile geometry: 2 * 16 * w * regions < 2^24
    auto kern = msda_fwd_tile_kernel<FUSED>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kTlLdsBytes);
    if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
    static const int dbg = []() { const char *e = getenv("RDETR_TILE_DBG"); return e ? atoi(e) : 0; }();   // timing experiments only
    static const int rect_cfg = []() { const char *e = getenv("RDETR_TILE_RECT"); return e ? atoi(e) : 0; }();
    const long long pairs = (long long)B * kTlHeads;
    long long splits = 256 / pairs;                 // one resident workgroup per CU
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    const long long nblk = pairs * splits;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(kTlThreads), (size_t)kTlLdsBytes, stream, value, shapes,
                       level_start, src_a, src_b, ref, ref_dim, S, (int)splits, (int)nblk, dbg, rect_cfg, out);
    return launch_status();
}

template int msda_tile_forward<false>(const uint16_t *, const int64_t *, const int64_t *, const void *, const void *,
                                      const float *, int, int, int, int, int, uint16_t *, hipStream_t);
template int msda_tile_forward<true>(const uint16_t *, const int64_t *, const int64_t *, const void *, const void *,
                                     const float *, int, int, int, int, int, uint16_t *, hipStream_t);

}  // namespace rdetr
