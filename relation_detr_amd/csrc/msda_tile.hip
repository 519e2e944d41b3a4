// Multi-scale deformable attention, forward -- LDS-tiled MFMA kernel for the ENCODER shape (queries = the pyramid's own
// pixels, Nq == S, L == 4, bf16 value) on gfx950 (MI355X).
//
// Why: the direct gather (msda_fwd.hip) is bound twice over -- by the ~0.23 L2 requests/clk/CU a missing L1 sustains (the
// footprint a wave samples is several times the L1) and by the VALU (1 wave-instruction/clk/CU: unpacking bf16 and the
// FMAs cost 3 instructions per sample).  This kernel removes both:
//   memory   per SPATIAL tile (a 16 x 12 pixel region of level 0 and the pixels of the coarser levels whose centres fall
//            into it: <= 192 + 64 queries) and per sampled level, the window ("rect") of the value plane the tile's
//            queries sample -- the bounding box of their ACTUAL sample corners, clipped to the buffer -- is copied
//            L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, coalesced, no VGPRs) one pass ahead of its use.  Tiles are
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
//   LDS 160 KiB = 2 KiB tables | 16 x 2 KiB per-wave staging (row offsets + bf16 weights; re-used as the patch and as the
//               output transpose) | buffer A 1344 px | buffer B 672 px  (64 B / pixel = one bf16 head row)
//   passes    = one per sampled level, order L0 (A), L2 (B), L1 (A), L3 (B): the buffers alternate, the DMA of the next
//               pass is issued before the MFMA loop of the current one; the next tile's locations are loaded and their
//               bounding boxes reduced two passes ahead.  One barrier per pass.
// Per corner the arithmetic is msda_fwd.hip's (same weights, same zero padding); the summation order differs and each
// weight carries a 2^-17 relative representation error (the bf16 output rounds at 2^-9).
#include <climits>

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
constexpr int kTlMiscBytes = 2048;
constexpr int kTlZeroOff = 512;                               // 64 zero bytes (inside the misc area)
constexpr int kTlFgoOff = 1024;                               // 16 waves x 64 B: global row offsets of the flagged samples in flight
constexpr int kTlStageOff = kTlMiscBytes;
constexpr int kTlStagePerWave = 2048;                         // [0,1K) row offsets O[query][point][corner], [1K,2K) weights
constexpr int kTlBufAOff = kTlStageOff + kTlWaves * kTlStagePerWave;
constexpr int kTlCapA = 1344, kTlSqWA = 42, kTlSqHA = 32;     // pixels; fallback rect of the buffer when both sides clip
constexpr int kTlBufBOff = kTlBufAOff + kTlCapA * (int)kTlPixB;
constexpr int kTlCapB = 672, kTlSqWB = 26, kTlSqHB = 25;
constexpr int kTlLdsBytes = kTlBufBOff + kTlCapB * (int)kTlPixB;
static_assert(kTlLdsBytes == 160 * 1024, "LDS map must fill exactly 160 KiB");

struct TileShared {
    int h[kTlLevels], w[kTlLevels], start[kTlLevels];
    int regions_x, regions_y, chunks;      // spatial tiles of level 0; coarse-query chunks per region (1 unless > 64 coarse)
    int max_coarse;
    int bbox[kTlLevels * 4];               // per level: min x, min y, max x, max y of the next tile's valid sample corners
    int desc[2][kTlLevels][4];             // per tile parity and level: rect x, y, width (== 2 mod 4), height (0 = none)
};
static_assert(sizeof(TileShared) <= kTlZeroOff, "tables overlap the zero row");

struct TileSamples {                       // lane (query = lane >> 2, point = lane & 3): its sample in each level
    f32x2 xy[kTlLevels];
    float a[kTlLevels];
    int q;                                 // query index of lane >> 2, -1 = none
};

template <bool MAX> __device__ __forceinline__ int tl_wave_reduce(int v)
{
#define RDETR_TL_STEP(ctrl)                                                          \
    {                                                                                \
        const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false);      \
        v = MAX ? (v > o ? v : o) : (v < o ? v : o);                                 \
    }
    RDETR_TL_STEP(0xB1)     // quad_perm [1,0,3,2]
    RDETR_TL_STEP(0x4E)     // quad_perm [2,3,0,1]
    RDETR_TL_STEP(0x141)    // row_half_mirror
    RDETR_TL_STEP(0x140)    // row_mirror
#undef RDETR_TL_STEP
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const int ab = MAX ? (a > b ? a : b) : (a < b ? a : b), cd = MAX ? (c > d ? c : d) : (c < d ? c : d);
    return MAX ? (ab > cd ? ab : cd) : (ab < cd ? ab : cd);
}

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

// first pixel coordinate of a level of size `n` whose centre lies in region `r` or beyond (regions of `reg` level-0
// pixels, level-0 size n0): the smallest x with (2x + 1) * n0 >= 2 * reg * n * r, clipped to n
__device__ __forceinline__ int tl_region_begin(int r, int reg, int n, int n0)
{
    const long long v = 2ll * reg * n * r;
    const int c = (int)((v + n0 - 1) / n0);
    const int x = c / 2;
    return x < n ? x : n;
}

// bf16 high part (round to nearest even) and low part of an fp32 weight: w = hi + lo up to 2^-17 |w|
__device__ __forceinline__ void tl_split(float w, unsigned &hi, unsigned &lo)
{
    hi = f32_to_bf16_bits(w);
    lo = f32_to_bf16_bits(w - bf16_bits_to_f32(hi));
}

template <bool FUSED>
__global__ __launch_bounds__(kTlThreads) void msda_fwd_tile_kernel(
    const uint16_t *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const void *__restrict__ src_a, const void *__restrict__ src_b, const float *__restrict__ ref, int ref_dim, int S,
    int splits, int nblk, uint16_t *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    TileShared &sh = *reinterpret_cast<TileShared *>(lds);
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
    if (tid < kTlLevels * 4) sh.bbox[tid] = (tid & 2) ? INT_MIN : INT_MAX;
    if (tid < 16) reinterpret_cast<unsigned *>(lds + kTlZeroOff)[tid] = 0u;
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

    const int logical = xcd_contiguous_block(blockIdx.x, nblk);
    const int pair = logical / splits, split = logical - pair * splits;
    const int b = pair / kTlHeads, m = pair - b * kTlHeads;
    const int chunks = sh.chunks;
    const int ntiles = sh.regions_x * sh.regions_y * chunks;
    const int t0 = (int)((long long)split * ntiles / splits), t1 = (int)((long long)(split + 1) * ntiles / splits);
    if (t0 >= t1) return;                                   // uniform for the workgroup

    const unsigned char *plane = reinterpret_cast<const unsigned char *>(value) + (size_t)b * S * kTlGPixB + (size_t)m * kTlPixB;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(plane), 0, (unsigned)S * kTlGPixB - (unsigned)m * kTlPixB, 0x00020000);

    // set-up role: query qx of the wave, point pp.   gather role: lane group g, row tq / piece tp of a transposed read;
    // as an A-operand lane: row am = lane & 15 = 4 * (its group) + ar
    const int qx = lane >> 2, pp = lane & 3;
    const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int am = lane & 15, ar = am & 3;
    const bool a_active = (am >> 2) == g;
    const unsigned a_mlo = (a_active && !(ar & 1)) ? 0xffffffffu : 0u, a_mhi = (a_active && (ar & 1)) ? 0xffffffffu : 0u;
    const unsigned stage_off = (unsigned)(kTlStageOff + wave * kTlStagePerWave);
    unsigned char *stage = lds + stage_off;
    const unsigned o_rd = stage_off + (unsigned)(g * 128 + tq * 4);                       // O[2g][.][tq]; + o' * 512 + p * 16
    const unsigned w_rd = stage_off + 1024u + (unsigned)((2 * g + (ar & 1)) * 64 + (ar >> 1) * 8);
    const unsigned c0 = (unsigned)(tp * 8 + (g & 1) * 32);
    unsigned char *fgo = lds + kTlFgoOff + wave * 64;

    // ---- helpers -----------------------------------------------------------------------------------------------
    // query owned by lane >> 2 of this wave in tile t (-1 = none)
    auto query_of = [&](int t) -> int {
        const int region = t / chunks, chunk = t - region * chunks;
        const int ry = region / sh.regions_x, rx = region - ry * sh.regions_x;
        if (wave < kTlCoarseWave0) {
            const int x = rx * kTlRegW + qx, y = ry * kTlRegH + wave;
            return (chunk == 0 && x < sh.w[0] && y < sh.h[0]) ? sh.start[0] + y * sh.w[0] + x : -1;
        }
        int j = chunk * kTlCoarseSlots + (wave - kTlCoarseWave0) * 16 + qx;
        int q = -1;
#pragma unroll
        for (int l = 1; l < kTlLevels; ++l) {
            const int xa = tl_region_begin(rx, kTlRegW, sh.w[l], sh.w[0]), xb = tl_region_begin(rx + 1, kTlRegW, sh.w[l], sh.w[0]);
            const int ya = tl_region_begin(ry, kTlRegH, sh.h[l], sh.h[0]), yb = tl_region_begin(ry + 1, kTlRegH, sh.h[l], sh.h[0]);
            const int nx = xb - xa, n = nx * (yb - ya);
            if (q < 0 && j >= 0 && j < n) {
                const int yy = j / nx;
                q = sh.start[l] + (ya + yy) * sh.w[l] + xa + (j - yy * nx);
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
                    sm.xy[l].x = rp[0] + sm.xy[l].x / (float)sh.w[l];
                    sm.xy[l].y = rp[1] + sm.xy[l].y / (float)sh.h[l];
                } else {
                    sm.xy[l].x = rp[0] + sm.xy[l].x * (1.0f / kTlPoints) * rp[2] * 0.5f;
                    sm.xy[l].y = rp[1] + sm.xy[l].y * (1.0f / kTlPoints) * rp[3] * 0.5f;
                }
            }
        } else {
            const float *loc_q = static_cast<const float *>(src_a) + hrow * 2;
            const float *att_q = static_cast<const float *>(src_b) + hrow;
#pragma unroll
            for (int l = 0; l < kTlLevels; ++l) {
                const int pt = l * kTlPoints + pp;
                sm.xy[l] = *reinterpret_cast<const f32x2 *>(loc_q + 2 * pt);
                sm.a[l] = att_q[pt];
            }
        }
    };

    // bounding box, per level, of the pixels the tile's valid sample corners touch -> LDS atomics
    auto bbox_accumulate = [&](const TileSamples &sm) {
#pragma unroll
        for (int l = 0; l < kTlLevels; ++l) {
            const TileCorner c = tl_corners(sm.xy[l], sm.a[l], sm.q >= 0, sh.w[l], sh.h[l]);
            const int mnx = tl_wave_reduce<false>(c.inside ? c.xa : INT_MAX);
            const int mny = tl_wave_reduce<false>(c.inside ? c.ya : INT_MAX);
            const int mxx = tl_wave_reduce<true>(c.inside ? c.xb : INT_MIN);
            const int mxy = tl_wave_reduce<true>(c.inside ? c.yb : INT_MIN);
            if (lane == 0 && mxx >= mnx) {
                atomicMin(&sh.bbox[l * 4 + 0], mnx);
                atomicMin(&sh.bbox[l * 4 + 1], mny);
                atomicMax(&sh.bbox[l * 4 + 2], mxx);
                atomicMax(&sh.bbox[l * 4 + 3], mxy);
            }
        }
    };

    // thread l: bounding box of level l -> rect (clipped to the level's buffer), then re-arm the box
    auto compute_desc = [&](int l, int par) {
        const int mnx = sh.bbox[l * 4 + 0], mny = sh.bbox[l * 4 + 1], mxx = sh.bbox[l * 4 + 2], mxy = sh.bbox[l * 4 + 3];
        sh.bbox[l * 4 + 0] = INT_MAX;
        sh.bbox[l * 4 + 1] = INT_MAX;
        sh.bbox[l * 4 + 2] = INT_MIN;
        sh.bbox[l * 4 + 3] = INT_MIN;
        const bool in_a = l < 2;
        const int cap = in_a ? kTlCapA : kTlCapB, sqw = in_a ? kTlSqWA : kTlSqWB, sqh = in_a ? kTlSqHA : kTlSqHB;
        int rx = 0, ry = 0, rw = 2, rh = 0;
        if (mxx >= mnx) {
            const int rw0 = mxx - mnx + 1, rh0 = mxy - mny + 1;
            const int rwp = ((rw0 + 1) & ~3) + 2;              // smallest width >= rw0 that is == 2 (mod 4)
            rx = mnx; ry = mny; rw = rwp; rh = rh0;
            if (rwp * rh0 > cap) {
                if (rh0 <= sqh) rw = (((cap / rh0) - 2) & ~3) + 2;
                else if (rwp <= sqw) rh = cap / rwp;
                else { rw = sqw; rh = sqh; }
                if (rw < rw0) rx = mnx + (rw0 - rw) / 2;
                if (rh < rh0) ry = mny + (rh0 - rh) / 2;
                if (rw * rh * 4 < rw0 * rh0) rh = 0;           // would cover < 25 % of the footprint: not worth a fill
            }
        }
        sh.desc[par][l][0] = rx;
        sh.desc[par][l][1] = ry;
        sh.desc[par][l][2] = rw;
        sh.desc[par][l][3] = rh;
    };

    // DMA the rect of level l (tile parity par) into its buffer: lane = (pixel, 16-byte chunk), 16 pixels per instruction
    auto fill = [&](int l, int par) {
        const int rx = __builtin_amdgcn_readfirstlane(sh.desc[par][l][0]);
        const int ry = __builtin_amdgcn_readfirstlane(sh.desc[par][l][1]);
        const int rw = __builtin_amdgcn_readfirstlane(sh.desc[par][l][2]);
        const int rh = __builtin_amdgcn_readfirstlane(sh.desc[par][l][3]);
        const int W = sh.w[l], st = sh.start[l];
        const unsigned buf = l < 2 ? (unsigned)kTlBufAOff : (unsigned)kTlBufBOff;
        const int n4 = rw * rh * 4;
        const float inv = 1.0f / (float)rw;
        for (int i = tid; i < n4; i += kTlThreads) {
            const int px = i >> 2, c = i & 3;
            const int r = (int)(((float)px + 0.5f) * inv);
            const int cx = px - r * rw;
            int gp = st + (ry + r) * W + rx + cx;                 // columns past the level's edge (width rounding) read
            gp = gp < S ? gp : S - 1;                              // valid-but-unused pixels
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(plane + (size_t)gp * kTlGPixB + (unsigned)c * 16u),
                (__attribute__((address_space(3))) void *)(lds + buf + (unsigned)(i & ~63) * 16u), 16, 0, 0);
        }
    };

    f32x4 acc[2][2];                       // [octet o'][X]: D[4g + r][channel i + 16 ((g & 1) ^ X)] of queries 8 o' + 2g + (r & 1)

    // one MFMA step: 8 samples (queries 8 o' .. 8 o' + 7, point p) x 32 channels.
    // oa / ob = LDS offsets of this lane's row (corner tq) of samples 2g / 2g + 1
    auto mfma_step = [&](int op, int p, unsigned oa, unsigned ob, f32x4 &d0, f32x4 &d1) {
        const u32x2 aw = *reinterpret_cast<const u32x2 *>(lds + w_rd + op * 512 + p * 16);
        const u32x4 af = {aw.x & a_mlo, aw.y & a_mlo, aw.x & a_mhi, aw.y & a_mhi};
        const unsigned ba = oa + c0, bb = ob + c0;
        const tl_s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tl_s16x4 *)(lds + ba));
        const tl_s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tl_s16x4 *)(lds + bb));
        const tl_s16x4 r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tl_s16x4 *)(lds + (ba ^ 32u)));
        const tl_s16x4 r3 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tl_s16x4 *)(lds + (bb ^ 32u)));
        const u32x2 x0 = __builtin_bit_cast(u32x2, r0), x1 = __builtin_bit_cast(u32x2, r1);
        const u32x2 y0 = __builtin_bit_cast(u32x2, r2), y1 = __builtin_bit_cast(u32x2, r3);
        const u32x4 b0 = {x0.x, x0.y, x1.x, x1.y}, b1 = {y0.x, y0.y, y1.x, y1.y};
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tl_bf16x8, af), __builtin_bit_cast(tl_bf16x8, b0), d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tl_bf16x8, af), __builtin_bit_cast(tl_bf16x8, b1), d1, 0, 0, 0);
    };

    // one level of the wave's 16 queries: set-up (lane = query x point) -> staging -> MFMA loop -> patch steps for flagged samples
    auto pass = [&](int l, int par, const TileSamples &sm) {
        const int rx = __builtin_amdgcn_readfirstlane(sh.desc[par][l][0]);
        const int ry = __builtin_amdgcn_readfirstlane(sh.desc[par][l][1]);
        const int rw = __builtin_amdgcn_readfirstlane(sh.desc[par][l][2]);
        const int rh = __builtin_amdgcn_readfirstlane(sh.desc[par][l][3]);
        const int W = sh.w[l], H = sh.h[l];
        const unsigned buf = l < 2 ? (unsigned)kTlBufAOff : (unsigned)kTlBufBOff;
        const TileCorner c = tl_corners(sm.xy[l], sm.a[l], sm.q >= 0, W, H);
        const bool in_rect = c.xa >= rx && c.xb < rx + rw && c.ya >= ry && c.yb < ry + rh;
        const bool flagged = c.inside && !in_rect;
        u32x4 o = {kTlZeroOff, kTlZeroOff, kTlZeroOff, kTlZeroOff};
        if (c.inside && in_rect) {           // LDS byte offsets inside the rect
            const unsigned o00 = buf + (unsigned)((c.ya - ry) * rw + (c.xa - rx)) * kTlPixB;
            const unsigned dx = (unsigned)(c.xb - c.xa) * kTlPixB, dy = (unsigned)(c.yb - c.ya) * (unsigned)rw * kTlPixB;
            o = u32x4{o00, o00 + dx, o00 + dy, o00 + dy + dx};
        }
        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
        tl_split(c.w00, h0, l0);
        tl_split(c.w01, h1, l1);
        tl_split(c.w10, h2, l2);
        tl_split(c.w11, h3, l3);
        *reinterpret_cast<u32x4 *>(stage + lane * 16) = o;
        *reinterpret_cast<u32x4 *>(stage + 1024 + lane * 16) = u32x4{h0 | (h1 << 16), h2 | (h3 << 16), l0 | (l1 << 16), l2 | (l3 << 16)};
        unsigned long long fmask = __ballot(flagged);          // remaining flagged samples (bit = set-up lane = query * 4 + point)

        // flagged samples: publish the global byte offsets of four of them, start their row loads (lane = sample g,
        // corner tq, 16-byte chunk tp) -- the loads land behind the pass's DMA fill, i.e. by the time the MFMA loop is done
        u32x4 pre = {0u, 0u, 0u, 0u};
        const unsigned g00 = (unsigned)(sh.start[l] + c.ya * W + c.xa) * kTlGPixB;
        const unsigned gdx = (unsigned)(c.xb - c.xa) * kTlGPixB, gdy = (unsigned)(c.yb - c.ya) * (unsigned)W * kTlGPixB;
        const int frank = __builtin_amdgcn_mbcnt_hi((unsigned)(fmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fmask, 0));
        auto issue_patch_loads = [&](int first_rank) {
            if (flagged && frank >= first_rank && frank < first_rank + 4)
                *reinterpret_cast<u32x4 *>(fgo + (frank - first_rank) * 16) = u32x4{g00, g00 + gdx, g00 + gdy, g00 + gdy + gdx};
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

#pragma unroll
        for (int op = 0; op < 2; ++op) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const unsigned *orow = reinterpret_cast<const unsigned *>(lds + o_rd + op * 512 + p * 16);
                mfma_step(op, p, orow[0], orow[16], acc[op][0], acc[op][1]);
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
                const int fq = id >> 2, fp = id & 3, fs = fq & 7;
                const unsigned prow = stage_off + (unsigned)(k * 256 + tq * 64);
                const unsigned oa = (g == (fs >> 1) && !(fs & 1)) ? prow : (unsigned)kTlZeroOff;
                const unsigned ob = (g == (fs >> 1) && (fs & 1)) ? prow : (unsigned)kTlZeroOff;
                if (fq < 8) mfma_step(0, fp, oa, ob, acc[0][0], acc[0][1]);
                else mfma_step(1, fp, oa, ob, acc[1][0], acc[1][1]);
            }
            done += 4;
            if (fmask != 0ull) issue_patch_loads(done);          // more than four: next batch (its latency is exposed; rare)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // reads before the next pass's staging writes
        __builtin_amdgcn_wave_barrier();
    };

    // out[query][channel] = D[hi row] + D[lo row]; transposed through the wave's staging area so that a lane stores 16 bytes
    auto store_tile = [&](const TileSamples &sm) {
        float *tr = reinterpret_cast<float *>(stage);
#pragma unroll
        for (int op = 0; op < 2; ++op) {
#pragma unroll
            for (int X = 0; X < 2; ++X) {
                const f32x4 d = acc[op][X];
                const int ch = (lane & 15) + 16 * ((g & 1) ^ X);
                tr[(8 * op + 2 * g) * 32 + ch] = d.x + d.z;
                tr[(8 * op + 2 * g + 1) * 32 + ch] = d.y + d.w;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(tr + qx * 32 + pp * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(tr + qx * 32 + pp * 8 + 4);
        if (sm.q >= 0) {
            u32x4 w;
            w.x = f32_to_bf16_bits(lo.x) | (f32_to_bf16_bits(lo.y) << 16);
            w.y = f32_to_bf16_bits(lo.z) | (f32_to_bf16_bits(lo.w) << 16);
            w.z = f32_to_bf16_bits(hi.x) | (f32_to_bf16_bits(hi.y) << 16);
            w.w = f32_to_bf16_bits(hi.z) | (f32_to_bf16_bits(hi.w) << 16);
            *reinterpret_cast<u32x4 *>(out + ((size_t)b * Nq + sm.q) * (kTlHeads * kTlHeadDim) + m * kTlHeadDim + pp * 8) = w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // ---- pipeline ----------------------------------------------------------------------------------------------
    TileSamples cur, nxt;
    load_samples(t0, cur);
    nxt = cur;
    bbox_accumulate(cur);
    __syncthreads();
    if (tid < kTlLevels) compute_desc(tid, t0 & 1);
    __syncthreads();
    fill(0, t0 & 1);
    __syncthreads();

    for (int t = t0; t < t1; ++t) {
        const int par = t & 1;
        const bool has_next = t + 1 < t1;
        const bool busy = __ballot(cur.q >= 0) != 0ull;  // any query in this wave?
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
            for (int X = 0; X < 2; ++X) acc[op][X] = f32x4{0.f, 0.f, 0.f, 0.f};

        fill(2, par);                                   // pass 0: level 0 from A   | level 2 -> B in flight
        if (has_next) load_samples(t + 1, nxt);
        if (busy) pass(0, par, cur);
        __syncthreads();

        fill(1, par);                                   // pass 1: level 2 from B   | level 1 -> A in flight
        if (busy) pass(2, par, cur);
        if (has_next) bbox_accumulate(nxt);
        __syncthreads();

        fill(3, par);                                   // pass 2: level 1 from A   | level 3 -> B in flight
        if (has_next && tid < kTlLevels) compute_desc(tid, par ^ 1);
        if (busy) pass(1, par, cur);
        __syncthreads();

        if (has_next) fill(0, par ^ 1);                 // pass 3: level 3 from B   | next tile's level 0 -> A in flight
        if (busy) {
            pass(3, par, cur);
            store_tile(cur);
        }
        cur = nxt;
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
    if ((long long)S * kTlGPixB >= (1ll << 31)) return RDETR_ERR_UNSUPPORTED;
    auto kern = msda_fwd_tile_kernel<FUSED>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kTlLdsBytes);
    if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
    const long long pairs = (long long)B * kTlHeads;
    long long splits = 256 / pairs;                 // one resident workgroup per CU
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    const long long nblk = pairs * splits;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(kTlThreads), (size_t)kTlLdsBytes, stream, value, shapes,
                       level_start, src_a, src_b, ref, ref_dim, S, (int)splits, (int)nblk, out);
    return launch_status();
}

template int msda_tile_forward<false>(const uint16_t *, const int64_t *, const int64_t *, const void *, const void *,
                                      const float *, int, int, int, int, int, uint16_t *, hipStream_t);
template int msda_tile_forward<true>(const uint16_t *, const int64_t *, const int64_t *, const void *, const void *,
                                     const float *, int, int, int, int, int, uint16_t *, hipStream_t);

}  // namespace rdetr
