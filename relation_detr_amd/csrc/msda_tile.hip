// Multi-scale deformable attention, forward -- LDS-tiled kernel for the ENCODER shape (queries = pyramid pixels,
// Nq == S, L == 4, bf16 value) on gfx950 (MI355X).
//
// Why: the direct gather (msda_fwd.hip) pulls 64 x the value tensor through the texture path and, in bf16, is bound by
// the ~0.23 L2 requests/clk/CU an L1 that misses can sustain.  Neighbouring queries sample neighbouring
// pixels, so this kernel turns the gather into  (1) a coalesced copy of the window ("rect") of each level that a
// 16 x 16 tile of queries samples, L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPRs), and  (2) a gather out of
// LDS (ds_read_b128, 4 x the L1 rate, bank-conflict free by construction).  Nothing is assumed about the sampling
// locations: the rect of a (tile, level) is the bounding box of the tile's actual sample footprint, clipped to the
// buffer; samples that fall outside it are fetched from global memory in the same loop ("mixed" loop), so the result
// never depends on the window and a decoder-like scatter merely degrades to the direct gather's speed.
//
//   workgroup = 1024 threads = 16 waves, persistent over a contiguous range of the tiles of ONE (image, head);
//               wave w owns row w of the 16 x 16 query tile (16 queries) for the whole tile: accumulators in VGPRs.
//   LDS (160 KiB) = 1 KiB tables | 16 x 2 KiB per-wave staging | buffer A 1392 px | buffer B 640 px  (64 B / pixel)
//   passes    = one per sampled level, in the order L0 (A), L2 (B), L1 (A), L3 (B): the buffers alternate, so the
//               DMA fill of the NEXT pass (other buffer) is issued before the gather of the current one and is hidden
//               behind it; the next tile's sampling locations are loaded and their bounding boxes reduced (wave DPP
//               reduction -> LDS atomics) two passes ahead.  One barrier per pass.
//   set-up    = lane (query, point): pixel coords, 4 corner LDS offsets, 4 corner weights (bilinear x attention), staged
//               per wave as 4 x (offset, weight); a sample whose corners are not all inside the rect is flagged and
//               carries global byte offsets instead.
//   gather    = lane (sample s of 4, corner of 4, 16-byte chunk of 4): ONE ds_read_b64 (its corner's offset + weight)
//               and ONE ds_read_b128 per sample; the rect's row stride is == 2 (mod 4) pixels, so the 4 corners of a
//               sample sit in 4 different bank quarters and the 16 lanes of a ds_read_b128 lane group never collide.
//               Each lane accumulates ITS corner over the 16 points (v_pk_fma_f32); corners are summed once per tile by
//               two DPP row rotations.  Corners outside the level: weight 0, address of a valid corner (broadcast).
// Same arithmetic as msda_fwd.hip per corner; only the summation order differs (fp32 accumulate).
#include <climits>

#include "common.h"

namespace rdetr {

constexpr int kTlThreads = 1024;
constexpr int kTlWaves = kTlThreads / kWave;
constexpr int kTlTile = 16;                                   // 16 x 16 queries; one tile row per wave
constexpr int kTlHeads = 8, kTlHeadDim = 32, kTlPoints = 4, kTlLevels = 4;
constexpr unsigned kTlPixB = 64;                              // LDS bytes per pixel (one bf16 head row)
constexpr unsigned kTlGPixB = kTlHeads * kTlHeadDim * 2;      // global bytes per pixel (512)
constexpr int kTlMiscBytes = 1024;
constexpr int kTlZeroOff = 512;                               // 64 zero bytes (inside the misc area)
constexpr int kTlStageOff = kTlMiscBytes;
constexpr int kTlStagePerWave = 64 * 32;                      // 64 samples x 4 corners x {offset, weight}
constexpr int kTlBufAOff = kTlStageOff + kTlWaves * kTlStagePerWave;
constexpr int kTlCapA = 1392, kTlSqWA = 38, kTlSqHA = 36;     // pixels; "square" fallback rect of the buffer
constexpr int kTlBufBOff = kTlBufAOff + kTlCapA * (int)kTlPixB;
constexpr int kTlCapB = 640, kTlSqWB = 26, kTlSqHB = 24;
constexpr int kTlLdsBytes = kTlBufBOff + kTlCapB * (int)kTlPixB;
static_assert(kTlLdsBytes == 160 * 1024, "LDS map must fill exactly 160 KiB");

struct TileShared {
    int h[kTlLevels], w[kTlLevels], start[kTlLevels];
    int tiles_x[kTlLevels];
    int tile_base[kTlLevels + 1];          // cumulative tile count per query level
    int bbox[kTlLevels * 4];               // per level: min x, min y, max x, max y of the next tile's valid sample corners
    int desc[2][kTlLevels][4];             // per tile parity and level: rect x, y, width (== 2 mod 4), height (0 = none)
};
static_assert(sizeof(TileShared) <= kTlZeroOff, "tables overlap the zero row");

struct TileSamples {                       // lane (query = lane >> 2, point = lane & 3): its sample in each level
    f32x2 xy[kTlLevels];
    float a[kTlLevels];
    bool qok;
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
template <int CTRL> __device__ __forceinline__ float tl_row_rot_add(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
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
        int base = 0;
        for (int l = 0; l < kTlLevels; ++l) {
            const int h = (int)shapes[2 * l], w = (int)shapes[2 * l + 1];
            sh.h[l] = h;
            sh.w[l] = w;
            sh.start[l] = (int)level_start[l];
            sh.tiles_x[l] = (w + kTlTile - 1) / kTlTile;
            sh.tile_base[l] = base;
            base += sh.tiles_x[l] * ((h + kTlTile - 1) / kTlTile);
        }
        sh.tile_base[kTlLevels] = base;
    }
    if (tid < kTlLevels * 4) sh.bbox[tid] = (tid & 2) ? INT_MIN : INT_MAX;
    if (tid < 16) reinterpret_cast<unsigned *>(lds + kTlZeroOff)[tid] = 0u;
    __syncthreads();

    const int logical = xcd_contiguous_block(blockIdx.x, nblk);
    const int pair = logical / splits, split = logical - pair * splits;
    const int b = pair / kTlHeads, m = pair - b * kTlHeads;
    const int ntiles = sh.tile_base[kTlLevels];
    const int t0 = (int)((long long)split * ntiles / splits), t1 = (int)((long long)(split + 1) * ntiles / splits);
    if (t0 >= t1) return;                                   // uniform for the workgroup

    const unsigned char *plane = reinterpret_cast<const unsigned char *>(value) + (size_t)b * S * kTlGPixB + (size_t)m * kTlPixB;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(plane), 0, (unsigned)S * kTlGPixB - (unsigned)m * kTlPixB, 0x00020000);

    // set-up role: query qx of the wave's tile row, point pp;  gather role: sample gs, corner gc, chunk gk
    const int qx = lane >> 2, pp = lane & 3;
    const int gs = lane >> 4, gc = (lane >> 2) & 3, gk = lane & 3;
    unsigned char *stage = lds + kTlStageOff + wave * kTlStagePerWave;
    const unsigned stage_rd = (unsigned)(kTlStageOff + wave * kTlStagePerWave + gs * 128 + gc * 8);
    const unsigned chunk16 = (unsigned)gk * 16u;

    // ---- helpers -----------------------------------------------------------------------------------------------
    auto tile_geom = [&](int t, int &lq, int &x0, int &y0) {
        lq = 0;
#pragma unroll
        for (int l = 1; l < kTlLevels; ++l) lq = (t >= sh.tile_base[l]) ? l : lq;
        const int r = t - sh.tile_base[lq];
        const int ty = r / sh.tiles_x[lq];
        x0 = (r - ty * sh.tiles_x[lq]) * kTlTile;
        y0 = ty * kTlTile;
    };

    // sampling locations / attention weights of this lane's (query, point) in every level
    auto load_samples = [&](int t, TileSamples &sm) {
        int lq, x0, y0;
        tile_geom(t, lq, x0, y0);
        const int x = x0 + qx, y = y0 + wave;
        sm.qok = x < sh.w[lq] && y < sh.h[lq];
        const int q = sm.qok ? sh.start[lq] + y * sh.w[lq] + x : 0;
        const size_t row = (size_t)b * Nq + q;
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
            const TileCorner c = tl_corners(sm.xy[l], sm.a[l], sm.qok, sh.w[l], sh.h[l]);
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

    f32x2 acc[4][4];

    // one level of the wave's 16 queries: set-up (lane = query x point) -> staging -> gather (lane = sample x corner x chunk)
    auto pass = [&](int l, int par, const TileSamples &sm) {
        const int rx = __builtin_amdgcn_readfirstlane(sh.desc[par][l][0]);
        const int ry = __builtin_amdgcn_readfirstlane(sh.desc[par][l][1]);
        const int rw = __builtin_amdgcn_readfirstlane(sh.desc[par][l][2]);
        const int rh = __builtin_amdgcn_readfirstlane(sh.desc[par][l][3]);
        const int W = sh.w[l], H = sh.h[l];
        const unsigned buf = l < 2 ? (unsigned)kTlBufAOff : (unsigned)kTlBufBOff;
        const TileCorner c = tl_corners(sm.xy[l], sm.a[l], sm.qok, W, H);
        const bool in_rect = c.xa >= rx && c.xb < rx + rw && c.ya >= ry && c.yb < ry + rh;
        const bool flagged = c.inside && !in_rect;
        unsigned o00, dx, dy;
        if (flagged) {                       // global byte offsets (bit 31 marks them)
            o00 = 0x80000000u | ((unsigned)(sh.start[l] + c.ya * W + c.xa) * kTlGPixB);
            dx = (unsigned)(c.xb - c.xa) * kTlGPixB;
            dy = (unsigned)(c.yb - c.ya) * (unsigned)W * kTlGPixB;
        } else if (c.inside) {               // LDS byte offsets inside the rect
            o00 = buf + (unsigned)((c.ya - ry) * rw + (c.xa - rx)) * kTlPixB;
            dx = (unsigned)(c.xb - c.xa) * kTlPixB;
            dy = (unsigned)(c.yb - c.ya) * (unsigned)rw * kTlPixB;
        } else {
            o00 = kTlZeroOff; dx = 0; dy = 0;
        }
        u32x4 r0, r1;
        r0.x = o00;           r0.y = __builtin_bit_cast(unsigned, c.w00);
        r0.z = o00 + dx;      r0.w = __builtin_bit_cast(unsigned, c.w01);
        r1.x = o00 + dy;      r1.y = __builtin_bit_cast(unsigned, c.w10);
        r1.z = o00 + dy + dx; r1.w = __builtin_bit_cast(unsigned, c.w11);
        *reinterpret_cast<u32x4 *>(stage + lane * 32) = r0;
        *reinterpret_cast<u32x4 *>(stage + lane * 32 + 16) = r1;
        const unsigned long long fmask = __ballot(flagged);
        // staging is private to the wave and LDS operations of one wave complete in order: wave-level fences suffice
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        if (fmask == 0ull) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const u32x2 rec = *reinterpret_cast<const u32x2 *>(lds + stage_rd + (16 * j + p) * 32);
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(lds + rec.x + chunk16);
                    const float w = __builtin_bit_cast(f32x2, rec).y;      // never bit_cast a vector COMPONENT (clang reads .x)
                    const f32x2 wv = {w, w};
                    acc[j][0] = __builtin_elementwise_fma(wv, f32x2{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u)}, acc[j][0]);
                    acc[j][1] = __builtin_elementwise_fma(wv, f32x2{__builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u)}, acc[j][1]);
                    acc[j][2] = __builtin_elementwise_fma(wv, f32x2{__builtin_bit_cast(float, v.z << 16), __builtin_bit_cast(float, v.z & 0xffff0000u)}, acc[j][2]);
                    acc[j][3] = __builtin_elementwise_fma(wv, f32x2{__builtin_bit_cast(float, v.w << 16), __builtin_bit_cast(float, v.w & 0xffff0000u)}, acc[j][3]);
                }
            }
        } else {                             // some sample of this wave lies outside the rect: those lanes read global memory
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const u32x2 rec = *reinterpret_cast<const u32x2 *>(lds + stage_rd + (16 * j + p) * 32);
                    const bool g = (int)rec.x < 0;
                    const unsigned lo = g ? (unsigned)kTlZeroOff : rec.x;
                    const unsigned go = g ? (rec.x & 0x7fffffffu) + chunk16 : 0x80000000u;   // out of range -> 0, no request
                    const u32x4 vl = *reinterpret_cast<const u32x4 *>(lds + lo + chunk16);
                    const u32x4 vg = __builtin_amdgcn_raw_buffer_load_b128(rsrc, go, 0, 0);
                    const u32x4 v = vl | vg;                                                  // one of the two is all zero
                    const float w = __builtin_bit_cast(f32x2, rec).y;      // never bit_cast a vector COMPONENT (clang reads .x)
                    const f32x2 wv = {w, w};
                    acc[j][0] = __builtin_elementwise_fma(wv, f32x2{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u)}, acc[j][0]);
                    acc[j][1] = __builtin_elementwise_fma(wv, f32x2{__builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u)}, acc[j][1]);
                    acc[j][2] = __builtin_elementwise_fma(wv, f32x2{__builtin_bit_cast(float, v.z << 16), __builtin_bit_cast(float, v.z & 0xffff0000u)}, acc[j][2]);
                    acc[j][3] = __builtin_elementwise_fma(wv, f32x2{__builtin_bit_cast(float, v.w << 16), __builtin_bit_cast(float, v.w & 0xffff0000u)}, acc[j][3]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // gather reads before the next pass's staging writes
        __builtin_amdgcn_wave_barrier();
    };

    // sum the four corner lanes of every sample and write the wave's 16 output rows (64 B each)
    auto store_tile = [&](int t) {
        int lq, x0, y0;
        tile_geom(t, lq, x0, y0);
        const int y = y0 + wave;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float r[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[2 * i] = tl_row_rot_add<0x124>(tl_row_rot_add<0x128>(acc[j][i].x));     // row_ror:8 then row_ror:4
                r[2 * i + 1] = tl_row_rot_add<0x124>(tl_row_rot_add<0x128>(acc[j][i].y));
            }
            const int x = x0 + 4 * j + gs;
            if (gc == 0 && x < sh.w[lq] && y < sh.h[lq]) {
                const size_t row = (size_t)b * Nq + (sh.start[lq] + y * sh.w[lq] + x);
                u32x4 o;
                o.x = f32_to_bf16_bits(r[0]) | (f32_to_bf16_bits(r[1]) << 16);
                o.y = f32_to_bf16_bits(r[2]) | (f32_to_bf16_bits(r[3]) << 16);
                o.z = f32_to_bf16_bits(r[4]) | (f32_to_bf16_bits(r[5]) << 16);
                o.w = f32_to_bf16_bits(r[6]) | (f32_to_bf16_bits(r[7]) << 16);
                *reinterpret_cast<u32x4 *>(out + row * (kTlHeads * kTlHeadDim) + m * kTlHeadDim + gk * 8) = o;
            }
        }
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
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = f32x2{0.f, 0.f};

        fill(2, par);                                   // pass 0: level 0 from A   | level 2 -> B in flight
        if (has_next) load_samples(t + 1, nxt);
        pass(0, par, cur);
        __syncthreads();

        fill(1, par);                                   // pass 1: level 2 from B   | level 1 -> A in flight
        pass(2, par, cur);
        if (has_next) bbox_accumulate(nxt);
        __syncthreads();

        fill(3, par);                                   // pass 2: level 1 from A   | level 3 -> B in flight
        if (has_next && tid < kTlLevels) compute_desc(tid, par ^ 1);
        pass(1, par, cur);
        __syncthreads();

        if (has_next) fill(0, par ^ 1);                 // pass 3: level 3 from B   | next tile's level 0 -> A in flight
        pass(3, par, cur);
        store_tile(t);
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
