// Multi-scale deformable attention, forward, ENCODER shape (queries = the pyramid's own pixels, Nq == S)
// -- LDS "sweep" kernel for gfx950 (MI355X).
//
// Why: the direct gather (msda_fwd.hip) moves 64 x the value tensor through the texture path
// (<= 64 B/clk/CU, L2->L1 fills ~9 TB/s chip-wide) and tops out near 15 % of the HBM roofline
// (profiles/r01).  In the encoder a query at pixel (x, y) samples every level within a few pixels of its
// own position, so a workgroup that walks down a vertical strip of the image can keep, for every level,
// a sliding band of rows in LDS and serve (almost) every bilinear corner from LDS (4x the L1 rate),
// loading each value row from L2/HBM once per strip.
//
// Decomposition
//   workgroup = (image b, head m, [channel half], strip sx, segment of steps); 512 threads.
//   The normalised image is cut into `nsx` vertical strips and `nstep` horizontal steps; a query (pixel
//   of any level) belongs to the strip / step that contains its centre.  For level l the band of step t
//   is rows [ylo, yhi) = rows within `margin` pixels of the step, x-window [wx0, wx1) likewise; rows live
//   in a ring of `rc` row slots (slot = (y + 1) mod rc), 64 B per pixel (fp32: 16 channels = half a head,
//   so fp32 runs two workgroups per head; bf16: the whole 32-channel head).  Pixels outside the level are
//   stored as zeros (the range-checked buffer load returns 0), so zero padding needs no tests.
//   While step t is computed the rows that step t+1 adds are fetched into registers and written to their
//   (free) ring slots before the single barrier that ends the step.
//   Compute: a wave pass handles 16 queries x 4 lanes; lane `sub` prepares point `sub` of each level and
//   quad-broadcasts it (DPP), every corner is one ds_read_b128.  A sample whose 2x2 footprint is not in the
//   band (offset larger than the margin) falls back to the range-checked global load -- results never
//   depend on the margin, only speed does.
//   The FUSED form also performs softmax(L*P) and the location arithmetic (see msda_fwd.hip).
#include <algorithm>

#include "common.h"

namespace rdetr {

constexpr int kSwThreads = 512;
constexpr int kSwWaves = kSwThreads / kWave;
constexpr int kSwHeads = 8, kSwHeadDim = 32, kSwPoints = 4;
constexpr int kSwRowBytes = 64;              // LDS bytes per pixel
constexpr int kSwMaxNewLoads = 4;            // 16-byte loads per thread for the rows one step adds
constexpr unsigned kSwInvalid = 0x80000000u;
constexpr int kSwTableBytes = 1152;          // step tables, level tables and a 128-byte zero row
constexpr int kSwZeroOff = 1024;
constexpr int kSwMaxLevels = 8;

struct SweepLevel {
    int h, w, start;
    int pitch;      // ring row pitch in pixels (max window width over strips)
    int rc;         // ring rows
    int base;       // LDS byte offset of the ring
};
struct SweepArgs {
    SweepLevel lv[kSwMaxLevels];
    int L, nsx, nstep, nseg, margin, S, ref_dim, nblk, B;
};
struct StepRow {
    int ylo, yhi, slot0, qy0, qy1, pre;      // band rows, ring slot of ylo, query rows, queries in lower levels
    int newpre, pad;                         // 16-byte load units (this step -> next) in lower levels
};

__host__ __device__ inline int sw_floordiv(int a, int b)
{
    const int q = a / b, r = a % b;
    return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}
// first pixel of part t when n pixels are split into N parts by pixel centre
__host__ __device__ inline int sw_part_first(int t, int n, int N)
{
    const int v = -sw_floordiv(-(2 * t * n - N), 2 * N);
    return v < 0 ? 0 : (v > n ? n : v);
}
__host__ __device__ inline int sw_win_lo(int t, int n, int N, int M)
{
    const int v = sw_floordiv(2 * t * n - N, 2 * N) - M;
    return v < -1 ? -1 : v;
}
__host__ __device__ inline int sw_win_hi(int t, int n, int N, int M)
{
    const int v = sw_floordiv(2 * (t + 1) * n - N, 2 * N) + M + 2;
    return v > n + 1 ? n + 1 : v;
}

template <int P> __device__ __forceinline__ int quad_bcast(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, P * 0x55, 0xf, 0xf, false);       // quad_perm [P,P,P,P]
}
template <int P> __device__ __forceinline__ float quad_bcast(float v)
{
    return __builtin_bit_cast(float, quad_bcast<P>(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float quad_max(float v)
{
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false)));
    return v;
}
__device__ __forceinline__ float quad_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));
    return v;
}
__device__ __forceinline__ u32x4 vsel(bool c, u32x4 a, u32x4 b)
{
    return u32x4{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w};
}
// exact for 0 <= a < 2^20, 1 <= b < 2^12: (a + 0.5) / b is never within rounding distance of an integer
__device__ __forceinline__ int small_div(int a, int b) { return (int)(((float)a + 0.5f) / (float)b); }

template <typename T> struct SweepIO;
template <> struct SweepIO<float> {
    static constexpr int kHalves = 2, kCh = 4;
    static __device__ __forceinline__ void unpack(u32x4 r, float (&v)[4])
    {
        // NB: __builtin_bit_cast(float, r.y) on a vector COMPONENT miscompiles (ROCm 7.2 clang reads r.x):
        // cast the whole vector instead.
        const f32x4 f = __builtin_bit_cast(f32x4, r);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    }
    static __device__ __forceinline__ u32x4 pack(const float (&a)[4])
    {
        return __builtin_bit_cast(u32x4, f32x4{a[0], a[1], a[2], a[3]});
    }
};
template <> struct SweepIO<uint16_t> {
    static constexpr int kHalves = 1, kCh = 8;
    static __device__ __forceinline__ void unpack(u32x4 r, float (&v)[8])
    {
        v[0] = __builtin_bit_cast(float, r.x << 16); v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, r.y << 16); v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
        v[4] = __builtin_bit_cast(float, r.z << 16); v[5] = __builtin_bit_cast(float, r.z & 0xffff0000u);
        v[6] = __builtin_bit_cast(float, r.w << 16); v[7] = __builtin_bit_cast(float, r.w & 0xffff0000u);
    }
    static __device__ __forceinline__ u32x4 pack(const float (&a)[8])
    {
        u32x4 o;
        o.x = f32_to_bf16_bits(a[0]) | (f32_to_bf16_bits(a[1]) << 16);
        o.y = f32_to_bf16_bits(a[2]) | (f32_to_bf16_bits(a[3]) << 16);
        o.z = f32_to_bf16_bits(a[4]) | (f32_to_bf16_bits(a[5]) << 16);
        o.w = f32_to_bf16_bits(a[6]) | (f32_to_bf16_bits(a[7]) << 16);
        return o;
    }
};

template <typename Q> __device__ __forceinline__ float sw_load_q(const Q *p);
template <> __device__ __forceinline__ float sw_load_q<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float sw_load_q<uint16_t>(const uint16_t *p) { return bf16_bits_to_f32(*p); }
template <typename Q> __device__ __forceinline__ f32x2 sw_load_q2(const Q *p);
template <> __device__ __forceinline__ f32x2 sw_load_q2<float>(const float *p) { return *reinterpret_cast<const f32x2 *>(p); }
template <> __device__ __forceinline__ f32x2 sw_load_q2<uint16_t>(const uint16_t *p)
{
    const unsigned u = *reinterpret_cast<const unsigned *>(p);
    return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}

// Uniform (same in every lane) value that came through a VGPR: move it to an SGPR.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct LevelStatic {
    int wx0, wx1, qx0, qxw;
};

// Per-lane inputs of one 16-query group: point `sub` of every level (raw projection outputs when FUSED).
template <int LMAX> struct GroupIn {
    int q;
    bool qok;
    f32x2 xy[LMAX];      // sampling location, or raw offset (FUSED)
    float a[LMAX];       // attention weight, or raw logit (FUSED)
    f32x2 rxy[LMAX];     // reference point x, y (FUSED only)
};

template <typename T, bool FUSED, int LMAX>
__global__ __launch_bounds__(kSwThreads) void msda_fwd_sweep_kernel(const T *__restrict__ value,
                                                                    const void *__restrict__ src_a,
                                                                    const void *__restrict__ src_b,
                                                                    const float *__restrict__ ref, T *__restrict__ out,
                                                                    const SweepArgs A)
{
    using IO = SweepIO<T>;
    constexpr int NH = IO::kHalves, CH = IO::kCh;
    constexpr unsigned PIXB = kSwHeads * kSwHeadDim * sizeof(T), HEADB = kSwHeadDim * sizeof(T);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    StepRow *table = reinterpret_cast<StepRow *>(lds);                               // [3][L]  (<= 768 B)
    LevelStatic *lstat = reinterpret_cast<LevelStatic *>(lds + 800);                 // [L]     (<= 128 B)
    int *new_total = reinterpret_cast<int *>(lds + 784);                             // [3]

    const int L = A.L;
    const int LP = L * kSwPoints;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int slot = lane >> 2, sub = lane & 3;

    // ---- which (image, head, half, strip, segment) ------------------------------------------------
    int id = xcd_contiguous_block(blockIdx.x, A.nblk);
    const int seg = id % A.nseg; id /= A.nseg;
    const int sx = id % A.nsx;   id /= A.nsx;
    const int half = id % NH;    id /= NH;
    const int m = id % kSwHeads;
    const int b = id / kSwHeads;
    const int t0 = (int)((long long)seg * A.nstep / A.nseg), t1 = (int)((long long)(seg + 1) * A.nstep / A.nseg);
    const int M = A.margin;
    const int Nq = A.S;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T *>(value) + (size_t)b * A.S * (kSwHeads * kSwHeadDim), 0, (unsigned)A.S * PIXB, 0x00020000);
    const unsigned chan_off = (unsigned)m * HEADB + (unsigned)half * (NH == 2 ? kSwRowBytes : 0);

    if (tid < 32) reinterpret_cast<unsigned *>(lds + kSwZeroOff)[tid] = 0u;         // 128-byte zero row
    if (tid < L) {
        LevelStatic ls;
        ls.wx0 = sw_win_lo(sx, A.lv[tid].w, A.nsx, M);
        ls.wx1 = sw_win_hi(sx, A.lv[tid].w, A.nsx, M);
        ls.qx0 = sw_part_first(sx, A.lv[tid].w, A.nsx);
        ls.qxw = sw_part_first(sx + 1, A.lv[tid].w, A.nsx) - ls.qx0;
        lstat[tid] = ls;
    }
    __syncthreads();

    // one lane of the LAST wave (normally the idle one) prepares the per-level rows of step t: band, query rows,
    // running query count, running count of the 16-byte units step t+1 will add
    auto fill_table = [&](int t) {
        if (tid == kSwThreads - kWave && t < t1) {
            int pre = 0, newpre = 0;
            for (int l = 0; l < L; ++l) {
                StepRow r;
                r.ylo = sw_win_lo(t, A.lv[l].h, A.nstep, M);
                r.yhi = sw_win_hi(t, A.lv[l].h, A.nstep, M);
                r.slot0 = (r.ylo + 1) % A.lv[l].rc;
                r.qy0 = sw_part_first(t, A.lv[l].h, A.nstep);
                r.qy1 = sw_part_first(t + 1, A.lv[l].h, A.nstep);
                r.pre = pre;
                r.newpre = newpre;
                r.pad = 0;
                pre += lstat[l].qxw * (r.qy1 - r.qy0);
                if (t + 1 < t1)
                    newpre += (sw_win_hi(t + 1, A.lv[l].h, A.nstep, M) - r.yhi) * (lstat[l].wx1 - lstat[l].wx0) * 4;
                table[(t % 3) * L + l] = r;
            }
            new_total[t % 3] = newpre;
        }
    };
    // global byte offset of this lane's 16 bytes of pixel (x, y) of a level, or the invalid marker
    auto pixel_offset = [&](int lh, int lw, int lstart, int x, int y, int s16) -> unsigned {
        const bool in = x >= 0 && y >= 0 && x < lw && y < lh;
        return in ? (unsigned)(lstart + y * lw + x) * PIXB + chan_off + (unsigned)s16 * 16u : kSwInvalid;
    };

    fill_table(t0);
    fill_table(t0 + 1);
    __syncthreads();

    // ---- initial band: rows [ylo(t0), yhi(t0)) of every level ----------------------------------------
    for (int l = 0; l < L; ++l) {
        const int ylo = uni(table[(t0 % 3) * L + l].ylo), yhi = uni(table[(t0 % 3) * L + l].yhi);
        const int wx0 = uni(lstat[l].wx0), ww = uni(lstat[l].wx1) - wx0;
        const int lh = A.lv[l].h, lw = A.lv[l].w, lstart = A.lv[l].start, rc = A.lv[l].rc;
        const int lbase = A.lv[l].base, pitch = A.lv[l].pitch;
        const int n = (yhi - ylo) * ww * 4;
        for (int u = tid; u < n; u += kSwThreads) {
            const int px = u >> 2, s16 = u & 3;
            const int ry = small_div(px, ww), cx = px - ry * ww;
            const int y = ylo + ry;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, pixel_offset(lh, lw, lstart, wx0 + cx, y, s16), 0, 0);
            const int rs = (y + 1) - small_div(y + 1, rc) * rc;
            *reinterpret_cast<u32x4 *>(lds + lbase + (rs * pitch + cx) * kSwRowBytes + s16 * 16) = v;
        }
    }
    __syncthreads();

    auto step_queries = [&](const StepRow *rows) {
        return uni(rows[L - 1].pre) + uni(lstat[L - 1].qxw) * (uni(rows[L - 1].qy1) - uni(rows[L - 1].qy0));
    };
    // issue (do not wait for) the global loads of group g of the step described by `rows`
    auto load_group = [&](const StepRow *rows, int g) {
        GroupIn<LMAX> gi;
        const int i = g * 16 + slot;
        gi.qok = i < step_queries(rows);
        gi.q = 0;
        for (int l = 0; l < L; ++l) {
            const int pre = uni(rows[l].pre), qxw = uni(lstat[l].qxw);
            const int cnt = qxw * (uni(rows[l].qy1) - uni(rows[l].qy0));
            if (gi.qok && i >= pre && i < pre + cnt) {
                const int r = i - pre;
                const int ry = small_div(r, qxw);
                gi.q = A.lv[l].start + (uni(rows[l].qy0) + ry) * A.lv[l].w + uni(lstat[l].qx0) + (r - ry * qxw);
            }
        }
        const size_t row = (size_t)b * Nq + gi.q;
        const size_t hrow = (row * kSwHeads + m) * (size_t)LP;
#pragma unroll
        for (int k = 0; k < LMAX; ++k) {
            const int kk = k < L ? k : 0;
            if constexpr (FUSED) {
                gi.xy[k] = sw_load_q2<T>(static_cast<const T *>(src_a) + (hrow + kk * kSwPoints + sub) * 2);
                gi.a[k] = sw_load_q<T>(static_cast<const T *>(src_b) + hrow + kk * kSwPoints + sub);
                gi.rxy[k] = *reinterpret_cast<const f32x2 *>(ref + (row * L + kk) * (size_t)A.ref_dim);
            } else {
                gi.xy[k] = *reinterpret_cast<const f32x2 *>(static_cast<const float *>(src_a) + (hrow + kk * kSwPoints + sub) * 2);
                gi.a[k] = static_cast<const float *>(src_b)[hrow + kk * kSwPoints + sub];
                gi.rxy[k] = f32x2{0.f, 0.f};
            }
        }
        return gi;
    };

    GroupIn<LMAX> gcur = load_group(table + (t0 % 3) * L, wave), gnext = gcur;

    for (int t = t0; t < t1; ++t) {
        const StepRow *cur = table + (t % 3) * L;
        const StepRow *nxt = table + ((t + 1) % 3) * L;
        const bool has_next = t + 1 < t1;

        // ---- (1) fetch the rows step t+1 adds: [yhi(t), yhi(t+1)) of every level -> registers ------------
        u32x4 nv[kSwMaxNewLoads];
        int naddr[kSwMaxNewLoads];
#pragma unroll
        for (int k = 0; k < kSwMaxNewLoads; ++k) naddr[k] = -1;
        const int n_new = has_next ? uni(new_total[t % 3]) : 0;
#pragma unroll
        for (int k = 0; k < kSwMaxNewLoads; ++k) {
            if (k * kSwThreads < n_new) {                               // wave-uniform
                const int u = tid + k * kSwThreads;
                int l = 0;
                for (int j = 1; j < L; ++j) l = (u >= cur[j].newpre) ? j : l;
                const int v = u - cur[l].newpre;
                const int wx0 = lstat[l].wx0, ww = lstat[l].wx1 - wx0;
                const int px = v >> 2, s16 = v & 3;
                const int ry = small_div(px, ww), cx = px - ry * ww;
                const int y = cur[l].yhi + ry;
                const bool live = u < n_new;
                const int rc = A.lv[l].rc;
                const int rs = (y + 1) - small_div(y + 1, rc) * rc;
                nv[k] = __builtin_amdgcn_raw_buffer_load_b128(
                    rsrc, live ? pixel_offset(A.lv[l].h, A.lv[l].w, A.lv[l].start, wx0 + cx, y, s16) : kSwInvalid, 0, 0);
                naddr[k] = live ? A.lv[l].base + (rs * A.lv[l].pitch + cx) * kSwRowBytes + s16 * 16 : -1;
            }
        }

        // ---- (2) compute the queries of step t ---------------------------------------------------------
        const int ngroups = (step_queries(cur) + 15) >> 4;
        if (has_next) gnext = load_group(nxt, wave);      // first group of step t+1: hides the HBM latency of loc / weights

        for (int g = wave; g < ngroups; g += kSwWaves) {
            const GroupIn<LMAX> gi = (g == wave) ? gcur : load_group(cur, g);
            const size_t row = (size_t)b * Nq + gi.q;
            const size_t hrow = (row * kSwHeads + m) * (size_t)LP;
            const bool qok = gi.qok;

            float smax = 0.f, sinv = 1.f;
            if constexpr (FUSED) {                         // softmax statistics over the L*P logits of (q, m)
                smax = -__builtin_inff();
#pragma unroll
                for (int k = 0; k < LMAX; ++k) smax = fmaxf(smax, k < L ? gi.a[k] : -__builtin_inff());
                smax = quad_max(smax);
                float ssum = 0.f;
#pragma unroll
                for (int k = 0; k < LMAX; ++k) ssum += k < L ? expf(gi.a[k] - smax) : 0.f;
                sinv = quad_sum(ssum);
            }

            float acc[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = 0.f;

#pragma unroll
            for (int k = 0; k < LMAX; ++k) {
                if (k >= L) break;
                // lane `sub` prepares point `sub` of level k
                const int h = A.lv[k].h, w = A.lv[k].w, lstart = A.lv[k].start;
                const int rc = A.lv[k].rc, pitch = A.lv[k].pitch, lbase = A.lv[k].base;
                const int wx0 = uni(lstat[k].wx0), wx1 = uni(lstat[k].wx1);
                const int ylo = uni(cur[k].ylo), yhi = uni(cur[k].yhi), slot0 = uni(cur[k].slot0);
                f32x2 xy = gi.xy[k];
                float a = gi.a[k];
                if constexpr (FUSED) {
                    a = expf(a - smax) / sinv;
                    if (A.ref_dim == 2) {
                        xy.x = gi.rxy[k].x + xy.x / (float)w;
                        xy.y = gi.rxy[k].y + xy.y / (float)h;
                    } else {
                        const float *rp = ref + (row * L + k) * 4;
                        xy.x = gi.rxy[k].x + xy.x * (1.0f / kSwPoints) * rp[2] * 0.5f;
                        xy.y = gi.rxy[k].y + xy.y * (1.0f / kSwPoints) * rp[3] * 0.5f;
                    }
                }
                const float x = xy.x * (float)w - 0.5f;
                const float y = xy.y * (float)h - 0.5f;
                const bool inside = qok && (y > -1.f) && (x > -1.f) && (y < (float)h) && (x < (float)w);
                const float xf = floorf(x), yf = floorf(y);
                const int x0 = inside ? (int)xf : 0, y0 = inside ? (int)yf : 0;
                const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
                const float w00 = inside ? hy * hx * a : 0.f, w01 = inside ? hy * lx * a : 0.f;
                const float w10 = inside ? ly * hx * a : 0.f, w11 = inside ? ly * lx * a : 0.f;
                const bool inwin = x0 >= wx0 && x0 + 1 < wx1 && y0 >= ylo && y0 + 1 < yhi;
                int r0 = slot0 + (y0 - ylo);
                r0 = r0 >= rc ? r0 - rc : r0;
                const int r1 = (r0 + 1 == rc) ? 0 : r0 + 1;
                const bool use_lds = inside && inwin;
                const int la = use_lds ? lbase + (r0 * pitch + (x0 - wx0)) * kSwRowBytes : kSwZeroOff;
                const int lb = use_lds ? lbase + (r1 * pitch + (x0 - wx0)) * kSwRowBytes : kSwZeroOff;
                const int fb = (inside && !inwin) ? 1 : 0;

                auto one_point = [&](int la_p, int lb_p, float w00_p, float w01_p, float w10_p, float w11_p, int fb_p,
                                     auto bcast_xy) {
                    u32x4 r00 = *reinterpret_cast<const u32x4 *>(lds + la_p + sub * 16);
                    u32x4 r01 = *reinterpret_cast<const u32x4 *>(lds + la_p + kSwRowBytes + sub * 16);
                    u32x4 r10 = *reinterpret_cast<const u32x4 *>(lds + lb_p + sub * 16);
                    u32x4 r11 = *reinterpret_cast<const u32x4 *>(lds + lb_p + kSwRowBytes + sub * 16);
                    if (__builtin_amdgcn_ballot_w64(fb_p != 0) != 0) {       // rare: footprint outside the band
                        const bool f = fb_p != 0;
                        int x0_p, y0_p;
                        bcast_xy(x0_p, y0_p);
                        const u32x4 g00 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, f ? pixel_offset(h, w, lstart, x0_p, y0_p, sub) : kSwInvalid, 0, 0);
                        const u32x4 g01 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, f ? pixel_offset(h, w, lstart, x0_p + 1, y0_p, sub) : kSwInvalid, 0, 0);
                        const u32x4 g10 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, f ? pixel_offset(h, w, lstart, x0_p, y0_p + 1, sub) : kSwInvalid, 0, 0);
                        const u32x4 g11 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, f ? pixel_offset(h, w, lstart, x0_p + 1, y0_p + 1, sub) : kSwInvalid, 0, 0);
                        // component-wise: `bool ? vec : vec` on ext_vector types is NOT a whole-vector select in clang
                        r00 = vsel(f, g00, r00); r01 = vsel(f, g01, r01); r10 = vsel(f, g10, r10); r11 = vsel(f, g11, r11);
                    }
                    float v00[CH], v01[CH], v10[CH], v11[CH];
                    IO::unpack(r00, v00); IO::unpack(r01, v01); IO::unpack(r10, v10); IO::unpack(r11, v11);
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        acc[c] += w00_p * v00[c];
                        acc[c] += w01_p * v01[c];
                        acc[c] += w10_p * v10[c];
                        acc[c] += w11_p * v11[c];
                    }
                };
                one_point(quad_bcast<0>(la), quad_bcast<0>(lb), quad_bcast<0>(w00), quad_bcast<0>(w01), quad_bcast<0>(w10),
                          quad_bcast<0>(w11), quad_bcast<0>(fb), [&](int &xx, int &yy) { xx = quad_bcast<0>(x0); yy = quad_bcast<0>(y0); });
                one_point(quad_bcast<1>(la), quad_bcast<1>(lb), quad_bcast<1>(w00), quad_bcast<1>(w01), quad_bcast<1>(w10),
                          quad_bcast<1>(w11), quad_bcast<1>(fb), [&](int &xx, int &yy) { xx = quad_bcast<1>(x0); yy = quad_bcast<1>(y0); });
                one_point(quad_bcast<2>(la), quad_bcast<2>(lb), quad_bcast<2>(w00), quad_bcast<2>(w01), quad_bcast<2>(w10),
                          quad_bcast<2>(w11), quad_bcast<2>(fb), [&](int &xx, int &yy) { xx = quad_bcast<2>(x0); yy = quad_bcast<2>(y0); });
                one_point(quad_bcast<3>(la), quad_bcast<3>(lb), quad_bcast<3>(w00), quad_bcast<3>(w01), quad_bcast<3>(w10),
                          quad_bcast<3>(w11), quad_bcast<3>(fb), [&](int &xx, int &yy) { xx = quad_bcast<3>(x0); yy = quad_bcast<3>(y0); });
            }
            if (qok) {
                T *o = out + row * (kSwHeads * kSwHeadDim) + m * kSwHeadDim + half * (NH == 2 ? 16 : 0) + sub * CH;
                *reinterpret_cast<u32x4 *>(o) = IO::pack(acc);
            }
        }
        gcur = gnext;

        // ---- (3) park the fetched rows in their (free) ring slots, (4) table for step t+2 -----------------
        if (has_next) {
#pragma unroll
            for (int k = 0; k < kSwMaxNewLoads; ++k)
                if (naddr[k] >= 0) *reinterpret_cast<u32x4 *>(lds + naddr[k]) = nv[k];
        }
        fill_table(t + 2);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ host plan
struct SweepPlan {
    SweepArgs args;
    int lds_bytes;
    bool ok;
};

static SweepPlan plan_sweep(const int64_t *host_shapes, int L, int B, int S, int halves)
{
    SweepPlan p{};
    p.ok = false;
    if (L < 1 || L > kSwMaxLevels) return p;
    int hh[kSwMaxLevels], ww[kSwMaxLevels], st[kSwMaxLevels];
    long long tot = 0;
    for (int l = 0; l < L; ++l) {
        hh[l] = (int)host_shapes[2 * l];
        ww[l] = (int)host_shapes[2 * l + 1];
        if (hh[l] <= 0 || ww[l] <= 0 || hh[l] > 4096 || ww[l] > 4096) return p;
        st[l] = (int)tot;
        tot += (long long)hh[l] * ww[l];
    }
    if (tot != S) return p;
    const int lds_cap = 160 * 1024 - kSwTableBytes;
    // Search order: widest margin first (fewer global-load fallbacks), then the widest strip that fits LDS.
    // Rows per step are chosen so that one step holds ~128 queries = one 16-query group per wave.
    static const int strip_w[] = {32, 24, 16, 12, 8};
    const double ratio = (double)S / ((double)hh[0] * ww[0]);           // queries of all levels per level-0 pixel
    for (int M = 8; M >= 2; --M)
        for (int wi = 0; wi < 5; ++wi) {
            const int nsx = std::max(1, (ww[0] + strip_w[wi] - 1) / strip_w[wi]);
            const double strip_px = (double)ww[0] / nsx;
            int r0 = (int)(kSwWaves * 16 / (strip_px * ratio) + 0.5);
            r0 = std::min(std::max(r0, 1), std::min(hh[0], 16));
            const int nstep = std::max(1, (hh[0] + r0 - 1) / r0);
            int bytes = 0, newmax = 0;
            SweepArgs a{};
            for (int l = 0; l < L; ++l) {
                int pitch = 1, rc = 1;
                for (int s = 0; s < nsx; ++s) pitch = std::max(pitch, sw_win_hi(s, ww[l], nsx, M) - sw_win_lo(s, ww[l], nsx, M));
                for (int t = 0; t < nstep; ++t) {
                    const int hi = sw_win_hi(std::min(t + 1, nstep - 1), hh[l], nstep, M);
                    rc = std::max(rc, hi - sw_win_lo(t, hh[l], nstep, M));
                }
                a.lv[l] = SweepLevel{hh[l], ww[l], st[l], pitch, rc, kSwTableBytes + bytes};
                bytes += rc * pitch * kSwRowBytes;
            }
            for (int t = 0; t + 1 < nstep; ++t) {
                int nn = 0;
                for (int l = 0; l < L; ++l)
                    nn += (sw_win_hi(t + 1, hh[l], nstep, M) - sw_win_hi(t, hh[l], nstep, M)) * a.lv[l].pitch;
                newmax = std::max(newmax, nn);
            }
            if (bytes > lds_cap || newmax * 4 > kSwMaxNewLoads * kSwThreads) continue;
            a.L = L;
            a.nsx = nsx;
            a.nstep = nstep;
            const long long per_seg = (long long)B * kSwHeads * halves * nsx;
            a.nseg = (int)std::min<long long>(nstep, std::max<long long>(1, (2 * 256 + per_seg - 1) / per_seg));
            a.margin = M;
            a.S = S;
            a.B = B;
            const long long nblk = per_seg * a.nseg;
            if (nblk > 0x7fffffffll) return p;
            a.nblk = (int)nblk;
            p.args = a;
            p.lds_bytes = kSwTableBytes + bytes;
            p.ok = true;
            return p;
        }
    return p;
}

template <typename T, bool FUSED>
int msda_sweep_forward(const T *value, const int64_t *host_shapes, const void *src_a, const void *src_b,
                              const float *ref, int ref_dim, int B, int S, int L, T *out, hipStream_t stream)
{
    if (B < 0 || S <= 0 || L <= 0) return RDETR_ERR_INVALID_ARG;
    if (B == 0) return RDETR_OK;
    if (!value || !host_shapes || !src_a || !src_b || !out || (FUSED && !ref)) return RDETR_ERR_INVALID_ARG;
    if (FUSED && ref_dim != 2 && ref_dim != 4) return RDETR_ERR_INVALID_ARG;
    auto al16 = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (!al16(value) || !al16(out) || reinterpret_cast<uintptr_t>(src_a) % 8 != 0) return RDETR_ERR_UNSUPPORTED;
    if ((long long)S * kSwHeads * kSwHeadDim * (long long)sizeof(T) >= (1ll << 31)) return RDETR_ERR_UNSUPPORTED;
    SweepPlan p = plan_sweep(host_shapes, L, B, S, SweepIO<T>::kHalves);
    if (!p.ok) return RDETR_ERR_UNSUPPORTED;
    p.args.ref_dim = ref_dim;
    dim3 grid((unsigned)p.args.nblk), block(kSwThreads);
    if (L <= 4) {
        auto kern = msda_fwd_sweep_kernel<T, FUSED, 4>;
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, grid, block, p.lds_bytes, stream, value, src_a, src_b, ref, out, p.args);
    } else {
        auto kern = msda_fwd_sweep_kernel<T, FUSED, kSwMaxLevels>;
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr != hipSuccess) return RDETR_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, grid, block, p.lds_bytes, stream, value, src_a, src_b, ref, out, p.args);
    }
    return launch_status();
}

template int msda_sweep_forward<float, false>(const float *, const int64_t *, const void *, const void *, const float *, int,
                                              int, int, int, float *, hipStream_t);
template int msda_sweep_forward<float, true>(const float *, const int64_t *, const void *, const void *, const float *, int,
                                             int, int, int, float *, hipStream_t);
template int msda_sweep_forward<uint16_t, false>(const uint16_t *, const int64_t *, const void *, const void *,
                                                 const float *, int, int, int, int, uint16_t *, hipStream_t);
template int msda_sweep_forward<uint16_t, true>(const uint16_t *, const int64_t *, const void *, const void *,
                                                const float *, int, int, int, int, uint16_t *, hipStream_t);

}  // namespace rdetr
