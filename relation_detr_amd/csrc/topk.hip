// Row-wise top-k (largest, sorted) of an fp32 matrix for the two selections on the transformer's dependency chain (gfx950):
//   the two-stage proposal choice  torch.topk(enc_outputs_class.max(-1)[0], 900, dim=1)   models/bricks/relation_transformer.py:93
//   PostProcess                    torch.topk(prob.view(B, -1), 300, dim=1)                models/bricks/post_process.py:30
// torch's multi-block radix select + sort takes 84 us for [4, 22,323] and 118 us for [4, 81,900]; the stack pays that on every
// image group's chain.  Here: a multi-workgroup radix select on the top 12 key bits (histogram, scan, partition: three short
// launches that use the whole chip; per-workgroup partial histograms, so the workspace needs no zeroing) and one workgroup per row that refines the boundary bin and sorts the k winners.
//
// Order: TOTAL and deterministic -- by value descending, equal values by index ascending, NaN above everything (torch.topk's
// order among equal values is unspecified; its NaN rule is the same).  Every comparison is made on a 64-bit composite
//     (ordered 32-bit key of the float) << 20  |  (0xfffff - index)            n < 2^20
// whose values are pairwise distinct, so the k-th largest composite is a sharp threshold and no tie ever needs a rule of its own.
#include "common.h"

namespace rdetr {

constexpr int kTkBins = 4096;                  // level 1: top 12 bits of the 32-bit key
constexpr int kTkItems = 16;                   // elements per thread in the streaming kernels
constexpr int kTkBlock = 256;
constexpr int kTkChunk = kTkBlock * kTkItems;  // 4096 elements per workgroup

struct TkCtrl {                                // per row, in the workspace
    unsigned bin, above, out_count, cand_count;
};

__device__ __forceinline__ unsigned tk_key(float v)
{
    const unsigned u = __builtin_bit_cast(unsigned, v);
    if (v != v) return 0xffffffffu;                                    // NaN: the largest
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);                 // ascending as unsigned
}
__device__ __forceinline__ float tk_load(const void *x, int is_bf16, size_t i)
{
    return is_bf16 ? bf16_bits_to_f32(static_cast<const uint16_t *>(x)[i]) : static_cast<const float *>(x)[i];
}
__device__ __forceinline__ float tk_value(unsigned key)
{
    if (key == 0xffffffffu) return __builtin_nanf("");
    return __builtin_bit_cast(float, (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}

// A: histogram of the level-1 bins.  grid (chunks, rows)
__global__ __launch_bounds__(kTkBlock) void topk_hist_kernel(const void *__restrict__ x, int is_bf16, int n, unsigned *__restrict__ hist)
{
    __shared__ unsigned h[kTkBins];
    for (int i = threadIdx.x; i < kTkBins; i += kTkBlock) h[i] = 0;
    __syncthreads();
    const size_t row = (size_t)blockIdx.y * n;
    const int base = blockIdx.x * kTkChunk;
#pragma unroll 4
    for (int j = 0; j < kTkItems; ++j) {
        const int i = base + j * kTkBlock + threadIdx.x;
        if (i < n) atomicAdd(&h[tk_key(tk_load(x, is_bf16, row + i)) >> 20], 1u);
    }
    __syncthreads();
    // every workgroup writes its own partial histogram (no zero-initialised global state, no global atomics); B adds them up
    unsigned *g = hist + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kTkBins;
    for (int i = threadIdx.x; i < kTkBins; i += kTkBlock) g[i] = h[i];
}

// B: the bin that holds the k-th largest element and the number of elements above it.  grid rows, 1024 threads (4 bins each)
__global__ __launch_bounds__(1024) void topk_scan_kernel(const unsigned *__restrict__ hist, int chunks, int k, TkCtrl *__restrict__ ctrl)
{
    __shared__ unsigned s[1024];
    const int t = threadIdx.x;
    unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int ch = 0; ch < chunks; ++ch) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(hist + ((size_t)blockIdx.x * chunks + ch) * kTkBins + 4 * t);
        c0 += v.x; c1 += v.y; c2 += v.z; c3 += v.w;
    }
    s[t] = c0 + c1 + c2 + c3;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                               // inclusive suffix sums over the threads
        const unsigned v = t + o < 1024 ? s[t + o] : 0u;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    unsigned run = t + 1 < 1024 ? s[t + 1] : 0u;                      // elements in bins above this thread's
    const unsigned c[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int j = 3; j >= 0; --j) {
        if (run < (unsigned)k && run + c[j] >= (unsigned)k) {
            ctrl[blockIdx.x].bin = (unsigned)(4 * t + j);
            ctrl[blockIdx.x].above = run;
        }
        run += c[j];
    }
    if (t == 0) {
        ctrl[blockIdx.x].out_count = 0;
        ctrl[blockIdx.x].cand_count = 0;
    }
}

// C: elements above the boundary bin go to the winners, elements of the boundary bin to the candidates (composites).  grid (chunks, rows)
// One global atomic per workgroup and class: the 16 items of a thread are classified first, positions inside the workgroup come from
// a prefix over its threads (a chain of per-wave atomics cost ~1 us each: 16 us per launch).
__global__ __launch_bounds__(kTkBlock) void topk_partition_kernel(const void *__restrict__ x, int is_bf16, int n, TkCtrl *ctrl,
                                                                 unsigned long long *__restrict__ winners, int k,
                                                                 unsigned long long *__restrict__ cand)
{
    __shared__ unsigned wave_w[kTkBlock / 64], wave_c[kTkBlock / 64], base_w, base_c;
    const int r = blockIdx.y;
    const size_t row = (size_t)r * n;
    const unsigned bstar = ctrl[r].bin;
    unsigned long long *w = winners + (size_t)r * k, *c = cand + (size_t)r * n;
    const int base = blockIdx.x * kTkChunk, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned key[kTkItems];
    unsigned nw = 0, nc = 0, fw = 0, fc = 0;                              // counts and per-item flags of this thread
#pragma unroll
    for (int j = 0; j < kTkItems; ++j) {
        const int i = base + j * kTkBlock + threadIdx.x;
        key[j] = 0;
        if (i < n) {
            key[j] = tk_key(tk_load(x, is_bf16, row + i));
            const unsigned bin = key[j] >> 20;
            if (bin > bstar) { fw |= 1u << j; ++nw; }
            else if (bin == bstar) { fc |= 1u << j; ++nc; }
        }
    }
    // exclusive prefix of (nw, nc) over the workgroup's threads
    unsigned pw = nw, pc = nc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned a = __shfl_up(pw, o, 64), bq = __shfl_up(pc, o, 64);
        if (lane >= o) { pw += a; pc += bq; }
    }
    if (lane == 63) { wave_w[wave] = pw; wave_c[wave] = pc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tw = 0, tc = 0;
        for (int i = 0; i < kTkBlock / 64; ++i) {
            const unsigned a = wave_w[i], bq = wave_c[i];
            wave_w[i] = tw; wave_c[i] = tc;
            tw += a; tc += bq;
        }
        base_w = tw ? atomicAdd(&ctrl[r].out_count, tw) : 0u;
        base_c = tc ? atomicAdd(&ctrl[r].cand_count, tc) : 0u;
    }
    __syncthreads();
    unsigned ow = base_w + wave_w[wave] + pw - nw, oc = base_c + wave_c[wave] + pc - nc;
#pragma unroll
    for (int j = 0; j < kTkItems; ++j) {
        const int i = base + j * kTkBlock + threadIdx.x;
        const unsigned long long comp = ((unsigned long long)key[j] << 20) | (unsigned long long)(0xfffffu - (unsigned)i);
        if (fw >> j & 1u) w[ow++] = comp;
        if (fc >> j & 1u) c[oc++] = comp;
    }
}

// D: one workgroup per row.  The `need` = k - above largest composites of the candidates (4 radix passes of 10 bits over the 40 low
// bits; the 12 high bits are equal in all of them), appended to the winners; bitonic sort of the k winners; values + indices out.
// Scans run inside the waves (shuffles) with one exchange of the 16 wave totals; the sort's strides below 64 are shuffles too.
__device__ __forceinline__ unsigned tk_suffix_exclusive(unsigned v, unsigned *wave_tot, int t)
{
    // sum of v over the threads ABOVE t in a 1024-thread workgroup
    const int lane = t & 63, wave = t >> 6;
    unsigned incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned a = __shfl_down(incl, o, 64);
        if (lane + o < 64) incl += a;
    }
    if (lane == 0) wave_tot[wave] = incl;                                  // the wave's total
    __syncthreads();
    unsigned above_waves = 0;
    for (int wv = wave + 1; wv < 16; ++wv) above_waves += wave_tot[wv];
    __syncthreads();
    return incl - v + above_waves;
}

__global__ __launch_bounds__(1024) void topk_finish_kernel(const TkCtrl *__restrict__ ctrl, const unsigned long long *__restrict__ cand, int n,
                                                          unsigned long long *__restrict__ winners, int k, float *__restrict__ values,
                                                          long long *__restrict__ indices)
{
    __shared__ unsigned hist[1024];
    __shared__ unsigned wave_tot[16];
    __shared__ unsigned long long keys[1024];
    __shared__ unsigned sel_bin, sel_above, append;
    const int r = blockIdx.x, t = threadIdx.x;
    const unsigned above = ctrl[r].above, cnt = ctrl[r].cand_count;
    const unsigned long long *c = cand + (size_t)r * n;
    unsigned long long *w = winners + (size_t)r * k;
    unsigned need = (unsigned)k - above;                               // 1 <= need <= cnt
    unsigned long long prefix = 0, pmask = 0;                          // bits of the threshold fixed so far (below bit 40)
    for (int shift = 30; shift >= 0; shift -= 10) {
        hist[t] = 0;
        __syncthreads();
        for (unsigned i = t; i < cnt; i += 1024) {
            const unsigned long long v = c[i];
            if ((v & pmask) == prefix) atomicAdd(&hist[(unsigned)(v >> shift) & 1023u], 1u);
        }
        __syncthreads();
        const unsigned mine = hist[t];
        const unsigned ab = tk_suffix_exclusive(mine, wave_tot, t);        // candidates with a larger digit
        if (ab < need && ab + mine >= need) {
            sel_bin = (unsigned)t;
            sel_above = ab;
        }
        __syncthreads();
        prefix |= (unsigned long long)sel_bin << shift;
        pmask |= 1023ull << shift;
        need -= sel_above;
        __syncthreads();
    }
    // threshold = the candidate whose low 40 bits equal `prefix`: composites >= it are in (all distinct -> exactly k - above of them)
    const unsigned long long low40 = (1ull << 40) - 1ull;
    if (t == 0) append = 0;
    __syncthreads();
    for (unsigned i = t; i < cnt; i += 1024) {
        const unsigned long long v = c[i];
        if ((v & low40) >= prefix) w[above + atomicAdd(&append, 1u)] = v;
    }
    __syncthreads();
    // bitonic sort, descending, of the k winners padded with zeros to 1024: one element per thread, strides < 64 by shuffle
    unsigned long long mine = t < k ? w[t] : 0ull;
    for (int size = 2; size <= 1024; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            unsigned long long other;
            if (stride >= 64) {
                keys[t] = mine;
                __syncthreads();
                other = keys[t ^ stride];
                __syncthreads();
            } else {
                const unsigned lo = __shfl_xor((unsigned)mine, stride, 64), hi = __shfl_xor((unsigned)(mine >> 32), stride, 64);
                other = ((unsigned long long)hi << 32) | lo;
            }
            const bool desc = (t & size) == 0, lower = (t & stride) == 0;    // the lower thread of a pair keeps the larger value when descending
            const bool take_max = desc == lower;
            mine = take_max ? (mine > other ? mine : other) : (mine < other ? mine : other);
        }
    if (t < k) {
        values[(size_t)r * k + t] = tk_value((unsigned)(mine >> 20));
        indices[(size_t)r * k + t] = (long long)(0xfffffu - (unsigned)(mine & 0xfffffu));
    }
}

}  // namespace rdetr

using namespace rdetr;

// values [rows, k] fp32 + indices [rows, k] int64 of the k largest elements of every row of x [rows, n] fp32 | bf16 (contiguous), sorted
// by value descending, equal values by index ascending, NaN first.  1 <= k <= min(n, 1024), n < 2^20.
// workspace: rdetr_topk_workspace_bytes(rows, n, k) bytes, 16-byte aligned; its contents are scratch.
extern "C" long long rdetr_topk_workspace_bytes(int rows, int n, int k)
{
    if (rows < 0 || n < 0 || k < 0) return -1;
    const long long chunks = (n + kTkChunk - 1) / kTkChunk;
    return (long long)rows * (chunks * kTkBins * 4 + 16 + 8ll * n + 8ll * k);
}

extern "C" int rdetr_topk(const void *x, int is_bf16, int rows, int n, int k, void *workspace, float *values, long long *indices, void *stream)
{
    if (rows < 0 || n <= 0 || k <= 0) return RDETR_ERR_INVALID_ARG;
    if (k > n || k > 1024 || n >= (1 << 20) || rows > 65535) return RDETR_ERR_UNSUPPORTED;
    if (rows == 0) return RDETR_OK;
    if (!x || !workspace || !values || !indices) return RDETR_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(workspace) % 16) return RDETR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    const int chunks = (n + kTkChunk - 1) / kTkChunk;
    unsigned *hist = reinterpret_cast<unsigned *>(ws);                         // [rows][chunks][4096]
    TkCtrl *ctrl = reinterpret_cast<TkCtrl *>(ws + (size_t)rows * chunks * kTkBins * 4);
    unsigned long long *cand = reinterpret_cast<unsigned long long *>(ws + (size_t)rows * ((size_t)chunks * kTkBins * 4 + 16));
    unsigned long long *winners = cand + (size_t)rows * n;
    const dim3 grid((unsigned)chunks, (unsigned)rows);
    hipLaunchKernelGGL(topk_hist_kernel, grid, dim3(kTkBlock), 0, st, x, is_bf16, n, hist);
    hipLaunchKernelGGL(topk_scan_kernel, dim3((unsigned)rows), dim3(1024), 0, st, hist, chunks, k, ctrl);
    hipLaunchKernelGGL(topk_partition_kernel, grid, dim3(kTkBlock), 0, st, x, is_bf16, n, ctrl, winners, k, cand);
    hipLaunchKernelGGL(topk_finish_kernel, dim3((unsigned)rows), dim3(1024), 0, st, ctrl, cand, n, winners, k, values, indices);
    return launch_status();
}
