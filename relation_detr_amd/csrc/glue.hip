// Small fused elementwise kernels of the decoder's box bookkeeping (gfx950): each replaces a chain of ~8 torch launches
// on a [B, N, 4] tensor -- inside a HIP graph every launch still costs a few microseconds of dependency latency.
//
//   box_refine      sigmoid(delta + inverse_sigmoid(ref))                 models/bricks/relation_transformer.py:363-381
//                   with inverse_sigmoid(x) = log(clamp(x,0,1).clamp(min=eps) / (1 - clamp(x,0,1)).clamp(min=eps)), eps = 1e-3
//                   (util/misc.py:31-35); delta in fp32 or bf16, ref / out fp32
//   sine_pos_embed  get_sine_pos_embed(pos, num_pos_feats = F, temperature, scale = 2*pi, exchange_xy = True)
//                   (models/bricks/position_encoding.py:115-138): [rows, n] -> [rows, n * F], coordinate order (y, x, rest),
//                   channel = coord * F + 2k + {sin, cos};  (pos * scale) / dim_t[k] in fp32 as the reference rounds it
// Bound: launch latency (tens of KB of traffic).
//
// and of the encoder's input / output side (HBM-bound, each one pass over its tensor):
//   zero_masked_rows   value.masked_fill(key_padding_mask[..., None], 0) in place (models/bricks/ms_deform_attn.py:316-319):
//                      reads the mask, WRITES ONLY the padded rows (torch's masked_fill reads and rewrites every row)
//   row_max            x.max(-1)[0] of the two-stage class logits (models/bricks/relation_transformer.py:105): 4 rows per wave
//   nchw_to_tokens     x.flatten(2).transpose(1, 2) of one pyramid level (+ a per-channel vector: the level embedding of
//                      models/bricks/relation_transformer.py:87-89) written into its row range of the level-packed
//                      [B, S, C] token tensor (models/bricks/base_transformer.py:17-23): LDS-tiled transpose, 64 x 64
#include "common.h"

namespace rdetr {

template <typename TD> __device__ __forceinline__ float glue_load(const TD *p);
template <> __device__ __forceinline__ float glue_load<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float glue_load<uint16_t>(const uint16_t *p) { return bf16_bits_to_f32(*p); }

template <typename TD>
__global__ __launch_bounds__(256) void box_refine_kernel(const TD *__restrict__ delta, const float *__restrict__ ref, long long n,
                                                         float eps, float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = ref[i];
    x = fminf(fmaxf(x, 0.f), 1.f);                       // NaN -> 0 like torch.clamp? torch propagates NaN: keep it
    if (ref[i] != ref[i]) x = ref[i];
    const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
    const float z = glue_load<TD>(delta + i) + logf(x1 / x2);
    out[i] = 1.f / (1.f + expf(-z));
}

template <typename TO> __device__ __forceinline__ void glue_store(TO *p, float v);
template <> __device__ __forceinline__ void glue_store<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void glue_store<uint16_t>(uint16_t *p, float v) { *p = (uint16_t)f32_to_bf16_bits(v); }

struct GlueDimT {
    float v[64];
};

template <typename TO>
__global__ __launch_bounds__(256) void sine_pos_embed_kernel(const float *__restrict__ pos, long long rows, int n, int F, float scale,
                                                             GlueDimT dim_t, TO *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (row, coord, k): a sin/cos pair
    const int half = F / 2;
    const long long total = rows * n * half;
    if (idx >= total) return;
    const int k = (int)(idx % half);
    const long long rc = idx / half;
    const int c = (int)(rc % n);
    const long long row = rc / n;
    const int src = c == 0 ? 1 : (c == 1 ? 0 : c);       // exchange_xy: output coordinate 0 is y, 1 is x
    const float a = (pos[row * n + src] * scale) / dim_t.v[k];
    TO *o = out + (row * n + c) * F + 2 * k;
    glue_store<TO>(o, sinf(a));
    glue_store<TO>(o + 1, cosf(a));
}


// Decoder layer entry (relation_transformer.py:335-343): ref_in[b][q][l] = reference[b][q] * (vr[b][l].x, .y, .x, .y) for every
// level, and the sine embedding of ref_in[:, :, 0, :] (exchange_xy = True) -- one launch for the multiply, the slice copy and the
// embedding.  One thread per (query, coord, k); the threads with k == 0 also write the coordinate's L scaled values.
template <typename TO>
__global__ __launch_bounds__(256) void decoder_reference_kernel(const float *__restrict__ ref, const float *__restrict__ ratios, int N,
                                                                int L, int F, float scale, GlueDimT dim_t, float *__restrict__ ref_in,
                                                                TO *__restrict__ emb, long long total)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int half = F / 2;
    const int k = (int)(idx % half);
    const long long rc = idx / half;
    const int c = (int)(rc % 4);
    const long long row = rc / 4;                         // b * N + q
    const int b = (int)(row / N);
    const int src = c == 0 ? 1 : (c == 1 ? 0 : c);        // exchange_xy: output coordinate 0 is y, 1 is x
    const float *vr = ratios + (size_t)b * L * 2;
    const float a = ((ref[row * 4 + src] * vr[src & 1]) * scale) / dim_t.v[k];
    TO *o = emb + (row * 4 + c) * F + 2 * k;
    glue_store<TO>(o, sinf(a));
    glue_store<TO>(o + 1, cosf(a));
    if (k == 0) {
        const float r = ref[row * 4 + c];
        for (int l = 0; l < L; ++l) ref_in[(row * L + l) * 4 + c] = r * vr[2 * l + (c & 1)];
    }
}

// query_pos = a * s and query + query_pos in one pass (relation_transformer.py:346-347, 452): the product is rounded to the storage
// type before it is added, as the two torch kernels do.  Contiguous tensors of n elements.
template <typename T>
__global__ __launch_bounds__(256) void scaled_pos_kernel(const T *__restrict__ a, const T *__restrict__ sc, const T *__restrict__ q,
                                                         long long n, T *__restrict__ pos, T *__restrict__ qp)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    glue_store<T>(pos + i, glue_load<T>(a + i) * glue_load<T>(sc + i));
    glue_store<T>(qp + i, glue_load<T>(q + i) + glue_load<T>(pos + i));
}

// PostProcess after its top-k (models/bricks/post_process.py:30-44): flat index -> (box, label), cxcywh -> xyxy, scale to pixels,
// pack (x1, y1, x2, y2, score, label) -- one launch for ~12 torch ones.  sizes: (h, w) per image as int64.
__global__ __launch_bounds__(256) void detections_kernel(const float *__restrict__ score, const long long *__restrict__ idx,
                                                         const float *__restrict__ boxes, const long long *__restrict__ sizes, int N, int C,
                                                         int K, long long total, float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;        // (image, rank)
    if (i >= total) return;
    const int b = (int)(i / K);
    const long long flat = idx[i];
    const long long bi = flat / C;
    const float label = (float)(flat - bi * C);
    const f32x4 bx = *reinterpret_cast<const f32x4 *>(boxes + ((size_t)b * N + bi) * 4);
    const float ih = (float)sizes[2 * b], iw = (float)sizes[2 * b + 1];
    float *o = out + i * 6;
    o[0] = (bx.x - 0.5f * bx.z) * iw;
    o[1] = (bx.y - 0.5f * bx.w) * ih;
    o[2] = (bx.x + 0.5f * bx.z) * iw;
    o[3] = (bx.y + 0.5f * bx.w) * ih;
    o[4] = score[i];
    o[5] = label;
}

// ---- zero the rows whose mask byte is set ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zero_masked_rows_kernel(unsigned char *__restrict__ x, const unsigned char *__restrict__ mask,
                                                               long long rows, int row_bytes, long long ld_bytes)
{
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;     // 64 rows per wave: lane = row
    const long long row = row0 + lane;
    unsigned long long todo = __ballot(row < rows && mask[row] != 0);
    while (todo) {                                                                     // uniform
        const int r = __builtin_ctzll(todo);
        todo &= todo - 1;
        unsigned char *dst = x + (row0 + r) * ld_bytes;
        for (int o = lane * 16; o < row_bytes; o += 64 * 16) *reinterpret_cast<u32x4 *>(dst + o) = u32x4{0u, 0u, 0u, 0u};
    }
}

// ---- row maximum (NaN propagates, as torch.max) ---------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void row_max_kernel(const T *__restrict__ x, long long rows, int C, long long ldx, T *__restrict__ out)
{
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);    // 16 lanes per row
    float m = -__builtin_inff();
    if (row < rows) {
        const T *xr = x + row * ldx;
        for (int c = sub; c < C; c += 16) {
            const float v = glue_load<T>(xr + c);
            m = (v > m || v != v) ? v : m;
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {
        const float v = __shfl_xor(m, o, 64);
        m = (v > m || v != v) ? v : m;
    }
    if (row < rows && sub == 0) glue_store<T>(out + row, m);
}

// ---- [B, C, P] -> rows [p0 .. p0 + P) of [B, S, C'] (row stride ld_out): 64 channels x 64 pixels per block through LDS --
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_tokens_kernel(const T *__restrict__ src, const T *__restrict__ add_vec, int C, int P,
                                                             long long out_image_stride, long long ld_out, T *__restrict__ out)
{
    __shared__ float tile[64][65];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;                  // 4 rows of 64 per pass
    const T *s = src + ((size_t)b * C + c0) * (size_t)P + p0;
#pragma unroll 4
    for (int i = ty; i < 64; i += 4)                                         // channel c0 + i, pixel p0 + tx: coalesced along pixels
        tile[i][tx] = (c0 + i < C && p0 + tx < P) ? glue_load<T>(s + (size_t)i * P + tx) : 0.f;
    __syncthreads();
    const float av = (add_vec && c0 + tx < C) ? glue_load<T>(add_vec + c0 + tx) : 0.f;
    T *o = out + (size_t)b * out_image_stride + (size_t)p0 * ld_out + c0;
#pragma unroll 4
    for (int i = ty; i < 64; i += 4)                                         // pixel p0 + i, channel c0 + tx: coalesced along channels
        if (p0 + i < P && c0 + tx < C) {
            glue_store<T>(o + (size_t)i * ld_out + tx, tile[tx][i] + av);    // two stored values, summed in fp32, rounded once (as torch)
        }
}

// bf16 with 16-byte accesses on both sides (P, C, ld_out multiples of 8, aligned bases): a lane reads 8 pixels of one channel and
// writes 8 channels of one pixel; the scalar kernel above moves 128 bytes per wave instruction (2 TB/s at the encoder's level 0).
// Same arithmetic per element (fp32 sum with the optional per-channel vector, one rounding).
__global__ __launch_bounds__(256) void nchw_to_tokens_bf16_vec_kernel(const uint16_t *__restrict__ src, const uint16_t *__restrict__ add_vec,
                                                                      int C, int P, long long out_image_stride, long long ld_out,
                                                                      uint16_t *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint16_t tile[64][72];          // [pixel][channel], rows 144 B apart
    const int b = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int it = tid + 256 * k, ch = it >> 3, pg = it & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (c0 + ch < C && p0 + 8 * pg < P)                                   // P % 8 == 0: a group of 8 pixels is inside or outside
            v = *reinterpret_cast<const u32x4 *>(src + ((size_t)b * C + c0 + ch) * (size_t)P + p0 + 8 * pg);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tile[8 * pg + 2 * j][ch] = (uint16_t)(w[j] & 0xffffu);
            tile[8 * pg + 2 * j + 1][ch] = (uint16_t)(w[j] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int it = tid + 256 * k, px = it >> 3, cg = it & 7;
        if (p0 + px >= P || c0 + 8 * cg >= C) continue;                       // C % 8 == 0
        u32x4 v = *reinterpret_cast<const u32x4 *>(&tile[px][8 * cg]);
        if (add_vec) {
            const u32x4 a = *reinterpret_cast<const u32x4 *>(add_vec + c0 + 8 * cg);
            const unsigned vw[4] = {v.x, v.y, v.z, v.w}, aw[4] = {a.x, a.y, a.z, a.w};
            unsigned o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = pack_bf16x2(bf16_bits_to_f32(vw[j] & 0xffffu) + bf16_bits_to_f32(aw[j] & 0xffffu),
                                   bf16_bits_to_f32(vw[j] >> 16) + bf16_bits_to_f32(aw[j] >> 16));
            v = u32x4{o[0], o[1], o[2], o[3]};
        }
        *reinterpret_cast<u32x4 *>(out + (size_t)b * out_image_stride + (size_t)(p0 + px) * ld_out + c0 + 8 * cg) = v;
    }
}

// ---- pyramid geometry of the two-stage transformer in two launches (torch: ~40 small ones per forward) ---------------------
struct PyrLevels {
    const unsigned char *mask[8];     // [B, h, w] bool per level
    int h[8], w[8], start[8];
};

// valid_ratios[b][l] = (unpadded columns of row 0 / w, unpadded rows of column 0 / h)   (base_transformer.py:42-51)
__global__ __launch_bounds__(64) void valid_ratios_kernel(PyrLevels lv, int L, float *__restrict__ out)
{
    const int l = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int h = lv.h[l], w = lv.w[l];
    const unsigned char *m = lv.mask[l] + (size_t)b * h * w;
    int cw = 0, ch = 0;
    for (int x = lane; x < w; x += 64) cw += m[x] == 0;
    for (int y = lane; y < h; y += 64) ch += m[(size_t)y * w] == 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        cw += __shfl_xor(cw, o, 64);
        ch += __shfl_xor(ch, o, 64);
    }
    if (lane == 0) {
        out[((size_t)b * L + l) * 2 + 0] = (float)cw / (float)w;
        out[((size_t)b * L + l) * 2 + 1] = (float)ch / (float)h;
    }
}

// per position s of level l (pixel x, y):  full = (x + .5, y + .5) / (valid_ratio[b][l] * (w, h))         (base_transformer.py:57-70)
//   reference[b][s][l'] = full * valid_ratio[b][l']                                                    [B, S, L, 2]
//   proposal = (full, 0.05 * 2^l, 0.05 * 2^l); valid = all(0.01 < proposal < 0.99) and not padded      (relation_transformer.py:162-176)
//   logit[b][s] = log(p / (1 - p)), +inf where not valid                                               [B, S, 4]
//   keep[b][s]  = valid ? 1 : 0 in fp32 / bf16 (the factor the encoder memory is multiplied with)       [B, S]
template <typename T>
__global__ __launch_bounds__(256) void pyramid_points_kernel(PyrLevels lv, int L, int S, const float *__restrict__ ratios,
                                                             const unsigned char *__restrict__ pad, float *__restrict__ reference,
                                                             float *__restrict__ logit, T *__restrict__ keep)
{
    const int s = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (s >= S) return;
    int l = 0;
    while (l + 1 < L && s >= lv.start[l + 1]) ++l;
    const int w = lv.w[l], h = lv.h[l], idx = s - lv.start[l];
    const int y = idx / w, x = idx - y * w;
    const float *vr = ratios + (size_t)b * L * 2;
    const float fx = ((float)x + 0.5f) / (vr[2 * l] * (float)w), fy = ((float)y + 0.5f) / (vr[2 * l + 1] * (float)h);
    float *r = reference + ((size_t)b * S + s) * L * 2;
    for (int k = 0; k < L; ++k) {
        r[2 * k] = fx * vr[2 * k];
        r[2 * k + 1] = fy * vr[2 * k + 1];
    }
    const float wh = 0.05f * (float)(1 << l);
    const float p[4] = {fx, fy, wh, wh};
    bool ok = !(pad && pad[(size_t)b * S + s]);
#pragma unroll
    for (int k = 0; k < 4; ++k) ok = ok && p[k] > 0.01f && p[k] < 0.99f;
    float *lg = logit + ((size_t)b * S + s) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) lg[k] = ok ? logf(p[k] / (1.0f - p[k])) : __builtin_inff();
    glue_store<T>(keep + (size_t)b * S + s, ok ? 1.0f : 0.0f);
}

}  // namespace rdetr

using namespace rdetr;

extern "C" int rdetr_pyramid_points(const unsigned char *const *level_masks, const int *level_hw, int L, int B,
                                    const unsigned char *pad_mask, int keep_is_bf16, float *valid_ratios, float *reference,
                                    float *logit, void *keep, void *stream)
{
    if (L <= 0 || B < 0 || !level_masks || !level_hw) return RDETR_ERR_INVALID_ARG;
    if (L > 8 || B > 65535) return RDETR_ERR_UNSUPPORTED;
    if (B == 0) return RDETR_OK;
    if (!valid_ratios || !reference || !logit || !keep) return RDETR_ERR_INVALID_ARG;
    PyrLevels lv;
    long long S = 0;
    for (int l = 0; l < L; ++l) {
        if (!level_masks[l] || level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0) return RDETR_ERR_INVALID_ARG;
        lv.mask[l] = level_masks[l];
        lv.h[l] = level_hw[2 * l];
        lv.w[l] = level_hw[2 * l + 1];
        lv.start[l] = (int)S;
        S += (long long)lv.h[l] * lv.w[l];
    }
    if (S >= (1ll << 31)) return RDETR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(valid_ratios_kernel, dim3((unsigned)L, (unsigned)B), dim3(64), 0, st, lv, L, valid_ratios);
    const dim3 grid((unsigned)((S + 255) / 256), (unsigned)B);
    if (keep_is_bf16)
        hipLaunchKernelGGL((pyramid_points_kernel<uint16_t>), grid, dim3(256), 0, st, lv, L, (int)S, valid_ratios, pad_mask, reference, logit,
                           static_cast<uint16_t *>(keep));
    else
        hipLaunchKernelGGL((pyramid_points_kernel<float>), grid, dim3(256), 0, st, lv, L, (int)S, valid_ratios, pad_mask, reference, logit,
                           static_cast<float *>(keep));
    return launch_status();
}

extern "C" int rdetr_box_refine_f32(const void *delta, int delta_is_bf16, const float *ref, long long n, float eps, float *out,
                                    void *stream)
{
    if (n < 0) return RDETR_ERR_INVALID_ARG;
    if (n == 0) return RDETR_OK;
    if (!delta || !ref || !out) return RDETR_ERR_INVALID_ARG;
    const long long nblk = (n + 255) / 256;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (delta_is_bf16)
        hipLaunchKernelGGL((box_refine_kernel<uint16_t>), dim3((unsigned)nblk), dim3(256), 0, st, static_cast<const uint16_t *>(delta),
                           ref, n, eps, out);
    else
        hipLaunchKernelGGL((box_refine_kernel<float>), dim3((unsigned)nblk), dim3(256), 0, st, static_cast<const float *>(delta), ref, n,
                           eps, out);
    return launch_status();
}

extern "C" int rdetr_sine_pos_embed(const float *pos, long long rows, int n, int F, float temperature, float scale, void *out,
                                    int out_is_bf16, void *stream)
{
    // n >= 2: the first two coordinates are exchanged (exchange_xy=True); the reference's index_select fails for n == 1 too
    if (rows < 0 || n < 2 || F <= 0) return RDETR_ERR_INVALID_ARG;
    if ((F & 1) || F > 128) return RDETR_ERR_UNSUPPORTED;
    if (rows == 0) return RDETR_OK;
    if (!pos || !out) return RDETR_ERR_INVALID_ARG;
    GlueDimT dt;
    // get_dim_t: temperature ** (2 * (arange(F) // 2) / F), fp32 (position_encoding.py:101-105)
    for (int k = 0; k < F / 2; ++k) dt.v[k] = powf(temperature, 2.0f * (float)k / (float)F);
    const long long total = rows * n * (F / 2), nblk = (total + 255) / 256;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (out_is_bf16)
        hipLaunchKernelGGL((sine_pos_embed_kernel<uint16_t>), dim3((unsigned)nblk), dim3(256), 0, st, pos, rows, n, F, scale, dt,
                           static_cast<uint16_t *>(out));
    else
        hipLaunchKernelGGL((sine_pos_embed_kernel<float>), dim3((unsigned)nblk), dim3(256), 0, st, pos, rows, n, F, scale, dt,
                           static_cast<float *>(out));
    return launch_status();
}

extern "C" int rdetr_detections_from_topk(const float *score, const long long *index, const float *boxes, const long long *image_sizes,
                                          int B, int N, int C, int K, float *out, void *stream)
{
    if (B < 0 || N <= 0 || C <= 0 || K < 0) return RDETR_ERR_INVALID_ARG;
    if (B == 0 || K == 0) return RDETR_OK;
    if (!score || !index || !boxes || !image_sizes || !out) return RDETR_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(boxes) % 16) return RDETR_ERR_UNSUPPORTED;
    const long long total = (long long)B * K, nblk = (total + 255) / 256;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(detections_kernel, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream), score, index, boxes,
                       image_sizes, N, C, K, total, out);
    return launch_status();
}

extern "C" int rdetr_scaled_pos(const void *a, const void *scale, const void *query, long long n, int is_bf16, void *pos, void *qp,
                                void *stream)
{
    if (n < 0) return RDETR_ERR_INVALID_ARG;
    if (n == 0) return RDETR_OK;
    if (!a || !scale || !query || !pos || !qp) return RDETR_ERR_INVALID_ARG;
    const long long nblk = (n + 255) / 256;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (is_bf16)
        hipLaunchKernelGGL((scaled_pos_kernel<uint16_t>), dim3((unsigned)nblk), dim3(256), 0, st, static_cast<const uint16_t *>(a),
                           static_cast<const uint16_t *>(scale), static_cast<const uint16_t *>(query), n, static_cast<uint16_t *>(pos),
                           static_cast<uint16_t *>(qp));
    else
        hipLaunchKernelGGL((scaled_pos_kernel<float>), dim3((unsigned)nblk), dim3(256), 0, st, static_cast<const float *>(a),
                           static_cast<const float *>(scale), static_cast<const float *>(query), n, static_cast<float *>(pos),
                           static_cast<float *>(qp));
    return launch_status();
}

extern "C" int rdetr_decoder_reference(const float *reference, const float *valid_ratios, int B, int N, int L, int F, float temperature,
                                       float scale, float *ref_in, void *emb, int emb_is_bf16, void *stream)
{
    if (B < 0 || N < 0 || L <= 0 || F <= 0) return RDETR_ERR_INVALID_ARG;
    if ((F & 1) || F > 128) return RDETR_ERR_UNSUPPORTED;
    if (B == 0 || N == 0) return RDETR_OK;
    if (!reference || !valid_ratios || !ref_in || !emb) return RDETR_ERR_INVALID_ARG;
    GlueDimT dt;
    for (int k = 0; k < F / 2; ++k) dt.v[k] = powf(temperature, 2.0f * (float)k / (float)F);
    const long long total = (long long)B * N * 4 * (F / 2), nblk = (total + 255) / 256;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (emb_is_bf16)
        hipLaunchKernelGGL((decoder_reference_kernel<uint16_t>), dim3((unsigned)nblk), dim3(256), 0, st, reference, valid_ratios, N, L, F,
                           scale, dt, ref_in, static_cast<uint16_t *>(emb), total);
    else
        hipLaunchKernelGGL((decoder_reference_kernel<float>), dim3((unsigned)nblk), dim3(256), 0, st, reference, valid_ratios, N, L, F, scale,
                           dt, ref_in, static_cast<float *>(emb), total);
    return launch_status();
}

extern "C" int rdetr_zero_masked_rows(void *x, const unsigned char *mask, long long rows, int row_bytes, long long ld_bytes,
                                      void *stream)
{
    if (rows < 0 || row_bytes <= 0 || ld_bytes < row_bytes) return RDETR_ERR_INVALID_ARG;
    if ((row_bytes & 15) || (ld_bytes & 15)) return RDETR_ERR_UNSUPPORTED;
    if (rows == 0) return RDETR_OK;
    if (!x || !mask || (reinterpret_cast<uintptr_t>(x) & 15)) return RDETR_ERR_INVALID_ARG;
    const long long nblk = (rows + 255) / 256;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(zero_masked_rows_kernel, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<unsigned char *>(x), mask, rows, row_bytes, ld_bytes);
    return launch_status();
}

extern "C" int rdetr_row_max(const void *x, int is_bf16, long long rows, int C, long long ldx, void *out, void *stream)
{
    if (rows < 0 || C <= 0 || ldx < C) return RDETR_ERR_INVALID_ARG;
    if (rows == 0) return RDETR_OK;
    if (!x || !out) return RDETR_ERR_INVALID_ARG;
    const long long nblk = (rows + 15) / 16;
    if (nblk > 0x7fffffffll) return RDETR_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (is_bf16)
        hipLaunchKernelGGL((row_max_kernel<uint16_t>), dim3((unsigned)nblk), dim3(256), 0, st, static_cast<const uint16_t *>(x), rows, C,
                           ldx, static_cast<uint16_t *>(out));
    else
        hipLaunchKernelGGL((row_max_kernel<float>), dim3((unsigned)nblk), dim3(256), 0, st, static_cast<const float *>(x), rows, C, ldx,
                           static_cast<float *>(out));
    return launch_status();
}

extern "C" int rdetr_nchw_to_tokens(const void *src, const void *add_vec, int is_bf16, int B, int C, int P,
                                    long long out_image_stride, long long ld_out, void *out, void *stream)
{
    if (B < 0 || C <= 0 || P < 0 || ld_out < C || out_image_stride < 0) return RDETR_ERR_INVALID_ARG;
    if (B == 0 || P == 0) return RDETR_OK;
    if (!src || !out) return RDETR_ERR_INVALID_ARG;
    if (B > 65535 || (C + 63) / 64 > 65535) return RDETR_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((P + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto al16 = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (is_bf16 && P % 8 == 0 && C % 8 == 0 && ld_out % 8 == 0 && out_image_stride % 8 == 0 && al16(src) && al16(out) &&
        (!add_vec || al16(add_vec)))
        hipLaunchKernelGGL(nchw_to_tokens_bf16_vec_kernel, grid, dim3(256), 0, st, static_cast<const uint16_t *>(src),
                           static_cast<const uint16_t *>(add_vec), C, P, out_image_stride, ld_out, static_cast<uint16_t *>(out));
    else if (is_bf16)
        hipLaunchKernelGGL((nchw_to_tokens_kernel<uint16_t>), grid, dim3(256), 0, st, static_cast<const uint16_t *>(src),
                           static_cast<const uint16_t *>(add_vec), C, P, out_image_stride, ld_out, static_cast<uint16_t *>(out));
    else
        hipLaunchKernelGGL((nchw_to_tokens_kernel<float>), grid, dim3(256), 0, st, static_cast<const float *>(src),
                           static_cast<const float *>(add_vec), C, P, out_image_stride, ld_out, static_cast<float *>(out));
    return launch_status();
}
