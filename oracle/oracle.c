/*
 * oracle.c -- scalar C restatement of the Relation-DETR hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so.  The shipped path is the HIP library
 * in relation_detr_amd/csrc; it never links or calls anything in here.
 *
 * Parity pin: tests/test_oracle_golden.py checks every entry point against golden vectors
 * generated from the reference's own Python (oracle/gen_golden.py, tests/golden/).
 *
 * Citations are into /root/reference:
 *   oracle_msda_forward_f32   models/bricks/ops/cuda/ms_deform_im2col_cuda.cuh:22-73 (bilinear),
 *                             :226-288 (forward kernel); same math as
 *                             models/bricks/ms_deform_attn.py:159-212.
 *   oracle_msda_backward_f32  ms_deform_im2col_cuda.cuh:76-148 (one-sample backward), :290-392.
 *   oracle_relation_bias_f32  models/bricks/relation_transformer.py:481-490,520-532 and
 *                             models/bricks/position_encoding.py:101-138.
 *   oracle_bias_softmax_f32   softmax(QK^T/sqrt(d) + bias) inside nn.MultiheadAttention as called
 *                             at models/bricks/relation_transformer.py:452-459.
 *
 * Layouts (all contiguous, row-major):
 *   value        [B, S, H, D]            levels packed along S in level_start order, row = y*W_l + x
 *   shapes       [L, 2] int64 (h, w)     level_start [L] int64
 *   loc          [B, Nq, H, L, P, 2]     (x, y) normalised to [0,1]
 *   attn         [B, Nq, H, L, P]
 *   out          [B, Nq, H*D]            channel = m*D + c
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* acc_double != 0 accumulates the L*P sum in double (a tighter reference for error budgets). */
int oracle_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                            const float *loc, const float *attn, int B, int S, int H, int D, int L,
                            int Nq, int P, int acc_double, float *out)
{
    if (B < 0 || S < 0 || H <= 0 || D <= 0 || L <= 0 || P <= 0 || Nq < 0) return -1;
    const int64_t pix_stride = (int64_t)H * D;
    const int64_t rows = (int64_t)B * Nq * H;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const int m = (int)(r % H);
        const int64_t bq = r / H;
        const int b = (int)(bq / Nq);
        const float *vb = value + (int64_t)b * S * pix_stride;
        const float *lp = loc + r * L * P * 2;
        const float *ap = attn + r * L * P;
        float *o = out + bq * pix_stride + (int64_t)m * D;
        for (int c = 0; c < D; ++c) {
            float accf = 0.f;
            double accd = 0.0;
            for (int l = 0; l < L; ++l) {
                const int hl = (int)shapes[2 * l], wl = (int)shapes[2 * l + 1];
                const float *vl = vb + level_start[l] * pix_stride + (int64_t)m * D + c;
                for (int p = 0; p < P; ++p) {
                    const float x = lp[(l * P + p) * 2 + 0] * wl - 0.5f;
                    const float y = lp[(l * P + p) * 2 + 1] * hl - 0.5f;
                    const float a = ap[l * P + p];
                    if (!(y > -1 && x > -1 && y < hl && x < wl)) continue;
                    const int y0 = (int)floorf(y), x0 = (int)floorf(x);
                    const int y1 = y0 + 1, x1 = x0 + 1;
                    const float ly = y - y0, lx = x - x0, hy = 1 - ly, hx = 1 - lx;
                    float v00 = 0, v01 = 0, v10 = 0, v11 = 0;
                    if (y0 >= 0 && x0 >= 0) v00 = vl[((int64_t)y0 * wl + x0) * pix_stride];
                    if (y0 >= 0 && x1 <= wl - 1) v01 = vl[((int64_t)y0 * wl + x1) * pix_stride];
                    if (y1 <= hl - 1 && x0 >= 0) v10 = vl[((int64_t)y1 * wl + x0) * pix_stride];
                    if (y1 <= hl - 1 && x1 <= wl - 1) v11 = vl[((int64_t)y1 * wl + x1) * pix_stride];
                    const float s = hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11;
                    if (acc_double) accd += (double)s * (double)a; else accf += s * a;
                }
            }
            o[c] = acc_double ? (float)accd : accf;
        }
    }
    return 0;
}

/* All three gradient buffers are zeroed here; the caller only allocates them. */
int oracle_msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *level_start,
                             const float *loc, const float *attn, const float *grad_out, int B, int S,
                             int H, int D, int L, int Nq, int P, float *grad_value, float *grad_loc,
                             float *grad_attn)
{
    if (B < 0 || S < 0 || H <= 0 || D <= 0 || L <= 0 || P <= 0 || Nq < 0) return -1;
    const int64_t pix_stride = (int64_t)H * D;
    memset(grad_value, 0, sizeof(float) * (size_t)B * S * pix_stride);
    memset(grad_loc, 0, sizeof(float) * (size_t)B * Nq * H * L * P * 2);
    memset(grad_attn, 0, sizeof(float) * (size_t)B * Nq * H * L * P);
    /* serial over (b,q,m): grad_value scatter has collisions, keep one deterministic order */
    for (int64_t r = 0; r < (int64_t)B * Nq * H; ++r) {
        const int m = (int)(r % H);
        const int64_t bq = r / H;
        const int b = (int)(bq / Nq);
        const float *vb = value + (int64_t)b * S * pix_stride;
        float *gvb = grad_value + (int64_t)b * S * pix_stride;
        const float *go = grad_out + bq * pix_stride + (int64_t)m * D;
        for (int l = 0; l < L; ++l) {
            const int hl = (int)shapes[2 * l], wl = (int)shapes[2 * l + 1];
            const int64_t lbase = level_start[l] * pix_stride + (int64_t)m * D;
            for (int p = 0; p < P; ++p) {
                const int64_t k = r * L * P + (int64_t)l * P + p;
                const float x = loc[2 * k] * wl - 0.5f, y = loc[2 * k + 1] * hl - 0.5f, a = attn[k];
                if (!(y > -1 && x > -1 && y < hl && x < wl)) continue;
                const int y0 = (int)floorf(y), x0 = (int)floorf(x), y1 = y0 + 1, x1 = x0 + 1;
                const float ly = y - y0, lx = x - x0, hy = 1 - ly, hx = 1 - lx;
                const int ok00 = y0 >= 0 && x0 >= 0, ok01 = y0 >= 0 && x1 <= wl - 1;
                const int ok10 = y1 <= hl - 1 && x0 >= 0, ok11 = y1 <= hl - 1 && x1 <= wl - 1;
                const int64_t o00 = lbase + ((int64_t)y0 * wl + x0) * pix_stride, o01 = o00 + pix_stride;
                const int64_t o10 = o00 + (int64_t)wl * pix_stride, o11 = o10 + pix_stride;
                double g_a = 0, g_x = 0, g_y = 0;
                for (int c = 0; c < D; ++c) {
                    const float g = go[c], ga = g * a;
                    float v00 = 0, v01 = 0, v10 = 0, v11 = 0, gy = 0, gx = 0;
                    if (ok00) { v00 = vb[o00 + c]; gy -= hx * v00; gx -= hy * v00; gvb[o00 + c] += hy * hx * ga; }
                    if (ok01) { v01 = vb[o01 + c]; gy -= lx * v01; gx += hy * v01; gvb[o01 + c] += hy * lx * ga; }
                    if (ok10) { v10 = vb[o10 + c]; gy += hx * v10; gx -= ly * v10; gvb[o10 + c] += ly * hx * ga; }
                    if (ok11) { v11 = vb[o11 + c]; gy += lx * v11; gx += ly * v11; gvb[o11 + c] += ly * lx * ga; }
                    g_a += g * (hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11);
                    g_x += (double)wl * gx * ga;
                    g_y += (double)hl * gy * ga;
                }
                grad_attn[k] = (float)g_a;
                grad_loc[2 * k] = (float)g_x;
                grad_loc[2 * k + 1] = (float)g_y;
            }
        }
    }
    return 0;
}

/* src [B,N1,4], tgt [B,N2,4] cxcywh; Wp [Hh, 4*F] (F = num_pos_feats, channel = coord*F + 2k + {sin,cos});
 * bp [Hh]; out [B,Hh,N1,N2].  Follows the reference's fp32 op order: (e*scale)/dim_t. */
int oracle_relation_bias_f32(const float *src, const float *tgt, const float *Wp, const float *bp, int B,
                             int N1, int N2, int Hh, int F, float scale, float temperature, float eps,
                             float *out)
{
    if (F <= 0 || (F & 1) || F > 128 || Hh <= 0 || Hh > 64) return -1;
    const int K = F / 2;
    float dim_t[64];
    for (int k = 0; k < K; ++k) dim_t[k] = powf(temperature, (float)k * 2.0f / (float)F);
#pragma omp parallel for schedule(static) collapse(2)
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N1; ++i) {
            const float *s = src + ((int64_t)b * N1 + i) * 4;
            float feat[4 * 128];
            for (int j = 0; j < N2; ++j) {
                const float *t = tgt + ((int64_t)b * N2 + j) * 4;
                float e[4];
                e[0] = logf(fabsf(s[0] - t[0]) / (s[2] + eps) + 1.0f);
                e[1] = logf(fabsf(s[1] - t[1]) / (s[3] + eps) + 1.0f);
                e[2] = logf((s[2] + eps) / (t[2] + eps));
                e[3] = logf((s[3] + eps) / (t[3] + eps));
                for (int c = 0; c < 4; ++c)
                    for (int k = 0; k < K; ++k) {
                        const float a = (e[c] * scale) / dim_t[k];
                        feat[c * F + 2 * k] = sinf(a);
                        feat[c * F + 2 * k + 1] = cosf(a);
                    }
                for (int h = 0; h < Hh; ++h) {
                    float acc = bp ? bp[h] : 0.f;
                    for (int ch = 0; ch < 4 * F; ++ch) acc += Wp[h * 4 * F + ch] * feat[ch];
                    out[(((int64_t)b * Hh + h) * N1 + i) * N2 + j] = acc > 0.f ? acc : 0.f;
                }
            }
        }
    return 0;
}

/* scores [BH, N1, N2] in place: softmax over the last axis of (scores + bias) with an optional
 * boolean mask [N1,N2] (non-zero = -inf).  bias may be NULL; bias may contain -inf. */
int oracle_bias_softmax_f32(float *scores, const float *bias, const uint8_t *mask, int BH, int N1, int N2)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)BH * N1; ++r) {
        float *row = scores + r * N2;
        const float *brow = bias ? bias + r * N2 : NULL;
        const uint8_t *mrow = mask ? mask + (r % N1) * N2 : NULL;
        float mx = -INFINITY;
        for (int j = 0; j < N2; ++j) {
            float v = row[j] + (brow ? brow[j] : 0.f);
            if (mrow && mrow[j]) v = -INFINITY;
            row[j] = v;
            if (v > mx) mx = v;
        }
        double sum = 0;
        for (int j = 0; j < N2; ++j) { row[j] = expf(row[j] - mx); sum += row[j]; }
        const float inv = (float)(1.0 / sum);
        for (int j = 0; j < N2; ++j) row[j] *= inv;
    }
    return 0;
}
