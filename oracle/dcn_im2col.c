/*
 * ORACLE (test infrastructure, not product code).
 *
 * CPU restatement of the reference's modulated deformable im2col (DCNv2 forward
 * gather).  Follows /root/reference/src/lib/models/networks/DCNv2/src/cpu/
 * dcn_v2_im2col_cpu.cpp:27-56 (bilinear sampler) and :125-195 (im2col kernel).
 * Written from the algorithm description, plain C, scalar; an optional OpenMP
 * pragma parallelises over (batch, channel) which does not change any result
 * (every output element is produced by exactly one iteration).
 *
 * Layouts (all float32, contiguous, identical to the reference):
 *   im     [B, C, H, W]
 *   offset [B, dg*2*kh*kw, Ho, Wo]   channel 2*t = dh of tap t, 2*t+1 = dw
 *   mask   [B, dg*kh*kw,   Ho, Wo]
 *   col    [B, C*kh*kw, Ho*Wo]       row index c*kh*kw + t
 */
#include <math.h>
#include <stddef.h>

static float bilinear_zero_pad(const float *plane, int H, int W, float h, float w)
{
    /* reference :27-56: corners outside the image contribute 0 */
    int h0 = (int)floorf(h), w0 = (int)floorf(w);
    int h1 = h0 + 1, w1 = w0 + 1;
    float lh = h - (float)h0, lw = w - (float)w0;
    float hh = 1.0f - lh, hw = 1.0f - lw;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    if (h0 >= 0 && w0 >= 0) v1 = plane[(size_t)h0 * W + w0];
    if (h0 >= 0 && w1 <= W - 1) v2 = plane[(size_t)h0 * W + w1];
    if (h1 <= H - 1 && w0 >= 0) v3 = plane[(size_t)h1 * W + w0];
    if (h1 <= H - 1 && w1 <= W - 1) v4 = plane[(size_t)h1 * W + w1];
    float w1_ = hh * hw, w2_ = hh * lw, w3_ = lh * hw, w4_ = lh * lw;
    return (w1_ * v1 + w2_ * v2 + w3_ * v3 + w4_ * v4);
}

void cp_oracle_dcn_im2col(const float *im, const float *offset, const float *mask,
                          int B, int C, int H, int W, int Ho, int Wo,
                          int kh, int kw, int ph, int pw, int sh, int sw,
                          int dh, int dw, int dg, float *col)
{
    const int T = kh * kw;
    const int cpg = C / dg;
    const size_t HWo = (size_t)Ho * Wo;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int c = 0; c < C; ++c) {
            const int g = c / cpg;
            const float *plane = im + ((size_t)b * C + c) * H * W;
            const float *off = offset + ((size_t)b * dg + g) * 2 * T * HWo;
            const float *msk = mask + ((size_t)b * dg + g) * T * HWo;
            float *out = col + ((size_t)b * C * T + (size_t)c * T) * HWo;
            for (int ho = 0; ho < Ho; ++ho) {
                for (int wo = 0; wo < Wo; ++wo) {
                    const size_t p = (size_t)ho * Wo + wo;
                    const int h_in = ho * sh - ph, w_in = wo * sw - pw;
                    for (int i = 0; i < kh; ++i) {
                        for (int j = 0; j < kw; ++j) {
                            const int t = i * kw + j;
                            const float oh = off[(size_t)(2 * t) * HWo + p];
                            const float ow = off[(size_t)(2 * t + 1) * HWo + p];
                            const float m = msk[(size_t)t * HWo + p];
                            const float h_im = (float)(h_in + i * dh) + oh;
                            const float w_im = (float)(w_in + j * dw) + ow;
                            float val = 0.f;
                            /* reference :180: open interval (-1, H) x (-1, W) */
                            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
                                val = bilinear_zero_pad(plane, H, W, h_im, w_im);
                            out[(size_t)t * HWo + p] = val * m;
                        }
                    }
                }
            }
        }
    }
}
