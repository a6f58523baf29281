// Generic modulated deformable convolution (DCNv2 forward) with the reference's NCHW layouts: any channel count, kernel
// size, stride, padding, dilation and `deformable_group` -- the part of `_ext.dcn_v2_forward`'s contract
// (/root/reference/src/lib/models/networks/DCNv2/src/dcn_v2.h:9-23, cuda/dcn_v2_cuda.cu:42-172) that CenterPose itself
// never exercises (pose_dla_dcn.py:384 always builds 3x3 / stride 1 / pad 1 / dg 1 on 64..512 channels; those shapes
// run on dcn16p.hip / dcn16.hip / igemm.hip<DCN>).  The reference's own self-checks do use it: testcpu.py:32-67
// (2 -> 2 channels, dg 1) and `example_dconv` :169-180 (dg 2).
//
// Not a fast path: float32 FMAs on the vector ALUs, no matrix cores.  A workgroup owns 64 consecutive output pixels of
// one image and up to 64 output channels; the K = C*kh*kw reduction is walked in chunks of 32 rows: the chunk's
// "columns" (bilinear sample x mask, exactly dcn_v2_im2col_cuda.cu:25-54,125-195: corners outside the image contribute
// 0, samples outside the open interval (-1, H) x (-1, W) are 0) are built once in LDS by all 256 lanes, then every lane
// accumulates 16 output channels of its pixel from them in ascending k.  The reference's `columns` buffer in HBM
// (C*kh*kw*Ho*Wo floats per image) never exists.
#include "cp_common.h"

namespace {

constexpr int GP = 64;   // output pixels per workgroup
constexpr int GK = 32;   // K rows per chunk
constexpr int GCO = 16;  // output channels per lane (x 4 lane groups = 64 per workgroup)

struct DcnGenericParams {
    const float* x;       // [B,C,H,W]
    const float* w;       // [Co,C,kh,kw]
    const float* bias;    // [Co]
    const float* offset;  // [B,dg*2*kh*kw,Ho,Wo]
    const float* mask;    // [B,dg*kh*kw,Ho,Wo]
    float* out;           // [B,Co,Ho,Wo]
    int B, C, H, W, Co, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, dg;
};

__device__ __forceinline__ float bilinear_zero(const float* __restrict__ plane, int H, int W, float h, float w) {
    const int h0 = (int)floorf(h), w0 = (int)floorf(w);
    const int h1 = h0 + 1, w1 = w0 + 1;
    const float lh = h - (float)h0, lw = w - (float)w0;
    const float hh = 1.f - lh, hw = 1.f - lw;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    if (h0 >= 0 && w0 >= 0) v1 = plane[(size_t)h0 * W + w0];
    if (h0 >= 0 && w1 <= W - 1) v2 = plane[(size_t)h0 * W + w1];
    if (h1 <= H - 1 && w0 >= 0) v3 = plane[(size_t)h1 * W + w0];
    if (h1 <= H - 1 && w1 <= W - 1) v4 = plane[(size_t)h1 * W + w1];
    // the reference's expression order: w1*v1 + w2*v2 + w3*v3 + w4*v4 with w_i the products of the 1-D weights
    return (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4;
}

__global__ __launch_bounds__(256) void dcn_generic_kernel(const DcnGenericParams p) {
    __shared__ float col[GK][GP + 1];
    const int T = p.kh * p.kw;
    const int K = p.C * T;
    const int HWo = p.Ho * p.Wo;
    const int tiles_per_img = (HWo + GP - 1) / GP;
    const int b = blockIdx.x / tiles_per_img;
    const int p0 = (blockIdx.x - b * tiles_per_img) * GP;
    const int co0 = blockIdx.y * (4 * GCO);
    const int lane_p = threadIdx.x & (GP - 1);  // pixel of this lane in phase 2
    const int grp = threadIdx.x >> 6;           // 0..3: output-channel group in phase 2
    const int cpg = p.C / p.dg;                 // channels per deformable group

    float acc[GCO];
#pragma unroll
    for (int i = 0; i < GCO; ++i) acc[i] = 0.f;

    for (int k0 = 0; k0 < K; k0 += GK) {
        // phase 1: columns of rows k0 .. k0+GK-1 for the tile's pixels (each lane GK*GP/256 = 8 elements)
        for (int e = threadIdx.x; e < GK * GP; e += 256) {
            const int kk = e / GP, pp = e - kk * GP;
            const int k = k0 + kk, pix = p0 + pp;
            float v = 0.f;
            if (k < K && pix < HWo) {
                const int c = k / T, t = k - c * T;
                const int i = t / p.kw, j = t - i * p.kw;
                const int g = c / cpg;
                const int ho = pix / p.Wo, wo = pix - ho * p.Wo;
                const size_t ob = ((size_t)b * p.dg + g) * 2 * T;
                const float oh = p.offset[(ob + 2 * t) * HWo + pix];
                const float ow = p.offset[(ob + 2 * t + 1) * HWo + pix];
                const float m = p.mask[(((size_t)b * p.dg + g) * T + t) * HWo + pix];
                const float h_im = (float)(ho * p.sh - p.ph + i * p.dh) + oh;
                const float w_im = (float)(wo * p.sw - p.pw + j * p.dw) + ow;
                if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W)
                    v = bilinear_zero(p.x + ((size_t)b * p.C + c) * p.H * p.W, p.H, p.W, h_im, w_im);
                v *= m;
            }
            col[kk][pp] = v;
        }
        __syncthreads();
        // phase 2: out[co][pix] += w[co][k] * col[k][pix], k ascending
        const int kn = min(GK, K - k0);
#pragma unroll
        for (int i = 0; i < GCO; ++i) {
            const int co = co0 + grp * GCO + i;
            if (co < p.Co) {
                const float* wr = p.w + (size_t)co * K + k0;
                float a = acc[i];
                for (int kk = 0; kk < kn; ++kk) a = fmaf(wr[kk], col[kk][lane_p], a);
                acc[i] = a;
            }
        }
        __syncthreads();
    }
    const int pix = p0 + lane_p;
    if (pix < HWo) {
#pragma unroll
        for (int i = 0; i < GCO; ++i) {
            const int co = co0 + grp * GCO + i;
            if (co < p.Co) p.out[((size_t)b * p.Co + co) * HWo + pix] = acc[i] + p.bias[co];
        }
    }
}

}  // namespace

int cp_launch_dcn_generic(hipStream_t s, const float* x, const float* w, const float* bias, const float* offset,
                          const float* mask, float* out, int B, int C, int H, int W, int Co, int Ho, int Wo, int kh, int kw,
                          int sh, int sw, int ph, int pw, int dh, int dw, int dg) {
    DcnGenericParams p{x, w, bias, offset, mask, out, B, C, H, W, Co, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, dg};
    const long long tiles = (long long)B * ((Ho * Wo + GP - 1) / GP);
    if (tiles < 1 || tiles > 0x7fffffffLL) return CP_ERR_INVALID;
    hipLaunchKernelGGL(dcn_generic_kernel, dim3((unsigned)tiles, (Co + 4 * GCO - 1) / (4 * GCO)), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}
