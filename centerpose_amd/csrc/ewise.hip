// HBM-bound data-movement / element-wise kernels of the backbone (float32 NHWC, 16 B per lane).
#include "cp_common.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TPB = 256;

inline int grid_for(size_t n, int cap = 256 * 16) {
    size_t g = (n + TPB - 1) / TPB;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// [B,C,H,W] -> [B,H,W,Cpad] (zero pad), used for the network inputs (image / pre_img / pre_hm / pre_hm_hp).
// One pixel per lane: C coalesced plane reads, Cpad/4 16-byte stores.
template <int CPAD>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int HW) {
    const int total = B * HW;  // < 2^31 for every supported shape
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / HW, p = i - b * HW;
        const float* src = in + (size_t)b * C * HW + p;
        float v[CPAD];
#pragma unroll
        for (int c = 0; c < CPAD; ++c) v[c] = c < C ? src[(size_t)c * HW] : 0.f;
        float4* dst = reinterpret_cast<float4*>(out + (size_t)i * CPAD);
#pragma unroll
        for (int q = 0; q < CPAD / 4; ++q) dst[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
}

// General [B,C,HW] -> [B,HW,C] through a 32 x 32 LDS tile (both sides coalesced); C % 4 == 0 not required.
__global__ void nchw_to_nhwc_tiled_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? in[((size_t)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (p < HW && c < C) out[((size_t)b * HW + p) * C + c] = tile[tx][r];
    }
}

// [B,H,W,ldi] (first C channels) -> [B,C,H,W]
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int HW,
                                    int ldi) {
    const size_t total = (size_t)B * C * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i % HW, bc = i / HW;
        const size_t c = bc % C, b = bc / C;
        out[i] = in[(b * HW + p) * ldi + c];
    }
}

__global__ void maxpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C4) {
    // in [B,H,W,C], out [B,H/2,W/2,C]; one float4 of channels per thread
    const int Ho = H >> 1, Wo = W >> 1;
    const size_t total = (size_t)B * Ho * Wo * C4;
    const float4* in4 = reinterpret_cast<const float4*>(in);
    float4* out4 = reinterpret_cast<float4*>(out);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const size_t base = (((size_t)b * H + 2 * ho) * W + 2 * wo) * C4 + c;
        const float4 a = in4[base], bb = in4[base + C4], cc = in4[base + (size_t)W * C4],
                     d = in4[base + (size_t)W * C4 + C4];
        float4 r;
        r.x = fmaxf(fmaxf(a.x, bb.x), fmaxf(cc.x, d.x));
        r.y = fmaxf(fmaxf(a.y, bb.y), fmaxf(cc.y, d.y));
        r.z = fmaxf(fmaxf(a.z, bb.z), fmaxf(cc.z, d.z));
        r.w = fmaxf(fmaxf(a.w, bb.w), fmaxf(cc.w, d.w));
        out4[i] = r;
    }
}

// Depth-wise ConvTranspose2d(C, C, k=2f, stride=f, padding=f/2, groups=C) + add (IDAUp.forward,
// pose_dla_dcn.py:411-417).  out[y,x,c] = add[y,x,c] + sum_{ky,kx} in[(y+p-ky)/f, (x+p-kx)/f, c] * w[c,ky,kx]
// over taps with (y+p-ky) % f == 0 and the source inside the image: exactly 2x2 taps per output pixel.
// w is [C,1,k,k] exactly as in the checkpoint.
__global__ void upsample_add_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                    const float* __restrict__ add, float* __restrict__ out, int B, int H, int W, int C,
                                    int f, unsigned* __restrict__ out_amax) {
    // per-channel kernels staged once per workgroup as [tap][channel] so a lane's 4 channels are one ds_read_b128
    // (16-byte aligned and indexed in float4 units below: as a plain float array with a float index the compiler read a lane's four
    // weights with two ds_read2_b32 -- lanes 16 bytes apart, a 4-way bank conflict on every read: three quarters of the kernel's LDS
    // cycles, a third of its clocks x CUs, profiles/r06_pmc_sq_counters.txt)
    extern __shared__ __attribute__((aligned(16))) float wt[];
    const int k = 2 * f, kk = k * k;
    for (int i = threadIdx.x; i < C * kk; i += blockDim.x) {
        const int c = i / kk, t = i - c * kk;
        wt[t * C + c] = w[i];
    }
    __syncthreads();
    // 32-bit index math throughout (B*Ho*Wo*C/4 < 2^31 for every supported shape)
    const int p = f / 2, Ho = H * f, Wo = W * f, C4 = C >> 2;
    const unsigned total = (unsigned)B * Ho * Wo * C4;
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((unsigned)B * H * W * C * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_add = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(add ? add : in), 0, add ? (int)(total * 16u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(total * 16u), 0x00020000);
    float amax = 0.f;
    // One output piece = 16 bytes (4 channels of one pixel): `add` + exactly 2 x 2 source taps.  All five loads are issued before
    // anything is used, without a branch -- a tap outside the source (and a piece beyond the tensor) is requested beyond the buffer
    // descriptor and comes back as zeros (s + 0 * w = s: the sum is the one of the taps inside) -- and a thread handles TWO pieces per
    // turn, so ten loads are in flight per lane: the kernel's waves spend three quarters of their cycles waiting for HBM.
    struct Piece {
        float4 v[2][2], acc;
        int wi;   // float4 index of tap (ky0, kx0)'s weights; the other taps are f and f * k further on
    };
    auto request = [&](unsigned i, Piece& q) {
        const unsigned c4 = i % (unsigned)C4;
        unsigned t = i / (unsigned)C4;
        const int x = (int)(t % (unsigned)Wo);
        t /= (unsigned)Wo;
        const int y = (int)(t % (unsigned)Ho);
        const int b = (int)(t / (unsigned)Ho);
        const bool live = i < total;
        // ky ranges over { (y+p) % f, (y+p) % f + f }
        const int ky0 = (y + p) % f, kx0 = (x + p) % f;
        const int iy0 = (y + p - ky0) / f, ix0 = (x + p - kx0) / f;  // source of tap (ky0, kx0); the other tap is one less
        q.wi = (ky0 * k + kx0) * C4 + (int)c4;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int iy = iy0 - a, ix = ix0 - bb;
                const bool ok = live && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(r_in, (int)(ok ? (((unsigned)(b * H + iy) * W + ix) * C4 + c4) * 16u : 0xffffffffu), 0, 0);
                q.v[a][bb] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w));
            }
        const u32x4 ra = __builtin_amdgcn_raw_buffer_load_b128(r_add, (int)(live ? i * 16u : 0xffffffffu), 0, 0);   // (no `add`: a descriptor of size 0)
        q.acc = make_float4(__uint_as_float(ra.x), __uint_as_float(ra.y), __uint_as_float(ra.z), __uint_as_float(ra.w));
    };
    auto finish = [&](unsigned i, const Piece& q) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const float4 wv = reinterpret_cast<const float4*>(wt)[q.wi + (a * k + bb) * f * C4];   // (float4 index: the alignment is provable)
                s.x += q.v[a][bb].x * wv.x;
                s.y += q.v[a][bb].y * wv.y;
                s.z += q.v[a][bb].z * wv.z;
                s.w += q.v[a][bb].w * wv.w;
            }
        float4 acc = q.acc;
        acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w))));   // (a piece beyond the tensor is all zeros)
        const u32x4 pk = {__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w)};
        __builtin_amdgcn_raw_buffer_store_b128(pk, r_out, (int)(i < total ? i * 16u : 0xffffffffu), 0, 0);
    };
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += 2 * stride) {
        Piece q0, q1;
        request(i, q0);
        request(i + stride, q1);
        finish(i, q0);
        finish(i + stride, q1);
    }
    if (out_amax) cp_amax_commit(out_amax, amax);
}

// out = a + b (+ c) (+ d)   (tracking stems: x = base + pre_img + pre_hm + pre_hm_hp, pose_dla_dcn.py:312-318)
__global__ void add_n_kernel(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                             const float4* __restrict__ d, float4* __restrict__ out, size_t n4,
                             unsigned* __restrict__ out_amax) {
    float amax = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 r = a[i];
        const float4 y = b[i];
        r.x += y.x; r.y += y.y; r.z += y.z; r.w += y.w;
        if (c) { const float4 z = c[i]; r.x += z.x; r.y += z.y; r.z += z.z; r.w += z.w; }
        if (d) { const float4 z = d[i]; r.x += z.x; r.y += z.y; r.z += z.z; r.w += z.w; }
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w))));
        out[i] = r;
    }
    if (out_amax) cp_amax_commit(out_amax, amax);
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// ConvGRUCell.forward (convGRU.py:32-39) given the two fused 64->192 convolutions:
//   x3 = [Wir x + b_ir | Wiz x + b_iz | Win x + b_in],  h3 = [Whr h | Whz h | Whn h]  (h3 == nullptr at step 0: h = 0)
//   r = sig(x3r + h3r); z = sig(x3z + h3z); n = tanh(x3n + r * h3n); h' = (1 - z) * n + z * h
__global__ void gru_gate_kernel(const float* __restrict__ x3, const float* __restrict__ h3,
                                const float* __restrict__ hprev, float* __restrict__ hout, size_t M,
                                unsigned* __restrict__ out_amax) {
    const size_t total = M * 16;  // 64 channels / 4
    float amax = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i >> 4;
        const int c = (int)(i & 15) * 4;
        const float4 xr = *reinterpret_cast<const float4*>(x3 + m * 192 + c);
        const float4 xz = *reinterpret_cast<const float4*>(x3 + m * 192 + 64 + c);
        const float4 xn = *reinterpret_cast<const float4*>(x3 + m * 192 + 128 + c);
        float4 hr = make_float4(0, 0, 0, 0), hz = hr, hn = hr, hp = hr;
        if (h3) {
            hr = *reinterpret_cast<const float4*>(h3 + m * 192 + c);
            hz = *reinterpret_cast<const float4*>(h3 + m * 192 + 64 + c);
            hn = *reinterpret_cast<const float4*>(h3 + m * 192 + 128 + c);
            hp = *reinterpret_cast<const float4*>(hprev + m * 64 + c);
        }
        float4 o;
#define CP_GRU(e)                                        \
    {                                                    \
        const float r = sigm(xr.e + hr.e);               \
        const float z = sigm(xz.e + hz.e);               \
        const float n = tanhf(xn.e + r * hn.e);          \
        o.e = (1.f - z) * n + z * hp.e;                  \
    }
        CP_GRU(x) CP_GRU(y) CP_GRU(z) CP_GRU(w)
#undef CP_GRU
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
        *reinterpret_cast<float4*>(hout + m * 64 + c) = o;
    }
    if (out_amax) cp_amax_commit(out_amax, amax);
}

// GroupNorm statistics: one workgroup per (b, slab of pixels); per-group sum / sum of squares in
// double, combined with atomics into stats[b][g][2].
__global__ void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, int HW, int C, int groups,
                                int slab) {
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * slab;
    const int p1 = min(HW, p0 + slab);
    const int cpg = C / groups;       // channels per group (8 for 256/32)
    const int C4 = C >> 2;
    // each thread owns one float4 column position c4 (fixed) and strides over pixels
    const int c4 = threadIdx.x % C4;
    const int prow = threadIdx.x / C4;
    const int rows = blockDim.x / C4;
    double s = 0.0, ss = 0.0;
    for (int p = p0 + prow; p < p1; p += rows) {
        const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)b * HW + p) * C + c4 * 4);
        s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    // a float4 never straddles groups when cpg % 4 == 0
    const int g = (c4 * 4) / cpg;
    __shared__ double sh[2 * 64];
    if (threadIdx.x < 2 * groups && threadIdx.x < 128) sh[threadIdx.x] = 0.0;
    __syncthreads();
    atomicAdd(&sh[2 * g], s);
    atomicAdd(&sh[2 * g + 1], ss);
    __syncthreads();
    if (threadIdx.x < 2 * groups) atomicAdd(&stats[(size_t)b * groups * 2 + threadIdx.x], sh[threadIdx.x]);
}

__global__ void gn_apply_relu_kernel(float* __restrict__ x, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, const double* __restrict__ stats, int B, int HW,
                                     int C, int groups, float eps, unsigned* __restrict__ out_amax) {
    const int C4 = C >> 2, cpg = C / groups;
    const size_t total = (size_t)B * HW * C4;
    const double cnt = (double)HW * cpg;
    float amax = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const size_t b = i / ((size_t)HW * C4);
        const int g = (c4 * 4) / cpg;
        const double s = stats[(b * groups + g) * 2], ss = stats[(b * groups + g) * 2 + 1];
        const double mean = s / cnt;
        double var = ss / cnt - mean * mean;
        if (var < 0) var = 0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float mu = (float)mean;
        float4 v = reinterpret_cast<float4*>(x)[i];
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c4 * 4);
        const float4 be = *reinterpret_cast<const float4*>(beta + c4 * 4);
        v.x = fmaxf((v.x - mu) * rstd * ga.x + be.x, 0.f);
        v.y = fmaxf((v.y - mu) * rstd * ga.y + be.y, 0.f);
        v.z = fmaxf((v.z - mu) * rstd * ga.z + be.z, 0.f);
        v.w = fmaxf((v.w - mu) * rstd * ga.w + be.w, 0.f);
        amax = fmaxf(amax, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        reinterpret_cast<float4*>(x)[i] = v;
    }
    if (out_amax) cp_amax_commit(out_amax, amax);
}

// PyTorch conv weight [Cout][Cin][KH*KW] -> implicit-GEMM B matrix wp[(tap*CinP + ci) * CoutPad + coff + co]
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int taps,
                                   int CinP, int CoutPad, int coff) {
    const size_t total = (size_t)Cout * Cin * taps;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % taps);
        const size_t r = i / taps;
        const int ci = (int)(r % Cin), co = (int)(r / Cin);
        wp[((size_t)t * CinP + ci) * CoutPad + coff + co] = w[i];
    }
}

__global__ void gn_finalize_kernel(const double* __restrict__ stats, float* __restrict__ mr, int n, double count,
                                   float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double mean = stats[2 * i] / count;
    double var = stats[2 * i + 1] / count - mean * mean;
    if (var < 0) var = 0;
    mr[2 * i] = (float)mean;
    mr[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ void gn_affine_kernel(const double* __restrict__ stats, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float* __restrict__ a, float* __restrict__ d, int B, int C,
                                 int groups, double count, float eps, const unsigned* __restrict__ x_amax,
                                 unsigned* __restrict__ y_amax) {
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i0 < B * C;
    const int i = live ? i0 : 0;
    const int b = i / C, c = i - b * C, g = c / (C / groups);
    const double mean = stats[2 * (b * groups + g)] / count;
    double var = stats[2 * (b * groups + g) + 1] / count - mean * mean;
    if (var < 0) var = 0;
    const float mu = (float)mean, rs = (float)(1.0 / sqrt(var + (double)eps));
    const float av = rs * gamma[c], dv = beta[c] - mu * av;
    if (live) {
        a[i] = av;
        d[i] = dv;
    }
    if (y_amax) {
        // |relu(a*x + d)| <= |a| * max|x| + |d|; a few ulps of slack for the rounding of the products
        const float xm = x_amax ? __uint_as_float(cp_amax_read(x_amax)) : 0.f;
        cp_amax_commit(y_amax, live ? (fabsf(av) * xm + fabsf(dv)) * 1.0001f : 0.f);
    }
}

// Final 1x1 convolution of a GroupNorm'd prediction head (pose_dla_dcn.py:505-520: conv3x3 -> GroupNorm -> ReLU ->
// conv1x1 with 1..16 output maps): y[m][n] = bias[n] + sum_c relu(a[b][c] x[m][c] + d[b][c]) w[c][n] on the vector ALUs
// in float32.  The layer reads a 256-channel hidden tensor (1 KB per pixel, 537 MB at batch 32) to write <= 64 bytes
// per pixel: it is an HBM stream, and the matrix-core version (igemm16p_kernel<.., GNIN>, 128-pixel tiles, a barrier
// per 32 channels) measured 3.8 TB/s on it.  Here 16 lanes share a pixel (16 channels each: four 16-byte loads, each
// 16-lane group reading 256 contiguous bytes), every lane accumulates its partial sums for all n, the 16 partials are
// combined with DPP row rotations, and a wave hands 64 consecutive pixels per output map to one coalesced store.
// Weights sit in LDS as [n / 4][row][4] with row = (16 (4 i + k) + j) for channel 4 j + 64 i + k: the 16 lanes of a pixel
// (j = 0..15, same i, k) read 16 consecutive 16-byte rows -- conflict-free.
template <int NQ>
__global__ __launch_bounds__(256) void gn_final_kernel(const float* __restrict__ x, const float* __restrict__ ga,
                                                       const float* __restrict__ gd, const float* __restrict__ wp,
                                                       const float* __restrict__ bias, float* __restrict__ out, int C,
                                                       int N, int wld, int HW, int sigmoid) {
    constexpr int NP = 4 * NQ;
    extern __shared__ __attribute__((aligned(16))) float gw[];  // [NQ][C][4]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < C * NP; i += 256) {
        const int c = i / NP, n = i - c * NP;
        const int row = 16 * (4 * (c >> 6) + (c & 3)) + ((c & 63) >> 2);
        gw[((n >> 2) * C + row) * 4 + (n & 3)] = n < N ? wp[(size_t)c * wld + n] : 0.f;
    }
    __syncthreads();
    const int j = lane & 15, g = lane >> 4;
    const size_t px0 = ((size_t)blockIdx.x * 4 + wid) * 64;  // this wave's 64 consecutive pixels (one image: HW % 64 == 0)
    const int b = (int)(px0 / HW);
    const int CI = C / 64;  // float4 per lane and pixel (channels 4 j + 64 i .. + 3), <= 4
    float keep[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) keep[n] = 0.f;
    float4 a4[4], d4[4];  // the GroupNorm affine of this lane's channels (one image per wave)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c0 = 4 * j + 64 * (i < CI ? i : 0);
        a4[i] = *reinterpret_cast<const float4*>(ga + (size_t)b * C + c0);
        d4[i] = *reinterpret_cast<const float4*>(gd + (size_t)b * C + c0);
    }
    for (int t = 0; t < 16; ++t) {
        const float* xp = x + (px0 + 4 * t + g) * C;
        float acc[NP];
#pragma unroll
        for (int n = 0; n < NP; ++n) acc[n] = 0.f;
        float4 xv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < CI) xv[i] = *reinterpret_cast<const float4*>(xp + 4 * j + 64 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= CI) break;
            const float4 v = xv[i];
            const float h[4] = {fmaxf(fmaf(a4[i].x, v.x, d4[i].x), 0.f), fmaxf(fmaf(a4[i].y, v.y, d4[i].y), 0.f),
                                fmaxf(fmaf(a4[i].z, v.z, d4[i].z), 0.f), fmaxf(fmaf(a4[i].w, v.w, d4[i].w), 0.f)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* wr = gw + (16 * (4 * i + k) + j) * 4;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wr + q * C * 4);
                    acc[4 * q] = fmaf(h[k], w4.x, acc[4 * q]);
                    acc[4 * q + 1] = fmaf(h[k], w4.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(h[k], w4.z, acc[4 * q + 2]);
                    acc[4 * q + 3] = fmaf(h[k], w4.w, acc[4 * q + 3]);
                }
            }
        }
        // sum over the 16 lanes of the pixel: rotations inside the DPP row (row_ror:8 / 4 / 2 / 1)
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            float v = acc[n];
            v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x128, 0xf, 0xf, false));
            v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x124, 0xf, 0xf, false));
            v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x122, 0xf, 0xf, false));
            v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x121, 0xf, 0xf, false));
            keep[n] = (j == t) ? v : keep[n];  // lane (g, j) ends up with pixel 4 j + g
        }
    }
    const size_t hw = px0 - (size_t)b * HW + 4 * j + g;
#pragma unroll
    for (int n = 0; n < NP; ++n)
        if (n < N) {
            float y = keep[n] + (bias ? bias[n] : 0.f);
            if (sigmoid) y = 1.f / (1.f + expf(-y));
            out[((size_t)b * N + n) * HW + hw] = y;
        }
}

// BaseDetector.pre_process on device (base_detector.py:127-134): cv2.warpAffine(INTER_LINEAR, constant 0 border) of an
// 8-bit HWC BGR frame to the network input size, then (x / 255 - mean) / std, written NCHW float32.
// The warp follows OpenCV's fixed-point arithmetic (imgwarp.cpp WarpAffineInvoker + remapBilinear, restated in
// oracle/cv_emul.py): source coordinates with 10 fractional bits (round-half-even of m * x * 1024, + 16), reduced to 5,
// integer tap weights (32 - fx)(32 - fy) * 32 that sum to 2^15, result (sum + 2^14) >> 15 as an 8-bit value -- so the
// network sees exactly the grey levels the reference's pre-process produces, not a float blend of them.
struct PreParams {
    double m[6];  // INVERSE map (destination -> source), float64 as cv::invertAffineTransform leaves it
    double mean[3], std[3];
};

__global__ void preprocess_kernel(const unsigned char* __restrict__ img, int H, int W, float* __restrict__ out, int OH,
                                  int OW, PreParams pp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= OH * OW) return;
    img += (size_t)blockIdx.y * H * W * 3;  // frame blockIdx.y of a batch (cp_preprocess_batch: one transform for all)
    out += (size_t)blockIdx.y * 3 * OH * OW;
    const int y = i / OW, x = i - y * OW;
    // rint() = lrint / cvRound: round half to even
    const long long adelta = (long long)rint(pp.m[0] * (double)x * 1024.0), bdelta = (long long)rint(pp.m[3] * (double)x * 1024.0);
    const long long X0 = (long long)rint((pp.m[1] * (double)y + pp.m[2]) * 1024.0) + 16;
    const long long Y0 = (long long)rint((pp.m[4] * (double)y + pp.m[5]) * 1024.0) + 16;
    const long long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const long long sx = X >> 5, sy = Y >> 5;
    const int fx = (int)(X & 31), fy = (int)(Y & 31);
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const long long xx = sx + dx, yy = sy + dy;
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            const int wgt = (dx ? fx : 32 - fx) * (dy ? fy : 32 - fy) * 32;
            const unsigned char* px = img + ((size_t)yy * W + (size_t)xx) * 3;
            acc[0] += wgt * px[0];
            acc[1] += wgt * px[1];
            acc[2] += wgt * px[2];
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int v8 = (acc[c] + (1 << 14)) >> 15;
        // numpy: (uint8 / 255. - mean) / std in float64, then .astype(float32)
        out[(size_t)c * OH * OW + i] = (float)(((double)v8 / 255.0 - pp.mean[c]) / pp.std[c]);
    }
}

// cv2.resize(img, (OW, OH)) (INTER_LINEAR) of an 8-bit HWC frame, OpenCV's fixed-point form (resize.cpp: coefficients
// with 11 fractional bits, horizontal pass in int32, vertical ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
// the `scale != 1` branch of pre_process (base_detector.py:128), multi-scale testing.
// Coefficient set-up in OpenCV's order (resize.cpp, the INTER_LINEAR branch of cv::resize): scale_x = 1. / inv_scale_x with
// inv_scale_x = (double)dsize / ssize; fx = (float)((dx + 0.5) * scale_x - 0.5) is rounded to float BEFORE cvFloor and the
// subtraction (fx -= sx, in float), then clamped at both borders.  PARITY UNPINNED until checked against a real cv2.
__device__ __forceinline__ void resize_coef(int d, int dst, int src, int* s0, int* s1, int* a0, int* a1) {
    const double inv_scale = (double)dst / (double)src;
    const double scale = 1.0 / inv_scale;
    float ff = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(ff);
    ff -= (float)s;
    if (s < 0) { ff = 0.f; s = 0; }
    if (s >= src - 1) { ff = 0.f; s = src - 1; }
    *s0 = s;
    *s1 = min(s + 1, src - 1);
    *a0 = (int)rintf((1.f - ff) * 2048.f);
    *a1 = (int)rintf(ff * 2048.f);
}

__global__ void resize_u8_kernel(const unsigned char* __restrict__ img, int H, int W, int C, unsigned char* __restrict__ out,
                                 int OH, int OW) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= OH * OW) return;
    const int y = i / OW, x = i - y * OW;
    int x0, x1, ax0, ax1, y0, y1, ay0, ay1;
    resize_coef(x, OW, W, &x0, &x1, &ax0, &ax1);
    resize_coef(y, OH, H, &y0, &y1, &ay0, &ay1);
    for (int c = 0; c < C; ++c) {
        const int S0 = img[((size_t)y0 * W + x0) * C + c] * ax0 + img[((size_t)y0 * W + x1) * C + c] * ax1;
        const int S1 = img[((size_t)y1 * W + x0) * C + c] * ax0 + img[((size_t)y1 * W + x1) * C + c] * ax1;
        int v = (((ay0 * (S0 >> 4)) >> 16) + ((ay1 * (S1 >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        out[(size_t)i * C + c] = (unsigned char)v;
    }
}

// kp_module merge of the stacked hourglass (large_hourglass.py:186-188): out = up1 + Upsample(scale_factor=2)(low3),
// nearest neighbour.  up1 / out: [B, 2H, 2W, C], low: [B, H, W, C]; one float4 per lane.
__global__ void upsample2_nearest_add_kernel(const float* __restrict__ up1, const float* __restrict__ low,
                                             float* __restrict__ out, int B, int H, int W, int C4,
                                             unsigned* __restrict__ out_amax) {
    const size_t total = (size_t)B * 2 * H * 2 * W * C4;
    float amax = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int x = (int)(r % (2 * W));
        r /= 2 * W;
        const int y = (int)(r % (2 * H));
        const size_t b = r / (2 * H);
        const float4 a = reinterpret_cast<const float4*>(up1)[i];
        const float4 l = reinterpret_cast<const float4*>(low)[((b * H + (y >> 1)) * W + (x >> 1)) * C4 + c];
        const float4 o = make_float4(a.x + l.x, a.y + l.y, a.z + l.z, a.w + l.w);
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
        reinterpret_cast<float4*>(out)[i] = o;
    }
    if (out_amax) cp_amax_commit(out_amax, amax);
}

inline int check() { return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH; }

}  // namespace

int cp_launch_nchw_to_nhwc(const float* in, float* out, int B, int C, int H, int W, int Cpad, hipStream_t s) {
    const int g = grid_for((size_t)B * H * W, 256 * 64);
    if (Cpad == 4) hipLaunchKernelGGL(nchw_to_nhwc_kernel<4>, dim3(g), dim3(TPB), 0, s, in, out, B, C, H * W);
    else if (Cpad == 8) hipLaunchKernelGGL(nchw_to_nhwc_kernel<8>, dim3(g), dim3(TPB), 0, s, in, out, B, C, H * W);
    else if (Cpad == C)
        hipLaunchKernelGGL(nchw_to_nhwc_tiled_kernel, dim3((H * W + 31) / 32, (C + 31) / 32, B), dim3(TPB), 0, s, in, out,
                           C, H * W);
    else return CP_ERR_INVALID;
    return check();
}

int cp_launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int H, int W, int ldi, hipStream_t s) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((size_t)B * C * H * W)), dim3(TPB), 0, s, in, out, B, C,
                       H * W, ldi);
    return check();
}

int cp_launch_maxpool2(const float* in, float* out, int B, int H, int W, int C, hipStream_t s) {
    if (C % 4 || H % 2 || W % 2) return CP_ERR_INVALID;
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for((size_t)B * (H / 2) * (W / 2) * (C / 4))), dim3(TPB), 0, s, in,
                       out, B, H, W, C / 4);
    return check();
}

int cp_launch_upsample_add(const float* in, const float* w, const float* add, float* out, int B, int H, int W, int C,
                           int f, unsigned* out_amax, hipStream_t s) {
    if (C % 4 || f < 2 || (f & 1)) return CP_ERR_INVALID;
    if ((size_t)B * H * f * W * f * (C / 4) >= ((size_t)1 << 31)) return CP_ERR_INVALID;
    const size_t lds = (size_t)C * 4 * f * f * sizeof(float);
    if (lds > 64 * 1024) return CP_ERR_INVALID;
    hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for((size_t)B * H * f * W * f * (C / 4), 256 * 8)), dim3(TPB), lds,
                       s, in, w, add, out, B, H, W, C, f, out_amax);
    return check();
}

int cp_launch_add_relu_sum(const float* a, const float* b, const float* c, const float* d, float* out, size_t n,
                           unsigned* out_amax, hipStream_t s) {
    if (n % 4) return CP_ERR_INVALID;
    hipLaunchKernelGGL(add_n_kernel, dim3(grid_for(n / 4)), dim3(TPB), 0, s, (const float4*)a, (const float4*)b,
                       (const float4*)c, (const float4*)d, (float4*)out, n / 4, out_amax);
    return check();
}

int cp_launch_upsample2_nearest_add(const float* up1, const float* low, float* out, int B, int H, int W, int C,
                                    unsigned* out_amax, hipStream_t s) {
    if (C % 4) return CP_ERR_INVALID;
    hipLaunchKernelGGL(upsample2_nearest_add_kernel, dim3(grid_for((size_t)B * 4 * H * W * (C / 4))), dim3(TPB), 0, s, up1,
                       low, out, B, H, W, C / 4, out_amax);
    return check();
}

int cp_launch_gru_gate(const float* x3, const float* h3, const float* hprev, float* hout, size_t M, unsigned* out_amax,
                       hipStream_t s) {
    hipLaunchKernelGGL(gru_gate_kernel, dim3(grid_for(M * 16)), dim3(TPB), 0, s, x3, h3, hprev, hout, M, out_amax);
    return check();
}

int cp_launch_groupnorm_relu(float* x, const float* gamma, const float* beta, double* stats_ws, int B, int HW, int C,
                             int groups, float eps, unsigned* out_amax, hipStream_t s) {
    if (C % 4 || (C / groups) % 4 || groups > 64 || TPB % (C / 4)) return CP_ERR_INVALID;
    if (hipMemsetAsync(stats_ws, 0, sizeof(double) * 2 * groups * B, s) != hipSuccess) return CP_ERR_LAUNCH;
    const int slab = 512;
    hipLaunchKernelGGL(gn_stats_kernel, dim3((HW + slab - 1) / slab, B), dim3(TPB), 0, s, x, stats_ws, HW, C, groups,
                       slab);
    hipLaunchKernelGGL(gn_apply_relu_kernel, dim3(grid_for((size_t)B * HW * (C / 4))), dim3(TPB), 0, s, x, gamma, beta,
                       stats_ws, B, HW, C, groups, eps, out_amax);
    return check();
}

int cp_launch_pack_weight(const float* w, float* wp, int Cout, int Cin, int taps, int CinP, int CoutPad, int coff,
                          hipStream_t s) {
    hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for((size_t)Cout * Cin * taps)), dim3(TPB), 0, s, w, wp, Cout, Cin,
                       taps, CinP, CoutPad, coff);
    return check();
}

int cp_launch_gn_affine(const double* stats, const float* gamma, const float* beta, float* a, float* d, int B, int C,
                        int groups, double count, float eps, const unsigned* x_amax, unsigned* y_amax, hipStream_t s) {
    if (C % groups) return CP_ERR_INVALID;
    hipLaunchKernelGGL(gn_affine_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, stats, gamma, beta, a, d, B, C, groups,
                       count, eps, x_amax, y_amax);
    return check();
}

// x [B * HW][C] NHWC hidden tensor, ga / gd [B][C] (cp_launch_gn_affine), wp [C][wld] (ConvW::wp of the 1x1), out NCHW
// [B][N][HW].  Requirements (else CP_ERR_INVALID, the caller keeps the matrix-core path): C % 64 == 0, HW % 64 == 0,
// 1 <= N <= 16.
int cp_launch_gn_final(const float* x, const float* ga, const float* gd, const float* wp, const float* bias, float* out,
                       int B, int HW, int C, int N, int wld, int sigmoid, hipStream_t s) {
    if (C % 64 != 0 || HW % 64 != 0 || N < 1 || N > 16 || C > 256) return CP_ERR_INVALID;
    const size_t M = (size_t)B * HW;
    if (M % 256 != 0) return CP_ERR_INVALID;
    const int nq = (N + 3) / 4 == 3 ? 4 : (N + 3) / 4;
    const dim3 grid((unsigned)(M / 256)), block(256);
    const size_t lds = (size_t)C * 4 * nq * sizeof(float);
    if (nq == 1) hipLaunchKernelGGL(gn_final_kernel<1>, grid, block, lds, s, x, ga, gd, wp, bias, out, C, N, wld, HW, sigmoid);
    else if (nq == 2) hipLaunchKernelGGL(gn_final_kernel<2>, grid, block, lds, s, x, ga, gd, wp, bias, out, C, N, wld, HW, sigmoid);
    else hipLaunchKernelGGL(gn_final_kernel<4>, grid, block, lds, s, x, ga, gd, wp, bias, out, C, N, wld, HW, sigmoid);
    return check();
}

int cp_launch_gn_finalize(const double* stats, float* mr, int n, double count, float eps, hipStream_t s) {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, s, stats, mr, n, count, eps);
    return check();
}

int cp_launch_preprocess(const unsigned char* img, int B, int H, int W, const double* trans6, const float* mean3,
                         const float* std3, float* out, int OH, int OW, hipStream_t s) {
    PreParams pp;
    // cv::invertAffineTransform (float64): warpAffine without WARP_INVERSE_MAP inverts the forward matrix first
    const double* M = trans6;
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1.0 / D : 0.0;
    const double A11 = M[4] * D, A22 = M[0] * D, A12 = -M[1] * D, A21 = -M[3] * D;
    pp.m[0] = A11; pp.m[1] = A12; pp.m[2] = -A11 * M[2] - A12 * M[5];
    pp.m[3] = A21; pp.m[4] = A22; pp.m[5] = -A21 * M[2] - A22 * M[5];
    for (int i = 0; i < 3; ++i) {
        pp.mean[i] = (double)mean3[i];
        pp.std[i] = (double)std3[i];
    }
    hipLaunchKernelGGL(preprocess_kernel, dim3((OH * OW + 255) / 256, B), dim3(256), 0, s, img, H, W, out, OH, OW, pp);
    return check();
}

int cp_launch_resize_u8(const unsigned char* img, int H, int W, int C, unsigned char* out, int OH, int OW, hipStream_t s) {
    hipLaunchKernelGGL(resize_u8_kernel, dim3((OH * OW + 255) / 256), dim3(256), 0, s, img, H, W, C, out, OH, OW);
    return check();
}
