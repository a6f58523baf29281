// Implicit-GEMM convolution and fused DCNv2 on the gfx950 matrix cores, float32 in / float32
// accumulate (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: exact f32, 157 TFLOP/s peak).
//
//   GEMM view   M = B*Ho*Wo output pixels,  N = Cout,  K = KH*KW*Cin  (ci fastest inside a tap)
//   A [M x K]   never exists in memory: each workgroup builds its BM x 16 slice in LDS per K-step,
//               either by shifted NHWC reads (plain conv, any kernel size / stride / virtual concat)
//               or by the modulated bilinear gather of DCNv2 (reference algorithm:
//               DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:25-54,125-195) — so the reference's `columns`
//               buffer (9x the input, dcn_v2_cuda.cu:90) and its two batched SGEMMs (:123-163) become
//               one kernel whose only HBM traffic is input + offsets + weights + output.
//   B [K x N]   weights pre-packed at load time as [tap][ci][co] (co contiguous).
//   epilogue    y = acc*scale[n] + shift[n] (+ residual) -> ReLU / sigmoid -> NHWC or NCHW store
//               (folded eval-mode BatchNorm, conv bias, BasicBlock residual: pose_dla_dcn.py:48-62,
//               DeformConv.actf :380-389).
//
// Tiling is for 64-wide wavefronts: 4 waves per workgroup, each wave owns MT x NT MFMA fragments;
// the A slice is stored k-major in LDS (row stride padded so both the transposing ds_write_b32 and
// the per-lane ds_read_b32 of the MFMA operand are bank-conflict free); global->register prefetch of
// tile t+1 overlaps the MFMA block of tile t; two LDS buffers, one barrier per K-step.
#include "cp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 16;
constexpr int NTHREADS = 256;

template <int FRAG> struct Frag;
template <> struct Frag<32> {
    typedef f32x16 acc_t;
    static constexpr int NACC = 16, KSTEP = 2;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    static __device__ __forceinline__ int row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};
template <> struct Frag<16> {
    typedef f32x4 acc_t;
    static constexpr int NACC = 4, KSTEP = 4;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + r
    static __device__ __forceinline__ int row(int r, int lane) { return (lane >> 4) * 4 + r; }
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Reference bilinear blend, evaluated in the reference's operation order without fused
// multiply-adds (dcn_v2_im2col_cuda.cu:47-52: w1*v1 + w2*v2 + w3*v3 + w4*v4, then * mask :190).
__device__ __forceinline__ float blend(float w1, float v1, float w2, float v2, float w3, float v3, float w4, float v4,
                                       float mk) {
    float s = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w1, v1), __fmul_rn(w2, v2)), __fmul_rn(w3, v3)),
                        __fmul_rn(w4, v4));
    return __fmul_rn(s, mk);
}

template <int FRAG, int MT, int NT, int WM, int WN, bool DCN>
__global__ __launch_bounds__(NTHREADS) void igemm_kernel(const ConvParams p, const int tiles_m, const int tiles_n) {
    typedef Frag<FRAG> F;
    typedef typename F::acc_t acc_t;
    constexpr int BM = FRAG * MT * WM, BN = FRAG * NT * WN;
    static_assert(WM * WN * 64 == NTHREADS, "4 waves");
    constexpr int LDA = BM + (FRAG == 16 ? 18 : 2);
    constexpr int LDB = BN + (FRAG == 16 ? 0 : 4);
    constexpr int A_SLOTS = BM * BK / 4 / NTHREADS;  // float4 per thread per K-step
    constexpr int B_F4 = BK * BN / 4;
    constexpr int B_SLOTS = (B_F4 + NTHREADS - 1) / NTHREADS;
    constexpr int A_SZ = BK * LDA, B_SZ = BK * LDB;
    __shared__ __attribute__((aligned(16))) float lds[2 * A_SZ + 2 * B_SZ];
    float* As = lds;
    float* Bs = lds + 2 * A_SZ;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;

    // XCD-aware, bijective block -> tile map: block b runs on XCD b % 8; give every XCD a contiguous
    // run of tiles (n fastest) so neighbouring tiles share their input halo / weights in that XCD's L2.
    int tile;
    {
        const int nt = tiles_m * tiles_n, bid = blockIdx.x;
        const int q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tn = tile % tiles_n, tm = tile / tiles_n;

    const int M = p.B * p.Ho * p.Wo;
    const int HWo = p.Ho * p.Wo;

    // ---- per-thread A-slot geometry (fixed over the K loop) ----
    const int k4 = tid & 3;
    int a_b[A_SLOTS], a_h0[A_SLOTS], a_w0[A_SLOTS];
    bool a_ok[A_SLOTS];
#pragma unroll
    for (int j = 0; j < A_SLOTS; ++j) {
        const int m = tm * BM + (tid >> 2) + j * 64;
        a_ok[j] = m < M;
        const int mm = a_ok[j] ? m : 0;
        const int b = mm / HWo, rem = mm - b * HWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        a_b[j] = b;
        a_h0[j] = ho * p.stride - p.pad;
        a_w0[j] = wo * p.stride - p.pad;
    }

    float4 a_reg[A_SLOTS];
    float4 b_reg[B_SLOTS];

    auto load_tile = [&](int kt) {
        // ---- B (weights): BK rows of BN floats ----
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) {
            const int f = tid + j * NTHREADS;
            if (B_F4 % NTHREADS == 0 || f < B_F4) {
                const int row = f / (BN / 4), c4 = f % (BN / 4);
                b_reg[j] = ld4(p.wp + (size_t)(kt * BK + row) * p.CoutPad + tn * BN + c4 * 4);
            }
        }
        // ---- A ----
        const int kk = kt * BK + k4 * 4;
        const int tap = kk / p.Cin;
        const int ci = kk - tap * p.Cin;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const bool kvalid = kk < p.K;
        if (!DCN) {
            const float* base = p.src[0];
            int sc = p.src_c[0], c = ci;
            if (p.nsrc > 1 && c >= sc) {
                c -= sc; base = p.src[1]; sc = p.src_c[1];
                if (p.nsrc > 2 && c >= sc) {
                    c -= sc; base = p.src[2]; sc = p.src_c[2];
                    if (p.nsrc > 3 && c >= sc) { c -= sc; base = p.src[3]; sc = p.src_c[3]; }
                }
            }
#pragma unroll
            for (int j = 0; j < A_SLOTS; ++j) {
                const int hi = a_h0[j] + kh, wi = a_w0[j] + kw;
                const bool ok = a_ok[j] && kvalid && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
                a_reg[j] = ok ? ld4(base + ((size_t)(a_b[j] * p.H + hi) * p.W + wi) * sc + c) : zero4();
            }
        } else {
            const float* base = p.src[0];
            const int C = p.Cin;
#pragma unroll
            for (int j = 0; j < A_SLOTS; ++j) {
                float4 v = zero4();
                if (a_ok[j]) {
                    // (ho, wo) = (a_h0 + pad, a_w0 + pad) since stride == 1
                    const size_t pix = (size_t)(a_b[j] * p.H + (a_h0[j] + p.pad)) * p.W + (a_w0[j] + p.pad);
                    const float* om = p.offmask + pix * 32;
                    const float dh = om[2 * tap], dw = om[2 * tap + 1], mk = om[18 + tap];
                    const float h_im = (float)(a_h0[j] + kh) + dh;
                    const float w_im = (float)(a_w0[j] + kw) + dw;
                    if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
                        const int h_lo = (int)floorf(h_im), w_lo = (int)floorf(w_im);
                        const int h_hi = h_lo + 1, w_hi = w_lo + 1;
                        const float lh = h_im - (float)h_lo, lw = w_im - (float)w_lo;
                        const float hh = 1.f - lh, hw = 1.f - lw;
                        const float* img = base + (size_t)a_b[j] * p.H * p.W * C + ci;
                        float4 v1 = zero4(), v2 = zero4(), v3 = zero4(), v4 = zero4();
                        if (h_lo >= 0 && w_lo >= 0) v1 = ld4(img + ((size_t)h_lo * p.W + w_lo) * C);
                        if (h_lo >= 0 && w_hi <= p.W - 1) v2 = ld4(img + ((size_t)h_lo * p.W + w_hi) * C);
                        if (h_hi <= p.H - 1 && w_lo >= 0) v3 = ld4(img + ((size_t)h_hi * p.W + w_lo) * C);
                        if (h_hi <= p.H - 1 && w_hi <= p.W - 1) v4 = ld4(img + ((size_t)h_hi * p.W + w_hi) * C);
                        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                        v.x = blend(w1, v1.x, w2, v2.x, w3, v3.x, w4, v4.x, mk);
                        v.y = blend(w1, v1.y, w2, v2.y, w3, v3.y, w4, v4.y, mk);
                        v.z = blend(w1, v1.z, w2, v2.z, w3, v3.z, w4, v4.z, mk);
                        v.w = blend(w1, v1.w, w2, v2.w, w3, v3.w, w4, v4.w, mk);
                    }
                }
                a_reg[j] = v;
            }
        }
    };

    auto store_tile = [&](int buf) {
        float* A = As + buf * A_SZ;
        float* Bt = Bs + buf * B_SZ;
#pragma unroll
        for (int j = 0; j < A_SLOTS; ++j) {
            const int ml = (tid >> 2) + j * 64;
            float* d = A + (k4 * 4) * LDA + ml;
            d[0] = a_reg[j].x;
            d[LDA] = a_reg[j].y;
            d[2 * LDA] = a_reg[j].z;
            d[3 * LDA] = a_reg[j].w;
        }
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) {
            const int f = tid + j * NTHREADS;
            if (B_F4 % NTHREADS == 0 || f < B_F4) {
                const int row = f / (BN / 4), c4 = f % (BN / 4);
                *reinterpret_cast<float4*>(Bt + row * LDB + c4 * 4) = b_reg[j];
            }
        }
    };

    acc_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) acc[i][j][r] = 0.f;

    const int nk = p.Kpad / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int lrow = lane / FRAG;  // k index inside a KSTEP
    const int lcol = lane % FRAG;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const float* A = As + buf * A_SZ + wm * (MT * FRAG) + lcol;
        const float* Bt = Bs + buf * B_SZ + wn * (NT * FRAG) + lcol;
#pragma unroll
        for (int ks = 0; ks < BK / F::KSTEP; ++ks) {
            const int krow = ks * F::KSTEP + lrow;
            float a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = A[krow * LDA + i * FRAG];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = Bt[krow * LDB + j * FRAG];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = F::mfma(a[i], b[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = tn * BN + wn * (NT * FRAG) + j * FRAG + lcol;
        const float sc = p.scale ? p.scale[n] : 1.f;
        const float sh = p.shift ? p.shift[n] : 0.f;
        const bool n_ok = n < p.Cout;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mbase = tm * BM + wm * (MT * FRAG) + i * FRAG;
            float v[F::NACC];
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) {
                const int m = mbase + F::row(r, lane);
                float y = acc[i][j][r] * sc + sh;
                if (p.res && n_ok && m < M) y += p.res[(size_t)m * p.res_ld + n];
                if (p.act == CP_ACT_RELU) y = fmaxf(y, 0.f);
                else if (p.act == CP_ACT_SIGMOID || (p.act == CP_ACT_SIGMOID_FROM && n >= p.act_from))
                    y = 1.f / (1.f + expf(-y));
                v[r] = y;
            }
            if (!n_ok) continue;
            if (p.store == CP_STORE_NHWC) {
#pragma unroll
                for (int r = 0; r < F::NACC; ++r) {
                    const int m = mbase + F::row(r, lane);
                    if (m < M) p.out[(size_t)m * p.ldo + p.coff + n] = v[r];
                }
            } else {
                // NCHW: rows r..r+3 of one register quad are 4 consecutive pixels
#pragma unroll
                for (int r4 = 0; r4 < F::NACC; r4 += 4) {
                    const int m = mbase + F::row(r4, lane);
                    if (m >= M) continue;
                    const int b = m / HWo, pix = m - b * HWo;
                    float* o = p.out + ((size_t)b * p.ldo + p.coff + n) * HWo + pix;
                    if ((HWo & 3) == 0 && m + 3 < M) {
                        *reinterpret_cast<float4*>(o) = make_float4(v[r4], v[r4 + 1], v[r4 + 2], v[r4 + 3]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int mq = m + q;
                            if (mq < M) {
                                const int bq = mq / HWo, pq = mq - bq * HWo;
                                p.out[((size_t)bq * p.ldo + p.coff + n) * HWo + pq] = v[r4 + q];
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int FRAG, int MT, int NT, int WM, int WN, bool DCN>
int launch(const ConvParams& p, hipStream_t stream) {
    constexpr int BM = FRAG * MT * WM, BN = FRAG * NT * WN;
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_m = (M + BM - 1) / BM;
    const int tiles_n = p.CoutPad / BN;
    if (p.CoutPad % BN != 0 || p.Kpad % BK != 0) return CP_ERR_INVALID;
    hipLaunchKernelGGL((igemm_kernel<FRAG, MT, NT, WM, WN, DCN>), dim3(tiles_m * tiles_n), dim3(NTHREADS), 0, stream,
                       p, tiles_m, tiles_n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

}  // namespace

int cp_conv_tile_n(int cout) {
    if (cout <= 16) return 16;
    if (cout <= 32) return 32;
    if (cout % 128 == 0) return 128;
    return 64;
}

// 0..3: plain conv with N tile 16/32/64/128; 4..5: fused DCNv2 with N tile 64/128
int cp_conv_variant(const ConvParams& p) {
    const int bn = cp_conv_tile_n(p.Cout);
    if (p.offmask) return bn == 128 ? 5 : 4;
    return bn == 16 ? 0 : bn == 32 ? 1 : bn == 64 ? 2 : 3;
}

const char* cp_conv_variant_name(int v) {
    static const char* names[6] = {"igemm_f32_16x16x4_m256n16", "igemm_f32_32x32x2_m256n32", "igemm_f32_32x32x2_m128n64",
                                   "igemm_f32_32x32x2_m128n128", "dcn_igemm_f32_32x32x2_m128n64",
                                   "dcn_igemm_f32_32x32x2_m128n128"};
    return (v >= 0 && v < 6) ? names[v] : "?";
}

int cp_launch_conv(const ConvParams& p, hipStream_t stream) {
    if (p.nsrc < 1 || p.nsrc > CP_MAX_SRC || p.Cin % 4 != 0) return CP_ERR_INVALID;
    for (int s = 0; s < p.nsrc; ++s)
        if (p.src_c[s] % 4 != 0 || (p.nsrc > 1 && p.src_c[s] % BK != 0)) return CP_ERR_INVALID;
    const int bn = cp_conv_tile_n(p.Cout);
    if (p.offmask) {
        // DCN: 3x3, stride 1, pad 1, one source, Cin % 16 == 0 (a K-step never straddles a tap)
        if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.nsrc != 1 || p.Cin % BK != 0 ||
            p.H != p.Ho || p.W != p.Wo)
            return CP_ERR_INVALID;
        if (bn == 128) return launch<32, 2, 2, 2, 2, true>(p, stream);
        if (bn == 64) return launch<32, 2, 1, 2, 2, true>(p, stream);
        return CP_ERR_INVALID;
    }
    switch (bn) {
        case 16: return launch<16, 4, 1, 4, 1, false>(p, stream);
        case 32: return launch<32, 2, 1, 4, 1, false>(p, stream);
        case 64: return launch<32, 2, 1, 2, 2, false>(p, stream);
        default: return launch<32, 2, 2, 2, 2, false>(p, stream);
    }
}
