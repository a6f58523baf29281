// Pieces shared by the implicit-GEMM kernels (igemm.hip: exact-f32 MFMA; igemm16.hip: split-f16 MFMA).
#pragma once
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

// XCD-aware, bijective block -> tile map: block b runs on XCD b % 8; give every XCD a contiguous run of tiles
// (n fastest) so neighbouring tiles share their input halo / weights in that XCD's L2.
__device__ __forceinline__ int tile_of_block(int tiles_m, int tiles_n) {
    const int nt = tiles_m * tiles_n, bid = blockIdx.x;
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Output-pixel index m -> (image b, row ho, column wo).  For m < 2^24 the quotient by float reciprocal is off by at
// most two (|error| <= q * 1.5 * 2^-23), so two correction steps make it exact: ~12 VALU per division instead of the
// ~35 of a 32-bit integer division (four slots x two divisions per thread were a fifth of the prologue).  `big`
// (wave-uniform: the tensor has >= 2^24 output pixels) selects the integer division.
__device__ __forceinline__ int div_small(int m, int d, float rcp_d, int* rem, bool big) {
    if (big) {
        const int q = m / d;
        *rem = m - q * d;
        return q;
    }
    int q = (int)((float)m * rcp_d);
    int r = m - q * d;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        if (r < 0) { --q; r += d; }
        else if (r >= d) { ++q; r -= d; }
    }
    *rem = r;
    return q;
}

// Bit t = kh*KW + kw of the result: tap (kh, kw) of an output pixel whose window starts at (h0, w0) reads inside the
// image.  Row and column validity are separable, so the KH*KW-deep double loop becomes KH + KW steps.
__device__ __forceinline__ unsigned tap_valid_mask(int h0, int w0, int H, int W, int KH, int KW) {
    if (KH == 3 && KW == 3) {  // wave-uniform fast path, no loops: the 3x3 layers are 80 % of the launches
        const unsigned wb = ((unsigned)w0 < (unsigned)W ? 1u : 0u) | ((unsigned)(w0 + 1) < (unsigned)W ? 2u : 0u) |
                            ((unsigned)(w0 + 2) < (unsigned)W ? 4u : 0u);
        const unsigned hb = ((unsigned)h0 < (unsigned)H ? 1u : 0u) | ((unsigned)(h0 + 1) < (unsigned)H ? 8u : 0u) |
                            ((unsigned)(h0 + 2) < (unsigned)H ? 64u : 0u);
        return wb * hb;  // bit (3*kh + kw)
    }
    if (KH == 1 && KW == 1) return ((unsigned)h0 < (unsigned)H && (unsigned)w0 < (unsigned)W) ? 1u : 0u;
    unsigned wbits = 0u;
    for (int kw = 0; kw < KW; ++kw)
        if ((unsigned)(w0 + kw) < (unsigned)W) wbits |= 1u << kw;
    unsigned vm = 0u;
    for (int kh = 0; kh < KH; ++kh)
        if ((unsigned)(h0 + kh) < (unsigned)H) vm |= wbits << (kh * KW);
    return vm;
}

// m -> (b, ho, wo) for all the slots of a thread: shifts when Ho*Wo and Wo are powers of two (every DLA-34 level at
// 512x512 and most other input sizes), else the float-reciprocal division above.  `sh_hw` / `sh_w` are -1 when the
// corresponding extent is not a power of two (wave-uniform).
struct PixelDecomp {
    int HWo, Wo, sh_hw, sh_w;
    float rcp_hwo, rcp_wo;
    bool big;
    __device__ __forceinline__ void init(int Ho, int Wo_, int M) {
        HWo = Ho * Wo_;
        Wo = Wo_;
        sh_hw = (HWo & (HWo - 1)) == 0 ? __builtin_ctz(HWo) : -1;
        sh_w = (Wo & (Wo - 1)) == 0 ? __builtin_ctz(Wo) : -1;
        rcp_hwo = 1.f / (float)HWo;
        rcp_wo = 1.f / (float)Wo;
        big = M >= (1 << 24);
    }
    __device__ __forceinline__ void split(int m, int* b, int* ho, int* wo) const {
        if (sh_hw >= 0 && sh_w >= 0) {
            *b = m >> sh_hw;
            const int rem = m & (HWo - 1);
            *ho = rem >> sh_w;
            *wo = rem & (Wo - 1);
            return;
        }
        int rem;
        *b = div_small(m, HWo, rcp_hwo, &rem, big);
        *ho = div_small(rem, Wo, rcp_wo, wo, big);
    }
};

// Raw buffer resource (SRD) over `bytes` bytes at p: loads beyond it return 0, stores beyond it are dropped.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// Activation pre-scale of an f16x3 launch (ConvParams::in_amax): 2^e_a and its inverse from the largest running |max|
// of the sources; (1, 1) when the launch carries no slots (exact-f32 kernels, stand-alone calls without tracking).
__device__ __forceinline__ void conv_in_scale(const ConvParams& p, float* fwd, float* inv) {
    *fwd = 1.f;
    *inv = 1.f;
    if (!p.in_amax[0]) return;
    unsigned mx = cp_amax_read(p.in_amax[0]);
    if (p.nsrc > 1 && p.in_amax[1]) mx = max(mx, cp_amax_read(p.in_amax[1]));
    if (p.nsrc > 2 && p.in_amax[2]) mx = max(mx, cp_amax_read(p.in_amax[2]));
    if (p.nsrc > 3 && p.in_amax[3]) mx = max(mx, cp_amax_read(p.in_amax[3]));
    cp_amax_to_scale(mx, fwd, inv);
}

// The same in two halves, so that a kernel can put its first global loads between them: the 32 scalar loads of the
// sub-slots are ISSUED by conv_in_scale_issue and only waited for / reduced by conv_in_scale_finish (the wait sits at the
// first use of the loaded values).  Called back to back they are conv_in_scale; with the first tile's loads in between, the
// scale's round trip (~1 us, once per block) overlaps theirs instead of preceding it.
struct AmaxRaw {
    unsigned v[CP_AMAX_SUB];  // the first source's sub-slots, un-reduced (uniform: scalar registers)
    unsigned extra;           // further sources of a virtual concat, already reduced (rare: Root nodes)
};
__device__ __forceinline__ AmaxRaw conv_in_scale_issue(const ConvParams& p) {
    AmaxRaw r;
    r.extra = 0u;
#pragma unroll
    for (int k = 0; k < CP_AMAX_SUB; ++k) r.v[k] = 0u;
    if (!p.in_amax[0]) return r;
#pragma unroll
    for (int k = 0; k < CP_AMAX_SUB; ++k) r.v[k] = p.in_amax[0][k * CP_AMAX_STRIDE];
    if (p.nsrc > 1 && p.in_amax[1]) r.extra = max(r.extra, cp_amax_read(p.in_amax[1]));
    if (p.nsrc > 2 && p.in_amax[2]) r.extra = max(r.extra, cp_amax_read(p.in_amax[2]));
    if (p.nsrc > 3 && p.in_amax[3]) r.extra = max(r.extra, cp_amax_read(p.in_amax[3]));
    return r;
}
__device__ __forceinline__ void conv_in_scale_finish(const ConvParams& p, const AmaxRaw& r, float* fwd, float* inv) {
    *fwd = 1.f;
    *inv = 1.f;
    if (!p.in_amax[0]) return;
    unsigned m = r.extra;
#pragma unroll
    for (int k = 0; k < CP_AMAX_SUB; ++k) m = max(m, r.v[k]);
    cp_amax_to_scale(m, fwd, inv);
}

// Epilogue of every implicit-GEMM tile: y = acc*scale[n] + shift[n] (+ residual) -> ReLU / sigmoid -> NHWC or NCHW
// store (folded eval-mode BatchNorm, conv bias, BasicBlock residual: pose_dla_dcn.py:48-62, DeformConv.actf :380-389).
template <int FRAG, int MT, int NT, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const ConvParams& p, typename Frag<FRAG>::acc_t (&acc)[MT][NT], int tm,
                                               int tn, int wm, int wn, int lane, float ainv = 1.f) {
    typedef Frag<FRAG> F;
    constexpr int BM = FRAG * MT * WM, BN = FRAG * NT * WN;
    const int M = p.B * p.Ho * p.Wo;
    const int HWo = p.Ho * p.Wo;
    const int lcol = lane % FRAG;
    // every mode decision below is wave-uniform and taken once per fragment, outside the per-element loops (as
    // per-element branches the epilogue was ~2000 instructions per wave -- more than ten K-steps of the main loop)
    const int act = p.act;
    const bool has_res = p.res != nullptr, has_gn = p.gn_stats != nullptr, nhwc = p.store == CP_STORE_NHWC;
    float amax = 0.f;  // running max|y| of this lane's outputs (ConvParams::out_amax)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = tn * BN + wn * (NT * FRAG) + j * FRAG + lcol;
        // ainv = 2^-e_a of the activation pre-scale (1 for the exact-f32 kernels); scale already carries 2^-e_w
        const float sc = (p.scale ? p.scale[n] : 1.f) * ainv;
        const float sh = p.shift ? p.shift[n] : 0.f;
        const bool n_ok = n < p.Cout;
        const bool sig_lane = act == CP_ACT_SIGMOID || (act == CP_ACT_SIGMOID_FROM && n >= p.act_from);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            // wave-uniform by construction (wm comes from threadIdx / 64): tell the compiler, or every buffer access below
            // is wrapped in a waterfall loop over a 'divergent' resource
            const int mbase = __builtin_amdgcn_readfirstlane(tm * BM + wm * (MT * FRAG) + i * FRAG);
            const bool full = mbase + FRAG <= M;  // every fragment but the ragged last ones
            float v[F::NACC];
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) v[r] = acc[i][j][r] * sc + sh;
            if (has_res) {
                if (full) {
                    // residual rows via buffer loads relative to the fragment's first row: per-lane offset once, the
                    // row step rides in the scalar offset
                    const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res + (size_t)mbase * p.res_ld, (unsigned)(FRAG * p.res_ld) * 4u);
                    const unsigned vr = n_ok ? (unsigned)(F::row(0, lane) * p.res_ld + n) * 4u : 0x80000000u;
                    float rv[F::NACC];
#pragma unroll
                    for (int r = 0; r < F::NACC; ++r)
                        rv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, (int)vr, F::row(r, 0) * p.res_ld * 4, 0));
#pragma unroll
                    for (int r = 0; r < F::NACC; ++r) v[r] += rv[r];
                } else {
#pragma unroll
                    for (int r = 0; r < F::NACC; ++r) {
                        const int m = mbase + F::row(r, lane);
                        if (n_ok && m < M) v[r] += p.res[(size_t)m * p.res_ld + n];
                    }
                }
            }
            if (act == CP_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < F::NACC; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (act == CP_ACT_SIGMOID || act == CP_ACT_SIGMOID_FROM) {
#pragma unroll
                for (int r = 0; r < F::NACC; ++r) v[r] = sig_lane ? 1.f / (1.f + expf(-v[r])) : v[r];
            }
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) amax = fmaxf(amax, fabsf(v[r]));
            if (has_gn) {
                // GroupNorm statistics of this 32-row x 32-channel fragment: rows live in registers, the 8 channels
                // of a group in 8 neighbouring lanes (and the other 16 rows in lane ^ 32)
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < F::NACC; ++r) {
                    const int m = mbase + F::row(r, lane);
                    if (n_ok && m < M) { s1 += v[r]; s2 += v[r] * v[r]; }
                }
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                if (FRAG == 32) { s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64); }
                else { s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64); s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64); }
                if ((lane & 7) == 0 && lane < FRAG && n_ok && mbase < M) {
                    const int b = mbase / HWo;  // launcher guarantees HWo % FRAG == 0: one image per fragment
                    double* st = p.gn_stats + ((size_t)b * p.gn_groups + n / p.gn_cpg) * 2;
                    atomicAdd(st, (double)s1);
                    atomicAdd(st + 1, (double)s2);
                }
            }
            if (nhwc && full) {
                // buffer stores relative to the fragment's first row: no 64-bit address arithmetic and no bounds
                // compare per element
                float* frag_out = p.out + (size_t)mbase * p.ldo + p.coff;
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(frag_out, (unsigned)(FRAG * p.ldo) * 4u);
                // a masked lane stays out of range with or without the scalar offset added
                const unsigned vo = n_ok ? (unsigned)(F::row(0, lane) * p.ldo + n) * 4u : 0x80000000u;
#pragma unroll
                for (int r = 0; r < F::NACC; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), ro, (int)vo, F::row(r, 0) * p.ldo * 4, 0);
                continue;
            }
            if (!n_ok) continue;
            if (nhwc) {
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
    if (p.out_amax) cp_amax_commit(p.out_amax, amax);
}

// split-K: raw accumulators of this K slice -> partial[slice][m][CoutPad]
template <int FRAG, int MT, int NT, int WM, int WN>
__device__ __forceinline__ void igemm_store_partial(const ConvParams& p, typename Frag<FRAG>::acc_t (&acc)[MT][NT], int tm,
                                                    int tn, int wm, int wn, int lane, int slice) {
    typedef Frag<FRAG> F;
    constexpr int BM = FRAG * MT * WM, BN = FRAG * NT * WN;
    const int M = p.B * p.Ho * p.Wo;
    float* dst = p.partial + (size_t)slice * M * p.CoutPad;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = tn * BN + wn * (NT * FRAG) + j * FRAG + lane % FRAG;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mbase = tm * BM + wm * (MT * FRAG) + i * FRAG;
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) {
                const int m = mbase + F::row(r, lane);
                if (m < M) dst[(size_t)m * p.CoutPad + n] = acc[i][j][r];
            }
        }
    }
}

// K-step range of this block's slice
__device__ __forceinline__ void splitk_range(const ConvParams& p, int nk, int* kt0, int* kt1) {
    if (p.splitk <= 1) { *kt0 = 0; *kt1 = nk; return; }
    const int per = (nk + p.splitk - 1) / p.splitk;
    *kt0 = min(nk, (int)blockIdx.y * per);
    *kt1 = min(nk, *kt0 + per);
}

}  // namespace
