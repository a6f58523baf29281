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

// Epilogue of every implicit-GEMM tile: y = acc*scale[n] + shift[n] (+ residual) -> ReLU / sigmoid -> NHWC or NCHW
// store (folded eval-mode BatchNorm, conv bias, BasicBlock residual: pose_dla_dcn.py:48-62, DeformConv.actf :380-389).
template <int FRAG, int MT, int NT, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const ConvParams& p, typename Frag<FRAG>::acc_t (&acc)[MT][NT], int tm,
                                               int tn, int wm, int wn, int lane) {
    typedef Frag<FRAG> F;
    constexpr int BM = FRAG * MT * WM, BN = FRAG * NT * WN;
    const int M = p.B * p.Ho * p.Wo;
    const int HWo = p.Ho * p.Wo;
    const int lcol = lane % FRAG;
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
            if (p.gn_stats) {
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
