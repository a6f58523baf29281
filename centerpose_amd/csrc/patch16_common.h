// Epilogue shared by the kernels whose M tile is an 8 x 16 patch of output pixels of one image (halo16.hip, dcn16p.hip):
// 4 waves, fragment i of wave wm = tile rows m0 .. m0 + 31 = patch rows (m0 / 16), (m0 / 16) + 1.
#pragma once
#include "igemm16_common.h"

namespace {

constexpr int PATCH_TH = 8, PATCH_TW = 16;

// y = acc * scale[n] * 2^-e_a + shift[n] (+ residual) -> activation -> NHWC store (full patches only); GroupNorm
// statistics and |max| tracking as igemm_epilogue.  (b, ty0, tx0): image and top-left output pixel of the patch.
// PERM: fragment row m of a wave is not pixel (m / 16, m % 16) of its two patch rows but patch_perm_row / patch_perm_col
// below (dcn16p.hip: every 16-lane group of a ds_read_b128 then covers 16 consecutive pixels of ONE row).
// Lanes 4 qd .. 4 qd + 3 (qd = m / 4): row [0,1,1,0,1,0,0,1][qd], columns 4 (qd / 2) .. + 3.
__device__ __forceinline__ int patch_perm_row(int m) { return (0x96 >> (m >> 2)) & 1; }
__device__ __forceinline__ int patch_perm_col(int m) { return 4 * (m >> 3) + (m & 3); }

template <int MT, int NT, int WM, int WN, bool PERM = false>
__device__ __forceinline__ void patch_epilogue(const ConvParams& p, Frag<32>::acc_t (&acc)[MT][NT], int b, int ty0, int tx0,
                                               int tn, int wm, int wn, int lane, float ainv) {
    typedef Frag<32> F;
    constexpr int TW = PATCH_TW;
    constexpr int BN = 32 * NT * WN;
    const int lcol = lane & 31;
    const int act = p.act;
    const bool has_res = p.res != nullptr, has_gn = p.gn_stats != nullptr;
    float amax = 0.f;
    const int h4 = lane >> 5;  // fragment rows (r & 3) + 8 (r >> 2) + 4 h4
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = tn * BN + wn * (NT * 32) + j * 32 + lcol;
        const float sc = (p.scale ? p.scale[n] : 1.f) * ainv;
        const float sh = p.shift ? p.shift[n] : 0.f;
        const bool n_ok = n < p.Cout;
        const bool sig_lane = act == CP_ACT_SIGMOID || (act == CP_ACT_SIGMOID_FROM && n >= p.act_from);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            // fragment = tile rows m0 .. m0 + 31 = patch rows y, y + 1 (16 pixels each)
            const int m0 = wm * (MT * 32) + i * 32;
            const int pix0 = __builtin_amdgcn_readfirstlane((b * p.H + ty0 + (m0 >> 4)) * p.W + tx0);
            float v[F::NACC];
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) v[r] = acc[i][j][r] * sc + sh;
            if (has_res) {
                const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res + (size_t)pix0 * p.res_ld, (unsigned)((p.W + TW) * p.res_ld) * 4u);
                // accumulator r of lane-half h4 = fragment row (r & 3) + 8 (r >> 2) + 4 h4; PERM: patch row h4 ^ [0,1,1,0][r >> 2],
                // column 4 (r >> 2) + (r & 3)
                const unsigned vr = n_ok ? (unsigned)((PERM ? h4 * p.W : 4 * h4) * p.res_ld + n) * 4u : 0x80000000u;
                const unsigned vr1 = n_ok ? (unsigned)((1 - h4) * p.W * p.res_ld + n) * 4u : 0x80000000u;
#pragma unroll
                for (int r = 0; r < F::NACC; ++r) {
                    const int so = PERM ? (4 * (r >> 2) + (r & 3)) * p.res_ld * 4
                                        : (((r >> 3) * p.W) + 8 * ((r >> 2) & 1) + (r & 3)) * p.res_ld * 4;
                    const bool flip = PERM && ((0x6 >> (r >> 2)) & 1);
                    v[r] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, (int)(flip ? vr1 : vr), so, 0));
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
                float s1 = 0.f, s2 = 0.f;
                if (n_ok) {
#pragma unroll
                    for (int r = 0; r < F::NACC; ++r) { s1 += v[r]; s2 += v[r] * v[r]; }
                }
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if ((lane & 7) == 0 && lane < 32 && n_ok) {
                    double* st = p.gn_stats + ((size_t)b * p.gn_groups + n / p.gn_cpg) * 2;
                    atomicAdd(st, (double)s1);
                    atomicAdd(st + 1, (double)s2);
                }
            }
            float* frag_out = p.out + (size_t)pix0 * p.ldo + p.coff;
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(frag_out, (unsigned)((p.W + TW) * p.ldo) * 4u);
            const unsigned vo = n_ok ? (unsigned)((PERM ? h4 * p.W : 4 * h4) * p.ldo + n) * 4u : 0x80000000u;
            const unsigned vo1 = n_ok ? (unsigned)((1 - h4) * p.W * p.ldo + n) * 4u : 0x80000000u;
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) {
                const int so = PERM ? (4 * (r >> 2) + (r & 3)) * p.ldo * 4
                                    : (((r >> 3) * p.W) + 8 * ((r >> 2) & 1) + (r & 3)) * p.ldo * 4;
                const bool flip = PERM && ((0x6 >> (r >> 2)) & 1);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), ro, (int)(flip ? vo1 : vo), so, 0);
            }
        }
    }
    if (p.out_amax) cp_amax_commit(p.out_amax, amax);
}

}  // namespace
