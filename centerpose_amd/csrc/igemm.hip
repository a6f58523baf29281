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
#include "igemm_common.h"

namespace {

// Reference bilinear blend, evaluated in the reference's operation order without fused
// multiply-adds (dcn_v2_im2col_cuda.cu:47-52: w1*v1 + w2*v2 + w3*v3 + w4*v4, then * mask :190).
__device__ __forceinline__ float blend(float w1, float v1, float w2, float v2, float w3, float v3, float w4, float v4,
                                       float mk) {
    float s = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w1, v1), __fmul_rn(w2, v2)), __fmul_rn(w3, v3)),
                        __fmul_rn(w4, v4));
    return __fmul_rn(s, mk);
}

// ALIGNED: Cin % 16 == 0 and every source's channel count % 16 == 0, so a K-step lies inside one tap and
// one concat source: the (tap, source, channel) walk is then wave-uniform scalar state advanced once per
// K-step, per-lane work is one add + one 64-bit mad + a predicated 16-byte load per slot, and tap
// validity comes from a per-slot bit-mask computed once.  (The unaligned variant serves the 7x7 stems.)
template <int FRAG, int MT, int NT, int WM, int WN, bool DCN, bool ALIGNED, bool MULTISRC>
__global__ __launch_bounds__(NTHREADS, 3) void igemm_kernel(const ConvParams p, const int tiles_m, const int tiles_n) {
    typedef Frag<FRAG> F;
    typedef typename F::acc_t acc_t;
    constexpr int BM = FRAG * MT * WM, BN = FRAG * NT * WN;
    static_assert(WM * WN * 64 == NTHREADS, "4 waves");
    static_assert(!DCN || ALIGNED, "DCN needs aligned channels");
    static_assert(!MULTISRC || (ALIGNED && !DCN), "virtual concat only on the aligned plain-conv path");
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
    const int tile = tile_of_block(tiles_m, tiles_n);
    const int tn = tile % tiles_n, tm = tile / tiles_n;

    const int M = p.B * p.Ho * p.Wo;

    // ---- per-thread A-slot geometry (fixed over the K loop) ----
    PixelDecomp pdec;
    pdec.init(p.Ho, p.Wo, M);
    const int k4 = tid & 3;
    int a_b[A_SLOTS], a_h0[A_SLOTS], a_w0[A_SLOTS];
    int a_pix0[A_SLOTS];       // (b*H + h0)*W + w0, may be negative (halo)
    unsigned a_vmask[A_SLOTS];  // bit t: tap t reads inside the image (aligned path: KH*KW <= 32)
    bool a_ok[A_SLOTS];
#pragma unroll
    for (int j = 0; j < A_SLOTS; ++j) {
        const int m = tm * BM + (tid >> 2) + j * 64;
        a_ok[j] = m < M;
        const int mm = a_ok[j] ? m : 0;
        int b, ho, wo;
        pdec.split(mm, &b, &ho, &wo);
        a_b[j] = b;
        a_h0[j] = ho * p.stride - p.pad;
        a_w0[j] = wo * p.stride - p.pad;
        a_pix0[j] = (b * p.H + a_h0[j]) * p.W + a_w0[j];
        unsigned vm = 0u;
        if (ALIGNED && !DCN && a_ok[j]) vm = tap_valid_mask(a_h0[j], a_w0[j], p.H, p.W, p.KH, p.KW);
        a_vmask[j] = vm;
    }

    float4 a_reg[A_SLOTS];
    float4 b_reg[B_SLOTS];

    const int nk = p.Kpad / BK;
    int kt0, kt1;
    splitk_range(p, nk, &kt0, &kt1);
    // wave-uniform K-walk state (ALIGNED): advanced once per K-step, lives in SGPRs
    int u_tap = 0, u_kh = 0, u_kw = 0, u_c0 = 0, u_src = 0, u_cs = 0;
    bool u_first = true;
    if (ALIGNED && kt0 > 0) {
        const int k0 = kt0 * BK;
        u_tap = k0 / p.Cin;
        u_c0 = k0 - u_tap * p.Cin;
        u_kh = u_tap / p.KW;
        u_kw = u_tap - u_kh * p.KW;
        u_cs = u_c0;
        if (MULTISRC) {
            for (int q = 0; q < 3; ++q) {
                const int cur = q == 0 ? p.src_c[0] : q == 1 ? p.src_c[1] : p.src_c[2];
                if (u_src == q && u_cs >= cur) { u_cs -= cur; ++u_src; }
            }
        }
    }
    // DCN: per-slot bilinear taps of the current kernel tap, recomputed only when the tap changes
    int d_idx[DCN ? A_SLOTS : 1][4];
    float d_w[DCN ? A_SLOTS : 1][4];

    const float* b_ptr[B_SLOTS];
#pragma unroll
    for (int j = 0; j < B_SLOTS; ++j) {
        const int f = tid + j * NTHREADS;
        const int row = f / (BN / 4), c4 = f % (BN / 4);
        b_ptr[j] = p.wp + (size_t)(kt0 * BK + row) * p.CoutPad + tn * BN + c4 * 4;
    }

    auto load_tile = [&](int kt) {
        // ---- B (weights): BK rows of BN floats ----
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) {
            const int f = tid + j * NTHREADS;
            if (B_F4 % NTHREADS == 0 || f < B_F4) {
                b_reg[j] = ld4(b_ptr[j]);
                b_ptr[j] += (size_t)BK * p.CoutPad;
            }
        }
        // ---- A ----
        if (ALIGNED && !DCN) {
            const float* base = p.src[0];
            int sc = p.src_c[0];
            if (MULTISRC) {
                if (u_src == 1) { base = p.src[1]; sc = p.src_c[1]; }
                else if (u_src == 2) { base = p.src[2]; sc = p.src_c[2]; }
                else if (u_src == 3) { base = p.src[3]; sc = p.src_c[3]; }
            }
            const int tap_pix = u_kh * p.W + u_kw;
            const int coff = u_cs + k4 * 4;
            const unsigned bit = 1u << u_tap;
#pragma unroll
            for (int j = 0; j < A_SLOTS; ++j) {
                const long long off = (long long)(a_pix0[j] + tap_pix) * sc + coff;
                a_reg[j] = (a_vmask[j] & bit) ? ld4(base + off) : zero4();
            }
            if (!MULTISRC && p.gn_in_mr) {
                // fused GroupNorm + affine + ReLU of the producer's raw output (pose_dla_dcn.py:499-503)
                const float4 ga = ld4(p.gn_in_gamma + coff), be = ld4(p.gn_in_beta + coff);
                const int g = coff / p.gn_cpg;
#pragma unroll
                for (int j = 0; j < A_SLOTS; ++j) {
                    if (a_vmask[j] & bit) {
                        const float mu = p.gn_in_mr[(a_b[j] * p.gn_groups + g) * 2];
                        const float rs = p.gn_in_mr[(a_b[j] * p.gn_groups + g) * 2 + 1];
                        float4 v = a_reg[j];
                        v.x = fmaxf((v.x - mu) * rs * ga.x + be.x, 0.f);
                        v.y = fmaxf((v.y - mu) * rs * ga.y + be.y, 0.f);
                        v.z = fmaxf((v.z - mu) * rs * ga.z + be.z, 0.f);
                        v.w = fmaxf((v.w - mu) * rs * ga.w + be.w, 0.f);
                        a_reg[j] = v;
                    }
                }
            }
        } else if (!DCN) {
            const int kk = kt * BK + k4 * 4;
            const int tap = kk / p.Cin;
            const int ci = kk - tap * p.Cin;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const bool kvalid = kk < p.K;
            const float* base = p.src[0];
            const int sc = p.src_c[0];  // unaligned convolutions are single-source (checked by the launcher)
#pragma unroll
            for (int j = 0; j < A_SLOTS; ++j) {
                const int hi = a_h0[j] + kh, wi = a_w0[j] + kw;
                const bool ok = a_ok[j] && kvalid && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
                a_reg[j] = ok ? ld4(base + ((size_t)(a_b[j] * p.H + hi) * p.W + wi) * sc + ci) : zero4();
            }
        } else {
            const float* base = p.src[0];
            const int C = p.Cin;
            if (u_c0 == 0 || u_first) {
                u_first = false;
                // new kernel tap: sample positions and (mask-folded) bilinear weights per slot
#pragma unroll
                for (int j = 0; j < A_SLOTS; ++j) {
                    int i0 = -1, i1 = -1, i2 = -1, i3 = -1;
                    float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
                    if (a_ok[j]) {
                        // (ho, wo) = (a_h0 + pad, a_w0 + pad) since stride == 1
                        const size_t pix = (size_t)(a_b[j] * p.H + (a_h0[j] + p.pad)) * p.W + (a_w0[j] + p.pad);
                        const float* om = p.offmask + pix * 32;
                        const float dh = om[2 * u_tap], dw = om[2 * u_tap + 1], mk = om[18 + u_tap];
                        const float h_im = (float)(a_h0[j] + u_kh) + dh;
                        const float w_im = (float)(a_w0[j] + u_kw) + dw;
                        if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
                            const int h_lo = (int)floorf(h_im), w_lo = (int)floorf(w_im);
                            const int h_hi = h_lo + 1, w_hi = w_lo + 1;
                            const float lh = h_im - (float)h_lo, lw = w_im - (float)w_lo;
                            const float hh = 1.f - lh, hw = 1.f - lw;
                            const int bb = a_b[j] * p.H;
                            if (h_lo >= 0 && w_lo >= 0) i0 = (bb + h_lo) * p.W + w_lo;
                            if (h_lo >= 0 && w_hi <= p.W - 1) i1 = (bb + h_lo) * p.W + w_hi;
                            if (h_hi <= p.H - 1 && w_lo >= 0) i2 = (bb + h_hi) * p.W + w_lo;
                            if (h_hi <= p.H - 1 && w_hi <= p.W - 1) i3 = (bb + h_hi) * p.W + w_hi;
                            w1 = hh * hw * mk; w2 = hh * lw * mk; w3 = lh * hw * mk; w4 = lh * lw * mk;
                        }
                    }
                    d_idx[j][0] = i0; d_idx[j][1] = i1; d_idx[j][2] = i2; d_idx[j][3] = i3;
                    d_w[j][0] = w1; d_w[j][1] = w2; d_w[j][2] = w3; d_w[j][3] = w4;
                }
            }
            const int coff = u_c0 + k4 * 4;
#pragma unroll
            for (int j = 0; j < A_SLOTS; ++j) {
                float4 v1 = zero4(), v2 = zero4(), v3 = zero4(), v4 = zero4();
                if (d_idx[j][0] >= 0) v1 = ld4(base + (long long)d_idx[j][0] * C + coff);
                if (d_idx[j][1] >= 0) v2 = ld4(base + (long long)d_idx[j][1] * C + coff);
                if (d_idx[j][2] >= 0) v3 = ld4(base + (long long)d_idx[j][2] * C + coff);
                if (d_idx[j][3] >= 0) v4 = ld4(base + (long long)d_idx[j][3] * C + coff);
                const float w1 = d_w[j][0], w2 = d_w[j][1], w3 = d_w[j][2], w4 = d_w[j][3];
                float4 v;
                v.x = fmaf(w4, v4.x, fmaf(w3, v3.x, fmaf(w2, v2.x, w1 * v1.x)));
                v.y = fmaf(w4, v4.y, fmaf(w3, v3.y, fmaf(w2, v2.y, w1 * v1.y)));
                v.z = fmaf(w4, v4.z, fmaf(w3, v3.z, fmaf(w2, v2.z, w1 * v1.z)));
                v.w = fmaf(w4, v4.w, fmaf(w3, v3.w, fmaf(w2, v2.w, w1 * v1.w)));
                a_reg[j] = v;
            }
        }
        if (ALIGNED) {  // advance the uniform K walk
            u_c0 += BK;
            u_cs += BK;
            if (u_c0 >= p.Cin) {
                u_c0 = 0; u_cs = 0; u_src = 0;
                ++u_tap;
                if (++u_kw == p.KW) { u_kw = 0; ++u_kh; }
            } else if (MULTISRC) {
                const int cur = u_src == 0 ? p.src_c[0] : u_src == 1 ? p.src_c[1] : u_src == 2 ? p.src_c[2] : p.src_c[3];
                if (u_cs >= cur) { u_cs = 0; ++u_src; }
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

    load_tile(kt0);
    store_tile(0);
    __syncthreads();

    const int lrow = lane / FRAG;  // k index inside a KSTEP
    const int lcol = lane % FRAG;
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) load_tile(kt + 1);
        const float* A = As + buf * A_SZ + wm * (MT * FRAG) + lcol;
        const float* Bt = Bs + buf * B_SZ + wn * (NT * FRAG) + lcol;
        // fragment reads for k-step s+1 are issued before the MFMAs of step s (register double buffer), so
        // the LDS latency hides under the matrix pipe instead of stalling the in-order wave
        constexpr int NKS = BK / F::KSTEP;
        float a[2][MT], b[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[0][i] = A[lrow * LDA + i * FRAG];
#pragma unroll
        for (int j = 0; j < NT; ++j) b[0][j] = Bt[lrow * LDB + j * FRAG];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < NKS) {
                const int krow = (ks + 1) * F::KSTEP + lrow;
#pragma unroll
                for (int i = 0; i < MT; ++i) a[nxt][i] = A[krow * LDA + i * FRAG];
#pragma unroll
                for (int j = 0; j < NT; ++j) b[nxt][j] = Bt[krow * LDB + j * FRAG];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ahead of this step's MFMAs
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = F::mfma(a[cur][i], b[cur][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kt + 1 < kt1) store_tile(buf ^ 1);
        __syncthreads();
    }

    if (p.splitk > 1) igemm_store_partial<FRAG, MT, NT, WM, WN>(p, acc, tm, tn, wm, wn, lane, blockIdx.y);
    else igemm_epilogue<FRAG, MT, NT, WM, WN>(p, acc, tm, tn, wm, wn, lane);
}

template <int FRAG, int MT, int NT, int WM, int WN, bool DCN, bool ALIGNED, bool MULTISRC>
int launch_a(const ConvParams& p, hipStream_t stream) {
    constexpr int BM = FRAG * MT * WM, BN = FRAG * NT * WN;
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_m = (M + BM - 1) / BM;
    // N tiles that hold real output columns (weights may be padded wider than this kernel's N tile, e.g. the <= 16-wide
    // final heads are packed to 32 columns for the f16x3 kernel: the all-padding tile is not launched)
    const int tiles_n = (p.Cout + BN - 1) / BN;
    if (p.CoutPad % BN != 0 || p.Kpad % BK != 0) return CP_ERR_INVALID;
    hipLaunchKernelGGL((igemm_kernel<FRAG, MT, NT, WM, WN, DCN, ALIGNED, MULTISRC>),
                       dim3(tiles_m * tiles_n, p.splitk > 1 ? p.splitk : 1), dim3(NTHREADS), 0, stream, p, tiles_m, tiles_n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

static bool conv_aligned(const ConvParams& p) {
    bool aligned = p.Cin % BK == 0 && p.KH * p.KW <= 32;
    for (int s = 0; s < p.nsrc; ++s) aligned = aligned && (p.src_c[s] % BK == 0);
    return aligned;
}

template <int FRAG, int MT, int NT, int WM, int WN, bool DCN>
int launch(const ConvParams& p, hipStream_t stream) {
    const bool aligned = conv_aligned(p);
    if (DCN) return aligned ? launch_a<FRAG, MT, NT, WM, WN, DCN, true, false>(p, stream) : CP_ERR_INVALID;
    if (aligned && p.nsrc > 1) return launch_a<FRAG, MT, NT, WM, WN, false, true, true>(p, stream);
    if (aligned) return launch_a<FRAG, MT, NT, WM, WN, false, true, false>(p, stream);
    if (p.nsrc != 1) return CP_ERR_INVALID;
    return launch_a<FRAG, MT, NT, WM, WN, false, false, false>(p, stream);
}

}  // namespace

int cp_conv_tile_n(int cout) {
    if (cout <= 16) return 16;
    if (cout <= 32) return 32;
    if (cout % 128 == 0) return 128;
    return 64;
}

// Kernel variant id = one template instantiation family (what rocprofv3 lists as one kernel name):
//   0..3  plain conv, N tile 16/32/64/128      4..5  fused DCNv2, N tile 64/128
//   6..9  plain conv over a virtual channel concat (Root), N tile 16/32/64/128
//   10..13 unaligned-channel plain conv (7x7 stems), N tile 16/32/64/128
int cp_conv_variant(const ConvParams& p) {
    const int bn = cp_conv_tile_n(p.Cout);
    if (p.offmask) return bn == 128 ? 5 : 4;
    const int t = bn == 16 ? 0 : bn == 32 ? 1 : bn == 64 ? 2 : 3;
    if (!conv_aligned(p)) return 10 + t;
    return (p.nsrc > 1 ? 6 : 0) + t;
}

const char* cp_conv_variant_name(int v) {
    static const char* names[CP_NUM_CONV_VARIANTS] = {
        "igemm_f32_16x16x4_m256n16", "igemm_f32_32x32x2_m256n32", "igemm_f32_32x32x2_m128n64",
        "igemm_f32_32x32x2_m128n128", "dcn_igemm_f32_32x32x2_m128n64", "dcn_igemm_f32_32x32x2_m128n128",
        "igemm_cat_f32_16x16x4_m256n16", "igemm_cat_f32_32x32x2_m256n32", "igemm_cat_f32_32x32x2_m128n64",
        "igemm_cat_f32_32x32x2_m128n128", "igemm_unaligned_f32_16x16x4_m256n16", "igemm_unaligned_f32_32x32x2_m256n32",
        "igemm_unaligned_f32_32x32x2_m128n64", "igemm_unaligned_f32_32x32x2_m128n128",
        "igemm16_f16x3_m128n32", "igemm16_f16x3_m128n64", "igemm16_f16x3_m128n128", "dcn_igemm16_f16x3_m128n64",
        "dcn_igemm16_f16x3_m128n128", "igemm16_cat_f16x3_m128n32", "igemm16_cat_f16x3_m128n64",
        "igemm16_cat_f16x3_m128n128", "igemm16_head_f16x3_m128n128",
        "lowc_stem7x7_f16x3", "lowc_3x3_c16_f16x3", "lowc_3x3s2_c16_f16x3", "igemm16_gru_f16x3_m128n96",
        "halo16_f16x3_m128n32", "halo16_f16x3_m128n64", "halo16_f16x3_m128n128", "dcn16p_f16x3_p128n64", "gn_final_f32_valu", "halo16_head_f16x3_m128n128",
        "halo16_gru_f16x3_m128n96", "pw16_f16x3_m128n64", "pw16_f16x3_m128n128", "dcn16s_f16x3_p128n64", "igemm16_f16x3_m64n64", "dcn16p_f16x3_p128n128",
        "dcn16t_f16x3_p128n64", "lowc_stem_level0_f16x3", "strm16_f16x3_w32n32", "lowc_3x3s2_c16_rows_f16x3"};
    return (v >= 0 && v < CP_NUM_CONV_VARIANTS) ? names[v] : "?";
}

void cp_conv_geometry(const ConvParams& p, bool f16x3, int* tiles, int* nk) {
    int bn = cp_conv_tile_n(p.Cout);
    if (f16x3 && bn < 32) bn = 32;
    if (f16x3 && p.tile_n && p.tile_n < bn && p.CoutPad % p.tile_n == 0 && (!p.offmask || p.tile_n == 64)) bn = p.tile_n;
    const int bm = f16x3 ? ((bn == 64 && p.tile_m == 64) ? 64 : 128) : (bn <= 32 ? 256 : 128);
    const int M = p.B * p.Ho * p.Wo;
    *tiles = ((M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn);
    *nk = f16x3 ? p.Kpad16 / 32 : p.Kpad / BK;
}

namespace {
// split-K epilogue: out = act((sum_slices partial) * scale + shift + residual), slices summed in index order
__global__ void splitk_epilogue_kernel(const ConvParams p) {
    const int M = p.B * p.Ho * p.Wo, HWo = p.Ho * p.Wo;
    const size_t total = (size_t)M * p.Cout;
    float afwd = 1.f, ainv = 1.f, amax = 0.f;
    const AmaxRaw amax_raw = conv_in_scale_issue(p);  // the 32 scalar loads go out first, their reduction waits (below)
    bool have_scale = false;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int m, n;
        if (p.store == CP_STORE_NHWC) { m = (int)(i / p.Cout); n = (int)(i - (size_t)m * p.Cout); }
        else { n = (int)(i / M); m = (int)(i - (size_t)n * M); }
        // eight slab loads in flight per lane, then summed in slice order (a missing slice adds an exact 0)
        float acc = 0.f;
        const float* src = p.partial + (size_t)m * p.CoutPad + n;
        const size_t slab = (size_t)M * p.CoutPad;
        for (int z0 = 0; z0 < p.splitk; z0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (z0 + j < p.splitk) ? src[(size_t)(z0 + j) * slab] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        // the activation scale (32 scalar loads issued at the top) is reduced behind the slab loads, not in front of them: this
        // kernel is one memory round trip long, and it runs 40-odd times per batch-1 frame
        const float sc = p.scale ? p.scale[n] : 1.f, sh = p.shift ? p.shift[n] : 0.f;
        const float rs = p.res ? p.res[(size_t)m * p.res_ld + n] : 0.f;
        if (!have_scale) {
            conv_in_scale_finish(p, amax_raw, &afwd, &ainv);
            have_scale = true;
        }
        float y = acc * (sc * ainv) + sh;
        if (p.res) y += rs;
        if (p.act == CP_ACT_RELU) y = fmaxf(y, 0.f);
        else if (p.act == CP_ACT_SIGMOID || (p.act == CP_ACT_SIGMOID_FROM && n >= p.act_from)) y = 1.f / (1.f + expf(-y));
        amax = fmaxf(amax, fabsf(y));
        if (p.store == CP_STORE_NHWC) p.out[(size_t)m * p.ldo + p.coff + n] = y;
        else {
            const int b = m / HWo, pix = m - b * HWo;
            p.out[((size_t)b * p.ldo + p.coff + n) * HWo + pix] = y;
        }
    }
    if (p.out_amax) cp_amax_commit(p.out_amax, amax);
}

// The same for NHWC outputs with whole channel quads (every split-K layer of the networks): one lane = four consecutive
// channels of a pixel, ALL slab reads (<= 32 slices x 16 bytes) in flight at once behind one buffer descriptor (slices beyond
// splitk read out of range = exact zeros: the descriptor's num_records check includes soffset on gfx9-family parts, which
// cp_common.h's architecture guard pins; a uniform `z < splitk` guard per load was tried instead and costs 76 scalar
// branches + 39 spilled SGPRs in a kernel that is one memory round trip long), then the sum in slice order.  The element-wise form above
// keeps 8 four-byte loads in flight and walks 32 slices in four dependent rounds: 6.5 us per launch, 48 launches = 0.31 ms of a
// 1.43 ms batch-1 frame (profiles/r05_frame_trace_dla_34.txt).  Same additions in the same order: bit-identical.
typedef uint32_t sk_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void splitk_epilogue4_kernel(const ConvParams p) {
    const int M = p.B * p.Ho * p.Wo, C4 = p.Cout >> 2;
    const int total = M * C4;
    float afwd = 1.f, ainv = 1.f, amax = 0.f;
    const AmaxRaw amax_raw = conv_in_scale_issue(p);
    bool have_scale = false;
    const unsigned slab = (unsigned)M * (unsigned)p.CoutPad * 4u;  // bytes per slice
    const __amdgpu_buffer_rsrc_t rp = make_rsrc(p.partial, slab * (unsigned)p.splitk);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int m = i / C4, n = (i - m * C4) << 2;
        const unsigned off = ((unsigned)m * (unsigned)p.CoutPad + (unsigned)n) * 4u;
        sk_u32x4 v[32];
#pragma unroll
        for (int z = 0; z < 32; ++z) v[z] = __builtin_amdgcn_raw_buffer_load_b128(rp, (int)off, (int)((unsigned)z * slab), 0);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int z = 0; z < 32; ++z) {
            a0 += __uint_as_float(v[z].x); a1 += __uint_as_float(v[z].y); a2 += __uint_as_float(v[z].z); a3 += __uint_as_float(v[z].w);
        }
        const float4 sc = p.scale ? *reinterpret_cast<const float4*>(p.scale + n) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 sh = p.shift ? *reinterpret_cast<const float4*>(p.shift + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 rs = p.res ? *reinterpret_cast<const float4*>(p.res + (size_t)m * p.res_ld + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (!have_scale) {
            conv_in_scale_finish(p, amax_raw, &afwd, &ainv);
            have_scale = true;
        }
        float y[4] = {a0 * (sc.x * ainv) + sh.x, a1 * (sc.y * ainv) + sh.y, a2 * (sc.z * ainv) + sh.z, a3 * (sc.w * ainv) + sh.w};
        const float re[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (p.res) y[e] += re[e];
            if (p.act == CP_ACT_RELU) y[e] = fmaxf(y[e], 0.f);
            else if (p.act == CP_ACT_SIGMOID || (p.act == CP_ACT_SIGMOID_FROM && n + e >= p.act_from)) y[e] = 1.f / (1.f + expf(-y[e]));
            amax = fmaxf(amax, fabsf(y[e]));
        }
        *reinterpret_cast<float4*>(p.out + (size_t)m * p.ldo + p.coff + n) = make_float4(y[0], y[1], y[2], y[3]);
    }
    if (p.out_amax) cp_amax_commit(p.out_amax, amax);
}
}  // namespace

int cp_launch_splitk_epilogue(const ConvParams& p, hipStream_t stream) {
    const size_t total = (size_t)p.B * p.Ho * p.Wo * p.Cout;
    const size_t slab_bytes = (size_t)p.B * p.Ho * p.Wo * p.CoutPad * 4;
    const bool quad = p.store == CP_STORE_NHWC && p.Cout % 4 == 0 && p.CoutPad % 4 == 0 && p.ldo % 4 == 0 && p.coff % 4 == 0 &&
                      (!p.res || p.res_ld % 4 == 0) && p.splitk <= 32 && slab_bytes * 32 < (size_t)0xf0000000u &&
                      total / 4 < (size_t)0x7fffffff && !(p.dbg & 8);  // (cp_set_debug 8: the element-wise form, A/B)
    if (quad) {
        int g = (int)((total / 4 + 255) / 256);
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(splitk_epilogue4_kernel, dim3(g), dim3(256), 0, stream, p);
        return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
    }
    int g = (int)((total + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(g), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
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
