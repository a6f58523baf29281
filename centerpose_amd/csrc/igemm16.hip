// Implicit-GEMM convolution / fused DCNv2 on the gfx950 f16 matrix cores with float32-class accuracy:
// every float32 operand x is split into two binary16 numbers x = hi + lo (hi = rtz16(x), lo = rtz16(x - hi),
// |lo| < 2^-10 |x|) and each product is evaluated as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  with float32
// accumulation inside v_mfma_f32_32x32x16_f16 (the dropped a_lo*b_lo term is < 2^-20 relative).  Three f16 MFMAs
// replace one f32 MFMA that has 1/16 of the rate, so the matrix ceiling rises from 157 TFLOP/s (exact-f32
// MFMA, igemm.hip) to 2.5 PFLOP/s / 3 = 833 TFLOP/s of algorithmic FLOPs, while activations, weights at the
// boundary, accumulation and every epilogue stay float32.  Measured end-to-end error vs the reference graph is
// reported by tests/ and DESIGN.md (budget 1e-3 on heat-maps).
//
// Same GEMM view, tile map, K-walk and epilogue as igemm.hip.  Differences that follow from the MFMA shape:
//   * operand fragments are 8 consecutive k per lane (A[m = lane%32][k = 8*(lane/32) .. +7]), so both tiles live
//     in LDS row-major with k contiguous ([m][k] / [n][k]); the NHWC float4 a lane loads IS 4 consecutive k of one
//     pixel, so the A tile needs no transpose; rows are 64 bytes, unpadded, with the 16-byte chunks XOR-swizzled by
//     the row (see swz()) so that ds_read_b128 / ds_write_b64 lane groups are bank-conflict free;
//   * weights are split and packed once at model load as [co][tap][ci] binary16 pairs (hi / lo arrays);
//   * BK = 32 (two 32x32x16 MFMA k-steps per LDS tile).
// Kernels: igemm16p_kernel (hand-pipelined K loop; all non-DCN layers, optional fused prediction head) and
// igemm16_kernel (previous loop structure; DCN gather layers).  Requirements: every source's channel count % 32 == 0;
// the 16-channel layers at the top of the network run in lowc.hip, the <= 16-wide GroupNorm'd final 1x1 heads on the
// exact-f32 kernel.
#include "igemm16_common.h"

namespace {

// PF2: two register sets for the global->LDS staging, i.e. tile t+2 is in flight while tile t is multiplied (a
// 32-deep K-step is only ~770 MFMA cycles per wave, shorter than an L2 round trip under load).
template <int MT, int NT, int WM, int WN, bool DCN, bool MULTISRC, bool PF2>
__global__ __launch_bounds__(NT16, (DCN && NT == 1) ? 3 : 2) void igemm16_kernel(const ConvParams p, const int tiles_m, const int tiles_n) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    static_assert(WM * WN * 64 == NT16, "4 waves");
    constexpr int A_SLOTS = BM * BK16 / 4 / NT16;          // float4 per thread per K-step
    constexpr int B_CHUNKS = BN * BK16 * 2 / 16;            // 16-byte chunks per array (hi or lo)
    constexpr int B_SLOTS = (B_CHUNKS + NT16 - 1) / NT16;   // per array
    constexpr int A_SZ = BM * LDH, B_SZ = BN * LDH;         // halfs per array per buffer
    constexpr int BUF = 2 * A_SZ + 2 * B_SZ;
    // two distinct LDS objects (not one array): in the fused steady state the stores go to one buffer and the
    // fragment reads to the other, and the compiler may only interleave them if it can prove they do not alias
    __shared__ __attribute__((aligned(16))) _Float16 lds0[BUF];
    __shared__ __attribute__((aligned(16))) _Float16 lds1[BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int tile = tile_of_block(tiles_m, tiles_n);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int M = p.B * p.Ho * p.Wo;
    PixelDecomp pdec;
    pdec.init(p.Ho, p.Wo, M);

    // ---- per-thread A-slot geometry: slot j covers pixel row (tid / 8) + 32 * j, float4 column tid % 8 ----
    const int k4 = tid & 7;
    int a_b[A_SLOTS], a_h0[A_SLOTS], a_w0[A_SLOTS], a_pix0[A_SLOTS];
    unsigned a_vmask[A_SLOTS], a_byte0[A_SLOTS];
    bool a_ok[A_SLOTS];
#pragma unroll
    for (int j = 0; j < A_SLOTS; ++j) {
        const int m = tm * BM + (tid >> 3) + j * 32;
        a_ok[j] = m < M;
        const int mm = a_ok[j] ? m : 0;
        int b, ho, wo;
        pdec.split(mm, &b, &ho, &wo);
        a_b[j] = b;
        a_h0[j] = ho * p.stride - p.pad;
        a_w0[j] = wo * p.stride - p.pad;
        a_pix0[j] = (b * p.H + a_h0[j]) * p.W + a_w0[j];
        a_byte0[j] = (unsigned)(a_pix0[j] * p.src_c[0] + k4 * 4) * 4u;  // single-source fast path (wraps for halo; masked)
        a_vmask[j] = (!DCN && a_ok[j]) ? tap_valid_mask(a_h0[j], a_w0[j], p.H, p.W, p.KH, p.KW) : 0u;
    }

    float4 a_reg0[A_SLOTS], a_reg1[PF2 ? A_SLOTS : 1];
    u32x4 bh_reg0[B_SLOTS], bl_reg0[B_SLOTS], bh_reg1[PF2 ? B_SLOTS : 1], bl_reg1[PF2 ? B_SLOTS : 1];
    const int nk = p.Kpad16 / BK16;
    int kt0, kt1;
    splitk_range(p, nk, &kt0, &kt1);
    int u_tap = 0, u_kh = 0, u_kw = 0, u_c0 = 0, u_src = 0, u_cs = 0;
    bool u_first = true;
    if (kt0 > 0) {
        const int k0 = kt0 * BK16;
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
    int d_idx[DCN ? A_SLOTS : 1][4];
    float d_w[DCN ? A_SLOTS : 1][4];
    // power-of-two pre-scale of the activations (ConvParams::in_amax); DCN folds it into the bilinear weights
    float afwd, ainv;
    conv_in_scale(p, &afwd, &ainv);

    // weights: [CoutPad][Kpad16] binary16, k contiguous; chunk f -> row n = f / 4, 16-byte column f % 4
    const _Float16* wh = reinterpret_cast<const _Float16*>(p.w16_hi);
    const _Float16* wl = reinterpret_cast<const _Float16*>(p.w16_lo);
    const unsigned w_bytes = (unsigned)((size_t)p.CoutPad * p.Kpad16 * 2);
    const __amdgpu_buffer_rsrc_t r_wh = make_rsrc(wh, w_bytes), r_wl = make_rsrc(wl, w_bytes);
    const unsigned img_px = (unsigned)p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t r_s0 = make_rsrc(p.src[0], img_px * p.src_c[0] * 4u);
    const __amdgpu_buffer_rsrc_t r_s1 = make_rsrc(MULTISRC ? p.src[1] : p.src[0], MULTISRC ? img_px * p.src_c[1] * 4u : 0u);
    const __amdgpu_buffer_rsrc_t r_s2 = make_rsrc(MULTISRC && p.nsrc > 2 ? p.src[2] : p.src[0], MULTISRC && p.nsrc > 2 ? img_px * p.src_c[2] * 4u : 0u);
    const __amdgpu_buffer_rsrc_t r_s3 = make_rsrc(MULTISRC && p.nsrc > 3 ? p.src[3] : p.src[0], MULTISRC && p.nsrc > 3 ? img_px * p.src_c[3] * 4u : 0u);
    unsigned b_off[B_SLOTS];  // byte offsets into the packed binary16 weights
#pragma unroll
    for (int j = 0; j < B_SLOTS; ++j) {
        const int f = tid + j * NT16;
        b_off[j] = (unsigned)(((size_t)(tn * BN + f / 4) * p.Kpad16 + (f % 4) * 8 + (size_t)kt0 * BK16) * 2);
    }

    auto load_tile = [&](float4* a_reg, u32x4* bh_reg, u32x4* bl_reg) {
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) {
            const int f = tid + j * NT16;
            if (B_CHUNKS % NT16 == 0 || f < B_CHUNKS) {
                bh_reg[j] = __builtin_amdgcn_raw_buffer_load_b128(r_wh, (int)b_off[j], 0, 0);
                bl_reg[j] = __builtin_amdgcn_raw_buffer_load_b128(r_wl, (int)b_off[j], 0, 0);
                b_off[j] += BK16 * 2;
            }
        }
        if (!DCN) {
            __amdgpu_buffer_rsrc_t rs = r_s0;
            int sc = p.src_c[0];
            if (MULTISRC) {
                if (u_src == 1) { rs = r_s1; sc = p.src_c[1]; }
                else if (u_src == 2) { rs = r_s2; sc = p.src_c[2]; }
                else if (u_src == 3) { rs = r_s3; sc = p.src_c[3]; }
            }
            const int tap_pix = u_kh * p.W + u_kw;
            const int coff = u_cs + k4 * 4;
            const unsigned bit = 1u << u_tap;
            if (MULTISRC) {
#pragma unroll
                for (int j = 0; j < A_SLOTS; ++j) {
                    const unsigned off = (unsigned)((a_pix0[j] + tap_pix) * sc + coff) * 4u;
                    a_reg[j] = buf_ld4(rs, (a_vmask[j] & bit) ? off : OOB);
                }
            } else {
                // per-slot byte base is fixed; the (tap, channel) part is wave-uniform scalar arithmetic
                const unsigned uoff = (unsigned)(tap_pix * sc + u_cs) * 4u;
#pragma unroll
                for (int j = 0; j < A_SLOTS; ++j) a_reg[j] = buf_ld4(rs, (a_vmask[j] & bit) ? a_byte0[j] + uoff : OOB);
            }
        } else {
            const int C = p.Cin;
            if (u_c0 == 0 || u_first) {
                u_first = false;
#pragma unroll
                for (int j = 0; j < A_SLOTS; ++j) {
                    int i0 = (int)OOB_BASE, i1 = (int)OOB_BASE, i2 = (int)OOB_BASE, i3 = (int)OOB_BASE;
                    float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
                    if (a_ok[j]) {
                        const size_t pix = (size_t)(a_b[j] * p.H + (a_h0[j] + p.pad)) * p.W + (a_w0[j] + p.pad);
                        const float* om = p.offmask + pix * 32;
                        const float dh = om[2 * u_tap], dw = om[2 * u_tap + 1], mk = om[18 + u_tap] * afwd;
                        const float h_im = (float)(a_h0[j] + u_kh) + dh;
                        const float w_im = (float)(a_w0[j] + u_kw) + dw;
                        if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
                            const int h_lo = (int)floorf(h_im), w_lo = (int)floorf(w_im);
                            const int h_hi = h_lo + 1, w_hi = w_lo + 1;
                            const float lh = h_im - (float)h_lo, lw = w_im - (float)w_lo;
                            const float hh = 1.f - lh, hw = 1.f - lw;
                            // byte offsets of the 4 corners' channel vectors (invalid corners keep OOB_BASE, which
                            // stays out of range after the small per-K-step channel offset is added)
                            const int bb = a_b[j] * p.H;
                            const int cb = C * 4;
                            if (h_lo >= 0 && w_lo >= 0) i0 = ((bb + h_lo) * p.W + w_lo) * cb;
                            if (h_lo >= 0 && w_hi <= p.W - 1) i1 = ((bb + h_lo) * p.W + w_hi) * cb;
                            if (h_hi <= p.H - 1 && w_lo >= 0) i2 = ((bb + h_hi) * p.W + w_lo) * cb;
                            if (h_hi <= p.H - 1 && w_hi <= p.W - 1) i3 = ((bb + h_hi) * p.W + w_hi) * cb;
                            w1 = hh * hw * mk; w2 = hh * lw * mk; w3 = lh * hw * mk; w4 = lh * lw * mk;
                        }
                    }
                    d_idx[j][0] = i0; d_idx[j][1] = i1; d_idx[j][2] = i2; d_idx[j][3] = i3;
                    d_w[j][0] = w1; d_w[j][1] = w2; d_w[j][2] = w3; d_w[j][3] = w4;
                }
            }
            const unsigned coff = (unsigned)(u_c0 + k4 * 4) * 4u;
#pragma unroll
            for (int j = 0; j < A_SLOTS; ++j) {
                const float4 v1 = buf_ld4(r_s0, (unsigned)d_idx[j][0] + coff);
                const float4 v2 = buf_ld4(r_s0, (unsigned)d_idx[j][1] + coff);
                const float4 v3 = buf_ld4(r_s0, (unsigned)d_idx[j][2] + coff);
                const float4 v4 = buf_ld4(r_s0, (unsigned)d_idx[j][3] + coff);
                const float w1 = d_w[j][0], w2 = d_w[j][1], w3 = d_w[j][2], w4 = d_w[j][3];
                float4 v;
                v.x = fmaf(w4, v4.x, fmaf(w3, v3.x, fmaf(w2, v2.x, w1 * v1.x)));
                v.y = fmaf(w4, v4.y, fmaf(w3, v3.y, fmaf(w2, v2.y, w1 * v1.y)));
                v.z = fmaf(w4, v4.z, fmaf(w3, v3.z, fmaf(w2, v2.z, w1 * v1.z)));
                v.w = fmaf(w4, v4.w, fmaf(w3, v3.w, fmaf(w2, v2.w, w1 * v1.w)));
                a_reg[j] = v;
            }
        }
        // advance the wave-uniform K walk by one 32-wide step
        u_c0 += BK16;
        u_cs += BK16;
        if (u_c0 >= p.Cin) {
            u_c0 = 0; u_cs = 0; u_src = 0;
            ++u_tap;
            if (++u_kw == p.KW) { u_kw = 0; ++u_kh; }
        } else if (MULTISRC) {
            const int cur = u_src == 0 ? p.src_c[0] : u_src == 1 ? p.src_c[1] : u_src == 2 ? p.src_c[2] : p.src_c[3];
            if (u_cs >= cur) { u_cs = 0; ++u_src; }
        }
    };

    auto store_tile = [&](int buf, const float4* a_reg, const u32x4* bh_reg, const u32x4* bl_reg) {
        _Float16* Ah = buf ? lds1 : lds0;
        _Float16* Al = Ah + A_SZ;
        _Float16* Bh = Al + A_SZ;
        _Float16* Bl = Bh + B_SZ;
#pragma unroll
        for (int j = 0; j < A_SLOTS; ++j) {
            const int row = (tid >> 3) + j * 32;
            const float as = DCN ? 1.f : afwd;
            const Split2 s0 = split2(a_reg[j].x * as, a_reg[j].y * as), s1 = split2(a_reg[j].z * as, a_reg[j].w * as);
            const int col = ((((k4 >> 1) ^ swz(row)) << 1) | (k4 & 1)) * 4;  // halfs
            *reinterpret_cast<u32x2*>(Ah + row * LDH + col) = u32x2{s0.hi, s1.hi};
            *reinterpret_cast<u32x2*>(Al + row * LDH + col) = u32x2{s0.lo, s1.lo};
        }
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) {
            const int f = tid + j * NT16;
            if (B_CHUNKS % NT16 == 0 || f < B_CHUNKS) {
                const int n = f / 4, c = f % 4;
                *reinterpret_cast<u32x4*>(Bh + n * LDH + (c ^ swz(n)) * 8) = bh_reg[j];
                *reinterpret_cast<u32x4*>(Bl + n * LDH + (c ^ swz(n)) * 8) = bl_reg[j];
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

    const int lrow = lane >> 5;  // which 8-wide k group of the 16-deep MFMA step
    const int lcol = lane & 31;

    auto mma_tile = [&](int buf) {
        // fragment rows are (tile base, a multiple of 32) + lcol, so the swizzle only depends on lcol
        const _Float16* base = buf ? lds1 : lds0;
        const _Float16* Ah = base + (wm * (MT * 32) + lcol) * LDH;
        const _Float16* Al = Ah + A_SZ;
        const _Float16* Bh = base + 2 * A_SZ + (wn * (NT * 32) + lcol) * LDH;
        const _Float16* Bl = Bh + B_SZ;
#pragma unroll
        for (int ks = 0; ks < BK16 / 16; ++ks) {
            const int co = (((ks * 2 + lrow) ^ swz(lcol)) * 8);
            h8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ah[i] = *reinterpret_cast<const h8*>(Ah + i * 32 * LDH + co);
                al[i] = *reinterpret_cast<const h8*>(Al + i * 32 * LDH + co);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bh[j] = *reinterpret_cast<const h8*>(Bh + j * 32 * LDH + co);
                bl[j] = *reinterpret_cast<const h8*>(Bl + j * 32 * LDH + co);
            }
            // term-major order: an accumulator is reused only after the MT*NT-1 other fragments' MFMAs, so no
            // MFMA waits on the result of the one issued just before it
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    };

    if (!PF2) {
        load_tile(a_reg0, bh_reg0, bl_reg0);
        store_tile(0, a_reg0, bh_reg0, bl_reg0);
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            if (kt + 1 < kt1) load_tile(a_reg0, bh_reg0, bl_reg0);
            mma_tile(buf);
            if (kt + 1 < kt1) store_tile(buf ^ 1, a_reg0, bh_reg0, bl_reg0);
            __syncthreads();
        }
    } else {
        // tiles t+1 and t+2 are in registers / in flight while tile t is multiplied out of LDS.  In the steady state
        // the conversion + LDS store of tile t+1 is issued in the same basic block as the MFMAs of tile t and the
        // scheduler is told to interleave them (one MFMA, a few VALU, an LDS op, ...): the in-order wave then
        // converts and stores under the 32-cycle shadow of each MFMA instead of after all of them.
        const int n = kt1 - kt0;
        load_tile(a_reg0, bh_reg0, bl_reg0);
        if (n > 1) load_tile(a_reg1, bh_reg1, bl_reg1);
        store_tile(0, a_reg0, bh_reg0, bl_reg0);
        __syncthreads();
        auto fused = [&](int bufc, const float4* a_reg, const u32x4* bh_reg, const u32x4* bl_reg) {
            store_tile(bufc ^ 1, a_reg, bh_reg, bl_reg);
            mma_tile(bufc);
#pragma unroll
            for (int g = 0; g < MT * NT * 6; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // three VALU (conversion)
                if (g & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // an LDS write
                else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // an LDS read (next fragments)
            }
        };
        int kt = 0;
        while (kt + 3 < n) {
            load_tile(a_reg0, bh_reg0, bl_reg0);      // tile kt+2 -> set 0
            fused(0, a_reg1, bh_reg1, bl_reg1);       // store tile kt+1 (set 1) into buffer 1, multiply buffer 0
            __syncthreads();
            load_tile(a_reg1, bh_reg1, bl_reg1);      // tile kt+3 -> set 1
            fused(1, a_reg0, bh_reg0, bl_reg0);       // store tile kt+2 (set 0) into buffer 0, multiply buffer 1
            __syncthreads();
            kt += 2;
        }
        for (; kt < n; kt += 2) {  // tail: same order with the loads / stores guarded
            if (kt + 2 < n) load_tile(a_reg0, bh_reg0, bl_reg0);
            if (kt + 1 < n) store_tile(1, a_reg1, bh_reg1, bl_reg1);
            mma_tile(0);
            __syncthreads();
            if (kt + 1 >= n) break;
            if (kt + 3 < n) load_tile(a_reg1, bh_reg1, bl_reg1);
            if (kt + 2 < n) store_tile(0, a_reg0, bh_reg0, bl_reg0);
            mma_tile(1);
            __syncthreads();
        }
    }
    if (p.splitk > 1) igemm_store_partial<32, MT, NT, WM, WN>(p, acc, tm, tn, wm, wn, lane, blockIdx.y);
    else igemm_epilogue<32, MT, NT, WM, WN>(p, acc, tm, tn, wm, wn, lane, ainv);
}

// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined K loop (non-DCN layers).  Same tiles, LDS layout, loaders and epilogue as igemm16_kernel, but the
// order in which a wave issues its work is written out by hand and pinned with sched_barrier, because the wave is
// in-order and the MFMA pipe is only busy while something else is NOT making the wave wait:
//
//   iteration t (tile t in LDS buffer t&1; fragment sets F0 = k-step 0, F1 = k-step 1; one register set G holds the
//   global data of tile t+1):
//     phase 1   MFMAs of (t, k-step 0) out of F0, and in their shadow:  ds_read F1 <- (t, k-step 1);
//               convert + ds_write tile t+1 (G) into buffer (t+1)&1
//     barrier   tile t+1 complete in LDS; every wave has read all of tile t  ->  buffer t&1 is free
//     phase 2   MFMAs of (t, k-step 1) out of F1, and in their shadow:  buffer_load tile t+2 -> G;
//               ds_read F0 <- (t+1, k-step 0)
//   so every ds_read is issued >= 3*MT*NT/2 MFMAs before its first use, every global load half a tile (>= 12 MFMAs
//   of this wave plus the co-resident wave's, > an L2 hit) before its conversion, the conversion VALU / LDS stores
//   sit between MFMAs instead of after them, and there is one barrier per tile.  Tiles past the end of K are loaded with out-of-range offsets (zeros, no memory traffic) and stored
//   into a buffer nobody reads, which keeps the loop body branch-free.
// ---------------------------------------------------------------------------------------------------------------
// FUSE (128x128 tiles only): fused prediction head.  The main MFMAs run with swapped operands, so a wave's accumulators
// hold hidden^T -- rows = 32 hidden channels of a fragment spread over (register, lane half), columns = 32 pixels over
// the lanes.  That is exactly the B-operand shape of a second MFMA whose k runs over hidden channels: 8 consecutive
// registers of a lane are 8 k-values of its pixel.  After bias + ReLU the accumulators are split to binary16 hi/lo in
// registers and multiplied by the 1x1 weights (A operand, pre-packed in the matching channel order) into a
// [32 final channels][pixels] accumulator -- no LDS transpose, no hidden tensor in HBM.  The two waves of a pixel
// range (hidden-channel halves) are summed through LDS, the N-tiles of the hidden dimension through fuse_out slices.
// GNIN (1x1 layers whose input is a GroupNorm'd tensor, dlav1 heads): the A loader applies the pre-folded
// normalisation + ReLU y = max(a*x + d, 0) (a, d per image and channel) before the hi/lo split; an output tile lies in one image.
// GRU (128x96 tiles: MT 1, NT 3, 4x1 waves): fused ConvGRU gate epilogue, see ConvParams::gru_x3.  Fragment j of a
// wave's accumulators is gate j (r, z, n) of the same 32 channels, so a lane holds all three pre-activations of its
// (pixel, channel) pairs and the gate arithmetic needs no exchange.
template <int MT, int NT, int WM, int WN, bool MULTISRC, bool FUSE = false, bool GNIN = false, bool GRU = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void igemm16p_kernel(const ConvParams p, const int tiles_m, const int tiles_n) {
    static_assert(!FUSE || (MT == 2 && NT == 2 && WM == 2 && WN == 2 && !MULTISRC), "fused head: 128x128 tiles");
    static_assert(!GRU || (MT == 1 && NT == 3 && WM == 4 && WN == 1 && !MULTISRC && !FUSE), "GRU: 128x96 tiles");
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    constexpr int NTH = WM * WN * 64;  // 4 waves (256 threads), or 8 for the high-occupancy 128x128 variant
    constexpr int RPP = NTH / 8;       // A-tile rows covered by one pass of the block (8 float4 per 32-wide row)
    constexpr int A_SLOTS = BM * BK16 / 4 / NTH;
    constexpr int B_CHUNKS = BN * BK16 * 2 / 16;
    constexpr int B_SLOTS = (B_CHUNKS + NTH - 1) / NTH;
    constexpr bool B_PART = B_CHUNKS % NTH != 0;
    constexpr int A_SZ = BM * LDH, B_SZ = BN * LDH;
    constexpr int BUF = 2 * A_SZ + 2 * B_SZ;
    constexpr int NM = 3 * MT * NT;        // MFMAs per phase
    constexpr int NR = 2 * MT + 2 * NT;    // fragment reads per phase
    // one LDS object, the two tile buffers are selected by a run-time offset: the loop body exists once, so the
    // accumulators / fragments / staging registers keep their allocation across the back edge (the order of the LDS
    // reads and writes is pinned by hand below, no alias analysis needed)
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int tile = tile_of_block(tiles_m, tiles_n);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int M = p.B * p.Ho * p.Wo;
    PixelDecomp pdec;
    pdec.init(p.Ho, p.Wo, M);

    const int k4 = tid & 7;
    int a_pix0[A_SLOTS];
    unsigned a_vmask[A_SLOTS], a_byte0[A_SLOTS];
#pragma unroll
    for (int j = 0; j < A_SLOTS; ++j) {
        const int m = tm * BM + (tid >> 3) + j * RPP;
        const bool ok = m < M;
        const int mm = ok ? m : 0;
        int b, ho, wo;
        pdec.split(mm, &b, &ho, &wo);
        const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
        a_pix0[j] = (b * p.H + h0) * p.W + w0;
        a_byte0[j] = (unsigned)(a_pix0[j] * p.src_c[0] + k4 * 4) * 4u;
        a_vmask[j] = ok ? tap_valid_mask(h0, w0, p.H, p.W, p.KH, p.KW) : 0u;
    }

    const int nk = p.Kpad16 / BK16;
    int kt0, kt1;
    splitk_range(p, nk, &kt0, &kt1);
    const int n = kt1 - kt0;
    int u_tap = 0, u_kh = 0, u_kw = 0, u_c0 = 0, u_src = 0, u_cs = 0;
    if (kt0 > 0) {
        const int k0 = kt0 * BK16;
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

    const unsigned w_bytes = (unsigned)((size_t)p.CoutPad * p.Kpad16 * 2);
    const __amdgpu_buffer_rsrc_t r_wh = make_rsrc(p.w16_hi, w_bytes), r_wl = make_rsrc(p.w16_lo, w_bytes);
    const unsigned img_px = (unsigned)p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t r_s0 = make_rsrc(p.src[0], img_px * p.src_c[0] * 4u);
    const __amdgpu_buffer_rsrc_t r_s1 = make_rsrc(MULTISRC ? p.src[1] : p.src[0], MULTISRC ? img_px * p.src_c[1] * 4u : 0u);
    const __amdgpu_buffer_rsrc_t r_s2 = make_rsrc(MULTISRC && p.nsrc > 2 ? p.src[2] : p.src[0], MULTISRC && p.nsrc > 2 ? img_px * p.src_c[2] * 4u : 0u);
    const __amdgpu_buffer_rsrc_t r_s3 = make_rsrc(MULTISRC && p.nsrc > 3 ? p.src[3] : p.src[0], MULTISRC && p.nsrc > 3 ? img_px * p.src_c[3] * 4u : 0u);
    unsigned b_off[B_SLOTS];
#pragma unroll
    for (int j = 0; j < B_SLOTS; ++j) {
        const int f = tid + j * NTH;
        b_off[j] = (!B_PART || f < B_CHUNKS)
                       ? (unsigned)(((size_t)(tn * BN + f / 4) * p.Kpad16 + (f % 4) * 8 + (size_t)kt0 * BK16) * 2)
                       : OOB;
    }
    int b_soff = 0;      // bytes the weight loads have advanced along K (wave-uniform -> the instruction's soffset)
    int tiles_left = n;  // tiles not yet issued; past the end every load goes out of range

    // ---- loader pieces: A slot j / B slot j of the tile the K walk points at, then the walk step ----
    float afwd = 1.f, ainv = 1.f;  // power-of-two pre-scale of the activations and its inverse (ConvParams::in_amax)
    const AmaxRaw amax_raw = conv_in_scale_issue(p);  // reduced behind the first tile's loads (prologue), except GNIN
    if (GNIN) conv_in_scale_finish(p, amax_raw, &afwd, &ainv);
    float4 gn_a4 = make_float4(1.f, 1.f, 1.f, 1.f), gn_d4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int gn_b = GNIN ? (tm * BM) / (p.Ho * p.Wo) : 0;  // the tile's image (the launcher guarantees Ho*Wo % BM == 0)
    auto issue_a = [&](float4(&ga)[A_SLOTS], int j) {
        if (GNIN && j == 0) {  // affine of the 4 channels this lane converts, for the K tile being issued
            const int c = (tiles_left > 0 ? u_cs : 0) + k4 * 4;
            gn_a4 = *reinterpret_cast<const float4*>(p.gn_in_a + (size_t)gn_b * p.Cin + c);
            gn_d4 = *reinterpret_cast<const float4*>(p.gn_in_d + (size_t)gn_b * p.Cin + c);
            // relu(a*x + d) * 2^e == relu((a*2^e)*x + d*2^e): the pre-scale rides in the affine
            gn_a4.x *= afwd; gn_a4.y *= afwd; gn_a4.z *= afwd; gn_a4.w *= afwd;
            gn_d4.x *= afwd; gn_d4.y *= afwd; gn_d4.z *= afwd; gn_d4.w *= afwd;
        }
        __amdgpu_buffer_rsrc_t rs = r_s0;
        int sc = p.src_c[0];
        if (MULTISRC) {
            if (u_src == 1) { rs = r_s1; sc = p.src_c[1]; }
            else if (u_src == 2) { rs = r_s2; sc = p.src_c[2]; }
            else if (u_src == 3) { rs = r_s3; sc = p.src_c[3]; }
        }
        const int tap_pix = u_kh * p.W + u_kw;
        const unsigned bit = tiles_left > 0 ? 1u << u_tap : 0u;
        unsigned off;
        if (MULTISRC) off = (unsigned)((a_pix0[j] + tap_pix) * sc + (u_cs + k4 * 4)) * 4u;
        else off = a_byte0[j] + (unsigned)(tap_pix * sc + u_cs) * 4u;
        ga[j] = buf_ld4(rs, (a_vmask[j] & bit) ? off : OOB);
    };
    auto issue_b = [&](u32x4(&gbh)[B_SLOTS], u32x4(&gbl)[B_SLOTS], int j) {
        const int so = tiles_left > 0 ? b_soff : 0;
        const unsigned vo = tiles_left > 0 ? b_off[j] : OOB;
        gbh[j] = __builtin_amdgcn_raw_buffer_load_b128(r_wh, (int)vo, so, 0);
        gbl[j] = __builtin_amdgcn_raw_buffer_load_b128(r_wl, (int)vo, so, 0);
    };
    auto advance_k = [&]() {
        --tiles_left;
        b_soff += BK16 * 2;
        u_c0 += BK16;
        u_cs += BK16;
        if (u_c0 >= p.Cin) {
            u_c0 = 0; u_cs = 0; u_src = 0;
            ++u_tap;
            if (++u_kw == p.KW) { u_kw = 0; ++u_kh; }
        } else if (MULTISRC) {
            const int cur = u_src == 0 ? p.src_c[0] : u_src == 1 ? p.src_c[1] : u_src == 2 ? p.src_c[2] : p.src_c[3];
            if (u_cs >= cur) { u_cs = 0; ++u_src; }
        }
    };
    auto issue_tile = [&](float4(&ga)[A_SLOTS], u32x4(&gbh)[B_SLOTS], u32x4(&gbl)[B_SLOTS]) {
        // same order as phase 2 of the loop (the compiler's vmcnt bookkeeping merges both at the loop header)
#pragma unroll
        for (int j = 0; j < A_SLOTS; ++j) issue_a(ga, j);
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) issue_b(gbh, gbl, j);
        advance_k();
    };

    // ---- conversion / LDS store pieces ----
    Split2 cs0[A_SLOTS], cs1[A_SLOTS];
    // piece q of A slot j: 0 split (x, y), 1 split (z, w), 2 store the hi halves, 3 store the lo halves
    auto store_a_piece = [&](int buf, const float4(&ga)[A_SLOTS], int j, int q) {
        _Float16* Ah = lds + buf * BUF;
        _Float16* Al = Ah + A_SZ;
        const int row = (tid >> 3) + j * RPP;
        const int col = ((((k4 >> 1) ^ swz(row)) << 1) | (k4 & 1)) * 4;
        if (q == 0) {
            float x = ga[j].x, y = ga[j].y;
            if (GNIN) { x = fmaxf(fmaf(x, gn_a4.x, gn_d4.x), 0.f); y = fmaxf(fmaf(y, gn_a4.y, gn_d4.y), 0.f); }
            else { x *= afwd; y *= afwd; }
            cs0[j] = split2(x, y);
        } else if (q == 1) {
            float z = ga[j].z, w = ga[j].w;
            if (GNIN) { z = fmaxf(fmaf(z, gn_a4.z, gn_d4.z), 0.f); w = fmaxf(fmaf(w, gn_a4.w, gn_d4.w), 0.f); }
            else { z *= afwd; w *= afwd; }
            cs1[j] = split2(z, w);
        }
        else if (q == 2) *reinterpret_cast<u32x2*>(Ah + row * LDH + col) = u32x2{cs0[j].hi, cs1[j].hi};
        else *reinterpret_cast<u32x2*>(Al + row * LDH + col) = u32x2{cs0[j].lo, cs1[j].lo};
    };
    // piece q of B slot j: 0 hi array, 1 lo array
    auto store_b_piece = [&](int buf, const u32x4(&gbh)[B_SLOTS], const u32x4(&gbl)[B_SLOTS], int j, int q) {
        _Float16* Bh = lds + buf * BUF + 2 * A_SZ;
        _Float16* Bl = Bh + B_SZ;
        const int f = tid + j * NTH;
        if (!B_PART || f < B_CHUNKS) {
            const int nn = f / 4, c = f % 4;
            if (q == 0) *reinterpret_cast<u32x4*>(Bh + nn * LDH + (c ^ swz(nn)) * 8) = gbh[j];
            else *reinterpret_cast<u32x4*>(Bl + nn * LDH + (c ^ swz(nn)) * 8) = gbl[j];
        }
    };
    auto store_all = [&](int buf, const float4(&ga)[A_SLOTS], const u32x4(&gbh)[B_SLOTS], const u32x4(&gbl)[B_SLOTS],
                         int) {
#pragma unroll
        for (int j = 0; j < A_SLOTS; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) store_a_piece(buf, ga, j, q);
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) {
            store_b_piece(buf, gbh, gbl, j, 0);
            store_b_piece(buf, gbh, gbl, j, 1);
        }
    };

    acc_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) acc[i][j][r] = 0.f;

    const int lrow = lane >> 5, lcol = lane & 31;
    const int a_frag = (wm * (MT * 32) + lcol) * LDH;
    const int b_frag = 2 * A_SZ + (wn * (NT * 32) + lcol) * LDH;
    const int co0 = ((0 * 2 + lrow) ^ swz(lcol)) * 8, co1 = ((1 * 2 + lrow) ^ swz(lcol)) * 8;

    // fragment read r of k-step ks: order al[*], bh[*], ah[*], bl[*] = the order the MFMA terms need them
    auto read_frag = [&](int buf, int ks, int r, h8(&ah)[MT], h8(&al)[MT], h8(&bh)[NT], h8(&bl)[NT]) {
        const _Float16* base = lds + buf * BUF;
        const int co = ks ? co1 : co0;
        if (r < MT) al[r] = *reinterpret_cast<const h8*>(base + a_frag + A_SZ + r * 32 * LDH + co);
        else if (r < MT + NT) bh[r - MT] = *reinterpret_cast<const h8*>(base + b_frag + (r - MT) * 32 * LDH + co);
        else if (r < 2 * MT + NT) ah[r - MT - NT] = *reinterpret_cast<const h8*>(base + a_frag + (r - MT - NT) * 32 * LDH + co);
        else bl[r - 2 * MT - NT] = *reinterpret_cast<const h8*>(base + b_frag + B_SZ + (r - 2 * MT - NT) * 32 * LDH + co);
    };
    auto mfma_slot = [&](int s, const h8(&ah)[MT], const h8(&al)[MT], const h8(&bh)[NT], const h8(&bl)[NT]) {
        const int term = s / (MT * NT), idx = s % (MT * NT), i = idx / NT, j = idx % NT;
        if (FUSE) {  // transposed product: rows = output channels, columns = pixels
            if (term == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
            else if (term == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
            return;
        }
        if (term == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
        else if (term == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    };

    float4 ga[A_SLOTS];
    u32x4 gbh[B_SLOTS], gbl[B_SLOTS];
    h8 f0ah[MT], f0al[MT], f0bh[NT], f0bl[NT], f1ah[MT], f1al[MT], f1bh[NT], f1bl[NT];

    // one iteration: `cur` = t & 1; the register set holds tile t+1 (stored in phase 1, refilled with tile t+2 in phase 2)
    // (Four register sets = four K tiles in flight for the batch-1 launches were measured in round 5 and not kept: the K loop
    // of such a launch is a single wave per SIMD issuing ~100 instructions per tile in order, not a memory round trip per
    // tile -- profiles/r05_small_launch_timeline.txt, NOTES.)
    auto iteration = [&](int cur) {
        // ---------------- phase 1 ----------------
        constexpr int P1 = NR + 4 * A_SLOTS + 2 * B_SLOTS;
        // the last MFMAs of a phase carry no side work: the LDS queue is then drained when the wave reaches the barrier
        // (phase 1) and the next tile's first fragments have landed when phase 1 starts (phase 2)
        constexpr int NS = NM > 4 ? NM - 2 : NM;
#pragma unroll
        for (int s = 0; s < NM; ++s) {
            mfma_slot(s, f0ah, f0al, f0bh, f0bl);
#pragma unroll
            for (int q = s * P1 / NS; q < (s + 1) * P1 / NS && s < NS; ++q) {
                // the k-step-1 fragment reads and the A conversion / store pieces alternate, the B stores come last
                constexpr int NI = 2 * NR;  // interleaved prefix: read, piece, read, piece, ...
                if (q < NI && (q & 1) == 0) read_frag(cur, 1, q / 2, f1ah, f1al, f1bh, f1bl);
                else {
                    const int z = q < NI ? q / 2 : q - NR;  // index into the store pieces
                    if (z < 4 * A_SLOTS) store_a_piece(cur ^ 1, ga, z / 4, z % 4);
                    else store_b_piece(cur ^ 1, gbh, gbl, (z - 4 * A_SLOTS) / 2, (z - 4 * A_SLOTS) % 2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // ---------------- phase 2 ----------------
        constexpr int NL = A_SLOTS + B_SLOTS + 1;
        constexpr int P2 = NR + NL;
#pragma unroll
        for (int s = 0; s < NM; ++s) {
            mfma_slot(s, f1ah, f1al, f1bh, f1bl);
#pragma unroll
            for (int q = s * P2 / NS; q < (s + 1) * P2 / NS && s < NS; ++q) {
                // loader pieces and the next tile's k-step-0 fragment reads alternate
                constexpr int NI = 2 * (NL < NR ? NL : NR);
                const bool is_load = q < NI ? (q & 1) == 0 : NL > NR;
                const int l = q < NI ? q / 2 : q - NR;   // loader piece index when is_load
                const int r = q < NI ? q / 2 : q - NL;   // read index otherwise
                if (is_load) {
                    if (l < A_SLOTS) issue_a(ga, l);
                    else if (l < A_SLOTS + B_SLOTS) issue_b(gbh, gbl, l - A_SLOTS);
                    else advance_k();
                } else read_frag(cur ^ 1, 0, r, f0ah, f0al, f0bh, f0bl);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: tile 0 in buffer 0, tile 1 in the register set ----
    issue_tile(ga, gbh, gbl);
    if (!GNIN) conv_in_scale_finish(p, amax_raw, &afwd, &ainv);
    store_all(0, ga, gbh, gbl, 2);
    issue_tile(ga, gbh, gbl);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NR; ++r) read_frag(0, 0, r, f0ah, f0al, f0bh, f0bl);

    for (int t = 0; t < n; ++t) iteration(t & 1);
    if constexpr (FUSE) {
        const int g = lane >> 5;
        // 1x1 weight fragments of this wave's 64 hidden channels: [tn][wn][j][s][lane] x 8 halfs
        const u32x4* w2h = reinterpret_cast<const u32x4*>(p.fuse_w2_hi) + (size_t)((tn * WN + wn) * 4) * 64 + lane;
        const u32x4* w2l = reinterpret_cast<const u32x4*>(p.fuse_w2_lo) + (size_t)((tn * WN + wn) * 4) * 64 + lane;
        h8 wh[2][2], wl[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u32x4 a = w2h[(j * 2 + ks) * 64], b = w2l[(j * 2 + ks) * 64];
                wh[j][ks] = *reinterpret_cast<const h8*>(&a);
                wl[j][ks] = *reinterpret_cast<const h8*>(&b);
            }
        acc_t acc2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;
        const bool relu = p.act == CP_ACT_RELU;
        // hidden activations (bias + ReLU applied) in place, and their |max| over the wave: the second product's
        // activation operand gets its own power-of-two pre-scale (hfwd), undone on acc2 before the cross-wave sums
        float hmax = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = tn * BN + wn * 64 + j * 32 + F::row(r, lane);
                const float sc = (p.scale ? p.scale[ch] : 1.f) * ainv, sh = p.shift ? p.shift[ch] : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float x = acc[i][j][r] * sc + sh;
                    if (relu) x = fmaxf(x, 0.f);
                    acc[i][j][r] = x;
                    hmax = fmaxf(hmax, fabsf(x));
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hmax = fmaxf(hmax, __shfl_xor(hmax, o, 64));
        float hfwd, hinv;
        cp_amax_to_scale(__float_as_uint(hmax), &hfwd, &hinv);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    uint32_t hh[4], hl[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = ks * 8 + q * 2;
                        const float x0 = acc[i][j][r] * hfwd, x1 = acc[i][j][r + 1] * hfwd;
                        const Split2 sp = split2(x0, x1);
                        hh[q] = sp.hi;
                        hl[q] = sp.lo;
                    }
                    const u32x4 vh = {hh[0], hh[1], hh[2], hh[3]}, vl = {hl[0], hl[1], hl[2], hl[3]};
                    const h8 bhh = *reinterpret_cast<const h8*>(&vh), bhl = *reinterpret_cast<const h8*>(&vl);
                    acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j][ks], bhh, acc2[i], 0, 0, 0);
                    acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j][ks], bhl, acc2[i], 0, 0, 0);
                    acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j][ks], bhh, acc2[i], 0, 0, 0);
                }
            }
        }
        // sum the two hidden-channel halves (wn = 0 / 1) of each pixel range through LDS, in a fixed order
        __syncthreads();  // every wave is done with the tile buffers
        float* red = reinterpret_cast<float*>(lds);  // [wm][i][r][lane]
        // back to true units: 2^-e of this wave's hidden pre-scale and of the 1x1 weights' per-channel scale
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float wi = (p.fuse_w2_inv ? p.fuse_w2_inv[F::row(r, lane)] : 1.f) * hinv;
            acc2[0][r] *= wi;
            acc2[1][r] *= wi;
        }
        if (wn == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wm * 2 + i) * 16 + r) * 64 + lane] = acc2[i][r];
        }
        __syncthreads();
        if (wn == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = tm * BM + wm * 64 + i * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = F::row(r, lane);
                    const float v = acc2[i][r] + red[((wm * 2 + i) * 16 + r) * 64 + lane];
                    if (c < p.fuse_c2 && m < M) p.fuse_out[((size_t)tn * p.fuse_c2 + c) * M + m] = v;
                }
            }
        }
        (void)g;
        return;
    }
    if constexpr (GRU) {
        // r = sig(x_r + h_r); z = sig(x_z + h_z); n = tanh(x_n + r * h_n); h' = (1 - z) * n + z * h   (convGRU.py:32-39,
        // the same float expressions as gru_gate_kernel)
        const int mbase = __builtin_amdgcn_readfirstlane(tm * BM + wm * 32);
        const int ch = tn * 32 + (lane & 31);
        {   // undo the operand pre-scales: 2^-e_a of the state tensor and 2^-e_w of this lane's three gate rows
            const int row0 = tn * 96 + (lane & 31);
            const float s0 = (p.scale ? p.scale[row0] : 1.f) * ainv, s1 = (p.scale ? p.scale[row0 + 32] : 1.f) * ainv,
                        s2 = (p.scale ? p.scale[row0 + 64] : 1.f) * ainv;
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) { acc[0][0][r] *= s0; acc[0][1][r] *= s1; acc[0][2][r] *= s2; }
        }
        float amax = 0.f;
        if (mbase + 32 <= M) {
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.gru_x3 + (size_t)mbase * 192, 32u * 192u * 4u);
            const __amdgpu_buffer_rsrc_t rh = make_rsrc(p.gru_hprev + (size_t)mbase * 64, 32u * 64u * 4u);
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + (size_t)mbase * 64, 32u * 64u * 4u);
            const int vx = (F::row(0, lane) * 192 + ch) * 4, vh = (F::row(0, lane) * 64 + ch) * 4;
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) {
                const int sx = F::row(r, 0) * 192 * 4, sh = F::row(r, 0) * 64 * 4;
                const float xr = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, sx, 0));
                const float xz = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx + 64 * 4, sx, 0));
                const float xn = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx + 128 * 4, sx, 0));
                const float hp = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rh, vh, sh, 0));
                const float rg = cp_fast_sigmoid(xr + acc[0][0][r]);
                const float zg = cp_fast_sigmoid(xz + acc[0][1][r]);
                const float ng = cp_fast_tanh(xn + rg * acc[0][2][r]);
                const float hv = (1.f - zg) * ng + zg * hp;
                amax = fmaxf(amax, fabsf(hv));
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hv), ro, vh, sh, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) {
                const int m = mbase + F::row(r, lane);
                if (m >= M) continue;
                const float* x3 = p.gru_x3 + (size_t)m * 192 + ch;
                const float hp = p.gru_hprev[(size_t)m * 64 + ch];
                const float rg = cp_fast_sigmoid(x3[0] + acc[0][0][r]);
                const float zg = cp_fast_sigmoid(x3[64] + acc[0][1][r]);
                const float ng = cp_fast_tanh(x3[128] + rg * acc[0][2][r]);
                const float hv = (1.f - zg) * ng + zg * hp;
                amax = fmaxf(amax, fabsf(hv));
                p.out[(size_t)m * 64 + ch] = hv;
            }
        }
        if (p.out_amax) cp_amax_commit(p.out_amax, amax);
        return;
    }
    if (p.splitk > 1) igemm_store_partial<32, MT, NT, WM, WN>(p, acc, tm, tn, wm, wn, lane, blockIdx.y);
    else igemm_epilogue<32, MT, NT, WM, WN>(p, acc, tm, tn, wm, wn, lane, ainv);
}

template <int MT, int NT, int WM, int WN, bool DCN, bool MULTISRC>
int launch16(const ConvParams& p, hipStream_t stream) {
    constexpr bool PF2 = false;  // the DCN loader blends on arrival: one register set
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = p.CoutPad / BN;
    if (p.CoutPad % BN != 0 || p.Kpad16 % BK16 != 0) return CP_ERR_INVALID;
    const dim3 grid(tiles_m * tiles_n, p.splitk > 1 ? p.splitk : 1);
    if constexpr (!DCN) {
        // every non-DCN layer runs the pipelined kernel; igemm16_kernel is only instantiated for the DCN gather layers
        hipLaunchKernelGGL((igemm16p_kernel<MT, NT, WM, WN, MULTISRC>), grid, dim3(NT16), 0, stream, p, tiles_m, tiles_n);
        return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((igemm16_kernel<MT, NT, WM, WN, DCN, MULTISRC, PF2>), grid, dim3(NT16), 0, stream, p, tiles_m,
                       tiles_n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

// 1x1 head weights [C2][Chid] -> MFMA A-operand fragments in the channel order the fused epilogue's accumulators have:
// fragment (tn, wn, j, ks), lane (c = lane % 32, g = lane / 32), element e holds hidden channel
// tn*128 + wn*64 + j*32 + (e & 3) + 8 * (2*ks + (e >> 2)) + 4*g of final channel c (zero for c >= C2).
__global__ void pack_head_w2_kernel(const float* __restrict__ w1, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                    const float* __restrict__ fwd, int C2, int Chid) {
    const int total = Chid * 32;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, frag = idx >> 9;
        const int ks = frag & 1, j = (frag >> 1) & 1, wave = frag >> 2;  // wave = tn * 2 + wn
        const int c = lane & 31, g = lane >> 5;
        const int ch = wave * 64 + j * 32 + (e & 3) + 8 * (2 * ks + (e >> 2)) + 4 * g;
        const float x = c < C2 ? w1[(size_t)c * Chid + ch] * fwd[c] : 0.f;
        const Split2 sp = split2(x, 0.f);
        reinterpret_cast<uint16_t*>(hi)[idx] = (uint16_t)(sp.hi & 0xffffu);
        reinterpret_cast<uint16_t*>(lo)[idx] = (uint16_t)(sp.lo & 0xffffu);
    }
}

// out[b][c][pix] = act(bias[c] + sum over slices (in index order) of slabs[slice][c][b*HW + pix])
__global__ void head_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ bias,
                                   float* __restrict__ out, int slices, int C2, int B, int HW, int sigmoid) {
    const size_t M = (size_t)B * HW, total = M * C2;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / M);
        const size_t m = i - (size_t)c * M;
        float acc = 0.f;
        for (int z = 0; z < slices; ++z) acc += slabs[((size_t)z * C2 + c) * M + m];
        float y = acc + (bias ? bias[c] : 0.f);
        if (sigmoid) y = 1.f / (1.f + expf(-y));
        const size_t b = m / HW, pix = m - b * HW;
        out[(b * C2 + c) * HW + pix] = y;
    }
}

// the same reduction for every head of a grouped launch (blockIdx.y = head)
__global__ void head_reduce_grouped_kernel(const float* __restrict__ slabs, const HeadReduceGroup g, int B, int HW) {
    const int h = blockIdx.y;
    const int C2 = g.c2[h], slices = g.slices;
    const float* __restrict__ sl = slabs + (size_t)g.base[h] * B * HW;
    const float* __restrict__ bias = g.bias[h];
    float* __restrict__ out = g.out[h];
    const bool sigmoid = g.sigmoid[h] != 0;
    const size_t M = (size_t)B * HW, total = M * C2;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / M);
        const size_t m = i - (size_t)c * M;
        float acc = 0.f;
        for (int z = 0; z < slices; ++z) acc += sl[((size_t)z * C2 + c) * M + m];
        float y = acc + (bias ? bias[c] : 0.f);
        if (sigmoid) y = 1.f / (1.f + expf(-y));
        const size_t b = m / HW, pix = m - b * HW;
        out[(b * C2 + c) * HW + pix] = y;
    }
}

// pack PyTorch [Cout][Cin][taps] float32 weights into split binary16 [CoutPad][Kpad16] (k = tap*Cin + ci)
__global__ void pack_weight16_kernel(const float* __restrict__ w, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                     int Cout, int Cin, int taps, int Kpad16, int coff, const float* __restrict__ fwd) {
    const size_t total = (size_t)Cout * Cin * taps;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % taps);
        const size_t r = i / taps;
        const int ci = (int)(r % Cin), co = (int)(r / Cin);
        const float x = w[i] * (fwd ? fwd[coff + co] : 1.f);
        const Split2 s = split2(x, 0.f);
        const size_t o = (size_t)(coff + co) * Kpad16 + (size_t)t * Cin + ci;
        reinterpret_cast<uint16_t*>(hi)[o] = (uint16_t)(s.hi & 0xffffu);
        reinterpret_cast<uint16_t*>(lo)[o] = (uint16_t)(s.lo & 0xffffu);
    }
}

// one 64-lane block per output channel: (2^e, 2^-e) with max|w[co,:]| * 2^e in [2^14, 2^15)
__global__ void weight_scale_kernel(const float* __restrict__ w, int per, float* __restrict__ fwd, float* __restrict__ inv) {
    const int co = blockIdx.x;
    float m = 0.f;
    for (int i = threadIdx.x; i < per; i += 64) m = fmaxf(m, fabsf(w[(size_t)co * per + i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (threadIdx.x == 0) {
        float f = 1.f, iv = 1.f;
        if (m > 0.f) cp_amax_to_scale(__float_as_uint(m), &f, &iv);
        fwd[co] = f;
        inv[co] = iv;
    }
}

__global__ void scale16_kernel(const float* __restrict__ scale, const float* __restrict__ inv, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (scale ? scale[i] : 1.f) * inv[i];
}

__global__ void absmax_kernel(const float4* __restrict__ x, size_t n4, unsigned* __restrict__ slot) {
    float m = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    cp_amax_commit(slot, m);
}

}  // namespace

int cp_launch_weight_scale(const float* w, int Cout, int per, float* fwd, float* inv, hipStream_t s) {
    hipLaunchKernelGGL(weight_scale_kernel, dim3(Cout), dim3(64), 0, s, w, per, fwd, inv);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

int cp_launch_scale16(const float* scale, const float* inv, float* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(scale16_kernel, dim3((n + 255) / 256), dim3(256), 0, s, scale, inv, out, n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

int cp_launch_absmax(const float* x, size_t n, unsigned* slot, hipStream_t s) {
    if (n % 4 || ((uintptr_t)x & 15)) return CP_ERR_INVALID;
    size_t g = (n / 4 + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)g), dim3(256), 0, s, (const float4*)x, n / 4, slot);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

// N tile of the f16x3 launch: cp_conv_tile_n, except that <= 16-wide outputs whose weights were packed to 32 columns
// (the GroupNorm'd final 1x1 heads) take the 32-wide tile
static int conv16_tile_n(const ConvParams& p) {
    const int bn = cp_conv_tile_n(p.Cout);
    if (p.tile_n && p.tile_n < bn && p.CoutPad % p.tile_n == 0 && (!p.offmask || p.tile_n == 64)) return p.tile_n;
    return (bn < 32 && p.CoutPad >= 32 && p.CoutPad % 32 == 0) ? 32 : bn;
}

// DCNv2 layers whose launch fills the chip gather from an LDS-staged halo (dcn16p.hip); small launches keep dcn16.hip
// (split-K).  cp_set_debug: 32768 = never, 65536 = every eligible layer (tests, A/B runs).
static bool dcn16p_wanted(const ConvParams& p) {
    if ((p.dbg & 32768) || (p.dbg & 1024) || !cp_dcn16p_supported(p)) return false;
    // from 64 blocks up (round 3; 256 before): at batch 1 the 128 x 128 and 64 x 64 maps are 128 / 64-block launches, and one
    // patch-resident launch beats the split-K gather kernel + its epilogue launch (frame 1.666 -> 1.627 ms, 2.089 -> 2.044;
    // thresholds 128 / 32 / none: 1.639 / 1.643 / 1.637)
    return (p.dbg & 65536) || cp_dcn16p_blocks(p) >= 64;
}

// ... and, when every resident workgroup gets many (patch, N tile) items, from the persistent streamed form of the same
// gather (dcn16s.hip): measured per layer shape at B = 64 (tools/dcn_ab.py, profiles/r04_dcn_ab.txt) it is ahead from 8 items
// per workgroup (64 -> 64 @ 128 x 128: -3 %, 128 -> 128 @ 64 x 64: -3 %) and behind below that (2 - 4 items: +3 ... +6 %).
// cp_set_debug: 1048576 = never, 2097152 = every eligible launch (tests, A/B runs).
static bool dcn16s_wanted(const ConvParams& p) {
    if ((p.dbg & 1048576) || !dcn16p_wanted(p) || !cp_dcn16s_supported(p)) return false;
    if (cp_dcn16p_wide(p) && !(p.dbg & 2097152)) return false;  // 128 output channels per tile: the 128-wide patch kernel (round 5)
    return (p.dbg & 2097152) || cp_dcn16s_items(p) >= 4096;
}

// ... and the layers without whole 128-channel tiles (Cout = 64: no wider tile to share a blend) from the three-workgroups-per-CU
// form (dcn16t.hip) once the launch has more workgroups than the chip holds at three per CU: B = 64, same box, alternating
// launches (profiles/r06_dcn16t_ab.txt): 64 -> 64 @128^2 (8192 workgroups) 393 (dcn16s) / 414 (dcn16p) -> 360 us, 128 -> 64 @64^2
// (2048) 183 -> 169, 256 -> 64 @32^2 (512) 87 -> 91 (stays on dcn16p).
// cp_set_debug: 33554432 = every eligible launch, 67108864 = never (tests, A/B runs).
static bool dcn16t_wanted(const ConvParams& p) {
    if ((p.dbg & 67108864) || !dcn16p_wanted(p) || !cp_dcn16t_supported(p)) return false;
    if (p.dbg & 33554432) return true;
    if (p.dbg & 2097152) return false;  // (dcn16s asked for by name)
    return !cp_dcn16p_wide(p) && cp_dcn16p_blocks(p) >= 1024;
}

// 64 -> <= 32 channel 3x3 layers (DCN offset / mask convolutions) with at least one (strip, band) job per wave slot of the chip: the
// row-streaming kernel of strm16.hip.  cp_set_debug: 268435456 = never, 536870912 = every eligible layer (tests).
static bool strm16_wanted(const ConvParams& p) {
    if ((p.dbg & 268435456) || (p.dbg & 4096) || !cp_strm16_supported(p)) return false;
    return (p.dbg & 536870912) || cp_strm16_jobs(p) >= 1024;
}

static bool halo16_wanted(const ConvParams& p, int bn) {
    if ((p.dbg & 4096) || p.gn_in_a || !cp_halo16_supported(p)) return false;
    // with the weight fragments coming straight from L2 (no barrier inside a chunk) the halo kernel beats the per-tap
    // implicit GEMM on every N tile; with an LDS weight tile only on the 32-wide one (8192: everywhere anyway, A/B runs)
    return bn == 32 || (p.w16f_hi && p.w16f_lo && !(p.dbg & 16384)) || (p.dbg & 8192);
}

// 1x1 layers: the register-only stream of pw16.hip (cp_set_debug 4194304: the LDS-staged loop instead, A/B runs)
static bool pw16_wanted(const ConvParams& p) { return !(p.dbg & 4194304) && cp_pw16_supported(p); }

bool cp_conv16_supported(const ConvParams& p) {
    if (!p.w16_hi || !p.w16_lo || p.Cin % BK16 != 0 || p.KH * p.KW > 32) return false;
    for (int s = 0; s < p.nsrc; ++s)
        if (p.src_c[s] % BK16 != 0) return false;
    const int bn = conv16_tile_n(p);
    if (bn < 32) return false;
    if (p.gn_in_mr) return false;  // the per-group form belongs to the exact-f32 loader
    if (p.gn_in_a || p.gn_in_d) {
        if (!p.gn_in_a || !p.gn_in_d || bn != 32 || p.KH != 1 || p.KW != 1 || p.pad != 0 || p.stride != 1 || p.nsrc != 1 ||
            p.offmask || (p.Ho * p.Wo) % 128 != 0)
            return false;
    }
    if (p.offmask && bn < 64) return false;
    // 32-bit byte offsets of the buffer loads
    for (int s = 0; s < p.nsrc; ++s)
        if ((size_t)p.B * p.H * p.W * p.src_c[s] * 4 >= (size_t)0xf0000000u) return false;
    if ((size_t)p.CoutPad * p.Kpad16 * 2 >= ((size_t)1 << 32)) return false;
    return true;
}

int cp_launch_conv16(const ConvParams& p, hipStream_t stream) {
    if (!cp_conv16_supported(p)) return CP_ERR_INVALID;
    const int bn = conv16_tile_n(p);
    const bool cat = p.nsrc > 1;
    if (p.gn_in_a) {
        const int M = p.B * p.Ho * p.Wo;
        const int tiles_m = (M + 127) / 128, tiles_n = p.CoutPad / 32;
        hipLaunchKernelGGL((igemm16p_kernel<1, 1, 4, 1, false, false, true>),
                           dim3(tiles_m * tiles_n, p.splitk > 1 ? p.splitk : 1), dim3(NT16), 0, stream, p, tiles_m, tiles_n);
        return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
    }
    if (p.offmask) {
        if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.nsrc != 1 || p.H != p.Ho || p.W != p.Wo)
            return CP_ERR_INVALID;
        // dcn16.hip: the software-pipelined gather kernel; cp_set_debug(1024) keeps the previous un-pipelined loop
        // (igemm16_kernel<DCN>) for A/B runs, 2048 selects the other wave count of the new kernel
        if (dcn16t_wanted(p)) return cp_launch_dcn16t(p, stream);
        if (dcn16s_wanted(p)) return cp_launch_dcn16s(p, stream);
        if (dcn16p_wanted(p)) return cp_launch_dcn16p(p, stream);
        if (!(p.dbg & 1024)) return cp_launch_dcn16(p, bn, (p.dbg & 2048) ? 1 : 0, stream);
        return bn == 128 ? launch16<2, 2, 2, 2, true, false>(p, stream) : launch16<2, 1, 2, 2, true, false>(p, stream);
    }
    // 3x3 / stride 1 layers with full 8x16 patches can run on the halo-resident kernel (halo16.hip).  Measured on the
    // dlav1_34 B=32 step (profiles/r02_halo_ab.txt): N tile 32 (conv_offset_mask) 62 -> 103 TFLOP/s, N 64 223 -> 212,
    // N 128 288 -> 290: with 64+ output channels the loop is bound by MFMA issue + fragment reads at the sustained
    // clock, not by the A-side loads / conversion the halo removes, so only the 32-wide tile uses it by default.
    // cp_set_debug: 4096 = never, 8192 = every eligible layer (A/B runs).
    // small launches on 64 x 64 tiles (ConvParams::tile_m): the LDS-staged loop, whatever the layer shape
    if (bn == 64 && p.tile_m == 64) return cat ? launch16<1, 1, 2, 2, false, true>(p, stream) : launch16<1, 1, 2, 2, false, false>(p, stream);
    if (bn == 32 && strm16_wanted(p)) return cp_launch_strm16(p, stream);
    if (halo16_wanted(p, bn)) return cp_launch_halo16(p, bn, stream);
    if (pw16_wanted(p)) return cp_launch_pw16(p, stream);
    if (bn == 128) return cat ? launch16<2, 2, 2, 2, false, true>(p, stream) : launch16<2, 2, 2, 2, false, false>(p, stream);
    if (bn == 64) return cat ? launch16<2, 1, 2, 2, false, true>(p, stream) : launch16<2, 1, 2, 2, false, false>(p, stream);
    return cat ? launch16<1, 1, 4, 1, false, true>(p, stream) : launch16<1, 1, 4, 1, false, false>(p, stream);
}

// kernel-variant ids continue after the exact-f32 ones (cp_conv_variant): 14.. = split-f16 instantiations
int cp_conv16_variant(const ConvParams& p) {
    const int bn = conv16_tile_n(p);
    if (p.offmask) return dcn16t_wanted(p) ? CP_VARIANT_DCN16T : dcn16s_wanted(p) ? CP_VARIANT_DCN16S : dcn16p_wanted(p) ? (cp_dcn16p_wide(p) ? CP_VARIANT_DCN16PW : CP_VARIANT_DCN16P) : bn == 128 ? 18 : 17;
    const int t = bn == 32 ? 0 : bn == 64 ? 1 : 2;
    if (bn == 64 && p.tile_m == 64) return CP_VARIANT_M64N64;
    if (bn == 32 && strm16_wanted(p)) return CP_VARIANT_STRM16;
    if (halo16_wanted(p, bn)) return 27 + t;
    if (pw16_wanted(p)) return CP_VARIANT_PW16 + (p.CoutPad % 128 == 0 ? 1 : 0);
    return (p.nsrc > 1 ? 19 : 14) + t;
}

int cp_launch_pack_weight16(const float* w, void* hi, void* lo, int Cout, int Cin, int taps, int Kpad16, int coff,
                            const float* fwd, hipStream_t s) {
    const size_t n = (size_t)Cout * Cin * taps;
    int g = (int)((n + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_weight16_kernel, dim3(g), dim3(256), 0, s, w, (_Float16*)hi, (_Float16*)lo, Cout, Cin, taps,
                       Kpad16, coff, fwd);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

bool cp_head_fuse_supported(const ConvParams& p, int c2) {
    return cp_conv16_supported(p) && !p.offmask && p.nsrc == 1 && p.CoutPad % 128 == 0 && p.Cout == p.CoutPad &&
           c2 >= 1 && c2 <= 32 && !p.res && !p.gn_stats && !p.gn_in_mr &&
           (p.act == CP_ACT_RELU || p.act == CP_ACT_NONE);
}

int cp_launch_conv16_fused_head(const ConvParams& p, hipStream_t stream) {
    if (!cp_head_fuse_supported(p, p.fuse_c2) || !p.fuse_w2_hi || !p.fuse_w2_lo || !p.fuse_out || p.splitk > 1)
        return CP_ERR_INVALID;
    if (cp_halo16_fused_head_supported(p)) return cp_launch_halo16_fused_head(p, stream);  // halo16.hip
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_m = (M + 127) / 128, tiles_n = p.CoutPad / 128;
    hipLaunchKernelGGL((igemm16p_kernel<2, 2, 2, 2, false, true>), dim3(tiles_m * tiles_n), dim3(NT16), 0, stream, p,
                       tiles_m, tiles_n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

int cp_launch_pack_head_w2(const float* w1, void* hi, void* lo, float* w2_inv, int C2, int Chid, hipStream_t s) {
    if (Chid % 128 != 0 || C2 < 1 || C2 > 32 || !w2_inv) return CP_ERR_INVALID;
    // w2_inv doubles as [32 inv | 32 fwd]: per-final-channel power-of-two scale of the 1x1 rows (pad rows: 1)
    float* fwd = w2_inv + 32;
    if (hipMemsetAsync(w2_inv, 0, 64 * sizeof(float), s) != hipSuccess) return CP_ERR_LAUNCH;
    hipLaunchKernelGGL(weight_scale_kernel, dim3(C2), dim3(64), 0, s, w1, Chid, fwd, w2_inv);
    hipLaunchKernelGGL(pack_head_w2_kernel, dim3((Chid * 32 + 255) / 256), dim3(256), 0, s, w1, (_Float16*)hi,
                       (_Float16*)lo, (const float*)fwd, C2, Chid);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

int cp_launch_head_reduce(const float* slabs, const float* bias, float* out_nchw, int slices, int C2, int B, int HW,
                          int sigmoid, hipStream_t s) {
    const size_t n = (size_t)B * HW * C2;
    size_t g = (n + 255) / 256;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(head_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, slabs, bias, out_nchw, slices, C2, B, HW,
                       sigmoid);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

int cp_launch_head_reduce_grouped(const float* slabs, const HeadReduceGroup& g, int B, int HW, hipStream_t s) {
    if (g.n < 1 || g.n > CP_MAX_HEAD_GROUP) return CP_ERR_INVALID;
    int cmax = 1;
    for (int h = 0; h < g.n; ++h) cmax = g.c2[h] > cmax ? g.c2[h] : cmax;
    size_t gx = ((size_t)B * HW * cmax + 255) / 256;
    if (gx > 16384) gx = 16384;
    hipLaunchKernelGGL(head_reduce_grouped_kernel, dim3((unsigned)gx, g.n), dim3(256), 0, s, slabs, g, B, HW);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}


// hidden-side ConvGRU convolution (64 -> [r|z|n] x 64, 3x3) with the gate arithmetic fused (ConvParams::gru_x3)
int cp_launch_conv16_gru(const ConvParams& p, hipStream_t stream) {
    if (!p.w16_hi || !p.w16_lo || !p.gru_x3 || !p.gru_hprev || !p.out || p.Cin != 64 || p.nsrc != 1 || p.src_c[0] != 64 ||
        p.CoutPad != 192 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.Kpad16 != 576 || p.splitk > 1 ||
        (size_t)p.B * p.H * p.W * 192 * 4 >= (size_t)0xf0000000u)
        return CP_ERR_INVALID;
    if (cp_halo16_gru_supported(p)) return cp_launch_halo16_gru(p, stream);  // halo16.hip
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_m = (M + 127) / 128, tiles_n = 2;
    hipLaunchKernelGGL((igemm16p_kernel<1, 3, 4, 1, false, false, false, true>), dim3(tiles_m * tiles_n), dim3(NT16), 0,
                       stream, p, tiles_m, tiles_n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}
