// Direct convolution for the full-resolution, low-channel layers at the top of DLA-34 (pose_dla_dcn.py:268-283:
// base_layer 7x7 3->16, level0 3x3 16->16, level1 3x3/2 16->32, each + BatchNorm + ReLU) in the split-f16 ("f16x3")
// precision mode.  These three layers hold 2 % of the network's FLOPs but touch its largest activations
// (B x 512 x 512 x 16 float32 = 537 MB at batch 32), so they are HBM-bound by nature; the generic implicit-GEMM
// kernels ran them at 8x / 3x / 1.8x their traffic floor because with 16 output channels every A element feeds only one
// MFMA column block and the per-element loader work (index math, float32 -> hi/lo split) dominates.
//
// Here a block stages its input tile (with halo) ONCE into LDS, already split into binary16 hi / lo planes in pixel-
// major order [row][col][CIN], so each input element is converted once instead of once per tap.  The A operand of
// v_mfma_f32_16x16x32_f16 (16 pixels x 32 k) is then read straight out of that image: 8 consecutive k of a lane are
// 8 consecutive halfs of the LDS image (two neighbouring pixels x 4 channels for the stem, 8 channels of one tap for
// the 16-channel layers), i.e. the im2col matrix is never formed.  Weights live in registers as B fragments for the
// whole kernel.  Products are hi*hi + hi*lo + lo*hi with float32 accumulation, as in igemm16.hip.
//
// K layout (what pack_lowc_weights mirrors):
//   CIN == 4  (stem; input planes 0..2 real, channel 3 zero):  k-step s = kh, lane chunk q = lane / 16 covers kernel
//             columns 2q, 2q+1 (x 4 channels); column 7 is padding (zero weights).  K = 7 x 32.  The 8-plane stem
//             (pre_hm_hp_layer, pose_dla_dcn.py:262-265) is NG = 2 such groups of 4 planes: two LDS images, K = 14 x 32.
//   CIN == 16 (3x3):  k-step s covers taps 2s, 2s+1 (tap = kh*3 + kw; tap 9 is padding), chunk q -> tap 2s + q/2,
//             channels 8*(q%2) .. +7.  K = 5 x 32.
#include "cp_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#ifdef CP_LOWC_STAMP
// tuning build: shader-clock stamps of wave 0 of one mid-launch block per kernel kind (tools/lowc_timeline.py)
__device__ unsigned long long g_lowc_clk[4][16];
#define LOWC_STAMP(i) do { if (blockIdx.x == gridDim.x / 2 + 1 && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"); \
                                g_lowc_clk[(CIN == 4 ? 0 : S) + (NG - 1) * 3][i] = clock64(); } } while (0)
extern "C" int cp_debug_read_lowc_clk(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lowc_clk), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#else
#define LOWC_STAMP(i) do { } while (0)
#endif

namespace {

struct LowcParams {
    const float* in;   // NCHW_IN: [B][planes][H][W]; else NHWC [B][H][W][CIN]
    float* out;        // NHWC [B][Ho][Wo][COUT]
    const void* w_hi;  // B fragments [KSTEPS][COUT/16][64 lanes][8 halfs]
    const void* w_lo;
    const float* scale;  // folded BatchNorm [COUT] x 2^-e_w of the per-channel weight pre-scale
    const float* shift;
    const unsigned* in_amax;  // running |max| of the input (ConvParams::in_amax), nullptr = no activation pre-scale
    unsigned* out_amax;       // receives the output's |max|, nullptr = not tracked
    int B, H, W, Ho, Wo, planes, pad;
};

__device__ __forceinline__ void split2(float a, float b, uint32_t* hi, uint32_t* lo) {
    fp16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
    *hi = *reinterpret_cast<uint32_t*>(&h);
    // a - float(hi) straight from the packed halves, rounded into the two halves of lo (see igemm16_common.h: split2)
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "v"(*hi));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(*hi));
    *lo = l;
}

template <int CIN, int KS, int S, int COUT, int TW, int TH, bool NCHW_IN, int NG = 1>
__global__ __launch_bounds__(256) void lowc_kernel(const LowcParams p) {
    static_assert(CIN == 4 || CIN == 16, "channel layouts");
    static_assert(NG == 1 || (CIN == 4 && NCHW_IN), "plane groups: NCHW stems only");
    constexpr int KSTEPS1 = CIN == 4 ? KS : (KS * KS * CIN + 31) / 32;  // per plane group
    constexpr int KSTEPS = NG * KSTEPS1;
    constexpr int NF = COUT / 16;
    constexpr int XF = TW / 16;                // x fragments per tile row
    constexpr int IH = (TH - 1) * S + KS + 1;  // + 1: the padding tap / column reads finite data, never out of bounds
    constexpr int IW = (TW - 1) * S + KS + 1;
    constexpr int PLANE = IH * IW * CIN;       // halfs per LDS plane
    __shared__ __attribute__((aligned(16))) _Float16 img_hi[NG * PLANE];
    __shared__ __attribute__((aligned(16))) _Float16 img_lo[NG * PLANE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;

    LOWC_STAMP(0);
    // ---- weights -> registers (B operand: lane = (n = lane % 16, k chunk = lane / 16)) ----
    h8 wh[KSTEPS][NF], wl[KSTEPS][NF];
    {
        const u32x4* gh = reinterpret_cast<const u32x4*>(p.w_hi) + lane;
        const u32x4* gl = reinterpret_cast<const u32x4*>(p.w_lo) + lane;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const u32x4 a = gh[(s * NF + f) * 64], c = gl[(s * NF + f) * 64];
                wh[s][f] = *reinterpret_cast<const h8*>(&a);
                wl[s][f] = *reinterpret_cast<const h8*>(&c);
            }
    }

    LOWC_STAMP(1);  // weights arrived
    float afwd = 1.f, ainv = 1.f;
    if (p.in_amax) cp_amax_to_scale(cp_amax_read(p.in_amax), &afwd, &ainv);
    LOWC_STAMP(2);  // activation scale arrived
    // ---- stage the input tile: float32 global -> binary16 hi / lo image in LDS, zero outside the picture ----
    // All loads of a round are issued before the first conversion (one HBM round trip per round of SR slots, not one per
    // slot: the kernels are streams of their input and output, latency is what they have to hide).
    constexpr int SR = 6;
    if (NCHW_IN) {
        const size_t plane_sz = (size_t)p.H * p.W;
        const float* base = p.in + (size_t)b * p.planes * plane_sz;
        // (the 3-plane stem measured faster one slot at a time: 0.218 vs 0.265 ms -- its 537 MB of stores want the waves
        // the extra staging registers cost)
        constexpr int NI = (IH * IW + 255) / 256, SRN = 1;
#pragma unroll 1
        for (int r0 = 0; r0 < NI; r0 += SRN) {
            float v[SRN][NG][4];
#pragma unroll
            for (int k = 0; k < SRN; ++k) {
                const int i = tid + (r0 + k) * 256;
                const int r = i / IW, c = i - r * IW;
                const int iy = iy0 + r, ix = ix0 + c;
#pragma unroll
                for (int g = 0; g < NG; ++g) v[k][g][0] = v[k][g][1] = v[k][g][2] = v[k][g][3] = 0.f;
                if (i < IH * IW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                    const float* q = base + (size_t)iy * p.W + ix;
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4)
                            if (p.planes > 4 * g + c4) v[k][g][c4] = q[(size_t)(4 * g + c4) * plane_sz];
                }
            }
#pragma unroll
            for (int k = 0; k < SRN; ++k) {
                const int i = tid + (r0 + k) * 256;
                if (i >= IH * IW) continue;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    uint32_t h0, l0, h1, l1;
                    split2(v[k][g][0] * afwd, v[k][g][1] * afwd, &h0, &l0);
                    split2(v[k][g][2] * afwd, v[k][g][3] * afwd, &h1, &l1);
                    *reinterpret_cast<u32x2*>(img_hi + g * PLANE + i * 4) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(img_lo + g * PLANE + i * 4) = u32x2{l0, l1};
                }
            }
        }
    } else {
        constexpr int V = CIN / 4;  // float4 per pixel
        const float* base = p.in + (size_t)b * p.H * p.W * CIN;
        constexpr int NI = (IH * IW * V + 255) / 256;
#pragma unroll 1
        for (int r0 = 0; r0 < NI; r0 += SR) {
            float4 x[SR];
#pragma unroll
            for (int k = 0; k < SR; ++k) {
                const int i = tid + (r0 + k) * 256;
                const int px = i / V, v = i - px * V;
                const int r = px / IW, c = px - r * IW;
                const int iy = iy0 + r, ix = ix0 + c;
                x[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < IH * IW * V && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    x[k] = *reinterpret_cast<const float4*>(base + ((size_t)iy * p.W + ix) * CIN + v * 4);
            }
#pragma unroll
            for (int k = 0; k < SR; ++k) {
                const int i = tid + (r0 + k) * 256;
                if (i >= IH * IW * V) continue;
                const int px = i / V, v = i - px * V;
                uint32_t h0, l0, h1, l1;
                split2(x[k].x * afwd, x[k].y * afwd, &h0, &l0);
                split2(x[k].z * afwd, x[k].w * afwd, &h1, &l1);
                *reinterpret_cast<u32x2*>(img_hi + px * CIN + v * 4) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(img_lo + px * CIN + v * 4) = u32x2{l0, l1};
            }
        }
    }
    LOWC_STAMP(3);  // this wave's share of the tile staged
    __syncthreads();
    LOWC_STAMP(4);  // barrier passed

    // ---- multiply: wave w owns x fragment (w % XF) of rows (w / XF), + 4 / XF, ... ----
    const int pl = lane & 15, q = lane >> 4;
    const int xf = wid % XF;
    float sc[NF], sh[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        sc[f] = (p.scale ? p.scale[f * 16 + pl] : 1.f) * ainv;
        sh[f] = p.shift ? p.shift[f * 16 + pl] : 0.f;
    }
    float amax = 0.f;
    for (int row = wid / XF; row < TH; row += 4 / XF) {
        const int oy = oy0 + row;
        if (oy >= p.Ho) break;
        if (row == wid / XF + 4 / XF) LOWC_STAMP(5);  // first row done (MFMAs retired, stores issued and acknowledged)
        f32x4 acc[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int lx = (xf * 16 + pl) * S;  // column of this lane's pixel inside the LDS image (before the tap offset)
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            h8 ah, al;
            if (CIN == 4) {
                const int off = (s / KSTEPS1) * PLANE + ((row * S + s % KSTEPS1) * IW + lx + 2 * q) * 4;  // 8-byte aligned
                const u32x2 h0 = *reinterpret_cast<const u32x2*>(img_hi + off), h1 = *reinterpret_cast<const u32x2*>(img_hi + off + 4);
                const u32x2 l0 = *reinterpret_cast<const u32x2*>(img_lo + off), l1 = *reinterpret_cast<const u32x2*>(img_lo + off + 4);
                const u32x4 hv = {h0.x, h0.y, h1.x, h1.y}, lv = {l0.x, l0.y, l1.x, l1.y};
                ah = *reinterpret_cast<const h8*>(&hv);
                al = *reinterpret_cast<const h8*>(&lv);
            } else {
                const int tap = 2 * s + (q >> 1);
                const int kh = tap / KS, kw = tap - kh * KS;  // tap 9 -> (3, 0): the extra LDS row, zero weights
                const int off = ((row * S + kh) * IW + lx + kw) * CIN + (q & 1) * 8;  // 16-byte aligned
                ah = *reinterpret_cast<const h8*>(img_hi + off);
                al = *reinterpret_cast<const h8*>(img_lo + off);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[s][f], acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[s][f], acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[s][f], acc[f], 0, 0, 0);
            }
        }
        // C layout: column (channel) = lane % 16, rows (pixels) = 4 * (lane / 16) + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ox = ox0 + xf * 16 + 4 * q + r;
            if (ox >= p.Wo) continue;
            float* o = p.out + (((size_t)b * p.Ho + oy) * p.Wo + ox) * COUT + pl;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float y = fmaxf(acc[f][r] * sc[f] + sh[f], 0.f);
                amax = fmaxf(amax, y);
                o[f * 16] = y;
            }
        }
    }
    LOWC_STAMP(6);  // all rows done
    if (p.out_amax) cp_amax_commit(p.out_amax, amax);
    LOWC_STAMP(7);
}

// PyTorch [COUT][cin][KS][KS] float32 -> hi / lo B fragments in the K layout described at the top of the file
// ci0: first input channel of this plane group (CIN == 4 only)
// 3x3 / 16-channel weights in the K order of the fused stem + level0 kernel: K step s = (kernel row s / 2, column group s % 2);
// group 0 = columns 0, 1 (chunk q -> column q / 2, channels 8 (q % 2) .. + 7), group 1 = column 2 in chunks 0, 1 and zeros in
// chunks 2, 3 -- every K step reads ONE row of the source image, so a wave can multiply a staged row into the three output
// rows it belongs to.  6 x 32 instead of 5 x 32 deep.
__global__ void pack_lowc_rows_weights(const float* __restrict__ w, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                       const float* __restrict__ fwd, int cout, int cin) {
    const int nf = cout / 16, total = 6 * nf * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63, sf = idx >> 9;
        const int f = sf % nf, s = sf / nf;
        const int co = f * 16 + (lane & 15), q = lane >> 4;
        const int kh = s >> 1, kw = (s & 1) ? ((q >> 1) == 0 ? 2 : -1) : (q >> 1), ci = (q & 1) * 8 + j;
        float v = 0.f;
        if (kw >= 0 && ci < cin) v = w[(((size_t)co * cin + ci) * 3 + kh) * 3 + kw];
        uint32_t h, l;
        split2(fwd ? v * fwd[co] : v, 0.f, &h, &l);
        hi[idx] = (uint16_t)(h & 0xffffu);
        lo[idx] = (uint16_t)(l & 0xffffu);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused stem (7x7, <= 4 planes -> 16) + level0 (3x3, 16 -> 16), both + BatchNorm + ReLU (pose_dla_dcn.py:268-283, :310-322): the
// 16-channel full-resolution tensor between them (16.8 MB per 512 x 512 image, written once and read once by the two-kernel
// form: 43 % of the first three layers' traffic) never exists.
//
// A WAVE streams down a column strip on its own -- no barrier, no LDS shared between waves, so every wave of the chip is at
// another row and the loads / stores / MFMAs of different waves overlap by themselves:
//   * strip = 14 level0 output columns = one 16-pixel fragment of stem outputs (the 3x3 needs one column either side) = 23
//     input columns; rows [y0, y0 + R) of level0 = stem rows [y0 - 1, y0 + R] = input rows [y0 - 4, y0 + R + 3];
//   * per input row j: the row (3 planes x 23 floats, requested three rows ahead) is split into hi / lo halves in a wave-private
//     LDS row, read back ONCE as the MFMA operand (16 pixels x 8 columns x 4 channels) and multiplied into the SEVEN rolling
//     stem accumulators it belongs to (kernel rows 6 .. 0 of stem rows j - 3 .. j + 3); stem row j - 3 is then complete:
//     BatchNorm + ReLU, zero outside the picture (level0's zero padding), x 2^e, split, one wave-private LDS row, read back as
//     the operands of level0's two K steps per kernel row and multiplied into the THREE rolling level0 accumulators; level0
//     row j - 4 is complete: BatchNorm + ReLU, one 16-byte store per lane (4 consecutive channels of its pixel);
//   * products are computed transposed (weights as the first operand): a lane's accumulator quad = channels 4 q .. 4 q + 3 of
//     pixel lane % 16 -- the layout both the re-split and the stores want;
//   * 39 MFMAs (16 x 16 x 32) per row and wave: 21 for the stem, 18 for level0; LDS traffic 0.5 KB + 3 KB per row.
// The intermediate's power-of-two pre-scale cannot come from its measured |max| (it is never stored): it comes from the bound
// |stem out| <= max_c (|bn scale_c| sum |w_c|) max|x| + max_c |shift_c| (bound_l, bound_s: computed once at model finalize) --
// a few bits looser than the measured maximum, deterministic, and independent of the other images of the batch.
// Per level0 row the products are summed in the order (kernel row, columns {0, 1}, column 2): another order than
// lowc_kernel<16, 3, 1> (tap pairs) -- results agree to float32 round-off.
struct Lowc2Params {
    const float* in;   // NCHW [B][planes][H][W]
    float* out;        // NHWC [B][H][W][16]: level0's output
    const void *w0_hi, *w0_lo;  // stem fragments [7][64][8] (pack_lowc_weights<4, 7>)
    const void *w1_hi, *w1_lo;  // level0 fragments [6][64][8] (pack_lowc_rows_weights)
    const float *scale0, *shift0, *scale1, *shift1;  // folded BatchNorm x 2^-e of the fragment rows
    const unsigned* in_amax;
    unsigned* out_amax;
    float bound_l, bound_s;
    int B, H, W, planes, rows, strips, bands;
};

constexpr int L2_SW = 14;                       // level0 columns per strip
constexpr int L2_INW = 24, L2_STW = 18;         // pixels of the wave-private input / stem rows (23 / 18 used)
constexpr int L2_PF = 3;                        // input rows requested ahead

__global__ __launch_bounds__(256, 3) void lowc2_kernel(const Lowc2Params p) {
    __shared__ __attribute__((aligned(16))) _Float16 rows_s[4][2 * (L2_INW * 4 + L2_STW * 16)];
    // level0's 12 weight fragments (hi, lo x 6 K steps) live in LDS, shared by the four waves and read per row (12 KB of the
    // ~15 KB a wave reads per row: a fifth of the LDS rate) -- in registers they were 48 of 192 and the kernel ran two waves per
    // SIMD; without them it fits three
    __shared__ __attribute__((aligned(16))) u32x4 w1_s[2][6][64];
    {
        const int t = threadIdx.x;
        for (int i = t; i < 2 * 6 * 64; i += 256) {
            const int pl_ = i / (6 * 64), r = i - pl_ * (6 * 64);
            w1_s[pl_][r / 64][r % 64] = reinterpret_cast<const u32x4*>(pl_ ? p.w1_lo : p.w1_hi)[r];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int pl = lane & 15, q = lane >> 4;
    int t = blockIdx.x;
    const int sg = t % ((p.strips + 3) / 4);
    t /= (p.strips + 3) / 4;
    const int band = t % p.bands, b = t / p.bands;
    const int sx = sg * 4 + wid, x0 = sx * L2_SW, y0 = band * p.rows;
    float amax = 0.f;
    if (x0 < p.W && y0 < p.H) {
        _Float16* in_hi = rows_s[wid];
        _Float16* in_lo = in_hi + L2_INW * 4;
        _Float16* st_hi = in_lo + L2_INW * 4;
        _Float16* st_lo = st_hi + L2_STW * 16;
        // ---- weights -> registers (first operand: lane = (output channel lane % 16, k chunk lane / 16)) ----
        h8 w0h[7], w0l[7];
        {
            const u32x4* g0h = reinterpret_cast<const u32x4*>(p.w0_hi) + lane;
            const u32x4* g0l = reinterpret_cast<const u32x4*>(p.w0_lo) + lane;
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                const u32x4 a = g0h[s * 64], c = g0l[s * 64];
                w0h[s] = *reinterpret_cast<const h8*>(&a);
                w0l[s] = *reinterpret_cast<const h8*>(&c);
            }
        }
        float afwd = 1.f, ainv = 1.f, hfwd = 1.f, hinv = 1.f;
        if (p.in_amax) {
            const unsigned am = cp_amax_read(p.in_amax);
            cp_amax_to_scale(am, &afwd, &ainv);
            cp_amax_to_scale(__float_as_uint(p.bound_l * __uint_as_float(am) + p.bound_s), &hfwd, &hinv);
        }
        float4 sc0 = p.scale0 ? *reinterpret_cast<const float4*>(p.scale0 + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 sh0 = p.shift0 ? *reinterpret_cast<const float4*>(p.shift0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 sc1 = p.scale1 ? *reinterpret_cast<const float4*>(p.scale1 + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 sh1 = p.shift1 ? *reinterpret_cast<const float4*>(p.shift1 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        sc0.x *= ainv; sc0.y *= ainv; sc0.z *= ainv; sc0.w *= ainv;
        sc1.x *= hinv; sc1.y *= hinv; sc1.z *= hinv; sc1.w *= hinv;
        // the two pad pixels of the stem row (read by the invalid output columns 14, 15 and by the zero-weight chunks) stay zero
        if (lane < 32) {
            reinterpret_cast<uint32_t*>(st_hi + 16 * 16)[lane & 15] = 0u;
            reinterpret_cast<uint32_t*>(st_lo + 16 * 16)[lane & 15] = 0u;
        }
        // ---- input row requests: lane l < 23 -> column x0 - 4 + l of planes 0 .. 2.  NO branch anywhere in the row loop: rows /
        //      columns / planes that do not exist are requested beyond the buffer descriptor (-> 0, no traffic) and masked stores go
        //      there too.  (With `if`s around the loads and the store, the compiler's s_waitcnt pass merged the paths conservatively
        //      and every row waited with vmcnt(0) for the loads it had just issued for three rows later: 3200 clocks per row.) ----
        const int icol = x0 - 4 + lane;
        const bool icol_ok = lane < 23 && (unsigned)icol < (unsigned)p.W;
        const unsigned plane_b = (unsigned)p.H * (unsigned)p.W * 4u;
        const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in + (size_t)b * p.planes * p.H * p.W), 0,
                                                                            (int)(plane_b * (unsigned)p.planes), 0x00020000);
        const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)b * p.H * p.W * 16, 0,
                                                                             (int)((unsigned)p.H * (unsigned)p.W * 64u), 0x00020000);
        const int so1 = p.planes > 1 ? (int)plane_b : 0x7ffffff0, so2 = p.planes > 2 ? (int)(2u * plane_b) : 0x7ffffff0;
        auto request = [&](int j, float (&v)[3]) {
            const unsigned vo = (icol_ok && (unsigned)j < (unsigned)p.H) ? (unsigned)(j * p.W + icol) * 4u : 0xffffffffu;
            v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_in, (int)vo, 0, 0));
            v[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_in, (int)vo, so1, 0));
            v[2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_in, (int)vo, so2, 0));
        };
        const int j0 = y0 - 4, j1 = min(y0 + p.rows, p.H) + 3;   // input rows j0 .. j1 (inclusive)
        float pv[L2_PF][3];
#pragma unroll
        for (int d = 0; d < L2_PF; ++d) request(j0 + d, pv[d]);
        f32x4 sacc[7], lacc[3];
#pragma unroll
        for (int k = 0; k < 7; ++k) sacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) lacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int scol = x0 - 1 + pl;  // this lane's stem pixel (column); its level0 pixel is x0 + pl
        const bool scol_ok = (unsigned)scol < (unsigned)p.W;
        const bool ocol_ok = pl < L2_SW && x0 + pl < p.W;
        const int y_end = min(y0 + p.rows, p.H);

        for (int j = j0; j <= j1; ++j) {
            // ---- input row j: registers -> hi / lo halves in LDS; request row j + 3 ----
            {
                uint32_t h0, l0, h1, l1;
                split2(pv[0][0] * afwd, pv[0][1] * afwd, &h0, &l0);
                split2(pv[0][2] * afwd, 0.f, &h1, &l1);
                if (lane < L2_INW) {
                    *reinterpret_cast<u32x2*>(in_hi + lane * 4) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(in_lo + lane * 4) = u32x2{l0, l1};
                }
#pragma unroll
                for (int d = 0; d + 1 < L2_PF; ++d) { pv[d][0] = pv[d + 1][0]; pv[d][1] = pv[d + 1][1]; pv[d][2] = pv[d + 1][2]; }
                request(j + L2_PF, pv[L2_PF - 1]);
            }
            __builtin_amdgcn_wave_barrier();
            h8 ah, al;
            {
                const int off = (pl + 2 * q) * 4;  // 8-byte aligned: columns pl + 2 q, pl + 2 q + 1
                const u32x2 a0 = *reinterpret_cast<const u32x2*>(in_hi + off), a1 = *reinterpret_cast<const u32x2*>(in_hi + off + 4);
                const u32x2 c0 = *reinterpret_cast<const u32x2*>(in_lo + off), c1 = *reinterpret_cast<const u32x2*>(in_lo + off + 4);
                const u32x4 hv = {a0.x, a0.y, a1.x, a1.y}, lv = {c0.x, c0.y, c1.x, c1.y};
                ah = *reinterpret_cast<const h8*>(&hv);
                al = *reinterpret_cast<const h8*>(&lv);
            }
            __builtin_amdgcn_wave_barrier();
            // ---- stem: input row j is kernel row 6 - k of the stem row in sacc[k] (rows j - 3 .. j + 3) ----
            // (term-major: the three products of an accumulator -- lo*hi, hi*lo, hi*hi, in that order -- are seven MFMAs apart, so no
            // MFMA waits for the one in front of it)
#pragma unroll
            for (int k = 0; k < 7; ++k) sacc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h[6 - k], al, sacc[k], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 7; ++k) sacc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0l[6 - k], ah, sacc[k], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 7; ++k) sacc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h[6 - k], ah, sacc[k], 0, 0, 0);
            const int i = j - 3;  // the stem row that is complete now
            const f32x4 sdone = sacc[0];
#pragma unroll
            for (int k = 0; k < 6; ++k) sacc[k] = sacc[k + 1];
            sacc[6] = f32x4{0.f, 0.f, 0.f, 0.f};
            // (stem rows i < y0 - 1 are incomplete -- their upper kernel rows were never fed -- and only reach level0 rows above the
            // band, which are not stored)
            {
                const bool ok = scol_ok && (unsigned)i < (unsigned)p.H;   // level0 sees zeros outside the picture
                const float v0 = ok ? fmaxf(sdone[0] * sc0.x + sh0.x, 0.f) * hfwd : 0.f;
                const float v1 = ok ? fmaxf(sdone[1] * sc0.y + sh0.y, 0.f) * hfwd : 0.f;
                const float v2 = ok ? fmaxf(sdone[2] * sc0.z + sh0.z, 0.f) * hfwd : 0.f;
                const float v3 = ok ? fmaxf(sdone[3] * sc0.w + sh0.w, 0.f) * hfwd : 0.f;
                uint32_t h0, l0, h1, l1;
                split2(v0, v1, &h0, &l0);
                split2(v2, v3, &h1, &l1);
                *reinterpret_cast<u32x2*>(st_hi + pl * 16 + q * 4) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(st_lo + pl * 16 + q * 4) = u32x2{l0, l1};
            }
            __builtin_amdgcn_wave_barrier();
            h8 bh[2], bl[2];
            {
                const int o0 = (pl + (q >> 1)) * 16 + (q & 1) * 8;   // columns 0, 1 of the kernel row
                const int o1 = (pl + 2) * 16 + (q & 1) * 8;          // column 2 (chunks 2, 3 meet zero weights)
                bh[0] = *reinterpret_cast<const h8*>(st_hi + o0);
                bl[0] = *reinterpret_cast<const h8*>(st_lo + o0);
                bh[1] = *reinterpret_cast<const h8*>(st_hi + o1);
                bl[1] = *reinterpret_cast<const h8*>(st_lo + o1);
            }
            __builtin_amdgcn_wave_barrier();
            // ---- level0: stem row i is kernel row 2 - k of the level0 row in lacc[k] (rows i - 1 .. i + 1) ----
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                h8 wh[3], wl[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const u32x4 a = w1_s[0][2 * (2 - k) + g][lane], c = w1_s[1][2 * (2 - k) + g][lane];
                    wh[k] = *reinterpret_cast<const h8*>(&a);
                    wl[k] = *reinterpret_cast<const h8*>(&c);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) lacc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[k], bl[g], lacc[k], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 3; ++k) lacc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[k], bh[g], lacc[k], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 3; ++k) lacc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[k], bh[g], lacc[k], 0, 0, 0);
            }
            const int o = i - 1;  // the level0 row that is complete now
            const f32x4 ldone = lacc[0];
            lacc[0] = lacc[1];
            lacc[1] = lacc[2];
            lacc[2] = f32x4{0.f, 0.f, 0.f, 0.f};
            {
                const bool st_ok = o >= y0 && o < y_end && ocol_ok;
                float4 y;
                y.x = fmaxf(ldone[0] * sc1.x + sh1.x, 0.f);
                y.y = fmaxf(ldone[1] * sc1.y + sh1.y, 0.f);
                y.z = fmaxf(ldone[2] * sc1.z + sh1.z, 0.f);
                y.w = fmaxf(ldone[3] * sc1.w + sh1.w, 0.f);
                const float m4 = fmaxf(fmaxf(y.x, y.y), fmaxf(y.z, y.w));
                amax = fmaxf(amax, st_ok ? m4 : 0.f);
                const u32x4 pk = {__float_as_uint(y.x), __float_as_uint(y.y), __float_as_uint(y.z), __float_as_uint(y.w)};
                __builtin_amdgcn_raw_buffer_store_b128(pk, r_out, (int)(st_ok ? (unsigned)((o * p.W + x0 + pl) * 64 + 16 * q) : 0x80000000u), 0, 0);
            }
        }
    }
    if (p.out_amax) cp_amax_commit(p.out_amax, amax);
}

template <int CIN, int KS>
__global__ void pack_lowc_weights(const float* __restrict__ w, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                  const float* __restrict__ fwd, int cout, int cin, int ci0) {
    constexpr int KSTEPS = CIN == 4 ? KS : (KS * KS * CIN + 31) / 32;
    const int nf = cout / 16, total = KSTEPS * nf * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63, sf = idx >> 9;
        const int f = sf % nf, s = sf / nf;
        const int co = f * 16 + (lane & 15), q = lane >> 4;
        float v = 0.f;
        if (CIN == 4) {
            const int kh = s, kw = 2 * q + (j >> 2), ci = ci0 + (j & 3);
            if (kw < KS && ci < cin) v = w[(((size_t)co * cin + ci) * KS + kh) * KS + kw];
        } else {
            const int tap = 2 * s + (q >> 1), ci = (q & 1) * 8 + j;
            if (tap < KS * KS && ci < cin) v = w[(((size_t)co * cin + ci) * KS + tap / KS) * KS + tap % KS];
        }
        uint32_t h, l;
        split2(fwd ? v * fwd[co] : v, 0.f, &h, &l);
        hi[idx] = (uint16_t)(h & 0xffffu);
        lo[idx] = (uint16_t)(l & 0xffffu);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// level1 (3x3 / stride 2, 16 -> 32, + BatchNorm + ReLU; pose_dla_dcn.py:310-322 `_make_conv_level(16, 32, 1, stride=2)`) as a row
// stream: 25.2 MB per 512 x 512 image against 0.3 GFLOP -- nothing but memory traffic, which the tile kernel above moves at
// 3.75 TB/s behind its barriers.  The structure of strm16.hip: a wave owns a strip of 32 output columns (65 input columns) and a band
// of output rows; an input row (65 px x 16 ch x 4 B = 4.1 KB, requested four rows ahead, branch-free) is split into hi / lo halves
// in a wave-private LDS row -- even and odd columns in separate planes, so the stride-2 fragment of a kernel column is a
// conflict-free ds_read_b128 at pitch 48 B -- and multiplied into the rolling accumulators of the output rows it belongs to: an even
// row 2y is kernel row 1 of output row y, an odd row 2y + 1 kernel row 2 of y (which is then complete) and kernel row 0 of y + 1.
// The nine weight fragments (one per tap: 32 output channels x 16 input channels, hi / lo, 18 KB) sit in LDS; products transposed
// (weights first: a lane holds four consecutive channels of ITS pixel per accumulator quad), and the finished values pass through
// wave-private transposition rows so that eight lanes hold one pixel's 128 bytes and a 16-byte store instruction writes 8 whole lines
// (straight from the accumulator layout it would touch 64 lines: measured 0.51 against 0.43 ms).
constexpr int L1_W = 32;                    // output columns per strip
constexpr int L1_PIECES = (2 * L1_W + 1) * 4;   // float4 pieces of an input row (65 pixels x 4)
constexpr int L1_NLD = (L1_PIECES + 63) / 64;   // loads per lane and row (5; the last round's spare lanes land in spare slots)
constexpr int L1_PITCH = 48;                // bytes per pixel and plane (16 halfs + 16 B)
constexpr int L1_PLANE = 40 * L1_PITCH;     // even or odd columns of a row: 33 / 32 used + the spare slots of pieces 260 .. 319
constexpr int L1_ROWB = 4 * L1_PLANE;       // hi even, hi odd, lo even, lo odd
constexpr int L1_WAVES = 8;
constexpr int L1_WB = 9 * 2 * 1024;         // weight fragments [tap][hi, lo][64 lanes][16 B]
constexpr int L1_TPITCH = 144;              // bytes per pixel of the epilogue's transposition rows (32 channels x 4 B + 16)
constexpr int L1_TRB = 32 * L1_TPITCH;
constexpr int L1_LDS = L1_WB + L1_WAVES * (L1_ROWB + L1_TRB);   // 114 KB: one workgroup per CU
constexpr int L1_PF = 4;                    // input rows requested ahead = register buffers used in turn (odd, even, odd, even)
static_assert(L1_LDS > 80 * 1024 && L1_LDS <= 160 * 1024, "one workgroup per CU");

struct Lowc1Params {
    const float* in;   // NHWC [B][H][W][16]
    float* out;        // NHWC [B][Ho][Wo][32]
    const void *w_hi, *w_lo;   // fragments [9 taps][64 lanes][8 halfs] (pack kind 5)
    const float *scale, *shift;
    const unsigned* in_amax;
    unsigned* out_amax;
    int B, H, W, Ho, Wo, rows, strips, bands, njobs;
};

__global__ __launch_bounds__(64 * L1_WAVES, 1) void lowc1s_kernel(const Lowc1Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char l1_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const u32x4* gh = reinterpret_cast<const u32x4*>(p.w_hi);
        const u32x4* gl = reinterpret_cast<const u32x4*>(p.w_lo);
        u32x4* ws = reinterpret_cast<u32x4*>(l1_smem);
        for (int i = tid; i < 9 * 64; i += 64 * L1_WAVES) {
            const int t = i >> 6, l = i & 63;
            ws[(2 * t) * 64 + l] = gh[i];
            ws[(2 * t + 1) * 64 + l] = gl[i];
        }
    }
    __syncthreads();
    float afwd = 1.f, ainv = 1.f;
    if (p.in_amax) cp_amax_to_scale(cp_amax_read(p.in_amax), &afwd, &ainv);
    afwd = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(afwd)));
    ainv = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ainv)));
    unsigned char* row_b = l1_smem + L1_WB + wid * (L1_ROWB + L1_TRB);   // [hi even | hi odd | lo even | lo odd]
    unsigned char* tr_b = row_b + L1_ROWB;
    const unsigned char* w_b = l1_smem + lane * 16;
    const int px = lane & 31, kg = lane >> 5;
    float4 sc[4], sh[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const int n0 = 8 * g4 + 4 * kg;
        sc[g4] = p.scale ? *reinterpret_cast<const float4*>(p.scale + n0) : make_float4(1.f, 1.f, 1.f, 1.f);
        sh[g4] = p.shift ? *reinterpret_cast<const float4*>(p.shift + n0) : make_float4(0.f, 0.f, 0.f, 0.f);
        sc[g4].x *= ainv; sc[g4].y *= ainv; sc[g4].z *= ainv; sc[g4].w *= ainv;
    }
    float amax = 0.f;
    // LDS slot of this lane's pieces: piece i = lane + 64 k = pixel i / 4 (input column 2 x0 - 1 + i / 4), channel quad i % 4
    int wo[L1_NLD];
#pragma unroll
    for (int k = 0; k < L1_NLD; ++k) {
        const int i = lane + 64 * k, pp = i >> 2;
        wo[k] = (pp & 1) * L1_PLANE + (pp >> 1) * L1_PITCH + (i & 3) * 8;
    }

    // (block-major: the eight waves of a workgroup take neighbouring strips of one band -- whole rows of the picture per workgroup)
    for (int job = (int)blockIdx.x * L1_WAVES + wid; job < p.njobs; job += (int)gridDim.x * L1_WAVES) {
        int t = job;
        const int sx = t % p.strips;
        t /= p.strips;
        const int band = t % p.bands, b = t / p.bands;
        const int x0 = sx * L1_W;
        const int y0 = band * p.rows, y1 = min(y0 + p.rows, p.Ho);
        const int jlast = 2 * y1 - 1;   // the last input row of the band
        const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in + (size_t)b * p.H * p.W * 16), 0,
                                                                            (int)((unsigned)p.H * (unsigned)p.W * 64u), 0x00020000);
        const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)b * p.Ho * p.Wo * 32, 0,
                                                                             (int)((unsigned)p.Ho * (unsigned)p.Wo * 128u), 0x00020000);
        unsigned cvo[L1_NLD];
#pragma unroll
        for (int k = 0; k < L1_NLD; ++k) {
            const int i = lane + 64 * k, c = 2 * x0 - 1 + (i >> 2);
            cvo[k] = (i < L1_PIECES && (unsigned)c < (unsigned)p.W) ? (unsigned)(c * 64 + (i & 3) * 16) : 0xffffffffu;
        }
        const float col_okf = x0 + px < p.Wo ? 1.f : 0.f;
        float4 pf[L1_PF][L1_NLD];
        auto request = [&](int j, float4 (&v)[L1_NLD]) {
            const bool row_ok = (unsigned)j < (unsigned)p.H && j <= jlast;   // (rows behind the band: nothing is fetched)
            const int so = row_ok ? j * p.W * 64 : 0;
#pragma unroll
            for (int k = 0; k < L1_NLD; ++k) {
                const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r_in, (int)(row_ok ? cvo[k] : 0xffffffffu), so, 0);
                v[k] = make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
            }
        };
#pragma unroll
        for (int d = 0; d < L1_PF; ++d) {
            request(2 * y0 - 1 + d, pf[d]);
            __builtin_amdgcn_sched_barrier(0);   // requests stay in row order (strm16.hip: what the other order costs)
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;

        // registers -> hi / lo halves in the wave's LDS row; the buffer then takes the request four rows on
        auto stage = [&](const int j, float4 (&buf)[L1_NLD]) {
#pragma unroll
            for (int k = 0; k < L1_NLD; ++k) {
                const float4 v = buf[k];
                uint32_t h0, l0, h1, l1;
                split2(v.x * afwd, v.y * afwd, &h0, &l0);
                split2(v.z * afwd, v.w * afwd, &h1, &l1);
                *reinterpret_cast<u32x2*>(row_b + wo[k]) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(row_b + 2 * L1_PLANE + wo[k]) = u32x2{l0, l1};
            }
            request(j + L1_PF, buf);
            __builtin_amdgcn_wave_barrier();
        };
        // acc += kernel row kh x the staged row: three taps, kernel column kw reads pixel 2 x + kw = even[x], odd[x], even[x + 1]
        auto mac = [&](const int kh, f32x16& a) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ao = (kw == 1 ? L1_PLANE : 0) + (px + (kw == 2 ? 1 : 0)) * L1_PITCH + kg * 16;
                const h8 xh = *reinterpret_cast<const h8*>(row_b + ao), xl = *reinterpret_cast<const h8*>(row_b + 2 * L1_PLANE + ao);
                const int tp = kh * 3 + kw;
                const h8 wh = *reinterpret_cast<const h8*>(w_b + (2 * tp) * 1024), wl = *reinterpret_cast<const h8*>(w_b + (2 * tp + 1) * 1024);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, a, 0, 0, 0);
            }
        };
        // odd input row j = 2 y + 1: closes output row y (kernel row 2), stores it, and opens row y + 1 (kernel row 0)
        auto odd_row = [&](const int j, float4 (&buf)[L1_NLD]) {
            stage(j, buf);
            mac(2, acc);
            f32x16 nxt;
#pragma unroll
            for (int r = 0; r < 16; ++r) nxt[r] = 0.f;
            mac(0, nxt);
            __builtin_amdgcn_wave_barrier();
            const int o = (j - 1) >> 1;   // (j = 2 y0 - 1: o = y0 - 1, masked)
            const unsigned okm = ~(unsigned)((o - y0) >> 31) & (unsigned)((o - y1) >> 31);
            const float okf = __uint_as_float(okm & __float_as_uint(col_okf));
            // BatchNorm + ReLU where a lane knows its channels (quad g4 = channels 8 g4 + 4 kg .. + 3 of pixel px), then through the
            // wave's transposition rows: stored straight from this layout a 16-byte store would touch 64 different 128-byte lines
            // (lane = pixel); after the exchange lane l holds quad l % 8 of pixel l / 8 + 8 i and a store writes 8 whole lines
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 y;
                y.x = fmaxf(acc[4 * g4] * sc[g4].x + sh[g4].x, 0.f);
                y.y = fmaxf(acc[4 * g4 + 1] * sc[g4].y + sh[g4].y, 0.f);
                y.z = fmaxf(acc[4 * g4 + 2] * sc[g4].z + sh[g4].z, 0.f);
                y.w = fmaxf(acc[4 * g4 + 3] * sc[g4].w + sh[g4].w, 0.f);
                amax = fmaxf(amax, fmaxf(fmaxf(y.x, y.y), fmaxf(y.z, y.w)) * okf);
                *reinterpret_cast<float4*>(tr_b + px * L1_TPITCH + (2 * g4 + kg) * 16) = y;
            }
            __builtin_amdgcn_wave_barrier();
            // (the two "no store" marks are OR-ed in AFTER the sum: added, a masked row and a masked column would wrap into range)
            const unsigned row_out = (unsigned)(o * p.Wo) * 128u, row_no = ~okm & 0x80000000u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pxo = (lane >> 3) + 8 * i;
                const u32x4 pk = *reinterpret_cast<const u32x4*>(tr_b + pxo * L1_TPITCH + (lane & 7) * 16);
                const unsigned col_no = x0 + pxo < p.Wo ? 0u : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(pk, r_out, (int)((row_out + (unsigned)((x0 + pxo) * 128 + (lane & 7) * 16)) | row_no | col_no), 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
            acc = nxt;
        };
        auto even_row = [&](const int j, float4 (&buf)[L1_NLD]) {
            stage(j, buf);
            mac(1, acc);
            __builtin_amdgcn_wave_barrier();
        };
        // rows 2 y0 - 1 .. 2 y1 - 1, four per turn; what a turn runs past the band is neither fetched nor stored
        for (int j = 2 * y0 - 1; j <= jlast; j += 4) {
            odd_row(j, pf[0]);
            even_row(j + 1, pf[1]);
            odd_row(j + 2, pf[2]);
            even_row(j + 3, pf[3]);
        }
    }
    if (p.out_amax) cp_amax_commit(p.out_amax, amax);
}

// level1's weights for lowc1s_kernel: one fragment per tap, first-operand order (lane = output channel lane % 32, input channels
// 8 (lane / 32) .. + 7)
__global__ void pack_lowc1s_weights(const float* __restrict__ w, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                    const float* __restrict__ fwd, int cin) {
    const int total = 9 * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63, tp = idx >> 9;
        const int co = lane & 31, ci = (lane >> 5) * 8 + j;
        float v = 0.f;
        if (ci < cin) v = w[(((size_t)co * cin + ci) * 3 + tp / 3) * 3 + tp % 3];
        uint32_t h, l;
        split2(fwd ? v * fwd[co] : v, 0.f, &h, &l);
        hi[idx] = (uint16_t)(h & 0xffffu);
        lo[idx] = (uint16_t)(l & 0xffffu);
    }
}

template <int CIN, int KS, int S, int COUT, int TW, int TH, bool NCHW_IN, int NG = 1>
int launch_lowc(const LowcParams& p, hipStream_t s) {
    const int tiles = ((p.Wo + TW - 1) / TW) * ((p.Ho + TH - 1) / TH) * p.B;
    hipLaunchKernelGGL((lowc_kernel<CIN, KS, S, COUT, TW, TH, NCHW_IN, NG>), dim3(tiles), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

}  // namespace

// kind: 0 stem 7x7 (NCHW input with `planes` <= 4 channels, pad 3) -> 16; 1 level0 3x3 16->16; 2 level1 3x3/2 16->32;
//       3 stem 7x7 with 5..8 input planes (two groups of 4) -> 16
//       4 level0's weights in the row order of the fused stem + level0 kernel (pack only; launched by cp_launch_lowc_fused)
//       5 level1's weights, one fragment per tap, for the row-streaming level1 kernel (launched as kind 5)
size_t cp_lowc_weight_halfs(int kind) {
    return kind == 0 ? (size_t)7 * 1 * 512 : kind == 1 ? (size_t)5 * 1 * 512 : kind == 2 ? (size_t)5 * 2 * 512 : kind == 4 ? (size_t)6 * 512 : kind == 5 ? (size_t)9 * 512 : (size_t)14 * 512;
}

int cp_launch_pack_lowc(int kind, const float* w, void* hi, void* lo, const float* fwd, int cin, hipStream_t s) {
    if (kind == 0) hipLaunchKernelGGL((pack_lowc_weights<4, 7>), dim3(16), dim3(256), 0, s, w, (uint16_t*)hi, (uint16_t*)lo, fwd, 16, cin, 0);
    else if (kind == 1) hipLaunchKernelGGL((pack_lowc_weights<16, 3>), dim3(16), dim3(256), 0, s, w, (uint16_t*)hi, (uint16_t*)lo, fwd, 16, cin, 0);
    else if (kind == 2) hipLaunchKernelGGL((pack_lowc_weights<16, 3>), dim3(16), dim3(256), 0, s, w, (uint16_t*)hi, (uint16_t*)lo, fwd, 32, cin, 0);
    else if (kind == 3) {  // plane groups 0..3 and 4..7: fragments [group][kh][lane][8]
        hipLaunchKernelGGL((pack_lowc_weights<4, 7>), dim3(16), dim3(256), 0, s, w, (uint16_t*)hi, (uint16_t*)lo, fwd, 16, cin, 0);
        hipLaunchKernelGGL((pack_lowc_weights<4, 7>), dim3(16), dim3(256), 0, s, w, (uint16_t*)hi + 7 * 512, (uint16_t*)lo + 7 * 512,
                           fwd, 16, cin, 4);
    } else if (kind == 4) {
        hipLaunchKernelGGL(pack_lowc_rows_weights, dim3(16), dim3(256), 0, s, w, (uint16_t*)hi, (uint16_t*)lo, fwd, 16, cin);
    } else if (kind == 5) {
        hipLaunchKernelGGL(pack_lowc1s_weights, dim3(16), dim3(256), 0, s, w, (uint16_t*)hi, (uint16_t*)lo, fwd, cin);
    } else return CP_ERR_INVALID;
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

// stem (<= 3 planes, NCHW) + level0 in one launch -> level0's output (NHWC, 16 channels)
int cp_launch_lowc_fused(const float* in, float* out, const void* w0_hi, const void* w0_lo, const float* scale0, const float* shift0,
                         const void* w1_hi, const void* w1_lo, const float* scale1, const float* shift1, float bound_l, float bound_s,
                         const unsigned* in_amax, unsigned* out_amax, int B, int H, int W, int planes, hipStream_t s) {
    if (planes < 1 || planes > 3 || B < 1 || H < 1 || W < 1) return CP_ERR_INVALID;
    Lowc2Params p;
    p.in = in; p.out = out;
    p.w0_hi = w0_hi; p.w0_lo = w0_lo; p.w1_hi = w1_hi; p.w1_lo = w1_lo;
    p.scale0 = scale0; p.shift0 = shift0; p.scale1 = scale1; p.shift1 = shift1;
    p.in_amax = in_amax; p.out_amax = out_amax;
    p.bound_l = bound_l; p.bound_s = bound_s;
    p.B = B; p.H = H; p.W = W; p.planes = planes;
    p.strips = (W + L2_SW - 1) / L2_SW;
    // rows per job: about 8192 wave jobs per launch (a job's first 5 rows are recomputed halo), at least 8 rows
    int bands = 8192 / (B * p.strips);
    if (bands < 1) bands = 1;
    int rows = (H + bands - 1) / bands;
    if (rows < 8) rows = 8;
    p.rows = rows;
    p.bands = (H + rows - 1) / rows;
    const int groups = (p.strips + 3) / 4;
    hipLaunchKernelGGL(lowc2_kernel, dim3(groups * p.bands * B), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

int cp_launch_lowc(int kind, const float* in, float* out, const void* w_hi, const void* w_lo, const float* scale,
                   const float* shift, const unsigned* in_amax, unsigned* out_amax, int B, int H, int W, int planes,
                   hipStream_t s) {
    LowcParams p;
    p.in_amax = in_amax;
    p.out_amax = out_amax;
    p.in = in;
    p.out = out;
    p.w_hi = w_hi;
    p.w_lo = w_lo;
    p.scale = scale;
    p.shift = shift;
    p.B = B;
    p.H = H;
    p.W = W;
    p.planes = planes;
    if (kind == 0) {
        if (planes < 1 || planes > 4) return CP_ERR_INVALID;
        p.Ho = H; p.Wo = W; p.pad = 3;
        return launch_lowc<4, 7, 1, 16, 64, 8, true>(p, s);
    }
    if (kind == 1) {
        p.Ho = H; p.Wo = W; p.pad = 1;
        // 32 x 8 tiles: 24 KB of LDS, six blocks per CU instead of three with 64 x 8 (0.73 -> 0.62 ms at batch 64; 16 x 8: 0.81)
        return launch_lowc<16, 3, 1, 16, 32, 8, false>(p, s);
    }
    if (kind == 2) {
        p.Ho = (H + 2 - 3) / 2 + 1; p.Wo = (W + 2 - 3) / 2 + 1; p.pad = 1;
        return launch_lowc<16, 3, 2, 32, 32, 8, false>(p, s);
    }
    if (kind == 3) {
        if (planes < 5 || planes > 8) return CP_ERR_INVALID;
        p.Ho = H; p.Wo = W; p.pad = 3;
        return launch_lowc<4, 7, 1, 16, 64, 8, true, 2>(p, s);
    }
    if (kind == 5) {
        Lowc1Params q;
        q.in = in; q.out = out; q.w_hi = w_hi; q.w_lo = w_lo; q.scale = scale; q.shift = shift; q.in_amax = in_amax; q.out_amax = out_amax;
        q.B = B; q.H = H; q.W = W;
        q.Ho = (H + 2 - 3) / 2 + 1; q.Wo = (W + 2 - 3) / 2 + 1;
        if ((size_t)H * W * 64 >= (size_t)0x70000000u || (size_t)q.Ho * q.Wo * 128 >= (size_t)0x70000000u) return CP_ERR_INVALID;
        static int cus_of[16] = {0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return CP_ERR_LAUNCH;
        if (!cus_of[dev]) {
            hipDeviceProp_t prop;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lowc1s_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L1_LDS) != hipSuccess)
                return CP_ERR_LAUNCH;
            cus_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }
        const int cus = cus_of[dev];
        q.strips = (q.Wo + L1_W - 1) / L1_W;
        // the tallest band of output rows that still gives every wave slot of the chip a job (B = 16 / 32 at 512 x 512: 16 / 32 rows,
        // one job per wave: 5.2 / 4.7 TB/s).  At twice that (B = 64: 1.6 GB through the kernel) the chip moves MORE with a quarter of
        // its CUs idle -- measured 0.43 - 0.44 ms on 256 CUs at every band height against 0.365 on 192 with 32-row bands dealt in three
        // turns (profiles/r06_level1_rows_ab.txt: it is the number of CUs injecting requests, not the waves per CU, the band
        // geometry or the image order) -- so large launches run on three quarters of the CUs.
        int rows = 64;
        while (rows > 8 && B * q.strips * ((q.Ho + rows - 1) / rows) < cus * L1_WAVES) rows >>= 1;
        int use_cus = cus;
        if (B * q.strips * ((q.Ho + 31) / 32) >= 2 * cus * L1_WAVES) {
            rows = 32;
            use_cus = cus * 3 / 4;
        }
        q.rows = rows;
        q.bands = (q.Ho + rows - 1) / rows;
        q.njobs = B * q.strips * q.bands;
        const int blocks = (q.njobs + L1_WAVES - 1) / L1_WAVES < use_cus ? (q.njobs + L1_WAVES - 1) / L1_WAVES : use_cus;
        hipLaunchKernelGGL(lowc1s_kernel, dim3(blocks), dim3(64 * L1_WAVES), L1_LDS, s, q);
        return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
    }
    return CP_ERR_INVALID;
}
