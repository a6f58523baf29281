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

template <int CIN, int KS, int S, int COUT, int TW, int TH, bool NCHW_IN, int NG = 1>
int launch_lowc(const LowcParams& p, hipStream_t s) {
    const int tiles = ((p.Wo + TW - 1) / TW) * ((p.Ho + TH - 1) / TH) * p.B;
    hipLaunchKernelGGL((lowc_kernel<CIN, KS, S, COUT, TW, TH, NCHW_IN, NG>), dim3(tiles), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

}  // namespace

// kind: 0 stem 7x7 (NCHW input with `planes` <= 4 channels, pad 3) -> 16; 1 level0 3x3 16->16; 2 level1 3x3/2 16->32;
//       3 stem 7x7 with 5..8 input planes (two groups of 4) -> 16
size_t cp_lowc_weight_halfs(int kind) {
    return kind == 0 ? (size_t)7 * 1 * 512 : kind == 1 ? (size_t)5 * 1 * 512 : kind == 2 ? (size_t)5 * 2 * 512 : (size_t)14 * 512;
}

int cp_launch_pack_lowc(int kind, const float* w, void* hi, void* lo, const float* fwd, int cin, hipStream_t s) {
    if (kind == 0) hipLaunchKernelGGL((pack_lowc_weights<4, 7>), dim3(16), dim3(256), 0, s, w, (uint16_t*)hi, (uint16_t*)lo, fwd, 16, cin, 0);
    else if (kind == 1) hipLaunchKernelGGL((pack_lowc_weights<16, 3>), dim3(16), dim3(256), 0, s, w, (uint16_t*)hi, (uint16_t*)lo, fwd, 16, cin, 0);
    else if (kind == 2) hipLaunchKernelGGL((pack_lowc_weights<16, 3>), dim3(16), dim3(256), 0, s, w, (uint16_t*)hi, (uint16_t*)lo, fwd, 32, cin, 0);
    else if (kind == 3) {  // plane groups 0..3 and 4..7: fragments [group][kh][lane][8]
        hipLaunchKernelGGL((pack_lowc_weights<4, 7>), dim3(16), dim3(256), 0, s, w, (uint16_t*)hi, (uint16_t*)lo, fwd, 16, cin, 0);
        hipLaunchKernelGGL((pack_lowc_weights<4, 7>), dim3(16), dim3(256), 0, s, w, (uint16_t*)hi + 7 * 512, (uint16_t*)lo + 7 * 512,
                           fwd, 16, cin, 4);
    } else return CP_ERR_INVALID;
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
    return CP_ERR_INVALID;
}
