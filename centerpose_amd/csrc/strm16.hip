// Streamed 3x3 / stride-1 / pad-1 convolution for 64 input channels and <= 32 output channels in the split-f16 ("f16x3") arithmetic:
// the DCNv2 offset / mask convolutions `conv_offset_mask` of the full-resolution layers (DCNv2/dcn_v2.py:105-111: 64 -> 27 @ 128 x 128,
// five per dla_34 forward) -- the layers on which the patch kernel (halo16.hip, 32-wide N tile) keeps the matrix pipe 27 % busy:
// per 8 x 16 patch a workgroup lives 24 k clocks of which the K loop is 7 k, the rest prologue, staging and epilogue behind barriers,
// and every wave pulls its own copy of every weight fragment (2 KB per three MFMAs) through the texture path.
//
// Here (the structure of lowc.hip's fused stem + level0 kernel, round 6):
//   * a WAVE streams down a 32-pixel-wide column strip on its own: no barrier after the start, no LDS shared between waves
//     except the read-only weights.  Per input row: 34 pixels x 64 channels (8.7 KB, one coalesced piece, requested two rows
//     ahead) -> x 2^e, hi / lo halves -> a wave-private LDS row (pixel pitch 144 B: conflict-free fragment reads) -> the 12 A
//     fragment pairs (3 columns x 4 channel groups) are read ONCE and each is multiplied into the THREE rolling accumulators of
//     the output rows the input row belongs to (kernel rows 2, 1, 0 of rows j - 1, j, j + 1): 108 MFMAs per input row and wave
//     against ~350 VALU (split, epilogue, accumulator rotation), 96 LDS fragment reads and 4 sixteen-byte stores per lane;
//   * the whole weight tensor (36 K steps x hi / lo x 1 KB = 72 KB in MFMA operand order) sits in LDS, loaded once per
//     workgroup: a weight fragment is a conflict-free ds_read_b128, not a texture-path load per wave;
//   * one persistent workgroup of 8 waves per CU (72 KB of weights + 8 x 10.1 KB of row buffers), jobs = (image, strip, band of
//     16 rows) dealt wave-major; a band re-reads one row above and one below, and the row products that belong to output rows
//     outside the band are formed and dropped (6 of 54: cheaper than a branch in front of every MFMA);
//   * products transposed (weights as the first operand): a lane's accumulator quad = four consecutive channels of its pixel ->
//     16-byte stores, scale / shift read as float4.
// K order per output row: (kernel row, kernel column, 16-channel group; lo.hi, hi.lo, hi.hi) -- for a 64-channel layer the order of
// halo16.hip (one chunk: tap, 16-channel group) and of the per-tap kernel: linear / ReLU outputs are bit-identical to theirs.  The
// mask sigmoid is exp2 / rcp here (cp_fast_sigmoid, |error| < 3e-7) where those kernels call expf.
// The row loop is branch-free on purpose (out-of-range buffer offsets, arithmetic masks, two register buffers used in turn):
// profiles/NOTES.md round 6 lists what each kind of branch costs; tests/test_host_cpu.py checks the built loop for counted waits.
#include "igemm16_common.h"

namespace {

constexpr int SM_W = 32;                  // output columns per strip
constexpr int SM_PX = SM_W + 2;           // input pixels per row
constexpr int SM_PITCH = 144;             // bytes per pixel and plane in the row buffer (64 halfs + 16 B)
constexpr int SM_ROWB = (SM_PX + 2) * SM_PITCH;   // one plane of a wave's row buffer: 34 pixels + 2 spare slots (pieces 544 .. 575 of the last load round)
constexpr int SM_G = 36;                  // K steps of 16: (tap, 16-channel group)
constexpr int SM_WB = 2 * SM_G * 1024;    // weight fragments, hi then lo
constexpr int SM_WAVES = 8;
constexpr int SM_LDS = SM_WB + SM_WAVES * 2 * SM_ROWB;
constexpr int SM_Q = SM_PX * 16;          // float4 pieces of an input row (544)
constexpr int SM_NLD = (SM_Q + 63) / 64;  // loads per lane and row (9)
constexpr int SM_PF = 2;                  // rows requested ahead (= register buffers used in turn)
static_assert(SM_LDS <= 160 * 1024, "one workgroup per CU");

__global__ __launch_bounds__(64 * SM_WAVES, 1) void strm16_kernel(const ConvParams p, const int rows, const int strips, const int bands,
                                                                  const int njobs) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- the weights, once per workgroup: fragment (K step g, plane) = 64 lanes x 16 B, already in operand order ----
    {
        const u32x4* gh = reinterpret_cast<const u32x4*>(p.w16f_hi);
        const u32x4* gl = reinterpret_cast<const u32x4*>(p.w16f_lo);
        u32x4* ws = reinterpret_cast<u32x4*>(smem);
        for (int i = tid; i < SM_G * 64; i += 64 * SM_WAVES) {
            ws[i] = gh[i];
            ws[SM_G * 64 + i] = gl[i];
        }
    }
    __syncthreads();
    float afwd, ainv;
    conv_in_scale(p, &afwd, &ainv);
    afwd = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(afwd)));
    ainv = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ainv)));
    unsigned char* row_hi = smem + SM_WB + wid * (2 * SM_ROWB);
    unsigned char* row_lo = row_hi + SM_ROWB;
    const unsigned char* w_hi = smem + lane * 16;
    const unsigned char* w_lo = w_hi + SM_G * 1024;
    const int px = lane & 31, kg = lane >> 5;   // this lane's pixel of the fragment and its 8-channel half of a K step
    const unsigned img_b = (unsigned)p.H * (unsigned)p.W * 256u;   // bytes per image (64 channels x 4)
    // epilogue constants: accumulator quad g4 = channels 8 g4 + 4 kg .. + 3
    float4 sc[4], sh[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const int n0 = 8 * g4 + 4 * kg;
        sc[g4] = p.scale ? *reinterpret_cast<const float4*>(p.scale + n0) : make_float4(1.f, 1.f, 1.f, 1.f);
        sh[g4] = p.shift ? *reinterpret_cast<const float4*>(p.shift + n0) : make_float4(0.f, 0.f, 0.f, 0.f);
        sc[g4].x *= ainv; sc[g4].y *= ainv; sc[g4].z *= ainv; sc[g4].w *= ainv;
    }
    const float relu_floor = p.act == CP_ACT_RELU ? 0.f : -__builtin_inff();
    const int act_from = p.act == CP_ACT_SIGMOID ? 0 : p.act == CP_ACT_SIGMOID_FROM ? p.act_from : 1 << 30;
    float amax = 0.f;

    // jobs dealt wave-major (wave w of block b: job w * blocks + b, + blocks * 8 per turn): with fewer jobs than wave slots every CU
    // still gets its share
    for (int job = wid * (int)gridDim.x + (int)blockIdx.x; job < njobs; job += (int)gridDim.x * SM_WAVES) {
        int t = job;
        const int sx = t % strips;
        t /= strips;
        const int band = t % bands, b = t / bands;
        const int x0 = sx * SM_W, y0 = band * rows, y1 = min(y0 + rows, p.H);
        const __amdgpu_buffer_rsrc_t r_in = make_rsrc(p.src[0] + (size_t)b * p.H * p.W * 64, img_b);
        const __amdgpu_buffer_rsrc_t r_out = make_rsrc(p.out + (size_t)b * p.H * p.W * p.ldo + p.coff, (unsigned)p.H * (unsigned)p.W * (unsigned)p.ldo * 4u);
        // ---- row requests: piece i = lane + 64 k of the row's 544 float4: pixel i / 16 (column x0 - 1 + i / 16), channel quad i % 16.
        //      No branch in the row loop: what does not exist is requested out of range (zeros, no traffic). ----
        unsigned cvo[SM_NLD];   // column part of the byte offset, or out of range
#pragma unroll
        for (int k = 0; k < SM_NLD; ++k) {
            const int i = lane + 64 * k, c = x0 - 1 + (i >> 4);
            cvo[k] = (i < SM_Q && (unsigned)c < (unsigned)p.W) ? (unsigned)(c * 256 + (i & 15) * 16) : 0xffffffffu;
        }
        const float col_okf = x0 + px < p.W ? 1.f : 0.f;
        float4 pf[SM_PF][SM_NLD];
        auto request = [&](int j, float4 (&v)[SM_NLD]) {
            const bool row_ok = (unsigned)j < (unsigned)p.H;
            const int so = row_ok ? j * p.W * 256 : 0;
#pragma unroll
            for (int k = 0; k < SM_NLD; ++k) {
                const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r_in, (int)(row_ok ? cvo[k] : 0xffffffffu), so, 0);
                v[k] = make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
            }
        };
#pragma unroll
        for (int d = 0; d < SM_PF; ++d) {
            request(y0 - 1 + d, pf[d]);
            // keep the requests in row order: the scheduler put row y0 first, the loop head then had to wait for the LAST request
            // of the prologue -- and, merged over the back edge, for everything in flight in every turn (s_waitcnt vmcnt(0))
            __builtin_amdgcn_sched_barrier(0);
        }
        acc_t racc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) racc[k][r] = 0.f;

        // one input row; `buf` holds row j and receives the request for row j + 2 (two buffers used in turn: a register copy
        // pf[0] = pf[1] would make every row wait for the loads of the NEXT one)
        auto row = [&](const int j, float4 (&buf)[SM_NLD]) {
            // ---- input row j: registers -> hi / lo halves in the wave's LDS row; request row j + 2 ----
#pragma unroll
            for (int k = 0; k < SM_NLD; ++k) {
                const int i = lane + 64 * k;
                const float4 v = buf[k];
                const Split2 s0 = split2(v.x * afwd, v.y * afwd), s1 = split2(v.z * afwd, v.w * afwd);
                // (pieces 544 .. 575 of the last round land in the two spare pixel slots 34, 35: no condition around the write)
                const int o = (i >> 4) * SM_PITCH + (i & 15) * 8;
                *reinterpret_cast<u32x2*>(row_hi + o) = u32x2{s0.hi, s1.hi};
                *reinterpret_cast<u32x2*>(row_lo + o) = u32x2{s0.lo, s1.lo};
            }
            request(j + SM_PF, buf);
            __builtin_amdgcn_wave_barrier();
            // ---- input row j is kernel row kh of output row j + 1 - kh, held in racc[2 - kh].  All three products are always formed:
            //      at the band's ends they land in accumulators that are never stored (rows y0 - 2, y0 - 1, y1, y1 + 1: 6 of 54
            //      products per band of 16) -- cheaper than a branch in front of every MFMA ----
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) {
                    const int ao = (px + kw) * SM_PITCH + cg * 32 + kg * 16;
                    const h8 ah = *reinterpret_cast<const h8*>(row_hi + ao), al = *reinterpret_cast<const h8*>(row_lo + ao);
                    h8 wh[3], wl[3];
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const int g = (kh * 3 + kw) * 4 + cg;
                        wh[kh] = *reinterpret_cast<const h8*>(w_hi + g * 1024);
                        wl[kh] = *reinterpret_cast<const h8*>(w_lo + g * 1024);
                    }
                    // term-major over the three accumulators: an MFMA never waits for the one in front of it
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) racc[2 - kh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kh], al, racc[2 - kh], 0, 0, 0);
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) racc[2 - kh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[kh], ah, racc[2 - kh], 0, 0, 0);
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) racc[2 - kh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kh], ah, racc[2 - kh], 0, 0, 0);
                }
            __builtin_amdgcn_wave_barrier();
            // ---- output row j - 1 is complete: scale / shift (+ sigmoid from channel act_from), one 16-byte store per channel quad ----
            const int o = j - 1;
            // all ones when y0 <= o < y1 -- by arithmetic: a selection on the row test became a scalar branch in the row loop, and
            // every branch there costs the counted waits on the row requests (s_waitcnt vmcnt(0) at the loop head)
            const unsigned okm = ~(unsigned)((o - y0) >> 31) & (unsigned)((o - y1) >> 31);
            const float okf = __uint_as_float(okm & __float_as_uint(col_okf));
            // scale / shift / activation where a lane knows its channels (quad g4 = channels 8 g4 + 4 kg .. + 3 of pixel px), then through
            // the wave's row buffer (free until the next row is staged; LDS operations of a wave execute in order): stored straight
            // from this layout, a 16-byte store touches 64 different 128-byte lines (lane = pixel); after the exchange lane l holds
            // channel quad l % 8 of pixel l / 8 + 8 i and a store writes 8 whole lines (measured on the level1 stream: -16 %)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n0 = 8 * g4 + 4 * kg;
                float v[4] = {racc[0][4 * g4] * sc[g4].x + sh[g4].x, racc[0][4 * g4 + 1] * sc[g4].y + sh[g4].y,
                              racc[0][4 * g4 + 2] * sc[g4].z + sh[g4].z, racc[0][4 * g4 + 3] * sc[g4].w + sh[g4].w};
                // sigmoid by hardware exp2 / rcp (cp_fast_sigmoid: |error| < 3e-7, the form the fused ConvGRU epilogues use): the
                // library expf + IEEE division is ~35 VALU per value and was compiled into a branch per value
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sg = cp_fast_sigmoid(v[e]);
                    v[e] = n0 + e >= act_from ? sg : v[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(v[e], relu_floor);
                    v[e] = n0 + e < p.Cout ? v[e] : 0.f;
                    amax = fmaxf(amax, fabsf(v[e]) * okf);
                }
                *reinterpret_cast<float4*>(row_hi + px * SM_PITCH + (2 * g4 + kg) * 16) = make_float4(v[0], v[1], v[2], v[3]);
            }
            __builtin_amdgcn_wave_barrier();
            // (the "no store" marks are OR-ed in after the sum: added, a masked row and a masked column could wrap into range)
            const unsigned row_out = (unsigned)(o * p.W * p.ldo) * 4u, row_no = ~okm & 0x80000000u;
            const unsigned q_no = 4 * (lane & 7) < p.Cout ? 0u : 0x80000000u;   // channel quads beyond Cout are not written
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pxo = (lane >> 3) + 8 * i;
                const u32x4 pk = *reinterpret_cast<const u32x4*>(row_hi + pxo * SM_PITCH + (lane & 7) * 16);
                const unsigned col_no = x0 + pxo < p.W ? 0u : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(pk, r_out, (int)((row_out + (unsigned)(((x0 + pxo) * p.ldo + (lane & 7) * 4) * 4)) | row_no | col_no | q_no), 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
            racc[0] = racc[1];
            racc[1] = racc[2];
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) racc[2][r] = 0.f;
        };
        // rows y0 - 1 .. y1, two per turn (an odd count runs one more row: its products and its store go nowhere)
        for (int j = y0 - 1; j <= y1; j += 2) {
            row(j, pf[0]);
            row(j + 1, pf[1]);
        }
    }
    if (p.out_amax) cp_amax_commit(p.out_amax, amax);
}

}  // namespace

// 3x3 / stride 1 / pad 1, ONE source of exactly 64 channels, one 32-wide N tile (Cout <= 32), NHWC output with 16-byte-aligned channel
// rows, no residual / GroupNorm / split-K, fragment-ordered weights present, 32-bit offsets
bool cp_strm16_supported(const ConvParams& p) {
    return p.w16f_hi && p.w16f_lo && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.nsrc == 1 && p.Cin == 64 &&
           p.src_c[0] == 64 && p.CoutPad == 32 && p.Cout <= 32 && !p.offmask && !p.res && !p.gn_stats && !p.gn_in_a && !p.gn_in_mr && !p.gru_x3 && !p.fuse_w2_hi &&
           p.splitk <= 1 && p.H == p.Ho && p.W == p.Wo && p.store == CP_STORE_NHWC && p.ldo % 4 == 0 && p.coff % 4 == 0 && p.coff + ((p.Cout + 3) & ~3) <= p.ldo &&
           p.Kpad16 == 576 && (p.act == CP_ACT_NONE || p.act == CP_ACT_RELU || p.act == CP_ACT_SIGMOID || p.act == CP_ACT_SIGMOID_FROM) &&
           (size_t)p.H * p.W * 256 < (size_t)0x70000000u && (size_t)p.H * p.W * p.ldo * 4 < (size_t)0x70000000u;
}

// (strip, band of 16 rows) jobs of the layer; the launcher halves the bands when that leaves wave slots empty
int cp_strm16_jobs(const ConvParams& p) { return p.B * ((p.W + SM_W - 1) / SM_W) * ((p.H + 15) / 16); }

int cp_launch_strm16(const ConvParams& p, hipStream_t stream) {
    if (!cp_strm16_supported(p)) return CP_ERR_INVALID;
    // per device (a process normally drives one): CU count, and the opt-in to more than 64 KB of dynamic LDS
    static int cus_of[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return CP_ERR_LAUNCH;
    if (!cus_of[dev]) {
        hipDeviceProp_t prop;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&strm16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS) != hipSuccess)
            return CP_ERR_LAUNCH;
        cus_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const int cus = cus_of[dev];
    // bands of 16 rows (2 of 18 input rows re-read, 6 of 54 row products unused); of 8 when that is what fills the chip's wave slots
    const int strips = (p.W + SM_W - 1) / SM_W;
    const int rows = p.B * strips * ((p.H + 15) / 16) >= cus * SM_WAVES * 3 / 4 ? 16 : 8;
    const int bands = (p.H + rows - 1) / rows;
    const int njobs = p.B * strips * bands;
    const int blocks = njobs < cus ? njobs : cus;
    hipLaunchKernelGGL(strm16_kernel, dim3(blocks), dim3(64 * SM_WAVES), SM_LDS, stream, p, rows, strips, bands, njobs);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}
