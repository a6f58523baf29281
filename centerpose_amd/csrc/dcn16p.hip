// Patch-resident DCNv2 (modulated deformable 3x3 convolution, stride 1, pad 1, dilation 1, one deformable group) in the
// split-f16 ("f16x3") arithmetic of igemm16.hip: the deformed samples are gathered from LDS, not through the texture
// path.  Replaces the reference's modulated_deformable_im2col + GEMM pair (DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:125-195,
// dcn_v2_cuda.cu:42-172) for the layers whose launch fills the chip; dcn16.hip keeps the others (split-K, ragged maps).
//
// Why.  dcn16.hip fetches the 4 bilinear corners of every (pixel, tap) as 16-byte vectors through the texture
// addresser: 36 corner reads per output pixel and input channel, 64 B/clk per CU -- its K tile costs >= 1152 clk of
// address processing against 384 clk of MFMA issue (N tile 64), and it measures at 60 % of that bound (PMC: TA 58 %
// busy, MFMA 13 %).  Here a block owns an 8 x 16 patch of output pixels and stages the (8+6) x (16+6) input halo of a
// 32-channel chunk ONCE into LDS as float32 (2.4 input pixels per output pixel instead of 36 corner vectors); the
// corners are then ds_read_b128s -- 256 B/clk per CU, four times the texture path -- and all 9 taps x 2 K steps of
// the chunk are served from the same image.
//   * Pixel rows are 144 bytes apart (128 + 16): consecutive pixels start 9 sixteen-byte bank groups apart, 9 is odd,
//     so the 16 lanes of a ds_read_b128 group (neighbouring pixels, same channel quad) hit 16 distinct groups; no XOR
//     swizzle, hence ONE address register per (lane, tap) and every corner / K step / quad is an immediate offset.
//   * A lane gathers exactly its MFMA A-fragment (pixel = lane % 32, 8 consecutive channels = 2 quads per corner),
//     blends in float32 (packed FMAs), splits to binary16 hi / lo in registers: the A tile never exists in LDS.
//     Each wave owns 32 pixels x the whole N tile, so nothing is gathered twice inside a block.
//   * The bilinear set-up (corner address, 4 weights with mask and activation pre-scale folded in) of a lane's 9 taps
//     lives in 45 registers for the whole kernel (2 waves per SIMD => 256 VGPRs each).
//   * Samples whose 2x2 corner block leaves the staged halo (|offset| > 2..3 px at the patch border: ~2 % of samples
//     at sigma = 1.5 px) are "exceptions": the set-up appends them to a block list (LDS atomic) and the staging copies
//     their 2x2 source pixels into spare patch rows, laid out so that the same four immediates address them -- the K
//     loop has no branch.  A block with more than ECAP exceptions (huge offsets everywhere) switches, as a whole, to
//     gathering through buffer loads like dcn16.hip (slower, same results).
// K order is (32-channel chunk, tap, 16-channel half): same products as dcn16.hip, different summation order.
// LDS: 440 x 144 B patch + 16 KB weight tiles (double-buffered) = 79.9 KB => two blocks per CU, whose staging /
// compute phases overlap.
#include "patch16_common.h"
#include <type_traits>

namespace {

constexpr int TH = PATCH_TH, TW = PATCH_TW, HALO = 3;
constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO, NPIX = PH * PW;  // 22 x 14 = 308 patch pixels
constexpr int CKC = 32;                                                // channels per staged chunk
constexpr int PSTR = CKC * 4 + 16;                                     // bytes between patch pixels
constexpr int EROWS = 6;                                               // spare patch rows: 2x2 blocks of exception samples
constexpr int EPR = PW / 2;                                            // exceptions per pair of spare rows
constexpr int ECAP = (EROWS / 2) * EPR;                                // 33
constexpr int NPIX_ALL = NPIX + EROWS * PW;                            // 440
constexpr int ST_REG = (NPIX * 8 + 255) / 256;                         // staging passes over the halo: 10
constexpr int ST_EXC = (ECAP * 32 + 255) / 256;                        // ... over the exception blocks: 5

__device__ __forceinline__ float4 buf_ld4s(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int NT>
__global__ __launch_bounds__(256, 2) void dcn16p_kernel(const ConvParams p, const int tiles_m, const int tiles_n) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    constexpr int BN = 32 * NT;
    constexpr int B_CHUNKS = BN * BK16 * 2 / 16;  // 16-byte chunks per weight array (hi or lo) per 32-deep K tile
    constexpr int B_SLOTS = (B_CHUNKS + 255) / 256;
    constexpr int B_SZ = BN * LDH;
    static_assert(B_CHUNKS % 256 == 0, "whole passes over the weight tile");
    __shared__ __attribute__((aligned(16))) unsigned char patch[NPIX_ALL * PSTR];
    __shared__ __attribute__((aligned(16))) _Float16 bt[2][2 * B_SZ];  // [buffer][hi | lo]
    __shared__ int exc_list[ECAP];
    __shared__ int exc_count;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int tile = tile_of_block(tiles_m, tiles_n);
    const int tn = tile % tiles_n;
    int tm = tile / tiles_n;
    const int txs = p.W / TW, tys = p.H / TH;
    const int tx0 = (tm % txs) * TW;
    tm /= txs;
    const int ty0 = (tm % tys) * TH, b = tm / tys;
    float afwd, ainv;
    conv_in_scale(p, &afwd, &ainv);
    // wave-uniform: keep both in scalar registers (the vector file is full)
    afwd = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(afwd)));
    ainv = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ainv)));

    const unsigned img_px = (unsigned)p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(p.src[0], img_px * (unsigned)p.Cin * 4u);
    const __amdgpu_buffer_rsrc_t r_om = make_rsrc(p.offmask, img_px * 128u);
    const unsigned w_bytes = (unsigned)((size_t)p.CoutPad * p.Kpad16 * 2);
    const __amdgpu_buffer_rsrc_t r_wh = make_rsrc(p.w16_hi, w_bytes), r_wl = make_rsrc(p.w16_lo, w_bytes);
    const int cb = p.Cin * 4, rowb = p.W * cb;

    if (tid == 0) exc_count = 0;
    // ---- this lane's pixel: tile row m = 32 wave + lane % 32 -> patch pixel (m / 16, m % 16) ----
    const int m = wid * 32 + lcol;
    const int y = ty0 + (m >> 4), x = tx0 + (m & 15);
    float om[28];  // the pixel's offset / mask record (27 used)
    {
        const unsigned rec = (unsigned)((b * p.H + y) * p.W + x) * 128u;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const float4 v = buf_ld4(r_om, rec + 16u * i);
            om[4 * i] = v.x; om[4 * i + 1] = v.y; om[4 * i + 2] = v.z; om[4 * i + 3] = v.w;
        }
    }
    __syncthreads();  // exc_count = 0 is visible

    // ---- bilinear set-up of the lane's 9 taps (dcn_v2_im2col_cuda.cu:25-54, 150-187), kept in registers ----
    // corner (h_lo, w_lo) of each tap: its byte address in `patch` + this lane's 32-byte channel half (fast mode) /
    // its byte offset into the input tensor | 4 corner-validity bits (buffer-load mode); one of them survives the set-up
    int a_lds[9], g_base[9];
    float bw[9][4];  // corner weights x mask x activation pre-scale
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int kh = t / 3, kw = t % 3;
        const float dh = om[2 * t], dw = om[2 * t + 1], mk = om[18 + t] * afwd;
        const float h_im = (float)(y - 1 + kh) + dh;
        const float w_im = (float)(x - 1 + kw) + dw;
        int q = 0, gb = 0;
        float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
        bool exc = false;
        int key = 0;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
            const int h_lo = (int)floorf(h_im), w_lo = (int)floorf(w_im);
            const int h_hi = h_lo + 1, w_hi = w_lo + 1;
            const float lh = h_im - (float)h_lo, lw = w_im - (float)w_lo;
            const float hh = 1.f - lh, hw = 1.f - lw;
            int vm = 0;
            if (h_lo >= 0 && w_lo >= 0) vm |= 1;
            if (h_lo >= 0 && w_hi <= p.W - 1) vm |= 2;
            if (h_hi <= p.H - 1 && w_lo >= 0) vm |= 4;
            if (h_hi <= p.H - 1 && w_hi <= p.W - 1) vm |= 8;
            gb = (((b * p.H + h_lo) * p.W + w_lo) * cb) | vm;
            w1 = hh * hw * mk; w2 = hh * lw * mk; w3 = lh * hw * mk; w4 = lh * lw * mk;
            const int qy = h_lo - (ty0 - HALO), qx = w_lo - (tx0 - HALO);
            if ((unsigned)qy <= (unsigned)(PH - 2) && (unsigned)qx <= (unsigned)(PW - 2)) q = qy * PW + qx;
            else {
                exc = true;
                key = ((h_lo + 1) << 16) | (w_lo + 1);
            }
        }
        if (exc && lrow == 0) {  // one of the two lanes that share the pixel files the exception
            const int e = atomicAdd(&exc_count, 1);
            if (e < ECAP) {
                exc_list[e] = key;
                q = NPIX + (e / EPR) * (2 * PW) + (e % EPR) * 2;
            }
        }
        q = __shfl(q, lcol, 64);
        a_lds[t] = q * PSTR + lrow * 32;
        g_base[t] = gb;
        bw[t][0] = w1; bw[t][1] = w2; bw[t][2] = w3; bw[t][3] = w4;
    }
    __syncthreads();
    const int nexc_all = __builtin_amdgcn_readfirstlane(exc_count);  // scalar: the mode branches below stay uniform
    const bool slow = nexc_all > ECAP;  // block-uniform
    const int nexc = nexc_all < ECAP ? nexc_all : ECAP;
    int addr[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) addr[t] = slow ? g_base[t] : a_lds[t];

    // ---- staging geometry (chunk-invariant): thread -> (patch pixel, 16-byte channel quad) per pass ----
    unsigned st_off[ST_REG], ex_off[ST_EXC];
    int ex_lds[ST_EXC];
    const int st_lds = (tid >> 3) * PSTR + (tid & 7) * 16;  // + 32 * PSTR per pass
#pragma unroll
    for (int s = 0; s < ST_REG; ++s) {
        const int pix = (tid >> 3) + 32 * s;
        const int ppy = pix / PW, ppx = pix - ppy * PW;
        const int iy = ty0 - HALO + ppy, ix = tx0 - HALO + ppx;
        const bool in = pix < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        st_off[s] = in ? (unsigned)(((b * p.H + iy) * p.W + ix) * cb + (tid & 7) * 16) : OOB;
    }
#pragma unroll
    for (int s = 0; s < ST_EXC; ++s) {
        const int idx = tid + 256 * s;
        const int e = idx >> 5, corner = (idx >> 3) & 3;
        ex_off[s] = OOB;
        ex_lds[s] = -1;
        if (e < nexc && !slow) {
            const int key = exc_list[e];
            const int iy = (key >> 16) - 1 + (corner >> 1), ix = (key & 0xffff) - 1 + (corner & 1);
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                ex_off[s] = (unsigned)(((b * p.H + iy) * p.W + ix) * cb + (tid & 7) * 16);
            ex_lds[s] = (NPIX + (e / EPR) * (2 * PW) + (e % EPR) * 2 + (corner >> 1) * PW + (corner & 1)) * PSTR + (tid & 7) * 16;
        }
    }

    // ---- weight tile: chunk f -> row n = f / 4, 16-byte column f % 4 of the 32-deep K tile ----
    unsigned b_off[B_SLOTS];
#pragma unroll
    for (int j = 0; j < B_SLOTS; ++j) {
        const int f = tid + j * 256;
        b_off[j] = (unsigned)(((size_t)(tn * BN + f / 4) * p.Kpad16 + (f % 4) * 8) * 2);
    }
    // Weight tiles are fetched TWO taps ahead (a tap is only 2 K steps = ~0.4 us of work, less than an L2 round trip
    // under load): tile of tap T lives in register set T % 3 from its issue (tap T - 2) to its LDS store (end of tap
    // T - 1); 9 taps per chunk keep the rotation consistent across chunks, and only two sets are ever live.
    u32x4 gbh[3][B_SLOTS], gbl[3][B_SLOTS];
    auto issue_b = [&](int set, int kbyte) {  // kbyte: byte offset of the K tile inside a weight row (wave-uniform)
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) {
            gbh[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wh, (int)b_off[j], kbyte, 0);
            gbl[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wl, (int)b_off[j], kbyte, 0);
        }
    };
    auto store_b = [&](int set, int buf) {
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) {
            const int f = tid + j * 256;
            const int nn = f / 4, c = f % 4;
            *reinterpret_cast<u32x4*>(bt[buf] + nn * LDH + (c ^ swz(nn)) * 8) = gbh[set][j];
            *reinterpret_cast<u32x4*>(bt[buf] + B_SZ + nn * LDH + (c ^ swz(nn)) * 8) = gbl[set][j];
        }
    };

    acc_t acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < F::NACC; ++r) acc[0][j][r] = 0.f;
    const int b_frag = lcol * LDH;

    const int nch = p.Cin / CKC;
    const int nkt = nch * 9;
    issue_b(0, 0);
    if (!slow) issue_b(1, (1 * p.Cin) * 2);  // (nkt >= 9)
    // the K loop, instantiated once per mode (SLOW is block-uniform): the fast instance has no control flow inside a
    // chunk, so the scheduler can hoist fragment reads and the next step's gather above the MFMAs
    auto k_loop = [&](auto mode) {
    constexpr bool SLOW = decltype(mode)::value;
    constexpr int PF = SLOW ? 1 : 2;  // weight tiles in flight ahead of the tap being multiplied
    int kt = 0;
    for (int ch = 0; ch < nch; ++ch) {
        // ---- stage the halo (and the exception blocks) of this 32-channel chunk; every wave left the previous
        //      chunk's patch at the barrier that ended its last tap ----
        if (!SLOW) {
            const int csoff = ch * (CKC * 4);
            // two rounds (halo rows, then the tail + the exception blocks): half the registers in flight
            constexpr int H1 = 8;
            {
                float4 sv[H1];
#pragma unroll
                for (int s = 0; s < H1; ++s) sv[s] = buf_ld4s(r_x, st_off[s], csoff);
#pragma unroll
                for (int s = 0; s < H1; ++s) *reinterpret_cast<float4*>(patch + st_lds + s * (32 * PSTR)) = sv[s];
            }
            {
                float4 sv[ST_REG - H1], ev[ST_EXC];
#pragma unroll
                for (int s = H1; s < ST_REG; ++s) sv[s - H1] = buf_ld4s(r_x, st_off[s], csoff);
#pragma unroll
                for (int s = 0; s < ST_EXC; ++s) ev[s] = buf_ld4s(r_x, ex_off[s], csoff);
#pragma unroll
                for (int s = H1; s < ST_REG; ++s)
                    if (s < ST_REG - 1 || (tid >> 3) + 32 * s < NPIX)
                        *reinterpret_cast<float4*>(patch + st_lds + s * (32 * PSTR)) = sv[s - H1];
#pragma unroll
                for (int s = 0; s < ST_EXC; ++s)
                    if (ex_lds[s] >= 0) *reinterpret_cast<float4*>(patch + ex_lds[s]) = ev[s];
            }
        }
        if (ch == 0) store_b(0, 0);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t, ++kt) {
            const int cur = kt & 1;
            if (kt + PF < nkt) {
                const int t2 = t + PF < 9 ? t + PF : t + PF - 9, ch2 = t + PF < 9 ? ch : ch + 1;
                issue_b((t + PF) % 3, (t2 * p.Cin + ch2 * CKC) * 2);
            }
            const _Float16* Bh = bt[cur] + b_frag;
            const _Float16* Bl = Bh + B_SZ;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float4 r[4][2];
                if (!SLOW) {
                    const unsigned char* ap = patch + addr[t] + ks * 64;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int co = ((c >> 1) * PW + (c & 1)) * PSTR;
                        r[c][0] = *reinterpret_cast<const float4*>(ap + co);
                        r[c][1] = *reinterpret_cast<const float4*>(ap + co + 16);
                    }
                } else {
                    // buffer-load mode: corner offsets into the tensor, invalid corners out of range (-> 0)
                    const int so = (ch * CKC + ks * 16) * 4;
                    const int base = (addr[t] & ~15) + lrow * 32;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int gi = (addr[t] & (1 << c)) ? base + (c >> 1) * rowb + (c & 1) * cb : (int)OOB_BASE;
                        r[c][0] = buf_ld4s(r_x, (unsigned)gi, so);
                        r[c][1] = buf_ld4s(r_x, (unsigned)gi + 16u, so);
                    }
                }
                h8 bh[NT], bl[NT];
                const int co = ((ks * 2 + lrow) ^ swz(lcol)) * 8;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    bh[j] = *reinterpret_cast<const h8*>(Bh + j * 32 * LDH + co);
                    bl[j] = *reinterpret_cast<const h8*>(Bl + j * 32 * LDH + co);
                }
                // fma(w4, v4, fma(w3, v3, fma(w2, v2, w1 * v1))) per channel (dcn16.hip's order), two per v_pk_fma_f32
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 w1 = {bw[t][0], bw[t][0]}, w2 = {bw[t][1], bw[t][1]}, w3 = {bw[t][2], bw[t][2]},
                            w4 = {bw[t][3], bw[t][3]};
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    const float4 v1 = r[0][hq], v2 = r[1][hq], v3 = r[2][hq], v4 = r[3][hq];
                    f32x2 lo2 = w1 * f32x2{v1.x, v1.y}, hi2 = w1 * f32x2{v1.z, v1.w};
                    lo2 = __builtin_elementwise_fma(w2, f32x2{v2.x, v2.y}, lo2);
                    hi2 = __builtin_elementwise_fma(w2, f32x2{v2.z, v2.w}, hi2);
                    lo2 = __builtin_elementwise_fma(w3, f32x2{v3.x, v3.y}, lo2);
                    hi2 = __builtin_elementwise_fma(w3, f32x2{v3.z, v3.w}, hi2);
                    lo2 = __builtin_elementwise_fma(w4, f32x2{v4.x, v4.y}, lo2);
                    hi2 = __builtin_elementwise_fma(w4, f32x2{v4.z, v4.w}, hi2);
                    const Split2 s0 = split2(lo2.x, lo2.y), s1 = split2(hi2.x, hi2.y);
                    hi[2 * hq] = s0.hi; hi[2 * hq + 1] = s1.hi;
                    lo[2 * hq] = s0.lo; lo[2 * hq + 1] = s1.lo;
                }
                const u32x4 ahv = {hi[0], hi[1], hi[2], hi[3]}, alv = {lo[0], lo[1], lo[2], lo[3]};
                const h8 ah = *reinterpret_cast<const h8*>(&ahv), al = *reinterpret_cast<const h8*>(&alv);
                // same term order as igemm16.hip (lo*hi, hi*lo, hi*hi)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[0][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[0][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[0][j], 0, 0, 0);
            }
            if (kt + 1 < nkt) store_b((t + 1) % 3, cur ^ 1);
            __syncthreads();
        }
    }
    };
    if (slow) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    patch_epilogue<1, NT, 4, 1>(p, acc, b, ty0, tx0, tn, wid, 0, lane, ainv);
}

template <int NT>
int launch_dcn16p(const ConvParams& p, hipStream_t stream) {
    constexpr int BN = 32 * NT;
    const int tiles_m = p.B * (p.H / TH) * (p.W / TW), tiles_n = p.CoutPad / BN;
    hipLaunchKernelGGL((dcn16p_kernel<NT>), dim3(tiles_m * tiles_n), dim3(256), 0, stream, p, tiles_m, tiles_n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

}  // namespace

// Full 8 x 16 patches, NHWC output, no split-K, 32-bit offsets; the caller (cp_launch_conv16) also asks for enough
// blocks to fill the chip before it prefers this kernel to dcn16.hip's.
bool cp_dcn16p_supported(const ConvParams& p) {
    return p.offmask && p.w16_hi && p.w16_lo && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.nsrc == 1 &&
           p.H == p.Ho && p.W == p.Wo && p.splitk <= 1 && p.Cin % CKC == 0 && p.H % TH == 0 && p.W % TW == 0 &&
           p.H < 65535 && p.W < 65535 && p.store == CP_STORE_NHWC && p.Kpad16 == 9 * p.Cin && p.CoutPad % 64 == 0 &&
           !p.gn_stats && (size_t)p.B * p.H * p.W * p.Cin * 4 < (size_t)0xf0000000u &&
           (size_t)p.B * p.H * p.W * 128 < (size_t)0xf0000000u && (size_t)p.B * p.H * p.W * p.ldo * 4 < (size_t)0xf0000000u;
}

int cp_dcn16p_blocks(const ConvParams& p) { return p.B * (p.H / TH) * (p.W / TW) * (p.CoutPad / 64); }

int cp_launch_dcn16p(const ConvParams& p, hipStream_t stream) {
    if (!cp_dcn16p_supported(p)) return CP_ERR_INVALID;
    return launch_dcn16p<2>(p, stream);
}
