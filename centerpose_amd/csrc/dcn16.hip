// Fused DCNv2 (modulated deformable 3x3 convolution, stride 1, pad 1, dilation 1, one deformable group) on the gfx950
// f16 matrix cores in the split-f16 ("f16x3") arithmetic of igemm16.hip.  Replaces the reference's
// modulated_deformable_im2col + GEMM pair (DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:125-195, dcn_v2_cuda.cu:42-172): the
// `columns` buffer (9x the input, written and re-read) never exists -- the four bilinear corners of a (pixel, tap) are
// gathered as 16-byte channel vectors, blended in registers, split into binary16 hi / lo and stored straight into the
// A tile of the implicit GEMM.
//
// What bounds it.  Per K tile (128 pixels x 32 channels of one tap) the block gathers 4 corners x 128 x 128 B = 64 KB
// through the texture-addresser / L1 path (64 B/clk/CU => >= 1024 clk) against 384 (N tile 64) or 768 (N tile 128) clk
// of MFMA issue: the gather, not the contraction, is the floor, so the loop is organised around keeping gather loads
// in flight at all times:
//   * software pipeline: the raw corner vectors of tile t+1 are issued BEFORE the MFMAs of tile t and blended /
//     converted / stored after them (one register set of A_SLOTS x 4 float4), so the L2 round trip of the gather
//     hides under the matrix work instead of preceding it;
//   * the bilinear set-up of all 9 taps of the tile's 128 pixels is computed once per block into an LDS table (not once
//     per lane and tap: 8 lanes share a pixel row), so a tap boundary costs two LDS reads and a few selects per row;
//   * the 4-corner blend runs on packed float32 FMAs (two channels per instruction);
//   * 8 waves per block for the 128-wide N tile (two pixel rows per thread instead of four: half the gather state per
//     thread, no spills next to the 64 accumulators' worth of output).
// Measured effect of the pipelining: +4 % (N 64) / +12 % (N 128) -- the gather is not latency- but issue-bound: a K tile
// costs 18 x 16-byte-per-lane load instructions per wave, 4 waves share the CU's one texture addresser at 16 clk per
// instruction = 1152 clk, i.e. <= 245 (N 64) / 490 (N 128) TFLOP/s however well the loads overlap.
// Same tiles, LDS layout (unpadded 64-byte rows, XOR-swizzled 16-byte chunks), tile map, epilogue and operand
// pre-scaling (ConvParams::in_amax) as igemm16.hip; results are bit-identical to the loop this replaces.
#include "igemm16_common.h"

namespace {

// OCC: waves per SIMD the register allocation must allow (HIP's second __launch_bounds__ argument)
template <int MT, int NT, int WM, int WN, int OCC>
__global__ __launch_bounds__(WM * WN * 64, OCC) void dcn16_kernel(const ConvParams p, const int tiles_m,
                                                                                  const int tiles_n) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    constexpr int NTH = WM * WN * 64;
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    constexpr int RPP = NTH / 8;                        // A rows covered by one pass of the block (8 float4 per row)
    constexpr int A_SLOTS = BM / RPP;                   // pixel rows per thread
    constexpr int B_CHUNKS = BN * BK16 * 2 / 16;        // 16-byte chunks per weight array (hi or lo)
    constexpr int B_SLOTS = (B_CHUNKS + NTH - 1) / NTH;
    constexpr bool B_PART = B_CHUNKS % NTH != 0;
    constexpr int A_SZ = BM * LDH, B_SZ = BN * LDH;
    constexpr int BUF = 2 * A_SZ + 2 * B_SZ;
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * BUF];
    // Bilinear set-up of every (pixel row, tap) of the tile, computed ONCE by the block (1152 entries over its threads)
    // instead of once per lane: the 8 lanes that share a pixel row used to repeat the floor / compare / index
    // arithmetic (with its quarter-rate integer multiplies) per tap -- PMC of the previous version: 302 VALU
    // instructions per wave and K tile against 12 MFMAs, i.e. the kernel was VALU-bound (42 % VALU busy, TA 58 %, MFMA
    // 13 %).  Entry: byte offset of corner (h_lo, w_lo) with the 4 corner-validity bits in its low bits (the offset is
    // a multiple of Cin * 4 >= 128), and the 4 corner weights (mask and activation pre-scale folded in).
    __shared__ int tab_base[BM * 9];
    __shared__ __attribute__((aligned(16))) float tab_w[BM * 9][4];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int tile = tile_of_block(tiles_m, tiles_n);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int M = p.B * p.Ho * p.Wo;
    PixelDecomp pdec;
    pdec.init(p.Ho, p.Wo, M);
    float afwd, ainv;
    conv_in_scale(p, &afwd, &ainv);

    // ---- per-thread pixel rows: row (tid / 8) + RPP * j, float4 column k4 = tid % 8 of the 32-channel K tile ----
    const int k4 = tid & 7;
    const unsigned img_px = (unsigned)p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(p.src[0], img_px * (unsigned)p.Cin * 4u);
    const __amdgpu_buffer_rsrc_t r_om = make_rsrc(p.offmask, img_px * 128u);
    {   // ---- set-up table (dcn_v2_im2col_cuda.cu:25-54, 150-187) ----
        const int cb = p.Cin * 4;
        for (int e = tid; e < BM * 9; e += NTH) {
            const int r = e % BM, tap = e / BM;  // consecutive lanes = consecutive pixels
            const int m = tm * BM + r;
            int base = 0;
            float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
            if (m < M) {
                int b, y, x;
                pdec.split(m, &b, &y, &x);
                const unsigned o = (unsigned)m * 128u;  // H == Ho, W == Wo: the record index is the output pixel index
                const float dh = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_om, (int)o + (2 * tap) * 4, 0, 0));
                const float dw = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_om, (int)o + (2 * tap + 1) * 4, 0, 0));
                const float mk = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_om, (int)o + (18 + tap) * 4, 0, 0)) * afwd;
                const int kh = tap / 3, kw = tap - kh * 3;
                const float h_im = (float)(y - 1 + kh) + dh;
                const float w_im = (float)(x - 1 + kw) + dw;
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
                    base = (((b * p.H + h_lo) * p.W + w_lo) * cb) | vm;  // may be "before" the tensor when h_lo / w_lo = -1:
                                                                          // only valid corners are ever dereferenced
                    w1 = hh * hw * mk; w2 = hh * lw * mk; w3 = lh * hw * mk; w4 = lh * lw * mk;
                }
            }
            tab_base[e] = base;
            *reinterpret_cast<float4*>(tab_w[e]) = make_float4(w1, w2, w3, w4);
        }
        __syncthreads();
    }
    const unsigned w_bytes = (unsigned)((size_t)p.CoutPad * p.Kpad16 * 2);
    const __amdgpu_buffer_rsrc_t r_wh = make_rsrc(p.w16_hi, w_bytes), r_wl = make_rsrc(p.w16_lo, w_bytes);

    const int nk = p.Kpad16 / BK16;
    int kt0, kt1;
    splitk_range(p, nk, &kt0, &kt1);
    const int n = kt1 - kt0;
    int u_tap = 0, u_kh = 0, u_kw = 0, u_c0 = 0;
    if (kt0 > 0) {
        const int k0 = kt0 * BK16;
        u_tap = k0 / p.Cin;
        u_c0 = k0 - u_tap * p.Cin;
        u_kh = u_tap / 3;
        u_kw = u_tap - u_kh * 3;
    }
    unsigned b_off[B_SLOTS];
#pragma unroll
    for (int j = 0; j < B_SLOTS; ++j) {
        const int f = tid + j * NTH;
        b_off[j] = (!B_PART || f < B_CHUNKS)
                       ? (unsigned)(((size_t)(tn * BN + f / 4) * p.Kpad16 + (f % 4) * 8 + (size_t)kt0 * BK16) * 2)
                       : OOB;
    }
    int b_soff = 0;

    // corner byte offsets / weights of the tap being gathered, expanded from the table
    int d_idx[A_SLOTS][4];
    float d_w[A_SLOTS][4];
    auto setup_tap = [&]() {
        const int cb = p.Cin * 4, rowb = p.W * cb;
#pragma unroll
        for (int j = 0; j < A_SLOTS; ++j) {
            const int e = u_tap * BM + (tid >> 3) + j * RPP;
            const int bv = tab_base[e];
            const float4 w = *reinterpret_cast<const float4*>(tab_w[e]);
            const int base = (bv & ~15) + k4 * 16;  // + this lane's float4 column; the K walk's channel offset is wave-uniform
                                                     // and rides in the loads' scalar offset
            d_idx[j][0] = (bv & 1) ? base : (int)OOB_BASE;
            d_idx[j][1] = (bv & 2) ? base + cb : (int)OOB_BASE;
            d_idx[j][2] = (bv & 4) ? base + rowb : (int)OOB_BASE;
            d_idx[j][3] = (bv & 8) ? base + rowb + cb : (int)OOB_BASE;
            d_w[j][0] = w.x; d_w[j][1] = w.y; d_w[j][2] = w.z; d_w[j][3] = w.w;
        }
    };

    // ---- issue: raw corner vectors + weight chunks of the tile the K walk points at, then advance the walk ----
    float4 raw[A_SLOTS][4];
    u32x4 gbh[B_SLOTS], gbl[B_SLOTS];
    bool need_setup = true;
    int tiles_left = n;
    auto issue = [&]() {
        const bool live = tiles_left > 0;
        if (need_setup && live) {
            setup_tap();
            need_setup = false;
        }
        const int csoff = live ? u_c0 * 4 : 0;  // bytes (range checking ignores the scalar offset: an OOB index stays OOB)
#pragma unroll
        for (int j = 0; j < A_SLOTS; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r_x, live ? d_idx[j][c] : (int)OOB, csoff, 0);
                raw[j][c] = make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
            }
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) {
            const unsigned vo = live ? b_off[j] : OOB;
            gbh[j] = __builtin_amdgcn_raw_buffer_load_b128(r_wh, (int)vo, live ? b_soff : 0, 0);
            gbl[j] = __builtin_amdgcn_raw_buffer_load_b128(r_wl, (int)vo, live ? b_soff : 0, 0);
        }
        --tiles_left;
        b_soff += BK16 * 2;
        u_c0 += BK16;
        if (u_c0 >= p.Cin) {
            u_c0 = 0;
            ++u_tap;
            if (++u_kw == 3) { u_kw = 0; ++u_kh; }
            need_setup = true;
        }
    };
    // ---- blend + split + LDS store of the issued tile (weights d_w are those of its tap: the next set-up only runs
    //      inside the next issue) ----
    auto blend_store = [&](int buf) {
        _Float16* Ah = lds + buf * BUF;
        _Float16* Al = Ah + A_SZ;
        _Float16* Bh = Al + A_SZ;
        _Float16* Bl = Bh + B_SZ;
#pragma unroll
        for (int j = 0; j < A_SLOTS; ++j) {
            // fma(w4, v4, fma(w3, v3, fma(w2, v2, w1 * v1))) per component, two components per v_pk_fma_f32
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 w1 = {d_w[j][0], d_w[j][0]}, w2 = {d_w[j][1], d_w[j][1]}, w3 = {d_w[j][2], d_w[j][2]},
                        w4 = {d_w[j][3], d_w[j][3]};
            const float4 v1 = raw[j][0], v2 = raw[j][1], v3 = raw[j][2], v4 = raw[j][3];
            f32x2 lo2 = w1 * f32x2{v1.x, v1.y}, hi2 = w1 * f32x2{v1.z, v1.w};
            lo2 = __builtin_elementwise_fma(w2, f32x2{v2.x, v2.y}, lo2);
            hi2 = __builtin_elementwise_fma(w2, f32x2{v2.z, v2.w}, hi2);
            lo2 = __builtin_elementwise_fma(w3, f32x2{v3.x, v3.y}, lo2);
            hi2 = __builtin_elementwise_fma(w3, f32x2{v3.z, v3.w}, hi2);
            lo2 = __builtin_elementwise_fma(w4, f32x2{v4.x, v4.y}, lo2);
            hi2 = __builtin_elementwise_fma(w4, f32x2{v4.z, v4.w}, hi2);
            float4 v;
            v.x = lo2.x; v.y = lo2.y; v.z = hi2.x; v.w = hi2.y;
            const int row = (tid >> 3) + j * RPP;
            const Split2 s0 = split2(v.x, v.y), s1 = split2(v.z, v.w);
            const int col = ((((k4 >> 1) ^ swz(row)) << 1) | (k4 & 1)) * 4;  // halfs
            *reinterpret_cast<u32x2*>(Ah + row * LDH + col) = u32x2{s0.hi, s1.hi};
            *reinterpret_cast<u32x2*>(Al + row * LDH + col) = u32x2{s0.lo, s1.lo};
        }
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) {
            const int f = tid + j * NTH;
            if (!B_PART || f < B_CHUNKS) {
                const int nn = f / 4, c = f % 4;
                *reinterpret_cast<u32x4*>(Bh + nn * LDH + (c ^ swz(nn)) * 8) = gbh[j];
                *reinterpret_cast<u32x4*>(Bl + nn * LDH + (c ^ swz(nn)) * 8) = gbl[j];
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
    const int lrow = lane >> 5, lcol = lane & 31;
    auto mma_tile = [&](int buf) {
        const _Float16* base = lds + buf * BUF;
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
            // same term order as igemm16.hip (lo*hi, hi*lo, hi*hi): bit-identical accumulation
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

    // ---- prologue: tile 0 into buffer 0 ----
    issue();
    blend_store(0);
    __syncthreads();
    for (int t = 0; t < n; ++t) {
        const int cur = t & 1;
        issue();              // tile t+1 (out-of-range loads past the end: zeros, no traffic)
        mma_tile(cur);        // ... in flight while tile t is multiplied
        blend_store(cur ^ 1);
        __syncthreads();
    }
    if (p.splitk > 1) igemm_store_partial<32, MT, NT, WM, WN>(p, acc, tm, tn, wm, wn, lane, blockIdx.y);
    else igemm_epilogue<32, MT, NT, WM, WN>(p, acc, tm, tn, wm, wn, lane, ainv);
}

template <int MT, int NT, int WM, int WN, int OCC>
int launch_dcn16(const ConvParams& p, hipStream_t stream) {
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = p.CoutPad / BN;
    if (p.CoutPad % BN != 0 || p.Kpad16 % BK16 != 0) return CP_ERR_INVALID;
    hipLaunchKernelGGL((dcn16_kernel<MT, NT, WM, WN, OCC>), dim3(tiles_m * tiles_n, p.splitk > 1 ? p.splitk : 1),
                       dim3(WM * WN * 64), 0, stream, p, tiles_m, tiles_n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

}  // namespace

// bn: N tile (64 or 128).  variant (cp_set_debug bits 1024 / 2048, tuning A/B): 0 = default shapes, 1 = the other wave
// count for that N tile.
int cp_launch_dcn16(const ConvParams& p, int bn, int variant, hipStream_t stream) {
    if (!p.offmask || !p.w16_hi || !p.w16_lo || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.nsrc != 1 ||
        p.H != p.Ho || p.W != p.Wo || p.Cin % BK16 != 0)
        return CP_ERR_INVALID;
    if ((size_t)p.B * p.H * p.W * 128 >= (size_t)0xf0000000u) return CP_ERR_INVALID;  // 32-bit offsets of the records
    // measured on the dlav1_34 B=32 step (profiles/r02_dcn_ab.txt): N 64: 4 waves of 64x32, two blocks per CU (110 TFLOP/s)
    // vs 8 waves at 128 VGPRs (108); N 128: 8 waves of 64x32, one block per CU (152) vs 4 waves of 64x64 at 256 VGPRs (136)
    if (bn == 64 && p.tile_m == 64) return launch_dcn16<1, 1, 2, 2, 2>(p, stream);  // small launches: 64 x 64 tiles (ConvParams::tile_m)
    if (bn == 64) return variant ? launch_dcn16<1, 1, 4, 2, 4>(p, stream) : launch_dcn16<2, 1, 2, 2, 2>(p, stream);
    if (bn == 128) return variant ? launch_dcn16<2, 2, 2, 2, 1>(p, stream) : launch_dcn16<2, 1, 2, 4, 2>(p, stream);
    return CP_ERR_INVALID;
}
