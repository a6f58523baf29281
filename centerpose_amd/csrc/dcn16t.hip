// Patch-resident DCNv2 (modulated deformable 3x3 convolution, stride 1, pad 1, dilation 1, one deformable group) in the
// split-f16 ("f16x3") arithmetic, built for THREE workgroups per CU (three waves per SIMD).  Same operator and the same
// structure as dcn16p.hip (which replaces the reference's modulated_deformable_im2col + GEMM pair, DCNv2/src/cuda/
// dcn_v2_im2col_cuda.cu:125-195, dcn_v2_cuda.cu:42-172); what differs is the budget it is written to.
//
// Why (profiles/NOTES.md rounds 4-5, r05_pmc_sq_counters.txt).  The K step of dcn16p / dcn16s is one wave's in-order stream of
// ~82 instructions per 6 MFMAs (44 VALU of blend + split, 8 LDS gathers, 4 weight loads, scalar bookkeeping): at ~4 clocks per
// issued instruction that IS the step's duration (345 clocks measured), two waves per SIMD leave the matrix pipe busy a quarter
// of the time and the SIMD's issue ports half idle, and every phase outside the K loop (staging, set-up, epilogue: 44 % of an
// item) overlaps with one other workgroup at best.  Both kernels hold 256 registers and ~79 KB of LDS, i.e. two workgroups per
// CU.  A third wave per SIMD needs <= 168 registers and <= 53 KB:
//   * 16-channel chunks (9 K steps: one per tap) instead of 32: the staged halo is 308 + 184 + 23 pixels x 80 B = 41 KB (pixel
//     pitch 64 + 16 B = 5 bank groups, odd: the 16 lanes of a ds_read_b128 group still start in 16 different groups);
//   * ONE gather register set (32 registers, as dcn16p's 128-wide tile): the next step's 8 ds_read_b128 are issued right behind
//     the blend that consumed the set and land under this step's MFMAs and the other two waves' work;
//   * TWO weight register sets (32) instead of three: the fragments of step s + 1 are requested at the top of step s, a whole
//     step ahead, into the set step s - 1 has just consumed (chunks are walked in pairs, so the set index stays a compile-time
//     constant although a chunk has an odd number of steps);
//   * no software pipelining inside the wave beyond that: with three waves per SIMD the overlap comes from the other waves.
//   * the phases outside the K loop are a workgroup's critical path once the K loop runs near the matrix rate (in-kernel
//     timeline of the first version, profiles/r06_dcn16t_timeline.txt: 62 k clocks per patch = prologue 15.6 k + 4 stagings
//     13 k + 36 K steps 28 k + epilogue 4.7 k), so they are kept short: every load of the prologue -- activation scale (scalar),
//     offset / mask record, the whole first chunk -- is in flight before anything waits; a chunk's staging is two patch rows per
//     round with one vector add per load (the first version derived (row, column) of every float4 by a division: ~100 VALU per
//     chunk and thread); the product is computed TRANSPOSED (weights as the first operand), so a lane ends up with four
//     consecutive output channels of its own pixel per accumulator quad and the epilogue is 8 sixteen-byte stores, not 32
//     four-byte ones, with scale / shift read 16 bytes at a time behind scalar descriptors (dcn16s.hip's epilogue).
// K order is (16-channel chunk, tap): the same products as dcn16p / dcn16s / dcn16.hip in dcn16s's summation order.
// Everything else -- the bilinear set-up (45 registers per lane: corner address + 4 weights x mask x activation pre-scale per
// tap, computed 5 + 4 by the two lanes that share a pixel and exchanged by v_permlane32_swap), the exception samples blended
// into spare pixels by the staging, the block-wide switch to buffer loads when a patch has more than 184 of them, the permuted
// pixel order of a wave's fragment rows, the epilogue -- is dcn16p's, restated for the narrower chunk.
#include <type_traits>

#include "patch16_common.h"

namespace {

constexpr int T_TH = PATCH_TH, T_TW = PATCH_TW, T_HALO = 3;
constexpr int T_PW = T_TW + 2 * T_HALO, T_PH = T_TH + 2 * T_HALO, T_NPIX = T_PH * T_PW;  // 22 x 14 = 308 patch pixels
constexpr int T_CK = 16;                                                                   // channels per staged chunk
constexpr int T_PSTR = T_CK * 4 + 16;                                                      // bytes between patch pixels (80)
constexpr int T_ECAP = 184;                                                                // exception samples per block
constexpr int T_NPIX_ALL = T_NPIX + T_ECAP + T_PW + 1;                                     // + the 3 other "corners" of the last one
constexpr int T_NSTEP = 9;                                                                 // K steps (one tap x 16 channels) per chunk
constexpr int T_RQ = T_PW * (T_CK / 4);                                                    // float4 pieces of a patch row (88)
constexpr int T_ROUNDS = T_PH / 2;                                                         // staging rounds: two patch rows each (7)
static_assert(2 * T_RQ <= 256 && T_PH % 2 == 0, "two patch rows per staging round");
static_assert(T_NPIX_ALL * T_PSTR + T_ECAP * 24 + 64 <= 53 * 1024, "three workgroups per CU");
static_assert((T_PSTR / 16) % 2 == 1, "odd pixel pitch in bank groups");

__device__ __forceinline__ float4 t_ld4s(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// both 32-lane halves of `v` for every lane: {lower half's value, upper half's value} (dcn16p.hip: both_halves5)
__device__ __forceinline__ void t_both_halves5(const uint32_t (&v)[5], uint32_t (&lo)[5], uint32_t (&hi)[5]) {
    uint32_t a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], b0 = v[0], b1 = v[1], b2 = v[2], b3 = v[3], b4 = v[4];
    asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %5\n\ts_nop 1\n\tv_permlane32_swap_b32 %1, %6\n\ts_nop 1\n\t"
                 "v_permlane32_swap_b32 %2, %7\n\ts_nop 1\n\tv_permlane32_swap_b32 %3, %8\n\ts_nop 1\n\t"
                 "v_permlane32_swap_b32 %4, %9\n\ts_nop 4"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4));
    lo[0] = a0; lo[1] = a1; lo[2] = a2; lo[3] = a3; lo[4] = a4;
    hi[0] = b0; hi[1] = b1; hi[2] = b2; hi[3] = b3; hi[4] = b4;
}

// a lane id the compiler cannot see through: what is derived from it is rebuilt where it is used instead of being held (or
// spilled) across the K loops
__device__ __forceinline__ int t_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

typedef float t_f32x2 __attribute__((ext_vector_type(2)));

template <int NT>
__global__ __launch_bounds__(256, 3) void dcn16t_kernel(const ConvParams p, const int tiles_m, const int tiles_n) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    static_assert(NT == 2, "64-wide N tile");
    __shared__ __attribute__((aligned(16))) unsigned char patch[T_NPIX_ALL * T_PSTR];
    __shared__ int exc_key[T_ECAP];   // (h_lo + 1) << 16 | (w_lo + 1) of the sample's top-left corner
    __shared__ int exc_goff[T_ECAP];  // that corner's byte offset into the input tensor
    __shared__ __attribute__((aligned(16))) float exc_w[T_ECAP][4];  // its four corner weights
    __shared__ int exc_count;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: a scalar register
    const int lrow = lane >> 5, lcol = lane & 31;
    const int tile = tile_of_block(tiles_m, tiles_n);
    const int tn = tile % tiles_n;
    int tm = tile / tiles_n;
    const int txs = p.W / T_TW, tys = p.H / T_TH;
    const int tx0 = (tm % txs) * T_TW;
    tm /= txs;
    const int ty0 = (tm % tys) * T_TH, b = tm / tys;
    // the activation pre-scale's 32 scalar loads go out first; they are reduced behind the record and first-chunk loads
    const AmaxRaw amax_raw = conv_in_scale_issue(p);
    const unsigned img_px = (unsigned)p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(p.src[0], img_px * (unsigned)p.Cin * 4u);
    const __amdgpu_buffer_rsrc_t r_om = make_rsrc(p.offmask, img_px * 128u);
    const unsigned w_bytes = (unsigned)((size_t)p.CoutPad * p.Kpad16 * 2);
    const __amdgpu_buffer_rsrc_t r_wh = make_rsrc(p.w16f_hi, w_bytes), r_wl = make_rsrc(p.w16f_lo, w_bytes);
    const int cb = p.Cin * 4, rowb = p.W * cb;
    // staging geometry: a round = two patch rows = 176 float4; thread t < 176 -> row t / 88 of the pair, column (t % 88) / 4,
    // channel quad t % 4.  One byte offset per thread (first round), + 2 rows per round; the row test is a compare per load.
    // (Rebuilt from an opaque thread id where it is used: nothing of it lives across the K loops.)
    auto stage_geo = [&](int t, int* goff, int* lds, int* row0, bool* col_ok) {
        const int rp = t >= T_RQ ? 1 : 0, within = t - rp * T_RQ, col = within >> 2;
        const int gx = tx0 - T_HALO + col;
        *row0 = ty0 - T_HALO + rp;
        *col_ok = t < 2 * T_RQ && (unsigned)gx < (unsigned)p.W;
        *goff = ((b * p.H + *row0) * p.W + gx) * cb + (within & 3) * 16;
        *lds = (rp * T_PW + col) * T_PSTR + (within & 3) * 16;
    };
    auto stage_issue = [&](float4 (&sv)[T_ROUNDS], int goff, int row0, bool col_ok, int csoff) {
#pragma unroll
        for (int k = 0; k < T_ROUNDS; ++k) {
            const bool ok = col_ok && (unsigned)(row0 + 2 * k) < (unsigned)p.H;
            sv[k] = t_ld4s(r_x, ok ? (unsigned)(goff + 2 * k * rowb) : OOB, csoff);
        }
    };
    auto stage_store = [&](const float4 (&sv)[T_ROUNDS], int lds, int t) {
        if (t < 2 * T_RQ) {
#pragma unroll
            for (int k = 0; k < T_ROUNDS; ++k) *reinterpret_cast<float4*>(patch + lds + k * (2 * T_PW * T_PSTR)) = sv[k];
        }
    };

    if (tid == 0) exc_count = 0;
    // the spare pixels start as zeros: an exception sample reads its blended value with weights (1, 0, 0, 0), and the three
    // zero-weight "corners" next to it must never be NaN / Inf bit patterns left behind by an earlier kernel
    for (int i = tid; i < (T_NPIX_ALL - T_NPIX) * (T_PSTR / 16); i += 256)
        *reinterpret_cast<float4*>(patch + T_NPIX * T_PSTR + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- this lane's pixel: fragment row lane % 32 of wave w -> patch rows 2 w, 2 w + 1 in the permuted order of
    //      patch16_common.h (the 16 lanes of a ds_read_b128 group read 16 consecutive pixels of one row) ----
    const int y = ty0 + 2 * wid + patch_perm_row(lcol), x = tx0 + patch_perm_col(lcol);
    const unsigned rec = (unsigned)((b * p.H + y) * p.W + x) * 128u;  // the pixel's offset / mask record (32 floats)
    // this lane's share of the record: taps 5 lrow .. 5 lrow + 4 (slot 4 of the upper half is a dummy, tap "9")
    float od[12], omk[5];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float4 v = buf_ld4(r_om, rec + (unsigned)lrow * 40u + 16u * i);
        od[4 * i] = v.x; od[4 * i + 1] = v.y; od[4 * i + 2] = v.z; od[4 * i + 3] = v.w;
    }
    {
        const float4 v = buf_ld4(r_om, rec + 72u + (unsigned)lrow * 20u);
        omk[0] = v.x; omk[1] = v.y; omk[2] = v.z; omk[3] = v.w;
        omk[4] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_om, (int)(rec + 88u + (unsigned)lrow * 20u), 0, 0));
    }
    // the first chunk's halo, requested now (the register file is still empty) and parked in LDS behind the set-up
    float4 sv0[T_ROUNDS];
    int g0_off, g0_lds, g0_row;
    bool g0_ok;
    stage_geo(tid, &g0_off, &g0_lds, &g0_row, &g0_ok);
    stage_issue(sv0, g0_off, g0_row, g0_ok, 0);
    float afwd, ainv;
    conv_in_scale_finish(p, amax_raw, &afwd, &ainv);
    afwd = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(afwd)));  // wave-uniform: scalar registers
    ainv = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ainv)));
    __syncthreads();  // exc_count = 0 is visible

    // ---- bilinear set-up (dcn_v2_im2col_cuda.cu:25-54, 150-187): 5 tap slots per lane, then both halves swap ----
    uint32_t sq[5], sw[5][4];  // patch pixel of corner (h_lo, w_lo); corner weights x mask x activation pre-scale
    const float fy0 = (float)(y - 1), fx0 = (float)(x - 1);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        // tap 5 lrow + j = (kh, kw): lower half (0,0) (0,1) (0,2) (1,0) (1,1); upper half (1,2) (2,0) (2,1) (2,2) (-)
        const float khf = lrow ? (float)((5 + j) / 3) : (float)(j / 3);
        const float kwf = lrow ? (float)((5 + j) % 3) : (float)(j % 3);
        float h_im = (fy0 + khf) + od[2 * j];
        float w_im = (fx0 + kwf) + od[2 * j + 1];
        const bool valid = h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W && !(lrow && j == 4);
        h_im = valid ? h_im : 0.f;  // keeps the arithmetic below finite; its weights are zeroed through the mask
        w_im = valid ? w_im : 0.f;
        const float mk = valid ? omk[j] * afwd : 0.f;
        const float fh = floorf(h_im), fw = floorf(w_im);
        const int h_lo = (int)fh, w_lo = (int)fw;
        const float lh = h_im - fh, lw = w_im - fw;
        const float hh = 1.f - lh, hw = 1.f - lw;
        sw[j][0] = __float_as_uint(hh * hw * mk);
        sw[j][1] = __float_as_uint(hh * lw * mk);
        sw[j][2] = __float_as_uint(lh * hw * mk);
        sw[j][3] = __float_as_uint(lh * lw * mk);
        const int qy = h_lo - (ty0 - T_HALO), qx = w_lo - (tx0 - T_HALO);
        const bool inp = (unsigned)qy <= (unsigned)(T_PH - 2) && (unsigned)qx <= (unsigned)(T_PW - 2);
        int q = inp ? qy * T_PW + qx : 0;
        if (valid && !inp) {  // exception sample: file its corner and weights; the staging blends it into spare pixel e,
                              // which this lane then reads with weights (1, 0, 0, 0)
            const int e = atomicAdd(&exc_count, 1);
            if (e < T_ECAP) {
                exc_key[e] = ((h_lo + 1) << 16) | (w_lo + 1);
                exc_goff[e] = ((b * p.H + h_lo) * p.W + w_lo) * cb;
                *reinterpret_cast<float4*>(exc_w[e]) = make_float4(__uint_as_float(sw[j][0]), __uint_as_float(sw[j][1]),
                                                                   __uint_as_float(sw[j][2]), __uint_as_float(sw[j][3]));
                sw[j][0] = __float_as_uint(1.f);
                sw[j][1] = sw[j][2] = sw[j][3] = 0u;
                q = T_NPIX + e;
            }
        }
        sq[j] = (uint32_t)q;
    }
    int addr[9];       // byte address in `patch` of corner (h_lo, w_lo) + this lane's 32-byte channel half; in the
                       // buffer-load mode: that corner's byte offset into the input tensor | 4 corner-validity bits
    t_f32x2 bw[9][2];  // {w1, w2}, {w3, w4}: corner weights x mask x activation pre-scale
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint32_t pack[5] = {sq[j], sw[j][0], sw[j][1], sw[j][2], sw[j][3]};
        uint32_t lo[5], hi[5];
        t_both_halves5(pack, lo, hi);
        addr[j] = (int)lo[0] * T_PSTR + lrow * 32;
        if (j < 4) addr[5 + j] = (int)hi[0] * T_PSTR + lrow * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bw[j][c >> 1][c & 1] = __uint_as_float(lo[1 + c]);
            if (j < 4) bw[5 + j][c >> 1][c & 1] = __uint_as_float(hi[1 + c]);
        }
    }
    stage_store(sv0, g0_lds, tid);  // (pixels 0 .. 307: disjoint from the spare pixels zeroed above)
    __syncthreads();
    const int nexc_all = __builtin_amdgcn_readfirstlane(exc_count);  // scalar: the mode branches below stay uniform
    const bool slow = nexc_all > T_ECAP;                             // block-uniform
    const int nexc = nexc_all < T_ECAP ? nexc_all : T_ECAP;

    // ---- weight fragments: (n tile j of 32, K step g of 16) = 1 KB in lane order at ((j G + g) 64 + lane) 16 B ----
    // step u of chunk ch = tap u, channels 16 ch .. + 15: K step g = u (Cin / 16) + ch.  Two register sets: step s of the patch
    // (s = 9 ch + u) uses set s % 2 and, at its top, requests step s + 1 into the other set, which step s - 1 has just consumed.
    const int G = p.Kpad16 / 16, gpt = p.Cin / 16;
    const int nch = p.Cin / T_CK;  // even (Cin % 32 == 0)
    unsigned b_voff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) b_voff[j] = (unsigned)(((tn * NT + j) * G) * 1024 + lane * 16);
    u32x4 wbh[2][NT], wbl[2][NT];
    // (No branch around the loads: past the last chunk the scalar offset points beyond the descriptor -- zeros, no memory
    // traffic.  With `if (ch >= nch) return` in every step the compiler's s_waitcnt pass merged the two paths conservatively and
    // made every step's MFMAs wait for the loads issued at the top of that SAME step: vmcnt(3..0) instead of vmcnt(7..4).)
    auto issue_w = [&](int set, int ch, int u) {
        int so = (u * gpt + ch) * 1024;
        if (u >= T_NSTEP) so = ch + 1 < nch ? ((u - T_NSTEP) * gpt + ch + 1) * 1024 : 0x7ffffff0;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            wbh[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wh, (int)b_voff[j], so, 0);
            wbl[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wl, (int)b_voff[j], so, 0);
        }
    };

    acc_t acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < F::NACC; ++r) acc[0][j][r] = 0.f;

    // blend + split of one gathered K step (fma(w4, v4, fma(w3, v3, fma(w2, v2, w1 * v1))) per channel: dcn16.hip's order, plain
    // v_fma_f32), `mid` (the next gather, into the registers the blend has just freed), then the step's 3 NT MFMAs in the term
    // order of igemm16.hip (lo*hi, hi*lo, hi*hi)
    auto mma_step = [&](const float4 (&r)[4][2], const t_f32x2 (&w)[2], const u32x4 (&bh)[NT], const u32x4 (&bl)[NT], auto&& mid) {
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            const float4 v1 = r[0][hq], v2 = r[1][hq], v3 = r[2][hq], v4 = r[3][hq];
            const float w1 = w[0].x, w2 = w[0].y, w3 = w[1].x, w4 = w[1].y;
            const float o0 = fmaf(w4, v4.x, fmaf(w3, v3.x, fmaf(w2, v2.x, w1 * v1.x)));
            const float o1 = fmaf(w4, v4.y, fmaf(w3, v3.y, fmaf(w2, v2.y, w1 * v1.y)));
            const float o2 = fmaf(w4, v4.z, fmaf(w3, v3.z, fmaf(w2, v2.z, w1 * v1.z)));
            const float o3 = fmaf(w4, v4.w, fmaf(w3, v3.w, fmaf(w2, v2.w, w1 * v1.w)));
            const Split2 t0 = split2(o0, o1), t1 = split2(o2, o3);
            hi[2 * hq] = t0.hi; hi[2 * hq + 1] = t1.hi;
            lo[2 * hq] = t0.lo; lo[2 * hq + 1] = t1.lo;
        }
        const u32x4 ahv = {hi[0], hi[1], hi[2], hi[3]}, alv = {lo[0], lo[1], lo[2], lo[3]};
        const h8 ah = *reinterpret_cast<const h8*>(&ahv), al = *reinterpret_cast<const h8*>(&alv);
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
        // weights as the first operand: the accumulators hold the TRANSPOSED tile (rows = output channels, columns = this wave's
        // pixels) -- accumulator 4 g + i of N tile j in lane (pixel lane % 32, half h4) = channel 32 j + 8 g + 4 h4 + i of that pixel
#pragma unroll
        for (int j = 0; j < NT; ++j)
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&bh[j]), al, acc[0][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j)
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&bl[j]), ah, acc[0][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j)
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&bh[j]), ah, acc[0][j], 0, 0, 0);
    };

    if (!slow) {
        // ================= fast mode: every sample is in LDS =================
        auto gather = [&](float4 (&r)[4][2], int a) {
            const unsigned char* ap = patch + a;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int co = ((c >> 1) * T_PW + (c & 1)) * T_PSTR;
                r[c][0] = *reinterpret_cast<const float4*>(ap + co);
                r[c][1] = *reinterpret_cast<const float4*>(ap + co + 16);
            }
        };
        // exception samples: 4 threads each (one channel quad per thread), 64 samples per pass: the four corners are blended here
        // (same FMA order as the K loop) into the sample's spare pixel; addresses are rebuilt per chunk from the block's list
        auto stage_exceptions = [&](int csoff, int t) {
            for (int e = t >> 2; e < nexc; e += 64) {
                const int key = exc_key[e], go = exc_goff[e] + (t & 3) * 16;
                const float4 w = *reinterpret_cast<const float4*>(exc_w[e]);
                const int iy = (key >> 16) - 1, ix = (key & 0xffff) - 1;
                const bool y0 = (unsigned)iy < (unsigned)p.H, y1 = (unsigned)(iy + 1) < (unsigned)p.H;
                const bool x0 = (unsigned)ix < (unsigned)p.W, x1 = (unsigned)(ix + 1) < (unsigned)p.W;
                const float4 v1 = t_ld4s(r_x, (y0 && x0) ? (unsigned)go : OOB, csoff);
                const float4 v2 = t_ld4s(r_x, (y0 && x1) ? (unsigned)(go + cb) : OOB, csoff);
                const float4 v3 = t_ld4s(r_x, (y1 && x0) ? (unsigned)(go + rowb) : OOB, csoff);
                const float4 v4 = t_ld4s(r_x, (y1 && x1) ? (unsigned)(go + rowb + cb) : OOB, csoff);
                float4 o;
                o.x = fmaf(w.w, v4.x, fmaf(w.z, v3.x, fmaf(w.y, v2.x, w.x * v1.x)));
                o.y = fmaf(w.w, v4.y, fmaf(w.z, v3.y, fmaf(w.y, v2.y, w.x * v1.y)));
                o.z = fmaf(w.w, v4.z, fmaf(w.z, v3.z, fmaf(w.y, v2.z, w.x * v1.z)));
                o.w = fmaf(w.w, v4.w, fmaf(w.z, v3.w, fmaf(w.y, v2.w, w.x * v1.w)));
                *reinterpret_cast<float4*>(patch + (T_NPIX + e) * T_PSTR + (t & 3) * 16) = o;
            }
        };
        // one chunk: stage (all five loads of a thread in flight, then the exception samples, then the LDS stores), barrier, the
        // 9 K steps.  PAR = chunk parity: step u uses weight set (PAR + u) % 2
        auto chunk = [&](auto par_c, int ch) {
            constexpr int PAR = decltype(par_c)::value;
            const int csoff = ch * (T_CK * 4);
            const int t = t_opaque(tid);
            if (ch > 0) {
                __syncthreads();  // every wave is done with the previous chunk's patch
                float4 sv[T_ROUNDS];
                int goff, lds, row0;
                bool col_ok;
                stage_geo(t, &goff, &lds, &row0, &col_ok);
                stage_issue(sv, goff, row0, col_ok, csoff);
                stage_exceptions(csoff, t);
                stage_store(sv, lds, t);
            } else {
                stage_exceptions(csoff, t);  // (chunk 0's halo was requested at the top of the kernel and is in LDS already)
            }
            __syncthreads();
            float4 raw[4][2];
            gather(raw, addr[0]);
#pragma unroll
            for (int u = 0; u < T_NSTEP; ++u) {
                // the order of the phases is pinned (sched_barrier): left alone, the scheduler sinks every load to just above its
                // first use -- no lead at all
                issue_w((PAR + u + 1) & 1, ch, u + 1);  // step s + 1's fragments, a whole step ahead
                __builtin_amdgcn_sched_barrier(0);
                mma_step(raw, bw[u], wbh[(PAR + u) & 1], wbl[(PAR + u) & 1], [&]() {
                    if (u + 1 < T_NSTEP) gather(raw, addr[u + 1]);
                });
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        issue_w(0, 0, 0);
        for (int ch = 0; ch < nch; ch += 2) {
            chunk(std::integral_constant<int, 0>(), ch);
            chunk(std::integral_constant<int, 1>(), ch + 1);
        }
    } else {
        // ================= buffer-load mode: every block sample through the texture path =================
        // corner offsets | validity bits and the weights of the lane's 9 taps, from the record again: the fast set-up
        // does not keep the offsets, and it replaced the weights of the samples it filed as exceptions
        {
            float o9[28];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const float4 v = buf_ld4(r_om, rec + 16u * i);
                o9[4 * i] = v.x; o9[4 * i + 1] = v.y; o9[4 * i + 2] = v.z; o9[4 * i + 3] = v.w;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float h_im = (float)(y - 1 + t / 3) + o9[2 * t];
                const float w_im = (float)(x - 1 + t % 3) + o9[2 * t + 1];
                int gb = 0;
                float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
                if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
                    const int h_lo = (int)floorf(h_im), w_lo = (int)floorf(w_im);
                    const float lh = h_im - (float)h_lo, lw = w_im - (float)w_lo;
                    const float hh = 1.f - lh, hw = 1.f - lw, mk = o9[18 + t] * afwd;
                    int vm = 0;
                    if (h_lo >= 0 && w_lo >= 0) vm |= 1;
                    if (h_lo >= 0 && w_lo + 1 <= p.W - 1) vm |= 2;
                    if (h_lo + 1 <= p.H - 1 && w_lo >= 0) vm |= 4;
                    if (h_lo + 1 <= p.H - 1 && w_lo + 1 <= p.W - 1) vm |= 8;
                    gb = (((b * p.H + h_lo) * p.W + w_lo) * cb) | vm;
                    w1 = hh * hw * mk; w2 = hh * lw * mk; w3 = lh * hw * mk; w4 = lh * lw * mk;
                }
                addr[t] = gb;
                bw[t][0] = t_f32x2{w1, w2};
                bw[t][1] = t_f32x2{w3, w4};
            }
        }
        // (the rare mode: no lead for the weights -- one register set, requested with the step's samples -- so that it does not set
        // the kernel's register budget; the fast mode above does)
        for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
            for (int u = 0; u < T_NSTEP; ++u) {
                float4 r[4][2];
                const int so = ch * (T_CK * 4);
                const int base = (addr[u] & ~15) + lrow * 32;
                issue_w(0, ch, u);
#pragma unroll
                for (int c = 0; c < 4; ++c) {  // invalid corners out of range (-> 0)
                    const int gi = (addr[u] & (1 << c)) ? base + (c >> 1) * rowb + (c & 1) * cb : (int)OOB_BASE;
                    r[c][0] = t_ld4s(r_x, (unsigned)gi, so);
                    r[c][1] = t_ld4s(r_x, (unsigned)gi + 16u, so);
                }
                __builtin_amdgcn_sched_barrier(0);
                mma_step(r, bw[u], wbh[0], wbl[0], []() {});
                __builtin_amdgcn_sched_barrier(0);  // keep the loads of later steps below
            }
        }
    }
    // ---- epilogue (dcn16s.hip's, on this kernel's pixel order): y = acc * scale[n] * 2^-e_a + shift[n] -> ReLU -> NHWC, one
    //      16-byte store per (N tile, channel group of 8): 2 NT x 4 per lane ----
    {
        // (the lane id is rebuilt, not kept in a register across the K loops)
        const int ln = t_opaque((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))), h4 = ln >> 5, lc = ln & 31;
        // fragment row lc of wave w = pixel (2 w + patch_perm_row(lc), patch_perm_col(lc)) of the patch
        const int pix0 = __builtin_amdgcn_readfirstlane((b * p.H + ty0 + 2 * wid) * p.W + tx0);
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + (size_t)pix0 * p.ldo + p.coff, (unsigned)((p.W + T_TW) * p.ldo) * 4u);
        const __amdgpu_buffer_rsrc_t r_sc = make_rsrc(p.scale, (unsigned)p.CoutPad * 4u), r_sh = make_rsrc(p.shift, (unsigned)p.CoutPad * 4u);
        const unsigned vpix = (unsigned)((patch_perm_row(lc) * p.W + patch_perm_col(lc)) * p.ldo) * 4u;
        const int nb0 = tn * (32 * NT);  // first channel of this N tile (scalar)
        const bool relu = p.act == CP_ACT_RELU;
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int nsc = (nb0 + 32 * j + 8 * g4) * 4;  // bytes (scalar); + 16 h4 per lane half
                const float4 sc = p.scale ? t_ld4s(r_sc, (unsigned)(16 * h4), nsc) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float4 sh = p.shift ? t_ld4s(r_sh, (unsigned)(16 * h4), nsc) : make_float4(0.f, 0.f, 0.f, 0.f);
                float v[4] = {acc[0][j][4 * g4] * (sc.x * ainv) + sh.x, acc[0][j][4 * g4 + 1] * (sc.y * ainv) + sh.y,
                              acc[0][j][4 * g4 + 2] * (sc.z * ainv) + sh.z, acc[0][j][4 * g4 + 3] * (sc.w * ainv) + sh.w};
                if (relu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                }
                const int n0 = nb0 + 32 * j + 8 * g4 + 4 * h4;  // (Cout % 4 == 0: a quad is inside or outside)
                const bool n_ok = n0 < p.Cout;
#pragma unroll
                for (int i = 0; i < 4; ++i) amax = fmaxf(amax, n_ok ? fabsf(v[i]) : 0.f);
                const u32x4 pk4 = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                __builtin_amdgcn_raw_buffer_store_b128(pk4, ro, (int)(n_ok ? vpix + (unsigned)n0 * 4u : 0x80000000u), 0, 0);
            }
        if (p.out_amax) cp_amax_commit(p.out_amax, amax);
    }
}

}  // namespace

// dcn16s's conditions: dcn16p's (full 8 x 16 patches, NHWC output, no split-K, 32-bit offsets, fragment-ordered weights) plus no
// residual / GroupNorm statistics, ReLU or no activation, whole channel quads (the 16-byte stores); Cin % 32 == 0 makes the
// chunk count even
bool cp_dcn16t_supported(const ConvParams& p) { return cp_dcn16s_supported(p); }

int cp_launch_dcn16t(const ConvParams& p, hipStream_t stream) {
    if (!cp_dcn16t_supported(p)) return CP_ERR_INVALID;
    const int tiles_m = p.B * (p.H / T_TH) * (p.W / T_TW), tiles_n = p.CoutPad / 64;
    hipLaunchKernelGGL((dcn16t_kernel<2>), dim3(tiles_m * tiles_n), dim3(256), 0, stream, p, tiles_m, tiles_n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}
