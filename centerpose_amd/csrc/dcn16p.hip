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
//     so lanes reading neighbouring pixels at the same channel quad spread over the banks; no XOR swizzle, hence ONE
//     address register per (lane, tap) and every corner / K step / quad is an immediate offset.
//   * A lane gathers exactly its MFMA A-fragment (pixel = lane % 32, 8 consecutive channels = 2 quads per corner),
//     blends in float32 (plain v_fma_f32: the packed form measured 4 % slower), splits to binary16 hi / lo in registers: the A tile never exists in LDS.
//     Each wave owns 32 pixels x the whole N tile, so nothing is gathered twice inside a block.  The N tile is 64 wide (NT = 2)
//     or -- round 5, layers with whole 128-channel tiles -- 128 wide (NT = 4): the same gather / blend / split then feeds twice
//     the MFMAs (see the weight-set comment in the kernel and cp_dcn16p_wide).
//   * The weight fragments come straight from global memory / L2 in MFMA operand order (ConvParams::w16f_*: one
//     coalesced 1 KB load per fragment, cp_launch_frag16_repack), three K steps ahead.  With neither operand staged
//     through a shared tile the K loop has NO barrier inside a chunk: the four waves of a block drift apart and one
//     wave's gather / blend overlaps another's MFMAs.  (The first version kept the weights in a double-buffered LDS
//     tile with a barrier per tap: 19 % MFMA busy, waves 45 % parked at s_waitcnt / s_barrier.)
//   * The gather of K step u + 1 is issued before the blend / MFMAs of step u (two register sets).
//   * The bilinear set-up (corner address, 4 weights with mask and activation pre-scale folded in) of a lane's 9 taps
//     lives in 45 registers for the whole kernel.  The two lanes that share a pixel (the two 8-channel halves of a
//     K step) compute 5 and 4 taps each and swap the results with v_permlane32_swap.
//   * Samples whose 2x2 corner block leaves the staged halo (|offset| > 2..3 px at the patch border: 2 % of samples
//     at sigma = 1.5 px, 7 % at 1.9 -- the synthetic network's layers measure 1.4 .. 1.93) are "exceptions": the
//     set-up appends them (corner, 4 weights) to a block list (LDS atomic) and takes weights (1, 0, 0, 0) itself; the
//     staging blends each one's four corners per chunk into a spare patch pixel, which the K loop then reads like any
//     other corner -- no branch.  A block with more than ECAP = 184 exceptions (offsets of sigma > 3 px everywhere)
//     switches, as a whole, to gathering through buffer loads like dcn16.hip (slower, same results).
// K order is (32-channel chunk, tap, 16-channel half): same products as dcn16.hip, different summation order.
// LDS: (308 + 207) x 144 B + 4.4 KB of lists = 78.6 KB => two blocks per CU, whose staging / compute phases overlap.
#include <type_traits>

#include "patch16_common.h"



namespace {

constexpr int TH = PATCH_TH, TW = PATCH_TW, HALO = 3;
constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO, NPIX = PH * PW;  // 22 x 14 = 308 patch pixels
constexpr int CKC = 32;                                                // channels per staged chunk
constexpr int PSTR = CKC * 4 + 16;                                     // bytes between patch pixels
constexpr int ECAP = 184;                                              // exception samples per block (one spare pixel each)
constexpr int NPIX_ALL = NPIX + ECAP + PW + 1;                         // + the 3 other "corners" of the last one: 515
constexpr int NSTEP = 18;                                              // K steps (16 channels of one tap) per chunk

__device__ __forceinline__ float4 buf_ld4s(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// both 32-lane halves of `v` for every lane: {lower half's value, upper half's value}
__device__ __forceinline__ void both_halves5(const uint32_t (&v)[5], uint32_t (&lo)[5], uint32_t (&hi)[5]) {
    // v_permlane32_swap_b32 vdst, vsrc exchanges vdst[32..63] with vsrc[0..31]; with both operands holding v every lane ends up
    // with {the lower half's value, the upper half's value}.  Written out by hand, each swap on its own pair of registers with
    // the wait states the hazard table asks for inside the statement (VALU write -> v_permlane*_swap read: 2).  (Round 4 padded
    // these swaps while hunting wrong set-up values; the swaps were innocent -- the cause was a packed-f32 op with a set op_sel
    // bit, profiles/NOTES.md round 5 -- but the hand-written form costs nothing and stays.)
    uint32_t a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], b0 = v[0], b1 = v[1], b2 = v[2], b3 = v[3], b4 = v[4];
    asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %5\n\ts_nop 1\n\tv_permlane32_swap_b32 %1, %6\n\ts_nop 1\n\t"
                 "v_permlane32_swap_b32 %2, %7\n\ts_nop 1\n\tv_permlane32_swap_b32 %3, %8\n\ts_nop 1\n\t"
                 "v_permlane32_swap_b32 %4, %9\n\ts_nop 4"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4));
    lo[0] = a0; lo[1] = a1; lo[2] = a2; lo[3] = a3; lo[4] = a4;
    hi[0] = b0; hi[1] = b1; hi[2] = b2; hi[3] = b3; hi[4] = b4;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NT>
__global__ __launch_bounds__(256, 2) void dcn16p_kernel(const ConvParams p, const int tiles_m, const int tiles_n) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    __shared__ __attribute__((aligned(16))) unsigned char patch[NPIX_ALL * PSTR];
    __shared__ int exc_key[ECAP];   // (h_lo + 1) << 16 | (w_lo + 1) of the sample's top-left corner
    __shared__ int exc_goff[ECAP];  // that corner's byte offset into the input tensor (may be "before" it: see validity)
    __shared__ __attribute__((aligned(16))) float exc_w[ECAP][4];  // its four corner weights
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
    const __amdgpu_buffer_rsrc_t r_wh = make_rsrc(p.w16f_hi, w_bytes), r_wl = make_rsrc(p.w16f_lo, w_bytes);
    const int cb = p.Cin * 4, rowb = p.W * cb;
    // staging geometry: thread -> (patch column tid / 8 (< PW: 22 of 32 busy), channel quad tid % 8), pass s = patch
    // row s; one base offset register, the row validity is wave-uniform
    const int spx = tid >> 3, six = tx0 - HALO + spx;
    const bool col_ok = spx < PW && (unsigned)six < (unsigned)p.W;
    const int st_base = ((b * p.H + ty0) * p.W + six) * cb + (tid & 7) * 16;  // row ty0 (always inside the image)
    const int st_lds = spx * PSTR + (tid & 7) * 16;                            // + PW * PSTR per row

    if (tid == 0) exc_count = 0;
    // the spare pixels start as zeros: an exception sample reads its blended value with weights (1, 0, 0, 0), and the three
    // zero-weight "corners" next to it must never be NaN / Inf bit patterns left behind by an earlier kernel
    for (int i = tid; i < (NPIX_ALL - NPIX) * (PSTR / 16); i += 256)
        *reinterpret_cast<float4*>(patch + NPIX * PSTR + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- this lane's pixel: fragment row lane % 32 of wave w -> patch rows 2 w, 2 w + 1 in the permuted order of
    //      patch16_common.h: the 16 lanes of a ds_read_b128 group read 16 consecutive pixels of one row, whose 144-byte
    //      pitch spreads them over all 16 bank groups (the plain order put 6 of 16 lanes on an occupied group: 54 % of
    //      the LDS cycles were conflicts) ----
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
        const int qy = h_lo - (ty0 - HALO), qx = w_lo - (tx0 - HALO);
        const bool inp = (unsigned)qy <= (unsigned)(PH - 2) && (unsigned)qx <= (unsigned)(PW - 2);
        int q = inp ? qy * PW + qx : 0;
        if (valid && !inp) {  // exception sample: file its corner and weights; the staging blends it into spare pixel e,
                              // which this lane then reads with weights (1, 0, 0, 0)
            const int e = atomicAdd(&exc_count, 1);
            if (e < ECAP) {
                exc_key[e] = ((h_lo + 1) << 16) | (w_lo + 1);
                exc_goff[e] = ((b * p.H + h_lo) * p.W + w_lo) * cb;
                *reinterpret_cast<float4*>(exc_w[e]) = make_float4(__uint_as_float(sw[j][0]), __uint_as_float(sw[j][1]),
                                                                   __uint_as_float(sw[j][2]), __uint_as_float(sw[j][3]));
                sw[j][0] = __float_as_uint(1.f);
                sw[j][1] = sw[j][2] = sw[j][3] = 0u;
                q = NPIX + e;
            }
        }
        sq[j] = (uint32_t)q;
    }
    int addr[9];     // byte address in `patch` of corner (h_lo, w_lo) + this lane's 32-byte channel half; in the
                     // buffer-load mode: that corner's byte offset into the input tensor | 4 corner-validity bits
    f32x2 bw[9][2];  // {w1, w2}, {w3, w4}: corner weights x mask x activation pre-scale
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint32_t pack[5] = {sq[j], sw[j][0], sw[j][1], sw[j][2], sw[j][3]};
        uint32_t lo[5], hi[5];
        both_halves5(pack, lo, hi);
        addr[j] = (int)lo[0] * PSTR + lrow * 32;
        if (j < 4) addr[5 + j] = (int)hi[0] * PSTR + lrow * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bw[j][c >> 1][c & 1] = __uint_as_float(lo[1 + c]);
            if (j < 4) bw[5 + j][c >> 1][c & 1] = __uint_as_float(hi[1 + c]);
        }
    }
    __syncthreads();
    const int nexc_all = __builtin_amdgcn_readfirstlane(exc_count);  // scalar: the mode branches below stay uniform
    const bool slow = nexc_all > ECAP;                               // block-uniform
    const int nexc = nexc_all < ECAP ? nexc_all : ECAP;

    // ---- weight fragments: (n tile j of 32, K step g of 16) = 1 KB in lane order at ((j G + g) 64 + lane) 16 B ----
    const int G = p.Kpad16 / 16, gpt = p.Cin / 16;  // K steps per weight row / per tap
    const int nch_w = p.Cin / CKC;
    unsigned b_voff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) b_voff[j] = (unsigned)(((tn * NT + j) * G) * 1024 + lane * 16);
    // step u of chunk ch = (tap u / 2, 16-channel half u % 2): K step g = tap * (Cin / 16) + 2 ch + (u % 2)
    // A weight register set holds the fragments of NW = 2 N tiles of one K step.  NT = 2: one set per step, set u % 3.  NT = 4 (the
    // 128-wide N tile, round 5): a step is NH = 2 half-steps -- the same A fragment against N tiles {0, 1}, then {2, 3} -- and the
    // sets rotate per half-step, so the weights' lead (two half-steps = 12 MFMAs) and their registers are those of the 64-wide kernel
    // while the gather, the blend and the split of a step are done once for twice the outputs.  (18 NH half-steps per chunk keep the
    // rotation consistent across chunks.)
    constexpr int NW = 2, NH = NT / NW, NHS = NSTEP * NH;
    static_assert(NT % NW == 0 && NHS % 3 == 0, "weight set rotation");
    u32x4 wbh[3][NW], wbl[3][NW];
    auto issue_b = [&](int set, int g, int h) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            wbh[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wh, (int)b_voff[h * NW + j], g * 1024, 0);
            wbl[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wl, (int)b_voff[h * NW + j], g * 1024, 0);
        }
    };
    // half-step s of chunk ch (s may run into the next chunk): K step u = s / NH = (tap u / 2, 16-channel half u % 2), N-tile pair s % NH
    auto issue_hs = [&](int s, int ch) {
        if (s >= NHS) { s -= NHS; ++ch; }
        if (ch >= nch_w) return;
        const int u = s / NH;
        issue_b(s % 3, (u >> 1) * gpt + 2 * ch + (u & 1), s % NH);
    };

    acc_t acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < F::NACC; ++r) acc[0][j][r] = 0.f;
    const int nch = p.Cin / CKC;

    // blend + split of one gathered K step, then its 3 NT MFMAs
    // `mid(h)` runs between the blend and the MFMAs of half-step h: the refill of the weight set the PREVIOUS half-step consumed (two
    // half-steps ahead; round 4 moved it here from behind the MFMAs on a suspicion that round 5 cleared -- tools/probe/
    // mfma_war_probe.hip finds no write-after-read hazard on MFMA operands -- and it measured the same in both places) and, in
    // the 128-wide kernel, the gather of the next step into the registers the blend has just freed.
    auto mma_step = [&](const float4 (&r)[4][2], const f32x2 (&w)[2], int s0, auto&& mid) {
        // fma(w4, v4, fma(w3, v3, fma(w2, v2, w1 * v1))) per channel (dcn16.hip's order), two per v_pk_fma_f32
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            const float4 v1 = r[0][hq], v2 = r[1][hq], v3 = r[2][hq], v4 = r[3][hq];
            // plain v_fma_f32: same products in the same order as the v_pk_fma_f32 form this replaced, which measured 4 % slower
            // -- packed float32 VALU beside MFMAs is an anti-lever on this part (MI355X_MICROARCH.md, instruction table)
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
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const u32x4 (&bh)[NW] = wbh[(s0 + h) % 3];
            const u32x4 (&bl)[NW] = wbl[(s0 + h) % 3];
            __builtin_amdgcn_sched_barrier(0);
            mid(h);
            __builtin_amdgcn_sched_barrier(0);
            // same term order as igemm16.hip (lo*hi, hi*lo, hi*hi)
#pragma unroll
            for (int j = 0; j < NW; ++j)
                acc[0][h * NW + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, *reinterpret_cast<const h8*>(&bh[j]), acc[0][h * NW + j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NW; ++j)
                acc[0][h * NW + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, *reinterpret_cast<const h8*>(&bl[j]), acc[0][h * NW + j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NW; ++j)
                acc[0][h * NW + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, *reinterpret_cast<const h8*>(&bh[j]), acc[0][h * NW + j], 0, 0, 0);
        }
    };

    if (!slow) {
        // ================= fast mode: every sample is in LDS =================
        auto gather = [&](float4 (&r)[4][2], int a) {  // a: addr[tap] + 64 (K step % 2)
            const unsigned char* ap = patch + a;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int co = ((c >> 1) * PW + (c & 1)) * PSTR;
                r[c][0] = *reinterpret_cast<const float4*>(ap + co);
                r[c][1] = *reinterpret_cast<const float4*>(ap + co + 16);
            }
        };
        // exception samples: 8 threads each (one channel quad per thread), 32 samples per pass: the four corners are
        // blended here (same FMA order as the K loop) into the sample's spare pixel; addresses are rebuilt per chunk from
        // the block's list (LDS) -- no registers held across the K loop
        auto stage_exceptions = [&](int csoff) {
            for (int e = tid >> 3; e < nexc; e += 32) {
                const int key = exc_key[e], go = exc_goff[e] + (tid & 7) * 16;
                const float4 w = *reinterpret_cast<const float4*>(exc_w[e]);
                const int iy = (key >> 16) - 1, ix = (key & 0xffff) - 1;
                const bool y0 = (unsigned)iy < (unsigned)p.H, y1 = (unsigned)(iy + 1) < (unsigned)p.H;
                const bool x0 = (unsigned)ix < (unsigned)p.W, x1 = (unsigned)(ix + 1) < (unsigned)p.W;
                const float4 v1 = buf_ld4s(r_x, (y0 && x0) ? (unsigned)go : OOB, csoff);
                const float4 v2 = buf_ld4s(r_x, (y0 && x1) ? (unsigned)(go + cb) : OOB, csoff);
                const float4 v3 = buf_ld4s(r_x, (y1 && x0) ? (unsigned)(go + rowb) : OOB, csoff);
                const float4 v4 = buf_ld4s(r_x, (y1 && x1) ? (unsigned)(go + rowb + cb) : OOB, csoff);
                float4 o;
                o.x = fmaf(w.w, v4.x, fmaf(w.z, v3.x, fmaf(w.y, v2.x, w.x * v1.x)));
                o.y = fmaf(w.w, v4.y, fmaf(w.z, v3.y, fmaf(w.y, v2.y, w.x * v1.y)));
                o.z = fmaf(w.w, v4.z, fmaf(w.z, v3.z, fmaf(w.y, v2.z, w.x * v1.z)));
                o.w = fmaf(w.w, v4.w, fmaf(w.z, v3.w, fmaf(w.y, v2.w, w.x * v1.w)));
                *reinterpret_cast<float4*>(patch + (NPIX + e) * PSTR + (tid & 7) * 16) = o;
            }
        };
        // weights of half-steps 0 and 1 in flight before the first chunk is staged (half-step s refills set (s + 2) % 3 with s + 2)
        issue_hs(0, 0);
        issue_hs(1, 0);
        for (int ch = 0; ch < nch; ++ch) {
            if (ch > 0) __syncthreads();  // every wave is done with the previous chunk's patch
            {
                const int csoff = ch * (CKC * 4);
                int sb = st_base;
                asm volatile("" : "+v"(sb));  // per chunk: keeps the 14 row offsets from being hoisted into 14 registers
                auto row_off = [&](int s) -> unsigned {  // patch row s of this thread's column, or out of range (-> 0)
                    const bool row_ok = (unsigned)(ty0 - HALO + s) < (unsigned)p.H;
                    return (row_ok && col_ok) ? (unsigned)(sb + (s - HALO) * rowb) : OOB;
                };
                // two rounds of 7 rows: half the registers in flight (one round of 14 measured the same)
                constexpr int H1 = PH / 2;
                {
                    float4 sv[H1];
#pragma unroll
                    for (int s = 0; s < H1; ++s) sv[s] = buf_ld4s(r_x, row_off(s), csoff);
#pragma unroll
                    for (int s = 0; s < H1; ++s)
                        if (spx < PW) *reinterpret_cast<float4*>(patch + st_lds + s * (PW * PSTR)) = sv[s];
                }
                {
                    float4 sv[PH - H1 > 0 ? PH - H1 : 1];
#pragma unroll
                    for (int s = H1; s < PH; ++s) sv[s - H1] = buf_ld4s(r_x, row_off(s), csoff);
                    stage_exceptions(csoff);
#pragma unroll
                    for (int s = H1; s < PH; ++s)
                        if (spx < PW) *reinterpret_cast<float4*>(patch + st_lds + s * (PW * PSTR)) = sv[s - H1];
                }
            }
            __syncthreads();
            // NT = 2: two gather register sets, step u + 1 requested before the blend of step u.  NT = 4: ONE set (the accumulators
            // took the other's 32 registers): step u + 1 is requested right after the blend of step u has consumed it, in front of
            // that step's 12 MFMAs, which cover the LDS round trip
            constexpr int RS = NT >= 4 ? 1 : 2;
            float4 raw[RS][4][2];
            gather(raw[0], addr[0]);
#pragma unroll
            for (int u = 0; u < NSTEP; ++u) {
                // the order of the three phases is pinned (sched_barrier): left alone, the scheduler sinks every load to
                // just above its first use -- no prefetch, a full LDS / L2 round trip exposed per step
                if (RS == 2 && u + 1 < NSTEP) gather(raw[(u + 1) % RS], addr[(u + 1) >> 1] + ((u + 1) & 1) * 64);
                __builtin_amdgcn_sched_barrier(0);
                mma_step(raw[u % RS], bw[u >> 1], u * NH, [&](int h) {
                    if (RS == 1 && h == 0 && u + 1 < NSTEP) gather(raw[0], addr[(u + 1) >> 1] + ((u + 1) & 1) * 64);
                    // the set half-step s - 1 consumed takes half-step s + 2 (of this chunk or the next)
                    issue_hs(u * NH + h + 2, ch);
                });
                __builtin_amdgcn_sched_barrier(0);
            }
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
                bw[t][0] = f32x2{w1, w2};
                bw[t][1] = f32x2{w3, w4};
            }
        }
        issue_hs(0, 0);
        issue_hs(1, 0);
        for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
            for (int u = 0; u < NSTEP; ++u) {
                const int t = u >> 1, ks = u & 1;
                float4 r[4][2];
                const int so = (ch * CKC + ks * 16) * 4;
                const int base = (addr[t] & ~15) + lrow * 32;
#pragma unroll
                for (int c = 0; c < 4; ++c) {  // invalid corners out of range (-> 0)
                    const int gi = (addr[t] & (1 << c)) ? base + (c >> 1) * rowb + (c & 1) * cb : (int)OOB_BASE;
                    r[c][0] = buf_ld4s(r_x, (unsigned)gi, so);
                    r[c][1] = buf_ld4s(r_x, (unsigned)gi + 16u, so);
                }
                mma_step(r, bw[t], u * NH, [&](int h) { issue_hs(u * NH + h + 2, ch); });
                __builtin_amdgcn_sched_barrier(0);  // keep the loads of later steps below: the register file is full
            }
        }
    }
    patch_epilogue<1, NT, 4, 1, true>(p, acc, b, ty0, tx0, tn, wid, 0, lane, ainv);
}

// [CoutPad][Kpad16] binary16 -> MFMA B-operand order: fragment (n tile j of 32, K step g of 16) = 64 lanes x 16 bytes,
// lane l = row 32 j + l % 32, k = 16 g + 8 (l / 32) .. + 7
__global__ void frag16_repack_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int CoutPad, int Kpad16) {
    const int G = Kpad16 / 16;
    const size_t total = (size_t)(CoutPad / 32) * G * 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(i & 63);
        const size_t jg = i >> 6;
        const int g = (int)(jg % G), j = (int)(jg / G);
        out[i] = in[((size_t)(32 * j + (l & 31)) * Kpad16 + 16 * g + 8 * (l >> 5)) / 8];
    }
}

template <int NT>
int launch_dcn16p(const ConvParams& p, hipStream_t stream) {
    constexpr int BN = 32 * NT;
    const int tiles_m = p.B * (p.H / TH) * (p.W / TW), tiles_n = p.CoutPad / BN;
    hipLaunchKernelGGL((dcn16p_kernel<NT>), dim3(tiles_m * tiles_n), dim3(256), 0, stream, p, tiles_m, tiles_n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

}  // namespace

// Full 8 x 16 patches, NHWC output, no split-K, 32-bit offsets, fragment-ordered weights present; the caller
// (cp_launch_conv16) also asks for enough blocks to fill the chip before it prefers this kernel to dcn16.hip's.
bool cp_dcn16p_supported(const ConvParams& p) {
    return p.offmask && p.w16f_hi && p.w16f_lo && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.nsrc == 1 &&
           p.H == p.Ho && p.W == p.Wo && p.splitk <= 1 && p.Cin % CKC == 0 && p.H % TH == 0 && p.W % TW == 0 &&
           p.H < 65535 && p.W < 65535 && p.store == CP_STORE_NHWC && p.Kpad16 == 9 * p.Cin && p.CoutPad % 64 == 0 &&
           !p.gn_stats && (size_t)p.B * p.H * p.W * p.Cin * 4 < (size_t)0xf0000000u &&
           (size_t)p.B * p.H * p.W * 128 < (size_t)0xf0000000u && (size_t)p.B * p.H * p.W * p.ldo * 4 < (size_t)0xf0000000u &&
           (size_t)p.CoutPad * p.Kpad16 * 2 < (size_t)0x7fffffff;
}

int cp_dcn16p_blocks(const ConvParams& p) { return p.B * (p.H / TH) * (p.W / TW) * (p.CoutPad / 64); }

// The 128-wide N tile (NT = 4: gather, blend and split once per 128 outputs instead of once per 64) for layers with whole
// 128-channel tiles whose launch still gives every CU a workgroup (measured at B = 64, profiles/r05_dcn_wide_ab.txt: 128 -> 128
// @64^2 379 -> 262 us, 256 -> 128 @32^2 171 -> 131, 256 -> 256 @32^2 364 -> 258, and 512 -> 256 @16^2 -- 256 workgroups -- 160 ->
// 151).  cp_set_debug 524288: never (A/B runs, tests).
bool cp_dcn16p_wide(const ConvParams& p) {
    return cp_dcn16p_supported(p) && p.CoutPad % 128 == 0 && !(p.dbg & 524288) &&
           ((p.dbg & 65536) || p.B * (p.H / TH) * (p.W / TW) * (p.CoutPad / 128) >= 256);  // (65536: launches of any size, tests)
}

int cp_launch_dcn16p(const ConvParams& p, hipStream_t stream) {
    if (!cp_dcn16p_supported(p)) return CP_ERR_INVALID;
    if (cp_dcn16p_wide(p)) return launch_dcn16p<4>(p, stream);
    return launch_dcn16p<2>(p, stream);
}

int cp_launch_frag16_repack(const void* w16, void* w16f, int CoutPad, int Kpad16, hipStream_t s) {
    if (CoutPad % 32 != 0 || Kpad16 % 16 != 0) return CP_ERR_INVALID;
    const size_t total = (size_t)CoutPad * Kpad16 / 8;
    int g = (int)((total + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(frag16_repack_kernel, dim3(g), dim3(256), 0, s, (const uint4*)w16, (uint4*)w16f, CoutPad, Kpad16);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}
