// Streamed, persistent DCNv2 (modulated deformable 3x3 convolution, stride 1, pad 1, dilation 1, one deformable group) in
// the split-f16 ("f16x3") arithmetic: dcn16p.hip's patch-resident gather with the memory side taken out of a block's
// critical path.  Replaces the reference's modulated_deformable_im2col + GEMM pair (DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:
// 125-195, dcn_v2_cuda.cu:42-172) for the launches with several patches per resident workgroup.
//
// Why (profiles/NOTES.md, round 3: in-kernel timeline and ablations of dcn16p).  A dcn16p block lives 47 k clocks of which
// the 36 K steps are 24 k: the rest is a prologue bound by the CU's fill rate (activation scale, offset / mask record, first
// chunk: 16 k), the second chunk's staging round trips (4 k) and the epilogue, none of it overlapped with arithmetic of the
// same block, and 8192 such blocks per launch pay the workgroup dispatch + 78 KB LDS set-up each.  Here
//   * the kernel is PERSISTENT: two workgroups per CU walk the (patch, N tile) items of an XCD-contiguous range; weights
//     pipeline, activation scale, LDS initialisation and the |max| commit happen once per workgroup, not once per patch;
//   * the input halo arrives by LDS-DMA (buffer_load_dwordx4 ... lds: no registers, no ds_write pass) in 16-CHANNEL
//     chunks into one of TWO buffers: chunk c + 1 (or chunk 0 of the NEXT patch) lands while the 9 K steps of chunk c run,
//     so a chunk boundary is one barrier, not a memory round trip, and a patch boundary is epilogue + set-up only;
//   * the next patch's offset / mask record is requested before the epilogue's stores and arrives under them;
//   * the exception samples' corners (2x2 blocks outside the staged halo) ride the same DMA stream into a corner buffer
//     and are blended LDS -> LDS by the wave that requested them (wave-local ordering, no extra barrier);
//   * halo 4 instead of 3 (16 x 24 patch pixels for 8 x 16 outputs): a third of the exception samples at the synthetic
//     network's offset spread, so 64 spare pixels suffice where dcn16p needs 184.
// LDS layout of a chunk buffer: pixel-major, 64 B per pixel (16 channels), rows of 24 pixels padded to 1552 B.  LDS-DMA
// writes lane-linearly (M0 base + 16 B x lane), so per-pixel padding is not available; bank conflicts of the gather are
// avoided by geometry instead: the 16 lanes of a ds_read_b128 group hold a 4 x 4 block of output pixels, whose un-deformed
// corners sit at (row + 4 col + quad) mod 16 sixteen-byte bank groups (1552 B = 97 groups = 1 mod 16): all distinct.
// Corner / K-step / quad offsets are immediates, one address register per (lane, tap), as in dcn16p.
// K order is (16-channel chunk, tap): same products as dcn16.hip / dcn16p.hip, different summation order.
// LDS: 2 x 30.6 KB buffers + 16 KB corner buffer + 1.5 KB lists = 79.2 KB => two workgroups per CU.
#include <type_traits>

#include "patch16_common.h"

namespace {

constexpr int S_TH = PATCH_TH, S_TW = PATCH_TW, S_HALO = 4;
constexpr int S_PW = S_TW + 2 * S_HALO, S_PH = S_TH + 2 * S_HALO;  // 24 x 16 patch pixels
constexpr int S_CK = 16;                                           // channels per chunk
constexpr int S_PXB = S_CK * 4;                                    // bytes per patch pixel
constexpr int S_ROWB = S_PW * S_PXB + 16;                          // bytes per patch row (= 1 mod 16 bank groups)
constexpr int S_ECAP = 64;                                         // exception samples per patch (one spare pixel each)
constexpr int S_SP0 = S_PH * S_ROWB;                               // first spare pixel
constexpr int S_BUFB = S_SP0 + S_ECAP * S_PXB + S_ROWB + 2 * S_PXB;  // + the other three "corners" of the last spare pixel
constexpr int S_CORN = 2 * S_BUFB;                                 // corner buffer: [exception][corner][quad] x 16 B
constexpr int S_EW = S_CORN + S_ECAP * 256;                        // float4 corner weights per exception
constexpr int S_EGOFF = S_EW + S_ECAP * 16;                        // top-left corner's byte offset into the input tensor
constexpr int S_EKEY = S_EGOFF + S_ECAP * 4;                       // (h_lo + 1) << 16 | (w_lo + 1)
constexpr int S_ECNT = S_EKEY + S_ECAP * 4;                        // two counters (patch parity)
constexpr int S_SCSH = S_ECNT + 16;                                // epilogue constants of the item's N tile: 64 scales, 64 shifts
constexpr int S_LDS = S_SCSH + 512;
constexpr int S_NSTEP = 9;                                         // K steps (one tap x 16 channels) per chunk
static_assert(S_BUFB % 16 == 0 && S_LDS <= 80 * 1024 - 128, "two workgroups per CU");
static_assert((S_ROWB / 16) % 16 == 1, "row pitch = 1 bank group mod 16");

__device__ __forceinline__ float4 s_ld4s(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ void s_both_halves5(const uint32_t (&v)[5], uint32_t (&lo)[5], uint32_t (&hi)[5]) {
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

// One LDS-DMA piece: lane l's 16 bytes at (voff + soff) of the buffer land at LDS byte lds_addr + 16 l (lanes masked off by
// EXEC write nothing; out-of-range offsets write zeros).  The compiler does not count it: completion = an explicit
// s_waitcnt vmcnt, visibility to other waves = a barrier after that.  s_nop 4: the operands may come straight from
// v_readfirstlane / v_cmp (VALU-written SGPRs read by VMEM); s_nop 0: M0 written by SALU, read by the DMA.
__device__ __forceinline__ void s_dma16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(r), "s"(soff)
                 : "memory", "m0");
}

typedef float s_f32x2 __attribute__((ext_vector_type(2)));

// The lane id behind an optimisation barrier: everything derived from it is recomputed where it is used (a handful of VALU
// per chunk) instead of being hoisted out of the item loop and held -- or spilled -- across the K loops.
__device__ __forceinline__ int s_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// (image, top-left output pixel, N tile) of item `it` (n fastest)
struct SItem {
    int b, ty0, tx0, tn;
};
__device__ __forceinline__ SItem s_item(int it, int tiles_n, int txs, int tys) {
    SItem r;
    r.tn = it % tiles_n;
    int tm = it / tiles_n;
    r.tx0 = (tm % txs) * S_TW;
    tm /= txs;
    r.ty0 = (tm % tys) * S_TH;
    r.b = tm / tys;
    return r;
}

template <int NT>
__global__ __launch_bounds__(256, 2) void dcn16s_kernel(const ConvParams p, const int tiles_m, const int tiles_n) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S_LDS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // lane -> output pixel of the wave's 4 x 8 sub-patch: the ds_read_b128 lane groups {0-3, 12-15, 20-27} / {4-11, 16-19,
    // 28-31} are the left / right 4 x 4 block (lane quad q8 = (lane % 32) / 4: block 0x96 >> q8, row q8 / 2)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);

    // ---- this workgroup's items: XCD x = blockIdx % 8 owns a contiguous range, its workgroups interleave inside it ----
    const int nitems = tiles_m * tiles_n, nblk = (int)gridDim.x >> 3;  // (the launcher makes gridDim a multiple of 8)
    const int xcd = (int)blockIdx.x & 7, bidx = (int)blockIdx.x >> 3;
    const int iq = nitems >> 3, ir = nitems & 7;
    const int it_lo = xcd < ir ? xcd * (iq + 1) : ir * (iq + 1) + (xcd - ir) * iq;
    const int it_end = it_lo + iq + (xcd < ir ? 1 : 0);
    int it = it_lo + bidx;
    if (it >= it_end) return;  // (block-uniform)
    const int txs = p.W / S_TW, tys = p.H / S_TH;

    const unsigned img_px = (unsigned)p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(p.src[0], img_px * (unsigned)p.Cin * 4u);
    const __amdgpu_buffer_rsrc_t r_om = make_rsrc(p.offmask, img_px * 128u);
    const unsigned w_bytes = (unsigned)((size_t)p.CoutPad * p.Kpad16 * 2);
    const __amdgpu_buffer_rsrc_t r_wh = make_rsrc(p.w16f_hi, w_bytes), r_wl = make_rsrc(p.w16f_lo, w_bytes);
    const int cb = p.Cin * 4, rowb = p.W * cb;
    const int nch = p.Cin / S_CK;                   // chunks per item (even: Cin % 32 == 0)
    const int Gk = p.Kpad16 / 16, gpt = p.Cin / 16;  // K steps per weight row / per tap

    // ---- halo DMA of chunk `ch` of the patch at (b, ty0, tx0) into buffer `buf`: wave w carries patch rows 4 w .. 4 w + 3,
    //      two pieces of 12 pixels x 4 quads each (lanes 0 .. 47) ----
    auto issue_halo = [&](const SItem& t, int ch, int buf) {
        const int ln = s_opaque(lane);
        // column validity of this lane's pixel in either half row (-> all-ones offset = out of range = zeros)
        const int gx0 = t.tx0 - S_HALO + (ln >> 2);
        const unsigned vh0 = (unsigned)gx0 < (unsigned)p.W ? (unsigned)(gx0 * cb + (ln & 3) * 16) : OOB;
        const unsigned vh1 = (unsigned)(gx0 + 12) < (unsigned)p.W ? (unsigned)((gx0 + 12) * cb + (ln & 3) * 16) : OOB;
        const unsigned ldsb = lds0 + (unsigned)(buf * S_BUFB + 4 * wid * S_ROWB);
        if (ln < 48) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int gy = t.ty0 - S_HALO + 4 * wid + (k >> 1);  // scalar
                const bool row_ok = (unsigned)gy < (unsigned)p.H;
                const unsigned soff = row_ok ? (unsigned)((t.b * p.H + gy) * rowb + ch * S_PXB) : 0u;
                const unsigned vo = row_ok ? ((k & 1) ? vh1 : vh0) : OOB;
                s_dma16(r_x, vo, soff, ldsb + (unsigned)((k >> 1) * S_ROWB + (k & 1) * (12 * S_PXB)));
            }
        }
    };

    // ---- offset / mask record of this lane's pixel: taps 5 lrow .. 5 lrow + 4 (slot 4 of the upper half is a dummy) ----
    float od[12], omk[5];
    auto load_record = [&](const SItem& t) {
        const int ln = s_opaque(lane), lrow = ln >> 5, q8 = (ln >> 2) & 7;
        const int y = t.ty0 + 4 * (wid >> 1) + (q8 >> 1), x = t.tx0 + 8 * (wid & 1) + 4 * ((0x96 >> q8) & 1) + (ln & 3);
        const unsigned rec = (unsigned)((t.b * p.H + y) * p.W + x) * 128u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float4 v = buf_ld4(r_om, rec + (unsigned)lrow * 40u + 16u * i);
            od[4 * i] = v.x; od[4 * i + 1] = v.y; od[4 * i + 2] = v.z; od[4 * i + 3] = v.w;
        }
        const float4 v = buf_ld4(r_om, rec + 72u + (unsigned)lrow * 20u);
        omk[0] = v.x; omk[1] = v.y; omk[2] = v.z; omk[3] = v.w;
        omk[4] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_om, (int)(rec + 88u + (unsigned)lrow * 20u), 0, 0));
    };

    // ---- weight fragments: (n tile j of 32, K step g of 16) = 1 KB in lane order at ((j Gk + g) 64 + lane) 16 B; three
    //      register sets, K step s of an item uses set s % 3 (9 steps per chunk keep the rotation aligned) ----
    u32x4 wbh[3][NT], wbl[3][NT];
    auto issue_b = [&](int set, int tn, int g) {
        const unsigned b_lane = (unsigned)(lane * 16);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int so = ((tn * NT + j) * Gk + g) * 1024;
            wbh[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wh, (int)b_lane, so, 0);
            wbl[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wl, (int)b_lane, so, 0);
        }
    };

    // the epilogue's per-channel scale / shift of N tile tn, once per workgroup (again only if a later item has another tn)
    auto write_scsh = [&](int tn) {
        if (tid < 32 * NT) {
            const int n = tn * (32 * NT) + tid;
            reinterpret_cast<float*>(smem + S_SCSH)[tid] = (p.scale && n < p.Cout) ? p.scale[n] : 1.f;
            reinterpret_cast<float*>(smem + S_SCSH)[32 * NT + tid] = (p.shift && n < p.Cout) ? p.shift[n] : 0.f;
        }
    };
    // =============================== prologue (once per workgroup) ===============================
    SItem cur = s_item(it, tiles_n, txs, tys);
    write_scsh(cur.tn);
    issue_halo(cur, 0, 0);
    load_record(cur);
    issue_b(0, cur.tn, 0 * gpt);
    issue_b(1, cur.tn, 1 * gpt);
    float afwd, ainv;
    conv_in_scale(p, &afwd, &ainv);
    afwd = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(afwd)));
    ainv = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ainv)));
    // spare pixels (+ the tail their other "corners" read) start as zeros: those corners carry weight 0 and must never be
    // NaN / Inf bit patterns left behind by an earlier kernel
    for (int i = tid; i < 2 * ((S_BUFB - S_SP0) / 16); i += 256) {
        const int bsel = i >= (S_BUFB - S_SP0) / 16 ? 1 : 0;
        const int k = i - bsel * ((S_BUFB - S_SP0) / 16);
        *reinterpret_cast<float4*>(smem + bsel * S_BUFB + S_SP0 + k * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 2) reinterpret_cast<int*>(smem + S_ECNT)[tid] = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first chunk has landed (this wave's pieces)
    __syncthreads();

    float amax = 0.f;
    int parity = 0;  // patch parity: which exception counter this item uses
    const int act = p.act;

    for (;;) {
        const int it_next = it + nblk;
        const bool has_next = it_next < it_end;
        const SItem nxt = s_item(has_next ? it_next : it, tiles_n, txs, tys);
        int* const ecnt = reinterpret_cast<int*>(smem + S_ECNT);

        // ---- bilinear set-up (dcn_v2_im2col_cuda.cu:25-54, 150-187): 5 tap slots per lane, then both halves swap ----
        const int ln0 = s_opaque(lane), lrow = ln0 >> 5, q8 = (ln0 >> 2) & 7;
        const int y = cur.ty0 + 4 * (wid >> 1) + (q8 >> 1), x = cur.tx0 + 8 * (wid & 1) + 4 * ((0x96 >> q8) & 1) + (ln0 & 3);
        uint32_t sq[5], sw[5][4];  // byte offset of corner (h_lo, w_lo) in a chunk buffer; corner weights x mask x pre-scale
        const float fy0 = (float)(y - 1), fx0 = (float)(x - 1);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            // tap 5 lrow + j = (kh, kw): lower half (0,0) (0,1) (0,2) (1,0) (1,1); upper half (1,2) (2,0) (2,1) (2,2) (-)
            const float khf = lrow ? (float)((5 + j) / 3) : (float)(j / 3);
            const float kwf = lrow ? (float)((5 + j) % 3) : (float)(j % 3);
            float h_im = (fy0 + khf) + od[2 * j];
            float w_im = (fx0 + kwf) + od[2 * j + 1];
            const bool valid = h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W && !(lrow && j == 4);
            h_im = valid ? h_im : 0.f;
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
            const int qy = h_lo - (cur.ty0 - S_HALO), qx = w_lo - (cur.tx0 - S_HALO);
            const bool inp = (unsigned)qy <= (unsigned)(S_PH - 2) && (unsigned)qx <= (unsigned)(S_PW - 2);
            int q = inp ? qy * S_ROWB + qx * S_PXB : 0;
            if (valid && !inp) {  // exception sample: file its corner and weights; this lane then reads spare pixel e with
                                  // weights (1, 0, 0, 0)
                const int e = atomicAdd(&ecnt[parity], 1);
                if (e < S_ECAP) {
                    reinterpret_cast<int*>(smem + S_EKEY)[e] = ((h_lo + 1) << 16) | (w_lo + 1);
                    reinterpret_cast<int*>(smem + S_EGOFF)[e] = ((cur.b * p.H + h_lo) * p.W + w_lo) * cb;
                    *reinterpret_cast<float4*>(smem + S_EW + e * 16) =
                        make_float4(__uint_as_float(sw[j][0]), __uint_as_float(sw[j][1]), __uint_as_float(sw[j][2]),
                                    __uint_as_float(sw[j][3]));
                    sw[j][0] = __float_as_uint(1.f);
                    sw[j][1] = sw[j][2] = sw[j][3] = 0u;
                    q = S_SP0 + e * S_PXB;
                }
            }
            sq[j] = (uint32_t)q;
        }
        int addr[9];       // fast mode: byte offset in a chunk buffer of corner (h_lo, w_lo) + this lane's 32-byte channel
                           // half; buffer-load mode: that corner's byte offset into the input tensor | 4 validity bits
        s_f32x2 bw[9][2];  // {w1, w2}, {w3, w4}
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint32_t pack[5] = {sq[j], sw[j][0], sw[j][1], sw[j][2], sw[j][3]};
            uint32_t lo[5], hi[5];
            s_both_halves5(pack, lo, hi);
            addr[j] = (int)lo[0] + lrow * 32;
            if (j < 4) addr[5 + j] = (int)hi[0] + lrow * 32;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bw[j][c >> 1][c & 1] = __uint_as_float(lo[1 + c]);
                if (j < 4) bw[5 + j][c >> 1][c & 1] = __uint_as_float(hi[1 + c]);
            }
        }
        __syncthreads();  // the exception list is complete
        const int nexc_all = __builtin_amdgcn_readfirstlane(ecnt[parity]);
        const bool slow = nexc_all > S_ECAP;  // block-uniform
        const int nexc = nexc_all < S_ECAP ? nexc_all : S_ECAP;
        if (tid == 0) ecnt[parity ^ 1] = 0;  // the next item's counter (its last readers passed the barrier above long ago)
        parity ^= 1;

        acc_t acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) acc[j][r] = 0.f;

        // blend + split of one gathered K step, then its 3 NT MFMAs (term order of igemm16.hip: lo*hi, hi*lo, hi*hi).  `mid` runs
        // between the two: the refill of the weight set the PREVIOUS step consumed -- never right behind that step's MFMAs, whose
        // B operand a fast-returning load would overwrite while the matrix pipe still reads it (dcn16p.hip, mma_step).
        auto mma_step = [&](const float4 (&r)[4][2], const s_f32x2 (&w)[2], const u32x4 (&bh)[NT], const u32x4 (&bl)[NT],
                            auto&& mid) {
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
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&bh[j]), al, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&bl[j]), ah, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&bh[j]), ah, acc[j], 0, 0, 0);
        };
        // inside step (ch, t): the weight set step t - 1 consumed takes the step two ahead -- of this chunk, of the next one, or of
        // the next item's first chunk
        auto refill = [&](int ch, int t) {
            const int t2 = t + 2 < S_NSTEP ? t + 2 : t + 2 - S_NSTEP;
            int ch2 = t + 2 < S_NSTEP ? ch : ch + 1;
            int tn2 = cur.tn;
            if (ch2 >= nch) {
                if (!has_next) return;
                ch2 = 0;
                tn2 = nxt.tn;
            }
            issue_b((t + 2) % 3, tn2, t2 * gpt + ch2);
        };

        if (!slow) {
            // ================= fast mode: every sample is in LDS =================
            // corner pieces of wave w: exceptions 4 (w + 4 j) + i, j, i < 4; lane = (j' = corner slot of the piece): the DMA
            // piece pc carries exceptions 4 pc .. 4 pc + 3 x 4 corners x 4 quads
            auto issue_corners = [&](int ch) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pc = wid + 4 * j;
                    if (4 * pc < nexc) {  // scalar
                        const int ln = s_opaque(lane);
                        const int e = 4 * pc + (ln >> 4), c = (ln >> 2) & 3;
                        const int key = reinterpret_cast<const int*>(smem + S_EKEY)[e];
                        const int go = reinterpret_cast<const int*>(smem + S_EGOFF)[e];
                        const int iy = (key >> 16) - 1 + (c >> 1), ix = (key & 0xffff) - 1 + (c & 1);
                        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                        const unsigned vo = ok ? (unsigned)(go + (c >> 1) * rowb + (c & 1) * cb + (ln & 3) * 16) : OOB;
                        s_dma16(r_x, vo, (unsigned)(ch * S_PXB), lds0 + (unsigned)(S_CORN + pc * 1024));
                    }
                }
            };
            // the wave's exceptions blended corner buffer -> spare pixels of buffer `buf` (same FMA order as the K loop);
            // lane = (j, i, quad): exception 4 (w + 4 j) + i
            auto blend_corners = [&](int buf) {
                const int ln = s_opaque(lane);
                const int e = 4 * (wid + 4 * (ln >> 4)) + ((ln >> 2) & 3);
                if (e < nexc) {
                    const unsigned char* cp0 = smem + S_CORN + e * 256 + (ln & 3) * 16;
                    const float4 v1 = *reinterpret_cast<const float4*>(cp0);
                    const float4 v2 = *reinterpret_cast<const float4*>(cp0 + 64);
                    const float4 v3 = *reinterpret_cast<const float4*>(cp0 + 128);
                    const float4 v4 = *reinterpret_cast<const float4*>(cp0 + 192);
                    const float4 w = *reinterpret_cast<const float4*>(smem + S_EW + e * 16);
                    float4 o;
                    o.x = fmaf(w.w, v4.x, fmaf(w.z, v3.x, fmaf(w.y, v2.x, w.x * v1.x)));
                    o.y = fmaf(w.w, v4.y, fmaf(w.z, v3.y, fmaf(w.y, v2.y, w.x * v1.y)));
                    o.z = fmaf(w.w, v4.z, fmaf(w.z, v3.z, fmaf(w.y, v2.z, w.x * v1.z)));
                    o.w = fmaf(w.w, v4.w, fmaf(w.z, v3.w, fmaf(w.y, v2.w, w.x * v1.w)));
                    *reinterpret_cast<float4*>(smem + buf * S_BUFB + S_SP0 + e * S_PXB + (ln & 3) * 16) = o;
                }
            };
            // chunk 0's exceptions: the only memory round trip of a patch that nothing hides
            if (nexc > 0) {
                issue_corners(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                blend_corners(0);
            }
            __syncthreads();

            // One K step of the software pipeline (all of it one wave's instruction stream, written slot by slot): the six MFMAs of
            // step t, each followed by an eighth-of-a-step of the NEXT step's blend (8 VALU) and two of the memory instructions that
            // belong to later steps (the gather of step t + 2, the weight fragments of step t + 2).  A wave on its own SIMD slot
            // hides ~5 issue slots under a 32-cycle MFMA (MI355X_MICROARCH.md); as blocks "48 VALU, then 6 MFMAs" the two waves of
            // a SIMD ran their phases in lock-step and nothing overlapped (every ablation saved its full share: profiles/NOTES.md).
            auto chunk = [&](auto par_c, int ch) {
                constexpr int PAR = decltype(par_c)::value;
                // ---- request the next chunk: halo into the other buffer, exception corners into the corner buffer ----
                const bool last = ch + 1 >= nch;
                if (!last) {
                    issue_halo(cur, ch + 1, PAR ^ 1);
                    issue_corners(ch + 1);
                } else if (has_next) {
                    issue_halo(nxt, 0, PAR ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* base = smem + PAR * S_BUFB;
                float4 raw[2][4][2];
                // corner c (0 .. 3), both quads of this lane's channel half
                auto gather_c = [&](float4 (&r)[4][2], int a, int c) {
                    const unsigned char* ap = base + a + (c >> 1) * S_ROWB + (c & 1) * S_PXB;
                    r[c][0] = *reinterpret_cast<const float4*>(ap);
                    r[c][1] = *reinterpret_cast<const float4*>(ap + 16);
                };
                uint32_t ahi[2][4], alo[2][4];  // split A operands of steps t (consumed by the MFMAs) and t + 1 (being produced)
                float o[4];
                // blend of quad hq in three pieces of 8 VALU: fma(w4, v4, fma(w3, v3, fma(w2, v2, w1 * v1))), then the hi / lo split
                auto blend_piece = [&](const float4 (&r)[4][2], const s_f32x2 (&w)[2], uint32_t (&hi)[4], uint32_t (&lo)[4], int hq,
                                       int piece) {
                    if (piece == 0) {
                        const float4 v1 = r[0][hq], v2 = r[1][hq];
                        o[0] = fmaf(w[0].y, v2.x, w[0].x * v1.x);
                        o[1] = fmaf(w[0].y, v2.y, w[0].x * v1.y);
                        o[2] = fmaf(w[0].y, v2.z, w[0].x * v1.z);
                        o[3] = fmaf(w[0].y, v2.w, w[0].x * v1.w);
                    } else if (piece == 1) {
                        const float4 v3 = r[2][hq], v4 = r[3][hq];
                        o[0] = fmaf(w[1].y, v4.x, fmaf(w[1].x, v3.x, o[0]));
                        o[1] = fmaf(w[1].y, v4.y, fmaf(w[1].x, v3.y, o[1]));
                        o[2] = fmaf(w[1].y, v4.z, fmaf(w[1].x, v3.z, o[2]));
                        o[3] = fmaf(w[1].y, v4.w, fmaf(w[1].x, v3.w, o[3]));
                    } else {
                        const Split2 t0 = split2(o[0], o[1]), t1 = split2(o[2], o[3]);
                        hi[2 * hq] = t0.hi; hi[2 * hq + 1] = t1.hi;
                        lo[2 * hq] = t0.lo; lo[2 * hq + 1] = t1.lo;
                    }
                };
                auto mma1 = [&](const uint32_t (&hi)[4], const uint32_t (&lo)[4], const u32x4 (&bh)[NT], const u32x4 (&bl)[NT], int k) {
                    // term order of igemm16.hip per accumulator: lo * hi, hi * lo, hi * hi
                    const u32x4 ahv = {hi[0], hi[1], hi[2], hi[3]}, alv = {lo[0], lo[1], lo[2], lo[3]};
                    const h8 ah = *reinterpret_cast<const h8*>(&ahv), al = *reinterpret_cast<const h8*>(&alv);
                    const int j = k % NT, term = k / NT;
                    const h8 a = term == 0 ? al : ah;
                    const h8 b = *reinterpret_cast<const h8*>(term == 1 ? &bl[j] : &bh[j]);
                    // weights as the first operand: the accumulators hold the TRANSPOSED tile (rows = output channels, columns =
                    // this wave's pixels), i.e. 4 consecutive channels of the lane's own pixel per accumulator quad -> 16-byte stores
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[j], 0, 0, 0);
                };
                // ---- pipeline fill: gathers of steps 0 and 1, blend of step 0 ----
#pragma unroll
                for (int c = 0; c < 4; ++c) gather_c(raw[0], addr[0], c);
#pragma unroll
                for (int c = 0; c < 4; ++c) gather_c(raw[1], addr[1], c);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 6; ++k) blend_piece(raw[0], bw[0], ahi[0], alo[0], k / 3, k % 3);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < S_NSTEP; ++t) {
#pragma unroll
                    for (int k = 0; k < 3 * NT; ++k) {
                        mma1(ahi[t & 1], alo[t & 1], wbh[t % 3], wbl[t % 3], k);
                        __builtin_amdgcn_sched_barrier(0);
                        // fillers of slot k
                        if (t + 1 < S_NSTEP) blend_piece(raw[(t + 1) & 1], bw[t + 1], ahi[(t + 1) & 1], alo[(t + 1) & 1], k / 3, k % 3);
                        if (k == 0) refill(ch, t);   // set (t + 2) % 3: step t - 1's MFMAs retired a whole blend ago
                        if (t + 2 < S_NSTEP && k >= 2) gather_c(raw[t & 1], addr[t + 2], k - 2);   // raw[t & 1]: consumed by blend(t)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // ---- chunk boundary: this wave's DMA pieces have landed once at most the 2 x 2 NT weight loads issued after
                //      them (steps 7 and 8: the next chunk's first two steps) are outstanding; its exceptions' corners are
                //      blended into the other buffer; one barrier ----
                if (!last || has_next) {
                    if (NT == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (!last) blend_corners(PAR ^ 1);
                __syncthreads();
            };
            for (int ch = 0; ch < nch; ch += 2) {
                chunk(std::integral_constant<int, 0>(), ch);
                chunk(std::integral_constant<int, 1>(), ch + 1);
            }
        } else {
            // ================= buffer-load mode: every sample of the patch through the texture path =================
            if (has_next) issue_halo(nxt, 0, 0);  // the next item's first chunk (this item's copy in buffer 0 is not used)
            {
                const unsigned rec = (unsigned)((cur.b * p.H + y) * p.W + x) * 128u;
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
                        gb = (((cur.b * p.H + h_lo) * p.W + w_lo) * cb) | vm;
                        w1 = hh * hw * mk; w2 = hh * lw * mk; w3 = lh * hw * mk; w4 = lh * lw * mk;
                    }
                    addr[t] = gb;
                    bw[t][0] = s_f32x2{w1, w2};
                    bw[t][1] = s_f32x2{w3, w4};
                }
            }
            for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
                for (int t = 0; t < S_NSTEP; ++t) {
                    float4 r[4][2];
                    const int so = ch * S_PXB;
                    const int base = (addr[t] & ~15) + lrow * 32;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {  // invalid corners out of range (-> 0)
                        const int gi = (addr[t] & (1 << c)) ? base + (c >> 1) * rowb + (c & 1) * cb : (int)OOB_BASE;
                        r[c][0] = s_ld4s(r_x, (unsigned)gi, so);
                        r[c][1] = s_ld4s(r_x, (unsigned)gi + 16u, so);
                    }
                    mma_step(r, bw[t], wbh[t % 3], wbl[t % 3], [&]() { refill(ch, t); });
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }

        // =============================== patch boundary ===============================
        if (has_next) load_record(nxt);  // arrives under the epilogue's stores
        {
            // transposed tile: accumulator 4 g + i of N tile j in lane (pixel = lane % 32, half h4) = output channel
            // 32 j + 8 g + 4 h4 + i of that lane's own pixel -> one 16-byte store per (j, g), 2 NT x 4 per lane instead of 16 NT
            // 4-byte ones (the epilogue was store-issue-bound: 4.2 k of an item's 41 k clocks, tools/dcn16s_timeline.py)
            const int ln = s_opaque(lane), h4 = ln >> 5, q8 = (ln >> 2) & 7;
            const int pix0 = (cur.b * p.H + cur.ty0 + 4 * (wid >> 1)) * p.W + cur.tx0 + 8 * (wid & 1);  // scalar
            float* frag_out = p.out + (size_t)pix0 * p.ldo + p.coff;
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(frag_out, (unsigned)((3 * p.W + 8) * p.ldo) * 4u);
            const unsigned vpix = (unsigned)(((q8 >> 1) * p.W + 4 * ((0x96 >> q8) & 1) + (ln & 3)) * p.ldo) * 4u;
            const int nb0 = cur.tn * (32 * NT) + 4 * h4;  // + 32 j + 8 g
            float4 sc[NT][4], sh[NT][4];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int nl = 4 * h4 + 32 * j + 8 * g4;  // channel inside the N tile (Cout % 4 == 0: a quad is inside or outside)
                    sc[j][g4] = *reinterpret_cast<const float4*>(smem + S_SCSH + nl * 4);
                    sh[j][g4] = *reinterpret_cast<const float4*>(smem + S_SCSH + (32 * NT + nl) * 4);
                }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int n0 = nb0 + 32 * j + 8 * g4;
                    float v[4] = {acc[j][4 * g4] * (sc[j][g4].x * ainv) + sh[j][g4].x, acc[j][4 * g4 + 1] * (sc[j][g4].y * ainv) + sh[j][g4].y,
                                  acc[j][4 * g4 + 2] * (sc[j][g4].z * ainv) + sh[j][g4].z, acc[j][4 * g4 + 3] * (sc[j][g4].w * ainv) + sh[j][g4].w};
                    if (act == CP_ACT_RELU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                    }
                    const bool n_ok = n0 < p.Cout;
#pragma unroll
                    for (int i = 0; i < 4; ++i) amax = fmaxf(amax, n_ok ? fabsf(v[i]) : 0.f);
                    const u32x4 pk4 = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(pk4, ro, (int)(n_ok ? vpix + (unsigned)n0 * 4u : 0x80000000u), 0, 0);
                }
        }
        if (!has_next) break;
        if (nxt.tn != cur.tn) {  // (block-uniform, rare: the grid stride is usually a multiple of the N tile count)
            __syncthreads();
            write_scsh(nxt.tn);
        }
        it = it_next;
        cur = nxt;
    }
    if (p.out_amax) cp_amax_commit(p.out_amax, amax);
}

template <int NT>
int launch_dcn16s(const ConvParams& p, int blocks, hipStream_t stream) {
    constexpr int BN = 32 * NT;
    const int tiles_m = p.B * (p.H / S_TH) * (p.W / S_TW), tiles_n = p.CoutPad / BN;
    hipLaunchKernelGGL((dcn16s_kernel<NT>), dim3(blocks), dim3(256), 0, stream, p, tiles_m, tiles_n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

}  // namespace

// dcn16p's conditions, plus: no residual / GroupNorm statistics, ReLU or no activation (what DeformConv and the stand-alone
// operator use), and the 16 x 24 patch's row arithmetic inside 32-bit offsets.
bool cp_dcn16s_supported(const ConvParams& p) {
    return cp_dcn16p_supported(p) && !p.res && (p.act == CP_ACT_NONE || p.act == CP_ACT_RELU) && p.Cin % 32 == 0 &&
           p.Cout % 4 == 0 && p.ldo % 4 == 0 && p.coff % 4 == 0;  // 16-byte stores of 4 consecutive output channels
}

int cp_dcn16s_items(const ConvParams& p) { return p.B * (p.H / S_TH) * (p.W / S_TW) * (p.CoutPad / 64); }

// Persistent grid: two workgroups per CU (a multiple of 8 so that every XCD runs the same number), fewer when the launch has
// fewer items.
int cp_launch_dcn16s(const ConvParams& p, hipStream_t stream) {
    if (!cp_dcn16s_supported(p)) return CP_ERR_INVALID;
    static int max_blocks = 0;
    if (!max_blocks) {
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        max_blocks = 2 * cus / 8 * 8;
        if (max_blocks < 8) max_blocks = 8;
    }
    const int items = cp_dcn16s_items(p);
    int blocks = items < max_blocks ? (items + 7) / 8 * 8 : max_blocks;
    if (p.dbg & 8388608) blocks = 8;  // tests: one workgroup per XCD, so that small problems walk several items per workgroup
    return launch_dcn16s<2>(p, blocks, stream);
}
