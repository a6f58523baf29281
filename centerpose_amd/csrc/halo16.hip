// Halo-resident 3x3 / stride-1 / pad-1 convolution in the split-f16 ("f16x3") arithmetic of igemm16.hip.
//
// The implicit-GEMM kernel of igemm16.hip re-loads, re-converts (float32 -> binary16 hi / lo) and re-stores the A tile
// for every one of the 9 taps: per 128-pixel tile and 64 input channels that is 9 x 32 KB of global loads, ~500 VALU
// per lane of conversion and 9 x 32 KB of LDS writes, all in the shadow of (and competing for issue slots with) the
// MFMAs -- the reason that kernel sits at ~35 % of the f16 matrix peak (41 % issuing, 38 % issue-stalled).
// Here a block owns an 8 x 16 patch of output pixels of one image and stages the (8+2) x (16+2) input halo of a
// 64-channel chunk ONCE into LDS, already split into hi / lo binary16 planes in pixel-major order.  Every tap's A
// fragment is then read straight out of that image (the 8 consecutive k of a lane are 8 consecutive halfs of one
// pixel), so per chunk the loop body is only: weight tile -> LDS (16 KB per tap), fragment reads, MFMAs.  A-side global
// traffic / conversion work / LDS writes drop by 9 / 1.41 (halo overhead) = 6.4x.  The same idea as lowc.hip, for the
// 64..512-channel layers: prediction-head 3x3s, ConvGRU input side, BasicBlock convolutions, DCN offset convolutions.
//
// K order is (64-channel chunk, tap, 32-channel half) instead of (tap, channel): same products, different summation
// order than igemm16p_kernel (results agree to float32 round-off, both inside the parity budget).
// Requirements (else the launcher falls back to igemm16p_kernel): 3x3, stride 1, pad 1, one source, Cin % 64 == 0,
// H % 8 == 0, W % 16 == 0, NHWC output, no split-K.
#include <type_traits>

#include "patch16_common.h"

#ifdef CP_HALO_STAMP
// tuning build: shader-clock stamps of wave 0 of one mid-launch workgroup (tools/halo16_timeline.py)
__device__ unsigned long long g_halo_clk[32];
#define HALO_STAMP(i) do { if (blockIdx.x == gridDim.x / 2 + 1 && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"); \
                                g_halo_clk[i] = clock64(); } } while (0)
extern "C" int cp_debug_read_halo_clk(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_halo_clk), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1;
}
#else
#define HALO_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int TH = 8, TW = 16, PW = TW + 2, PH = TH + 2, NPIX = PH * PW;  // 180 patch pixels
constexpr int CK = 64;                                                    // channels per staged chunk
constexpr int PROW = CK + 8;                                              // halfs per patch pixel and plane: 128 B of data + 16 B pad

// Pixel pitch 144 B = 9 sixteen-byte bank groups (round 5; rounds 2-4: 128 B with the chunk index XOR-ed by the pixel number).
// Lanes of a fragment read consecutive pixels at the same chunk: 9 is odd, so the 16 lanes of a ds_read_b128 group start in 16
// different bank groups -- conflict-free without a swizzle, and therefore every fragment address of a tile is ONE per-lane base
// (pixel of the fragment row, k half) plus a compile-time offset (tap, channel chunk).  With the XOR each of the 72 (pixel, chunk)
// combinations of a tile needed its own ~3 VALU of address arithmetic, which the compiler hoists in front of every tile's K loop:
// 270 of the ~850 non-MFMA VALU per hidden tile of the fused heads (SQ_INSTS_VALU, profiles/r05_pmc_sq_counters.txt).

// BDIRECT: the weight fragments of the 32-wide N tile come from global memory / L2 straight into the MFMA operand
// registers, two K tiles ahead, out of the fragment-ordered copy of the weights (ConvParams::w16f_*, one coalesced 1 KB
// load per fragment), so that the K loop has no barrier inside a chunk: with N = 32 a K tile is only 6 MFMAs per wave,
// and the barrier + LDS round trip of the shared weight tile cost more than that.  (A first attempt read the fragments
// from the k-contiguous [Cout][K] layout: every lane its own cache line, 80 TFLOP/s against 106 for the LDS tile --
// profiles/r02_halo_ab.txt.)
// EPI: 0 = plain epilogue (patch16_common.h); 1 = fused prediction head (igemm16.hip: FUSE -- transposed main product,
// bias + ReLU, second MFMA product with the 1x1 weights, slices to fuse_out); 2 = fused ConvGRU gates (igemm16.hip: GRU).
template <int MT, int NT, int WM, int WN, bool BDIRECT = false, int EPI = 0, int FT = 0>
__global__ __launch_bounds__(256, (BDIRECT && EPI == 0 && MT * NT <= 2) ? 3 : 2) void halo16_kernel(const ConvParams p, const int tiles_m, const int tiles_n) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    static_assert(WM * WN == 4 && 32 * MT * WM == TH * TW, "4 waves over a 128-pixel patch");
    static_assert(FT == 0 || EPI == 1, "FT: hidden tiles a fused-head workgroup walks (ConvParams::fuse_final), 0 = one tile, slabs");
    static_assert(EPI != 1 || (MT == 2 && NT == 2 && WM == 2 && WN == 2), "fused head: 128 x 128 tiles");
    static_assert(EPI != 2 || (MT == 1 && NT == 3 && WM == 4 && WN == 1), "GRU: 128 x 96 tiles");
    constexpr int BN = 32 * NT * WN;
    constexpr int B_CHUNKS = BN * BK16 * 2 / 16;  // 16-byte chunks per weight array (hi or lo) per 32-deep K tile
    constexpr int B_SLOTS = (B_CHUNKS + 255) / 256;
    constexpr bool B_PART = B_CHUNKS % 256 != 0;
    constexpr int B_SZ = BN * LDH;
    __shared__ __attribute__((aligned(16))) _Float16 patch_hi[NPIX * PROW];
    __shared__ __attribute__((aligned(16))) _Float16 patch_lo[NPIX * PROW];
    __shared__ __attribute__((aligned(16))) _Float16 bt[BDIRECT ? 1 : 2][BDIRECT ? 8 : 2 * B_SZ];  // [buffer][hi | lo]
    __shared__ float red_s[EPI == 1 ? 2 * 2 * 16 * 64 : 1];  // fused head: the second hidden half's partial maps [wm][i][r][lane]
                                                              // (FT: two buffers of 8 rows, used alternately)
    __shared__ float sum_s[(EPI == 1 && FT) ? 2 * 2 * 8 * 64 : 1];  // fuse_final: the maps summed over the tiles walked so far (8 real rows)

    const int tid = threadIdx.x, lane = tid & 63;
    HALO_STAMP(0);  // start
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: scalar register
    const int wm = wid / WN, wn = wid % WN;
    const int tile = tile_of_block(tiles_m, tiles_n);
    int tn = tile % tiles_n;
    int tm = tile / tiles_n;
    if (EPI == 1 && p.fuse_ngroups > 0) {
        // several heads in one launch.  Tile order = bands of HEAD_BAND patches; inside a band head by head, inside a head
        // patch by patch, n fastest: an XCD's contiguous run of tiles keeps one band's input (~1.5 MB with halos) and one
        // head's weights (0.6 MB) in its 4 MB L2, so the shared input is read from HBM once -- head-slowest over the whole
        // map streamed it once per head (2.17 GB per launch at B = 64 against 0.42 GB algorithmic, profiles/pmc_traffic.json
        // of the first grouped form), n-fastest over all heads cycled through 4.1 MB of weights per patch (3 % slower)
        const int bsel = (p.dbg >> 18) & 3;  // cp_set_debug bits 18-19 (A/B): band of 8 / 128 patches / the whole map
        const int HEAD_BAND = bsel == 0 ? 32 : bsel == 1 ? 8 : bsel == 2 ? 128 : tiles_m;
        // fuse_final: one workgroup per (patch, head) -- tiles_n counts heads, tn becomes the head's first hidden tile below
        const int gt = FT ? 1 : p.fuse_gtiles, per_band = HEAD_BAND * tiles_n;
        const int band = tile / per_band, m0 = band * HEAD_BAND;
        const int bsz = min(HEAD_BAND, tiles_m - m0);  // patches in this band (the last one may be short)
        const int rem = tile - band * per_band;
        const int g = rem / (bsz * gt), rr = rem - g * (bsz * gt);
        tn = g * gt + rr % gt;
        tm = m0 + rr / gt;
    }
    constexpr int ntl = FT ? FT : 1;  // hidden tiles this workgroup walks (a compile-time count: the walk is unrolled, so the
                                      // staging code and its operands are not carried through a loop -- as a run-time loop the
                                      // kernel spilled 15 registers and wrote 1.4 GB of scratch lines back per launch)
    if (FT) tn *= FT;
    const int txs = p.W / TW, tys = p.H / TH;
    const int tx0 = (tm % txs) * TW;
    tm /= txs;
    const int ty0 = (tm % tys) * TH, b = tm / tys;
    // activation pre-scale: its 32 scalar loads are issued here and reduced only behind the first staging loads (below)
    float afwd = 1.f, ainv = 1.f;
    const AmaxRaw amax_raw = conv_in_scale_issue(p);
    bool have_scale = false;

    const unsigned img_bytes = (unsigned)p.B * p.H * p.W * (unsigned)p.Cin * 4u;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(p.src[0], img_bytes);
    const unsigned w_bytes = (unsigned)((size_t)p.CoutPad * p.Kpad16 * 2);
    const __amdgpu_buffer_rsrc_t r_wh = make_rsrc(BDIRECT ? p.w16f_hi : p.w16_hi, w_bytes),
                                 r_wl = make_rsrc(BDIRECT ? p.w16f_lo : p.w16_lo, w_bytes);

    // ---- staging geometry: slot s of this thread = patch pixel (tid / 16) + 16 s, float4 column tid % 16 ----
    constexpr int ST = (NPIX + 15) / 16;  // 12 passes of 16 pixels x 16 float4

    // ---- weight tile loads: chunk f -> row n = f / 4, 16-byte column f % 4 of the 32-deep K tile ----
    unsigned b_off[B_SLOTS];
#pragma unroll
    for (int j = 0; j < B_SLOTS; ++j) {
        const int f = tid + j * 256;
        b_off[j] = (!B_PART || f < B_CHUNKS) ? (unsigned)(((size_t)(tn * BN + f / 4) * p.Kpad16 + (f % 4) * 8) * 2) : OOB;
    }
    u32x4 gbh[B_SLOTS], gbl[B_SLOTS];
    auto issue_b = [&](int kbyte) {  // kbyte: byte offset of the K tile inside a weight row (wave-uniform)
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) {
            gbh[j] = __builtin_amdgcn_raw_buffer_load_b128(r_wh, (int)b_off[j], kbyte, 0);
            gbl[j] = __builtin_amdgcn_raw_buffer_load_b128(r_wl, (int)b_off[j], kbyte, 0);
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) {
            const int f = tid + j * 256;
            if (!B_PART || f < B_CHUNKS) {
                const int nn = f / 4, c = f % 4;
                *reinterpret_cast<u32x4*>(bt[buf] + nn * LDH + (c ^ swz(nn)) * 8) = gbh[j];
                *reinterpret_cast<u32x4*>(bt[buf] + B_SZ + nn * LDH + (c ^ swz(nn)) * 8) = gbl[j];
            }
        }
    };

    acc_t acc[MT][NT];
    // BDIRECT: fragment (n tile j, K step g of 16) = 1 KB in lane order at ((j G + g) 64 + lane) 16 bytes
    const int G = p.Kpad16 / 16, gpt = p.Cin / 16;  // K steps per weight row / per tap
    unsigned bd_off[NT];
    constexpr int NSET = MT * NT == 1 ? 3 : 2;  // K tiles in flight + 1 (18 tiles per chunk: both rotations stay consistent)
    u32x4 dbh[NSET][2][NT], dbl[NSET][2][NT];  // [register set = K tile % NSET][k-step][fragment]
    const int nchunks = p.Cin / CK;
    // K tile kt of chunk c = (tap kt / 2, 32-channel half kt % 2) -> first K step g = tap (Cin / 16) + 4 c + 2 (kt % 2)
    auto issue_bd = [&](int set, int c, int kt) {
        if (kt >= 18) { kt -= 18; ++c; }
        if (c >= nchunks) return;
        const int g = (kt >> 1) * gpt + 4 * c + 2 * (kt & 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                dbh[set][ks][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wh, (int)bd_off[j], (g + ks) * 1024, 0);
                dbl[set][ks][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wl, (int)bd_off[j], (g + ks) * 1024, 0);
            }
    };
    // hand-pipelined K loop (BDIRECT): weight fragments WD K steps ahead in WS = WD + 1 register sets, A fragments AD steps ahead in
    // AS = AD + 1 sets; both set counts divide the 36 steps of a chunk, so the rotation is the same in every chunk.  WD keeps the
    // weights' lead at 18 .. 24 MFMAs (an L2 round trip under load), AD the A fragments' at >= 6 (an LDS round trip)
    // Measured per tile shape (profiles/r05_halo16_pipe_ab.txt, same box, alternating runs).  With the XOR-swizzled patch image
    // (rounds 2-4) the pipelined loop paid only where registers were left: 128-wide plain tile -2.1 %, 32-wide tile -1.5 %, but
    // 64-wide tile +5 % and fused heads +5.6 % (their extra A set spilled).  With the padded image (no address registers: 176 ..
    // 201 VGPRs instead of 256) it pays everywhere: heads -2.6 %, 64-wide tile -3 %, the rest -0.5 % -- so it is the loop of every
    // BDIRECT kernel.  A third workgroup per CU for the 128-wide plain tile (168 registers) measured +-0 and is not used.
#ifdef CP_HALO_OLDK
    constexpr bool PIPE = false;
#else
    constexpr bool PIPE = BDIRECT;
#endif
    constexpr int WD = MT * NT >= 4 ? 2 : MT * NT >= 2 ? 3 : 5, WS = WD + 1;
    constexpr int AD = MT * NT >= 2 ? 1 : 2, AS = AD + 1;
    static_assert(36 % WS == 0 && 36 % AS == 0, "set rotation must close over a chunk");
    u32x4 pwh[WS][NT], pwl[WS][NT];
    h8 pah[AS][MT], pal[AS][MT];
    auto issue_w = [&](int set, int c, int st) {
        if (st >= 36) { st -= 36; ++c; }
        if (c >= nchunks) return;
        const int kt = st >> 1;
        const int g = (kt >> 1) * gpt + 4 * c + 2 * (kt & 1) + (st & 1);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            pwh[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wh, (int)bd_off[j], g * 1024, 0);
            pwl[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wl, (int)bd_off[j], g * 1024, 0);
        }
    };
    // ---- stage the halo patch of one 64-channel chunk: float32 global -> hi / lo binary16 planes (called per chunk; fused heads
    // that walk several tiles: once) ----
    auto stage_chunk = [&](int ch) {
        // all loads of a chunk are issued before the first conversion: one HBM round trip per chunk
        constexpr int SR = 12;  // (rounds 2-4: two rounds of 6 where the XOR-swizzled image left no registers; one round now: offset convolutions -2.4 %)
        // the thread id is rebuilt per chunk (not held in a vector register across the K loop)
        int lane_c = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane_c));
        const int tid_c = wid * 64 + lane_c, c4c = tid_c & 15;
#pragma unroll
        for (int s0 = 0; s0 < ST; s0 += SR) {
            float4 sv[SR];
#pragma unroll
            for (int s = 0; s < SR; ++s) {
                const int q = (tid_c >> 4) + 16 * (s0 + s);
                const int py = q / PW, px = q - py * PW;
                const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
                const bool in = q < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const unsigned off = in ? (unsigned)(((b * p.H + iy) * p.W + ix) * p.Cin + ch * CK + c4c * 4) * 4u : OOB;
                sv[s] = buf_ld4(r_x, off);
            }
            if (!have_scale) {  // once per block, with the first round of staging loads already in flight
                conv_in_scale_finish(p, amax_raw, &afwd, &ainv);
                have_scale = true;
            }
#pragma unroll
            for (int s = 0; s < SR; ++s) {
                const int q = (tid_c >> 4) + 16 * (s0 + s);
                if (q < NPIX) {
                    const float4 v = sv[s];
                    const Split2 h0 = split2(v.x * afwd, v.y * afwd), h1 = split2(v.z * afwd, v.w * afwd);
                    const int col = c4c * 4;  // halfs
                    *reinterpret_cast<u32x2*>(patch_hi + q * PROW + col) = u32x2{h0.hi, h1.hi};
                    *reinterpret_cast<u32x2*>(patch_lo + q * PROW + col) = u32x2{h0.lo, h1.lo};
                }
            }
        }
    };
    if (FT) {  // one chunk (Cin = 64), staged once for every hidden tile the workgroup walks
        stage_chunk(0);
        __syncthreads();
    }
    // fuse_final 2: the workgroup walks every head of its patch (one staging for all of them), head by head
    const int nwalk = (FT && p.fuse_final == 2) ? p.fuse_ngroups : 1;
#pragma unroll 1
    for (int hw = 0; hw < nwalk; ++hw)
#pragma unroll
    for (int t2 = 0; t2 < ntl; ++t2) {
        // (FT && PIPE: one chunk per tile -- the first K step's MFMAs take a literal zero as C instead: no 64 v_mov per tile)
        if (!(FT && PIPE)) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < F::NACC; ++r) acc[i][j][r] = 0.f;
        }

        // ---- fragment geometry: row m of the tile = pixel (m / 16, m % 16); lane reads rows lcol + 32 i + 32 MT wm ----
        // (derived from a lane id the compiler cannot see through, once per hidden tile: otherwise the unrolled walk shares the
        // per-tap LDS addresses of its K loops and carries them -- 15 registers -- across the epilogue in between, in scratch)
        int lane_t = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane_t));
        const int lrow = lane_t >> 5, lcol = lane_t & 31;
        int q0[MT];  // patch pixel of the fragment row at tap (0, 0)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = wm * (MT * 32) + i * 32 + lcol;
            q0[i] = (m >> 4) * PW + (m & 15);
        }
        const int b_frag = (wn * (NT * 32) + lcol) * LDH;
#pragma unroll
        for (int j = 0; j < NT; ++j) bd_off[j] = (unsigned)((((tn * (BN / 32) + wn * NT + j) * G) * 64 + lane_t) * 16);

        if (BDIRECT) {  // the first K steps' weights in flight while the first patch is staged
            if (PIPE) {
#pragma unroll
                for (int st = 0; st < WD; ++st) issue_w(st, 0, st);
            } else {
                issue_bd(0, 0, 0);
                if (NSET == 3) issue_bd(1, 0, 1);
            }
        }

        for (int ch = 0; ch < nchunks; ++ch) {
            // first weight tile of the chunk in flight while the patch is staged
            if (!BDIRECT) issue_b(((0 * p.Cin) + ch * CK) * 2);
            if (!FT) {
                if (ch > 0) __syncthreads();  // every wave is done reading the previous chunk's patch
                if (ch < 4) HALO_STAMP(1 + 4 * ch);  // chunk: previous K loop over, barrier passed
                stage_chunk(ch);
                if (ch < 4) HALO_STAMP(2 + 4 * ch);  // this wave's share staged
                if (!BDIRECT) store_b(0);
                __syncthreads();
                if (ch < 4) HALO_STAMP(3 + 4 * ch);  // barrier passed: K loop starts
            }
            // ---- 18 K tiles: (tap, 32-channel half) ----
            auto k_tile = [&](int kt, const _Float16* Bh, const _Float16* Bl, const u32x4 (&fh)[2][NT], const u32x4 (&fl)[2][NT]) {
                const int tap = kt >> 1, half = kt & 1;
                const int kh = tap / 3, kw = tap - kh * 3;
                const int dq = kh * PW + kw;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    h8 ah[MT], al[MT], bh[NT], bl[NT];
                    const int c8 = half * 4 + ks * 2 + lrow;  // 16-byte chunk of the 64-channel pixel row
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int q = q0[i] + dq;
                        const int o = q * PROW + c8 * 8;
                        ah[i] = *reinterpret_cast<const h8*>(patch_hi + o);
                        al[i] = *reinterpret_cast<const h8*>(patch_lo + o);
                    }
                    const int co = ((ks * 2 + lrow) ^ swz(lcol)) * 8;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if (BDIRECT) {
                            bh[j] = *reinterpret_cast<const h8*>(&fh[ks][j]);
                            bl[j] = *reinterpret_cast<const h8*>(&fl[ks][j]);
                        } else {
                            bh[j] = *reinterpret_cast<const h8*>(Bh + j * 32 * LDH + co);
                            bl[j] = *reinterpret_cast<const h8*>(Bl + j * 32 * LDH + co);
                        }
                    }
                    // (fused head: transposed product -- rows = output channels, columns = pixels)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = EPI == 1 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = EPI == 1 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = EPI == 1 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            };
            if (PIPE) {
                // ---- 36 K steps (tap, 32-channel half, 16-channel quarter), software-pipelined by hand (round 5): at step s the
                //      weight fragments of step s + WD and the A fragments of step s + AD are requested, then the MT x NT x 3 MFMAs
                //      of step s run on operands that arrived under earlier steps' MFMAs.  Before, a K tile read each A fragment
                //      right in front of its first MFMA (`ds_read ; s_waitcnt lgkmcnt(0) ; v_mfma` -- the scheduler sinks loads to
                //      their use when the register file is full), so each wave exposed an LDS round trip four times per 12 MFMAs
                //      and the pipe depended on the other wave of the SIMD being ready at exactly those moments. ----
                auto load_a = [&](int set, int st) {
                    const int kt = st >> 1, ks = st & 1, tap = kt >> 1, half = kt & 1;
                    const int kh = tap / 3, kw = tap - kh * 3, dq = kh * PW + kw;
                    const int c8 = half * 4 + ks * 2 + lrow;  // 16-byte chunk of the 64-channel pixel row
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int q = q0[i] + dq;
                        const int o = q * PROW + c8 * 8;
                        pah[set][i] = *reinterpret_cast<const h8*>(patch_hi + o);
                        pal[set][i] = *reinterpret_cast<const h8*>(patch_lo + o);
                    }
                };
#pragma unroll
                for (int st = 0; st < AD; ++st) load_a(st, st);
#pragma unroll
                for (int st = 0; st < 36; ++st) {
                    issue_w((st + WD) % WS, ch, st + WD);            // (that set was consumed by step st - 1)
                    if (st + AD < 36) load_a((st + AD) % AS, st + AD);
                    __builtin_amdgcn_sched_barrier(0);
                    const h8 (&ah)[MT] = pah[st % AS];
                    const h8 (&al)[MT] = pal[st % AS];
                    const u32x4 (&fh)[NT] = pwh[st % WS];
                    const u32x4 (&fl)[NT] = pwl[st % WS];
                    // (fused head: transposed product -- rows = output channels, columns = pixels)
                    const acc_t zero_c = {};
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const acc_t c_in = (FT && st == 0) ? zero_c : acc[i][j];
                            acc[i][j] = EPI == 1 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&fh[j]), al[i], c_in, 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], *reinterpret_cast<const h8*>(&fh[j]), c_in, 0, 0, 0);
                        }
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = EPI == 1 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&fl[j]), ah[i], acc[i][j], 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], *reinterpret_cast<const h8*>(&fl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = EPI == 1 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&fh[j]), ah[i], acc[i][j], 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], *reinterpret_cast<const h8*>(&fh[j]), acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (BDIRECT) {
#pragma unroll
                for (int kt = 0; kt < 18; ++kt) {
                    // (phase order pinned: the scheduler would otherwise sink the loads to just above their use)
                    // set (kt + NSET - 1) % NSET = (kt - 1) % NSET was consumed by the previous tile
                    issue_bd((kt + NSET - 1) % NSET, ch, kt + NSET - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    k_tile(kt, nullptr, nullptr, dbh[kt % NSET], dbl[kt % NSET]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll 2
                for (int kt = 0; kt < 18; ++kt) {
                    const int cur = kt & 1;
                    if (kt + 1 < 18) {
                        const int tap1 = (kt + 1) >> 1, half1 = (kt + 1) & 1;
                        issue_b((tap1 * p.Cin + ch * CK + half1 * 32) * 2);
                    }
                    k_tile(kt, bt[cur] + b_frag, bt[cur] + b_frag + B_SZ, dbh[0], dbl[0]);
                    if (kt + 1 < 18) store_b(cur ^ 1);
                    __syncthreads();
                }
            }
        }

        HALO_STAMP(20);  // last K loop done
        // (the lane id is rebuilt here instead of living in a vector register across the K loop -- opaquely, so that the
        // unrolled tile walk does not keep one tile's epilogue addresses for the next tile's)
        int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (EPI == 1) asm volatile("" : "+v"(lane_e));
        if constexpr (EPI == 1) {
            // ---- fused prediction head (the arithmetic of igemm16.hip's FUSE epilogue on this kernel's pixel order) ----
            // Round 5: every table of this epilogue (scale / shift, 1x1 fragments, w2_inv, bias, the output maps) sits behind a
            // scalar buffer descriptor and is read 16 bytes at a time -- the per-lane part of an address is one small offset
            // (half-wave x 16 B, or lane x 16 B), everything else scalar.  Before, each of the ~130 table reads and 32 stores of a
            // tile was a 4-byte access with its own 64-bit address arithmetic (292 v_lshl_add_u64 + 236 v_add_u32 + 535 v_mov in
            // the kernel: 2.4 non-MFMA VALU per MFMA, all of it issued while the wave's MFMAs stand still).  Heads that finish in
            // the kernel (FT) have at most 16 final channels = accumulator rows r < 8 of the second product: the rest is padding
            // and is no longer scaled, exchanged or summed.  Same operations on the same values in the same order: bit-identical.
            const int M = p.B * p.H * p.W;
            // head of this N tile (several heads in one launch: ConvParams::fuse_ngroups), its channel count and first plane
            const int hg = p.fuse_ngroups > 0 ? tn / p.fuse_gtiles : 0;
            const int c2 = p.fuse_ngroups > 0 ? p.fuse_gc2[hg] : p.fuse_c2;
            const int plane0 = p.fuse_ngroups > 0 ? p.fuse_gbase[hg] + (tn - hg * p.fuse_gtiles) * c2 : tn * p.fuse_c2;
            constexpr int RN = FT ? 8 : 16;  // accumulator rows of the second product that hold real channels
            const int h4e = lane_e >> 5;
            const unsigned v16 = (unsigned)(h4e * 16), vlane = (unsigned)(lane_e * 16);
            auto ld4so = [](__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) -> float4 {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
                return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            };
            const __amdgpu_buffer_rsrc_t r_w2h = make_rsrc(p.fuse_w2_hi, 0x7ffffff0u), r_w2l = make_rsrc(p.fuse_w2_lo, 0x7ffffff0u);
            const int w2s = ((tn * WN + wn) * 4) * 1024;  // this N tile's four 1 KB fragments (scalar)
            h8 wh[2][2], wl[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r_w2h, (int)vlane, w2s + (j * 2 + ks) * 1024, 0);
                    const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(r_w2l, (int)vlane, w2s + (j * 2 + ks) * 1024, 0);
                    wh[j][ks] = *reinterpret_cast<const h8*>(&a);
                    wl[j][ks] = *reinterpret_cast<const h8*>(&c);
                }
            acc_t acc2[2];  // (started by a literal-zero C in the first product below)
            const bool relu = p.act == CP_ACT_RELU;
            float hmax = 0.f;
            {
                // accumulator r of a lane = hidden channel (r & 3) + 8 (r >> 2) + 4 h4 of the fragment: four groups of four
                // consecutive channels -> one 16-byte read per group of the scale and of the shift table
                const __amdgpu_buffer_rsrc_t r_sc = make_rsrc(p.scale, (unsigned)p.CoutPad * 4u), r_sh = make_rsrc(p.shift, (unsigned)p.CoutPad * 4u);
                const int ch0 = (tn * BN + wn * 64) * 4;  // bytes (scalar)
                // (the activation is decided once per tile, not per element: as `if (relu) x = max(x, 0)` inside the loops it compiled
                // to a v_max + v_cndmask pair per accumulator)
                auto scale_shift = [&](auto relu_c) {
                    constexpr bool RELU = decltype(relu_c)::value;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 s4 = p.scale ? ld4so(r_sc, v16, ch0 + (j * 32 + 8 * g) * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
                            const float4 b4 = p.shift ? ld4so(r_sh, v16, ch0 + (j * 32 + 8 * g) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                            const float se[4] = {s4.x, s4.y, s4.z, s4.w}, be[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = 4 * g + e;
                                const float sc = se[e] * ainv, sh = be[e];
                                float x0 = acc[0][j][r] * sc + sh, x1 = acc[1][j][r] * sc + sh;
                                if (RELU) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
                                acc[0][j][r] = x0;
                                acc[1][j][r] = x1;
                                hmax = fmaxf(fmaxf(hmax, fabsf(x0)), fabsf(x1));  // (one v_max3_f32)
                            }
                        }
                    }
                };
                if (relu) scale_shift(std::true_type());
                else scale_shift(std::false_type());
            }
            // the wave's maximum without LDS round trips (six ds_bpermute + waits stood here): four DPP steps leave every row of 16
            // lanes with its maximum, four v_readlane + scalar max finish it -- the result and the scale pair are wave-uniform
            {
                auto dppmax = [](float v, auto ctrl) {
                    return fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(v), __float_as_uint(v), decltype(ctrl)::value, 0xf, 0xf, false)));
                };
                hmax = dppmax(hmax, std::integral_constant<int, 0xB1>());   // quad_perm [1,0,3,2]
                hmax = dppmax(hmax, std::integral_constant<int, 0x4E>());   // quad_perm [2,3,0,1]
                hmax = dppmax(hmax, std::integral_constant<int, 0x141>());  // row_half_mirror
                hmax = dppmax(hmax, std::integral_constant<int, 0x140>());  // row_mirror
                const unsigned hb = __float_as_uint(hmax);  // (non-negative floats order like their bit patterns)
                const unsigned m01 = max((unsigned)__builtin_amdgcn_readlane((int)hb, 0), (unsigned)__builtin_amdgcn_readlane((int)hb, 16));
                const unsigned m23 = max((unsigned)__builtin_amdgcn_readlane((int)hb, 32), (unsigned)__builtin_amdgcn_readlane((int)hb, 48));
                hmax = __uint_as_float(max(m01, m23));
            }
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
                            const Split2 sp = split2(acc[i][j][r] * hfwd, acc[i][j][r + 1] * hfwd);
                            hh[q] = sp.hi;
                            hl[q] = sp.lo;
                        }
                        const u32x4 vh = {hh[0], hh[1], hh[2], hh[3]}, vl = {hl[0], hl[1], hl[2], hl[3]};
                        const h8 bhh = *reinterpret_cast<const h8*>(&vh), bhl = *reinterpret_cast<const h8*>(&vl);
                        const acc_t zero_c2 = {};
                        acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j][ks], bhh, (j == 0 && ks == 0) ? zero_c2 : acc2[i], 0, 0, 0);
                        acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j][ks], bhl, acc2[i], 0, 0, 0);
                        acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j][ks], bhh, acc2[i], 0, 0, 0);
                    }
                }
            }
            // exchange buffer of the two hidden halves' partial maps, [wm][i][r][lane].  FT (8 real rows): two half-size buffers
            // used alternately, so ONE barrier per tile orders everything -- tile t + 2 re-uses tile t's buffer, and tile t's
            // readers have arrived at tile t + 1's barrier by then.  Slab form: one buffer, a barrier in front as well.
            if (!FT) __syncthreads();  // the previous tile's partial maps have been read
            float* red = FT ? red_s + ((hw * ntl + t2) & 1) * (2 * 2 * 8 * 64) : red_s;
            constexpr int RS = FT ? 8 : 16;  // rows per (wm, i) block of `red`
            {
                const __amdgpu_buffer_rsrc_t r_wi = make_rsrc(p.fuse_w2_inv, 0x7ffffff0u);
#pragma unroll
                for (int g = 0; g < RN / 4; ++g) {
                    const float4 w4 = p.fuse_w2_inv ? ld4so(r_wi, v16, (hg * 64 + 8 * g) * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
                    const float we[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float wi = we[e] * hinv;
                        acc2[0][4 * g + e] *= wi;
                        acc2[1][4 * g + e] *= wi;
                    }
                }
            }
            if (wn == 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < RN; ++r) red[((wm * 2 + i) * RS + r) * 64 + lane_e] = acc2[i][r];
            }
            __syncthreads();
            if (wn == 0) {
                const int HW = p.H * p.W;
                // the finished maps of head hg, image b: channel c of this lane's row group, pixel pix -> byte (c HW + pix) 4 of
                // [c2][HW]; channels >= c2 fall outside the descriptor (and are skipped before the sigmoid anyway)
                const __amdgpu_buffer_rsrc_t r_go = make_rsrc(FT ? p.fuse_gout[hg] + (size_t)b * c2 * HW : nullptr, (unsigned)(c2 * HW) * 4u);
                const __amdgpu_buffer_rsrc_t r_gb = make_rsrc(FT ? p.fuse_gbias[hg] : nullptr, (unsigned)c2 * 4u);
                float bias_e[RN] = {};
                if (FT && t2 + 1 == ntl) {
#pragma unroll
                    for (int g = 0; g < RN / 4; ++g) {
                        const float4 b4 = p.fuse_gbias[hg] ? ld4so(r_gb, v16, 8 * g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                        bias_e[4 * g] = b4.x; bias_e[4 * g + 1] = b4.y; bias_e[4 * g + 2] = b4.z; bias_e[4 * g + 3] = b4.w;
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ml = wm * 64 + i * 32 + (lane_e & 31);  // tile row -> patch pixel (ml / 16, ml % 16)
                    const int m = (b * p.H + ty0 + (ml >> 4)) * p.W + tx0 + (ml & 15);
                    const int pix = (ty0 + (ml >> 4)) * p.W + tx0 + (ml & 15);
                    const unsigned vout = (unsigned)((4 * h4e) * HW + pix) * 4u;
#pragma unroll
                    for (int r = 0; r < RN; ++r) {
                        const int c = F::row(r, lane_e);
                        const float v = acc2[i][r] + red[((wm * 2 + i) * RS + r) * 64 + lane_e];
                        if (!FT) {
                            if (c < c2) p.fuse_out[((size_t)plane0 + c) * M + m] = v;
                            continue;
                        }
                        // head_reduce_grouped_kernel's arithmetic: tiles summed in index order from 0, + bias, sigmoid
                        // (the running sum lives in LDS, each lane its own word: 32 registers less across the K loop)
                        float* run = sum_s + ((wm * 2 + i) * 8 + r) * 64 + lane_e;
                        const float vs = (t2 == 0 ? 0.f : *run) + v;
                        if (t2 + 1 < ntl) *run = vs;
                        else if (c < c2) {
                            float y = vs + bias_e[r];
                            if (p.fuse_gsig[hg]) y = 1.f / (1.f + expf(-y));
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), r_go, (int)vout, ((r & 3) + 8 * (r >> 2)) * HW * 4, 0);
                        }
                    }
                }
            }
        }
        if (EPI == 1) ++tn;
    }  // hidden tiles
    if constexpr (EPI == 1) return;
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if constexpr (EPI == 2) {
        // ---- fused ConvGRU gates (igemm16.hip's GRU epilogue): r = sig(x_r + h_r); z = sig(x_z + h_z);
        //      n = tanh(x_n + r * h_n); h' = (1 - z) * n + z * h   (convGRU.py:32-39) ----
        const int lc = lane_e & 31, h4 = lane_e >> 5;
        const int ch = tn * 32 + lc;
        {
            const int row0 = tn * 96 + lc;
            const float s0 = (p.scale ? p.scale[row0] : 1.f) * ainv, s1 = (p.scale ? p.scale[row0 + 32] : 1.f) * ainv,
                        s2 = (p.scale ? p.scale[row0 + 64] : 1.f) * ainv;
#pragma unroll
            for (int r = 0; r < F::NACC; ++r) { acc[0][0][r] *= s0; acc[0][1][r] *= s1; acc[0][2][r] *= s2; }
        }
        float amax = 0.f;
        // fragment of wave wm = patch rows 2 wm, 2 wm + 1; accumulator r of lane half h4 = pixel (r / 8, 8 (r / 4 % 2) + r % 4 + 4 h4)
        const int pix0 = __builtin_amdgcn_readfirstlane((b * p.H + ty0 + 2 * wm) * p.W + tx0);
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.gru_x3 + (size_t)pix0 * 192, (unsigned)(p.W + TW) * 192u * 4u);
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(p.gru_hprev + (size_t)pix0 * 64, (unsigned)(p.W + TW) * 64u * 4u);
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + (size_t)pix0 * 64, (unsigned)(p.W + TW) * 64u * 4u);
        const int vx = (4 * h4 * 192 + ch) * 4, vh = (4 * h4 * 64 + ch) * 4;
#pragma unroll
        for (int r = 0; r < F::NACC; ++r) {
            const int po = (r >> 3) * p.W + 8 * ((r >> 2) & 1) + (r & 3);
            const int sx = po * 192 * 4, sh = po * 64 * 4;
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
        if (p.out_amax) cp_amax_commit(p.out_amax, amax);
        return;
    }
    patch_epilogue<MT, NT, WM, WN>(p, acc, b, ty0, tx0, tn, wm, wn, lane_e, ainv);
    HALO_STAMP(21);  // epilogue done (stores acknowledged)
}

template <int MT, int NT, int WM, int WN, bool BDIRECT = false, int EPI = 0, int FT = 0>
int launch_halo(const ConvParams& p, hipStream_t stream) {
    constexpr int BN = 32 * NT * WN;
    // (fused heads that finish in the kernel: one workgroup per patch and head, ConvParams::fuse_final)
    const int tiles_m = p.B * (p.H / TH) * (p.W / TW), tiles_n = FT ? (p.fuse_final == 2 ? 1 : p.fuse_ngroups) : p.CoutPad / BN;
    hipLaunchKernelGGL((halo16_kernel<MT, NT, WM, WN, BDIRECT, EPI, FT>), dim3(tiles_m * tiles_n), dim3(256), 0, stream, p,
                       tiles_m, tiles_n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

}  // namespace

bool cp_halo16_supported(const ConvParams& p) {
    return p.w16_hi && p.w16_lo && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.nsrc == 1 && !p.offmask &&
           !p.gn_in_a && !p.gn_in_mr && p.splitk <= 1 && p.Cin % CK == 0 && p.H % TH == 0 && p.W % TW == 0 && p.H == p.Ho &&
           p.W == p.Wo && p.store == CP_STORE_NHWC && p.Kpad16 == 9 * p.Cin &&
           (size_t)p.B * p.H * p.W * p.Cin * 4 < (size_t)0xf0000000u && (size_t)p.B * p.H * p.W * p.ldo * 4 < (size_t)0xf0000000u;
}

// bn: N tile the weights were padded for (32 / 64 / 128)
int cp_launch_halo16(const ConvParams& p, int bn, hipStream_t stream) {
    if (!cp_halo16_supported(p) || p.CoutPad % bn != 0) return CP_ERR_INVALID;
    // weight fragments straight from the fragment-ordered copy when the layer has one (cp_set_debug 16384: the LDS-staged
    // weight tile instead, A/B runs)
    const bool direct = p.w16f_hi && p.w16f_lo && !(p.dbg & 16384);
    if (bn == 128) return direct ? launch_halo<2, 2, 2, 2, true>(p, stream) : launch_halo<2, 2, 2, 2>(p, stream);
    if (bn == 64) return direct ? launch_halo<2, 1, 2, 2, true>(p, stream) : launch_halo<2, 1, 2, 2>(p, stream);
    if (bn == 32) return direct ? launch_halo<1, 1, 4, 1, true>(p, stream) : launch_halo<1, 1, 4, 1>(p, stream);
    return CP_ERR_INVALID;
}

// the geometry / operand conditions of the two fused forms (their own launchers check the epilogue operands)
static bool halo16_fused_geometry(const ConvParams& p) {
    return p.w16f_hi && p.w16f_lo && !(p.dbg & 4096) && !(p.dbg & 16384) && p.KH == 3 && p.KW == 3 && p.stride == 1 &&
           p.pad == 1 && p.nsrc == 1 && !p.offmask && !p.gn_in_a && !p.gn_in_mr && !p.gn_stats && p.splitk <= 1 &&
           p.Cin % CK == 0 && p.H % TH == 0 && p.W % TW == 0 && p.H == p.Ho && p.W == p.Wo && p.Kpad16 == 9 * p.Cin &&
           (size_t)p.B * p.H * p.W * p.Cin * 4 < (size_t)0xf0000000u;
}

// fused prediction head on the halo-resident kernel (same operands as cp_launch_conv16_fused_head)
bool cp_halo16_fused_head_supported(const ConvParams& p) {
    if (p.fuse_final && (p.fuse_ngroups < 1 || p.Cin != CK || p.fuse_gtiles != 2 || p.CoutPad != p.fuse_ngroups * 256)) return false;
    if (p.fuse_final)  // the in-kernel finish keeps 16 final channels per head (accumulator rows r < 8 of the second product)
        for (int g = 0; g < p.fuse_ngroups; ++g)
            if (p.fuse_gc2[g] > 16) return false;
    return halo16_fused_geometry(p) && p.CoutPad % 128 == 0 && p.fuse_w2_hi && p.fuse_w2_lo && (p.fuse_out || p.fuse_final);
}
int cp_launch_halo16_fused_head(const ConvParams& p, hipStream_t stream) {
    if (!cp_halo16_fused_head_supported(p)) return CP_ERR_INVALID;
    if (p.fuse_final) return launch_halo<2, 2, 2, 2, true, 1, 2>(p, stream);  // 256 hidden channels = two tiles per head
    return launch_halo<2, 2, 2, 2, true, 1>(p, stream);
}

// fused ConvGRU hidden-side step on the halo-resident kernel (same operands as cp_launch_conv16_gru)
bool cp_halo16_gru_supported(const ConvParams& p) {
    return halo16_fused_geometry(p) && p.CoutPad == 192 && p.Cin == 64 && p.gru_x3 && p.gru_hprev && p.out &&
           (size_t)p.B * p.H * p.W * 192 * 4 < (size_t)0xf0000000u;
}
int cp_launch_halo16_gru(const ConvParams& p, hipStream_t stream) {
    if (!cp_halo16_gru_supported(p)) return CP_ERR_INVALID;
    return launch_halo<1, 3, 4, 1, true, 2>(p, stream);
}
