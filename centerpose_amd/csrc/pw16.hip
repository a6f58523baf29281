// Point-wise (1x1, stride 1) convolution in the split-f16 ("f16x3") arithmetic of igemm16.hip, as a register-only stream:
// the Root nodes of DLA-34 (conv1x1 over a virtual concat of 2..4 tensors, pose_dla_dcn.py:160-168), the `project`
// convolutions (:211-224) and every other 1x1 of the backbone.
//
// Why not the per-tap implicit GEMM.  With K = 32 .. 1280 these layers are 1 - 40 K tiles long: the LDS-staged loop of
// igemm16p_kernel (global -> registers -> convert -> LDS -> barrier -> fragments) spends its time filling and draining
// its pipeline and measures 35 - 150 TFLOP/s at 2.3 - 2.6 TB/s of activation traffic, under both of its rooflines.  A 1x1
// has no tap reuse, so the LDS A tile buys nothing: a lane's MFMA A fragment (pixel = lane % 32, 8 consecutive channels)
// is 32 contiguous bytes of the NHWC tensor.  Here
//   * every wave owns 32 pixels x the whole N tile (64 or 128 output channels) and runs on its own: no LDS, no barrier;
//   * A fragments come straight from global memory (two 16-byte buffer loads per lane and K step of 16 channels), six
//     K steps ahead (they stream from HBM), are scaled by the tensor's power-of-two pre-scale and split to hi / lo in
//     registers;
//   * B fragments come from the fragment-ordered weight copy (cp_launch_frag16_repack: one coalesced 1 KB load per
//     fragment, L2-resident), three K steps ahead;
//   * a virtual concat is a change of base pointer at a K-step boundary (every source has a multiple of 32 channels).
// Products are hi*hi + hi*lo + lo*hi with float32 accumulation, K order = channel order, as in igemm16p_kernel.
// (Round 6: the A operand is loaded in whole lines by pw16s_kernel below -- the default -- and this kernel is its A/B partner.)
// Measured on the dlav1_34 batch-32 step: the ten 1x1 layers 0.79 -> 0.60 ms; the short-K layers now stream at
// 2.7 - 3.8 TB/s of activation traffic, the long-K Root nodes (K = 448 .. 1280) reach 150 - 190 TFLOP/s where the texture
// path's load-instruction rate (10 sixteen-byte loads per wave and K step against 12 MFMAs) is what binds.
#include "igemm16_common.h"

namespace {

constexpr int DA = 6;  // A register sets = K steps in flight
constexpr int DB = 3;  // B register sets

template <int NT>
__global__ __launch_bounds__(256, 2) void pw16_kernel(const ConvParams p, const int tiles_m, const int tiles_n) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = tile_of_block(tiles_m, tiles_n);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int M = p.B * p.Ho * p.Wo;
    float afwd, ainv;
    conv_in_scale(p, &afwd, &ainv);

    // ---- A: this lane's pixel row and 8-channel half of a K step; one resource and one row offset per source ----
    const int row = tm * 128 + wid * 32 + (lane & 31), half = lane >> 5;
    const int n0 = p.src_c[0] >> 4, n1 = p.nsrc > 1 ? p.src_c[1] >> 4 : 0, n2 = p.nsrc > 2 ? p.src_c[2] >> 4 : 0;  // K steps
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(p.src[0], (unsigned)M * (unsigned)p.src_c[0] * 4u);
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(p.nsrc > 1 ? p.src[1] : p.src[0], p.nsrc > 1 ? (unsigned)M * (unsigned)p.src_c[1] * 4u : 0u);
    const __amdgpu_buffer_rsrc_t r2 = make_rsrc(p.nsrc > 2 ? p.src[2] : p.src[0], p.nsrc > 2 ? (unsigned)M * (unsigned)p.src_c[2] * 4u : 0u);
    const __amdgpu_buffer_rsrc_t r3 = make_rsrc(p.nsrc > 3 ? p.src[3] : p.src[0], p.nsrc > 3 ? (unsigned)M * (unsigned)p.src_c[3] * 4u : 0u);
    const bool live = row < M;
    const unsigned v0 = live ? (unsigned)(row * p.src_c[0] + half * 8) * 4u : OOB;
    const unsigned v1 = live && p.nsrc > 1 ? (unsigned)(row * p.src_c[1] + half * 8) * 4u : OOB;
    const unsigned v2 = live && p.nsrc > 2 ? (unsigned)(row * p.src_c[2] + half * 8) * 4u : OOB;
    const unsigned v3 = live && p.nsrc > 3 ? (unsigned)(row * p.src_c[3] + half * 8) * 4u : OOB;
    const int G = p.Kpad16 >> 4;  // K steps of 16 channels

    // ---- B: fragment (n tile j, K step g) = 1 KB in lane order at ((j G + g) 64 + lane) 16 bytes ----
    const unsigned w_bytes = (unsigned)((size_t)p.CoutPad * p.Kpad16 * 2);
    const __amdgpu_buffer_rsrc_t r_wh = make_rsrc(p.w16f_hi, w_bytes), r_wl = make_rsrc(p.w16f_lo, w_bytes);
    unsigned bd_off[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bd_off[j] = (unsigned)((((tn * NT + j) * G) * 64 + lane) * 16);

    u32x4 ra[DA][2];            // raw float32 A: 8 channels of this lane's pixel
    u32x4 rbh[DB][NT], rbl[DB][NT];
    auto issue_a = [&](int set, int g) {  // K step g -> (source, step inside it): wave-uniform scalar work
        if (g >= G) return;
        int kk = g;
        if (kk < n0) {
            ra[set][0] = __builtin_amdgcn_raw_buffer_load_b128(r0, (int)v0, kk * 64, 0);
            ra[set][1] = __builtin_amdgcn_raw_buffer_load_b128(r0, (int)v0, kk * 64 + 16, 0);
            return;
        }
        kk -= n0;
        if (kk < n1) {
            ra[set][0] = __builtin_amdgcn_raw_buffer_load_b128(r1, (int)v1, kk * 64, 0);
            ra[set][1] = __builtin_amdgcn_raw_buffer_load_b128(r1, (int)v1, kk * 64 + 16, 0);
            return;
        }
        kk -= n1;
        if (kk < n2) {
            ra[set][0] = __builtin_amdgcn_raw_buffer_load_b128(r2, (int)v2, kk * 64, 0);
            ra[set][1] = __builtin_amdgcn_raw_buffer_load_b128(r2, (int)v2, kk * 64 + 16, 0);
            return;
        }
        kk -= n2;
        ra[set][0] = __builtin_amdgcn_raw_buffer_load_b128(r3, (int)v3, kk * 64, 0);
        ra[set][1] = __builtin_amdgcn_raw_buffer_load_b128(r3, (int)v3, kk * 64 + 16, 0);
    };
    auto issue_b = [&](int set, int g) {
        if (g >= G) return;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            rbh[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wh, (int)bd_off[j], g * 1024, 0);
            rbl[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wl, (int)bd_off[j], g * 1024, 0);
        }
    };

    acc_t acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < F::NACC; ++r) acc[0][j][r] = 0.f;

#pragma unroll
    for (int u = 0; u < DB; ++u) issue_b(u, u);
#pragma unroll
    for (int u = 0; u < DA; ++u) issue_a(u, u);

    for (int g0 = 0; g0 < G; g0 += DA) {
#pragma unroll
        for (int u = 0; u < DA; ++u) {
            const int g = g0 + u;
            if (g < G) {  // wave-uniform
                // split this step's A to hi / lo, then refill its registers with step g + DA before the MFMAs
                const u32x4 x0 = ra[u][0], x1 = ra[u][1];
                const Split2 s0 = split2(__uint_as_float(x0.x) * afwd, __uint_as_float(x0.y) * afwd);
                const Split2 s1 = split2(__uint_as_float(x0.z) * afwd, __uint_as_float(x0.w) * afwd);
                const Split2 s2 = split2(__uint_as_float(x1.x) * afwd, __uint_as_float(x1.y) * afwd);
                const Split2 s3 = split2(__uint_as_float(x1.z) * afwd, __uint_as_float(x1.w) * afwd);
                const u32x4 hv = {s0.hi, s1.hi, s2.hi, s3.hi}, lv = {s0.lo, s1.lo, s2.lo, s3.lo};
                const h8 ah = *reinterpret_cast<const h8*>(&hv), al = *reinterpret_cast<const h8*>(&lv);
                issue_a(u, g + DA);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int DBm = DB;
                const int sb = u % DBm;  // DA is a multiple of DB: the B set index is static too
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, *reinterpret_cast<const h8*>(&rbh[sb][j]), acc[0][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, *reinterpret_cast<const h8*>(&rbl[sb][j]), acc[0][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, *reinterpret_cast<const h8*>(&rbh[sb][j]), acc[0][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                issue_b(sb, g + DB);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    igemm_epilogue<32, 1, NT, 4, 1>(p, acc, tm, tn, wid, 0, lane, ainv);
}

// The same stream with the A operand loaded in WHOLE LINES and turned into fragments by a wave-private LDS round trip (round 6).
// pw16_kernel's fragment-shaped loads take 16 bytes from each of 64 different 128-byte lines per instruction (a lane = one pixel's
// 8 channels), and every line is visited by four instructions of two K steps -- the texture path moves a line per lane, not a line
// per 8 lanes.  Here a K chunk is 32 channels = exactly one 128-byte line per pixel: lane l loads piece l % 8 of pixel l / 8 + 8 i
// (four 1 KB instructions per chunk, each 8 whole lines), scales and splits it, writes hi / lo halves to the wave's 5 KB staging
// rows (pixel pitch 80 B: conflict-free b64 writes and b128 fragment reads), and reads back the two K steps' fragments.  No
// barrier: the rows are private to the wave and a wave's LDS operations execute in order.  Chunks are requested PA = 3 ahead
// (96 channels, as pw16_kernel's six K steps); B fragments, epilogue and summation order are pw16_kernel's: bit-identical results.
constexpr int PA = 3;              // A chunk register sets = 32-channel chunks in flight
constexpr int PW_PITCH = 80;       // bytes per pixel and plane of the staging rows (32 halfs + 16 B)
constexpr int PW_PLANE = 32 * PW_PITCH;

template <int NT>
__global__ __launch_bounds__(256, NT == 2 ? 3 : 2) void pw16s_kernel(const ConvParams p, const int tiles_m, const int tiles_n) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    __shared__ __attribute__((aligned(16))) unsigned char stage_s[4][2 * PW_PLANE];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = tile_of_block(tiles_m, tiles_n);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int M = p.B * p.Ho * p.Wo;
    float afwd, ainv;
    conv_in_scale(p, &afwd, &ainv);
    unsigned char* st_hi = stage_s[wid];
    unsigned char* st_lo = st_hi + PW_PLANE;

    // ---- A: piece c4 = lane % 8 (16 bytes = 4 channels) of pixels lane / 8 + 8 i, i = 0 .. 3 ----
    const int c4 = lane & 7, row0 = tm * 128 + wid * 32 + (lane >> 3);
    const int n0 = p.src_c[0] >> 5, n1 = p.nsrc > 1 ? p.src_c[1] >> 5 : 0, n2 = p.nsrc > 2 ? p.src_c[2] >> 5 : 0;  // chunks per source
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(p.src[0], (unsigned)M * (unsigned)p.src_c[0] * 4u);
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(p.nsrc > 1 ? p.src[1] : p.src[0], p.nsrc > 1 ? (unsigned)M * (unsigned)p.src_c[1] * 4u : 0u);
    const __amdgpu_buffer_rsrc_t r2 = make_rsrc(p.nsrc > 2 ? p.src[2] : p.src[0], p.nsrc > 2 ? (unsigned)M * (unsigned)p.src_c[2] * 4u : 0u);
    const __amdgpu_buffer_rsrc_t r3 = make_rsrc(p.nsrc > 3 ? p.src[3] : p.src[0], p.nsrc > 3 ? (unsigned)M * (unsigned)p.src_c[3] * 4u : 0u);
    const int GC = p.Kpad16 >> 5;  // 32-channel chunks

    const unsigned w_bytes = (unsigned)((size_t)p.CoutPad * p.Kpad16 * 2);
    const __amdgpu_buffer_rsrc_t r_wh = make_rsrc(p.w16f_hi, w_bytes), r_wl = make_rsrc(p.w16f_lo, w_bytes);
    const int G = p.Kpad16 >> 4;
    unsigned bd_off[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bd_off[j] = (unsigned)((((tn * NT + j) * G) * 64 + lane) * 16);

    u32x4 ra[PA][4];
    u32x4 rbh[DB][NT], rbl[DB][NT];
    auto load4 = [&](u32x4 (&dst)[4], const __amdgpu_buffer_rsrc_t r, const int cs, const int kk) {
        // lane offset = its first pixel's row, the other three pixels (8 rows on each) and the chunk in the scalar offset; rows >= M
        // are beyond the descriptor (the range check counts the scalar offset in: cp_common.h's architecture guard) -> zeros
        const unsigned v = (unsigned)row0 * (unsigned)(cs * 4) + (unsigned)(c4 * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)v, kk * 128 + i * 8 * cs * 4, 0);
    };
    auto issue_a = [&](int set, int c) {  // chunk c -> (source, chunk inside it): wave-uniform scalar work
        if (c >= GC) return;
        int kk = c;
        if (kk < n0) { load4(ra[set], r0, p.src_c[0], kk); return; }
        kk -= n0;
        if (kk < n1) { load4(ra[set], r1, p.src_c[1], kk); return; }
        kk -= n1;
        if (kk < n2) { load4(ra[set], r2, p.src_c[2], kk); return; }
        kk -= n2;
        load4(ra[set], r3, p.src_c[3], kk);
    };
    auto issue_b = [&](int set, int g) {
        if (g >= G) return;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            rbh[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wh, (int)bd_off[j], g * 1024, 0);
            rbl[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r_wl, (int)bd_off[j], g * 1024, 0);
        }
    };

    acc_t acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < F::NACC; ++r) acc[0][j][r] = 0.f;

#pragma unroll
    for (int u = 0; u < DB; ++u) issue_b(u, u);
#pragma unroll
    for (int u = 0; u < PA; ++u) issue_a(u, u);

    const int wr_off = (lane >> 3) * PW_PITCH + c4 * 8;              // + 8 i pixels
    const int rd_off = (lane & 31) * PW_PITCH + (lane >> 5) * 16;    // + 32 h for the chunk's K step h
    static_assert((2 * PA) % DB == 0, "the B register set of a K step must be a compile-time index");
    for (int c0 = 0; c0 < GC; c0 += PA) {
#pragma unroll
        for (int u = 0; u < PA; ++u) {
            const int c = c0 + u;
            if (c < GC) {  // wave-uniform
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 x = ra[u][i];
                    const Split2 s0 = split2(__uint_as_float(x.x) * afwd, __uint_as_float(x.y) * afwd);
                    const Split2 s1 = split2(__uint_as_float(x.z) * afwd, __uint_as_float(x.w) * afwd);
                    *reinterpret_cast<u32x2*>(st_hi + wr_off + 8 * i * PW_PITCH) = u32x2{s0.hi, s1.hi};
                    *reinterpret_cast<u32x2*>(st_lo + wr_off + 8 * i * PW_PITCH) = u32x2{s0.lo, s1.lo};
                }
                issue_a(u, c + PA);
                __builtin_amdgcn_wave_barrier();
                h8 ah[2], al[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    ah[h] = *reinterpret_cast<const h8*>(st_hi + rd_off + 32 * h);
                    al[h] = *reinterpret_cast<const h8*>(st_lo + rd_off + 32 * h);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int g = 2 * c + h;
                    const int sb = (2 * u + h) % DB;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[h], *reinterpret_cast<const h8*>(&rbh[sb][j]), acc[0][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[h], *reinterpret_cast<const h8*>(&rbl[sb][j]), acc[0][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[h], *reinterpret_cast<const h8*>(&rbh[sb][j]), acc[0][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    issue_b(sb, g + DB);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    igemm_epilogue<32, 1, NT, 4, 1>(p, acc, tm, tn, wid, 0, lane, ainv);
}

template <int NT>
int launch_pw16s(const ConvParams& p, hipStream_t stream) {
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_m = (M + 127) / 128, tiles_n = p.CoutPad / (32 * NT);
    hipLaunchKernelGGL((pw16s_kernel<NT>), dim3(tiles_m * tiles_n), dim3(256), 0, stream, p, tiles_m, tiles_n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

template <int NT>
int launch_pw16(const ConvParams& p, hipStream_t stream) {
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_m = (M + 127) / 128, tiles_n = p.CoutPad / (32 * NT);
    hipLaunchKernelGGL((pw16_kernel<NT>), dim3(tiles_m * tiles_n), dim3(256), 0, stream, p, tiles_m, tiles_n);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

}  // namespace

static_assert(DA % DB == 0, "the B register set of a K step must be a compile-time index");

// 1x1 / stride 1 / no padding, NHWC in and out of the same size, fragment-ordered weights present, every source a
// multiple of 32 channels, 32-bit byte offsets, no split-K and none of the fused loaders / epilogues of the other kernels
bool cp_pw16_supported(const ConvParams& p) {
    if (!p.w16f_hi || !p.w16f_lo || p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad != 0 || p.H != p.Ho || p.W != p.Wo)
        return false;
    if (p.offmask || p.gn_in_a || p.gn_in_d || p.gn_in_mr || p.gn_stats || p.splitk > 1 || p.nsrc < 1 || p.nsrc > 4)
        return false;
    if (p.CoutPad % 64 != 0 || p.Kpad16 != p.Cin || p.Cin % 32 != 0) return false;
    int c = 0;
    const size_t M = (size_t)p.B * p.Ho * p.Wo;
    for (int s = 0; s < p.nsrc; ++s) {
        if (p.src_c[s] % 32 != 0 || M * p.src_c[s] * 4 >= (size_t)0xf0000000u) return false;
        c += p.src_c[s];
    }
    return c == p.Cin && (size_t)p.CoutPad * p.Kpad16 * 2 < (size_t)0x7fffffff && M * p.ldo * 4 < (size_t)0xf0000000u;
}

int cp_launch_pw16(const ConvParams& p, hipStream_t stream) {
    if (!cp_pw16_supported(p)) return CP_ERR_INVALID;
    // whole-line A loads through wave-private staging rows (cp_set_debug 4: the fragment-shaped loads of pw16_kernel, A/B runs)
    if (!(p.dbg & 4)) return p.CoutPad % 128 == 0 ? launch_pw16s<4>(p, stream) : launch_pw16s<2>(p, stream);
    return p.CoutPad % 128 == 0 ? launch_pw16<4>(p, stream) : launch_pw16<2>(p, stream);
}
