// centerpose_amd — shared declarations for the HIP kernels (gfx950 / CDNA4 only).
//
// Activations are float32 NHWC ("pixel-major": all channels of one output pixel are
// contiguous).  That is the MI355X-first layout choice for this path: the DCNv2 bilinear gather
// and the implicit-GEMM loaders then read whole channel vectors (16 B per lane, 64-256 B per
// pixel) instead of the reference's per-channel NCHW planes (dcn_v2_im2col_cuda.cu:125-195
// touches 4 scattered floats per (channel, tap)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The kernels of this library rely on the raw-buffer range check of the gfx9 / CDNA family as gfx950 implements it: a lane whose
// (voffset + soffset [+ immediate]) lies beyond the descriptor's num_records reads 0 -- per dword for a 16-byte access that
// straddles the end -- and its stores / LDS-DMA writes are dropped (zeros land in LDS).  Out-of-picture halo pixels, padded
// weight / bias rows, missing split-K slices (igemm.hip: splitk_epilogue4_kernel) and "invalid corner" gathers are all
// expressed that way (OOB / OOB_BASE in igemm16_common.h) instead of as branches.  Other architectures check differently
// (gfx10+ exclude soffset in some modes): refuse to build device code for anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "centerpose_amd device code is written for gfx950 only (raw-buffer range-check semantics, MFMA shapes, LDS-DMA)"
#endif

#define CP_OK 0
#define CP_ERR_INVALID (-1)
#define CP_ERR_LAUNCH (-2)
#define CP_ERR_ALLOC (-3)
#define CP_ERR_STATE (-4)

enum CpAct : int {
    CP_ACT_NONE = 0,
    CP_ACT_RELU = 1,
    CP_ACT_SIGMOID = 2,       // all channels
    CP_ACT_SIGMOID_FROM = 3,  // sigmoid on channels >= act_from (DCN mask logits), identity below
};

enum CpStore : int {
    CP_STORE_NHWC = 0,  // out[(pixel) * ldo + coff + c]
    CP_STORE_NCHW = 1,  // out[((b * Cout_total) + coff + c) * HoWo + p]   (reference head layout)
};

#define CP_MAX_SRC 4
#define CP_MAX_HEAD_GROUP 12

#ifdef __HIPCC__
// ---- |max| tracking and power-of-two operand scaling for the split-f16 kernels (see ConvParams::in_amax) ----
// amax bits -> (2^e, 2^-e) with amax * 2^e in [2^14, 2^15): e = 14 - floor(log2(amax)).  Exponents are clamped so that
// both factors stay normal float32 numbers (amax = 0 or < 2^-111: 2^125; inf / NaN inputs stay inf / NaN).
__device__ __forceinline__ void cp_amax_to_scale(unsigned amax_bits, float* fwd, float* inv) {
    int e = (int)((amax_bits >> 23) & 0xffu);
    e = e < 16 ? 16 : (e > 253 ? 253 : e);
    *fwd = __uint_as_float((unsigned)(268 - e) << 23);
    *inv = __uint_as_float((unsigned)(e - 14) << 23);
}
// A tensor's |max| lives in CP_AMAX_SUB sub-slots CP_AMAX_STRIDE uints apart (different cache lines): all waves of a
// small kernel finish at about the same time, all find the slot still at its old value and all issue their atomic --
// thousands of same-address atomics serialise at the memory side (measured: +10 us per kernel at batch 1).  Spreading
// the writers over 32 addresses by workgroup (one atomic per workgroup) cuts that to a handful per address; readers
// take the max.
#define CP_AMAX_SUB 32
#define CP_AMAX_STRIDE 2048  // = |max| slots (tensors) per forward pass
__device__ __forceinline__ unsigned cp_amax_read(const unsigned* slot) {
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < CP_AMAX_SUB; ++k) m = max(m, slot[k * CP_AMAX_STRIDE]);
    return m;
}
// Block-reduce the lanes' local max|v| (wave shuffle, then one LDS word per wave) and fold it into the block's sub-slot
// (float bits of non-negative numbers order like unsigned integers).  The sub-slot is read first: later blocks usually
// find their maximum covered and skip the atomic.  Every thread of the block must call it (it contains a barrier).
__device__ __forceinline__ void cp_amax_commit(unsigned* slot, float local) {
    __shared__ float cp_amax_red[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local = fmaxf(local, __shfl_xor(local, o, 64));
    if ((threadIdx.x & 63) == 0) cp_amax_red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (int)(blockDim.x + 63) >> 6;
        float m = cp_amax_red[0];
        for (int i = 1; i < nw; ++i) m = fmaxf(m, cp_amax_red[i]);
        unsigned* sub = slot + (blockIdx.x & (CP_AMAX_SUB - 1)) * CP_AMAX_STRIDE;
        const unsigned b = __float_as_uint(m);
        if (b > __hip_atomic_load(sub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(sub, b);
    }
}

// Gate non-linearities of the fused ConvGRU epilogues (f16x3 mode): hardware exp2 / rcp, absolute error < 3e-7 (sigmoid)
// and < 6e-7 (tanh) against expf / tanhf, i.e. at float32 round-off of the gate values.  The library forms cost ~80 VALU
// instructions per output (range reduction, IEEE division, tanhf's branches) -- with 48 outputs per lane that was more
// vector-ALU time than the K loop has matrix time.  The exact-f32 mode and the stand-alone gate kernel keep expf / tanhf.
__device__ __forceinline__ float cp_fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float cp_fast_tanh(float x) {
    return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.f;
}
#endif

// One implicit-GEMM convolution:  out[m, n] = act( (sum_k A[m,k] * Wp[k,n]) * scale[n] + shift[n] + res[m,n] )
//   m = output pixel (b, ho, wo);  k = (tap, ci) with ci fastest;  n = output channel.
// A is read on the fly from up to CP_MAX_SRC NHWC sources that form a virtual channel concat
// (Root's torch.cat, pose_dla_dcn.py:162), or — in DCN mode — produced by the modulated bilinear
// gather of the reference's deformable im2col (dcn_v2_im2col_cuda.cu:125-195) without ever
// materialising the `columns` buffer.
struct ConvParams {
    const float* src[CP_MAX_SRC];
    int src_c[CP_MAX_SRC];  // channels per source (multiples of 4)
    int nsrc;
    int Cin;  // total input channels
    int B, H, W, Ho, Wo;
    int KH, KW, stride, pad;
    int K;     // KH*KW*Cin (un-padded)
    int Kpad;  // rows of wp (multiple of BK)
    const float* wp;  // packed weights [Kpad][CoutPad]
    int Cout, CoutPad;
    const float* scale;  // [CoutPad] or nullptr (=1)
    const float* shift;  // [CoutPad] or nullptr (=0)
    const float* res;    // NHWC [M][res_ld] or nullptr
    int res_ld;
    int act;
    int act_from;
    float* out;
    int store;       // CpStore
    int ldo;         // NHWC: channel stride of out; NCHW: total channels of out tensor
    int coff;        // channel offset inside out
    const void* w16_hi;    // split-f16 path: packed weights [CoutPad][Kpad16] binary16 (hi / lo), or nullptr
    const void* w16_lo;
    // the same two arrays in MFMA B-operand order (cp_launch_frag16_repack), optional: dcn16p.hip loads its weight
    // fragments straight from them
    const void* w16f_hi;
    const void* w16f_lo;
    int Kpad16;
    // GroupNorm fusion (dlav1 heads, GN.py:4-9): the producing conv accumulates per-(image, group) sum / sum of
    // squares of its biased output into gn_stats[B][groups][2] (doubles, zeroed by the caller); the consuming 1x1
    // conv normalises + affine + ReLU on load from gn_in_mr[B][groups] = (mean, rstd).
    double* gn_stats;
    int gn_groups, gn_cpg;
    const float* gn_in_mr;
    const float* gn_in_gamma;
    const float* gn_in_beta;
    // the same fusion for the f16x3 1x1 kernel: the normalisation pre-folded per (image, channel) to y = relu(a*x + d),
    // planes [B][Cin] each (cp_launch_gn_affine)
    const float* gn_in_a;
    const float* gn_in_d;
    // fused ConvGRU step (convGRU.py:32-39) on the hidden-side 3x3 convolution: weights packed so that an N tile of 96
    // holds [r | z | n] of the same 32 channels; the epilogue reads gru_x3 ([M,192] = input-side pre-activations) and
    // gru_hprev ([M,64]) and writes h' = (1-z)*n + z*h into out ([M,64]).  The [M,192] hidden-side tensor never exists.
    const float* gru_x3;
    const float* gru_hprev;
    // deterministic split-K (few output tiles, long K: low-resolution layers at small batch): blockIdx.y = K slice,
    // raw accumulators go to partial[slice][M][CoutPad]; cp_launch_splitk_epilogue sums the slices in order and
    // applies the usual epilogue.
    int splitk;
    float* partial;
    // tile of a split-K launch when it should be smaller than the default 128 x cp_conv_tile_n(Cout) (0 = default): the slab
    // volume is slices x M x Cout x 4 bytes, and smaller tiles reach the same workgroup count with fewer slices
    int tile_m, tile_n;
    // Fused prediction head (dlav1 heads: conv3x3 -> ReLU -> conv1x1, pose_dla_dcn.py head Sequential): the 3x3 tile
    // is multiplied by the 1x1 weights inside the kernel and only [slices = CoutPad/128][fuse_c2][M] partial sums of the
    // final maps are written (fuse_out); cp_launch_head_reduce adds the slices + bias (+ sigmoid) into NCHW.  The
    // 256-channel hidden tensor never exists in memory.  fuse_w2_*: 1x1 weights packed as MFMA fragments
    // (cp_launch_pack_head_w2).
    const void* fuse_w2_hi;
    const void* fuse_w2_lo;
    float* fuse_out;
    int fuse_c2;
    // Several fused heads that read the same input in ONE launch (halo16.hip, EPI = 1): the 3x3 weights, scale / shift, 1x1
    // fragments and w2_inv tables of the heads are concatenated along N (CoutPad = sum of the heads' hidden widths); N
    // tile tn belongs to head g = tn / fuse_gtiles, whose final maps have fuse_gc2[g] channels and whose slabs start at
    // plane fuse_gbase[g] of fuse_out ([plane][M]; head g owns fuse_gtiles * fuse_gc2[g] planes, slice-major).
    // fuse_ngroups == 0: one head, fuse_c2 channels, planes from 0 (the layout above).
    int fuse_ngroups, fuse_gtiles;
    int fuse_gc2[CP_MAX_HEAD_GROUP];
    int fuse_gbase[CP_MAX_HEAD_GROUP];
    // fuse_final (halo16 grouped launch, Cin == 64): a workgroup stages its patch once, walks all fuse_gtiles hidden tiles of
    // its head and writes the finished maps -- sum of the tiles in index order + bias (+ sigmoid), NCHW, the arithmetic of
    // head_reduce_grouped_kernel -- so neither the slabs nor the reduction launch exist.  fuse_out is then unused.
    int fuse_final;
    int fuse_gsig[CP_MAX_HEAD_GROUP];
    const float* fuse_gbias[CP_MAX_HEAD_GROUP];
    float* fuse_gout[CP_MAX_HEAD_GROUP];
    // ---- range-safe split-f16 arithmetic (f16x3 kernels) ----
    // Binary16 only has 5 exponent bits, so the hi/lo split is exact to 2^-21 only while the operand sits well inside
    // the normal range.  Both operands are therefore pre-scaled by exact powers of two: weights per output channel at
    // pack time (the inverse is folded into `scale`, which for f16x3 launches points at scale * 2^-e_w), activations per
    // tensor at run time from the tensor's running |max| (in_amax: float bits of max|x| written by the producing
    // kernel's epilogue into a 4-byte slot; several sources of a virtual concat share the largest).  The loader
    // multiplies by 2^e_a before the split and the epilogue by 2^-e_a: every partial sum is scaled by the same power
    // of two, so results are bit-identical to the unscaled arithmetic wherever that one was inside the normal range.
    const unsigned* in_amax[CP_MAX_SRC];  // nullptr: operand used as is (scale 1)
    unsigned* out_amax;                   // nullptr: the output's |max| is not tracked
    const float* fuse_w2_inv;             // fused head: 2^-e of the 1x1 weights per final channel [32]
    int dbg;  // tuning ablations (tools/conv_bench.py --dbg): 1 skip A loads, 2 skip B loads, 4 skip LDS stores, 8 skip MFMA
    const float* offmask;  // DCN mode: NHWC [B,H,W,32] = 18 offsets (dh,dw interleaved per tap) + 9 masks (already sigmoided) + 5 pad
};

int cp_launch_conv(const ConvParams& p, hipStream_t stream);
// Tile N-width the launcher will pick for `cout` (weights must be padded to a multiple of it).
int cp_conv_tile_n(int cout);
int cp_conv_variant(const ConvParams& p);
int cp_launch_splitk_epilogue(const ConvParams& p, hipStream_t stream);
// K steps (of 16 for the f32 kernels, 32 for f16x3) and output tiles of the launch cp_launch_conv[16] would make
void cp_conv_geometry(const ConvParams& p, bool f16x3, int* tiles, int* nk);
const char* cp_conv_variant_name(int v);
#define CP_NUM_CONV_VARIANTS 43
#define CP_VARIANT_STRM16 41
#define CP_VARIANT_LOWC1S 42
#define CP_VARIANT_GRU 26
// split-f16 ("f16x3") implicit GEMM (igemm16.hip)
bool cp_conv16_supported(const ConvParams& p);
int cp_launch_conv16(const ConvParams& p, hipStream_t stream);
int cp_conv16_variant(const ConvParams& p);
// halo-resident 3x3 / stride-1 convolution (halo16.hip): eligibility and launch (bn = N tile 32 / 64 / 128)
bool cp_halo16_supported(const ConvParams& p);
bool cp_halo16_fused_head_supported(const ConvParams& p);
int cp_launch_halo16_fused_head(const ConvParams& p, hipStream_t stream);
bool cp_halo16_gru_supported(const ConvParams& p);
int cp_launch_halo16_gru(const ConvParams& p, hipStream_t stream);
int cp_launch_halo16(const ConvParams& p, int bn, hipStream_t stream);
// strm16.hip: 64 -> <= 32 channel 3x3 / stride-1 layers (DCN offset / mask convolutions) as wave-private row streams, weights in LDS
bool cp_strm16_supported(const ConvParams& p);
int cp_strm16_jobs(const ConvParams& p);
int cp_launch_strm16(const ConvParams& p, hipStream_t stream);
// fused DCNv2 gather + contraction (dcn16.hip); bn = N tile (64 / 128), variant = alternative wave count (tuning)
int cp_launch_dcn16(const ConvParams& p, int bn, int variant, hipStream_t stream);
// dcn16p.hip: patch-resident DCNv2 (gather from an LDS-staged halo); N tile 64, or 128 where cp_dcn16p_wide says so
// pw16.hip: 1x1 / stride-1 layers (incl. virtual concats) as a register-only stream, weight fragments from w16f_*
bool cp_pw16_supported(const ConvParams& p);
int cp_launch_pw16(const ConvParams& p, hipStream_t stream);
bool cp_dcn16p_supported(const ConvParams& p);
int cp_dcn16p_blocks(const ConvParams& p);
int cp_launch_dcn16p(const ConvParams& p, hipStream_t stream);
bool cp_dcn16p_wide(const ConvParams& p);  // the launch takes the 128-wide N tile (NT = 4)
// dcn16s.hip: the same gather as a persistent kernel with the halo streamed by LDS-DMA into two 16-channel buffers
// (launches with several (patch, N tile) items per resident workgroup)
#define CP_VARIANT_DCN16S 36
#define CP_VARIANT_M64N64 37  // igemm16p on 64 x 64 tiles (small launches)
#define CP_VARIANT_DCN16PW 38  // dcn16p on the 128-wide N tile
bool cp_dcn16s_supported(const ConvParams& p);
int cp_dcn16s_items(const ConvParams& p);
int cp_launch_dcn16s(const ConvParams& p, hipStream_t stream);
int cp_launch_frag16_repack(const void* w16, void* w16f, int CoutPad, int Kpad16, hipStream_t s);
// dcn16t.hip: dcn16p's gather written for three workgroups per CU (16-channel chunks, one gather set, two weight sets)
#define CP_VARIANT_DCN16T 39
bool cp_dcn16t_supported(const ConvParams& p);
int cp_launch_dcn16t(const ConvParams& p, hipStream_t stream);
#define CP_VARIANT_DCN16P 30
#define CP_VARIANT_GN_FINAL 31
#define CP_VARIANT_HALO_HEAD 32
#define CP_VARIANT_HALO_GRU 33
#define CP_VARIANT_PW16 34  // + 1 for the 128-wide N tile
// `fwd` (may be nullptr = 1): per-output-channel power-of-two factor applied before the split, indexed [coff + co]
int cp_launch_pack_weight16(const float* w, void* hi, void* lo, int Cout, int Cin, int taps, int Kpad16, int coff,
                            const float* fwd, hipStream_t s);
// per-output-channel power-of-two scale of a [Cout][per] weight: fwd[co] = 2^e with max|w[co,:]| * 2^e in [2^14, 2^15),
// inv[co] = 2^-e (all-zero rows: 1)
int cp_launch_weight_scale(const float* w, int Cout, int per, float* fwd, float* inv, hipStream_t s);
// out[i] = (scale ? scale[i] : 1) * inv[i]
int cp_launch_scale16(const float* scale, const float* inv, float* out, int n, hipStream_t s);
// |max| of x[0..n) folded into the slot's sub-slots (CP_AMAX_SUB x CP_AMAX_STRIDE uints, zeroed by the caller);
// n % 4 == 0, 16-byte aligned
int cp_launch_absmax(const float* x, size_t n, unsigned* slot, hipStream_t s);
// fused head (see ConvParams::fuse_*): is this 3x3 (+ReLU) -> 1x1 pair eligible; launch; 1x1 weight packing
// (w1: [C2][Chid] float32 -> two arrays of Chid*32 binary16); slice reduction + bias (+ sigmoid) -> NCHW
bool cp_head_fuse_supported(const ConvParams& p, int c2);
int cp_launch_conv16_fused_head(const ConvParams& p, hipStream_t stream);
// w2_inv: [32] floats, receives 2^-e per final channel (the packed rows are scaled by 2^e)
int cp_launch_pack_head_w2(const float* w1, void* hi, void* lo, float* w2_inv, int C2, int Chid, hipStream_t s);
int cp_launch_head_reduce(const float* slabs, const float* bias, float* out_nchw, int slices, int C2, int B, int HW,
                          int sigmoid, hipStream_t s);
// the same for a group of heads in one launch: head g reads planes base[g] .. base[g] + slices * c2[g] of `slabs`
struct HeadReduceGroup {
    int n, slices;
    int c2[CP_MAX_HEAD_GROUP], base[CP_MAX_HEAD_GROUP], sigmoid[CP_MAX_HEAD_GROUP];
    const float* bias[CP_MAX_HEAD_GROUP];
    float* out[CP_MAX_HEAD_GROUP];
};
int cp_launch_head_reduce_grouped(const float* slabs, const HeadReduceGroup& g, int B, int HW, hipStream_t s);
int cp_launch_conv16_gru(const ConvParams& p, hipStream_t stream);
#define CP_VARIANT_FUSED_HEAD 22
// direct low-channel convolutions of the network's first three layers in f16x3 mode (lowc.hip).
// kind: 0 stem 7x7 (NCHW input, `planes` <= 4) -> 16; 1 level0 3x3 16->16; 2 level1 3x3 stride 2 16->32
size_t cp_lowc_weight_halfs(int kind);
// fwd: per-output-channel power-of-two weight pre-scale (cp_launch_weight_scale) or nullptr; `scale` of cp_launch_lowc
// must then carry the inverse.  in_amax / out_amax: see ConvParams.
int cp_launch_pack_lowc(int kind, const float* w, void* hi, void* lo, const float* fwd, int cin, hipStream_t s);
int cp_launch_lowc(int kind, const float* in, float* out, const void* w_hi, const void* w_lo, const float* scale,
                   const float* shift, const unsigned* in_amax, unsigned* out_amax, int B, int H, int W, int planes,
                   hipStream_t s);
#define CP_VARIANT_LOWC0 23
// stem + level0 in one launch (lowc.hip: lowc2_kernel): the 16-channel full-resolution tensor between them is never stored
#define CP_VARIANT_LOWC01 40
int cp_launch_lowc_fused(const float* in, float* out, const void* w0_hi, const void* w0_lo, const float* scale0, const float* shift0,
                         const void* w1_hi, const void* w1_lo, const float* scale1, const float* shift1, float bound_l, float bound_s,
                         const unsigned* in_amax, unsigned* out_amax, int B, int H, int W, int planes, hipStream_t s);
#define CP_PREC_F32 0
#define CP_PREC_F16X3 1

// ---- element-wise / data-movement kernels (ewise.hip) ----
int cp_launch_nchw_to_nhwc(const float* in, float* out, int B, int C, int H, int W, int Cpad, hipStream_t s);
int cp_launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int H, int W, int ldi, hipStream_t s);
int cp_launch_maxpool2(const float* in, float* out, int B, int H, int W, int C, hipStream_t s);
// the element-wise producers below take an optional |max| slot for their output (ConvParams::out_amax)
// depth-wise ConvTranspose2d(k=2f, stride=f, pad=f/2) of `in` [B,H,W,C] plus `add` [B,fH,fW,C] -> out
int cp_launch_upsample_add(const float* in, const float* w, const float* add, float* out, int B, int H, int W,
                           int C, int f, unsigned* out_amax, hipStream_t s);
int cp_launch_add_relu_sum(const float* a, const float* b, const float* c, const float* d, float* out, size_t n,
                           unsigned* out_amax, hipStream_t s);
// hourglass merge: out[B,2H,2W,C] = up1 + nearest-x2(low[B,H,W,C])  (large_hourglass.py:186-188)
int cp_launch_upsample2_nearest_add(const float* up1, const float* low, float* out, int B, int H, int W, int C,
                                    unsigned* out_amax, hipStream_t s);
// ConvGRU gates (convGRU.py:32-39).  x3/h3: [M,192] = (r,z,n) pre-activations, h: [M,64]
int cp_launch_gru_gate(const float* x3, const float* h3, const float* hprev, float* hout, size_t M, unsigned* out_amax,
                       hipStream_t s);
// GroupNorm(32 groups) over NHWC [B, HW, C]: stats then in-place normalise + affine + ReLU
int cp_launch_groupnorm_relu(float* x, const float* gamma, const float* beta, double* stats_ws, int B, int HW, int C,
                             int groups, float eps, unsigned* out_amax, hipStream_t s);
// (sum, sumsq) doubles -> (mean, rstd) floats per (image, group)
// (sum, sum of squares) per (image, group) -> per (image, channel) affine a = rstd*gamma, d = beta - mean*rstd*gamma
// x_amax / y_amax (optional): |max| slot of the un-normalised tensor, and the slot that receives the bound
// max over (image, channel) of |a| * max|x| + |d| >= max|relu(a*x + d)| for the consumer's activation pre-scale
int cp_launch_gn_final(const float* x, const float* ga, const float* gd, const float* wp, const float* bias, float* out,
                       int B, int HW, int C, int N, int wld, int sigmoid, hipStream_t s);
int cp_launch_gn_affine(const double* stats, const float* gamma, const float* beta, float* a, float* d, int B, int C,
                        int groups, double count, float eps, const unsigned* x_amax, unsigned* y_amax, hipStream_t s);
int cp_launch_gn_finalize(const double* stats, float* mr, int n, double count, float eps, hipStream_t s);
// PyTorch [Cout][Cin][taps] weights -> packed GEMM operand (buffer must be pre-zeroed for padding)
int cp_launch_pack_weight(const float* w, float* wp, int Cout, int Cin, int taps, int CinP, int CoutPad, int coff,
                          hipStream_t s);

// ---- decode (decode.hip) ----
// detection record: 118 float32 per detection, J = 8 joints (field order of decode.py:347-361)
#define CP_DET_BBOX 0
#define CP_DET_SCORE 4
#define CP_DET_KPS 5
#define CP_DET_CLS 21
#define CP_DET_SCALE 22
#define CP_DET_SCALE_UNC 25
#define CP_DET_TRACKING 28
#define CP_DET_TRACKING_HP 30
#define CP_DET_KPS_DISP_MEAN 46
#define CP_DET_KPS_DISP_STD 62
#define CP_DET_KPS_HM_MEAN 78
#define CP_DET_KPS_HM_STD 94
#define CP_DET_KPS_HM_HEIGHT 110
#define CP_DET_STRIDE 118
size_t cp_decode_ws_bytes(int B, int J, int K);
int cp_launch_decode(hipStream_t s, int B, int J, int H, int W, float* hm, const float* hps, const float* wh,
                     const float* hps_unc, const float* scale, const float* scale_unc, const float* reg, float* hm_hp,
                     const float* hp_offset, const float* tracking, const float* tracking_hp, int K, int rep_mode,
                     int fit_gaussian, float balance, int legacy_bool_mask, int apply_sigmoid, float* det, void* ws);

// ---- generic DCNv2 forward on the reference's NCHW layouts (dcn_generic.hip): any C / kernel / stride / dilation / dg ----
int cp_launch_dcn_generic(hipStream_t s, const float* x, const float* w, const float* bias, const float* offset,
                          const float* mask, float* out, int B, int C, int H, int W, int Co, int Ho, int Wo, int kh, int kw,
                          int sh, int sw, int ph, int pw, int dh, int dw, int dg);

// ---- batched PnP (pnp.hip) ----
#define CP_PNP_STRIDE 40
size_t cp_pnp_ws_bytes(int N);
int cp_launch_pnp(hipStream_t s, const float* pts, const float* scale, const double* cam, int N, int npts, double* out,
                  void* ws);

int cp_launch_preprocess(const unsigned char* img, int B, int H, int W, const double* trans6, const float* mean3,
                         const float* std3, float* out, int OH, int OW, hipStream_t s);
int cp_launch_resize_u8(const unsigned char* img, int H, int W, int C, unsigned char* out, int OH, int OW, hipStream_t s);

// device post-process + soft-NMS (post.hip); record layout CP_POST_* in include/centerpose_hip.h
int cp_launch_postprocess(const float* det, int B, int K, const double* meta, double vis_thresh, int nms,
                          float div_scale, double* out, int* count, double* ws, hipStream_t s);
int cp_launch_pnp_assemble(const double* post, const int* count, int B, int K, int npts, const double* cam_img, float* pts,
                           float* scale, double* cam, hipStream_t s);
int cp_launch_render_gaussians(const double* recs, int N, float* out, int C, int H, int W, hipStream_t s);

// ---- CenterPoseTrack bookkeeping on the device (track.hip; TrackParams: track_common.h) ----
struct TrackParams;
size_t cp_track_state_bytes_impl(int B, int cap);
size_t cp_track_ws_bytes_impl(int B, int K, int cap);
int cp_launch_track_step(hipStream_t s, const TrackParams& P, const double* vmeta, const double* post, const int* count,
                         const double* det_pnp, int B, int K, void* state, double* recs, void* ws);
