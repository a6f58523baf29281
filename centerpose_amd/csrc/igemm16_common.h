// Pieces shared by the split-f16 ("f16x3") kernels: igemm16.hip (plain / fused convolutions) and dcn16.hip (DCNv2).
#pragma once
#include "igemm_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK16 = 32;
constexpr int LDH = 32;  // halfs per LDS row: 64 bytes, no padding; 16-byte chunks are XOR-swizzled by the row
// chunk c of row r lives at chunk position c ^ ((r >> 2) & 3): every ds_read_b128 lane group ({0-3,12-15,20-27}, ...)
// then hits 16 distinct 16-byte slots of the 256-byte bank row, and two consecutive rows written by a ds_write_b64 /
// b128 lane group cover 32 distinct banks (measured before the swizzle: 33 % of LDS cycles were bank conflicts).
__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }
constexpr int NT16 = 256;

// Raw buffer loads (SRD + 32-bit byte offset): an out-of-range offset returns 0, so halo / invalid taps need no
// exec-mask branch around the load -- and without control flow between the loads the compiler can wait for tile
// t+1 with a counted s_waitcnt vmcnt(N) while tile t+2 stays in flight.
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
constexpr unsigned OOB = 0xffffffffu;
constexpr unsigned OOB_BASE = 0xf0000000u;  // + any channel offset (< 64 KiB) is still beyond every supported tensor

struct Split2 {
    uint32_t hi, lo;  // two binary16 values each
};

__device__ __forceinline__ uint32_t pk(float a, float b) {
    fp16x2 v = __builtin_amdgcn_cvt_pkrtz(a, b);
    return *reinterpret_cast<uint32_t*>(&v);
}

// (a, b) -> packed hi halves and packed lo halves.  hi = RTZ(a), RTZ(b) (one v_cvt_pkrtz); the residuals a - float(hi) are
// formed AND rounded to binary16 by v_fma_mixlo_f16 / v_fma_mixhi_f16, which read the binary16 half of the packed hi register
// directly (a * 1.0 - hi: exact in float32, then one rounding to nearest even) and write the low / high half of the lo
// register: 3 VALU per pair.  (Rounds 1-4 used v_fma_mix_f32 x 2 + a second v_cvt_pkrtz: 4 VALU per pair, lo rounded towards
// zero; CP_SPLIT4 keeps that form for A/B builds.)  The split is most of the loaders' and of the DCN blend's VALU work.
__device__ __forceinline__ Split2 split2(float a, float b) {
    Split2 s;
    fp16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
    s.hi = *reinterpret_cast<uint32_t*>(&h);
#ifdef CP_SPLIT4
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "v"(s.hi));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "v"(s.hi));
    s.lo = pk(ra, rb);
#else
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(s.lo) : "v"(a), "v"(s.hi));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(s.lo) : "v"(b), "v"(s.hi));
#endif
    return s;
}

}  // namespace
