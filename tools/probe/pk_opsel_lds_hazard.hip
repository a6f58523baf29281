// Reproducer (gfx950 / MI355X, ROCm 7.2): a packed-f32 VALU op whose LOW result takes the HIGH word of a source (a set op_sel bit,
// e.g.  v_pk_add_f32 v[10:11], v[14:15], v[10:11] op_sel:[0,1] op_sel_hi:[1,0] ) returns a wrong low result in lanes 48-63 when
// another wave of the same SIMD issues MFMAs and LDS reads; DS writes of the own wave shortly before raise the rate by four
// orders of magnitude (their EXEC mask does not matter), and neither s_waitcnt lgkmcnt(0) nor 64 wait states in between cure it.
// The broadcast forms (op_sel_hi only: high result <- low word), the un-swizzled op and v_pk_fma_f32 op_sel_hi:[0,1,1] are clean.
// Root cause of the wrong bilinear set-up values in dcn16p's early-prologue build (profiles/NOTES.md, round 5): there the SLP
// vectorizer folds {w_im, h_im} = {wb, hb} + {dx, dy} into exactly that op, and the previous tap's exception filing
// (ds_write under a sparse EXEC) sits in front of it.  Output of a run: profiles/r05_pk_opsel_lds_hazard.txt.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/probe/pk_opsel_lds_hazard.hip -o /tmp/hz && /tmp/hz
// Workgroup = 8 waves: waves 0-3 (first wave of each SIMD) run [PRE; FORM; compare with scalar adds], waves 4-7 (second wave of
// each SIMD) loop over 8 ds_read_b128 + 6 MFMAs (NB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { if ((x) != hipSuccess) { printf("hip error: %s\n", #x); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// FORM 0: op_sel:[0,1] (low <- high word of src1)  1: op_sel_hi:[1,0] (high <- low word)  2: both (the kernel's)  3: no swizzle
//      4: v_pk_fma_f32 op_sel_hi:[0,1,1] (dcn16.hip's broadcast FMA)
// PRE bit 0: EXEC narrowed (lanes 48-63 off) and restored around bit 1: 14 ds_write_b128; bit 2: s_waitcnt lgkmcnt(0) behind them;
//     bit 3: 64 wait states behind them.      NB bit 0: the neighbour wave issues MFMAs, bit 1: LDS reads
template <int FORM, int PRE, int NB>
__global__ __launch_bounds__(512) void k(unsigned* bad, float* keep, int iters) {
    __shared__ float4 lds[14 * 256];
    const int tid = threadIdx.x & 255, lane = tid & 63;
    const unsigned la = (unsigned)(size_t)&lds[tid];
    if (threadIdx.x >= 256) {  // the neighbour (second wave of each SIMD)
        h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {2, 1, 2, 1, 2, 1, 2, 1};
        f32x16 c0 = {}, c1 = {};
        f32x4 g[8] = {};
        for (int it = 0; it < (NB ? iters * 3 : 0); ++it) {
            asm volatile(".if %13 & 2\n\t.set o, 0\n\t.irp r,%2,%3,%4,%5,%6,%7,%8,%9\n\tds_read_b128 \\r, %12 offset:o\n\t.set o, o + 4096\n\t.endr\n\t.endif\n\t"
                         ".if %13 & 1\n\t.rept 3\n\tv_mfma_f32_32x32x16_f16 %0, %10, %11, %0\n\tv_mfma_f32_32x32x16_f16 %1, %10, %11, %1\n\t.endr\n\t.endif\n\t"
                         "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7"
                         : "+v"(c0), "+v"(c1), "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7])
                         : "v"(a), "v"(b), "v"(la), "n"(NB) : "memory");
            c0[0] += g[0].x + g[7].w;
        }
        if (c0[0] + c1[3] == 1.2345e-30f) keep[0] = c0[0];
        return;
    }
    f32x2 s0 = {(float)lane, 100.f + (float)lane}, s1 = {0.25f * (float)tid, -3.f - 0.5f * (float)tid};
    const f32x4 junk = {1.f, 2.f, 3.f, 4.f};
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 d = s1;
        asm volatile(".if %5 & 1\n\ts_mov_b64 s[44:45], exec\n\ts_mov_b32 exec_hi, 0xffff\n\t.endif\n\t"
                     ".if %5 & 2\n\t.set o, 0\n\t.rept 14\n\tds_write_b128 %2, %3 offset:o\n\t.set o, o + 4096\n\t.endr\n\t.endif\n\t"
                     ".if %5 & 1\n\ts_mov_b64 exec, s[44:45]\n\t.endif\n\t"
                     ".if %5 & 4\n\ts_waitcnt lgkmcnt(0)\n\t.endif\n\t.if %5 & 8\n\t.rept 4\n\ts_nop 15\n\t.endr\n\t.endif\n\t"
                     ".if %4 == 0\n\tv_pk_add_f32 %0, %1, %0 op_sel:[0,1]\n\t.endif\n\t"
                     ".if %4 == 1\n\tv_pk_add_f32 %0, %1, %0 op_sel_hi:[1,0]\n\t.endif\n\t"
                     ".if %4 == 2\n\tv_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]\n\t.endif\n\t"
                     ".if %4 == 3\n\tv_pk_add_f32 %0, %1, %0\n\t.endif\n\t"
                     ".if %4 == 4\n\tv_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]\n\t.endif\n\t"
                     "s_nop 4\n\ts_waitcnt lgkmcnt(0)"
                     : "+v"(d) : "v"(s0), "v"(la), "v"(junk), "n"(FORM), "n"(PRE) : "s44", "s45", "memory");
        const f32x2 e = FORM == 0 ? f32x2{s0.x + s1.y, s0.y + s1.y} : FORM == 1 ? f32x2{s0.x + s1.x, s0.y + s1.x}
                      : FORM == 2 ? f32x2{s0.x + s1.y, s0.y + s1.x} : FORM == 3 ? f32x2{s0.x + s1.x, s0.y + s1.y}
                                  : f32x2{__builtin_fmaf(s0.x, s1.x, s1.x), __builtin_fmaf(s0.x, s1.y, s1.y)};
        nbad += (d.x != e.x ? 1 : 0) + (d.y != e.y ? 0x10000 : 0);
        s1.x += 0.125f; s1.y -= 0.25f;
    }
    if (nbad) { atomicAdd(&bad[lane], nbad & 0xffff); atomicAdd(&bad[64 + lane], nbad >> 16); }
}

static unsigned* bad;
static float* keep;
static int iters = 20000;
template <int FORM, int PRE, int NB>
static void run(const char* what) {
    CK(hipMemset(bad, 0, 128 * sizeof(unsigned)));
    hipLaunchKernelGGL((k<FORM, PRE, NB>), dim3(512), dim3(512), 0, 0, bad, keep, iters);
    CK(hipDeviceSynchronize());
    unsigned h[128];
    CK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
    unsigned long long q[8] = {};
    for (int i = 0; i < 128; ++i) q[i / 16] += h[i];
    printf("%-44s | in front: EXEC rewritten %d, DS writes %d, lgkmcnt(0) %d, 64 states %d | neighbour: MFMA %d LDS %d | of %.2g: wrong LOW by "
           "lane quarter [%llu %llu %llu %llu] HIGH [%llu %llu %llu %llu]\n", what, PRE & 1, (PRE >> 1) & 1, (PRE >> 2) & 1, (PRE >> 3) & 1, NB & 1,
           (NB >> 1) & 1, 512.0 * 256 * iters, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]);
}

int main(int argc, char** argv) {
    if (argc > 1) iters = atoi(argv[1]);
    CK(hipMalloc(&bad, 128 * sizeof(unsigned)));
    CK(hipMalloc(&keep, 16));
    const char* kf = "op_sel:[0,1] op_sel_hi:[1,0] (the kernel's)";
    run<0, 3, 3>("op_sel:[0,1] (low <- high word)");            // which form
    run<1, 3, 3>("op_sel_hi:[1,0] (high <- low word)");
    run<2, 3, 3>(kf);
    run<3, 3, 3>("no swizzle");
    run<4, 3, 3>("v_pk_fma_f32 op_sel_hi:[0,1,1] (dcn16.hip)");
    run<2, 0, 3>(kf); run<2, 1, 3>(kf); run<2, 2, 3>(kf);         // what has to come in front
    run<2, 3 | 4, 3>(kf); run<2, 3 | 8, 3>(kf); run<2, 3 | 4 | 8, 3>(kf);
    run<2, 3, 0>(kf); run<2, 3, 1>(kf); run<2, 3, 2>(kf);         // what the neighbour has to do
    return 0;
}
