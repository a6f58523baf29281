// Hazard probe (gfx950): is a load (LDS / VMEM) that targets the SrcA / SrcB registers of a just-issued v_mfma_f32_32x32x16_f16
// allowed to land while that MFMA (queued behind earlier ones of the same wave) has not read its operands yet?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_war_probe.hip -o /tmp/mfma_war_probe && /tmp/mfma_war_probe
// Each wave runs: 6 MFMAs (two accumulator chains, as the K step of dcn16p / halo16) -> NOPS wait states -> ds_read_b128 (or
// buffer_load_dwordx4) INTO the A (or B) operand registers of the last MFMAs.  The loaded data differs from the operand, so a
// too-early landing changes the accumulators; rows 16..31 of the 32 x 32 tile are reported separately.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NOPS, int MODE>  // MODE 0: ds_read into A, 1: ds_read into B of the last MFMA, 2: global load into A, 3: nothing (control / reference)
__global__ __launch_bounds__(256, 1) void k(const u32x4* ga, const u32x4* gb, const u32x4* gjunk, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) u32x4 junk[256];
    const int lane = threadIdx.x & 63;
    junk[threadIdx.x] = gjunk[threadIdx.x];
    __syncthreads();
    u32x4 a = ga[lane], b0 = gb[lane], b1 = gb[64 + lane];
    const u32x4 a_keep = a, b1_keep = b1;
    f32x16 c0 = {}, c1 = {};
    const unsigned ldsaddr = (unsigned)(size_t)(&junk[threadIdx.x]);
    const u32x4* gp = gjunk + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)
            asm volatile(
                "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %1\n\t"
                "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %1\n\t"
                "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %1\n\t"
                ".rept %6\n\ts_nop 0\n\t.endr\n\t"
                "ds_read_b128 %2, %5\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7"
                : "+v"(c0), "+v"(c1), "+v"(a) : "v"(b0), "v"(b1), "v"(ldsaddr), "n"(NOPS) : "memory");
        else if (MODE == 1)
            asm volatile(
                "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %1\n\t"
                "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %1\n\t"
                "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %1\n\t"
                ".rept %6\n\ts_nop 0\n\t.endr\n\t"
                "ds_read_b128 %4, %5\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7"
                : "+v"(c0), "+v"(c1), "+v"(a), "+v"(b0), "+v"(b1) : "v"(ldsaddr), "n"(NOPS) : "memory");
        else if (MODE == 2)
            asm volatile(
                "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %1\n\t"
                "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %1\n\t"
                "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %1\n\t"
                ".rept %6\n\ts_nop 0\n\t.endr\n\t"
                "global_load_dwordx4 %2, %5, off\n\ts_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7"
                : "+v"(c0), "+v"(c1), "+v"(a) : "v"(b0), "v"(b1), "v"(gp), "n"(NOPS) : "memory");
        else
            asm volatile(
                "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %1\n\t"
                "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %1\n\t"
                "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, %1\n\t"
                ".rept %6\n\ts_nop 0\n\t.endr\n\t"
                "s_nop 7\n\ts_nop 7"
                : "+v"(c0), "+v"(c1), "+v"(a) : "v"(b0), "v"(b1), "v"(ldsaddr), "n"(NOPS) : "memory");
        a = a_keep;   // restore the operands for the next round
        b1 = b1_keep;
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b1));
    }
    float* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 32;
    for (int r = 0; r < 16; ++r) { o[r] = c0[r]; o[16 + r] = c1[r]; }
}

template <int NOPS, int MODE>
int run(const char* what, const u32x4* ga, const u32x4* gb, const u32x4* gj, float* dout, const std::vector<float>& ref, int blocks) {
    const int iters = 64;
    hipLaunchKernelGGL((k<NOPS, MODE>), dim3(blocks), dim3(256), 0, 0, ga, gb, gj, dout, iters);
    std::vector<float> o((size_t)blocks * 256 * 32);
    (void)hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    long bad_lo = 0, bad_hi = 0;
    for (size_t t = 0; t < (size_t)blocks * 256; ++t)
        for (int r = 0; r < 32; ++r)
            if (o[t * 32 + r] != ref[(t & 63) * 32 + r]) { if ((r & 15) >= 8) ++bad_hi; else ++bad_lo; }
    printf("%-28s nops %2d: wrong accumulators rows 0-15: %8ld  rows 16-31: %8ld\n", what, NOPS, bad_lo, bad_hi);
    return bad_lo + bad_hi != 0;
}

int main() {
    std::vector<uint16_t> ha(64 * 8), hb(128 * 8), hj(256 * 8);
    auto f16 = [](float v) { _Float16 h = (_Float16)v; uint16_t u; std::memcpy(&u, &h, 2); return u; };
    srand(3);
    for (auto& v : ha) v = f16((float)(rand() % 17 - 8) / 8.f);
    for (auto& v : hb) v = f16((float)(rand() % 17 - 8) / 8.f);
    for (auto& v : hj) v = f16((float)(rand() % 17 - 8) * 4.f);
    u32x4 *ga, *gb, *gj; float* dout;
    const int blocks = 512;
    (void)hipMalloc(&ga, 64 * 16); (void)hipMalloc(&gb, 128 * 16); (void)hipMalloc(&gj, 256 * 16); (void)hipMalloc(&dout, (size_t)blocks * 256 * 32 * 4);
    (void)hipMemcpy(ga, ha.data(), 64 * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(gb, hb.data(), 128 * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(gj, hj.data(), 256 * 16, hipMemcpyHostToDevice);
    // reference: the same kernel with a long gap (64 wait states) -- values are small integers / 64: exact in f32
    hipLaunchKernelGGL((k<64, 3>), dim3(1), dim3(64), 0, 0, ga, gb, gj, dout, 64);
    std::vector<float> ref(64 * 32);
    (void)hipMemcpy(ref.data(), dout, ref.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
#define R(N, M, W) bad += run<N, M>(W, ga, gb, gj, dout, ref, blocks)
    R(0, 0, "ds_read -> A of MFMAs");  R(2, 0, "ds_read -> A of MFMAs");  R(4, 0, "ds_read -> A of MFMAs");  R(8, 0, "ds_read -> A of MFMAs");
    R(16, 0, "ds_read -> A of MFMAs"); R(32, 0, "ds_read -> A of MFMAs");
    R(0, 1, "ds_read -> B of last MFMA"); R(4, 1, "ds_read -> B of last MFMA"); R(8, 1, "ds_read -> B of last MFMA"); R(16, 1, "ds_read -> B of last MFMA");
    R(0, 2, "global_load -> A");       R(8, 2, "global_load -> A");       R(16, 2, "global_load -> A");
    R(0, 3, "no overwrite (control)");
    printf(bad ? "HAZARD OBSERVED\n" : "no hazard observed\n");
    return 0;
}
