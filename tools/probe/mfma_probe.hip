// Calibration probe (tuning aid): what rate does v_mfma_f32_32x32x16_f16 sustain on this part when it is fed the way
// igemm16p feeds it?  Each variant adds one ingredient of the real loop to a bare MFMA loop.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_clk[2];

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    __shared__ __attribute__((aligned(16))) _Float16 lds[32768];  // 64 KB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32768; i += 256) lds[i] = (_Float16)(float)(i & 7);
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    h8 fa[4], fb[4];
    for (int q = 0; q < 4; ++q) {
        fa[q] = *reinterpret_cast<const h8*>(lds + (q * 64 + lane) * 8);
        fb[q] = *reinterpret_cast<const h8*>(lds + 4096 + (q * 64 + lane) * 8);
    }
    const u32x4* gin = reinterpret_cast<const u32x4*>(in) + (size_t)blockIdx.x * 4096 + tid;
    u32x4 g[8];
    for (int q = 0; q < 8; ++q) g[q] = u32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const int base = (it & 1) * 16384;
        if (MODE >= 3) {
#pragma unroll
            for (int q = 0; q < 8; ++q) g[q] = gin[q * 256 + (it & 1) * 2048];
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (MODE >= 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    fa[q] = *reinterpret_cast<const h8*>(lds + base + ((ks * 8 + q) * 64 + lane) * 8);
                    fb[q] = *reinterpret_cast<const h8*>(lds + base + ((ks * 8 + 4 + q) * 64 + lane) * 8);
                }
            }
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[(a + term) & 3], fb[(a * 2 + term) & 3], acc[a], 0, 0, 0);
        }
        if (MODE >= 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                u32x4 v = g[q];
                v.x += it;  // keep the stores alive
                *reinterpret_cast<u32x4*>(lds + (base ^ 16384) + (q * 256 + tid) * 8) = v;
            }
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) {  // shader clock vs the constant 100 MHz counter -> actual frequency of this CU
        g_clk[0] = clock64() - c0;
        g_clk[1] = wall_clock64() - w0;
    }
}

template <int MODE>
double run(const float* in, float* out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof(h));
    printf("[shader clock %4.0f MHz] ", (double)h[0] / ((double)h[1] / 100.0));
    const double flops = 5.0 * blocks * 4.0 * iters * 24.0 * 32768.0;  // 4 waves x 24 MFMAs per iteration
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    const int blocks = 512 * 8, iters = 512;
    float *in, *out;
    hipMalloc(&in, (size_t)blocks * 4096 * 16 + (1 << 20));
    hipMemset(in, 0, (size_t)blocks * 4096 * 16 + (1 << 20));
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    double r;
    r = run<0>(in, out, blocks, iters); printf("bare MFMA loop                          : %7.1f TFLOP/s\n", r);
    r = run<1>(in, out, blocks, iters); printf("+ 16 ds_read_b128 per 24 MFMA           : %7.1f TFLOP/s\n", r);
    r = run<2>(in, out, blocks, iters); printf("+ 8 ds_write_b128 + barrier             : %7.1f TFLOP/s\n", r);
    r = run<3>(in, out, blocks, iters); printf("+ 8 global_load_dwordx4 per thread      : %7.1f TFLOP/s\n", r);
    return 0;
}
