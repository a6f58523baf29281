// Calibration probe: L2-resident streaming bandwidth vs bytes in flight per CU (loads per lane x waves per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NL, int OCC>
__global__ __launch_bounds__(256, OCC) void stream(const u32x4* __restrict__ in, uint32_t* __restrict__ out, size_t stride16,
                                                   size_t span16, size_t total16, int iters) {
    const u32x4* p = in + ((size_t)blockIdx.x * stride16) % (total16 - span16 + 1) + threadIdx.x;
    uint32_t acc = 0;
    const int n = (int)(span16 / (256 * NL));
    for (int it = 0; it < iters; ++it)
        for (int o = 0; o < n; ++o) {
            u32x4 v[NL];
#pragma unroll
            for (int q = 0; q < NL; ++q) v[q] = __builtin_nontemporal_load(p + (size_t)o * 256 * NL + q * 256);
#pragma unroll
            for (int q = 0; q < NL; ++q) acc += v[q].x ^ v[q].w;
        }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int NL, int OCC>
void run(const u32x4* in, uint32_t* out, const char* what, size_t stride, size_t span, size_t total, int iters) {
    const int blocks = 256 * OCC * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((stream<NL, OCC>), dim3(blocks), dim3(256), 0, 0, in, out, stride / 16, span / 16, total / 16, iters);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r)
        hipLaunchKernelGGL((stream<NL, OCC>), dim3(blocks), dim3(256), 0, 0, in, out, stride / 16, span / 16, total / 16, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %2d loads/lane x %2d waves/CU = %4d KB in flight/CU : %6.2f TB/s\n", what, NL, OCC * 4, NL * OCC * 4,
           3.0 * blocks * (double)span * iters / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t total = (size_t)1 << 30;
    u32x4* in;
    uint32_t* out;
    hipMalloc(&in, total);
    hipMemset(in, 1, total);
    hipMalloc(&out, (size_t)65536 * 256 * 4);
    // all workgroups stream a shared 16 MB window (L2 + infinity cache resident), staggered starts
    run<4, 2>(in, out, "16 MB shared window", 64 << 10, 1 << 20, 16 << 20, 4);
    run<8, 2>(in, out, "16 MB shared window", 64 << 10, 1 << 20, 16 << 20, 4);
    run<16, 2>(in, out, "16 MB shared window", 64 << 10, 1 << 20, 16 << 20, 4);
    run<8, 4>(in, out, "16 MB shared window", 64 << 10, 1 << 20, 16 << 20, 4);
    run<16, 4>(in, out, "16 MB shared window", 64 << 10, 1 << 20, 16 << 20, 4);
    run<8, 8>(in, out, "16 MB shared window", 64 << 10, 1 << 20, 16 << 20, 4);
    run<8, 2>(in, out, "2 MB shared window", 64 << 10, 1 << 20, 2 << 20, 4);
    run<16, 4>(in, out, "2 MB shared window", 64 << 10, 1 << 20, 2 << 20, 4);
    return 0;
}
