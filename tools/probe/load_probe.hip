// Calibration probe (tuning aid): sustained bandwidth of 16-byte-per-lane loads as a function of the footprint
// (L1 / L2 / infinity cache / shared by all workgroups), with 8 waves per CU like the implicit-GEMM kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/load_probe.hip -o /tmp/load_probe && /tmp/load_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// every workgroup streams `span` bytes starting at (blockIdx * stride) % total, `iters` times
__global__ __launch_bounds__(256, 2) void stream(const u32x4* __restrict__ in, uint32_t* __restrict__ out, size_t stride16,
                                                 size_t span16, size_t total16, int iters) {
    // (the host picks stride / span so that base + span <= total: no wrap-around arithmetic in the loop)
    const u32x4* p = in + ((size_t)blockIdx.x * stride16) % (total16 - span16 + 1) + threadIdx.x;
    uint32_t acc = 0;
    const int n = (int)(span16 / (256 * 8));
    for (int it = 0; it < iters; ++it)
        for (int o = 0; o < n; ++o) {
            u32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = p[(size_t)o * 2048 + q * 256];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q].x ^ v[q].w;
        }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

double run(const u32x4* in, uint32_t* out, int blocks, size_t stride, size_t span, size_t total, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(stream, dim3(blocks), dim3(256), 0, 0, in, out, stride / 16, span / 16, total / 16, iters);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r)
        hipLaunchKernelGGL(stream, dim3(blocks), dim3(256), 0, 0, in, out, stride / 16, span / 16, total / 16, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return 3.0 * blocks * (double)span * iters / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t total = (size_t)1 << 30;
    u32x4* in;
    uint32_t* out;
    hipMalloc(&in, total);
    hipMemset(in, 1, total);
    hipMalloc(&out, (size_t)8192 * 256 * 4);
    const int blocks = 512 * 4;
    printf("per-WG private 16 KB  (L1-resident)            : %6.2f TB/s\n", run(in, out, blocks, 16 << 10, 16 << 10, total, 256));
    printf("per-WG private 64 KB  (> L1, all WGs = 32 MB)  : %6.2f TB/s\n", run(in, out, blocks, 64 << 10, 64 << 10, total, 64));
    printf("all WGs share 16 MB, staggered starts (L2)      : %6.2f TB/s\n", run(in, out, blocks, 64 << 10, 1 << 20, 16 << 20, 4));
    printf("all WGs share the same 288 KB (weights-like)   : %6.2f TB/s\n", run(in, out, blocks, 0, 288 << 10, total, 16));
    printf("per-WG private 1 MB of 1 GB (HBM/MALL stream)  : %6.2f TB/s\n", run(in, out, blocks, 1 << 20, 1 << 20, total, 2));
    return 0;
}
