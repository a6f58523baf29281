// Calibration probe: the K-loop skeleton of igemm16p (two phases of 12 MFMAs, fragment reads of the other k-step in
// their shadow, one barrier per tile) without any global load / conversion / LDS store, bisected by template flags.
//   PIN  : sched_barrier(0) after every MFMA slot (the hand-pinned order of the real kernel)
//   BAR  : one __syncthreads() per tile
//   READS: ds_read_b128 fragment reads (else the fragments stay constant)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool PIN, bool BAR, bool READS, int OCC>
__global__ __launch_bounds__(256, OCC) void skel(float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[32768];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 32768; i += 256) lds[i] = (_Float16)(float)(i & 3);
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    h8 f0[8], f1[8];  // al0 al1 bh0 bh1 ah0 ah1 bl0 bl1
    for (int q = 0; q < 8; ++q) {
        f0[q] = *reinterpret_cast<const h8*>(lds + (q * 64 + lane) * 8);
        f1[q] = *reinterpret_cast<const h8*>(lds + 4096 + (q * 64 + lane) * 8);
    }
    auto slot = [&](int s, const h8(&f)[8]) {
        const int term = s / 4, idx = s % 4, i = idx / 2, j = idx % 2;
        if (term == 0) acc[idx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], f[2 + j], acc[idx], 0, 0, 0);
        else if (term == 1) acc[idx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[4 + i], f[6 + j], acc[idx], 0, 0, 0);
        else acc[idx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[4 + i], f[2 + j], acc[idx], 0, 0, 0);
    };
    for (int it = 0; it < iters; ++it) {
        const int base = (it & 1) * 16384 + wid * 1024;
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            slot(s, f0);
            if (READS && s < 8) f1[s] = *reinterpret_cast<const h8*>(lds + base + 8192 + (s * 64 + lane) * 8);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __syncthreads();
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            slot(s, f1);
            if (READS && s < 8) f0[s] = *reinterpret_cast<const h8*>(lds + (base ^ 16384) + (s * 64 + lane) * 8);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) sum += acc[a][r];
    out[(size_t)blockIdx.x * 256 + tid] = sum;
}

template <bool PIN, bool BAR, bool READS, int OCC>
void run(float* out, const char* what) {
    const int blocks = 256 * OCC * 8, iters = 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((skel<PIN, BAR, READS, OCC>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((skel<PIN, BAR, READS, OCC>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s occ %d : %7.1f TFLOP/s issued\n", what, OCC, 3.0 * blocks * 4.0 * iters * 24.0 * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    hipMalloc(&out, (size_t)8192 * 256 * 4);
    run<false, false, false, 2>(out, "MFMA only");
    run<false, false, true, 2>(out, "MFMA + reads (compiler order)");
    run<true, false, true, 2>(out, "MFMA + reads, order pinned per slot");
    run<false, true, true, 2>(out, "MFMA + reads + barrier per tile (compiler order)");
    run<true, true, true, 2>(out, "MFMA + reads + barrier per tile, order pinned");
    run<true, true, true, 1>(out, "MFMA + reads + barrier per tile, order pinned");
    run<true, true, false, 2>(out, "MFMA + barrier per tile, pinned, no reads");
    return 0;
}
