// LDS-DMA semantics probe for dcn16s.hip (tuning / bring-up aid): buffer_load_dwordx4 ... lds
//   hipcc --offload-arch=gfx950 -O3 tools/probe/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
// Checks: (0) lane-linear destination M0 + 16 lane; (1) out-of-range lanes write ZEROS; (2) EXEC-masked lanes write nothing,
// a 16-byte-aligned (not 1 KB-aligned) base works; (3) soffset is added to the source; (4) destinations above 64 KB work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned lds_base) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_base), "v"(voff), "s"(r), "s"(soff) : "memory", "m0");
}
constexpr int NF = 30 * 1024;  // floats: 120 KB
__global__ void k(const float* in, float* out, unsigned nbytes, int mode) {
    __shared__ __attribute__((aligned(16))) float lds[NF];
    for (int i = threadIdx.x; i < NF; i += blockDim.x) lds[i] = -7.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = make_rsrc(in, nbytes);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned voff = (unsigned)(lane * 16 + w * 1024);
    if (mode == 1 && (lane & 3) == 1) voff = 0xffffffffu;
    unsigned base = (unsigned)(size_t)(lds) + (unsigned)w * 2048u;
    if (mode == 4) base += 100u * 1024u;
    base = __builtin_amdgcn_readfirstlane(base);
    if (mode == 2) { if (lane < 48) dma16(r, voff, 0, base + 16); }
    else dma16(r, voff, mode == 3 ? 64 : 0, base);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < NF; i += blockDim.x) out[i] = lds[i];
    if (threadIdx.x == 0) out[NF] = (float)((unsigned)(size_t)(lds));
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)(i + 1);
    float *din, *dout;
    (void)hipMalloc(&din, n * 4); (void)hipMalloc(&dout, (NF + 1) * 4);
    (void)hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int mode = 0; mode < 5; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, din, dout, (unsigned)(n * 4), mode);
        std::vector<float> o(NF + 1);
        (void)hipMemcpy(o.data(), dout, (NF + 1) * 4, hipMemcpyDeviceToHost);
        int err = 0, touched = 0;
        for (int i = 0; i < NF; ++i) {
            float want = -7.f;
            for (int w = 0; w < 2; ++w) {
                const int b0 = w * 512 + (mode == 2 ? 4 : 0) + (mode == 4 ? 25600 : 0);  // float index of the wave's base
                const int j = i - b0;
                const int nl = mode == 2 ? 48 : 64;
                if (j >= 0 && j < nl * 4) {
                    const int l = j / 4;
                    want = (float)(w * 256 + j + 1 + (mode == 3 ? 16 : 0));
                    if (mode == 1 && (l & 3) == 1) want = 0.f;
                }
            }
            if (o[i] != want) { if (err < 6) printf("  mode %d: lds[%d] = %g, want %g\n", mode, i, o[i], want); ++err; }
            if (o[i] != -7.f) ++touched;
        }
        printf("mode %d: %s (%d mismatches, %d floats written, lds base %g)\n", mode, err ? "FAIL" : "ok", err, touched, o[NF]);
        bad += err != 0;
    }
    printf(bad ? "DMA PROBE FAILED\n" : "DMA PROBE OK\n");
    return bad;
}
