// CenterPoseTrack's track bookkeeping on the device, for B concurrent videos: replaces, per frame, the
// host-side Python of `BaseDetector.run` between `merge_outputs` and the next frame's `_get_additional_inputs`
// (/root/reference/src/lib/detectors/base_detector.py:501-544 Gaussian fusion, :547-654 boxes, :660-665 `tracker.step`,
// :150-388 which Gaussians are drawn; utils/tracker.py:112-302 `Tracker.step`; utils/pnp/cuboid_pnp_shell.py:26-91).
// The arithmetic lives in track_common.h (scalar functions, also compiled for the host by the logic tests); this file
// is the data movement: five launches per frame, no host synchronisation.
//
//   1  track_prepare_kernel    one lane per (video, detection slot): candidate record = post-processed fields, fusion,
//                              packaged detection PnP answer, "has a box" flag
//   2  track_associate_kernel  one wavefront per video: lane 0 walks the greedy association (inherently sequential:
//                              detection i may only take a track no earlier detection took), then all 64 lanes copy
//                              the records of the new list (520 doubles each) into the other half of the state
//   3  track_advance_kernel    one lane per (video, track): Kalman predict + update (eight 4 x 4 blocks) or init, scale
//                              pool, read-out; writes the inputs of the filtered PnP
//   4  pnp_kernel / pnp_rare_kernel (pnp.hip) on B * cap problems of 8 points; empty slots answer -1 at once
//   5  track_finish_kernel     one lane per (video, track): packaging + visibility rejects of the filtered answer, the
//                              `boxes` flag, and the Gaussian records (centre + 8 vertices) of next frame's inputs
// Work per frame is a few hundred KFLOP in float64 -- latency, not throughput; what matters is that nothing waits for
// the host (the Python loop it replaces cost 0.5 ms per track and frame: profiles/r03_track_e2e.txt).
#include "cp_common.h"
#include "track_common.h"

namespace {

struct StateView {
    int* hdr;        // [0] parity of the current list, then per video (n, id_count, overflow, pad)
    double* tracks;  // [2][B][cap][STRIDE]
};

__host__ __device__ inline size_t state_hdr_bytes(int B) { return ((size_t)(4 + 4 * B) * sizeof(int) + 255) / 256 * 256; }

__device__ __forceinline__ StateView view(void* state, int B) {
    StateView v;
    v.hdr = (int*)state;
    v.tracks = (double*)((char*)state + state_hdr_bytes(B));
    return v;
}

// (launch bounds = the 64 threads these kernels are launched with: without them the compiler assumes 1024-thread blocks,
// caps the kernels at 128 registers and spills the Kalman blocks to scratch)
__global__ __launch_bounds__(64) void track_prepare_kernel(const TrackParams P, const double* __restrict__ vmeta,
                                                           const double* __restrict__ post, const int* __restrict__ count,
                                                           const double* __restrict__ det_pnp, int B, int K,
                                                           double* __restrict__ dets, int* __restrict__ use) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K, k = i - b * K;
    int ok = 0;
    if (k < count[b])
        ok = trk_prepare_det(P, vmeta + (size_t)b * CP_VMETA_STRIDE, post + (size_t)i * 120,
                             det_pnp ? det_pnp + (size_t)i * 40 : nullptr, dets + (size_t)i * CP_TRACK_STRIDE);
    use[i] = ok;
}

__host__ __device__ inline size_t munkres_bytes(int K, int cap) {  // per video: C [K cap] doubles, marked [K cap] bytes, covers, path
    return ((size_t)K * cap * 9 + (size_t)(K + cap) * (2 + 8) + 255) / 256 * 256;
}

__global__ __launch_bounds__(64) void track_associate_kernel(const TrackParams P, const int* __restrict__ count, int B, int K,
                                                             const double* __restrict__ dets, int* __restrict__ use,
                                                             void* state, unsigned char* __restrict__ munkres_ws) {
    __shared__ int plan[3 * 128];
    __shared__ int det_idx[128];
    __shared__ unsigned char taken[128];
    __shared__ int s_n;
    // work space of the optimal assignment (P.hungarian): one entry per detection / track, LS = 129
    __shared__ double ls_u[129], ls_v[129], ls_spc[129];
    __shared__ int ls_path[129], ls_c4r[129], ls_r4c[129], ls_rem[129], ls_match[129];
    __shared__ unsigned char ls_sr[129], ls_sc[129];
    const int b = blockIdx.x, lane = threadIdx.x;
    StateView S = view(state, B);
    const int par = S.hdr[0];
    int* h = S.hdr + 4 + 4 * b;
    const double* prev = S.tracks + ((size_t)par * B + b) * P.cap * CP_TRACK_STRIDE;
    double* next = S.tracks + ((size_t)(par ^ 1) * B + b) * P.cap * CP_TRACK_STRIDE;
    const double* d = dets + (size_t)b * K * CP_TRACK_STRIDE;
    int* u = use + (size_t)b * K;
    const int nd = count[b] < K ? count[b] : K;
    if (lane == 0) {
        int any = 0;
        for (int k = 0; k < nd; ++k) any |= u[k];
        if (!any)
            for (int k = 0; k < nd; ++k) u[k] = 1;  // no box at all: every detection takes part (tracker.py:116-117)
        int idc = h[1], dropped = 0;
        const TrkLsapWork W = {ls_u, ls_v, ls_spc, ls_path, ls_c4r, ls_r4c, ls_rem, ls_sr, ls_sc};
        // Munkres work space of this video (global memory: the reduced cost matrix does not fit LDS)
        unsigned char* mw = munkres_ws + (size_t)b * munkres_bytes(K, P.cap);
        TrkMunkresWork MW;
        MW.C = (double*)mw;
        MW.path = (int*)(mw + (size_t)K * P.cap * 8);
        MW.marked = mw + (size_t)K * P.cap * 8 + (size_t)(K + P.cap) * 8;
        MW.row_unc = MW.marked + (size_t)K * P.cap;
        MW.col_unc = MW.row_unc + (K + P.cap);
        const int n = trk_associate(P, d, u, nd, prev, h[0], plan, &idc, det_idx, taken, &dropped, &W, ls_match, &MW);
        h[1] = idc;
        h[2] += dropped;  // sticky count of list entries dropped because a frame needed more than `cap` (cp_track_status)
        s_n = n;
    }
    __syncthreads();
    const int n = s_n;
    for (int t = 0; t < n; ++t)
        for (int e = lane; e < CP_TRACK_STRIDE; e += 64)
            trk_materialise(plan + 3 * t, d, prev, next + (size_t)t * CP_TRACK_STRIDE, e, e + 1);
    if (lane == 0) h[3] = n;  // published as h[0] by the advance kernel's launch (same stream: ordered)
}

__global__ __launch_bounds__(64) void track_advance_kernel(const TrackParams P, const double* __restrict__ vmeta, int B,
                                                           void* state, float* __restrict__ pts, float* __restrict__ scale,
                                                           double* __restrict__ cam) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * P.cap) return;
    const int b = i / P.cap, t = i - b * P.cap;
    StateView S = view(state, B);
    const int par = S.hdr[0];
    const int n = S.hdr[4 + 4 * b + 3];
    const double* prev = S.tracks + ((size_t)par * B + b) * P.cap * CP_TRACK_STRIDE;
    double* next = S.tracks + ((size_t)(par ^ 1) * B + b) * P.cap * CP_TRACK_STRIDE;
    float p16[16], s3[3] = {1.f, 1.f, 1.f};
    for (int e = 0; e < 16; ++e) p16[e] = -10000.f;
    if (t < n) trk_track_stage(P, next + (size_t)t * CP_TRACK_STRIDE, prev, p16, s3);
    for (int e = 0; e < 16; ++e) pts[(size_t)i * 16 + e] = p16[e];
    for (int e = 0; e < 3; ++e) scale[(size_t)i * 3 + e] = s3[e];
    for (int e = 0; e < 4; ++e) cam[(size_t)i * 4 + e] = vmeta[(size_t)b * CP_VMETA_STRIDE + VM_CAM + e];
}

__global__ __launch_bounds__(64) void track_finish_kernel(const TrackParams P, const double* __restrict__ vmeta, int B,
                                                          void* state, const double* __restrict__ rows,
                                                          double* __restrict__ recs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * P.cap) return;
    const int b = i / P.cap, t = i - b * P.cap;
    StateView S = view(state, B);
    const int par = S.hdr[0];
    const int n = S.hdr[4 + 4 * b + 3];
    double* next = S.tracks + ((size_t)(par ^ 1) * B + b) * P.cap * CP_TRACK_STRIDE;
    double* rec = recs + (size_t)i * 45;
    if (t < n)
        trk_finish_stage(P, vmeta + (size_t)b * CP_VMETA_STRIDE, next + (size_t)t * CP_TRACK_STRIDE,
                         rows ? rows + (size_t)i * 40 : nullptr, b, B + 8 * b, rec);
    else
        for (int e = 0; e < 9; ++e) rec[5 * e] = -1.0;
}

// the new list becomes the current one
__global__ __launch_bounds__(64) void track_flip_kernel(int B, void* state) {
    int* hdr = (int*)state;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) hdr[4 + 4 * b] = hdr[4 + 4 * b + 3];
    if (b == 0) hdr[0] ^= 1;
}

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

size_t cp_track_state_bytes_impl(int B, int cap) {
    return state_hdr_bytes(B) + (size_t)2 * B * cap * CP_TRACK_STRIDE * sizeof(double);
}

size_t cp_track_ws_bytes_impl(int B, int K, int cap) {
    const size_t n = (size_t)B * cap;
    return align256((size_t)B * K * CP_TRACK_STRIDE * 8) + align256((size_t)B * K * 4) + align256(n * 16 * 4) +
           align256(n * 3 * 4) + align256(n * 4 * 8) + align256(n * 40 * 8) + align256(cp_pnp_ws_bytes((int)n)) +
           (size_t)B * munkres_bytes(K, cap);
}

int cp_launch_track_step(hipStream_t s, const TrackParams& P, const double* vmeta, const double* post, const int* count,
                         const double* det_pnp, int B, int K, void* state, double* recs, void* ws) {
    char* w = (char*)ws;
    double* dets = (double*)w;
    w += align256((size_t)B * K * CP_TRACK_STRIDE * 8);
    int* use = (int*)w;
    w += align256((size_t)B * K * 4);
    const size_t n = (size_t)B * P.cap;
    float* pts = (float*)w;
    w += align256(n * 16 * 4);
    float* scale = (float*)w;
    w += align256(n * 3 * 4);
    double* cam = (double*)w;
    w += align256(n * 4 * 8);
    double* rows = (double*)w;
    w += align256(n * 40 * 8);
    unsigned char* munkres_ws = (unsigned char*)w + align256(cp_pnp_ws_bytes((int)n));  // behind the PnP work space (at `w`)
    hipLaunchKernelGGL(track_prepare_kernel, dim3((B * K + 63) / 64), dim3(64), 0, s, P, vmeta, post, count, det_pnp, B, K,
                       dets, use);
    hipLaunchKernelGGL(track_associate_kernel, dim3(B), dim3(64), 0, s, P, count, B, K, dets, use, state, munkres_ws);
    hipLaunchKernelGGL(track_advance_kernel, dim3(((int)n + 63) / 64), dim3(64), 0, s, P, vmeta, B, state, pts, scale, cam);
    const bool pnp = P.use_pnp && (P.kalman || P.scale_pool);
    if (pnp) {
        const int rc = cp_launch_pnp(s, pts, scale, cam, (int)n, 8, rows, w);
        if (rc != CP_OK) return rc;
    }
    hipLaunchKernelGGL(track_finish_kernel, dim3(((int)n + 63) / 64), dim3(64), 0, s, P, vmeta, B, state,
                       pnp ? rows : nullptr, recs);
    hipLaunchKernelGGL(track_flip_kernel, dim3((B + 63) / 64), dim3(64), 0, s, B, state);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}
