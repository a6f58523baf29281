// Post-process + soft-NMS on the device: what ObjectPoseDetector.post_process + merge_outputs do per
// image on the host (detectors/object_pose.py:167-197 -> utils/post_process.py:12-68, utils/image.py:23-74, and
// soft_nms_nvidia, object_pose.py:27-124), for a whole batch in one launch.  The chain backbone -> decode ->
// post-process -> PnP then only leaves the device with the final, already filtered records.
//
// One workgroup (one wave) per image.
//   1. every detection of the image: output-grid -> original-image coordinates with the image's inverse affine
//      (transform_preds: float32 point, float64 matrix, float64 result; (-10000, -10000) passes through), the
//      `ratio` / 0.32 scalings in float32 exactly as numpy evaluates them, ct = bbox centre;
//   2. keep score > vis_thresh in decode order (merge_outputs :185);
//   3. Gaussian soft-NMS (method 2, sigma 0.5, threshold = vis_thresh) -- the reference's selection-sort formulation
//      with its swap-with-last removal is inherently sequential and order dependent, so one lane walks it over
//      (score, bbox, record index) arrays in LDS, in float64 like the Python floats of the reference;
//   4. the surviving records are written in their final order: out[b][0 .. count[b]).
#include <cmath>  // (the host-testable headers below are included inside the namespace: their std includes come first)
#include <cfloat>

#include "cp_common.h"
#include "../../include/centerpose_hip.h"

namespace {

constexpr int MAXK = 128;

#include "post_common.h"  // xform, post_transform_record, post_filter_nms (host-testable)

__global__ __launch_bounds__(64) void postprocess_kernel(const float* __restrict__ det, int K,
                                                         const double* __restrict__ meta, double vis_thresh, int nms,
                                                         float div_scale, double* __restrict__ out,
                                                         int* __restrict__ count, double* __restrict__ ws) {
    __shared__ double s_score[MAXK];
    __shared__ double s_box[MAXK][4];
    __shared__ int s_idx[MAXK];
    __shared__ int s_n;
    const int b = blockIdx.x, lane = threadIdx.x;
    const double* t = meta + (size_t)b * 8;  // inverse affine (6), ratio (1), pad
    const float ratio = (float)t[6];
    double* ob = ws + (size_t)b * K * CP_POST_STRIDE;  // all K transformed records, decode order (workspace)

    // ---- 1. transform every record (two passes of 64 lanes for K = 100), decode order ----
    for (int k = lane; k < K; k += 64) {
        const float* d = det + ((size_t)b * K + k) * CP_DET_STRIDE;
        double* o = ob + (size_t)k * CP_POST_STRIDE;
        post_transform_record(d, t, ratio, div_scale, o);
        s_score[k] = o[0];
        for (int i = 0; i < 4; ++i) s_box[k][i] = o[24 + i];
    }
    __syncthreads();

    // ---- 2 + 3. threshold filter and soft-NMS: sequential by construction, one lane ----
    if (lane == 0) {
        const int N = post_filter_nms(s_score, s_box, s_idx, K, vis_thresh, nms);
        s_n = N;
        count[b] = N;
    }
    __syncthreads();

    // ---- 4. gather the survivors in their final order ----
    const int N = s_n;
    double* fin = out + (size_t)b * K * CP_POST_STRIDE;
    for (int r = 0; r < N; ++r) {
        const double* src = ob + (size_t)s_idx[r] * CP_POST_STRIDE;
        for (int e = lane; e < CP_POST_STRIDE; e += 64) fin[(size_t)r * CP_POST_STRIDE + e] = e == 0 ? s_score[r] : src[e];
    }
}

// Tracking-input render (first half): `draw_umich_gaussian` (utils/image.py:135-150) for a list of
// (channel, x, y, radius, k) records into the pre_hm / pre_hm_hp planes that CenterPoseTrack feeds back into the
// network (base_detector.py:150-388 decides WHICH points; this kernel draws them).  One workgroup per record walks its
// (2r+1)^2 window, clipped to the map exactly like the reference's slice arithmetic; the value is
// float32(exp(-(dx^2+dy^2) / (2 sigma^2)) * k) with sigma = (2r+1)/6 evaluated in float64, merged with an integer
// atomicMax on the bit pattern (all values are >= 0, so the order of records does not matter: deterministic).
__global__ void render_gaussians_kernel(const double* __restrict__ recs, float* __restrict__ out, int C, int H, int W) {
    const double* r = recs + (size_t)blockIdx.x * 5;
    const int c = (int)r[0], x = (int)r[1], y = (int)r[2], radius = (int)r[3];
    const double k = r[4];
    if (c < 0 || c >= C || radius < 0) return;
    const int d = 2 * radius + 1;
    const double sigma = (double)d / 6.0;
    const double den = 2 * sigma * sigma;
    for (int i = threadIdx.x; i < d * d; i += blockDim.x) {
        const int dy = i / d - radius, dx = i - (i / d) * d - radius;
        const int px = x + dx, py = y + dy;
        if (px < 0 || px >= W || py < 0 || py >= H) continue;
        double g = exp(-((double)dx * dx + (double)dy * dy) / den);
        if (g < 2.220446049250313e-16) g = 0.0;  // gaussian2D: h[h < eps * h.max()] = 0, h.max() = 1 at the centre
        const float v = (float)(g * k);
        if (v > 0.f) atomicMax(reinterpret_cast<int*>(out) + ((size_t)c * H + py) * W + px, __float_as_int(v));
    }
}

}  // namespace

int cp_launch_render_gaussians(const double* recs, int N, float* out, int C, int H, int W, hipStream_t s) {
    if (N <= 0) return CP_OK;
    hipLaunchKernelGGL(render_gaussians_kernel, dim3(N), dim3(256), 0, s, recs, out, C, H, W);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

// PnP input assembly for every post-processed slot (base_detector.py:547-566): slot (b, k) is a detection iff
// k < count[b]; its 8 (rep_mode 0/3/4: `kps`) or 16 (rep_mode 1: displacement / heat-map pairs interleaved per vertex)
// image points, relative size and the image's intrinsics are laid out for cp_pnp_solve.  Absent slots get all points
// invalid, so the solver returns status -1 for them without work.
__global__ void pnp_assemble_kernel(const double* __restrict__ post, const int* __restrict__ count, int B, int K, int npts,
                                    const double* __restrict__ cam_img, float* __restrict__ pts, float* __restrict__ scale,
                                    double* __restrict__ cam) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K, k = i - b * K;
    const double* r = post + (size_t)i * CP_POST_STRIDE;
    float* P = pts + (size_t)i * npts * 2;
    const bool live = k < count[b];
    for (int v = 0; v < 8; ++v) {
        if (npts == 16) {
            P[4 * v + 0] = live ? (float)r[64 + 2 * v] : -10000.f;  // kps_displacement_mean
            P[4 * v + 1] = live ? (float)r[65 + 2 * v] : -10000.f;
            P[4 * v + 2] = live ? (float)r[80 + 2 * v] : -10000.f;  // kps_heatmap_mean
            P[4 * v + 3] = live ? (float)r[81 + 2 * v] : -10000.f;
        } else {
            P[2 * v + 0] = live ? (float)r[30 + 2 * v] : -10000.f;  // kps
            P[2 * v + 1] = live ? (float)r[31 + 2 * v] : -10000.f;
        }
    }
    // relative size as the reference forms it (cuboid_pnp_shell.py:12: scale / scale[1]) and as the host path hands it to
    // cp_pnp_solve: the float64 quotient of the float32-valued fields, rounded to float32
    for (int d = 0; d < 3; ++d) scale[(size_t)i * 3 + d] = live ? (float)(r[2 + d] / r[3]) : 1.f;
    for (int d = 0; d < 4; ++d) cam[(size_t)i * 4 + d] = cam_img[(size_t)b * 4 + d];
}

int cp_launch_pnp_assemble(const double* post, const int* count, int B, int K, int npts, const double* cam_img, float* pts,
                           float* scale, double* cam, hipStream_t s) {
    hipLaunchKernelGGL(pnp_assemble_kernel, dim3((B * K + 127) / 128), dim3(128), 0, s, post, count, B, K, npts, cam_img,
                       pts, scale, cam);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}

int cp_launch_postprocess(const float* det, int B, int K, const double* meta, double vis_thresh, int nms,
                          float div_scale, double* out, int* count, double* ws, hipStream_t s) {
    if (K < 1 || K > MAXK) return CP_ERR_INVALID;
    hipLaunchKernelGGL(postprocess_kernel, dim3(B), dim3(64), 0, s, det, K, meta, vis_thresh, nms, div_scale, out,
                       count, ws);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}
