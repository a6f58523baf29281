/*
 * centerpose_hip.h — C ABI of libcenterpose_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the CenterPose inference hot path.  Every entry point takes plain
 * pointers and sizes (no torch / ATen types), returns 0 on success or a negative CP_ERR_* code
 * (never prints-and-continues like the reference's launchers, dcn_v2_im2col_cuda.cu:346-350),
 * launches on the caller's stream and never synchronises.  Unless stated otherwise pointers are
 * DEVICE pointers to float32.  Paths below are relative to the reference tree
 * (/root/reference/src/lib/...).
 *
 * Threading: stateless entry points (cp_dcnv2_forward, cp_conv2d_nhwc, cp_decode, cp_postprocess, cp_pnp_solve,
 * cp_preprocess, cp_render_gaussians) may be called concurrently on different streams.  A cp_model is not re-entrant:
 * one forward / detect at a time per model (its workspace, profile records and hipGraph cache are per model).
 * cp_last_error() is per calling thread (the message of that thread's last failing call); cp_set_default_precision() is
 * process-global.  The kernel-selection hooks the test-suite uses are NOT part of this header: centerpose_hip_testing.h.
 */
#ifndef CENTERPOSE_HIP_H
#define CENTERPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cp_stream_t; /* hipStream_t */
typedef struct cp_model cp_model;

#define CP_OK 0
#define CP_ERR_INVALID (-1)
#define CP_ERR_LAUNCH (-2)
#define CP_ERR_ALLOC (-3)
#define CP_ERR_STATE (-4)

/* Library / device info.  cp_version() returns a static string.
 * CP_ABI_VERSION counts incompatible changes of this header; cp_abi_version() returns the value the library was built
 * with, so a caller compiled against another revision can refuse to run instead of passing arguments with a stale
 * meaning.  History: 1 = round-1 header; 2 = cp_preprocess takes the FORWARD 2x3 affine as double[6] and inverts it
 * itself (round 1: the inverse as float[6]); 3 = cp_dcnv2_forward accepts every shape of the reference op (generic
 * kernel), cp_num_kernel_variants() / cp_num_roles() size the profile buffers; 4 = cp_track_* added; 5 = cp_track_status, list truncation instead of reset on overflow;
 * 6 = cp_preprocess_batch, cp_linear_assignment, CP_NUM_KERNEL_VARIANTS 43, cp_set_debug moved out of this header (centerpose_hip_testing.h). */
#define CP_ABI_VERSION 6
const char* cp_version(void);
int cp_abi_version(void);
const char* cp_last_error(void);

/* ------------------------------------------------------------------------------------------
 * DCNv2 forward — replaces `_ext.dcn_v2_forward`
 *   models/networks/DCNv2/src/vision.cpp:5, dcn_v2.h:9-46, cuda/dcn_v2_cuda.cu:42-172 and its
 *   raw-pointer launcher `modulated_deformable_im2col_cuda` (cuda/dcn_v2_im2col_cuda.h:67-79).
 * Same tensor layouts as the reference (all contiguous NCHW float32):
 *   input [B,C,H,W], weight [Co,C,kh,kw], bias [Co], offset [B,dg*2*kh*kw,Ho,Wo] ((dh,dw)
 *   interleaved per tap), mask [B,dg*kh*kw,Ho,Wo], output [B,Co,Ho,Wo].
 * Ho = (H + 2*ph - (dh*(kh-1)+1)) / sh + 1 (Wo likewise), as dcn_v2_cuda.cu:75-76.
 * Every shape the reference op accepts is accepted (C % deformable_group == 0).  Two paths, same results:
 *   - what CenterPose uses (pose_dla_dcn.py:384: kh=kw=3, stride 1, pad 1, dilation 1, deformable_group 1,
 *     C % 16 == 0, Co > 32): the fused gather + matrix-core kernels (dcn16p.hip / dcn16.hip / igemm.hip);
 *   - anything else (the reference's own self-checks: DCNv2/testcpu.py:32-67 with 2 channels, :169-180 with
 *     deformable_group 2; other kernel sizes / strides / dilations): a generic float32 kernel (dcn_generic.hip),
 *     correct but not tuned.
 * `workspace` must hold cp_dcnv2_workspace_bytes(...) bytes (NHWC staging + packed weights; the generic path does not
 * touch it).
 * ------------------------------------------------------------------------------------------ */
size_t cp_dcnv2_workspace_bytes(int B, int C, int H, int W, int Co);
int cp_dcnv2_forward(cp_stream_t stream, const float* input, const float* weight, const float* bias,
                     const float* offset, const float* mask, float* output, int B, int C, int H, int W, int Co,
                     int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int deformable_group,
                     void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------
 * Backbone + heads — replaces `create_model` / `load_model` / `model(images, pre_images,
 *   pre_hms, pre_hm_hp)[-1]`  (models/model.py:26-87, models/networks/pose_dla_dcn.py:457-570,
 *   detectors/object_pose.py:135-138).
 *
 * arch: "dla_34" (DLA-34 + DCNv2 up-sampling) or "dlav1_34" (+ ConvGRU + GroupNorm heads).
 * heads: names/classes in the order of `opt.heads` (opts.py:394-426).
 * Parameters are fed one tensor at a time under the reference's state_dict names (HOST float32
 * pointers; `module.` prefixes are the caller's business as in model.py:43-48); finalize folds
 * eval-mode BatchNorm into per-channel scale/shift, re-packs every convolution as [tap][ci][co]
 * and uploads.  Unknown names are ignored (CP_OK) like load_model's "Drop parameter" branch;
 * finalize fails with CP_ERR_STATE if a required tensor is missing.
 * ------------------------------------------------------------------------------------------ */
int cp_model_create(const char* arch, int tracking_task, int num_heads, const char* const* head_names,
                    const int* head_classes, int head_conv, cp_model** out);
int cp_model_set_param(cp_model* m, const char* name, const float* host_data, int64_t numel);
int cp_model_finalize(cp_model* m);
void cp_model_destroy(cp_model* m);

/* Bytes of device scratch needed by cp_model_forward for a batch of B images of H x W. */
size_t cp_model_workspace_bytes(cp_model* m, int B, int H, int W);

/* images [B,3,H,W] NCHW (H, W multiples of 32).  pre_img [B,3,H,W], pre_hm [B,1,H,W],
 * pre_hm_hp [B,8,H,W] may each be NULL (pose_dla_dcn.py:312-318).  head_out[i] receives head i
 * as [B,classes_i,H/4,W/4] NCHW — raw logits, except that with sigmoid_hm != 0 the 'hm' and
 * 'hm_hp' heads are returned post-sigmoid (object_pose.py:136-138 fused into the epilogue). */
int cp_model_forward(cp_model* m, cp_stream_t stream, int B, int H, int W, const float* images,
                     const float* pre_img, const float* pre_hm, const float* pre_hm_hp, float* const* head_out,
                     int sigmoid_hm, void* workspace, size_t workspace_bytes);

/* One frame batch end to end on the device: backbone + heads + sigmoid(hm, hm_hp) + decode — what
 * `ObjectPoseDetector.process` does (detectors/object_pose.py:131-165) — in ONE call.  Arguments as in
 * cp_model_forward and cp_decode; head_out[] receives the head tensors (hm / hm_hp post-sigmoid), det the
 * [B,K,118] records.  With use_graph != 0 the launch sequence is captured into a hipGraph on first use (keyed
 * by every pointer / size argument, so buffers must be reused) and replayed afterwards: a frame costs one graph
 * launch instead of ~120 kernel launches.  Needs a non-default stream; ignored while profiling is armed. */
size_t cp_model_detect_workspace_bytes(cp_model* m, int B, int H, int W, int K);
int cp_model_detect(cp_model* m, cp_stream_t stream, int B, int H, int W, const float* images, const float* pre_img,
                    const float* pre_hm, const float* pre_hm_hp, float* const* head_out, int K, int rep_mode,
                    int fit_gaussian, float balance, int legacy_bool_mask, float* det, void* workspace,
                    size_t workspace_bytes, int use_graph);

/* Debug/parity aid: same as cp_model_forward but additionally copies the named intermediate
 * activation (names follow the reference module paths, e.g. "base.level3", "dla_up.ida_2.node_3",
 * "feat", "convGRU.step1") to tap_out as NCHW.  tap_dims receives {C,H,W}. */
int cp_model_forward_tap(cp_model* m, cp_stream_t stream, int B, int H, int W, const float* images,
                         const float* pre_img, const float* pre_hm, const float* pre_hm_hp, float* const* head_out,
                         int sigmoid_hm, void* workspace, size_t workspace_bytes, const char* tap_name,
                         float* tap_out, int* tap_dims);

/* Arithmetic of the convolution / DCNv2 contractions (activations, weights at the boundary, accumulation and
 * epilogues are float32 in both modes):
 *   CP_PREC_F32   (0) exact float32 matrix instructions (v_mfma_f32_32x32x2_f32), 157 TFLOP/s ceiling
 *   CP_PREC_F16X3 (1) every float32 operand split into two binary16 numbers, products evaluated as
 *                     hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with float32 accumulation (relative product
 *                     error < 2^-20), 833 TFLOP/s ceiling; layers whose channel counts are not multiples of 32
 *                     (the 16-channel stem levels, final 1x1 heads) stay on the exact path.
 * cp_set_default_precision affects models created afterwards and the stand-alone cp_conv2d_nhwc / cp_dcnv2_forward. */
#define CP_PREC_F32 0
#define CP_PREC_F16X3 1
int cp_set_default_precision(int precision);
int cp_model_set_precision(cp_model* m, int precision);

/* Per-launch timing of the implicit-GEMM kernels with HIP events recorded on the launch stream
 * (replaces the reference's wall-clock `torch.cuda.synchronize()` fences, base_detector.py:466-498).
 * cp_model_profile(m, 1) arms it; every conv / DCN launch of later forwards is bracketed by an event
 * pair.  cp_model_profile_read drains them: out[v*4 + 0..3] = {launches, total milliseconds, total
 * algorithmic FLOPs (2*M*Cout*KH*KW*Cin), total algorithmic bytes (input + output + weights
 * [+ offsets/mask] [+ residual], float32)} per kernel variant v in [0, CP_NUM_KERNEL_VARIANTS). */
#define CP_NUM_KERNEL_VARIANTS 43
int cp_num_kernel_variants(void); /* the value the LIBRARY was built with: size cp_model_profile_read's buffer from it */
int cp_model_profile(cp_model* m, int enable);
int cp_model_profile_read(cp_model* m, double* out, int num_variants);
const char* cp_kernel_variant_name(int v);
/* The same launches grouped by what they compute (the figures BASELINE.json's north_star names: DCNv2 traffic, 1x1
 * convolution MFMA rate, decode time).  Valid after cp_model_profile_read: out[r*4 + 0..3] = {launches, milliseconds,
 * algorithmic FLOPs, algorithmic bytes} of role r since the previous read. */
#define CP_ROLE_CONV 0        /* 3x3 / 7x7 convolutions of the base network and the hourglass */
#define CP_ROLE_CONV1X1 1     /* 1x1 projections, Root nodes, hourglass skips */
#define CP_ROLE_DCN 2         /* DCNv2 gather + contraction (pose_dla_dcn.py:386-389) */
#define CP_ROLE_DCN_OFFSET 3  /* conv_offset_mask Cin->27 (dcn_v2.py:105-111) */
#define CP_ROLE_HEAD 4        /* 3x3 of a prediction head (fused heads: 3x3 + 1x1) */
#define CP_ROLE_HEAD_FINAL 5  /* final 1x1 of an un-fused head */
#define CP_ROLE_GRU 6         /* ConvGRU convolutions (convGRU.py:32-39) */
#define CP_ROLE_LOWC 7        /* stem / level0 / level1 direct kernels (f16x3 mode) */
#define CP_ROLE_DECODE 8      /* cp_model_detect's decode launch (both kernels) */
#define CP_NUM_ROLES 9
int cp_num_roles(void);
int cp_model_profile_roles(cp_model* m, double* out, int num_roles);
const char* cp_role_name(int role);

/* ------------------------------------------------------------------------------------------
 * Generic NHWC convolution (exposed for unit tests of the implicit-GEMM kernel).
 *   x [B,H,W,Cin] NHWC, w [Cout,Cin,KH,KW] (reference/PyTorch layout, DEVICE), scale/shift/
 *   residual may be NULL.  out [B,Ho,Wo,Cout] NHWC.  act: 0 none, 1 relu, 2 sigmoid.
 * ------------------------------------------------------------------------------------------ */
size_t cp_conv2d_workspace_bytes(int Cin, int Cout, int KH, int KW);
int cp_conv2d_nhwc(cp_stream_t stream, const float* x, const float* w, const float* scale, const float* shift,
                   const float* residual, float* out, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                   int stride, int pad, int act, void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------
 * Heat-map decode — replaces `object_pose_decode(..., Inference=True)` (models/decode.py:72-375,
 *   models/utils.py:43-47; called from detectors/object_pose.py:154-161) including the 13
 *   device->host copies and the per-point Python loop (decode.py:191-252).
 * Inputs are the head tensors, NCHW float32 on the device, H x W = output grid (<= 32768 pixels, W % 4 == 0),
 * one category, 8 joints: hm [B,1,H,W], hps [B,16,H,W], wh [B,2,H,W], hm_hp [B,8,H,W] are
 * required (the detector's Inference configuration); hps_uncertainty [B,16], scale [B,3],
 * scale_uncertainty [B,3], reg [B,2], hp_offset [B,2], tracking [B,2], tracking_hp [B,16] may be
 * NULL (decode.py:304-345 zero-fill / +0.5 rules).  hm and hm_hp must already be sigmoided unless
 * apply_sigmoid != 0, in which case they hold logits and are overwritten with their sigmoid
 * (object_pose.py:136-138).
 *   K                 opt.K (<= 128)                      rep_mode   opt.rep_mode (0..4)
 *   fit_gaussian      opt.tracking_task || opt.refined_Kalman || rep_mode == 2 (decode.py:222)
 *   balance           opt.balance_coefficient[opt.c] (decode.py:309)
 *   legacy_bool_mask  0: `mask_2 == 7` is the AND of its 7 conditions (torch <= 1.1, what the
 *                     published models were used with); 1: reproduce torch >= 1.2, where the sum of
 *                     bool tensors can never equal 7 and every kps_heatmap_* stays -10000.
 * Output det [B,K,118]: bboxes[0:4] score[4] kps[5:21] cls[21] obj_scale[22:25]
 *   obj_scale_uncertainty[25:28] tracking[28:30] tracking_hp[30:46] kps_displacement_mean[46:62]
 *   kps_displacement_std[62:78] kps_heatmap_mean[78:94] kps_heatmap_std[94:110]
 *   kps_heatmap_height[110:118]  — the 13 keys of decode.py:347-361, output-grid units.
 * Ordering: (score desc, pixel index asc); torch.topk's order among exactly equal scores is
 * implementation-defined, so parity is defined on distinct scores.
 * ------------------------------------------------------------------------------------------ */
#define CP_DET_STRIDE 118
size_t cp_decode_workspace_bytes(int B, int K);
int cp_decode(cp_stream_t stream, int B, int H, int W, float* hm, const float* hps, const float* wh,
              const float* hps_uncertainty, const float* scale, const float* scale_uncertainty, const float* reg,
              float* hm_hp, const float* hp_offset, const float* tracking, const float* tracking_hp, int K,
              int rep_mode, int fit_gaussian, float balance, int legacy_bool_mask, int apply_sigmoid, float* det,
              void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------
 * Pre-process — replaces `BaseDetector.pre_process`'s image work
 *   (detectors/base_detector.py:127-134: cv2.resize when scale != 1, cv2.warpAffine(..., INTER_LINEAR), then
 *    (x/255 - mean)/std, HWC->CHW).
 * image_hwc_bgr: DEVICE uint8 [H,W,3] (BGR as cv2.imread gives); trans6: HOST double[6], the row-major 2x3 FORWARD
 * matrix `trans_input` (source -> network input), inverted inside exactly as cv::invertAffineTransform does;
 * mean3/std3: HOST float[3] (opts.py:436-437); out_chw: DEVICE float32 [3,out_h,out_w].
 * Both kernels follow OpenCV's fixed-point arithmetic (5-bit bilinear weights for the warp, 11-bit coefficients for the
 * resize; oracle/cv_emul.py restates it), so the network input is built from the same rounded 8-bit values as the
 * reference's.  cp_resize_u8: in / out DEVICE uint8 [H,W,C] -> [out_h,out_w,C].
 * ------------------------------------------------------------------------------------------ */
int cp_preprocess(cp_stream_t stream, const unsigned char* image_hwc_bgr, int H, int W, const double* trans6,
                  const float* mean3, const float* std3, float* out_chw, int out_h, int out_w);
/* The same warp + normalise for B frames of one size that share the transform (fix_res batches, run_batch): images DEVICE
 * uint8 [B,H,W,3] contiguous -> out DEVICE float32 [B,3,out_h,out_w]; one launch. */
int cp_preprocess_batch(cp_stream_t stream, const unsigned char* images_bhwc_bgr, int B, int H, int W, const double* trans6,
                        const float* mean3, const float* std3, float* out_bchw, int out_h, int out_w);
int cp_resize_u8(cp_stream_t stream, const unsigned char* image_hwc, int H, int W, int C, unsigned char* out_hwc,
                 int out_h, int out_w);

/* ------------------------------------------------------------------------------------------
 * Post-process + soft-NMS — replaces `ObjectPoseDetector.post_process` + `merge_outputs`
 *   (detectors/object_pose.py:167-197 -> utils/post_process.py:12-68 `object_pose_post_process`,
 *    utils/image.py:23-32 `transform_preds`, object_pose.py:27-124 `soft_nms_nvidia` with Nt=0.5, method=2,
 *    threshold=vis_thresh).
 * det:   DEVICE float32 [B,K,118] from cp_decode.
 * meta:  DEVICE float64 [B,8]: 0..5 = get_affine_transform(c, s, 0, (out_w,out_h), inv=1) row-major (image.py:35-68),
 *        6 = s / max(out_w, out_h), 7 unused.
 * out:   DEVICE float64 [B,K,CP_POST_STRIDE]; image b's kept detections, in the reference's final order, are
 *        out[b][0 .. count[b]) with fields: score 0 | cls 1 | obj_scale 2 | obj_scale_uncertainty 5 |
 *        kps_displacement_std 8 | bbox 24 | ct 28 | kps 30 | tracking 46 | tracking_hp 48 | kps_displacement_mean 64 |
 *        kps_heatmap_mean 80 | kps_heatmap_std 96 | kps_heatmap_height 112.
 * count: DEVICE int32 [B].   nms: 0 = threshold filter only (opt.nms False), 1 = Gaussian soft-NMS.
 * vis_thresh is a double because the reference compares float64 scores with the Python float opt.vis_thresh.
 * div_scale: the `scale` of multi-scale testing (object_pose.py:171-176), 1 for the demo configuration.
 * workspace: cp_postprocess_workspace_bytes(B, K) bytes.
 * ------------------------------------------------------------------------------------------ */
#define CP_POST_STRIDE 120
size_t cp_postprocess_workspace_bytes(int B, int K);
int cp_postprocess(cp_stream_t stream, const float* det, int B, int K, const double* meta, double vis_thresh, int nms,
                   float div_scale, double* out, int* count, void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------
 * Tracking-input render — the drawing half of `BaseDetector._get_additional_inputs`
 *   (detectors/base_detector.py:150-388 -> utils/image.py:135-150 `draw_umich_gaussian`, :126-132 `gaussian2D`).
 * recs: DEVICE float64 [N,5] = (channel, x, y, radius, k) per Gaussian (x, y, radius integral; k = peak value);
 * out:  DEVICE float32 [C,H,W] (pre_hm: C=1, pre_hm_hp: C=8, or both stacked); cleared first when clear != 0.
 * Each record is drawn clipped to the map and merged with max(), exactly as the reference's in-place np.maximum.
 * ------------------------------------------------------------------------------------------ */
int cp_render_gaussians(cp_stream_t stream, const double* recs, int N, float* out, int C, int H, int W, int clear);

/* ------------------------------------------------------------------------------------------
 * Batched cuboid PnP — replaces the per-detection loop `pnp_shell` -> `CuboidPNPSolver.solve_pnp`
 *   -> `cv2.solvePnPGeneric(SOLVEPNP_ITERATIVE | SOLVEPNP_EPNP)` + `cv2.projectPoints`
 *   (utils/pnp/cuboid_pnp_shell.py:11-24, utils/pnp/cuboid_pnp_solver.py:141-239,
 *   detectors/base_detector.py:547-654).
 *   pts   [N, npts, 2] float32 image points (the reference hands cv2 float64 values, cuboid_pnp_solver.py:153; the
 *         float32 boundary perturbs a coordinate of a few hundred pixels by <= 3e-5 px, far inside the 1 degree / 1 %
 *         pose tolerance; all arithmetic after the load is float64), npts = 8 (rep_mode 0/3/4: `kps`) or 16 (rep_mode 1:
 *         displacement/heat-map pairs interleaved per vertex, base_detector.py:558-566); a point
 *         with x or y < -5000 is invalid (cuboid_pnp_solver.py:145)
 *   scale [N, 3] float32 relative cuboid size (divided by its y component inside, shell :12)
 *   cam   [N, 4] float64 (fx, fy, cx, cy) of each detection's image
 *   out   [N, 40] float64:
 *     [0] status: 1 solved, 2 solved but t_z < 0 (reference drops it, solver :207-220), -1 < 4 valid points,
 *                 0 failure (degenerate correspondences, e.g. 4-5 coplanar points handed to EPnP)
 *                 Branches as the reference selects them (solver :157-171): >= 6 valid non-planar points
 *                 SOLVEPNP_ITERATIVE (DLT + LM), coplanar model points its homography initialisation + LM,
 *                 4-5 valid points SOLVEPNP_EPNP (no refinement; approximate for exactly 4 points, as published)
 *     [1:4] rvec  [4:7] tvec (OpenCV frame)  [7] RMS reprojection error
 *     [8:24] the 8 cuboid vertices projected with (rvec, tvec), pixels
 *     [24:28] quaternion xyzw (OpenCV frame)   [28:31] location, [31:35] quaternion xyzw in the
 *     OpenGL frame the evaluation uses (solver :179-196)   [35] valid points  [36] LM iterations
 * ------------------------------------------------------------------------------------------ */
#define CP_PNP_STRIDE 40
size_t cp_pnp_workspace_bytes(int N);
int cp_pnp_solve(cp_stream_t stream, const float* pts, const float* scale, const double* cam, int N, int npts,
                 double* out, void* workspace, size_t workspace_bytes);

/* PnP of every post-processed detection of a batch without leaving the device -- replaces the per-detection loop of
 *   `BaseDetector.run` (detectors/base_detector.py:547-566 point assembly by rep_mode, :652 `pnp_shell`) between
 *   `merge_outputs` and the packaging of `cuboid_pnp_shell.py:26-91`.
 *   post / count: outputs of cp_postprocess ([B,K,CP_POST_STRIDE] float64, [B] int32).
 *   rep_mode: 0 / 3 / 4 -> 8 points from `kps`; 1 -> 16 points, (kps_displacement_mean, kps_heatmap_mean) per vertex.
 *   cam: DEVICE float64 [B,4] (fx, fy, cx, cy) per image.
 *   The assembly casts the float64 record fields to cp_pnp_solve's float32 `pts` / `scale` inputs (same rounding as the
 *   host path, which builds float32 arrays from the same float64 values): row (b,k) is bit-identical to cp_pnp_solve on
 *   points assembled on the host by the reference rule (tests/test_gpu_pose_chain.py).
 *   out: DEVICE float64 [B,K,CP_PNP_STRIDE]; row (b,k) as cp_pnp_solve for k < count[b], status -1 beyond.
 * No host synchronisation and no host-visible count: the whole chain backbone -> decode -> post-process -> PnP is a
 * fixed launch sequence. */
size_t cp_pnp_from_post_workspace_bytes(int B, int K);
int cp_pnp_from_post(cp_stream_t stream, const double* post, const int* count, int B, int K, int rep_mode,
                     const double* cam, double* out, void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------
 * CenterPoseTrack bookkeeping for B concurrent videos, on the device -- replaces, per frame, the host-side Python of
 *   `BaseDetector.run` between `merge_outputs` and the next frame's inputs (detectors/base_detector.py:501-544 Gaussian
 *   fusion, :547-654 `boxes`, :660-665 `self.tracker.step`, :150-388 which Gaussians `_get_additional_inputs` draws;
 *   utils/tracker.py:112-302 `Tracker.step`: greedy association, 32-state Kalman filter per track, scale pool, filtered
 *   PnP; utils/pnp/cuboid_pnp_shell.py:26-91 packaging and visibility rejects).
 * Supported configuration = the demo's (src/demo.py:117-129): `tracking_task` with `kalman` and / or `scale_pool`,
 * greedy or Hungarian association (`hungarian`), `Tracker` or -- `baseline`, the reference's `--refined_Kalman` --
 * `Tracker_baseline` (utils/tracker_baseline.py:14-310), no ground-truth seeding; anything else stays on the host mirror
 * (centerpose_amd/lib/utils/tracker.py).  Per video the state holds at most `cap` (<= CP_TRACK_CAP) tracks.
 *
 *   params        HOST struct (opt fields; cat_rule: 0 camera / bottle / cup, 1 book / chair / cereal_box, 2 bike / laptop /
 *                 shoe -- the visibility reject of cuboid_pnp_shell.py:70-84; K = slots per image of post / det_pnp)
 *   vmeta         DEVICE float64 [B,16] per video: trans_input 2x3 row-major | width height inp_width inp_height |
 *                 fx fy cx cy | 2 pad  (the `meta` of base_detector.py:142-147)
 *   post, count   outputs of cp_postprocess ([B,K,CP_POST_STRIDE] float64, [B] int32)
 *   det_pnp       DEVICE float64 [B,K,CP_PNP_STRIDE] from cp_pnp_from_post, or NULL when params->use_pnp == 0
 *   state         DEVICE, cp_track_state_bytes(B, cap) bytes, zeroed by cp_track_reset:
 *                   int32 hdr[4 + 4 B]: hdr[0] = which half holds the current lists; per video b at hdr[4 + 4 b]:
 *                   n tracks, last id given out, sticky count of list entries DROPPED because a frame needed more than cap
 *                   tracks (the list then keeps its first cap entries in the reference's order -- matched, new by score,
 *                   coasting -- and no id is spent on a dropped detection: read it with cp_track_status), scratch;  then (256-byte aligned) float64 tracks[2][B][cap][CP_TRACK_STRIDE]
 *                 track record (doubles): 0 tracking_id | 1 age | 2 active | 3 flags (bit 0 location / quaternion /
 *                   projected_cuboid / kps_3d_cam / kps_pnp valid, 1 kps_pnp_kf / kps_3d_cam_kf / kps_ori_kf valid, 2 in
 *                   this frame's `boxes`, 3 kps_ori valid, 4 filter state valid) | 4 the CP_POST_STRIDE detection fields |
 *                   124 kps_fusion_mean[16] | 140 kps_fusion_std[16] | 156 location[3] | 159 quaternion_xyzw[4] |
 *                   163 projected_cuboid[16] | 179 kps_pnp[18] | 197 kps_3d_cam[27] | 224 kps_ori[18] | 242 kf.x[32] |
 *                   274 kf.P as 8 blocks of 4x4 | 402 scale-pool sums[7] | 409 kps_mean_kf[16] | 425 kps_std_kf[16] |
 *                   441 obj_scale_kf[3] | 444 obj_scale_uncertainty_kf[3] | 447 vertex confidences[8] | 455 kps_pnp_kf[18] |
 *                   473 kps_3d_cam_kf[27] | 500 kps_ori_kf[18]
 *   render_recs   DEVICE float64 [B,cap,9,5]: next frame's Gaussians as (plane, x, y, radius, k) records for
 *                 cp_render_gaussians(recs, B*cap*9, out, C = 9 B, ...): plane b = pre_hm of video b, plane B + 8 b + j =
 *                 pre_hm_hp[j] of video b; plane -1 = nothing to draw
 * No host synchronisation; five kernel launches + the batched PnP of the filtered vertices.
 * ------------------------------------------------------------------------------------------ */
#define CP_TRACK_STRIDE 520
#define CP_TRACK_CAP 128
typedef struct cp_track_params {
    double new_thresh, pre_thresh, R, conf_lo, conf_hi;
    int max_age, kalman, scale_pool, use_pnp, hps_uncertainty, show_axes, cat_rule, render_hm_mode, render_hmhp_mode, pre_hm,
        pre_hm_hp, K, cap;
    int hungarian; /* != 0: optimal assignment (tracker.py:154-170) instead of the greedy walk.
                    *   1 = the solver the reference calls (tracker.py:6,157): sklearn.utils.linear_assignment_ of the pinned
                    *       scikit-learn 0.22.2, i.e. the Kuhn-Munkres state machine, restated operation for operation
                    *       (csrc/track_common.h: trk_munkres; the module no longer exists in scikit-learn >= 0.23);
                    *   2 = scipy.optimize.linear_sum_assignment's rectangular shortest-augmenting-path solver, restated
                    *       operation for operation (trk_lsap) -- what a host with a current scipy / scikit-learn would run, and
                    *       the faster of the two (O(n^2 m) against the state machine's O(n^3 m) worst case on one lane).
                    * Both are optimal; with many 1e18 "forbidden" entries the optimum is degenerate and the two can undo
                    * different forbidden pairs: same matching cost, possibly another order of the left-over detections and
                    * hence of new tracking ids / coasting tracks.  Goldens exist for both (tests/golden/tracker_ref.json). */
    int baseline;  /* 1: Tracker_baseline (--refined_Kalman, utils/tracker_baseline.py:14-310): only (x, y) of a vertex observed,
                      plain scale average, association on raw centres against velocity-advanced track centres */
} cp_track_params;
size_t cp_track_state_bytes(int B, int cap);
size_t cp_track_workspace_bytes(int B, int K, int cap);
int cp_track_reset(cp_stream_t stream, void* state, int B, int cap);
int cp_track_step(cp_stream_t stream, const cp_track_params* params, const double* vmeta, const double* post, const int* count,
                  const double* det_pnp, int B, void* state, double* render_recs, void* workspace, size_t workspace_bytes);
/* dropped_out: HOST int32 [B], the sticky overflow counters above.  Copies 16 + 16 B bytes and synchronises `stream`. */
int cp_track_status(cp_stream_t stream, const void* state, int B, int* dropped_out);

/* The tracker's assignment on the HOST (no device work, no stream): replaces `linear_assignment(dist)` of tracker.py:157 for the
 * host tracker (centerpose_amd/lib/utils/tracker.py) with the very routine the device tracker runs.
 *   cost       HOST float64 [n_rows, n_cols] row-major (detections x tracks)
 *   solver     1 = scikit-learn 0.22.2's Munkres, 2 = scipy's rectangular LSAP (cp_track_params.hungarian)
 *   match_out  HOST int32 [n_rows]: column of each row, -1 for rows left out (min(n_rows, n_cols) rows get one) */
int cp_linear_assignment(const double* cost, int n_rows, int n_cols, int solver, int* match_out);

#ifdef __cplusplus
}
#endif
#endif /* CENTERPOSE_HIP_H */
