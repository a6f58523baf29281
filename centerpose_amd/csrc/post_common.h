// The scalar logic of post.hip's postprocess_kernel (post-process of one record, threshold filter + the reference's
// selection-sort soft-NMS), written so that the host compiler can build it too (tests/native/post_host.cpp,
// tests/test_post_logic_cpu.py pins it to the reference's own output, tests/golden/host_post.json).
#pragma once
#include <cmath>

// detection record (cp_common.h) and post-processed record (include/centerpose_hip.h) layouts, repeated here so that the
// host build needs neither header; an identical redefinition is legal, a drifting one is a compile error
#define CP_DET_BBOX 0
#define CP_DET_SCORE 4
#define CP_DET_KPS 5
#define CP_DET_CLS 21
#define CP_DET_SCALE 22
#define CP_DET_SCALE_UNC 25
#define CP_DET_TRACKING 28
#define CP_DET_TRACKING_HP 30
#define CP_DET_KPS_DISP_MEAN 46
#define CP_DET_KPS_DISP_STD 62
#define CP_DET_KPS_HM_MEAN 78
#define CP_DET_KPS_HM_STD 94
#define CP_DET_KPS_HM_HEIGHT 110
#define CP_DET_STRIDE 118
#define CP_POST_STRIDE 120

// No fused multiply-adds in either build: the arithmetic below restates numpy / Python expressions whose every operation
// rounds (hipcc's default, -ffp-contract=fast, would otherwise fuse e.g. the soft-NMS union area on the device, and the CPU
// test of this same source would not pin the device's bits).  Restored at the end of the header.
#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif
#ifdef __HIPCC__
#define POST_HD __host__ __device__ __forceinline__
#define POST_FMUL(a, b) __fmul_rn((a), (b))
#define POST_DADD(a, b) __dadd_rn((a), (b))
#define POST_DMUL(a, b) __dmul_rn((a), (b))
#else
#define POST_HD static inline
#define POST_FMUL(a, b) ((float)(a) * (float)(b))
#define POST_DADD(a, b) ((double)(a) + (double)(b))
#define POST_DMUL(a, b) ((double)(a) * (double)(b))
#endif

// record layout of the output (doubles), mirroring the dict of utils/post_process.py:21-58
// score 0 | cls 1 | obj_scale 2..4 | obj_scale_uncertainty 5..7 | kps_displacement_std 8..23 | bbox 24..27 | ct 28..29
// | kps 30..45 | tracking 46..47 | tracking_hp 48..63 | kps_displacement_mean 64..79 | kps_heatmap_mean 80..95
// | kps_heatmap_std 96..111 | kps_heatmap_height 112..119
POST_HD void xform(const double* t, float x, float y, double* ox, double* oy) {
    if (x == -10000.f && y == -10000.f) {
        *ox = -10000.0;
        *oy = -10000.0;
        return;
    }
    // np.dot(t, [x, y, 1]) in float64, no contraction
    *ox = POST_DADD(POST_DADD(POST_DMUL(t[0], (double)x), POST_DMUL(t[1], (double)y)), t[2]);
    *oy = POST_DADD(POST_DADD(POST_DMUL(t[3], (double)x), POST_DMUL(t[4], (double)y)), t[5]);
}

// one decoded record d[CP_DET_STRIDE] (float32) -> o[CP_POST_STRIDE] (float64): t = the image's inverse affine (6) + ratio
POST_HD void post_transform_record(const float* d, const double* t, float ratio, float div_scale, double* o) {
    const float coef = 0.32f;
    o[0] = (double)d[CP_DET_SCORE];
    o[1] = (double)(int)d[CP_DET_CLS];
    for (int i = 0; i < 3; ++i) {
        o[2 + i] = (double)d[CP_DET_SCALE + i];
        o[5 + i] = (double)d[CP_DET_SCALE_UNC + i];
    }
    for (int i = 0; i < 16; ++i) {
        o[8 + i] = (double)POST_FMUL(POST_FMUL(d[CP_DET_KPS_DISP_STD + i], ratio), coef);
        o[96 + i] = (double)POST_FMUL(POST_FMUL(d[CP_DET_KPS_HM_STD + i], ratio), coef);
        o[48 + i] = (double)POST_FMUL(d[CP_DET_TRACKING_HP + i], ratio);
    }
    for (int i = 0; i < 2; ++i) {
        xform(t, d[CP_DET_BBOX + 2 * i], d[CP_DET_BBOX + 2 * i + 1], &o[24 + 2 * i], &o[25 + 2 * i]);
        o[46 + i] = (double)POST_FMUL(d[CP_DET_TRACKING + i], ratio);
    }
    o[28] = (o[24] + o[26]) / 2;
    o[29] = (o[25] + o[27]) / 2;
    for (int i = 0; i < 8; ++i) {
        xform(t, d[CP_DET_KPS + 2 * i], d[CP_DET_KPS + 2 * i + 1], &o[30 + 2 * i], &o[31 + 2 * i]);
        xform(t, d[CP_DET_KPS_DISP_MEAN + 2 * i], d[CP_DET_KPS_DISP_MEAN + 2 * i + 1], &o[64 + 2 * i], &o[65 + 2 * i]);
        xform(t, d[CP_DET_KPS_HM_MEAN + 2 * i], d[CP_DET_KPS_HM_MEAN + 2 * i + 1], &o[80 + 2 * i], &o[81 + 2 * i]);
        o[112 + i] = (double)d[CP_DET_KPS_HM_HEIGHT + i];
    }
    if (div_scale != 1.f) {  // multi-scale testing (object_pose.py:171-176): float32 arrays divided by `scale`
        const int segs[7][2] = {{24, 4}, {30, 16}, {8, 16}, {46, 2}, {48, 16}, {64, 16}, {80, 16}};
        for (int sgi = 0; sgi < 7; ++sgi)
            for (int i = 0; i < segs[sgi][1]; ++i)
                o[segs[sgi][0] + i] = (double)((float)o[segs[sgi][0] + i] / div_scale);
    }
}

// keep score > vis_thresh in decode order, then (nms) the reference's Gaussian soft-NMS walk; arrays are rewritten in
// place, idx[r] = decode-order index of the r-th survivor; returns the number of survivors
POST_HD int post_filter_nms(double* s_score, double (*s_box)[4], int* s_idx, int K, double vis_thresh, int nms) {
    int N = 0;
    for (int k = 0; k < K; ++k)
        if (s_score[k] > vis_thresh) {
            s_score[N] = s_score[k];
            for (int i = 0; i < 4; ++i) s_box[N][i] = s_box[k][i];
            s_idx[N] = k;
            ++N;
        }
    if (nms) {
        const double sigma = 0.5, threshold = vis_thresh;
        for (int i = 0; i < N; ++i) {
            double maxscore = s_score[i];
            int maxpos = i;
            for (int pos = i + 1; pos < N; ++pos)
                if (maxscore < s_score[pos]) { maxscore = s_score[pos]; maxpos = pos; }
            // swap record i <-> maxpos (bbox, score and the rest of the dict = the record index)
            for (int q = 0; q < 4; ++q) { const double v = s_box[i][q]; s_box[i][q] = s_box[maxpos][q]; s_box[maxpos][q] = v; }
            { const double v = s_score[i]; s_score[i] = s_score[maxpos]; s_score[maxpos] = v; }
            { const int v = s_idx[i]; s_idx[i] = s_idx[maxpos]; s_idx[maxpos] = v; }
            const double tx1 = s_box[i][0], ty1 = s_box[i][1], tx2 = s_box[i][2], ty2 = s_box[i][3];
            int pos = i + 1;
            while (pos < N) {
                const double x1 = s_box[pos][0], y1 = s_box[pos][1], x2 = s_box[pos][2], y2 = s_box[pos][3];
                const double area = (x2 - x1 + 1) * (y2 - y1 + 1);
                const double iw = fmin(tx2, x2) - fmax(tx1, x1) + 1;
                if (iw > 0) {
                    const double ih = fmin(ty2, y2) - fmax(ty1, y1) + 1;
                    if (ih > 0) {
                        const double ua = (tx2 - tx1 + 1) * (ty2 - ty1 + 1) + area - iw * ih;
                        const double ov = iw * ih / ua;
                        const double weight = exp(-(ov * ov) / sigma);
                        s_score[pos] = weight * s_score[pos];
                        if (s_score[pos] < threshold) {
                            // the reference overwrites `pos` with the last record and swaps the dict payloads
                            for (int q = 0; q < 4; ++q) s_box[pos][q] = s_box[N - 1][q];
                            s_score[pos] = s_score[N - 1];
                            { const int v = s_idx[pos]; s_idx[pos] = s_idx[N - 1]; s_idx[N - 1] = v; }
                            --N;
                            --pos;
                        }
                    }
                }
                ++pos;
            }
        }
    }
    return N;
}

#if defined(__clang__) && defined(__HIPCC__)
#pragma clang fp contract(fast)  // hipcc's default again for whatever the including file defines after this header
#endif
