// CenterPoseTrack's per-frame track bookkeeping as plain scalar functions that compile for the device (track.hip) and
// for the host (tests/native/track_host.cpp builds the very same functions with g++ so that the logic can be checked
// against the reference-pinned Python tracker without a GPU).  No HIP, no libm beyond sqrt / exp / log / pow / floor.
//
// What is restated (paths relative to /root/reference/src/lib):
//   detectors/base_detector.py:501-544   Gaussian fusion of the displacement and heat-map keypoint estimates
//   utils/pnp/cuboid_pnp_shell.py:26-91  packaging of a PnP answer (3-D vertices in the camera frame, normalised
//                                        projections, category-dependent visibility rejects)
//   utils/tracker.py:112-302             Tracker.step: greedy association on displaced centres, matched / new / coasting
//                                        tracks, 32-state constant-velocity Kalman filter per track (filterpy's update
//                                        order: S = P + R, K = P S^-1, Joseph-form covariance), precision-weighted scale
//                                        pool, filtered read-out, confidence from the combined std
//   detectors/base_detector.py:150-388   which Gaussians are drawn into next frame's pre_hm / pre_hm_hp inputs
// The Kalman state is kept as eight independent 4 x 4 blocks (x, y, vx, vy of one vertex): F, H = I, Q = I and the
// diagonal R never couple two vertices, and numpy's dense 32 x 32 arithmetic produces exact zeros outside the blocks.
#pragma once
#include <math.h>

// no fused multiply-adds: the float32 passages below restate numpy expressions whose every operation rounds
#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

#ifdef __HIPCC__
#define CP_HD __host__ __device__ __forceinline__
#else
#define CP_HD static inline
#endif

// ---- one track (or one detection on its way to becoming one): CP_TRACK_STRIDE doubles ----
#define TR_ID 0
#define TR_AGE 1
#define TR_ACTIVE 2
#define TR_FLAGS 3          // bit 0: location / quaternion / projected_cuboid / kps_3d_cam / kps_pnp set (a PnP answer was
                            // packaged); 1: kps_pnp_kf / kps_3d_cam_kf / kps_ori_kf set; 2: in this frame's `boxes`;
                            // 3: kps_ori set (the detection came with a box, Tracker.step :116-123); 4: Kalman / pool valid
#define TR_POST 4           // the 120 doubles of the post-processed detection (CP_POST_* layout of centerpose_hip.h)
#define TR_FUS_MEAN 124     // [16]
#define TR_FUS_STD 140      // [16]
#define TR_LOC 156          // [3]   location
#define TR_QUAT 159         // [4]   quaternion_xyzw
#define TR_PROJ 163         // [16]  projected_cuboid
#define TR_KPS_PNP 179      // [18]  kps_pnp (centroid first, normalised)
#define TR_KPS_3D 197       // [27]  kps_3d_cam
#define TR_KPS_ORI 224      // [18]  kps_ori
#define TR_KF_X 242         // [32]
#define TR_KF_P 274         // [8][4][4]
#define TR_POOL_PREC 402    // [3]
#define TR_POOL_ACC 405     // [3]
#define TR_POOL_N 408
#define TR_MEAN_KF 409      // [16]  kps_mean_kf (-10000 where the confidence is below 0.15)
#define TR_STD_KF 425       // [16]
#define TR_SCALE_KF 441     // [3]
#define TR_SCALE_UNC_KF 444 // [3]
#define TR_CONF 447         // [8]
#define TR_KPS_PNP_KF 455    // [18]
#define TR_KPS_3D_KF 473     // [27]
#define TR_KPS_ORI_KF 500    // [18]
#define TR_SRC 518          // scratch: index of the previous-frame track a matched detection continues (-1: new)
#define CP_TRACK_STRIDE 520

// offsets inside the post-processed record (include/centerpose_hip.h: cp_postprocess)
#define PO_SCORE 0
#define PO_CLS 1
#define PO_SCALE 2
#define PO_SCALE_UNC 5
#define PO_DISP_STD 8
#define PO_BBOX 24
#define PO_CT 28
#define PO_KPS 30
#define PO_TRACKING 46
#define PO_TRACKING_HP 48
#define PO_DISP_MEAN 64
#define PO_HM_MEAN 80
#define PO_HM_STD 96
#define PO_HM_HEIGHT 112

struct TrackParams {
    double new_thresh, pre_thresh, R, conf_lo, conf_hi;
    int max_age, kalman, scale_pool, use_pnp, hps_uncertainty, show_axes;
    int cat_rule;  // visibility reject of cuboid_pnp_shell.py:70-84: 0 = camera / bottle / cup (3 points), 1 = book / chair /
                   // cereal_box (6 points), 2 = bike / laptop / shoe (none)
    int render_hm_mode, render_hmhp_mode, pre_hm, pre_hm_hp;
    int K, cap;
    int hungarian;  // association by optimal assignment (tracker.py:154-170) instead of the greedy walk
    int baseline;   // Tracker_baseline (--refined_Kalman, utils/tracker_baseline.py): position-only filter, plain scale average
};

// per video: trans_input (2 x 3, row-major) | width height inp_w inp_h | fx fy cx cy
#define VM_TIN 0
#define VM_WIDTH 6
#define VM_HEIGHT 7
#define VM_INP_W 8
#define VM_INP_H 9
#define VM_CAM 10
#define CP_VMETA_STRIDE 16

// ---------------------------------------------------------------------------------------------------------------------
// Rectangular linear sum assignment exactly as scipy.optimize.linear_sum_assignment computes it (the reference calls
// sklearn 0.22's linear_assignment, tracker.py:157; sklearn is absent and the goldens come from scipy's solver, which returns
// the same optimum): Crouse's shortest augmenting path (2016) with scipy's conventions -- rows are the shorter side (a tall
// matrix is transposed), the candidate list is filled in reverse, ties prefer a column that ends the path, the duals are
// updated after every augmentation -- and the same float64 operations in the same order, so that even the assignments among
// 1e18-"forbidden" pairs (whose duals swallow the low bits of the real costs) come out as scipy's do.
// cost(i, j): det i x track j; nd x nt; match[i] = track of det i or -1.  Work space: u / spc [LS] doubles, v [LS] doubles,
// path / col4row / row4col / remaining [LS] ints, SR / SC [LS] bytes with LS >= max(nd, nt).
struct TrkLsapWork {
    double* u; double* v; double* spc;
    int* path; int* col4row; int* row4col; int* remaining;
    unsigned char* SR; unsigned char* SC;
};
template <class Cost>
CP_HD void trk_lsap(const Cost& cost, int nd, int nt, int* match, const TrkLsapWork& W) {
    for (int i = 0; i < nd; ++i) match[i] = -1;
    if (nd == 0 || nt == 0) return;
    const bool tr = nt < nd;                 // tall: rows = tracks, columns = detections
    const int nr = tr ? nt : nd, nc = tr ? nd : nt;
    auto C = [&](int i, int j) -> double { return tr ? cost(j, i) : cost(i, j); };
    for (int i = 0; i < nr; ++i) { W.u[i] = 0.0; W.col4row[i] = -1; }
    for (int j = 0; j < nc; ++j) { W.v[j] = 0.0; W.path[j] = -1; W.row4col[j] = -1; }
    const double INF = __builtin_huge_val();
    for (int cur = 0; cur < nr; ++cur) {
        // ---- shortest augmenting path from row `cur` ----
        double minVal = 0.0;
        int num_remaining = nc;
        for (int it = 0; it < nc; ++it) W.remaining[it] = nc - it - 1;
        for (int i = 0; i < nr; ++i) W.SR[i] = 0;
        for (int j = 0; j < nc; ++j) { W.SC[j] = 0; W.spc[j] = INF; }
        int sink = -1, i = cur;
        while (sink == -1) {
            int index = -1;
            double lowest = INF;
            W.SR[i] = 1;
            for (int it = 0; it < num_remaining; ++it) {
                const int j = W.remaining[it];
                const double r = minVal + C(i, j) - W.u[i] - W.v[j];
                if (r < W.spc[j]) { W.path[j] = i; W.spc[j] = r; }
                if (W.spc[j] < lowest || (W.spc[j] == lowest && W.row4col[j] == -1)) { lowest = W.spc[j]; index = it; }
            }
            minVal = lowest;
            if (!(minVal < INF)) return;  // infeasible (cannot happen: every cost is finite)
            const int j = W.remaining[index];
            if (W.row4col[j] == -1) sink = j;
            else i = W.row4col[j];
            W.SC[j] = 1;
            W.remaining[index] = W.remaining[--num_remaining];
        }
        // ---- dual variables ----
        W.u[cur] += minVal;
        for (int r = 0; r < nr; ++r)
            if (W.SR[r] && r != cur) W.u[r] += minVal - W.spc[W.col4row[r]];
        for (int j = 0; j < nc; ++j)
            if (W.SC[j]) W.v[j] -= minVal - W.spc[j];
        // ---- augment ----
        int j = sink;
        for (;;) {
            const int r = W.path[j];
            W.row4col[j] = r;
            const int t = W.col4row[r];
            W.col4row[r] = j;
            j = t;
            if (r == cur) break;
        }
    }
    for (int r = 0; r < nr; ++r) {
        if (tr) match[W.col4row[r]] = r;  // row = track, its column = the detection
        else match[r] = W.col4row[r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The assignment the reference really calls: sklearn.utils.linear_assignment_.linear_assignment of scikit-learn 0.22.2
// (tracker.py:6,157; requirements.txt:13) -- the Kuhn-Munkres state machine (row reduction and greedy starring, cover the
// starred columns, prime uncovered zeros, augment along the prime / star path, add the smallest uncovered value to covered
// rows and subtract it from uncovered columns).  The module is absent from every scikit-learn since 0.23; oracle/munkres.py
// restates it in numpy and this is the same restatement in scalar form for the device and the host build: the SAME float64
// operations on the same elements in the same order (row minimum subtracted; in the adjustment step first `+= minval` on
// covered rows, then `-= minval` on uncovered columns -- two roundings where both apply), zeros are `== 0` exactly, and every
// search returns the FIRST hit in numpy's order (row-major for the uncovered zero, lowest index for the star / prime of a
// line), because which optimum comes out of a degenerate problem -- the tracker's matrices are full of 1e18 -- depends on all
// of that, and new tracking ids are handed out in the order of the left-over detections.
// cost(i, j): det i x track j; nd x nt; match[i] = track of det i or -1 (min(nd, nt) detections / tracks get a partner).
// Work space for n = min(nd, nt) rows and m = max(nd, nt) columns: C [n m] doubles, marked [n m] bytes (0 / 1 star / 2 prime),
// row_unc [n] / col_unc [m] bytes, path [2 (n + m)] ints.  Cost: O(n m) per search and per adjustment, O(n^2 m) .. O(n^3 m)
// in all on one lane -- microseconds for a frame's ten detections, long for a hundred mutually tied ones; the scipy form
// above (hungarian = 2) is the fast one.
struct TrkMunkresWork {
    double* C; unsigned char* marked; unsigned char* row_unc; unsigned char* col_unc; int* path;
};
template <class Cost>
CP_HD void trk_munkres(const Cost& cost, int nd, int nt, int* match, const TrkMunkresWork& W) {
    for (int i = 0; i < nd; ++i) match[i] = -1;
    if (nd == 0 || nt == 0) return;
    const bool tr = nt < nd;  // more rows (detections) than columns: the transpose is solved, the pairs swapped back
    const int n = tr ? nt : nd, m = tr ? nd : nt;
    double* C = W.C;
    unsigned char* mk = W.marked;
    // ---- step 1: row reduction; star zeros whose row and column hold no star yet (row-major scan) ----
    for (int i = 0; i < n; ++i) {
        double lo = tr ? cost(0, i) : cost(i, 0);
        for (int j = 0; j < m; ++j) {
            const double c = tr ? cost(j, i) : cost(i, j);
            C[i * m + j] = c;
            if (c < lo) lo = c;
        }
        for (int j = 0; j < m; ++j) C[i * m + j] -= lo;
    }
    for (int i = 0; i < n; ++i) W.row_unc[i] = 1;
    for (int j = 0; j < m; ++j) W.col_unc[j] = 1;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            mk[i * m + j] = 0;
            if (C[i * m + j] == 0.0 && W.col_unc[j] && W.row_unc[i]) {
                mk[i * m + j] = 1;
                W.col_unc[j] = 0;
                W.row_unc[i] = 0;
            }
        }
    for (;;) {
        // ---- step 3: covers cleared, starred columns covered; n stars = a complete assignment ----
        for (int i = 0; i < n; ++i) W.row_unc[i] = 1;
        for (int j = 0; j < m; ++j) W.col_unc[j] = 1;
        int stars = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < m; ++j)
                if (mk[i * m + j] == 1) { W.col_unc[j] = 0; ++stars; }
        if (stars >= n) break;
        int zr = -1, zc = -1;
        for (;;) {
            // ---- step 4: prime the first uncovered zero (row-major) until one has no star in its row ----
            for (;;) {
                int row = -1, col = -1;
                for (int i = 0; i < n && row < 0; ++i) {
                    if (!W.row_unc[i]) continue;
                    for (int j = 0; j < m; ++j)
                        if (W.col_unc[j] && C[i * m + j] == 0.0) { row = i; col = j; break; }
                }
                if (row < 0) break;  // none left: adjust the matrix
                mk[row * m + col] = 2;
                int star = -1;
                for (int j = 0; j < m; ++j)
                    if (mk[row * m + j] == 1) { star = j; break; }
                if (star < 0) { zr = row; zc = col; break; }
                W.row_unc[row] = 0;
                W.col_unc[star] = 1;
            }
            if (zr >= 0) break;
            // ---- step 6: smallest uncovered value: + on covered rows, then - on uncovered columns ----
            bool any_r = false, any_c = false;
            for (int i = 0; i < n; ++i) any_r = any_r || W.row_unc[i];
            for (int j = 0; j < m; ++j) any_c = any_c || W.col_unc[j];
            if (any_r && any_c) {
                double minval = 0.0;
                bool have = false;
                for (int i = 0; i < n; ++i) {
                    if (!W.row_unc[i]) continue;
                    for (int j = 0; j < m; ++j)
                        if (W.col_unc[j] && (!have || C[i * m + j] < minval)) { minval = C[i * m + j]; have = true; }
                }
                for (int i = 0; i < n; ++i)
                    if (!W.row_unc[i])
                        for (int j = 0; j < m; ++j) C[i * m + j] += minval;
                for (int j = 0; j < m; ++j)
                    if (W.col_unc[j])
                        for (int i = 0; i < n; ++i) C[i * m + j] -= minval;
            }
        }
        // ---- step 5: alternating path from the primed zero; stars on it go, its primes become stars; primes erased ----
        int count = 0;
        W.path[0] = zr;
        W.path[1] = zc;
        for (;;) {
            const int pc = W.path[2 * count + 1];
            int row = -1;
            for (int i = 0; i < n; ++i)
                if (mk[i * m + pc] == 1) { row = i; break; }
            if (row < 0) break;
            ++count;
            W.path[2 * count] = row;
            W.path[2 * count + 1] = pc;
            int pcol = -1;
            for (int j = 0; j < m; ++j)
                if (mk[row * m + j] == 2) { pcol = j; break; }
            ++count;
            W.path[2 * count] = row;
            W.path[2 * count + 1] = pcol;  // (a star on the path always has a prime in its row)
        }
        for (int k = 0; k <= count; ++k) {
            unsigned char& e = mk[W.path[2 * k] * m + W.path[2 * k + 1]];
            e = e == 1 ? 0 : 1;
        }
        for (int i = 0; i < n * m; ++i)
            if (mk[i] == 2) mk[i] = 0;
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j)
            if (mk[i * m + j] == 1) {
                if (tr) match[j] = i;  // row = track, column = detection
                else match[i] = j;
            }
}

// ---------------------------------------------------------------------------------------------------------------------
// Gaussian fusion (base_detector.py:503-536), hps_uncertainty branch and the fixed-variance branch.  The reference's standard
// deviations are float32 numpy scalars and its means float64 ones, so `std ** -2`, their sum and `** -0.5` are float32
// operations (numpy keeps float32 ** python-int / python-float in float32) and only the products with the means promote to
// float64; `hs / np.sqrt(2)` is float64 (np.sqrt(2) is a float64 scalar).  Restated with the same widths: the fused values
// then agree with the reference to the last float32 digit of powf instead of ~1e-7.
CP_HD void trk_fuse(const double* post, int hps_uncertainty, double* mean, double* std) {
    for (int i = 0; i < 16; ++i) {
        const double dm = post[PO_DISP_MEAN + i], hm = post[PO_HM_MEAN + i];
        const float ds = (float)post[PO_DISP_STD + i], hs = (float)post[PO_HM_STD + i];
        const bool missing = hm < 0 || hs < 0;
        double s, m;
        if (hps_uncertainty) {
            if (missing) { s = post[PO_DISP_STD + i]; m = dm; }  // (passed through as stored)
            else {
                const float a = powf(ds, -2.0f), b = powf(hs, -2.0f);
                const float sf = powf(a + b, -0.5f);
                s = (double)sf;
                m = (double)(sf * sf) * ((double)a * dm + (double)b * hm);
            }
        } else if (missing) { s = 20.0; m = dm; }
        else {
            const float b = powf(hs, -2.0f);
            s = (double)hs / sqrt(2.0);
            m = s * s * ((double)b * dm + (double)b * hm);
        }
        mean[i] = m;
        std[i] = s;
    }
}

CP_HD void trk_quat_to_matrix(const double* q, double* R) {  // scipy Rotation.from_quat(q).as_matrix() (normalises)
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

// Everything pnp_shell does after a successful solve (cuboid_pnp_shell.py:26-91).  `row`: the 40 doubles of
// cp_pnp_solve; `scale3`: the (un-normalised) relative size handed to pnp_shell; `kps16`: the detection's `kps`.
// Writes location / quaternion / projected_cuboid / kps_3d_cam / kps_pnp into `t` (the reference sets them before its
// visibility rejects) and kps_ori into `ori`; returns 1 when the detection survives the rejects.
CP_HD int trk_finish(const TrackParams& P, const double* vm, const double* row, const double* scale3, const double* kps16,
                     double* t, double* ori, int opencv_frame) {
    // OPENCV_RETURN of pnp_shell: opt.show_axes, except for Tracker_baseline's filtered PnP (default frame, :276)
    const double* loc = opencv_frame ? row + 4 : row + 28;
    const double* quat = opencv_frame ? row + 24 : row + 31;
    for (int i = 0; i < 3; ++i) t[TR_LOC + i] = loc[i];
    for (int i = 0; i < 4; ++i) t[TR_QUAT + i] = quat[i];
    for (int i = 0; i < 16; ++i) t[TR_PROJ + i] = row[8 + i];
    // cuboid vertices of size scale / scale_y (cuboid_objectron.py:80-110), posed
    const double w = scale3[0] / scale3[1], h = 1.0, d = scale3[2] / scale3[1];
    double R[9];
    trk_quat_to_matrix(quat, R);
    double cm[3] = {0, 0, 0};
    for (int v = 0; v < 8; ++v) {
        const double x = (v & 4) ? w / 2 : -w / 2, y = (v & 2) ? h / 2 : -h / 2, z = (v & 1) ? d / 2 : -d / 2;
        for (int r = 0; r < 3; ++r) {
            const double c = R[3 * r] * x + R[3 * r + 1] * y + R[3 * r + 2] * z + loc[r];
            t[TR_KPS_3D + 3 * (v + 1) + r] = c;
            cm[r] += c;
        }
    }
    for (int r = 0; r < 3; ++r) t[TR_KPS_3D + r] = cm[r] / 8;
    // projected points, centroid first, normalised by the image size
    const double W0 = vm[VM_WIDTH], H0 = vm[VM_HEIGHT];
    double mx = 0, my = 0;
    for (int v = 0; v < 8; ++v) { mx += row[8 + 2 * v]; my += row[9 + 2 * v]; }
    t[TR_KPS_PNP] = (mx / 8) / W0;
    t[TR_KPS_PNP + 1] = (my / 8) / H0;
    for (int v = 0; v < 8; ++v) {
        t[TR_KPS_PNP + 2 * (v + 1)] = row[8 + 2 * v] / W0;
        t[TR_KPS_PNP + 2 * (v + 1) + 1] = row[9 + 2 * v] / H0;
    }
    if (P.cat_rule != 2) {
        const int thr = P.cat_rule == 1 ? 6 : 3;
        int n_out = 0;
        for (int v = 0; v < 9; ++v) {
            const double px = t[TR_KPS_PNP + 2 * v], py = t[TR_KPS_PNP + 2 * v + 1];
            if (px < 0 || px > 1 || py < 0 || py > 1) ++n_out;
        }
        if (n_out >= thr) return 0;
    }
    {
        const double px = t[TR_KPS_PNP], py = t[TR_KPS_PNP + 1];
        if (!(px > 0 && px < 1 && py > 0 && py < 1)) return 0;
    }
    mx = my = 0;
    for (int v = 0; v < 8; ++v) { mx += kps16[2 * v]; my += kps16[2 * v + 1]; }
    ori[0] = (mx / 8) / W0;
    ori[1] = (my / 8) / H0;
    for (int v = 0; v < 8; ++v) {
        ori[2 * (v + 1)] = kps16[2 * v] / W0;
        ori[2 * (v + 1) + 1] = kps16[2 * v + 1] / H0;
    }
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// 4 x 4 helpers (row-major)
CP_HD void m4_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
            C[4 * i + j] = s;
        }
}
CP_HD void m4_mul_bt(const double* A, const double* B, double* C) {  // A B^T
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * j + k];
            C[4 * i + j] = s;
        }
}
// inverse by Gauss-Jordan with partial pivoting (what LAPACK's getrf / getri amount to on a well-conditioned block)
CP_HD void m4_inv(const double* A, double* X) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = A[4 * i + j]; a[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (piv != c)
            for (int j = 0; j < 8; ++j) { const double tmp = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = tmp; }
        const double inv = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= inv;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = a[r][c];
                if (f != 0.0)
                    for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
            }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) X[4 * i + j] = a[i][4 + j];
}

// observation (x, y, vx, vy) and its noise for vertex v (tracker.py:55-79): position = fused keypoint, velocity = minus
// the predicted displacement to the previous frame; R = diag(std_x^2, std_y^2, opt.R, opt.R)
CP_HD void trk_obs(const TrackParams& P, const double* t, int v, double* z, double* r) {
    z[0] = t[TR_FUS_MEAN + 2 * v];
    z[1] = t[TR_FUS_MEAN + 2 * v + 1];
    z[2] = -t[TR_POST + PO_TRACKING_HP + 2 * v];
    z[3] = -t[TR_POST + PO_TRACKING_HP + 2 * v + 1];
    const double sx = t[TR_FUS_STD + 2 * v], sy = t[TR_FUS_STD + 2 * v + 1];
    r[0] = sx * sx;
    r[1] = sy * sy;
    r[2] = r[3] = P.R;
}

CP_HD void m2_inv(const double* A, double* X) {  // the same elimination on a 2 x 2 block
    double a[2][4] = {{A[0], A[1], 1.0, 0.0}, {A[2], A[3], 0.0, 1.0}};
    for (int c = 0; c < 2; ++c) {
        const int piv = (c == 0 && fabs(a[1][0]) > fabs(a[0][0])) ? 1 : c;
        if (piv != c)
            for (int j = 0; j < 4; ++j) { const double tmp = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = tmp; }
        const double inv = 1.0 / a[c][c];
        for (int j = 0; j < 4; ++j) a[c][j] *= inv;
        const int r = 1 - c;
        const double f = a[r][c];
        if (f != 0.0)
            for (int j = 0; j < 4; ++j) a[r][j] -= f * a[c][j];
    }
    X[0] = a[0][2]; X[1] = a[0][3]; X[2] = a[1][2]; X[3] = a[1][3];
}

// Tracker_baseline.init_kf (tracker_baseline.py:54-78): only (x, y) observed; x = (mean_x, mean_y, 0, 0); P = I with its
// position block set to [[vx, vy], [vx, vy]] (the reference assigns a 2-vector to the 2 x 2 block, which broadcasts over rows)
CP_HD void trk_kf_init_base(double* t) {
    for (int v = 0; v < 8; ++v) {
        const double sx = t[TR_FUS_STD + 2 * v], sy = t[TR_FUS_STD + 2 * v + 1];
        double* Pb = t + TR_KF_P + 16 * v;
        for (int i = 0; i < 16; ++i) Pb[i] = 0.0;
        for (int i = 0; i < 4; ++i) Pb[5 * i] = 1.0;
        Pb[0] = sx * sx; Pb[1] = sy * sy;
        Pb[4] = sx * sx; Pb[5] = sy * sy;
        t[TR_KF_X + 4 * v] = t[TR_FUS_MEAN + 2 * v];
        t[TR_KF_X + 4 * v + 1] = t[TR_FUS_MEAN + 2 * v + 1];
        t[TR_KF_X + 4 * v + 2] = t[TR_KF_X + 4 * v + 3] = 0.0;
    }
}

// Tracker_baseline: predict() then update(z, R) with H = [I2 0] (tracker_baseline.py:80-92; filterpy's Joseph form):
// y = z - H x, S = H P H^T + R, K = P H^T S^-1, x += K y, P = (I - K H) P (I - K H)^T + K R K^T
CP_HD void trk_kf_step_base(double* t) {
    const double F[16] = {1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int v = 0; v < 8; ++v) {
        double* x = t + TR_KF_X + 4 * v;
        double* Pb = t + TR_KF_P + 16 * v;
        double xp[4], T[16], Pp[16];
        const double z[2] = {t[TR_FUS_MEAN + 2 * v], t[TR_FUS_MEAN + 2 * v + 1]};
        const double sx = t[TR_FUS_STD + 2 * v], sy = t[TR_FUS_STD + 2 * v + 1];
        const double r[2] = {sx * sx, sy * sy};
        for (int i = 0; i < 4; ++i) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += F[4 * i + k] * x[k];
            xp[i] = s;
        }
        m4_mul(F, Pb, T);
        m4_mul_bt(T, F, Pp);
        for (int i = 0; i < 4; ++i) Pp[5 * i] += 1.0;
        const double S[4] = {Pp[0] + r[0], Pp[1], Pp[4], Pp[5] + r[1]};
        double Si[4], Kg[8];  // K: 4 x 2 = (P H^T) S^-1, P H^T = the first two columns of P
        m2_inv(S, Si);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j) Kg[2 * i + j] = Pp[4 * i] * Si[j] + Pp[4 * i + 1] * Si[2 + j];
        const double y0 = z[0] - xp[0], y1 = z[1] - xp[1];
        for (int i = 0; i < 4; ++i) x[i] = xp[i] + (Kg[2 * i] * y0 + Kg[2 * i + 1] * y1);
        double IK[16], A[16], B[16];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) IK[4 * i + j] = (i == j ? 1.0 : 0.0) - (j < 2 ? Kg[2 * i + j] : 0.0);
        m4_mul(IK, Pp, A);
        m4_mul_bt(A, IK, B);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) Pb[4 * i + j] = B[4 * i + j] + (Kg[2 * i] * r[0] * Kg[2 * j] + Kg[2 * i + 1] * r[1] * Kg[2 * j + 1]);
    }
}

CP_HD void trk_kf_init(const TrackParams& P, double* t) {  // init_kf: x = observation, P = R
    if (P.baseline) { trk_kf_init_base(t); return; }
    for (int v = 0; v < 8; ++v) {
        double z[4], r[4];
        trk_obs(P, t, v, z, r);
        double* Pb = t + TR_KF_P + 16 * v;
        for (int i = 0; i < 16; ++i) Pb[i] = 0.0;
        for (int i = 0; i < 4; ++i) {
            t[TR_KF_X + 4 * v + i] = z[i];
            Pb[5 * i] = r[i];
        }
    }
}

// filterpy predict() then update(z, R) on the state in `t` (already holding the previous frame's x and P)
CP_HD void trk_kf_step(const TrackParams& P, double* t) {
    if (P.baseline) { trk_kf_step_base(t); return; }
    const double F[16] = {1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int v = 0; v < 8; ++v) {
        double* x = t + TR_KF_X + 4 * v;
        double* Pb = t + TR_KF_P + 16 * v;
        double z[4], r[4], xp[4], T[16], Pp[16];
        trk_obs(P, t, v, z, r);
        // predict: x = F x, P = F P F^T + I
        for (int i = 0; i < 4; ++i) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += F[4 * i + k] * x[k];
            xp[i] = s;
        }
        m4_mul(F, Pb, T);
        m4_mul_bt(T, F, Pp);
        for (int i = 0; i < 4; ++i) Pp[5 * i] += 1.0;
        // update (H = I): y = z - x, S = P + R, K = P S^-1, x += K y, P = (I - K) P (I - K)^T + K R K^T
        double S[16], Si[16], Kg[16], IK[16], A[16], B[16], KR[16];
        for (int i = 0; i < 16; ++i) S[i] = Pp[i];
        for (int i = 0; i < 4; ++i) S[5 * i] += r[i];
        m4_inv(S, Si);
        m4_mul(Pp, Si, Kg);
        for (int i = 0; i < 4; ++i) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += Kg[4 * i + k] * (z[k] - xp[k]);
            x[i] = xp[i] + s;
        }
        for (int i = 0; i < 16; ++i) IK[i] = -Kg[i];
        for (int i = 0; i < 4; ++i) IK[5 * i] += 1.0;
        m4_mul(IK, Pp, A);
        m4_mul_bt(A, IK, B);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) KR[4 * i + j] = Kg[4 * i + j] * r[j];
        m4_mul_bt(KR, Kg, A);
        for (int i = 0; i < 16; ++i) Pb[i] = B[i] + A[i];
    }
}

// scale pool (tracker.py:98-110): one more (mean, uncertainty) sample; the running sums equal the reference's re-summation
// of its list in the same order
CP_HD void trk_pool_add(const TrackParams& P, double* t) {
    for (int i = 0; i < 3; ++i) {
        const double u = t[TR_POST + PO_SCALE_UNC + i];
        const double w = P.baseline ? 1.0 : 1.0 / (u * u);  // Tracker_baseline: a plain average (tracker_baseline.py:94-101)
        t[TR_POOL_PREC + i] += w;
        t[TR_POOL_ACC + i] += w * t[TR_POST + PO_SCALE + i];
    }
    t[TR_POOL_N] += 1.0;
}

CP_HD double trk_conf(const TrackParams& P, double comb) {  // tracker.py:254-262 / base_detector.py:277-305
    const double decay = exp(log(0.15) / (P.conf_lo - P.conf_hi));
    const double c = 1.0 - pow(decay, comb - P.conf_hi);
    return c > 0 ? c : 0.0;
}

// Filter read-out of one track (tracker.py:238-275): kps_mean_kf / kps_std_kf / confidences / fused scale.  Returns the
// 8 image points and the relative size the filtered PnP is asked to explain (points of low confidence = -10000).
CP_HD void trk_readout(const TrackParams& P, double* t, double* pts16, double* scale3) {
    if (P.kalman) {
        for (int v = 0; v < 8; ++v) {
            double vx = t[TR_KF_P + 16 * v + 0], vy = t[TR_KF_P + 16 * v + 5];
            if (P.baseline) {
                // the reference reads P[2v, 2v] and P[2v+1, 2v+1] of the 32 x 32 matrix here (tracker_baseline.py:251-254):
                // diagonal entries 2v, 2v+1 -- position variances of vertex v/2 for even v, velocity variances for odd v
                const int k0 = 2 * v, k1 = 2 * v + 1;
                vx = t[TR_KF_P + 16 * (k0 >> 2) + 5 * (k0 & 3)];
                vy = t[TR_KF_P + 16 * (k1 >> 2) + 5 * (k1 & 3)];
            }
            t[TR_MEAN_KF + 2 * v] = t[TR_KF_X + 4 * v];
            t[TR_MEAN_KF + 2 * v + 1] = t[TR_KF_X + 4 * v + 1];
            t[TR_STD_KF + 2 * v] = sqrt(vx);
            t[TR_STD_KF + 2 * v + 1] = sqrt(vy);
            const double c = trk_conf(P, sqrt(vx + vy));
            t[TR_CONF + v] = c;
            if (c < 0.15) t[TR_MEAN_KF + 2 * v] = t[TR_MEAN_KF + 2 * v + 1] = -10000.0;
        }
        for (int i = 0; i < 16; ++i) pts16[i] = t[TR_MEAN_KF + i];
    } else {
        for (int v = 0; v < 8; ++v) t[TR_CONF + v] = 0.0;  // `conf` stays an empty list: its mean is 0
        for (int i = 0; i < 16; ++i) pts16[i] = t[TR_POST + PO_KPS + i];
    }
    if (P.scale_pool) {
        for (int i = 0; i < 3; ++i) {
            if (P.baseline) {  // mean of the samples, placeholder uncertainty (tracker_baseline.py:97-101, :269)
                t[TR_SCALE_KF + i] = t[TR_POOL_ACC + i] / t[TR_POOL_N];
                t[TR_SCALE_UNC_KF + i] = 0.0;
            } else {
                const double std = 1.0 / sqrt(t[TR_POOL_PREC + i]);
                t[TR_SCALE_KF + i] = t[TR_POOL_ACC + i] * (std * std);
                t[TR_SCALE_UNC_KF + i] = std;
            }
            scale3[i] = t[TR_SCALE_KF + i];
        }
    } else {
        for (int i = 0; i < 3; ++i) scale3[i] = t[TR_POST + PO_SCALE + i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// float32 helpers for the places where the reference computes in numpy float32
CP_HD float f32(double x) { return (float)x; }

// affine_transform (utils/image.py:71-74): float32 (x, y, 1) through the float64 2 x 3 matrix
CP_HD void trk_affine(const double* T, double x, double y, double* ox, double* oy) {
    const double fx = (double)f32(x), fy = (double)f32(y);
    *ox = T[0] * fx + T[1] * fy + T[2] * 1.0;
    *oy = T[3] * fx + T[4] * fy + T[5] * 1.0;
}

CP_HD double trk_gaussian_radius(double h, double w) {  // utils/image.py:103-123, min_overlap 0.7
    const double o = 0.7;
    const double b1 = h + w, c1 = w * h * (1 - o) / (1 + o);
    const double r1 = (b1 + sqrt(b1 * b1 - 4 * 1 * c1)) / 2;
    const double b2 = 2 * (h + w), c2 = (1 - o) * w * h;
    const double r2 = (b2 + sqrt(b2 * b2 - 4 * 4 * c2)) / 2;
    const double a3 = 4 * o, b3 = -2 * o * (h + w), c3 = (o - 1) * w * h;
    const double r3 = (b3 + sqrt(b3 * b3 - 4 * a3 * c3)) / 2;
    double r = r1 < r2 ? r1 : r2;
    return r < r3 ? r : r3;
}

// The Gaussians `_get_additional_inputs` draws for one track (base_detector.py:150-388, the 'pnp' / 'kps' branches of the
// inference configuration): rec[0] = the centre blob for pre_hm, rec[1..8] = the vertex blobs for pre_hm_hp; a record is
// (channel, x, y, radius, k) with channel = -1 when nothing is drawn.  hm_plane / hp_plane0: planes of this video.
CP_HD void trk_render_records(const TrackParams& P, const double* vm, const double* t, int hm_plane, int hp_plane0,
                              double* rec) {
    for (int i = 0; i < 9; ++i) rec[5 * i] = -1.0;
    const double score = t[TR_POST + PO_SCORE];
    if (score < P.pre_thresh) return;
    const double* T = vm + VM_TIN;
    const double iw = vm[VM_INP_W], ih = vm[VM_INP_H], W0 = vm[VM_WIDTH], H0 = vm[VM_HEIGHT];
    // _trans_bbox: float32 box, both corners through trans_input, clipped to the input
    float bx[4];
    {
        double ox, oy;
        trk_affine(T, (double)f32(t[TR_POST + PO_BBOX]), (double)f32(t[TR_POST + PO_BBOX + 1]), &ox, &oy);
        bx[0] = f32(ox); bx[1] = f32(oy);
        trk_affine(T, (double)f32(t[TR_POST + PO_BBOX + 2]), (double)f32(t[TR_POST + PO_BBOX + 3]), &ox, &oy);
        bx[2] = f32(ox); bx[3] = f32(oy);
        const float xm = (float)(iw - 1), ym = (float)(ih - 1);
        bx[0] = bx[0] < 0.f ? 0.f : (bx[0] > xm ? xm : bx[0]);
        bx[2] = bx[2] < 0.f ? 0.f : (bx[2] > xm ? xm : bx[2]);
        bx[1] = bx[1] < 0.f ? 0.f : (bx[1] > ym ? ym : bx[1]);
        bx[3] = bx[3] < 0.f ? 0.f : (bx[3] > ym ? ym : bx[3]);
    }
    const float h = bx[3] - bx[1], w = bx[2] - bx[0];
    if (!(h > 0.f && w > 0.f)) return;
    const double rr = trk_gaussian_radius(ceil((double)h), ceil((double)w));
    const int radius = (int)rr > 0 ? (int)rr : 0;
    const int ctx = (int)((bx[0] + bx[2]) / 2.f), cty = (int)((bx[1] + bx[3]) / 2.f);
    if (P.pre_hm && (P.render_hm_mode == 0 || P.render_hm_mode == 1)) {
        rec[0] = hm_plane; rec[1] = ctx; rec[2] = cty; rec[3] = radius;
        rec[4] = P.render_hm_mode == 1 ? score : 1.0;
    }
    if (!P.pre_hm_hp) return;
    const int flags = (int)t[TR_FLAGS];
    // which vertex estimate is re-drawn (normalised image coordinates), :253-268
    double src[16];
    int nsrc = 8;
    const bool pnp_mode = P.use_pnp != 0;
    if (!pnp_mode) {
        for (int i = 0; i < 16; ++i) src[i] = t[TR_POST + PO_KPS + i];
    } else {
        if (P.render_hmhp_mode == 0 || P.render_hmhp_mode == 1) {
            if (!(flags & 8)) return;  // no kps_ori on this record (the reference would raise a KeyError)
            for (int i = 0; i < 16; ++i) src[i] = t[TR_KPS_ORI + 2 + i];
        } else if (P.kalman || P.scale_pool) {
            if (flags & 2) {
                for (int i = 0; i < 16; ++i) src[i] = t[TR_KPS_PNP_KF + 2 + i];
            } else {  // kps_mean_kf[1:]: seven rows (the reference drops the first vertex here)
                nsrc = 7;
                for (int i = 0; i < 14; ++i) src[i] = t[TR_MEAN_KF + 2 + i];
            }
        } else if (flags & 1) {
            for (int i = 0; i < 16; ++i) src[i] = t[TR_KPS_PNP + 2 + i];
        } else {
            for (int i = 0; i < 16; ++i) src[i] = 0.0;
        }
        for (int v = 0; v < nsrc; ++v) { src[2 * v] *= W0; src[2 * v + 1] *= H0; }
    }
    for (int j = 0; j < 8; ++j) {
        // COCO-style visibility in an int64 table: floats truncate toward zero
        long long px = 0, py = 0;
        int vis = 0;
        if (j < nsrc) {
            const double qx = src[2 * j], qy = src[2 * j + 1];
            const bool outside = qx >= W0 || qx < 0 || qy < 0 || qy >= H0;
            px = (long long)qx;
            py = (long long)qy;
            vis = outside ? 1 : 2;
        }
        double ox, oy;
        trk_affine(T, (double)px, (double)py, &ox, &oy);
        px = (long long)ox;
        py = (long long)oy;
        if (!(vis > 1 && px >= 0 && (double)px < iw && py >= 0 && (double)py < ih)) continue;
        double k = 1.0;
        bool draw;
        if (pnp_mode && (P.render_hmhp_mode == 0 || P.render_hmhp_mode == 2)) {
            const double* spread = P.hps_uncertainty ? t + TR_FUS_STD : t + TR_POST + PO_HM_STD;
            draw = (int)spread[2 * j] > 0;  // astype(int32): the heat-map estimate is sometimes missing
            if (P.kalman && (flags & 16)) {
                const double vx = t[TR_KF_P + 16 * j], vy = t[TR_KF_P + 16 * j + 5];
                k = trk_conf(P, sqrt(vx + vy));
            } else if (P.hps_uncertainty) {
                k = trk_conf(P, sqrt(t[TR_FUS_STD + 2 * j] + t[TR_FUS_STD + 2 * j + 1]));
            } else {
                k = t[TR_POST + PO_HM_HEIGHT + j];
            }
        } else {
            draw = true;
        }
        if (!draw) continue;
        double* r = rec + 5 * (1 + j);
        r[0] = hp_plane0 + j; r[1] = (double)(int)px; r[2] = (double)(int)py; r[3] = radius; r[4] = k;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Association of one video (Tracker.step :125-236), sequential: which record each entry of this frame's list is made of.
//   dets [nd_all][STRIDE]  this frame's candidate detections (stage 1), in decode order
//   use [nd_all]           1 = takes part (when any detection has a box, only those do: :116-123)
//   prev [np][STRIDE]      last frame's tracks
//   plan [cap][3]          entry e of the new list = (kind, detection index, prev-track index):
//                          kind 0 matched detection (inherits id / filter of the prev track), 1 new track (id in slot 2),
//                          2 coasting prev track.  Order as the reference builds its list: matched (detection order),
//                          new (detection order), coasting (track order).
// Returns the length of the new list (<= P.cap).  A frame that would need more than P.cap entries keeps the first P.cap in the
// reference's own order -- every matched track (there are at most P.cap of them), then new tracks in detection (= score)
// order, then coasting tracks -- and reports the number of entries it dropped in *dropped: the state stays well defined (no id
// is given to a dropped detection; a dropped coasting track is simply retired early) and the caller can surface the event.
CP_HD int trk_associate(const TrackParams& P, const double* dets, const int* use, int nd_all, const double* prev, int np,
                        int* plan, int* id_count, int* det_idx /* scratch [K] */, unsigned char* taken /* scratch [cap] */,
                        int* dropped, const TrkLsapWork* lsap = nullptr, int* lsap_match /* scratch [K] */ = nullptr,
                        const TrkMunkresWork* munkres = nullptr) {
    *dropped = 0;
    int nd = 0;
    for (int k = 0; k < nd_all; ++k)
        if (use[k]) det_idx[nd++] = k;
    // centres compared (float32 arrays in the reference).  Tracker: detection centre moved by its tracking offset against the
    // track's centre (tracker.py:130, :143); Tracker_baseline: the raw detection centre against the track's centre moved by the
    // mean filtered velocity of its eight vertices (tracker_baseline.py:121, :134-140; float64 sum, then the float32 array)
    auto det_centre = [&](const double* d, float* cx, float* cy) {
        if (P.baseline) { *cx = f32(d[TR_POST + PO_CT]); *cy = f32(d[TR_POST + PO_CT + 1]); }
        else {
            *cx = f32(d[TR_POST + PO_CT] + d[TR_POST + PO_TRACKING]);
            *cy = f32(d[TR_POST + PO_CT + 1] + d[TR_POST + PO_TRACKING + 1]);
        }
    };
    auto trk_centre = [&](const double* tr, float* cx, float* cy) {
        if (P.baseline) {
            double vx = 0, vy = 0;
            for (int i = 0; i < 8; ++i) { vx += tr[TR_KF_X + 4 * i + 2]; vy += tr[TR_KF_X + 4 * i + 3]; }
            *cx = f32((double)f32(tr[TR_POST + PO_CT]) + vx / 8);
            *cy = f32((double)f32(tr[TR_POST + PO_CT + 1]) + vy / 8);
        } else { *cx = f32(tr[TR_POST + PO_CT]); *cy = f32(tr[TR_POST + PO_CT + 1]); }
    };
    for (int t = 0; t < np; ++t) taken[t] = 0;
    if (P.hungarian && lsap) {
        // ---- optimal assignment over the same cost matrix (:130-157), forbidden pairs undone afterwards (:167-174) ----
        auto cost = [&](int i, int t) -> double {
            const double* d = dets + (long long)det_idx[i] * CP_TRACK_STRIDE;
            const double* tr = prev + (long long)t * CP_TRACK_STRIDE;
            float dcx, dcy, tcx, tcy;
            det_centre(d, &dcx, &dcy);
            trk_centre(tr, &tcx, &tcy);
            const float darea = f32((d[TR_POST + PO_BBOX + 2] - d[TR_POST + PO_BBOX]) *
                                    (d[TR_POST + PO_BBOX + 3] - d[TR_POST + PO_BBOX + 1]));
            const float ex = tcx - dcx, ey = tcy - dcy;
            const float ex2 = ex * ex, ey2 = ey * ey;
            const float c32 = ex2 + ey2;
            const float tarea = f32((tr[TR_POST + PO_BBOX + 2] - tr[TR_POST + PO_BBOX]) *
                                    (tr[TR_POST + PO_BBOX + 3] - tr[TR_POST + PO_BBOX + 1]));
            const bool bad = c32 > tarea || c32 > darea || (int)d[TR_POST + PO_CLS] != (int)tr[TR_POST + PO_CLS];
            const double c = (double)c32 + (bad ? 1e18 : 0.0);
            return c > 1e18 ? 1e18 : c;  // dist[dist > 1e18] = 1e18
        };
        // hungarian = 1: the reference's dependency (sklearn 0.22.2's Munkres); 2: scipy's rectangular LSAP -- the same optimum
        // value, possibly another optimum among tied / forbidden pairs (include/centerpose_hip.h: cp_track_params)
        if (P.hungarian == 2 || !munkres) trk_lsap(cost, nd, np, lsap_match, *lsap);
        else trk_munkres(cost, nd, np, lsap_match, *munkres);
        for (int i = 0; i < nd; ++i) {
            const int match = lsap_match[i];
            if (match >= 0 && cost(i, match) > 1e16) {
                // a forbidden pair of the optimum is undone (:167-174): both ends go to the END of the unmatched lists, in pair
                // order -- after the detections / tracks the assignment left out altogether
                taken[match] = 2;
                det_idx[i] |= (1 << 30) | ((match + 1) << 16);
            } else {
                if (match >= 0) taken[match] = 1;
                det_idx[i] |= (match + 1) << 16;
            }
        }
    } else
    // ---- greedy: detections in order, each takes its nearest still-free admissible track (:305-314) ----
    for (int i = 0; i < nd; ++i) {
        const double* d = dets + (long long)det_idx[i] * CP_TRACK_STRIDE;
        float dcx, dcy;
        det_centre(d, &dcx, &dcy);
        const float darea = f32((d[TR_POST + PO_BBOX + 2] - d[TR_POST + PO_BBOX]) *
                                (d[TR_POST + PO_BBOX + 3] - d[TR_POST + PO_BBOX + 1]));
        const int dcls = (int)d[TR_POST + PO_CLS];
        int best = -1;
        double bestc = 0;
        for (int t = 0; t < np; ++t) {
            const double* tr = prev + (long long)t * CP_TRACK_STRIDE;
            float tcx, tcy;
            trk_centre(tr, &tcx, &tcy);
            const float ex = tcx - dcx, ey = tcy - dcy;
            const float ex2 = ex * ex, ey2 = ey * ey;
            const float c32 = ex2 + ey2;  // float32: each square rounded, then their sum
            const float tarea = f32((tr[TR_POST + PO_BBOX + 2] - tr[TR_POST + PO_BBOX]) *
                                    (tr[TR_POST + PO_BBOX + 3] - tr[TR_POST + PO_BBOX + 1]));
            const bool bad = c32 > tarea || c32 > darea || dcls != (int)tr[TR_POST + PO_CLS];
            double c = (double)c32 + (bad ? 1e18 : 0.0);
            if (taken[t]) c = 1e18;  // the column was overwritten by an earlier match
            if (best < 0 || c < bestc) { best = t; bestc = c; }
        }
        int match = -1;
        if (np > 0 && bestc < 1e16) { match = best; taken[best] = 1; }
        det_idx[i] |= (match + 1) << 16;  // remember the match next to the detection index
    }
    int n_out = 0;
    for (int i = 0; i < nd; ++i) {  // matched, in detection order
        const int k = det_idx[i] & 0xffff, match = ((det_idx[i] >> 16) & 0x3fff) - 1;
        if (match < 0 || (det_idx[i] & (1 << 30))) continue;
        if (n_out >= P.cap) { *dropped += 1; continue; }  // (cannot happen: matches <= np <= cap)
        plan[3 * n_out] = 0; plan[3 * n_out + 1] = k; plan[3 * n_out + 2] = match;
        ++n_out;
    }
    // unmatched detections above new_thresh start tracks: first the ones the association left out, then (Hungarian only) the
    // ones whose forbidden pair was undone -- the order in which the reference's list grows and ids are handed out
    for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < nd; ++i) {
            const int k = det_idx[i] & 0xffff, match = ((det_idx[i] >> 16) & 0x3fff) - 1;
            const bool undone = (det_idx[i] & (1 << 30)) != 0;
            if (pass == 0 ? (match >= 0 || undone) : !undone) continue;
            if (!(dets[(long long)k * CP_TRACK_STRIDE + TR_POST + PO_SCORE] > P.new_thresh)) continue;
            if (n_out >= P.cap) { *dropped += 1; continue; }
            *id_count += 1;
            plan[3 * n_out] = 1; plan[3 * n_out + 1] = k; plan[3 * n_out + 2] = *id_count;
            ++n_out;
        }
    for (int t = 0; t < np; ++t) {  // unmatched tracks coast until max_age: the left-out ones in track order ...
        if (taken[t]) continue;
        if (!(prev[(long long)t * CP_TRACK_STRIDE + TR_AGE] < P.max_age)) continue;
        if (n_out >= P.cap) { *dropped += 1; continue; }
        plan[3 * n_out] = 2; plan[3 * n_out + 1] = -1; plan[3 * n_out + 2] = t;
        ++n_out;
    }
    for (int i = 0; i < nd; ++i) {  // ... then the tracks of undone pairs, in pair (= detection) order
        if (!(det_idx[i] & (1 << 30))) continue;
        const int t = ((det_idx[i] >> 16) & 0x3fff) - 1;
        if (!(prev[(long long)t * CP_TRACK_STRIDE + TR_AGE] < P.max_age)) continue;
        if (n_out >= P.cap) { *dropped += 1; continue; }
        plan[3 * n_out] = 2; plan[3 * n_out + 1] = -1; plan[3 * n_out + 2] = t;
        ++n_out;
    }
    return n_out;
}

// Elements [e0, e1) of list entry `o` according to its plan triple (the device copies with a whole wavefront); call with
// the full range once (or last) so that the header fields are written.
CP_HD void trk_materialise(const int* pl, const double* dets, const double* prev, double* o, int e0, int e1) {
    const double* src = pl[0] == 2 ? prev + (long long)pl[2] * CP_TRACK_STRIDE : dets + (long long)pl[1] * CP_TRACK_STRIDE;
    for (int e = e0; e < e1; ++e) {
        double v = src[e];
        if (pl[0] == 0) {  // matched detection: id / activity of the track it continues
            const double* old = prev + (long long)pl[2] * CP_TRACK_STRIDE;
            if (e == TR_ID) v = old[TR_ID];
            else if (e == TR_AGE) v = 1;
            else if (e == TR_ACTIVE) v = old[TR_ACTIVE] + 1;
            else if (e == TR_SRC) v = pl[2];
        } else if (pl[0] == 1) {
            if (e == TR_ID) v = pl[2];
            else if (e == TR_AGE) v = 1;
            else if (e == TR_ACTIVE) v = 1;
            else if (e == TR_SRC) v = -1;
        } else {
            if (e == TR_AGE) v = src[TR_AGE] + 1;
            else if (e == TR_ACTIVE) v = 0;
            else if (e == TR_SRC) v = -2;
        }
        o[e] = v;
    }
}

// Kalman / pool state of one entry of the new list (:178-218) followed by the read-out (:238-275).
CP_HD void trk_advance(const TrackParams& P, double* t, const double* prev, double* pts16, double* scale3) {
    const int src = (int)t[TR_SRC];
    int flags = (int)t[TR_FLAGS];
    if (src >= 0) {  // matched: inherits the filter and the pool, then predict + update / append
        const double* old = prev + (long long)src * CP_TRACK_STRIDE;
        if (P.kalman) {
            for (int i = 0; i < 32 + 128; ++i) t[TR_KF_X + i] = old[TR_KF_X + i];
            trk_kf_step(P, t);
        }
        if (P.scale_pool) {
            for (int i = 0; i < 7; ++i) t[TR_POOL_PREC + i] = old[TR_POOL_PREC + i];
            trk_pool_add(P, t);
        }
        flags |= 16;
    } else if (src == -1) {  // new track
        if (P.kalman) trk_kf_init(P, t);
        if (P.scale_pool) {
            for (int i = 0; i < 7; ++i) t[TR_POOL_PREC + i] = 0.0;
            trk_pool_add(P, t);
        }
        flags |= 16;
    }
    // the filtered PnP fields of an earlier frame survive only in a coasting track's own record
    if (src != -2) flags &= ~(2 | 4);
    else flags &= ~4;
    t[TR_FLAGS] = flags;
    trk_readout(P, t, pts16, scale3);
}

// ---------------------------------------------------------------------------------------------------------------------
// The stages of one frame, in the order the kernels of track.hip (and the host test harness) run them.

// stage 1, per detection slot k < count: the candidate record (post-processed fields, fusion, packaged PnP answer).
// Returns 1 when the detection comes with a box (`boxes` of base_detector.py:652-654).
CP_HD int trk_prepare_det(const TrackParams& P, const double* vm, const double* post, const double* pnp_row, double* d) {
    for (int e = 0; e < CP_TRACK_STRIDE; ++e) d[e] = 0.0;
    for (int e = 0; e < 120; ++e) d[TR_POST + e] = post[e];
    trk_fuse(post, P.hps_uncertainty, d + TR_FUS_MEAN, d + TR_FUS_STD);
    int flags = 0, ok = 0;
    if (P.use_pnp && pnp_row && (int)pnp_row[0] == 1) {
        double ori[18];
        ok = trk_finish(P, vm, pnp_row, post + PO_SCALE, post + PO_KPS, d, ori, P.show_axes);
        flags |= 1;
        if (ok) {
            for (int i = 0; i < 18; ++i) d[TR_KPS_ORI + i] = ori[i];
            flags |= 8;
        }
    }
    d[TR_FLAGS] = flags;
    return ok;
}

// stage 3, per entry of the new list: filter state + read-out, and the inputs of the filtered PnP (8 points, relative
// size as float32(scale / scale_y), like the host path's solve_pnp_batch)
CP_HD void trk_track_stage(const TrackParams& P, double* t, const double* prev, float* pts16, float* scale3) {
    double p16[16], s3[3];
    trk_advance(P, t, prev, p16, s3);
    for (int i = 0; i < 16; ++i) pts16[i] = (float)p16[i];
    for (int i = 0; i < 3; ++i) scale3[i] = (float)(s3[i] / s3[1]);
}

// stage 5, per track: package the filtered PnP answer (tracker.py:276-294) and emit next frame's Gaussians
CP_HD void trk_finish_stage(const TrackParams& P, const double* vm, double* t, const double* pnp_row, int hm_plane,
                            int hp_plane0, double* rec) {
    int flags = (int)t[TR_FLAGS];
    if (P.use_pnp && pnp_row && (int)pnp_row[0] == 1) {
        double ori[18], s3[3];
        for (int i = 0; i < 3; ++i) s3[i] = P.scale_pool ? t[TR_SCALE_KF + i] : t[TR_POST + PO_SCALE + i];
        const int ok = trk_finish(P, vm, pnp_row, s3, t + TR_POST + PO_KPS, t, ori, P.baseline ? 0 : P.show_axes);
        flags |= 1;
        if (ok) {
            for (int i = 0; i < 18; ++i) t[TR_KPS_PNP_KF + i] = t[TR_KPS_PNP + i];
            for (int i = 0; i < 27; ++i) t[TR_KPS_3D_KF + i] = t[TR_KPS_3D + i];
            for (int i = 0; i < 18; ++i) t[TR_KPS_ORI_KF + i] = ori[i];
            flags |= 2;
            double cs = 0;
            for (int v = 0; v < 8; ++v) cs += t[TR_CONF + v];
            if (cs / 8 > 0.25) flags |= 4;
        }
    }
    t[TR_FLAGS] = flags;
    if (rec) trk_render_records(P, vm, t, hm_plane, hp_plane0, rec);
}

#if defined(__clang__) && defined(__HIPCC__)
#pragma clang fp contract(fast)  // hipcc's default again: the switch above must not leak into the rest of engine.hip / track.hip
#endif
