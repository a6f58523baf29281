// TEST HARNESS (not product code): the scalar functions of centerpose_amd/csrc/track_common.h compiled for the host with
// g++, so that the device tracker's logic can be compared with the reference-pinned Python tracker on a machine without a
// GPU (tests/test_track_logic_cpu.py).  The product runs the same functions inside track.hip's kernels.
#include "../../centerpose_amd/csrc/track_common.h"

#include <vector>

static int g_last_dropped = 0;
extern "C" {
int cp_track_host_last_dropped(void) { return g_last_dropped; }

int cp_track_host_stride(void) { return CP_TRACK_STRIDE; }
int cp_track_host_params_bytes(void) { return (int)sizeof(TrackParams); }

// stages 1-3 of one frame of one video.  post [count][120], pnp_rows [count][40] or null, prev [*np][STRIDE] ->
// next [cap][STRIDE] (returns its length, -1 on overflow), pts [cap][16] float, scale [cap][3] float
int cp_track_host_update(const TrackParams* P, const double* vm, const double* post, int count, const double* pnp_rows,
                         const double* prev, int np, int* id_count, double* next, float* pts, float* scale) {
    std::vector<double> dets((size_t)(count > 0 ? count : 1) * CP_TRACK_STRIDE);
    std::vector<int> use(count > 0 ? count : 1), idx(count > 0 ? count : 1);
    std::vector<unsigned char> taken(np > 0 ? np : 1);
    int any = 0;
    for (int k = 0; k < count; ++k) {
        use[k] = trk_prepare_det(*P, vm, post + (size_t)k * 120, pnp_rows ? pnp_rows + (size_t)k * 40 : nullptr,
                                 dets.data() + (size_t)k * CP_TRACK_STRIDE);
        any |= use[k];
    }
    if (!any)
        for (int k = 0; k < count; ++k) use[k] = 1;
    std::vector<int> plan((size_t)3 * (P->cap > 0 ? P->cap : 1));
    int dropped = 0;
    const int LS = (count > np ? count : np) + 1;
    std::vector<double> wu(LS), wv(LS), ws(LS);
    std::vector<int> wp(LS), wc(LS), wr(LS), wrem(LS), lm(LS);
    std::vector<unsigned char> wsr(LS), wsc(LS);
    const TrkLsapWork W = {wu.data(), wv.data(), ws.data(), wp.data(), wc.data(), wr.data(), wrem.data(), wsr.data(), wsc.data()};
    std::vector<double> mC((size_t)LS * LS);
    std::vector<unsigned char> mM((size_t)LS * LS), mR(LS), mCu(LS);
    std::vector<int> mP(4 * (size_t)LS);
    const TrkMunkresWork MW = {mC.data(), mM.data(), mR.data(), mCu.data(), mP.data()};
    const int n = trk_associate(*P, dets.data(), use.data(), count, prev, np, plan.data(), id_count, idx.data(), taken.data(), &dropped,
                                &W, lm.data(), &MW);
    g_last_dropped = dropped;
    for (int t = 0; t < n; ++t)
        trk_materialise(plan.data() + 3 * t, dets.data(), prev, next + (size_t)t * CP_TRACK_STRIDE, 0, CP_TRACK_STRIDE);
    for (int t = 0; t < n; ++t)
        trk_track_stage(*P, next + (size_t)t * CP_TRACK_STRIDE, prev, pts + (size_t)t * 16, scale + (size_t)t * 3);
    return n;
}

// the assignment alone, on an explicit nd x nt cost matrix (tests: against scipy.optimize.linear_sum_assignment)
void cp_track_host_lsap(const double* cost, int nd, int nt, int* match) {
    const int LS = (nd > nt ? nd : nt) + 1;
    std::vector<double> wu(LS), wv(LS), ws(LS);
    std::vector<int> wp(LS), wc(LS), wr(LS), wrem(LS);
    std::vector<unsigned char> wsr(LS), wsc(LS);
    const TrkLsapWork W = {wu.data(), wv.data(), ws.data(), wp.data(), wc.data(), wr.data(), wrem.data(), wsr.data(), wsc.data()};
    trk_lsap([&](int i, int j) { return cost[(size_t)i * nt + j]; }, nd, nt, match, W);
}

// sklearn 0.22.2's Munkres on an explicit nd x nt cost matrix (tests: against oracle/munkres.py)
void cp_track_host_munkres(const double* cost, int nd, int nt, int* match) {
    const size_t LS = (size_t)(nd > nt ? nd : nt) + 1;
    std::vector<double> mC((size_t)nd * nt + 1);
    std::vector<unsigned char> mM((size_t)nd * nt + 1), mR(LS), mCu(LS);
    std::vector<int> mP(4 * LS);
    const TrkMunkresWork MW = {mC.data(), mM.data(), mR.data(), mCu.data(), mP.data()};
    trk_munkres([&](int i, int j) { return cost[(size_t)i * nt + j]; }, nd, nt, match, MW);
}

// stage 5: tracks [n][STRIDE] (in place), pnp_rows [n][40] or null, recs [n][9][5]
void cp_track_host_finish(const TrackParams* P, const double* vm, double* tracks, int n, const double* pnp_rows, int hm_plane,
                          int hp_plane0, double* recs) {
    for (int t = 0; t < n; ++t)
        trk_finish_stage(*P, vm, tracks + (size_t)t * CP_TRACK_STRIDE, pnp_rows ? pnp_rows + (size_t)t * 40 : nullptr,
                         hm_plane, hp_plane0, recs + (size_t)t * 45);
}

}  // extern "C"
