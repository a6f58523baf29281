// Host build of centerpose_amd/csrc/post_common.h for tests/test_post_logic_cpu.py: one image through the same source the
// device's postprocess_kernel runs (transform every record, threshold filter, soft-NMS, gather in the final order).
#include <cstddef>
#include <vector>

#include "../../centerpose_amd/csrc/post_common.h"

extern "C" int cp_post_host_image(const float* det, int K, const double* meta8, double vis_thresh, int nms, float div_scale,
                                  double* out /*[K][120]*/) {
    std::vector<double> ob((size_t)K * CP_POST_STRIDE), score(K);
    std::vector<int> idx(K);
    std::vector<double> box((size_t)K * 4);
    double(*bx)[4] = reinterpret_cast<double(*)[4]>(box.data());
    const float ratio = (float)meta8[6];
    for (int k = 0; k < K; ++k) {
        double* o = ob.data() + (size_t)k * CP_POST_STRIDE;
        post_transform_record(det + (size_t)k * CP_DET_STRIDE, meta8, ratio, div_scale, o);
        score[k] = o[0];
        for (int i = 0; i < 4; ++i) bx[k][i] = o[24 + i];
    }
    const int N = post_filter_nms(score.data(), bx, idx.data(), K, vis_thresh, nms);
    for (int r = 0; r < N; ++r) {
        const double* src = ob.data() + (size_t)idx[r] * CP_POST_STRIDE;
        for (int e = 0; e < CP_POST_STRIDE; ++e) out[(size_t)r * CP_POST_STRIDE + e] = e == 0 ? score[r] : src[e];
    }
    return N;
}
