// DLA-34 (+DCNv2 up-sampling, optional ConvGRU + GroupNorm heads) inference engine and the C ABI
// (include/centerpose_hip.h).  Host-side C++ only orchestrates: parameters are folded / packed once
// at cp_model_finalize, the forward pass is a fixed sequence of HIP kernel launches on the caller's
// stream out of a caller-provided workspace (deterministic arena, no allocation, no sync).
//
// Topology follows the reference modules (paths relative to /root/reference/src/lib/models/networks):
//   DLA.forward pose_dla_dcn.py:310-322, Tree.forward :211-224, Root :160-168, BasicBlock :48-62,
//   DLAUp :437-443, IDAUp :411-417, DeformConv :386-389 (DCN: DCNv2/dcn_v2.py:118-128),
//   DLASeg.forward :523-570, ConvGRU convGRU.py:72-94, GroupNorm GN.py:4-9.
#include "../../include/centerpose_hip.h"
#include "../../include/centerpose_hip_testing.h"
#undef CP_OK
#undef CP_ERR_INVALID
#undef CP_ERR_LAUNCH
#undef CP_ERR_ALLOC
#undef CP_ERR_STATE
#undef CP_DET_STRIDE
#undef CP_PNP_STRIDE
#undef CP_TRACK_STRIDE
#include "cp_common.h"
#include "track_common.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// Deterministic first-fit arena over a caller-provided workspace.  A dry run (base == nullptr)
// replays the same allocation sequence to measure the peak, so cp_model_workspace_bytes() and
// cp_model_forward() always agree.
// ---------------------------------------------------------------------------------------------
struct Arena {
    char* base = nullptr;
    size_t cap = 0, peak = 0;
    bool overflow = false;
    std::vector<std::pair<size_t, size_t>> free_;  // (offset, size), sorted by offset

    void reset(void* b, size_t c) {
        base = (char*)b;
        cap = c;
        peak = 0;
        overflow = false;
        free_.clear();
        free_.push_back({0, (size_t)1 << 62});
    }
    size_t alloc(size_t bytes) {
        bytes = align_up(bytes, 256);
        for (size_t i = 0; i < free_.size(); ++i) {
            if (free_[i].second >= bytes) {
                const size_t off = free_[i].first;
                free_[i].first += bytes;
                free_[i].second -= bytes;
                if (free_[i].second == 0) free_.erase(free_.begin() + i);
                if (off + bytes > peak) peak = off + bytes;
                if (base && off + bytes > cap) overflow = true;
                return off;
            }
        }
        overflow = true;
        return 0;
    }
    void release(size_t off, size_t bytes) {
        bytes = align_up(bytes, 256);
        size_t i = 0;
        while (i < free_.size() && free_[i].first < off) ++i;
        free_.insert(free_.begin() + i, {off, bytes});
        if (i + 1 < free_.size() && free_[i].first + free_[i].second == free_[i + 1].first) {
            free_[i].second += free_[i + 1].second;
            free_.erase(free_.begin() + i + 1);
        }
        if (i > 0 && free_[i - 1].first + free_[i - 1].second == free_[i].first) {
            free_[i - 1].second += free_[i].second;
            free_.erase(free_.begin() + i);
        }
    }
};

struct Block {
    Arena* a;
    size_t off, bytes;
    Block(Arena* a_, size_t b) : a(a_), off(a_->alloc(b)), bytes(b) {}
    ~Block() { a->release(off, bytes); }
};

// NHWC activation handle; memory returns to the arena when the last handle dies (the single stream
// orders reuse after the last enqueued consumer).
struct Tensor {
    std::shared_ptr<Block> blk;
    int C = 0, H = 0, W = 0;
    unsigned* amax = nullptr;  // 4-byte slot holding the float bits of max|x| (f16x3 mode; ConvParams::in_amax)
    float* ptr() const { return blk->a->base ? (float*)(blk->a->base + blk->off) : nullptr; }
    bool valid() const { return (bool)blk; }
};

struct ConvW {
    float* wp = nullptr;     // [Kpad][CoutPad]
    float* scale = nullptr;  // [CoutPad] or nullptr
    float* shift = nullptr;  // [CoutPad] or nullptr
    int Cin = 0, CinP = 0, Cout = 0, CoutPad = 0, KH = 0, KW = 0, K = 0, Kpad = 0;
    void* w16_hi = nullptr;  // split-f16 copies [CoutPad][K] (only when every K-step of 32 stays inside one tap)
    void* w16_lo = nullptr;
    int Kpad16 = 0;
    void* w16f_hi = nullptr;  // DCN main convolutions: the same in MFMA fragment order (dcn16p.hip)
    void* w16f_lo = nullptr;
    // per-output-channel power-of-two pre-scale of the split-f16 copies: rows are stored times wfwd[co] = 2^e,
    // winv = 2^-e, scale16 = (scale or 1) * winv is what the f16x3 kernels' epilogue multiplies with
    float* wfwd = nullptr;
    float* winv = nullptr;
    float* scale16 = nullptr;
};

struct LowcW {
    void* hi = nullptr;
    void* lo = nullptr;
    float* scale16 = nullptr;  // folded BatchNorm scale x 2^-e of the fragment rows
};

int g_default_precision = CP_PREC_F32;
// split-K policy: launches with fewer output tiles than kSplitTiles (and >= 8 K steps) are cut into K slices until
// about kSplitTarget workgroups exist
constexpr int kSplitTiles = 128, kSplitTarget = 384;  // (384 / 512 measured: B=32 equal, hourglass B=1 latency +7 %)
int g_dbg = 0;  // cp_set_debug (include/centerpose_hip_testing.h: kernel SELECTION switches for the parity tests and A/B runs; every choice computes the layer correctly): 4 1x1 layers with fragment-shaped A loads (pw16_kernel) instead of whole lines through staging rows (pw16s_kernel), 8 split-K epilogue element-wise (not the quad form), 1 grouped heads write slabs + reduction launch, 2 grouped heads one workgroup per head (not per patch), 16 small launches on 128-row tiles, 32 no head fusion, 64 no lowc kernels, 128 GN heads' 1x1 on the f32 kernel, 256 unfused ConvGRU step, 512 no activation |max| tracking / pre-scale, 1024 previous DCN loop, 2048 alternative DCN wave counts, 4096 / 8192 halo kernel never / everywhere, 16384 LDS-staged weights in the N=32 halo kernel, 32768 / 65536 patch-resident DCN never / everywhere, 524288 patch-resident DCN never on the 128-wide N tile, 1048576 / 2097152 streamed DCN (dcn16s) never / everywhere, 33554432 / 67108864 three-workgroup DCN (dcn16t) everywhere / never, 134217728 stem and level0 as two kernels (not the fused one), 262144 / 1073741824 level1 never / always on the row-streaming kernel, 268435456 / 536870912 row-streamed 64 -> <= 32 channel 3x3 layers (strm16) never / at any size, 131072 GroupNorm'd heads' 1x1 on the matrix cores, 4194304 1x1 layers on the LDS-staged loop instead of pw16.hip, 8388608 cp_dcnv2_forward always on the generic kernel, 16777216 fused heads one launch per head instead of one grouped launch

struct DeformW {
    ConvW offset;  // conv_offset_mask (27 -> 32 padded), shift = bias
    ConvW main;    // DCN weight, scale/shift = folded bias + BN
};

struct HeadW {
    std::string name;
    int classes = 0;
    ConvW c0, c1;
    void* w2_hi = nullptr;  // fused-head form of c1 (cp_launch_pack_head_w2); null when the pair is not eligible
    void* w2_lo = nullptr;
    float* w2_inv = nullptr;  // [32] 2^-e per final channel (+ [32] 2^e used while packing)
    float* gn_gamma = nullptr;
    float* gn_beta = nullptr;
};

}  // namespace

struct cp_model {
    std::string arch;
    bool gru = false, tracking = false, finalized = false, hourglass = false;
    int precision = g_default_precision;
    int head_conv = 256;
    std::vector<std::pair<std::string, int>> heads;
    std::map<std::string, std::vector<float>> params;  // host copies until finalize
    std::map<std::string, ConvW> convs;
    std::map<std::string, DeformW> deforms;
    std::map<std::string, float*> ups;
    std::vector<HeadW> headw;
    // every fused head of the model in ONE launch (they all read the same feature map): the heads' 3x3 fragments,
    // scale / shift, 1x1 fragments and w2_inv tables concatenated along N (ConvParams::fuse_ngroups)
    struct HeadGroup {
        bool ok = false;
        void* w16f_hi = nullptr;
        void* w16f_lo = nullptr;
        void* w2_hi = nullptr;
        void* w2_lo = nullptr;
        float* scale16 = nullptr;
        float* shift = nullptr;
        float* w2_inv = nullptr;
        int Cin = 0, hid = 0, Kpad16 = 0;
    } head_group;
    std::map<std::string, LowcW> lowc;  // hi / lo weight fragments of the lowc.hip layers
    int ws_key[4] = {0, 0, 0, -1};  // (B, H, W, g_dbg) of the cached work-space query below
    size_t ws_cached = 0;
    int dry_variant = 0;  // work-space query: 1 = the dry run takes the fused stem + level0 path where the model allows it (the query
                          // runs both forms and returns the larger peak: switches and taps may select either form later)
    float stem_bound_l = 0.f, stem_bound_s = 0.f;  // |base_layer out| <= stem_bound_l * max|image| + stem_bound_s (fused stem + level0)
    ConvW gru_x, gru_h;
    void* gru_h16_hi = nullptr;  // hidden-side GRU weights re-ordered [tile][r|z|n][32] for the fused-gate kernel
    void* gru_h16_lo = nullptr;
    void* gru_h16f_hi = nullptr;  // ... and in MFMA fragment order (halo16.hip)
    void* gru_h16f_lo = nullptr;
    float* gru_h16_fwd = nullptr;  // [192] per-row 2^e of the fused-order copies, and the matching 2^-e
    float* gru_h16_inv = nullptr;
    std::vector<void*> device_allocs;
    Arena arena;
    // forward-call state
    hipStream_t stream = nullptr;
    int B = 0;
    bool dry = false;
    int status = CP_OK;
    const char* tap_name = nullptr;
    float* tap_out = nullptr;
    int* tap_dims = nullptr;
    // optional per-launch profiling of the implicit-GEMM kernels (HIP events on the launch stream)
    struct ProfRec {
        int variant;
        int role = 0;  // CP_ROLE_*
        double flops, bytes;
        int M, N, K, kh, stride;
        hipEvent_t e0, e1;
    };
    std::map<std::vector<uint64_t>, hipGraphExec_t> graphs;  // captured detect() launches, keyed by every argument
    bool profile = false;
    std::vector<ProfRec> prof;
    double roles[CP_NUM_ROLES * 4] = {0};  // per-role totals of the last cp_model_profile_read
    std::vector<hipEvent_t> event_pool;
    hipEvent_t get_event() {
        if (!event_pool.empty()) {
            hipEvent_t e = event_pool.back();
            event_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};

namespace {

// ------------------------------- parameter packing -------------------------------------------
struct Packer {
    cp_model* m;
    int status = CP_OK;
    std::string missing;

    // a failing HIP runtime call while packing makes cp_model_finalize fail (first error wins)
    bool hip_ok(hipError_t e) {
        if (e != hipSuccess && status == CP_OK) {
            status = CP_ERR_LAUNCH;
            missing = std::string("HIP runtime: ") + hipGetErrorString(e);
        }
        return e == hipSuccess;
    }
    const std::vector<float>* get(const std::string& n, size_t numel) {
        auto it = m->params.find(n);
        if (it == m->params.end() || it->second.size() != numel) {
            if (status == CP_OK) missing = n;
            status = CP_ERR_STATE;
            return nullptr;
        }
        return &it->second;
    }
    float* dev_alloc(size_t nfloat, bool zero = true) {
        void* p = nullptr;
        if (hipMalloc(&p, nfloat * sizeof(float)) != hipSuccess) {
            status = CP_ERR_ALLOC;
            return nullptr;
        }
        if (zero) hip_ok(hipMemset(p, 0, nfloat * sizeof(float)));
        m->device_allocs.push_back(p);
        return (float*)p;
    }
    float* upload(const std::vector<float>& h) {
        float* d = dev_alloc(h.size(), false);
        if (d) hip_ok(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        return d;
    }
    // fragment-ordered copies of the split-f16 weights (3x3 layers): dcn16p.hip and halo16.hip load their MFMA B operands
    // straight from them
    void frag_copies(ConvW& c) {
        if (c.w16f_hi) return;
        const size_t halfs = (size_t)c.CoutPad * c.Kpad16;
        c.w16f_hi = dev_alloc((halfs + 1) / 2);
        c.w16f_lo = dev_alloc((halfs + 1) / 2);
        if (!c.w16f_hi || !c.w16f_lo) return;
        int rc = cp_launch_frag16_repack(c.w16_hi, c.w16f_hi, c.CoutPad, c.Kpad16, nullptr);
        if (rc == CP_OK) rc = cp_launch_frag16_repack(c.w16_lo, c.w16f_lo, c.CoutPad, c.Kpad16, nullptr);
        hip_ok(hipDeviceSynchronize());
        if (rc != CP_OK) status = rc;
    }
    // Pack several PyTorch-layout weights side by side along Cout (GRU gates) into one GEMM operand.
    ConvW pack(const std::vector<std::string>& wnames, int cout_each, int cin, int kh, int kw, int cin_pad = 0,
               int cout_pad_min = 0) {
        ConvW c;
        c.Cin = cin;
        c.CinP = cin_pad ? cin_pad : cin;
        c.Cout = cout_each * (int)wnames.size();
        c.CoutPad = (int)align_up(c.Cout, cp_conv_tile_n(c.Cout));
        if (c.CoutPad < cout_pad_min) c.CoutPad = cout_pad_min;  // <= 16-wide heads: 32 columns for the f16x3 N tile
        c.KH = kh;
        c.KW = kw;
        c.K = kh * kw * c.CinP;
        c.Kpad = (int)align_up(c.K, 16);
        c.wp = dev_alloc((size_t)c.Kpad * c.CoutPad);
        if (!c.wp) return c;
        const bool want16 = (c.CinP == cin) && (cin % 32 == 0) && c.CoutPad >= 32 && c.CoutPad % 32 == 0 && kh * kw <= 32;
        if (want16) {
            c.Kpad16 = c.K;
            const size_t halfs = (size_t)c.CoutPad * c.Kpad16;
            c.w16_hi = dev_alloc((halfs + 1) / 2);
            c.w16_lo = dev_alloc((halfs + 1) / 2);
            const std::vector<float> ones(c.CoutPad, 1.f);
            c.wfwd = upload(ones);
            c.winv = upload(ones);
            c.scale16 = upload(ones);
            if (!c.w16_hi || !c.w16_lo || !c.wfwd || !c.winv || !c.scale16) return c;
        }
        for (size_t i = 0; i < wnames.size(); ++i) {
            const auto* w = get(wnames[i], (size_t)cout_each * cin * kh * kw);
            if (!w) return c;
            float* tmp = nullptr;
            if (hipMalloc((void**)&tmp, w->size() * sizeof(float)) != hipSuccess) {
                status = CP_ERR_ALLOC;
                return c;
            }
            hip_ok(hipMemcpy(tmp, w->data(), w->size() * sizeof(float), hipMemcpyHostToDevice));
            int rc = cp_launch_pack_weight(tmp, c.wp, cout_each, cin, kh * kw, c.CinP, c.CoutPad, (int)i * cout_each,
                                           nullptr);
            if (rc == CP_OK && c.w16_hi && c.w16_lo) {
                const int coff = (int)i * cout_each;
                rc = cp_launch_weight_scale(tmp, cout_each, cin * kh * kw, c.wfwd + coff, c.winv + coff, nullptr);
                if (rc == CP_OK)
                    rc = cp_launch_pack_weight16(tmp, c.w16_hi, c.w16_lo, cout_each, cin, kh * kw, c.Kpad16, coff, c.wfwd,
                                                 nullptr);
            }
            hip_ok(hipDeviceSynchronize());
            (void)hipFree(tmp);
            if (rc != CP_OK) status = rc;
        }
        if (c.w16_hi && c.w16_lo && ((kh == 3 && kw == 3) || (kh == 1 && kw == 1)) && status == CP_OK) frag_copies(c);
        if (c.scale16 && status == CP_OK) {  // no affine yet: scale16 = winv (set_affine folds a scale in later)
            const int rc = cp_launch_scale16(nullptr, c.winv, c.scale16, c.CoutPad, nullptr);
            hip_ok(hipDeviceSynchronize());
            if (rc != CP_OK) status = rc;
        }
        return c;
    }
    // scale/shift vectors padded to CoutPad (scale pad = 1, shift pad = 0)
    void set_affine(ConvW& c, const std::vector<float>* scale, const std::vector<float>& shift) {
        std::vector<float> sh(c.CoutPad, 0.f);
        for (size_t i = 0; i < shift.size(); ++i) sh[i] = shift[i];
        c.shift = upload(sh);
        if (scale) {
            std::vector<float> sc(c.CoutPad, 1.f);
            for (size_t i = 0; i < scale->size(); ++i) sc[i] = (*scale)[i];
            c.scale = upload(sc);
            if (c.scale && c.scale16) {
                const int rc = cp_launch_scale16(c.scale, c.winv, c.scale16, c.CoutPad, nullptr);
                hip_ok(hipDeviceSynchronize());
                if (rc != CP_OK) status = rc;
            }
        }
    }
    // eval-mode BatchNorm folded to y = x*scale + shift; optional conv bias folded in as well
    bool bn_fold(const std::string& bn, int c, const std::vector<float>* conv_bias, std::vector<float>& scale,
                 std::vector<float>& shift) {
        const auto* g = get(bn + ".weight", c);
        const auto* b = get(bn + ".bias", c);
        const auto* mu = get(bn + ".running_mean", c);
        const auto* var = get(bn + ".running_var", c);
        if (!g || !b || !mu || !var) return false;
        scale.resize(c);
        shift.resize(c);
        for (int i = 0; i < c; ++i) {
            const double s = (double)(*g)[i] / std::sqrt((double)(*var)[i] + 1e-5);
            double t = (double)(*b)[i] - (double)(*mu)[i] * s;
            if (conv_bias) t += (double)(*conv_bias)[i] * s;
            scale[i] = (float)s;
            shift[i] = (float)t;
        }
        return true;
    }
    void conv_bn(const std::string& key, const std::string& conv, const std::string& bn, int cout, int cin, int k,
                 int cin_pad = 0) {
        ConvW c = pack({conv + ".weight"}, cout, cin, k, k, cin_pad);
        std::vector<float> sc, sh;
        if (bn_fold(bn, cout, nullptr, sc, sh)) set_affine(c, &sc, sh);
        m->convs[key] = c;
    }
    void block(const std::string& p, int cin, int cout) {
        conv_bn(p + ".conv1", p + ".conv1", p + ".bn1", cout, cin, 3);
        conv_bn(p + ".conv2", p + ".conv2", p + ".bn2", cout, cout, 3);
    }
    void tree(const std::string& p, int levels, int cin, int cout, bool level_root, int root_dim = 0) {
        if (root_dim == 0) root_dim = 2 * cout;
        if (level_root) root_dim += cin;
        if (levels == 1) {
            block(p + ".tree1", cin, cout);
            block(p + ".tree2", cout, cout);
            conv_bn(p + ".root", p + ".root.conv", p + ".root.bn", cout, root_dim, 1);
            if (cin != cout) conv_bn(p + ".project", p + ".project.0", p + ".project.1", cout, cin, 1);
        } else {
            tree(p + ".tree1", levels - 1, cin, cout, false, 0);
            tree(p + ".tree2", levels - 1, cout, cout, false, root_dim + cout);
            // the outer project of a 2-level tree never influences the output (Tree.forward :214-217)
        }
    }
    void deform(const std::string& p, int chi, int cho) {
        DeformW d;
        d.offset = pack({p + ".conv.conv_offset_mask.weight"}, 27, chi, 3, 3);
        if (const auto* b = get(p + ".conv.conv_offset_mask.bias", 27)) set_affine(d.offset, nullptr, *b);
        d.main = pack({p + ".conv.weight"}, cho, chi, 3, 3);
        const auto* bias = get(p + ".conv.bias", cho);
        std::vector<float> sc, sh;
        if (bias && bn_fold(p + ".actf.0", cho, bias, sc, sh)) set_affine(d.main, &sc, sh);
        m->deforms[p] = d;
    }
    void ida(const std::string& p, int o, const std::vector<int>& channels, const std::vector<int>& up_f) {
        for (size_t i = 1; i < channels.size(); ++i) {
            const std::string k = std::to_string(i);
            deform(p + ".proj_" + k, channels[i], o);
            deform(p + ".node_" + k, o, o);
            const int f = up_f[i];
            if (const auto* w = get(p + ".up_" + k + ".weight", (size_t)o * 4 * f * f)) m->ups[p + ".up_" + k] = upload(*w);
        }
    }
    // ---- stacked hourglass (large_hourglass.py) ----
    void hg_residual(const std::string& p, int cin, int cout, int stride) {
        conv_bn(p + ".conv1", p + ".conv1", p + ".bn1", cout, cin, 3);
        conv_bn(p + ".conv2", p + ".conv2", p + ".bn2", cout, cout, 3);
        if (stride != 1 || cin != cout) conv_bn(p + ".skip", p + ".skip.0", p + ".skip.1", cout, cin, 1);
    }
    void hg_kp(const std::string& p, int n, const int* dims, const int* mods) {
        const int cur = dims[0], nxt = dims[1], cm = mods[0], nm = mods[1];
        for (int i = 0; i < cm; ++i) hg_residual(p + ".up1." + std::to_string(i), cur, cur, 1);
        for (int i = 0; i < cm; ++i) hg_residual(p + ".low1." + std::to_string(i), i == 0 ? cur : nxt, nxt, i == 0 ? 2 : 1);
        if (n > 1) hg_kp(p + ".low2", n - 1, dims + 1, mods + 1);
        else
            for (int i = 0; i < nm; ++i) hg_residual(p + ".low2." + std::to_string(i), nxt, nxt, 1);
        for (int i = 0; i < cm; ++i) hg_residual(p + ".low3." + std::to_string(i), nxt, i < cm - 1 ? nxt : cur, 1);
    }
    void run_hourglass() {
        static const int dims[6] = {256, 256, 384, 384, 384, 512}, mods[6] = {2, 2, 2, 2, 2, 4};
        conv_bn("pre.0", "pre.0.conv", "pre.0.bn", 128, 3, 7, 4);
        hg_residual("pre.1", 128, 256, 2);
        for (int k = 0; k < 2; ++k) {
            const std::string ks = std::to_string(k);
            hg_kp("kps." + ks, 5, dims, mods);
            conv_bn("cnvs." + ks, "cnvs." + ks + ".conv", "cnvs." + ks + ".bn", 256, 256, 3);
        }
        hg_residual("inters.0", 256, 256, 1);
        conv_bn("inters_.0", "inters_.0.0", "inters_.0.1", 256, 256, 1);
        conv_bn("cnvs_.0", "cnvs_.0.0", "cnvs_.0.1", 256, 256, 1);
        // heads of the LAST stack only: the detector takes model(x)[-1] (object_pose.py:135); the first stack's head
        // tensors do not feed anything downstream
        for (auto& h : m->heads) {
            HeadW hw;
            hw.name = h.first;
            hw.classes = h.second;
            const std::string b = h.first + ".1";
            hw.c0 = pack({b + ".0.conv.weight"}, 256, 256, 3, 3);
            if (const auto* bias = get(b + ".0.conv.bias", 256)) set_affine(hw.c0, nullptr, *bias);
            hw.c1 = pack({b + ".1.weight"}, h.second, 256, 1, 1);
            if (const auto* bias = get(b + ".1.bias", h.second)) set_affine(hw.c1, nullptr, *bias);
            if (h.second <= 32 && hw.c0.w16_hi) {
                if (const auto* w1 = get(b + ".1.weight", (size_t)h.second * 256)) {
                    float* tmp = upload(*w1);
                    hw.w2_hi = dev_alloc((size_t)256 * 32 / 2);
                    hw.w2_lo = dev_alloc((size_t)256 * 32 / 2);
                    hw.w2_inv = dev_alloc(64);
                    if (tmp && hw.w2_hi && hw.w2_lo && hw.w2_inv) {
                        const int rc = cp_launch_pack_head_w2(tmp, hw.w2_hi, hw.w2_lo, hw.w2_inv, h.second, 256, nullptr);
                        hip_ok(hipDeviceSynchronize());
                        if (rc != CP_OK) status = rc;
                    }
                }
            }
            m->headw.push_back(hw);
        }
        group_heads();
    }

    // concatenate the fused heads' operands for the grouped launch (all heads must be fusable and of one shape)
    void group_heads() {
        auto& g = m->head_group;
        const size_t n = m->headw.size();
        if (n < 2 || n > CP_MAX_HEAD_GROUP || status != CP_OK) return;
        const ConvW& c = m->headw[0].c0;
        for (const HeadW& h : m->headw)
            if (!h.w2_hi || !h.w2_lo || !h.w2_inv || !h.c0.w16f_hi || !h.c0.w16f_lo || !h.c0.scale16 || !h.c0.shift ||
                h.c0.Cin != c.Cin || h.c0.CoutPad != c.CoutPad || h.c0.Cout != c.CoutPad || h.c0.Kpad16 != c.Kpad16 ||
                h.c0.KH != 3 || h.c0.KW != 3 || c.CoutPad % 128 != 0)
                return;
        const size_t wbytes = (size_t)c.CoutPad * c.Kpad16 * 2, w2bytes = (size_t)c.CoutPad * 32 * 2;
        g.w16f_hi = dev_alloc(n * wbytes / 4, false);
        g.w16f_lo = dev_alloc(n * wbytes / 4, false);
        g.w2_hi = dev_alloc(n * w2bytes / 4, false);
        g.w2_lo = dev_alloc(n * w2bytes / 4, false);
        g.scale16 = dev_alloc(n * c.CoutPad, false);
        g.shift = dev_alloc(n * c.CoutPad, false);
        g.w2_inv = dev_alloc(n * 64, false);
        if (!g.w16f_hi || !g.w16f_lo || !g.w2_hi || !g.w2_lo || !g.scale16 || !g.shift || !g.w2_inv) return;
        for (size_t i = 0; i < n; ++i) {
            const HeadW& h = m->headw[i];
            const auto d2d = hipMemcpyDeviceToDevice;
            hip_ok(hipMemcpy((char*)g.w16f_hi + i * wbytes, h.c0.w16f_hi, wbytes, d2d));
            hip_ok(hipMemcpy((char*)g.w16f_lo + i * wbytes, h.c0.w16f_lo, wbytes, d2d));
            hip_ok(hipMemcpy((char*)g.w2_hi + i * w2bytes, h.w2_hi, w2bytes, d2d));
            hip_ok(hipMemcpy((char*)g.w2_lo + i * w2bytes, h.w2_lo, w2bytes, d2d));
            hip_ok(hipMemcpy(g.scale16 + i * c.CoutPad, h.c0.scale16, (size_t)c.CoutPad * 4, d2d));
            hip_ok(hipMemcpy(g.shift + i * c.CoutPad, h.c0.shift, (size_t)c.CoutPad * 4, d2d));
            hip_ok(hipMemcpy(g.w2_inv + i * 64, h.w2_inv, 64 * 4, d2d));
        }
        g.Cin = c.Cin;
        g.hid = c.CoutPad;
        g.Kpad16 = c.Kpad16;
        g.ok = status == CP_OK;
    }

    // weight fragments for the direct low-channel kernels (f16x3 mode); the folded BatchNorm comes from the ConvW
    void lowc(const std::string& name, const std::string& wname, int kind, int cout, int cin, int k, const std::string& affine = "") {
        const auto* w = get(wname + ".weight", (size_t)cout * cin * k * k);
        if (!w) return;
        float* tmp = upload(*w);
        const size_t halfs = cp_lowc_weight_halfs(kind);
        void* hi = dev_alloc(halfs / 2);
        void* lo = dev_alloc(halfs / 2);
        float* fwd = dev_alloc(cout);
        float* inv = dev_alloc(cout);
        LowcW lw;
        lw.hi = hi;
        lw.lo = lo;
        lw.scale16 = dev_alloc(cout);
        if (!tmp || !hi || !lo || !fwd || !inv || !lw.scale16) return;
        int rc = cp_launch_weight_scale(tmp, cout, cin * k * k, fwd, inv, nullptr);
        if (rc == CP_OK) rc = cp_launch_pack_lowc(kind, tmp, hi, lo, fwd, cin, nullptr);
        // the folded BatchNorm of the same layer lives in the ConvW packed under the same name (conv_bn ran first)
        auto it = m->convs.find(affine.empty() ? name : affine);
        if (rc == CP_OK) rc = cp_launch_scale16(it != m->convs.end() ? it->second.scale : nullptr, inv, lw.scale16, cout, nullptr);
        hip_ok(hipDeviceSynchronize());
        if (rc != CP_OK) status = rc;
        m->lowc[name] = lw;
    }
    void run() {
        conv_bn("base.base_layer", "base.base_layer.0", "base.base_layer.1", 16, 3, 7, 4);
        // previous-frame stems: each exists iff its own flag was set when the checkpoint was made
        // (pose_dla_dcn.py:253-271), i.e. iff its weights were supplied
        const bool has_pre_img = m->params.count("base.pre_img_layer.0.weight") != 0;
        const bool has_pre_hm = m->params.count("base.pre_hm_layer.0.weight") != 0;
        const bool has_pre_hm_hp = m->params.count("base.pre_hm_hp_layer.0.weight") != 0;
        if (has_pre_img) conv_bn("base.pre_img_layer", "base.pre_img_layer.0", "base.pre_img_layer.1", 16, 3, 7, 4);
        if (has_pre_hm) conv_bn("base.pre_hm_layer", "base.pre_hm_layer.0", "base.pre_hm_layer.1", 16, 1, 7, 4);
        if (has_pre_hm_hp) conv_bn("base.pre_hm_hp_layer", "base.pre_hm_hp_layer.0", "base.pre_hm_hp_layer.1", 16, 8, 7, 8);
        conv_bn("base.level0", "base.level0.0", "base.level0.1", 16, 16, 3);
        conv_bn("base.level1", "base.level1.0", "base.level1.1", 32, 16, 3);
        // f16x3 fragments of the same layers (after conv_bn: they take the folded BatchNorm from the ConvW)
        lowc("base.base_layer", "base.base_layer.0", 0, 16, 3, 7);
        lowc("base.level0", "base.level0.0", 1, 16, 16, 3);
        {   // fused stem + level0 (lowc2_kernel): level0's weights in kernel-row order, and the bound that replaces the measured
            // |max| of the tensor between the two layers: |relu(bn(conv(x)))_c| <= |s_c| sum|w_c| max|x| + |t_c|
            lowc("base.level0.rows", "base.level0.0", 4, 16, 16, 3, "base.level0");
            const auto* w = get("base.base_layer.0.weight", (size_t)16 * 3 * 49);
            std::vector<float> sc, sh;
            if (w && bn_fold("base.base_layer.1", 16, nullptr, sc, sh)) {
                double bl = 0, bs = 0;
                for (int c = 0; c < 16; ++c) {
                    double l1 = 0;
                    for (int i = 0; i < 147; ++i) l1 += std::fabs((double)(*w)[(size_t)c * 147 + i]);
                    bl = std::max(bl, std::fabs((double)sc[c]) * l1);
                    bs = std::max(bs, std::fabs((double)sh[c]));
                }
                m->stem_bound_l = (float)(bl * 1.0001);
                m->stem_bound_s = (float)(bs * 1.0001);
            }
        }
        lowc("base.level1", "base.level1.0", 2, 32, 16, 3);
        lowc("base.level1.rows", "base.level1.0", 5, 32, 16, 3, "base.level1");   // the row-streaming level1 kernel's fragments
        if (has_pre_img) lowc("base.pre_img_layer", "base.pre_img_layer.0", 0, 16, 3, 7);
        if (has_pre_hm) lowc("base.pre_hm_layer", "base.pre_hm_layer.0", 0, 16, 1, 7);
        if (has_pre_hm_hp) lowc("base.pre_hm_hp_layer", "base.pre_hm_hp_layer.0", 3, 16, 8, 7);
        tree("base.level2", 1, 32, 64, false);
        tree("base.level3", 2, 64, 128, true);
        tree("base.level4", 2, 128, 256, true);
        tree("base.level5", 1, 256, 512, true);
        ida("dla_up.ida_0", 256, {256, 512}, {1, 2});
        ida("dla_up.ida_1", 128, {128, 256, 256}, {1, 2, 2});
        ida("dla_up.ida_2", 64, {64, 128, 128, 128}, {1, 2, 2, 2});
        ida("ida_up", 64, {64, 128, 256}, {1, 2, 4});
        if (m->gru) {
            const std::string c = "convGRU.cell0.";
            m->gru_x = pack({c + "Wir.weight", c + "Wiz.weight", c + "Win.weight"}, 64, 64, 3, 3);
            std::vector<float> b;
            for (const char* g : {"Wir", "Wiz", "Win"}) {
                const auto* v = get(c + g + ".bias", 64);
                if (v) b.insert(b.end(), v->begin(), v->end());
            }
            if (b.size() == 192) set_affine(m->gru_x, nullptr, b);
            m->gru_h = pack({c + "Whr.weight", c + "Whz.weight", c + "Whn.weight"}, 64, 64, 3, 3);
            {   // the same weights in the fused-gate order: N tile t (96 wide) = [r | z | n] of channels 32t .. 32t+31
                const size_t halfs = (size_t)192 * 576;
                m->gru_h16_hi = dev_alloc(halfs / 2);
                m->gru_h16_lo = dev_alloc(halfs / 2);
                m->gru_h16_fwd = dev_alloc(192);
                m->gru_h16_inv = dev_alloc(192);
                const char* gates[3] = {"Whr", "Whz", "Whn"};
                for (int g = 0; g < 3 && m->gru_h16_hi && m->gru_h16_lo && m->gru_h16_fwd && m->gru_h16_inv; ++g) {
                    const auto* w = get(c + gates[g] + ".weight", (size_t)64 * 64 * 9);
                    if (!w) break;
                    float* tmp = upload(*w);
                    if (!tmp) break;
                    for (int t = 0; t < 2; ++t) {
                        const int row = t * 96 + g * 32;
                        int rc = cp_launch_weight_scale(tmp + (size_t)32 * t * 64 * 9, 32, 64 * 9, m->gru_h16_fwd + row,
                                                        m->gru_h16_inv + row, nullptr);
                        if (rc == CP_OK)
                            rc = cp_launch_pack_weight16(tmp + (size_t)32 * t * 64 * 9, m->gru_h16_hi, m->gru_h16_lo, 32, 64, 9,
                                                         576, row, m->gru_h16_fwd, nullptr);
                        if (rc != CP_OK) status = rc;
                    }
                    hip_ok(hipDeviceSynchronize());
                }
                if (m->gru_h16_hi && m->gru_h16_lo && status == CP_OK) {
                    m->gru_h16f_hi = dev_alloc(halfs / 2);
                    m->gru_h16f_lo = dev_alloc(halfs / 2);
                    if (m->gru_h16f_hi && m->gru_h16f_lo) {
                        int rc = cp_launch_frag16_repack(m->gru_h16_hi, m->gru_h16f_hi, 192, 576, nullptr);
                        if (rc == CP_OK) rc = cp_launch_frag16_repack(m->gru_h16_lo, m->gru_h16f_lo, 192, 576, nullptr);
                        hip_ok(hipDeviceSynchronize());
                        if (rc != CP_OK) status = rc;
                    }
                }
            }
        }
        const int hc = m->head_conv;
        for (auto& h : m->heads) {
            HeadW hw;
            hw.name = h.first;
            hw.classes = h.second;
            const std::string last = h.first + (m->gru ? ".3" : ".2");
            hw.c0 = pack({h.first + ".0.weight"}, hc, 64, 3, 3);
            if (const auto* b = get(h.first + ".0.bias", hc)) set_affine(hw.c0, nullptr, *b);
            hw.c1 = pack({last + ".weight"}, h.second, hc, 1, 1, 0, m->gru ? 32 : 0);
            if (const auto* b = get(last + ".bias", h.second)) set_affine(hw.c1, nullptr, *b);
            if (!m->gru && hc % 128 == 0 && h.second <= 32 && hw.c0.w16_hi) {
                // conv3x3 -> ReLU -> conv1x1 head: keep the 1x1 weights as MFMA fragments for the fused kernel too
                if (const auto* w1 = get(last + ".weight", (size_t)h.second * hc)) {
                    float* tmp = upload(*w1);
                    hw.w2_hi = dev_alloc((size_t)hc * 32 / 2);
                    hw.w2_lo = dev_alloc((size_t)hc * 32 / 2);
                    hw.w2_inv = dev_alloc(64);
                    if (tmp && hw.w2_hi && hw.w2_lo && hw.w2_inv) {
                        const int rc = cp_launch_pack_head_w2(tmp, hw.w2_hi, hw.w2_lo, hw.w2_inv, h.second, hc, nullptr);
                        hip_ok(hipDeviceSynchronize());
                        if (rc != CP_OK) status = rc;
                    }
                }
            }
            if (m->gru) {
                const auto* g = get(h.first + ".1.weight", hc);
                const auto* be = get(h.first + ".1.bias", hc);
                if (g && be) {
                    hw.gn_gamma = upload(*g);
                    hw.gn_beta = upload(*be);
                }
            }
            m->headw.push_back(hw);
        }
        if (!m->gru) group_heads();
    }
};

// ------------------------------------ forward -------------------------------------------------
struct Fwd {
    cp_model* m;
    int B;
    hipStream_t s;
    // GroupNorm fusion hooks for the next conv() call (reset after use)
    double* gn_stats_out = nullptr;
    const float* gn_in_mr = nullptr;
    const float* gn_in_a = nullptr;  // f16x3 form of the same fusion: per (image, channel) a, d planes
    const float* gn_in_d = nullptr;
    const float* gn_in_gamma = nullptr;
    const float* gn_in_beta = nullptr;
    int role = -1;  // CP_ROLE_* of the next conv() call when the shape does not say it (heads, GRU); reset after use
    const unsigned* gn_in_amax = nullptr;  // bound on max|relu(a*x + d)| for the GNIN loader's pre-scale
    // |max| slots of this forward's tensors (f16x3 range-safe scaling, ConvParams::in_amax): one zeroed block at the
    // start of the arena, a slot per Tensor in creation order
    static constexpr int kMaxSlots = CP_AMAX_STRIDE;
    static constexpr size_t kSlotBytes = (size_t)CP_AMAX_SUB * CP_AMAX_STRIDE * sizeof(unsigned);
    Tensor slots_t;
    unsigned* slots = nullptr;
    int nslots = 0;
    void init_slots() {
        slots_t.blk = std::make_shared<Block>(&m->arena, kSlotBytes);
        if (m->dry || m->precision != CP_PREC_F16X3 || (g_dbg & 512)) return;  // 512: A/B switch, operands used unscaled
        slots = (unsigned*)slots_t.ptr();
        if (hipMemsetAsync(slots, 0, kSlotBytes, s) != hipSuccess) chk(CP_ERR_LAUNCH);
    }
    unsigned* new_slot() {
        if (!slots) return nullptr;
        if (nslots >= kMaxSlots) {
            chk(fail(CP_ERR_STATE, "out of |max| slots"));
            return nullptr;
        }
        return slots + nslots++;
    }
    // |max| of a caller-owned input (network images): one extra read of the tensor
    unsigned* input_slot(const float* x, size_t n) {
        unsigned* sl = new_slot();
        if (sl) chk(cp_launch_absmax(x, n, sl, s));
        return sl;
    }

    void chk(int rc) {
        if (rc != CP_OK && m->status == CP_OK) m->status = rc;
    }
    Tensor make(int C, int H, int W) {
        Tensor t;
        t.C = C;
        t.H = H;
        t.W = W;
        t.blk = std::make_shared<Block>(&m->arena, (size_t)B * H * W * C * sizeof(float));
        t.amax = new_slot();
        return t;
    }
    void tap(const char* name, const Tensor& t, int c_valid = 0) {
        if (m->dry || !m->tap_name || std::strcmp(name, m->tap_name) != 0) return;
        const int C = c_valid ? c_valid : t.C;
        chk(cp_launch_nhwc_to_nchw(t.ptr(), m->tap_out, B, C, t.H, t.W, t.C, s));
        if (m->tap_dims) {
            m->tap_dims[0] = C;
            m->tap_dims[1] = t.H;
            m->tap_dims[2] = t.W;
        }
    }
    void tap(const std::string& name, const Tensor& t, int c_valid = 0) { tap(name.c_str(), t, c_valid); }

    // conv3x3 (+bias, ReLU) -> conv1x1 (+bias, optional sigmoid) of a prediction head in one kernel + a slice reduction;
    // returns false (nothing launched) when the launch would want split-K or the shapes are not eligible
    bool fused_head(const HeadW& hw, const Tensor& x, bool sigmoid, float* out_nchw) {
        const ConvW& w = hw.c0;
        ConvParams p;
        std::memset(&p, 0, sizeof(p));
        p.nsrc = 1;
        p.src[0] = x.ptr();
        p.src_c[0] = x.C;
        if (x.C != w.CinP) return false;
        p.Cin = x.C;
        p.B = B;
        p.H = x.H;
        p.W = x.W;
        p.Ho = x.H + 2 - w.KH + 1;
        p.Wo = x.W + 2 - w.KW + 1;
        p.KH = w.KH;
        p.KW = w.KW;
        p.stride = 1;
        p.pad = 1;
        p.K = w.K;
        p.Kpad = w.Kpad;
        p.wp = w.wp;
        p.Cout = w.Cout;
        p.CoutPad = w.CoutPad;
        p.scale = w.scale16;
        p.shift = w.shift;
        p.in_amax[0] = x.amax;
        p.fuse_w2_inv = hw.w2_inv;
        p.act = CP_ACT_RELU;
        p.w16_hi = w.w16_hi;
        p.w16_lo = w.w16_lo;
        p.w16f_hi = w.w16f_hi;
        p.w16f_lo = w.w16f_lo;
        p.Kpad16 = w.Kpad16;
        p.splitk = 1;
        p.dbg = g_dbg;
        p.fuse_w2_hi = hw.w2_hi;
        p.fuse_w2_lo = hw.w2_lo;
        p.fuse_c2 = hw.classes;
        if (w.KH != 3 || w.KW != 3 || !cp_head_fuse_supported(p, hw.classes)) return false;
        int tiles = 0, nk = 0;
        cp_conv_geometry(p, true, &tiles, &nk);
        if (tiles < kSplitTiles && nk >= 8) return false;  // small launches keep the split-K path (conv())
        const int slices = p.CoutPad / 128;
        Tensor slabs = make(slices * hw.classes, p.Ho, p.Wo);
        p.fuse_out = slabs.ptr();
        auto launch = [&]() -> int {
            int rc = cp_launch_conv16_fused_head(p, s);
            if (rc == CP_OK)
                rc = cp_launch_head_reduce(slabs.ptr(), hw.c1.shift, out_nchw, slices, hw.classes, B, p.Ho * p.Wo,
                                           sigmoid ? 1 : 0, s);
            return rc;
        };
        if (m->dry) return true;
        if (m->profile) {
            cp_model::ProfRec r;
            r.variant = cp_halo16_fused_head_supported(p) ? CP_VARIANT_HALO_HEAD : CP_VARIANT_FUSED_HEAD;
            r.role = CP_ROLE_HEAD;
            const double M = (double)B * p.Ho * p.Wo;
            r.flops = 2.0 * M * w.Cout * (double)(w.KH * w.KW * w.Cin) + 2.0 * M * hw.classes * (double)w.Cout;
            // algorithmic bytes: input once + final maps once + both weight sets (the hidden tensor is not counted:
            // it is not part of the head's definition, only of the unfused implementation)
            r.bytes = 4.0 * ((double)B * x.H * x.W * w.Cin + M * hw.classes + (double)w.KH * w.KW * w.Cin * w.Cout +
                             (double)w.Cout * hw.classes);
            r.M = (int)M; r.N = w.Cout; r.K = w.KH * w.KW * w.Cin; r.kh = w.KH; r.stride = 1;
            r.e0 = m->get_event();
            r.e1 = m->get_event();
            (void)hipEventRecord(r.e0, s);
            chk(launch());
            (void)hipEventRecord(r.e1, s);
            m->prof.push_back(r);
        } else {
            chk(launch());
        }
        return true;
    }

    // every fused head of the model in one launch + one slice reduction (cp_model::head_group); false = nothing launched
    bool fused_heads_grouped(const Tensor& x, float* const* head_out, int sigmoid_hm) {
        const auto& g = m->head_group;
        const int n = (int)m->headw.size();
        if (!g.ok || m->precision != CP_PREC_F16X3 || m->tap_name || (g_dbg & 32) || (g_dbg & 16777216) || x.C != g.Cin)
            return false;
        ConvParams p;
        std::memset(&p, 0, sizeof(p));
        p.nsrc = 1;
        p.src[0] = x.ptr();
        p.src_c[0] = x.C;
        p.Cin = x.C;
        p.B = B;
        p.H = p.Ho = x.H;
        p.W = p.Wo = x.W;
        p.KH = p.KW = 3;
        p.stride = 1;
        p.pad = 1;
        p.K = p.Kpad = p.Kpad16 = g.Kpad16;
        p.Cout = p.CoutPad = n * g.hid;
        p.scale = g.scale16;
        p.shift = g.shift;
        p.in_amax[0] = x.amax;
        p.act = CP_ACT_RELU;
        p.w16f_hi = g.w16f_hi;
        p.w16f_lo = g.w16f_lo;
        p.splitk = 1;
        p.dbg = g_dbg;
        p.fuse_w2_hi = g.w2_hi;
        p.fuse_w2_lo = g.w2_lo;
        p.fuse_w2_inv = g.w2_inv;
        p.fuse_ngroups = n;
        p.fuse_gtiles = g.hid / 128;
        p.fuse_out = (float*)0x1000;  // placeholder for the eligibility check
        // the kernel walks a head's hidden tiles and writes the finished maps itself (cp_set_debug 1: slabs + reduction launch)
        // 2: every head of a patch in one workgroup (one staging for all of them) -- when the patches alone fill the device
        // several times over; below that (small batches) one workgroup per patch and head
        p.fuse_final = (g.Cin == 64 && g.hid == 256 && !(g_dbg & 1)) ? (((g_dbg & 2) || B * (x.H / 8) * (x.W / 16) < 2048) ? 1 : 2) : 0;
        if (!cp_halo16_fused_head_supported(p)) return false;
        if (B * (x.H / 8) * (x.W / 16) * (p.CoutPad / 128) < kSplitTiles) return false;  // small maps: per-head split-K path
        HeadReduceGroup rg;
        std::memset(&rg, 0, sizeof(rg));
        rg.n = n;
        rg.slices = p.fuse_gtiles;
        int planes = 0;
        double flops = 0.0, bytes = 0.0;
        const double M = (double)B * x.H * x.W;
        for (int i = 0; i < n; ++i) {
            const HeadW& hw = m->headw[i];
            p.fuse_gc2[i] = rg.c2[i] = hw.classes;
            p.fuse_gbase[i] = rg.base[i] = planes;
            planes += p.fuse_gtiles * hw.classes;
            rg.sigmoid[i] = sigmoid_hm && (hw.name == "hm" || hw.name == "hm_hp");
            rg.bias[i] = hw.c1.shift;
            rg.out[i] = m->dry ? nullptr : head_out[i];
            p.fuse_gsig[i] = rg.sigmoid[i];
            p.fuse_gbias[i] = rg.bias[i];
            p.fuse_gout[i] = rg.out[i];
            flops += 2.0 * M * g.hid * (9.0 * g.Cin) + 2.0 * M * hw.classes * (double)g.hid;
            bytes += 4.0 * (M * hw.classes + 9.0 * g.Cin * g.hid + (double)g.hid * hw.classes);
        }
        bytes += 4.0 * M * g.Cin;  // the shared input is read once
        Tensor slabs;
        if (!p.fuse_final) slabs = make(planes, x.H, x.W);
        if (m->dry) return true;
        p.fuse_out = p.fuse_final ? nullptr : slabs.ptr();
        auto launch = [&]() -> int {
            int rc = cp_launch_halo16_fused_head(p, s);
            if (rc == CP_OK && !p.fuse_final) rc = cp_launch_head_reduce_grouped(slabs.ptr(), rg, B, x.H * x.W, s);
            return rc;
        };
        if (m->profile) {
            cp_model::ProfRec r;
            r.variant = CP_VARIANT_HALO_HEAD;
            r.role = CP_ROLE_HEAD;
            r.flops = flops;
            r.bytes = bytes;
            r.M = (int)M; r.N = p.CoutPad; r.K = 9 * g.Cin; r.kh = 3; r.stride = 1;
            r.e0 = m->get_event();
            r.e1 = m->get_event();
            (void)hipEventRecord(r.e0, s);
            chk(launch());
            (void)hipEventRecord(r.e1, s);
            m->prof.push_back(r);
        } else {
            chk(launch());
        }
        return true;
    }

    // generic conv into a fresh NHWC tensor (or into user NCHW memory when out_nchw != nullptr)
    Tensor conv(const ConvW& w, const std::vector<const Tensor*>& srcs, int stride, int pad, int act,
                const Tensor* res = nullptr, const Tensor* offmask = nullptr, int act_from = 0,
                float* out_nchw = nullptr, int out_ld = 0) {
        const Tensor& x0 = *srcs[0];
        ConvParams p;
        std::memset(&p, 0, sizeof(p));
        int cin = 0;
        p.nsrc = (int)srcs.size();
        for (int i = 0; i < p.nsrc; ++i) {
            p.src[i] = srcs[i]->ptr();
            p.src_c[i] = srcs[i]->C;
            cin += srcs[i]->C;
        }
        if (cin != w.CinP) {
            chk(fail(CP_ERR_INVALID, "conv: channel mismatch"));
            return Tensor();
        }
        p.Cin = cin;
        p.B = B;
        p.H = x0.H;
        p.W = x0.W;
        p.Ho = (x0.H + 2 * pad - w.KH) / stride + 1;
        p.Wo = (x0.W + 2 * pad - w.KW) / stride + 1;
        p.KH = w.KH;
        p.KW = w.KW;
        p.stride = stride;
        p.pad = pad;
        p.K = w.K;
        p.Kpad = w.Kpad;
        p.wp = w.wp;
        p.Cout = w.Cout;
        p.CoutPad = w.CoutPad;
        p.scale = w.scale;
        p.shift = w.shift;
        p.res = res ? res->ptr() : nullptr;
        p.res_ld = res ? res->C : 0;
        p.act = act;
        p.act_from = act_from;
        p.offmask = offmask ? offmask->ptr() : nullptr;
        p.dbg = g_dbg;
        p.gn_stats = gn_stats_out;
        p.gn_groups = 32;
        p.gn_cpg = w.Cout / 32 > 0 ? w.Cout / 32 : 1;
        if (gn_in_a) {
            p.gn_in_a = gn_in_a;
            p.gn_in_d = gn_in_d;
        }
        if (gn_in_mr) {
            p.gn_in_mr = gn_in_mr;
            p.gn_in_gamma = gn_in_gamma;
            p.gn_in_beta = gn_in_beta;
            p.gn_cpg = w.Cin / 32;
        }
        p.w16f_hi = w.w16f_hi;
        p.w16f_lo = w.w16f_lo;
        p.w16_hi = w.w16_hi;
        p.w16_lo = w.w16_lo;
        p.Kpad16 = w.Kpad16;
        const bool use16 = m->precision == CP_PREC_F16X3 && cp_conv16_supported(p);
        if (use16) {
            p.scale = w.scale16;
            for (int i = 0; i < p.nsrc; ++i) p.in_amax[i] = srcs[i]->amax;
            if (p.gn_in_a) p.in_amax[0] = gn_in_amax;
        }
        if (p.gn_in_a && !use16) {  // the per-channel affine form only exists in the f16x3 1x1 kernel
            chk(fail(CP_ERR_INVALID, "conv: GroupNorm affine input without an f16x3 kernel"));
            return Tensor();
        }
        Tensor out;
        if (out_nchw) {
            p.out = out_nchw;
            p.store = CP_STORE_NCHW;
            p.ldo = out_ld;
            p.coff = 0;
        } else {
            // offset/mask maps keep their padded width so the DCN loader can index [pixel*32 + c]
            const int cstore = (act == CP_ACT_SIGMOID_FROM) ? w.CoutPad : w.Cout;
            out = make(cstore, p.Ho, p.Wo);
            p.out = out.ptr();
            p.out_amax = act == CP_ACT_SIGMOID_FROM ? nullptr : out.amax;  // offset/mask maps are never a GEMM operand
            p.store = CP_STORE_NHWC;
            p.ldo = cstore;
            p.coff = 0;
        }
        // deterministic split-K for launches with too few output tiles to fill 256 CUs (low-resolution layers at
        // small batch): slices write slabs, a small epilogue kernel sums them in order
        Tensor partial;
        p.splitk = 1;
        {
            int tiles = 0, nk = 0;
            cp_conv_geometry(p, use16, &tiles, &nk);
            // small launches of the f16x3 path run on 64 x 64 tiles (four times the workgroups per slice): a quarter of the
            // slices and of the slab bytes (slices x M x Cout x 4) for the same workgroup count, and no split at all where that
            // already gives kSplitTiles workgroups.  cp_set_debug 16: the 128-row tiles everywhere (A/B runs).
            // Measured at B = 1 / 2 / 4 / 8 (profiles/NOTES.md): pays up to 32 tiles of 128 rows, up to 64 when K is short.
            if (use16 && tiles > 0 && (tiles <= 32 || (tiles <= 64 && nk <= 36)) && nk >= 8 && !p.gn_stats && !p.gn_in_a &&
                p.CoutPad % 64 == 0 && w.Cout >= 64 && !(g_dbg & 16)) {
                p.tile_m = p.tile_n = 64;
                cp_conv_geometry(p, use16, &tiles, &nk);
            }
            // (64 x 64 tiles are a quarter of the work each: they are still cut along K below one workgroup per CU)
            // (cp_set_debug 536870912 -- tests: the row-streaming kernel at any size -- keeps such a layer whole)
            const bool force_strm = use16 && (g_dbg & 536870912) && cp_strm16_supported(p);
            if (tiles > 0 && tiles < (p.tile_m == 64 ? 256 : kSplitTiles) && nk >= 8 && !p.gn_stats && !force_strm) {
                int want = (kSplitTarget + tiles - 1) / tiles;
                if (want > nk / 2) want = nk / 2;
                if (want > 32) want = 32;
                if (want > 1) {
                    const int per = (nk + want - 1) / want;
                    const int sk = (nk + per - 1) / per;
                    if (sk > 1) {
                        p.splitk = sk;
                        partial = make(sk * p.CoutPad, p.Ho, p.Wo);
                        p.partial = partial.ptr();
                    }
                }
            }
        }
        auto launch = [&]() -> int {
            int rc = use16 ? cp_launch_conv16(p, s) : cp_launch_conv(p, s);
            if (rc == CP_OK && p.splitk > 1) rc = cp_launch_splitk_epilogue(p, s);
            return rc;
        };
        if (!m->dry) {
            if (m->profile) {
                cp_model::ProfRec r;
                r.variant = use16 ? cp_conv16_variant(p) : cp_conv_variant(p);
                r.role = role >= 0 ? role : offmask ? CP_ROLE_DCN : act == CP_ACT_SIGMOID_FROM ? CP_ROLE_DCN_OFFSET
                         : (w.KH == 1 && w.KW == 1) ? CP_ROLE_CONV1X1 : CP_ROLE_CONV;
                const double M = (double)B * p.Ho * p.Wo;
                const int cin_real = w.Cin;  // un-padded input channels
                r.flops = 2.0 * M * w.Cout * (double)(w.KH * w.KW * cin_real);
                // algorithmic bytes: input once + output once + weights (+ offsets/mask for DCN, + residual)
                r.bytes = 4.0 * ((double)B * x0.H * x0.W * cin_real + M * w.Cout +
                                 (double)w.KH * w.KW * cin_real * w.Cout + (offmask ? M * 27 : 0.0) +
                                 (res ? M * w.Cout : 0.0));
                r.M = (int)M; r.N = w.Cout; r.K = w.KH * w.KW * cin_real; r.kh = w.KH; r.stride = stride;
                r.e0 = m->get_event();
                r.e1 = m->get_event();
                (void)hipEventRecord(r.e0, s);
                chk(launch());
                (void)hipEventRecord(r.e1, s);
                m->prof.push_back(r);
            } else {
                chk(launch());
            }
        }
        gn_stats_out = nullptr;
        gn_in_mr = nullptr;
        gn_in_a = nullptr;
        gn_in_d = nullptr;
        gn_in_amax = nullptr;
        role = -1;
        return out;
    }
    const ConvW& cw(const std::string& k) { return m->convs.at(k); }

    Tensor maxpool(const Tensor& x) {
        Tensor o = make(x.C, x.H / 2, x.W / 2);
        o.amax = x.amax;  // max|maxpool(x)| <= max|x|: the input's slot is a valid bound
        if (!m->dry) chk(cp_launch_maxpool2(x.ptr(), o.ptr(), B, x.H, x.W, x.C, s));
        return o;
    }

    Tensor basic_block(const std::string& p, const Tensor& x, int stride, const Tensor& residual) {
        Tensor t = conv(cw(p + ".conv1"), {&x}, stride, 1, CP_ACT_RELU);
        Tensor o = conv(cw(p + ".conv2"), {&t}, 1, 1, CP_ACT_RELU, &residual);
        tap(p, o);
        return o;
    }

    // one-level Tree (Tree.forward with levels == 1); `bottom` may be supplied by the caller when it
    // already computed maxpool(x) (the outer two-level tree needs the same tensor as a root child)
    Tensor tree1(const std::string& p, const Tensor& x, int cin, int cout, int stride, bool level_root,
                 std::vector<const Tensor*> children, const Tensor* bottom_in = nullptr) {
        Tensor bottom_own;
        const Tensor* bottom = &x;
        if (stride > 1) {
            if (bottom_in) bottom = bottom_in;
            else {
                bottom_own = maxpool(x);
                bottom = &bottom_own;
            }
        }
        Tensor proj;
        const Tensor* residual = bottom;
        if (cin != cout) {
            proj = conv(cw(p + ".project"), {bottom}, 1, 0, CP_ACT_NONE);
            residual = &proj;
        }
        if (level_root) children.insert(children.begin(), bottom);
        Tensor x1 = basic_block(p + ".tree1", x, stride, *residual);
        proj = Tensor();
        Tensor x2 = basic_block(p + ".tree2", x1, 1, x1);
        std::vector<const Tensor*> srcs = {&x2, &x1};
        for (auto* c : children) srcs.push_back(c);
        Tensor o = conv(cw(p + ".root"), srcs, 1, 0, CP_ACT_RELU);
        tap(p + ".root", o);
        return o;
    }
    // two-level Tree with level_root = true (base.level3 / base.level4)
    Tensor tree2(const std::string& p, const Tensor& x, int cin, int cout) {
        Tensor bottom = maxpool(x);
        Tensor x1 = tree1(p + ".tree1", x, cin, cout, 2, false, {}, &bottom);
        return tree1(p + ".tree2", x1, cout, cout, 1, false, {&bottom, &x1});
    }

    Tensor deform(const std::string& p, const Tensor& x) {
        const DeformW& d = m->deforms.at(p);
        Tensor om = conv(d.offset, {&x}, 1, 1, CP_ACT_SIGMOID_FROM, nullptr, nullptr, 18);
        tap(p + ".offmask", om, 27);
        Tensor o = conv(d.main, {&x}, 1, 1, CP_ACT_RELU, nullptr, &om);
        tap(p, o);
        return o;
    }
    Tensor upsample_add(const std::string& p, const Tensor& x, int f, const Tensor& add) {
        Tensor o = make(x.C, x.H * f, x.W * f);
        if (!m->dry)
            chk(cp_launch_upsample_add(x.ptr(), m->ups.at(p), add.ptr(), o.ptr(), B, x.H, x.W, x.C, f, o.amax, s));
        return o;
    }
    // IDAUp.forward: layers[i] = node(up(proj(layers[i])) + layers[i-1])
    void ida(const std::string& p, std::vector<Tensor>& layers, int startp, int endp, const std::vector<int>& up_f) {
        for (int i = startp + 1; i < endp; ++i) {
            const std::string k = std::to_string(i - startp);
            Tensor t = deform(p + ".proj_" + k, layers[i]);
            Tensor u = upsample_add(p + ".up_" + k, t, up_f[i - startp], layers[i - 1]);
            t = Tensor();
            layers[i] = deform(p + ".node_" + k, u);
        }
    }

    // the network's first layers through lowc.hip (f16x3 mode only); returns an invalid Tensor when not applicable
    Tensor lowc(const std::string& name, int kind, const float* in, int H, int W, int planes, const unsigned* in_amax) {
        if (m->precision != CP_PREC_F16X3 || (g_dbg & 64)) return Tensor();
        const int Ho = kind == 2 ? (H - 1) / 2 + 1 : H, Wo = kind == 2 ? (W - 1) / 2 + 1 : W;
        // level1: the row-streaming kernel (lowc1s_kernel, kind 5) from the batch at which bands of >= 8 output rows give every wave
        // slot of the chip a strip (cp_set_debug 262144: never, 1073741824: at any size -- tests)
        if (kind == 2 && m->lowc.count(name + ".rows") && !(g_dbg & 262144) &&
            ((g_dbg & 1073741824) || (long)B * ((Wo + 31) / 32) * ((Ho + 7) / 8) >= 2048))
            kind = 5;
        auto it = m->lowc.find(kind == 5 ? name + ".rows" : name);
        if (it == m->lowc.end()) return Tensor();
        const ConvW& w = cw(name);
        const bool stem = kind == 0 || kind == 3;  // 3: the 8-plane stem (two groups of 4 planes)
        const bool l1 = kind == 2 || kind == 5;
        const int cout = l1 ? 32 : 16, cin = stem ? planes : 16, k = stem ? 7 : 3;
        Tensor out = make(cout, Ho, Wo);
        if (m->dry) return out;
        auto launch = [&]() {
            return cp_launch_lowc(kind, in, out.ptr(), it->second.hi, it->second.lo, it->second.scale16, w.shift, in_amax,
                                  out.amax, B, H, W, planes, s);
        };
        if (m->profile) {
            cp_model::ProfRec r;
            r.variant = kind == 5 ? CP_VARIANT_LOWC1S : CP_VARIANT_LOWC0 + (kind == 3 ? 0 : kind);
            r.role = CP_ROLE_LOWC;
            const double M = (double)B * Ho * Wo;
            r.flops = 2.0 * M * cout * (double)(k * k * cin);
            r.bytes = 4.0 * ((double)B * H * W * cin + M * cout + (double)k * k * cin * cout);
            r.M = (int)M; r.N = cout; r.K = k * k * cin; r.kh = k; r.stride = l1 ? 2 : 1;
            r.e0 = m->get_event();
            r.e1 = m->get_event();
            (void)hipEventRecord(r.e0, s);
            chk(launch());
            (void)hipEventRecord(r.e1, s);
            m->prof.push_back(r);
        } else {
            chk(launch());
        }
        return out;
    }

    Tensor to_nhwc(const float* nchw, int C, int Cpad, int H, int W) {
        Tensor t = make(Cpad, H, W);
        t.amax = nullptr;  // re-laid network inputs only feed the exact-f32 stems
        if (!m->dry) chk(cp_launch_nchw_to_nhwc(nchw, t.ptr(), B, C, H, W, Cpad, s));
        return t;
    }

    // ---- stacked hourglass forward (large_hourglass.py:50-78, 129-189, 266-286) ----
    Tensor hg_residual(const std::string& p, const Tensor& x, int stride) {
        Tensor t = conv(cw(p + ".conv1"), {&x}, stride, 1, CP_ACT_RELU);
        if (m->convs.count(p + ".skip")) {
            Tensor sk = conv(cw(p + ".skip"), {&x}, stride, 0, CP_ACT_NONE);
            return conv(cw(p + ".conv2"), {&t}, 1, 1, CP_ACT_RELU, &sk);  // relu(bn2(conv2) + skip)
        }
        return conv(cw(p + ".conv2"), {&t}, 1, 1, CP_ACT_RELU, &x);
    }
    Tensor hg_seq(const std::string& p, Tensor x, int n, int first_stride) {
        for (int i = 0; i < n; ++i) x = hg_residual(p + "." + std::to_string(i), x, i == 0 ? first_stride : 1);
        return x;
    }
    Tensor hg_kp(const std::string& p, const Tensor& x, int n, const int* mods) {
        const int cm = mods[0], nm = mods[1];
        Tensor up1 = hg_seq(p + ".up1", x, cm, 1);
        Tensor low = hg_seq(p + ".low1", x, cm, 2);
        low = n > 1 ? hg_kp(p + ".low2", low, n - 1, mods + 1) : hg_seq(p + ".low2", low, nm, 1);
        low = hg_seq(p + ".low3", low, cm, 1);
        Tensor out = make(up1.C, up1.H, up1.W);
        if (!m->dry)
            chk(cp_launch_upsample2_nearest_add(up1.ptr(), low.ptr(), out.ptr(), B, low.H, low.W, low.C, out.amax, s));
        tap(p, out);
        return out;
    }
    void run_hourglass(int H, int W, const float* images, float* const* head_out, int sigmoid_hm) {
        static const int mods[6] = {2, 2, 2, 2, 2, 4};
        init_slots();
        Tensor inter;
        {
            Tensor in = to_nhwc(images, 3, 4, H, W);
            Tensor p0 = conv(cw("pre.0"), {&in}, 2, 3, CP_ACT_RELU);
            inter = hg_residual("pre.1", p0, 2);
        }
        tap("pre", inter);
        Tensor cnv;
        for (int k = 0; k < 2; ++k) {
            const std::string ks = std::to_string(k);
            Tensor kp = hg_kp("kps." + ks, inter, 5, mods);
            cnv = conv(cw("cnvs." + ks), {&kp}, 1, 1, CP_ACT_RELU);
            tap("cnvs." + ks, cnv);
            if (k == 0) {
                Tensor a = conv(cw("inters_.0"), {&inter}, 1, 0, CP_ACT_NONE);
                Tensor b = conv(cw("cnvs_.0"), {&cnv}, 1, 0, CP_ACT_RELU, &a);  // relu(inters_(inter) + cnvs_(cnv))
                inter = hg_residual("inters.0", b, 1);
            }
        }
        if (fused_heads_grouped(cnv, head_out, sigmoid_hm)) return;
        for (size_t i = 0; i < m->headw.size(); ++i) {
            const HeadW& hw = m->headw[i];
            const bool sg = sigmoid_hm && (hw.name == "hm" || hw.name == "hm_hp");
            if (hw.w2_hi && m->precision == CP_PREC_F16X3 && !m->tap_name && !(g_dbg & 32) &&
                fused_head(hw, cnv, sg, m->dry ? (float*)0x1000 : head_out[i]))
                continue;
            role = CP_ROLE_HEAD;
            Tensor hid = conv(hw.c0, {&cnv}, 1, 1, CP_ACT_RELU);
            role = CP_ROLE_HEAD_FINAL;
            conv(hw.c1, {&hid}, 1, 0, sg ? CP_ACT_SIGMOID : CP_ACT_NONE, nullptr, nullptr, 0,
                 m->dry ? (float*)0x1000 : head_out[i], hw.classes);
        }
    }

    void run(int H, int W, const float* images, const float* pre_img, const float* pre_hm, const float* pre_hm_hp,
             float* const* head_out, int sigmoid_hm) {
        init_slots();
        const bool use_lowc = m->precision == CP_PREC_F16X3 && !(g_dbg & 64) && m->lowc.count("base.base_layer");
        // stem + level0 in one launch when nothing is added to the stem's output (no previous-frame stems) and nobody asks for it
        // (cp_set_debug 134217728: the two kernels, A/B runs and tests)
        const bool no_pre = m->dry ? m->dry_variant == 1 : (!pre_img && !pre_hm && !pre_hm_hp);
        const bool fuse01 = use_lowc && no_pre && m->lowc.count("base.level0.rows") && m->stem_bound_l > 0.f &&
                            !(g_dbg & 134217728) && !(m->tap_name && std::strcmp(m->tap_name, "base.base_layer") == 0) &&
                            !m->convs.count("base.pre_img_layer") && !m->convs.count("base.pre_hm_layer") && !m->convs.count("base.pre_hm_hp_layer");
        Tensor l0f;
        if (fuse01) {
            const unsigned* in_slot = !m->dry ? input_slot(images, (size_t)B * 3 * H * W) : nullptr;
            l0f = make(16, H, W);
            if (!m->dry) {
                const LowcW& a = m->lowc["base.base_layer"];
                const LowcW& c = m->lowc["base.level0.rows"];
                auto launch = [&]() {
                    return cp_launch_lowc_fused(images, l0f.ptr(), a.hi, a.lo, a.scale16, cw("base.base_layer").shift, c.hi, c.lo, c.scale16,
                                                cw("base.level0").shift, m->stem_bound_l, m->stem_bound_s, in_slot, l0f.amax, B, H, W, 3, s);
                };
                if (m->profile) {
                    cp_model::ProfRec r;
                    r.variant = CP_VARIANT_LOWC01;
                    r.role = CP_ROLE_LOWC;
                    const double M = (double)B * H * W;
                    r.flops = 2.0 * M * 16 * (147.0 + 144.0);
                    r.bytes = 4.0 * (M * 3 + M * 16);
                    r.M = (int)M; r.N = 16; r.K = 147 + 144; r.kh = 7; r.stride = 1;
                    r.e0 = m->get_event();
                    r.e1 = m->get_event();
                    (void)hipEventRecord(r.e0, s);
                    chk(launch());
                    (void)hipEventRecord(r.e1, s);
                    m->prof.push_back(r);
                } else {
                    chk(launch());
                }
            }
        }
        Tensor x0 = fuse01 ? Tensor() : lowc("base.base_layer", 0, images, H, W, 3,
                         use_lowc && !m->dry ? input_slot(images, (size_t)B * 3 * H * W) : nullptr);
        if (!x0.valid() && !fuse01) {
            Tensor in = to_nhwc(images, 3, 4, H, W);
            x0 = conv(cw("base.base_layer"), {&in}, 1, 3, CP_ACT_RELU);
        }
        // a dry run (workspace query) is sized for every stem the model has
        if (m->dry) {
            if (!m->convs.count("base.pre_img_layer")) pre_img = nullptr;
            if (!m->convs.count("base.pre_hm_layer")) pre_hm = nullptr;
            if (!m->convs.count("base.pre_hm_hp_layer")) pre_hm_hp = nullptr;
        }
        if ((pre_img && !m->convs.count("base.pre_img_layer")) || (pre_hm && !m->convs.count("base.pre_hm_layer")) ||
            (pre_hm_hp && !m->convs.count("base.pre_hm_hp_layer"))) {
            chk(fail(CP_ERR_INVALID, "a previous-frame input was given to a model built without that pre_* layer"));
            return;
        }
        if (pre_img || pre_hm || pre_hm_hp) {
            Tensor a, b, c;
            if (pre_img) {
                a = lowc("base.pre_img_layer", 0, pre_img, H, W, 3,
                         use_lowc && !m->dry ? input_slot(pre_img, (size_t)B * 3 * H * W) : nullptr);
                if (!a.valid()) {
                    Tensor in = to_nhwc(pre_img, 3, 4, H, W);
                    a = conv(cw("base.pre_img_layer"), {&in}, 1, 3, CP_ACT_RELU);
                }
            }
            if (pre_hm) {
                b = lowc("base.pre_hm_layer", 0, pre_hm, H, W, 1,
                         use_lowc && !m->dry ? input_slot(pre_hm, (size_t)B * H * W) : nullptr);
                if (!b.valid()) {
                    Tensor in = to_nhwc(pre_hm, 1, 4, H, W);
                    b = conv(cw("base.pre_hm_layer"), {&in}, 1, 3, CP_ACT_RELU);
                }
            }
            if (pre_hm_hp) {
                c = lowc("base.pre_hm_hp_layer", 3, pre_hm_hp, H, W, 8,
                         use_lowc && !m->dry ? input_slot(pre_hm_hp, (size_t)B * 8 * H * W) : nullptr);
                if (!c.valid()) {
                    Tensor in = to_nhwc(pre_hm_hp, 8, 8, H, W);
                    c = conv(cw("base.pre_hm_hp_layer"), {&in}, 1, 3, CP_ACT_RELU);
                }
            }
            // x = x + pre_img_layer(..) + pre_hm_layer(..) + pre_hm_hp_layer(..)  (left-to-right, :312-318)
            std::vector<const Tensor*> adds;
            for (Tensor* t : {&a, &b, &c})
                if (t->valid()) adds.push_back(t);
            Tensor sum = make(16, H, W);
            if (!m->dry)
                chk(cp_launch_add_relu_sum(x0.ptr(), adds[0]->ptr(), adds.size() > 1 ? adds[1]->ptr() : nullptr,
                                           adds.size() > 2 ? adds[2]->ptr() : nullptr, sum.ptr(),
                                           (size_t)B * H * W * 16, sum.amax, s));
            x0 = sum;
        }
        if (!fuse01) tap("base.base_layer", x0);
        Tensor l0 = fuse01 ? l0f : lowc("base.level0", 1, x0.ptr(), H, W, 16, x0.amax);
        l0f = Tensor();  // (one owner: the block returns to the arena when l0 is dropped below)
        if (!l0.valid()) l0 = conv(cw("base.level0"), {&x0}, 1, 1, CP_ACT_RELU);
        tap("base.level0", l0);
        x0 = Tensor();
        Tensor l1 = lowc("base.level1", 2, l0.ptr(), H, W, 16, l0.amax);
        if (!l1.valid()) l1 = conv(cw("base.level1"), {&l0}, 2, 1, CP_ACT_RELU);
        tap("base.level1", l1);
        l0 = Tensor();
        std::vector<Tensor> L(6);
        L[2] = tree1("base.level2", l1, 32, 64, 2, false, {});
        l1 = Tensor();
        L[3] = tree2("base.level3", L[2], 64, 128);
        L[4] = tree2("base.level4", L[3], 128, 256);
        L[5] = tree1("base.level5", L[4], 256, 512, 2, true, {});
        tap("base.level2", L[2]);
        tap("base.level3", L[3]);
        tap("base.level4", L[4]);
        tap("base.level5", L[5]);

        // DLAUp.forward (:437-443): out = [after ida_2, after ida_1, after ida_0, L5]
        ida("dla_up.ida_0", L, 4, 6, {1, 2});
        Tensor o2 = L[5];  // 256 @ 1/8... (after ida_0: 256 ch at L4 resolution)
        ida("dla_up.ida_1", L, 3, 6, {1, 2, 2});
        Tensor o1 = L[5];
        ida("dla_up.ida_2", L, 2, 6, {1, 2, 2, 2});
        Tensor o0 = L[5];
        for (auto& t : L) t = Tensor();
        // DLASeg.forward (:531-536): ida_up over [o0, o1, o2]
        std::vector<Tensor> y = {o0, o1, o2};
        o0 = o1 = o2 = Tensor();
        ida("ida_up", y, 0, 3, {1, 2, 4});
        Tensor feat = y[2];
        y.clear();
        tap("feat", feat);

        std::vector<Tensor> gru_out;
        if (m->gru) {
            const int steps = m->tracking ? 4 : 3;
            role = CP_ROLE_GRU;
            Tensor x3 = conv(m->gru_x, {&feat}, 1, 1, CP_ACT_NONE);
            Tensor h;
            for (int st = 0; st < steps; ++st) {
                Tensor hn = make(64, feat.H, feat.W);
                const size_t M = (size_t)B * feat.H * feat.W;
                if (st == 0) {
                    // h0 = 0: the three hidden-side convolutions are identically zero (convGRU.py:51,80-84)
                    if (!m->dry) chk(cp_launch_gru_gate(x3.ptr(), nullptr, nullptr, hn.ptr(), M, hn.amax, s));
                } else if (m->precision == CP_PREC_F16X3 && m->gru_h16_hi && !(g_dbg & 256) &&
                           (size_t)M * 192 * 4 < (size_t)0xf0000000u) {
                    // hidden-side convolution with the gate arithmetic in its epilogue: h3 is never written
                    if (!m->dry) {
                        ConvParams p;
                        std::memset(&p, 0, sizeof(p));
                        p.nsrc = 1;
                        p.src[0] = h.ptr();
                        p.src_c[0] = 64;
                        p.Cin = 64;
                        p.B = B; p.H = feat.H; p.W = feat.W; p.Ho = feat.H; p.Wo = feat.W;
                        p.KH = 3; p.KW = 3; p.stride = 1; p.pad = 1;
                        p.K = 576; p.Kpad = 576; p.Kpad16 = 576;
                        p.Cout = 192; p.CoutPad = 192;
                        p.w16_hi = m->gru_h16_hi; p.w16_lo = m->gru_h16_lo;
                        p.w16f_hi = m->gru_h16f_hi; p.w16f_lo = m->gru_h16f_lo;
                        p.dbg = g_dbg;
                        p.scale = m->gru_h16_inv;  // 2^-e of the fused-order weight rows (the hidden-side convs have no affine)
                        p.in_amax[0] = h.amax;
                        p.out_amax = hn.amax;
                        p.out = hn.ptr();
                        p.gru_x3 = x3.ptr();
                        p.gru_hprev = h.ptr();
                        p.splitk = 1;
                        auto launch = [&]() { return cp_launch_conv16_gru(p, s); };
                        if (m->profile) {
                            cp_model::ProfRec r;
                            r.variant = cp_halo16_gru_supported(p) ? CP_VARIANT_HALO_GRU : CP_VARIANT_GRU;
                            r.role = CP_ROLE_GRU;
                            r.flops = 2.0 * (double)M * 192 * 576;
                            r.bytes = 4.0 * ((double)M * (64 + 192 + 64 + 64) + 576.0 * 192);
                            r.M = (int)M; r.N = 192; r.K = 576; r.kh = 3; r.stride = 1;
                            r.e0 = m->get_event();
                            r.e1 = m->get_event();
                            (void)hipEventRecord(r.e0, s);
                            chk(launch());
                            (void)hipEventRecord(r.e1, s);
                            m->prof.push_back(r);
                        } else {
                            chk(launch());
                        }
                    }
                } else {
                    role = CP_ROLE_GRU;
                    Tensor h3 = conv(m->gru_h, {&h}, 1, 1, CP_ACT_NONE);
                    if (!m->dry) chk(cp_launch_gru_gate(x3.ptr(), h3.ptr(), h.ptr(), hn.ptr(), M, hn.amax, s));
                }
                h = hn;
                gru_out.push_back(h);
                tap(("convGRU.step" + std::to_string(st)).c_str(), h);
            }
        }

        // GroupNorm statistics of every head (32 groups x (sum, sumsq) doubles per image = 128 floats per image and head):
        // one block, zeroed by one memset per forward pass instead of one per head
        if (!m->gru && fused_heads_grouped(feat, head_out, sigmoid_hm)) return;
        Tensor stats_all;
        if (m->gru) {
            stats_all = make(128 * (int)m->headw.size(), 1, 1);
            if (!m->dry && hipMemsetAsync(stats_all.ptr(), 0, sizeof(double) * 64 * B * m->headw.size(), s) != hipSuccess)
                chk(CP_ERR_LAUNCH);
        }
        for (size_t i = 0; i < m->headw.size(); ++i) {
            const HeadW& hw = m->headw[i];
            const Tensor* src = &feat;
            if (m->gru) {
                int r = -1;
                const std::string& n = hw.name;
                if (m->tracking) {
                    if (n == "tracking" || n == "tracking_hp") r = 0;
                    else if (n == "hm" || n == "wh" || n == "reg") r = 1;
                    else if (n == "hm_hp" || n == "hp_offset" || n == "hps" || n == "hps_uncertainty") r = 2;
                    else if (n == "scale" || n == "scale_uncertainty") r = 3;
                } else {
                    if (n == "hm" || n == "wh" || n == "reg") r = 0;
                    else if (n == "hm_hp" || n == "hp_offset" || n == "hps") r = 1;
                    else if (n == "scale") r = 2;
                }
                if (r < 0) {  // unreachable: cp_model_create refuses heads outside the routing table
                    chk(fail(CP_ERR_STATE, "head without a ConvGRU step"));
                    continue;
                }
                src = &gru_out[r];
            }
            Tensor mr, ad;
            double* stats = m->gru && !m->dry ? (double*)stats_all.ptr() + (size_t)i * 64 * B : nullptr;
            const bool fuse_gn = m->gru && ((src->H * src->W) % 32 == 0) && hw.c0.Cout % 32 == 0 && (hw.c0.Cout / 32) % 4 == 0;
            if (m->gru) {
                mr = make(64, 1, 1);
                if (fuse_gn && !m->dry) gn_stats_out = stats;
            }
            const bool sg = sigmoid_hm && (hw.name == "hm" || hw.name == "hm_hp");
            if (!m->gru && hw.w2_hi && m->precision == CP_PREC_F16X3 && !m->tap_name && !(g_dbg & 32) &&
                fused_head(hw, *src, sg, m->dry ? (float*)0x1000 : head_out[i]))
                continue;
            role = CP_ROLE_HEAD;
            Tensor hid = conv(hw.c0, {src}, 1, 1, m->gru ? CP_ACT_NONE : CP_ACT_RELU);
            if (m->gru) {
                if (fuse_gn) {
                    // statistics came out of the conv epilogue; normalise + affine + ReLU happens in the 1x1 loader
                    const bool affine16 = m->precision == CP_PREC_F16X3 && hw.c1.w16_hi && (hid.H * hid.W) % 128 == 0 &&
                                          !(g_dbg & 128);
                    if (affine16) {
                        // f16x3 1x1 kernel: the normalisation pre-folded to y = relu(a*x + d) per (image, channel)
                        ad = make(2 * hid.C, 1, 1);
                        if (!m->dry) {
                            float* ap = ad.ptr();
                            float* dp = ap + (size_t)B * hid.C;
                            unsigned* bound = new_slot();
                            chk(cp_launch_gn_affine((const double*)stats, hw.gn_gamma, hw.gn_beta, ap, dp, B, hid.C, 32,
                                                    (double)hid.H * hid.W * (hid.C / 32), 1e-5f, hid.amax, bound, s));
                            gn_in_a = ap;
                            gn_in_d = dp;
                            gn_in_amax = bound;
                        }
                    } else if (!m->dry) {
                        chk(cp_launch_gn_finalize((const double*)stats, mr.ptr(), B * 32,
                                                  (double)hid.H * hid.W * (hid.C / 32), 1e-5f, s));
                        gn_in_mr = mr.ptr();
                        gn_in_gamma = hw.gn_gamma;
                        gn_in_beta = hw.gn_beta;
                    }
                } else if (!m->dry) {
                    // in place: the slot keeps the larger of the raw and the normalised |max| -- a valid bound
                    chk(cp_launch_groupnorm_relu(hid.ptr(), hw.gn_gamma, hw.gn_beta, stats, B,
                                                 hid.H * hid.W, hid.C, 32, 1e-5f, hid.amax, s));
                }
            }
            role = CP_ROLE_HEAD_FINAL;
            if (gn_in_a && !(g_dbg & 131072) && hid.C % 64 == 0 && hid.C <= 256 && (hid.H * hid.W) % 64 == 0 &&
                ((size_t)B * hid.H * hid.W) % 256 == 0 && hw.classes <= 16 && !m->tap_name) {
                // float32 vector-ALU kernel (ewise.hip: gn_final_kernel): the layer is an HBM stream of the hidden tensor
                auto launch = [&]() -> int {
                    return cp_launch_gn_final(hid.ptr(), gn_in_a, gn_in_d, hw.c1.wp, hw.c1.shift, head_out[i], B, hid.H * hid.W,
                                              hid.C, hw.classes, hw.c1.CoutPad, sg ? 1 : 0, s);
                };
                if (m->profile) {
                    cp_model::ProfRec r;
                    r.variant = CP_VARIANT_GN_FINAL;
                    r.role = CP_ROLE_HEAD_FINAL;
                    const double M = (double)B * hid.H * hid.W;
                    r.flops = 2.0 * M * hw.classes * (double)hid.C;
                    r.bytes = 4.0 * (M * hid.C + M * hw.classes + (double)hid.C * hw.classes);
                    r.M = (int)M; r.N = hw.classes; r.K = hid.C; r.kh = 1; r.stride = 1;
                    r.e0 = m->get_event();
                    r.e1 = m->get_event();
                    (void)hipEventRecord(r.e0, s);
                    chk(launch());
                    (void)hipEventRecord(r.e1, s);
                    m->prof.push_back(r);
                } else {
                    chk(launch());
                }
                gn_in_a = gn_in_d = nullptr;
                gn_in_amax = nullptr;
                role = -1;
                continue;
            }
            conv(hw.c1, {&hid}, 1, 0, sg ? CP_ACT_SIGMOID : CP_ACT_NONE, nullptr, nullptr, 0,
                 m->dry ? (float*)0x1000 : head_out[i], hw.classes);
        }
    }
};

int forward_impl(cp_model* m, hipStream_t stream, int B, int H, int W, const float* images, const float* pre_img,
                 const float* pre_hm, const float* pre_hm_hp, float* const* head_out, int sigmoid_hm, void* ws,
                 size_t ws_bytes, bool dry) {
    if (!m || !m->finalized) return fail(CP_ERR_STATE, "model not finalized");
    if (B < 1 || H % 32 || W % 32 || H < 32 || W < 32) return fail(CP_ERR_INVALID, "H and W must be multiples of 32");
    if (m->hourglass && (H % 128 || W % 128))
        return fail(CP_ERR_INVALID, "hourglass: H and W must be multiples of 128 (stride 4, then five stride-2 levels)");
    m->arena.reset(dry ? nullptr : ws, ws_bytes);
    m->dry = dry;
    m->status = CP_OK;
    Fwd f{m, B, stream};
    if (m->hourglass) f.run_hourglass(H, W, images, head_out, sigmoid_hm);
    else f.run(H, W, images, pre_img, pre_hm, pre_hm_hp, head_out, sigmoid_hm);
    if (!dry && m->arena.overflow)
        return fail(CP_ERR_INVALID, "workspace too small: " + std::to_string(ws_bytes) + " bytes given, this launch sequence peaks at " +
                                        std::to_string(m->arena.peak));
    return m->status;
}

}  // namespace

// ============================================ C ABI ==============================================
extern "C" {

const char* cp_version(void) { return "centerpose_hip 0.3.0 (gfx950; f32 and split-f16 MFMA)"; }
int cp_abi_version(void) { return CP_ABI_VERSION; }
static_assert(CP_NUM_KERNEL_VARIANTS == CP_NUM_CONV_VARIANTS, "public and internal kernel-variant counts");
int cp_num_kernel_variants(void) { return CP_NUM_KERNEL_VARIANTS; }
int cp_num_roles(void) { return CP_NUM_ROLES; }
const char* cp_last_error(void) { return g_err.c_str(); }

int cp_model_create(const char* arch, int tracking_task, int num_heads, const char* const* head_names,
                    const int* head_classes, int head_conv, cp_model** out) {
    if (!arch || !out || num_heads < 1 || !head_names || !head_classes) return fail(CP_ERR_INVALID, "null argument");
    std::string a(arch);
    if (a != "dla_34" && a != "dlav1_34" && a != "hourglass")
        return fail(CP_ERR_INVALID, "arch must be dla_34, dlav1_34 or hourglass");
    if (a == "hourglass" && tracking_task)
        return fail(CP_ERR_INVALID, "the hourglass takes a single frame (large_hourglass.py:266)");
    if (head_conv <= 0 || head_conv % 32 != 0) return fail(CP_ERR_INVALID, "head_conv must be a positive multiple of 32");
    cp_model* m = new cp_model();
    m->arch = a;
    m->gru = (a == "dlav1_34");
    m->hourglass = (a == "hourglass");
    m->tracking = tracking_task != 0;
    m->head_conv = head_conv;
    for (int i = 0; i < num_heads; ++i) m->heads.push_back({head_names[i], head_classes[i]});
    if (m->gru) {
        // ConvGRU models route each head to a fixed step (pose_dla_dcn.py:545-563); the reference leaves any other head
        // out of its output dict (the detector then fails with a KeyError).  Refuse it here instead of returning an
        // unwritten tensor.
        static const char* pose[] = {"hm", "wh", "reg", "hm_hp", "hp_offset", "hps", "scale"};
        static const char* track[] = {"tracking", "tracking_hp", "hps_uncertainty", "scale_uncertainty"};
        for (auto& h : m->heads) {
            bool ok = false;
            for (const char* n : pose) ok = ok || h.first == n;
            if (m->tracking)
                for (const char* n : track) ok = ok || h.first == n;
            if (!ok) {
                const std::string msg = "dlav1_34: head '" + h.first + "' has no ConvGRU step in the reference routing (" +
                                        (m->tracking ? "tracking" : "non-tracking") + " table, pose_dla_dcn.py:545-563)";
                delete m;
                return fail(CP_ERR_INVALID, msg);
            }
        }
    }
    *out = m;
    return CP_OK;
}

int cp_model_set_param(cp_model* m, const char* name, const float* host_data, int64_t numel) {
    if (!m || !name || !host_data || numel < 0) return fail(CP_ERR_INVALID, "null argument");
    if (m->finalized) return fail(CP_ERR_STATE, "model already finalized");
    m->params[name] = std::vector<float>(host_data, host_data + numel);
    return CP_OK;
}

int cp_model_finalize(cp_model* m) {
    if (!m) return fail(CP_ERR_INVALID, "null model");
    if (m->finalized) return CP_OK;
    Packer pk{m};
    if (m->hourglass) pk.run_hourglass();
    else pk.run();
    pk.hip_ok(hipDeviceSynchronize());
    if (pk.status != CP_OK)
        return fail(pk.status, (pk.status == CP_ERR_STATE ? "missing or mis-shaped parameter: " : "finalize failed: ") + pk.missing);
    m->params.clear();
    m->finalized = true;
    return CP_OK;
}

int cp_set_debug(int flags) {
    g_dbg = flags;
    return CP_OK;
}

int cp_set_default_precision(int precision) {
    if (precision != CP_PREC_F32 && precision != CP_PREC_F16X3) return fail(CP_ERR_INVALID, "precision must be 0 or 1");
    g_default_precision = precision;
    return CP_OK;
}

int cp_model_set_precision(cp_model* m, int precision) {
    if (!m || (precision != CP_PREC_F32 && precision != CP_PREC_F16X3)) return fail(CP_ERR_INVALID, "bad argument");
    m->precision = precision;
    m->ws_cached = 0;  // (the two arithmetic modes run different launch sequences)
    return CP_OK;
}

int cp_model_profile(cp_model* m, int enable) {
    if (!m) return fail(CP_ERR_INVALID, "null model");
    m->profile = enable != 0;
    return CP_OK;
}

// Drains the recorded launches (synchronises their events).  out[v*4 + {0,1,2,3}] = {launches, total ms,
// total algorithmic FLOPs, total algorithmic bytes} for kernel variant v (names: cp_conv_variant_name).
int cp_model_profile_read(cp_model* m, double* out, int num_variants) {
    if (!m || !out || num_variants < CP_NUM_CONV_VARIANTS) return fail(CP_ERR_INVALID, "bad argument");
    for (int i = 0; i < num_variants * 4; ++i) out[i] = 0.0;
    for (int i = 0; i < CP_NUM_ROLES * 4; ++i) m->roles[i] = 0.0;
    const char* dump = getenv("CP_PROFILE_DUMP");  // optional per-launch CSV for kernel tuning
    FILE* df = dump ? fopen(dump, "a") : nullptr;
    for (auto& r : m->prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess)
            return fail(CP_ERR_LAUNCH, "event timing failed");
        if (df)
            fprintf(df, "%s,%d,%d,%d,%d,%d,%.4f,%.2f,%s\n", r.variant >= 0 ? cp_conv_variant_name(r.variant) : "decode", r.M,
                    r.N, r.K, r.kh, r.stride, ms, r.flops / (ms * 1e-3) / 1e12, cp_role_name(r.role));
        if (r.variant >= 0) {
            out[r.variant * 4 + 0] += 1.0;
            out[r.variant * 4 + 1] += ms;
            out[r.variant * 4 + 2] += r.flops;
            out[r.variant * 4 + 3] += r.bytes;
        }
        if (r.role >= 0 && r.role < CP_NUM_ROLES) {
            m->roles[r.role * 4 + 0] += 1.0;
            m->roles[r.role * 4 + 1] += ms;
            m->roles[r.role * 4 + 2] += r.flops;
            m->roles[r.role * 4 + 3] += r.bytes;
        }
        m->event_pool.push_back(r.e0);
        m->event_pool.push_back(r.e1);
    }
    m->prof.clear();
    if (df) fclose(df);
    return CP_OK;
}

const char* cp_kernel_variant_name(int v) { return cp_conv_variant_name(v); }

const char* cp_role_name(int role) {
    static const char* names[CP_NUM_ROLES] = {"conv", "conv1x1", "dcn", "dcn_offset", "head", "head_final", "gru", "lowc",
                                              "decode"};
    return (role >= 0 && role < CP_NUM_ROLES) ? names[role] : "?";
}

int cp_model_profile_roles(cp_model* m, double* out, int num_roles) {
    if (!m || !out || num_roles < CP_NUM_ROLES) return fail(CP_ERR_INVALID, "bad argument");
    for (int i = 0; i < CP_NUM_ROLES * 4; ++i) out[i] = m->roles[i];
    return CP_OK;
}

void cp_model_destroy(cp_model* m) {
    if (!m) return;
    for (auto& r : m->prof) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    for (auto e : m->event_pool) (void)hipEventDestroy(e);
    for (auto& kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
    for (void* p : m->device_allocs) (void)hipFree(p);
    delete m;
}

size_t cp_model_workspace_bytes(cp_model* m, int B, int H, int W) {
    // The launch sequence -- and with it the arena's allocation order -- has variants the caller may select later: the first
    // layers fused or not (engine: fuse01), and a tap request, which turns the fused heads off.  The query runs the dry pass for
    // every combination and returns the largest peak.
    if (m && m->ws_cached && m->ws_key[0] == B && m->ws_key[1] == H && m->ws_key[2] == W && m->ws_key[3] == g_dbg && m->finalized)
        return m->ws_cached;   // (cp_model_detect asks on every call: four dry passes per frame would show in the batch-1 latency)
    size_t peak = 0;
    const char* const tap_before = m->tap_name;
    for (int v = 0; v < 4; ++v) {
        m->dry_variant = v & 1;
        m->tap_name = (v & 2) ? "" : nullptr;   // "" matches no tensor name: only the routing changes
        const int rc = forward_impl(m, nullptr, B, H, W, nullptr, (const float*)1, (const float*)1, (const float*)1, nullptr, 0,
                                    nullptr, 0, true);
        m->dry_variant = 0;
        m->tap_name = tap_before;
        if (rc != CP_OK) return 0;
        if (m->arena.peak > peak) peak = m->arena.peak;
    }
    m->ws_key[0] = B; m->ws_key[1] = H; m->ws_key[2] = W; m->ws_key[3] = g_dbg;
    m->ws_cached = peak;
    return peak;
}

int cp_model_forward(cp_model* m, cp_stream_t stream, int B, int H, int W, const float* images, const float* pre_img,
                     const float* pre_hm, const float* pre_hm_hp, float* const* head_out, int sigmoid_hm,
                     void* workspace, size_t workspace_bytes) {
    if (!images || !head_out || !workspace) return fail(CP_ERR_INVALID, "null argument");
    m->tap_name = nullptr;
    return forward_impl(m, (hipStream_t)stream, B, H, W, images, pre_img, pre_hm, pre_hm_hp, head_out, sigmoid_hm,
                        workspace, workspace_bytes, false);
}

size_t cp_model_detect_workspace_bytes(cp_model* m, int B, int H, int W, int K) {
    const size_t a = cp_model_workspace_bytes(m, B, H, W);
    return a ? align_up(a, 256) + cp_decode_ws_bytes(B, 8, K) : 0;
}

// backbone + heads + sigmoid + decode in one call (what ObjectPoseDetector.process does, object_pose.py:131-165),
// optionally replayed from a captured hipGraph (the ~120 launches of a frame become one graph launch).
int cp_model_detect(cp_model* m, cp_stream_t stream, int B, int H, int W, const float* images, const float* pre_img,
                    const float* pre_hm, const float* pre_hm_hp, float* const* head_out, int K, int rep_mode,
                    int fit_gaussian, float balance, int legacy_bool_mask, float* det, void* workspace,
                    size_t workspace_bytes, int use_graph) {
    if (!m || !images || !head_out || !det || !workspace) return fail(CP_ERR_INVALID, "null argument");
    if (!m->finalized) return fail(CP_ERR_STATE, "model not finalized");
    const size_t model_ws = align_up(cp_model_workspace_bytes(m, B, H, W), 256);
    if (model_ws == 0) return CP_ERR_INVALID;
    if (workspace_bytes < model_ws + cp_decode_ws_bytes(B, 8, K)) return fail(CP_ERR_INVALID, "workspace too small");
    // decode inputs by head name (opts.py:394-426)
    float* hp[11] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const char* names[11] = {"hm", "hps", "wh", "hps_uncertainty", "scale", "scale_uncertainty", "reg", "hm_hp",
                             "hp_offset", "tracking", "tracking_hp"};
    for (size_t i = 0; i < m->headw.size(); ++i)
        for (int j = 0; j < 11; ++j)
            if (m->headw[i].name == names[j]) hp[j] = head_out[i];
    if (!hp[0] || !hp[1] || !hp[2] || !hp[7]) return fail(CP_ERR_INVALID, "detect needs the hm, hps, wh and hm_hp heads");
    hipStream_t s = (hipStream_t)stream;
    auto enqueue = [&]() -> int {
        m->tap_name = nullptr;
        int rc = forward_impl(m, s, B, H, W, images, pre_img, pre_hm, pre_hm_hp, head_out, 1, workspace, model_ws, false);
        if (rc != CP_OK) return rc;
        cp_model::ProfRec r;
        if (m->profile) {
            // algorithmic bytes of the decode: one read of hm + hm_hp, the gathers at the K centres
            // (<= 60 channels) and at the 8K joint peaks (2 channels), the records written
            const double hw = (double)(H / 4) * (W / 4);
            r.variant = -1;
            r.role = CP_ROLE_DECODE;
            r.flops = 0.0;
            r.bytes = (double)B * (9.0 * hw * 4 + K * 60.0 * 4 + 8.0 * K * 2 * 4 + (double)K * CP_DET_STRIDE * 4);
            r.M = B; r.N = K; r.K = (int)hw; r.kh = 0; r.stride = 0;
            r.e0 = m->get_event();
            r.e1 = m->get_event();
            (void)hipEventRecord(r.e0, s);
        }
        rc = cp_launch_decode(s, B, 8, H / 4, W / 4, hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7], hp[8], hp[9],
                              hp[10], K, rep_mode, fit_gaussian, balance, legacy_bool_mask, 0, det,
                              (char*)workspace + model_ws);
        if (m->profile) {
            (void)hipEventRecord(r.e1, s);
            m->prof.push_back(r);
        }
        if (rc != CP_OK) return fail(rc, "detect: decode failed (need K <= 128 <= H*W/16 <= 32768)");
        return rc;
    };
    if (!use_graph || m->profile) return enqueue();
    std::vector<uint64_t> key = {(uint64_t)B, (uint64_t)H, (uint64_t)W, (uint64_t)images, (uint64_t)pre_img,
                                 (uint64_t)pre_hm, (uint64_t)pre_hm_hp, (uint64_t)K, (uint64_t)rep_mode,
                                 (uint64_t)fit_gaussian, (uint64_t)legacy_bool_mask, (uint64_t)det, (uint64_t)workspace,
                                 (uint64_t)m->precision, (uint64_t)(balance * 1e6f), (uint64_t)s, (uint64_t)(unsigned)g_dbg};
    for (size_t i = 0; i < m->headw.size(); ++i) key.push_back((uint64_t)head_out[i]);
    auto it = m->graphs.find(key);
    if (it == m->graphs.end()) {
        if (s == nullptr) return fail(CP_ERR_INVALID, "graph capture needs a non-default stream");
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess)
            return fail(CP_ERR_LAUNCH, "hipStreamBeginCapture failed");
        const int rc = enqueue();
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(s, &g);
        if (rc != CP_OK || e != hipSuccess || !g) {
            if (g) (void)hipGraphDestroy(g);
            return rc != CP_OK ? rc : fail(CP_ERR_LAUNCH, "hipStreamEndCapture failed");
        }
        hipGraphExec_t ex = nullptr;
        if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGraphDestroy(g);
            return fail(CP_ERR_LAUNCH, "hipGraphInstantiate failed");
        }
        (void)hipGraphDestroy(g);
        if (m->graphs.size() >= 16) {  // bound the cache
            for (auto& kv : m->graphs) (void)hipGraphExecDestroy(kv.second);
            m->graphs.clear();
        }
        it = m->graphs.emplace(key, ex).first;
    }
    return hipGraphLaunch(it->second, s) == hipSuccess ? CP_OK : fail(CP_ERR_LAUNCH, "hipGraphLaunch failed");
}

int cp_model_forward_tap(cp_model* m, cp_stream_t stream, int B, int H, int W, const float* images,
                         const float* pre_img, const float* pre_hm, const float* pre_hm_hp, float* const* head_out,
                         int sigmoid_hm, void* workspace, size_t workspace_bytes, const char* tap_name, float* tap_out,
                         int* tap_dims) {
    if (!images || !head_out || !workspace) return fail(CP_ERR_INVALID, "null argument");
    m->tap_name = tap_name;
    m->tap_out = tap_out;
    m->tap_dims = tap_dims;
    int rc = forward_impl(m, (hipStream_t)stream, B, H, W, images, pre_img, pre_hm, pre_hm_hp, head_out, sigmoid_hm,
                          workspace, workspace_bytes, false);
    m->tap_name = nullptr;
    return rc;
}

// ------------------------------------------------------------------------------------------------
size_t cp_conv2d_workspace_bytes(int Cin, int Cout, int KH, int KW) {
    const size_t kpad = align_up((size_t)KH * KW * Cin, 16);
    const size_t cpad = align_up((size_t)Cout, cp_conv_tile_n(Cout));
    // f32 packed weights + (split-f16 path) two binary16 copies + per-channel weight scales (2^e, 2^-e, scale * 2^-e)
    // + the input's |max| slot
    // (+ the fragment-ordered copies of the binary16 weights)
    return align_up(kpad * cpad * sizeof(float), 256) + 4 * align_up(kpad * cpad * 2, 256) +
           3 * align_up(cpad * sizeof(float), 256) + (size_t)CP_AMAX_SUB * CP_AMAX_STRIDE * sizeof(unsigned);
}

int cp_conv2d_nhwc(cp_stream_t stream, const float* x, const float* w, const float* scale, const float* shift,
                   const float* residual, float* out, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                   int stride, int pad, int act, void* workspace, size_t workspace_bytes) {
    if (!x || !w || !out || !workspace) return fail(CP_ERR_INVALID, "null argument");
    if (Cin % 4) return fail(CP_ERR_INVALID, "Cin must be a multiple of 4");
    if (workspace_bytes < cp_conv2d_workspace_bytes(Cin, Cout, KH, KW)) return fail(CP_ERR_INVALID, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int bn = cp_conv_tile_n(Cout);
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.K = KH * KW * Cin;
    p.Kpad = (int)align_up(p.K, 16);
    p.CoutPad = (int)align_up(Cout, bn);
    // scale/shift are read up to CoutPad: only allow un-padded Cout when they are given
    if ((scale || shift) && p.CoutPad != Cout) return fail(CP_ERR_INVALID, "scale/shift need Cout % tile_n == 0");
    float* wp = (float*)workspace;
    if (hipMemsetAsync(wp, 0, (size_t)p.Kpad * p.CoutPad * sizeof(float), s) != hipSuccess) return CP_ERR_LAUNCH;
    int rc = cp_launch_pack_weight(w, wp, Cout, Cin, KH * KW, Cin, p.CoutPad, 0, s);
    if (rc != CP_OK) return rc;
    p.src[0] = x;
    p.src_c[0] = Cin;
    p.nsrc = 1;
    p.Cin = Cin;
    p.B = B;
    p.H = H;
    p.W = W;
    p.Ho = (H + 2 * pad - KH) / stride + 1;
    p.Wo = (W + 2 * pad - KW) / stride + 1;
    p.KH = KH;
    p.KW = KW;
    p.stride = stride;
    p.pad = pad;
    p.wp = wp;
    p.Cout = Cout;
    p.scale = scale;
    p.shift = shift;
    p.res = residual;
    p.res_ld = Cout;
    p.act = act;
    p.out = out;
    p.store = CP_STORE_NHWC;
    p.ldo = Cout;
    p.dbg = g_dbg;
    if (g_default_precision == CP_PREC_F16X3 && Cin % 32 == 0 && KH * KW <= 32 && bn >= 32) {
        char* w16 = (char*)workspace + align_up((size_t)p.Kpad * p.CoutPad * sizeof(float), 256);
        const size_t sz = align_up((size_t)p.Kpad * p.CoutPad * 2, 256);
        if (hipMemsetAsync(w16, 0, 2 * sz, s) != hipSuccess) return CP_ERR_LAUNCH;
        p.w16_hi = w16;
        p.w16_lo = w16 + sz;
        p.Kpad16 = p.K;
        // range-safe operands: per-channel power-of-two weight scale, per-tensor activation scale from one |max| pass
        const size_t csz = align_up((size_t)p.CoutPad * sizeof(float), 256);
        float* wfwd = (float*)(w16 + 2 * sz);
        float* winv = (float*)((char*)wfwd + csz);
        float* sc16 = (float*)((char*)winv + csz);
        unsigned* slot = (unsigned*)((char*)sc16 + csz);
        if (hipMemsetAsync(wfwd, 0, 3 * csz + (size_t)CP_AMAX_SUB * CP_AMAX_STRIDE * sizeof(unsigned), s) != hipSuccess)
            return CP_ERR_LAUNCH;
        rc = cp_launch_weight_scale(w, Cout, Cin * KH * KW, wfwd, winv, s);
        if (rc == CP_OK) rc = cp_launch_pack_weight16(w, (void*)p.w16_hi, (void*)p.w16_lo, Cout, Cin, KH * KW, p.Kpad16, 0, wfwd, s);
        if (rc == CP_OK) rc = cp_launch_scale16(scale, winv, sc16, Cout, s);
        if (rc == CP_OK) rc = cp_launch_absmax(x, (size_t)B * H * W * Cin, slot, s);
        if (rc == CP_OK && p.CoutPad % 32 == 0 && p.Kpad16 % 16 == 0) {
            char* w16f = (char*)slot + (size_t)CP_AMAX_SUB * CP_AMAX_STRIDE * sizeof(unsigned);
            const size_t fsz = align_up((size_t)p.Kpad * p.CoutPad * 2, 256);
            rc = cp_launch_frag16_repack(p.w16_hi, w16f, p.CoutPad, p.Kpad16, s);
            if (rc == CP_OK) rc = cp_launch_frag16_repack(p.w16_lo, w16f + fsz, p.CoutPad, p.Kpad16, s);
            p.w16f_hi = w16f;
            p.w16f_lo = w16f + fsz;
        }
        if (rc != CP_OK) return rc;
        if (cp_conv16_supported(p)) {
            p.scale = sc16;
            p.in_amax[0] = slot;
            return cp_launch_conv16(p, s);
        }
    }
    return cp_launch_conv(p, s);
}

int cp_preprocess(cp_stream_t stream, const unsigned char* image_hwc_bgr, int H, int W, const double* trans6,
                  const float* mean3, const float* std3, float* out_chw, int out_h, int out_w) {
    if (!image_hwc_bgr || !trans6 || !mean3 || !std3 || !out_chw || H < 1 || W < 1 || out_h < 1 || out_w < 1)
        return fail(CP_ERR_INVALID, "bad argument");
    return cp_launch_preprocess(image_hwc_bgr, 1, H, W, trans6, mean3, std3, out_chw, out_h, out_w, (hipStream_t)stream);
}

int cp_preprocess_batch(cp_stream_t stream, const unsigned char* images_bhwc_bgr, int B, int H, int W, const double* trans6,
                        const float* mean3, const float* std3, float* out_bchw, int out_h, int out_w) {
    if (!images_bhwc_bgr || !trans6 || !mean3 || !std3 || !out_bchw || B < 1 || B > 65535 || H < 1 || W < 1 || out_h < 1 || out_w < 1)
        return fail(CP_ERR_INVALID, "bad argument");
    return cp_launch_preprocess(images_bhwc_bgr, B, H, W, trans6, mean3, std3, out_bchw, out_h, out_w, (hipStream_t)stream);
}

int cp_resize_u8(cp_stream_t stream, const unsigned char* image_hwc, int H, int W, int C, unsigned char* out_hwc, int out_h,
                 int out_w) {
    if (!image_hwc || !out_hwc || H < 1 || W < 1 || C < 1 || C > 4 || out_h < 1 || out_w < 1)
        return fail(CP_ERR_INVALID, "bad argument");
    return cp_launch_resize_u8(image_hwc, H, W, C, out_hwc, out_h, out_w, (hipStream_t)stream);
}

int cp_render_gaussians(cp_stream_t stream, const double* recs, int N, float* out, int C, int H, int W, int clear) {
    if (!out || C < 1 || H < 1 || W < 1 || N < 0 || (N > 0 && !recs)) return fail(CP_ERR_INVALID, "bad argument");
    if (clear && hipMemsetAsync(out, 0, (size_t)C * H * W * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return fail(CP_ERR_LAUNCH, "memset failed");
    return cp_launch_render_gaussians(recs, N, out, C, H, W, (hipStream_t)stream);
}

size_t cp_postprocess_workspace_bytes(int B, int K) {
    return B > 0 && K > 0 ? (size_t)B * K * CP_POST_STRIDE * sizeof(double) : 0;
}

int cp_postprocess(cp_stream_t stream, const float* det, int B, int K, const double* meta, double vis_thresh, int nms,
                   float div_scale, double* out, int* count, void* workspace, size_t workspace_bytes) {
    if (!det || !meta || !out || !count || !workspace || B < 1) return fail(CP_ERR_INVALID, "bad argument");
    if (K < 1 || K > 128) return fail(CP_ERR_INVALID, "K must be in [1, 128]");
    if (workspace_bytes < cp_postprocess_workspace_bytes(B, K)) return fail(CP_ERR_INVALID, "workspace too small");
    if (!(div_scale > 0.f)) return fail(CP_ERR_INVALID, "div_scale must be positive");
    return cp_launch_postprocess(det, B, K, meta, vis_thresh, nms, div_scale, out, count, (double*)workspace,
                                 (hipStream_t)stream);
}

size_t cp_pnp_workspace_bytes(int N) { return cp_pnp_ws_bytes(N); }

int cp_pnp_solve(cp_stream_t stream, const float* pts, const float* scale, const double* cam, int N, int npts,
                 double* out, void* workspace, size_t workspace_bytes) {
    if (N == 0) return CP_OK;
    if (!pts || !scale || !cam || !out || !workspace || N < 0) return fail(CP_ERR_INVALID, "null argument");
    if (npts != 8 && npts != 16) return fail(CP_ERR_INVALID, "npts must be 8 or 16");
    if (workspace_bytes < cp_pnp_ws_bytes(N)) return fail(CP_ERR_INVALID, "workspace too small");
    return cp_launch_pnp((hipStream_t)stream, pts, scale, cam, N, npts, out, workspace);
}

// workspace: [pts B*K*16*2 f32][scale B*K*3 f32][cam B*K*4 f64][solver workspace]
size_t cp_pnp_from_post_workspace_bytes(int B, int K) {
    if (B < 1 || K < 1) return 0;
    const size_t n = (size_t)B * K;
    return align_up(n * 32 * 4, 256) + align_up(n * 3 * 4, 256) + align_up(n * 4 * 8, 256) + cp_pnp_ws_bytes((int)n);
}

int cp_pnp_from_post(cp_stream_t stream, const double* post, const int* count, int B, int K, int rep_mode,
                     const double* cam, double* out, void* workspace, size_t workspace_bytes) {
    if (!post || !count || !cam || !out || !workspace || B < 1 || K < 1) return fail(CP_ERR_INVALID, "bad argument");
    if (rep_mode < 0 || rep_mode > 4 || rep_mode == 2)
        return fail(CP_ERR_INVALID, "rep_mode 0, 1, 3 or 4 (2 samples a GMM with numpy's RNG: host only)");
    if (workspace_bytes < cp_pnp_from_post_workspace_bytes(B, K)) return fail(CP_ERR_INVALID, "workspace too small");
    const size_t n = (size_t)B * K;
    const int npts = rep_mode == 1 ? 16 : 8;
    char* w = (char*)workspace;
    float* pts = (float*)w;
    w += align_up(n * 32 * 4, 256);
    float* scale = (float*)w;
    w += align_up(n * 3 * 4, 256);
    double* camn = (double*)w;
    w += align_up(n * 4 * 8, 256);
    int rc = cp_launch_pnp_assemble(post, count, B, K, npts, cam, pts, scale, camn, (hipStream_t)stream);
    if (rc != CP_OK) return fail(rc, "pnp assemble launch failed");
    rc = cp_launch_pnp((hipStream_t)stream, pts, scale, camn, (int)n, npts, out, w);
    return rc == CP_OK ? CP_OK : fail(rc, "pnp launch failed");
}

static_assert(sizeof(cp_track_params) == sizeof(TrackParams), "cp_track_params mirrors TrackParams field by field");

size_t cp_track_state_bytes(int B, int cap) {
    return (B >= 1 && cap >= 1 && cap <= CP_TRACK_CAP) ? cp_track_state_bytes_impl(B, cap) : 0;
}

size_t cp_track_workspace_bytes(int B, int K, int cap) {
    return (B >= 1 && K >= 1 && K <= 128 && cap >= 1 && cap <= CP_TRACK_CAP) ? cp_track_ws_bytes_impl(B, K, cap) : 0;
}

int cp_track_reset(cp_stream_t stream, void* state, int B, int cap) {
    const size_t n = cp_track_state_bytes(B, cap);
    if (!state || n == 0) return fail(CP_ERR_INVALID, "cp_track_reset: bad argument");
    return hipMemsetAsync(state, 0, n, (hipStream_t)stream) == hipSuccess ? CP_OK : fail(CP_ERR_LAUNCH, "memset failed");
}

int cp_track_step(cp_stream_t stream, const cp_track_params* params, const double* vmeta, const double* post, const int* count,
                  const double* det_pnp, int B, void* state, double* render_recs, void* workspace, size_t workspace_bytes) {
    if (!params || !vmeta || !post || !count || !state || !render_recs || !workspace || B < 1)
        return fail(CP_ERR_INVALID, "cp_track_step: null argument");
    TrackParams P;
    std::memcpy(&P, params, sizeof(P));
    if (P.K < 1 || P.K > 128 || P.cap < 1 || P.cap > CP_TRACK_CAP)
        return fail(CP_ERR_INVALID, "cp_track_step: K must be in [1, 128] and cap in [1, CP_TRACK_CAP]");
    if (!P.kalman && !P.scale_pool)
        return fail(CP_ERR_INVALID, "cp_track_step: needs opt.kalman and / or opt.scale_pool (host tracker otherwise)");
    if (P.use_pnp && !det_pnp) return fail(CP_ERR_INVALID, "cp_track_step: use_pnp needs the detections' PnP rows");
    if (P.cat_rule < 0 || P.cat_rule > 2) return fail(CP_ERR_INVALID, "cp_track_step: cat_rule must be 0, 1 or 2");
    if (workspace_bytes < cp_track_workspace_bytes(B, P.K, P.cap)) return fail(CP_ERR_INVALID, "workspace too small");
    const int rc = cp_launch_track_step((hipStream_t)stream, P, vmeta, post, count, P.use_pnp ? det_pnp : nullptr, B, P.K,
                                        state, render_recs, workspace);
    return rc == CP_OK ? CP_OK : fail(rc, "cp_track_step: launch failed");
}

int cp_linear_assignment(const double* cost, int n_rows, int n_cols, int solver, int* match_out) {
    if (n_rows < 0 || n_cols < 0 || (n_rows > 0 && !match_out) || (n_rows > 0 && n_cols > 0 && !cost) || (solver != 1 && solver != 2))
        return fail(CP_ERR_INVALID, "cp_linear_assignment: bad argument (solver: 1 Munkres, 2 scipy LSAP)");
    const size_t ls = (size_t)(n_rows > n_cols ? n_rows : n_cols) + 1;
    auto c = [&](int i, int j) -> double { return cost[(size_t)i * n_cols + j]; };
    try {
        if (solver == 2) {
            std::vector<double> u(ls), v(ls), spc(ls);
            std::vector<int> path(ls), c4r(ls), r4c(ls), rem(ls);
            std::vector<unsigned char> sr(ls), sc(ls);
            const TrkLsapWork W = {u.data(), v.data(), spc.data(), path.data(), c4r.data(), r4c.data(), rem.data(), sr.data(), sc.data()};
            trk_lsap(c, n_rows, n_cols, match_out, W);
        } else {
            std::vector<double> C((size_t)n_rows * n_cols + 1);
            std::vector<unsigned char> marked((size_t)n_rows * n_cols + 1), ru(ls), cu(ls);
            std::vector<int> path(2 * ((size_t)n_rows + n_cols) + 2);
            const TrkMunkresWork W = {C.data(), marked.data(), ru.data(), cu.data(), path.data()};
            trk_munkres(c, n_rows, n_cols, match_out, W);
        }
    } catch (const std::bad_alloc&) {
        return fail(CP_ERR_ALLOC, "cp_linear_assignment: out of host memory");
    }
    return CP_OK;
}

// Sticky per-video overflow counters of the device tracker (list entries dropped because a frame needed more than `cap`
// tracks): dropped_out [B] HOST int32.  Synchronises `stream` (one small copy); call it every few frames, not every frame.
int cp_track_status(cp_stream_t stream, const void* state, int B, int* dropped_out) {
    if (!state || !dropped_out || B < 1) return fail(CP_ERR_INVALID, "cp_track_status: bad argument");
    std::vector<int> hdr(4 + 4 * (size_t)B);
    if (hipMemcpyAsync(hdr.data(), state, hdr.size() * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return fail(CP_ERR_LAUNCH, "cp_track_status: copy failed");
    for (int b = 0; b < B; ++b) dropped_out[b] = hdr[4 + 4 * b + 2];
    return CP_OK;
}

size_t cp_decode_workspace_bytes(int B, int K) { return cp_decode_ws_bytes(B, 8, K); }

int cp_decode(cp_stream_t stream, int B, int H, int W, float* hm, const float* hps, const float* wh,
              const float* hps_uncertainty, const float* scale, const float* scale_uncertainty, const float* reg,
              float* hm_hp, const float* hp_offset, const float* tracking, const float* tracking_hp, int K,
              int rep_mode, int fit_gaussian, float balance, int legacy_bool_mask, int apply_sigmoid, float* det,
              void* workspace, size_t workspace_bytes) {
    if (!hm || !hps || !wh || !hm_hp || !det || !workspace)
        return fail(CP_ERR_INVALID, "hm, hps, wh, hm_hp, det and workspace are required");
    if (B < 1 || rep_mode < 0 || rep_mode > 4) return fail(CP_ERR_INVALID, "bad B / rep_mode");
    if (workspace_bytes < cp_decode_workspace_bytes(B, K)) return fail(CP_ERR_INVALID, "workspace too small");
    int rc = cp_launch_decode((hipStream_t)stream, B, 8, H, W, hm, hps, wh, hps_uncertainty, scale, scale_uncertainty,
                              reg, hm_hp, hp_offset, tracking, tracking_hp, K, rep_mode, fit_gaussian, balance,
                              legacy_bool_mask, apply_sigmoid, det, workspace);
    if (rc != CP_OK) return fail(rc, "decode: unsupported shape (need K <= 128 <= H*W <= 32768, W % 4 == 0) or launch failure");
    return CP_OK;
}

// DCNv2 forward with the reference's NCHW layouts (see header).  Workspace layout:
//   [x NHWC B*H*W*C][offmask NHWC B*H*W*32][y NHWC B*H*W*Co][packed weights][shift CoutPad]
size_t cp_dcnv2_workspace_bytes(int B, int C, int H, int W, int Co) {
    const size_t px = (size_t)B * H * W;
    const size_t cpad = align_up((size_t)Co, cp_conv_tile_n(Co));
    return align_up(px * C * 4, 256) + align_up(px * 32 * 4, 256) + align_up(px * Co * 4, 256) +
           align_up((size_t)9 * C * cpad * 4, 256) + align_up(cpad * 4, 256) + 4 * align_up((size_t)9 * C * cpad * 2, 256) +
           3 * align_up(cpad * 4, 256) + (size_t)CP_AMAX_SUB * CP_AMAX_STRIDE * sizeof(unsigned);
}

}  // extern "C"

namespace {
__global__ void dcn_offmask_pack_kernel(const float* __restrict__ offset, const float* __restrict__ mask,
                                        float* __restrict__ om, int B, int HW) {
    // offset [B,18,HW], mask [B,9,HW] -> om [B,HW,32]
    const size_t total = (size_t)B * HW * 32;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & 31);
        const size_t px = i >> 5;
        const size_t b = px / HW, p = px - b * HW;
        float v = 0.f;
        if (c < 18) v = offset[(b * 18 + c) * HW + p];
        else if (c < 27) v = mask[(b * 9 + (c - 18)) * HW + p];
        om[i] = v;
    }
}
}  // namespace

extern "C" int cp_dcnv2_forward(cp_stream_t stream, const float* input, const float* weight, const float* bias,
                                const float* offset, const float* mask, float* output, int B, int C, int H, int W,
                                int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                int deformable_group, void* workspace, size_t workspace_bytes) {
    if (!input || !weight || !bias || !offset || !mask || !output || !workspace)
        return fail(CP_ERR_INVALID, "null argument");
    if (B < 1 || C < 1 || H < 1 || W < 1 || Co < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1 || ph < 0 || pw < 0 || dh < 1 ||
        dw < 1 || deformable_group < 1 || C % deformable_group != 0)
        return fail(CP_ERR_INVALID, "dcn_v2_forward: bad shape argument (C must be divisible by deformable_group)");
    hipStream_t s = (hipStream_t)stream;
    const bool fast = kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && dh == 1 && dw == 1 &&
                      deformable_group == 1 && C % 16 == 0 && cp_conv_tile_n(Co) >= 64 && !(g_dbg & 8388608);
    if (!fast) {
        // everything CenterPose does not use (other kernels / strides / dilations, deformable groups, tiny channel
        // counts): the generic float32 kernel on the reference's own layouts, no workspace
        const int Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1, Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
        if (Ho < 1 || Wo < 1) return fail(CP_ERR_INVALID, "dcn_v2_forward: empty output");
        const int rc = cp_launch_dcn_generic(s, input, weight, bias, offset, mask, output, B, C, H, W, Co, Ho, Wo, kh, kw, sh,
                                             sw, ph, pw, dh, dw, deformable_group);
        return rc == CP_OK ? CP_OK : fail(rc, "dcn_v2_forward: generic kernel launch failed");
    }
    if (workspace_bytes < cp_dcnv2_workspace_bytes(B, C, H, W, Co)) return fail(CP_ERR_INVALID, "workspace too small");
    const size_t px = (size_t)B * H * W;
    const int cpad = (int)align_up((size_t)Co, cp_conv_tile_n(Co));
    char* w8 = (char*)workspace;
    float* x_nhwc = (float*)w8;
    w8 += align_up(px * C * 4, 256);
    float* om = (float*)w8;
    w8 += align_up(px * 32 * 4, 256);
    float* y_nhwc = (float*)w8;
    w8 += align_up(px * Co * 4, 256);
    float* wp = (float*)w8;
    w8 += align_up((size_t)9 * C * cpad * 4, 256);
    float* shift = (float*)w8;
    int rc = cp_launch_nchw_to_nhwc(input, x_nhwc, B, C, H, W, C, s);
    if (rc != CP_OK) return rc;
    hipLaunchKernelGGL(dcn_offmask_pack_kernel, dim3(2048), dim3(256), 0, s, offset, mask, om, B, H * W);
    if (hipMemsetAsync(wp, 0, (size_t)9 * C * cpad * 4, s) != hipSuccess) return CP_ERR_LAUNCH;
    if (hipMemsetAsync(shift, 0, (size_t)cpad * 4, s) != hipSuccess) return CP_ERR_LAUNCH;
    if (hipMemcpyAsync(shift, bias, (size_t)Co * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return CP_ERR_LAUNCH;
    rc = cp_launch_pack_weight(weight, wp, Co, C, 9, C, cpad, 0, s);
    if (rc != CP_OK) return rc;
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.src[0] = x_nhwc;
    p.src_c[0] = C;
    p.nsrc = 1;
    p.Cin = C;
    p.B = B;
    p.H = p.Ho = H;
    p.W = p.Wo = W;
    p.KH = p.KW = 3;
    p.stride = 1;
    p.pad = 1;
    p.K = p.Kpad = 9 * C;
    p.wp = wp;
    p.Cout = Co;
    p.CoutPad = cpad;
    p.shift = shift;
    p.act = CP_ACT_NONE;
    p.out = y_nhwc;
    p.store = CP_STORE_NHWC;
    p.ldo = Co;
    p.offmask = om;
    p.dbg = g_dbg;
    if (g_default_precision == CP_PREC_F16X3 && C % 32 == 0) {
        char* w16 = (char*)shift + align_up((size_t)cpad * 4, 256);
        const size_t sz = align_up((size_t)9 * C * cpad * 2, 256);
        if (hipMemsetAsync(w16, 0, 2 * sz, s) != hipSuccess) return CP_ERR_LAUNCH;
        p.w16_hi = w16;
        p.w16_lo = w16 + sz;
        p.Kpad16 = 9 * C;
        const size_t csz = align_up((size_t)cpad * 4, 256);
        float* wfwd = (float*)(w16 + 2 * sz);
        float* winv = (float*)((char*)wfwd + csz);
        float* sc16 = (float*)((char*)winv + csz);
        unsigned* slot = (unsigned*)((char*)sc16 + csz);
        if (hipMemsetAsync(wfwd, 0, 3 * csz + (size_t)CP_AMAX_SUB * CP_AMAX_STRIDE * sizeof(unsigned), s) != hipSuccess)
            return CP_ERR_LAUNCH;
        rc = cp_launch_weight_scale(weight, Co, C * 9, wfwd, winv, s);
        if (rc == CP_OK) rc = cp_launch_pack_weight16(weight, (void*)p.w16_hi, (void*)p.w16_lo, Co, C, 9, p.Kpad16, 0, wfwd, s);
        if (rc == CP_OK) rc = cp_launch_scale16(nullptr, winv, sc16, Co, s);
        if (rc == CP_OK) rc = cp_launch_absmax(x_nhwc, px * C, slot, s);
        char* w16f = (char*)slot + (size_t)CP_AMAX_SUB * CP_AMAX_STRIDE * sizeof(unsigned);
        if (rc == CP_OK) rc = cp_launch_frag16_repack(p.w16_hi, w16f, cpad, p.Kpad16, s);
        if (rc == CP_OK) rc = cp_launch_frag16_repack(p.w16_lo, w16f + sz, cpad, p.Kpad16, s);
        p.w16f_hi = w16f;
        p.w16f_lo = w16f + sz;
        if (rc != CP_OK) return rc;
        if (cp_conv16_supported(p)) {
            p.scale = sc16;
            p.in_amax[0] = slot;
        } else {
            p.w16_hi = p.w16_lo = nullptr;
        }
    }
    rc = (p.w16_hi && cp_conv16_supported(p)) ? cp_launch_conv16(p, s) : cp_launch_conv(p, s);
    if (rc != CP_OK) return rc;
    return cp_launch_nhwc_to_nchw(y_nhwc, output, B, Co, H, W, Co, s);
}
