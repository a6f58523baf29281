// Heat-map decode on device: 3x3 max-pool NMS + exact top-K per map, centre-indexed gathers,
// displacement <-> heat-map keypoint association, inference filter and per-point heat-map statistics.
// Restates /root/reference/src/lib/models/decode.py:72-375 (object_pose_decode, Inference=True) and
// utils.py:43-47; replaces ~60 ATen kernels, 5-9 full-map transposes and a B x 8 x K Python loop with
// D2H syncs (decode.py:191-252) by two launches.
//
//   peaks_kernel   one workgroup (1024 lanes = 16 wavefronts) per (image, map): the map is read once (coalesced
//                  16-byte loads, optional in-place sigmoid) into LDS, NMS from LDS, then an exact radix select of
//                  the K largest (value desc, index asc) on register-resident keys -- over the positive keys only
//                  when they fill the top K -- and a rank sort of the winners in LDS.
//   assoc_kernel   one workgroup per (image, joint): candidate table in LDS, one lane per detection
//                  scans the K candidates (float ops in the reference's order, no FMA contraction so
//                  argmin / threshold decisions match the CPU bit for bit).
//
// Output: det[B][K][118] float32 records (field offsets in cp_common.h / centerpose_hip.h).
#include "cp_common.h"

namespace {

constexpr int PK_THREADS = 1024;
constexpr float NEGV = -10000.0f;

__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(b);
}

// block-wide exclusive prefix sum of one int per lane (1024 lanes); returns exclusive prefix, total in *total
__device__ int block_excl_scan(int v, int* sh /*[17]*/, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) sh[w] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < PK_THREADS / 64; ++i) {
            const int t = sh[i];
            sh[i] = acc;
            acc += t;
        }
        sh[16] = acc;
    }
    __syncthreads();
    const int r = sh[w] + x - v;
    *total = sh[16];
    __syncthreads();
    return r;
}

// block-wide sums of two ints per lane (1024 lanes)
__device__ void block_sum2(int a, int b, int* sh /*[34]*/, int* ta, int* tb) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w] = a; sh[16 + w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int x = 0, y = 0;
        for (int i = 0; i < PK_THREADS / 64; ++i) { x += sh[i]; y += sh[16 + i]; }
        sh[32] = x;
        sh[33] = y;
    }
    __syncthreads();
    *ta = sh[32];
    *tb = sh[33];
    __syncthreads();
}

// maps: NCHW.  Block (m, b): m == 0 -> hm[b, 0], m >= 1 -> hm_hp[b, m-1].
//
// A lane owns PK_G groups of four consecutive pixels, group g of lane t = pixels 4 (1024 g + t) .. + 3: the map is read
// once with coalesced 16-byte loads (the first version gave each lane 16 consecutive pixels = a 64-byte lane stride and
// read the 8 NMS neighbours of every pixel from global memory) and staged in LDS, from which the 3x3 maximum of a group
// takes 3 x (one 16-byte + two 4-byte) reads.  After NMS nearly every key is the suppressed value +0: histogramming
// those through LDS atomics meant ~15000 serialised updates of ONE bin per pass, which was most of the kernel's time.
// Keys are therefore classed first -- positive / zero / negative -- and the radix select only runs over the positive
// ones when they already fill the top K (always, for sigmoid heat-maps); the zeros are taken by index when they do not;
// only maps that need negative values walk the general path.
constexpr uint32_t ZKEY = 0x80000000u;  // f2ord(+0.0f)

// PK_G = 4: maps up to 16384 pixels (128 x 128, 64 KB of LDS); PK_G = 8: up to 32768 (e.g. --keep_res 480 x 640 frames ->
// 120 x 160, or --input_res 1024 x 512 -> 256 x 128; 128 KB of the CU's 160 KB LDS)
template <int PK_G>
__global__ __launch_bounds__(PK_THREADS) void peaks_kernel(float* __restrict__ hm, float* __restrict__ hm_hp,
                                                           int J, int H, int W, int K, int apply_sigmoid,
                                                           float* __restrict__ pk_score, int* __restrict__ pk_ind) {
    const int mi = blockIdx.x, b = blockIdx.y, nm = gridDim.x;
    const int HW = H * W, n4 = HW >> 2;
    float* map = (mi == 0) ? hm + (size_t)b * HW : hm_hp + ((size_t)b * J + (mi - 1)) * HW;
    __shared__ __attribute__((aligned(16))) float smap[PK_THREADS * PK_G * 4];  // the whole map
    __shared__ int hist[256];
    __shared__ int scan_sh[34];
    __shared__ int sel[2];  // digit, need
    __shared__ unsigned long long list[128], sorted[128];
    __shared__ int cnt;
    const int tid = threadIdx.x;

    float4 v[PK_G];
#pragma unroll
    for (int g = 0; g < PK_G; ++g) {
        const int i4 = g * PK_THREADS + tid;
        v[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i4 < n4) {
            v[g] = reinterpret_cast<const float4*>(map)[i4];
            if (apply_sigmoid) {
                v[g].x = 1.f / (1.f + expf(-v[g].x));
                v[g].y = 1.f / (1.f + expf(-v[g].y));
                v[g].z = 1.f / (1.f + expf(-v[g].z));
                v[g].w = 1.f / (1.f + expf(-v[g].w));
                reinterpret_cast<float4*>(map)[i4] = v[g];
            }
            reinterpret_cast<float4*>(smap)[i4] = v[g];
        }
    }
    __syncthreads();
    // ---- NMS: key = ordered bits of (v * (hmax == v)) ----
    uint32_t key[PK_G * 4];
    int n_pos = 0, n_zero = 0;
    const float NINF = -__builtin_huge_valf();
#pragma unroll
    for (int g = 0; g < PK_G; ++g) {
        const int i4 = g * PK_THREADS + tid;
        if (i4 < n4) {
            const int p = i4 << 2;
            const int y = p / W, x0 = p - y * W;  // W % 4 == 0: the four pixels share a row
            float m0 = NINF, m1 = NINF, m2 = NINF, m3 = NINF;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                const int rb = yy * W + x0;
                const float4 c = *reinterpret_cast<const float4*>(smap + rb);
                const float l = x0 > 0 ? smap[rb - 1] : NINF, r = x0 + 4 < W ? smap[rb + 4] : NINF;
                m0 = fmaxf(m0, fmaxf(l, fmaxf(c.x, c.y)));
                m1 = fmaxf(m1, fmaxf(c.x, fmaxf(c.y, c.z)));
                m2 = fmaxf(m2, fmaxf(c.y, fmaxf(c.z, c.w)));
                m3 = fmaxf(m3, fmaxf(c.z, fmaxf(c.w, r)));
            }
            const float vv[4] = {v[g].x, v[g].y, v[g].z, v[g].w}, mm[4] = {m0, m1, m2, m3};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float kept = (mm[e] == vv[e]) ? vv[e] : vv[e] * 0.0f;  // heat * keep (decode.py:23)
                uint32_t k = f2ord(kept + 0.0f);                            // -0 -> +0 so equal values tie
                if (k == 0u) k = 1u;
                key[g * 4 + e] = k;
                n_pos += k > ZKEY ? 1 : 0;
                n_zero += k == ZKEY ? 1 : 0;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) key[g * 4 + e] = 0u;  // below every real key
        }
    }
    int tot_pos, tot_zero;
    block_sum2(n_pos, n_zero, scan_sh, &tot_pos, &tot_zero);

    // ---- K-th largest key: radix select over the class that contains it ----
    uint32_t prefix = 0u;
    int need = K;
    if (tot_pos < K && tot_pos + tot_zero >= K) {
        prefix = ZKEY;  // every positive key, then zeros by ascending index
        need = K - tot_pos;
    } else {
        const uint32_t floor_excl = tot_pos >= K ? ZKEY : 0u;  // histogram only keys above this
        uint32_t maskb = 0u;
        for (int pass = 3; pass >= 0; --pass) {
            const int shift = pass * 8;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PK_G * 4; ++i)
                if (key[i] > floor_excl && (key[i] & maskb) == prefix) atomicAdd(&hist[(key[i] >> shift) & 255u], 1);
            __syncthreads();
            if (tid < 64) {
                // lane l owns bins 4l..4l+3; suffix sums over lanes (high digits first)
                const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
                const int s = h0 + h1 + h2 + h3;
                int suf = s;  // inclusive suffix sum: lanes >= tid
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int y = __shfl_down(suf, o, 64);
                    if (tid + o < 64) suf += y;
                }
                const int above = suf - s;  // keys in higher lanes' bins
                if (above < need && need <= suf) {
                    int c = above, d = 4 * tid + 3;
                    const int hh[4] = {h0, h1, h2, h3};
                    for (int q = 3; q >= 0; --q) {
                        if (c + hh[q] >= need) { d = 4 * tid + q; break; }
                        c += hh[q];
                    }
                    sel[0] = d;
                    sel[1] = need - c;
                }
            }
            __syncthreads();
            prefix |= ((uint32_t)sel[0]) << shift;
            maskb |= 0xffu << shift;
            need = sel[1];
            __syncthreads();
        }
    }
    // prefix = K-th largest key; need = how many keys == prefix to take (lowest pixel indices first)
    int my_eq = 0, dummy = 0;
#pragma unroll
    for (int i = 0; i < PK_G * 4; ++i) my_eq += (key[i] == prefix) ? 1 : 0;
    int total_eq;
    block_sum2(my_eq, 0, scan_sh, &total_eq, &dummy);
    if (tid == 0) cnt = 0;
    if (tid < 128) list[tid] = sorted[tid] = 0ull;
    __syncthreads();
    const bool all_eq = total_eq <= need;  // no tie at the threshold: every equal key is taken, no ranking needed
    int taken_before = 0;                  // equal keys in lower-indexed groups (pixel order: group, lane, element)
#pragma unroll
    for (int g = 0; g < PK_G; ++g) {
        int rank = 0;
        if (!all_eq) {
            int c = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) c += key[g * 4 + e] == prefix ? 1 : 0;
            int tot_g;
            rank = taken_before + block_excl_scan(c, scan_sh, &tot_g);
            taken_before += tot_g;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t k = key[g * 4 + e];
            bool take = k > prefix;
            if (k == prefix) {
                take = all_eq || rank < need;
                ++rank;
            }
            if (take && k != 0u) {
                const int slot = atomicAdd(&cnt, 1);
                const uint32_t p = (uint32_t)(((g * PK_THREADS + tid) << 2) + e);
                if (slot < 128) list[slot] = ((unsigned long long)k << 32) | (unsigned long long)(0xffffffffu - p);
            }
        }
    }
    __syncthreads();
    // ---- order the (at most 128, pairwise different) winners, largest first: every key counts the keys above it and goes to
    //      that position.  128 broadcast LDS reads per lane and ONE barrier where the bitonic network of rounds 1-4 took 28
    //      barriers of a 16-wave workgroup (a fifth of the kernel at batch 1).  Empty slots (0) all land behind the winners. ----
    if (tid < 128) {
        const unsigned long long mine = list[tid];
        int rank = 0;
#pragma unroll 16
        for (int j = 0; j < 128; ++j) rank += list[j] > mine ? 1 : 0;
        sorted[mine ? rank : 127] = mine;  // (K <= 128: when 128 real keys exist there is no empty slot to collide with rank 127)
    }
    __syncthreads();
    if (tid < K) {
        const unsigned long long e = sorted[tid];
        const size_t o = ((size_t)b * nm + mi) * K + tid;
        pk_score[o] = ord2f((uint32_t)(e >> 32));
        pk_ind[o] = (int)(0xffffffffu - (uint32_t)(e & 0xffffffffull));
    }
}

struct AssocParams {
    const float *hps, *wh, *hps_unc, *scale, *scale_unc, *reg, *hm_hp, *hp_offset, *tracking, *tracking_hp;
    const float* pk_score;
    const int* pk_ind;
    float* det;
    int B, J, H, W, K, rep_mode, fit_gaussian, legacy_bool_mask;
    float balance;
};

__device__ __forceinline__ float sel_mix(float m, float a, float b) {
    // (1 - m) * a + m * b evaluated like the reference's float tensors (m is exactly 0 or 1)
    return __fadd_rn(__fmul_rn(__fsub_rn(1.f, m), a), __fmul_rn(m, b));
}

// python-style int(): truncate toward zero
__device__ __forceinline__ int pyint(float v) { return (int)v; }

// gpfit.moments (gpfit.py:13-26) on a zero-padded window + scipy's make_strictly_feasible nudge
// (least_squares(..., max_nfev=1) returns its strictly feasible start; gpfit.py:38).
__device__ void window_moments(const float* data, int H, int W, float xf, float yf, double out[5]) {
    const int ran = 5, HP = H + 2 * ran, WP = W + 2 * ran;
    // python slice [int(y) : int(y + 11)] on the padded array (negative indices wrap, then clip)
    int r0 = pyint(yf), r1 = pyint(__fadd_rn(yf, 11.f)), c0 = pyint(xf), c1 = pyint(__fadd_rn(xf, 11.f));
    if (r0 < 0) r0 = max(0, r0 + HP);
    if (r1 < 0) r1 = max(0, r1 + HP);
    if (c0 < 0) c0 = max(0, c0 + WP);
    if (c1 < 0) c1 = max(0, c1 + WP);
    r0 = min(r0, HP); r1 = min(r1, HP); c0 = min(c0, WP); c1 = min(c1, WP);
    const int nr = max(0, r1 - r0), nc = max(0, c1 - c0);
    auto at = [&](int r, int c) -> double {
        const int y = r0 + r - ran, x = c0 + c - ran;
        return (y >= 0 && y < H && x >= 0 && x < W) ? (double)data[y * W + x] : 0.0;
    };
    double total = 0, sx = 0, sy = 0, hmax = -1e300;
    for (int r = 0; r < nr; ++r)
        for (int c = 0; c < nc; ++c) {
            const double v = at(r, c);
            total += v;
            sx += r * v;
            sy += c * v;
            hmax = fmax(hmax, v);
        }
    const double x = sx / total, y = sy / total;  // x: row centroid, y: col centroid (gpfit.py:18-19)
    double num = 0, den = 0;
    const int cy = (int)y;
    for (int r = 0; r < nr; ++r) {  // col = data[:, int(y)]; width_x uses (arange - y)
        const double v = (cy >= 0 && cy < nc) ? at(r, cy) : 0.0;
        num += ((double)r - y) * ((double)r - y) * v;
        den += v;
    }
    const double wx = sqrt(fabs(num) / den);
    num = 0; den = 0;
    const int cx = (int)x;
    for (int c = 0; c < nc; ++c) {  // row = data[int(x), :]; width_y uses (arange - x)
        const double v = (cx >= 0 && cx < nr) ? at(cx, c) : 0.0;
        num += ((double)c - x) * ((double)c - x) * v;
        den += v;
    }
    const double wy = sqrt(fabs(num) / den);
    double p[5] = {hmax, x, y, wx, wy};
    const double ub[5] = {1e300, (double)nr, (double)nc, 1e300, 1e300};
    for (int i = 0; i < 5; ++i) {  // scipy.optimize._lsq.common.make_strictly_feasible, rstep = 1e-10
        const double lo = p[i], up = ub[i] - p[i];
        const bool fin_ub = (i == 1 || i == 2);
        const double thr_u = 1e-10 * fmax(1.0, fabs(ub[i]));
        const bool lower = lo <= fmin(fin_ub ? up : 1e300, 1e-10);
        const bool upper = fin_ub && (up <= fmin(lo, thr_u));
        if (lower) p[i] = 1e-10;
        else if (upper) p[i] = ub[i] - thr_u;
        if (fin_ub && (p[i] < 0 || p[i] > ub[i])) p[i] = 0.5 * ub[i];
        out[i] = p[i];
    }
}

__global__ __launch_bounds__(128) void assoc_kernel(const AssocParams p) {
    const int j = blockIdx.x, b = blockIdx.y, k = threadIdx.x;
    const int K = p.K, J = p.J, H = p.H, W = p.W, HW = H * W, NM = J + 1;
    extern __shared__ float sh[];
    float* cx = sh;
    float* cy = sh + K;
    float* cs = sh + 2 * K;
    const float thresh = 0.1f;

    // ---- candidate table for joint j (decode.py:128-144) ----
    if (k < K) {
        const size_t o = ((size_t)b * NM + (j + 1)) * K + k;
        const float s = p.pk_score[o];
        const int ind = p.pk_ind[o];
        float x = (float)(ind % W), y = (float)(ind / W);
        if (p.hp_offset) {
            x = __fadd_rn(x, p.hp_offset[((size_t)b * 2 + 0) * HW + ind]);
            y = __fadd_rn(y, p.hp_offset[((size_t)b * 2 + 1) * HW + ind]);
        } else {
            x = __fadd_rn(x, 0.5f);
            y = __fadd_rn(y, 0.5f);
        }
        const float m = (s > thresh) ? 1.f : 0.f;
        cs[k] = sel_mix(m, -1.f, s);
        cy[k] = sel_mix(m, NEGV, y);
        cx[k] = sel_mix(m, NEGV, x);
    }
    __syncthreads();
    if (k >= K) return;

    // ---- centre-indexed gathers (decode.py:86-109) ----
    const size_t oc = ((size_t)b * NM) * K + k;
    const float score = p.pk_score[oc];
    const int ind = p.pk_ind[oc];
    const float xs_i = (float)(ind % W), ys_i = (float)(ind / W);
    float xs = xs_i, ys = ys_i;
    if (p.reg) {
        xs = __fadd_rn(xs_i, p.reg[((size_t)b * 2 + 0) * HW + ind]);
        ys = __fadd_rn(ys_i, p.reg[((size_t)b * 2 + 1) * HW + ind]);
    } else {
        xs = __fadd_rn(xs_i, 0.5f);
        ys = __fadd_rn(ys_i, 0.5f);
    }
    const float w2 = p.wh[((size_t)b * 2 + 0) * HW + ind] / 2.f, h2 = p.wh[((size_t)b * 2 + 1) * HW + ind] / 2.f;
    const float l = __fsub_rn(xs, w2), t = __fsub_rn(ys, h2), r = __fadd_rn(xs, w2), bt = __fadd_rn(ys, h2);
    const float kx = __fadd_rn(p.hps[((size_t)b * 2 * J + 2 * j) * HW + ind], xs_i);
    const float ky = __fadd_rn(p.hps[((size_t)b * 2 * J + 2 * j + 1) * HW + ind], ys_i);

    // ---- nearest heat-map peak (decode.py:147-156) ----
    float best = 0.f;
    int bi = 0;
    for (int c = 0; c < K; ++c) {
        const float dx = __fsub_rn(kx, cx[c]), dy = __fsub_rn(ky, cy[c]);
        const float d = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
        if (c == 0 || d < best) {
            best = d;
            bi = c;
        }
    }
    const float hs = cs[bi], hx = cx[bi], hy = cy[bi];
    const float size = fmaxf(__fsub_rn(bt, t), __fsub_rn(r, l));
    const bool rej = (hx < l) || (hx > r) || (hy < t) || (hy > bt) || (hs < thresh) || (best > __fmul_rn(size, 0.3f));
    const float m = rej ? 1.f : 0.f;
    float ox, oy;
    if (p.rep_mode == 3) { ox = kx; oy = ky; }
    else if (p.rep_mode == 4) { ox = hx; oy = hy; }
    else { ox = sel_mix(m, hx, kx); oy = sel_mix(m, hy, ky); }

    // ---- inference filter (decode.py:178-189); torch<=1.1 semantics unless legacy_bool_mask ----
    bool m2 = (hx > __fmul_rn(0.8f, l)) && (hx < __fmul_rn(1.2f, r)) && (hy > __fmul_rn(0.8f, t)) &&
              (hy < __fmul_rn(1.2f, bt)) && (hs > thresh) && (best < __fmul_rn(size, 0.5f)) && (score > thresh);
    if (p.legacy_bool_mask) m2 = false;
    const float m2f = m2 ? 1.f : 0.f;
    const float fx = __fadd_rn(__fmul_rn(m2f, hx), __fmul_rn(__fsub_rn(1.f, m2f), NEGV));
    const float fy = __fadd_rn(__fmul_rn(m2f, hy), __fmul_rn(__fsub_rn(1.f, m2f), NEGV));

    // ---- per-point heat-map statistics (decode.py:195-252) ----
    float mean_x = NEGV, mean_y = NEGV, std_x = NEGV, std_y = NEGV, height = NEGV;
    if ((p.rep_mode == 1 || p.rep_mode == 2) && !(fx == NEGV || fy == NEGV)) {
        const float* data = p.hm_hp + ((size_t)b * J + j) * HW;  // un-suppressed copy (decode.py:114)
        if (p.fit_gaussian) {
            double q[5];
            window_moments(data, H, W, fx, fy, q);
            // decode.py:236,248-249: (height, mu_x, mu_y, std_x, std_y) = params, i.e. the ROW centroid is added to x
            mean_x = (float)((double)fx + q[1] - 5.0);
            mean_y = (float)((double)fy + q[2] - 5.0);
            std_x = (float)q[3];
            std_y = (float)q[4];
            height = (float)q[0];
        } else {
            int iy = pyint(fy), ix = pyint(fx);
            if (iy < 0) iy += H;  // python negative indexing
            if (ix < 0) ix += W;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                // numpy float32 scalar arithmetic: (x + 5) - 5 in float32
                mean_x = __fsub_rn(__fadd_rn(fx, 5.f), 5.f);
                mean_y = __fsub_rn(__fadd_rn(fy, 5.f), 5.f);
                height = data[iy * W + ix];
                std_x = 1.f;
                std_y = 1.f;
            }  // else: the reference raises IndexError; the record keeps -10000
        }
    }

    float* d = p.det + ((size_t)b * K + k) * CP_DET_STRIDE;
    d[CP_DET_KPS + 2 * j] = ox;
    d[CP_DET_KPS + 2 * j + 1] = oy;
    d[CP_DET_KPS_DISP_MEAN + 2 * j] = kx;
    d[CP_DET_KPS_DISP_MEAN + 2 * j + 1] = ky;
    d[CP_DET_KPS_HM_MEAN + 2 * j] = mean_x;
    d[CP_DET_KPS_HM_MEAN + 2 * j + 1] = mean_y;
    d[CP_DET_KPS_HM_STD + 2 * j] = std_x;
    d[CP_DET_KPS_HM_STD + 2 * j + 1] = std_y;
    d[CP_DET_KPS_HM_HEIGHT + j] = height;
    // per-joint slices of the centre-indexed optional heads
    for (int q = 0; q < 2; ++q) {
        const int c = 2 * j + q;
        float v = 0.f;
        if (p.hps_unc) v = __fmul_rn(sqrtf(expf(p.hps_unc[((size_t)b * 2 * J + c) * HW + ind])), p.balance);
        d[CP_DET_KPS_DISP_STD + c] = v;
        d[CP_DET_TRACKING_HP + c] = p.tracking_hp ? p.tracking_hp[((size_t)b * 2 * J + c) * HW + ind] : 0.f;
    }
    if (j == 0) {
        d[CP_DET_BBOX + 0] = l;
        d[CP_DET_BBOX + 1] = t;
        d[CP_DET_BBOX + 2] = r;
        d[CP_DET_BBOX + 3] = bt;
        d[CP_DET_SCORE] = score;
        d[CP_DET_CLS] = 0.f;  // single category (opts.py:435): clses = topk_ind / K = 0
        for (int c = 0; c < 3; ++c) {
            d[CP_DET_SCALE + c] = p.scale ? p.scale[((size_t)b * 3 + c) * HW + ind] : 0.f;
            d[CP_DET_SCALE_UNC + c] = p.scale_unc ? sqrtf(expf(p.scale_unc[((size_t)b * 3 + c) * HW + ind])) : 0.f;
        }
        for (int c = 0; c < 2; ++c) d[CP_DET_TRACKING + c] = p.tracking ? p.tracking[((size_t)b * 2 + c) * HW + ind] : 0.f;
    }
}

}  // namespace

size_t cp_decode_ws_bytes(int B, int J, int K) { return (size_t)B * (J + 1) * K * 8 + 256; }

int cp_launch_decode(hipStream_t s, int B, int J, int H, int W, float* hm, const float* hps, const float* wh,
                     const float* hps_unc, const float* scale, const float* scale_unc, const float* reg, float* hm_hp,
                     const float* hp_offset, const float* tracking, const float* tracking_hp, int K, int rep_mode,
                     int fit_gaussian, float balance, int legacy_bool_mask, int apply_sigmoid, float* det, void* ws) {
    if (H * W > PK_THREADS * 32 || H * W < K || K < 1 || K > 128 || J < 1 || W % 4 != 0) return CP_ERR_INVALID;
    float* pk_score = (float*)ws;
    int* pk_ind = (int*)((char*)ws + (size_t)B * (J + 1) * K * 4);
    if (H * W <= PK_THREADS * 16)
        hipLaunchKernelGGL(peaks_kernel<4>, dim3(J + 1, B), dim3(PK_THREADS), 0, s, hm, hm_hp, J, H, W, K, apply_sigmoid,
                           pk_score, pk_ind);
    else
        hipLaunchKernelGGL(peaks_kernel<8>, dim3(J + 1, B), dim3(PK_THREADS), 0, s, hm, hm_hp, J, H, W, K, apply_sigmoid,
                           pk_score, pk_ind);
    AssocParams p;
    p.hps = hps; p.wh = wh; p.hps_unc = hps_unc; p.scale = scale; p.scale_unc = scale_unc; p.reg = reg;
    p.hm_hp = hm_hp; p.hp_offset = hp_offset; p.tracking = tracking; p.tracking_hp = tracking_hp;
    p.pk_score = pk_score; p.pk_ind = pk_ind; p.det = det;
    p.B = B; p.J = J; p.H = H; p.W = W; p.K = K; p.rep_mode = rep_mode; p.fit_gaussian = fit_gaussian;
    p.legacy_bool_mask = legacy_bool_mask; p.balance = balance;
    hipLaunchKernelGGL(assoc_kernel, dim3(J, B), dim3(128), 3 * K * sizeof(float), s, p);
    return hipGetLastError() == hipSuccess ? CP_OK : CP_ERR_LAUNCH;
}
