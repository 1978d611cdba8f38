// k_feature.hip -- K7 quad pairing (featureRecovery / featureOrganization / cornerObtain),
// K8 sub-pixel edge refinement (edgeRefine), K9 marker grouping, cross-ratio IDs and dictionary decode
// (markerOrganization / featureExtraction / markerDecoder / match_dictionary).
//   /root/reference/corner_detector.cpp:465-598, :600-951, :976-1324; call order CylinderTag.cpp:92-128.
// Same arithmetic, types and evaluation order as the reference so the discrete decisions
// match the oracle bit for bit; the parallel structure is this implementation's own:
//   K7  one workgroup per frame; for each unvisited quad i the match test runs on all j > i at once and
//       the smallest matching j wins (== the reference's first match in its sequential j loop).
//   K8  one workgroup per (feature, quad); one lane per edge sample, the 41-step normal search is the
//       lane's inner loop; the weighted moment sums are then accumulated in sample order by 12 lanes
//       (6 sums x {next,last} weighting) so the doubles equal the reference's sequential sums.
//   K9  one wave per frame; dictionary matching evaluates all (row, column, direction) hypotheses on lanes
//       and replays the reference's order-dependent max / second-max bookkeeping afterwards.
#include "ctag_internal.h"
#include "ctag_math.h"
#include "ctag_refine.h"

namespace ctag {

constexpr double kPi = 3.1415926535897932384626433832795;

struct P2 {
    float x, y;
};
__device__ __forceinline__ float dist2p(P2 a, P2 b) { return ctm::sqrt32((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y)); }
// atan2(float,float)*180/CV_PI as the reference writes it: float atan2, float*int, then a double division
__device__ __forceinline__ double angdeg(float dy, float dx) { return ctm::atan2_32(dy, dx) * 180 / kPi; }
__device__ __forceinline__ bool solve2x2f(float a00, float a01, float a10, float a11, float b0, float b1, float& x0, float& x1) {
    double d = (double)a00 * a11 - (double)a01 * a10;
    if (d == 0.) return false;
    d = 1. / d;
    const float t = (float)(((double)b0 * a11 - (double)b1 * a01) * d);
    x1 = (float)(((double)b1 * a00 - (double)b0 * a10) * d);
    x0 = t;
    return true;
}

// =====================================================================================================
// K7
// =====================================================================================================
struct QuadDerived {  // per accepted quad: centre, side lengths, the two mean side directions (:473-481)
    float cx, cy, d[4], a1, a2;
    float e[4];  // the four edge directions the pair test may ask for (:497-539), computed once per quad instead of per pair
};
static_assert(sizeof(QuadDerived) == 48, "workspace carve in ctag_api.hip");
struct FeatPtrs {
    const int32_t* ncand;
    const QuadOut* quads;
    QuadDerived* derived;  // [F][kQuadStride]
    int32_t* quad_index;   // [F][kQuadStride] accepted-quad -> candidate index (the first kQuadStride of them: more than CTAG_MAX_QUADS ends the frame)
    int32_t* nquads;
    int32_t* nfeat;
    int32_t* status;
    uint32_t* frame_flags;
    FeatureDev* feat0;
    FeatureDev* feat1;
    FeatureDev* feat2;
    unsigned long long* stamps;  // developer aid (CTAG_FEAT_STAMPS=1): clock ticks per phase of k_features, else null
    float threshold_angle;       // include/ctag.h ctag_params [5]
    int32_t* frame_long;         // [F] reset here for K8 (k_edge_refine<1> sets it)
    int cand_cap;                // candidates per frame the workspace holds (stride of `quads`)
};
static_assert(kQuadStride > CTAG_MAX_QUADS, "K7 keeps every quad of a frame it goes on with");

// developer aid: phase clock of a block (thread 0), summed over blocks into `stamps[base + phase]`
struct PhaseClock {
    unsigned long long* stamps;
    unsigned long long t_prev;
    __device__ explicit PhaseClock(unsigned long long* s) : stamps(s), t_prev(s ? __builtin_amdgcn_s_memtime() : 0ull) {}
    __device__ void mark(int slot) {
        if (stamps && threadIdx.x == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            atomicAdd(&stamps[slot], t - t_prev);
            t_prev = t;
        }
    }
};

__device__ __forceinline__ bool near_ang(float a, float b, float thr) {
    return ctm::fabs32(a - b) < thr || ctm::fabs32(ctm::fabs32(a - b) - 180) < thr || ctm::fabs32(ctm::fabs32(a - b) - 360) < thr;
}

// featureOrganization (:571-598), eight lanes per feature: lane k of the group evaluates ONE of the eight corner angles (an atan2 each, what the
// function costs), the rest -- a handful of float operations -- every lane repeats and lane 0 stores
__device__ __forceinline__ void feature_organization(const float* q1, const float* q2, float c1x, float c1y, float c2x, float c2y, float feature_angle, FeatureDev* F, int k, int lane0, bool store) {
    const int kk = k & 3;
    const float* q = k < 4 ? q1 : q2;
    const float ang = (float)angdeg((k < 4 ? c1y : c2y) - q[2 * kk + 1], (k < 4 ? c1x : c2x) - q[2 * kk]);
    float a1[4], a2[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a1[i] = __shfl(ang, lane0 + i);
        a2[i] = __shfl(ang, lane0 + 4 + i);
    }
    float angle_max = 0, angle_min = 360;
    int pos1 = -1, pos2 = -1;
    auto fold = [&](float a) { return fminf(360 - ctm::fabs32(a - feature_angle), ctm::fabs32(a - feature_angle)); };
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float v1 = fold(a1[(i + 2) % 4]) + fold(a1[(i + 3) % 4]);
        if (v1 < angle_min) {
            angle_min = v1;
            pos1 = i;
        }
        const float v2 = fold(a2[(i + 2) % 4]) + fold(a2[(i + 3) % 4]);
        if (v2 > angle_max) {
            angle_max = v2;
            pos2 = i;
        }
    }
    if (!store) return;
    if (pos1 < 0) pos1 = 0;
    if (pos2 < 0) pos2 = 0;
    for (int i = 0; i < 4; i++) {
        F->c[2 * i] = q1[2 * ((i + pos1) % 4)];
        F->c[2 * i + 1] = q1[2 * ((i + pos1) % 4) + 1];
        F->c[8 + 2 * i] = q2[2 * ((i + pos2) % 4)];
        F->c[8 + 2 * i + 1] = q2[2 * ((i + pos2) % 4) + 1];
    }
    F->center[0] = (F->c[0] + F->c[2] + F->c[8] + F->c[10]) / 4;
    F->center[1] = (F->c[1] + F->c[3] + F->c[9] + F->c[11]) / 4;
    F->angle = feature_angle;
    F->pad = 0;
}

// the pair test of featureRecovery (:483-548) for quads i < j; pure in (Di, Dj)
__device__ __forceinline__ bool feature_pair(const QuadDerived& Di, const QuadDerived& Dj, float thr /* threshold_angle [5] */) {
    bool tag1 = false, tag2 = false;
    float d1s = 0, d1l = 0, d2s = 0, d2l = 0, ea1 = 0, ea2 = 0;
    const float fa = (float)angdeg(Di.cy - Dj.cy, Di.cx - Dj.cx);
    if (near_ang(fa, Di.a1, thr)) {
        tag1 = true;
        d1l = (Di.d[0] + Di.d[2]) / 2;
        d1s = fminf(Di.d[1], Di.d[3]);
        ea1 = Di.d[1] < Di.d[3] ? Di.e[0] : Di.e[1];
    }
    if (near_ang(fa, Di.a2, thr)) {
        tag1 = true;
        d1s = fminf(Di.d[0], Di.d[2]);
        d1l = (Di.d[1] + Di.d[3]) / 2;
        ea1 = Di.d[0] > Di.d[2] ? Di.e[2] : Di.e[3];
    }
    if (near_ang(fa, Dj.a1, thr)) {
        tag2 = true;
        d2l = (Dj.d[0] + Dj.d[2]) / 2;
        d2s = fminf(Dj.d[1], Dj.d[3]);
        ea2 = Dj.d[1] < Dj.d[3] ? Dj.e[0] : Dj.e[1];
    }
    if (near_ang(fa, Dj.a2, thr)) {
        tag2 = true;
        d2s = fminf(Dj.d[0], Dj.d[2]);
        d2l = (Dj.d[1] + Dj.d[3]) / 2;
        ea2 = Dj.d[0] > Dj.d[2] ? Dj.e[2] : Dj.e[3];
    }
    const float fl = dist2p(P2{Di.cx, Di.cy}, P2{Dj.cx, Dj.cy});
    return (tag1 && tag2) && (d1l > d1s || d2l > d2s) && near_ang(ea1, ea2, thr * 10) && (ctm::fabs32(d1s - d2s) < fminf(d1s, d2s) * 0.33) &&
           ((d1l + d2l) > (d1s + d2s)) && ((d1l + d2l) < 15 * (d1s + d2s)) && (fl - (d1l + d2l) / 2 < 0.3 * (fl + (d1l + d2l) / 2));
}

// THREADS: 128 for batches (a frame per block, many blocks per CU); 512 for calls of a few frames, where the kernel is as long as one block's
// chain of phases and every phase is a loop over work items (quads, pairs, features x 8 angles) that more lanes finish in fewer trips
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_features(FeatPtrs P, int nframes, int feature_size) {
    constexpr int NW = THREADS / 64;
    __shared__ unsigned char s_vis[CTAG_MAX_QUADS];
    __shared__ int s_scan[NW];
    __shared__ int s_best;
    __shared__ int s_nf;
    constexpr int kPairCap = 192;  // quads per frame handled by the all-pairs path (a frame has 50-80)
    __shared__ QuadDerived s_der[kPairCap];
    __shared__ uint32_t s_pred[kPairCap * (kPairCap / 32)];
    __shared__ uint32_t s_match[CTAG_MAX_FEATURES];
    constexpr int kPairList = 2048;  // close pairs listed for the dense pass (a frame has a few hundred); more are evaluated in place
    static_assert(kPairCap <= 256, "a listed pair is two bytes");
    __shared__ uint16_t s_plist[kPairList];
    __shared__ float s_reach[kPairCap];
    __shared__ float s_qc[kPairCap][8];  // the quads' corners: featureOrganization reads them again (two dependent global loads per feature otherwise)
    __shared__ int s_npair;
    const int frame = blockIdx.x;
    if (frame >= nframes) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nc = P.ncand[frame];
    const QuadOut* quads = P.quads + (size_t)frame * P.cand_cap;
    QuadDerived* der = P.derived + (size_t)frame * kQuadStride;
    int32_t* qidx = P.quad_index + (size_t)frame * kQuadStride;
    FeatureDev* f0 = P.feat0 + (size_t)frame * CTAG_MAX_FEATURES;
    FeatureDev* f1 = P.feat1 + (size_t)frame * CTAG_MAX_FEATURES;
    FeatureDev* f2 = P.feat2 + (size_t)frame * CTAG_MAX_FEATURES;

    PhaseClock clk(P.stamps);
    // compact accepted quads in candidate (= OpenCV label) order
    int Q = 0;
    for (int base = 0; base < nc; base += THREADS) {
        const int i = base + tid;
        const int v = (i < nc && quads[i].valid) ? 1 : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        int pre = inc - v, all = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const int c = s_scan[w];
            if (w < wave) pre += c;
            all += c;
        }
        if (v && Q + pre < kQuadStride) qidx[Q + pre] = i;
        Q += all;
        __syncthreads();
    }
    if (tid == 0) {
        P.nquads[frame] = Q;
        s_nf = 0;
    }
    if (Q == 0) {
        if (tid == 0) {
            P.nfeat[frame] = 0;
            P.status[frame] = (P.frame_flags[frame] & CTAG_FLAG_POOL_OVERFLOW) ? CTAG_ERR_LIMIT : CTAG_NO_CORNER;
        }
        return;
    }
    if (Q > CTAG_MAX_QUADS) {
        if (tid == 0) {
            atomicOr(&P.frame_flags[frame], CTAG_FLAG_QUAD_OVERFLOW);
            P.nfeat[frame] = 0;
            P.status[frame] = CTAG_ERR_LIMIT;
        }
        return;
    }
    __syncthreads();  // qidx visible to the block (global memory, same workgroup)
    clk.mark(0);
    // per quad: centre, side lengths and the five side directions the pair test asks for (:473-481, :497-539) -- eight lanes per quad, lane k < 5
    // of the group evaluates ONE atan2 (what the phase costs); lane 0 gathers them
    for (int q0 = 0; q0 < Q; q0 += THREADS / 8) {  // uniform trip count: the shuffles below need the whole group
        const int q = q0 + (tid >> 3), k = tid & 7, l0 = lane & ~7;
        const bool have = q < Q;
        const float* c = quads[qidx[have ? q : 0]].c;
        // A = (c1-c3, c0-c2)  B = (c7-c5, c6-c4)  C = (c3-c5, c2-c4)  D = (c1-c7, c0-c6)  E = (c5-c7, c4-c6)
        const int ya = k == 0 ? 1 : k == 1 ? 7 : k == 2 ? 3 : k == 3 ? 1 : 5, yb = k == 0 ? 3 : k == 1 ? 5 : k == 2 ? 5 : k == 3 ? 7 : 7;
        const double ang = k < 5 ? angdeg(c[ya] - c[yb], c[ya - 1] - c[yb - 1]) : 0.;
        double A[5];
#pragma unroll
        for (int u = 0; u < 5; u++) A[u] = __shfl(ang, l0 + u);
        if (have && k == 0) {
            QuadDerived D;
            D.cx = (c[0] + c[2] + c[4] + c[6]) / 4;
            D.cy = (c[1] + c[3] + c[5] + c[7]) / 4;
            for (int j = 0; j < 4; j++) {
                const int kk = (j + 1) % 4;
                D.d[j] = ctm::sqrt32((c[2 * j] - c[2 * kk]) * (c[2 * j] - c[2 * kk]) + (c[2 * j + 1] - c[2 * kk + 1]) * (c[2 * j + 1] - c[2 * kk + 1]));
            }
            D.a1 = (float)((A[0] + A[1]) / 2);
            D.a2 = (float)((A[2] + A[3]) / 2);
            D.e[0] = (float)A[3];
            D.e[1] = (float)A[2];
            D.e[2] = (float)A[0];
            D.e[3] = (float)A[4];
            der[q] = D;
            s_vis[q] = 0;
            if (q < kPairCap) {
                s_der[q] = D;
                s_reach[q] = fmaxf((D.d[0] + D.d[2]) / 2, (D.d[1] + D.d[3]) / 2);
#pragma unroll
                for (int u = 0; u < 8; u++) s_qc[q][u] = c[u];
            }
        }
    }
    for (int w = tid; w < min(Q, kPairCap) * (kPairCap / 32); w += THREADS) s_pred[w] = 0u;
    if (tid == 0) s_npair = 0;
    __syncthreads();
    clk.mark(1);
    // ---- greedy pairing (:483-553): quad i takes the first unvisited j > i that passes the pair test.  The test does not
    // depend on the visited flags, so for Q <= kPairCap it is evaluated for all pairs at once (every lane busy, one atan2 per
    // pair) into a bit matrix, one lane replays the greedy order on the bits, and the matches are organised in parallel.
    if (Q <= kPairCap) {
        // The pair test costs an atan2 in double and most pairs are far apart, so a cheap necessary condition goes first: the
        // test ends with  fl - L/2 < 0.3 (fl + L/2)  (fl = centre distance, L = d1l + d2l), and whichever branches are taken
        // d1l <= reach_i = max((d0+d2)/2, (d1+d3)/2) -- the same float expressions -- hence L <= reach_i + reach_j =: S
        // (rounding is monotonic).  With fl >= S the left side is >= fl/2 and the right side <= 0.45 fl: the test fails by a
        // margin no rounding closes.  Pairs that survive are listed and evaluated densely, a pair per lane.
        {
            // every wave holds (cx, cy, reach) of all quads in registers -- quad j in lane j & 63, register j >> 6 -- and takes every NW-th row i:
            // the row's three values come out of those registers with v_readlane (i is wave-uniform), not out of LDS with a round trip per row
            static_assert(kPairCap <= 192, "three quads per lane");
            float qx[3], qy[3], qr[3];
#pragma unroll
            for (int b = 0; b < 3; b++) {
                const int j = lane + 64 * b;
                const bool hj = j < Q;
                qx[b] = hj ? s_der[j].cx : 0.f;
                qy[b] = hj ? s_der[j].cy : 0.f;
                qr[b] = hj ? s_reach[j] : 0.f;
            }
            auto consider = [&](int i, int j) {
                const int at = atomicAdd(&s_npair, 1);
                if (at < kPairList) s_plist[at] = (uint16_t)(i | (j << 8));
                else if (feature_pair(s_der[i], s_der[j], P.threshold_angle)) atomicOr(&s_pred[i * (kPairCap / 32) + (j >> 5)], 1u << (j & 31));  // list full: in place
            };
            auto bcast = [&](const float (&v)[3], int i) {  // v of quad i (uniform)
                const int l = i & 63;
                const int b = i >> 6;
                const int r0 = __builtin_amdgcn_readlane(__float_as_int(v[0]), l), r1 = __builtin_amdgcn_readlane(__float_as_int(v[1]), l),
                          r2 = __builtin_amdgcn_readlane(__float_as_int(v[2]), l);
                return __int_as_float(b == 0 ? r0 : (b == 1 ? r1 : r2));
            };
            for (int i = wave; i + 1 < Q; i += NW) {
                const float xi = bcast(qx, i), yi = bcast(qy, i), ri = bcast(qr, i);
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    const int j = lane + 64 * b;
                    if (j < Q && j > i) {
                        const float S = ri + qr[b], dx = xi - qx[b], dy = yi - qy[b];
                        if (!(dx * dx + dy * dy > S * S)) consider(i, j);
                    }
                }
            }
        }
        __syncthreads();
        for (int t = tid; t < min(s_npair, kPairList); t += THREADS) {
            const int i = (int)(s_plist[t] & 0xffu), j = (int)(s_plist[t] >> 8);
            if (feature_pair(s_der[i], s_der[j], P.threshold_angle)) atomicOr(&s_pred[i * (kPairCap / 32) + (j >> 5)], 1u << (j & 31));
        }
        __syncthreads();
        clk.mark(2);
        if (wave == 0) {
            // the greedy replay (first unvisited j of every unvisited i, ascending i) on wave 0: row q of the bit matrix sits in lane q & 63 (three rows
            // per lane, three 64-bit words per row), the visited set and the rows still to look at are wave-uniform 64-bit words, so a step is a
            // v_readlane of the row's words and a handful of scalar bit operations -- no LDS round trip, no vector compare, no ballot
            constexpr int kW = kPairCap / 32;
            static_assert(kPairCap == 192, "three 64-bit words per row, three rows per lane");
            unsigned long long row[3][3];  // [row block a: row = lane + 64 a][word b: columns 64 b .. 64 b + 63]
#pragma unroll
            for (int a3 = 0; a3 < 3; a3++) {
                const int q = a3 * 64 + lane;
#pragma unroll
                for (int b3 = 0; b3 < 3; b3++)
                    row[a3][b3] = q < Q ? ((unsigned long long)s_pred[q * kW + 2 * b3] | ((unsigned long long)s_pred[q * kW + 2 * b3 + 1] << 32)) : 0ull;
            }
            unsigned long long todo[3], vis[3] = {0ull, 0ull, 0ull};
#pragma unroll
            for (int a3 = 0; a3 < 3; a3++) todo[a3] = __ballot((row[a3][0] | row[a3][1] | row[a3][2]) != 0ull);  // rows with a candidate at all
            auto rd = [&](unsigned long long v, int l) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
                return (unsigned long long)lo | ((unsigned long long)hi << 32);
            };
            int nm = 0;
#pragma unroll
            for (int a3 = 0; a3 < 3; a3++) {
                unsigned long long t = todo[a3];
                while (t) {  // uniform
                    const int l = __builtin_ctzll(t);
                    t &= t - 1;
                    if ((vis[a3] >> l) & 1ull) continue;
                    // a row holds bits j > i only: the words below the row's own are empty
                    unsigned long long m[3];
#pragma unroll
                    for (int b3 = 0; b3 < 3; b3++) m[b3] = b3 >= a3 ? (rd(row[a3][b3], l) & ~vis[b3]) : 0ull;
                    int jb = -1;
#pragma unroll
                    for (int b3 = 2; b3 >= 0; b3--)
                        if (m[b3]) jb = b3;
                    if (jb < 0) continue;
                    const unsigned long long mj = jb == 0 ? m[0] : (jb == 1 ? m[1] : m[2]);
                    const int jl = __builtin_ctzll(mj);
                    const int i = a3 * 64 + l, j = jb * 64 + jl;
                    vis[a3] |= 1ull << l;
#pragma unroll
                    for (int b3 = 0; b3 < 3; b3++)
                        if (b3 == jb) vis[b3] |= 1ull << jl;
                    if (lane == 0 && nm < CTAG_MAX_FEATURES) s_match[nm] = (uint32_t)i | ((uint32_t)j << 16);
                    nm++;
                }
            }
            if (lane == 0) s_nf = nm;
        }
        __syncthreads();
        clk.mark(3);
        {
            const int nm = min(s_nf, CTAG_MAX_FEATURES);
            for (int k0 = 0; k0 < nm; k0 += THREADS / 8) {  // eight lanes per feature (uniform trip count: the group shuffles)
                const int k = k0 + (tid >> 3);
                const bool have = k < nm;
                const uint32_t mk = s_match[have ? k : 0];
                const int i = (int)(mk & 0xffffu), j = (int)(mk >> 16);
                const QuadDerived &Di = s_der[i], &Dj = s_der[j];
                const float fa = (float)angdeg(Di.cy - Dj.cy, Di.cx - Dj.cx);
                feature_organization(s_qc[i], s_qc[j], Di.cx, Di.cy, Dj.cx, Dj.cy, fa, &f0[have ? k : 0], tid & 7, lane & ~7, have && (tid & 7) == 0);
            }
        }
    } else {
        for (int i = 0; i + 1 < Q; i++) {
            if (s_vis[i]) continue;  // uniform
            if (tid == 0) s_best = 0x7fffffff;
            __syncthreads();
            const QuadDerived Di = der[i];
            int mine = 0x7fffffff;
            for (int j = i + 1 + tid; j < Q && mine == 0x7fffffff; j += THREADS) {
                if (s_vis[j]) continue;
                if (feature_pair(Di, der[j], P.threshold_angle)) mine = j;
            }
            if (mine != 0x7fffffff) atomicMin(&s_best, mine);
            __syncthreads();
            const int j = s_best;
            __syncthreads();
            if (j != 0x7fffffff) {
                if (tid < 8) {
                    const int nf = s_nf;
                    const QuadDerived Dj = der[j];
                    const float fa = (float)angdeg(Di.cy - Dj.cy, Di.cx - Dj.cx);
                    feature_organization(quads[qidx[i]].c, quads[qidx[j]].c, Di.cx, Di.cy, Dj.cx, Dj.cy, fa, &f0[min(nf, CTAG_MAX_FEATURES - 1)], tid, 0, tid == 0 && nf < CTAG_MAX_FEATURES);
                }
                __syncthreads();
                if (tid == 0) {
                    s_vis[i] = 1;
                    s_vis[j] = 1;
                    s_nf = s_nf + 1;
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
    clk.mark(4);
    const int nf = s_nf;
    int status = CTAG_OK;
    if (nf < feature_size) status = CTAG_NO_FEATURE;
    else if (nf > CTAG_MAX_FEATURES) status = CTAG_ERR_LIMIT;
    // a GPU-only capacity was exceeded upstream (tile runs / slots, component pool, candidates, edge pools): part of the
    // frame was dropped, so the frame is reported as such instead of as a (possibly wrong) detection
    if (P.frame_flags[frame] & CTAG_FLAG_POOL_OVERFLOW) status = CTAG_ERR_LIMIT;
    if (tid == 0) {
        P.frame_long[frame] = 0;
        P.nfeat[frame] = min(nf, CTAG_MAX_FEATURES);
        P.status[frame] = status;
        if (nf > CTAG_MAX_FEATURES) atomicOr(&P.frame_flags[frame], CTAG_FLAG_FEATURE_OVERFLOW);
    }
    // cornerObtain (:561-569): half-res -> full-res coordinates
    for (int k = tid; k < min(nf, CTAG_MAX_FEATURES); k += THREADS) {
        FeatureDev F = f0[k];
        for (int q = 0; q < 16; q++) F.c[q] = (F.c[q] - 0.5f) * 2 + 0.5f;
        F.center[0] = (F.c[0] + F.c[2] + F.c[8] + F.c[10]) / 4;
        F.center[1] = (F.c[1] + F.c[3] + F.c[9] + F.c[11]) / 4;
        f1[k] = F;
        f2[k] = F;
    }
    clk.mark(5);
}

// =====================================================================================================
// K8
// =====================================================================================================
struct RefinePtrs {
    const uint8_t* frames;
    ptrdiff_t frame_stride, row_stride;
    const int32_t* nfeat;
    const int32_t* status;
    const FeatureDev* feat1;
    FeatureDev* feat2;
    double* n0;          // [F][CTAG_MAX_FEATURES * 2][4][kRefineSamples]: the search kernel's result per sample (NaN: no edge point), MODE 1 -> 2
    int32_t* frame_long; // [F] 1: the frame has a quad with an edge of more than kRefineSamples samples (those quads take the one-kernel form)
    int ch;              // 1: gray frames; 3: BGR frames (3 bytes per pixel), converted -- cvtColor(BGR2GRAY), gray_of -- where a pixel is loaded
};
constexpr int kRefineSamples = 128;               // samples of an edge searched per pass (the reference's minimum sample count, :615)
constexpr int kRefineThreads = 2 * kRefineSamples; // two edges side by side: waves 0-1 edge e, waves 2-3 edge e + 1
#ifndef CTAG_REFINE_SPLIT
#define CTAG_REFINE_SPLIT 1  // batches: searches and ordered sums as two kernels (k_edge_refine<MODE>)
#endif
#ifndef CTAG_REFINE_STAGE_DEEP
#define CTAG_REFINE_STAGE_DEEP 12
#endif
#ifndef CTAG_REFINE_REGION
#define CTAG_REFINE_REGION 21504                   // bytes of the quad's pixel neighbourhood staged in LDS (with the rest: 39.8 KB per block, 4 blocks per CU)
#endif
constexpr int kRefineRegion = CTAG_REFINE_REGION;
#ifndef CTAG_REFINE_REGION_LARGE
#define CTAG_REFINE_REGION_LARGE 49152             // ... of the search kernel's build for frames above 1920x1200: boxes up to ~220 x 220 px, three blocks per CU
#endif
constexpr int kRefineRegionLarge = CTAG_REFINE_REGION_LARGE;

// MODE 0: the whole of edgeRefine for a (feature, quad) in one block of 256 -- calls of a few frames, and quads with an edge of more
//         than kRefineSamples samples (edges longer than 1024 px; `only_long`).
// MODE 1: the searches only (256 threads): per sample n0 = Mn / Mcount goes to P.n0 (NaN: no edge point).  No sample rows in LDS, so
//         a block is the 21.5 KB pixel box and little else.
// The ordered sums, line parameters and corners of a batch are kernels of their own: k_edge_refine_sums (one wave per quad: rebuilds every sample's point from
// n0 with the search's own expressions) and k_edge_refine_tail, below.
// Batches run MODE 1, the sums, the tails (then MODE 0 for the rare long quads).  In one kernel the serial tail -- 48 chains of 128 dependent FP64 adds on one
// wave, then divisions / atan2 / sin / cos on 8 lanes -- held a 40 KB, four-wave block while three of its waves idled, at four
// blocks per CU; apart, the search blocks are smaller and the tails of many quads overlap each other.
#ifndef CTAG_REFINE_SEARCH_WAVES
#define CTAG_REFINE_SEARCH_WAVES 4
#endif
// Tail of edgeRefine for one quad, from its 48 ordered sums A[edge * 12 + pass * 6 + {Mx, My, Mxx, Mxy, Myy, N}]:
// refine_line: the line of (edge, pass) -> L = {Ex, Ey, nx, ny} (:667-678 / :743-754); refine_corner: corner `it` from the lines (:757-776).
__device__ __forceinline__ void refine_line(const double* A, double* L) {
    const double Mx = A[0], My = A[1], Mxx = A[2], Mxy = A[3], Myy = A[4], N = A[5];  // (L may be A: everything is read first)
    const ctm::Recip64 RN = ctm::recip64(N);  // five IEEE quotients by one denominator (ctag_math.h)
    const double Ex = ctm::div64(Mx, RN), Ey = ctm::div64(My, RN);
    const double Cxx = ctm::div64(Mxx, RN) - Ex * Ex;
    const double Cxy = ctm::div64(Mxy, RN) - Ex * Ey;
    const double Cyy = ctm::div64(Myy, RN) - Ey * Ey;
    const double normal_theta = .5 * ctm::atan2_32((float)(-2 * Cxy), (float)(Cyy - Cxx));
    L[0] = Ex;
    L[1] = Ey;
    float sn, cs;
    ctm::sincos32((float)normal_theta, &sn, &cs);  // == sin32, cos32 of the same angle: one reduction, no quadrant divergence
    L[2] = cs;
    L[3] = sn;
}
// A: the quad's 48-double block after refine_line ran IN PLACE on each (edge, pass) group of six (L = A + edge * 12 + pass * 6)
__device__ __forceinline__ void refine_corner(const double* A, int it, int off, const FeatureDev* F, FeatureDev* O) {
    const double* Ln = A + it * 12;                    // edge `it`, weighting towards the next corner
    const double* Ll = A + ((it + 1) & 3) * 12 + 6;    // the following edge, weighting towards the last corner
    const double A00 = Ln[3], A01 = -Ll[3];
    const double A10 = -Ln[2], A11 = Ll[2];
    const double B0 = -Ln[0] + Ll[0];
    const double B1 = -Ln[1] + Ll[1];
    const double det = A00 * A11 - A10 * A01;
    const double W00 = A11 / det, W01 = -A01 / det;
    const double L0 = W00 * B0 + W01 * B1;
    const int idx = ((it + 1) & 3) + off;
    if (ctm::fabs64(det) > 0.001) {
        O->c[2 * idx] = (float)(Ln[0] + L0 * A00);
        O->c[2 * idx + 1] = (float)(Ln[1] + L0 * A10);
    } else {
        O->c[2 * idx] = F->c[2 * idx];
        O->c[2 * idx + 1] = F->c[2 * idx + 1];
    }
}

// Barrier of the edgeRefine blocks: their phases exchange data through LDS only, so it waits for LDS traffic and NOT for vector memory -- __syncthreads() would also
// wait for the block's n0 stores (the vector-memory counter counts stores on gfx9) before the next quad's box may be requested.  A value a thread loaded itself is
// waited for where it is used, as always.
// Holds only while no phase of these blocks hands data to another thread through GLOBAL memory (none does: n0 is re-read by other kernels only).  A build with
// EXTRA=-DCTAG_REFINE_PLAIN_SYNC=1 replaces both macros by __syncthreads(): the parity tests pass with either (tests/test_refine_sync_gpu.py runs them on that build).
#if defined(CTAG_REFINE_PLAIN_SYNC) && CTAG_REFINE_PLAIN_SYNC
#define REFINE_SYNC() __syncthreads()
#else
#define REFINE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
// REGION: bytes of the quad's pixel neighbourhood staged in LDS.  kRefineRegion holds the box of every quad of a 1080p-class frame; frames above 1920x1200 have quads
// of twice the size (a diagonal strip's box is ~190 x 190 px, up to 240 x 240) and batches of them run the search kernel with kRefineRegionLarge -- three blocks per CU
// instead of four, but searches that gather from LDS: a box that does not fit leaves its searches to byte gathers from global memory (round 5: a 4K quad cost 2.3 x a
// 1080p quad for the same 512 searches).
template <int MODE, int REGION = kRefineRegion>
__device__ __forceinline__ bool refine_quad(const RefinePtrs& P, int rows, int cols, int subpix, int frame, int qidx, int only_long, double alpha128 = 0.0) {
    // per edge and sample: refined point and its position parameter; the 48 running sums (4 edges x {next,last}
    // weighting x 6 moments) are then accumulated in sample order, all at once, one sum per lane
    // 16 columns of per-sample values: rows 0-3 x of edge 0-3, 4-7 y, 8-11 weight towards the next corner, 12-15 towards the
    // last one (0 for a sample without an edge point).  The odd row length puts the 16 rows on 16 different LDS bank
    // pairs: the accumulation reads one element of up to 9 rows per instruction.
    // row 16 holds ones: the factors "1" of the sums Mx = (x * 1) * w, N = (1 * 1) * w are read like any other column, so that
    // every lane of the accumulation walks three unit-stride columns (immediate offsets, no address arithmetic in the loop)
    constexpr int kPitch = kRefineSamples + 1;
    constexpr int kRows = 16;
    // The row of ones sits 16 doubles (32 banks) further than a 17th row would: a row of 130 doubles starts 4 banks after its predecessor, so
    // rows 0-15 tile the 64 banks and a 17th row would share row 0's; the instructions that read ones (factor A or B of a sum) read x / y rows 0-7
    // beside it, never the weight rows 8-15 whose banks it now shares (round 4: the sums kernel's 19 % bank conflicts were these two rows)
#ifndef CTAG_ONES_SHIFT
#define CTAG_ONES_SHIFT 16
#endif
    __shared__ __attribute__((aligned(16))) double s_vall[MODE == 1 ? kPitch : (kRows + 1) * kPitch + CTAG_ONES_SHIFT];
    double (*s_v)[kPitch] = reinterpret_cast<double (*)[kPitch]>(s_vall);
    double* const s_ones = MODE == 1 ? s_vall : s_vall + kRows * kPitch + CTAG_ONES_SHIFT;
    double (*s_bx)[kPitch] = s_v, (*s_by)[kPitch] = s_v + (MODE == 1 ? 0 : 4);
    __shared__ double s_nrm[4][2];                // unit normal of each edge
    __shared__ uint64_t s_step[4][2];             // ctr::fast_step of the normal's components
    __shared__ double s_acc[48];
    // The pixels the searches of this quad can touch -- the bounding box of its corners grown by the search length --
    // staged once with coalesced row loads: the 4 x 128 x 49 scattered byte loads of the searches then gather from LDS
    // instead of going through the texture-address path, which was as busy as the vector ALUs (70 % TA busy,
    // profiles/r02a_before_round2_work_instmix.json).  A box that does not fit stays in global memory (uniform per block).
    __shared__ __attribute__((aligned(16))) uint8_t s_reg[REGION];
    const int fi = qidx >> 1, quad = qidx & 1;  // (the caller's loop keeps qidx below twice the frame's feature count: no load of it here, once per quad)
    const int tid = threadIdx.x;
    const FeatureDev* F = P.feat1 + (size_t)frame * CTAG_MAX_FEATURES + fi;
    const uint8_t* __restrict__ img = P.frames + (ptrdiff_t)frame * P.frame_stride;
    const uint32_t rs = (uint32_t)P.row_stride;
    const int off = quad * 4;
    __shared__ float s_cx[4], s_cy[4];
    __shared__ int s_ns[4];
    __shared__ int s_box[4];  // x0, y0 of the staged box, its pitch, its rows (0 = not staged)
    if (tid < 4) {
        s_cx[tid] = F->c[2 * (off + tid)];
        s_cy[tid] = F->c[2 * (off + tid) + 1];
    }
    double* const accp = s_acc;
    if (MODE != 1 && tid < 48) accp[tid] = 0.0;
    if constexpr (MODE != 1) {
        for (int k = tid; k < kPitch; k += (int)blockDim.x) s_ones[k] = 1.0;
    }
    REFINE_SYNC();
    if (tid < 4) {  // :609-615
        const int a = tid, b = (tid + 1) & 3;
        const double nx = s_cy[b] - s_cy[a];
        const double ny = -s_cx[b] + s_cx[a];
        const double mag = ctm::sqrt64(nx * nx + ny * ny);
        const double ns_d = mag / 8 > 128.0 ? mag / 8 : 128.0;
        s_ns[tid] = (int)ns_d;
        s_nrm[tid][0] = nx / mag;
        s_nrm[tid][1] = ny / mag;
        s_step[tid][0] = ctr::fast_step(nx / mag);  // the searches' pixel step along this edge's normal, once per edge instead of per sample
        s_step[tid][1] = ctr::fast_step(ny / mag);
    }
    if (tid == 64) {  // the box: corners +- (search length + 2), clipped to the image; columns from a multiple of 4
        const float m = (float)(subpix + 3);
        const float fx0 = fminf(fminf(s_cx[0], s_cx[1]), fminf(s_cx[2], s_cx[3])) - m, fx1 = fmaxf(fmaxf(s_cx[0], s_cx[1]), fmaxf(s_cx[2], s_cx[3])) + m;
        const float fy0 = fminf(fminf(s_cy[0], s_cy[1]), fminf(s_cy[2], s_cy[3])) - m, fy1 = fmaxf(fmaxf(s_cy[0], s_cy[1]), fmaxf(s_cy[2], s_cy[3])) + m;
        int staged = 0, bx0 = 0, by0 = 0, pitch = 4;
        // (a corner far outside the image cannot come out of detect(); the comparisons also reject NaN)
        if (fx0 > -1e6f && fx1 < 1e6f && fy0 > -1e6f && fy1 < 1e6f) {
            bx0 = max((int)floorf(fx0), 0) & ~3;
            by0 = max((int)floorf(fy0), 0);
            const int bx1 = min((int)floorf(fx1) + 1, cols - 1), by1 = min((int)floorf(fy1) + 1, rows - 1);
            if (bx1 >= bx0 && by1 >= by0) {
                pitch = (bx1 - bx0 + 4) & ~3;
                if (((pitch >> 2) & 1) == 0) pitch += 4;  // odd number of banks per row: a column of pixels spreads over the banks
                staged = ((long long)pitch * (by1 - by0 + 1) <= REGION && (pitch >> 2) <= kRefineThreads) ? by1 - by0 + 1 : 0;
            }
        }
        s_box[0] = bx0;
        s_box[1] = by0;
        s_box[2] = pitch;
        s_box[3] = staged;  // rows staged, 0 = the box stays in global memory
    }
    REFINE_SYNC();
    const int max_ns = max(max(s_ns[0], s_ns[1]), max(s_ns[2], s_ns[3]));
    {   // which form takes this quad: an edge of more than kRefineSamples samples needs several passes -> the one-kernel form
        const bool long_quad = max_ns > kRefineSamples;
        if (MODE == 1 && long_quad && tid == 0) P.frame_long[frame] = 1;
        if (MODE != 0 && long_quad) return false;
        if (MODE == 0 && only_long && !long_quad) return false;
    }
    const int box_x0 = s_box[0], box_y0 = s_box[1], box_pitch = s_box[2], box_rows = s_box[3];
    const bool staged = box_rows > 0;
    if (staged) {
        // rows of the box, 4 bytes per lane: aligned words when the frame's rows allow it, bytes otherwise.  A thread keeps its
        // column and walks down the rows, four rows requested before the first is stored (the loads of a row-major loop with
        // a division per word were issued and awaited one at a time: staging a large box then cost more than it saved).
        const int wpr = box_pitch >> 2;              // words per row
        const int rpp = kRefineThreads / wpr;        // rows per pass (>= 1: a staged box is at most kRefineThreads words wide)
        const bool aligned = ((rs & 3u) == 0u) && ((reinterpret_cast<uintptr_t>(img) & 3u) == 0u);
        const int r0 = tid / wpr, c4 = (tid - r0 * wpr) * 4;
        const int gx = box_x0 + c4;
        if (rpp > 0 && r0 < rpp) {
            // `load`: the word of row r in this thread's column.  Two forms, chosen per BLOCK: when every word of the box is an aligned word
            // inside its row (almost always) the loads are plain -- with the general form's per-thread test inside, every load of the
            // unrolled groups below came out as byte loads + a wait + a word load, one round trip to memory PER ROW (the staging phase
            // was 64 % of a search block's lifetime).
            // `load`: what row r of this thread's column needs from memory (requested for a batch of rows before anything is used), `conv`: that -> the word of
            // four gray pixels
            auto stage_rows = [&](auto load, auto conv) {
                int r = r0;
                if constexpr (MODE == 1) {
                    // the search kernel has its registers free at this point: twelve rows in flight (the usual 10 KB box: 10 per thread)
                    constexpr int kDeep = CTAG_REFINE_STAGE_DEEP;
                    for (; r + (kDeep - 1) * rpp < box_rows; r += kDeep * rpp) {
                        decltype(load(0)) v[kDeep];
#pragma unroll
                        for (int u = 0; u < kDeep; u++) v[u] = load(r + u * rpp);
#pragma unroll
                        for (int u = 0; u < kDeep; u++) *reinterpret_cast<uint32_t*>(s_reg + (r + u * rpp) * box_pitch + c4) = conv(v[u]);
                    }
                }
                for (; r + 3 * rpp < box_rows; r += 4 * rpp) {
                    const auto v0 = load(r), v1 = load(r + rpp), v2 = load(r + 2 * rpp), v3 = load(r + 3 * rpp);
                    *reinterpret_cast<uint32_t*>(s_reg + r * box_pitch + c4) = conv(v0);
                    *reinterpret_cast<uint32_t*>(s_reg + (r + rpp) * box_pitch + c4) = conv(v1);
                    *reinterpret_cast<uint32_t*>(s_reg + (r + 2 * rpp) * box_pitch + c4) = conv(v2);
                    *reinterpret_cast<uint32_t*>(s_reg + (r + 3 * rpp) * box_pitch + c4) = conv(v3);
                }
                for (; r < box_rows; r += rpp) *reinterpret_cast<uint32_t*>(s_reg + r * box_pitch + c4) = conv(load(r));
            };
            auto same = [](uint32_t v) -> uint32_t { return v; };
            if (P.ch == 3) {  // BGR frames: four pixels are twelve bytes, converted as they are staged
                struct W3 {
                    uint32_t a, b, c;
                };
                if (aligned && box_x0 + box_pitch <= cols) {
                    stage_rows(
                        [&](int r) -> W3 {
                            const uint32_t* src = reinterpret_cast<const uint32_t*>(img + (size_t)__umul24((unsigned)(box_y0 + r), rs) + 3 * gx);
                            return W3{src[0], src[1], src[2]};
                        },
                        [](const W3& w) -> uint32_t { return gray4_of(w.a, w.b, w.c); });
                } else {
                    stage_rows(
                        [&](int r) -> uint32_t {
                            const uint8_t* src = img + (size_t)__umul24((unsigned)(box_y0 + r), rs) + 3 * gx;
                            uint32_t v = 0;
                            for (int k = 0; k < 4; k++)
                                if (gx + k < cols) v |= gray_of(src[3 * k], src[3 * k + 1], src[3 * k + 2]) << (8 * k);
                            return v;
                        },
                        same);
                }
            } else if (aligned && box_x0 + box_pitch <= cols) {
                stage_rows([&](int r) -> uint32_t { return *reinterpret_cast<const uint32_t*>(img + (size_t)__umul24((unsigned)(box_y0 + r), rs) + gx); }, same);
            } else {
                stage_rows(
                    [&](int r) -> uint32_t {
                        const uint8_t* src = img + (size_t)__umul24((unsigned)(box_y0 + r), rs) + gx;
                        uint32_t v = 0;
                        for (int k = 0; k < 4; k++)
                            if (gx + k < cols) v |= (uint32_t)src[k] << (8 * k);
                        return v;
                    },
                    same);
            }
        }
        REFINE_SYNC();
    }
    const int half = tid >> 7, st = tid & (kRefineSamples - 1);
    double* const n0_quad = P.n0 + ((size_t)frame * (CTAG_MAX_FEATURES * 2) + qidx) * (4 * kRefineSamples);
    for (int sbase = 0; sbase < max_ns; sbase += kRefineSamples) {
        const int s = sbase + st;
#pragma nounroll
        for (int epair = 0; epair < 2; epair++) {  // rolled on purpose: one copy of the search loops
            const int edge = 2 * epair + half;
            const int a = edge, b = (edge + 1) & 3;
            const float ax = s_cx[a], ay = s_cy[a], bx = s_cx[b], by = s_cy[b];
            const int nsamples = s_ns[edge];
            const double nx = s_nrm[edge][0], ny = s_nrm[edge][1];
            const bool axis = nx == 0.0 || ny == 0.0;
            const uint64_t step_xy[2] = {s_step[edge][0], s_step[edge][1]};
            bool ok = false;
            double bestx = 0, besty = 0, alpha = 0;
            if (s < nsamples) {
                // (the search kernel's edges all have kRefineSamples samples: the quotient depends on the thread only, and its caller has it)
                alpha = MODE == 1 ? alpha128 : (15.0 + s) / (nsamples + 30);
                const double x0 = alpha * ax + (1 - alpha) * bx;
                const double y0 = alpha * ay + (1 - alpha) * by;
                // the normal search of this sample (:623-657): ctag_refine.h -- the fast form where every pixel of the search is
                // inside the image, the reference's own arithmetic otherwise or when the fast form declines
                double Mn = 0, Mcount = 0;
                const bool bgr = P.ch == 3;
                auto px = [&](int x, int y) -> unsigned {  // 32-bit pixel offsets (checked by the API)
                    if (bgr) {
                        const uint8_t* q = img + __umul24((unsigned)y, rs) + 3u * (unsigned)x;
                        return gray_of(q[0], q[1], q[2]);
                    }
                    return img[__umul24((unsigned)y, rs) + (unsigned)x];
                };
                bool done = false;
                if (subpix <= ctr::kFastMaxSubpix && ctr::interior(x0, y0, nx, ny, subpix, rows, cols)) {
                    if (staged) {  // the walk runs in box coordinates: the address is one multiply-add
                        auto px_lds = [&](int x, int y) -> unsigned { return s_reg[__umul24((unsigned)y, (unsigned)box_pitch) + (unsigned)x]; };
                        // an axis-aligned edge (uniform over the two waves of an edge) steps by exactly +-1/4 px: from corners at x.5 every
                        // fourth step of every sample is on a pixel border, the fast form would decline them all -> middle form at once
                        if (!axis)
                            done = subpix == 5 ? ctr::search_fast<5>(x0, y0, nx, ny, 5, px_lds, Mn, Mcount, box_x0, box_y0, step_xy)  // main.cpp:39,57
                                               : ctr::search_fast<0>(x0, y0, nx, ny, subpix, px_lds, Mn, Mcount, box_x0, box_y0, step_xy);
                        if (!done) {
                            ctr::search_mid<0>(x0, y0, nx, ny, subpix, px_lds, Mn, Mcount, box_x0, box_y0);
                            done = true;
                        }
                    } else {
                        done = ctr::search_fast(x0, y0, nx, ny, subpix, px, Mn, Mcount);
                    }
                }
                if (!done) ctr::search_exact(x0, y0, nx, ny, subpix, rows, cols, px, Mn, Mcount);
                if (Mcount != 0) {
                    const double n0 = Mn / Mcount;
                    bestx = x0 + n0 * nx;
                    besty = y0 + n0 * ny;
                    ok = true;
                    if constexpr (MODE == 1) n0_quad[edge * kRefineSamples + st] = n0;
                }
            }
            if constexpr (MODE == 1) {
                if (!ok) n0_quad[edge * kRefineSamples + st] = ctm::bits_to_f64(0x7ff8000000000000ULL);  // no edge point (also s >= nsamples: never, ns == 128 here)
            } else {
                s_bx[edge][st] = bestx;  // 0 when !ok
                s_by[edge][st] = besty;
                s_v[8 + edge][st] = ok ? 1 - alpha : 0.0;
                s_v[12 + edge][st] = ok ? alpha : 0.0;
            }
        }
        if constexpr (MODE == 1) return false;  // one pass: every edge of the quad has kRefineSamples samples
        REFINE_SYNC();
        if (tid < 48) {  // sequential (sample-order) accumulation: bit-identical to the reference's running sums
            // every sum has the form (A * B) * w with A, B in {x, y, 1} (x * 1 and 1 * 1 are exact); a sample without an
            // edge point has x = y = w = 0 and adds +0.0, which equals the reference skipping it
            const int edge = tid / 12, r = tid - edge * 12;
            const int pass = r / 6, which = r - pass * 6;
            const double* pa = (which == 0 || which == 2 || which == 3) ? s_bx[edge] : (which == 5 ? s_ones : s_by[edge]);
            const double* pb = which == 2 ? s_bx[edge] : ((which == 3 || which == 4) ? s_by[edge] : s_ones);
            const double* pw = s_v[8 + 4 * pass + edge];
            double acc = accp[tid];
            const int cntS = min(kRefineSamples, s_ns[edge] - sbase);
            int k = 0;
            {
                for (; k + 16 <= cntS; k += 16) {
#pragma unroll
                    for (int u = 0; u < 16; u++) acc += (pa[k + u] * pb[k + u]) * pw[k + u];
                }
            }
            for (; k < cntS; k++) acc += (pa[k] * pb[k]) * pw[k];
            accp[tid] = acc;
        }
        REFINE_SYNC();
    }
    if (tid < 8) {  // line of (edge, pass)
        const int edge = tid >> 1, pass = tid & 1;
        refine_line(s_acc + edge * 12 + pass * 6, s_acc + edge * 12 + pass * 6);
    }
    REFINE_SYNC();
    if (tid < 4) refine_corner(s_acc, tid, off, F, P.feat2 + (size_t)frame * CTAG_MAX_FEATURES + fi);  // one refined corner per lane
    return true;
}

template <int MODE, int REGION = kRefineRegion>
__global__ __launch_bounds__(kRefineThreads) __attribute__((amdgpu_waves_per_eu(MODE == 1 ? (REGION > 36864 ? 3 : CTAG_REFINE_SEARCH_WAVES) : 1, MODE == 1 ? CTAG_REFINE_SEARCH_WAVES : 8)))
void k_edge_refine(RefinePtrs P, int rows, int cols, int subpix, int nframes, int per_frame) {
    // per_frame > 0: a 1-D grid of per_frame blocks per frame in which blocks b and b + 8 -- one XCD -- belong to the same frame: the
    // boxes of a frame's quads overlap, and on one XCD the shared pixels come out of its L2 instead of HBM
    int frame = blockIdx.y, bx = blockIdx.x, gx = gridDim.x;
    if (per_frame > 0) {
        const int b = blockIdx.x;
        frame = ((b >> 3) / per_frame) * 8 + (b & 7);
        bx = (b >> 3) % per_frame;
        gx = per_frame;
    }
    if (frame >= nframes) return;
    if (P.status[frame] != CTAG_OK) return;
    // blocks loop over the frame's quads: a grid of one block per possible quad (2 * CTAG_MAX_FEATURES) would launch more blocks
    // that find nothing to do than blocks that work
    const int nq = 2 * min(P.nfeat[frame], CTAG_MAX_FEATURES);
    const double alpha128 = (15.0 + (double)((int)threadIdx.x & (kRefineSamples - 1))) / (kRefineSamples + 30);  // == (15.0 + s) / (nsamples + 30) at 128 samples
    for (int q = bx; q < nq; q += gx) {
        refine_quad<MODE, REGION>(P, rows, cols, subpix, frame, q, 0, alpha128);
        if (q + gx < nq) REFINE_SYNC();
    }
}
// ---- the sums kernel (round 5): TERMS in LDS.  Its round-4 form (k_edge_refine<2>, docs/history.md) kept rows of x, y, the products and the weights, and every step of
// a sum was a load of two operands, a multiplication and the (ordered) addition; 64 registers of operand buffers held it to three waves per SIMD and half of its time it
// waited: 1.03 ms per 4096 frames.  Here the lane that builds a sample also multiplies -- (A B) w, the same two roundings in the same order, for the sample's twelve sums --
// and a sum's step is one addition: segments of 16 samples (64 lanes = 4 edges x 16), 48 rows of 16 terms (6.9 KB), a chain's row index IS its lane
// (12 edge + 6 pass + moment), ~100 registers, four waves per SIMD.
// What bounds it now is LDS BANDWIDTH: a quad's 48 x 128 terms are written once and read once, 98 KB, and 393 K quads per 4096 frames are 38.5 GB against the
// 78 TB/s the 256 CUs' LDS move at 128 B per clock: 0.49 ms; the kernel runs 0.70-0.73.  Timing builds without its global loads, without its stores, without
// both: 4.40 / 4.40 / 4.39 against 4.45 ms edge_refine -- memory is not what it waits for; halving its vector instructions (the position parameters' divisions,
// a table now) did not move it either.  The transposition sample-lane -> sum-lane is the work.
#ifndef CTAG_REFINE_SUMS2_WAVES
#define CTAG_REFINE_SUMS2_WAVES 4
#endif
constexpr int kSumSeg = 16, kSumSegs = kRefineSamples / kSumSeg;
constexpr int kSumPitch = kSumSeg + 2;  // doubles per row: 36 words -- an odd multiple of four, so the ds_read_b128 of 16 consecutive lanes (rows) cover the 64 banks exactly once
// The sums kernel's block is ONE wave, and LDS serves a wave's accesses in order: its phases are ordered by a wait for LDS traffic alone.  __syncthreads() would also wait
// for vector memory -- for the NEXT quad's n0, requested a quad ahead precisely so that nobody waits for it (with __syncthreads() the kernel ran at 0.38 of its
// vector-issue rate: profiles/r05_pmc_instmix.json).
// No barrier at all: correct ONLY for a block of exactly one wave64 -- k_edge_refine_sums (its only user) checks its launch size and traps otherwise.
#if defined(CTAG_REFINE_PLAIN_SYNC) && CTAG_REFINE_PLAIN_SYNC
#define SUMS_SYNC() __syncthreads()
#else
#define SUMS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(__GFX9__) && defined(__AMDGCN_WAVEFRONT_SIZE__) && __AMDGCN_WAVEFRONT_SIZE__ != 64
#error "SUMS_SYNC assumes 64-wide wavefronts"
#endif
struct SumsPrefetch {
    double n0[kSumSegs];  // of sample 16 g + (tid & 15) of edge tid >> 4
    float cx, cy;
};
__device__ __forceinline__ void sums_prefetch(const RefinePtrs& P, int frame, int qidx, SumsPrefetch& R) {
    const int tid = threadIdx.x;
    const double* src = P.n0 + ((size_t)frame * (CTAG_MAX_FEATURES * 2) + qidx) * (4 * kRefineSamples) + (tid >> 4) * kRefineSamples + (tid & 15);
#pragma unroll
    for (int g = 0; g < kSumSegs; g++) R.n0[g] = src[g * kSumSeg];
    const FeatureDev* F = P.feat1 + (size_t)frame * CTAG_MAX_FEATURES + (qidx >> 1);
    const int c = (qidx & 1) * 4 + (tid & 3);
    R.cx = F->c[2 * c];
    R.cy = F->c[2 * c + 1];
}
// the 48 ordered sums of one quad -> acc_out[edge * 12 + pass * 6 + {Mx, My, Mxx, Mxy, Myy, N}]; false: no such quad, or one with an edge of more than kRefineSamples samples
// (the caller has checked qidx against the frame's feature count; s_alpha: the samples' position parameters, the same for every quad)
__device__ __forceinline__ bool refine_sums_quad(const SumsPrefetch& pre, const double* s_alpha, double* acc_out) {
    __shared__ __attribute__((aligned(16))) double s_t[48][kSumPitch];
    __shared__ float s_cx[4], s_cy[4];
    __shared__ double s_nrm[4][2];
    __shared__ int s_ns[4];
    const int tid = threadIdx.x;
    if (tid < 4) {
        s_cx[tid] = pre.cx;
        s_cy[tid] = pre.cy;
    }
    SUMS_SYNC();
    if (tid < 4) {  // :609-615
        const int a = tid, b = (tid + 1) & 3;
        const double nx = s_cy[b] - s_cy[a];
        const double ny = -s_cx[b] + s_cx[a];
        const double mag = ctm::sqrt64(nx * nx + ny * ny);
        const double ns_d = mag / 8 > 128.0 ? mag / 8 : 128.0;
        s_ns[tid] = (int)ns_d;
        s_nrm[tid][0] = nx / mag;
        s_nrm[tid][1] = ny / mag;
    }
    SUMS_SYNC();
    if (max(max(s_ns[0], s_ns[1]), max(s_ns[2], s_ns[3])) > kRefineSamples) return false;  // k_edge_refine_long's
    const int edge = tid >> 4, loc = tid & 15;
    const float ax = s_cx[edge], ay = s_cy[edge], bx = s_cx[(edge + 1) & 3], by = s_cy[(edge + 1) & 3];
    const double nx = s_nrm[edge][0], ny = s_nrm[edge][1];
    double* const mine = &s_t[edge * 12][loc];
    const double2* const row2 = reinterpret_cast<const double2*>(s_t[tid < 48 ? tid : 0]);
    double acc = 0.0;
#pragma unroll
    for (int g = 0; g < kSumSegs; g++) {
        // the sample's point with the search's own expressions: x0 = alpha ax + (1 - alpha) bx, best = x0 + n0 nx (:619-621, :656-657)
        const double alpha = s_alpha[kSumSeg * g + loc];  // (15.0 + s) / (nsamples + 30) (:619): every edge of this quad has kRefineSamples samples, so the quotient is per sample index
        const double x0 = alpha * ax + (1 - alpha) * bx;
        const double y0 = alpha * ay + (1 - alpha) * by;
        const double n0 = pre.n0[g];
        const bool ok = n0 == n0;  // NaN: no edge point
        const double x = ok ? x0 + n0 * nx : 0.0, y = ok ? y0 + n0 * ny : 0.0;
        const double wn = ok ? 1 - alpha : 0.0, wl = ok ? alpha : 0.0;  // weights towards the next / the last corner; a sample without an edge point adds +0.0 to every sum
        const double xx = x * x, xy = x * y, yy = y * y;
        if (g) SUMS_SYNC();  // the sums of the segment before are done with the rows
        mine[0 * kSumPitch] = x * wn;   // (x 1) w: x 1 is exact
        mine[1 * kSumPitch] = y * wn;
        mine[2 * kSumPitch] = xx * wn;  // (A B) w: the product first, as the reference's left-to-right evaluation rounds it
        mine[3 * kSumPitch] = xy * wn;
        mine[4 * kSumPitch] = yy * wn;
        mine[5 * kSumPitch] = wn;       // (1 1) w
        mine[6 * kSumPitch] = x * wl;
        mine[7 * kSumPitch] = y * wl;
        mine[8 * kSumPitch] = xx * wl;
        mine[9 * kSumPitch] = xy * wl;
        mine[10 * kSumPitch] = yy * wl;
        mine[11 * kSumPitch] = wl;
        SUMS_SYNC();
        if (tid < 48) {  // sequential (sample-order) accumulation: bit-identical to the reference's running sums
            double t[kSumSeg];
#pragma unroll
            for (int u = 0; u < kSumSeg / 2; u++) {
                const double2 v = row2[u];
                t[2 * u] = v.x, t[2 * u + 1] = v.y;
            }
#pragma unroll
            for (int u = 0; u < kSumSeg; u++) acc += t[u];
        }
    }
    if (tid < 48) acc_out[tid] = acc;
    SUMS_SYNC();
    return true;
}
// The sums go back to global memory -- into the first 49 doubles of the quad's own n0 block, which is dead once they are formed: [0, 48) the sums, [48] 1.0 when the
// quad has them (0.0: no such quad here, or k_edge_refine_long's) -- and the tails (line parameters, corners) are a kernel of their own, k_edge_refine_tail: five
// divisions, atan2, sin / cos per line in double are what the sums kernel's registers went to, and a chain as long as a quad's sums that only 8 lanes walk.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CTAG_REFINE_SUMS2_WAVES, 8)))
void k_edge_refine_sums(RefinePtrs P, int nframes, int per_frame) {
    // per_frame blocks per frame, blocks b and b + 8 -- one XCD -- on the same frame; a block loops over the frame's quads with the next quad's inputs in flight
    if (blockDim.x != 64) __builtin_trap();  // SUMS_SYNC orders the phases of ONE wave; any other launch size would race silently
    const int b = blockIdx.x;
    const int frame = ((b >> 3) / per_frame) * 8 + (b & 7);
    const int bx = (b >> 3) % per_frame;
    if (frame >= nframes) return;
    if (P.status[frame] != CTAG_OK) return;
    const int nq = 2 * min(P.nfeat[frame], CTAG_MAX_FEATURES);
    int q = bx;
    if (q >= nq) return;
    __shared__ double s_acc[48];
    __shared__ double s_alpha[kRefineSamples];  // a double division per sample and quad otherwise: more than half of the kernel's vector instructions
    const int tid = threadIdx.x;
    for (int k = tid; k < kRefineSamples; k += 64) s_alpha[k] = (15.0 + k) / (kRefineSamples + 30);
    // The loop is ROTATED so that nothing a quad needs has been requested less than a quad's time ago -- and it took the ISA to see that it was not so: with
    // "request next, work, store, cur = next" the compiler's wait-count pass, merging the loop's entry (cur's own loads outstanding) with its back edge, put
    // s_waitcnt vmcnt(0) before the first use of `cur`, right behind the requests for the next quad (so the prefetch hid nothing, here and in round 4's kernel),
    // and the copy at the end waited for the stores just issued.  Now: cur's first loads are awaited explicitly before the loop (the pass understands
    // S_WAITCNT), and an iteration is work(cur) -> cur = next (a wait for loads a quad old) -> request the quad after -> store this quad's sums.
    SumsPrefetch cur, nxt;
    sums_prefetch(P, frame, q, cur);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    int qn = q + per_frame;
    if (qn < nq) sums_prefetch(P, frame, qn, nxt);
    for (;;) {
        const bool have = refine_sums_quad(cur, s_alpha, s_acc);
        const double mine = tid < 48 ? s_acc[tid] : (have ? 1.0 : 0.0);
        double* const out = P.n0 + ((size_t)frame * (CTAG_MAX_FEATURES * 2) + q) * (4 * kRefineSamples);
        SUMS_SYNC();
        q = qn;
        qn += per_frame;
        if (q < nq) {
            cur = nxt;
            if (qn < nq) sums_prefetch(P, frame, qn, nxt);
        }
        if (tid < 48 ? have : tid == 48) out[tid] = mine;  // [0, 48): the sums; [48]: 1.0 when the quad has them
        if (q >= nq) break;
    }
}
// lines and corners of the quads whose sums k_edge_refine_sums left: 8 quads per wave -- a lane per (edge, pass) line (:667-678 / :743-754), then a lane per corner (:757-776)
__global__ __launch_bounds__(64) void k_edge_refine_tail(RefinePtrs P, int nframes, int per_frame) {
    const int frame = blockIdx.y;
    if (frame >= nframes || P.status[frame] != CTAG_OK) return;
    const int nq = 2 * min(P.nfeat[frame], CTAG_MAX_FEATURES);
    __shared__ double s_L[8][48];
    __shared__ int s_ok[8];
    const int tid = threadIdx.x, sl = tid >> 3, ep = tid & 7;
    for (int q0 = (int)blockIdx.x * 8; q0 < nq; q0 += per_frame * 8) {
        const int q = q0 + sl;
        const double* A = P.n0 + ((size_t)frame * (CTAG_MAX_FEATURES * 2) + min(q, nq - 1)) * (4 * kRefineSamples);
        const bool ok = q < nq && A[48] == 1.0;
        if (ep == 0) s_ok[sl] = ok ? 1 : 0;
        if (ok) {
            double a6[6];
#pragma unroll
            for (int k = 0; k < 6; k++) a6[k] = A[(ep >> 1) * 12 + (ep & 1) * 6 + k];
            double* L = s_L[sl] + (ep >> 1) * 12 + (ep & 1) * 6;
            refine_line(a6, L);
        }
        __syncthreads();
        const int sc = tid >> 2;
        if (sc < 8 && s_ok[sc]) {
            const int qq = q0 + sc, fi = qq >> 1;
            refine_corner(s_L[sc], tid & 3, (qq & 1) * 4, P.feat1 + (size_t)frame * CTAG_MAX_FEATURES + fi, P.feat2 + (size_t)frame * CTAG_MAX_FEATURES + fi);
        }
        __syncthreads();
    }
}
// the quads k_edge_refine<1> / <2> left out (an edge of more than kRefineSamples samples), a few blocks per frame looping over the
// frame's quads: nothing to do in almost every frame, so it must cost nothing there
__global__ __launch_bounds__(kRefineThreads) void k_edge_refine_long(RefinePtrs P, int rows, int cols, int subpix, int nframes) {
    const int frame = blockIdx.y;
    if (frame >= nframes) return;
    if (P.status[frame] != CTAG_OK || P.frame_long[frame] == 0) return;
    const int nq = 2 * min(P.nfeat[frame], CTAG_MAX_FEATURES);
    for (int q = blockIdx.x; q < nq; q += gridDim.x) {
        refine_quad<0>(P, rows, cols, subpix, frame, q, 1);
        __syncthreads();
    }
}

// =====================================================================================================
// K9
// =====================================================================================================
struct MarkerPtrs {
    const int32_t* nfeat;
    const int32_t* status;
    const uint32_t* frame_flags;
    const FeatureDev* feat;
    const int32_t* dict;
    ctag_frame_result* pre;  // optional (debug)
    ctag_frame_result* out;
    unsigned long long* stamps;  // developer aid (CTAG_FEAT_STAMPS=1), slots 16..
    const uint32_t* dict_pos;    // [dict_rows][64] columns holding symbol v, bit c = column c; null when the dictionary has > 32 columns
    KParams kp;                  // tunables: angle, vertical, cross-ratio tables
    PendingCtx pend;             // frames that exceeded this workspace's pools (device-memory calls list them for the any-frame pass)
    int big;                     // this IS the any-frame workspace: an overflow is final
};

// featureExtraction for one feature (:1056-1207); C = 8 corners (x,y), swapped in place when direction == 0
__device__ void feature_ids(float* C, int direction, int& ID_left, int& ID_right, float& crl, float& crr, int& id, int& idl, int& idr, const float* IDc,
                            const float* covL, const float* covR) {  // ID_cr_correspond, cr_covariance_left / _right (ctag_params)
    auto Pt = [&](int k) { return P2{C[2 * k], C[2 * k + 1]}; };
    if (!direction) {
        if (C[0] > C[8]) {
            for (int k = 0; k < 8; k++) {
                const float t = C[k];
                C[k] = C[8 + k];
                C[8 + k] = t;
            }
        }
    }
    float l1[4], l2[4];
    l1[0] = dist2p(Pt(0), Pt(3));
    l1[1] = dist2p(Pt(3), Pt(6));
    l1[2] = dist2p(Pt(6), Pt(5));
    l1[3] = dist2p(Pt(0), Pt(5));
    l2[0] = dist2p(Pt(1), Pt(2));
    l2[1] = dist2p(Pt(2), Pt(7));
    l2[2] = dist2p(Pt(7), Pt(4));
    l2[3] = dist2p(Pt(1), Pt(4));
    crl = (l1[0] + l1[1]) * (l1[2] + l1[1]) / ((l1[1] * l1[3]));
    crr = (l2[0] + l2[1]) * (l2[2] + l2[1]) / ((l2[1] * l2[3]));
    struct P3 {
        float x, y, z;
    };
    auto mkline = [](P2 p, P2 q, P2 on) {
        P3 l;
        l.x = p.y - q.y;
        l.y = q.x - p.x;
        l.z = -l.x * on.x - l.y * on.y;
        return l;
    };
    const P3 line1 = mkline(Pt(5), Pt(4), Pt(5));
    const P3 line2 = mkline(Pt(0), Pt(1), Pt(0));
    const P3 lc1 = mkline(Pt(0), Pt(4), Pt(0));
    const P3 lc2 = mkline(Pt(5), Pt(1), Pt(5));
    const P3 ll = mkline(Pt(5), Pt(0), Pt(5));
    const P3 lr = mkline(Pt(1), Pt(4), Pt(1));
    P2 vanish{0, 0}, middle{0, 0}, mleft{0, 0}, mright{0, 0};
    solve2x2f(line1.x, line1.y, line2.x, line2.y, -line1.z, -line2.z, vanish.x, vanish.y);
    solve2x2f(lc1.x, lc1.y, lc2.x, lc2.y, -lc1.z, -lc2.z, middle.x, middle.y);
    P3 ml;
    ml.x = middle.y - vanish.y;
    ml.y = vanish.x - middle.x;
    ml.z = -ml.x * middle.x - ml.y * middle.y;
    solve2x2f(ml.x, ml.y, ll.x, ll.y, -ml.z, -ll.z, mleft.x, mleft.y);
    solve2x2f(ml.x, ml.y, lr.x, lr.y, -ml.z, -lr.z, mright.x, mright.y);
    float d1, d2, d3, d4;
    bool is_long = false;
    d1 = dist2p(mleft, Pt(0));
    d2 = dist2p(mleft, Pt(3));
    d3 = dist2p(mleft, Pt(5));
    d4 = dist2p(mleft, Pt(6));
    if (d2 * d3 < d1 * d4) is_long = true;
    for (int j = 0; j < 4; j++) {
        if ((IDc[j] >= crl) && (IDc[j] - crl < covL[j])) ID_left = is_long ? 7 - j : j;
        if ((IDc[j] < crl) && (crl - IDc[j] < covR[j])) ID_left = is_long ? 7 - j : j;
    }
    is_long = false;
    d1 = dist2p(mleft, Pt(1));  // measured from middle_left again
    d2 = dist2p(mleft, Pt(2));
    d3 = dist2p(mleft, Pt(4));
    d4 = dist2p(mleft, Pt(7));
    if (d2 * d3 < d1 * d4) is_long = true;
    for (int j = 0; j < 4; j++) {
        if ((IDc[j] >= crr) && (IDc[j] - crr < covL[j])) ID_right = is_long ? 7 - j : j;
        if ((IDc[j] < crr) && (crr - IDc[j] < covR[j])) ID_right = is_long ? 7 - j : j;
    }
    if (ctm::fabs32(l1[1] - l2[1]) > 0.05 * (l1[1] + l2[1])) {
        idl = -1;
        idr = -1;
        id = -2;
    } else {
        idl = ID_left;
        idr = ID_right;
        id = ID_left * 8 + ID_right;
    }
}

// WAVES: 1 for batches (a frame per wave, eight frames per CU); 8 for calls of a few frames: the pair predicate and the per-slot phases spread over
// all lanes, the union-find stays on wave 0 (its state lives in that wave's registers), and the markers are decoded a wave each -- code
// positions, dictionary coverage and the choice are independent per marker; only the places of the accepted markers in the record depend on the
// markers before them, and those are handed out in marker order once a round of WAVES markers is decoded.
#define WSYNC()                                                \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_markers(MarkerPtrs P, int nframes, int feature_size, int drows, int dcols) {
    constexpr int NT = 64 * WAVES;
    constexpr int SL = (CTAG_MAX_FEATURES + NT - 1) / NT;  // feature slots per lane in the per-slot phases
    // the features as K8 left them and, later, the records built from them share one region (records are built from registers):
    // with the coverage table in bytes the block needs 19.6 KB instead of 31.6 KB of LDS -- 8 frames per CU instead of 5
    __shared__ __attribute__((aligned(16))) unsigned char s_fr[CTAG_MAX_FEATURES * sizeof(ctag_feature_rec)];
    static_assert(sizeof(ctag_feature_rec) >= sizeof(FeatureDev), "records overlay the features");
    FeatureDev* s_feat = reinterpret_cast<FeatureDev*>(s_fr);
    ctag_feature_rec* s_rec = reinterpret_cast<ctag_feature_rec*>(s_fr);
    __shared__ int s_father[CTAG_MAX_FEATURES];
    __shared__ int s_group[CTAG_MAX_FEATURES];   // marker index of each feature
    __shared__ int s_order[CTAG_MAX_FEATURES];   // features grouped by marker, in marker order, sorted
    __shared__ int s_mfirst[CTAG_MAX_FEATURES + 1];
    __shared__ uint8_t s_cov_w[WAVES][2 * kMaxDictCells];  // coverage per (dir, row, col): at most CTAG_MAX_CODE_POS
    __shared__ int s_code_w[WAVES][CTAG_MAX_CODE_POS];
    __shared__ int s_misc_w[WAVES][8];
    __shared__ int s_cnt;
    __shared__ unsigned long long s_pair[CTAG_MAX_FEATURES][2];
    __shared__ int s_pos_w[WAVES][CTAG_MAX_CODE_POS];
    __shared__ uint8_t s_dict[kMaxDictCells];  // dictionary entries are 0..63 (checked at load): one byte each
    const int frame = blockIdx.x;
    if (frame >= nframes) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ctag_frame_result* out = P.out + frame;
    PhaseClock clk(P.stamps);
    // every byte of a result record is defined (unused slots are zero): records are compared / gathered as raw bytes
    {
        uint32_t* w = reinterpret_cast<uint32_t*>(out);
        for (int i = tid; i < (int)(sizeof(ctag_frame_result) / 4); i += NT) w[i] = 0u;
        if (P.pre) {
            uint32_t* wp = reinterpret_cast<uint32_t*>(P.pre + frame);
            for (int i = tid; i < (int)(sizeof(ctag_frame_result) / 4); i += NT) wp[i] = 0u;
        }
    }
    __syncthreads();
    const int status = P.status[frame];
    const uint32_t flags = P.frame_flags[frame];
    const int nf = P.nfeat[frame];
    if (status != CTAG_OK || nf == 0) {
        if (tid == 0) {
            int st = status;
            if ((flags & CTAG_FLAG_POOL_OVERFLOW) && !P.big) {
                // the batch workspace's pools were too small for this frame: not a result yet -- the library runs it again through the
                // workspace that holds any frame (finish_pending, ctag_api.hip); device-memory calls leave what that takes in a list
                st = CTAG_PENDING;
                if (P.pend.list) {
                    const int at = atomicAdd(P.pend.count, 1);
                    if (at < P.pend.cap) {
                        PendingRec r;
                        r.src = P.pend.src + (int64_t)frame * P.pend.frame_stride;
                        r.out = out;
                        r.row_stride = P.pend.row_stride;
                        r.rows = P.pend.rows;
                        r.cols = P.pend.cols;
                        r.ch = P.pend.ch;
                        r.tw = P.pend.tw;
                        r.subpix = P.pend.subpix;
                        r.dist = P.pend.dist;
                        P.pend.list[at] = r;
                    } else {
                        st = CTAG_ERR_LIMIT;  // the list is full (65 536 frames between two synchronisation points): a terminal status, the flag says why
                    }
                }
            }
            out->status = st;
            out->n_markers = 0;
            out->n_features = 0;
            out->flags = flags;
            if (P.pre) {
                P.pre[frame].status = st;
                P.pre[frame].n_markers = 0;
                P.pre[frame].n_features = 0;
                P.pre[frame].flags = flags;
            }
        }
        return;
    }
    for (int i = tid; i < nf; i += NT) s_feat[i] = P.feat[(size_t)frame * CTAG_MAX_FEATURES + i];
    for (int i = tid; i < drows * dcols; i += NT) s_dict[i] = (uint8_t)P.dict[i];
    __syncthreads();
    clk.mark(16);
    // ---- markerOrganization (:976-1052).  The O(F^2) pair predicate is evaluated for all pairs i < j at once, a pair per lane
    // (the triangle folded into a rectangle: rows r and nf-2-r together hold nf entries), into a bit matrix; the unions of
    // the true pairs are then replayed in the reference's (i, j) order, which is all the union-find state depends on.
    {
        uint32_t* pair32 = reinterpret_cast<uint32_t*>(&s_pair[0][0]);
        for (int w = tid; w < 4 * nf; w += NT) pair32[w] = 0u;
        __syncthreads();
        const int rows2 = nf / 2;  // ceil((nf - 1) / 2) folded rows
        for (int idx = tid; idx < rows2 * nf; idx += NT) {
            const int r = idx / nf, c = idx - r * nf;
            int i, j;
            if (c < nf - 1 - r) {
                i = r;
                j = r + 1 + c;
            } else {
                i = nf - 2 - r;
                j = i + 1 + (c - (nf - 1 - r));
                if (i == r) continue;  // the middle row pairs with itself: its entries are the first part
            }
            const FeatureDev &A = s_feat[i], &B = s_feat[j];
            const float threshold_angle = P.kp.angle, threshold_vertical = P.kp.vertical;
            const float vlx = A.c[0] - A.c[10], vly = A.c[1] - A.c[11];
            const double reach = 0.3 * dist2p(P2{A.c[0], A.c[1]}, P2{A.c[10], A.c[11]});
            const float vcx = A.center[0] - B.center[0], vcy = A.center[1] - B.center[1];
            const float center_angle = (vcx * vlx + vcy * vly) / ctm::sqrt32((vcx * vcx + vcy * vcy) * (vlx * vlx + vly * vly));
            const bool hit = (ctm::fabs32(A.angle - B.angle) < threshold_angle * 2 || ctm::fabs32(180 - ctm::fabs32(A.angle - B.angle)) < threshold_angle) &&
                             (dist2p(P2{A.center[0], A.center[1]}, P2{B.center[0], B.center[1]}) < reach) && (ctm::fabs32(center_angle) < threshold_vertical);
            if (hit) atomicOr(&pair32[j * 4 + (i >> 5)], 1u << (i & 31));  // column j of the matrix: the i < j paired with it
        }
    }
    __syncthreads();
    clk.mark(17);
    // The union-find runs on the whole wave in lockstep with its state in registers: father[k] lives in lane k & 63 (fa for k < 64,
    // fb above), read with v_readlane (the index is wave-uniform) instead of a chain of dependent LDS round trips.
    if (wave == 0) {
        int cnt;
        int fa = tid, fb = 64 + tid;
        auto fget = [&](int x) { return x < 64 ? __builtin_amdgcn_readlane(fa, x) : __builtin_amdgcn_readlane(fb, x - 64); };
        auto fset = [&](int x, int v) {
            if (tid == (x & 63)) {
                if (x < 64) fa = v;
                else fb = v;
            }
        };
        auto uf = [&](int x) {
            int r = x;
            for (int f = fget(r); f != r; f = fget(r)) r = f;
            for (int nx = fget(x); nx != r; nx = fget(x)) {  // path compression as the recursive union_find does
                fset(x, r);
                x = nx;
            }
            return r;
        };
        // column j of the bit matrix sits in lane j & 63 (columns 64.. in the second pair of registers), so row i -- the j paired
        // with i -- is a ballot over "my column has bit i", with no cross-lane read at all
        const unsigned long long c0a = tid < nf ? s_pair[tid][0] : 0ull, c0b = tid < nf ? s_pair[tid][1] : 0ull;
        const unsigned long long c1a = tid + 64 < nf ? s_pair[tid + 64][0] : 0ull, c1b = tid + 64 < nf ? s_pair[tid + 64][1] : 0ull;
        for (int i = 0; i < nf - 1; i++) {
            unsigned long long m0, m1;
            if (i < 64) {
                m0 = __ballot((c0a >> i) & 1ull);
                m1 = __ballot((c1a >> i) & 1ull);
            } else {
                m0 = __ballot((c0b >> (i - 64)) & 1ull);
                m1 = __ballot((c1b >> (i - 64)) & 1ull);
            }
            if (!(m0 | m1)) continue;
            // Most true pairs join features that already hang directly under the same root: for those the reference's two
            // union_find calls write nothing.  With r = father[i] a root, every j whose father is r is such a pair, now and for the
            // rest of row i (a union only hangs another root under r), so one ballot removes them from the row.
            {
                const int ri = fget(i);
                const unsigned long long roots = ri < 64 ? __ballot(fa == tid) >> ri : __ballot(fb == tid + 64) >> (ri - 64);
                if (roots & 1ull) {
                    m0 &= ~__ballot(fa == ri);
                    m1 &= ~__ballot(fb == ri);
                }
            }
            for (int w = 0; w < 2; w++) {
                unsigned long long m = w ? m1 : m0;
                while (m) {
                    const int j = w * 64 + __builtin_ctzll(m);
                    m &= m - 1;
                    const int fi = uf(i), fj = uf(j);
                    if (fi != fj) fset(fj, fi);
                }
            }
        }
        // groups in first-seen order (:993-1019).  father[0] is captured BEFORE the flattening loop, exactly as the reference
        // pushes it (it may be a stale non-root: literal quirk).  The flattening father[i] = union_find(father[i]), i >= 1, leaves
        // every such entry at the root of its set whatever the order, so it is done by pointer jumping on all lanes; the groups
        // are then numbered a set at a time, in the order of their first member i >= 1 (father_database[0] = the captured value).
        const int db0 = fget(0);
        for (;;) {
            const int ga2 = __shfl(fa, fa & 63), gb2 = __shfl(fb, fa & 63), ha2 = __shfl(fa, fb & 63), hb2 = __shfl(fb, fb & 63);
            const int na = fa < 64 ? ga2 : gb2, nb = fb < 64 ? ha2 : hb2;  // father[father[k]]
            const bool moved = (tid >= 1 && na != fa) || nb != fb;
            if (tid >= 1) fa = na;  // entry 0 is not flattened by the reference (and not read below)
            fb = nb;
            if (!__ballot(moved)) break;
        }
        int ga = 0, gb = 0;  // marker index of feature tid / tid + 64
        cnt = 1;
        {
            unsigned long long ra = __ballot(tid >= 1 && tid < nf), rb = __ballot(tid + 64 < nf);  // features not numbered yet
            while (ra | rb) {
                const int lead = ra ? __builtin_ctzll(ra) : 64 + __builtin_ctzll(rb);
                const int r = fget(lead);
                const int g = r == db0 ? 0 : cnt++;
                const unsigned long long ma = __ballot(fa == r) & ra, mb = __ballot(fb == r) & rb;
                if ((ma >> tid) & 1ull) ga = g;
                if ((mb >> tid) & 1ull) gb = g;
                ra &= ~ma;
                rb &= ~mb;
            }
        }
        // features of each marker in ascending feature index (the order marker_ID[j] is filled)
        const unsigned long long lt = tid ? (~0ull >> (64 - tid)) : 0ull;
        int pos = 0;
        for (int m = 0; m < cnt; m++) {
            const unsigned long long ba = __ballot(tid < nf && ga == m), bb = __ballot(tid + 64 < nf && gb == m);
            if (tid == 0) s_mfirst[m] = pos;
            if (tid < nf && ga == m) {
                const int slot = pos + __popcll(ba & lt);
                s_order[slot] = tid;
                s_father[slot] = m;  // s_father: marker owning slot k
            }
            if (tid + 64 < nf && gb == m) {
                const int slot = pos + __popcll(ba) + __popcll(bb & lt);
                s_order[slot] = tid + 64;
                s_father[slot] = m;
            }
            pos += __popcll(ba) + __popcll(bb);
        }
        if (tid == 0) {
            s_mfirst[cnt] = pos;
            s_cnt = cnt;
        }
    }
    __syncthreads();
    const int cnt = s_cnt;
    clk.mark(18);
    // ---- per marker: orientation (the angles of its features are summed in order, :1021-1031) and the stable sort of its
    // features: angles and ranks are computed a feature slot per lane, only the sum is a lane per marker
    double* s_ang = reinterpret_cast<double*>(&s_pair[0][0]);  // the bit matrix is dead: folded direction of the feature in slot k
    for (int k = tid; k < nf; k += NT) {
        const FeatureDev& F = s_feat[s_order[k]];
        double angle_now = ctm::fast_atan2_deg(F.c[1] - F.c[11], F.c[0] - F.c[10]);
        if (angle_now > 180) angle_now -= 180;
        s_ang[k] = angle_now;
    }
    __syncthreads();
    for (int m = tid; m < cnt; m += NT) {
        const int a = s_mfirst[m], b = s_mfirst[m + 1], n = b - a;
        float marker_angle = 0;
        for (int k = a; k < b; k++) marker_angle = (float)(marker_angle + s_ang[k]);
        marker_angle /= (float)n;
        s_group[m] = (ctm::fabs32(marker_angle) < 45 || ctm::fabs32(marker_angle) > 135) ? 0 : 1;
    }
    __syncthreads();
    {
        // rank of slot k among its marker's slots: std::sort on <= 16 elements is an insertion sort, i.e. stable
        int newv[SL], newk[SL], dirk[SL];
#pragma unroll
        for (int q = 0; q < SL; q++) {
            newv[q] = 0, newk[q] = -1, dirk[q] = 0;
            const int k = tid + NT * q;
            if (k >= nf) continue;
            const int m = s_father[k], a = s_mfirst[m], b = s_mfirst[m + 1], direc = s_group[m];
            const int v = s_order[k];
            const float kv = direc == 0 ? s_feat[v].center[1] : s_feat[v].center[0];
            int rank = 0;
            for (int x = a; x < b; x++) {
                const int vx = s_order[x];
                const float kx = direc == 0 ? s_feat[vx].center[1] : s_feat[vx].center[0];
                const bool x_before = direc == 0 ? (kx > kv) : (kx < kv);
                const bool k_before = direc == 0 ? (kv > kx) : (kv < kx);
                rank += (x_before || (!k_before && x < k)) ? 1 : 0;
            }
            newv[q] = v;
            newk[q] = a + rank;
            dirk[q] = direc;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SL; q++) {
            if (newk[q] < 0) continue;
            s_order[newk[q]] = newv[q];
        }
        __syncthreads();
        // s_group becomes: direction of the marker owning slot k (s_father keeps the marker of slot k; a marker's slots do not move)
#pragma unroll
        for (int q = 0; q < SL; q++) {
            const int k = tid + NT * q;
            if (k < nf) s_group[k] = dirk[q];
        }
    }
    __syncthreads();
    clk.mark(19);
    // ---- featureExtraction (lane per feature slot).  ID_left / ID_right persist from one feature to the next when no
    // cross-ratio band matches, so each lane reports "matched value or carry" and thread 0 replays the carry.
    constexpr int kCarry = -99;
    static_assert(CTAG_MAX_FEATURES <= 128, "two feature slots per lane of a one-wave block");
    FeatureDev Fk[SL];
#pragma unroll
    for (int q = 0; q < SL; q++) {
        const int k = tid + NT * q;
        if (k < nf) Fk[q] = s_feat[s_order[k]];
    }
    __syncthreads();  // every feature is in registers: the region becomes the records
#pragma unroll
    for (int q = 0; q < SL; q++) {
        const int k = tid + NT * q;
        if (k >= nf) continue;
        const FeatureDev& F = Fk[q];
        ctag_feature_rec& R = s_rec[k];
        for (int c = 0; c < 16; c++) R.corners[c] = F.c[c];
        R.center[0] = F.center[0];
        R.center[1] = F.center[1];
        R.edge_length = (dist2p(P2{F.c[0], F.c[1]}, P2{F.c[2], F.c[3]}) + dist2p(P2{F.c[8], F.c[9]}, P2{F.c[10], F.c[11]}) / 2);
        R.pos = -1;
        int id, idl, idr, nl = kCarry, nr = kCarry;
        feature_ids(R.corners, s_group[k], nl, nr, R.cr_left, R.cr_right, id, idl, idr, P.kp.cr_id, P.kp.cr_lo, P.kp.cr_hi);
        R.id = id == -2 ? -2 : 0;  // -2: rejected by the edge-length test; 0: ids filled in by the replay
        R.id_left = nl;
        R.id_right = nr;
    }
    __syncthreads();
    clk.mark(20);
    if (tid == 0) {
        int ID_left = 0, ID_right = 0;  // reset per detect()
        for (int k = 0; k < nf; k++) {
            ctag_feature_rec& R = s_rec[k];
            if (R.id_left != kCarry) ID_left = R.id_left;
            if (R.id_right != kCarry) ID_right = R.id_right;
            if (R.id == -2) {
                R.id_left = -1;
                R.id_right = -1;
            } else {
                R.id_left = ID_left;
                R.id_right = ID_right;
                R.id = ID_left * 8 + ID_right;
            }
        }
    }
    __syncthreads();
    if (P.pre) {  // debug copy of the markers before decoding
        ctag_frame_result* pre = P.pre + frame;
        if (tid == 0) {
            pre->status = status;
            pre->n_markers = cnt;
            pre->n_features = nf;
            pre->flags = flags;
        }
        for (int m = tid; m < cnt; m += NT) {
            pre->markers[m].marker_id = -1;
            pre->markers[m].first_feature = s_mfirst[m];
            pre->markers[m].n_features = s_mfirst[m + 1] - s_mfirst[m];
            pre->markers[m].n_pos = 0;
        }
        for (int k = tid; k < nf; k += NT) pre->features[k] = s_rec[k];
    }
    clk.mark(21);
    // ---- markerDecoder (:1211-1250) + match_dictionary (:1269-1324)
    int out_markers = 0, out_features = 0;
    uint32_t oflags = flags;
    for (int m0 = 0; m0 < cnt; m0 += WAVES) {
        __syncthreads();  // the records are complete / the previous round's markers are stored
        int* const s_code = s_code_w[wave];
        uint8_t* const s_cov = s_cov_w[wave];
        int* const s_misc = s_misc_w[wave];
        int* const s_pos = s_pos_w[wave];
        const int m = m0 + wave;
        const int a = m < cnt ? s_mfirst[m] : 0, b = m < cnt ? s_mfirst[m + 1] : 0, n = b - a;
        if (lane == 0) {
            s_misc[3] = 0;  // no code overflow
            s_misc[5] = 0;  // not accepted
        }
        WSYNC();
        if (m < cnt && n >= feature_size) do {  // (wave-uniform)
        if (n <= 64) {
            // code positions (:1218-1227) a feature per lane: gap_j from features j-1 and j, position = running sum of the gaps;
            // the sequential loop stops at the first bad gap or position, which any lane reports; of several features on one
            // position the last one stays
            const int j = lane;
            int gap = 0, bad = 0, idj = -1;
            if (j < n) {
                const ctag_feature_rec& Rj = s_rec[a + j];
                idj = Rj.id;
                if (j >= 1) {
                    const ctag_feature_rec& Rp = s_rec[a + j - 1];
                    const float dist_fea = dist2p(P2{Rj.center[0], Rj.center[1]}, P2{Rp.center[0], Rp.center[1]});
                    const float gap_f = ctm::round32(dist_fea / ((Rj.edge_length + Rp.edge_length) * 3 / 4));
                    if (!(gap_f >= 0.f && gap_f < (float)CTAG_MAX_CODE_POS)) bad = 1;
                    else gap = (int)gap_f;
                }
            }
            int pos = gap;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(pos, d);
                if (lane >= d) pos += o;
            }
            if (j < n && pos >= CTAG_MAX_CODE_POS) bad = 1;
            const int overflow = __ballot(bad != 0) != 0ull;
            if (lane < CTAG_MAX_CODE_POS) s_code[lane] = -1;
            WSYNC();
            const int pos_next = __shfl_down(pos, 1);
            if (!overflow && j < n && (j == n - 1 || pos_next != pos)) s_code[pos] = idj;
            WSYNC();
            const unsigned long long lg = __ballot(lane < CTAG_MAX_CODE_POS && s_code[lane < CTAG_MAX_CODE_POS ? lane : 0] > -1);
            const int pos_last = __shfl(pos, n - 1);
            if (lane == 0) {
                s_misc[2] = pos_last;
                s_misc[3] = overflow;
                s_misc[4] = __popcll(lg);
            }
        } else if (lane == 0) {
            for (int k = 0; k < CTAG_MAX_CODE_POS; k++) s_code[k] = -1;
            int pos_now = 0, overflow = 0;
            s_code[0] = s_rec[a].id;
            for (int j = 1; j < n; j++) {
                const ctag_feature_rec &Rj = s_rec[a + j], &Rp = s_rec[a + j - 1];
                const float dist_fea = dist2p(P2{Rj.center[0], Rj.center[1]}, P2{Rp.center[0], Rp.center[1]});
                const float gap_f = ctm::round32(dist_fea / ((Rj.edge_length + Rp.edge_length) * 3 / 4));
                if (!(gap_f >= 0.f && gap_f < (float)CTAG_MAX_CODE_POS)) {
                    overflow = 1;
                    break;
                }
                pos_now += (int)gap_f;
                if (pos_now >= CTAG_MAX_CODE_POS || pos_now < 0) {
                    overflow = 1;
                    break;
                }
                s_code[pos_now] = Rj.id;
            }
            int legal = 0;
            for (int k = 0; k < CTAG_MAX_CODE_POS; k++)
                if (s_code[k] > -1) legal++;
            s_misc[2] = pos_now;
            s_misc[3] = overflow;
            s_misc[4] = legal;
        }
        WSYNC();
        clk.mark(22);
        if (s_misc[3]) break;  // CTAG_FLAG_CODE_OVERFLOW (set where the round's markers are stored)
        const int length = s_misc[2], legal = s_misc[4];
        // coverage of every hypothesis (dir, row, col), stored in the reference's scan order
        const int hyp = drows * dcols;
        const int per = (2 * hyp + 63) / 64;
        const int h_lo = min(lane * per, 2 * hyp), h_hi = min(h_lo + per, 2 * hyp);
        int lane_max = -1;
        if (P.dict_pos) {
            // Bit-parallel form, a dictionary row per lane: dict_pos[row][v] is the set of columns holding symbol v (built once per
            // handle), so position k of the code contributes the column set of its symbol rotated by k -- bit j of the rotated
            // set says "hypothesis (row, j) matches at k" -- and the matches of all columns are counted at once in five
            // bit planes.  Reversed direction: the inverted symbol, rotation the other way, and a column that the reference's
            // (j - k + dcols) leaves negative (k > j + dcols) never matches.
            const uint32_t dmask = dcols >= 32 ? 0xffffffffu : ((1u << dcols) - 1u);
            for (int i = lane; i < drows; i += 64) {
                const uint32_t* Ti = P.dict_pos + (size_t)i * 64;
                uint32_t f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0;
                int sh = 0;  // k % dcols
#pragma unroll
                for (int k = 0; k < CTAG_MAX_CODE_POS; k++) {
                    if (k <= length) {  // uniform
                        const int cd = s_code[k];
                        const bool have = cd >= 0;  // -1: no feature at this position; matches nothing in either direction
                        const uint32_t mf = have ? Ti[cd & 63] : 0u;
                        const uint32_t mb = have ? Ti[((7 - cd / 8) + (7 - cd % 8) * 8) & 63] : 0u;
                        uint32_t x = sh ? (((mf >> sh) | (mf << (dcols - sh))) & dmask) : mf;
                        uint32_t y = sh ? (((mb << sh) | (mb >> (dcols - sh))) & dmask) : mb;
                        if (k > dcols) y &= ~((1u << (k - dcols)) - 1u);
                        uint32_t c;
                        c = f0 & x, f0 ^= x, x = c;
                        c = f1 & x, f1 ^= x, x = c;
                        c = f2 & x, f2 ^= x, x = c;
                        c = f3 & x, f3 ^= x, x = c;
                        f4 ^= x;
                        c = b0 & y, b0 ^= y, y = c;
                        c = b1 & y, b1 ^= y, y = c;
                        c = b2 & y, b2 ^= y, y = c;
                        c = b3 & y, b3 ^= y, y = c;
                        b4 ^= y;
                    }
                    if (++sh == dcols) sh = 0;
                }
                for (int j = 0; j < dcols; j++) {
                    s_cov[i * dcols + j] = (uint8_t)(((f0 >> j) & 1u) | (((f1 >> j) & 1u) << 1) | (((f2 >> j) & 1u) << 2) | (((f3 >> j) & 1u) << 3) | (((f4 >> j) & 1u) << 4));
                    s_cov[hyp + i * dcols + j] =
                        (uint8_t)(((b0 >> j) & 1u) | (((b1 >> j) & 1u) << 1) | (((b2 >> j) & 1u) << 2) | (((b3 >> j) & 1u) << 3) | (((b4 >> j) & 1u) << 4));
                }
            }
            WSYNC();
            for (int h = h_lo; h < h_hi; h++) lane_max = max(lane_max, (int)s_cov[h]);
        } else {
            // more than 32 dictionary columns: a contiguous run of hypotheses per lane, the code and its reversed+inverted form in
            // registers, the CTAG_MAX_CODE_POS dictionary bytes of a hypothesis requested together
            {
                int codef[CTAG_MAX_CODE_POS], codeb[CTAG_MAX_CODE_POS];
    #pragma unroll
                for (int k = 0; k < CTAG_MAX_CODE_POS; k++) {
                    const int cd = s_code[k];
                    codef[k] = k <= length ? cd : -2;                                  // -2 / -3: never equal to a dictionary byte
                    codeb[k] = k <= length ? ((7 - cd / 8) + (7 - cd % 8) * 8) : -3;  // a -1 gives 71: no match either
                }
                int dir = h_lo >= hyp ? 1 : 0;
                const int rc0 = h_lo - dir * hyp;
                int i = rc0 / dcols, j = rc0 - i * dcols;  // one division per lane; (dir, i, j) advance incrementally
                for (int h = h_lo; h < h_hi; h++) {
                    const uint8_t* row = s_dict + i * dcols;
                    int cov = 0;
                    int cf = j;  // forward: column (j + k) % dcols
    #pragma unroll
                    for (int k = 0; k < CTAG_MAX_CODE_POS; k++) {
                        int cb = j - k + dcols;  // reversed: (j - k + dcols) % dcols with C semantics; a negative column never matches
                        if (cb >= dcols) cb -= dcols;
                        const int c = dir ? cb : cf;
                        const int want = dir ? codeb[k] : codef[k];
                        const int got = (int)row[max(c, 0)];  // unconditional load: the 20 bytes of a hypothesis are requested together
                        cov += (int)(c >= 0) & (int)(got == want);
                        if (++cf == dcols) cf = 0;
                    }
                    s_cov[h] = (uint8_t)cov;
                    lane_max = max(lane_max, cov);
                    if (++j == dcols) {
                        j = 0;
                        if (++i == drows) {
                            i = 0;
                            dir = 1;
                        }
                    }
                }
            }
        }
        clk.mark(23);
        // order-dependent max / second bookkeeping (:1280-1311): an element updates `second` iff it does not raise the
        // running maximum, so every lane replays its run from the exclusive prefix maximum of the lanes before it
        int run = lane_max;
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(run, d);
            if (lane >= d) run = max(run, o);
        }
        const int gmax = __shfl(run, 63);
        run = __shfl_up(run, 1);
        if (lane == 0) run = -1;
        int sec = -1, first_max = 0x7fffffff;
        for (int h = h_lo; h < h_hi; h++) {
            const int cov = s_cov[h];
            if (cov > run) {
                run = cov;
                if (cov == gmax) first_max = h;
            } else if (cov > sec) {
                sec = cov;
            }
        }
        for (int d = 32; d; d >>= 1) {
            sec = max(sec, __shfl_xor(sec, d));
            first_max = min(first_max, __shfl_xor(first_max, d));
        }
        if (lane == 0) {
            const int max_cov = gmax, second = sec;
            const int dir = first_max >= hyp, rc = first_max - dir * hyp;
            const int mx = rc / dcols, my = rc - mx * dcols;
            const int direc = dir == 0 ? 1 : -1;
            const double lim = 0.8 * legal < legal - 1.0 ? 0.8 * legal : legal - 1.0;
            const int good = (max_cov >= lim && max_cov > second) ? 1 : 0;
            s_misc[5] = good;
            s_misc[6] = direc;
            s_misc[1] = mx;
            if (good) {
                int np = 0;
                for (int i = 0; i <= length; i++) {
                    if (s_code[i] != -1) {
                        s_pos[np] = (my + direc * i + dcols) % dcols;
                        np++;
                    }
                }
                s_misc[7] = np;
            }
        }
        } while (0);
        clk.mark(23);
        __syncthreads();
        // the round's markers take their places in the record, in marker order
        for (int w = 0; w < WAVES && m0 + w < cnt; w++) {
            const int* const r_misc = s_misc_w[w];
            if (r_misc[3]) oflags |= CTAG_FLAG_CODE_OVERFLOW;
            if (!r_misc[5]) continue;
            const int ra = s_mfirst[m0 + w], rn = s_mfirst[m0 + w + 1] - ra;
            const int direc = r_misc[6], np = r_misc[7];
            if (tid == 0) {
                ctag_marker_rec& M = out->markers[out_markers];
                M.marker_id = r_misc[1];
                M.first_feature = out_features;
                M.n_features = rn;
                M.n_pos = np;
            }
            for (int k = tid; k < rn; k += NT) {
                ctag_feature_rec R = s_rec[ra + k];
                if (direc == -1) {
                    for (int q = 0; q < 8; q++) {
                        const float t = R.corners[q];
                        R.corners[q] = R.corners[8 + q];
                        R.corners[8 + q] = t;
                    }
                }
                R.pos = k < np ? s_pos_w[w][k] : -1;
                out->features[out_features + k] = R;
            }
            out_markers++;
            out_features += rn;
        }
        clk.mark(24);
    }
    if (tid == 0) {
        out->status = CTAG_OK;
        out->n_markers = out_markers;
        out->n_features = out_features;
        out->flags = oflags;
    }
}

// =====================================================================================================
// launchers
// =====================================================================================================
// developer aid (CTAG_FEAT_STAMPS=1): phase clocks of k_features (slots 0-5) and k_markers (16-24); `report` waits and prints
static unsigned long long* feat_stamps(hipStream_t s, bool report) {
    static const bool want = getenv("CTAG_FEAT_STAMPS") != nullptr;
    static unsigned long long* d = nullptr;
    if (!want) return nullptr;
    if (!d) {
        (void)hipMalloc(reinterpret_cast<void**>(&d), 32 * 8);
        (void)hipMemset(d, 0, 32 * 8);
    }
    if (report) {
        unsigned long long h[32];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipMemset(d, 0, 32 * 8);
        fprintf(stderr, "[k_features ticks] compact %llu derive %llu pairs %llu greedy %llu organise %llu obtain %llu\n", h[0], h[1], h[2], h[3], h[4], h[5]);
        fprintf(stderr, "[k_markers ticks] load %llu pairs %llu union %llu sort %llu ids %llu carry+pre %llu | code %llu coverage %llu pick+store %llu\n", h[16], h[17], h[18],
                h[19], h[20], h[21], h[22], h[23], h[24]);
    }
    return d;
}

hipError_t launch_features(int nframes, const Workspace& ws, const DetectParams& p, hipStream_t s) {
    FeatPtrs P{ws.ncand, ws.quads, reinterpret_cast<QuadDerived*>(ws.quad_derived), ws.quad_index, ws.nquads, ws.nfeat, ws.status, ws.frame_flags, ws.feat0, ws.feat1, ws.feat2, feat_stamps(s, false), ws.kp.angle, ws.frame_long, ws.cand_cap};
    if (nframes <= kLatencyFrames) hipLaunchKernelGGL(k_features<512>, dim3(nframes), dim3(512), 0, s, P, nframes, p.feature_size);
    else hipLaunchKernelGGL(k_features<128>, dim3(nframes), dim3(128), 0, s, P, nframes, p.feature_size);
    return hipGetLastError();
}
hipError_t launch_edge_refine(const uint8_t* frames, ptrdiff_t frame_stride, ptrdiff_t row_stride, int nframes, const Workspace& ws, const DetectParams& p, hipStream_t s) {
    if (!p.corner_subpix) return hipSuccess;
    RefinePtrs P{frames, frame_stride, row_stride, ws.nfeat, ws.status, ws.feat1, ws.feat2, ws.refine_n0, ws.frame_long, p.channels == 3 ? 3 : 1};
    static const int refine_gx = getenv("CTAG_REFINE_GX") ? atoi(getenv("CTAG_REFINE_GX")) : 32;  // looping blocks per frame of the other forms
    const bool few = nframes <= kLatencyFrames || !CTAG_REFINE_SPLIT;
    const dim3 grid(few ? CTAG_MAX_FEATURES * 2 : refine_gx, nframes);  // a few frames: a block per quad -- the call is as long as its longest block, and looping blocks triple it
    static const int refine_sums_gx = getenv("CTAG_REFINE_SUMS_GX") ? atoi(getenv("CTAG_REFINE_SUMS_GX")) : 12;  // looping blocks per frame of k_edge_refine<2>
    if (few) {  // one kernel, one launch
        hipLaunchKernelGGL(k_edge_refine<0>, grid, dim3(kRefineThreads), 0, s, P, ws.g.rows, ws.g.cols, p.subpix_dist, nframes, 0);
    } else {
        static const int xcd = getenv("CTAG_REFINE_XCD") ? atoi(getenv("CTAG_REFINE_XCD")) : 3;
        const int f8 = ((nframes + 7) / 8) * 8;
        // (round 5: searches and sums alternating over slices of 256 / 512 / 1024 frames, so that a slice's n0 is read back while the memory-side cache
        // still holds it, lost -- 5.13 / 4.95 / 4.84 against 4.71-4.76 ms per 4096 frames: the round trip through HBM is not what the sums kernel waits for)
        static const int large_env = getenv("CTAG_REFINE_LARGE") ? atoi(getenv("CTAG_REFINE_LARGE")) : 1;  // 0: the small staging region for every frame size (A/B)
        const bool large = large_env && (long long)ws.g.rows * ws.g.cols > 1920LL * 1200;
        if (large) hipLaunchKernelGGL((k_edge_refine<1, kRefineRegionLarge>), dim3(f8 * refine_gx), dim3(kRefineThreads), 0, s, P, ws.g.rows, ws.g.cols, p.subpix_dist, nframes, refine_gx);
        else if (xcd & 1) hipLaunchKernelGGL(k_edge_refine<1>, dim3(f8 * refine_gx), dim3(kRefineThreads), 0, s, P, ws.g.rows, ws.g.cols, p.subpix_dist, nframes, refine_gx);
        else hipLaunchKernelGGL(k_edge_refine<1>, grid, dim3(kRefineThreads), 0, s, P, ws.g.rows, ws.g.cols, p.subpix_dist, nframes, 0);
        hipLaunchKernelGGL(k_edge_refine_sums, dim3(f8 * refine_sums_gx), dim3(64), 0, s, P, nframes, refine_sums_gx);
        hipLaunchKernelGGL(k_edge_refine_tail, dim3(13, nframes), dim3(64), 0, s, P, nframes, 13);  // 13 x 8 quads: the synthetic frames' 96; a block loops when a frame has more
        hipLaunchKernelGGL(k_edge_refine_long, dim3(4, nframes), dim3(kRefineThreads), 0, s, P, ws.g.rows, ws.g.cols, p.subpix_dist, nframes);
    }
    return hipGetLastError();
}
hipError_t launch_markers(int nframes, const Workspace& ws, const DetectParams& p, ctag_frame_result* out, const PendingCtx& pend, hipStream_t s) {
    MarkerPtrs P{ws.nfeat, ws.status, ws.frame_flags, ws.feat2, p.dict, ws.premarkers, out, feat_stamps(s, false), p.dict_pos, ws.kp, pend, ws.big ? 1 : 0};
    if (nframes <= kLatencyFrames) hipLaunchKernelGGL(k_markers<8>, dim3(nframes), dim3(512), 0, s, P, nframes, p.feature_size, p.dict_rows, p.dict_cols);
    else hipLaunchKernelGGL(k_markers<1>, dim3(nframes), dim3(64), 0, s, P, nframes, p.feature_size, p.dict_rows, p.dict_cols);
    (void)feat_stamps(s, true);
    return hipGetLastError();
}

}  // namespace ctag
