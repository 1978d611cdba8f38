// ctag_testkit.hip -- implementation of include/ctag_testkit.h (libctag_testkit.so): test and bench scaffolding that lives
// OUTSIDE the product library.  Links against libctag_hip.so; reaches into a handle only through ctag::handle_view and
// ctag::gather_unpack_gathered (cylindertag_amd/csrc/ctag_internal.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "../include/ctag_testkit.h"
#include "../cylindertag_amd/csrc/ctag_internal.h"
#include "../cylindertag_amd/csrc/ctag_math.h"
#include "ctag_synth.h"

using namespace ctag;

#define TK_TRY(expr)                              \
    do {                                          \
        if ((expr) != hipSuccess) return CTAG_ERR_HIP; \
    } while (0)

namespace {

__global__ void k_math_probe(int op, int n, const double* a, const double* b, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b[i];
    double r = 0;
    switch (op) {
        case 0: r = ctm::atan2_64(x, y); break;
        case 1: r = ctm::sin64(x); break;
        case 2: r = ctm::cos64(x); break;
        case 3: r = ctm::exp64(x); break;
        case 4: r = ctm::acos64(x); break;
        case 5: r = ctm::atan2_32((float)x, (float)y); break;
        case 6: r = ctm::sin32((float)x); break;
        case 7: r = ctm::cos32((float)x); break;
        case 8: r = ctm::exp32((float)x); break;
        case 9: r = ctm::fast_atan2_deg((float)x, (float)y); break;
        case 10: r = x / y; break;
        case 11: r = ctm::sqrt64(x); break;
        case 12: r = (float)x / (float)y; break;
        case 13: r = ctm::sqrt32((float)x); break;
        case 14: r = ctm::round32((float)x); break;
        case 15: r = ctm::exp32_nonpos((float)x); break;  // x <= 0, not NaN
        case 16: r = ctm::div64(x, ctm::recip64(y)); break;  // x / y through the shared-reciprocal form (device) or `/` (host)
        default: break;
    }
    out[i] = r;
}

__global__ void k_label_roots(const uint16_t* labels, const int32_t* tile_base, const int32_t* root_of, int32_t* out, FrameGeom g) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= g.hcols || y >= g.hrows) return;
    const unsigned l = labels[(size_t)y * g.lp + x];
    int v = 0;
    const int tile = (y / kTileH) * g.tiles_x + (x / kTileW);
    if (l & 0x8000u) v = -(1 + tile * 32768 + (int)(l & 0x7fffu));  // an unpublished speck of the second CCL pass: a private negative id
    else if (l) v = 1 + root_of[tile_base[tile] + (int)l - 1];
    out[(size_t)y * g.hcols + x] = v;
}

__global__ __launch_bounds__(256) void k_synth(const ctag_synth::Frame* frames, uint8_t* out, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride) {
    __shared__ ctag_synth::Frame F;
    const int f = blockIdx.z;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(frames + f);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&F);
        for (int i = threadIdx.x; i < (int)(sizeof(ctag_synth::Frame) / 4); i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= cols || y >= rows) return;
    out[(ptrdiff_t)f * frame_stride + (ptrdiff_t)y * row_stride + x] = ctag_synth::pixel(F, x, y, rows, cols);
}

// renders frames [first_frame, first_frame + n) of a synthetic scene into device memory; layouts are computed on the host
int synth_frames_device(ctag_handle* h, uint8_t* frames_dev, int first_frame, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                        uint64_t seed, int markers_per_frame, bool scene3d, double fx, double fy, double cx, double cy) {
    if (!h || !frames_dev || n < 0 || rows < 1 || cols < 1 || row_stride < cols) return CTAG_ERR_ARG;
    if (n == 0) return CTAG_OK;
    HandleView v{};
    handle_view(h, &v);
    TK_TRY(hipSetDevice(v.device));
    hipStream_t s = static_cast<hipStream_t>(ctag_stream(h));
    const int step = 512;
    ctag_synth::Frame* d_lay = nullptr;
    TK_TRY(hipMalloc(reinterpret_cast<void**>(&d_lay), sizeof(ctag_synth::Frame) * step));
    std::vector<ctag_synth::Frame> lay(step);
    int rc = CTAG_OK;
    for (int f0 = 0; f0 < n && rc == CTAG_OK; f0 += step) {
        const int m = std::min(step, n - f0);
        for (int i = 0; i < m; i++) {
            if (scene3d)
                ctag_synth::layout3d(v.dict, v.dict_rows, v.dict_cols, seed, first_frame + f0 + i, rows, cols, markers_per_frame, fx, fy, cx, cy, &lay[i], nullptr);
            else
                ctag_synth::layout(v.dict, v.dict_rows, v.dict_cols, seed, first_frame + f0 + i, rows, cols, markers_per_frame, &lay[i], nullptr);
        }
        if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(d_lay, lay.data(), sizeof(ctag_synth::Frame) * m, hipMemcpyHostToDevice) != hipSuccess) {
            rc = CTAG_ERR_HIP;
            break;
        }
        hipLaunchKernelGGL(k_synth, dim3((cols + 255) / 256, rows, m), dim3(256), 0, s, d_lay, frames_dev + (ptrdiff_t)f0 * frame_stride, rows, cols, row_stride,
                           frame_stride);
        if (hipGetLastError() != hipSuccess) rc = CTAG_ERR_HIP;
    }
    if (hipStreamSynchronize(s) != hipSuccess) rc = CTAG_ERR_HIP;
    (void)hipFree(d_lay);
    return rc;
}

void fill_truth(const ctag_synth::Truth& T, ctag_synth_truth* truth) {
    std::memset(truth, 0, sizeof(*truth));
    truth->n_markers = T.n;
    for (int k = 0; k < T.n && k < 8; k++) {
        truth->dict_row[k] = T.dict_row[k];
        truth->strip_len[k] = T.strip_len[k];
        for (int q = 0; q < 8; q++) truth->corners[k][q] = T.corners[k][q];
    }
}

}  // namespace

extern "C" {

long ctag_debug_fetch(ctag_handle* h, int frame, int what, void* dst, size_t cap) {
    if (!h) return -1;
    HandleView v{};
    handle_view(h, &v);
    if (!v.ws || !v.ws->base || frame < 0 || frame >= v.last_chunk_frames) return -1;
    hipStream_t s = static_cast<hipStream_t>(ctag_stream(h));
    if (hipSetDevice(v.device) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -2;
    const Workspace& W = *v.ws;
    const FrameGeom& g = W.g;
    auto d2h = [&](void* d, const void* src, size_t bytes) { return hipMemcpy(d, src, bytes, hipMemcpyDeviceToHost) == hipSuccess; };
    switch (what) {
        case CTAG_DBG_HALF: {
            const size_t n = (size_t)g.hrows * g.hcols;
            if (dst && cap >= n) {
                if (v.fused) {
                    // the fused sweep never writes the half-size image (its place holds the threshold mask): decimate this frame again with
                    // the stand-alone kernel into a scratch image (the fused kernel's own pixels are checked through CTAG_DBG_MASK / labels)
                    if (!v.frames) return -1;
                    Workspace T = W;
                    uint8_t* tmp = nullptr;
                    if (hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)g.hrows * g.hp + 256) != hipSuccess) return -2;
                    T.half = tmp;
                    bool ok = launch_decimate(v.frames + (ptrdiff_t)frame * v.frame_stride, v.frame_stride, v.row_stride, 1, T, s, false) == hipSuccess &&
                              hipStreamSynchronize(s) == hipSuccess;
                    ok = ok && hipMemcpy2D(dst, g.hcols, tmp, g.hp, g.hcols, g.hrows, hipMemcpyDeviceToHost) == hipSuccess;
                    (void)hipFree(tmp);
                    if (!ok) return -2;
                } else if (hipMemcpy2D(dst, g.hcols, W.half + (size_t)frame * g.hrows * g.hp, g.hp, g.hcols, g.hrows, hipMemcpyDeviceToHost) != hipSuccess) {
                    return -2;
                }
            }
            return (long)n;
        }
        case CTAG_DBG_MASK: {
            if (!v.fused) return -1;
            const size_t n = (size_t)g.hrows * g.hcols;
            if (dst && cap >= n) {
                const size_t mb = (size_t)g.hrows * (g.hcols >> 3);
                std::vector<uint8_t> bits(mb);
                if (!d2h(bits.data(), W.half + (size_t)frame * mb, mb)) return -2;
                uint8_t* o = static_cast<uint8_t*>(dst);
                for (size_t i = 0; i < n; i++) o[i] = (bits[i >> 3] >> (i & 7)) & 1u;
            }
            return (long)n;
        }
        case CTAG_DBG_GRAY: {
            if (!v.gray) return -1;
            const size_t n = (size_t)g.rows * g.cols;
            if (dst && cap >= n) {
                if (hipMemcpy2D(dst, g.cols, v.gray + (size_t)frame * v.gray_frame_stride, v.gray_row_stride, g.cols, g.rows, hipMemcpyDeviceToHost) != hipSuccess)
                    return -2;
            }
            return (long)n;
        }
        case CTAG_DBG_LABELS: {
            const size_t n = (size_t)g.hrows * g.hcols;
            if (dst && cap >= n) {
                int32_t* tmp = nullptr;
                if (hipMalloc(reinterpret_cast<void**>(&tmp), n * 4) != hipSuccess) return -2;
                hipLaunchKernelGGL(k_label_roots, dim3((g.hcols + 255) / 256, g.hrows), dim3(256), 0, s, W.labels + (size_t)frame * g.hrows * g.lp,
                                   W.tile_base + (size_t)frame * g.tiles_x * g.tiles_y, W.root_of + (size_t)frame * g.pool_cap, tmp, g);
                const bool ok = hipStreamSynchronize(s) == hipSuccess && d2h(dst, tmp, n * 4);
                (void)hipFree(tmp);
                if (!ok) return -2;
            }
            return (long)n;
        }
        case CTAG_DBG_CANDIDATES:
        case CTAG_DBG_CAND_QUADS: {
            int nc = 0;
            if (!d2h(&nc, W.ncand + frame, 4)) return -2;
            if (dst && cap >= (size_t)nc * 8 && nc > 0) {
                std::vector<Candidate> c(nc);
                std::vector<QuadOut> q(nc);
                if (!d2h(c.data(), W.cand + (size_t)frame * W.cand_cap, sizeof(Candidate) * nc)) return -2;
                if (!d2h(q.data(), W.quads + (size_t)frame * W.cand_cap, sizeof(QuadOut) * nc)) return -2;
                if (what == CTAG_DBG_CANDIDATES) {
                    int32_t* o = static_cast<int32_t*>(dst);
                    for (int i = 0; i < nc; i++) {
                        o[8 * i + 0] = c[i].area;
                        o[8 * i + 1] = c[i].x_min;
                        o[8 * i + 2] = c[i].y_min;
                        o[8 * i + 3] = c[i].x_max;
                        o[8 * i + 4] = c[i].y_max;
                        o[8 * i + 5] = q[i].valid;
                        o[8 * i + 6] = q[i].n_boundary;
                        o[8 * i + 7] = c[i].root;
                    }
                } else {
                    float* o = static_cast<float*>(dst);
                    for (int i = 0; i < nc; i++)
                        for (int k = 0; k < 8; k++) o[8 * i + k] = q[i].valid ? q[i].c[k] : 0.f;
                }
            }
            return (long)nc * 8;
        }
        case CTAG_DBG_FEATURES0:
        case CTAG_DBG_FEATURES1:
        case CTAG_DBG_FEATURES2: {
            int nf = 0;
            if (!d2h(&nf, W.nfeat + frame, 4)) return -2;
            int fst = 0;
            if (!d2h(&fst, W.status + frame, 4)) return -2;
            if (dst && cap >= (size_t)nf * 19 && nf > 0 && fst != CTAG_OK && what != CTAG_DBG_FEATURES0) {
                // early return ("No feature detected!", fewer features than featureSize): cornerObtain / edgeRefine never ran
                std::memset(dst, 0, (size_t)nf * 19 * sizeof(float));
            } else if (dst && cap >= (size_t)nf * 19 && nf > 0) {
                std::vector<FeatureDev> f(nf);
                const FeatureDev* src = what == CTAG_DBG_FEATURES0 ? W.feat0 : what == CTAG_DBG_FEATURES1 ? W.feat1 : W.feat2;
                if (!d2h(f.data(), src + (size_t)frame * CTAG_MAX_FEATURES, sizeof(FeatureDev) * nf)) return -2;
                float* o = static_cast<float*>(dst);
                for (int i = 0; i < nf; i++) {
                    for (int k = 0; k < 16; k++) o[19 * i + k] = f[i].c[k];
                    o[19 * i + 16] = f[i].center[0];
                    o[19 * i + 17] = f[i].center[1];
                    o[19 * i + 18] = f[i].angle;
                }
            }
            return (long)nf * 19;
        }
        case CTAG_DBG_LINES: {
            int nl = 0;
            if (!d2h(&nl, W.line_count + frame, 4)) return -2;
            nl = std::min(nl, W.line_cap);
            if (dst && cap >= (size_t)nl && nl > 0) {
                std::vector<LineDesc> d(nl);
                if (!d2h(d.data(), W.line_desc + (size_t)frame * W.line_cap, sizeof(LineDesc) * nl)) return -2;
                int32_t* o = static_cast<int32_t*>(dst);
                for (int i = 0; i < nl; i++) o[i] = d[i].n;
            }
            return (long)nl;
        }
        case CTAG_DBG_PREMARKERS: {
            if (!v.keep_pre) return -1;
            if (dst && cap >= 1) {
                if (!d2h(dst, W.premarkers + frame, sizeof(ctag_frame_result))) return -2;
            }
            return 1;
        }
        default: return -1;
    }
}

int ctag_math_probe(ctag_handle* h, int op, int n, const double* a, const double* b, double* out) {
    if (!h || n < 0 || !a || !out) return CTAG_ERR_ARG;
    if (n == 0) return CTAG_OK;
    HandleView v{};
    handle_view(h, &v);
    TK_TRY(hipSetDevice(v.device));
    hipStream_t s = static_cast<hipStream_t>(ctag_stream(h));
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    int rc = CTAG_OK;
    if (hipMalloc(reinterpret_cast<void**>(&da), (size_t)n * 8) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&db), (size_t)n * 8) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&dout), (size_t)n * 8) != hipSuccess)
        rc = CTAG_ERR_HIP;
    if (rc == CTAG_OK && hipMemcpy(da, a, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) rc = CTAG_ERR_HIP;
    if (rc == CTAG_OK && (b ? hipMemcpy(db, b, (size_t)n * 8, hipMemcpyHostToDevice) : hipMemset(db, 0, (size_t)n * 8)) != hipSuccess) rc = CTAG_ERR_HIP;
    if (rc == CTAG_OK) {
        hipLaunchKernelGGL(k_math_probe, dim3((n + 255) / 256), dim3(256), 0, s, op, n, da, db, dout);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess || hipMemcpy(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess)
            rc = CTAG_ERR_HIP;
    }
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    return rc;
}

int ctag_testkit_unpack_gathered(ctag_handle* h, const void* gathered_dev, int n_total, int world, uint64_t width, ctag_frame_result* out_dev) {
    return gather_unpack_gathered(h, gathered_dev, n_total, world, width, out_dev);
}

__global__ void k_stall(long long ticks) {  // s_memrealtime counts at 100 MHz
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
int ctag_testkit_stall_stream(ctag_handle* h, int milliseconds) {
    if (!h || milliseconds < 0 || milliseconds > 10000) return CTAG_ERR_ARG;
    hipLaunchKernelGGL(k_stall, dim3(1), dim3(1), 0, static_cast<hipStream_t>(ctag_stream(h)), (long long)milliseconds * 100000ll);
    return hipGetLastError() == hipSuccess ? CTAG_OK : CTAG_ERR_HIP;
}

int ctag_synth_frames_device(ctag_handle* h, uint8_t* frames_dev, int first_frame, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                             uint64_t seed, int markers_per_frame) {
    return synth_frames_device(h, frames_dev, first_frame, n, rows, cols, row_stride, frame_stride, seed, markers_per_frame, false, 1, 1, 0, 0);
}

int ctag_synth3d_frames_device(ctag_handle* h, uint8_t* frames_dev, int first_frame, int n, int rows, int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                               uint64_t seed, int markers_per_frame, double fx, double fy, double cx, double cy) {
    if (!(fx > 0) || !(fy > 0)) return CTAG_ERR_ARG;
    return synth_frames_device(h, frames_dev, first_frame, n, rows, cols, row_stride, frame_stride, seed, markers_per_frame, true, fx, fy, cx, cy);
}

int ctag_synth3d_frame_host(const int32_t* state, int dict_rows, int dict_cols, uint8_t* frame, int frame_index, int rows, int cols, ptrdiff_t row_stride,
                            uint64_t seed, int markers_per_frame, double fx, double fy, double cx, double cy, ctag_synth3d_truth* truth) {
    if (!state || !frame || rows < 1 || cols < 1 || row_stride < cols || dict_rows < 1 || dict_cols < 1 || !(fx > 0) || !(fy > 0)) return CTAG_ERR_ARG;
    ctag_synth::Frame F;
    ctag_synth::Truth3D T;
    ctag_synth::layout3d(state, dict_rows, dict_cols, seed, frame_index, rows, cols, markers_per_frame, fx, fy, cx, cy, &F, &T);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) frame[(ptrdiff_t)y * row_stride + x] = ctag_synth::pixel(F, x, y, rows, cols);
    if (truth) {
        std::memset(truth, 0, sizeof(*truth));
        truth->n_markers = T.n;
        for (int k = 0; k < T.n && k < 8; k++) {
            truth->dict_row[k] = T.dict_row[k];
            for (int i = 0; i < 9; i++) truth->R[k][i] = T.R[k][i];
            for (int i = 0; i < 3; i++) truth->t[k][i] = T.t[k][i];
            truth->radius[k] = T.radius[k];
        }
    }
    return CTAG_OK;
}

int ctag_synth3d_model(const int32_t* state, int dict_rows, int dict_cols, float* corners) {
    if (!state || !corners || dict_rows < 1 || dict_cols < 1 || dict_cols > ctag_synth::kMaxCols) return CTAG_ERR_ARG;
    for (int r = 0; r < dict_rows; r++) ctag_synth::model_corners(state, dict_cols, r, corners + (size_t)r * dict_cols * 24);
    return CTAG_OK;
}

int ctag_synth_layout_truth(const int32_t* state, int dict_rows, int dict_cols, int frame_index, int rows, int cols, uint64_t seed, int markers_per_frame,
                            ctag_synth_truth* truth) {
    if (!state || !truth || rows < 1 || cols < 1 || dict_rows < 1 || dict_cols < 1) return CTAG_ERR_ARG;
    ctag_synth::Frame F;
    ctag_synth::Truth T;
    ctag_synth::layout(state, dict_rows, dict_cols, seed, frame_index, rows, cols, markers_per_frame, &F, &T);
    fill_truth(T, truth);
    return CTAG_OK;
}

int ctag_synth_frame_host(const int32_t* state, int dict_rows, int dict_cols, uint8_t* frame, int frame_index, int rows, int cols, ptrdiff_t row_stride,
                          uint64_t seed, int markers_per_frame, ctag_synth_truth* truth) {
    if (!state || !frame || rows < 1 || cols < 1 || row_stride < cols || dict_rows < 1 || dict_cols < 1) return CTAG_ERR_ARG;
    ctag_synth::Frame F;
    ctag_synth::Truth T;
    ctag_synth::layout(state, dict_rows, dict_cols, seed, frame_index, rows, cols, markers_per_frame, &F, &T);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) frame[(ptrdiff_t)y * row_stride + x] = ctag_synth::pixel(F, x, y, rows, cols);
    if (truth) fill_truth(T, truth);
    return CTAG_OK;
}

}  // extern "C"
