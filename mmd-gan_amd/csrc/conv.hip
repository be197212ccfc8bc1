// C-ABI entry points of the convolution family: validate, then dispatch to the MFMA implicit-GEMM
// kernels (conv_igemm.hip) when the geometry is tile-aligned, to the thin first/last-layer kernels
// (conv_thin.hip) when one channel count is tiny, else to the generic direct kernels.
#include <stdlib.h>

#include "conv_internal.h"
#include <stdint.h>

using namespace mmdgan;

namespace {
int validate(const mmdgan_conv_geom *g, const char *what) {
    MMDGAN_REQUIRE(g, "%s: null geometry", what);
    MMDGAN_REQUIRE(g->N >= 1 && g->H >= 1 && g->W >= 1 && g->C >= 1 && g->K >= 1, "%s: bad shape", what);
    MMDGAN_REQUIRE(g->R >= 1 && g->R <= 7 && g->stride >= 1 && g->stride <= 7, "%s: bad kernel %d / stride %d", what,
                   g->R, g->stride);
    return MMDGAN_OK;
}
// MMDGAN_FORCE_DIRECT=1 routes everything to the direct kernels (A/B debugging aid)
bool force_direct() { return tuning().force_direct != 0; }
// a kernel smaller than its stride (the 1x1 stride-2 transposed conv a residual block on 'tc' may have as its shortcut,
// layer_func.py:1725-1745: taps that skip input pixels / output pixels no tap reaches): the generic direct kernels, whose
// index arithmetic assumes nothing about R against the stride; the tiled families are built for R >= stride
bool generic_only(const ConvDims &d) { return force_direct() || d.R < d.stride; }
// MMDGAN_THIN_VALU=1 keeps the thin first/last layers on the VALU kernels (A/B against the MFMA ones)
bool force_valu_thin() { return tuning().thin_valu != 0; }
inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }    // also true for nullptr
thread_local bool t_addend_applied = false;
thread_local double *t_bn_totals = nullptr;
thread_local bool t_bn_applied = false;
}  // namespace
namespace mmdgan {
void addend_applied() { t_addend_applied = true; }
double *bn_stats_request() { return t_bn_totals; }
void bn_stats_applied() { t_bn_applied = true; }
}

namespace {
// dact_batch: number of images dact_of holds (0 = as many as the output); the trailing
// (N - dact_batch) output images reuse the LAST (N - dact_batch) images of dact_of
int make_wrap(int N, int dact_batch, long per_image, const char *what, long *from, long *sub) {
    *from = kNoWrap; *sub = 0;
    if (dact_batch == 0 || dact_batch == N) return MMDGAN_OK;
    MMDGAN_REQUIRE(dact_batch > 0 && dact_batch < N && N - dact_batch <= dact_batch, "%s: bad dact_batch %d for N %d", what,
                   dact_batch, N);
    *from = (long)dact_batch * per_image;
    *sub = (long)(N - dact_batch) * per_image;
    return MMDGAN_OK;
}
}  // namespace

static int conv2d_fwd_impl(const mmdgan_conv_geom *g, const float *x, const float *w, const float *bias, const float *scale,
                           int act, const float *dact_of, int dact_batch, const float *addend, float *y, void *stream);
static int conv2d_dgrad_impl(const mmdgan_conv_geom *g, const float *dy, const float *w, const float *bias, const float *scale,
                             int act, const float *dact_of, int dact_batch, const float *addend, float *dx, void *stream);
// out = conv(...) + addend: the kernel's own epilogue where it has one for it, an axpby pass behind it otherwise
template <class F>
static int with_addend(const float *addend, float *out, long n, void *stream, F &&launch) {
    t_addend_applied = false;
    if (int rc = launch()) return rc;
    if (addend && !t_addend_applied) return mmdgan_axpby(out, 1.f, addend, 1.f, out, n, stream);
    return MMDGAN_OK;
}

extern "C" int mmdgan_conv2d_fwd(const mmdgan_conv_geom *g, const float *x, const float *w, const float *bias,
                                 const float *scale, int act, const float *dact_of, int dact_batch, float *y,
                                 void *stream) {
    return conv2d_fwd_impl(g, x, w, bias, scale, act, dact_of, dact_batch, nullptr, y, stream);
}
extern "C" int mmdgan_conv2d_fwd_add(const mmdgan_conv_geom *g, const float *x, const float *w, const float *bias,
                                     const float *scale, int act, const float *dact_of, int dact_batch, const float *addend,
                                     float *y, void *stream) {
    if (int rc = validate(g, "conv2d_fwd_add")) return rc;
    MMDGAN_REQUIRE(addend && al16(addend) && addend != y, "conv2d_fwd_add: the addend must be a 16-byte aligned tensor of the output's shape, not the output itself");
    MMDGAN_REQUIRE(!(act & MMDGAN_ACT_FLAG_OUT_ZEROED), "conv2d_fwd_add: not with MMDGAN_ACT_FLAG_OUT_ZEROED (split accumulation)");
    const ConvDims d = conv_dims(*g);
    return with_addend(addend, y, (long)d.N * d.P * d.Q * d.K, stream,
                       [&]() { return conv2d_fwd_impl(g, x, w, bias, scale, act, dact_of, dact_batch, addend, y, stream); });
}
static int conv2d_fwd_impl(const mmdgan_conv_geom *g, const float *x, const float *w, const float *bias, const float *scale,
                           int act, const float *dact_of, int dact_batch, const float *addend, float *y, void *stream) {
    if (int rc = validate(g, "conv2d_fwd")) return rc;
    MMDGAN_REQUIRE(x && w && y, "conv2d_fwd: null pointer");
    const bool out_zeroed = (act & MMDGAN_ACT_FLAG_OUT_ZEROED) != 0, w_wino = (act & MMDGAN_ACT_FLAG_W_WINOGRAD) != 0;
    const bool w_wino43 = (act & MMDGAN_ACT_FLAG_W_WINOGRAD43) != 0;
    act &= ~(MMDGAN_ACT_FLAG_OUT_ZEROED | MMDGAN_ACT_FLAG_W_WINOGRAD | MMDGAN_ACT_FLAG_W_WINOGRAD43);
    MMDGAN_REQUIRE(act >= MMDGAN_ACT_LINEAR && act <= MMDGAN_ACT_TANH, "conv2d_fwd: unknown activation %d", act);
    MMDGAN_REQUIRE(!(w_wino && w_wino43), "conv2d_fwd: MMDGAN_ACT_FLAG_W_WINOGRAD and MMDGAN_ACT_FLAG_W_WINOGRAD43 together");
    const ConvDims d = conv_dims(*g);
    long wf, ws;
    if (int rc = make_wrap(d.N, dact_of ? dact_batch : 0, (long)d.P * d.Q * d.K, "conv2d_fwd", &wf, &ws)) return rc;
    const ConvEpilogue ep{bias, scale, dact_of, act, wf, ws, out_zeroed, addend};
    if (w_wino) {
        MMDGAN_REQUIRE(wino_eligible(d, false) || wino2_eligible(d, false),
                       "conv2d_fwd: MMDGAN_ACT_FLAG_W_WINOGRAD on a geometry mmdgan_wino_eligible() rejects");
        return d.R == 3 ? wino_fwd(d, ep, x, nullptr, w, y, (hipStream_t)stream) : wino2_fwd(d, ep, x, nullptr, w, y, (hipStream_t)stream);
    }
    if (w_wino43) {
        MMDGAN_REQUIRE(wino43_eligible(d, false), "conv2d_fwd: MMDGAN_ACT_FLAG_W_WINOGRAD43 on a geometry whose mmdgan_wino_algo() is not MMDGAN_WINO_F43");
        return wino43_fwd(d, ep, x, nullptr, w, y, (hipStream_t)stream);
    }
    if (generic_only(d)) return direct_fwd(d, ep, x, w, y, (hipStream_t)stream);
    if (!force_direct() && wino43_eligible(d, false) && workspace(sizeof(float) * 36 * (size_t)d.C * d.K))
        return wino43_fwd(d, ep, x, w, nullptr, y, (hipStream_t)stream);
    if (!force_direct() && wino2_fwd_ok(d)) return wino2_fwd(d, ep, x, w, nullptr, y, (hipStream_t)stream);
    if (!force_direct() && wino_fwd_ok(d)) return wino_fwd(d, ep, x, w, nullptr, y, (hipStream_t)stream);
    if (!force_direct() && igemm_fwd_ok(d)) return igemm_fwd(d, ep, x, w, y, (hipStream_t)stream);
    if (!force_direct() && !force_valu_thin() && (thinm_fwd_n2w_ok(d) || thinm_fwd_w2n_ok(d)) && al16(x) && al16(y) &&
        al16(bias) && al16(dact_of))
        return thinm_fwd(d, ep, x, w, y, (hipStream_t)stream);
    if (!force_direct() && (thin_fwd_in_ok(d) || thin_fwd_out_ok(d))) return thin_fwd(d, ep, x, w, y, (hipStream_t)stream);
    return direct_fwd(d, ep, x, w, y, (hipStream_t)stream);
}

// y = conv2d_fwd(...) and bn_totals += [sum y, sum y^2] per output channel (the layout of mmdgan_bn_workspace_bytes; the caller
// zeroes it or runs under mmdgan_set_outputs_prezeroed(1)): what mmdgan_bn_fwd_apply normalises with - layer_func.py:953-966
// behind a convolution, without a pass over y for the statistics where the launch's last pass can form them
template <class F>
static int with_bn_stats(double *totals, const float *out, long rows, int C, void *stream, const char *what, F &&launch) {
    MMDGAN_REQUIRE(totals, "%s: null totals", what);
    if (!outputs_prezeroed() && memset_async(totals, 0, sizeof(double) * bn_slot_count(C) * 2 * C, (hipStream_t)stream) != hipSuccess)
        return check_launch("conv2d stats memset");
    t_bn_totals = totals;
    t_bn_applied = false;
    const int rc = launch();
    t_bn_totals = nullptr;
    if (rc) return rc;
    return t_bn_applied ? MMDGAN_OK : bn_stats_pass(out, rows, C, totals, (hipStream_t)stream);
}
extern "C" int mmdgan_conv2d_fwd_stats(const mmdgan_conv_geom *g, const float *x, const float *w, const float *bias,
                                       const float *scale, int act, float *y, void *bn_totals, void *stream) {
    if (int rc = validate(g, "conv2d_fwd_stats")) return rc;
    const ConvDims d = conv_dims(*g);
    return with_bn_stats((double *)bn_totals, y, (long)d.N * d.P * d.Q, d.K, stream, "conv2d_fwd_stats",
                         [&]() { return conv2d_fwd_impl(g, x, w, bias, scale, act, nullptr, 0, nullptr, y, stream); });
}
extern "C" int mmdgan_conv2d_dgrad_stats(const mmdgan_conv_geom *g, const float *dy, const float *w, const float *bias,
                                         const float *scale, int act, float *dx, void *bn_totals, void *stream) {
    if (int rc = validate(g, "conv2d_dgrad_stats")) return rc;
    return with_bn_stats((double *)bn_totals, dx, (long)g->N * g->H * g->W, g->C, stream, "conv2d_dgrad_stats",
                         [&]() { return conv2d_dgrad_impl(g, dy, w, bias, scale, act, nullptr, 0, nullptr, dx, stream); });
}

extern "C" int mmdgan_conv2d_dgrad(const mmdgan_conv_geom *g, const float *dy, const float *w, const float *bias,
                                   const float *scale, int act, const float *dact_of, int dact_batch, float *dx,
                                   void *stream) {
    return conv2d_dgrad_impl(g, dy, w, bias, scale, act, dact_of, dact_batch, nullptr, dx, stream);
}
extern "C" int mmdgan_conv2d_dgrad_add(const mmdgan_conv_geom *g, const float *dy, const float *w, const float *bias,
                                       const float *scale, int act, const float *dact_of, int dact_batch, const float *addend,
                                       float *dx, void *stream) {
    if (int rc = validate(g, "conv2d_dgrad_add")) return rc;
    MMDGAN_REQUIRE(addend && al16(addend) && addend != dx, "conv2d_dgrad_add: the addend must be a 16-byte aligned tensor of the output's shape, not the output itself");
    MMDGAN_REQUIRE(!(act & MMDGAN_ACT_FLAG_OUT_ZEROED), "conv2d_dgrad_add: not with MMDGAN_ACT_FLAG_OUT_ZEROED (split accumulation)");
    return with_addend(addend, dx, (long)g->N * g->H * g->W * g->C, stream,
                       [&]() { return conv2d_dgrad_impl(g, dy, w, bias, scale, act, dact_of, dact_batch, addend, dx, stream); });
}
static int conv2d_dgrad_impl(const mmdgan_conv_geom *g, const float *dy, const float *w, const float *bias, const float *scale,
                             int act, const float *dact_of, int dact_batch, const float *addend, float *dx, void *stream) {
    if (int rc = validate(g, "conv2d_dgrad")) return rc;
    MMDGAN_REQUIRE(dy && w && dx, "conv2d_dgrad: null pointer");
    const bool out_zeroed = (act & MMDGAN_ACT_FLAG_OUT_ZEROED) != 0, w_wino = (act & MMDGAN_ACT_FLAG_W_WINOGRAD) != 0;
    const bool w_wino43 = (act & MMDGAN_ACT_FLAG_W_WINOGRAD43) != 0;
    act &= ~(MMDGAN_ACT_FLAG_OUT_ZEROED | MMDGAN_ACT_FLAG_W_WINOGRAD | MMDGAN_ACT_FLAG_W_WINOGRAD43);
    MMDGAN_REQUIRE(act >= MMDGAN_ACT_LINEAR && act <= MMDGAN_ACT_TANH, "conv2d_dgrad: unknown activation %d", act);
    MMDGAN_REQUIRE(!(w_wino && w_wino43), "conv2d_dgrad: MMDGAN_ACT_FLAG_W_WINOGRAD and MMDGAN_ACT_FLAG_W_WINOGRAD43 together");
    const ConvDims d = conv_dims(*g);
    long wf, ws;
    if (int rc = make_wrap(d.N, dact_of ? dact_batch : 0, (long)d.H * d.W * d.C, "conv2d_dgrad", &wf, &ws)) return rc;
    const ConvEpilogue ep{bias, scale, dact_of, act, wf, ws, out_zeroed, addend};
    if (w_wino) {
        MMDGAN_REQUIRE(wino_eligible(d, true) || wino2_eligible(d, true),
                       "conv2d_dgrad: MMDGAN_ACT_FLAG_W_WINOGRAD on a geometry mmdgan_wino_eligible() rejects");
        return d.R == 3 ? wino_dgrad(d, ep, dy, nullptr, w, dx, (hipStream_t)stream) : wino2_dgrad(d, ep, dy, nullptr, w, dx, (hipStream_t)stream);
    }
    if (w_wino43) {
        MMDGAN_REQUIRE(wino43_eligible(d, true), "conv2d_dgrad: MMDGAN_ACT_FLAG_W_WINOGRAD43 on a geometry whose mmdgan_wino_algo() is not MMDGAN_WINO_F43");
        return wino43_dgrad(d, ep, dy, nullptr, w, dx, (hipStream_t)stream);
    }
    if (generic_only(d)) return direct_dgrad(d, ep, dy, w, dx, (hipStream_t)stream);
    if (!force_direct() && wino43_eligible(d, true) && workspace(sizeof(float) * 36 * (size_t)d.C * d.K))
        return wino43_dgrad(d, ep, dy, w, nullptr, dx, (hipStream_t)stream);
    if (!force_direct() && wino2_dgrad_ok(d)) return wino2_dgrad(d, ep, dy, w, nullptr, dx, (hipStream_t)stream);
    if (!force_direct() && wino_dgrad_ok(d)) return wino_dgrad(d, ep, dy, w, nullptr, dx, (hipStream_t)stream);
    if (!force_direct() && igemm_dgrad_ok(d)) return igemm_dgrad(d, ep, dy, w, dx, (hipStream_t)stream);
    if (!force_direct() && !force_valu_thin() && (thinm_dgrad_n2w_ok(d) || thinm_dgrad_w2n_ok(d)) && al16(dy) && al16(dx) &&
        al16(bias) && al16(dact_of))
        return thinm_dgrad(d, ep, dy, w, dx, (hipStream_t)stream);
    if (!force_direct() && (thin_dgrad_in_ok(d) || thin_dgrad_out_ok(d)))
        return thin_dgrad(d, ep, dy, w, dx, (hipStream_t)stream);
    return direct_dgrad(d, ep, dy, w, dx, (hipStream_t)stream);
}

extern "C" int mmdgan_wino_eligible(const mmdgan_conv_geom *g, int dgrad) {
    if (!g || g->N < 1 || g->H < 1 || g->W < 1 || g->C < 1 || g->K < 1 || g->R < 1 || g->stride < 1) return 0;
    const ConvDims d = conv_dims(*g);
    return !force_direct() && (wino_eligible(d, dgrad != 0) || wino2_eligible(d, dgrad != 0)) ? 1 : 0;
}

extern "C" int mmdgan_wino_algo(const mmdgan_conv_geom *g, int dgrad) {
    if (!g || g->N < 1 || g->H < 1 || g->W < 1 || g->C < 1 || g->K < 1 || g->R < 1 || g->stride < 1 || force_direct()) return MMDGAN_WINO_NONE;
    const ConvDims d = conv_dims(*g);
    if (wino43_eligible(d, dgrad != 0)) return MMDGAN_WINO_F43;
    if (wino_eligible(d, dgrad != 0)) return MMDGAN_WINO_F23;
    if (wino2_eligible(d, dgrad != 0)) return MMDGAN_WINO_F22S2;
    return MMDGAN_WINO_NONE;
}

extern "C" int mmdgan_wgrad_algo(const mmdgan_conv_geom *g) {
    if (!g || g->N < 1 || g->H < 1 || g->W < 1 || g->C < 1 || g->K < 1 || g->R < 1 || g->stride < 1 || force_direct()) return MMDGAN_WINO_NONE;
    const ConvDims d = conv_dims(*g);
    if (wino43_wgrad_ok(d)) return MMDGAN_WINO_F43;
    if (wino_wgrad_ok(d)) return MMDGAN_WINO_F23;
    if (wino2_wgrad_ok(d)) return MMDGAN_WINO_F22S2;
    return MMDGAN_WINO_NONE;
}

extern "C" size_t mmdgan_wino_algo_weight_bytes(const mmdgan_conv_geom *g, int algo) {
    if (!g) return 0;
    const size_t ck = sizeof(float) * (size_t)g->C * g->K;
    if (algo == MMDGAN_WINO_F23) return g->R == 3 ? 16 * ck : 0;
    if (algo == MMDGAN_WINO_F43) return g->R == 3 ? 36 * ck : 0;
    if (algo == MMDGAN_WINO_F22S2) return g->R == 4 ? 36 * ck : 0;
    return 0;
}

extern "C" int mmdgan_wino_transform_algo(const mmdgan_conv_geom *g, const float *w, int dgrad, int algo, float *u, void *stream) {
    if (algo != MMDGAN_WINO_F43) {
        MMDGAN_REQUIRE(g && ((algo == MMDGAN_WINO_F23 && g->R == 3) || (algo == MMDGAN_WINO_F22S2 && g->R == 4) || algo == MMDGAN_WINO_NONE),
                       "wino_transform_algo: algorithm %d does not fit the kernel size", algo);
        return mmdgan_wino_transform(g, w, dgrad, u, stream);
    }
    if (int rc = validate(g, "wino_transform_algo")) return rc;
    MMDGAN_REQUIRE(w && u, "wino_transform_algo: null pointer");
    MMDGAN_REQUIRE(g->R == 3 && g->stride == 1, "wino_transform_algo: F(4x4,3x3) is for 3x3 stride-1 kernels (got %dx%d stride %d)", g->R, g->R,
                   g->stride);
    return wino43_transform(conv_dims(*g), w, dgrad != 0, u, (hipStream_t)stream);
}

extern "C" size_t mmdgan_wino_weight_bytes(const mmdgan_conv_geom *g) {
    return g ? sizeof(float) * (g->R == 3 ? 16 : 36) * (size_t)g->C * g->K : 0;
}

extern "C" int mmdgan_wino_transform(const mmdgan_conv_geom *g, const float *w, int dgrad, float *u, void *stream) {
    if (int rc = validate(g, "wino_transform")) return rc;
    MMDGAN_REQUIRE(w && u, "wino_transform: null pointer");
    MMDGAN_REQUIRE((g->R == 3 && g->stride == 1) || (g->R == 4 && g->stride == 2),
                   "wino_transform: 3x3 stride 1 or 4x4 stride 2 kernels only (got %dx%d stride %d)", g->R, g->R, g->stride);
    const ConvDims d = conv_dims(*g);
    return g->R == 3 ? wino_transform(d, w, dgrad != 0, u, (hipStream_t)stream) : wino2_transform(d, w, dgrad != 0, u, (hipStream_t)stream);
}

// wdot / dot (optional): dot[0] = <dw, wdot> - the scalar of the spectral-norm fix-up (mmdgan_conv2d_wgrad_sn)
static int wgrad_impl(const mmdgan_conv_geom *g, const float *x, const float *dy, float *dw, float *dbias, void *stream,
                      const char *what, const float *wdot = nullptr, float *dot = nullptr) {
    if (int rc = validate(g, what)) return rc;
    MMDGAN_REQUIRE(x && dy && dw, "%s: null pointer", what);
    const ConvDims d = conv_dims(*g);
    const long nw = (long)d.R * d.R * d.C * d.K;
    if (!force_direct() && wino43_wgrad_ok(d)) {             // F(4x4,3x3): 1 = no workspace for its slabs -> the kernels below
        bool db_done = false, dot_done = false;
        if (wdot && zero_output(dot, sizeof(float), (hipStream_t)stream) != hipSuccess) return check_launch("conv2d_wgrad memset");
        int rcw = wino43_wgrad(d, x, dy, dw, dbias, &db_done, (hipStream_t)stream, wdot, dot, &dot_done);
        if (rcw != 1) {
            if (rcw == 0 && dbias && !db_done) rcw = mmdgan_colsum(dy, (long)d.N * d.P * d.Q, d.K, dbias, stream);
            if (rcw == 0 && wdot && !dot_done) rcw = mmdgan_dot(dw, wdot, nw, dot, stream);
            return rcw;
        }
    }
    if (!force_direct() && (wino_wgrad_ok(d) || wino2_wgrad_ok(d))) {
        bool db_done = false, dot_done = false;    // the slab kernels sum dy on the way, their reduction pass forms <dw, w>
        if (wdot && zero_output(dot, sizeof(float), (hipStream_t)stream) != hipSuccess) return check_launch("conv2d_wgrad memset");
        int rcw = d.R == 3 ? wino_wgrad(d, x, dy, dw, dbias, &db_done, (hipStream_t)stream, wdot, dot, &dot_done)
                           : wino2_wgrad(d, x, dy, dw, dbias, &db_done, (hipStream_t)stream, wdot, dot, &dot_done);
        if (rcw == 0 && dbias && !db_done) rcw = mmdgan_colsum(dy, (long)d.N * d.P * d.Q, d.K, dbias, stream);
        if (rcw == 0 && wdot && !dot_done) rcw = mmdgan_dot(dw, wdot, nw, dot, stream);
        return rcw;
    }
    // (not a slab kernel: it has no prologue for the slabs a previous call left behind under mmdgan_wgrad_defer - they are
    // summed by the stand-alone pass now, so that "complete behind the next weight-gradient call" holds for every geometry)
    if (wgrad_flush_pending()) return MMDGAN_E_LAUNCH;
    int rc = 1;
    bool dot_done = false;
    const bool generic = generic_only(d);
    if (!generic && igemm_wgrad_ok(d)) rc = igemm_wgrad(d, x, dy, dw, dbias, (hipStream_t)stream);   // sums dy on the way
    else {
        if (!generic && !force_valu_thin() && thinm_wgrad_ok(d) && d.N > 1) {
            // (its reduction pass forms <dw, w> where it has the finished dw)
            if (wdot && zero_output(dot, sizeof(float), (hipStream_t)stream) != hipSuccess) return check_launch("conv2d_wgrad memset");
            rc = thinm_wgrad(d, x, dy, dw, (hipStream_t)stream, wdot, dot);
            dot_done = rc == 0;
        }
        if (rc > 0) {                                              // 1: no workspace registered -> VALU kernel
            if (!generic && thin_wgrad_ok(d)) rc = thin_wgrad(d, x, dy, dw, (hipStream_t)stream);
            else rc = direct_wgrad(d, x, dy, dw, (hipStream_t)stream);
        }
        if (rc == 0 && dbias) rc = mmdgan_colsum(dy, (long)d.N * d.P * d.Q, d.K, dbias, stream);
    }
    if (rc == 0 && wdot && !dot_done) rc = mmdgan_dot(dw, wdot, nw, dot, stream);
    return rc;
}

extern "C" int mmdgan_conv2d_wgrad(const mmdgan_conv_geom *g, const float *x, const float *dy, float *dw, void *stream) {
    return wgrad_impl(g, x, dy, dw, nullptr, stream, "conv2d_wgrad");
}

extern "C" int mmdgan_conv2d_wgrad_bias(const mmdgan_conv_geom *g, const float *x, const float *dy, float *dw, float *dbias,
                                        void *stream) {
    MMDGAN_REQUIRE(dbias, "conv2d_wgrad_bias: null pointer");
    return wgrad_impl(g, x, dy, dw, dbias, stream, "conv2d_wgrad_bias");
}

extern "C" int mmdgan_conv2d_wgrad_sn(const mmdgan_conv_geom *g, const float *x, const float *dy, float *dw, float *dbias,
                                      const float *w, float *dot_gw, void *stream) {
    MMDGAN_REQUIRE(w && dot_gw, "conv2d_wgrad_sn: null pointer");
    return wgrad_impl(g, x, dy, dw, dbias, stream, "conv2d_wgrad_sn", w, dot_gw);
}
