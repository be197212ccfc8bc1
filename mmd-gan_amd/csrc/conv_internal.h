// internal declarations shared by the convolution translation units
#pragma once
#include "common.h"
#include "tuning.h"
#include "slab_reduce.h"

namespace mmdgan {

// epilogue shared by every conv/dgrad kernel: forward form act(v + bias) or backward form
// (v + bias) * act'(dact[o]);  v arrives already multiplied by the SN scale.
// dact may cover fewer images than the output: output elements at offset >= wrap_from read
// dact[o - wrap_sub] (the discriminator back-propagates [loss_dis rows (2B) ; loss_gen rows (B)] in one
// 3B-row launch, and the last B rows reuse the activations of the fake half).
struct ConvEpilogue {
    const float *bias, *scale, *dact;
    int act;
    long wrap_from, wrap_sub;
    bool out_zeroed;          // host-side hint only (MMDGAN_ACT_FLAG_OUT_ZEROED)
    // added LAST, after the activation / its derivative: out = epilogue(v) + addend[o] (same shape as the output, another buffer).  A residual block's branch sum (layer_func.py:1842) and the fan-in of two gradients at a block's
    // input ride on the launch that produces the second term instead of a pass of their own (mmdgan_conv2d_*_add).
    // A launcher that applies it in its kernel says so with addend_applied(); for any other the entry point adds it
    // with an axpby pass afterwards - every kernel gives the same result, the native ones save the pass.
    const float *addend = nullptr;
    __device__ __forceinline__ long dact_index(long o) const { return o >= wrap_from ? o - wrap_sub : o; }
    __device__ __forceinline__ float apply(float v, int ch, long o) const {
        if (bias) v += bias[ch];
        v = dact ? v * act_bwd_from_out(dact[dact_index(o)], act) : act_fwd(v, act);
        return addend ? v + addend[o] : v;
    }
    __device__ __forceinline__ float4 add4(float4 v, long o) const {      // o: element offset of v.x (16-byte aligned)
        if (addend) {
            const float4 a = *reinterpret_cast<const float4 *>(addend + o);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        return v;
    }
};
void addend_applied();        // conv.hip: the launch just issued applies ep.addend itself (thread-local note for the entry point)
// Batch-norm statistics riding on a convolution (mmdgan_conv2d_*_stats): the entry asks for the per-channel sums of its output
// and of its squares in the totals layout of bn.hip ([slot][2][C] fp64, atomics).  A launcher whose last pass writes the output
// (the slab pass of a reduction-split Winograd launch) takes the request with bn_stats_request() and confirms with
// bn_stats_applied(); otherwise the entry appends the statistics pass of bn.hip - every geometry gives the same totals.
double *bn_stats_request();   // the totals the current call should accumulate into, or nullptr
void bn_stats_applied();
int bn_stats_pass(const float *x, long rows, int C, double *totals, hipStream_t st);   // bn.hip: totals += [sum x, sum x^2]
int bn_slot_count(int C);     // bn.hip: copies of the totals (row split i adds into slot i % count)
constexpr long kNoWrap = 0x7fffffffffffffffL;

int direct_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, float *y, hipStream_t st);
int direct_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, float *dx, hipStream_t st);
int direct_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, hipStream_t st);

// thin first/last-layer kernels (conv_thin.hip)
bool thin_fwd_in_ok(const ConvDims &d);
bool thin_fwd_out_ok(const ConvDims &d);
bool thin_dgrad_in_ok(const ConvDims &d);
bool thin_dgrad_out_ok(const ConvDims &d);
bool thin_wgrad_ok(const ConvDims &d);
int thin_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, float *y, hipStream_t st);
int thin_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, float *dx, hipStream_t st);
int thin_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, hipStream_t st);

// Workgroups (= CUs: 8 waves with the whole register file each) the one-round weight-gradient kernels size their grid for.
// 224, not 256: they run on a stream of their own beside the main stream's launches, and a small main-stream kernel that
// finds every CU held by a 60 us workgroup waits for it; with an eighth of the CUs left over the step is no slower where it
// does not matter and a little faster where it does (CIFAR 256 / 224 / 192 / 160 / 128: 1.923 / 1.914 / 1.912 / 1.950 /
// 1.931 ms; CelebA 13.57 / 13.55 / 13.57 / 13.94 / 13.81).  MMDGAN_WGRAD_CUS.
inline int wgrad_cus() { return tuning().wgrad_cus; }
int thin_wgrad_reduce(const float *partials, int nblocks, long nout, float *dw, hipStream_t st, const float *wdot = nullptr,
                      float *dot = nullptr);   // dot (zero on entry) += <dw, wdot>

// MFMA versions of the same thin layers for 3x3 / stride 1 (conv_thin_mfma.hip)
bool thinm_fwd_n2w_ok(const ConvDims &d);
bool thinm_fwd_w2n_ok(const ConvDims &d);
bool thinm_dgrad_n2w_ok(const ConvDims &d);
bool thinm_dgrad_w2n_ok(const ConvDims &d);
bool thinm_wgrad_ok(const ConvDims &d);
int thinm_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, float *y, hipStream_t st);
int thinm_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, float *dx, hipStream_t st);
int thinm_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, hipStream_t st, const float *wdot = nullptr,
                float *dot = nullptr);   // 1 = no workspace; dot (zero on entry) += <dw, wdot> in the reduction pass

// Winograd F(2x2,3x3) for 3x3 / stride-1 layers (conv_wino.hip); needs the library workspace for G g G^T
bool wino_fwd_ok(const ConvDims &d);
bool wino_dgrad_ok(const ConvDims &d);
bool wino_eligible(const ConvDims &d, bool dgrad);
int wino_transform(const ConvDims &d, const float *w, bool flip, float *U, hipStream_t st);
// U = weights already transformed by wino_transform, or nullptr (then w is transformed into the workspace)
int wino_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, const float *U, float *y, hipStream_t st);
int wino_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, const float *U, float *dx, hipStream_t st);

// Winograd F(4x4,3x3) for the same layers where H and W are multiples of 4 (conv_wino43.hip); U = 36 * C * K floats
bool wino43_eligible(const ConvDims &d, bool dgrad);
int wino43_transform(const ConvDims &d, const float *w, bool flip, float *U, hipStream_t st);
int wino43_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, const float *U, float *y, hipStream_t st);
int wino43_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, const float *U, float *dx, hipStream_t st);
// ... and their weight gradient (conv_wino43w.hip): slabs in the workspace like the F(2x2,3x3) form; returns 1 when there is no
// workspace for them (the caller takes another kernel)
bool wino43_wgrad_ok(const ConvDims &d);
int wino43_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, float *dbias, bool *dbias_done, hipStream_t st,
                 const float *wdot = nullptr, float *dot = nullptr, bool *dot_done = nullptr);

// in-place bias / activation / activation-derivative pass after a split-reduction launch (conv_wino.hip)
int epilogue_pass(float *y, long total, int Ko, const ConvEpilogue &ep, hipStream_t st);
bool wino_wgrad_ok(const ConvDims &d);
// wdot / dot (optional): dot[0] += <dw, wdot> (dot zeroed by the caller); *dot_done says whether it was produced on the way
int wino_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, float *dbias, bool *dbias_done, hipStream_t st,
               const float *wdot = nullptr, float *dot = nullptr, bool *dot_done = nullptr);
// dw[n] = sum of nsplit slabs of n floats; dbias[k] = sum of nsplit rows of k floats (k = 0: none); dot[0] += <dw, wdot>.
// conv_wino2.hip
// Issued right away as a stand-alone pass, or - under mmdgan_wgrad_defer - left to the prologue of the stream's next slab
// weight-gradient launch (slab_reduce.h).  Returns 0 or an error code.
int slab_reduce(const float *part, int nsplit, size_t n, float *dw, const float *dbpart, int k, float *dbias, hipStream_t st,
                const float *wdot = nullptr, float *dot = nullptr);

// Winograd F(2x2,2x2) for 4x4 / stride-2 layers and their input-gradient (conv_wino2.hip); U = 36*C*K floats
// out = epilogue(sum of nslabs partial results, `total` floats each): the second pass of a reduction-split Winograd launch
int slab_epilogue(const float *slabs, int nslabs, long total, int Ko, const ConvEpilogue &ep, float *out, hipStream_t st);
bool wino2_eligible(const ConvDims &d, bool dgrad);
bool wino2_fwd_ok(const ConvDims &d);
bool wino2_dgrad_ok(const ConvDims &d);
int wino2_transform(const ConvDims &d, const float *w, bool dgrad, float *U, hipStream_t st);
int wino2_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, const float *U, float *y, hipStream_t st);
int wino2_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, const float *U, float *dx, hipStream_t st);
bool wino2_wgrad_ok(const ConvDims &d);
int wino2_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, float *dbias, bool *dbias_done, hipStream_t st,
                const float *wdot = nullptr, float *dot = nullptr, bool *dot_done = nullptr);

// MFMA implicit-GEMM path (conv_igemm.hip); *_ok() say whether a geometry is eligible
bool igemm_fwd_ok(const ConvDims &d);
bool igemm_dgrad_ok(const ConvDims &d);
bool igemm_wgrad_ok(const ConvDims &d);
int igemm_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, float *y, hipStream_t st);
int igemm_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, float *dx, hipStream_t st);
int igemm_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, float *dbias, hipStream_t st);   // dbias: optional column sums of dy

}  // namespace mmdgan
