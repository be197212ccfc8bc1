/* libmmdgan_hip - C ABI of the MI355X (gfx950) hot path of MMD-GAN training.
 *
 * The reference (richardwth/MMD-GAN, TensorFlow 1.x) has NO native code and no FFI: its seam is
 * the Python API.  This header is therefore the boundary the build's own
 * Python host code (mmd-gan_amd/mmdgan_hip/, mirroring GeneralTools/ and DeepLearning/) binds with
 * ctypes; each entry cites the reference call site (file:line under /root/reference) whose stock
 * TF op it replaces.  INTEGRATION.md shows the binding stub.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every function returns 0 on success or a negative
 *     MMDGAN_E_* code and never throws; mmdgan_last_error() gives the thread-local message.
 *   - the CALLER owns all device memory (torch allocates); the library allocates nothing.
 *     Scratch is passed in explicitly (sizes from the *_workspace_bytes queries).
 *   - every call is asynchronous and ordered on the hipStream_t passed as `stream` (void*; NULL =
 *     the default stream).  No call synchronises the device.
 *   - activations are NHWC fp32; conv kernels HWIO `[R,S,C,K]` exactly as the reference stores
 *     them (layer_func.py:584), transposed-conv kernels `[R,S,Kout,Cin]` (layer_func.py:595),
 *     dense kernels `[in,out]` (layer_func.py:577).
 *   - "scale" arguments are DEVICE pointers to one float (NULL = 1.0): the spectral-norm multiplier
 *     act_k/sigma (layer_func.py:886-887) is consumed from device memory, never read by the host.
 */
#ifndef MMDGAN_HIP_H
#define MMDGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMDGAN_OK 0
#define MMDGAN_E_ARG (-1)      /* bad argument (shape, enum, null pointer) */
#define MMDGAN_E_LAUNCH (-2)   /* hipLaunch / runtime error */
#define MMDGAN_E_UNSUPPORTED (-3)

/* activation enum - layer_func.py:104-151 (lrelu alpha = 0.1, layer_func.py:112) */
#define MMDGAN_ACT_LINEAR 0
#define MMDGAN_ACT_RELU 1
#define MMDGAN_ACT_LRELU 2
#define MMDGAN_ACT_TANH 3
/* OR-ed into `act` of conv2d_fwd / conv2d_dgrad / gemm: the output buffer is zero on entry and the entry may
 * ACCUMULATE into it (lets a launch with too few tiles for 256 CUs split its reduction even when
 * mmdgan_set_outputs_prezeroed(1) is in force).  Only honoured with a linear epilogue (no dact_of). */
#define MMDGAN_ACT_FLAG_OUT_ZEROED 0x100
/* OR-ed into `act` of conv2d_fwd / conv2d_dgrad: `w` is not the HWIO kernel but the tensor mmdgan_wino_transform()
 * produced from it (same dgrad flag as the call), for a geometry mmdgan_wino_eligible() accepts.  Lets a caller
 * whose weights change once per step transform them once, off the critical path, instead of inside every call. */
#define MMDGAN_ACT_FLAG_W_WINOGRAD 0x200
/* the same for a tensor mmdgan_wino_transform_algo(..., MMDGAN_WINO_F43, ...) produced: the call runs F(4x4,3x3)
 * (geometry: mmdgan_wino_algo() == MMDGAN_WINO_F43).  One of the two flags, not both. */
#define MMDGAN_ACT_FLAG_W_WINOGRAD43 0x400
/* Winograd algorithms (mmdgan_wino_algo): which transformed-weight layout a geometry's call expects */
#define MMDGAN_WINO_NONE 0
#define MMDGAN_WINO_F23 1      /* F(2x2,3x3): 3x3 stride 1, 16 * C * K floats (csrc/conv_wino.hip) */
#define MMDGAN_WINO_F22S2 2    /* F(2x2,2x2) on the 4 parity segments of 4x4 stride 2, 36 * C * K floats (csrc/conv_wino2.hip) */
#define MMDGAN_WINO_F43 3      /* F(4x4,3x3): 3x3 stride 1 with H, W multiples of 4, 36 * C * K floats (csrc/conv_wino43.hip) */

/* loss enum - math_func.py:2644-2647 */
#define MMDGAN_LOSS_REP 0
#define MMDGAN_LOSS_RMB 1
#define MMDGAN_LOSS_MMD_G 2    /* 'mmd_g' / 'fixed_g': five-scale Gaussian mixture, math_func.py:2160-2173 */
#define MMDGAN_LOSS_MGB 3      /* 'mgb': sigma-1 Gaussian with bounds (0.25, 4) on the D side, math_func.py:2175-2193 */
#define MMDGAN_LOSS_HINGE 4    /* math_func.py:2137-2143 (w, bounds, masks, dist, workspace unused) */
#define MMDGAN_LOSS_LOGISTIC 5 /* 'logistic' / '': non-saturating, math_func.py:2128-2135 */
/* 'mmd_g_mix' / 'fixed_g_mix' (GANLoss._mmd_g_mix_, math_func.py:2195-2228) and 'sgm' (_single_mmd_g_mix_, :2230-2263):
 * served by mmdgan_mmd_mix_loss, not by mmdgan_mmd_loss */
#define MMDGAN_LOSS_MMD_G_MIX 6
#define MMDGAN_LOSS_SGM 7
/* OR-ed into loss_type: write the four gradient blocks of mmdgan_mmd_loss in the order
 * [dL_dis/ds_x, dL_dis/ds_gen, dL_gen/ds_gen, dL_gen/ds_x] instead of [dL_gen/ds_gen, dL_gen/ds_x, dL_dis/ds_gen,
 * dL_dis/ds_x]: the first 3B rows are then exactly the score gradient a discriminator fed [real ; fake] (and the
 * fake half again for the generator loss) back-propagates - no gather copies. */
#define MMDGAN_LOSS_FLAG_GRADS_DIS_FIRST 0x100

const char *mmdgan_last_error(void);
/* ABI version of this header: bumped whenever an entry's argument list, a workspace size or a calling rule changes
 * (200: rounds 1-4 - although mmdgan_bn_bwd gained `beta` and mmdgan_bn_workspace_bytes grew in round 4 without a bump;
 * 500: round 5 - mmdgan_wgrad_defer / mmdgan_wgrad_flush, BN workspace documented as [slots][2][C];
 * 600: round 6 - mmdgan_wino_algo / _algo_weight_bytes / _transform_algo, mmdgan_wino_job.algo, MMDGAN_ACT_FLAG_W_WINOGRAD43;
 * 610: mmdgan_wgrad_algo, the F(4x4,3x3) weight gradient in the default selection).  A caller compares
 * mmdgan_version() of the library it loaded with the MMDGAN_VERSION it was built against (mmdgan_hip/_lib.py does). */
#define MMDGAN_VERSION 610
int mmdgan_version(void);
/* Which kernel a convolution call takes is decided by the geometry and by a handful of process-wide switches (environment
 * variables MMDGAN_*, read once: csrc/tuning.h lists them with their defaults - the defaults are the configuration the
 * bench runs and tests/test_production_gpu.py pins).  This writes "name=value ..." of all of them, a '*' behind every value
 * that is off its default, so a log can state the selection it ran under.  Returns the size a full listing needs. */
long mmdgan_tuning_describe(char *buf, size_t cap);
/* 1 if a gfx950 device is visible to this process, 0 otherwise (never an error) */
int mmdgan_device_ok(void);
/* ------------------------------------------------------------------------------------------------
 * Handles.  All mutable library state - the registered workspace, the prezeroed mode, recorded launch plans and the
 * events of the stream helpers - lives in an opaque handle, one per engine ("one handle per (process,
 * device); a handle is not thread-safe, distinct handles are; no global mutable state").  The entries below and every
 * compute entry act on the CALLING THREAD's current handle; a thread that never called mmdgan_make_current uses the
 * process default handle, so a single-engine caller needs none of this.  Two engines in one process each create a
 * handle and make it current around their own calls: their workspaces and plans no longer meet.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mmdgan_handle mmdgan_handle;
int mmdgan_create(mmdgan_handle **out);
int mmdgan_destroy(mmdgan_handle *h);          /* destroys its plans and events; the caller's device memory is untouched */
int mmdgan_make_current(mmdgan_handle *h);     /* NULL: back to the process default handle */

/* Optional caller-owned device scratch (the library itself allocates nothing).  Kernels that would
 * otherwise combine per-workgroup partial results with contended atomics write their partials here
 * and reduce them in a second pass when the registered region is large enough; without it they fall
 * back to atomics (or to the direct kernels).  Users: the thin-layer weight gradients, the
 * Winograd-domain weight gradients of the 3x3 / stride-1 and 4x4 / stride-2 layers (one slab of
 * dw's size per split of the pixel range: <= 256 workgroups x 4 x 64 x 128 floats = 33.6 MB and
 * <= 256 x 9 x 32 x 128 floats = 37.8 MB, plus the bias-gradient rows), and the in-call Winograd
 * weight transforms of callers that do not pass transformed weights.  64 MB covers every shape of
 * the reference's architectures.  One region per HANDLE; calls that use it must be ordered on ONE
 * stream (the engines issue every weight gradient on their weight-gradient stream and transform
 * weights once per step).  ptr == NULL unregisters. */
int mmdgan_set_workspace(void *ptr, size_t bytes);
/* Several entries accumulate into their output with atomics (split reductions, column sums, dot)
 * and zero it first with an internal memset node.  A caller that zeroes those outputs itself - e.g.
 * one memset over a whole gradient arena per step instead of ~40 small ones - sets this to 1 and the
 * internal memsets are skipped; the outputs MUST then be zero on entry.  In this mode conv2d_fwd /
 * conv2d_dgrad split their reduction (and so accumulate) only for batch-1 geometries (N == 1, the
 * spectral-norm power iteration) and gemm splits only when the call carries MMDGAN_ACT_FLAG_OUT_ZEROED.  Default 0. */
int mmdgan_set_outputs_prezeroed(int on);
/* Deferred slab reduction.  The Winograd-domain weight gradients (mmdgan_conv2d_wgrad* on 3x3 / stride-1 and 4x4 /
 * stride-2 geometries with a workspace) write one partial result per split of the pixel range into workspace slabs and sum
 * them in a second, bandwidth-only launch.  With on = 1 that launch is not issued: the NEXT such weight-gradient call on the
 * same stream sums the previous call's slabs in the prologue of its own kernel (every workgroup its 1/grid share; the two
 * calls' slabs lie in different parts of the workspace), so a chain of weight gradients - a backward pass - carries no
 * reduction launches except the last.  The sums are bit-identical in both modes.  Under on = 1 the outputs (dw, dbias,
 * dot_gw) of such a call are complete only once one of these has been issued behind it: the next mmdgan_conv2d_wgrad* call
 * (a slab one on the same stream carries the sum in its prologue, any other issues the stand-alone pass first), any other
 * workspace user of that stream, mmdgan_wgrad_flush(), or mmdgan_wgrad_defer(0) (which flushes).  A reader on another stream must be ordered behind that point.  Per handle; default 0.  (TF autodiff of
 * layer_func.py:914 is what the chain replaces.) */
int mmdgan_wgrad_defer(int on);
int mmdgan_wgrad_flush(void);

/* ------------------------------------------------------------------------------------------------
 * Launch plans: take the host out of a static step.  The reference's step is one `sess.run` of a fixed graph
 * (graph_func.py:851-854); here it is a fixed sequence of ~200 launches on a few streams.  Between mmdgan_plan_begin()
 * and mmdgan_plan_end() every entry of this library still EXECUTES normally and is also RECORDED - kernel, grid,
 * arguments by value, stream, plus the memsets and the stream dependencies below - so one ordinary step records itself.
 * mmdgan_plan_replay() then re-issues the recorded work from one C call: same streams, same overlap as the eager
 * issue, none of the per-entry host work (argument marshalling, dispatch, geometry / tile selection).
 *   - pointers are recorded by value: a replayed step reads and writes the SAME buffers (keep them allocated; feed
 *     new inputs by copying into them before the replay)
 *   - device-side state (Adam's step counter, spectral-norm vectors, the *_mix coin's averages) advances normally
 *   - work the library does not issue (an RCCL all-reduce through torch.distributed) is NOT recorded: close a segment
 *     with mmdgan_plan_mark() where it goes, and replay segment by segment with that work issued in between
 *   mmdgan_plan_mark    returns the index of the segment that starts at this point
 *   mmdgan_plan_replay  segment = -1: the whole plan; else segment 0 .. mmdgan_plan_segments()-1
 * Stream plumbing of a recorded step (what a framework's wait_stream / event / memset / copy would do un-recorded):
 *   mmdgan_stream_wait(w, s)   everything issued so far on stream s completes before anything issued later on stream w
 *   mmdgan_event_record(slot, s) / mmdgan_event_wait(slot, w)   the same, with the wait issued later than the record
 *                              (slot 0..63, per handle)
 *   mmdgan_memset_zero, mmdgan_memset_zero_multi, mmdgan_copy (device to device)
 * ---------------------------------------------------------------------------------------------- */
int mmdgan_plan_begin(void);
int mmdgan_plan_mark(void);
int mmdgan_plan_end(int *plan_id);
int mmdgan_plan_abort(void);                   /* drop a recording in progress (an entry failed) */
int mmdgan_plan_segments(int plan_id);         /* < 0: no such plan */
long mmdgan_plan_nodes(int plan_id);           /* recorded launches / memsets / dependencies; < 0: no such plan */
/* the kernel launches of a plan in issue order, one text line each: "<demangled kernel>\t<workgroups>\t<threads per
 * workgroup>\t<stream number, by first use>\n" - which kernels a recorded step consists of (the reference's step is one fixed
 * graph, graph_func.py:851-854: tests and the bench compare this list with a committed one).  Writes at most cap-1 bytes
 * plus a terminator into buf (buf may be NULL); returns the size a complete listing needs, < 0: no such plan. */
long mmdgan_plan_describe(int plan_id, char *buf, size_t cap);
int mmdgan_plan_replay(int plan_id, int segment);
int mmdgan_plan_destroy(int plan_id);
int mmdgan_stream_wait(void *waiting_stream, void *signalling_stream);
int mmdgan_event_record(int slot, void *stream);
int mmdgan_event_wait(int slot, void *stream);
int mmdgan_memset_zero(void *ptr, size_t bytes, void *stream);
/* n buffers zeroed by one launch (16-byte aligned sizes and pointers; anything else falls back to plain memsets) */
int mmdgan_memset_zero_multi(void *const *ptrs, const size_t *bytes, int n, void *stream);
int mmdgan_copy(void *dst, const void *src, size_t bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Gradient exchange (the reference's dormant tower helper graph_func.py:69-94 averages the towers'
 * gradients): RCCL from inside the library, one replica = one process = one GPU.  RCCL is bound at run time (dlopen):
 * without it these entries return MMDGAN_E_UNSUPPORTED and everything else works.
 *   mmdgan_comm_unique_id(out)        rank 0: 128 bytes to hand to the other ranks (any side channel)
 *   mmdgan_comm_init(id, nranks, rank) every rank: ncclCommInitRank
 *   mmdgan_allreduce_bucket(buf, count, stream)   in-place SUM over the replicas of `count` floats, asynchronous on
 *                                     `stream`; the mean is Adam's grad_scale = 1/nranks.  Recorded by launch plans like
 *                                     any other entry: a data-parallel step replays from one mmdgan_plan_replay call.
 * A caller may instead keep its own communicator (torch.distributed, as mmdgan_hip/engine.py does by default) and cut
 * its plan into segments around the collectives (mmdgan_plan_mark).
 * ---------------------------------------------------------------------------------------------- */
int mmdgan_comm_unique_id(void *out128);
int mmdgan_comm_init(const void *id128, int nranks, int rank);
int mmdgan_comm_size(void);                     /* 0 = no communicator */
int mmdgan_comm_destroy(void);
int mmdgan_allreduce_bucket(float *buf, size_t count, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution family.  Geometry: input [N,H,W,C], kernel [R,R,C,K], stride, 'SAME' padding with
 * pad_before = max((ceil(H/stride)-1)*stride + R - H, 0)/2 (tf.nn.conv2d, layer_func.py:914),
 * output [N,P,Q,K], P = ceil(H/stride).  1 <= R <= 7, 1 <= stride <= 7.  A kernel SMALLER than its stride is accepted
 * (a 1x1 stride-2 transposed conv is what a residual block on 'tc' has as its shortcut when its 'kernel' list says so,
 * layer_func.py:1725-1745): taps skip input pixels, the input-gradient leaves the pixels no tap reaches at act(bias);
 * those geometries run on the generic kernels (csrc/conv_direct.hip), the tiled families are built for R >= stride.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int N, H, W, C;    /* input  (NHWC) */
    int K;             /* output channels */
    int R;             /* square kernel size */
    int stride;
} mmdgan_conv_geom;

/* y = act(scale * conv(x, w) + bias)                       tf.nn.conv2d      layer_func.py:913-916
 * also the input-gradient of a transposed conv (with dact_of != NULL, see below).
 * If dact_of != NULL the epilogue is the BACKWARD form  y = scale*conv(x,w) * act'(dact_of)  where
 * dact_of holds the forward OUTPUT of the activation `act` at the same coordinates as y.
 * dact_batch (conv entries) / dact_rows (gemm): how many images / rows dact_of holds; 0 = as many as
 * the output.  If fewer, the trailing N - dact_batch output images reuse the LAST N - dact_batch
 * images of dact_of: the discriminator back-propagates its loss_dis rows (2B) and its loss_gen
 * rows (B, fake half only) in ONE 3B-row launch per layer. */
int mmdgan_conv2d_fwd(const mmdgan_conv_geom *g, const float *x, const float *w, const float *bias,
                      const float *scale, int act, const float *dact_of, int dact_batch, float *y, void *stream);

/* dx = scale * conv_input_grad(dy, w) [* act'(dact_of)]      autodiff of conv2d, my_sngan.py:302-304
 * Forward form (dact_of == NULL): dx = act(scale*conv_transpose(dy,w) + bias)
 *                                                          tf.nn.conv2d_transpose layer_func.py:926
 * g describes the CONV whose input-gradient this is: dy is [N,P,Q,K], dx is [N,H,W,C]. */
int mmdgan_conv2d_dgrad(const mmdgan_conv_geom *g, const float *dy, const float *w, const float *bias,
                        const float *scale, int act, const float *dact_of, int dact_batch, float *dx, void *stream);

/* The same two entries with a tensor added LAST - out = epilogue(conv) + addend, addend of the output's shape (16-byte aligned,
 * not the output buffer: a kernel without an epilogue for it gets an axpby pass behind it).  A residual block's branch sum  x + f(x)  (layer_func.py:1842) and the fan-in of the two
 * gradients that meet at a block's input ride on the launch that produces the second term instead of a pass of their own.
 * Kernels with an epilogue for it apply it in their stores (implicit GEMM, F(2x2,3x3), the slab pass of every split launch,
 * thin, direct); for the others the entry appends an axpby pass - the same result from every kernel.
 * Not with MMDGAN_ACT_FLAG_OUT_ZEROED (a split launch zeroes and accumulates into its output). */
int mmdgan_conv2d_fwd_add(const mmdgan_conv_geom *g, const float *x, const float *w, const float *bias, const float *scale,
                          int act, const float *dact_of, int dact_batch, const float *addend, float *y, void *stream);
int mmdgan_conv2d_dgrad_add(const mmdgan_conv_geom *g, const float *dy, const float *w, const float *bias, const float *scale,
                            int act, const float *dact_of, int dact_batch, const float *addend, float *dx, void *stream);

/* dw[R,R,C,K] = sum over pixels x (x) dy                   autodiff of conv2d / conv2d_transpose
 * dw is overwritten.  With a workspace registered the 3x3 / stride-1 (C % 32, K % 128) and 4x4 / stride-2 (C % 64,
 * K % 128) layers run in the Winograd domain with slab partial sums: no atomics, bit-reproducible. */
int mmdgan_conv2d_wgrad(const mmdgan_conv_geom *g, const float *x, const float *dy, float *dw, void *stream);
/* the same plus dbias[K] = column sums of dy (tf.nn.bias_add's gradient, layer_func.py:946): the MFMA kernel adds up
 * the dy tiles it streams anyway instead of a second pass over dy.  dbias is overwritten. */
int mmdgan_conv2d_wgrad_bias(const mmdgan_conv_geom *g, const float *x, const float *dy, float *dw, float *dbias,
                             void *stream);
/* the weight gradient of a SPECTRALLY NORMALISED kernel: as above (dbias may be NULL) plus dot_gw[0] = <dw, w>, the scalar of
 * the fix-up dL/dW = scale * dw - (scale / sigma) * <dw, w> * dsigma/dW (layer_func.py:884-887 + autodiff through
 * math_func.py:661-672).  dw stays the RAW gradient w.r.t. the scaled kernel: the fix-up is linear in it and
 * is applied where the gradient is consumed (mmdgan_adam_segments), after any data-parallel sum.  On the slab paths the
 * dot product is formed by the slabs' reduction pass (no extra launch); dot_gw is overwritten (accumulated into when
 * mmdgan_set_outputs_prezeroed(1) is in force). */
int mmdgan_conv2d_wgrad_sn(const mmdgan_conv_geom *g, const float *x, const float *dy, float *dw, float *dbias,
                           const float *w, float *dot_gw, void *stream);

/* Winograd F(2x2,3x3) for 3x3 / stride-1 layers (csrc/conv_wino.hip) and F(2x2,2x2) on the parity decomposition of
 * 4x4 / stride-2 layers and their transposes (csrc/conv_wino2.hip); same tf.nn.conv2d / conv2d_transpose / autodiff
 * call sites as above.
 * conv2d_fwd / conv2d_dgrad use it on their own when the geometry is eligible and a workspace is registered for
 * the transformed weights; a caller can instead transform once per weight update:
 *   mmdgan_wino_eligible(g, dgrad)        1 if conv2d_fwd (dgrad = 0) / conv2d_dgrad (1) of this geometry runs Winograd
 *   mmdgan_wino_weight_bytes(g)           size of the transformed tensor (16 * C * K floats for 3x3, 36 * C * K for 4x4)
 *   mmdgan_wino_transform(g, w, dgrad, u) u = G w G^T per (c,k), laid out for the forward (dgrad = 0) or the
 *                                         input-gradient (1: taps flipped, channel roles swapped)
 * and pass u as `w` with MMDGAN_ACT_FLAG_W_WINOGRAD. */
int mmdgan_wino_eligible(const mmdgan_conv_geom *g, int dgrad);
size_t mmdgan_wino_weight_bytes(const mmdgan_conv_geom *g);
int mmdgan_wino_transform(const mmdgan_conv_geom *g, const float *w, int dgrad, float *u, void *stream);
/* Round 6: a 3x3 / stride-1 geometry whose H and W are multiples of 4 can run F(4x4,3x3) (36 instead of 64 multiplies per
 * 4x4 outputs; layer_func.py:912-916) - another transformed-weight layout, so the caller says which one it holds:
 *   mmdgan_wino_algo(g, dgrad)                   MMDGAN_WINO_* the library prefers for conv2d_fwd / conv2d_dgrad of g
 *   mmdgan_wino_algo_weight_bytes(g, algo)       size of that algorithm's transformed tensor (0: algo does not fit g's kernel)
 *   mmdgan_wino_transform_algo(g, w, dgrad, algo, u)
 * and pass u as `w` with MMDGAN_ACT_FLAG_W_WINOGRAD (F23, F22S2) or MMDGAN_ACT_FLAG_W_WINOGRAD43 (F43).  The three entries
 * above keep their meaning (F23 for 3x3, F22S2 for 4x4 stride 2). */
int mmdgan_wino_algo(const mmdgan_conv_geom *g, int dgrad);
size_t mmdgan_wino_algo_weight_bytes(const mmdgan_conv_geom *g, int algo);
int mmdgan_wino_transform_algo(const mmdgan_conv_geom *g, const float *w, int dgrad, int algo, float *u, void *stream);
/* The weight gradient needs no transformed tensor from the caller (both of its operands are activations); which algorithm
 * mmdgan_conv2d_wgrad* takes for g WITH a workspace registered: MMDGAN_WINO_F43 (csrc/conv_wino43w.hip: 3x3 stride 1, H and W
 * multiples of 4, C and K of 32, from MMDGAN_WINO43_WGRAD_MIN_TILES 4x4 tiles on), _F23, _F22S2 or _NONE (implicit GEMM /
 * thin / direct kernels).  F(4x4,3x3) rounds at 1-4e-6 of the tensor's scale against F(2x2,3x3)'s 5-9e-7 (tools/wino43_gate.py):
 * a test that holds an identity to fp32 rounding asks here which floor applies.  (tf.gradients of layer_func.py:912-916.) */
int mmdgan_wgrad_algo(const mmdgan_conv_geom *g);
/* the same transform for MANY kernels in one launch (a training step re-transforms every eligible kernel of a network after
 * each weight update: one dispatch instead of one per kernel and form).  jobs is a HOST array, read during the call. */
typedef struct mmdgan_wino_job {
    const float *w;      /* HWIO kernel [R,R,C,K] */
    float *u;            /* mmdgan_wino_weight_bytes() of the geometry */
    int C, K, R, stride; /* R = 3 / stride 1 or R = 4 / stride 2 */
    int dgrad;           /* 0: forward form, 1: input-gradient form */
    int algo;            /* MMDGAN_WINO_* (0 = the default of the kernel size: F23 for 3x3, F22S2 for 4x4 stride 2) */
} mmdgan_wino_job;
int mmdgan_wino_transform_multi(const mmdgan_wino_job *jobs, int n_jobs, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Dense (tf.matmul, layer_func.py:909-911).  Row-major.  trans flags as in BLAS:
 *   C[M,N] = act(scale * op(A) op(B) + bias[N]),  op(A) is [M,K], op(B) is [K,N].
 * ---------------------------------------------------------------------------------------------- */
int mmdgan_gemm(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                const float *bias, const float *scale, int act, const float *dact_of, int dact_rows, float *C, int ldc,
                void *stream);

/* ------------------------------------------------------------------------------------------------
 * Column reductions / elementwise helpers
 * ---------------------------------------------------------------------------------------------- */
/* out[c] = sum_r x[r, c]  (bias gradient = column sum of the upstream gradient)  layer_func.py:946 */
int mmdgan_colsum(const float *x, long rows, int cols, float *out, void *stream);
/* out[0] = sum_i a[i]*b[i]  (double accumulation) */
int mmdgan_dot(const float *a, const float *b, long n, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Batch norm, training mode, NHWC / [rows, C]          tf.layers.batch_normalization
 *   layer_func.py:960-966: momentum .99, eps 1e-3, biased batch variance for normalisation,
 *   moving_variance updated with the unbiased variance when the input is 4-D (fused kernel).
 * fwd: y = act((x - mean)/sqrt(var+eps) * gamma + beta); writes save_mean/save_invstd [C] and
 *      new_moving_mean/var (may alias the inputs' storage: all reads precede all writes).
 * bwd: dx, dgamma, dbeta from dy (gradient w.r.t. y, i.e. AFTER the activation), y.  y may be NULL for act linear / relu /
 *      lrelu when beta is given: the derivative needs the SIGN of the forward value only, and the entry recomputes that value
 *      from x with the operations of the forward entry, bit for bit (a third less traffic than reading y back); beta is
 *      not read otherwise and may be NULL then.
 * workspace: mmdgan_bn_workspace_bytes(C) bytes of device scratch - fp64 totals [slots][2][C], accumulated with atomics,
 * the number of slots a function of C (up to 8): ALWAYS size it with mmdgan_bn_workspace_bytes, never as 2*C doubles.  The
 * entries zero it themselves, unless mmdgan_set_outputs_prezeroed(1) is in force: then it must be zero on entry and
 * distinct per call within a step (the forward and the backward call of a layer need separate totals).
 * bwd with y == NULL recomputes the forward value from x, gamma, beta, save_mean and save_invstd: those five must still hold
 * what the forward call read / wrote (no optimiser step on gamma / beta and no in-place change of x between the two calls).
 * ---------------------------------------------------------------------------------------------- */
size_t mmdgan_bn_workspace_bytes(int C);
/* A batch norm behind a convolution (layer_func.py:913-966): mmdgan_conv2d_fwd_stats / _dgrad_stats are mmdgan_conv2d_fwd /
 * _dgrad (no dact_of) that ALSO accumulate the per-channel sums of their output and of its squares into `bn_totals`
 * (mmdgan_bn_workspace_bytes(output channels) bytes; zeroed by the entry unless mmdgan_set_outputs_prezeroed(1)) - where the
 * launch's last pass writes the output (the slab pass of a reduction-split Winograd launch) the sums are formed there, otherwise
 * by the statistics pass of mmdgan_bn_fwd_train behind the convolution: the totals are the same for every geometry.
 * mmdgan_bn_fwd_apply is mmdgan_bn_fwd_train with those totals given: it normalises, applies the activation and updates the
 * moving statistics, and does not read x for the statistics again.  (version 500) */
int mmdgan_conv2d_fwd_stats(const mmdgan_conv_geom *g, const float *x, const float *w, const float *bias, const float *scale,
                            int act, float *y, void *bn_totals, void *stream);
int mmdgan_conv2d_dgrad_stats(const mmdgan_conv_geom *g, const float *dy, const float *w, const float *bias, const float *scale,
                              int act, float *dx, void *bn_totals, void *stream);
int mmdgan_bn_fwd_apply(const float *x, long rows, int C, const float *gamma, const float *beta, float eps,
                        float momentum, int unbiased_moving_var, int act, float *y, float *save_mean,
                        float *save_invstd, const float *moving_mean, const float *moving_var,
                        float *new_moving_mean, float *new_moving_var, void *workspace, void *stream);
int mmdgan_bn_fwd_train(const float *x, long rows, int C, const float *gamma, const float *beta, float eps,
                        float momentum, int unbiased_moving_var, int act, float *y, float *save_mean,
                        float *save_invstd, const float *moving_mean, const float *moving_var,
                        float *new_moving_mean, float *new_moving_var, void *workspace, void *stream);
int mmdgan_bn_fwd_infer(const float *x, long rows, int C, const float *gamma, const float *beta, float eps,
                        int act, const float *moving_mean, const float *moving_var, float *y, void *stream);
int mmdgan_bn_bwd(const float *x, const float *y, const float *dy, long rows, int C, const float *gamma,
                  const float *beta, const float *save_mean, const float *save_invstd, int act, float *dx,
                  float *dgamma, float *dbeta, void *workspace, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Spectral normalisation helpers (math_func.py:639-672).  The linear maps themselves are the
 * batch-1 conv / dgrad / gemm entries above; these are the norm steps between them.
 *   sn_norm:      out_norm[0] = ||v||_2 ; if v_normalised != NULL: v_normalised = v / (||v|| + 1e-10)
 *   sn_scale:     scale_out[0] = act_k / sigma[0]                      layer_func.py:886-887
 *   sn_wgrad_fixup: dw = scale*G - (scale/sigma) * <G,W> * dsigma_dw   (SURVEY A.2), in place on G.
 *                 `dot` is a device scalar holding <G,W>.
 * ---------------------------------------------------------------------------------------------- */
int mmdgan_sn_norm(const float *v, long n, float *out_norm, float *v_normalised, void *stream);
/* sn_norm + sn_scale in one launch: also writes scale_out[0] = act_k / ||v|| (scale_out may be NULL) */
int mmdgan_sn_norm_scale(const float *v, long n, float act_k, float *out_norm, float *scale_out, float *v_normalised,
                         void *stream);
int mmdgan_sn_scale(const float *sigma, float act_k, float *scale_out, void *stream);
int mmdgan_sn_wgrad_fixup(float *g_inout, const float *dsigma_dw, const float *dot, const float *sigma,
                          const float *scale, long n, void *stream);

/* One power-iteration step of MANY kernels at once (csrc/sn_chain.hip): every stage of all the chains is one launch - eight
 * per call for up to 8 kernels (more are processed in groups of 8) instead of five per kernel.  Per kernel, as the
 * reference's SpectralNorm (math_func.py:661-672, 739-744; the 'tc' branch :520-528 swaps the two conv forms):
 *     u = F(x);  sigma[0] = ||u||;  scale[0] = act_k / sigma[0];  un = u / (sigma + 1e-10)
 *     update != 0 (a training step's UPDATE_OPS):  dsigma = d sigma / d W  (the kernel's own layout),
 *                  xb = F^T(un);  xb_norm[0] = ||xb|| (if given);  x = xb / (||xb|| + 1e-10)   IN PLACE
 * form 0: F = conv2d_fwd with kernel w [R,R,C,K] ('SAME'): x [1,H,W,C], u / un [1,P,Q,K]
 * form 1: F = conv2d_dgrad: x [1,P,Q,K], u / un [1,H,W,C]                 (P, Q = ceil(H / stride), ceil(W / stride))
 * form 2: dense, w [C,K]: u [1,K] = x [1,C] w        form 3: dense: u [1,C] = x [1,K] w^T
 * (a conv kernel in 'sn_paper' mode is the dense form on its [R*R*C, K] view).  xb has x's shape.
 * col: scratch of 2 * P*Q * R*R*C floats per convolution kernel (the patch matrices), unused for the dense forms.
 * u, xb and dsigma are accumulated into, and so is the half of col that holds a PRODUCT (form 0: the second half, y W^T;
 * form 1: the first, x W^T) when that product has few tiles - ceil(P*Q / 64) * ceil(R*R*C / 64) < 128: the entry zeroes them
 * first unless mmdgan_set_outputs_prezeroed(1) says the caller did.  `layers` is a HOST array, read during the call. */
typedef struct mmdgan_sn_layer {
    const float *w;
    float *x, *u, *un, *xb, *col, *dsigma, *sigma, *scale, *xb_norm;
    float *norm_acc;     /* 4 floats (two doubles: the sums of squares of u and xb), 8-byte aligned; CONSECUTIVE over the layers of a call */
    float act_k;
    int form;
    int H, W, C, K, R, stride;
    float *x_out;        /* NULL: x is updated in place.  Else the normalised F^T(un) goes HERE and x is only read - with `sigma`
                          * pointed at a shadow too, a caller can run the iteration of step t+1 as soon as step t's optimiser
                          * has written W (beside the rest of step t) and commit x / sigma at the start of step t+1, so that
                          * both read between steps exactly as if the iteration had run inside step t+1 (math_func.py:661-672,
                          * 739-744: UPDATE_OPS semantics; version 500) */
} mmdgan_sn_layer;
int mmdgan_sn_power_iteration(const mmdgan_sn_layer *layers, int n_layers, int update, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Pairwise squared distances + Gaussian kernel + repulsive ('rep') / bounded ('rmb') MMD losses,
 * forward and backward in one launch (the B x B matrices never touch HBM).
 *   math_func.py: get_squared_dist :799-840, matrix_mean_wo_diagonal :1064, mmd_g :1312-1343,
 *   mmd_g_bounded :1380-1422, GANLoss._repulsive_mmd_g_(bounded_) :2505-2550.
 *   s_gen, s_x   [B,d]  discriminator scores of generated / real samples (x = gen, y = real)
 *   MMDGAN_LOSS_MMD_G / MGB (GANLoss._mmd_g_ :2160-2173 with mixture_mmd_g :1435-1462, _mmd_g_bound_ :2175-2193)
 *   run the same launch with their own kernel terms (w0, w1 ignored; e_k* of the mixture are summed over its five
 *   scales); MMDGAN_LOSS_HINGE / LOGISTIC (:2137-2143, :2128-2135) have no pairwise term: one small launch over
 *   the B*d scores, out_scalars = loss_gen, loss_dis, the two means of loss_dis, zeros; masks / dist must be NULL.
 *   out_scalars  [8]    loss_gen, loss_dis, e_kxx, e_kxy, e_kyy, e_kxx_b, e_kyy_b, e_kxy_b
 *   grads        [4,B,d] dLgen/ds_gen, dLgen/ds_x, dLdis/ds_gen, dLdis/ds_x   (NULL = forward only)
 *   masks        [3,B,B] bytes: dist_gg < lb, dist_gd > ub, dist_dd > ub       (NULL = skip)
 *   dist         [3,B,B] dist_gg, dist_gd, dist_dd                             (NULL = skip)
 *   workspace    mmdgan_mmd_workspace_bytes(B, d) bytes of device scratch
 * ---------------------------------------------------------------------------------------------- */
size_t mmdgan_mmd_workspace_bytes(int B, int d);
int mmdgan_mmd_loss(const float *s_gen, const float *s_x, int B, int d, int loss_type, float w0, float w1,
                    float lower_bound, float upper_bound, float *out_scalars, float *grads,
                    unsigned char *masks, float *dist, void *workspace, void *stream);

/* The `*_mix` losses: MMD with a coin that swaps generated and real scores between the two sets when the generator
 * loss has been above a threshold (math_func.py: get_mix_coin :2061-2085, slice_pairwise_distance :2038-2058,
 * moving_average_copy / moving_average_update :1979-2035, GANLoss._mmd_g_mix_ :2195-2228, _single_mmd_g_mix_ :2230-2263).
 *   loss_type     MMDGAN_LOSS_MMD_G_MIX (five-scale mixture) or MMDGAN_LOSS_SGM (sigma 1), | MMDGAN_LOSS_FLAG_GRADS_DIS_FIRST
 *   uni           [B] DEVICE floats: this step's tf.random_uniform([B], 0, 1) draw (:2079) - an input, so the
 *                 reference's boolean masks are reproducible bit for bit:  mix_indices = uni > mix_prob (:2080)
 *   mix_threshold 1.0 for mmd_g_mix, 0.2 for sgm in the reference (:2195, :2230)
 *   loss_average_update, mix_prob_update   both 0.01 in the reference (:2062)
 *   state         [2] DEVICE floats {loss_average ('coin/gen_average'), mix_prob ('coin/prob')}: read for this step,
 *                 then updated in place (the UPDATE_OPS; both right-hand sides use the pre-update values):
 *                   loss_average <- (1 - rho) loss_average + rho loss_gen
 *                   mix_prob     <- clip(mix_prob + rho (loss_average - mix_threshold), 0, 0.5)
 *   out_scalars   [8] loss_gen (un-mixed MMD), loss_dis (= -MMD between the mixed groups), e_kxx, e_kxy, e_kyy of the
 *                 un-mixed sets, loss_average and mix_prob as used (pre-update), number of true mix_indices
 *   grads         [4,B,d] as mmdgan_mmd_loss (NULL = forward only)
 *   masks         [5B] bytes: mix_indices [B], mix_group_1 = [idx, !idx] [2B], mix_group_2 = [!idx, idx] [2B]  (:2052-2053;
 *                 NULL = skip)
 *   workspace     mmdgan_mmd_mix_workspace_bytes(B, d) bytes, zero before the first call (B <= 8192) */
size_t mmdgan_mmd_mix_workspace_bytes(int B, int d);
int mmdgan_mmd_mix_loss(const float *s_gen, const float *s_x, int B, int d, int loss_type, const float *uni,
                        float mix_threshold, float loss_average_update, float mix_prob_update, float *state,
                        float *out_scalars, float *grads, unsigned char *masks, void *workspace, void *stream);

/* ------------------------------------------------------------------------------------------------
 * TF-semantics Adam over a list of tensors in one launch          graph_func.py:518-527
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 *   p -= lr_t * m / (sqrt(v) + eps)
 * ptrs is a DEVICE array of 4*n_tensors pointers (p, g, m, v per tensor), sizes a DEVICE array of
 * n_tensors element counts; grad_scale multiplies g first (1/world_size after a sum all-reduce).
 * Step count t: if step_counter (DEVICE int*) is non-NULL it is incremented on the device and used
 * (hipGraph-capturable: nothing host-computed changes between replays), else `step` is used.
 * lr_t_scratch: one DEVICE float of scratch.
 * ---------------------------------------------------------------------------------------------- */
int mmdgan_adam_multi(const void *const *ptrs, const long *sizes, int n_tensors, long max_size, float lr,
                      float beta1, float beta2, float eps, int step, int *step_counter, float *lr_t_scratch,
                      float grad_scale, void *stream);

/* The same update over ONE flat arena (params / grads / adam_m / adam_v share element offsets), cut into segments, with the
 * spectral-norm fix-up of mmdgan_conv2d_wgrad_sn folded into the gradient read: a segment with dsigma != NULL uses
 *     g_eff = grad_scale * (scale[0] * g - (scale[0] / sigma[0]) * dot[0] * dsigma)
 * (all DEVICE pointers; dsigma 16-byte aligned, n elements), any other segment g_eff = grad_scale * g.
 * segments_dev: DEVICE array of n_segments descriptors; blocks_dev: DEVICE array of n_blocks (segment index, block index
 * within the segment) int pairs, one per 1024 elements of every segment - the caller builds both once.
 * apply_fixup = 0: every segment is read plainly (the caller has applied mmdgan_sn_wgrad_fixup itself - what a data-parallel
 * replica does BEFORE its all-reduce: sigma and dsigma/dW carry the order of each replica's own atomics in their last bits,
 * and a fix-up applied after the sum would let the replicas' weights drift apart by an ulp per step).
 * Elements of the arena that no segment covers are left alone. */
typedef struct mmdgan_adam_segment {
    long off, n;
    const float *dsigma, *dot, *sigma, *scale;
} mmdgan_adam_segment;
int mmdgan_adam_segments(float *params, const float *grads, float *adam_m, float *adam_v,
                         const mmdgan_adam_segment *segments_dev, int n_segments, const int *blocks_dev, long n_blocks,
                         float lr, float beta1, float beta2, float eps, int step, int *step_counter, float *lr_t_scratch,
                         float grad_scale, int apply_fixup, void *stream);
/* The step-count half of either update on its own: advances the device counter (or takes `step`) and leaves the bias-corrected
 * learning rate in lr_t_scratch[0].  It depends on no gradient, so a caller can issue it early, off the critical path, and
 * then call mmdgan_adam_segments with step_counter = NULL and step = MMDGAN_ADAM_PREPARED (lr / betas of that call ignored
 * for the learning rate: lr_t_scratch is used as it is). */
#define MMDGAN_ADAM_PREPARED (-1)
int mmdgan_adam_prepare(float lr, float beta1, float beta2, int step, int *step_counter, float *lr_t_scratch, void *stream);
/* ... of several updates (G's and D's: graph_func.py:851-874 runs both optimisers in one sess.run) as ONE launch; the table
 * (host memory, n <= 8) travels by value. */
typedef struct {
    float lr, beta1, beta2;
    int step;                  /* used when step_counter is NULL */
    int *step_counter;
    float *lr_t_scratch;
} mmdgan_adam_prepare_job;
int mmdgan_adam_prepare_multi(const mmdgan_adam_prepare_job *jobs, int n, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise pieces of the residual blocks (layer_func.py:1687-1842), NHWC fp32.
 *   resample_down: y[n,p,q,:] (+)= scale * sum of the factor x factor window of x [N, P*factor, Q*factor, C].
 *                  scale = 1/factor^2 is ImageScaling 'avg' (:1155-1159); scale = 1 is the gradient of 'unpool'.
 *   resample_up:   y[n,h,w,:] (+)= scale * x[n, h/factor, w/factor, :], x [N,P,Q,C], y [N, P*factor, Q*factor, C].
 *                  scale = 1 is ImageScaling 'unpool' (:1160-1163, nearest-neighbour); scale = 1/factor^2 is the
 *                  gradient of 'avg'.
 *   act_fwd / act_bwd: a block's pre-activation on its input (:1785) and its gradient, taken from the
 *                  activation's output; dx (+)= dy * act'(y).
 *   axpby:         out = alpha*a + beta*b (the branch sum :1842 and gradient fan-in); out may alias a or b.
 * `accumulate` != 0 adds to what the output holds.
 * ---------------------------------------------------------------------------------------------- */
int mmdgan_resample_down(const float *x, float *y, int N, int P, int Q, int C, int factor, float scale, int accumulate,
                         void *stream);
int mmdgan_resample_up(const float *x, float *y, int N, int P, int Q, int C, int factor, float scale, int accumulate,
                       void *stream);
/* ImageScaling 'ps' (layer_func.py:197-244, 1125-1127): tf.depth_to_space (to_big = 1) / tf.space_to_depth (0) with
 * the reference's NCHW block-major channel order, on NHWC tensors:
 *   big[n, h*f + i, w*f + j, c] <-> small[n, h, w, (i*f + j)*C + c];  small is [N,H,W,f*f*C], big [N,H*f,W*f,C].
 * Each direction is the other's gradient. */
int mmdgan_periodic_shuffle(const float *src, float *dst, int N, int H, int W, int C, int factor, int to_big, void *stream);
/* ImageScaling 'bil' (layer_func.py:1128-1137): tf.image.resize_bilinear(align_corners=True), [N,H,W,C] -> [N,OH,OW,C]
 * (grad = 0), or its adjoint (grad = 1: src = dy [N,OH,OW,C], dst = dx [N,H,W,C], accumulated with atomics into a dx
 * the entry zeroes itself unless mmdgan_set_outputs_prezeroed(1) is in force). */
int mmdgan_bilinear_resize(const float *src, float *dst, int N, int H, int W, int C, int OH, int OW, int grad, void *stream);
/* ImageScaling 'bic' (layer_func.py:1138-1147): tf.image.resize_bicubic(align_corners=True) with TF 1.x's legacy sampling
 * (Keys cubic A = -0.75, fractional position rounded to 1/1024, taps clamped to the image; resize_bicubic_op.cc), same
 * arguments and gradient convention as mmdgan_bilinear_resize. */
int mmdgan_bicubic_resize(const float *src, float *dst, int N, int H, int W, int C, int OH, int OW, int grad, void *stream);
/* ImageScaling 'max' (layer_func.py:1149-1153): tf.nn.max_pool with window = stride = factor on x [N, P*factor, Q*factor, C].
 * dy == NULL: out [N,P,Q,C] = window maxima.  dy != NULL ([N,P,Q,C]): out [N, P*factor, Q*factor, C] = the gradient
 * w.r.t. x - dy at the first maximum of each window (row-major), zero elsewhere; every element of out is written. */
int mmdgan_max_pool(const float *x, const float *dy, float *out, int N, int P, int Q, int C, int factor, void *stream);
/* The scaling op of a residual block folded into the 3x3 stride-1 conv next to it (layer_func.py:1687-1842 order of ops):
 *   mode 0: avg-pool /2 AFTER the conv  = one 4x4 stride-2 conv;  dst [4,4,C,K] = 1/4 * (1_2 (*) w), for mmdgan_conv2d_fwd
 *   mode 1: 'unpool' x2 BEFORE the conv = one 4x4 stride-2 transposed conv; dst [4,4,K,C] (flipped, transposed), for the
 *           forward form of mmdgan_conv2d_dgrad
 * with w [3,3,C,K] the block's HWIO kernel.  grad = 1 is the adjoint: src = the gradient w.r.t. the 4x4 kernel (same
 * layout as dst above), dst [3,3,C,K] = the gradient w.r.t. w (overwritten). */
int mmdgan_compose_scaled_conv(const float *src, float *dst, int C, int K, int mode, int grad, void *stream);
/* The 'padding' and 'dilation' keys of a conv layer (layer_func.py:541-556, 912-916; math_func.py:172-193) as compositions
 * around the 'SAME' kernels above, NHWC:
 *   'VALID', kernel R, stride s, dilation d:  y = slice(conv_SAME_stride1(x)) with off = d * ((R-1)/2), step = s,
 *        P = ceil((H - (R-1) d) / s);  backward: the adjoint scatter (every element of the large tensor written), then the
 *        stride-1 input- / weight-gradient
 *   dilation d (stride 1, odd R):  y = from_batch(conv_SAME(to_batch(x))) on the d*d phase images of ceil(H/d) x ceil(W/d)
 * strided_slice: adjoint = 0: src [N,H,W,C] -> dst [N,P,Q,C]; adjoint = 1: src [N,P,Q,C] -> dst [N,H,W,C].
 * space_batch: to_batch = 1: src [N,H,W,C] -> dst [N*d*d, ceil(H/d), ceil(W/d), C] (zeros beyond the image); 0: the reverse. */
int mmdgan_strided_slice(const float *src, float *dst, int N, int H, int W, int C, int off, int step, int P, int Q,
                         int adjoint, void *stream);
int mmdgan_space_batch(const float *src, float *dst, int N, int H, int W, int C, int dilation, int to_batch, void *stream);
int mmdgan_act_fwd(const float *x, float *y, long n, int act, void *stream);
int mmdgan_act_bwd(const float *dy, const float *y, float *dx, long n, int act, int accumulate, void *stream);
int mmdgan_axpby(const float *a, float alpha, const float *b, float beta, float *out, long n, void *stream);

/* layout seam helpers: NCHW <-> NHWC (the reference API speaks NCHW, misc_fun.py:50-51) */
int mmdgan_nchw_to_nhwc(const float *src, float *dst, int N, int C, int H, int W, void *stream);
int mmdgan_nhwc_to_nchw(const float *src, float *dst, int N, int C, int H, int W, void *stream);
/* a batch of uint8 image records -> fp32 NHWC in [-1,1], the reference's input preprocessing
 * (input_func.py:797-801 decode_raw + cast, :839-842 image / 127.5 - 1, reshape (channels, height, width)).
 * src: N records of C*H*W bytes on the DEVICE, stored [C,H,W] (src_is_chw = 1, how the reference's converters
 * write them) or [H,W,C] (0).  Bit-exact: IEEE fp32 divide and subtract. */
int mmdgan_u8_records_to_nhwc(const unsigned char *src, int src_is_chw, float *dst, int N, int C, int H, int W,
                              void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MMDGAN_HIP_H */
