/*
 * exposure_hip.h -- C-ABI of libexposure_hip.so: the MI355X (gfx950) implementation
 * of Exposure's differentiable per-pixel filter stack.
 *
 * The reference (yuanming-hu/exposure, TF-1 Python) has NO FFI for this path: the
 * filters are Python classes composed of TF elementwise ops.  This header is the
 * boundary a native replacement of that path binds; each entry point cites the
 * reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - Images are contiguous NHWC, C = 3, dtype EXPO_F16 (default) or EXPO_F32.
 *   - Every pointer is a caller-owned DEVICE pointer that must stay valid until the
 *     work enqueued on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream) has completed.  No allocation, no ownership transfer, no host sync.
 *   - Stateless and re-entrant; ordering only through `stream`.
 *   - Reduction workspace.  Every entry point that produces per-image sums (parameter
 *     gradients, penalty, statistics) takes `workspace` / `workspace_bytes`: a caller-owned,
 *     4-byte aligned device scratch buffer of at least expo_workspace_bytes(n, h, w, dtype)
 *     bytes (expo_chain_bwd: that many bytes PER STEP).  It needs no initialisation and
 *     carries nothing from call to call: the streaming kernel leaves one record of partial
 *     sums per block in it and a small finish launch enqueued behind it adds an image's
 *     records in a fixed order -- no float atomics, no zero-fill launch, bit-reproducible
 *     results.  One workspace serves one stream at a time (calls that may run concurrently on
 *     different streams need their own).
 *   - Filter ids follow cfg.filters (config_example.py:22-25):
 *       0 E  ExposureFilter              P = 1   params[n][0] = EV stops
 *       1 G  GammaFilter                 P = 1   gamma
 *       2 W  ImprovedWhiteBalanceFilter  P = 3   per-channel scale
 *       3 S+ SaturationPlusFilter        P = 1   blend in [0,1]
 *       4 T  ToneFilter                  P = 8   curve slopes k_0..k_7
 *       5 Ct ContrastFilter              P = 1   blend in [-1,1]
 *       6 BW WNBFilter                   P = 1   blend in [0,1]
 *       7 C  ColorFilter                 P = 24  k[channel*8 + knot]
 *       8 Le LevelFilter                 P = 2   lower, upper-1   (not in cfg.filters)
 *     `params` are the outputs of Filter.filter_param_regressor (filters.py:177-179,
 *     201-203, 224-235, 481-482, 306-310, 411-413, 435-436, 256-262) flattened to
 *     float32 [N][P].
 *   - All functions return EXPO_OK (0) or a negative EXPO_E_* code; the message of
 *     the last failure on the calling thread is available from expo_last_error().
 */
#ifndef EXPOSURE_HIP_H_
#define EXPOSURE_HIP_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXPO_ABI_VERSION 7 /* 2: caller-owned reduction workspace (no float atomics, no fills); 3: derivatives of the
                              critic statistics and of the penalty, VignetFilter, bias + lrelu, masked per-image dispatch;
                              4: Tone / Color curves of any cfg.curve_steps (expo_curve_*); 5: the convnets' convolution
                              (expo_conv4x4s2_*); 6: its mask / bias variants, the hand-scheduled critic update's
                              kernels, expo_build_info; 7: expo_net_inputs, the first FC layer with its K dimension split
                              (expo_fc_*; expo_critic_head_fwd / _bwd take the partial sums) */

#define EXPO_OK 0
#define EXPO_E_BADARG (-1)
#define EXPO_E_BADDTYPE (-2)
#define EXPO_E_HIP (-3)

#define EXPO_F16 0
#define EXPO_F32 1

#define EXPO_NUM_FILTERS 9
#define EXPO_MAX_PARAMS 24
#define EXPO_MAX_HEADS 16 /* FC heads one expo_heads_regress_* call serves (cfg.filters has 8) */
#define EXPO_CURVE_MAX_STEPS 16 /* largest cfg.curve_steps of expo_curve_fwd / _bwd */

#define EXPO_FILTER_EXPOSURE 0
#define EXPO_FILTER_GAMMA 1
#define EXPO_FILTER_WB 2
#define EXPO_FILTER_SATPLUS 3
#define EXPO_FILTER_TONE 4
#define EXPO_FILTER_CONTRAST 5
#define EXPO_FILTER_WNB 6
#define EXPO_FILTER_COLOR 7
#define EXPO_FILTER_LEVEL 8 /* LevelFilter (filters.py:449-466): defined by the reference, not in cfg.filters */

/* hsv_grad_mode for SaturationPlus backward: 0 = TF-1.x faithful (RGBToHSV / HSVToRGB
 * are registered NotDifferentiable, so no gradient flows through full_color);
 * 1 = analytic gradient through the HSV round trip. */
#define EXPO_HSV_GRAD_TF1 0
#define EXPO_HSV_GRAD_ANALYTIC 1

/* Library / ABI version (EXPO_ABI_VERSION of the built library). */
int expo_version(void);

/* Message of the last error raised on the calling thread ("" if none). */
const char* expo_last_error(void);

/* (ABI 6) The sha256 (hex) of the sources this binary was built from -- csrc/build.sh hashes include/exposure_hip.h and
 * the .hip / .h files and build.sh of csrc/ (byte order of their names) and passes the digest to the compiler.  The
 * Python binding recomputes it from the tree and refuses a stale binary (EXPO_ALLOW_STALE_LIB=1 overrides); "unknown"
 * for a build that bypassed build.sh. */
const char* expo_build_info(void);

/* Replaces Filter.get_num_filter_parameters() (filters.py:24-26); -1 on bad id. */
int expo_num_filter_params(int filter_id);

/*
 * Bytes of reduction workspace ONE reducing call needs for images of this shape (the maximum
 * over the entry points; expo_chain_bwd needs `steps` times as much).  0 on invalid arguments.
 * New with ABI 2; the reference has no counterpart (TF allocates its reduction scratch itself).
 */
size_t expo_workspace_bytes(int n, int h, int w, int dtype);

/*
 * y = <Filter>.process(x, params)      replaces filters.py:181-182 (E), 205-206 (G),
 * 237-238 (W), 484-498 (S+), 312-322 (T), 415-419 (Ct), 438-440 (BW), 264-273 (C).
 * With cfg.masking = False (config_example.py:36) this is also Filter.apply's
 * lerp(img, process(img, p), ones) (filters.py:88, 111-113).
 * x, y: [N][H][W][3] of dtype; y may alias x.  params: float32 [N][P].
 * Any H, W >= 1 (the same parameters may be applied to the 64x64 proxy and to the
 * full-resolution image: filters.py:89-96).
 */
int expo_filter_fwd(int filter_id, const void* x, void* y, const float* params,
                    int n, int h, int w, int dtype, void* stream);

/*
 * Backward of expo_filter_fwd: what tf.gradients produces for process() in the
 * reference graph (net.py:222-251 via ly.optimize_loss).
 *   dx      [N][H][W][3] dtype, may be NULL (parameter gradient only), may alias dy.
 *   dparams float32 [N][P], OVERWRITTEN with sum_{h,w,c} dy * d y/d param.
 * y is recomputed from x (never read).
 */
int expo_filter_bwd(int filter_id, const void* x, const void* dy, void* dx,
                    const float* params, float* dparams, int n, int h, int w,
                    int dtype, int hsv_grad_mode, void* workspace, size_t workspace_bytes,
                    void* stream);

/*
 * Same as expo_filter_bwd but dparams is ACCUMULATED into (dparams += ...): the caller owns
 * the initial value.  This is gradient accumulation over several
 * backward calls that share a parameter tensor (e.g. the low-resolution and the high-resolution
 * application of one filter, filters.py:88-96, or micro-batches).
 */
int expo_filter_bwd_accumulate(int filter_id, const void* x, const void* dy, void* dx,
                               const float* params, float* dparams, int n, int h, int w,
                               int dtype, int hsv_grad_mode, void* workspace,
                               size_t workspace_bytes, void* stream);

/*
 * Filter.apply with the spatial mask enabled (cfg.masking = True; filters.py:62-99, 110-148):
 *   out = (1 - mask) * x + mask * process(x, params)
 *   mask = sigmoid((gx*m0 + gy*m1 + m2*(lum(x) - .5) + 2*m3) * maximum_sharpness * m4 / 5)
 *          * (m5/5 * .5 + .5) * (1 - minimum_strength) + minimum_strength
 * mask_params: float32 [N][6] = tanh_range(-5, 5)(raw mask parameters) -- the range squashing
 * (filters.py:121-123) stays with the caller so its gradient is handled by the host autograd.
 * gx, gy: the constant coordinate grid of filters.py:124-133.  cfg.maximum_sharpness and
 * cfg.minimum_strength are config_example.py:37-38.
 */
int expo_filter_apply_fwd(int filter_id, const void* x, void* y, const float* params,
                          const float* mask_params, float maximum_sharpness, float minimum_strength,
                          int n, int h, int w, int dtype, void* stream);

/* Backward of expo_filter_apply_fwd; dparams [N][P] and dmask_params [N][6] are overwritten. */
int expo_filter_apply_bwd(int filter_id, const void* x, const void* dy, void* dx,
                          const float* params, float* dparams, const float* mask_params,
                          float* dmask_params, float maximum_sharpness, float minimum_strength,
                          int n, int h, int w, int dtype, int hsv_grad_mode, void* workspace,
                          size_t workspace_bytes, void* stream);

/*
 * The agent's step with cfg.masking = True (agent.py:58-77, 119-125 around filters.py:62-99, 110-148): the reference runs
 * EVERY filter's masked apply on the whole batch, stacks the 8 results and reduces with one_hot(selected_filter_id).
 * The one-hot zeroes the seven others (values and gradients), so per image only the selected filter's masked apply runs:
 *   y[n] = expo_filter_apply_fwd(filter_ids[n], x[n], params[n], mask_params[n])      filter_ids[n] = -1 -> y[n] = 0
 * params / dparams: float32 [N][EXPO_MAX_PARAMS] (row = packed parameters of filter filter_ids[n], rest ignored / 0);
 * mask_params / dmask_params: float32 [N][6] = tanh_range(-5, 5)(raw mask parameters of THAT filter).
 */
int expo_filter_apply_dispatch_fwd(const int32_t* filter_ids, const void* x, void* y, const float* params,
                                   const float* mask_params, float maximum_sharpness, float minimum_strength,
                                   int n, int h, int w, int dtype, void* stream);
int expo_filter_apply_dispatch_bwd(const int32_t* filter_ids, const void* x, const void* dy, void* dx,
                                   const float* params, float* dparams, const float* mask_params,
                                   float* dmask_params, float maximum_sharpness, float minimum_strength, int n,
                                   int h, int w, int dtype, int hsv_grad_mode, void* workspace,
                                   size_t workspace_bytes, void* stream);

/*
 * Per-image filter choice == the reference's "compute all 8 filters, stack, multiply
 * by one_hot(selected_filter_id), reduce_sum" (agent.py:58-77, 119-125) without the
 * 7 discarded outputs.  filter_ids: int32 [N] in [-1, 8]; -1 (pdf_sample with noise
 * 0, pdf_sample_layer.py:5-10) selects nothing: y = 0, all gradients 0.
 * params / dparams: float32 [N][EXPO_MAX_PARAMS]; row n holds the P values of filter
 * filter_ids[n] in its first P slots (remaining slots ignored / written as 0).
 * penalty (nullable): float32 [N], overwritten with mean_{h,w,c} max(y - 1, 0)^2,
 * the over-exposure term of agent.py:249-251, fused into the same pass (workspace needed
 * only when penalty != NULL).
 */
int expo_filter_dispatch_fwd(const int32_t* filter_ids, const void* x, void* y,
                             const float* params, float* penalty, int n, int h, int w,
                             int dtype, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Backward of expo_filter_dispatch_fwd.  dpenalty (nullable): float32 [N], upstream
 * gradient of the fused penalty; its contribution 2*max(y-1,0)/(H*W*3)*dpenalty[n]
 * is added to dy before back-propagating through the selected filter.
 */
int expo_filter_dispatch_bwd(const int32_t* filter_ids, const void* x, const void* dy,
                             void* dx, const float* params, float* dparams,
                             const float* dpenalty, int n, int h, int w, int dtype,
                             int hsv_grad_mode, void* workspace, size_t workspace_bytes,
                             void* stream);

/*
 * The two halves of expo_filter_bwd, for callers that finish several passes with one launch
 * (expo_chain_bwd is exactly: the records pass of every step, then one expo_finish_bwd) and for
 * per-kernel timing.  expo_filter_bwd_records runs the streaming pass alone: dx is written, the
 * per-block partial sums stay in `records` (>= expo_workspace_bytes(n, h, w, dtype) bytes), dparams
 * is not touched.  expo_finish_bwd turns `steps` record slices -- slice k at byte offset
 * k * expo_workspace_bytes(...) of `workspace` -- into the parameter gradients dparams[k] of the
 * filters filter_ids[k] (host arrays, as in expo_chain_bwd).  No reference counterpart: TF fuses
 * nothing here (net.py:222-251 builds one reduction per parameter tensor).
 */
int expo_filter_bwd_records(int filter_id, const void* x, const void* dy, void* dx,
                            const float* params, int n, int h, int w, int dtype,
                            int hsv_grad_mode, void* records, size_t records_bytes, void* stream);
int expo_finish_bwd(const int* filter_ids, int steps, const float* const* params,
                    float* const* dparams, int n, int h, int w, int dtype, void* workspace,
                    size_t workspace_bytes, void* stream);

/*
 * The benchmark construct of SURVEY.md section 8(d): `steps` filters applied
 * sequentially, one kernel per step, enqueued by a single call.
 *   filter_ids  host int [steps]
 *   acts        host array of steps+1 device image pointers: acts[0] = input,
 *               acts[i+1] = output of step i (kept: the backward reads them)
 *   params      host array of `steps` device pointers, float32 [N][P_i]
 *
 * Images are independent, so for batches whose tensors are between 40 and 256 MiB the chain entry points split the
 * batch in two halves and run them on two streams -- `stream` and a library-owned helper stream, forked and joined
 * with events INSIDE the call (one half's kernels fill the ramp / tail bubbles of the other's 17 dependent launches:
 * -3.5 % per step at 64x512x512).  Nothing changes for the caller: all work is ordered after what `stream` held
 * before the call and before what it receives afterwards, and the pattern is capturable into a hipGraph.
 * expo_chain_streams() returns the number of streams (1 or 2) a shape gets (EXPO_CHAIN_STREAMS=1|2 overrides).
 *
 * The helper is chosen per (device, caller stream) and must sit on another hardware queue than `stream`: the pairing
 * is probed ONCE per caller stream (two one-wave kernels, < 0.1 ms; a rejected helper costs 0.5 ms -- the kernel on
 * `stream` waits that long for the helper's -- at most 4 candidates; both streams are synchronised), see
 * exposure_hip.hip.  By default the first EAGER two-stream chain call of a caller stream does that; an integrator who
 * does not want a stall inside a data-path call runs it at set-up time with expo_chain_prepare(stream) (idempotent;
 * nothing to do under stream capture, where the graph executor assigns the queues).  EXPO_CHAIN_HELPER_PROBE=0 in the
 * environment switches the probe off altogether (the first helper stream is taken as it comes).  Only callers on the
 * SAME stream wait for a probe.  expo_chain_release(stream) forgets a pairing and destroys its helper stream -- call it
 * before hipStreamDestroy of a stream that made chain calls (where the runtime has hipStreamGetId -- HIP >= 7.1 -- a destroyed
 * stream whose handle is reused is recognised and probed again either way).  Rejected helper streams stay parked (at most 8 per process) so
 * that their hardware queue is not handed out again.  expo_chain_helper_stats() reports how many pairings this
 * process probed and how many helpers it rejected (diagnostics; either pointer may be NULL).
 */
int expo_chain_streams(int n, int h, int w, int dtype);
int expo_chain_helper_stats(int* probed, int* rejected);
int expo_chain_prepare(void* stream);
int expo_chain_release(void* stream);
int expo_chain_fwd(const int* filter_ids, int steps, void* const* acts,
                   const float* const* params, int n, int h, int w, int dtype,
                   void* stream);

/*
 * Backward of expo_chain_fwd.
 *   grads    host array of steps+1 device image pointers: grads[steps] = upstream
 *            dy (read), grads[i] = gradient w.r.t. acts[i] (written).  Entries may
 *            ping-pong between two buffers as long as grads[i] != grads[i+1] is not
 *            required (dx may alias dy).
 *   dparams  host array of `steps` device pointers, float32 [N][P_i], overwritten.
 *   workspace_bytes >= steps * expo_workspace_bytes(n, h, w, dtype): every step keeps its block
 *   records until the ONE finish launch at the end of the chain has produced all dparams.
 */
int expo_chain_bwd(const int* filter_ids, int steps, void* const* acts,
                   void* const* grads, const float* const* params,
                   float* const* dparams, int n, int h, int w, int dtype,
                   int hsv_grad_mode, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Fused multi-step forward for the high-resolution inference path (net.py:796-821, evaluate.py:8-31):
 * the agent regresses every step's parameters on the 64x64 proxy, so the per-image sequence
 * (filter id, parameters) x steps is known before the full-resolution image is touched.  All
 * `steps` filters are applied while a pixel group sits in registers (fp32 between steps): one
 * read and one write of the image instead of `steps` of each.
 *   filter_ids  device int32 [N][steps] in [-1, 8]   (-1: the image becomes 0 from that step on)
 *   params      device float32 [N][steps][EXPO_MAX_PARAMS], row = packed params of that step's filter
 */
int expo_chain_fused_fwd(const int32_t* filter_ids, const float* params, int steps,
                         const void* x, void* y, int n, int h, int w, int dtype, void* stream);

/*
 * One-pass backward of the same fixed per-image sequence: dx = d(loss)/dx and every step's parameter
 * gradients from x and dy = d(loss)/dy alone -- ONE read of x and dy, ONE write of dx (18 B/pixel
 * whatever the number of steps); the activations are recomputed in registers and no intermediate
 * image exists.  A benchmark construct (bench.py --workload chain_fused): the reference has no caller
 * for it, because there the parameters of step k+1 depend on the image after step k through the CNN
 * (agent.py:30-125) and the backward has to be step by step (expo_chain_bwd / expo_filter_bwd).
 * Each step is linearised at its input rounded to the storage dtype (exact for fp32 storage; for fp16
 * storage the value the per-step chain stores between its launches); the gradient travels between the
 * steps in fp32.  Tie conventions, hsv_grad_mode and id -1 (the image becomes 0: no gradient reaches
 * the input or the earlier steps, that step's dparams row is 0) as in expo_filter_bwd / the dispatch.
 *   filter_ids  device int32 [N][steps] in [-1, 8],  steps <= EXPO_FUSED_BWD_MAX_STEPS
 *   params      device float32 [N][steps][EXPO_MAX_PARAMS]
 *   dparams     device float32 [N][steps][EXPO_MAX_PARAMS], fully overwritten (unused tail of a row = 0)
 *   dx may alias dy (not x).  workspace: expo_workspace_bytes(...) * steps, as for expo_chain_bwd.
 */
#define EXPO_FUSED_BWD_MAX_STEPS 8
int expo_chain_fused_bwd(const int32_t* filter_ids, const float* params, int steps, const void* x,
                         const void* dy, void* dx, float* dparams, int n, int h, int w, int dtype,
                         int hsv_grad_mode, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Per-image statistics the critic appends as constant feature planes
 * (critics.py:48-62): stats[n] = { mean(lum), variance(lum), mean(sat) } with
 * lum = .27R + .67G + .06B + 1e-5 and
 * sat = (max - min) / (min(max + min, 2 - max - min) + 1e-2) on the [0,1]-clipped image.
 * stats: float32 [N][3], overwritten.
 */
int expo_critic_stats(const void* x, float* stats, int n, int h, int w, int dtype,
                      void* workspace, size_t workspace_bytes, void* stream);

/*
 * mean_{h,w,c} max(y - 1, 0)^2 per image (agent.py:249-251). penalty: float32 [N].
 */
int expo_overexposure_penalty(const void* y, float* penalty, int n, int h, int w,
                              int dtype, void* workspace, size_t workspace_bytes, void* stream);

/*
 * VignetFilter.apply (filters.py:341-396; defined by the reference, in no cfg.filters).  process() is img * 0
 * (filters.py:351-352), so the filter IS its mask:  y = x * (1 - mask),
 *   mask = sigmoid(((gx*m0)^2 + (gy*m1)^2 + m2 - 5) * maximum_sharpness * m3 / 5) * (m4/5 * .5 + .5)
 * with gx, gy the constant coordinate grid of filters.py:371-380 (the one expo_filter_apply_fwd uses).
 * mask_params: float32 [N][5] = tanh_range(-5, 5)(raw mask parameters) (the squashing stays with the caller, as
 * for expo_filter_apply_fwd).  masking = 0 reproduces cfg.masking = False: mask forced to 1 (filters.py:390-392),
 * i.e. y = 0, dx = 0, dmask_params = 0.  The backward overwrites dx (nullable) and dmask_params [N][5].
 */
int expo_vignet_apply_fwd(const void* x, void* y, const float* mask_params, float maximum_sharpness, int masking,
                          int n, int h, int w, int dtype, void* stream);
int expo_vignet_apply_bwd(const void* x, const void* dy, void* dx, const float* mask_params,
                          float* dmask_params, float maximum_sharpness, int masking, int n, int h, int w,
                          int dtype, void* workspace, size_t workspace_bytes, void* stream);

/*
 * d penalty / d y of expo_overexposure_penalty (what tf.gradients produces for agent.py:249-251 when the
 * penalty is NOT taken from the fused dispatch pass -- cfg.masking or cfg.clamp, agent.py:240-241):
 *   dy[n] = 2 * max(y[n] - 1, 0) * dpenalty[n] / (H*W*3).      dy: [N][H][W][3] dtype, overwritten.
 */
int expo_overexposure_penalty_bwd(const void* y, const float* dpenalty, void* dy, int n, int h, int w,
                                  int dtype, void* stream);

/*
 * Derivatives of expo_critic_stats.  The statistics sit inside the training graph of the reference: the
 * generator's reward flows through critic(fake_output) (net.py:68-90) and the WGAN-GP term takes
 * tf.gradients(inte_logit, [interpolated]) and then differentiates THAT with respect to the critic's weights
 * (net.py:174-194), i.e. a double backward through critics.py:48-73.  With S = the three statistics of one
 * image and J = dS/dx (3 x H*W*3):
 *   expo_critic_stats_bwd   dx  = J^T dstats              first derivative (tf.gradients of critics.py:48-62)
 *   expo_critic_stats_jvp   jv  = J v                     d <dx, v> / d dstats: the path of the double backward
 *                                                         that reaches the critic's weights
 *   expo_critic_stats_hvp   out = d <J^T dstats, v> / dx  the second-order term in the image itself
 * stats: float32 [N][3] as written by expo_critic_stats for the SAME x (the mean luminance is read from it);
 * dstats, jv: float32 [N][3]; v, dx, out: images of x's dtype (the gradients are O(1/(H*W)): use EXPO_F32 images
 * when the values matter -- exposure_amd/critics.py always does).  TF conventions: population variance
 * (tf.nn.moments), reduce_max / reduce_min split the gradient evenly between tied channels, clip_by_value
 * passes on 0 <= x <= 1 inclusive, tf.minimum(x, y) sends ties to x.
 */
int expo_critic_stats_bwd(const void* x, const float* stats, const float* dstats, void* dx, int n, int h,
                          int w, int dtype, void* stream);
int expo_critic_stats_jvp(const void* x, const float* stats, const void* v, float* jv, int n, int h, int w,
                          int dtype, void* workspace, size_t workspace_bytes, void* stream);
int expo_critic_stats_hvp(const void* x, const float* dstats, const float* jv, const void* v, void* out,
                          int n, int h, int w, int dtype, void* stream);

/*
 * The activation of the convnets around the filter path -- lrelu(x) = f1*x + f2*|x| with leak 0.2 (util.py:225-229),
 * used after every ly.conv2d / ly.fully_connected of feature_extractor (agent.py:21-32), cnn (critics.py:13-35) and
 * the FC heads (filters.py:31-42, agent.py:87-99, critics.py:94-97) -- fused with the bias add in front of it.
 *   expo_bias_lrelu_fwd   z[i] = lrelu(y[i] + bias[i % channels])     bias may be NULL (then channels is ignored)
 *   expo_lrelu_bwd        dy[i] = dz[i] * (z[i] > 0 ? 1 : z[i] < 0 ? leak : (1 + leak) / 2)
 * float32, `count` contiguous elements, the channel is the fastest dimension (NHWC conv outputs, (N, C) FC
 * outputs); z may alias y, dy may alias dz.  The slope is taken from the OUTPUT z (lrelu keeps sign and zero);
 * exactly at 0 it is TF's sub-gradient f1 (tf.abs has gradient 0 at 0).  expo_lrelu_bwd is linear in dz and
 * therefore its own double backward (net.py:174-194).
 */
int expo_bias_lrelu_fwd(const float* y, const float* bias, float* z, size_t count, int channels, float leak,
                        void* stream);
int expo_lrelu_bwd(const float* z, const float* dz, float* dy, size_t count, float leak, void* stream);
/* (ABI 4) The same backward WITH the layer's bias gradient: dy as above and dbias[c] = sum over rows of dy[row][c]
 * (what tf.gradients returns for the biases of ly.conv2d / ly.fully_connected: agent.py:21-32, critics.py:13-35) in one
 * pass over dz instead of a second reduction pass over dy.  channels: a power of two in [4, 256]; count a multiple of
 * channels; 16-byte aligned pointers; dbias [channels] is overwritten; workspace >= expo_lrelu_bwd_bias_workspace_bytes(
 * channels) bytes, 16-byte aligned, uninitialised.  No atomics; bit-reproducible. */
size_t expo_lrelu_bwd_bias_workspace_bytes(int channels);
int expo_lrelu_bwd_bias(const float* z, const float* dz, float* dy, float* dbias, size_t count, int channels,
                        float leak, void* workspace, size_t workspace_bytes, void* stream);

/*
 * (ABI 4) The tail of the agent's FC heads: every filter's `filter_param_regressor` (filters.py:177-179 E, 201-203 G,
 * 224-235 W, 481-482 S+, 306-310 T, 411-413 Ct, 435-436 BW, 256-262 C, 457-458 Le) followed by the one-hot selection of
 * the chosen filter's parameters (agent.py:58-77, 119-125), per image, in one launch:
 *   params[n][0..P) = regressor_{abi_ids[j]}(raw[j][n][0..P)),  j = selected[n];  the rest of the row, and the whole row
 *   for j = -1, is 0.
 * raw: HOST array of `heads` device pointers, head j holding float32 [n][widths[j]] (its second FC's output: P_j filter
 * features followed by the mask features); abi_ids[j]: the C-ABI filter id of head j; selected: device int32 [n]
 * (position in cfg.filters, -1 = none); params / dparams: float32 [n][EXPO_MAX_PARAMS]; ranges: HOST float[9] =
 * {exposure_range, log(gamma_range), tone lo, tone hi, tone bias, colour lo, colour hi, colour bias, exposure bias}
 * (cfg.exposure_range, cfg.gamma_range, cfg.tone_curve_range, cfg.color_curve_range; the biases are util.py:281-294's
 * atanh(2 (initial - l) / (r - l) - 1), 0 for every shipped range).  The white-balance log range is the reference's
 * constant 0.5.  expo_heads_regress_bwd writes d raw[j] for EVERY head (zeros outside the selected head's slice).
 * 8-step curves only (cfg.curve_steps = 8: the rows hold EXPO_MAX_PARAMS values).
 */
int expo_heads_regress_fwd(const float* const* raw, const int* widths, const int* abi_ids, int heads,
                           const float* ranges, const int32_t* selected, float* params, int n, void* stream);
int expo_heads_regress_bwd(const float* const* raw, float* const* draw, const int* widths, const int* abi_ids,
                           int heads, const float* ranges, const int32_t* selected, const float* dparams, int n,
                           void* stream);

/*
 * (ABI 4) Action selection, state update and the image-independent part of the penalty (agent.py:87-125, 207-252;
 * pdf_sample_layer.py:5-10), per image, one launch:
 *   pdf = softmax(logits) + 1e-37;  pdf = pdf (1 - exploration) + exploration / K;  pdf /= sum(pdf) + 1e-30
 *   entropy = -sum pdf log pdf;  random id = #(exclusive cumsum(pdf / (rowsum(pdf) + 1e-36)) < noise) - 1 with the
 *   explicit association orders of the reference's CPU kernels (K = 8: ((p0+p4)+(p2+p6))+((p1+p5)+(p3+p7)); the scan
 *   left to right) -- noise 0 gives -1;  selected = is_train ? random id : argmax(pdf);  onehot;
 *   surrogate = log(pdf[selected] + 1e-10) (0 for -1);  new_states = [submitted, submitted, step + 1, max(usage, onehot)],
 *   submitted = |step + 1 - test_steps| < 1e-4;  penalty_base = (1 - progress) c_e (log K - entropy) + <usage, onehot> c_u
 *   + (1 - submitted) submitted c_s.
 * logits [n][k]; noise: element n * noise_stride (column 0 of z); states / new_states [n][state_dim] = [reward, stopped,
 * step, usage x k, ...]; progress: DEVICE float[1] (a graph input); consts: HOST float[5] = {cfg.exploration,
 * cfg.exploration_penalty, cfg.filter_usage_penalty, cfg.early_stop_penalty, cfg.test_steps}; outputs pdf / onehot
 * [n][k], entropy / surrogate / penalty_base [n], selected int32 [n].  expo_agent_select_bwd: d logits from the
 * gradients of surrogate and penalty_base (through the entropy); k <= 16.
 */
int expo_agent_select_fwd(const float* logits, const float* noise, int noise_stride, const float* states,
                          const float* progress, const float* consts, int k, int state_dim, int is_train, float* pdf,
                          float* entropy, int32_t* selected, float* onehot, float* surrogate, float* new_states,
                          float* penalty_base, int n, void* stream);
int expo_agent_select_bwd(const float* logits, const int32_t* selected, const float* progress, const float* consts,
                          int k, int state_dim, const float* d_surrogate, const float* d_penalty_base, float* d_logits,
                          int n, void* stream);

/*
 * (ABI 4) The critic step's loss glue (net.py:126-194) -- the callers of the critic around the filter path:
 *   expo_gp_inputs         cat[0:n] = real, cat[n:2n] = fake (float32) and the gradient penalty's interpolation
 *                          interp = real + alpha[n] (fake - real) (net.py:170-172) in one pass; real / fake of `dtype`,
 *                          elems_per_image = H*W*3; interp (and alpha) may be NULL: conversion + concatenation only
 *                          (the generator step's image pairs, exposure_amd/generator_direct.py)
 *   expo_gp_inputs_rows    the same with the batch's images named by ROW: image i is row real_rows[i] of `real` / row
 *                          fake_rows[i] of `fake` (device int64[n]; NULL: row i) -- the replayed records are read straight
 *                          out of the replay memory's pool and the resident data set (replay_memory.py:168-185 builds
 *                          that batch on the host), no gather launch in front
 *   expo_grad_penalty_fwd  per image: norm = sqrt(1e-6 + sum g^2), term = max(norm - 1, 0)^2 (net.py:185-187; the
 *                          penalty is lambda * mean(term)); g float32 [n][elems_per_image]
 *   expo_grad_penalty_bwd  dg = g * dterm[n] * 2 max(norm - 1, 0) / norm   (the gradient TF takes of that term with
 *                          respect to `gradients`; it then flows on into the critic's double backward)
 * No workspace; one block per image for the reduction (fixed order).
 */
int expo_gp_inputs(const void* real, const void* fake, const float* alpha, float* cat_out, float* interp, int n,
                   size_t elems_per_image, int dtype, void* stream);
int expo_gp_inputs_rows(const void* real, const int64_t* real_rows, const void* fake, const int64_t* fake_rows,
                        const float* alpha, float* cat_out, float* interp, int n, size_t elems_per_image, int dtype,
                        void* stream);
int expo_grad_penalty_fwd(const float* g, float* norm, float* term, int n, size_t elems_per_image, void* stream);
int expo_grad_penalty_bwd(const float* g, const float* norm, const float* dterm, float* dg, int n,
                          size_t elems_per_image, void* stream);

/*
 * ToneFilter / ColorFilter with ANY number of curve steps L = cfg.curve_steps (config_example.py:27; the reference
 * loops `for i in range(self.cfg.curve_steps)`, filters.py:264-273, 312-322):
 *     y = (L / S) sum_{i<L} clip(x - i/L, 0, 1/L) k_i,   S = sum_i k_i + 1e-30
 * `curves` = 1 (Tone: one curve for the three channels, params float32 [N][L]) or 3 (Color: one per channel,
 * params [N][3][L], channel-major like filter 7).  1 <= L <= EXPO_CURVE_MAX_STEPS.  Filters 4 and 7 of the entry
 * points above ARE these with L = 8, on kernels tuned for that count; other counts run here (one element-wise
 * pass per direction, O(1) work per element).  Gradient conventions as everywhere (tf.clip_by_value passes on both
 * inclusive bounds; x exactly on a knot receives both neighbouring slopes).  expo_curve_bwd: dx nullable, may alias
 * dy; dparams [N][curves * L] is overwritten; `workspace` >= expo_curve_workspace_bytes(n, h, w, curves, steps)
 * bytes, uninitialised, no atomics, bit-reproducible (block records + a finish launch, like expo_filter_bwd).
 */
size_t expo_curve_workspace_bytes(int n, int h, int w, int curves, int steps);
int expo_curve_fwd(const void* x, void* y, const float* params, int n, int h, int w, int dtype, int curves,
                   int steps, void* stream);
int expo_curve_bwd(const void* x, const void* dy, void* dx, const float* params, float* dparams, int n, int h,
                   int w, int dtype, int curves, int steps, void* workspace, size_t workspace_bytes, void* stream);

/*
 * One Adam update of a LIST of fp32 tensors in one launch -- the three optimisers of a training iteration
 * (net.py:222-251 `ly.optimize_loss(..., optimizer=cfg.optimizer)`; config_example.py:158
 * `tf.train.AdamOptimizer(learning_rate=lr, beta1=0.5, beta2=0.9)`), replacing torch's fused multi-tensor Adam
 * (one block per 65 536 elements: 43 us for the 1-5 M parameters of one network).
 *   params / grads / exp_avg / exp_avg_sq   host arrays of `count` device pointers; tensor j has numel[j] floats,
 *                                           all four in the same element order (tables of EXPO_ADAM_MAX_TENSORS
 *                                           tensors per launch; more tensors -> more launches)
 *   lr     device float        (a captured launch reads the learning rate of the replay)
 *   step   device float        t - 1 on entry; the update computes with t = step + 1 and a one-thread launch behind it
 *                              stores t (capturable, no host involvement; ABI 4-5 let the update's last block do it,
 *                              found through a ticket atomic per block -- most of the update's time)
 *   ticket device uint32, owned by the optimiser; accepted for ABI compatibility, not touched since ABI 6
 *   step_advanced  (ABI 7) 0: as above.  1: `step` already holds t -- a kernel the caller launched in front of this
 *                              update moved the counter (expo_critic_report's / expo_generator_losses' adam_step
 *                              pointers): the update computes with t = step and launches nothing behind it
 * Update rule (torch.optim.Adam without weight decay / amsgrad; TF-1's differs only in where epsilon sits:
 * sqrt(v) + eps' with eps' = eps sqrt(1 - beta2^t)):
 *   m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;
 *   p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 */
#define EXPO_ADAM_MAX_TENSORS 64
int expo_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const size_t* numel, const float* lr, float* step, void* ticket,
                   float beta1, float beta2, float eps, int step_advanced, void* stream);

/*
 * The input tensor of a convnet: image channels + per-image values broadcast as constant planes, minus `offset` --
 * critics.py:64-76 (`tf.concat([images, state planes, statistics planes], axis=3) - 0.5` in front of `cnn`) and
 * agent.py:17-19, 47-53 (`enrich_image_input`, then `net - 0.5` in feature_extractor): dtype conversion,
 * concatenation and subtraction in one launch.
 *   images  device [N][pixels][3] in `dtype` (NULL: zeros -- the adjoint's own adjoint has no image part)
 *   vec     device float32 [N][V] (0 <= V <= 61)
 *   out     device float32 [N][pixels][3 + V] = (c < 3 ? images : vec[n][c - 3]) - offset
 * Linear, so its derivative is slicing / a per-image sum (the host side leaves those to autograd).
 */
int expo_planes_concat(const void* images, const float* vec, float* out, int n, size_t pixels_per_image, int v,
                       int dtype, float offset, void* stream);

/*
 * The whole input side of a critic / value-net pass as ONE launch (critics.py:42-76 in front of `cnn`; net.py:170-172):
 * what expo_gp_inputs_rows + expo_critic_stats + expo_planes_concat compute in four, for images of at most 4096 pixels
 * (a block holds its image in LDS; EXPO_E_BADARG beyond -- the callers then run the separate launches).
 *   a, b        device [.][H][W][3] in `dtype`; a_rows / b_rows (nullable, int64 [n]): image j of the block is row rows[j]
 *   alpha       nullable float32 [n]: a third block of rows a + alpha (b - a) (the gradient penalty's interpolation)
 *   vec_a/_b    nullable float32 [n][v0]: per-image values of the two blocks (the states critics.py:64-70 broadcasts)
 *   planes      float32 [m][H][W][3 + v0 + 3] = concat(image, values, statistics) - offset,  m = 2 n (3 n with alpha)
 *   stats       float32 [m][3] = {mean luminance, luminance variance, mean saturation} (critics.py:48-62)
 *   x_out       nullable float32 [x_count][H][W][3]: the images of rows x_first .. x_first + x_count as float32 (the rows
 *               whose backward needs them: expo_critic_penalty_tangent, expo_critic_stats_bwd)
 */
int expo_net_inputs(const void* a, const int64_t* a_rows, const void* b, const int64_t* b_rows, const float* alpha,
                    const float* vec_a, const float* vec_b, int v0, float* planes, float* stats, float* x_out, int x_first,
                    int x_count, int n, int h, int w, int dtype, float offset, void* stream);

/*
 * The generator step's loss glue (net.py:92-160 with cfg.gan == 'w'; util.py:13-16 state columns): per image
 *   stopped = new_states[1], step = new_states[2];  nv = new_value * [step <= max_len];  gate = a + (1 - a) stopped
 *   reward = gate (fake_logit - fake_input_logit) m - penalty;  q = reward + (1 - stopped) gamma nv;  adv = q - old_value
 *   use_td:  g term = -q plm - surrogate adv      (net.py:135-140, 152-157)
 *   else:    g term = -reward - surrogate reward
 *   losses[0] = g_loss = mean g term,  losses[1] = v_loss = mean adv^2
 * consts = host float[5] {a = all_reward, m = critic_logit_multiplier, gamma = discount_factor,
 * plm = parameter_lr_mul, max_len = maximum_trajectory_length}; penalty may be NULL (cfg.use_penalty off).
 * All vectors float32 [N]; new_states [N][state_dim].  reward / q_value [N] are reported values;
 * coef float32 [5][N] receives d g_loss / d {fake_logit, new_value, surrogate, penalty} and d v_loss / d old_value
 * (stop-gradients of the reference applied: the weight of the surrogate and q inside adv are constants), so the
 * backward pass is coef times the upstream scalar.  One block; N is a minibatch.
 * adam_step_a / adam_step_b (ABI 7; nullable device floats) += 1: the step counters of the generator's and the value net's
 * updates behind this launch (expo_adam_step with step_advanced = 1).
 */
int expo_generator_losses(const float* fake_logit, const float* fake_input_logit, const float* new_value,
                          const float* old_value, const float* new_states, int state_dim, const float* penalty,
                          const float* surrogate, const float* consts, int use_td, float* losses, float* reward,
                          float* q_value, float* coef, int n, float* adam_step_a, float* adam_step_b, void* stream);

/* ---- the convnets' convolution (round 5) ------------------------------------------------------------------------
 * `ly.conv2d(net, C_out, kernel_size=4, stride=2)` (SAME padding; agent.py:21-32, critics.py:13-35) on NHWC float32
 * tensors as implicit-GEMM kernels on the f32 matrix cores (csrc/conv_ops.hip), with the layer's bias + lrelu
 * (util.py:225-229) in the epilogue:
 *   y[n][oh][ow][co] = f(bias[co] + sum_{kh,kw,ci} x[n][2 oh - 1 + kh][2 ow - 1 + kw][ci] w[co][kh][kw][ci])
 *   x     float32 [n][h][w][cin] (h, w even)      w     float32 [cout][4][4][cin] (16-byte aligned: the channels_last
 *   y     float32 [n][h/2][w/2][cout]                    memory order of an nn.Conv2d weight (cout, cin, 4, 4))
 *   bias  float32 [cout] or NULL                  act   0: f = identity, 1: f = lrelu(., leak)
 * Deterministic (fixed summation order, no atomics), no workspace.  Replaces aten::miopen_convolution +
 * expo_bias_lrelu_fwd for these layers. */
int expo_conv4x4s2_fwd(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int cin,
                       int cout, int act, float leak, void* stream);
/* The data gradient of the same convolution (the transposed convolution of dy): four dense GEMMs, one per parity
 * class of the input pixel.  Replaces aten::convolution_backward(input gradient only) and the zero fill in front of
 * MIOpen's split-K kernel.
 *   dy  float32 [n][h/2][w/2][cout] (cout % 4 == 0)     dx  float32 [n][h][w][cin], every element written */
int expo_conv4x4s2_bwd_data(const float* dy, const float* w, float* dx, int n, int h, int wd, int cin, int cout,
                            void* stream);
/* The weight gradient: dw[co][kh][kw][ci] = sum over the batch's output pixels of dy[.][co] x[.][kh][kw][ci], written in
 * the weight's own memory order, every element (dw 16-byte aligned).  The pixel sum of a 32 x 128 tile of dw is spread
 * over P blocks; for P > 1 the blocks write P full-size copies of dw into `workspace` (caller-owned scratch, no
 * initialisation needed, at least expo_conv4x4s2_wrw_workspace_bytes(...) bytes, 16-byte aligned) and a second,
 * element-wise launch adds them in block order: deterministic, no atomics, no zero fill of dw.  w / 2 must be even.
 * Replaces aten::convolution_backward(weight gradient only) + the zero fill in front of MIOpen's split-K kernel. */
size_t expo_conv4x4s2_wrw_workspace_bytes(int n, int h, int wd, int cin, int cout);
int expo_conv4x4s2_wrw(const float* x, const float* dy, float* dw, int n, int h, int wd, int cin, int cout,
                       void* workspace, size_t workspace_bytes, void* stream);
/* (ABI 6) The same three primitives with the ACTIVATION'S gradient and the BIAS gradient folded in -- what a layer
 * `z = lrelu(conv(x, w) + b)` (agent.py:21-32, critics.py:13-35; util.py:225-229) needs around its convolutions in the
 * backward and in the gradient penalty's double backward (net.py:174-194), without launches of their own:
 *   slope(z) = 1 (z > 0), leak (z < 0), (1 + leak) / 2 (z == 0: TF's sub-gradient of f1 x + f2 |x|)
 *   expo_conv4x4s2_bwd_data_mask   dx = D(dy, w) * slope(zmask):  zmask float32 [n][h][w][cin] = the activation of the
 *                                  layer BELOW -- its lrelu backward in this kernel's epilogue
 *   expo_conv4x4s2_fwd_mask        y = F(x, w) * slope(zmask):    zmask float32 [n][h/2][w/2][cout]; y MAY ALIAS zmask --
 *                                  the tangent of the double backward, t_l = F(t_{l-1}, w_l) slope(z_l), written over z_l
 *   expo_conv4x4s2_wrw_bias        expo_conv4x4s2_wrw and dbias[co] = sum of dy[.][co] over the output pixels of the
 *                                  first `bias_images` images (0 <= bias_images <= n; cout % 4 == 0, dbias 16-byte
 *                                  aligned): the column sums come from the values the kernel feeds to the matrix cores
 *                                  anyway.  Same workspace as expo_conv4x4s2_wrw (the bias copies are part of it).
 * The data gradient of the layers with 6 or 17 input planes (critic, value net) runs on the vector ALUs with the weights
 * in scalar registers (conv_bwd_small_kernel: a 32-wide matrix-core tile would be 81 % / 47 % padding). */
int expo_conv4x4s2_bwd_data_mask(const float* dy, const float* w, const float* zmask, float* dx, int n, int h, int wd,
                                 int cin, int cout, float leak, void* stream);
int expo_conv4x4s2_fwd_mask(const float* x, const float* w, const float* zmask, float* y, int n, int h, int wd, int cin,
                            int cout, float leak, void* stream);
int expo_conv4x4s2_wrw_bias(const float* x, const float* dy, float* dw, float* dbias, int bias_images, int n, int h,
                            int wd, int cin, int cout, void* workspace, size_t workspace_bytes, void* stream);
/* (ABI 7) Two problems of ONE geometry as one grid: the forward (bias + activation) and the data gradient (activation
 * gradient of the layer below in the epilogue) of two independent layers -- the agent's filter and selector extractors read
 * the same input through different weights (agent.py:47-56), the critic's and the value net's pair passes of a G / V step run
 * side by side -- at batch 64 / 128, where one of them alone leaves CUs idle.  The decomposition is chosen for the grid that
 * runs (twice the blocks), so the results equal the two separate calls up to the order in which the K slices are added
 * (float32 rounding); a pair call is bit-reproducible. */
int expo_conv4x4s2_fwd_pair(const float* x_a, const float* w_a, const float* bias_a, float* y_a, const float* x_b,
                            const float* w_b, const float* bias_b, float* y_b, int n, int h, int wd, int cin, int cout, int act,
                            float leak, void* stream);
/* (ABI 7) expo_conv4x4s2_fwd (zmask NULL) / expo_conv4x4s2_fwd_mask (zmask given, bias NULL, act 0) of a FIRST layer whose input
 * comes from expo_planes_concat / expo_net_inputs / expo_critic_penalty_tangent: channels 3 .. cin - 1 of x are per-image
 * CONSTANTS (the states and statistics critics.py:64-76 / agent.py:17-19 broadcast as planes; the kernel reads their values
 * from the image's first pixel).  The convolution is linear, so the constant planes contribute one value per image, channel
 * and border class (which of the 4 x 4 taps fall inside the image): K shrinks from 16 cin to 48.  Same sum in another order
 * (float32 rounding).  64-wide inputs of 4 .. 20 planes and at most 32 output channels; EXPO_E_BADARG otherwise (take
 * expo_conv4x4s2_fwd).  The caller vouches for the planes being constant. */
int expo_conv4x4s2_fwd_planes(const float* x, const float* w, const float* bias, const float* zmask, float* y, int n, int h,
                              int wd, int cin, int cout, int act, float leak, void* stream);
int expo_conv4x4s2_fwd_planes_pair(const float* x_a, const float* w_a, const float* bias_a, float* y_a, const float* x_b,
                                   const float* w_b, const float* bias_b, float* y_b, int n, int h, int wd, int cin, int cout,
                                   int act, float leak, void* stream);
int expo_conv4x4s2_bwd_data_mask_pair(const float* dy_a, const float* w_a, const float* zmask_a, float* dx_a, const float* dy_b,
                                      const float* w_b, const float* zmask_b, float* dx_b, int n, int h, int wd, int cin,
                                      int cout, float leak, void* stream);
/* The weight (and bias) gradients of up to 8 layers -- a whole stack -- with ONE reduce launch for all of them: per layer the
 * arguments of expo_conv4x4s2_wrw_bias (arrays of `count` entries; dbias[l] may be NULL: no bias gradient for that
 * layer), each layer with its own workspace of expo_conv4x4s2_wrw_workspace_bytes(...) bytes.  Same results, bit for
 * bit, as `count` separate calls. */
int expo_conv4x4s2_wrw_group(int count, const float* const* x, const float* const* dy, float* const* dw,
                             float* const* dbias, const int* bias_images, const int* n, const int* h, const int* wd,
                             const int* cin, const int* cout, void* const* workspace, const size_t* workspace_bytes,
                             void* stream);
/* Probes and tests: waves per block (1 .. 4) and blocks per tile of the weight-gradient kernel (negative = leave as
 * is, 0 = the library's choice; initial values from EXPO_CONV_WRW_SLICES / EXPO_CONV_PARTS).  Independent of
 * expo_conv_tuning's `slices`. */
int expo_conv_wrw_tuning(int slices, int parts);
/* Probes and tests: override how the forward / data-gradient kernels decompose a problem (process-wide; negative = leave as is,
 * 0 = the library's own choice).  tile 1-4: an LDS-tiled forward shape, 5: the flat kernel, 6: the
 * first layers' row-staged kernel where it applies; nt 1 | 2: column tiles per
 * wave; slices 1, 2, 4, 8 or 16: K slices per tile (any other value is rejected: the kernels cut K into 4 S2 segments).
 * The initial values come from EXPO_CONV_TILE / _NT / _SLICES, read once. */
int expo_conv_tuning(int tile, int nt, int slices);

/* ---- the hand-scheduled critic update (round 6; csrc/critic_step.hip, exposure_amd/critic_direct.py) ------------------
 * net.py:126-199, 245-251: c_loss = mean(D(fake) - D(real)) + lambda mean(max(||grad_x^ D(x^)|| - 1, 0)^2).  The real,
 * fake and interpolated images run as ONE batch [real | fake | interpolated] through critics.py:42-98; the kernels
 * below are the per-row / per-image reductions between its convolutions and GEMMs.  Fixed summation orders.
 *   expo_critic_head_fwd  hpre float32 [slabs][M][hidden]: fc1's pre-activation (critics.py:94-96) as `slabs` partial sums
 *                         (expo_fc_fwd_slabs; 1 = the finished GEMM) added here in slab order, + b1 (nullable: already
 *                         inside), M = n_real + n_fake + n_interp rows.  h = lrelu(hpre); logits[m] = h[m] . w2 + b2 (critics.py:97);
 *                         dh[m] = dlogit[m] w2 slope(h[m]) with dlogit = -inv_n (real), +inv_n (fake), 1 (interpolated:
 *                         the inner gradient starts from ones, net.py:174-183)
 *   expo_critic_report    out[0..4] = {c_loss = mean fake - mean real + lambda mean term, emd = mean real - mean fake, mean
 *                         norm, lambda mean term, c_average = (mean fake + mean real) / 2} (net.py:188-199) from the logits
 *                         [real | fake | interpolated] and the per-image norm / term; ema (nullable, device float) advances
 *                         as ema += (1 - decay) (c_average - ema) (update_average, net.py:165-168, 267-268); adam_step
 *                         (nullable, device float) += 1: the step counter of the update behind this launch
 *                         (expo_adam_step with step_advanced = 1)
 *   expo_critic_head_bwd  gb1 = sum over the real + fake rows of dh;  gw2 = sum over those rows of dlogit h + sum over
 *                         the interpolated rows of thpre slope(h) (thpre float32 [th_slabs][n_interp][hidden]: the penalty's
 *                         tangent in front of fc1's activation, as partial sums like hpre);  gb2 = sum of dlogit
 *   expo_plane_sums       sums[n][c - first] = sum over the pixels of x[n][.][c], first <= c < channels (<= 16 planes):
 *                         the gradient reaching the per-image values planes_concat broadcast (critics.py:64-76)
 *   expo_gp_direct        g = u[..., 0:3] + ds (u float32 [n][pixels][u_channels], ds [n][pixels][3]);
 *                         norm = sqrt(1e-6 + sum g^2), term = max(norm - 1, 0)^2 (net.py:185-187) and
 *                         v = scale 2 max(norm - 1, 0) / norm g  (the gradient of scale sum(term)) in one launch
 *   expo_critic_penalty_tangent  everything between the first layer's data gradient and the tangent pass as ONE launch (a
 *                         block per image): u float32 [n][h w][6] (image + statistics planes), x float32 [n][h w][3] the
 *                         interpolated images, stats [n][3] their statistics ->  gs = plane sums of u[..., 3:6];
 *                         g = u[..., 0:3] + J^T gs (expo_critic_stats_bwd);  norm, term as expo_gp_direct;
 *                         v = scale 2 max(norm - 1, 0) / norm g;  t0 [n][h w][6] = [v | J v broadcast]
 *                         (expo_critic_stats_jvp + expo_planes_concat with offset 0): the tangent's input */
int expo_critic_head_fwd(const float* hpre, const float* b1, int slabs, const float* w2, const float* b2, int n_real, int n_fake,
                         int n_interp, int hidden, float inv_n, float leak, float* logits, float* h, float* dh, void* stream);
int expo_critic_report(const float* logits, const float* norm, const float* term, int n_real, int n_fake, int n_interp,
                       float lambda, float decay, float* out, float* ema, float* adam_step, void* stream);
int expo_critic_head_bwd(const float* dh, const float* h, const float* thpre, int th_slabs, int n_real, int n_fake,
                         int n_interp, int hidden, float inv_n, float leak, float* gb1, float* gw2, float* gb2, void* stream);

/* ---- the first FC layer behind a convnet (critics.py:27-31 / agent.py:33-35: ly.fully_connected(flat, 128)) in the
 * hand-scheduled passes: the library GEMM takes ~10 us for it whatever the batch is (one workgroup per 16 x 16 tile walks
 * K = 4096), 14 calls per training iteration.
 *   expo_fc_fwd_slabs_count  S = the number of partial sums expo_fc_fwd_slabs writes for M rows of K features (0: K is not
 *                         a multiple of 128 -- use the library GEMM)
 *   expo_fc_fwd_slabs     slabs float32 [S][M][N]: slab s = x[:, K_s] w[:, K_s]^T over the s-th of S ranges of K (x float32
 *                         [M][K], w float32 [N][K] = nn.Linear's weight).  Their sum in slab order (+ bias) is the
 *                         pre-activation; expo_critic_head_fwd / _bwd add them while they read it.  Fixed summation order.
 *   expo_fc_bwd_data_mask gy float32 [M][C] = (dh w) slope(z): the layer's data gradient (dh float32 [M][J], w float32
 *                         [J][C], J % 16 == 0, C % 128 == 0) times the activation gradient of the feature map z [M][C] below
 *                         it (util.py:225-229; TF's sub-gradient at 0) -- torch.mm + expo_lrelu_bwd in one launch */
int expo_fc_fwd_slabs_count(int m, int k);
int expo_fc_fwd_slabs(const float* x, const float* w, float* slabs, int m, int n, int k, void* stream);
int expo_fc_bwd_data_mask(const float* dh, const float* w, const float* z, float* gy, int m, int j, int c, float leak,
                          void* stream);
/*   expo_fc_wrw           dw float32 [J][C] = dh^T x (dh float32 [M][J], x float32 [M][C]): the layer's weight gradient over M
 *                         rows -- a GEMM whose K dimension is the batch; OVERWRITTEN, fixed summation order */
int expo_fc_wrw(const float* dh, const float* x, float* dw, int m, int j, int c, void* stream);
int expo_plane_sums(const float* x, float* sums, int n, size_t pixels_per_image, int channels, int first, void* stream);
int expo_gp_direct(const float* u, int u_channels, const float* ds, float scale, float* v, float* norm, float* term, int n,
                   size_t pixels_per_image, void* stream);
int expo_critic_penalty_tangent(const float* u, const float* x, const float* stats, float scale, float* t0, float* norm,
                                float* term, int n, int h, int w, void* stream);
#ifdef __cplusplus
}
#endif
#endif /* EXPOSURE_HIP_H_ */
