/*
 * helmnet_hip.h -- C ABI of libhelmnet_hip.so, the MI355X (gfx950) implementation of the
 * helmnet IterativeSolver inference hot path.
 *
 * The reference (ucl-bug/helmnet) is pure Python/PyTorch and defines NO plugin / operator /
 * FFI interface: its boundary is the Python surface of
 * `IterativeSolver`.  This header is the C boundary introduced underneath that surface; each
 * entry point names the reference code it replaces (paths relative to the reference root).
 * The Python class `helmnet_amd.IterativeSolver` binds these symbols with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - every function returns 0 on success and a negative hn_status on failure; the message
 *     is available from hn_last_error(ctx) (or hn_last_error(NULL) for hn_create failures).
 *     No exception crosses the ABI, no global mutable state except that last-error string; environment
 *     variables are read once per context, in hn_create, as defaults for hn_set_unet_precision / hn_set_option.
 *   - every entry point selects the context's device for its own duration and restores the caller's.
 *   - all tensor arguments are DEVICE pointers to contiguous fp32, NCHW, owned by the caller,
 *     who guarantees their lifetime until `stream` has been synchronised.
 *   - all work is enqueued asynchronously on the caller's `stream` (a hipStream_t passed as
 *     void*; NULL = the default stream).  A ctx is bound to one device and is NOT thread-safe.
 *     ONE exception, "side streams": hn_step and hn_train_grad run part of their work on library streams beside the caller's, and HIP
 *     deals streams onto a few hardware queues -- a library stream that shares the caller's queue silently serialises (inference -9 %,
 *     the training step up to 3 x).  So the FIRST call that meets a new caller stream (per context and entry point) PROBES the library's
 *     candidate streams against it: it synchronises that stream (hipStreamSynchronize), runs a 200 us spin kernel on it and an empty
 *     kernel on a candidate, up to three times per candidate (majority), ~1 ms in all, and remembers the answer for that stream -- later
 *     calls on any already-probed stream are asynchronous again.  The probe never runs while the caller's stream (or a stream it is
 *     compared with) is being captured; a first call under capture uses candidate 0 unprobed.  A capture in global mode on ANOTHER thread
 *     cannot be detected: make the first call before it, or disable probing.  HN_SIDE_PRIORITY=1 / 2 / 3 in the environment (lowest /
 *     normal / highest priority candidate) disables probing altogether.  hn_get_counter(HN_CNT_STREAM_PROBES) counts the probes.
 *   - the library owns only its ctx: a re-packed copy of the weights, the spectral tables
 *     and the activation workspace (grown on demand by hn_reserve / the first call).
 */
#ifndef HELMNET_HIP_H
#define HELMNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hn_ctx hn_ctx;

enum hn_status {
    HN_OK = 0,
    HN_ERR_ARG = -1,      /* bad argument (shape, NULL pointer, unsupported size)            */
    HN_ERR_STATE = -2,    /* call order: weights / domain not set yet                        */
    HN_ERR_UNSUPPORTED = -3, /* architecture / activation the kernels do not implement       */
    HN_ERR_HIP = -4,      /* a HIP runtime call failed                                       */
    HN_ERR_NOMEM = -5
};

/* activation kinds accepted by hn_load_weights (architectures.py:5-44 getActivationFunction): every name the
 * reference knows except 'relu_batchnorm' (it carries BatchNorm2d parameters and running statistics in the
 * state_dict); an unknown kind returns HN_ERR_UNSUPPORTED, mirroring the reference's NotImplementedError.
 * The piecewise-linear ones (0..2) take their slope from the blob (constant for relu / leakyrelu); 3..7 are
 * evaluated in fp32 in the kernels' epilogues: celu (alpha 1), tanh, gelu (erf form), tanhshrink, softplus
 * (beta 1, threshold 20). */
enum hn_act { HN_ACT_PRELU = 0, HN_ACT_RELU = 1, HN_ACT_LEAKYRELU = 2, HN_ACT_CELU = 3, HN_ACT_TANH = 4, HN_ACT_GELU = 5,
              HN_ACT_TANHSHRINK = 6, HN_ACT_SOFTPLUS = 7 };

/* Arithmetic of the UNet convolutions (hn_set_unet_precision).  Everything in HBM (activations, hidden state,
 * wavefield, residual), the 1x1 out-conv, the wavefield update and the spectral residual are fp32 in every mode.
 *   HN_PREC_FP32     every product in fp32 on the f32-input matrix core (default; the reference's arithmetic)
 *   HN_PREC_BF16X3   3-term bf16 split of both operands, 6 product terms, fp32 accumulate: fp32-accurate emulation
 *   HN_PREC_FP16     fp16 operands, fp32 accumulate: the "mixed fp16 UNet / fp32 spectral residual" configuration
 *   HN_PREC_BF16X2   2-term bf16 split, 3 product terms (~2^-16 relative, full fp32 exponent range)
 *   HN_PREC_FP32_VALU  fp32 on the vector ALU (direct convolution; A/B reference for the matrix-core kernels) */
enum hn_precision { HN_PREC_FP32 = 0, HN_PREC_BF16X3 = 1, HN_PREC_FP16 = 2, HN_PREC_BF16X2 = 3, HN_PREC_FP32_VALU = 4 };

/* Tuning knobs (hn_set_option).  Bit-exact ones -- the same kernels and summation order, only launched differently: LANES,
 * SIDE_STREAM, SPECTRAL_COLS (same butterflies, other memory access).  The others select a different kernel for the same
 * fp32 arithmetic and agree to fp32 rounding, like two fp32 implementations of the reference do: DEEP (other summation order,
 * 2e-6 * max), SPECTRAL_PFA (FFT instead of the dense operator), SPECTRAL_RADIX16 (other butterfly order), DC_VALU. */
enum hn_option {
    HN_OPT_LANES = 0,        /* 1..8 sub-batches pipelined on internal streams (default 1; [measured, r5] 2 loses 3 % at 256^2 x 32 and 2 % at
                              * 256^2 x 64 -- one chain of 64 maps is the faster form since the r5 level-0 kernels --, +1 % at batch 128) */
    HN_OPT_SIDE_STREAM = 1,  /* conv_state kernels: 0 in line; on a library side stream released 1 after the last `down`,
                              * 2 level by level behind conv_signal, 3 behind the fused deep level               */
    HN_OPT_DEEP = 3,         /* the deep levels as one launch.  1: deepest level (32^2) + bottleneck in one per-sample LDS kernel (hn_deep.hip; 256^2 at
                              * depth 4).  2 (default): where the last encoder level is 64^2 (512^2) or the last two are 64^2 and 32^2 (256^2), those
                              * levels + the bottleneck as ONE launch with EIGHT workgroups per sample that exchange halo rows through flag-guarded
                              * global memory (hn_deepx.hip; batches up to 32 maps per call, every activation; fp32 arithmetic, which is what the 16-bit
                              * modes use below level 1 anyway -- they take this launch too), else as 1.  0: layer
                              * by layer.  Its waits are bounded like HN_OPT_SIDE_SYNC's and report through hn_check_async_errors                   */
    HN_OPT_SPECTRAL_PFA = 4, /* 0/1: prime-factor FFT for n = 3 * 2^k, 5 * 2^k, 7 * 2^k instead of the dense n x n operator (default 1;
                              * read by the next hn_set_domain)                                                     */
    HN_OPT_SPECTRAL_RADIX16 = 5, /* 256-point lines: 0 the radix-4 kernels, 1 radix-16 columns + 8x4x8 rows (default), 2 radix-16 rows too */
    HN_OPT_DC_VALU = 6,      /* fp32 DoubleConvs of the largest level (W >= 256) on the packed vector FMA (every FMA useful, same peak as the
                              * fp32 MFMA, whose 3x3 packing fills 75 % of its slots): 0 none (matrix core); 1 inc and the decoder, 2 all three on the
                              * compiler-scheduled kernel (hn_dcv.hip); 3 inc and the decoder, 4 (default) all three on the hand-scheduled kernel
                              * (hn_dca.hip: all 8 mid channels per wavefront, LDS-direct staging; [measured, r5] +1 .. 3 % it/s over 1).  Smooth
                              * activations and unaligned tensors always take hn_dcv.hip                                          */
    HN_OPT_DC_PAIR = 12,     /* 0/1 (default 1): where inc and conv_signal_0 both run on hn_dca.hip they are ONE launch -- conv_signal's blocks wait, tile by
                              * tile, on a flag the inc blocks of the tiles they read publish (write-through stores, agent-scope flag).  The same launch
                              * eagerly, under stream capture (HN_OPT_GRAPH: the launch's epoch is derived on the device there) and in every pipeline
                              * lane.  Bit-identical to the two launches.
                              * ASSUMES that the workgroups of a launch are dispatched in index order (true of the hardware dispatcher; a tool that
                              * re-orders or caps dispatch must run with 0): a conv_signal block whose inputs have not arrived after ~2 M polls
                              * writes NaN into its tile AND raises the context's sticky device error (HN_ERR_STATE from this or the next hn_step) */
    HN_OPT_SIDE_SYNC = 13,   /* 0/1 (default 1): between the iterations of ONE hn_step call the side stream is joined -- and, at 256^2, released -- through
                              * device words that kernels of the main chain store / poll on their way (one thread each) instead of event packets,
                              * each of which holds the main stream for ~7 us; a call forks the side stream with one event and its LAST iteration uses events.
                              * hn_step's single-lane eager path, fp32, HN_OPT_SIDE_STREAM 1; waits are bounded (2 s, then hn_step fails: the call in
                              * which a wait gave up returns HN_ERR_STATE if the error word is up by the time its launches are enqueued, else the next
                              * call does; hn_check_async_errors after a stream synchronise is the reliable test).
                              * The same option switches hn_train_grad's side stream (HN_OPT_TRAIN_OVERLAP 2) to device words (words 64 .. 160 of the
                              * context's 256 sync words; eager calls only) and then opens that side-stream window at every problem size instead of
                              * 200 k .. 1 M pixels; hn_train_grad reports a timed-out wait of an EARLIER call as HN_ERR_STATE at entry.
                              * Same kernels, same results; [measured, r5] +1 % it/s at 256^2 x 32, +8 % at batch 8.  A tool that runs ONE kernel at
                              * a time across all queues (counter collection: rocprofv3 --pmc) can starve such a wait: hn_create then defaults
                              * to 0 (ROCPROF_COUNTER_COLLECTION / ROCPROF_COUNTERS / ROCP_METRICS in the environment); any other such tool gets
                              * HN_ERR_STATE from hn_step after 2 s and must set HN_SIDE_SYNC=0                                              */
    HN_OPT_STATE_KERNEL = 14, /* 0/1 (default 1): the hidden-state DoubleConvs (10 -> 2 -> 2) of the levels at least 64 wide on the streaming kernel
                              * (hn_cs.hip: the tile's ten input planes through a ring of LDS-direct loads); 0: the general direct kernel.
                              * Bit-identical                                                                                          */
    HN_OPT_HIST_COPY = 15,   /* 0 (default): hn_step's residual / wavefield histories are written in place of the copies (see hn_step); 1: every
                              * iteration works in the caller's wf / res and copies them into the history slots (A/B; bit-identical)       */
    HN_OPT_INC_SIGMA_MAP = 16, /* 0/1 (default 1): the sigma maps are constants of the domain and two of the six input channels of every UNet evaluation the solver
                              * makes (hybridnet.py:564-566), so their share of the input layer's first convolution is precomputed per domain (float64 on the
                              * host, at hn_load_weights / hn_set_domain) and the hand-scheduled input layer (HN_OPT_DC_VALU 3 / 4) convolves four
                              * channels, adding the map in the tiles near the border (it is zero elsewhere).  Agrees with 0 to fp32 rounding  */
    HN_OPT_SPECTRAL_COLS = 7, /* 256-point column pass: 0 the r2 kernel (16-byte global accesses), 1 (default) / 2: coalesced float4 row
                              * segments transposed through LDS, 16 / 32 columns per workgroup                      */
    HN_OPT_TRAIN_FUSED = 10, /* hn_train_grad: sum of 1 (forward pass: an 8-channel DoubleConv is ONE launch of the fused matrix-core kernels of
                              * the inference path, which also store the pre-activation mid tensor to the tape; same tape within fp32
                              * rounding), 2 (backward pass: both backward-data convolutions of a big level's DoubleConv as one tiled
                              * launch; bit-identical gradients), 4 (the hidden-state DoubleConvs of all levels as one launch per
                              * direction instead of two; bit-identical), 16 (the backward-data pass of every 8-channel DoubleConv on the fp32 matrix
                              * core, k_dc_bwd_mfma_p: one launch, the g_z tile in LDS; the f32 matrix instruction is an exact fmaf chain in the vector
                              * kernels' order: bit-identical gradients at the training sizes) and 32 (with 4 and 16: conv_state's backward-data pass
                              * rides in the decoder's launch of the same level, k_dc_bwd_mfma_aux; d loss / d out is added up in one accumulator: fp32
                              * rounding); default 55, 0: every convolution as its own launch (round 3).  (Value 8 is a lab setting: the small
                              * levels' backward DoubleConvs on the tiled kernel; measured slower.)                                       */
    HN_OPT_TRAIN_OVERLAP = 11, /* hn_train_grad, backward pass: where the three weight-gradient launches of unrolled iteration t run.  0: in line on the
                              * caller's stream.  2 (default): on a library stream beside the backward chain of iteration t - 1 (two sets of gradient
                              * buffers) AND, where the chain is latency-bound (200 k .. 1 M pixels per call), with at most ~2 of their blocks per CU --
                              * each walks more tiles, the chain keeps half of every CU: 9.13 -> 8.77 ms at 96^2 x 32 -- and the forward sweep's
                              * hidden-state launch on that stream beside the decoder.  The cap changes how many partial sums a weight gradient is added
                              * up from: a fixed order (reproducible), not mode 0's.  (Value 1 is a lab setting: the side stream without
                              * the cap; bit-identical to 0 and measured equal to it.)                                                    */
    /* ---- laboratory knobs (same setter): implemented, tested bit-identical / equal within rounding, and MEASURED NOT TO HELP; they stay for A/B runs
     * and are not part of the supported surface (DESIGN_NOTEBOOK.md has the measurements) ---- */
    HN_EXP_GRAPH = 100,      /* 0: launch every kernel (default; measured 4 % faster); 1: replay one captured iteration per HIP graph
                              * launch; even n <= 64: n iterations per graph.  Bit-identical                              */
    HN_EXP_TRAIN_LANES = 101 /* hn_train_grad: 1 (default) the whole batch as one chain of launches; 2: the two halves of the batch as
                              * two chains on two streams.  Same gradient up to the order of the final sum over the halves; measured equal to 1 */
};
/* Diagnostics counters (hn_get_counter). */
enum hn_counter { HN_CNT_GRAPH_REPLAYS = 0, HN_CNT_EAGER_ITERATIONS = 1, HN_CNT_GRAPHS_CAPTURED = 2,
                  HN_CNT_STREAM_PROBES = 3,     /* reference-stream sets probed so far (see "side streams" in the conventions above) */
                  HN_CNT_SIDE_CANDIDATE = 4,    /* candidate (0 .. 3) hn_step's side stream was last picked from; -1 before the first pick */
                  HN_CNT_TRAIN_FWD_EVENTS = 5,  /* times hn_train_grad recorded the registered forward event (hn_train_set_forward_event): a caller compares the
                                                 * counter before and after a call to know whether event and table are valid for it (not under capture) */
                  HN_CNT_FLAG_SYNC_ITERATIONS = 6 };/* hn_step iterations whose side-stream hand-overs went through device words (HN_OPT_SIDE_SYNC): n_iter - 1 per
                                                 * eligible call, 0 with the option off, under stream capture, under counter collection, with several lanes */

#define HN_ABI_VERSION 7
int hn_abi_version(void);

/* Create / destroy a context on HIP device `device_id`. */
int hn_create(hn_ctx** out, int device_id);
void hn_destroy(hn_ctx* ctx);
const char* hn_last_error(const hn_ctx* ctx);

/* Number of fp32 values hn_load_weights expects for (features, depth, state_ch): the `f.*`
 * tensors of the checkpoint in state_dict order (architectures.py:317-388). */
size_t hn_weight_count(int features, int depth, int state_ch);

/* Upload the HybridNet parameters (HOST pointer; copied and re-packed, caller keeps `blob`).
 * Replaces: HybridNet.__init__ + load_state_dict (architectures.py:317-388).
 * Order of `blob`: inc.double_conv.{0.weight,0.bias,1.weight,2.weight,2.bias};
 *   for d in 0..depth-1: enc.d.conv_signal (5 tensors), enc.d.down.{weight,bias},
 *                        enc.d.conv_state (5 tensors);
 *   decode.0..depth (5 tensors each); up.0..depth-1 {weight,bias}; outc.conv.{weight,bias}.
 * Supported: features == 8, state_ch == 2, 1 <= depth <= 6, state_depth == depth. */
int hn_load_weights(hn_ctx* ctx, const float* blob, size_t n_floats, int features, int depth,
                    int state_ch, int act_kind);

/* Select the arithmetic of the UNet convolutions for every later call on this context (per context, not per
 * process: two contexts may differ).  The default is HN_PREC_FP32, or what the environment variable HN_UNET_IMPL
 * (fp32 | bf16x3 | fp16 | bf16x2 | valu) named when hn_create ran.  The reference has a single fp32 path
 * (architectures.py:439-465); the 16-bit modes are this library's extension (BASELINE.json configs[4]). */
int hn_set_unet_precision(hn_ctx* ctx, int precision);
int hn_get_unet_precision(const hn_ctx* ctx);
int hn_set_option(hn_ctx* ctx, int option, int value);
int64_t hn_get_counter(const hn_ctx* ctx, int counter);
/* Device-side waits of this library (HN_OPT_SIDE_SYNC, HN_OPT_DC_PAIR) are bounded and report through a sticky host-visible word; a call can only
 * see what is up by the time it returns, and the last hn_step of a solve has no successor.  HN_OK, or HN_ERR_STATE (with the message) if any wait
 * of any earlier call on this context gave up -- call it after synchronising the stream the work ran on.  No GPU work, no synchronisation. */
int hn_check_async_errors(hn_ctx* ctx);

/* Build the spectral-operator constants for an n x n domain (float64 on the host, fp32 on the
 * device): k grids, PML coefficients ax/bx/ay/by, sigma maps, FFT twiddles.
 * Replaces: FastLaplacianWithPML.init_variables / get_gamma_functions (spectral.py:267-363),
 * FourierDerivative k-grid (spectral.py:126-146), IterativeSolver.set_laplacian
 * (hybridnet.py:110-131).  n must be divisible by 2^depth (16 for the shipped net). */
int hn_set_domain(hn_ctx* ctx, int n, int pml, float sigma_max, float k);

/* Copy the sigma maps [2, n, n] (sigma_x, sigma_y) to `out` (device). hybridnet.py:126-131. */
int hn_get_sigmas(hn_ctx* ctx, float* out, void* stream);

/* Total hidden-state length per channel, sum_d (n / 2^d)^2 (architectures.py:390-404). */
int64_t hn_state_len(const hn_ctx* ctx);

/* Pre-allocate the activation workspace for batches up to `max_batch` (optional; otherwise
 * grown by the first call, which then must not be under stream capture). */
int hn_reserve(hn_ctx* ctx, int max_batch);

/* out[B,2,n,n] = L(wf[B,2,n,n]), the spectral Laplacian with PML.
 * Replaces: IterativeSolver.apply_laplacian (hybridnet.py:540-542) +
 * fast_laplacian_with_pml (spectral.py:31-79). */
int hn_laplacian(hn_ctx* ctx, const float* wf, float* out, int batch, void* stream);

/* res[B,2,n,n] = L(wf) + k_sq * wf - src;  k_sq is [B,1,n,n]; src is [src_batch,2,n,n] with
 * src_batch == 1 (broadcast) or == batch.  Replaces get_residual (hybridnet.py:544-556). */
int hn_residual(hn_ctx* ctx, const float* wf, const float* k_sq, const float* src, int src_batch,
                float* res, int batch, void* stream);

/* out[B,2,n,n] = J^T g: the vector-Jacobian product of hn_residual with respect to the wavefield, i.e. the adjoint operator
 * L^H(g) + k_sq * g (real k_sq; L^H is the conjugate transpose of the spectral Laplacian with PML).  What torch.autograd
 * computes when the reference back-propagates through get_residual (hybridnet.py:544-556) in training_step (:385-413); also
 * the building block of adjoint-state / normal-equation solvers.  g must not alias out. */
int hn_residual_vjp(hn_ctx* ctx, const float* g, const float* k_sq, float* out, int batch, void* stream);

/* rmse[B] = sqrt(mean_{c,h,w} res^2).  Replaces test_loss_function (hybridnet.py:295-297). */
int hn_rmse(hn_ctx* ctx, const float* res, float* rmse, int batch, void* stream);

/* d[B,2,n,n] = HybridNet(in6[B,6,n,n]); the hidden states are read from `states_in` and the new
 * ones written to `states_out`, both in the reference's flat layout [B, 2, hn_state_len()]
 * (architectures.py:419-437).  states_in must not alias states_out.
 * Replaces HybridNet.forward + EncoderBlock.forward (architectures.py:439-465, 240-252). */
int hn_unet(hn_ctx* ctx, const float* in6, const float* states_in, float* states_out, float* d_out,
            int batch, void* stream);

/* The UNet's sub-modules on their own -- what DoubleConv.forward (architectures.py:83-84), the 8x8 stride-2 convolution /
 * transposed convolution of an EncoderBlock / the decoder (:209-211, :375-382, as called in EncoderBlock.forward :252 and
 * HybridNet.forward :456) and OutConv.forward (:57-60) compute, for the channel shapes the UNet is made of.  Utility
 * entry points: the weights come as HOST pointers in PyTorch layout and are re-packed and uploaded by every call (the
 * solver never uses them; hn_unet / hn_step run the whole network from the blob of hn_load_weights).  Tensors are
 * device pointers, NCHW fp32; no domain or network needs to be loaded.
 *   hn_double_conv  x[B,cin,H,W] -> out[B,cout,H,W];  (cin, cout) in {(6,8), (8,8), (10,8), (16,8), (10,2)}, mid = cout channels;
 *                   weights = conv1.weight [cout,cin,3,3], conv1.bias [cout], slope [1] (PReLU weight, or the constant slope of
 *                   relu / leakyrelu; ignored by the smooth activations), conv2.weight [cout,cout,3,3], conv2.bias [cout], concatenated
 *   hn_conv8x8      transposed = 0: Conv2d(8, 8, 8, stride 2, padding 3), x[B,8,H,W] -> out[B,8,H/2,W/2] (H, W even), weights =
 *                   weight [8,8,8,8] (out, in, kh, kw), bias [8];  transposed = 1: ConvTranspose2d(8, 8, 8, stride 2, padding 3),
 *                   x[B,8,H,W] -> out[B,8,2H,2W], weight [8,8,8,8] (in, out, kh, kw), bias [8]
 *   hn_out_conv     Conv2d(8, 2, 1): x[B,8,H,W] -> out[B,2,H,W], weights = weight [2,8], bias [2]                                  */
int hn_double_conv(hn_ctx* ctx, const float* x, int cin, int cout, const float* weights_host, int act_kind, float* out,
                   int batch, int h, int w, void* stream);
int hn_conv8x8(hn_ctx* ctx, const float* x, const float* weights_host, int transposed, float* out, int batch, int h, int w,
               void* stream);
int hn_out_conv(hn_ctx* ctx, const float* x, const float* weights_host, float* out, int batch, int h, int w, void* stream);

/* n_iter fused solver iterations (hybridnet.py:558-584 single_step, looped as in
 * forward :654-697 / n_steps :586-623):
 *     d = HybridNet(cat[wf, 1e3*res, sigmas]); wf += d/1e3; res = L(wf) + k_sq*wf - src
 * wf, res and states (flat layout) are updated IN PLACE.
 * Optional outputs (NULL to skip):
 *     res_hist  [n_iter, B, 2, n, n]  residual after every iteration (the reference keeps them all)
 *     wf_hist   [n_iter, B, 2, n, n]  wavefield after every iteration (return_wavefields=True)
 *     st_hist   [n_iter, B, 2, L]     flat hidden state after every iteration (return_states=True)
 *     rmse_hist [n_iter, B]           per-sample residual RMSE after every iteration
 * Histories cost no copies (v7): the residual / wavefield of iteration `it` is WRITTEN into slot `it` of res_hist / wf_hist by the kernels
 * that compute it and read there by iteration it + 1 -- the reference keeps every residual for free too (it appends tensors, :676-697);
 * wf and res receive the last slot with one device-to-device copy per call.  Requires the slots not to alias wf / res; with captured
 * iterations (HN_EXP_GRAPH), several lanes or HN_OPT_HIST_COPY 1 the r1 - r6 form (in place + one copy per iteration and history) runs. */
int hn_step(hn_ctx* ctx, float* wf, float* res, float* states, const float* k_sq, const float* src,
            int src_batch, int batch, int n_iter, float* res_hist, float* wf_hist, float* st_hist,
            float* rmse_hist, void* stream);

/* ---- training step (hybridnet.py:385-413 training_step, :250-283 configure_optimizers, :172-176
 * on_after_backward).  The caller owns every training tensor: the weights as ONE flat device blob in the order of
 * hn_load_weights (PyTorch layouts inside: conv [out,in,kh,kw], transposed conv [in,out,kh,kw]), its gradient, and the
 * Adam moments -- so a data-parallel job all-reduces `grad` as a single 193 KB bucket (RCCL) between the two calls and a
 * checkpoint is those tensors.  The context supplies the architecture (depth / activation of the last hn_load_weights),
 * the spectral tables of hn_set_domain and the activation tape (grown on demand; hn_train_reserve pre-allocates).
 * fp32 throughout; gradients are accumulated in a fixed order (per-block partial sums + one reduction, no atomics):
 * bit-reproducible.  Levels without state (state_depth < depth) are handled by the caller's packing: zero-padded weights whose entries
 * it marks non-trainable in hn_adam_step. */

/* Pre-allocate the tape for `batch` samples x `n_unroll` unrolled iterations (optional). */
int hn_train_reserve(hn_ctx* ctx, int batch, int n_unroll);

/* Loss and gradients of n_unroll unrolled solver iterations started from (wf, res, states):
 *     for t < n_unroll: d = HybridNet_w(cat[wf, 1e3*res, sigmas]); wf += d/1e3; res = L(wf) + k_sq*wf - src   (single_step :558-584)
 *     loss = loss_scale * mean over (n_unroll, B, 2, n, n) of res_t^2        (training_step :403-409, loss_scale = 1e4)
 * Inputs (device, not modified): weights [hn_weight_count], wf / res [B,2,n,n], states [B,2,L], k_sq [B,1,n,n],
 * src [src_batch,2,n,n].  Outputs (device): wf_hist / res_hist [n_unroll,B,2,n,n] and st_hist [n_unroll,B,2,L] (required: the
 * lists n_steps(..., True, True) returns, :586-623 -- they are also the tape of the backward pass); loss [1];
 * grad [hn_weight_count] = d loss / d weights (overwritten); grad_wf0 / grad_res0 [B,2,n,n], grad_st0 [B,2,L] =
 * d loss / d (wf, res, states) (optional, NULL to skip).
 * Stream capture (a caller recording the step into a HIP graph): supported with the workspace in place -- call hn_train_reserve (or run
 * one eager hn_train_grad of the same shape) first; a captured call whose workspace would have to grow returns HN_ERR_STATE instead of
 * breaking the capture.  The launch tables of a captured call live in a pinned buffer of their own that eager calls never rewrite; capturing a
 * SECOND graph on the same context rewrites it (one captured training step per context at a time).  Once a call has been captured, an eager call that
 * needs a LARGER workspace (bigger batch, more unrolled iterations, another domain) returns HN_ERR_STATE instead of freeing memory the graph points into:
 * destroy the graph and call hn_train_reserve (the caller's word that no captured step is replayed any more; it may re-allocate). */
int hn_train_grad(hn_ctx* ctx, const float* weights, const float* wf, const float* res, const float* states, const float* k_sq,
                  const float* src, int src_batch, int batch, int n_unroll, float loss_scale, float* wf_hist, float* res_hist,
                  float* st_hist, float* loss, float* grad, float* grad_wf0, float* grad_res0, float* grad_st0, void* stream);

/* ABI v5.  `event` (a hipEvent_t, or NULL to clear): every later hn_train_grad on this context records it on the caller's stream BEHIND THE FORWARD SWEEP, i.e. when
 * wf_hist / res_hist / st_hist are complete and before the backward pass starts.  The reference's training_step decides from the residual of one
 * unrolled iteration which replay-buffer slots to refill (`res.pow(2).mean() < 1`, hybridnet.py:431-463) -- a host decision; with this event the host
 * takes it while the backward pass runs instead of draining the queue after it (helmnet_amd.training.Trainer).
 * `sumsq_host` (optional; pinned host memory of `sumsq_capacity` floats): the forward sweep's own table of sum over (2, N, N) of res^2, one float per
 * (iteration t, sample b) at [t * batch + b], is copied there before the event is recorded -- the refill rule needs nothing else, so no kernel of the
 * caller has to run between the two sweeps.  The sums are accumulated with float atomics (their last bits are not reproducible, like `loss`).
 * hn_train_grad returns HN_ERR_ARG if n_unroll * batch exceeds the capacity.  The caller owns event and table; not recorded / copied by a call that is
 * being captured into a HIP graph. */
int hn_train_set_forward_event(hn_ctx* ctx, void* event, float* sumsq_host, int64_t sumsq_capacity);

/* ABI v5.  Rows of the caller's replay buffer (replaybuffer.py:20-47; hybridnet.py:388-397 sample, :436-463 refill).  The reference keeps a Python list of
 * `capacity` Experience tuples; on the device that is one [capacity, row_floats[f]] fp32 array per field f (wavefield, hidden state, k_sq, residual,
 * source: n_fields <= 8), owned by the caller.  `slots`: `count` HOST integers in [0, capacity) -- they travel in the kernel arguments, so neither
 * call copies anything to the device or waits for it.
 *   hn_rows_gather : out[f][j, :] = buffers[f][slots[j], :]            (ReplayBuffer.sample's `stack`, all fields in one launch)
 *   hn_rows_scatter: buffers[f][slots[j], :] = rows[f][j, :]           (ReplayBuffer.append for a batch of slots); rows[f] == NULL writes zeros (the
 *                    wavefield and hidden state of a fresh experience, :456-458); rows_stride[f] = floats between consecutive new rows, 0 = the same row for
 *                    every slot (the one source map, :462); rows_stride == NULL means row_floats.  Slots must be distinct (as `sample` draws them).
 * HN_ERR_ARG for a slot outside the buffer, a NULL array, more than 8 fields. */
int hn_rows_gather(hn_ctx* ctx, int n_fields, const float* const* buffers, const int64_t* row_floats, int64_t capacity, const int32_t* slots,
                   int count, float* const* out, void* stream);
int hn_rows_scatter(hn_ctx* ctx, int n_fields, float* const* buffers, const int64_t* row_floats, int64_t capacity, const int32_t* slots, int count,
                    const float* const* rows, const int64_t* rows_stride, void* stream);

/* One optimiser step on caller-owned device arrays of n floats: gradient value clipping to [-clip_value, clip_value]
 * (clip_value <= 0: none; torch.nn.utils.clip_grad_value_, hybridnet.py:172-176), then torch.optim.Adam as configured by the
 * reference (hybridnet.py:250-258: betas (0.9, 0.95), L2 weight decay added to the gradient, no amsgrad):
 *     g += weight_decay * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)
 * `step` counts from 1.  `trainable` (n bytes on the device): entries with 0 are left untouched.  It may be NULL only for a PReLU network
 * with state at every level: for the parameter-free activations the blob's slope slots hold CONSTANTS (leakyrelu 0.01 ...) with zero
 * gradient, which weight decay + Adam's normalisation would move by about lr per step, and the zero padding of a level without state
 * (state_depth < depth) likewise -- helmnet_amd.training.trainable_mask builds the mask.  A NaN gradient entry stays NaN through the
 * clipping (as torch's clamp_) and makes that weight NaN: a diverged step is visible, not silently applied. */
int hn_adam_step(hn_ctx* ctx, float* weights, const float* grad, float* exp_avg, float* exp_avg_sq, const unsigned char* trainable,
                 size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, float clip_value, int64_t step,
                 void* stream);

/* Test / debugging aid: copy one tensor of the LAST unrolled iteration processed by hn_train_grad's backward sweep (i.e.
 * iteration 0 of the call) out of the library's workspace into `out` (device, at most max_floats; returns the number of floats
 * the tensor has, or a negative hn_status).  kind: 0 x_d (level input), 1 conv_signal mid, 2 out_d (skip), 3 conv_state mid,
 * 4 upsampled u_d, 5 decoder mid (level == depth: bottleneck), 6 decoder output y_d, 7 inc mid; 16 + k: the gradient of the
 * loss with respect to tensor kind k in {0, 2, 4, 6}. */
int64_t hn_train_peek(hn_ctx* ctx, int kind, int level, float* out, int64_t max_floats, void* stream);

/* Optional per-kernel timing (measurement only; bench.py's roofline block uses it).  While a
 * kernel id's bit is set in `kernel_mask`, every launch of that kernel inside hn_step / hn_unet /
 * hn_residual is bracketed by a HIP event pair recorded on the caller's stream.  Kernel ids:
 *   0 inc | 1+3d conv_signal(d) | 2+3d conv_state(d) | 3+3d down(d) | 19 bottleneck |
 *   20+2d up(d) | 21+2d decoder(d) (d=0 includes outc + wavefield update) |
 *   32 spectral column pass | 33 spectral row pass (or the dense operator) |
 *   34 the fused deepest level (conv_signal, conv_state, down, bottleneck, up, decoder of level depth-1 in one kernel;
 *      those six ids then do not occur) | 35 both spectral passes under ONE event pair
 *      | 36 inc and conv_signal_0 as one launch (HN_OPT_DC_PAIR; ids 0 and 1 then do not occur).
 * hn_profile_collect synchronises the recorded events, returns per-id total milliseconds and
 * launch counts for ids [0, n_ids) and resets the accumulators. */
#define HN_KERNEL_IDS 37
int hn_profile_enable(hn_ctx* ctx, uint64_t kernel_mask);
/* Bracket only every `every_nth` launch of a selected kernel (default 1), starting every_nth / 2 launches in.  An event pair
 * costs a few microseconds of stream gap, so timed runs sample instead of bracketing every launch. */
int hn_profile_stride(hn_ctx* ctx, int every_nth);
int hn_profile_collect(hn_ctx* ctx, double* total_ms, int64_t* count, int n_ids);
/* Shortest bracketed launch per kernel id in the interval closed by the last hn_profile_collect
 * (robust against host-side launch stalls, which inflate event-bracketed times). */
int hn_profile_min(hn_ctx* ctx, double* min_ms, int n_ids);

#ifdef __cplusplus
}
#endif
#endif /* HELMNET_HIP_H */
