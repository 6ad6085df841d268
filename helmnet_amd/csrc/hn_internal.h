// Internal declarations shared by the translation units of libhelmnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "helmnet_hip.h"

namespace hn {

constexpr int kFeat = 8;    // feature channels of the shipped net (hparams.features)
// device-side launch counters (hn_dca.hip: pair_done, hn_deepx.hip: dx_done): one 128-byte line per sample slot -- atomics on ONE line are served one after
// the other at the memory side, ~26 ns each [measured, r6: 4096 of them doubled a 106 us kernel]
constexpr int kCounterStride = 32;
constexpr int kState = 2;   // hidden-state channels per level (hparams.state_channels)
constexpr int kInCh = 6;    // [wf_re, wf_im, 1e3*res_re, 1e3*res_im, sigma_x, sigma_y]
constexpr int kMaxDepth = 6;

// A read-only planar tensor view: element (b, c, y, x) at p[b*sb + c*sc + y*W + x].
struct Src {
    const float* p;
    long sb;      // sample stride (0 broadcasts over the batch)
    long sc;      // channel stride
    float scale;  // multiplied in while staging (the reference's 1e3 * residual)
};
struct Dst {
    float* p;
    long sb;
    long sc;
};
// flag sync (hn_ctx::sync_flags): what a main-chain kernel does for the side stream on the way -- thread 0 of block 0 stores `store_epoch` to *store (signal
// memory the side stream's command processor waits on) when the kernel starts, i.e. when everything launched before it on its stream is complete, and / or,
// after its own work, waits until *wait has reached wait_epoch
struct SyncHook {
    unsigned* store = nullptr; unsigned store_epoch = 0;
    const unsigned* wait = nullptr; unsigned wait_epoch = 0;
    int* err = nullptr;   // host-mapped: a wait that gave up
};
__device__ __forceinline__ void sync_hook_begin(const SyncHook& h) {
    if (h.store != nullptr && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0)
        __hip_atomic_store(h.store, h.store_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sync_wait_ge(const unsigned* flag, unsigned epoch, int* err) {   // wrap-around safe; bounded by 2 s of the 100 MHz counter
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
        __builtin_amdgcn_s_sleep(16);
    }
}
__device__ __forceinline__ void sync_hook_end(const SyncHook& h) {
    if (h.wait != nullptr && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) sync_wait_ge(h.wait, h.wait_epoch, h.err);
}

// Weights of one DoubleConv, re-packed for scalar (SGPR) broadcast loads:
//   w1 [cin][3][3][cmid], b1 [cmid], slope (1 float), w2 [cmid][3][3][cout], b2 [cout]
struct DcW {
    const float* w1;
    const float* b1;
    const float* slope;
    const float* w2;
    const float* b2;
    int act;   // hn_act
    const float* w1q;   // conv1 again as [cin][2 channel halves][3][3][4] (8-channel DoubleConvs only; hn_dcv.hip)
    const float* wa;    // conv1 again as [cin][3 kx][3 ky][8] with the input scales folded in (8-channel DoubleConvs only; hn_dca.hip)
    const float* wa2;   // conv2 likewise: [8 cm][3 kx][3 ky][8]
};

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (private L2 each; MI355X_MICROARCH.md, "Workgroup
// dispatch"), so with the plain grid order the tiles of one XCD are never neighbours and every halo row / column a tile shares
// with the next one is fetched into two L2s.  The linear workgroup id is remapped such that each XCD walks a contiguous run of
// tiles ((id % 8) * (T / 8) + id / 8, bijective when T % 8 == 0; identity otherwise).  [measured] conv_signal0 74.6 -> 66.4 us,
// decode0 75.2 -> 71.6 us.  Placement is a speed matter only: nothing depends on it.
struct TileId { int x, y, z; };
__device__ __forceinline__ TileId xcd_tile() {
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * (int)gridDim.z;
    int id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
#ifndef HN_NO_XCD
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
#endif
    const int q = id / gx;
    return TileId{id - q * gx, q % gy, q / gy};
}

// The reference's smooth activations (architectures.py:22-39: nn.CELU(), nn.Tanh(), nn.GELU(), nn.Tanhshrink(),
// nn.Softplus() with their default arguments), evaluated in fp32.  Kernels carry them as a separate template
// instance (GEN): the piecewise-linear instances used by the shipped checkpoint do not change by one instruction.
__device__ __forceinline__ float act_general(float x, int act) {
    switch (act) {
        case HN_ACT_CELU: return fmaxf(x, 0.f) + fminf(0.f, expm1f(x));
        case HN_ACT_TANH: return tanhf(x);
        case HN_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
        case HN_ACT_TANHSHRINK: return x - tanhf(x);
        case HN_ACT_SOFTPLUS: return x > 20.f ? x : log1pf(expf(x));
        default: return x;
    }
}
// d act(x) / dx (the backward kernels); GEN = false: the piecewise-linear kinds (torch: the slope applies at x == 0)
template <bool GEN>
__device__ __forceinline__ float act_grad(float x, int kind, float slope) {
    if (!GEN) return x > 0.f ? 1.f : slope;
    switch (kind) {
        case HN_ACT_CELU: return x > 0.f ? 1.f : expf(x);
        case HN_ACT_TANH: { const float t = tanhf(x); return 1.f - t * t; }
        case HN_ACT_GELU: return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
        case HN_ACT_TANHSHRINK: { const float t = tanhf(x); return t * t; }
        case HN_ACT_SOFTPLUS: return x > 20.f ? 1.f : 1.f / (1.f + expf(-x));
        default: return x > 0.f ? 1.f : slope;   // prelu / relu / leakyrelu
    }
}
// 8x8 stride-2 conv / transposed conv weights re-packed [cin][8][8][cout], bias [cout].
struct K8W {
    const float* w;
    const float* b;
};

struct SpecTables {
    int n = 0;
    bool pow2 = false;
    // pow2 path (all device pointers)
    float2* tw = nullptr;      // exp(-2 pi i m / n), m in [0, n)
    float* k1 = nullptr;       // fp32 wavenumber grid, Nyquist at -pi
    float* k2 = nullptr;       // -(k1*k1) evaluated in fp32 like the reference
    float2* a = nullptr;       // PML first-derivative coefficient  (-gamma' / gamma^3)
    float2* b = nullptr;       // PML second-derivative coefficient (1 / gamma^2)
    // prime-factor path (n = P * Q, P in {3, 5}, Q a power of two): Q-point twiddles and the derivative multipliers in
    // the (k1, k2) order of the Good-Thomas output map; a / b above stay in natural order
    int pfa_p = 0, pfa_q = 0;
    float2* tw_q = nullptr;    // exp(-2 pi i m / Q)
    float* k1_pfa = nullptr;   // [P][Q]
    float* k2_pfa = nullptr;
    // dense fallback (any other n): complex n x n operator, stored transposed
    float2* dense_t = nullptr; // dense_t[m*n + j] = M[j][m]
    float2* dense_adj_t = nullptr; // the same for the adjoint operator M^H (training, hn_train.hip)
    float* sigmas = nullptr;   // [2, n, n]
};

}  // namespace hn

struct hn_ctx {
    int device = 0;
    std::string err;
    // network
    bool have_weights = false;
    int depth = 0;
    int act_kind = HN_ACT_PRELU;
    float* wdev = nullptr;  // all re-packed weights
    hn::DcW inc{}, sig[hn::kMaxDepth]{}, st[hn::kMaxDepth]{}, dec[hn::kMaxDepth + 1]{};
    hn::K8W down[hn::kMaxDepth]{}, up[hn::kMaxDepth]{};
    // MFMA A-operand fragments (hn_mfma.hip): [cin][3][64] per 3x3 conv, [8][8][64] per 8x8 conv
    float* fragdev = nullptr;
    const float *f_inc[2]{}, *f_sig[hn::kMaxDepth][2]{}, *f_dec[hn::kMaxDepth + 1][2]{};
    const float *f_down[hn::kMaxDepth]{}, *f_up[hn::kMaxDepth]{};
    const float* f_down2[hn::kMaxDepth]{};   // the 8x8 stride-2 convolution again in the column-pair packing of hn_deepx.hip: [8 ci][2 h][10 kx'][64]
    const float *f_dec0c = nullptr, *dec0c_b = nullptr;   // decode[0] conv2 composed with the out-conv: [8][5][64] row-triple fragments, bias [2]
    const float* v_dec0c = nullptr;   // the same composed convolution for the vector-pipe kernel: [8 cm][3][3][2] (hn_dcv.hip)
    const float* f_st[hn::kMaxDepth][2]{};   // conv_state (2 output channels in rows 0..3 of M): [10][3][64], [2][3][64] (hn_deep.hip)
    bool deep_attr_set = false, pfa_attr_set = false, pfa_adj_attr_set = false;
    // arithmetic of the UNet convolutions (hn_set_unet_precision; default from HN_UNET_IMPL at hn_create only)
    int precision = HN_PREC_FP32;
    // tuning knobs (hn_set_option; defaults from HN_STREAMS / HN_SIDE_STREAM / HN_GRAPH at hn_create only)
    int opt_lanes = 1;         // hn_step pipelines this many sub-batches on internal streams
    int opt_side_stream = 1;   // conv_state kernels on a side stream, overlapping the deep levels
    int opt_side_priority = 0;     // HN_SIDE_PRIORITY: 0 (default) probe (above); 1 / 2 / 3 (A/B): the lowest- / normal- / highest-priority candidate without probing
    int opt_defer_join = 0;    // HN_DEFER_JOIN=1 (A/B): hn_step joins the side stream behind the NEXT iteration's input layer instead of at the end of
                               // the UNet -- measured no gain (1902 / 1927 / 1938 vs 1932 / 1933 / 1946 it/s, r3): the ~6 us bubble is the event
                               // packet itself, wherever it sits
    int opt_graph = 0;         // hn_step replays captured iterations (HIP graph; n > 1: n iterations per graph) instead of launching
                               // ~25 kernels per iteration.  Off by default: [measured] replay is 4 % SLOWER (586 vs 563 us per
                               // iteration at 256^2 x 32, graphs of 1, 2 or 8 iterations alike) -- the host keeps ahead of the GPU
                               // anyway (~100 us of launch calls per 560 us iteration) and the graph's node-to-node dependencies
                               // cost more than in-order stream launches
    int opt_pfa = 1;           // prime-factor FFT for n = 3 * 2^k, 5 * 2^k (0: dense operator, A/B; takes effect at hn_set_domain)
    int opt_radix16 = 1;       // 256-point transforms as two register-resident radix-16 passes (0: the radix-4 Stockham kernels)
    int opt_cols_t = 1;        // 256-point column pass through an LDS transpose: 0 the r2 kernel (16-byte global accesses), 1: 16 columns per block, 2: 32
    bool cols_t_attr_set = false, cols512_attr_set = false;
    int opt_deep = 2;          // HN_OPT_DEEP: 1 deepest encoder level + bottleneck + deepest decoder level as one per-sample LDS kernel (hn_deep.hip, deepest level 32^2);
                               // 2 (default) the last one or two levels + bottleneck as ONE launch with eight workgroups per sample where it applies (hn_deepx.hip:
                               // a 64-wide level, with a 32-wide one below it or not), else as 1; 0 layer by layer
    unsigned* dx_flags = nullptr;   // hn_deepx.hip: [sample slot][8 bands][8 hand-offs] epoch words
    unsigned* dx_done = nullptr;    // [sample slot][kCounterStride]: bands that have ended, ever (the launch's epoch is derived from it on the device)
    bool deepx_attr_set = false;
    int opt_dc_valu = 4;       // fp32 DoubleConvs of the big levels on the packed vector FMA: 0 none, 1 inc + decoder (hn_dcv.hip), 2 all three;
                               // 3 / 4 (default): the same two / three on the hand-scheduled kernel (hn_dca.hip)
    int opt_dc_pair = 1;       // HN_OPT_DC_PAIR: inc and conv_signal_0 as one launch with per-tile flags (hn_dca.hip, k_dc_asm_pair); hn_step's single-lane eager path
    unsigned* pair_flags = nullptr;   // one word per level-0 tile of the reserved batch: the epoch of the launch whose inc block wrote that tile
    long pair_flags_cap = 0;
    unsigned pair_epoch = 0;          // launches of the merged kernel outside stream capture (their epoch travels in the arguments, top bit set)
    unsigned* pair_done = nullptr;    // [sample slot][kCounterStride]: conv_signal blocks of the merged launch that have ended, ever (the launch's epoch is derived from it on the device)
    // Side stream <-> main stream without event packets ("flag sync", hn_unet.hip).  An event record holds the recording stream for ~7 us, a stream-wait for ~6,
    // and so does ANY extra kernel, however small [measured, r5: profiles/r5_side_sync.txt].  So between the iterations of one hn_step call the hand-overs
    // ride on kernels the main chain launches anyway (SyncHook):
    //   join     a one-thread kernel behind the hidden-state kernels stores the epoch to word 32; ONE thread of up_0 polls it after its own work (one
    //            spinning thread cannot keep the side stream's kernels off the CUs; a grid of polling conv_signal blocks could);
    //   release  (where the deep kernel exists) its first thread stores the epoch to word 0 -- everything before it on the main stream is complete --
    //            and a one-wave gate kernel in front of the hidden-state kernels polls it.  The gate is resident from the end of one iteration's
    //            hidden-state kernels to the next release and costs the level-0 kernels a block slot (decode_0 70 -> 74 us: half of what the missing
    //            event packet saves); hipStreamWaitValue64 is itself a spinning kernel here and slower.
    // Every store is enqueued before the kernel that waits for it (a tool that runs one kernel at a time in submission order cannot deadlock), every wait is
    // bounded (2 s, then hn_step fails).
    int opt_hist_copy = 0;     // HN_OPT_HIST_COPY: 0 (default) hn_step writes the residual / wavefield of iteration `it` straight into the caller's history slot and
                               // reads it there in iteration it + 1; 1: in place in the caller's wf / res plus one device-to-device copy per iteration and history (r1 - r5; A/B)
    int opt_state_kernel = 1;  // HN_OPT_STATE_KERNEL: 1 the hidden-state DoubleConvs (10 -> 2 -> 2) of the levels >= 64 wide on the streaming kernel (hn_cs.hip), 0 k_double_conv
    int opt_side_sync = 1;     // HN_OPT_SIDE_SYNC: 1 device words between the iterations of one hn_step call, 0 events everywhere
    unsigned* sync_flags = nullptr;   // device, 256 words: 0 release / 32 join of hn_step; 64, 96 forward and 128, 160 backward sweep of hn_train_grad
    unsigned sync_epoch = 0;   // (compared wrap-around safe)
    const float* step_wf_in = nullptr;   // argument of the NEXT decode_0 launch: the wavefield its update starts from when that is not the buffer it writes
                               // (hn_step's zero-copy wavefield history: reads slot it - 1, writes slot it); set and cleared by unet_forward
    int dca_dec_pad = 0;       // dynamic LDS of the next decode_0 launch on hn_dca.hip: 7168 (3 blocks per CU) while the gate kernel is resident, else 0
    int* sync_err = nullptr;          // host-mapped: a bounded device-side wait that gave up stores its code here (sticky; checked by hn_step)
    int* sync_err_dev = nullptr;
    // the sigma channels' share of the input layer's conv1 as a per-domain map (hn_dca.hip: SigmaMap; built by hn_load_weights / hn_set_domain, whichever comes second)
    float2* inc_sigma_map = nullptr;   // [4 channel pairs][n][n]
    int inc_sigma_band = 0;            // the map is zero farther than this many pixels from the border
    float inc_w_sigma[144]{};          // inc.conv1.weight[:, 4:6] ([8][2][3][3], host copy)
    int opt_inc_sigma_map = 1;         // HN_OPT_INC_SIGMA_MAP: 0 = the input layer convolves its six channels as in r5
    const float* zero_page = nullptr;   // 256 zero bytes (out-of-image float4s of the LDS-direct staging loads, hn_dca.hip)
    const float* outc_w = nullptr;  // [8][2]
    const float* outc_b = nullptr;  // [2]
    // domain
    hn::SpecTables tab;
    int64_t state_len = 0;
    int64_t state_off[hn::kMaxDepth]{};
    // workspace
    int cap_batch = 0;
    float* buf_a[hn::kMaxDepth + 1]{};  // x_d, later reused for the upsampled tensor u_d
    float* buf_o[hn::kMaxDepth]{};      // out_d (skip connections)
    float* buf_y[hn::kMaxDepth + 1]{};  // decoder outputs y_d
    float* st_tmp = nullptr;            // second flat state buffer for hn_step ping-pong
    // hn_step pipelines sub-batches on internal streams (samples are independent): while one
    // sub-batch walks the small, latency-bound UNet levels the other one keeps the CUs busy
    // conv_state kernels run on a side stream per pipeline lane (HN_SIDE_STREAM, hn_step only)
    struct SideLane {
        hipStream_t stream = nullptr; hipEvent_t ev[hn::kMaxDepth]{}; hipEvent_t done = nullptr;
        bool pending = false;   // `done` has been recorded on the side stream and not been waited for yet (deferred join, hn_step)
    };
    SideLane side[8];          // (lane 0's stream belongs to picks[0]: side_stream_for)
    // Side-stream picker (hn_api.hip: side_stream_for).  HIP streams are dealt onto a few hardware queues, and two streams on ONE queue run in submission order:
    // a library side stream that lands on the caller's queue silently loses its overlap (inference -9 %) or serialises behind capped launches (training step
    // 7.2 -> 21 ms) -- which streams share a queue depends on how many streams of which priority the PROCESS has created, not on anything the library controls
    // [measured, r4: profiles/r4_caller_stream.txt, r4_wg_prio.txt, r4_stream_probe.txt].  So per (slot, caller stream) the library PROBES its candidates once:
    // a 200 us spin kernel on the caller's stream, an empty one on the candidate -- did the second finish first? -- and keeps the first candidate that overlaps.
    struct SidePick {
        hipStream_t cand[4]{};
        struct Known { hipStream_t ref[3]{}; int nref = 0; int chosen = 0; };
        std::vector<Known> known;     // every reference-stream set probed so far and the candidate that overlapped with it
        int last_chosen = 0;
    } picks[8];   // 0: hn_step's side stream; 1, 2: hn_train_grad lanes;
                                                                                                               // 3 .. 6: two pipeline lanes (chain 0, chain 1, side 0, side 1: mutually overlapping)
    int n_streams = 0;         // internal streams created so far
    hipStream_t sub_stream[8]{};
    hipEvent_t ev_fork = nullptr, ev_join[8]{}, ev_stagger[8]{};
    int* it_counter = nullptr;  // device, one per lane: iterations done in the running hn_step (row index of rmse_hist)
    // captured iterations of hn_step, keyed by every pointer / size / mode baked into the kernel arguments
    struct StepGraph {
        const void *wf = nullptr, *res = nullptr, *states = nullptr, *k_sq = nullptr, *src = nullptr, *rmse = nullptr;
        int src_batch = 0, batch = 0, precision = 0, lanes = 0, side = 0;
        hipGraphExec_t exec[2] = {nullptr, nullptr};  // [0]: states user -> tmp, [1]: tmp -> user
        long last_use = 0;
    };
    std::vector<StepGraph> graphs;
    long graph_clock = 0;
    hipStream_t cap_stream = nullptr;              // iterations are captured here (the caller's stream may be the legacy default stream)
    long graph_replays = 0, eager_iterations = 0, graphs_captured = 0, probes_run = 0, train_fwd_events = 0, flag_sync_iterations = 0;  // diagnostics (hn_get_counter)
    // training workspace (hn_train.hip): activation tape of the unrolled iterations, gradient buffers, partial sums
    struct TrainWs {
        int batch = 0, n_unroll = 0, n = 0, depth = 0;
        float* tape = nullptr;       // [n_unroll][step_floats]
        size_t step_floats = 0;
        size_t o_zinc = 0, o_x[hn::kMaxDepth + 1]{}, o_zsig[hn::kMaxDepth]{}, o_out[hn::kMaxDepth]{}, o_zst[hn::kMaxDepth]{},
               o_u[hn::kMaxDepth]{}, o_zdec[hn::kMaxDepth + 1]{}, o_y[hn::kMaxDepth + 1]{};
        float* gbuf = nullptr;       // gradient buffers, carved below
        float *g_x[hn::kMaxDepth + 1]{}, *g_out[hn::kMaxDepth]{}, *g_u[hn::kMaxDepth]{}, *g_y[hn::kMaxDepth + 1]{};
        float *gz[3 * hn::kMaxDepth + 2]{};   // gradient of every DoubleConv's mid tensor (it feeds conv1's weight gradient at the end of the iteration)
        float *g_wf[2]{}, *g_res = nullptr, *g_st[3]{};   // g_st: read one, write the next, the third is still read by the previous iteration's weight gradients
        // two sets of the per-iteration gradient buffers above (g_x .. gz point into the set of the iteration being processed): the weight-gradient
        // launches of iteration t run on a stream of their own beside the backward chain of iteration t - 1, which writes the other set
        struct GSet { float *g_x[hn::kMaxDepth + 1]{}, *g_y[hn::kMaxDepth + 1]{}, *g_out[hn::kMaxDepth]{}, *g_u[hn::kMaxDepth]{}, *gz[3 * hn::kMaxDepth + 2]{}; } gset[2];
        hipStream_t wg_stream = nullptr;
        hipEvent_t wg_ready[2]{}, wg_done[2]{};
        bool wg_pending[2]{};
        hipEvent_t st_fork = nullptr, st_done = nullptr;   // forward sweep: the hidden-state DoubleConvs of an iteration on wg_stream (idle then) beside its decoder
        bool st_pending = false;
        unsigned wg_flag_epoch[2]{}, bwd_release_epoch = 0;   // flag sync of the training step (hn_train.hip): join epochs of the two job sets; release still to be stored
        float* part = nullptr;       // [640 rows][blob]: per-block sums of the weight-gradient kernels, all layers and iterations
        size_t part_floats = 0;
        double* slope_part = nullptr; // [3 depth + 2 DoubleConvs][slope_stride]: per-block sums of the PReLU-slope gradients (float64)
        size_t slope_stride = 0;
        float* w3 = nullptr;         // 3x3 weights packed [cin][9][cout]: forward arrangement, then backward-data
        float* k8 = nullptr;         // 8x8 weights as fp32 matrix-core fragments: [depth][4][4096]
        float* f3 = nullptr;         // 3x3 weights of the 8-channel DoubleConvs as fp32 matrix-core fragments [cin][3][64] (forward pass)
        float* zero8 = nullptr;      // 8 zeros (bias of the backward-data convolutions)
        float* sumsq = nullptr;      // [n_unroll][batch] per-sample sum of squared residuals
        // job tables of the batched weight-gradient launches: one region per unrolled iteration, filled on the host (pinned), copied
        // behind the iteration's backward-data chain, read by the three launches that follow the copy
        unsigned char *jobs_host = nullptr, *jobs_dev = nullptr;
        size_t jobs_region = 0;      // bytes per iteration
        // the pinned side rotates over kJobSets copies, one per call: the host may enqueue that many calls ahead of the GPU before it
        // has to wait for a table to have been copied out (a single copy made every call wait for the previous call's LAST launch)
        static constexpr int kJobSets = 4;
        hipEvent_t jobs_copied[kJobSets]{};   // recorded behind the last copy of the call that used the set
        bool jobs_in_flight[kJobSets]{};
        int jobs_set = 0, jobs_rows = 0;      // set of the current call; iterations per set
        bool captured = false;       // a call using this workspace has been captured into a HIP graph: growing it needs an explicit hn_train_reserve
        int last_batch = 0;          // samples of the last hn_train_grad call in this workspace (hn_train_peek)
        int sumsq_batch = 0;         // samples per row of sumsq (lane 0 holds the whole batch's rows)
    } tr, tr_b;                      // tr_b: the second half of the batch when hn_train_grad runs as two lanes
    int opt_train_overlap = 2;     // HN_OPT_TRAIN_OVERLAP: weight-gradient launches on a side stream beside the next iteration's backward chain.  0: in line;
                                   // 1: side stream, same launches ([measured, r4] no gain: 9.58 vs 9.61 ms at batch 32 -- the overlap is real, 2.3 ms of kernel time per
                                   // step run concurrently, but the weight-gradient blocks hold the CUs' LDS and the chain's kernels slow down by as much); 2 (default):
                                   // side stream AND the launches capped at ~2 blocks per CU where the chain is latency-bound (9.13 -> 8.77 ms; see hn_train_grad)
    int opt_train_fused = 55;       // HN_OPT_TRAIN_FUSED: bit 0 the forward pass's 8-channel DoubleConvs as fused matrix-core launches with a z-store epilogue,
                                   // bit 1 the backward-data pass of a big level's DoubleConv as one tiled launch (k_dc_bwd_tile), bit 2 the hidden-state
                                   // DoubleConvs of all levels as one launch per direction (k_dc_state_batch)
    hipEvent_t train_fwd_event = nullptr;   // caller-owned: recorded behind the forward sweep of hn_train_grad (hn_train_set_forward_event)
    float* train_fwd_sumsq = nullptr;       // caller-owned pinned host floats: the [n_unroll, batch] sums of res^2 land here before that event
    int64_t train_fwd_sumsq_cap = 0;
    int opt_train_lanes = 1;       // HN_OPT_TRAIN_LANES: 2 = the halves of the batch as two chains on two streams (measured: no gain)
    hipStream_t train_stream = nullptr;            // lane 1 (created on first use)
    hipEvent_t train_fork = nullptr, train_join = nullptr;
    // optional per-kernel timing with HIP events on the caller's stream (hn_profile_*)
    uint64_t prof_mask = 0;
    int prof_stride = 1;          // bracket every prof_stride-th launch of a selected kernel
    long prof_seen[64]{};
    struct ProfRec { int id; hipEvent_t a, b; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[64]{};
    double prof_min[64]{}, prof_min_last[64]{};  // shortest bracketed launch (robust to host stalls): running / last collect
    long prof_cnt[64]{};
};

namespace hn {

int fail(hn_ctx* ctx, int code, const char* fmt, ...);
void clear_step_graphs(hn_ctx* ctx);  // captured iterations bake in pointers, sizes and the precision mode
void set_global_error(const char* msg);

#define HN_HIP(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return hn::fail((ctx), HN_ERR_HIP, "%s failed: %s (%s:%d)", #call,              \
                            hipGetErrorString(e_), __FILE__, __LINE__);                     \
    } while (0)

// Kernel ids of the per-iteration launch sequence (hn_profile_*).
enum KernelId : int {
    KID_INC = 0,          // DoubleConv 6->8->8 at level 0
    KID_SIG0 = 1,         // + 3*d : conv_signal DoubleConv 10->8->8 at level d
    KID_STATE0 = 2,       // + 3*d : conv_state  DoubleConv 10->2->2 at level d
    KID_DOWN0 = 3,        // + 3*d : 8x8 stride-2 conv at level d
    KID_BOTTLENECK = 19,  // DoubleConv 8->8->8 at level depth
    KID_UP0 = 20,         // + 2*d : 8x8 stride-2 transposed conv producing level d
    KID_DEC0 = 21,        // + 2*d : decoder DoubleConv 16->8->8 at level d (d = 0: + outc + wf update)
    KID_SPEC_COLS = 32,   // spectral column pass
    KID_SPEC_ROWS = 33,   // spectral row pass + residual terms (or the dense operator)
    KID_DEEP = 34,        // deepest level in one per-sample kernel: conv_signal, conv_state, down, bottleneck, up, decoder
    KID_SPEC_PAIR = 35,   // one bracket around both spectral passes (the HBM-bound part of the path as a whole)
    KID_INC_SIG0 = 36,    // inc and conv_signal_0 as one launch (hn_dca.hip, k_dc_asm_pair); KID_INC and KID_SIG0 then do not occur
    KID_COUNT = 37
};

// RAII event pair around one launch when that kernel id is selected by hn_profile_enable.
// Selects the context's device for the duration of an entry point and restores the caller's (the library must not
// change the current device under PyTorch).
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(const hn_ctx* ctx) {
        int cur = -1;
        if (ctx && hipGetDevice(&cur) == hipSuccess && cur != ctx->device && hipSetDevice(ctx->device) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// Energy / time experiment only (tools/energy_probe.py; -DHN_EXP_REPEAT builds): launch the kernel(s) of one id `count` times instead of once -- the
// difference to the plain loop in time and in board energy is what one launch costs.  count 0 skips the kernel.  The results of such a run are not the solver's.
#ifdef HN_EXP_REPEAT
extern int g_exp_repeat[64];
#define HN_REP(kid) for (int r_ = g_exp_repeat[kid]; r_ > 0; --r_)
#else
#define HN_REP(kid)
#endif
struct ProfScope {
    hn_ctx* c;
    hipStream_t s;
    int slot = -1;
    ProfScope(hn_ctx* ctx, int id, hipStream_t st);
    ~ProfScope();
};

// ---- spectral (hn_spectral.hip) ----
int spec_build(hn_ctx* ctx, int n, int pml, double sigma_max, double k);
void spec_free(SpecTables& t);
// out = L(wf) [+ ksq*wf - src]; `accum_sumsq` (nullable) receives sum over (c,h,w) of out^2 per sample, in row
// `*it_counter - 1` of a [.., sumsq_stride] table when `it_counter` is given (the column pass increments it: one
// captured iteration can then be replayed for every row of the RMSE history), else in row 0.
int spec_apply(hn_ctx* ctx, const float* wf, float* out, const float* ksq, const float* src, int src_batch,
               int batch, float* accum_sumsq, hipStream_t s, int* it_counter = nullptr, int sumsq_stride = 0);
// out = L^H(g) + ksq * g [+ add]: the adjoint (conjugate transpose) of the residual operator, i.e. the vector-Jacobian product
// of hn_residual with respect to the wavefield (training).  `add` may alias `out`.
int spec_adjoint(hn_ctx* ctx, const float* g, float* out, const float* ksq, const float* add, int batch, hipStream_t s);
void train_free(hn_ctx* ctx);   // hn_train.hip

// ---- matrix-core kernels (hn_mfma.hip) ----
void pack_frag_3x3(const float* w_oihw, int cin, float* dst);  // -> [cin][3][64]
size_t frag_3x3_split_floats(int cin);                          // storage of the split-bf16 twin, in floats
void pack_frag_3x3_split(const float* w_oihw, int cin, float* dst);  // -> [ceil(cin/8)][3 dy][3 parts][64][8 bf16]
size_t frag_3x3_half_floats(int cin);
void pack_frag_3x3_half(const float* w_oihw, int cin, float* dst);   // -> [ceil(cin/8)][3 dy][64][8 half]
// conv2 [8][8][3][3] (+ bias) composed with the 1x1 out-conv [2][8] (+ bias): fragments [8 cm][5 variants][64] of the row-triple
// packing (frag != nullptr) and / or the composed bias [2] (bias != nullptr)
void pack_frag_outc3x3(const float* w2, const float* b2, const float* wo, const float* bo, float* frag, float* bias);
void pack_frag_down(const float* w_oihw, float* dst);          // -> [8][8][64]
void pack_frag_up(const float* w_iohw, float* dst);            // -> [8][2][4][64]
size_t k8_split_floats();
size_t k8_half_floats();
void pack_frag_down_x16(const float* w_oihw, float* dst_split, float* dst_half);
void pack_frag_up_x16(const float* w_iohw, float* dst_split, float* dst_half);
// kind: 0 inc (2+2+2 ch), 1 conv_signal (8+2), 2 bottleneck (8), 3 decoder (8+8; final_epi adds outc + wf update)
int launch_dc8(hn_ctx* ctx, int kind, Src a, Src b, Src c, Dst out, const DcW& w, const float* frag1, const float* frag2,
               bool final_epi, float* d_out, float* wf, int H, int W, int batch, hipStream_t s);
// training forward: the fused matrix-core DoubleConv with the pre-activation mid tensor stored to `z` ([B, 8, H, W]); fragments as pack_frag_3x3
bool dc8_tape_applies(int H, int W);
// Backward-data pass of an 8-channel DoubleConv (cin -> 8 -> 8) on the fp32 matrix core (hn_mfma.hip, k_dc_bwd_mfma_p): g_z = conv2^T(g) * act'(z),
// g_in = conv1^T(g_z), one launch, the g_z tile (with its halo) in LDS.  Channel groups of g_in as in the forward concatenation.
struct McBwdDst { float* p; long sb, sc; int nch; float scale; int accum; };   // p == nullptr: the group is discarded
struct McBwd {
    const float* g; long g_sb, g_sc;       // d loss / d output, 8 channels
    const float* a1;                       // conv2^T as A fragments [8 c2][3][64]
    const float* a2;                       // conv1^T as A fragments [passes][8 cm][3][64], pass p = input channels 8p .. 8p + 7 of the forward conv1
    const float* z; long z_sb, z_sc;       // the tape's pre-activation mid tensor, 8 channels
    float* gz; long gz_sb, gz_sc;          // g_z (8 channels): the weight-gradient kernels read it
    const float* slope; int act;
    double* slope_part;                    // PReLU: row [tile] += sum over the tile of conv2^T(g) * min(z, 0); nullptr otherwise
    McBwdDst dst[3];
};
// The hidden-state DoubleConv of the same level (new_state = DC(cat[out, state]), 10 -> 2 -> 2) riding in the decoder's backward launch: both read gradients on
// the same tile grid, and d loss / d out is the SUM of the decoder's skip gradient and conv_state's -- one store instead of a launch that opens the sum.
struct McBwdAux {
    const float* g; long g_sb, g_sc;       // d loss / d new state, 2 channels
    const float* a1;                       // conv2^T fragments [2][3][64]
    const float* a2;                       // conv1^T fragments [2 passes][2][3][64]: pass 0 -> d / d out (added to the decoder's channels 8 .. 15), pass 1 -> d / d old state
    const float* z; long z_sb, z_sc;       // 2 channels
    float* gz; long gz_sb, gz_sc;
    const float* slope; double* slope_part;
    McBwdDst dst;                          // d loss / d old state
};
int launch_dc8_bwd_aux(hn_ctx* ctx, const McBwd& a, const McBwdAux& x, int H, int W, int batch, hipStream_t s);
bool dc8_bwd_applies(int H, int W);
int dc8_bwd_tiles(int H, int W, int batch);   // rows of slope_part a launch adds to
int launch_dc8_bwd(hn_ctx* ctx, const McBwd& a, int cin, int H, int W, int batch, hipStream_t s);
int launch_dc8_tape(hn_ctx* ctx, int kind, Src a, Src b, Src c, Dst out, const float* frag1, const float* b1, const float* slope, const float* frag2,
                    const float* b2, int act, float* z, int H, int W, int batch, hipStream_t s);
void launch_down(const hn_ctx* ctx, Src in, Dst out, const float* frag, const float* bias, int Hin, int Win, int batch, hipStream_t s,
                 SyncHook hook = SyncHook{});   // hook: fp32 matrix-core kernels only (k_down_mfma)
void launch_up(const hn_ctx* ctx, Src in, Dst out, const float* frag, const float* bias, int Hin, int Win, int batch, hipStream_t s, bool accumulate = false,
               SyncHook hook = SyncHook{});   // hook: fp32 matrix-core kernels only (k_up_mfma)

// ---- vector-pipe DoubleConv of the big levels (hn_dcv.hip) ----
void pack_valu_q(const float* w_oihw, int cin, float* dst);            // conv1 [8][cin][3][3] -> [cin][2][9][4]
void pack_outc3x3_valu(const float* w2, const float* wo, float* dst);   // conv2 composed with the out-conv -> [8 cm][3][3][2]
bool dc_valu_applies(const hn_ctx* ctx, int act, Src a, Src b, Src c, int kind, int H, int W);
void launch_dc_valu(hn_ctx* ctx, int kind, Src a, Src b, Src c, Dst out, const DcW& w, bool final_epi, float* d_out, float* wf, int H,
                    int W, int batch, hipStream_t s);

// ---- the same with a hand-scheduled conv1 loop and LDS-direct staging (hn_dca.hip) ----
void pack_dca(const float* w_oihw, int cin, const float* scale, float* dst);   // conv1 [8][cin][3][3] -> [cin][3 kx][3 ky][8]
bool dc_asm_applies(const hn_ctx* ctx, int act, Src a, Src b, Src c, int kind, int H, int W);
void launch_dc_asm(hn_ctx* ctx, int kind, Src a, Src b, Src c, Dst out, const DcW& w, bool final_epi, float* d_out, float* wf, int H, int W,
                   int batch, hipStream_t s);
// ---- hidden-state DoubleConv as a streaming kernel (hn_cs.hip) ----
bool conv_state_applies(const hn_ctx* ctx, const DcW& w, Src a, Src b, Dst out, int H, int W);
void launch_conv_state(hn_ctx* ctx, int n, const Src* a, const Src* b, const Dst* out, const DcW* w, const int* H, const int* W, int batch, hipStream_t s);   // n levels, one launch
// inc and conv_signal_0 as ONE launch with a flag per tile (k_dc_asm_pair): x0_out / x0 are the same tensor as inc's output and conv_signal's input
bool dc_asm_pair_applies(const hn_ctx* ctx, Src wf, Src res, Src sig, Src x0, Src st, int H, int W, int batch, int ws_off);
void launch_dc_asm_pair(hn_ctx* ctx, Src wf, Src res, Src sig, Dst x0_out, Src x0, Src st, Dst out0, int H, int W, int batch, int ws_off, bool capturing, hipStream_t s);

// ---- deep levels in one per-sample kernel (hn_deep.hip) ----
void pack_frag_3x3_c2(const float* w_oihw, int cin, float* dst);  // 2 output channels -> [cin][3][64], rows 4..15 of M zero
bool deep_applies(const hn_ctx* ctx);
int launch_deep(hn_ctx* ctx, const float* x_in, long x_sb, const float* st_in, float* st_out, long st_sb, long st_sc, float* y_out,
                long y_sb, int batch, hipStream_t s, SyncHook hook = SyncHook{});

// ---- the last one or two levels + bottleneck with eight workgroups per sample (hn_deepx.hip) ----
void pack_frag_down2(const float* w_oihw, float* dst);   // -> [8 ci][2 h][10 kx'][64]
int deepx_levels(const hn_ctx* ctx, int batch);   // 2 / 1 / 0: levels the kernel would fuse for the current domain, options and precision
int launch_deepx(hn_ctx* ctx, int levels, const float* states_in, float* states_out, int ws_off, int batch, hipStream_t s, SyncHook hook = SyncHook{});

// ---- unet (hn_unet.hip) ----
// One HybridNet forward.  wf/res/sigma sources are generic views; if wf_update != nullptr the
// wavefield is updated (wf_update = wf_prev + d / 1e3; in place when wf_prev is nullptr) by the last kernel; if d_out != nullptr d is stored.
int unet_forward(hn_ctx* ctx, Src in_wf, Src in_res, Src in_sig, const float* states_in, float* states_out,
                 float* d_out, float* wf_update, int batch, hipStream_t s, int ws_off = 0, hipEvent_t after_down0 = nullptr,
                 hn_ctx::SideLane* side_lane = nullptr, bool defer_join = false, const float* wf_prev = nullptr);
// wf_prev (with wf_update): the update reads the old wavefield THERE and writes wf_update (in_wf should view the same tensor); nullptr: in place
// make stream s wait for the hidden-state kernels of the previous unet_forward(..., defer_join = true) on this lane
int side_join(hn_ctx* ctx, hn_ctx::SideLane* side_lane, hipStream_t s);
bool side_flags_apply(hn_ctx* ctx, hipStream_t s);
// flag sync, the side stream's halves (hn_unet.hip): a one-wave kernel that holds stream s until *flag has reached epoch / a one-thread kernel that stores it
int ensure_sync_words(hn_ctx* ctx);
int build_inc_sigma_map(hn_ctx* ctx);   // (hn_api.hip; needs weights and domain, synchronises)
int check_async(hn_ctx* ctx, const char* who);   // HN_ERR_STATE once a bounded device-side wait has given up (sticky word: 1 side-stream flag, 2 merged level-0 launch, 3 deep kernel)
int zero_async(hn_ctx* ctx, void* p, size_t bytes, hipStream_t s);   // (hn_api.hip: a kernel of this library, never hipMemsetAsync, on a caller's stream)
void launch_sync_gate(hn_ctx* ctx, const unsigned* flag, unsigned epoch, hipStream_t s);
void launch_sync_signal(unsigned* flag, unsigned epoch, hipStream_t s);   // would unet_forward(defer_join = true) on hn_step's single lane use the device flags?
int side_stream_for(hn_ctx* ctx, int slot, const hipStream_t* refs, int nrefs, bool may_sync, hipStream_t* out);   // a stream that overlaps with every stream in refs

// standalone sub-modules (hn_double_conv / hn_conv8x8 / hn_out_conv): fp32 vector kernels of hn_unet.hip on packed device weights
int module_double_conv(hn_ctx* ctx, const float* x, int cin, int cout, const DcW& w, float* out, int batch, int H, int W, hipStream_t s);
int module_conv8x8(hn_ctx* ctx, const float* x, const K8W& w, bool transposed, float* out, int batch, int H, int W, hipStream_t s);
int module_out_conv(hn_ctx* ctx, const float* x, const float* w_io, const float* b, float* out, int batch, int H, int W, hipStream_t s);

}  // namespace hn
