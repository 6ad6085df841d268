// Training step of the helmnet IterativeSolver for gfx950: loss and weight gradients of n unrolled solver iterations
// (back-propagation through time) and the Adam update.
//
// Reference: helmnet/hybridnet.py:385-413 (training_step: replay-buffer sample -> set_states -> n_steps(unrolling_steps = 10)
// -> loss = 1e4 * mean(cat(residuals)^2)), :558-584 (single_step), :172-176 (on_after_backward: clip_grad_value_), :250-258
// (Adam, betas (0.9, 0.95), weight decay); the network is helmnet/architectures.py:63-84 (DoubleConv), :240-252
// (EncoderBlock.forward), :439-465 (HybridNet.forward).  PyTorch's autograd does the backward pass there; here it is written
// out, layer by layer, as the vector-Jacobian products of those same functions.
//
// Design
//   * Forward with a tape: every convolution runs as its own launch and stores its PRE-activation output, so a DoubleConv
//     leaves (mid z, output) behind; the activation is applied when a consumer stages its input.  The tape of all unrolled
//     iterations stays in HBM (2.8 MB per sample-iteration at 96^2, 20 MB at 256^2: 0.9 / 6.3 GB for the reference's batch of
//     32 x 10 iterations -- nothing next to 288 GB, so nothing is recomputed).
//   * Backward-data of a 3x3 convolution is the same direct kernel (k_conv3) on the forward weights transposed and flipped
//     (both arrangements are packed on the device once per call and read as wave-uniform scalar loads); the activation
//     derivative and the PReLU-slope gradient ride in its epilogue.  Backward-data of the 8x8 stride-2 convolution is the
//     transposed-convolution matrix-core kernel of the inference path (hn_mfma.hip) with the forward weights read as
//     [in, out, kh, kw], and vice versa; their A-operand fragments are packed on the device too.
//   * Weight gradients: a block accumulates the [cout, cin, kh, kw] (+ bias) sums of its run of tiles in registers and adds
//     them to ITS row of a [640 rows][blob] table that collects all layers and all unrolled iterations; ONE reduction kernel
//     at the end of the call sums the rows in a fixed order into the gradient blob.  A table cell belongs to one block within a
//     launch and launches are ordered by the stream, so the updates (plain read-modify-writes, or no-return atomics where they
//     are coalesced) happen in a fixed order: gradients are bit-reproducible.
//     Weight gradients are off the backward chain: the pass files them and runs the 36 of an iteration as three batched launches
//     from a job table (flush_wgrads).
//   * A small dependent launch costs ~10 us whatever it computes (tools/ubench_launch_floor.hip), and a step is one chain of them:
//     launches that are not on the chain are batched (the hidden-state DoubleConvs of all levels: k_conv3_batch), and at the
//     levels up to 32^2 a DoubleConv, or its backward-data pass, is ONE per-sample launch with the image in LDS (k_dc_small).
//   * The spectral operator is linear: its backward is the adjoint pass spec_adjoint (hn_spectral.hip).
//   * Everything is enqueued on the caller's stream (HN_OPT_TRAIN_LANES 2: the second half of the batch on an internal stream,
//     forked and joined with events); the host waits only when the workspace has to grow, or for a job table of four calls ago
//     to have left its pinned buffer.
#include <cmath>
#include <cstring>
#include <vector>

#include "hn_internal.h"

namespace hn {
namespace {

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
void select_gset(hn_ctx::TrainWs& W, int k);

// ---- raw (PyTorch-layout) blob offsets, the order of hn_load_weights -------------------------------------------------
struct RawDc { size_t w1, b1, slope, w2, b2; int cin, cm, co; };
struct RawK8 { size_t w, b; };
struct RawLayout {
    RawDc inc, sig[kMaxDepth], st[kMaxDepth], dec[kMaxDepth + 1];
    RawK8 down[kMaxDepth], up[kMaxDepth];
    size_t outc_w, outc_b, total;
};
RawLayout raw_layout(int depth) {
    RawLayout L{};
    size_t pos = 0;
    auto dc = [&](int cin, int cm, int co) {
        RawDc d{};
        d.cin = cin; d.cm = cm; d.co = co;
        d.w1 = pos; pos += (size_t)cm * cin * 9;
        d.b1 = pos; pos += cm;
        d.slope = pos; pos += 1;
        d.w2 = pos; pos += (size_t)co * cm * 9;
        d.b2 = pos; pos += co;
        return d;
    };
    auto k8 = [&]() { RawK8 k{}; k.w = pos; pos += (size_t)kFeat * kFeat * 64; k.b = pos; pos += kFeat; return k; };
    L.inc = dc(kInCh, kFeat, kFeat);
    for (int d = 0; d < depth; ++d) {
        L.sig[d] = dc(kFeat + kState, kFeat, kFeat);
        L.down[d] = k8();
        L.st[d] = dc(kFeat + kState, kState, kState);
    }
    for (int d = 0; d <= depth; ++d) L.dec[d] = dc(d < depth ? 2 * kFeat : kFeat, kFeat, kFeat);
    for (int d = 0; d < depth; ++d) L.up[d] = k8();
    L.outc_w = pos; pos += 2 * kFeat;
    L.outc_b = pos; pos += 2;
    L.total = pos;
    return L;
}

// ---- activations and their derivatives (architectures.py:5-44) ---------------------------------------------------------
// GEN = false: the piecewise-linear activations only (prelu / relu / leakyrelu -- the shipped network).  The kernels are instantiated
// for both: the smooth activations' code (expm1f, tanhf, erff, log1pf inlined at every staged element) made a 3x3 kernel 35 KB of
// instructions, which every launch of the latency-bound training step fetched cold.
template <bool GEN>
__device__ __forceinline__ float act_fwd(float x, int kind, float slope) {
    if (!GEN || kind <= HN_ACT_LEAKYRELU) return x > 0.f ? x : slope * x;
    return act_general(x, kind);
}

// A channel group of an implicit concatenation: element (b, c, y, x) at p[b*sb + c*sc + y*W + x]
struct TSrc { const float* p; long sb, sc; int nch; float scale; int act; };   // act: the activation is applied while staging (p holds pre-activations)
struct TDst { float* p; long sb, sc; int nch; float scale; int accum; };      // p == nullptr: the group is discarded

// Stage one [rows x cols] window (top-left corner (y0, x0) in image coordinates, zero outside the H x W image) of every channel
// of an implicit concatenation into LDS, channel c at dst + c * cstride, row pitch `pitch`.  The window positions of a thread
// are the same for every channel, so their offsets and masks are computed once; per channel the loads are issued
// unconditionally from clamped addresses (all in flight together, several channels deep) and masked when they are stored.
template <int ROWS, int COLS, int NT>
struct WindowStager {
    static constexpr int NE = (ROWS * COLS + NT - 1) / NT;
    int goff[NE], lidx[NE];
    unsigned okmask = 0, inmask = 0;
    __device__ __forceinline__ void setup(int tid, int y0, int x0, int H, int W, int pitch) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * NT, ir = e / COLS, ic = e - ir * COLS;
            const int y = y0 + ir, x = x0 + ic;
            const bool in = e < ROWS * COLS, ok = in && y >= 0 && y < H && x >= 0 && x < W;
            goff[i] = ok ? y * W + x : 0;
            lidx[i] = ir * pitch + ic;
            okmask |= (ok ? 1u : 0u) << i;
            inmask |= (in ? 1u : 0u) << i;
        }
    }
    // channels [0, nch) of the concatenation src[0..2]; `slope` / `act_kind` for groups staged through the activation
    template <bool GEN, int G = 8>   // G: channels whose loads are in flight together (one global-memory round trip per group)
    __device__ __forceinline__ void stage(const TSrc (&src)[3], int nch, int b, float* dst, int cstride, int act_kind, float slope) const {
        // the three groups' fields as plain values (statically indexed reads, once): a job-table copy of `src` then lives in registers --
        // selecting among src[k].field inside the loop is turned back into an indexed access of the struct in scratch memory
        const float *p0 = src[0].p, *p1 = src[1].p, *p2 = src[2].p;
        const long sb0 = src[0].sb, sb1 = src[1].sb, sb2 = src[2].sb, sc0 = src[0].sc, sc1 = src[1].sc, sc2 = src[2].sc;
        const float f0 = src[0].scale, f1 = src[1].scale, f2 = src[2].scale;
        const int a0 = src[0].act, a1 = src[1].act, a2 = src[2].act;
        const int n0 = src[0].nch, n01 = n0 + src[1].nch;
        // Two phases per group of 8 channels, written out so that the compiler cannot interleave them: first EVERY load of the group
        // (clamped channel index: no branch between the loads), then the activation / masking / LDS stores.  With the per-channel
        // branches of the activation between one channel's loads and the next channel's, the loads were issued one channel at a time:
        // a launch paid `nch` dependent global-memory round trips before its barrier (r3: 10-14 us per small launch).
        for (int c0 = 0; c0 < nch; c0 += G) {
            float v[G][NE];
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int c = c0 + k < nch ? c0 + k : nch - 1;
                const bool g1 = c >= n0, g2 = c >= n01;
                const int cs = g2 ? c - n01 : g1 ? c - n0 : c;
                const float* sp = g2 ? p2 : g1 ? p1 : p0;
                const long ssb = g2 ? sb2 : g1 ? sb1 : sb0, ssc = g2 ? sc2 : g1 ? sc1 : sc0;
                const float* p = sp + (long)b * ssb + (long)cs * ssc;
#pragma unroll
                for (int i = 0; i < NE; ++i) v[k][i] = p[goff[i]];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int c = c0 + k;
                if (c < nch) {
                    const bool g1 = c >= n0, g2 = c >= n01;
                    const float sscale = g2 ? f2 : g1 ? f1 : f0;
                    const int sact = g2 ? a2 : g1 ? a1 : a0;
#pragma unroll
                    for (int i = 0; i < NE; ++i)
                        if (inmask >> i & 1u) {
                            float x = v[k][i];
                            if (sact) x = act_fwd<GEN>(x, act_kind, slope);
                            dst[c * cstride + lidx[i]] = (okmask >> i & 1u) ? x * sscale : 0.f;
                        }
                }
            }
        }
    }
};

// Wave-uniform reads through the constant address space become scalar loads (SGPR operands of the FMAs).
typedef const float __attribute__((address_space(4))) * CfPtr;
__device__ __forceinline__ CfPtr cf(const float* p) { return (CfPtr)(uintptr_t)p; }
// ... and channel PAIRS of the packed 3x3 weights as SGPR pairs of v_pk_fma_f32 (the packed arrays start at even offsets: pk_off)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const f32x2 __attribute__((address_space(4))) * Cf2Ptr;
__device__ __forceinline__ Cf2Ptr cf2(const float* p) { return (Cf2Ptr)(uintptr_t)p; }
__host__ __device__ constexpr long pk_off(long raw_off) { return (raw_off + 1) & ~1L; }   // the float after a 3x3 weight tensor is a bias: never packed

// The 9-tap x CO-channel FMA block of the direct 3x3 kernels for one input channel.  HN_TRAIN_PK 1: packed over channel pairs (v_pk_fma_f32 with an
// SGPR-pair weight operand, half the issue slots); 0: scalar v_fma_f32.  [measured, r4, same box] packed is SLOWER in these latency-bound kernels
// (step 9.82 vs 9.60 ms with fused forward, 11.51 vs 11.15 without: the pair operands cost SGPR alignment / spills): scalar is the default.
#ifndef HN_TRAIN_PK
#define HN_TRAIN_PK 0
#endif
template <int CO>
__device__ __forceinline__ void fma_taps(float (&acc)[CO], const float* wpk_ci, const float (&v)[9]) {
#if HN_TRAIN_PK
    const Cf2Ptr wq = cf2(wpk_ci);
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int c = 0; c < CO / 2; ++c) {
            const f32x2 r = __builtin_elementwise_fma(wq[k * (CO / 2) + c], (f32x2){v[k], v[k]}, (f32x2){acc[2 * c], acc[2 * c + 1]});
            acc[2 * c] = r[0]; acc[2 * c + 1] = r[1];
        }
#else
    const CfPtr wq = cf(wpk_ci);
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] = fmaf(wq[k * CO + c], v[k], acc[c]);
#endif
}

// Entry `j` of a job table, read word by word through the constant address space: the index is the same for the whole block, so
// the struct arrives in SGPRs and everything derived from it (loop bounds, base pointers) stays wave-uniform.
typedef const unsigned __attribute__((address_space(4))) * CuPtr;
template <class T>
__device__ __forceinline__ T load_job(const T* table, int j) {
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    const CuPtr q = (CuPtr)(uintptr_t)(table + j);
    T t;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) {
        const unsigned w = q[i];
        __builtin_memcpy(reinterpret_cast<char*>(&t) + 4 * i, &w, 4);
    }
    return t;
}
// Batched launches: the jobs of one launch own consecutive runs of blocks [blk0, blk0 + nblk); `first` = &table[0].blk0.
template <class T>
__device__ __forceinline__ int find_job(const T* table, int njobs, int block) {
    const CuPtr q = (CuPtr)(uintptr_t)&table[0].blk0;
    constexpr int stride = sizeof(T) / 4;
    int j = 0;
    while (j + 1 < njobs && block >= (int)q[(j + 1) * stride]) ++j;
    return j;
}

// ------------------------------------------------------------------------------------------------------------------
// 3x3 convolution, padding 1, any (cin <= 16) -> CO channels, direct fp32 on the vector ALU.
//   out[co] = bias[co] + sum_ci sum_k wpk[ci][k][co] * in[ci](. + k - 1)
// wpk is the arrangement k_pack3 wrote: the forward weights as they are (forward pass) or transposed and flipped
// (backward-data: out = gradient of the input).
//   EPI_ACT: out *= act'(z) (z: the pre-activation tensor the gradient flows back into) and, for PReLU, this block's partial sum
//            of out_before * min(z, 0) (d loss / d slope) is added to slope_part[block].
// Tile 16 x 32, 512 threads, thread = ONE pixel x all CO channels; ALL input channels staged at once (one load phase, one barrier).
// At the reference's training size (96^2 x 32) a layer is 295 k pixels: one pixel per thread makes it 4.5 wavefronts per SIMD, and
// what a launch costs is the serial load -> barrier -> compute chain of one wavefront -- with 1 x 4 strips per thread (1.1 wavefronts
// per SIMD, four times the chain) the same kernels measured 22-31 us.
// ------------------------------------------------------------------------------------------------------------------
struct Conv3Args {
    TSrc src[3];
    TDst dst[3];
    const float* wpk;
    const float* bias;
    int H, W;
    int act_kind;
    const float* slope;
    const float* z;
    long z_sb, z_sc;
    double* slope_part;
};
constexpr int kC3TH = 16, kC3TW = 32, kC3PI = 36;

// one 16 x 32 tile of sample b; `a` lives in the kernel-argument segment (scalar loads, also where it is indexed)
template <int CO, bool EPI_ACT, bool GEN>
__device__ __forceinline__ void conv3_tile(const Conv3Args& a, int x0, int y0, int b, int block_in_layer) {
    constexpr int TH = kC3TH, TW = kC3TW, PI = kC3PI, IR = TH + 2, IC = TW + 2;
    extern __shared__ __attribute__((aligned(16))) float s_in[];   // [CI][IR][PI] + 8
    __shared__ double s_red[8];
    const int tid = threadIdx.x;
    const int CI = a.src[0].nch + a.src[1].nch + a.src[2].nch;
    const float slope = a.slope != nullptr ? a.slope[0] : 0.f;
    {
        WindowStager<IR, IC, 512> st;
        st.setup(tid, y0 - 1, x0 - 1, a.H, a.W, PI);
        st.template stage<GEN>(a.src, CI, b, s_in, IR * PI, a.act_kind, slope);
    }
    const int ry = tid >> 5, cx = tid & 31;   // one output pixel per thread: (y0 + ry, x0 + cx), all CO channels
    const int y = y0 + ry, x = x0 + cx;
    const bool live = y < a.H && x < a.W;
    const long pix = live ? (long)y * a.W + x : 0;
    // Everything the epilogue needs from memory is requested HERE, behind the staging loads and in front of the barrier, so that it
    // arrives while the FMA loop runs: the biases (scalar loads), z (EPI_ACT) and the old values of accumulated destinations.  (As
    // part of the epilogue they were CO dependent round trips -- a vector load of bias[c], scalar loads of dst[.], a
    // read-modify-write -- one after the other: most of what a small launch cost, r3.)
    const float *dp0 = a.dst[0].p, *dp1 = a.dst[1].p, *dp2 = a.dst[2].p;   // the groups' fields as values: no indexed access of `a`
    const long dsb0 = a.dst[0].sb, dsb1 = a.dst[1].sb, dsb2 = a.dst[2].sb, dsc0 = a.dst[0].sc, dsc1 = a.dst[1].sc, dsc2 = a.dst[2].sc;
    const float df0 = a.dst[0].scale, df1 = a.dst[1].scale, df2 = a.dst[2].scale;
    const int da0 = a.dst[0].accum, da1 = a.dst[1].accum, da2 = a.dst[2].accum;
    const int dn0 = a.dst[0].nch, dn01 = dn0 + a.dst[1].nch;
    float bias[CO], zz[CO], old[CO];
    {
        const CfPtr bp = cf(a.bias);
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            bias[c] = a.bias != nullptr ? bp[c] : 0.f;
            zz[c] = 0.f;
            if (EPI_ACT) zz[c] = a.z[(long)b * a.z_sb + (long)c * a.z_sc + pix];
            const bool g1 = c >= dn0, g2 = c >= dn01;
            const float* dp = g2 ? dp2 : g1 ? dp1 : dp0;
            const int acc_on = g2 ? da2 : g1 ? da1 : da0;
            old[c] = 0.f;
            if (dp != nullptr && acc_on) {
                const int cd = g2 ? c - dn01 : g1 ? c - dn0 : c;
                old[c] = dp[(long)b * (g2 ? dsb2 : g1 ? dsb1 : dsb0) + (long)cd * (g2 ? dsc2 : g1 ? dsc1 : dsc0) + pix];
            }
        }
    }
    __syncthreads();
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
#pragma unroll CO <= 8 ? 2 : 1
    for (int ci = 0; ci < CI; ++ci) {
        const float* t = &s_in[(ci * IR + ry) * PI + cx];
        float v[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = t[dy * PI + dx];
        fma_taps<CO>(acc, a.wpk + ci * 9 * CO, v);
    }
    double sp = 0.0;   // the slope gradient is ONE number summed over every pixel of the layer with both signs: kept in float64
    if (live) {
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            float v = acc[c] + bias[c];
            if (EPI_ACT) {
                if (zz[c] <= 0.f) sp += (double)v * (double)zz[c];
                v *= act_grad<GEN>(zz[c], a.act_kind, slope);
            }
            const bool g1 = c >= dn0, g2 = c >= dn01;
            float* dp = const_cast<float*>(g2 ? dp2 : g1 ? dp1 : dp0);
            if (dp != nullptr) {
                const int cd = g2 ? c - dn01 : g1 ? c - dn0 : c;
                float* q = dp + (long)b * (g2 ? dsb2 : g1 ? dsb1 : dsb0) + (long)cd * (g2 ? dsc2 : g1 ? dsc1 : dsc0) + pix;
                *q = fmaf(v, g2 ? df2 : g1 ? df1 : df0, 0.f) + old[c];
            }
        }
    }
    if (EPI_ACT && a.slope_part != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sp += __shfl_down(sp, o, 64);
        if ((tid & 63) == 0) s_red[tid >> 6] = sp;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) tot += s_red[wv];
            a.slope_part[block_in_layer] += tot;
        }
    }
}

template <int CO, bool EPI_ACT, bool GEN>
__global__ __launch_bounds__(512) void k_conv3(Conv3Args a) {
    conv3_tile<CO, EPI_ACT, GEN>(a, blockIdx.x * kC3TW, blockIdx.y * kC3TH, blockIdx.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

// The same layer type at several levels of the UNet in ONE launch (the hidden-state DoubleConvs: off the chain within an
// iteration, so the levels' instances are independent): job j owns blocks [blk0[j], blk0[j + 1]), tiles in (x, y, sample) order.
struct Conv3Batch {
    Conv3Args job[kMaxDepth];
    int blk0[kMaxDepth + 1];
    int tiles_x[kMaxDepth], tiles_y[kMaxDepth];
    int njobs;
};
template <int CO, bool EPI_ACT, bool GEN>
__global__ __launch_bounds__(512) void k_conv3_batch(Conv3Batch q) {
    int j = 0;
    while (j + 1 < q.njobs && (int)blockIdx.x >= q.blk0[j + 1]) ++j;
    const int bid = (int)blockIdx.x - q.blk0[j];
    const int tx = bid % q.tiles_x[j], r = bid / q.tiles_x[j], ty = r % q.tiles_y[j], b = r / q.tiles_y[j];
    conv3_tile<CO, EPI_ACT, GEN>(q.job[j], tx * kC3TW, ty * kC3TH, b, bid);
}

// ------------------------------------------------------------------------------------------------------------------
// A whole DoubleConv (or its backward-data pass) of a SMALL level as one launch: one workgroup per sample, the S x S image
// (S <= 32) entirely in LDS.  At the training size the three deepest levels are 24^2, 12^2 and 6^2: a 3x3 launch there is 32
// workgroups and costs its 10 us floor whatever it computes (tools/ubench_launch_floor.hip), so two convolutions in one launch
// are one floor instead of two.
//   first : C1 channels = conv(a1.src) (+ bias, EPI: * act'(z), slope sum)  -> a1.dst[0] (global: the tape's z, or g_z) and the
//           LDS mid planes (through the activation when a2.src[0].act is set: the forward pass)
//   second: C2 channels = conv(mid) (+ bias) -> a2.dst[0..2] (accumulating where asked)
// ------------------------------------------------------------------------------------------------------------------
struct DcSmallArgs { Conv3Args a1, a2; };
constexpr int kSmallS = 32, kSmallR = kSmallS + 2, kSmallP = kSmallS + 3;   // planes of 34 rows x 35 floats: image + zero border
constexpr int kSmallNT = 1024;   // one thread per pixel of the largest image: a convolution is ONE pass (with 512 threads 24^2 took two)

template <int CO, bool EPI_ACT, bool GEN>
__device__ __forceinline__ void small_conv(const Conv3Args& a, int CI, const float* s_in, float* s_mid, bool mid_act, int b, int S, float slope,
                                           double& sp) {
    constexpr int R = kSmallR, P = kSmallP;
    const int tid = threadIdx.x;
    const float *dp0 = a.dst[0].p, *dp1 = a.dst[1].p, *dp2 = a.dst[2].p;
    const long dsb0 = a.dst[0].sb, dsb1 = a.dst[1].sb, dsb2 = a.dst[2].sb, dsc0 = a.dst[0].sc, dsc1 = a.dst[1].sc, dsc2 = a.dst[2].sc;
    const float df0 = a.dst[0].scale, df1 = a.dst[1].scale, df2 = a.dst[2].scale;
    const int da0 = a.dst[0].accum, da1 = a.dst[1].accum, da2 = a.dst[2].accum;
    const int dn0 = a.dst[0].nch, dn01 = dn0 + a.dst[1].nch;
    const CfPtr bp = cf(a.bias);
    for (int p = tid; p < S * S; p += kSmallNT) {
        const int y = p / S, x = p - y * S;
        const long pix = (long)y * S + x;
        float bias[CO], zz[CO], old[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) {   // requested ahead of the FMA loop (see conv3_tile)
            bias[c] = a.bias != nullptr ? bp[c] : 0.f;
            zz[c] = 0.f;
            if (EPI_ACT) zz[c] = a.z[(long)b * a.z_sb + (long)c * a.z_sc + pix];
            const bool g1 = c >= dn0, g2 = c >= dn01;
            const float* dp = g2 ? dp2 : g1 ? dp1 : dp0;
            old[c] = 0.f;
            if (dp != nullptr && (g2 ? da2 : g1 ? da1 : da0)) {
                const int cd = g2 ? c - dn01 : g1 ? c - dn0 : c;
                old[c] = dp[(long)b * (g2 ? dsb2 : g1 ? dsb1 : dsb0) + (long)cd * (g2 ? dsc2 : g1 ? dsc1 : dsc0) + pix];
            }
        }
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] = 0.f;
#pragma unroll CO <= 8 ? 2 : 1
        for (int ci = 0; ci < CI; ++ci) {
            const float* t = &s_in[(ci * R + y) * P + x];
            float v[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = t[dy * P + dx];
            fma_taps<CO>(acc, a.wpk + ci * 9 * CO, v);
        }
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            float v = acc[c] + bias[c];
            if (EPI_ACT) {
                if (zz[c] <= 0.f) sp += (double)v * (double)zz[c];
                v *= act_grad<GEN>(zz[c], a.act_kind, slope);
            }
            if (s_mid != nullptr) s_mid[(c * R + y + 1) * P + x + 1] = mid_act ? act_fwd<GEN>(v, a.act_kind, slope) : v;
            const bool g1 = c >= dn0, g2 = c >= dn01;
            float* dp = const_cast<float*>(g2 ? dp2 : g1 ? dp1 : dp0);
            if (dp != nullptr) {
                const int cd = g2 ? c - dn01 : g1 ? c - dn0 : c;
                float* q = dp + (long)b * (g2 ? dsb2 : g1 ? dsb1 : dsb0) + (long)cd * (g2 ? dsc2 : g1 ? dsc1 : dsc0) + pix;
                *q = fmaf(v, g2 ? df2 : g1 ? df1 : df0, 0.f) + old[c];
            }
        }
    }
}

template <int C1, int C2, bool EPI_ACT, bool GEN>
__global__ __launch_bounds__(kSmallNT) void k_dc_small(DcSmallArgs q) {
    constexpr int R = kSmallR, P = kSmallP;
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];   // input planes [CI][34][35], then the mid planes [C1][34][35]
    __shared__ double s_red[kSmallNT / 64];
    const int tid = threadIdx.x, b = blockIdx.x;
    const Conv3Args& a1 = q.a1;
    const int S = a1.H;
    const int CI = a1.src[0].nch + a1.src[1].nch + a1.src[2].nch;
    const float slope = a1.slope != nullptr ? a1.slope[0] : 0.f;
    float* const s_in = s_dyn;
    float* const s_mid = s_dyn + CI * R * P;
    {
        WindowStager<R, R, kSmallNT> st;   // the 34 x 34 window around the image: zero outside it
        st.setup(tid, -1, -1, S, S, P);
        st.template stage<GEN>(a1.src, CI, b, s_in, R * P, a1.act_kind, slope);
        for (int i = tid; i < C1 * R * P; i += kSmallNT) s_mid[i] = 0.f;   // zero border of the mid planes (the interior is overwritten)
    }
    __syncthreads();
    double sp = 0.0;
    small_conv<C1, EPI_ACT, GEN>(a1, CI, s_in, s_mid, q.a2.src[0].act != 0, b, S, slope, sp);
    if (EPI_ACT && a1.slope_part != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sp += __shfl_down(sp, o, 64);
        if ((tid & 63) == 0) s_red[tid >> 6] = sp;
    }
    __syncthreads();
    if (EPI_ACT && a1.slope_part != nullptr && tid == 0) {
        double tot = 0.0;
#pragma unroll
        for (int wv = 0; wv < kSmallNT / 64; ++wv) tot += s_red[wv];
        a1.slope_part[b] += tot;
    }
    double unused = 0.0;
    small_conv<C2, false, GEN>(q.a2, C1, s_mid, nullptr, false, b, S, slope, unused);
}

// ------------------------------------------------------------------------------------------------------------------
// Backward-data of a DoubleConv at a BIG level as ONE launch per 16 x 32 tile (r4; at the small levels k_dc_small does the same per sample):
//   g_z  = conv(g_out; W2^T) * act'(z)   on the tile + a halo of 1 (recomputed by the neighbouring tiles: 612 instead of 512 positions),
//          its own 16 x 32 positions stored (the weight gradient of conv1 reads g_z), the PReLU-slope sum taken over those only;
//   g_in = conv(g_z; W1^T)                from the LDS copy -> the channel groups of the DoubleConv's input concatenation.
// Two launches (k_conv3<8, EPI> + k_conv3<C2>) with a g_z round trip through HBM become one; same arithmetic per output, same
// per-block slope partial sums (the tiling is k_conv3's), so the gradients do not change by a bit.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kDtPI = kC3TW + 5, kDtPM = kC3TW + 3;   // row pitches of the staged g_out window (20 x 36) and of the g_z planes (18 x 34)

// One 16 x 32 tile of a two-convolution chain.  BWD: the backward-data pass described above (first convolution's output * act'(z), no biases).
// !BWD (the hidden-state DoubleConvs' forward pass): mid = conv(in) + b1 stored pre-activation (the tape's z), act(mid) in LDS, out = conv(act(mid)) + b2.
// CIMAX = most input channels staged (8: g_out; 10: cat[out_d, state_d]); C1 = channels of the tensor between the two convolutions, C2 = of the result.
template <int CIMAX, int C1, int C2, bool BWD, bool GEN>
__device__ __forceinline__ void dc_tile(const DcSmallArgs& q, int x0, int y0, int b, int block_in_layer) {
    constexpr int TH = kC3TH, TW = kC3TW, IR = TH + 4, IC = TW + 4, MR = TH + 2, MC = TW + 2;
    __shared__ __attribute__((aligned(16))) float s_in[CIMAX * IR * kDtPI];
    __shared__ __attribute__((aligned(16))) float s_mid[C1 * MR * kDtPM];
    __shared__ double s_red[8];
    const Conv3Args& a1 = q.a1;
    const Conv3Args& a2 = q.a2;
    const int tid = threadIdx.x;
    const int H = a1.H, W = a1.W;
    const int CI = a1.src[0].nch + a1.src[1].nch + a1.src[2].nch;
    const float slope = a1.slope != nullptr ? a1.slope[0] : 0.f;
    {
        WindowStager<IR, IC, 512> st;
        st.setup(tid, y0 - 2, x0 - 2, H, W, kDtPI);
        st.template stage<GEN>(a1.src, CI, b, s_in, IR * kDtPI, a1.act_kind, slope);
    }
    __syncthreads();
    double sp = 0.0;
    {
        float* const gz = a1.dst[0].p;
        const long gz_sb = a1.dst[0].sb, gz_sc = a1.dst[0].sc;
        const CfPtr bp = cf(a1.bias);
        for (int p = tid; p < MR * MC; p += 512) {
            const int my = p / MC, mx = p - my * MC;
            const int y = y0 - 1 + my, x = x0 - 1 + mx;
            const bool in = y >= 0 && y < H && x >= 0 && x < W;
            const bool own = in && my >= 1 && my <= TH && mx >= 1 && mx <= TW;
            const long pix = in ? (long)y * W + x : 0;
            float zz[C1];
#pragma unroll
            for (int c = 0; c < C1; ++c) zz[c] = BWD ? a1.z[(long)b * a1.z_sb + (long)c * a1.z_sc + pix] : (a1.bias != nullptr ? bp[c] : 0.f);   // (!BWD: the bias)
            float acc[C1];
#pragma unroll
            for (int c = 0; c < C1; ++c) acc[c] = 0.f;
#pragma unroll 2
            for (int ci = 0; ci < CI; ++ci) {
                const float* t = &s_in[(ci * IR + my) * kDtPI + mx];
                float v[9];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = t[dy * kDtPI + dx];
                fma_taps<C1>(acc, a1.wpk + ci * 9 * C1, v);
            }
#pragma unroll
            for (int c = 0; c < C1; ++c) {
                float v = acc[c];
                if (BWD) {
                    if (own && zz[c] <= 0.f) sp += (double)v * (double)zz[c];
                    v *= act_grad<GEN>(zz[c], a1.act_kind, slope);
                } else v += zz[c];
                if (own && gz != nullptr) gz[(long)b * gz_sb + (long)c * gz_sc + pix] = v;
                s_mid[(c * MR + my) * kDtPM + mx] = in ? (BWD ? v : act_fwd<GEN>(v, a1.act_kind, slope)) : 0.f;
            }
        }
    }
    if (BWD && a1.slope_part != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sp += __shfl_down(sp, o, 64);
        if ((tid & 63) == 0) s_red[tid >> 6] = sp;
    }
    __syncthreads();
    if (BWD && a1.slope_part != nullptr && tid == 0) {
        double tot = 0.0;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) tot += s_red[wv];
        a1.slope_part[block_in_layer] += tot;
    }
    // ---- second convolution from the LDS planes: one pixel per thread, C2 channels, to the destination's channel groups ----
    const int ry = tid >> 5, cx = tid & 31;
    const int y = y0 + ry, x = x0 + cx;
    const bool live = y < H && x < W;
    const long pix = live ? (long)y * W + x : 0;
    const float *dp0 = a2.dst[0].p, *dp1 = a2.dst[1].p, *dp2 = a2.dst[2].p;
    const long dsb0 = a2.dst[0].sb, dsb1 = a2.dst[1].sb, dsb2 = a2.dst[2].sb, dsc0 = a2.dst[0].sc, dsc1 = a2.dst[1].sc, dsc2 = a2.dst[2].sc;
    const float df0 = a2.dst[0].scale, df1 = a2.dst[1].scale, df2 = a2.dst[2].scale;
    const int da0 = a2.dst[0].accum, da1 = a2.dst[1].accum, da2 = a2.dst[2].accum;
    const int dn0 = a2.dst[0].nch, dn01 = dn0 + a2.dst[1].nch;
    float old[C2], o[C2], bias2[C2];
    {
        const CfPtr bp2 = cf(a2.bias);
#pragma unroll
        for (int c = 0; c < C2; ++c) {   // the old values of accumulated destinations and the biases, requested ahead of the FMA loop
            const bool g1 = c >= dn0, g2 = c >= dn01;
            const float* dp = g2 ? dp2 : g1 ? dp1 : dp0;
            old[c] = 0.f;
            if (dp != nullptr && (g2 ? da2 : g1 ? da1 : da0)) {
                const int cd = g2 ? c - dn01 : g1 ? c - dn0 : c;
                old[c] = dp[(long)b * (g2 ? dsb2 : g1 ? dsb1 : dsb0) + (long)cd * (g2 ? dsc2 : g1 ? dsc1 : dsc0) + pix];
            }
            o[c] = 0.f;
            bias2[c] = (!BWD && a2.bias != nullptr) ? bp2[c] : 0.f;   // added last, as k_conv3 does: the fused path is bit-identical to the two launches
        }
    }
#pragma unroll C2 <= 8 ? 2 : 1
    for (int ci = 0; ci < C1; ++ci) {
        const float* t = &s_mid[(ci * MR + ry) * kDtPM + cx];
        float v[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = t[dy * kDtPM + dx];
        fma_taps<C2>(o, a2.wpk + ci * 9 * C2, v);
    }
    if (live) {
#pragma unroll
        for (int c = 0; c < C2; ++c) {
            const float v = o[c] + bias2[c];
            const bool g1 = c >= dn0, g2 = c >= dn01;
            float* dp = const_cast<float*>(g2 ? dp2 : g1 ? dp1 : dp0);
            if (dp != nullptr) {
                const int cd = g2 ? c - dn01 : g1 ? c - dn0 : c;
                dp[(long)b * (g2 ? dsb2 : g1 ? dsb1 : dsb0) + (long)cd * (g2 ? dsc2 : g1 ? dsc1 : dsc0) + pix] = fmaf(v, g2 ? df2 : g1 ? df1 : df0, 0.f) + old[c];
            }
        }
    }
}

template <int C2, bool GEN>
__global__ __launch_bounds__(512) void k_dc_bwd_tile(DcSmallArgs q) {
    dc_tile<kFeat, kFeat, C2, true, GEN>(q, blockIdx.x * kC3TW, blockIdx.y * kC3TH, blockIdx.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

// The hidden-state DoubleConvs (10 -> 2 -> 2) of EVERY level in one launch per direction (they are off the chain within an iteration, so the levels'
// instances are independent): forward z = conv(cat[out_d, state_d]) + b (tape), new state = conv(act(z)) + b; backward g_z = conv(g_new; W2^T) act'(z),
// (g_out_d, g_state_d) = conv(g_z; W1^T).  r3 ran each direction as two batched launches with the 2-channel tensor going through HBM in between.
struct DcBatch {
    DcSmallArgs job[kMaxDepth];
    int blk0[kMaxDepth + 1];
    int tiles_x[kMaxDepth], tiles_y[kMaxDepth];
    int njobs;
};
static_assert(sizeof(DcBatch) <= 4096, "kernel-argument block");
template <bool BWD, bool GEN>
__global__ __launch_bounds__(512) void k_dc_state_batch(DcBatch q) {
    int j = 0;
    while (j + 1 < q.njobs && (int)blockIdx.x >= q.blk0[j + 1]) ++j;
    const int bid = (int)blockIdx.x - q.blk0[j];
    const int tx = bid % q.tiles_x[j], r = bid / q.tiles_x[j], ty = r % q.tiles_y[j], b = r / q.tiles_y[j];
    if (BWD) dc_tile<kState, kState, kFeat + kState, true, GEN>(q.job[j], tx * kC3TW, ty * kC3TH, b, bid);
    else dc_tile<kFeat + kState, kState, kState, false, GEN>(q.job[j], tx * kC3TW, ty * kC3TH, b, bid);
}

// ------------------------------------------------------------------------------------------------------------------
// Weight + bias gradient of a 3x3 convolution:  dW[co][ci][ky][kx] = sum_{b,y,x} g[co](y, x) * in[ci](y + ky - 1, x + kx - 1),
// db[co] = sum g[co].  Thread = one (ci, ky) row of taps (the bias is one more "row" with in == 1) x 3 kx x all CO channels x a
// subset of the tile rows: walking along x it keeps a 3-wide sliding window of the input, so a pixel costs one LDS read of
// the input and one broadcast read of the CO gradients for 3 * CO FMAs.  A block walks a strided run of kWgTH x 32 tiles and ADDS
// its sums to its row of the partials table (columns laid out like the blob: weight [CO][CI][3][3], then bias [CO]).
// ------------------------------------------------------------------------------------------------------------------
struct Wg3Args {
    TSrc src[3];
    const float* g;
    long g_sb, g_sc;
    int H, W, tiles_x, tiles_y, batch;
    int act_kind;
    const float* slope;
    float* part;        // &table[0][column of this layer's weight]
    long row_stride;    // floats between the rows of the table
    int blk0, nblk;     // this job's run of blocks within the batched launch
};

constexpr int kWgTH = 8;   // tile rows of the weight-gradient kernel: 32 KB of LDS per block, 4 blocks per CU (16-row tiles: 56 KB, 2 per CU; in the
                           // batched launch 133 -> 118 us.  As one launch per layer the two measured the same: the launch floor hid it)
template <int CO, bool GEN>
__global__ __launch_bounds__(256) void k_conv3_wgrad(const Wg3Args* __restrict__ jobs, int njobs) {
    const Wg3Args a = load_job(jobs, find_job(jobs, njobs, (int)blockIdx.x));
    const int bid = (int)blockIdx.x - a.blk0;
    constexpr int TH = kWgTH, TW = 32, IR = TH + 2, IC = TW + 2, PI = 35;
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];   // gradient tile [TH][TW][CO], then the input window [CI][IR][PI]; at the end the reduction scratch
    float* const s_g = s_dyn;
    float* const s_x = s_dyn + TH * TW * CO;
    const int tid = threadIdx.x;
    const int CI = a.src[0].nch + a.src[1].nch + a.src[2].nch;
    const int P = CI * 3 + 1, S = 256 / P;
    const int pair = tid % P, split = tid / P;
    const bool active = split < S, isb = pair == P - 1;
    const int ci = pair / 3, ky = pair - ci * 3;
    const float slope = a.slope != nullptr ? a.slope[0] : 0.f;
    // accumulators as channel PAIRS (HN_WG_PK, default): the gradient values of a pixel are adjacent in LDS, so a pair is one 8-byte read and a
    // v_pk_fma_f32 with two VGPR operands does two of the 3 x CO multiply-adds of a pixel (no SGPR-pair operand as in the 3x3 data kernels,
    // where the packed form measured slower)
#ifndef HN_WG_PK
#define HN_WG_PK 1
#endif
    f32x2 acc2[3][CO / 2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < CO / 2; ++c) acc2[k][c] = (f32x2){0.f, 0.f};
    const int ntiles = a.tiles_x * a.tiles_y * a.batch;
    for (int tile = bid; tile < ntiles; tile += a.nblk) {
        const int tx = tile % a.tiles_x, r0 = tile / a.tiles_x, ty = r0 % a.tiles_y, b = r0 / a.tiles_y;
        const int x0 = tx * TW, y0 = ty * TH;
        __syncthreads();
        {   // gradient tile (channel-interleaved in LDS: s_g[(y * TW + x) * CO + c]): requested first, in flight behind the input window's loads
            constexpr int NG = TH * TW / 256;
            float gv[CO][NG];
#pragma unroll
            for (int c = 0; c < CO; ++c)
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    const int r = tid + i * 256, iy = r / TW, ix = r - iy * TW;
                    const int y = y0 + iy, x = x0 + ix;
                    const bool ok = y < a.H && x < a.W;
                    gv[c][i] = a.g[(long)b * a.g_sb + (long)c * a.g_sc + (ok ? (long)y * a.W + x : 0)];
                    if (!ok) gv[c][i] = 0.f;
                }
            WindowStager<IR, IC, 256> st;
            st.setup(tid, y0 - 1, x0 - 1, a.H, a.W, PI);
            st.template stage<GEN, 16>(a.src, CI, b, s_x, IR * PI, a.act_kind, slope);   // every channel in ONE round trip (with the gradient tile's loads)
            // a pixel's CO gradient values are adjacent in LDS: stored as 16- / 8-byte vectors ([measured, r4 PMC] as CO scalar stores at a
            // stride of CO dwords they were 8-way bank conflicts: 3.8 conflict cycles per LDS instruction over the whole kernel)
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                if constexpr (CO % 4 == 0) {
#pragma unroll
                    for (int c = 0; c < CO; c += 4)
                        *reinterpret_cast<float4*>(&s_g[(tid + i * 256) * CO + c]) = make_float4(gv[c][i], gv[c + 1][i], gv[c + 2][i], gv[c + 3][i]);
                } else {
#pragma unroll
                    for (int c = 0; c < CO; c += 2) *reinterpret_cast<float2*>(&s_g[(tid + i * 256) * CO + c]) = make_float2(gv[c][i], gv[c + 1][i]);
                }
            }
        }
        __syncthreads();
        if (active) {
            for (int u = split; u < 2 * TH; u += S) {   // work units: (row, half of the columns), dealt round-robin to the S row subsets
                const int r = u >> 1, c0 = (u & 1) * (TW / 2);
                const float* xr = &s_x[(ci * IR + r + ky) * PI + c0];
                const float* gr = &s_g[(r * TW + c0) * CO];
                float xa = isb ? 1.f : xr[0], xb = isb ? 0.f : xr[1];
#pragma unroll 4
                for (int c = 0; c < TW / 2; ++c) {
                    const float xc = isb ? 0.f : xr[c + 2];
#pragma unroll
                    for (int o = 0; o < CO / 2; ++o) {
                        const f32x2 gv = *reinterpret_cast<const f32x2*>(&gr[c * CO + 2 * o]);
#if HN_WG_PK
                        acc2[0][o] = __builtin_elementwise_fma(gv, (f32x2){xa, xa}, acc2[0][o]);
                        acc2[1][o] = __builtin_elementwise_fma(gv, (f32x2){xb, xb}, acc2[1][o]);
                        acc2[2][o] = __builtin_elementwise_fma(gv, (f32x2){xc, xc}, acc2[2][o]);
#else
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            acc2[0][o][h] = fmaf(gv[h], xa, acc2[0][o][h]);
                            acc2[1][o][h] = fmaf(gv[h], xb, acc2[1][o][h]);
                            acc2[2][o][h] = fmaf(gv[h], xc, acc2[2][o][h]);
                        }
#endif
                    }
                    xa = isb ? 1.f : xb;
                    xb = xc;
                }
            }
        }
    }
    // reduce over the row subsets (fixed order), then one thread per tap row adds the block's sums to its row of the table
    __syncthreads();
    float* s_red = s_dyn;   // S * P * 3 * CO <= 256 * 3 * CO floats, over the (dead) gradient tile and input window
    if (active) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int o = 0; o < CO; ++o) s_red[((split * P + pair) * 3 + k) * CO + o] = acc2[k][o >> 1][o & 1];
    }
    __syncthreads();
    // every thread sums the row subsets (fixed order) of a few cells of the block's table row and adds them to it: consecutive
    // threads own consecutive cells, so the read-modify-write is coalesced (a cell belongs to this block alone within a launch;
    // one thread per tap row doing all of it serially was a third of the kernel's fixed cost, scattered 4-byte atomics another)
    float* row = a.part + (size_t)bid * a.row_stride;
    for (int i = tid; i < CO * CI * 9 + CO; i += 256) {
        int o, p, k;
        if (i >= CO * CI * 9) { o = i - CO * CI * 9; p = P - 1; k = 0; }
        else { o = i / (CI * 9); const int r = i - o * (CI * 9), c = r / 9, t9 = r - c * 9, y3 = t9 / 3; p = c * 3 + y3; k = t9 - y3 * 3; }
        float sum = 0.f;
        for (int q = 0; q < S; ++q) sum += s_red[((q * P + p) * 3 + k) * CO + o];
        row[i] += sum;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 8x8 stride-2 (transposed) convolution, padding 3 (architectures.py:209-211, 375-382):
//     dW[a][b][ky][kx] = sum_{n,Y,X} sm[a](Y, X) * bg[b](2Y + ky - 3, 2X + kx - 3)
//   down  (Conv2d, weight [out, in, 8, 8]):           sm = grad of the output (a = out), bg = input (b = in)
//   up    (ConvTranspose2d, weight [in, out, 8, 8]):  sm = input (a = in),               bg = grad of the output (b = out)
// Thread = (b, ky, kx) x 8 values of a; block walks a run of 8 x 16 tiles of the small tensor and adds to its row of the table.
// ------------------------------------------------------------------------------------------------------------------
struct Wg8Args {
    const float* sm; long sm_sb, sm_sc; int hs, ws;
    const float* bg; long bg_sb, bg_sc;
    int tiles_x, tiles_y, batch;
    float* part;
    long row_stride;
    int bias_from_big;   // the bias gradient (per-channel sum of the OUTPUT gradient) behind the weight in the blob: 0 sum of sm (down), 1 sum of bg (up)
    int blk0, nblk;      // this job's run of blocks within the batched launch
    int pad_;
};
static_assert(sizeof(Wg8Args) % 4 == 0 && sizeof(Wg3Args) % 4 == 0, "job tables are read dword by dword");

__global__ __launch_bounds__(512) void k_conv8_wgrad(const Wg8Args* __restrict__ jobs, int njobs) {
    const Wg8Args a = load_job(jobs, find_job(jobs, njobs, (int)blockIdx.x));
    const int bid = (int)blockIdx.x - a.blk0;
    constexpr int TY = 8, TX = 16, BR = 2 * TY + 6, BC = 2 * TX + 6, PB = 39;
    __shared__ float s_b[kFeat * BR * PB];
    __shared__ __attribute__((aligned(16))) float s_s[TY * TX * kFeat];
    const int tid = threadIdx.x;
    const int bch = tid >> 6, k = tid & 63, ky = k >> 3, kx = k & 7;
    const int hb = 2 * a.hs, wb = 2 * a.ws;
    float acc[kFeat], bsum[kFeat];
#pragma unroll
    for (int c = 0; c < kFeat; ++c) acc[c] = bsum[c] = 0.f;
    const int ntiles = a.tiles_x * a.tiles_y * a.batch;
    for (int tile = bid; tile < ntiles; tile += a.nblk) {
        const int tx = tile % a.tiles_x, r0 = tile / a.tiles_x, ty = r0 % a.tiles_y, n = r0 / a.tiles_y;
        const int X0 = tx * TX, Y0 = ty * TY;
        __syncthreads();
        {
            const TSrc big[3] = {TSrc{a.bg, a.bg_sb, a.bg_sc, kFeat, 1.f, 0}, TSrc{nullptr, 0, 0, 0, 1.f, 0}, TSrc{nullptr, 0, 0, 0, 1.f, 0}};
            WindowStager<BR, BC, 512> st;
            st.setup(tid, 2 * Y0 - 3, 2 * X0 - 3, hb, wb, PB);
            st.template stage<false>(big, kFeat, n, s_b, BR * PB, 0, 0.f);
        }
        {   // small tensor tile, channel-interleaved: s_s[(Y * TX + X) * 8 + c]; 128 positions x 8 channels = 2 per thread
            float sv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * 512, c = e >> 7, r = e & 127, iy = r / TX, ix = r - iy * TX;
                const int y = Y0 + iy, x = X0 + ix;
                const bool ok = y < a.hs && x < a.ws;
                sv[i] = a.sm[(long)n * a.sm_sb + (long)c * a.sm_sc + (ok ? (long)y * a.ws + x : 0)];
                if (!ok) sv[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * 512;
                s_s[(e & 127) * kFeat + (e >> 7)] = sv[i];
            }
        }
        __syncthreads();
        const float* bp = &s_b[(bch * BR + ky) * PB + kx];
#pragma unroll 2
        for (int Y = 0; Y < TY; ++Y)
#pragma unroll 4
            for (int X = 0; X < TX; ++X) {
                const float xv = bp[2 * Y * PB + 2 * X];
                const float* sp = &s_s[(Y * TX + X) * kFeat];
#pragma unroll
                for (int c = 0; c < kFeat; ++c) acc[c] = fmaf(sp[c], xv, acc[c]);
            }
        // bias gradient: this thread's share of the tile's own pixels (positions outside the image were staged as zeros)
        if (a.bias_from_big) {   // the 16 x 32 big-tensor pixels of the tile: one per thread, all 8 channels
            const int py = tid >> 5, px = tid & 31;
#pragma unroll
            for (int c = 0; c < kFeat; ++c) bsum[c] += s_b[(c * BR + 3 + py) * PB + 3 + px];
        } else if (tid < TY * TX) {   // the 8 x 16 small-tensor pixels
#pragma unroll
            for (int c = 0; c < kFeat; ++c) bsum[c] += s_s[tid * kFeat + c];
        }
    }
    float* row = a.part + (size_t)bid * a.row_stride;
#pragma unroll
    for (int c = 0; c < kFeat; ++c) unsafeAtomicAdd(&row[(c * kFeat + bch) * 64 + k], acc[c]);
    // per-channel sums over the block (fixed order: wave shuffles, then the 8 wave partials), behind the 4096 weights
    __syncthreads();
    float* s_part = s_s;   // [8 waves][8 channels]
#pragma unroll
    for (int c = 0; c < kFeat; ++c) {
        float v = bsum[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((tid & 63) == 0) s_part[(tid >> 6) * kFeat + c] = v;
    }
    __syncthreads();
    if (tid < kFeat) {
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) v += s_part[wv * kFeat + tid];
        row[kFeat * kFeat * 64 + tid] += v;
    }
}

// grad[j] = sum over the rows of the table, in a fixed order (every weight-gradient kernel of the call has added to its row)
// A block owns 64 columns; its 8 wavefronts sum 8 runs of rows (coalesced 256-byte row segments), then the runs are added in a fixed order.
// (One thread per column over all 640 rows was 188 blocks for a 123 MB table: 76 us = 1.6 TB/s [measured, r4].)
__global__ __launch_bounds__(512) void k_reduce_rows(const float* __restrict__ part, int rows, int count, float* __restrict__ out, int accumulate) {
    __shared__ float s_run[8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const int per = (rows + 7) / 8, r0 = wave * per, r1 = r0 + per < rows ? r0 + per : rows;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < count) {
        int r = r0;
        for (; r + 4 <= r1; r += 4) {
            s0 += part[(size_t)r * count + j];
            s1 += part[(size_t)(r + 1) * count + j];
            s2 += part[(size_t)(r + 2) * count + j];
            s3 += part[(size_t)(r + 3) * count + j];
        }
        for (; r < r1; ++r) s0 += part[(size_t)r * count + j];
    }
    s_run[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && j < count) {
        const float v = ((s_run[0][lane] + s_run[1][lane]) + (s_run[2][lane] + s_run[3][lane])) + ((s_run[4][lane] + s_run[5][lane]) + (s_run[6][lane] + s_run[7][lane]));
        out[j] = accumulate ? out[j] + v : v;   // the second lane's table is added to the first lane's sums
    }
}
// PReLU slope gradients: block j sums the rows[j] per-block partial sums of DoubleConv j (fixed order) into grad[off[j]]
struct SlopeJobs { int rows[3 * kMaxDepth + 2]; int off[3 * kMaxDepth + 2]; };
__global__ __launch_bounds__(256) void k_reduce_slopes(const double* __restrict__ part, int stride, SlopeJobs jobs, const double* __restrict__ part_b,
                                                       int stride_b, SlopeJobs jobs_b, float* __restrict__ grad) {
    __shared__ double s_red[4];
    const int j = blockIdx.x;
    double s = 0.0;
    for (int r = threadIdx.x; r < jobs.rows[j]; r += 256) s += part[(size_t)j * stride + r];
    if (part_b != nullptr)   // second lane (kept in float64 to the end: the slope gradient is a sum with both signs)
        for (int r = threadIdx.x; r < jobs_b.rows[j]; r += 256) s += part_b[(size_t)j * stride_b + r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) grad[jobs.off[j]] = (float)((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
}

// ---- output layer: d = Conv1x1(8 -> 2)(y0); wf_next = wf + d / 1e3   (architectures.py:57-60, hybridnet.py:570) ----------
__global__ __launch_bounds__(256) void k_outc_fwd(const float* __restrict__ y0, const float* __restrict__ w, const float* __restrict__ bias,
                                                  const float* __restrict__ wf, float* __restrict__ wf_next, long plane, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / plane, p = i - b * plane;
    const float* yp = y0 + b * kFeat * plane + p;
    float d0 = bias[0], d1 = bias[1];
#pragma unroll
    for (int c = 0; c < kFeat; ++c) {
        const float v = yp[c * plane];
        d0 = fmaf(w[c], v, d0);             // raw weight [2][8]
        d1 = fmaf(w[kFeat + c], v, d1);
    }
    const long o = b * 2 * plane + p;
    wf_next[o] = d0 / 1e3f + wf[o];
    wf_next[o + plane] = d1 / 1e3f + wf[o + plane];
}
// backward: gd = g_wfnext / 1e3;  g_y0[c] = sum_o w[o][c] gd[o];  sums of dW[o][c] = sum gd[o] y0[c], db[o] = sum gd[o] added to the block's row of the table
__global__ __launch_bounds__(256) void k_outc_bwd(const float* __restrict__ g_wfn, const float* __restrict__ y0, const float* __restrict__ w,
                                                  float* __restrict__ g_y0, float* __restrict__ part, long row_stride, long plane, long total) {
    __shared__ float s_red[4][18];
    float acc[18];
#pragma unroll
    for (int q = 0; q < 18; ++q) acc[q] = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / plane, p = i - b * plane;
        const float g0 = g_wfn[b * 2 * plane + p] / 1e3f, g1 = g_wfn[b * 2 * plane + plane + p] / 1e3f;
        const float* yp = y0 + b * kFeat * plane + p;
        float* gp = g_y0 + b * kFeat * plane + p;
#pragma unroll
        for (int c = 0; c < kFeat; ++c) {
            const float v = yp[c * plane];
            gp[c * plane] = fmaf(w[c], g0, w[kFeat + c] * g1);
            acc[c] = fmaf(g0, v, acc[c]);
            acc[kFeat + c] = fmaf(g1, v, acc[kFeat + c]);
        }
        acc[16] += g0;
        acc[17] += g1;
    }
#pragma unroll
    for (int q = 0; q < 18; ++q) {
        float s = acc[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][q] = s;
    }
    __syncthreads();
    if (threadIdx.x < 18) unsafeAtomicAdd(&part[(size_t)blockIdx.x * row_stride + threadIdx.x], (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]));
}

// g = [g_in +] c * res   (the loss term of one unrolled iteration: d/d res of scale * mean(res^2))
__global__ __launch_bounds__(256) void k_loss_seed(float* __restrict__ g, const float* __restrict__ res, float c, int has_in, long total, SyncHook hook) {
    sync_hook_begin(hook);   // (flag sync of the backward sweep: hn_train.hip, backward_step)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) g[i] = has_in ? fmaf(c, res[i], g[i]) : c * res[i];
    sync_hook_end(hook);
}
// loss = scale * sum(sumsq) / count, summed in a fixed order by one wavefront
__global__ void k_loss_finalize(const float* __restrict__ sumsq, int n, float scale_over_count, float* __restrict__ loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += sumsq[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (threadIdx.x == 0) loss[0] = s * scale_over_count;
}
// A-operand fragments of the fp32 matrix-core 8x8 kernels (hn_mfma.hip: pack_frag_down / pack_frag_up), built on the device.
//   up == 0: raw is [out][in][8][8] -> [in][kx][64 lanes]:      lane -> (co = (l & 15) >> 1, h = l & 1, k = l >> 4): raw[co][ci][4h + k][kx]
//   up == 1: raw is [in][out][8][8] -> [in][px][bb][64 lanes]:  lane -> (co, py = l & 1, a = l >> 4): raw[ci][co][6 + py - 2a][7 - px - 2bb]
// One launch for every level: blockIdx.y = 4 level + which (0 down forward, 1 down backward-data, 2 up forward, 3 up backward-data).
struct PackK8Jobs { int off[4 * kMaxDepth]; int up[4 * kMaxDepth]; };   // blob offset of the raw weight; read as a transposed convolution
__global__ __launch_bounds__(256) void k_pack_frag8(const float* __restrict__ blob, float* __restrict__ k8, PackK8Jobs jobs) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= kFeat * kFeat * 64) return;
    const float* raw = blob + jobs.off[blockIdx.y];
    float* dst = k8 + (size_t)blockIdx.y * 4096;
    const int l = e & 63, blk = (e >> 6) & 7, ci = e >> 9;
    const int co = (l & 15) >> 1, j = l & 1, q = l >> 4;
    if (!jobs.up[blockIdx.y]) dst[e] = raw[((co * kFeat + ci) * 8 + 4 * j + q) * 8 + blk];
    else {
        const int px = blk >> 2, bb = blk & 3;
        dst[e] = raw[((ci * kFeat + co) * 8 + (6 + j - 2 * q)) * 8 + (7 - px - 2 * bb)];
    }
}
// 3x3 weights raw [O][I][3][3] -> both arrangements k_conv3 reads: forward [I][9][O] at dst + off, backward-data [O][9][I]
// (taps flipped) at dst + total + off.  One launch packs every 3x3 convolution of the network (blockIdx.y = job).
struct Pack3Jobs { int n; int off[2 * (3 * kMaxDepth + 2)]; short o[2 * (3 * kMaxDepth + 2)], i[2 * (3 * kMaxDepth + 2)]; };
__global__ __launch_bounds__(256) void k_pack3(const float* __restrict__ raw, float* __restrict__ dst, long total, Pack3Jobs jobs) {
    const int j = blockIdx.y, O = jobs.o[j], I = jobs.i[j], off = jobs.off[j];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= O * I * 9) return;
    {   // forward: index (ci * 9 + k) * O + co
        const int co = e % O, r = e / O, k = r % 9, ci = r / 9;
        dst[pk_off(off) + e] = raw[off + (co * I + ci) * 9 + k];
    }
    {   // backward-data: input channel = forward output o, output channel = forward input i: index (o * 9 + k) * I + i
        const int i = e % I, r = e / I, k = r % 9, o = r / 9;
        dst[total + pk_off(off) + e] = raw[off + (o * I + i) * 9 + (8 - k)];
    }
}
// 3x3 weights of the 8-channel DoubleConvs as A-operand fragments of the fp32 matrix-core kernels (hn_mfma.hip, pack_frag_3x3):
// raw [8][cin][3][3] -> [cin][3][64], lane l -> (co = (l & 15) >> 1, dxo = l & 1, t = l >> 4): raw[co][ci][dy][t - dxo] where that tap exists.
struct F3Layout { size_t inc[2], sig[kMaxDepth][2], dec[kMaxDepth + 1][2], incb[2], sigb[kMaxDepth][2], decb[kMaxDepth + 1][2], stb[kMaxDepth][2], total; };
F3Layout f3_layout(int depth) {
    F3Layout F{};
    size_t pos = 0;
    auto take = [&](size_t (&o)[2], int cin) { o[0] = pos; pos += (size_t)cin * 192; o[1] = pos; pos += (size_t)kFeat * 192; };
    take(F.inc, kInCh);
    for (int d = 0; d < depth; ++d) take(F.sig[d], kFeat + kState);
    for (int d = 0; d <= depth; ++d) take(F.dec[d], d < depth ? 2 * kFeat : kFeat);
    // backward-data fragments (k_dc_bwd_mfma_p): [0] conv2^T, 8 fragment channels; [1] conv1^T, ceil(cin / 8) passes of 8
    auto takeb = [&](size_t (&o)[2], int cin) { o[0] = pos; pos += (size_t)kFeat * 192; o[1] = pos; pos += (size_t)cdiv(cin, kFeat) * kFeat * 192; };
    takeb(F.incb, kInCh);
    for (int d = 0; d < depth; ++d) takeb(F.sigb[d], kFeat + kState);
    for (int d = 0; d <= depth; ++d) takeb(F.decb[d], d < depth ? 2 * kFeat : kFeat);
    // the hidden-state DoubleConv's (10 -> 2 -> 2): conv2^T [2][3][64]; conv1^T two passes of [2][3][64] (k_dc_bwd_mfma_aux)
    for (int d = 0; d < depth; ++d) { F.stb[d][0] = pos; pos += (size_t)kState * 192; F.stb[d][1] = pos; pos += (size_t)2 * kState * 192; }
    F.total = pos;
    return F;
}
// mode 0: the forward convolution raw [8][cin][3][3] -> [cin][3][64].  mode 1: the transposed convolution of raw [8 co][8 cm][3][3] (conv2) -- fragment
// channel = co, output = cm, taps flipped.  mode 2: the transposed convolution of raw [8 cm][cin][3][3] (conv1) -- fragment channel = cm, output = forward
// input channel 8 pass + .. (zero beyond cin), taps flipped, `rows` = passes * 8 fragment channels.
struct PackF3Jobs { int n; int raw[5 * (2 * kMaxDepth + 2)], dst[5 * (2 * kMaxDepth + 2)]; short cin[5 * (2 * kMaxDepth + 2)], rows[5 * (2 * kMaxDepth + 2)], mode[5 * (2 * kMaxDepth + 2)], cm[5 * (2 * kMaxDepth + 2)]; };
__global__ __launch_bounds__(256) void k_pack_frag3(const float* __restrict__ blob, float* __restrict__ f3, PackF3Jobs jobs) {
    const int j = blockIdx.y, cin = jobs.cin[j], rows = jobs.rows[j], mode = jobs.mode[j], cmid = jobs.cm[j];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * 192) return;
    const int l = e & 63, dy = (e >> 6) % 3, ci = e / 192;
    const int co = (l & 15) >> 1, dx = (l >> 4) - (l & 1);
    float v = 0.f;
    if (dx >= 0 && dx <= 2) {
        if (mode == 0) v = blob[jobs.raw[j] + ((co * cin + ci) * 3 + dy) * 3 + dx];
        else if (mode == 1) {          // raw [rows = co'][cmid][3][3]: fragment channel ci = co', output co = mid channel (zero beyond cmid)
            if (co < cmid) v = blob[jobs.raw[j] + ((ci * cmid + co) * 3 + (2 - dy)) * 3 + (2 - dx)];
        } else {                       // raw [cmid][cin][3][3]: fragment channel = (pass, mid channel), output = forward input channel 8 pass + co (zero beyond cin)
            const int pass = ci / cmid, cm = ci - pass * cmid, fi = pass * kFeat + co;
            if (fi < cin) v = blob[jobs.raw[j] + ((cm * cin + fi) * 3 + (2 - dy)) * 3 + (2 - dx)];
        }
    }
    f3[jobs.dst[j] + e] = v;
}
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
                                              const unsigned char* __restrict__ trainable, size_t n, float step_size, float inv_sqrt_bc2, float b1,
                                              float b2, float eps, float wd, float clip) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || (trainable != nullptr && trainable[i] == 0)) return;
    float g = grad[i];
    if (clip > 0.f) g = g < -clip ? -clip : (g > clip ? clip : g);   // comparisons, not fminf / fmaxf: a NaN gradient stays NaN (torch.clamp_ propagates it too)
    const float w = p[i];
    if (wd != 0.f) g = fmaf(wd, w, g);
    const float mi = m[i] + (g - m[i]) * (1.f - b1);             // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = fmaf(g * (1.f - b2), g, v[i] * b2);         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = w - step_size * (mi / denom);
}

// ---- host-side drivers -------------------------------------------------------------------------------------------------
template <int CO>
void launch_conv3_co(const Conv3Args& a, bool epi, dim3 grid, size_t lds, hipStream_t s) {
    const bool gen = a.act_kind > HN_ACT_LEAKYRELU;
    if (epi && gen) hipLaunchKernelGGL((k_conv3<CO, true, true>), grid, dim3(512), lds, s, a);
    else if (epi) hipLaunchKernelGGL((k_conv3<CO, true, false>), grid, dim3(512), lds, s, a);
    else if (gen) hipLaunchKernelGGL((k_conv3<CO, false, true>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((k_conv3<CO, false, false>), grid, dim3(512), lds, s, a);
}
int launch_conv3(hn_ctx* ctx, int co, bool epi, const Conv3Args& a, int batch, hipStream_t s) {
    const dim3 grid(cdiv(a.W, kC3TW), cdiv(a.H, kC3TH), batch);
    const int ci = a.src[0].nch + a.src[1].nch + a.src[2].nch;
    const size_t lds = sizeof(float) * ((size_t)ci * (kC3TH + 2) * kC3PI + 8);
    switch (co) {
        case 2: launch_conv3_co<2>(a, epi, grid, lds, s); break;
        case 6: launch_conv3_co<6>(a, epi, grid, lds, s); break;
        case 8: launch_conv3_co<8>(a, epi, grid, lds, s); break;
        case 10: launch_conv3_co<10>(a, epi, grid, lds, s); break;
        case 16: launch_conv3_co<16>(a, epi, grid, lds, s); break;
        default: return fail(ctx, HN_ERR_UNSUPPORTED, "internal: no 3x3 kernel for %d output channels", co);
    }
    return HN_OK;
}

// q.job[0 .. njobs) filled by the caller; block runs, tile counts and the LDS size here
int launch_conv3_batch(hn_ctx* ctx, int co, bool epi, Conv3Batch& q, int batch, hipStream_t s) {
    int total = 0;
    size_t lds = 0;
    for (int j = 0; j < q.njobs; ++j) {
        const Conv3Args& a = q.job[j];
        q.tiles_x[j] = cdiv(a.W, kC3TW);
        q.tiles_y[j] = cdiv(a.H, kC3TH);
        q.blk0[j] = total;
        total += q.tiles_x[j] * q.tiles_y[j] * batch;
        const int ci = a.src[0].nch + a.src[1].nch + a.src[2].nch;
        const size_t l = sizeof(float) * ((size_t)ci * (kC3TH + 2) * kC3PI + 8);
        if (l > lds) lds = l;
    }
    q.blk0[q.njobs] = total;
    if (total == 0) return HN_OK;
    const bool gen = q.job[0].act_kind > HN_ACT_LEAKYRELU;
    if (co == 2 && !epi && !gen) hipLaunchKernelGGL((k_conv3_batch<2, false, false>), dim3(total), dim3(512), lds, s, q);
    else if (co == 2 && !epi) hipLaunchKernelGGL((k_conv3_batch<2, false, true>), dim3(total), dim3(512), lds, s, q);
    else if (co == 2 && epi && !gen) hipLaunchKernelGGL((k_conv3_batch<2, true, false>), dim3(total), dim3(512), lds, s, q);
    else if (co == 2 && epi) hipLaunchKernelGGL((k_conv3_batch<2, true, true>), dim3(total), dim3(512), lds, s, q);
    else if (co == 10 && !epi && !gen) hipLaunchKernelGGL((k_conv3_batch<10, false, false>), dim3(total), dim3(512), lds, s, q);
    else if (co == 10 && !epi) hipLaunchKernelGGL((k_conv3_batch<10, false, true>), dim3(total), dim3(512), lds, s, q);
    else return fail(ctx, HN_ERR_UNSUPPORTED, "internal: no batched 3x3 kernel for %d output channels (epilogue %d)", co, (int)epi);
    return HN_OK;
}

// C1 = channels of the first convolution's output (the mid tensor), C2 = of the second's
int launch_dc_small(hn_ctx* ctx, int c1, int c2, bool epi, const DcSmallArgs& q, int batch, hipStream_t s) {
    const int ci = q.a1.src[0].nch + q.a1.src[1].nch + q.a1.src[2].nch;
    const size_t lds = sizeof(float) * (size_t)(ci + c1) * kSmallR * kSmallP;
    const bool gen = q.a1.act_kind > HN_ACT_LEAKYRELU;
    const void* fn = nullptr;
#define HN_DCS(A, B, E) (gen ? reinterpret_cast<const void*>(k_dc_small<A, B, E, true>) : reinterpret_cast<const void*>(k_dc_small<A, B, E, false>))
    if (c1 == 8 && c2 == 8 && !epi) fn = HN_DCS(8, 8, false);
    else if (c1 == 8 && c2 == 8 && epi) fn = HN_DCS(8, 8, true);
    else if (c1 == 8 && c2 == 10 && epi) fn = HN_DCS(8, 10, true);
    else if (c1 == 8 && c2 == 16 && epi) fn = HN_DCS(8, 16, true);
#undef HN_DCS
    if (fn == nullptr) return fail(ctx, HN_ERR_UNSUPPORTED, "internal: no small DoubleConv kernel for %d -> %d channels (epilogue %d)", c1, c2, (int)epi);
    static bool attr_done[2][4] = {};   // the > 64 KB dynamic-LDS opt-in is per function (process-wide, idempotent)
    const int slot = !epi ? 0 : c2 == 8 ? 1 : c2 == 10 ? 2 : 3;
    if (!attr_done[gen][slot]) {
        HN_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * (16 + 8) * kSmallR * kSmallP)));
        attr_done[gen][slot] = true;
    }
    DcSmallArgs args = q;
    void* params[] = {&args};
    HN_HIP(ctx, hipLaunchKernel(fn, dim3(batch), dim3(kSmallNT), params, lds, s));
    return HN_OK;
}

int launch_dc_state_batch(hn_ctx* ctx, bool bwd, DcBatch& q, int batch, hipStream_t s) {
    int total = 0;
    for (int j = 0; j < q.njobs; ++j) {
        q.tiles_x[j] = cdiv(q.job[j].a1.W, kC3TW);
        q.tiles_y[j] = cdiv(q.job[j].a1.H, kC3TH);
        q.blk0[j] = total;
        total += q.tiles_x[j] * q.tiles_y[j] * batch;
    }
    q.blk0[q.njobs] = total;
    if (total == 0) return HN_OK;
    const bool gen = q.job[0].a1.act_kind > HN_ACT_LEAKYRELU;
    if (bwd && gen) hipLaunchKernelGGL((k_dc_state_batch<true, true>), dim3(total), dim3(512), 0, s, q);
    else if (bwd) hipLaunchKernelGGL((k_dc_state_batch<true, false>), dim3(total), dim3(512), 0, s, q);
    else if (gen) hipLaunchKernelGGL((k_dc_state_batch<false, true>), dim3(total), dim3(512), 0, s, q);
    else hipLaunchKernelGGL((k_dc_state_batch<false, false>), dim3(total), dim3(512), 0, s, q);
    (void)ctx;
    return HN_OK;
}

int launch_dc_bwd_tile(hn_ctx* ctx, int c2, const DcSmallArgs& q, int batch, hipStream_t s) {
    const dim3 grid(cdiv(q.a1.W, kC3TW), cdiv(q.a1.H, kC3TH), batch);
    const bool gen = q.a1.act_kind > HN_ACT_LEAKYRELU;
#define HN_DBT(C) do { if (gen) hipLaunchKernelGGL((k_dc_bwd_tile<C, true>), grid, dim3(512), 0, s, q); else hipLaunchKernelGGL((k_dc_bwd_tile<C, false>), grid, dim3(512), 0, s, q); } while (0)
    switch (c2) {
        case 6: HN_DBT(6); break;
        case 8: HN_DBT(8); break;
        case 10: HN_DBT(10); break;
        case 16: HN_DBT(16); break;
        default: return fail(ctx, HN_ERR_UNSUPPORTED, "internal: no tiled DoubleConv backward kernel for %d input channels", c2);
    }
#undef HN_DBT
    return HN_OK;
}

constexpr int kPartRows = 640;   // rows of the partials table = the most blocks ONE job of a weight-gradient launch uses (96^2 x 32: 1152 tiles of 8 x 32, ~2 each;
                                 // 256 rows measured the same at batch 32 and 3 % slower at batch 128)

struct Trainer {
    hn_ctx* ctx;
    hipStream_t s;
    const float* w;      // device blob (raw layouts)
    RawLayout L;
    int B, n, depth, act;
    long Lst;            // flat state length per channel
    hn_ctx::TrainWs* ws; // this lane's workspace (tape, gradient buffers, partial-sum tables, job tables)
    float* sumsq0;       // &sumsq[0][first sample of this lane]; rows are sumsq_stride floats apart
    int sumsq_stride;

    int side(int d) const { return n >> d; }
    long plane(int d) const { return (long)side(d) * side(d); }
    hn_ctx::TrainWs& T() const { return *ws; }
    float* tape(int t, size_t off) const { return T().tape + (size_t)t * T().step_floats + off; }
    TSrc feat(const float* p, int d, int nch = kFeat, int act_on_load = 0) const { return TSrc{p, nch * plane(d), plane(d), nch, 1.f, act_on_load}; }
    TDst featdst(float* p, int d, int nch = kFeat, int accum = 0) const { return TDst{p, nch * plane(d), plane(d), nch, 1.f, accum}; }
    static TSrc nosrc() { return TSrc{nullptr, 0, 0, 0, 1.f, 0}; }
    static TDst nodst() { return TDst{nullptr, 0, 0, 0, 1.f, 0}; }
    TSrc state_src(const float* flat, int d) const { return TSrc{flat + ctx->state_off[d], 2 * Lst, Lst, kState, 1.f, 0}; }
    TDst state_dst(float* flat, int d, int accum = 0) const { return TDst{flat + ctx->state_off[d], 2 * Lst, Lst, kState, 1.f, accum}; }
    Src msrc(const float* p, int d) const { return Src{p, kFeat * plane(d), plane(d), 1.f}; }
    Dst mdst(float* p, int d) const { return Dst{p, kFeat * plane(d), plane(d)}; }
    const float* wfwd(size_t off) const { return ctx->tr.w3 + pk_off((long)off); }            // (packed weights: one copy, in the first lane's workspace)                // k_pack3: forward arrangement at the raw offset
    const float* wbwd(size_t off) const { return ctx->tr.w3 + pk_off((long)L.total) + 2 + pk_off((long)off); }      // backward-data arrangement behind it
    float* table(size_t col) const { return T().part + col; }                   // &table[0][col]; rows are L.total floats apart
    const float* frag8(int d, int which) const { return ctx->tr.k8 + ((size_t)d * 4 + which) * 4096; }   // 0 down fwd, 1 down bwd-data, 2 up fwd, 3 up bwd-data

    Conv3Args fwd_args(const TSrc (&src)[3], size_t w_off, size_t b_off, size_t slope_off, TDst dst, int d) const {
        Conv3Args a{};
        for (int i = 0; i < 3; ++i) a.src[i] = src[i];
        a.dst[0] = dst; a.dst[1] = nodst(); a.dst[2] = nodst();
        a.wpk = wfwd(w_off); a.bias = w + b_off;
        a.H = a.W = side(d);
        a.act_kind = act; a.slope = w + slope_off;
        return a;
    }
    int conv_fwd(const TSrc (&src)[3], size_t w_off, size_t b_off, int w_o, size_t slope_off, TDst dst, int d) {
        return launch_conv3(ctx, w_o, false, fwd_args(src, w_off, b_off, slope_off, dst, d), B, s);
    }
    // DoubleConv forward with tape: z = conv1(in) (stored), out = conv2(act(z))
    bool small_level(int d) const { return side(d) <= kSmallS; }
    F3Layout F3;
    bool overlap = true;     // HN_OPT_TRAIN_OVERLAP 0 (A/B): the weight-gradient launches in line on the chain's stream (the r3 path)
    int wg_cap = 0;          // > 0: at most this many blocks per weight-gradient launch (HN_OPT_TRAIN_OVERLAP 2, see hn_train_grad)
    bool tile_small = false; // HN_OPT_TRAIN_FUSED bit 3 (A/B): the small levels' backward DoubleConvs on the tiled kernel too instead of the per-sample k_dc_small
    bool fused_state = true; // HN_OPT_TRAIN_FUSED bit 2 (A/B): the hidden-state DoubleConvs as two batched launches per direction (the r3 path)
    unsigned fwd_epoch = 0;  // (flag sync: the running forward iteration's epoch)
    int deferred_t = -1, deferred_par = 0;   // (flag sync: the backward iteration whose weight-gradient launches are enqueued behind the NEXT iteration's loss-seed kernel)
    bool flag_sync = false;  // HN_OPT_SIDE_SYNC (hn_internal.h: sync_flags): the side stream's releases and joins of the sweeps through device words carried by k_down_mfma / k_up_mfma /
                             // k_loss_seed instead of event packets (words 64 / 96: forward release / join; 128 / 160: backward).  One lane, not under capture.
    bool side_state_fwd = false; // forward sweep: the hidden-state DoubleConvs of an iteration on the (then idle) weight-gradient stream, joined in front of the next iteration's conv_signal
    bool merge_state = true; // (with bits 2 and 4) the hidden-state DoubleConv's backward-data pass rides in the decoder's launch of the same level (k_dc_bwd_mfma_aux)
    bool mfma_bwd = true;    // HN_OPT_TRAIN_FUSED bit 4: the backward-data pass of the 8-channel DoubleConvs on the fp32 matrix core (k_dc_bwd_mfma_p) instead of the vector-pipe kernels
    bool fused_bwd = true;   // HN_OPT_TRAIN_FUSED bit 1 (A/B): the big levels' backward DoubleConvs as two k_conv3 launches (the r3 path)
    bool fused_fwd = true;   // HN_OPT_TRAIN_FUSED 0 (A/B): every convolution of the forward pass as its own direct launch (the r3 path)
    int dc_fwd(const RawDc& dc, const TSrc (&in)[3], float* z, TDst out, int d, const size_t (*f3)[2] = nullptr) {
        // 8-channel DoubleConvs: the fused matrix-core kernels of the inference path, which also store the pre-activation mid tensor
        // (one launch instead of two, at 0.4-0.6 of the fp32 peak instead of 0.16; VERDICT r3 #2a)
        if (fused_fwd && f3 != nullptr && dc.cm == kFeat && dc.co == kFeat && dc8_tape_applies(side(d), side(d)) && out.scale == 1.f && !out.accum &&
            in[0].act == 0 && in[1].act == 0 && in[2].act == 0) {
            const int kind = dc.cin == kInCh ? 0 : dc.cin == kFeat + kState ? 1 : dc.cin == kFeat ? 2 : 3;
            auto ms = [](const TSrc& t) { return Src{t.p, t.sb, t.sc, t.scale}; };
            return launch_dc8_tape(ctx, kind, ms(in[0]), ms(in[1]), ms(in[2]), Dst{out.p, out.sb, out.sc}, ctx->tr.f3 + (*f3)[0], w + dc.b1, w + dc.slope,
                                   ctx->tr.f3 + (*f3)[1], w + dc.b2, act, z, side(d), side(d), B, s);
        }
        if (small_level(d) && dc.cm == kFeat && dc.co == kFeat) {   // both convolutions in one launch (k_dc_small)
            const TSrc mid[3] = {feat(z, d, dc.cm, 1), nosrc(), nosrc()};
            const DcSmallArgs q{fwd_args(in, dc.w1, dc.b1, dc.slope, featdst(z, d, dc.cm), d), fwd_args(mid, dc.w2, dc.b2, dc.slope, out, d)};
            return launch_dc_small(ctx, dc.cm, dc.co, false, q, B, s);
        }
        int rc = conv_fwd(in, dc.w1, dc.b1, dc.cm, dc.slope, featdst(z, d, dc.cm), d);
        if (rc != HN_OK) return rc;
        const TSrc mid[3] = {feat(z, d, dc.cm, 1), nosrc(), nosrc()};
        return conv_fwd(mid, dc.w2, dc.b2, dc.co, dc.slope, out, d);
    }
    // Weight gradients are not on the backward chain: every call below only files a job; flush_wgrads() at the end of the
    // iteration runs them as three launches (3x3 with 8 / with 2 output channels, 8x8) whose grids are the concatenation of the
    // jobs' block runs.  (One launch per layer -- 36 per iteration -- cost 6.2 of the 16.4 ms of a step at 96^2 x 32: at the small
    // levels a launch is 1-2 tiles per block on a fraction of the chip.)
    std::vector<Wg3Args> jobs8, jobs2;
    std::vector<Wg8Args> jobsk;
    size_t lds8 = 0, lds2 = 0;
    int wgrad3(const TSrc (&in)[3], TSrc g, int co, size_t grad_off, int d, size_t slope_off) {
        Wg3Args a{};
        for (int i = 0; i < 3; ++i) a.src[i] = in[i];
        a.g = g.p; a.g_sb = g.sb; a.g_sc = g.sc;
        a.H = a.W = side(d);
        a.tiles_x = cdiv(a.W, 32); a.tiles_y = cdiv(a.H, kWgTH); a.batch = B;
        a.act_kind = act; a.slope = w + slope_off;
        a.part = table(grad_off); a.row_stride = (long)L.total;
        const int ntiles = a.tiles_x * a.tiles_y * B;
        a.nblk = ntiles < kPartRows ? ntiles : kPartRows;
        const int ci = in[0].nch + in[1].nch + in[2].nch;
        size_t lds_f = (size_t)kWgTH * 32 * co + (size_t)ci * (kWgTH + 2) * 35;   // gradient tile + input window; the reduction scratch
        if (lds_f < (size_t)256 * 3 * co) lds_f = (size_t)256 * 3 * co;           // (256 * 3 * CO floats) takes their place at the end
        const size_t lds = sizeof(float) * lds_f;
        if (co == 8) { jobs8.push_back(a); if (lds > lds8) lds8 = lds; }
        else if (co == 2) { jobs2.push_back(a); if (lds > lds2) lds2 = lds; }
        else return fail(ctx, HN_ERR_UNSUPPORTED, "internal: no weight-gradient kernel for %d output channels", co);
        return HN_OK;
    }
    template <class J>
    static int number_blocks(std::vector<J>& jobs, int cap = 0) {
        int total = 0;
        for (J& j : jobs) total += j.nblk;
        if (cap > 0 && total > cap) {   // fewer, longer-running blocks (a block walks more tiles): the launch leaves CU slots to a concurrent chain
            const int t0 = total;
            for (J& j : jobs) { const int nb = (int)((long)j.nblk * cap / t0); j.nblk = nb < 1 ? 1 : nb; }
        }
        total = 0;
        for (J& j : jobs) { j.blk0 = total; total += j.nblk; }
        return total;
    }
    // the jobs filed during the backward pass of iteration t: table -> device (stream-ordered copy out of this iteration's own
    // region of the pinned buffer), then the three launches
    int flush_wgrads(int t, hipStream_t s) {   // (`s` shadows the chain's stream: the launches go where the caller says)
        auto& W = T();
        const size_t b8 = jobs8.size() * sizeof(Wg3Args), b2 = jobs2.size() * sizeof(Wg3Args), bk = jobsk.size() * sizeof(Wg8Args);
        if (b8 + b2 + bk > W.jobs_region) return fail(ctx, HN_ERR_STATE, "internal: weight-gradient job table overflow (%zu > %zu bytes)", b8 + b2 + bk, W.jobs_region);
        const int n8 = number_blocks(jobs8, wg_cap), n2 = number_blocks(jobs2, wg_cap / 4), nk = number_blocks(jobsk, wg_cap / 2);
        unsigned char* h = W.jobs_host + ((size_t)W.jobs_set * W.jobs_rows + t) * W.jobs_region;
        unsigned char* dv = W.jobs_dev + (size_t)t * W.jobs_region;
        if (b8) std::memcpy(h, jobs8.data(), b8);
        if (b2) std::memcpy(h + b8, jobs2.data(), b2);
        if (bk) std::memcpy(h + b8 + b2, jobsk.data(), bk);
        HN_HIP(ctx, hipMemcpyAsync(dv, h, b8 + b2 + bk, hipMemcpyHostToDevice, s));
        const bool gen = act > HN_ACT_LEAKYRELU;
        const Wg3Args* d8 = reinterpret_cast<const Wg3Args*>(dv);
        const Wg3Args* d2 = reinterpret_cast<const Wg3Args*>(dv + b8);
        if (n8 && gen) hipLaunchKernelGGL((k_conv3_wgrad<8, true>), dim3(n8), dim3(256), lds8, s, d8, (int)jobs8.size());
        else if (n8) hipLaunchKernelGGL((k_conv3_wgrad<8, false>), dim3(n8), dim3(256), lds8, s, d8, (int)jobs8.size());
        if (n2 && gen) hipLaunchKernelGGL((k_conv3_wgrad<2, true>), dim3(n2), dim3(256), lds2, s, d2, (int)jobs2.size());
        else if (n2) hipLaunchKernelGGL((k_conv3_wgrad<2, false>), dim3(n2), dim3(256), lds2, s, d2, (int)jobs2.size());
        if (nk) hipLaunchKernelGGL(k_conv8_wgrad, dim3(nk), dim3(512), 0, s, reinterpret_cast<const Wg8Args*>(dv + b8 + b2), (int)jobsk.size());
        jobs8.clear(); jobs2.clear(); jobsk.clear();
        return HN_OK;
    }
    // DoubleConv backward: g_out (co channels) -> weight gradients, gradients of the inputs into `gin` (channel groups of the concatenation)
    // the two backward-data convolutions of a DoubleConv: g_z = conv2^T(g_out) * act'(z) (+ d slope), g_in = conv1^T(g_z)
    Conv3Args bwd2_args(const RawDc& dc, int slot, const float* z, TSrc g_out, int d) const {
        Conv3Args a{};
        a.src[0] = g_out; a.src[1] = nosrc(); a.src[2] = nosrc();
        a.dst[0] = featdst(T().gz[slot], d, dc.cm); a.dst[1] = nodst(); a.dst[2] = nodst();
        a.wpk = wbwd(dc.w2); a.bias = nullptr;
        a.H = a.W = side(d);
        a.act_kind = act; a.slope = w + dc.slope;
        a.z = z; a.z_sb = dc.cm * plane(d); a.z_sc = plane(d);
        a.slope_part = act == HN_ACT_PRELU ? T().slope_part + (size_t)slot * T().slope_stride : nullptr;
        return a;
    }
    Conv3Args bwd1_args(const RawDc& dc, int slot, const TDst (&gin)[3], int d) const {
        Conv3Args a{};
        a.src[0] = feat(T().gz[slot], d, dc.cm); a.src[1] = nosrc(); a.src[2] = nosrc();
        for (int i = 0; i < 3; ++i) a.dst[i] = gin[i];
        a.wpk = wbwd(dc.w1); a.bias = nullptr;
        a.H = a.W = side(d);
        a.act_kind = act; a.slope = nullptr;
        return a;
    }
    // both weight gradients of a DoubleConv (filed; gz[slot] -- its own buffer per DoubleConv -- is read at the end of the iteration)
    int dc_wgrads(const RawDc& dc, int slot, const TSrc (&in)[3], const float* z, TSrc g_out, int d) {
        const TSrc mid[3] = {feat(z, d, dc.cm, 1), nosrc(), nosrc()};
        int rc = wgrad3(mid, g_out, dc.co, dc.w2, d, dc.slope);                              // dW2, db2 from (act(z), g_out)
        if (rc != HN_OK) return rc;
        return wgrad3(in, feat(T().gz[slot], d, dc.cm), dc.cm, dc.w1, d, dc.slope);          // dW1, db1 from (in, g_z)
    }
    // DoubleConv backward: g_out (co channels) -> weight gradients, gradients of the inputs into `gin` (channel groups of the concatenation)
    int dc_bwd(const RawDc& dc, int slot, const TSrc (&in)[3], const float* z, TSrc g_out, const TDst (&gin)[3], int d, const size_t (*fb)[2] = nullptr) {
        int rc;
        if ((rc = dc_wgrads(dc, slot, in, z, g_out, d)) != HN_OK) return rc;
        if (mfma_bwd && fb != nullptr && dc.cm == kFeat && dc.co == kFeat && g_out.act == 0 && g_out.scale == 1.f && dc8_bwd_applies(side(d), side(d)) &&
            dc8_bwd_tiles(side(d), side(d), B) <= (int)T().slope_stride) {
            McBwd a{};
            a.g = g_out.p; a.g_sb = g_out.sb; a.g_sc = g_out.sc;
            a.a1 = ctx->tr.f3 + (*fb)[0]; a.a2 = ctx->tr.f3 + (*fb)[1];
            a.z = z; a.z_sb = dc.cm * plane(d); a.z_sc = plane(d);
            a.gz = T().gz[slot]; a.gz_sb = dc.cm * plane(d); a.gz_sc = plane(d);
            a.slope = w + dc.slope; a.act = act;
            a.slope_part = act == HN_ACT_PRELU ? T().slope_part + (size_t)slot * T().slope_stride : nullptr;
            for (int k = 0; k < 3; ++k) a.dst[k] = McBwdDst{gin[k].p, gin[k].sb, gin[k].sc, gin[k].p != nullptr ? gin[k].nch : 0, gin[k].scale, gin[k].accum};
            // a discarded group in the middle keeps its channel count (the groups behind it keep their channel numbers)
            for (int k = 0; k < 3; ++k) if (gin[k].p == nullptr) a.dst[k].nch = gin[k].nch;
            return launch_dc8_bwd(ctx, a, dc.cin, side(d), side(d), B, s);
        }
        if (!tile_small && small_level(d) && dc.cm == kFeat && (dc.cin == kFeat || dc.cin == kFeat + kState || dc.cin == 2 * kFeat)) {
            const DcSmallArgs q{bwd2_args(dc, slot, z, g_out, d), bwd1_args(dc, slot, gin, d)};
            return launch_dc_small(ctx, dc.cm, dc.cin, true, q, B, s);
        }
        if (fused_bwd && dc.cm == kFeat && dc.co == kFeat && g_out.act == 0) {   // big levels: both backward-data convolutions in one tiled launch
            const DcSmallArgs q{bwd2_args(dc, slot, z, g_out, d), bwd1_args(dc, slot, gin, d)};
            return launch_dc_bwd_tile(ctx, dc.cin, q, B, s);
        }
        if ((rc = launch_conv3(ctx, dc.cm, true, bwd2_args(dc, slot, z, g_out, d), B, s)) != HN_OK) return rc;
        return launch_conv3(ctx, dc.cin, false, bwd1_args(dc, slot, gin, d), B, s);
    }
    // conv_state's backward-data pass inside the decoder's launch: needs the matrix-core backward kernels at every level and the z-stride layout of the tape
    bool merged_state() const {
        if (!(merge_state && mfma_bwd && fused_state)) return false;
        for (int d = 0; d < depth; ++d)
            if (!dc8_bwd_applies(side(d), side(d)) || dc8_bwd_tiles(side(d), side(d), B) > (int)T().slope_stride) return false;
        return true;
    }
    int slot_inc() const { return 0; }
    int slot_sig(int d) const { return 1 + d; }
    int slot_st(int d) const { return 1 + depth + d; }
    int slot_dec(int d) const { return 1 + 2 * depth + d; }
    int slope_rows(int d) const { (void)d; return (int)T().slope_stride; }   // every row of the slot (unused rows are zero): the kernels of a slot differ in their tile counts

    void wgrad8(const float* sm, int d_small, const float* bg, size_t grad_off, int bias_from_big) {   // weight [4096] and bias [8] are adjacent in the blob
        Wg8Args a{};
        a.bias_from_big = bias_from_big;
        a.sm = sm; a.sm_sb = kFeat * plane(d_small); a.sm_sc = plane(d_small); a.hs = a.ws = side(d_small);
        a.bg = bg; a.bg_sb = kFeat * plane(d_small - 1); a.bg_sc = plane(d_small - 1);
        a.tiles_x = cdiv(a.ws, 16); a.tiles_y = cdiv(a.hs, 8); a.batch = B;
        a.part = table(grad_off); a.row_stride = (long)L.total;
        const int ntiles = a.tiles_x * a.tiles_y * B;
        a.nblk = ntiles < kPartRows ? ntiles : kPartRows;
        jobsk.push_back(a);
    }
    // one unrolled iteration, forward (hybridnet.py:558-584), filling step t of the tape
    int forward_step(int t, const float* wf, const float* res, const float* st_in, float* wf_next, float* res_next, float* st_next,
                     const float* ksq, const float* src, int src_batch) {
        auto& W = T();
        int rc;
        const long p0 = plane(0);
        {
            const TSrc in[3] = {TSrc{wf, 2 * p0, p0, 2, 1.f, 0}, TSrc{res, 2 * p0, p0, 2, 1e3f, 0}, TSrc{ctx->tab.sigmas, 0, p0, 2, 1.f, 0}};
            if ((rc = dc_fwd(L.inc, in, tape(t, W.o_zinc), featdst(tape(t, W.o_x[0]), 0), 0, &F3.inc)) != HN_OK) return rc;
        }
        if (W.st_pending) { HN_HIP(ctx, hipStreamWaitEvent(s, W.st_done, 0)); W.st_pending = false; }   // the previous iteration's new states
        for (int d = 0; d < depth; ++d) {
            const TSrc in_sig[3] = {feat(tape(t, W.o_x[d]), d), state_src(st_in, d), nosrc()};
            if ((rc = dc_fwd(L.sig[d], in_sig, tape(t, W.o_zsig[d]), featdst(tape(t, W.o_out[d]), d), d, &F3.sig[d])) != HN_OK) return rc;
            SyncHook hk;   // flag sync: conv_signal of the last level is complete when its `down` starts: the hidden-state launch may go
            if (flag_sync && fused_state && side_state_fwd && d == depth - 1) { fwd_epoch = ++ctx->sync_epoch; hk.store = ctx->sync_flags + 64; hk.store_epoch = fwd_epoch; }
            launch_down(ctx, msrc(tape(t, W.o_out[d]), d), mdst(tape(t, W.o_x[d + 1]), d + 1), frag8(d, 0), w + L.down[d].b, side(d), side(d), B, s, hk);
        }
        {   // new_state_d = conv_state_d(cat[out_d, state_d]) (architectures.py:248) for every level at once: nothing of this iteration
            // reads the new states, so the levels' DoubleConvs are two launches (first convolutions, second convolutions) instead of 2 depth
            Conv3Batch q1{}, q2{};
            DcBatch qf{};
            q1.njobs = q2.njobs = qf.njobs = depth;
            for (int d = 0; d < depth; ++d) {
                const RawDc& dc = L.st[d];
                const TSrc in_st[3] = {feat(tape(t, W.o_out[d]), d), state_src(st_in, d), nosrc()};
                q1.job[d] = fwd_args(in_st, dc.w1, dc.b1, dc.slope, featdst(tape(t, W.o_zst[d]), d, dc.cm), d);
                const TSrc mid[3] = {feat(tape(t, W.o_zst[d]), d, dc.cm, 1), nosrc(), nosrc()};
                q2.job[d] = fwd_args(mid, dc.w2, dc.b2, dc.slope, state_dst(st_next, d), d);
                qf.job[d] = DcSmallArgs{q1.job[d], q2.job[d]};
            }
            if (fused_state && side_state_fwd && flag_sync) {   // ... beside the decoder, released and joined through device words (up_0 below polls)
                launch_sync_gate(ctx, ctx->sync_flags + 64, fwd_epoch, W.wg_stream);
                if ((rc = launch_dc_state_batch(ctx, false, qf, B, W.wg_stream)) != HN_OK) return rc;
                launch_sync_signal(ctx->sync_flags + 96, fwd_epoch, W.wg_stream);
            } else if (fused_state && side_state_fwd) {   // ... beside the decoder: nothing reads the new states before the next iteration
                HN_HIP(ctx, hipEventRecord(W.st_fork, s));
                HN_HIP(ctx, hipStreamWaitEvent(W.wg_stream, W.st_fork, 0));
                if ((rc = launch_dc_state_batch(ctx, false, qf, B, W.wg_stream)) != HN_OK) return rc;
                HN_HIP(ctx, hipEventRecord(W.st_done, W.wg_stream));
                W.st_pending = true;
            } else if (fused_state) {   // both convolutions of every level's hidden-state DoubleConv in ONE launch (k_dc_state_batch)
                if ((rc = launch_dc_state_batch(ctx, false, qf, B, s)) != HN_OK) return rc;
            } else {
                if ((rc = launch_conv3_batch(ctx, kState, false, q1, B, s)) != HN_OK) return rc;
                if ((rc = launch_conv3_batch(ctx, kState, false, q2, B, s)) != HN_OK) return rc;
            }
        }
        {
            const TSrc in[3] = {feat(tape(t, W.o_x[depth]), depth), nosrc(), nosrc()};
            if ((rc = dc_fwd(L.dec[depth], in, tape(t, W.o_zdec[depth]), featdst(tape(t, W.o_y[depth]), depth), depth, &F3.dec[depth])) != HN_OK) return rc;
        }
        for (int d = depth - 1; d >= 0; --d) {
            SyncHook hk;   // flag sync: the iteration's new hidden states are complete when up_0 is
            if (flag_sync && fused_state && side_state_fwd && d == 0) { hk.wait = ctx->sync_flags + 96; hk.wait_epoch = fwd_epoch; hk.err = ctx->sync_err_dev; }
            launch_up(ctx, msrc(tape(t, W.o_y[d + 1]), d + 1), mdst(tape(t, W.o_u[d]), d), frag8(d, 2), w + L.up[d].b, side(d + 1), side(d + 1), B, s, false, hk);
            const TSrc in[3] = {feat(tape(t, W.o_u[d]), d), feat(tape(t, W.o_out[d]), d), nosrc()};
            if ((rc = dc_fwd(L.dec[d], in, tape(t, W.o_zdec[d]), featdst(tape(t, W.o_y[d]), d), d, &F3.dec[d])) != HN_OK) return rc;
        }
        const long total = (long)B * p0;
        hipLaunchKernelGGL(k_outc_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, tape(t, W.o_y[0]), w + L.outc_w, w + L.outc_b, wf, wf_next, p0, total);
        return spec_apply(ctx, wf_next, res_next, ksq, src, src_batch, B, sumsq0 + (size_t)t * sumsq_stride, s);
    }

    // backward of iteration t.  On entry g_wf[cur_wf] / g_res / g_st[cur_st] hold d loss / d (wf, res, states) AFTER iteration t without
    // this iteration's own loss term (zeros for the last iteration); on exit they hold the gradients with respect to the
    // iteration's inputs (the buffer indices flip).
    int backward_step(int t, const float* wf_in, const float* res_in, const float* st_in, const float* res_out, const float* ksq,
                      float loss_c, int& cur_wf, int& cur_st) {
        auto& W = T();
        int rc;
        const long p0 = plane(0), tot2 = (long)B * 2 * p0;
        const int par = t & 1;
        if (overlap) {   // this iteration writes buffer set `par`: the weight gradients of iteration t + 2, which read it, must be done
            if (W.wg_pending[par] && !(flag_sync && W.wg_flag_epoch[par] != 0)) { HN_HIP(ctx, hipStreamWaitEvent(s, W.wg_done[par], 0)); W.wg_pending[par] = false; }
            select_gset(W, par);
        } else select_gset(W, 0);
        // flag sync: the loss-seed kernel, first of the iteration's chain and no user of the buffer sets, carries both hand-overs -- its first thread stores the
        // release word of iteration t + 1's weight-gradient launches (everything before it on the chain is complete) and, after its own work, waits for the
        // join word of iteration t + 2's, whose buffer set this iteration overwrites
        SyncHook hk;
        if (flag_sync && W.bwd_release_epoch != 0) { hk.store = ctx->sync_flags + 128; hk.store_epoch = W.bwd_release_epoch; W.bwd_release_epoch = 0; }
        if (flag_sync && overlap && W.wg_pending[par] && W.wg_flag_epoch[par] != 0) {
            hk.wait = ctx->sync_flags + 160; hk.wait_epoch = W.wg_flag_epoch[par]; hk.err = ctx->sync_err_dev;
            W.wg_flag_epoch[par] = 0; W.wg_pending[par] = false;
        }
        // this iteration's loss term, then the adjoint of the residual operator: G = g_wf + L^H(g_res) + ksq * g_res
        hipLaunchKernelGGL(k_loss_seed, dim3((unsigned)((tot2 + 255) / 256)), dim3(256), 0, s, W.g_res, res_out, loss_c, 1, tot2, hk);
        if (deferred_t >= 0) {
            // the weight-gradient launches of iteration t + 1, gate first: enqueued HERE, behind the kernel that stores their release word, so that every
            // store precedes its wait in submission order too (a tool or setting that runs kernels strictly in that order cannot starve the gate).
            // Their jobs are still the filed ones: nothing of this iteration has been filed yet
            const unsigned e = hk.store_epoch;
            launch_sync_gate(ctx, ctx->sync_flags + 128, e, W.wg_stream);
            if ((rc = flush_wgrads(deferred_t, W.wg_stream)) != HN_OK) return rc;
            launch_sync_signal(ctx->sync_flags + 160, e, W.wg_stream);
            HN_HIP(ctx, hipEventRecord(W.wg_done[deferred_par], W.wg_stream));
            W.wg_pending[deferred_par] = true;
            W.wg_flag_epoch[deferred_par] = e;
            deferred_t = -1;
        }
        if ((rc = spec_adjoint(ctx, W.g_res, W.g_wf[cur_wf ^ 1], ksq, W.g_wf[cur_wf], B, s)) != HN_OK) return rc;
        cur_wf ^= 1;
        float* G = W.g_wf[cur_wf];   // d loss / d wf_next; wf_next = wf + d / 1e3, so it is also the direct part of d loss / d wf
        {
            const long total = (long)B * p0;
            const int blocks = (int)((total + 255) / 256) < kPartRows ? (int)((total + 255) / 256) : kPartRows;
            hipLaunchKernelGGL(k_outc_bwd, dim3(blocks), dim3(256), 0, s, G, tape(t, W.o_y[0]), w + L.outc_w, W.g_y[0], table(L.outc_w), (long)L.total, p0, total);
        }
        {   // conv_state of every level first (new_state = DC(cat[out, state]): its gradient arrives from iteration t + 1 alone), as two
            // batched launches; it OPENS the sums g_out[d] and g_st[.][d] that the decoder, `down` and conv_signal then add to
            Conv3Batch q2{}, q1{};
            q2.njobs = q1.njobs = depth;
            for (int d = 0; d < depth; ++d) {
                const RawDc& dc = L.st[d];
                const TSrc in[3] = {feat(tape(t, W.o_out[d]), d), state_src(st_in, d), nosrc()};
                const TSrc g_new = state_src(W.g_st[cur_st], d);
                if ((rc = dc_wgrads(dc, slot_st(d), in, tape(t, W.o_zst[d]), g_new, d)) != HN_OK) return rc;
                q2.job[d] = bwd2_args(dc, slot_st(d), tape(t, W.o_zst[d]), g_new, d);
                const TDst gin[3] = {featdst(W.g_out[d], d, kFeat, 0), state_dst(W.g_st[(cur_st + 1) % 3], d, 0), nodst()};
                q1.job[d] = bwd1_args(dc, slot_st(d), gin, d);
            }
            if (merged_state()) {
                // (k_dc_bwd_mfma_aux: each level's pass rides in the decoder's backward launch below)
            } else if (fused_state) {
                DcBatch qb{};
                qb.njobs = depth;
                for (int d = 0; d < depth; ++d) qb.job[d] = DcSmallArgs{q2.job[d], q1.job[d]};
                if ((rc = launch_dc_state_batch(ctx, true, qb, B, s)) != HN_OK) return rc;
            } else {
                if ((rc = launch_conv3_batch(ctx, kState, true, q2, B, s)) != HN_OK) return rc;
                if ((rc = launch_conv3_batch(ctx, kFeat + kState, false, q1, B, s)) != HN_OK) return rc;
            }
        }
        for (int d = 0; d < depth; ++d) {   // decoder, top down
            const TSrc in[3] = {feat(tape(t, W.o_u[d]), d), feat(tape(t, W.o_out[d]), d), nosrc()};
            if (merged_state()) {   // the decoder's backward-data pass and conv_state's in one launch: g_out[d] = skip gradient + conv_state's, written once
                const RawDc &dc = L.dec[d], &ds = L.st[d];
                if ((rc = dc_wgrads(dc, slot_dec(d), in, tape(t, W.o_zdec[d]), feat(W.g_y[d], d), d)) != HN_OK) return rc;
                McBwd a{};
                a.g = W.g_y[d]; a.g_sb = kFeat * plane(d); a.g_sc = plane(d);
                a.a1 = ctx->tr.f3 + F3.decb[d][0]; a.a2 = ctx->tr.f3 + F3.decb[d][1];
                a.z = tape(t, W.o_zdec[d]); a.z_sb = kFeat * plane(d); a.z_sc = plane(d);
                a.gz = W.gz[slot_dec(d)]; a.gz_sb = kFeat * plane(d); a.gz_sc = plane(d);
                a.slope = w + dc.slope; a.act = act;
                a.slope_part = act == HN_ACT_PRELU ? W.slope_part + (size_t)slot_dec(d) * W.slope_stride : nullptr;
                a.dst[0] = McBwdDst{W.g_u[d], kFeat * plane(d), plane(d), kFeat, 1.f, 0};
                a.dst[1] = McBwdDst{W.g_out[d], kFeat * plane(d), plane(d), kFeat, 1.f, 0};
                a.dst[2] = McBwdDst{nullptr, 0, 0, 0, 1.f, 0};
                McBwdAux x{};
                const TSrc g_new = state_src(W.g_st[cur_st], d);
                x.g = g_new.p; x.g_sb = g_new.sb; x.g_sc = g_new.sc;
                x.a1 = ctx->tr.f3 + F3.stb[d][0]; x.a2 = ctx->tr.f3 + F3.stb[d][1];
                x.z = tape(t, W.o_zst[d]); x.z_sb = kState * plane(d); x.z_sc = plane(d);
                x.gz = W.gz[slot_st(d)]; x.gz_sb = kState * plane(d); x.gz_sc = plane(d);
                x.slope = w + ds.slope;
                x.slope_part = act == HN_ACT_PRELU ? W.slope_part + (size_t)slot_st(d) * W.slope_stride : nullptr;
                const TDst g_old = state_dst(W.g_st[(cur_st + 1) % 3], d, 0);
                x.dst = McBwdDst{g_old.p, g_old.sb, g_old.sc, kState, 1.f, 0};
                if ((rc = launch_dc8_bwd_aux(ctx, a, x, side(d), side(d), B, s)) != HN_OK) return rc;
            } else {
                const TDst gin[3] = {featdst(W.g_u[d], d), featdst(W.g_out[d], d, kFeat, 1), nodst()};
                if ((rc = dc_bwd(L.dec[d], slot_dec(d), in, tape(t, W.o_zdec[d]), feat(W.g_y[d], d), gin, d, &F3.decb[d])) != HN_OK) return rc;
            }
            // up[d]: backward-data = the stride-2 convolution kernel on the transposed-convolution weights read as [out, in, kh, kw]
            launch_down(ctx, msrc(W.g_u[d], d), mdst(W.g_y[d + 1], d + 1), frag8(d, 3), ctx->tr.zero8, side(d), side(d), B, s);
            wgrad8(tape(t, W.o_y[d + 1]), d + 1, W.g_u[d], L.up[d].w, 1);
        }
        {
            const TSrc in[3] = {feat(tape(t, W.o_x[depth]), depth), nosrc(), nosrc()};
            const TDst gin[3] = {featdst(W.g_x[depth], depth), nodst(), nodst()};
            if ((rc = dc_bwd(L.dec[depth], slot_dec(depth), in, tape(t, W.o_zdec[depth]), feat(W.g_y[depth], depth), gin, depth, &F3.decb[depth])) != HN_OK) return rc;
        }
        for (int d = depth - 1; d >= 0; --d) {   // encoder, bottom up
            // down[d]: backward-data = the transposed-convolution kernel on the convolution weights read as [in, out, kh, kw];
            // added to the skip gradient
            launch_up(ctx, msrc(W.g_x[d + 1], d + 1), mdst(W.g_out[d], d), frag8(d, 1), ctx->tr.zero8, side(d + 1), side(d + 1), B, s, true);
            wgrad8(W.g_x[d + 1], d + 1, tape(t, W.o_out[d]), L.down[d].w, 0);
            {   // conv_signal: out = DC(cat[x, state])
                const TSrc in[3] = {feat(tape(t, W.o_x[d]), d), state_src(st_in, d), nosrc()};
                const TDst gin[3] = {featdst(W.g_x[d], d), state_dst(W.g_st[(cur_st + 1) % 3], d, 1), nodst()};
                if ((rc = dc_bwd(L.sig[d], slot_sig(d), in, tape(t, W.o_zsig[d]), feat(W.g_out[d], d), gin, d, &F3.sigb[d])) != HN_OK) return rc;
            }
        }
        {   // inc: DC(cat[wf, 1e3 * res, sigmas]); the sigma channels need no gradient
            const TSrc in[3] = {TSrc{wf_in, 2 * p0, p0, 2, 1.f, 0}, TSrc{res_in, 2 * p0, p0, 2, 1e3f, 0}, TSrc{ctx->tab.sigmas, 0, p0, 2, 1.f, 0}};
            const TDst gin[3] = {TDst{G, 2 * p0, p0, 2, 1.f, 1}, TDst{W.g_res, 2 * p0, p0, 2, 1e3f, 0}, nodst()};
            if ((rc = dc_bwd(L.inc, slot_inc(), in, tape(t, W.o_zinc), feat(W.g_x[0], 0), gin, 0, &F3.incb)) != HN_OK) return rc;
        }
        cur_st = (cur_st + 1) % 3;
        if (overlap && t == 0 && wg_cap > 0) {
            // the last launches have no chain left to hide behind: in line and at full width, behind the side stream's (they add to the same table rows)
            for (int k = 0; k < 2; ++k)
                if (W.wg_pending[k]) { HN_HIP(ctx, hipStreamWaitEvent(s, W.wg_done[k], 0)); W.wg_pending[k] = false; W.wg_flag_epoch[k] = 0; }
            const int cap = wg_cap;
            wg_cap = 0;
            rc = flush_wgrads(t, s);
            wg_cap = cap;
            if (rc != HN_OK) return rc;
        } else if (overlap && flag_sync && t > 0) {   // ... released by the NEXT iteration's loss-seed kernel (t - 1 exists), joined through a device word or, at the end, the event
            W.bwd_release_epoch = ++ctx->sync_epoch;
            deferred_t = t;        // (gate, launches, signal and the wg_done record follow that loss-seed kernel: top of backward_step(t - 1))
            deferred_par = par;
        } else if (overlap) {   // the filed weight-gradient jobs: three launches on the side stream, beside the backward chain of iteration t - 1
            HN_HIP(ctx, hipEventRecord(W.wg_ready[par], s));
            HN_HIP(ctx, hipStreamWaitEvent(W.wg_stream, W.wg_ready[par], 0));
            if ((rc = flush_wgrads(t, W.wg_stream)) != HN_OK) return rc;
            HN_HIP(ctx, hipEventRecord(W.wg_done[par], W.wg_stream));
            W.wg_pending[par] = true;
            W.wg_flag_epoch[par] = 0;
        } else if ((rc = flush_wgrads(t, s)) != HN_OK) return rc;
        HN_HIP(ctx, hipGetLastError());
        return HN_OK;
    }
};

void select_gset(hn_ctx::TrainWs& W, int k) {   // g_x .. gz name the buffers of the iteration being processed
    const auto& G = W.gset[k];
    for (int d = 0; d <= kMaxDepth; ++d) { W.g_x[d] = G.g_x[d]; W.g_y[d] = G.g_y[d]; }
    for (int d = 0; d < kMaxDepth; ++d) { W.g_out[d] = G.g_out[d]; W.g_u[d] = G.g_u[d]; }
    for (int i = 0; i < 3 * kMaxDepth + 2; ++i) W.gz[i] = G.gz[i];
}

void train_free_ws(hn_ctx::TrainWs& W) {
    for (void* p : {(void*)W.tape, (void*)W.gbuf, (void*)W.part, (void*)W.slope_part, (void*)W.w3, (void*)W.k8, (void*)W.f3, (void*)W.zero8, (void*)W.sumsq, (void*)W.jobs_dev}) (void)hipFree(p);
    if (W.jobs_host != nullptr) (void)hipHostFree(W.jobs_host);
    for (hipEvent_t e : W.jobs_copied)
        if (e != nullptr) (void)hipEventDestroy(e);
    W.wg_stream = nullptr;   // (one of ctx->picks[1 / 2]'s candidates: side_stream_for)
    if (W.st_fork != nullptr) (void)hipEventDestroy(W.st_fork);
    if (W.st_done != nullptr) (void)hipEventDestroy(W.st_done);
    W.st_fork = W.st_done = nullptr; W.st_pending = false;
    for (int k = 0; k < 2; ++k) {
        if (W.wg_ready[k] != nullptr) (void)hipEventDestroy(W.wg_ready[k]);
        if (W.wg_done[k] != nullptr) (void)hipEventDestroy(W.wg_done[k]);
    }
    W = hn_ctx::TrainWs{};
}

// workspace W for `batch` samples x n_unroll iterations; sumsq rows of `sumsq_batch` samples (the first lane keeps the whole batch's)
int train_reserve(hn_ctx* ctx, hn_ctx::TrainWs& W, int batch, int n_unroll, int sumsq_batch, bool allow_regrow_captured = false) {
    const int n = ctx->tab.n, depth = ctx->depth;
    if (W.tape != nullptr && W.n == n && W.depth == depth && batch <= W.batch && n_unroll <= W.n_unroll && sumsq_batch <= W.sumsq_batch) return HN_OK;
    // a graph captured from hn_train_grad points into this workspace (tape, gradient buffers, the pinned job tables it re-copies on every replay): growing it
    // would free that memory under the graph.  hn_train_reserve -- an explicit call -- is how the caller says the graph is gone
    if (W.captured && !allow_regrow_captured)
        return fail(ctx, HN_ERR_STATE, "hn_train_grad: a training step captured into a HIP graph uses this workspace and the call needs a larger one: destroy the "
                                      "graph and call hn_train_reserve first");
    HN_HIP(ctx, hipDeviceSynchronize());
    const bool fresh = W.n != n || W.depth != depth;
    const int nb = fresh || batch > W.batch ? batch : W.batch;
    const int nu = !fresh && n_unroll < W.n_unroll ? W.n_unroll : n_unroll;
    const int nsq = !fresh && sumsq_batch < W.sumsq_batch ? W.sumsq_batch : sumsq_batch;
    train_free_ws(W);
    auto plane = [&](int d) { return (size_t)(n >> d) * (n >> d); };
    size_t pos = 0;
    auto take = [&](size_t ch, int d) { const size_t o = pos; pos += (size_t)nb * ch * plane(d); return o; };
    W.o_zinc = take(kFeat, 0);
    for (int d = 0; d <= depth; ++d) {
        W.o_x[d] = take(kFeat, d);
        W.o_zdec[d] = take(kFeat, d);
        W.o_y[d] = take(kFeat, d);
        if (d < depth) {
            W.o_zsig[d] = take(kFeat, d);
            W.o_out[d] = take(kFeat, d);
            W.o_zst[d] = take(kState, d);
            W.o_u[d] = take(kFeat, d);
        }
    }
    W.step_floats = pos;
    HN_HIP(ctx, hipMalloc((void**)&W.tape, sizeof(float) * W.step_floats * nu));
    size_t g = 0;
    auto gtake = [&](size_t floats) { const size_t o = g; g += floats; return o; };
    size_t o_gx[2][kMaxDepth + 1], o_gy[2][kMaxDepth + 1], o_go[2][kMaxDepth], o_gu[2][kMaxDepth];
    size_t o_gz[2][3 * kMaxDepth + 2];   // slots as Trainer::slot_*: inc, sig[d], st[d], dec[d]
    for (int k = 0; k < 2; ++k) {
        for (int d = 0; d <= depth; ++d) {
            o_gx[k][d] = gtake((size_t)nb * kFeat * plane(d));
            o_gy[k][d] = gtake((size_t)nb * kFeat * plane(d));
            if (d < depth) { o_go[k][d] = gtake((size_t)nb * kFeat * plane(d)); o_gu[k][d] = gtake((size_t)nb * kFeat * plane(d)); }
        }
        o_gz[k][0] = gtake((size_t)nb * kFeat * plane(0));
        for (int d = 0; d < depth; ++d) {
            o_gz[k][1 + d] = gtake((size_t)nb * kFeat * plane(d));
            o_gz[k][1 + depth + d] = gtake((size_t)nb * kState * plane(d));
        }
        for (int d = 0; d <= depth; ++d) o_gz[k][1 + 2 * depth + d] = gtake((size_t)nb * kFeat * plane(d));
    }
    const size_t o_wf0 = gtake((size_t)nb * 2 * plane(0)), o_wf1 = gtake((size_t)nb * 2 * plane(0)), o_res = gtake((size_t)nb * 2 * plane(0));
    size_t o_st[3];
    for (size_t& o : o_st) o = gtake((size_t)nb * kState * ctx->state_len);
    HN_HIP(ctx, hipMalloc((void**)&W.gbuf, sizeof(float) * g));
    for (int k = 0; k < 2; ++k) {
        auto& G = W.gset[k];
        for (int d = 0; d <= depth; ++d) {
            G.g_x[d] = W.gbuf + o_gx[k][d];
            G.g_y[d] = W.gbuf + o_gy[k][d];
            if (d < depth) { G.g_out[d] = W.gbuf + o_go[k][d]; G.g_u[d] = W.gbuf + o_gu[k][d]; }
        }
        for (int i = 0; i < 3 * depth + 2; ++i) G.gz[i] = W.gbuf + o_gz[k][i];
    }
    select_gset(W, 0);
    W.g_wf[0] = W.gbuf + o_wf0; W.g_wf[1] = W.gbuf + o_wf1; W.g_res = W.gbuf + o_res;
    for (int k = 0; k < 3; ++k) W.g_st[k] = W.gbuf + o_st[k];
    HN_HIP(ctx, hipEventCreateWithFlags(&W.st_fork, hipEventDisableTiming));
    HN_HIP(ctx, hipEventCreateWithFlags(&W.st_done, hipEventDisableTiming));
    for (int k = 0; k < 2; ++k) {
        HN_HIP(ctx, hipEventCreateWithFlags(&W.wg_ready[k], hipEventDisableTiming));
        HN_HIP(ctx, hipEventCreateWithFlags(&W.wg_done[k], hipEventDisableTiming));
    }
    const size_t total = raw_layout(depth).total;
    W.part_floats = (size_t)kPartRows * total;
    HN_HIP(ctx, hipMalloc((void**)&W.part, sizeof(float) * W.part_floats));
    W.slope_stride = (size_t)nb * cdiv(n, 16) * cdiv(n, 8);   // one row per tile of the finest tiling in use (k_dc_bwd_mfma_p: 8 x 32, 8 x 16 for W <= 16)
    HN_HIP(ctx, hipMalloc((void**)&W.slope_part, sizeof(double) * W.slope_stride * (3 * depth + 2)));
    HN_HIP(ctx, hipMalloc((void**)&W.w3, sizeof(float) * 2 * (total + 4)));   // both arrangements, each tensor at an even offset (pk_off)
    HN_HIP(ctx, hipMalloc((void**)&W.k8, sizeof(float) * (size_t)depth * 4 * 4096));
    HN_HIP(ctx, hipMalloc((void**)&W.f3, sizeof(float) * f3_layout(depth).total));
    HN_HIP(ctx, hipMalloc((void**)&W.zero8, sizeof(float) * 8));
    HN_HIP(ctx, hipMemset(W.zero8, 0, sizeof(float) * 8));
    HN_HIP(ctx, hipMalloc((void**)&W.sumsq, sizeof(float) * (size_t)nu * nsq));
    W.sumsq_batch = nsq;
    // per iteration: 2 jobs per DoubleConv (3 depth + 2 of them) + 2 depth 8x8 jobs
    W.jobs_region = (size_t)(6 * depth + 4) * sizeof(Wg3Args) + (size_t)2 * depth * sizeof(Wg8Args);
    W.jobs_rows = nu;
    HN_HIP(ctx, hipHostMalloc((void**)&W.jobs_host, W.jobs_region * nu * hn_ctx::TrainWs::kJobSets, hipHostMallocDefault));
    HN_HIP(ctx, hipMalloc((void**)&W.jobs_dev, W.jobs_region * nu));
    for (hipEvent_t& e : W.jobs_copied) HN_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    W.batch = nb; W.n_unroll = nu; W.n = n; W.depth = depth;
    return HN_OK;
}

int train_ready(hn_ctx* ctx, int batch, int n_unroll) {
    if (!ctx) return HN_ERR_ARG;
    if (!ctx->have_weights) return fail(ctx, HN_ERR_STATE, "hn_load_weights has not been called (it defines depth and activation)");
    if (ctx->tab.n == 0) return fail(ctx, HN_ERR_STATE, "hn_set_domain has not been called");
    if (batch <= 0 || n_unroll <= 0) return fail(ctx, HN_ERR_ARG, "batch and n_unroll must be positive (got %d, %d)", batch, n_unroll);
    if (ctx->tab.n % (1 << ctx->depth) != 0)
        return fail(ctx, HN_ERR_ARG, "domain size %d is not divisible by 2^depth = %d", ctx->tab.n, 1 << ctx->depth);
    return HN_OK;
}

}  // namespace

void train_free(hn_ctx* ctx) {
    train_free_ws(ctx->tr);
    train_free_ws(ctx->tr_b);
    if (ctx->train_fork != nullptr) (void)hipEventDestroy(ctx->train_fork);
    if (ctx->train_join != nullptr) (void)hipEventDestroy(ctx->train_join);
    ctx->train_stream = nullptr;
    ctx->train_fork = ctx->train_join = nullptr;
}

}  // namespace hn

using namespace hn;

extern "C" {

int hn_train_reserve(hn_ctx* ctx, int batch, int n_unroll) {
    int rc = train_ready(ctx, batch, n_unroll);
    if (rc != HN_OK) return rc;
    DeviceGuard guard(ctx);
    const int lanes = ctx->opt_train_lanes >= 2 && batch >= 2 ? 2 : 1;
    const int b0 = lanes == 2 ? (batch + 1) / 2 : batch;
    ctx->tr.captured = ctx->tr_b.captured = false;   // an explicit reserve is the caller's word that no captured training step is replayed any more
    rc = train_reserve(ctx, ctx->tr, b0, n_unroll, batch, true);
    if (rc == HN_OK && lanes == 2) rc = train_reserve(ctx, ctx->tr_b, batch - b0, n_unroll, 1, true);
    return rc;
}

int hn_train_grad(hn_ctx* ctx, const float* weights, const float* wf, const float* res, const float* states, const float* k_sq,
                  const float* src, int src_batch, int batch, int n_unroll, float loss_scale, float* wf_hist, float* res_hist,
                  float* st_hist, float* loss, float* grad, float* grad_wf0, float* grad_res0, float* grad_st0, void* stream) {
    if (!ctx || !weights || !wf || !res || !states || !k_sq || !src || !wf_hist || !res_hist || !st_hist || !loss || !grad)
        return fail(ctx, HN_ERR_ARG, "hn_train_grad: NULL argument");
    int rc = train_ready(ctx, batch, n_unroll);
    if (rc != HN_OK) return rc;
    if ((rc = check_async(ctx, "hn_train_grad (an earlier call)")) != HN_OK) return rc;   // a device-side wait that gave up: tape / gradients since then are not to be trusted
    if (src_batch != 1 && src_batch != batch) return fail(ctx, HN_ERR_ARG, "source batch %d must be 1 or equal to the batch %d", src_batch, batch);
    if (ctx->train_fwd_sumsq != nullptr && (int64_t)n_unroll * batch > ctx->train_fwd_sumsq_cap)
        return fail(ctx, HN_ERR_ARG, "hn_train_grad: the host table of hn_train_set_forward_event holds %lld floats, this call writes %lld", (long long)ctx->train_fwd_sumsq_cap,
                    (long long)n_unroll * batch);
    DeviceGuard guard(ctx);
    hipStream_t s = (hipStream_t)stream;
    const int n = ctx->tab.n, depth = ctx->depth;
    // Two lanes: samples are independent (hybridnet.py:399-409 is a batched call), and at the training size the chain of ~800 launches
    // is latency-bound (a step of 16 samples takes 0.83 of the time of 32), so the two halves of the batch run as two chains on
    // two streams.  Each lane has its own workspace and partial-sum tables; the packed weights and the sumsq rows are shared.
    const int lanes = ctx->opt_train_lanes >= 2 && batch >= 2 ? 2 : 1;
    const int lane_b0[2] = {0, lanes == 2 ? (batch + 1) / 2 : batch};
    const int lane_nb[2] = {lane_b0[1], batch - lane_b0[1]};
    hn_ctx::TrainWs* const ws[2] = {&ctx->tr, &ctx->tr_b};
    {   // a caller recording the step into a HIP graph: the workspace must already be large enough (growing it synchronises the device
        // and frees / allocates memory, which would invalidate the capture with an opaque HIP error)
        hipStreamCaptureStatus cap0 = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing((hipStream_t)stream, &cap0);
        auto fits = [&](const hn_ctx::TrainWs& W, int nb, int nsq) {
            return W.tape != nullptr && W.n == n && W.depth == depth && nb <= W.batch && n_unroll <= W.n_unroll && nsq <= W.sumsq_batch;
        };
        if (cap0 == hipStreamCaptureStatusActive && (!fits(ctx->tr, lane_nb[0], batch) || (lanes == 2 && (!fits(ctx->tr_b, lane_nb[1], 1) || ctx->train_stream == nullptr))))
            return fail(ctx, HN_ERR_STATE, "hn_train_grad under stream capture needs its workspace in place: call hn_train_reserve(ctx, %d, %d) (or one eager "
                        "hn_train_grad of this shape) before capturing", batch, n_unroll);
    }
    if ((rc = train_reserve(ctx, ctx->tr, lane_nb[0], n_unroll, batch)) != HN_OK) return rc;
    if (lanes == 2 && (rc = train_reserve(ctx, ctx->tr_b, lane_nb[1], n_unroll, 1)) != HN_OK) return rc;
    if (lanes == 2) {   // lane 1's chain: a stream that demonstrably overlaps with the caller's (hn_internal.h: SidePick)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing((hipStream_t)stream, &cs);
        const hipStream_t ref = (hipStream_t)stream;
        if ((rc = side_stream_for(ctx, 7, &ref, 1, cs == hipStreamCaptureStatusNone, &ctx->train_stream)) != HN_OK) return rc;
        if (ctx->train_fork == nullptr) {
            HN_HIP(ctx, hipEventCreateWithFlags(&ctx->train_fork, hipEventDisableTiming));
            HN_HIP(ctx, hipEventCreateWithFlags(&ctx->train_join, hipEventDisableTiming));
        }
    }
    const hipStream_t ls[2] = {s, lanes == 2 ? ctx->train_stream : s};
    // Under stream capture (a caller recording the step into a HIP graph) nothing may wait on the host: the job tables of a captured
    // call are copied from the pinned buffer at every REPLAY, so the caller must not interleave other training calls of this context
    // with replays (they would rewrite the tables); the event bookkeeping of the eager path is skipped.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    const bool capturing = cap == hipStreamCaptureStatusActive;
    if (!capturing && ctx->opt_side_sync == 1 && (rc = ensure_sync_words(ctx)) != HN_OK) return rc;
    for (int l = 0; l < lanes; ++l) {
        auto& W = *ws[l];
        // set 0 of the pinned job tables belongs to CAPTURED calls (their graph copies the tables out of it at every replay, so eager calls
        // must never rewrite it); eager calls rotate over sets 1 .. kJobSets - 1
        if (capturing) { W.jobs_set = 0; W.captured = true; }
        else W.jobs_set = W.jobs_set % (hn_ctx::TrainWs::kJobSets - 1) + 1;
        if (!capturing && W.jobs_in_flight[W.jobs_set]) {   // the tables of the call that last used this set have left the pinned buffer (normally long ago)
            HN_HIP(ctx, hipEventSynchronize(W.jobs_copied[W.jobs_set]));
            W.jobs_in_flight[W.jobs_set] = false;
        }
        W.last_batch = lane_nb[l];
    }
    if (lanes == 1) ctx->tr_b.last_batch = 0;
    const RawLayout L = raw_layout(depth);
    const long p2 = 2L * n * n, pst = (long)kState * ctx->state_len;   // floats per sample of a wavefield / of the flat states
    Trainer tr[2] = {
        Trainer{ctx, ls[0], weights, L, lane_nb[0], n, depth, ctx->act_kind, (long)ctx->state_len, ws[0], ctx->tr.sumsq, batch},
        Trainer{ctx, ls[1], weights, L, lane_nb[1], n, depth, ctx->act_kind, (long)ctx->state_len, ws[1], ctx->tr.sumsq + lane_b0[1], batch}};
    for (Trainer& t : tr) { t.F3 = f3_layout(depth); t.fused_fwd = (ctx->opt_train_fused & 1) != 0; t.fused_bwd = (ctx->opt_train_fused & 2) != 0; t.fused_state = (ctx->opt_train_fused & 4) != 0; t.tile_small = (ctx->opt_train_fused & 8) != 0; t.mfma_bwd = (ctx->opt_train_fused & 16) != 0; t.merge_state = (ctx->opt_train_fused & 32) != 0; }
    // Weight gradients beside the chain (HN_OPT_TRAIN_OVERLAP).  They are leaves of the backward pass, ~2 of the 9 ms of a step at 96^2 x 32, and the chain of
    // data-gradient kernels they wait behind is latency-bound (VALU activity ~0.12) -- but launched as they are on a second stream they gain nothing [measured]:
    // five of their 256-thread blocks hold 150 of a CU's 160 KB of LDS for ~100 us, so the chain's short kernels queue for workgroup slots behind them.  With the
    // launch CAPPED at ~2 blocks per CU (each block walks more tiles: 2 x the time alone) the chain keeps half of every CU and the two overlap: 9.13 -> 8.77 ms.
    // The window is narrow [measured, profiles/r4_train_wgcap.txt]: below ~200 k pixels per call the step is all launch floor (nothing to hide behind), above
    // ~1 M the chain fills the chip by itself (a cap costs what it hides; the plain side stream is still -2 %).  Mode 1 keeps the launches' block counts (bit-identical
    // to mode 0); mode 2 (default) applies the cap, i.e. another -- fixed -- order of the partial sums.
    {   // the side stream(s) that demonstrably overlap with the chain(s) -- and, with two lanes, with each other (hn_internal.h: SidePick)
        const hipStream_t r0[2] = {ls[0], ls[1]};
        if ((rc = side_stream_for(ctx, 1, r0, lanes, !capturing, &ws[0]->wg_stream)) != HN_OK) return rc;
        const hipStream_t r1[3] = {ls[0], ls[1], ws[0]->wg_stream};
        if (lanes == 2 && (rc = side_stream_for(ctx, 2, r1, 3, !capturing, &ws[1]->wg_stream)) != HN_OK) return rc;
    }
    for (int l = 0; l < lanes; ++l) {
        const long px = (long)lane_nb[l] * n * n;
        const int mode = ctx->opt_train_overlap;
        tr[l].flag_sync = ctx->opt_side_sync == 1 && !capturing && lanes == 1 && ctx->sync_flags != nullptr;   // (hn_internal.h: sync_flags; a captured step keeps events)
        // with device-word hand-overs the forks cost the chain nothing and the window below opens at any size [measured, r5: 96^2 x 8 5.10 -> 4.55 ms, x 4 4.60 -> 4.18,
        // 64^2 x 32 5.19 -> 4.49, 64^2 x 8 4.13 -> 3.69, 48^2 x 8 4.47 -> 4.07; 128^2 x 32 and 256^2 x 8 unchanged]
        const long px_min = tr[l].flag_sync ? 0 : 200000;
        tr[l].overlap = mode == 1 || (mode == 2 && px >= px_min);
        // ... and the forward sweep's hidden-state launch on that (then idle) stream beside the decoder: a fork whose join comes an iteration later.  [measured,
        // profiles/r4_side_state_fwd.txt] 7.36 -> 7.19 ms at 96^2 x 32, -2 % at 128^2 x 32 / 256^2 x 8 / 96^2 x 128; +7 % at 96^2 x 8 and 64^2 x 32, where the event packets cost
        // more than the 24 us kernel they take off the chain -- hence the same window
        tr[l].side_state_fwd = mode == 2 && px >= px_min;
#ifndef HN_WG_CAP_DIV
#define HN_WG_CAP_DIV 350   // pixels per weight-gradient block of the capped launches (tools/build_variant.sh ... -DHN_WG_CAP_DIV=...)
#endif
        // (the cap keeps its window whatever the hand-overs: it changes the order in which a weight gradient's partial sums are added, and a captured step -- events --
        // must reproduce the eager one's bits)
        tr[l].wg_cap = mode == 2 && px >= 200000 && px < 1000000 ? (int)(px / HN_WG_CAP_DIV < 512 ? 512 : px / HN_WG_CAP_DIV) : 0;
    }
    const size_t fwf = (size_t)batch * p2, fst = (size_t)batch * pst;
    // the training pass is fp32 whatever arithmetic the context's inference path is set to (the 8x8 launchers read it)
    struct PrecisionGuard { hn_ctx* c; int saved; ~PrecisionGuard() { c->precision = saved; } } pg{ctx, ctx->precision};
    ctx->precision = HN_PREC_FP32;
    {   // weights in the layouts the kernels read: 3x3 both arrangements (one launch), 8x8 matrix-core fragments
        auto& W = ctx->tr;
        Pack3Jobs jobs{};
        auto add = [&](const RawDc& dc) {
            jobs.off[jobs.n] = (int)dc.w1; jobs.o[jobs.n] = (short)dc.cm; jobs.i[jobs.n] = (short)dc.cin; ++jobs.n;
            jobs.off[jobs.n] = (int)dc.w2; jobs.o[jobs.n] = (short)dc.co; jobs.i[jobs.n] = (short)dc.cm; ++jobs.n;
        };
        add(L.inc);
        for (int d = 0; d < depth; ++d) { add(L.sig[d]); add(L.st[d]); }
        for (int d = 0; d <= depth; ++d) add(L.dec[d]);
        hipLaunchKernelGGL(k_pack3, dim3(cdiv(16 * 9 * 8, 256), jobs.n), dim3(256), 0, s, weights, W.w3, pk_off((long)L.total) + 2, jobs);
        PackK8Jobs kj{};
        for (int d = 0; d < depth; ++d) {
            kj.off[4 * d + 0] = (int)L.down[d].w; kj.up[4 * d + 0] = 0;   // down, forward
            kj.off[4 * d + 1] = (int)L.down[d].w; kj.up[4 * d + 1] = 1;   // down, backward-data: [out, in] read as [in, out] by the transposed kernel
            kj.off[4 * d + 2] = (int)L.up[d].w;   kj.up[4 * d + 2] = 1;   // up, forward
            kj.off[4 * d + 3] = (int)L.up[d].w;   kj.up[4 * d + 3] = 0;   // up, backward-data: [in, out] read as [out, in] by the convolution kernel
        }
        hipLaunchKernelGGL(k_pack_frag8, dim3(16, 4 * depth), dim3(256), 0, s, weights, W.k8, kj);
        {
            const F3Layout F = f3_layout(depth);
            PackF3Jobs fj{};
            auto put = [&](size_t raw, size_t dst, int cin, int rows, int mode, int cm) {
                fj.raw[fj.n] = (int)raw; fj.dst[fj.n] = (int)dst; fj.cin[fj.n] = (short)cin; fj.rows[fj.n] = (short)rows; fj.mode[fj.n] = (short)mode; fj.cm[fj.n] = (short)cm; ++fj.n;
            };
            auto addf = [&](const RawDc& dc, const size_t (&o)[2], const size_t (&ob)[2]) {
                put(dc.w1, o[0], dc.cin, dc.cin, 0, kFeat);
                put(dc.w2, o[1], dc.cm, dc.cm, 0, kFeat);
                put(dc.w2, ob[0], kFeat, kFeat, 1, kFeat);
                put(dc.w1, ob[1], dc.cin, cdiv(dc.cin, kFeat) * kFeat, 2, kFeat);
            };
            addf(L.inc, F.inc, F.incb);
            for (int d = 0; d < depth; ++d) addf(L.sig[d], F.sig[d], F.sigb[d]);
            for (int d = 0; d <= depth; ++d) addf(L.dec[d], F.dec[d], F.decb[d]);
            for (int d = 0; d < depth; ++d) {   // conv_state: [2][2][3][3] and [2][10][3][3]
                put(L.st[d].w2, F.stb[d][0], kState, kState, 1, kState);
                put(L.st[d].w1, F.stb[d][1], kFeat + kState, 2 * kState, 2, kState);
            }
            hipLaunchKernelGGL(k_pack_frag3, dim3(cdiv(16 * 192, 256), fj.n), dim3(256), 0, s, weights, W.f3, fj);
        }
        if ((rc = zero_async(ctx, W.sumsq, sizeof(float) * (size_t)n_unroll * batch, s)) != HN_OK) return rc;
    }
    if (lanes == 2) {   // the second lane starts behind the packed weights (and whatever the caller's stream held before)
        HN_HIP(ctx, hipEventRecord(ctx->train_fork, s));
        HN_HIP(ctx, hipStreamWaitEvent(ctx->train_stream, ctx->train_fork, 0));
    }
    for (int l = 0; l < lanes; ++l) {
        auto& W = *ws[l];
        if ((rc = zero_async(ctx, W.part, sizeof(float) * W.part_floats, ls[l])) != HN_OK) return rc;
        if ((rc = zero_async(ctx, W.slope_part, sizeof(double) * W.slope_stride * (3 * depth + 2), ls[l])) != HN_OK) return rc;
    }
    // per-lane views of the caller's tensors: sample b0 of a [batch][...] tensor
    auto in_wf = [&](int t, int l) { return (t == 0 ? wf : wf_hist + (size_t)(t - 1) * fwf) + (size_t)lane_b0[l] * p2; };
    auto in_res = [&](int t, int l) { return (t == 0 ? res : res_hist + (size_t)(t - 1) * fwf) + (size_t)lane_b0[l] * p2; };
    auto in_st = [&](int t, int l) { return (t == 0 ? states : st_hist + (size_t)(t - 1) * fst) + (size_t)lane_b0[l] * pst; };
    // An error inside the sweeps must not leave the library's streams forked from the caller's: they are joined before returning.
    auto join_streams = [&]() {
        for (int l = 0; l < lanes; ++l)
            for (int k = 0; k < 2; ++k)
                if (ws[l]->wg_pending[k]) { (void)hipStreamWaitEvent(ls[l], ws[l]->wg_done[k], 0); ws[l]->wg_pending[k] = false; }
        for (int l = 0; l < lanes; ++l)
            if (ws[l]->st_pending) { (void)hipStreamWaitEvent(ls[l], ws[l]->st_done, 0); ws[l]->st_pending = false; }
        if (lanes == 2) {
            (void)hipEventRecord(ctx->train_join, ctx->train_stream);
            (void)hipStreamWaitEvent(s, ctx->train_join, 0);
        }
    };
    if (tr[0].flag_sync && (tr[0].overlap || tr[0].side_state_fwd)) {
        // flag sync: the side stream's gate kernels spin from the moment that stream is free, so it first waits for the caller's stream to get HERE (one event per
        // CALL, as hn_step does): a caller's stream that is seconds behind cannot make a bounded wait give up
        HN_HIP(ctx, hipEventRecord(ws[0]->st_fork, s));
        HN_HIP(ctx, hipStreamWaitEvent(ws[0]->wg_stream, ws[0]->st_fork, 0));
    }
    for (int t = 0; t < n_unroll && rc == HN_OK; ++t)
        for (int l = 0; l < lanes && rc == HN_OK; ++l) {   // the lanes' launches are enqueued alternately, so both streams always hold work
            const size_t o2 = (size_t)t * fwf + (size_t)lane_b0[l] * p2, ost = (size_t)t * fst + (size_t)lane_b0[l] * pst;
            rc = tr[l].forward_step(t, in_wf(t, l), in_res(t, l), in_st(t, l), wf_hist + o2, res_hist + o2, st_hist + ost,
                                    k_sq + (size_t)lane_b0[l] * n * n, src + (src_batch == 1 ? 0 : (size_t)lane_b0[l] * p2), src_batch == 1 ? 1 : lane_nb[l]);
        }
    if (rc != HN_OK) { join_streams(); return rc; }
    for (int l = 0; l < lanes; ++l)   // the last iteration's new states: st_hist is complete behind this
        if (ws[l]->st_pending) { HN_HIP(ctx, hipStreamWaitEvent(ls[l], ws[l]->st_done, 0)); ws[l]->st_pending = false; }
    if (ctx->train_fwd_event != nullptr && !capturing) {   // the histories are complete: the caller's host logic may read them while the backward pass runs (not recorded into a captured graph)
        if (lanes == 2) {                    // (the second lane's forward sweep joins first)
            HN_HIP(ctx, hipEventRecord(ctx->train_join, ctx->train_stream));
            HN_HIP(ctx, hipStreamWaitEvent(s, ctx->train_join, 0));
        }
        if (ctx->train_fwd_sumsq != nullptr)   // sum over (2, N, N) of res^2 per (iteration, sample): what the reference's refill rule thresholds (hybridnet.py:437-438)
            HN_HIP(ctx, hipMemcpyAsync(ctx->train_fwd_sumsq, ctx->tr.sumsq, sizeof(float) * (size_t)n_unroll * batch, hipMemcpyDeviceToHost, s));
        HN_HIP(ctx, hipEventRecord(ctx->train_fwd_event, s));
        ++ctx->train_fwd_events;
    }
    // backward sweep
    int cur_wf[2] = {0, 0}, cur_st[2] = {0, 0};
    for (int l = 0; l < lanes; ++l) {
        auto& W = *ws[l];
        if ((rc = zero_async(ctx, W.g_wf[0], sizeof(float) * (size_t)lane_nb[l] * p2, ls[l])) != HN_OK) return rc;
        if ((rc = zero_async(ctx, W.g_res, sizeof(float) * (size_t)lane_nb[l] * p2, ls[l])) != HN_OK) return rc;
        if ((rc = zero_async(ctx, W.g_st[0], sizeof(float) * (size_t)lane_nb[l] * pst, ls[l])) != HN_OK) return rc;
    }
    const double count = (double)n_unroll * batch * 2.0 * n * n;
    const float loss_c = (float)(2.0 * (double)loss_scale / count);
    for (int t = n_unroll - 1; t >= 0 && rc == HN_OK; --t)
        for (int l = 0; l < lanes && rc == HN_OK; ++l)
            rc = tr[l].backward_step(t, in_wf(t, l), in_res(t, l), in_st(t, l), res_hist + (size_t)t * fwf + (size_t)lane_b0[l] * p2,
                                     k_sq + (size_t)lane_b0[l] * n * n, loss_c, cur_wf[l], cur_st[l]);
    if (rc != HN_OK) { join_streams(); return rc; }
    for (int l = 0; l < lanes; ++l) {
        auto& W = *ws[l];
        for (int k = 0; k < 2; ++k)   // the side stream's weight-gradient launches join the lane's stream
            if (W.wg_pending[k]) { HN_HIP(ctx, hipStreamWaitEvent(ls[l], W.wg_done[k], 0)); W.wg_pending[k] = false; }
        if (!capturing) {
            HN_HIP(ctx, hipEventRecord(W.jobs_copied[W.jobs_set], ls[l]));
            W.jobs_in_flight[W.jobs_set] = true;
        }
        // gradients with respect to the inputs: each lane's samples, on its own stream
        if (grad_wf0) HN_HIP(ctx, hipMemcpyAsync(grad_wf0 + (size_t)lane_b0[l] * p2, W.g_wf[cur_wf[l]], sizeof(float) * (size_t)lane_nb[l] * p2, hipMemcpyDeviceToDevice, ls[l]));
        if (grad_res0) HN_HIP(ctx, hipMemcpyAsync(grad_res0 + (size_t)lane_b0[l] * p2, W.g_res, sizeof(float) * (size_t)lane_nb[l] * p2, hipMemcpyDeviceToDevice, ls[l]));
        if (grad_st0) HN_HIP(ctx, hipMemcpyAsync(grad_st0 + (size_t)lane_b0[l] * pst, W.g_st[cur_st[l]], sizeof(float) * (size_t)lane_nb[l] * pst, hipMemcpyDeviceToDevice, ls[l]));
    }
    if (lanes == 2) {
        HN_HIP(ctx, hipEventRecord(ctx->train_join, ctx->train_stream));
        HN_HIP(ctx, hipStreamWaitEvent(s, ctx->train_join, 0));
    }
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, s, ctx->tr.sumsq, n_unroll * batch, (float)((double)loss_scale / count), loss);
    // the tables' rows -> the gradient blob (first lane, then the second lane added: a fixed order), then the slope entries from their
    // own per-block sums
    for (int l = 0; l < lanes; ++l)
        hipLaunchKernelGGL(k_reduce_rows, dim3(cdiv((int)L.total, 64)), dim3(512), 0, s, ws[l]->part, kPartRows, (int)L.total, grad, l);
    if (ctx->act_kind == HN_ACT_PRELU) {
        SlopeJobs sj[2] = {};
        for (int l = 0; l < lanes; ++l) {
            auto put = [&](int slot, const RawDc& dc, int d) { sj[l].rows[slot] = tr[l].slope_rows(d); sj[l].off[slot] = (int)dc.slope; };
            put(tr[l].slot_inc(), L.inc, 0);
            for (int d = 0; d < depth; ++d) { put(tr[l].slot_sig(d), L.sig[d], d); put(tr[l].slot_st(d), L.st[d], d); }
            for (int d = 0; d <= depth; ++d) put(tr[l].slot_dec(d), L.dec[d], d);
        }
        hipLaunchKernelGGL(k_reduce_slopes, dim3(3 * depth + 2), dim3(256), 0, s, ws[0]->slope_part, (int)ws[0]->slope_stride, sj[0],
                           lanes == 2 ? ws[1]->slope_part : nullptr, (int)ws[1]->slope_stride, sj[1], grad);
    }
    HN_HIP(ctx, hipGetLastError());
    return check_async(ctx, "hn_train_grad");   // (what is up by now; hn_check_async_errors behind a synchronise sees the rest)
}

int hn_train_set_forward_event(hn_ctx* ctx, void* event, float* sumsq_host, int64_t sumsq_capacity) {
    if (!ctx) return HN_ERR_ARG;
    if (sumsq_host != nullptr && (event == nullptr || sumsq_capacity < 1))
        return fail(ctx, HN_ERR_ARG, "hn_train_set_forward_event: the host table needs the event that says when it is complete, and a capacity");
    ctx->train_fwd_event = (hipEvent_t)event;
    ctx->train_fwd_sumsq = sumsq_host;
    ctx->train_fwd_sumsq_cap = sumsq_host != nullptr ? sumsq_capacity : 0;
    return HN_OK;
}

int hn_adam_step(hn_ctx* ctx, float* weights, const float* grad, float* exp_avg, float* exp_avg_sq, const unsigned char* trainable,
                 size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, float clip_value, int64_t step,
                 void* stream) {
    if (!ctx || !weights || !grad || !exp_avg || !exp_avg_sq) return fail(ctx, HN_ERR_ARG, "hn_adam_step: NULL argument");
    if (n == 0) return HN_OK;
    if (step < 1) return fail(ctx, HN_ERR_ARG, "hn_adam_step: step counts from 1 (got %lld)", (long long)step);
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f)) return fail(ctx, HN_ERR_ARG, "hn_adam_step: betas must lie in [0, 1)");
    DeviceGuard guard(ctx);
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weights, grad, exp_avg, exp_avg_sq, trainable, n,
                       (float)((double)lr / bc1), (float)(1.0 / std::sqrt(bc2)), beta1, beta2, eps, weight_decay, clip_value);
    HN_HIP(ctx, hipGetLastError());
    return HN_OK;
}

int64_t hn_train_peek(hn_ctx* ctx, int kind, int level, float* out, int64_t max_floats, void* stream) {
    if (!ctx || !out) return fail(ctx, HN_ERR_ARG, "hn_train_peek: NULL argument");
    if (ctx->tr.tape == nullptr) return fail(ctx, HN_ERR_STATE, "hn_train_peek: hn_train_grad has not run");
    const int depth = ctx->tr.depth;
    if (level < 0 || level > depth) return fail(ctx, HN_ERR_ARG, "hn_train_peek: level %d outside [0, %d]", level, depth);
    const size_t plane = (size_t)(ctx->tr.n >> level) * (ctx->tr.n >> level);
    const bool enc = level < depth;
    DeviceGuard guard(ctx);
    int64_t total = 0;
    // the lanes of the last call hold consecutive runs of samples: first lane, then second
    for (hn_ctx::TrainWs* wp : {&ctx->tr, &ctx->tr_b}) {
        auto& W = *wp;
        if (W.tape == nullptr || W.last_batch == 0) continue;
        const float* p = nullptr;
        size_t ch = kFeat;
        switch (kind) {
            case 0: p = W.tape + W.o_x[level]; break;
            case 1: if (enc) p = W.tape + W.o_zsig[level]; break;
            case 2: if (enc) p = W.tape + W.o_out[level]; break;
            case 3: if (enc) { p = W.tape + W.o_zst[level]; ch = kState; } break;
            case 4: if (enc) p = W.tape + W.o_u[level]; break;
            case 5: p = W.tape + W.o_zdec[level]; break;
            case 6: p = W.tape + W.o_y[level]; break;
            case 7: if (level == 0) p = W.tape + W.o_zinc; break;
            case 16: p = W.g_x[level]; break;
            case 18: if (enc) p = W.g_out[level]; break;
            case 20: if (enc) p = W.g_u[level]; break;
            case 22: p = W.g_y[level]; break;
            default: break;
        }
        if (p == nullptr) return fail(ctx, HN_ERR_ARG, "hn_train_peek: no tensor of kind %d at level %d", kind, level);
        const int64_t count = (int64_t)((size_t)W.last_batch * ch * plane);
        const int64_t room = max_floats - total;
        const int64_t ncopy = count < room ? count : (room > 0 ? room : 0);
        if (ncopy > 0 && hipMemcpyAsync(out + total, p, sizeof(float) * (size_t)ncopy, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            return fail(ctx, HN_ERR_HIP, "hn_train_peek: copy failed");
        total += count;
    }
    return total;
}

}  // extern "C"
